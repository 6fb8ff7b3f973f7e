"""Turn a reference ``state_dict`` into the device-resident packed weights
libmsf2 consumes (layout documented in include/msf2.h, struct msf2_weights).

Folding done here, once per load:
  * ScaleNorm gain g -> rows of the following Linear (FFConvM,
    reference: mossformer2_block.py:79-100);
  * LayerNorm / GroupNorm affine -> columns of the following projection plus
    two vectors (c1, c2) that let the kernel apply mean / rstd in its epilogue;
  * to_hidden and to_qk concatenated into one projection (same input);
  * depthwise filters transposed to tap-major [taps][C].
"""
import ctypes as C

import torch

from . import _lib
from .config import MossFormer2Config
from .schema import MDL


class PackedWeights:
    """Owns the packed device tensors and the ctypes structs pointing at them."""

    def __init__(self, state, cfg: MossFormer2Config, device, precision=_lib.PREC_FP32):
        self.cfg = cfg
        self.device = torch.device(device)
        self._keep = []
        d = cfg.dim
        g = lambda k: state[k].detach().to("cpu", torch.float64)

        bf16 = precision == _lib.PREC_BF16

        def dev(t):
            t = t.to(torch.float32).contiguous().to(self.device)
            self._keep.append(t)
            return t.data_ptr()

        def devw(t):
            """A projection matrix [N, K]: the B operand of a contraction kernel
            (bf16 in MSF2_PREC_BF16 mode, fp32 otherwise)."""
            if not bf16:
                return dev(t)
            t = t.to(torch.float32).to(torch.bfloat16).contiguous().to(self.device)
            self._keep.append(t)
            return t.data_ptr()

        def fold_norm(W, b, gain, offset):
            """norm(x) W^T + b -> (W diag(gain), c1 = W offset + b, c2 = W gain).
            c2 is summed over the weights as the kernel will see them (bf16-rounded in
            bf16 mode) so that the mean term cancels exactly."""
            Wg = W * gain[None, :]
            c1 = W @ offset + (b if b is not None else 0.0)
            Wr = Wg.to(torch.float32).to(torch.bfloat16).to(torch.float64) if bf16 else Wg
            c2 = Wr.sum(dim=1)
            return devw(Wg), dev(c1), dev(c2)

        self.layers = (_lib.LayerWeights * max(cfg.num_layers, 1))()
        fsmn = cfg.recurrent == "fsmn"
        self.fsmn_blocks = (_lib.FsmnWeights * max(cfg.num_layers, 1))()
        for i in range(cfg.num_layers):
            lw = self.layers[i]
            p = f"{MDL}.layers.{i}"
            Wh = g(f"{p}.to_hidden.mdl.1.weight") * g(f"{p}.to_hidden.mdl.0.g")
            Wq = g(f"{p}.to_qk.mdl.1.weight") * g(f"{p}.to_qk.mdl.0.g")
            lw.w_hq = devw(torch.cat((Wh, Wq), 0))
            lw.b_hq = dev(torch.cat((g(f"{p}.to_hidden.mdl.1.bias"), g(f"{p}.to_qk.mdl.1.bias")), 0))
            dwh = g(f"{p}.to_hidden.mdl.3.sequential.1.conv.weight")[:, 0, :]
            dwq = g(f"{p}.to_qk.mdl.3.sequential.1.conv.weight")[:, 0, :]
            lw.dw_hq = dev(torch.cat((dwh, dwq), 0).t())
            lw.os_gamma = dev(g(f"{p}.qk_offset_scale.gamma"))
            lw.os_beta = dev(g(f"{p}.qk_offset_scale.beta"))
            lw.rot_freqs = dev(g(f"{p}.rotary_pos_emb.freqs"))
            lw.w_out = devw(g(f"{p}.to_out.mdl.1.weight") * g(f"{p}.to_out.mdl.0.g"))
            lw.b_out = dev(g(f"{p}.to_out.mdl.1.bias"))
            lw.dw_out = dev(g(f"{p}.to_out.mdl.3.sequential.1.conv.weight")[:, 0, :].t())
            q = f"{MDL}.fsmn.{i}"
            if fsmn:
                blk, _, inner, keep = pack_fsmn_block(state, self.device, q, precision)
                assert inner == cfg.fsmn_inner, (inner, cfg.fsmn_inner)
                self._keep.extend(keep)
                self.fsmn_blocks[i] = blk
                lw.fsmn = C.pointer(self.fsmn_blocks[i])
                continue
            lw.w_gg = devw(g(f"{q}.global_gate.0.weight"))
            lw.b_gg = dev(g(f"{q}.global_gate.0.bias"))
            lw.w_in, lw.c1_in, lw.c2_in = fold_norm(g(f"{q}.proj_in.weight"), g(f"{q}.proj_in.bias"),
                                                    g(f"{q}.norm.weight"), g(f"{q}.norm.bias"))
            lw.w_po = devw(g(f"{q}.proj_out.weight"))
            lw.b_po = dev(g(f"{q}.proj_out.bias"))
            for j in range(cfg.tcn_depth):
                t = f"{q}.tcn.{j}"
                tw = lw.tcn[j]
                tw.dw_w = dev(g(f"{t}.net.0.weight")[:, 0, :].t())
                tw.dw_b = dev(g(f"{t}.net.0.bias"))
                tw.w_gate = devw(g(f"{t}.gate.0.weight")[:, :, 0])
                tw.b_gate = dev(g(f"{t}.gate.0.bias"))
                tw.w_pw, tw.c1, tw.c2 = fold_norm(g(f"{t}.net.3.weight")[:, :, 0], g(f"{t}.net.3.bias"),
                                                  g(f"{t}.net.2.weight"), g(f"{t}.net.2.bias"))

        w = _lib.Weights()
        w.enc_w = dev(g("enc.conv1d.weight")[:, 0, :].t())
        w.w_ce, w.c1_ce, w.c2_ce = fold_norm(g("mask_net.conv1d_encoder.weight")[:, :, 0], None,
                                             g("mask_net.norm.weight"), g("mask_net.norm.bias"))
        w.pe_inv_freq = dev(g("mask_net.pos_enc.inv_freq"))
        w.pe_scale = float(state["mask_net.pos_enc.scale"].detach().float().cpu().item())
        w.prelu_a = float(state["mask_net.prelu.weight"].detach().float().cpu().item())
        w.fin_ln_w = dev(g("mask_net.mdl.intra_mdl.norm.weight"))
        w.fin_ln_b = dev(g("mask_net.mdl.intra_mdl.norm.bias"))
        w.fin_gn_w = dev(g("mask_net.mdl.intra_norm.weight"))
        w.fin_gn_b = dev(g("mask_net.mdl.intra_norm.bias"))
        w.w_co = devw(g("mask_net.conv1d_out.weight")[:, :, 0])
        w.b_co = dev(g("mask_net.conv1d_out.bias"))
        w.w_o = devw(g("mask_net.output.0.weight")[:, :, 0])
        w.b_o = dev(g("mask_net.output.0.bias"))
        w.w_og = devw(g("mask_net.output_gate.0.weight")[:, :, 0])
        w.b_og = dev(g("mask_net.output_gate.0.bias"))
        w.w_cd = devw(g("mask_net.conv1_decoder.weight")[:, :, 0])
        w.dec_w = dev(g("dec.weight")[:, 0, :])
        w.layers = C.cast(self.layers, C.POINTER(_lib.LayerWeights))
        self.struct = w
        self.c_cfg = _lib.Config(cfg.dim, cfg.num_layers, cfg.enc_kernel, cfg.num_spks, precision,
                                 _lib.REC_FSMN if fsmn else _lib.REC_TCN, cfg.fsmn_inner)

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep)


def pack_fsmn_block(state, device, prefix="", precision=_lib.PREC_FP32):
    """Device weights of one Gated_FSMN_Block_Dilated (reference: mossformer2_block.py:434-463) for
    msf2_gated_fsmn / the FSMN form of msf2_forward.  state: the block's state_dict (reference key names,
    optionally under prefix).  precision: MSF2_PREC_BF16 stores the projection matrices as bf16 (what the bf16
    contraction kernels read).  Returns (FsmnWeights struct, dim, inner, keep-alive list)."""
    bf16 = precision == _lib.PREC_BF16
    pre = (prefix + ".") if prefix else ""
    keep = []

    def g(k):
        return state[pre + k].detach().to(torch.float32).cpu()

    def dev(t):
        t = t.to(torch.float32).contiguous().to(device)
        keep.append(t)
        return t.data_ptr()

    def devw(t):
        if not bf16:
            return dev(t)
        t = t.to(torch.float32).to(torch.bfloat16).contiguous().to(device)
        keep.append(t)
        return t.data_ptr()

    w = _lib.FsmnWeights()
    w1 = g("conv1.0.weight")[:, :, 0]
    inner, dim = w1.shape
    w.w1, w.b1, w.prelu1 = devw(w1), dev(g("conv1.0.bias")), float(g("conv1.1.weight")[0])
    w.ln1_g, w.ln1_b = dev(g("norm1.weight")), dev(g("norm1.bias"))
    Ws, c1s, c2s, dws = [], [], [], []
    for br in ("to_u", "to_v"):
        q = f"gated_fsmn.{br}.mdl"
        W, b = g(f"{q}.1.weight").double(), g(f"{q}.1.bias").double()
        gain, beta = g(f"{q}.0.weight").double(), g(f"{q}.0.bias").double()
        Wf = (W * gain[None, :]).float()                  # LayerNorm(x) W^T + b = rstd (x W'^T - mean c2) + c1
        Ws.append(Wf)
        c1s.append((W @ beta + b).float())
        c2s.append((Wf.to(torch.bfloat16) if bf16 else Wf).double().sum(1).float())   # over the weights as the kernel sees them
        dws.append(g(f"{q}.3.sequential.1.conv.weight")[:, 0, :])
    w.w_uv, w.c1_uv, w.c2_uv = devw(torch.cat(Ws, 0)), dev(torch.cat(c1s, 0)), dev(torch.cat(c2s, 0))
    w.dw_uv = dev(torch.cat(dws, 0).t())
    f = "gated_fsmn.fsmn"
    w.w_lin, w.b_lin, w.w_proj = devw(g(f"{f}.linear.weight")), dev(g(f"{f}.linear.bias")), devw(g(f"{f}.project.weight"))
    w.w_c1 = dev(g(f"{f}.conv.conv1.weight")[:, 0, :, 0].t())                  # [39][C]
    w.in1_g, w.in1_b, w.pr1 = dev(g(f"{f}.conv.norm1.weight")), dev(g(f"{f}.conv.norm1.bias")), dev(g(f"{f}.conv.prelu1.weight"))
    w.w_c2 = dev(g(f"{f}.conv.conv2.weight")[:, :, :, 0].permute(2, 1, 0))       # [C][2][39] -> [39][2][C]
    w.in2_g, w.in2_b, w.pr2 = dev(g(f"{f}.conv.norm2.weight")), dev(g(f"{f}.conv.norm2.bias")), dev(g(f"{f}.conv.prelu2.weight"))
    w.ln2_g, w.ln2_b = dev(g("norm2.weight")), dev(g("norm2.bias"))
    w.w2, w.b2 = devw(g("conv2.weight")[:, :, 0]), dev(g("conv2.bias"))
    return w, int(dim), int(inner), keep
