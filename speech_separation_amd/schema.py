"""``state_dict()`` key schema of the separation network.

Key names and shapes are those a checkpoint written by the reference carries
(reference modules: models/mossformer2/mossformer2.py:174-183 (enc), :561-586
(mask net), :705-711 (dec), :321-330; models/mossformer2/mossformer2_block.py:
46, 57-59, 71-72, 88-94, 197-217, 513-528, 538-554, 621-643;
models/mossformer2/conv_module.py:293-296).  The list holding the gated TCN
blocks is called ``fsmn`` in the reference (mossformer2_block.py:635) and the
name is kept, since checkpoints are matched by key.
"""
from collections import OrderedDict

from .config import MossFormer2Config

MDL = "mask_net.mdl.intra_mdl.mossformerM"


def _ffconvm(prefix, d_in, d_out, k, out):
    out[f"{prefix}.mdl.0.g"] = (1,)
    out[f"{prefix}.mdl.1.weight"] = (d_out, d_in)
    out[f"{prefix}.mdl.1.bias"] = (d_out,)
    out[f"{prefix}.mdl.3.sequential.1.conv.weight"] = (d_out, 1, k)


def _fsmn_block(p, d, inner, k, out):
    """Gated_FSMN_Block_Dilated (reference: mossformer2_block.py:447-455, Gated_FSMN_dilated :370-382, FFConvM with
    nn.LayerNorm :88-94, UniDeepFsmn_dilated fsmn.py:93-104, DilatedDenseNet fsmn.py:64-82 with lorder 20 -> 39 taps)."""
    out[f"{p}.conv1.0.weight"] = (inner, d, 1)
    out[f"{p}.conv1.0.bias"] = (inner,)
    out[f"{p}.conv1.1.weight"] = (1,)
    out[f"{p}.norm1.weight"] = (inner,)
    out[f"{p}.norm1.bias"] = (inner,)
    for br in ("to_u", "to_v"):
        q = f"{p}.gated_fsmn.{br}.mdl"
        out[f"{q}.0.weight"] = (inner,)
        out[f"{q}.0.bias"] = (inner,)
        out[f"{q}.1.weight"] = (inner, inner)
        out[f"{q}.1.bias"] = (inner,)
        out[f"{q}.3.sequential.1.conv.weight"] = (inner, 1, k)
    f = f"{p}.gated_fsmn.fsmn"
    out[f"{f}.linear.weight"] = (inner, inner)
    out[f"{f}.linear.bias"] = (inner,)
    out[f"{f}.project.weight"] = (inner, inner)
    out[f"{f}.conv.conv1.weight"] = (inner, 1, 39, 1)
    out[f"{f}.conv.norm1.weight"] = (inner,)
    out[f"{f}.conv.norm1.bias"] = (inner,)
    out[f"{f}.conv.prelu1.weight"] = (inner,)
    out[f"{f}.conv.conv2.weight"] = (inner, 2, 39, 1)
    out[f"{f}.conv.norm2.weight"] = (inner,)
    out[f"{f}.conv.norm2.bias"] = (inner,)
    out[f"{f}.conv.prelu2.weight"] = (inner,)
    out[f"{p}.norm2.weight"] = (inner,)
    out[f"{p}.norm2.bias"] = (inner,)
    out[f"{p}.conv2.weight"] = (d, inner, 1)
    out[f"{p}.conv2.bias"] = (d,)


def state_dict_schema(cfg: MossFormer2Config) -> "OrderedDict[str, tuple]":
    """Ordered mapping key -> shape, in the order ``nn.Module.state_dict`` of
    the reference model yields them."""
    d, L = cfg.dim, cfg.num_layers
    o: "OrderedDict[str, tuple]" = OrderedDict()
    o["enc.conv1d.weight"] = (d, 1, cfg.enc_kernel)
    o["mask_net.norm.weight"] = (d,)
    o["mask_net.norm.bias"] = (d,)
    o["mask_net.conv1d_encoder.weight"] = (d, d, 1)
    o["mask_net.pos_enc.scale"] = (1,)
    o["mask_net.pos_enc.inv_freq"] = (d // 2,)
    for i in range(L):
        p = f"{MDL}.layers.{i}"
        o[f"{p}.rotary_pos_emb.freqs"] = (cfg.rot_dim // 2,)
        _ffconvm(f"{p}.to_hidden", d, cfg.hidden, cfg.dw_kernel, o)
        _ffconvm(f"{p}.to_qk", d, cfg.qk_dim, cfg.dw_kernel, o)
        o[f"{p}.qk_offset_scale.gamma"] = (4, cfg.qk_dim)
        o[f"{p}.qk_offset_scale.beta"] = (4, cfg.qk_dim)
        _ffconvm(f"{p}.to_out", cfg.vdim, d, cfg.dw_kernel, o)
    for i in range(L if cfg.recurrent == "fsmn" else 0):
        _fsmn_block(f"{MDL}.fsmn.{i}", d, cfg.fsmn_inner, cfg.dw_kernel, o)
    for i in range(0 if cfg.recurrent == "fsmn" else L):
        p = f"{MDL}.fsmn.{i}"
        o[f"{p}.norm.weight"] = (d,)
        o[f"{p}.norm.bias"] = (d,)
        o[f"{p}.proj_in.weight"] = (d, d)
        o[f"{p}.proj_in.bias"] = (d,)
        o[f"{p}.proj_out.weight"] = (d, d)
        o[f"{p}.proj_out.bias"] = (d,)
        for j in range(cfg.tcn_depth):
            t = f"{p}.tcn.{j}"
            o[f"{t}.net.0.weight"] = (d, 1, 3)
            o[f"{t}.net.0.bias"] = (d,)
            o[f"{t}.net.2.weight"] = (d,)
            o[f"{t}.net.2.bias"] = (d,)
            o[f"{t}.net.3.weight"] = (d, d, 1)
            o[f"{t}.net.3.bias"] = (d,)
            o[f"{t}.gate.0.weight"] = (d, d, 1)
            o[f"{t}.gate.0.bias"] = (d,)
        o[f"{p}.global_gate.0.weight"] = (d, d)
        o[f"{p}.global_gate.0.bias"] = (d,)
    o["mask_net.mdl.intra_mdl.norm.weight"] = (d,)
    o["mask_net.mdl.intra_mdl.norm.bias"] = (d,)
    o["mask_net.mdl.intra_norm.weight"] = (d,)
    o["mask_net.mdl.intra_norm.bias"] = (d,)
    o["mask_net.conv1d_out.weight"] = (d * cfg.num_spks, d, 1)
    o["mask_net.conv1d_out.bias"] = (d * cfg.num_spks,)
    o["mask_net.conv1_decoder.weight"] = (d, d, 1)
    o["mask_net.prelu.weight"] = (1,)
    o["mask_net.output.0.weight"] = (d, d, 1)
    o["mask_net.output.0.bias"] = (d,)
    o["mask_net.output_gate.0.weight"] = (d, d, 1)
    o["mask_net.output_gate.0.bias"] = (d,)
    o["dec.weight"] = (d, 1, cfg.enc_kernel)
    return o
