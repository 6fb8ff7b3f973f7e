"""CPU oracle: a flat functional restatement of the MossFormer2_SS forward.

TEST INFRASTRUCTURE ONLY.  Nothing under ``speech_separation_amd/`` imports
this file; it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / timed CPU port, never as
the shipped compute path.

Parity status: PINNED for everything except the rotary embedding against
outputs of the reference model itself (``oracle/gen_golden.py`` imports the
reference in the build container and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` replays them).  The rotary arithmetic lives in
the third-party package ``rotary-embedding-torch==0.8.3`` (reference:
requirements.txt:25, call sites models/mossformer2/mossformer2_block.py:12,
:270-271, :621) which is not installed and has no test in the reference:
ROTARY PARITY IS UNPINNED; ``rotary()`` restates the published algorithm of
that release (interleaved pairs, theta 10000, first 32 dims).

Layout: activations are token-major ``[B, S, C]`` throughout (the reference
flips between ``[B, C, S]`` and ``[B, S, C]``); weights come as a dict keyed by
the reference's ``state_dict`` names.
"""
import math

import torch
import torch.nn.functional as F

MDL = "mask_net.mdl.intra_mdl.mossformerM"


def group_norm1(x, w, b, eps):
    """GroupNorm(1, C): statistics over (C, S) per utterance, biased variance,
    per-channel affine.  x: [B, S, C].
    (reference: models/mossformer2/mossformer2.py:139-148, mossformer2_block.py:521)"""
    mu = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def layer_norm(x, w, b, eps):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def scale_norm(x, g, eps=1e-5):
    """reference: mossformer2_block.py:41-50"""
    n = torch.sqrt((x * x).sum(dim=-1, keepdim=True)) * (x.shape[-1] ** -0.5)
    return x / n.clamp(min=eps) * g


def dwconv_time(x, w, bias=None, dilation=1):
    """Depthwise conv along time, 'same' zero padding.  x: [B, S, C]; w: [C, 1, k]."""
    k = w.shape[-1]
    pad = (k - 1) * dilation // 2
    y = F.conv1d(x.transpose(1, 2), w, bias, padding=pad, dilation=dilation, groups=w.shape[0])
    return y.transpose(1, 2)


def ffconvm(x, W, p):
    """ScaleNorm -> Linear -> SiLU -> (h + dwconv17(h)); dropout is identity in eval.
    (reference: mossformer2_block.py:79-100, conv_module.py:298-299)"""
    n = scale_norm(x, W[f"{p}.mdl.0.g"])
    h = F.silu(n @ W[f"{p}.mdl.1.weight"].t() + W[f"{p}.mdl.1.bias"])
    return h + dwconv_time(h, W[f"{p}.mdl.3.sequential.1.conv.weight"])


def rotary(t, freqs):
    """rotary-embedding-torch 0.8.3 ``rotate_queries_or_keys`` with defaults:
    positions 0..S-1, angle = pos * freqs[i] repeated over interleaved pairs,
    only the first 2*len(freqs) features are rotated.  t: [B, S, D]."""
    S = t.shape[-2]
    rd = 2 * freqs.shape[0]
    pos = torch.arange(S, dtype=freqs.dtype)
    ang = pos[:, None] * freqs[None, :]                     # [S, rd/2]
    ang = ang.repeat_interleave(2, dim=-1)                  # [S, rd]
    tr, tp = t[..., :rd], t[..., rd:]
    x1, x2 = tr[..., 0::2], tr[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).reshape(tr.shape)  # rotate_half, interleaved
    out = tr * ang.cos().to(t.dtype) + rot * ang.sin().to(t.dtype)
    return torch.cat((out, tp), dim=-1)


def attention_core(v, u, qk, gam, bet, freqs, group=256):
    """OffsetScale + rotary + grouped quadratic attention + global linear
    attention + output gating (reference: mossformer2_block.py:255-258, :262-325).
    v, u: [B, S, E]; qk: [B, S, 128]; returns o = (att_u * v) * sigmoid(att_v * u)."""
    B, S, _ = v.shape
    heads = [qk * gam[h] + bet[h] for h in range(4)]
    quad_q, lin_q, quad_k, lin_k = [rotary(t, freqs) for t in heads]
    pad = (group - S % group) % group
    if pad:
        quad_q, quad_k, lin_q, lin_k, v_p, u_p = [F.pad(t, (0, 0, 0, pad))
                                                  for t in (quad_q, quad_k, lin_q, lin_k, v, u)]
    else:
        v_p, u_p = v, u
    G = (S + pad) // group
    gq, gk, lq, lk, gv, gu = [t.reshape(B, G, group, t.shape[-1])
                              for t in (quad_q, quad_k, lin_q, lin_k, v_p, u_p)]
    sim = torch.einsum("bgid,bgjd->bgij", gq, gk) / group
    attn = F.relu(sim) ** 2
    if pad:   # padded keys are masked in the reference (:279-280, :294-295)
        keep = torch.arange(G * group).reshape(1, G, 1, group) < S
        attn = attn.masked_fill(~keep, 0.0)
    quad_v = torch.einsum("bgij,bgjd->bgid", attn, gv)
    quad_u = torch.einsum("bgij,bgjd->bgid", attn, gu)
    lin_kv = torch.einsum("bgnd,bgne->bde", lk, gv) / S
    lin_ku = torch.einsum("bgnd,bgne->bde", lk, gu) / S
    lin_v = torch.einsum("bgnd,bde->bgne", lq, lin_kv)
    lin_u = torch.einsum("bgnd,bde->bgne", lq, lin_ku)
    att_v = (quad_v + lin_v).reshape(B, G * group, -1)[:, :S]
    att_u = (quad_u + lin_u).reshape(B, G * group, -1)[:, :S]
    return (att_u * v) * torch.sigmoid(att_v * u)


def token_shift(x):
    """reference: mossformer2_block.py:245-248"""
    half = x.shape[-1] // 2
    return torch.cat((F.pad(x[..., :half], (0, 0, 1, -1)), x[..., half:]), dim=-1)


def flash_layer(x, W, p, group=256):
    """FLASH_ShareA_FFConvM.forward (reference: mossformer2_block.py:221-260)."""
    shifted = token_shift(x)
    v, u = ffconvm(shifted, W, f"{p}.to_hidden").chunk(2, dim=-1)
    qk = ffconvm(shifted, W, f"{p}.to_qk")
    out = attention_core(v, u, qk, W[f"{p}.qk_offset_scale.gamma"], W[f"{p}.qk_offset_scale.beta"],
                         W[f"{p}.rotary_pos_emb.freqs"], group)
    return x + ffconvm(out, W, f"{p}.to_out")


def tcn_block(h, W, p, dilation):
    """TCNBlock.forward (reference: mossformer2_block.py:508-531), h: [B, S, C]."""
    t = F.silu(dwconv_time(h, W[f"{p}.net.0.weight"], W[f"{p}.net.0.bias"], dilation))
    t = group_norm1(t, W[f"{p}.net.2.weight"], W[f"{p}.net.2.bias"], 1e-5)
    net = t @ W[f"{p}.net.3.weight"][:, :, 0].t() + W[f"{p}.net.3.bias"]
    gate = torch.sigmoid(h @ W[f"{p}.gate.0.weight"][:, :, 0].t() + W[f"{p}.gate.0.bias"])
    return h + gate * net


def gated_tcn(x, W, p, depth=8):
    """GatedTCNBlock.forward (reference: mossformer2_block.py:556-566)."""
    g = torch.sigmoid(x @ W[f"{p}.global_gate.0.weight"].t() + W[f"{p}.global_gate.0.bias"])
    h = layer_norm(x, W[f"{p}.norm.weight"], W[f"{p}.norm.bias"], 1e-5)
    h = h @ W[f"{p}.proj_in.weight"].t() + W[f"{p}.proj_in.bias"]
    for j in range(depth):
        h = tcn_block(h, W, f"{p}.tcn.{j}", 2 ** j)
    return x + ((g * h) @ W[f"{p}.proj_out.weight"].t() + W[f"{p}.proj_out.bias"])


def pos_encoding(S, inv_freq, scale):
    """ScaledSinuEmbedding (reference: mossformer2_block.py:54-66) -> [S, C]."""
    t = torch.arange(S, dtype=inv_freq.dtype)
    sinu = t[:, None] * inv_freq[None, :]
    return torch.cat((sinu.sin(), sinu.cos()), dim=-1) * scale


def forward(W, mix, cfg, taps=None):
    """Full forward.  W: dict key -> torch tensor (CPU); mix: [B, T].
    Returns a list of ``num_spks`` tensors [B, T]
    (reference: models/mossformer2/mossformer2.py:712-736).
    ``taps`` (optional dict) receives named intermediates, token-major."""
    B, T = mix.shape
    K, st, d = cfg.enc_kernel, cfg.stride, cfg.dim
    # encoder (reference: mossformer2.py:186-210)
    enc = F.relu(F.conv1d(mix[:, None, :], W["enc.conv1d.weight"], stride=st)).transpose(1, 2)  # [B,S,C]
    S = enc.shape[1]
    # mask net front (reference: mossformer2.py:609-618)
    y = group_norm1(enc, W["mask_net.norm.weight"], W["mask_net.norm.bias"], 1e-8)
    y = y @ W["mask_net.conv1d_encoder.weight"][:, :, 0].t()
    y = y + pos_encoding(S, W["mask_net.pos_enc.inv_freq"], W["mask_net.pos_enc.scale"]).to(y.dtype)
    skip = y
    x = y
    if taps is not None:
        taps["enc"] = enc
        taps["x0"] = x
    for l in range(cfg.num_layers):
        x = flash_layer(x, W, f"{MDL}.layers.{l}", cfg.group_size)
        if taps is not None and l == 0:
            taps["flash0"] = x
        if getattr(cfg, "recurrent", "tcn") == "fsmn":
            x = gated_fsmn_block_dilated(x, W, f"{MDL}.fsmn.{l}")
        else:
            x = gated_tcn(x, W, f"{MDL}.fsmn.{l}", cfg.tcn_depth)
        if taps is not None and l == 0:
            taps["tcn0"] = x
    # reference: mossformer2.py:350, :500-506
    x = layer_norm(x, W["mask_net.mdl.intra_mdl.norm.weight"], W["mask_net.mdl.intra_mdl.norm.bias"], 1e-6)
    x = group_norm1(x, W["mask_net.mdl.intra_norm.weight"], W["mask_net.mdl.intra_norm.bias"], 1e-8) + skip
    if taps is not None:
        taps["mdl"] = x
    # mask head (reference: mossformer2.py:623-644)
    a = W["mask_net.prelu.weight"]
    x = torch.where(x >= 0, x, a * x)
    z = x @ W["mask_net.conv1d_out.weight"][:, :, 0].t() + W["mask_net.conv1d_out.bias"]   # [B,S,spk*C]
    outs = []
    for s in range(cfg.num_spks):
        zs = z[..., s * d:(s + 1) * d]
        o = torch.tanh(zs @ W["mask_net.output.0.weight"][:, :, 0].t() + W["mask_net.output.0.bias"])
        g = torch.sigmoid(zs @ W["mask_net.output_gate.0.weight"][:, :, 0].t() + W["mask_net.output_gate.0.bias"])
        m = F.relu((o * g) @ W["mask_net.conv1_decoder.weight"][:, :, 0].t())
        if taps is not None:
            taps[f"mask{s}"] = m
        sep = enc * m
        # decoder (reference: mossformer2.py:238-260, :719-731)
        est = F.conv_transpose1d(sep.transpose(1, 2), W["dec.weight"], stride=st)[:, 0, :]
        if est.shape[1] < T:
            est = F.pad(est, (0, T - est.shape[1]))
        else:
            est = est[:, :T]
        outs.append(est)
    return outs


def to_torch(state, dtype=torch.float32):
    return {k: torch.as_tensor(v).to(dtype) for k, v in state.items()}


# ---------------------------------------------------------------------------
# decode harness (reference: utils/decode.py:25-83) restated on top of any
# callable ``model(x[B,T]) -> list of [B,T]``.
# ---------------------------------------------------------------------------
def decode_pad_len(t, window, stride):
    if t < window:
        return window
    if t < window + stride:
        return window + stride
    if (t - window) % stride != 0:
        return t + (t - (t - window) // stride * stride)
    return t


def decode_one_audio(model_fn, inputs, sampling_rate=8000, decode_window=1,
                     one_time_decode_length=60, num_spks=2):
    import numpy as np
    window = sampling_rate * decode_window
    stride = int(window * 0.75)
    b, t = inputs.shape
    segment = t > window * one_time_decode_length
    tp = decode_pad_len(t, window, stride)
    x = np.concatenate([inputs, np.zeros((b, tp - t))], 1) if tp > t else inputs
    x = torch.from_numpy(np.float32(x))
    t = x.shape[1]
    out = []
    if segment:
        outputs = np.zeros((num_spks, t))
        give = (window - stride) // 2
        idx = 0
        while idx + window <= t:
            o = model_fn(x[:, idx:idx + window])
            for s in range(num_spks):
                a = o[s][0, :].cpu().numpy()
                if idx == 0:
                    outputs[s, idx:idx + window - give] = a[:-give]
                else:
                    outputs[s, idx + give:idx + window - give] = a[give:-give]
            idx += stride
        out = [outputs[s, :] for s in range(num_spks)]
    else:
        o = model_fn(x)
        out = [o[s][0, :].cpu().numpy() for s in range(num_spks)]
    m = max(float(np.max(np.abs(a))) for a in out)
    return [a / m for a in out]


# ---------------------------------------------------------------------------
# SI-SNR with 2-source permutation invariance (reference: losses/loss.py:653-723 cal_si_snr, full-length mask;
# PitWrapper :24-178 for two sources = the better of the two assignments by mean loss)
# ---------------------------------------------------------------------------
def cal_si_snr(source, estimate):
    """source, estimate: [T, B, C] -> SI-SNR in dB [B, C] (the reference returns its negative, as a loss)."""
    EPS = 1e-8
    n = float(source.shape[0])
    zs = source - source.sum(0, keepdim=True) / n
    ze = estimate - estimate.sum(0, keepdim=True) / n
    dot = (ze * zs).sum(0, keepdim=True)
    energy = (zs ** 2).sum(0, keepdim=True) + EPS
    proj = dot * zs / energy
    noise = ze - proj
    ratio = (proj ** 2).sum(0) / ((noise ** 2).sum(0) + EPS)
    return 10 * torch.log10(ratio + EPS)


def pit_si_snr(est, src):
    """est, src: [2, B, T] -> (si_snr [B, 2 (source), 2 (estimate)], best [B], perm [B])."""
    B = est.shape[1]
    si = torch.empty((B, 2, 2), dtype=est.dtype)
    for s_ in range(2):
        for e_ in range(2):
            si[:, s_, e_] = cal_si_snr(src[s_].t().unsqueeze(-1), est[e_].t().unsqueeze(-1))[:, 0]
    keep = 0.5 * (si[:, 0, 0] + si[:, 1, 1])
    swap = 0.5 * (si[:, 0, 1] + si[:, 1, 0])
    return si, torch.maximum(keep, swap), (swap > keep).to(torch.int32)


# ---------------------------------------------------------------------------
# FSMN-family recurrent block (dormant in the reference's live model, kept behind a constructor switch):
# Gated_FSMN_Block_Dilated (reference: mossformer2_block.py:434-463), Gated_FSMN_dilated (:361-393),
# FFConvM with LayerNorm (:79-100), UniDeepFsmn_dilated and DilatedDenseNet (fsmn.py:62-114),
# CLayerNorm (layer_norm.py:14-35).  W: state_dict of the block; x: [B, S, dim].
# ---------------------------------------------------------------------------
def _ffconvm_ln(x, W, p):
    n = layer_norm(x, W[f"{p}.mdl.0.weight"], W[f"{p}.mdl.0.bias"], 1e-5)
    h = F.silu(n @ W[f"{p}.mdl.1.weight"].t() + W[f"{p}.mdl.1.bias"])
    return h + dwconv_time(h, W[f"{p}.mdl.3.sequential.1.conv.weight"])


def _instance_norm_prelu(t, gain, bias, slope, eps=1e-5):
    """InstanceNorm2d(affine) over time per (utterance, channel), then per-channel PReLU.  t: [B, S, C]."""
    mu = t.mean(dim=1, keepdim=True)
    var = ((t - mu) ** 2).mean(dim=1, keepdim=True)
    y = (t - mu) / torch.sqrt(var + eps) * gain + bias
    return torch.where(y >= 0, y, y * slope)


def dilated_dense_net(p1, W, p, lorder=20):
    """depth 2.  Stage 1: depthwise 39-tap conv (dilation 1).  Stage 2: 39-tap conv with dilation 2 over
    cat[out1, p1] in 256 groups of 2 input channels: group g reads channels 2g, 2g+1 of the concatenation."""
    C = p1.shape[-1]
    k = 2 * lorder - 1
    w1 = W[f"{p}.conv1.weight"][:, :, :, 0]                      # [C, 1, k]
    t1 = F.conv1d(p1.transpose(1, 2), w1, None, padding=(k - 1) // 2, groups=C).transpose(1, 2)
    out1 = _instance_norm_prelu(t1, W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"], W[f"{p}.prelu1.weight"])
    skip = torch.cat((out1, p1), dim=-1)                          # [B, S, 2C]
    w2 = W[f"{p}.conv2.weight"][:, :, :, 0]                      # [C, 2, k]
    t2 = F.conv1d(skip.transpose(1, 2), w2, None, padding=(k - 1), dilation=2, groups=C).transpose(1, 2)
    return _instance_norm_prelu(t2, W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"], W[f"{p}.prelu2.weight"])


def gated_fsmn_block_dilated(x, W, p=""):
    pre = (p + ".") if p else ""
    c1 = x @ W[f"{pre}conv1.0.weight"][:, :, 0].t() + W[f"{pre}conv1.0.bias"]
    c1 = torch.where(c1 >= 0, c1, c1 * W[f"{pre}conv1.1.weight"])
    n1 = layer_norm(c1, W[f"{pre}norm1.weight"], W[f"{pre}norm1.bias"], 1e-5)
    g = f"{pre}gated_fsmn"
    xu = _ffconvm_ln(n1, W, f"{g}.to_u")
    xv = _ffconvm_ln(n1, W, f"{g}.to_v")
    f1 = torch.relu(xu @ W[f"{g}.fsmn.linear.weight"].t() + W[f"{g}.fsmn.linear.bias"])
    p1 = f1 @ W[f"{g}.fsmn.project.weight"].t()
    xu = xu + dilated_dense_net(p1, W, f"{g}.fsmn.conv")
    seq = xv * xu + n1
    n2 = layer_norm(seq, W[f"{pre}norm2.weight"], W[f"{pre}norm2.bias"], 1e-5)
    return n2 @ W[f"{pre}conv2.weight"][:, :, 0].t() + W[f"{pre}conv2.bias"] + x
