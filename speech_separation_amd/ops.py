"""Thin torch-tensor wrappers over the single-operator entry points of libmsf2
(include/msf2.h).  Host plumbing only: shape checks, output / workspace
allocation from torch's allocator, the current HIP stream.  Every function
requires CUDA (HIP) fp32 contiguous tensors and raises if the library is
missing."""
import torch

from . import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def _chk(*ts):
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("expected contiguous fp32 tensors on the GPU")


def encoder(mix, enc_w_tapmajor, eps=1e-8):
    """mix [B,T], enc_w [16,C] -> (enc [B,S,C], stats [B,2] = (mean, rstd))."""
    lib = _lib.load()
    _chk(mix, enc_w_tapmajor)
    B, T = mix.shape
    C = enc_w_tapmajor.shape[1]
    S = lib.msf2_num_frames(T)
    enc = torch.empty((B, S, C), device=mix.device)
    stats = torch.empty((B, 2), device=mix.device)
    n = lib.msf2_stats_ws_bytes(B, S)
    ws = _ws(n, mix.device)
    _lib.check(lib.msf2_encoder(mix.data_ptr(), enc_w_tapmajor.data_ptr(), enc.data_ptr(), stats.data_ptr(),
                                eps, B, T, C, ws.data_ptr(), ws.numel(), _stream(mix)), "msf2_encoder")
    return enc, stats


def dwconv17(x, dw_tapmajor, out=None, accumulate=False):
    """x [B,S,C] -> x + dwconv17(x); with accumulate, added onto `out`."""
    lib = _lib.load()
    _chk(x, dw_tapmajor)
    B, S, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    _chk(out)
    _lib.check(lib.msf2_dwconv17(x.data_ptr(), C, dw_tapmajor.data_ptr(), out.data_ptr(), C, B, S, C,
                                 int(accumulate), _stream(x)), "msf2_dwconv17")
    return out


def linear(x, W, b=None, shift=False, scale_norm=False, act=0, precision=_lib.PREC_FP32):
    """x [B,S,K], W [N,K] -> act(rowscale(shift(x)) W^T + b) [B,S,N]."""
    lib = _lib.load()
    _chk(x, W)
    B, S, K = x.shape
    N = W.shape[0]
    if precision == _lib.PREC_BF16:          # the bf16 kernels take the weight matrix as bf16
        W = W.to(torch.bfloat16).contiguous()
    y = torch.empty((B, S, N), device=x.device)
    n = lib.msf2_linear_ws_bytes(B, S)
    ws = _ws(n, x.device)
    _lib.check(lib.msf2_linear(x.data_ptr(), K, W.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                               N, B, S, K, N, int(shift), int(scale_norm), act, precision, ws.data_ptr(), ws.numel(),
                               _stream(x)), "msf2_linear")
    return y


def flash_attention(hu, gamma, beta, freqs, precision=_lib.PREC_FP32):
    """hu [B,S,2E+128] = v|u|qk -> o [B,S,E]."""
    lib = _lib.load()
    _chk(hu, gamma, beta, freqs)
    B, S, W = hu.shape
    E = (W - _lib.QK_DIM) // 2
    o = torch.empty((B, S, E), device=hu.device)
    n = lib.msf2_flash_attention_ws_bytes(B, S, E)
    ws = _ws(n, hu.device)
    _lib.check(lib.msf2_flash_attention(hu.data_ptr(), W, gamma.data_ptr(), beta.data_ptr(), freqs.data_ptr(),
                                        o.data_ptr(), B, S, E, precision, ws.data_ptr(), ws.numel(), _stream(hu)),
               "msf2_flash_attention")
    return o


def gated_tcn(x, layer_weights, precision=_lib.PREC_FP32):
    """In place on x [B,S,C]; layer_weights: a _lib.LayerWeights (from PackedWeights.layers[i])."""
    lib = _lib.load()
    _chk(x)
    B, S, C = x.shape
    n = lib.msf2_gated_tcn_ws_bytes(B, S, C)
    ws = _ws(n, x.device)
    _lib.check(lib.msf2_gated_tcn(x.data_ptr(), layer_weights, B, S, C, precision, ws.data_ptr(), ws.numel(),
                                  _stream(x)), "msf2_gated_tcn")
    return x


def decoder(sep, dec_w, T):
    """sep [B,S,C] (encoder output times mask), dec_w [C,16] -> est [B,T]."""
    lib = _lib.load()
    _chk(sep, dec_w)
    B, S, C = sep.shape
    est = torch.empty((B, T), device=sep.device)
    n = lib.msf2_decoder_ws_bytes(B, S)
    ws = _ws(n, sep.device)
    _lib.check(lib.msf2_decoder(sep.data_ptr(), dec_w.data_ptr(), est.data_ptr(), B, S, C, T, ws.data_ptr(),
                                ws.numel(), _stream(sep)), "msf2_decoder")
    return est


def tcn_pointwise(h, t, bstat, tcn_weights, precision=_lib.PREC_FP32):
    """h, t [B,S,C]; bstat [B,2]; tcn_weights: a _lib.TcnWeights -> new h (one kernel launch)."""
    lib = _lib.load()
    _chk(h, t, bstat)
    B, S, C = h.shape
    out = torch.empty_like(h)
    _lib.check(lib.msf2_tcn_pointwise(h.data_ptr(), t.data_ptr(), bstat.data_ptr(), tcn_weights, out.data_ptr(),
                                      B, S, C, precision, _stream(h)), "msf2_tcn_pointwise")
    return out


def pit_si_snr(est, src):
    """est, src: [2, B, T] fp32 on the GPU -> (si_snr [B, 2 (source), 2 (estimate)] dB, best [B] dB,
    perm [B] int32: 0 identity, 1 swapped).  Reference: losses/loss.py:653-723 with the 2-source PitWrapper."""
    lib = _lib.load()
    _chk(est, src)
    assert est.shape == src.shape and est.shape[0] == 2 and est.dim() == 3
    _, B, T = est.shape
    si = torch.empty((B, 2, 2), dtype=torch.float32, device=est.device)
    best = torch.empty((B,), dtype=torch.float32, device=est.device)
    perm = torch.empty((B,), dtype=torch.int32, device=est.device)
    ws = _ws(lib.msf2_pit_si_snr_ws_bytes(B, T), est.device)
    _lib.check(lib.msf2_pit_si_snr(est.data_ptr(), src.data_ptr(), si.data_ptr(), best.data_ptr(), perm.data_ptr(), B, T,
                                   ws.data_ptr(), ws.numel(), _stream(est)), "msf2_pit_si_snr")
    return si, best, perm


def pit_si_snr_noalloc(est, src, si, best, perm):
    """pit_si_snr for callers that cannot allocate: every buffer, outputs included, comes from the caller and the
    library's partial sums live in its fixed device array (msf2_pit_si_snr_noalloc).  Nothing is allocated here.
    The library uses the array of the thread's current device, so the call is made with the operands' device current."""
    lib = _lib.load()
    _chk(est, src, si, best)
    if est.shape != src.shape or est.dim() != 3 or est.shape[0] != 2:
        raise ValueError("est and src must both be [2, B, T]")
    _, B, T = est.shape
    if not (perm.is_cuda and perm.dtype == torch.int32 and perm.is_contiguous() and tuple(perm.shape) == (B,)):
        raise ValueError("perm must be a contiguous int32 tensor [B] on the GPU")
    if tuple(si.shape) != (B, 2, 2) or tuple(best.shape) != (B,):
        raise ValueError("si must be [B, 2, 2] and best [B]")
    if any(t.device != est.device for t in (src, si, best, perm)):
        raise ValueError("all tensors must live on one GPU")
    with torch.cuda.device(est.device):
        _lib.check(lib.msf2_pit_si_snr_noalloc(est.data_ptr(), src.data_ptr(), si.data_ptr(), best.data_ptr(), perm.data_ptr(),
                                               B, T, _stream(est)), "msf2_pit_si_snr_noalloc")


def gated_fsmn(x, packed):
    """x: [B, S, dim] fp32 on the GPU; packed = packing.pack_fsmn_block(...).  Reference:
    Gated_FSMN_Block_Dilated.forward (mossformer2_block.py:456-463)."""
    import ctypes as C
    lib = _lib.load()
    w, dim, inner, _keep = packed
    _chk(x)
    B, S, D = x.shape
    assert D == dim
    y = torch.empty_like(x)
    ws = _ws(lib.msf2_gated_fsmn_ws_bytes(B, S, dim, inner), x.device)
    _lib.check(lib.msf2_gated_fsmn(x.data_ptr(), C.byref(w), y.data_ptr(), B, S, dim, inner, ws.data_ptr(), ws.numel(),
                                   _stream(x)), "msf2_gated_fsmn")
    return y
