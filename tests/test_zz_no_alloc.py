"""msf2_pit_si_snr_noalloc -- the entry point for callers that cannot allocate -- against
msf2_pit_si_snr (caller-owned workspace), through the C ABI: identical results (the same two kernels on the
same inputs; the partial moments merely live in the library's fixed device array), on two calls in a row, the
second with a smaller input, so that what the first call left in the array must not leak into the second."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speech_separation_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def _case(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(2, B, T, generator=g) + 0.2
    est = src.flip(0) * 0.7 + 0.2 * torch.randn(2, B, T, generator=g)
    est[:, 0] = src[:, 0] + 0.1 * torch.randn(2, T, generator=g)
    return est.to(DEV), src.to(DEV)


def _run_noalloc(est, src):
    B = est.shape[1]
    si = torch.full((B, 2, 2), float("nan"), device=DEV)
    best = torch.full((B,), float("nan"), device=DEV)
    perm = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(DEV)["allocation.all.allocated"]
    ops.pit_si_snr_noalloc(est, src, si, best, perm)
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(DEV)["allocation.all.allocated"] == before      # torch's allocator was not asked either
    return si, best, perm


def test_noalloc_entry_point_matches_workspace_entry_point():
    lib = _lib.load()
    big = _case(7, 100003, 1)            # 7 x 13 blocks of partial moments
    small = _case(3, 9001, 2)            # 3 x 2 blocks: a prefix of what the first call wrote
    for est, src in (big, small):
        want = ops.pit_si_snr(est, src)
        got = _run_noalloc(est, src)
        for g, w in zip(got, want):
            assert torch.equal(g, w)                           # bit-exact: same kernels, same order
    assert got[2].cpu().tolist()[0] == 0 and set(got[2].cpu().tolist()[1:]) == {1}
    # beyond the fixed array: rejected before anything is launched, outputs untouched
    B, T = 2, 8192 * 4096 + 1       # 2 x 4097 blocks of 8192 samples > 8192
    est = torch.zeros(2, B, T, device=DEV)
    si = torch.full((B, 2, 2), 7.0, device=DEV)
    rc = lib.msf2_pit_si_snr_noalloc(est.data_ptr(), est.data_ptr(), si.data_ptr(), None, None, B, T, None)
    torch.cuda.synchronize()
    assert rc == -2 and bool((si == 7.0).all())
    assert lib.msf2_pit_si_snr_noalloc(None, None, None, None, None, 1, 100, None) == -1


def test_noalloc_entry_point_refuses_operands_off_the_current_device():
    """The fixed array used is the one of the thread's current device: waveforms that do not live in that device's
    memory (here: plain host memory) are refused before anything is launched, and the wrapper checks shapes and types
    of the caller's output buffers (a short or mistyped `perm` would be written out of bounds)."""
    lib = _lib.load()
    host = np.zeros((2, 1, 4096), dtype=np.float32)
    best = torch.full((1,), 5.0, device=DEV)
    rc = lib.msf2_pit_si_snr_noalloc(host.ctypes.data, host.ctypes.data, None, best.data_ptr(), None, 1, 4096, None)
    torch.cuda.synchronize()
    assert rc == -1 and float(best[0]) == 5.0
    est, src = _case(2, 5000, 3)
    si = torch.empty((2, 2, 2), device=DEV)
    with pytest.raises(ValueError):
        ops.pit_si_snr_noalloc(est, src, si, torch.empty(2, device=DEV), torch.empty(2, dtype=torch.int64, device=DEV))
    with pytest.raises(ValueError):
        ops.pit_si_snr_noalloc(est, src, si, torch.empty(2, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV))
    with pytest.raises(ValueError):
        ops.pit_si_snr_noalloc(est, src[:, :1].contiguous(), si, torch.empty(2, device=DEV),
                               torch.empty(2, dtype=torch.int32, device=DEV))
