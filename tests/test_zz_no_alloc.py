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
