"""Diagnostic: per-wave cycle shares of one step of the bf16 DMA contraction (TCN pointwise) from a
timing-only build (-DMSF2_DIAG_STAMP, selected with MSF2_LIB).  Never used by the product."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_separation_amd import _lib, ops
from speech_separation_amd.config import MossFormer2Config
from speech_separation_amd.packing import PackedWeights
from speech_separation_amd.synth import make_state_dict
dev = "cuda:0"
cfg = MossFormer2Config(num_layers=1)
sd = {k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()}
pk = PackedWeights(sd, cfg, dev, 1)
lib = _lib.load()
B, S, Cc = 64, 3999, 512
x = torch.randn(B, S, Cc, device=dev)
for _ in range(2):
    ops.gated_tcn(x, pk.layers[0], 1)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (2048 * 16))()
raw.msf2_diag_read.argtypes = [C.c_void_p, C.c_size_t]
raw.msf2_diag_read(buf, 2048 * 16)
a = np.array(buf, dtype=np.float64).reshape(2048, 4, 4)
buf2 = (C.c_ulonglong * (2048 * 8))()
raw.msf2_diag_read2.argtypes = [C.c_void_p, C.c_size_t]
raw.msf2_diag_read2(buf2, 2048 * 8)
b = np.array(buf2, dtype=np.uint64).reshape(2048, 4, 2)
steps = (b[..., 0] & np.uint64((1 << 20) - 1)).astype(np.float64)
epi = (b[..., 0] >> np.uint64(20)).astype(np.float64)
tot = b[..., 1].astype(np.float64)
ok = steps > 0
print("blocks with data", ok[:, 0].sum(), "steps per block", steps[ok].mean())
for i, n in enumerate(("vmcnt wait", "barrier wait", "whole step")):
    v = a[..., i][ok] / steps[ok]
    print(f"{n:14s} cycles/step: mean {v.mean():7.0f}  p10 {np.percentile(v,10):7.0f}  p90 {np.percentile(v,90):7.0f}")
print(f"epilogues cycles/block: {epi[ok].mean():9.0f}   whole block: {tot[ok].mean():9.0f}   main loops: {(a[..., 2][ok]).mean():9.0f}")
