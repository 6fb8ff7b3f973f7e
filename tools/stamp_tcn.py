"""Diagnostic: per-wave cycle shares of the contraction main loop (stage / compute / land)
from a timing-only build (-DMSF2_DIAG_STAMP). Never used by the product."""
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
pk = PackedWeights(sd, cfg, dev)
lib = _lib.load()
B, S, Cc = 16, 3999, 512
x = torch.randn(B, S, Cc, device=dev); t = torch.randn(B, S, Cc, device=dev)
bstat = torch.tensor([[0.1, 1.3]] * B, device=dev)
for _ in range(3):
    ops.tcn_pointwise(x, t, bstat, pk.layers[0].tcn[0])
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (2048 * 16))()
raw.msf2_diag_read.argtypes = [C.c_void_p, C.c_size_t]
print("rc", raw.msf2_diag_read(buf, 2048 * 16))
a = np.array(buf, dtype=np.float64).reshape(2048, 4, 4)
it = a[..., 3].mean()
print("iterations", it)
for i, n in enumerate(("stage", "compute", "land")):
    v = a[..., i] / a[..., 3]
    print(f"{n:8s} cycles/iter: mean {v.mean():8.0f}  p10 {np.percentile(v,10):8.0f}  p90 {np.percentile(v,90):8.0f}")

buf2 = (C.c_ulonglong * (2048 * 8))()
raw.msf2_diag_read2.argtypes = [C.c_void_p, C.c_size_t]
raw.msf2_diag_read2(buf2, 2048 * 8)
b = np.array(buf2, dtype=np.float64).reshape(2048, 4, 2)
print(f"main (prologues + both K loops) cycles/block: mean {b[...,0].mean():8.0f} p10 {np.percentile(b[...,0],10):8.0f} p90 {np.percentile(b[...,0],90):8.0f}")
print(f"epilogue cycles/block:                        mean {b[...,1].mean():8.0f} p10 {np.percentile(b[...,1],10):8.0f} p90 {np.percentile(b[...,1],90):8.0f}")
