"""Diagnostic: a few bf16 (or fp32) forwards of the 24-layer model with kernel-choice knobs set, for profiling under
rocprofv3 (per-kernel times of alternative kernel forms on one box).
    python tools/knob_forward.py [name=value,...] [batch] [fp32|bf16] [reps] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_separation_amd import _lib
from speech_separation_amd.config import FULL_8K as cfg
from speech_separation_amd.model import MossFormerHIP
from speech_separation_amd.synth import make_state_dict, synth_batch
knobs = [kv.split("=") for kv in (sys.argv[1] if len(sys.argv) > 1 else "").split(",") if kv and kv != "-"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
prec = _lib.PREC_FP32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else _lib.PREC_BF16
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
T = int(float(sys.argv[5]) * 8000) if len(sys.argv) > 5 else 32000
lib = _lib.load()
m = MossFormerHIP(cfg, compute_device="cuda:0", precision=prec)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()})
x = torch.from_numpy(synth_batch(B, T, seed0=0)).to("cuda:0")
for k, v in knobs:
    _lib.check(lib.msf2_debug_kernel_choice(k.encode(), v.encode()))
m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = m(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"knobs {sys.argv[1] if len(sys.argv) > 1 else ''}: {1e3 * dt:8.2f} ms per forward, {B * T / 8000 / dt:8.1f} audio-s/s, finite {bool(torch.isfinite(torch.stack(out)).all())}", flush=True)
