"""Diagnostic: bf16 batch-64 forward time with a kernel-choice knob on and off (same process, same box).
    python tools/ab_knob.py attn_fused [batch [fp32]]        (AB_FIXED=name=value,... sets further knobs for both legs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_separation_amd import _lib
from speech_separation_amd.config import FULL_8K as cfg
from speech_separation_amd.model import MossFormerHIP
from speech_separation_amd.synth import make_state_dict, synth_batch
knob = sys.argv[1].encode()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
prec = _lib.PREC_FP32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else _lib.PREC_BF16
lib = _lib.load()
m = MossFormerHIP(cfg, compute_device="cuda:0", precision=prec)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()})
x = torch.from_numpy(synth_batch(B, 32000, seed0=0)).to("cuda:0")
fixed = [kv.split("=") for kv in os.environ.get("AB_FIXED", "").split(",") if kv]     # e.g. AB_FIXED=no_big=1
vals = [v.encode() for v in os.environ.get("AB_VALUES", "1,0,1,0").split(",")]
for val in vals:
    for k, v in fixed:
        _lib.check(lib.msf2_debug_kernel_choice(k.encode(), v.encode()))
    _lib.check(lib.msf2_debug_kernel_choice(knob, val))
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        out = m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print(f"{knob.decode()}={val.decode()}: {1e3 * dt:8.2f} ms per forward, {B * 4 / dt:8.1f} audio-s/s", flush=True)
