"""Diagnostic: forward of a 2-layer full-width bf16 model with a kernel-choice knob set to 1 and to 0; prints how far the two
outputs are apart (bit-identical forms print 0 differing samples).
    python tools/cmp_knob.py <knob> <batch> <samples>"""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from speech_separation_amd import _lib
from speech_separation_amd.config import FULL_8K
from speech_separation_amd.model import MossFormerHIP
from speech_separation_amd.synth import make_state_dict, synth_batch, si_sdr
knob = sys.argv[1].encode(); B = int(sys.argv[2]); T = int(sys.argv[3])
cfg2 = type(FULL_8K)(num_layers=2)
m = MossFormerHIP(cfg2, compute_device="cuda:0", precision=_lib.PREC_BF16)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg2, 0).items()})
x = torch.from_numpy(synth_batch(B, T, seed0=7)).to("cuda:0")
lib = _lib.load()
outs = {}
for val in (b"1", b"0"):
    _lib.check(lib.msf2_debug_kernel_choice(None, None))
    _lib.check(lib.msf2_debug_kernel_choice(knob, val))
    outs[val] = torch.stack(m(x), 0).cpu().numpy()
a, b = outs[b"1"], outs[b"0"]
print("max abs diff", np.abs(a - b).max(), "peak", np.abs(b).max(), "n differing", int((a != b).sum()), "of", a.size)
print("si-sdr", [si_sdr(a[s, i], b[s, i]) for s in range(2) for i in range(B)])
d = np.abs(a - b)[0, 0]
nz = np.nonzero(d)[0]
print("first/last differing sample", nz[:5], nz[-5:])
