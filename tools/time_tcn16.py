"""Diagnostic: run the bf16 gated TCN block of one layer a few times (for rocprofv3 --kernel-trace --stats);
the library is chosen with MSF2_LIB, so timing-only builds can be compared by kernel time.  Never used by the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_separation_amd import ops, _lib
from speech_separation_amd.config import MossFormer2Config
from speech_separation_amd.packing import PackedWeights
from speech_separation_amd.synth import make_state_dict
dev = "cuda:0"
cfg = MossFormer2Config(num_layers=1)
sd = {k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()}
pk = PackedWeights(sd, cfg, dev, 1)
x = torch.randn(64, 3999, 512, device=dev)
for kv in os.environ.get("AB_FIXED", "").split(","):       # AB_FIXED=name=value,...: kernel-choice knobs for this run
    if kv:
        k, v = kv.split("=")
        _lib.check(_lib.load().msf2_debug_kernel_choice(k.encode(), v.encode()))
for _ in range(6):
    ops.gated_tcn(x, pk.layers[0], 1)
torch.cuda.synchronize()
print("done")
