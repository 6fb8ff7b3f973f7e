"""Does the forward read workspace bytes it has not written?  Runs msf2_forward on a workspace pre-filled with
0x00, 0xFF (NaN patterns) and 0x7F and compares the outputs bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speech_separation_amd import _lib  # noqa: E402
from speech_separation_amd.config import FULL_8K, TINY, MossFormer2Config  # noqa: E402
from speech_separation_amd.model import MossFormerHIP  # noqa: E402
from speech_separation_amd.synth import make_state_dict, synth_batch  # noqa: E402

lib = _lib.load()
for cfg, B, T in ((TINY, 1, 4000), (TINY, 2, 8003), (MossFormer2Config(num_layers=2), 1, 32000), (MossFormer2Config(num_layers=2), 3, 12000)):
    for prec in (0, 1):
        m = MossFormerHIP(cfg, compute_device="cuda:0", precision=prec)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()})
        pk = m.packed()
        x = torch.from_numpy(synth_batch(B, T, seed0=0)).to("cuda:0")
        n = lib.msf2_workspace_bytes(pk.c_cfg, B, T)
        outs = []
        for fill in (0x00, 0xFF, 0x7F):
            ws = torch.full((n,), fill, dtype=torch.uint8, device="cuda:0")
            out = torch.empty((2, B, T), device="cuda:0")
            _lib.check(lib.msf2_forward(pk.c_cfg, pk.struct, x.data_ptr(), out.data_ptr(), B, T, ws.data_ptr(), n, None, None))
            torch.cuda.synchronize()
            outs.append(out)
        d1 = (outs[0] - outs[1]).abs().max().item()
        d2 = (outs[0] - outs[2]).abs().max().item()
        print(f"dim {cfg.dim} layers {cfg.num_layers} B {B} T {T} prec {prec}: finite {[bool(torch.isfinite(o).all()) for o in outs]} "
              f"max diff 0x00 vs 0xFF {d1:.3e}, vs 0x7F {d2:.3e}, peak {outs[0].abs().max().item():.3e}", flush=True)
