import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_separation_amd import _lib
from speech_separation_amd.config import FULL_8K
from speech_separation_amd.model import MossFormerHIP
from speech_separation_amd.synth import make_state_dict, synth_batch
cfg = FULL_8K
m = MossFormerHIP(cfg, compute_device="cuda:0", precision=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()})
x = torch.from_numpy(synth_batch(1, 32000, seed0=0)).to("cuda:0")
ref = torch.stack(m(x), 0)
torch.cuda.synchronize()
m.use_graphs = True
outs = [torch.stack(m(x), 0) for _ in range(6)]          # back to back, no sync
torch.cuda.synchronize()
print("back-to-back replays: max diff to eager", [float((o - ref).abs().max()) for o in outs], "peaks", [float(o.abs().max()) for o in outs])
outs = []
for _ in range(4):
    outs.append(torch.stack(m(x), 0))
    torch.cuda.synchronize()
print("synced replays: max diff", [float((o - ref).abs().max()) for o in outs])
graph, xin, out, ws = m._graphs[(1, 32000)]
print("xin == x", torch.equal(xin, x), "x peak", float(x.abs().max()), "xin peak", float(xin.abs().max()))
m.use_graphs = False
e = torch.stack(m(x), 0)
print("eager again: diff", float((e - ref).abs().max()))
xin.copy_(x); torch.cuda.synchronize()
graph.replay(); torch.cuda.synchronize()
print("manual replay after sync: diff", float((out - ref[:, :, :]).abs().max()), "out peak", float(out.abs().max()))
graph.replay(); graph.replay(); torch.cuda.synchronize()
print("two more: diff", float((out - ref).abs().max()))
