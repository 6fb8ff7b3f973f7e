"""Latency of the callers' own shape (one utterance per call): eager launch sequence vs HIP-graph replay.

    python tools/latency_b1.py [--seconds 4] [--reps 20]
Prints one JSON line per precision."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speech_separation_amd import _lib  # noqa: E402
from speech_separation_amd.config import FULL_8K as cfg  # noqa: E402
from speech_separation_amd.model import MossFormerHIP  # noqa: E402
from speech_separation_amd.synth import make_state_dict, synth_batch  # noqa: E402


def timed(model, x, reps):
    for _ in range(3):
        out = model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = model(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, torch.stack(out, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    T = int(a.seconds * 8000)
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()}
    x = torch.from_numpy(synth_batch(a.batch, T, seed0=0)).to("cuda:0")
    for name, prec in (("fp32", _lib.PREC_FP32), ("bf16", _lib.PREC_BF16)):
        m = MossFormerHIP(cfg, compute_device="cuda:0", precision=prec)
        m.load_state_dict(sd)
        eager, o1 = timed(m, x, a.reps)
        m.use_graphs = True
        graph, o2 = timed(m, x, a.reps)
        print(json.dumps({"workload": f"batch {a.batch} x {a.seconds:g} s", "precision": name, "eager_ms": eager,
                          "graph_ms": graph, "identical": bool(torch.equal(o1, o2)),
                          "max_diff": float((o1 - o2).abs().max()), "peak": float(o1.abs().max())}), flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
