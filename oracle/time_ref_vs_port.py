"""Time the REFERENCE model's CPU forward beside the CPU port (oracle/cpu_ref.py) on one host: the same 4-second
mixture, the same seeded weights, the same thread count, median of 3 passes after 1 warm-up each.

TEST INFRASTRUCTURE ONLY, build container only (it imports /root/reference, which never travels to the GPU box).
bench.py times the port on the GPU box's host cores (`cpu_baseline`, kind "port"); the ratio written here relates that
figure to the reference's own path (MossFormer.forward, models/mossformer2/mossformer2.py:712-736).

    python oracle/time_ref_vs_port.py [--threads 8] [--out profiles/r3_cpu_ref_vs_port.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import cpu_ref, gen_golden  # noqa: E402
from speech_separation_amd.config import FULL_8K  # noqa: E402
from speech_separation_amd.synth import synth_batch  # noqa: E402


def median_of_3(fn):
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        out = fn()
        times.append(time.perf_counter() - t0)
    return sorted(times[1:])[1], times, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r3_cpu_ref_vs_port.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    gen_golden.install_shims()
    model, sd = gen_golden.build_reference(FULL_8K)
    x = torch.from_numpy(synth_batch(1, 32000, seed0=0))
    W = cpu_ref.to_torch(sd)
    with torch.no_grad():
        t_port, port_passes, o_port = median_of_3(lambda: cpu_ref.forward(W, x, FULL_8K))
        t_ref, ref_passes, o_ref = median_of_3(lambda: model(x))
    err = float(np.abs(torch.stack(o_port, 0).numpy() - torch.stack(o_ref, 0).numpy()).max())
    rec = {"input": "one synthetic 4 s 8 kHz mixture (seed 0), batch 1, fp32, 24-layer MossFormer2_SS_8K, seeded weights",
           "threads": torch.get_num_threads(), "host": "build container (8 vCPU)",
           "reference_forward_s": t_ref, "reference_passes_s": ref_passes,
           "port_forward_s": t_port, "port_passes_s": port_passes,
           "reference_audio_s_per_s": 4.0 / t_ref, "port_audio_s_per_s": 4.0 / t_port,
           "port_speedup_over_reference": t_ref / t_port, "max_abs_difference_of_outputs": err}
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
