"""Secondary workloads of the benchmark list, one JSON line each.

    python tools/bench_configs.py --config c4 [--precision fp32|bf16]
        one 60 s 8 kHz mixture as the decode harness pads it (492000 samples, 61 499 frames, 241 groups), batch 1
    python tools/bench_configs.py --config c5 [--precision ...] [--utterances 3000]
        Libri2Mix-test-sized sweep: synthetic utterances of 3-15 s (lengths on a 0.5 s grid so that
        equal-length utterances can share a batch), utterance-sharded over the ranks of
        torch.distributed.run when launched under it; the timed region covers host->device input
        copies, the forward passes and the gather of all outputs to rank 0.
Waveforms are synthetic; weights follow the seeded recipe.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=("c4", "c5"), required=True)
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32")
    ap.add_argument("--utterances", type=int, default=3000)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from speech_separation_amd import _lib
    from speech_separation_amd.config import FULL_8K as cfg
    from speech_separation_amd.model import MossFormerHIP
    from speech_separation_amd.parallel import separate_sharded
    from speech_separation_amd.synth import make_state_dict, synth_batch

    prec = _lib.PREC_BF16 if a.precision == "bf16" else _lib.PREC_FP32
    model = MossFormerHIP(cfg, compute_device=dev, precision=prec)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()})
    model.eval()

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if a.config == "c4":
        # what the decode harness feeds the model for a 60 s file: 480000 samples padded to 492000
        # (reference: utils/decode.py:43-51); the rate is quoted on the 60 s of audio it came from
        from speech_separation_amd.decode import padded_length
        T = padded_length(60 * 8000, 8000, 6000)
        assert T == 492000
        x = torch.from_numpy(synth_batch(1, T, seed0=7)).to(dev)
        model(x)
        sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            out = model(x)
        sync()
        dt = (time.perf_counter() - t0) / a.reps
        res = {"config": "c4: one 60 s 8 kHz mixture, batch 1 (492000 samples after the decode padding rule)", "precision": a.precision, "n_gpus": world,
               "audio_s_per_s": world * 60.0 / dt, "ms_per_utterance": 1e3 * dt,
               "finite": bool(torch.isfinite(torch.stack(out)).all().item())}
    else:
        rng = np.random.default_rng(0)
        lengths = (rng.integers(6, 31, size=a.utterances) * 4000).astype(int)        # 3.0 .. 15.0 s on a 0.5 s grid
        pool = {int(n): synth_batch(1, int(n), seed0=int(n) // 4000)[0] for n in np.unique(lengths)}
        utts = [np.roll(pool[int(n)], 37 * i) for i, n in enumerate(lengths)]
        budget = (64 if prec == _lib.PREC_FP32 else 256) * 8000                       # samples per batch
        calls = [0]

        def fn(xb):
            calls[0] += 1
            if rank == 0 and calls[0] % 25 == 0:
                print(f"[rank 0] batch {calls[0]}", file=sys.stderr, flush=True)
            return model(xb)
        separate_sharded(fn, utts[:32], rank, world, max_batch=64, device=dev, max_samples=budget)   # warm-up
        sync()
        t0 = time.perf_counter()
        outs, _ = separate_sharded(fn, utts, rank, world, max_batch=64, device=dev, max_samples=budget)
        sync()
        dt = time.perf_counter() - t0
        total_s = float(lengths.sum()) / 8000.0
        res = {"config": f"c5: {a.utterances} synthetic utterances of 3-15 s, utterance-sharded x{world}, outputs gathered on rank 0",
               "precision": a.precision, "n_gpus": world, "audio_s": total_s, "seconds": dt, "audio_s_per_s": total_s / dt}
        if rank == 0:
            res["outputs"] = len(outs)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
