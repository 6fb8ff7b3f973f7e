"""Throughput benchmark of the MossFormer2_SS_8K separation forward on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 16] [--seconds 4]

One step = one forward pass over a batch of synthetic 8 kHz two-speaker
mixtures already resident in HBM (the configuration the headline metric is
quoted on: batch 16 of 4-second mixtures, fp32).  With N > 1 every rank (one
process per GPU, launched by torch.distributed.run) separates its own batch --
utterances are independent, so the path shards with no data-path collective;
the only exchange is the gather of the separated waveforms and SI-SDR sums,
done over RCCL after the timed region's compute and included in it.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def dominant_kernel_probe(model, step, B, T):
    """Average launch duration of the dominant kernel -- the fused pointwise half of a TCN
    block (gemm_kernel<ProjNT<0,5>> in fp32, gemm_kernel_bf16_dma<ProjNT<0,5>> in bf16):
    [B*S, 512] x two 512x512 weights with the gate / GroupNorm / residual epilogue, 8 launches
    per layer x 24 layers = 53 % of the forward's FLOPs.  Measured in place: one more step of
    the benchmark workload runs with the library's kernel timer on, which brackets each of its
    192 launches with HIP events on the stream they are launched on."""
    import ctypes
    from speech_separation_amd import _lib
    lib = _lib.load()
    S = lib.msf2_num_frames(T)
    C = model.cfg.dim
    _lib.check(lib.msf2_timing_begin())
    step()
    total, n = ctypes.c_double(0.0), ctypes.c_int(0)
    _lib.check(lib.msf2_timing_end(ctypes.byref(total), ctypes.byref(n)))
    if n.value == 0:
        raise SystemExit("kernel timer saw no launch of the TCN pointwise kernel")
    flops = 2.0 * B * S * 2 * C * C          # two [M,C]x[C,C] products per launch
    return total.value / n.value, flops, n.value


def measured_traffic(precision, B, T):
    """HBM bytes per launch of the dominant kernel from the PMC passes kept under profiles/
    (FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 runs, FETCH_SIZE doubled as gfx950
    requires); None when no measurement exists for this precision and shape."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_tcn_pointwise_pmc.json")
    try:
        with open(path) as f:
            return json.load(f)[f"{precision}_b{B}_t{T}"]["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def secondary_bf16(cfg, sd, dev, T, seconds, refn0, B=64, steps=3, warmup=2):
    """bf16 MFMA mode on batch 64: whole-forward throughput and, when the CPU reference of utterance 0 is at
    hand, the SI-SDR difference to it (the contract for this mode is 0.05 dB)."""
    from speech_separation_amd import _lib
    from speech_separation_amd.model import MossFormerHIP
    from speech_separation_amd.synth import synth_batch, synth_mixture, permute_si_sdr
    model = MossFormerHIP(cfg, compute_device=dev, precision=_lib.PREC_BF16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    x = torch.from_numpy(synth_batch(B, T, seed0=0)).to(dev)
    for _ in range(warmup):
        out = model(x)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model(x)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    rec = {"value": B * T / 8000.0 * steps / dt, "unit": "audio-s/s", "dtype": "bf16", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * dt / steps,
           "config": {"workload": f"MossFormer2_SS_8K batch={B} of synthetic {seconds:g} s mixtures, bf16 MFMA path"}}
    if refn0 is not None:
        got = torch.stack(out, 0)[:, :1].cpu().numpy()
        mix, s1, s2 = synth_mixture(T, 0)
        rec["si_sdr_delta_db"] = (permute_si_sdr(got[0, 0], got[1, 0], s1, s2)
                                  - permute_si_sdr(refn0[0, 0], refn0[1, 0], s1, s2))
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                    help="fp32: exact fp32 MFMA (the headline configuration); bf16: bf16 MFMA operands, fp32 accumulate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra bf16 batch-64 measurement that the default single-GPU fp32 run appends as 'secondary'")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from speech_separation_amd.config import FULL_8K as cfg
    from speech_separation_amd.model import MossFormerHIP
    from speech_separation_amd.synth import make_state_dict, synth_batch, synth_mixture, permute_si_sdr

    B, T = a.batch, int(round(a.seconds * 8000))
    sd = make_state_dict(cfg, 0)
    from speech_separation_amd import _lib
    prec = _lib.PREC_BF16 if a.precision == "bf16" else _lib.PREC_FP32
    model = MossFormerHIP(cfg, compute_device=dev, precision=prec)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    x = torch.from_numpy(synth_batch(B, T, seed0=rank * B)).to(dev)

    gathered = [torch.empty((cfg.num_spks, B, T), device=dev) for _ in range(world)] if world > 1 else None

    def step():
        out = torch.stack(model(x), 0)
        if world > 1:
            dist.all_gather(gathered, out)
        return out

    for _ in range(a.warmup):
        out = step()

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    audio_s = world * B * T / 8000.0 * a.steps
    res = {
        "metric": "audio-sec separated/sec (8 kHz)",
        "value": audio_s / dt,
        "unit": "audio-s/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if prec == _lib.PREC_FP32 else "bf16",
        "data": "synthetic",
        "config": {"workload": f"MossFormer2_SS_8K batch={B} of synthetic {a.seconds:g} s Libri2Mix-min-shaped 8 kHz mixtures, "
                               f"{a.precision}, random-init weights (seeded recipe)",
                   "batch_per_gpu": B, "samples": T, "layers": cfg.num_layers, "dim": cfg.dim,
                   "parallelism": f"utterance-sharded x{world}"},
        "model_tflops_per_s": world * B * cfg.flops(T) * a.steps / dt / 1e12,
    }

    if rank == 0:
        ms, flops, nl = dominant_kernel_probe(model, step, B, T)
        traffic = measured_traffic(a.precision, B, T)
        if prec == _lib.PREC_FP32:
            peak = 157.3
            res["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel<ProjNT<0,5>> (TCN pointwise, fp32 MFMA 32x32x2)",
                               "achieved": flops / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                               "frac": flops / (ms * 1e-3) / 1e12 / peak, "traffic": traffic,
                               "launch_ms": ms, "launches_timed": nl}
        else:
            # at bf16 MFMA rates the same kernel is bandwidth bound: it must read the bf16 images of h and t
            # and write the bf16 image of h' once (2 bytes per element each)
            S_ = model.cfg.frames(T)
            gbytes = 3.0 * B * S_ * cfg.dim * 2 / 1e9
            res["roofline"] = {"bound": "hbm", "kernel": "gemm_kernel_bf16_dma<ProjNT<0,5>> (TCN pointwise, bf16 MFMA 32x32x16)",
                               "achieved": gbytes / (ms * 1e-3), "peak": 8000.0, "unit": "GB/s",
                               "frac": gbytes / (ms * 1e-3) / 8000.0, "traffic": traffic, "launch_ms": ms, "launches_timed": nl,
                               "tflops": flops / (ms * 1e-3) / 1e12}
        if not a.no_cpu_baseline and world == 1:
            from oracle import cpu_ref           # the CPU port that is timed, never the product path
            cores = min(os.cpu_count() or 1, 16)
            torch.set_num_threads(cores)
            W = cpu_ref.to_torch(sd)
            xs = torch.from_numpy(synth_batch(1, T, seed0=0))
            with torch.no_grad():
                c0 = time.perf_counter()
                ref = cpu_ref.forward(W, xs, cfg)
                cdt = time.perf_counter() - c0
            got = out[:, :1].cpu().numpy()       # utterance 0 of the last timed step (same seed as the CPU sample)
            refn = torch.stack(ref, 0).numpy()
            mix, s1, s2 = synth_mixture(T, 0)
            sdr_cpu = permute_si_sdr(refn[0, 0], refn[1, 0], s1, s2)
            sdr_gpu = permute_si_sdr(got[0, 0], got[1, 0], s1, s2)
            res["cpu_baseline"] = {"value": (T / 8000.0) / cdt, "unit": "audio-s/s", "cores": cores, "kind": "port",
                                   "sample": f"one {a.seconds:g} s mixture (batch 1), fp32, one pass, oracle/cpu_ref.py"}
            res["max_abs_err_vs_cpu"] = float(np.abs(got - refn).max())
            res["si_sdr_delta_db"] = sdr_gpu - sdr_cpu
        if world == 1 and prec == _lib.PREC_FP32 and not a.no_secondary:
            # the bf16 MFMA configuration (batch 64 of the same mixtures), measured the same way, as a second record
            refn0 = refn if "cpu_baseline" in res else None
            del model
            torch.cuda.empty_cache()
            res["secondary"] = secondary_bf16(cfg, sd, dev, T, a.seconds, refn0)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
