"""Throughput benchmark of the MossFormer2_SS_8K separation forward on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 16] [--seconds 4]

Workloads (--workload; "auto" = c2 on one GPU, c5 on several):
  c2  one step = one forward pass over a batch of synthetic 8 kHz two-speaker
      mixtures already resident in HBM (the configuration the headline metric
      is quoted on: batch 16 of 4-second mixtures, fp32).  With N > 1 every
      rank separates its own batch (weak scaling) and the separated waveforms
      are gathered over RCCL after each step, inside the timed region.
  c4  the same with one 60-second mixture as the decode harness pads it
      (492000 samples, 61 499 frames); the rate is quoted on its 60 s.
  c5  the 3000-utterance sweep of 3-15 s mixtures, STRONG scaling: the sweep is
      planned once (exact-length batches, longest-processing-time-first over
      the ranks), every rank's waveforms are resident in HBM before the timed
      region, and the K steps are K consecutive slices of each rank's share --
      the timed region is exactly one pass over the whole sweep plus the
      reduction of the SI-SDR sums and the gather of all separated waveforms
      over RCCL.  No barrier separates the steps.

Launching: under ``torch.distributed.run`` the ranks are the processes it
started (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).
Started plainly with ``--gpus N`` (N > 1) this process only LAUNCHES: it starts
N fresh children, one per GPU, before anything here touches a GPU, waits for
them and exits with the first failure; rank 0's JSON line is the output.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_FILE = os.path.join(ROOT, "profiles", "tcn_pointwise_pmc.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("auto", "c2", "c4", "c5"), default="auto",
                    help="auto: c2 (batch of 4 s mixtures, weak scaling) on one GPU, c5 (3000-utterance sweep, strong scaling) on several")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--utterances", type=int, default=3000, help="c5: utterances in the sweep")
    ap.add_argument("--sweep-batch-seconds", type=float, default=0.0,
                    help="c5: audio seconds per batch (0: 64 in fp32, 256 in bf16)")
    ap.add_argument("--host-copy-pass", action="store_true",
                    help="c5: after the timed pass, separate the sweep once more from host memory (pinned staging, copies "
                         "overlapped with compute) and report that rate as value_incl_host_copies")
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                    help="fp32: exact fp32 MFMA (the headline configuration); bf16: bf16 MFMA operands, fp32 accumulate")
    ap.add_argument("--recurrent", choices=("tcn", "fsmn"), default="tcn",
                    help="block after each FLASH layer: tcn = GatedTCNBlock (the reference's live model, the headline); "
                         "fsmn = Gated_FSMN_Block_Dilated (the upstream architecture's block)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra records the default single-GPU fp32 run appends (bf16 batch 64, batch-1 latency)")
    ap.add_argument("--collectives-at-one-rank", action="store_true",
                    help="rehearsal on a one-GPU box: form the process group (RCCL) with a single rank and run every collective "
                         "of the multi-rank path (barriers, output gather, time gather / reduce) exactly as N > 1 ranks do")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="control-flow rehearsal without a GPU: gloo backend, CPU tensors and a trivial stand-in for the "
                         "model (checks launching, collectives per rank and the JSON line; measures nothing)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher (no GPU call, no torch import on this path)
# ------------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    """Start n fresh copies of this script, one per GPU, and wait for them.  The parent never
    touches a GPU (a process that has initialised the GPU must not be replaced or forked into
    ranks).  Returns the exit code: 0 when every rank succeeded."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0 writes the JSON line to our stdout; the other ranks have nothing to print
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
    def stop_children(signum=None, frame=None):
        for q in procs:
            if q.poll() is None:
                q.terminate()                        # the exact children started above
        for q in procs:
            try:
                q.wait(timeout=20)
            except subprocess.TimeoutExpired:
                q.kill()
        if signum is not None:
            sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, stop_children)
    signal.signal(signal.SIGINT, stop_children)
    rc = 0
    pending = set(range(n))
    try:
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            if pending:
                time.sleep(0.05)
    finally:
        stop_children()
    return rc


# ------------------------------------------------------------------------------------------------
# measurement helpers
# ------------------------------------------------------------------------------------------------
def dominant_kernel_probe(model, forward_once, B, T, which=0):
    """Average launch duration of the dominant kernel, measured in place: one more forward of the
    benchmark workload runs with the library's kernel timer on, which brackets each launch of the
    chosen kernel with HIP events on the stream it is launched on.  which = 0: the fused pointwise
    half of a TCN block (gemm_kernel<ProjNT<0,5>> in fp32, gemm_kernel_bf16_dma<ProjNT<0,5>> in
    bf16): [B*S, 512] x two 512x512 weights with the gate / GroupNorm / residual epilogue, 8 launches
    per layer x 24 layers = 53 % of the forward's FLOPs.  which = 1: the token-shifted to_hidden | to_qk
    projection of a FLASH layer ([B*S, 512] x [2176, 512]), one launch per layer -- the largest kernel
    when the recurrent block is the FSMN-family one.  `forward_once` must not contain a collective:
    only rank 0 runs the probe.  Returns (ms per launch, algorithmic FLOP per launch, launches)."""
    import ctypes
    from speech_separation_amd import _lib
    lib = _lib.load()
    S = lib.msf2_num_frames(T)
    C = model.cfg.dim
    _lib.check(lib.msf2_timing_begin_kernel(which))
    forward_once()
    total, n = ctypes.c_double(0.0), ctypes.c_int(0)
    _lib.check(lib.msf2_timing_end(ctypes.byref(total), ctypes.byref(n)))
    if n.value == 0:
        raise SystemExit("kernel timer saw no launch of the kernel it was asked to time")
    if which == 0:
        flops = 2.0 * B * S * 2 * C * C          # two [M,C]x[C,C] products per launch
    else:
        flops = 2.0 * B * S * C * (4 * C + 128)  # [M,C] x [4C+128,C]
    return total.value / n.value, flops, n.value


# Copy-only bound of the bf16 two-product contraction at 64 x 4 s (tools/lds_copy_rate.hip in the kernel's own access
# shape, no MFMAs / LDS reads / epilogue: profiles/r2_lds_copy_rate_shared.txt), ms per launch; scaled by rows for other shapes.
BF16_TCN_COPY_ONLY_MS_AT_255936_ROWS = 0.247
PEAK_FP32_MFMA, PEAK_BF16_MFMA, PEAK_HBM = 157.3, 2500.0, 8000.0        # TFLOP/s, TFLOP/s, GB/s (MI355X)


def roofline_record(model, prec_name, which, ms, flops, nl, B, T, traffic):
    """The `roofline` object of the JSON line for the kernel the probe timed."""
    C = model.cfg.dim
    M = B * model.cfg.frames(T)
    tf = flops / (ms * 1e-3) / 1e12
    if prec_name == "fp32":
        name = ("gemm_kernel<ProjNT<0,5>> (TCN pointwise, two products, fp32 MFMA 32x32x2)" if which == 0 else
                "gemm_kernel<ProjNT<1,0>> (token-shifted to_hidden|to_qk projection N=2176, fp32 MFMA 32x32x2)")
        return {"bound": "mfma", "kernel": name, "achieved": tf, "peak": PEAK_FP32_MFMA, "unit": "TFLOP/s",
                "frac": tf / PEAK_FP32_MFMA, "traffic": traffic, "launch_ms": ms, "launches_timed": nl}
    # bf16: neither roof binds -- the kernels are paced by the L2 -> LDS operand copies (see the copy-only bound); the record
    # carries the fraction of both roofs, `frac` is the MFMA one (the larger).  Algorithmic bytes: bf16 images of the
    # operands read once and the result written once.
    import torch
    resident = which == 0 and C == 512 and -(-model.cfg.frames(T) // 128) * B >= torch.cuda.get_device_properties(0).multi_processor_count
    if which == 0:
        # with a 128-row workgroup per CU or more, 7 of a layer's 8 launches are the register-resident form (the gated last block
        # stays on the tile kernel); the timer averages over all 8
        name = ("proj_astat_bf16_kernel<0,64> (TCN pointwise, two products; 32 rows of h and of t resident in registers per wave, weights "
                "streamed; bf16 MFMA 32x32x16) x7 + gemm_kernel_bf16_dma<ProjNT<0,6>> x1 per layer" if resident else
                "gemm_kernel_bf16_dma<ProjNT<0,5>> (TCN pointwise, two products, bf16 MFMA 32x32x16)")
        gbytes = 3.0 * M * C * 2 / 1e9                                    # h, t in; h' out
        copy_ms = None if resident else BF16_TCN_COPY_ONLY_MS_AT_255936_ROWS * M / 255936.0
    else:
        name = ("proj_astat_bf16_kernel<1,1> (token-shifted to_hidden|to_qk projection N=2176, activation rows resident in "
                "registers, bf16 MFMA 32x32x16)")
        gbytes = (M * C + M * (4 * C + 128)) * 2 / 1e9                    # x in; [v|u|qk] out
        copy_ms = None
    hbm = gbytes / (ms * 1e-3)
    binding = ("l2->lds operand copy path (neither roof)" if which == 0 and not resident else
               "in-order issue of one wave per SIMD: fragment reads, weight copies and epilogue arithmetic add to the MFMA time; "
               "the chip holds ~2.0-2.2 GHz of its 2.4 under this MFMA stream (neither roof)")
    return {"bound": "mfma", "binding_resource": binding, "kernel": name,
            "achieved": tf, "peak": PEAK_BF16_MFMA, "unit": "TFLOP/s", "frac": tf / PEAK_BF16_MFMA,
            "frac_mfma": tf / PEAK_BF16_MFMA, "frac_hbm": hbm / PEAK_HBM, "achieved_hbm_gb_s": hbm,
            "copy_only_bound_ms": copy_ms, "traffic": traffic, "launch_ms": ms, "launches_timed": nl}


def measured_traffic(precision, B, T, lib_version):
    """HBM bytes per launch of the dominant kernel from the PMC passes kept under profiles/
    (FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 runs, FETCH_SIZE doubled as gfx950
    requires).  None when no measurement exists for this precision and shape, or when it was taken
    on another version of the library (the kernel may have changed since)."""
    try:
        with open(PMC_FILE) as f:
            rec = json.load(f)[f"{precision}_b{B}_t{T}"]
        if int(rec.get("lib_version", -1)) != int(lib_version):
            return None
        return rec["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError, TypeError):
        return None


def timed_forwards(model, x, dev, steps, warmup):
    import torch
    for _ in range(warmup):
        out = model(x)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model(x)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps, out


def secondary_bf16(cfg, sd, dev, T, seconds, refn0, lib_version, B=64, steps=10, warmup=3):
    """bf16 MFMA mode on batch 64: whole-forward throughput, the roofline record of its dominant kernel and, when
    the CPU reference of utterance 0 is at hand, the SI-SDR difference to it (the contract for this mode is 0.05 dB)."""
    import torch
    from speech_separation_amd import _lib
    from speech_separation_amd.model import MossFormerHIP
    from speech_separation_amd.synth import synth_batch, synth_mixture, permute_si_sdr
    model = MossFormerHIP(cfg, compute_device=dev, precision=_lib.PREC_BF16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    x = torch.from_numpy(synth_batch(B, T, seed0=0)).to(dev)
    dt, out = timed_forwards(model, x, dev, steps, warmup)
    rec = {"value": B * T / 8000.0 / dt, "unit": "audio-s/s", "dtype": "bf16", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * dt, "model_tflops_per_s": B * cfg.flops(T) / dt / 1e12,
           "config": {"workload": f"MossFormer2_SS_8K batch={B} of synthetic {seconds:g} s mixtures, bf16 MFMA path"}}
    if refn0 is not None:
        got = torch.stack(out, 0)[:, :1].cpu().numpy()
        mix, s1, s2 = synth_mixture(T, 0)
        rec["si_sdr_delta_db"] = (permute_si_sdr(got[0, 0], got[1, 0], s1, s2)
                                  - permute_si_sdr(refn0[0, 0], refn0[1, 0], s1, s2))
    ms, flops, nl = dominant_kernel_probe(model, lambda: model(x), B, T)
    rec["roofline"] = roofline_record(model, "bf16", 0, ms, flops, nl, B, T, measured_traffic("bf16", B, T, lib_version))
    # the callers' own shape (inference.py / app.py feed one utterance per call), same weights
    x1 = x[:1].contiguous()
    dt1, _ = timed_forwards(model, x1, dev, 10, 3)
    rec["latency_b1_ms"] = 1e3 * dt1
    return rec


class _StubModel:
    """Stand-in for the separation network in --stub-cpu rehearsals: two scaled copies of the input."""
    class cfg:
        num_spks, dim, num_layers = 2, 512, 24

        @staticmethod
        def flops(T):
            return 0.0

    def __call__(self, x):
        return [0.5 * x, -0.5 * x]


# ------------------------------------------------------------------------------------------------
# the utterance sweep (strong scaling)
# ------------------------------------------------------------------------------------------------
class _SweepData:
    """The synthetic sweep: n utterances of 3.0 .. 15.0 s on a 0.5 s grid (25 distinct lengths, so that equal-length
    utterances can share a batch), each a rotation of its length's (mixture, source 1, source 2) triple.  Waveforms
    are produced on demand: a rank only ever touches its own share."""

    def __init__(self, n, unit=4000):
        import numpy as np
        self.np = np
        rng = np.random.default_rng(0)
        self.lengths = [int(v) for v in rng.integers(6, 31, size=n) * unit]
        self.unit = unit
        self._pool = {}

    def _triple(self, n):
        if n not in self._pool:
            from speech_separation_amd.synth import synth_mixture
            self._pool[n] = synth_mixture(n, n // self.unit)
        return self._pool[n]

    class _View:
        def __init__(self, data, which):
            self.data, self.which = data, which

        def __getitem__(self, i):
            d = self.data
            m, s1, s2 = d._triple(d.lengths[i])
            if self.which == 0:
                return d.np.roll(m, 37 * i)
            return (d.np.roll(s1, 37 * i), d.np.roll(s2, 37 * i))

    @property
    def utterances(self):
        return self._View(self, 0)

    @property
    def references(self):
        return self._View(self, 1)


def run_sweep(a, model, cfg, dev, rank, world, use_dist, stub, sync):
    """Workload c5: one pass over the whole sweep, utterance-sharded over the ranks (strong scaling)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from speech_separation_amd.parallel import ShardedSweep

    data = _SweepData(a.utterances, unit=(100 if stub else 4000))
    rate = 100.0 if stub else 8000.0
    budget_s = a.sweep_batch_seconds or (256.0 if a.precision == "bf16" else 64.0)
    sw = ShardedSweep(data.lengths, rank, world, max_batch=64, max_samples=int(budget_s * rate), num_spks=cfg.num_spks, device=dev)
    sw.alloc_flat()
    sw.stage_resident(data.utterances, data.references)          # inputs (and the sources of the statistic) resident before timing
    K, W = max(1, a.steps), max(0, a.warmup)
    chunks = [list(c) for c in np.array_split(np.arange(len(sw.mine)), K)]
    collectives = [0]

    def fence():
        sync()
        if use_dist:
            dist.barrier()
            collectives[0] += 1
        sync()

    def exchange():
        stats = sw.reduce_stats(use_dist)
        allbuf = None
        if use_dist:
            collectives[0] += 2
            allbuf = sw.gather_flat()
        return stats, allbuf

    # a line on stderr now and then: a pass over the whole sweep takes minutes on one GPU (nothing else is printed meanwhile)
    last_note = [time.perf_counter()]

    def note(what):
        if rank == 0 and time.perf_counter() - last_note[0] > 20.0:
            last_note[0] = time.perf_counter()
            print(f"[bench.py c5] {what}", file=sys.stderr, flush=True)

    sw.reset_stats()
    for w in range(W):                                           # untimed: the first W slices, and the exchange once
        sw.run_resident(model, chunks[w % K])
        note(f"warm-up slice {w + 1} of {W} enqueued")
    if W:
        exchange()
    sw.reset_stats()
    fence()
    t0 = time.perf_counter()
    for k in range(K):
        sw.run_resident(model, chunks[k])
        note(f"slice {k + 1} of {K} enqueued")
    sync()
    busy_mine = time.perf_counter() - t0                        # this rank's share separated
    stats, allbuf = exchange()
    fence()
    dt_mine = time.perf_counter() - t0

    t = torch.tensor([dt_mine, busy_mine], device=dev, dtype=torch.float64)
    if use_dist:
        allt = torch.empty(2 * world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, t)
        collectives[0] += 1
        allt = allt.cpu().view(world, 2)
        per_rank, busy = [float(v) for v in allt[:, 0]], [float(v) for v in allt[:, 1]]
    else:
        per_rank, busy = [dt_mine], [busy_mine]
    dt = max(per_rank)
    total_s = float(sum(data.lengths)) / rate
    n_sum, n_sq, n_cnt = stats
    res = {
        "metric": "audio-sec separated/sec (8 kHz)",
        "value": total_s / dt,
        "unit": "audio-s/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": 1e3 * dt / K,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if a.precision == "fp32" else "bf16",
        "data": "synthetic",
        "config": {"workload": f"MossFormer2_SS_8K sweep of {a.utterances} synthetic Libri2Mix-shaped 8 kHz utterances of 3-15 s "
                               f"({len(set(data.lengths))} exact-length buckets), {a.precision}, utterance-sharded over {world} GPU(s), "
                               "SI-SDR sums reduced and all separated waveforms gathered over RCCL inside the timed region; "
                               "a step is 1/steps of every rank's share; random-init weights (seeded recipe)",
                   "utterances": a.utterances, "audio_s": total_s, "batch_budget_audio_s": budget_s,
                   "layers": cfg.num_layers, "dim": cfg.dim,
                   "parallelism": f"utterance-sharded x{world} (longest-processing-time-first over exact-length batches)"},
        "world_size_seen": dist.get_world_size() if use_dist else 1,
        "backend": (dist.get_backend() if use_dist else None),
        "ms_per_step_per_rank": [1e3 * v / K for v in per_rank],
        "busy_s_per_rank": busy,                                 # separation only, before the exchange
        "exchange_s": dt - max(busy),
        "lpt_imbalance": sw.imbalance,                           # max rank load / mean rank load of the plan (samples)
        "items_per_rank": [len(p) for p in sw.plan],
        "mean_best_perm_si_sdr_db": (n_sum / n_cnt) if n_cnt else None,
        "utterances_in_statistic": n_cnt,
        "model_tflops_per_s": (sum(cfg.flops(n) for n in data.lengths) / dt / 1e12) if not stub else 0.0,
    }
    if stub or a.collectives_at_one_rank:
        res["collectives_per_rank"] = collectives[0]
    if stub:
        res["stub"] = True
    if rank == 0:
        # correctness of what was timed: everything finite, and the longest and the shortest utterance of rank 0's share equal
        # to a separation of that utterance alone (exact-length batching makes batch composition irrelevant)
        flat = allbuf[0] if allbuf is not None else sw.buf
        res["outputs_finite"] = bool(torch.isfinite(allbuf if allbuf is not None else sw.buf).all().item())
        worst = 0.0
        for k in (0, len(sw.mine) - 1) if sw.mine else ():
            n, idx = sw.mine[k]
            x1 = sw.staged[k][0][:1].contiguous()
            alone = torch.stack(list(model(x1)), 0)[:, 0]                                  # [spk, n]
            got = flat[sw.offsets[k]:sw.offsets[k] + cfg.num_spks * n].view(cfg.num_spks, n)
            worst = max(worst, float((alone - got).abs().max().item()) / max(float(alone.abs().max().item()), 1e-30))
        res["spot_check_rel_err_vs_alone"] = worst
    if a.host_copy_pass and not stub:
        # the same sweep from host memory: pinned staging, copies under compute, SI-SDR sums on the device
        from speech_separation_amd.parallel import separate_sharded
        done = [0]

        def model_noting(xb):
            done[0] += 1
            note(f"host-copy pass: batch {done[0]} enqueued")
            return model(xb)
        fence()
        t0 = time.perf_counter()
        separate_sharded(model_noting, data.utterances, rank, world, max_batch=64, device=dev, references=data.references,
                         max_samples=int(budget_s * rate), lengths=data.lengths)
        fence()
        th = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
        res["value_incl_host_copies"] = total_s / float(th.item())
    if rank == 0 and not stub and world == 1:
        # roofline of the dominant kernel on this workload's largest batch
        n, idx = sw.mine[0]
        x0 = sw.staged[0][0]
        ms, flops, nl = dominant_kernel_probe(model, lambda: model(x0), len(idx), n)
        res["roofline"] = roofline_record(model, a.precision, 0, ms, flops, nl, len(idx), n, None)
        res["roofline"]["shape"] = f"batch {len(idx)} x {n / 8000:g} s"
    return res


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def run_rank(a):
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    stub = a.stub_cpu
    if stub and os.environ.get("MSF2_BENCH_STUB_FAIL_RANK") == str(rank):
        raise SystemExit(3)                      # rehearsal of a rank that dies before the rendezvous
    if stub:
        dev = torch.device("cpu")
        sync = lambda: None                                           # noqa: E731
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = lambda: torch.cuda.synchronize(dev)                    # noqa: E731
    use_dist = world > 1 or a.collectives_at_one_rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise SystemExit("MASTER_PORT is not set: start the ranks with `bench.py --gpus N` or torch.distributed.run")
            with socket.socket() as sk:                  # one-rank rehearsal: any free port will do
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    workload = a.workload if a.workload != "auto" else ("c2" if world == 1 else "c5")
    B, T = a.batch, int(round(a.seconds * 8000))
    audio_s_per_utt = T / 8000.0
    if workload == "c4":
        # what the decode harness feeds the model for a 60 s file: 480000 samples padded to 492000
        # (reference: utils/decode.py:43-51); the rate is quoted on the 60 s of audio it came from
        from speech_separation_amd.decode import padded_length
        B, T, audio_s_per_utt = 1, padded_length(60 * 8000, 8000, 6000), 60.0
    if stub:
        model, cfg, sd, prec, lib = _StubModel(), _StubModel.cfg, None, 0, None
        x = torch.from_numpy(np.random.default_rng(rank).standard_normal((B, T)).astype(np.float32))
    else:
        from speech_separation_amd import _lib
        from speech_separation_amd.config import MossFormer2Config
        from speech_separation_amd.model import MossFormerHIP
        from speech_separation_amd.synth import make_state_dict, synth_batch
        cfg = MossFormer2Config(recurrent=a.recurrent)
        lib = _lib.load()
        sd = make_state_dict(cfg, 0)
        prec = _lib.PREC_BF16 if a.precision == "bf16" else _lib.PREC_FP32
        model = MossFormerHIP(cfg, compute_device=dev, precision=prec)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval()
        x = None if workload == "c5" else torch.from_numpy(synth_batch(B, T, seed0=rank * B)).to(dev)

    if workload == "c5":
        res = run_sweep(a, model, cfg, dev, rank, world, use_dist, stub, sync)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    gathered = torch.empty((world * cfg.num_spks, B, T), device=dev) if use_dist else None     # rank-major concatenation
    collectives = [0]

    def step():
        out = torch.stack(model(x), 0)
        if use_dist:
            dist.all_gather_into_tensor(gathered, out)      # every rank ends a step holding all separated waveforms
            collectives[0] += 1
        return out

    def fence():
        sync()
        if use_dist:
            dist.barrier()
            collectives[0] += 1
        sync()

    for _ in range(a.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    dt_mine = time.perf_counter() - t0

    # per-rank times (every rank joins), the job's time is the slowest rank's
    t = torch.tensor([dt_mine], device=dev, dtype=torch.float64)
    if use_dist:
        allt = torch.empty(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, t)
        collectives[0] += 1
        per_rank = [float(v) for v in allt.cpu()]
    else:
        per_rank = [dt_mine]
    dt = max(per_rank)

    # the same step with the host copies the callers pay for (waveforms host -> device, separated waveforms
    # device -> host, pinned buffers); no collective in here
    e2e_steps = max(1, min(a.steps, 5))
    if stub:
        e2e_mine = dt_mine / a.steps
    else:
        xh = x.cpu().pin_memory()
        oh = torch.empty((cfg.num_spks, B, T), dtype=torch.float32).pin_memory()
        xd = torch.empty_like(x)
        sync()
        t0 = time.perf_counter()
        for _ in range(e2e_steps):
            xd.copy_(xh, non_blocking=True)
            oh.copy_(torch.stack(model(xd), 0), non_blocking=True)
        sync()
        e2e_mine = (time.perf_counter() - t0) / e2e_steps
    t = torch.tensor([e2e_mine], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        collectives[0] += 1
    e2e = float(t.item())

    audio_s_step = world * B * audio_s_per_utt
    wl_name = (f"MossFormer2_SS_8K batch={B} of synthetic {a.seconds:g} s Libri2Mix-min-shaped 8 kHz mixtures" if workload == "c2" else
               "MossFormer2_SS_8K long form: one synthetic 60 s 8 kHz mixture as the decode harness pads it (492000 samples), batch 1")
    res = {
        "metric": "audio-sec separated/sec (8 kHz)",
        "value": audio_s_step * a.steps / dt,
        "unit": "audio-s/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if prec == 0 else "bf16",
        "data": "synthetic",
        "config": {"workload": f"{wl_name}, {a.precision}, random-init weights (seeded recipe)"
                               + ("" if a.recurrent == "tcn" else ", Gated_FSMN_Block_Dilated as the recurrent block"),
                   "batch_per_gpu": B, "samples": T, "layers": cfg.num_layers, "dim": cfg.dim,
                   "parallelism": f"utterance-sharded x{world}"},
        "world_size_seen": dist.get_world_size() if use_dist else 1,
        "backend": (dist.get_backend() if use_dist else None),
        "ms_per_step_per_rank": [1e3 * v / a.steps for v in per_rank],
        # inputs resident in HBM ("value") vs. host waveforms in, host waveforms out
        "value_incl_host_copies": audio_s_step / e2e,
        "ms_per_step_incl_host_copies": 1e3 * e2e,
        "model_tflops_per_s": world * B * cfg.flops(T) * a.steps / dt / 1e12,
    }
    if stub:
        res["stub"] = True
        res["collectives_per_rank"] = collectives[0]
    if a.collectives_at_one_rank:
        res["rehearsal"] = "process group formed with one rank; every collective of the multi-rank path executed"
        res["collectives_per_rank"] = collectives[0]

    if rank == 0 and not stub:
        from speech_separation_amd import _lib
        from speech_separation_amd.synth import synth_batch, synth_mixture, permute_si_sdr
        which = 0 if a.recurrent == "tcn" else 1          # FSMN form: no TCN pointwise kernel; its largest kernel is the N = 2176 projection
        ms, flops, nl = dominant_kernel_probe(model, lambda: model(x), B, T, which)
        traffic = measured_traffic(a.precision, B, T, lib.msf2_version()) if which == 0 else None
        res["roofline"] = roofline_record(model, a.precision, which, ms, flops, nl, B, T, traffic)
        if not a.no_cpu_baseline and world == 1 and workload == "c2":
            from oracle import cpu_ref           # the CPU port that is timed, never the product path
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            want = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(avail, 16)
            torch.set_num_threads(max(1, min(want, avail)))
            cores = torch.get_num_threads()
            W = cpu_ref.to_torch(sd)
            xs = torch.from_numpy(synth_batch(1, T, seed0=0))
            times = []
            with torch.no_grad():
                for i in range(4):               # one warm-up pass, then the median of three
                    c0 = time.perf_counter()
                    ref = cpu_ref.forward(W, xs, cfg)
                    times.append(time.perf_counter() - c0)
            cdt = sorted(times[1:])[1]
            got = out[:, :1].cpu().numpy()       # utterance 0 of the last timed step (same seed as the CPU sample)
            refn = torch.stack(ref, 0).numpy()
            mix, s1, s2 = synth_mixture(T, 0)
            sdr_cpu = permute_si_sdr(refn[0, 0], refn[1, 0], s1, s2)
            sdr_gpu = permute_si_sdr(got[0, 0], got[1, 0], s1, s2)
            res["cpu_baseline"] = {"value": (T / 8000.0) / cdt, "unit": "audio-s/s", "cores": cores, "kind": "port",
                                   "sample": f"one {a.seconds:g} s mixture (batch 1), fp32, oracle/cpu_ref.py on {cores} "
                                             f"threads ({avail} CPUs available to the process): median of 3 passes after 1 warm-up",
                                   "pass_s": times}
            res["max_abs_err_vs_cpu"] = float(np.abs(got - refn).max())
            res["si_sdr_delta_db"] = sdr_gpu - sdr_cpu
        if world == 1 and not a.no_secondary and workload == "c2":
            # the callers' own shape: one utterance per call (inference.py:35-41, app.py:269)
            dt1, _ = timed_forwards(model, x[:1].contiguous(), dev, 10, 3)
            res["latency"] = {"workload": f"batch 1 x {a.seconds:g} s, {a.precision}", "ms": 1e3 * dt1,
                              "audio_s_per_s": (T / 8000.0) / dt1}
        if world == 1 and prec == _lib.PREC_FP32 and not a.no_secondary and a.recurrent == "tcn" and workload == "c2":
            # the bf16 MFMA configuration (batch 64 of the same mixtures), measured the same way, as a second record
            refn0 = refn if "cpu_baseline" in res else None
            del model
            torch.cuda.empty_cache()
            res["secondary"] = secondary_bf16(cfg, sd, dev, T, a.seconds, refn0, lib.msf2_version())
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, argv))
    run_rank(a)


if __name__ == "__main__":
    main()
