"""bench.py started plainly with --gpus N launches its own ranks.  Rehearsed here without a GPU:
gloo backend, CPU tensors and a stand-in model (--stub-cpu); what is checked is the launching, that
every rank joins the same collectives, and the single JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra_env=None, gpus=2, steps=3, warmup=1, workload="c2", extra=()):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH, "--gpus", str(gpus), "--steps", str(steps), "--warmup", str(warmup),
                           "--batch", "2", "--seconds", "0.5", "--stub-cpu"]
                          + (["--workload", workload] if workload else []) + list(extra),
                          env=env, capture_output=True, text=True, timeout=300)


def _line(r):
    # (gloo itself prints a "[Gloo] Rank 0 is connected ..." notice on stdout; RCCL does not)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_sweep_is_the_default_on_several_ranks_and_scales_strongly():
    """Without --workload, N > 1 runs the utterance sweep: one plan over all ranks, a step = 1/steps of each rank's
    share, SI-SDR sums reduced and every output gathered inside the timed region."""
    r = _run(workload=None, steps=4, warmup=2, extra=("--utterances", "90"))
    assert r.returncode == 0, r.stderr
    rec = _line(r)
    assert rec["n_gpus"] == 2 and rec["world_size_seen"] == 2 and rec["scaling"] == "strong"
    assert rec["steps"] == 4 and rec["warmup"] == 2
    assert rec["config"]["utterances"] == 90 and rec["utterances_in_statistic"] == 90.0
    assert len(rec["busy_s_per_rank"]) == 2 and len(rec["items_per_rank"]) == 2 and sum(rec["items_per_rank"]) > 0
    assert 1.0 <= rec["lpt_imbalance"] < 1.1
    assert rec["ms_per_step"] == max(rec["ms_per_step_per_rank"])
    assert rec["outputs_finite"] and rec["spot_check_rel_err_vs_alone"] == 0.0
    # warm-up exchange (reduce + gather), two fences, the timed exchange (reduce + gather), the gather of rank times
    assert rec["collectives_per_rank"] == 2 + 2 + 2 + 1
    assert rec["value"] > 0


def test_sweep_on_one_rank_needs_no_collective():
    r = _run(gpus=1, workload="c5", extra=("--utterances", "40"))
    assert r.returncode == 0, r.stderr
    rec = _line(r)
    assert rec["n_gpus"] == 1 and rec["scaling"] == "strong" and rec["collectives_per_rank"] == 0
    assert rec["lpt_imbalance"] == 1.0 and rec["utterances_in_statistic"] == 40.0


def test_plain_invocation_spawns_ranks_and_prints_one_line():
    r = _run()
    assert r.returncode == 0, r.stderr
    # (gloo itself prints a "[Gloo] Rank 0 is connected ..." notice on stdout; RCCL does not)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size_seen"] == 2 and rec["backend"] == "gloo"
    assert rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert len(rec["ms_per_step_per_rank"]) == 2
    assert rec["ms_per_step"] == max(rec["ms_per_step_per_rank"])
    # warm-up + timed steps (one gather each), two fences, the gather of rank times, the max of the
    # end-to-end times: the same count on every rank or the run would have hung
    assert rec["collectives_per_rank"] == (1 + 3) + 2 + 1 + 1
    assert rec["value"] > 0 and rec["value_incl_host_copies"] > 0


def test_single_rank_needs_no_launcher():
    r = _run(gpus=1)
    assert r.returncode == 0, r.stderr
    rec = json.loads(r.stdout.strip())
    assert rec["n_gpus"] == 1 and rec["world_size_seen"] == 1 and rec["collectives_per_rank"] == 0


def test_failed_rank_fails_the_launch():
    r = _run({"MSF2_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr
    assert "{" not in r.stdout


def test_under_an_external_launcher_env_is_respected():
    """torch.distributed.run sets WORLD_SIZE itself: bench.py must then be a rank, not a launcher."""
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}
    r = _run(env, gpus=1)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_under_torch_distributed_run():
    """The way a multi-GPU run is launched from outside: `python -m torch.distributed.run --nproc-per-node N bench.py
    --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher).  bench.py must then be a rank, join the
    launcher's rendezvous and print its single line on rank 0."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--batch", "2", "--seconds", "0.5", "--stub-cpu"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size_seen"] == 2 and len(rec["ms_per_step_per_rank"]) == 2
