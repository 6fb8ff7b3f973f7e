"""bench.py's multi-rank code path on real hardware, as far as a one-GPU box allows: the process group is formed
over RCCL with ONE rank and every collective the N > 1 ranks execute (barriers on both sides of the timed region,
the gather of the separated waveforms each step, the gather / reduction of the per-rank times) runs on GPU tensors.
The launcher and the N = 2 control flow are rehearsed on the CPU (tests/test_bench_launcher_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_collectives_of_the_multi_rank_path_on_rccl():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--seconds", "1", "--collectives-at-one-rank", "--no-secondary", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["backend"] == "nccl" and d["world_size_seen"] == 1 and d["n_gpus"] == 1
    # 1 warm-up + 2 timed gathers, 2 fences (one barrier each), the time gather, the end-to-end reduction
    assert d["collectives_per_rank"] == 3 + 2 + 1 + 1
    assert d["value"] > 0 and len(d["ms_per_step_per_rank"]) == 1


SHARDED = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from speech_separation_amd.config import TINY
from speech_separation_amd.model import MossFormerHIP
from speech_separation_amd.parallel import separate_sharded
from speech_separation_amd.synth import make_state_dict, synth_mixture
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
m = MossFormerHIP(TINY, compute_device=dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(TINY, 0).items()})
utts, refs = [], []
for i, n in enumerate((8000, 12000, 8000, 9001, 12000)):
    mix, s1, s2 = synth_mixture(n, i)
    utts.append(mix); refs.append((s1, s2))
plain, st0 = separate_sharded(m, utts, device=dev, references=refs, max_batch=2)
gathered, st1 = separate_sharded(m, utts, device=dev, references=refs, max_batch=2, collectives_at_one_rank=True)
ok = all(np.array_equal(a, b) for a, b in zip(plain, gathered)) and np.allclose(st0, st1)
dist.barrier(); dist.destroy_process_group()
print(json.dumps({"ok": bool(ok), "n": len(gathered), "count": st1[2]}))
"""


def test_sharded_sweep_reduction_and_gather_on_rccl():
    """The utterance-sharded sweep (config 5's path): its SI-SDR reduction and its output gather over RCCL with one rank
    give what the collective-free single-process path gives."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    p = subprocess.run([sys.executable, "-c", SHARDED % ROOT], env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d == {"ok": True, "n": 5, "count": 5.0}


def test_sweep_workload_collectives_on_rccl():
    """bench.py --workload c5 (what N > 1 runs by default) with the process group formed over RCCL by ONE rank: the plan,
    the resident staging, the SI-SDR reduction and the gather of every output run as they do on N ranks."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "c5", "--utterances", "24",
                        "--steps", "3", "--warmup", "1", "--collectives-at-one-rank"],
                       env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["scaling"] == "strong" and d["n_gpus"] == 1
    assert d["utterances_in_statistic"] == 24.0 and d["outputs_finite"] and d["spot_check_rel_err_vs_alone"] <= 1e-5
    # warm-up exchange (reduce + gather), two fences, the timed exchange, the gather of the rank times
    assert d["collectives_per_rank"] == 2 + 2 + 2 + 1
    assert d["lpt_imbalance"] == 1.0 and len(d["busy_s_per_rank"]) == 1
