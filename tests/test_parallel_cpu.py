"""Utterance sharding and the gather, on CPU with the gloo backend and
world_size 2 (the model is the CPU oracle on the tiny configuration)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref
from speech_separation_amd.config import TINY
from speech_separation_amd.parallel import assign_items, make_work_items, separate_sharded
from speech_separation_amd.synth import make_state_dict, synth_mixture


def _utterances():
    lens = [4000, 6000, 4000, 8000, 6000, 4000, 4000]
    mixes, refs = [], []
    for i, n in enumerate(lens):
        m, s1, s2 = synth_mixture(n, 100 + i)
        mixes.append(m)
        refs.append((s1, s2))
    return mixes, refs


def _model():
    W = cpu_ref.to_torch(make_state_dict(TINY, 0))

    def fn(x):
        with torch.no_grad():
            return cpu_ref.forward(W, x, TINY)
    return fn


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    mixes, refs = _utterances()
    outs, stats = separate_sharded(_model(), mixes, rank, world, max_batch=2, device="cpu", references=refs)
    if rank == 0:
        q.put(([o.tolist() for o in outs], stats))
    else:
        assert outs is None
        q.put((None, stats))
    dist.barrier()
    dist.destroy_process_group()


def test_plan_is_balanced_and_complete():
    lens = [24000 + 4000 * (i % 25) for i in range(3000)]
    items = make_work_items(lens, 16)
    assert sorted(i for _, idx in items for i in idx) == list(range(3000))
    assert all(len(idx) <= 16 for _, idx in items)
    assert all(len({lens[i] for i in idx}) == 1 for _, idx in items)       # exact-length batches only
    for world in (1, 2, 4, 8):
        plan = assign_items(items, world)
        loads = [sum(n * len(idx) for n, idx in p) for p in plan]
        assert sum(loads) == sum(lens)
        assert max(loads) <= 1.02 * (sum(loads) / world)                    # near-linear utterance sharding
    assert assign_items(items, 4) == assign_items(items, 4)                 # deterministic


def test_two_rank_gloo_matches_single_process():
    mixes, refs = _utterances()
    single, stats1 = separate_sharded(_model(), mixes, 0, 1, max_batch=2, references=refs)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    outs = next(o for o, _ in results if o is not None)
    for got, ref, mix in zip(outs, single, mixes):
        got = np.asarray(got, dtype=np.float32)
        assert got.shape == (2, len(mix))
        assert np.abs(got - ref).max() <= 1e-6          # batch composition differs between the two runs
    for _, st in results:
        assert st[2] == len(mixes)
        assert st[0] == pytest.approx(stats1[0], abs=1e-3)
