"""Utterance-sharded separation over the GPUs of one node.

The forward pass has no intra-utterance shard: GroupNorm(1, C) statistics and
the linear-attention key/value summary couple every frame of an utterance
(reference: models/mossformer2/mossformer2_block.py:318-322, :521;
models/mossformer2/mossformer2.py:148).  Utterances are independent, so a
sweep is partitioned by utterance: one process per GPU (``torch.distributed``,
backend ``nccl`` = RCCL on ROCm, ``gloo`` on CPU for tests), every rank holds a
full weight replica, and the only exchange is the gather of separated
waveforms and of three SI-SDR sums after the compute.

Batching is only legal between utterances of EQUAL padded length (zero padding
changes the global statistics), so work items are exact-length buckets.
"""
from collections import defaultdict

import numpy as np
import torch
import torch.distributed as dist


# One call of the forward addresses its widest activation ([B * frames, 4 * 512 + 128] fp32) with 32-bit byte
# offsets, which caps a batch at about 3.9 M samples for the 512-channel model; batches are split to stay below it.
LIBRARY_MAX_SAMPLES = 3_800_000


def make_work_items(lengths, max_batch, max_samples=LIBRARY_MAX_SAMPLES):
    """Group utterance indices by exact length, split groups into batches of at
    most max_batch utterances (and, if given, at most max_samples samples in
    total, but never fewer than one utterance).  Returns [(length, [indices])],
    longest first."""
    groups = defaultdict(list)
    for i, n in enumerate(lengths):
        groups[int(n)].append(i)
    items = []
    for n in sorted(groups, reverse=True):
        idx = groups[n]
        step = max_batch if max_samples is None else max(1, min(max_batch, max_samples // n))
        for j in range(0, len(idx), step):
            items.append((n, idx[j:j + step]))
    return items


def assign_items(items, world):
    """Longest-processing-time-first greedy assignment of work items to ranks
    (cost = samples in the item).  Deterministic, so every rank computes the
    same plan without communicating.  Returns a list (per rank) of item lists."""
    load = [0] * world
    plan = [[] for _ in range(world)]
    for it in sorted(items, key=lambda t: (-t[0] * len(t[1]), t[1][0])):
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append(it)
        load[r] += it[0] * len(it[1])
    return plan


def separate_sharded(model_fn, utterances, rank=0, world=1, max_batch=16, device="cpu",
                     references=None, gather=True, max_samples=LIBRARY_MAX_SAMPLES, collectives_at_one_rank=False):
    """Separate a list of 1-D float32 utterances, sharded over ``world`` ranks.

    model_fn(x[B, T] tensor on device) -> list of num_spks tensors [B, T].
    Returns (outputs, stats): on rank 0 (or every rank if world == 1) outputs
    is a list (per utterance) of arrays [num_spks, T]; elsewhere None.
    stats = (sum, sum_sq, count) of best-permutation SI-SDR over all ranks when
    ``references`` (list of (s1, s2)) is given, else None.
    collectives_at_one_rank: with world == 1 and an initialised process group, run the reduction and the gather anyway
    (a one-GPU rehearsal of what N > 1 ranks execute).
    """
    use_dist = world > 1 or (collectives_at_one_rank and dist.is_initialized())
    from .synth import permute_si_sdr
    lengths = [len(u) for u in utterances]
    plan = assign_items(make_work_items(lengths, max_batch, max_samples), world)
    mine = plan[rank]
    local = {}
    on_gpu = torch.device(device).type == "cuda"
    acc = np.zeros(3)
    for n, idx in mine:
        x = torch.from_numpy(np.stack([utterances[i] for i in idx], 0).astype(np.float32)).to(device)
        out = model_fn(x)
        if references is not None and on_gpu:
            # the statistic is reduced where the waveforms are (msf2_pit_si_snr): B floats leave the device
            from . import ops
            est = torch.stack(list(out), 0).contiguous()                       # [spk, B, T]
            src = torch.from_numpy(np.stack([np.stack([references[i][k] for i in idx], 0) for k in range(2)], 0)
                                   .astype(np.float32)).to(device)
            _, best, _ = ops.pit_si_snr(est, src)
            v = best.double().cpu().numpy()
            acc += (v.sum(), (v * v).sum(), float(len(idx)))
        out = torch.stack(list(out), 1)                    # [B, spk, T]
        for j, i in enumerate(idx):
            local[i] = out[j]
    num_spks = next(iter(local.values())).shape[0] if local else 2

    stats = None
    if references is not None:
        if not on_gpu:
            for i, o in local.items():
                e = o.detach().cpu().numpy()
                v = permute_si_sdr(e[0], e[1], references[i][0], references[i][1])
                acc += (v, v * v, 1.0)
        t = torch.tensor(acc, dtype=torch.float64, device=device)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        stats = tuple(float(v) for v in t.cpu())

    if not use_dist:
        return [local[i].detach().cpu().numpy() for i in range(len(utterances))], stats
    if not gather:
        return None, stats

    # one flat fp32 buffer per rank, padded to the largest rank payload; the layout
    # of every rank's buffer follows from the shared plan
    def payload(items):
        return sum(n * len(idx) for n, idx in items) * num_spks
    size = max(max(payload(p) for p in plan), 1)
    buf = torch.zeros(size, dtype=torch.float32, device=device)
    off = 0
    for n, idx in mine:
        for i in idx:
            m = num_spks * n
            buf[off:off + m] = local[i].reshape(-1)
            off += m
    allbuf = torch.empty(world * size, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(allbuf, buf)
    if rank != 0:
        return None, stats
    allbuf = allbuf.cpu().numpy().reshape(world, size)
    outputs = [None] * len(utterances)
    for r in range(world):
        off = 0
        for n, idx in plan[r]:
            for i in idx:
                m = num_spks * n
                outputs[i] = allbuf[r, off:off + m].reshape(num_spks, n).copy()
                off += m
    return outputs, stats
