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

Nothing in the per-batch loop waits for the GPU: waveforms are staged through
two pinned host buffers and copied on a side stream while the batch before is
being separated, the SI-SDR sums stay on the device until the end, and the
separated waveforms either land in the flat device buffer that the final
gather sends (several ranks) or are copied to the host batch by batch through
two pinned buffers (one process).
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


class ShardedSweep:
    """One rank's share of an utterance sweep: the plan every rank derives from the lengths alone, this rank's
    items, its slice of the flat output layout, and the loop that separates them.

    Flat layout (the payload of the final gather): this rank's items in plan order, each item a contiguous
    [B, num_spks, n] block, i.e. utterance after utterance as [num_spks, n]."""

    def __init__(self, lengths, rank=0, world=1, max_batch=16, max_samples=LIBRARY_MAX_SAMPLES, num_spks=2, device="cpu"):
        self.lengths = [int(n) for n in lengths]
        self.rank, self.world, self.num_spks = int(rank), int(world), int(num_spks)
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.plan = assign_items(make_work_items(self.lengths, max_batch, max_samples), self.world)
        self.mine = self.plan[self.rank]
        self.loads = [sum(n * len(idx) for n, idx in p) for p in self.plan]          # samples per rank
        self.size = max(max(self.loads) * self.num_spks, 1)                          # floats per rank in the gather
        self.offsets, off = [], 0
        for n, idx in self.mine:
            self.offsets.append(off)
            off += self.num_spks * n * len(idx)
        self.max_item = max((n * len(idx) for n, idx in self.mine), default=0)       # samples in my largest item
        self.buf = None                       # flat device buffer (several ranks / resident mode)
        self.acc = None                       # (sum, sum of squares, count) of best-permutation SI-SDR, on the device
        self.staged = None                    # resident mode: [(x, src or None)] per item, on the device
        self.local = {}                       # single-process streaming mode: utterance index -> host array

    @property
    def imbalance(self):
        """max rank load / mean rank load of the plan (1.0 = perfectly even); the only scaling loss of the sweep."""
        mean = sum(self.loads) / float(self.world)
        return max(self.loads) / mean if mean > 0 else 1.0

    # -- buffers ---------------------------------------------------------------------------------
    def alloc_flat(self):
        if self.buf is None:
            self.buf = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        return self.buf

    def reset_stats(self):
        self.acc = torch.zeros(3, dtype=torch.float64, device=self.device)

    def stage_resident(self, utterances, references=None):
        """Copy this rank's items to the device once (inputs resident in HBM before any timed region)."""
        self.staged = []
        for n, idx in self.mine:
            x = torch.from_numpy(np.stack([utterances[i] for i in idx], 0).astype(np.float32)).to(self.device)
            src = None
            if references is not None:
                src = torch.from_numpy(np.stack([np.stack([references[i][k] for i in idx], 0) for k in range(2)], 0)
                                       .astype(np.float32)).to(self.device)
            self.staged.append((x, src))
        return self.staged

    # -- one item ----------------------------------------------------------------------------------
    def _accumulate(self, out, src, idx, references):
        """Best-permutation SI-SDR of one batch added to the running sums (no host synchronisation on the GPU)."""
        if self.on_gpu:
            from . import ops
            est = torch.stack(list(out), 0).contiguous()                       # [spk, B, T]
            _, best, _ = ops.pit_si_snr(est, src)
            v = best.double()
            self.acc += torch.stack([v.sum(), (v * v).sum(), torch.tensor(float(len(idx)), dtype=torch.float64, device=self.device)])
        else:
            from .synth import permute_si_sdr
            for j, i in enumerate(idx):
                e = [o[j].detach().cpu().numpy() for o in out]
                r = (src[0, j].numpy(), src[1, j].numpy()) if src is not None else references[i]
                v = permute_si_sdr(e[0], e[1], r[0], r[1])
                self.acc += torch.tensor([v, v * v, 1.0], dtype=torch.float64)

    def run_item(self, k, model_fn, x, src=None, references=None, into_flat=True):
        """Separate item k (x: [B, n] on the device).  Writes the block into the flat buffer when into_flat, updates
        the SI-SDR sums when sources are given, and returns the model's output list."""
        n, idx = self.mine[k]
        out = model_fn(x)
        if (src is not None or references is not None) and self.acc is not None:
            self._accumulate(out, src, idx, references)
        if into_flat:
            B = len(idx)
            dst = self.buf[self.offsets[k]:self.offsets[k] + B * self.num_spks * n].view(B, self.num_spks, n)
            for s in range(self.num_spks):
                dst[:, s].copy_(out[s])
        return out

    def run_resident(self, model_fn, items=None, into_flat=True):
        """Separate (a subset of) the items staged by stage_resident."""
        for k in (range(len(self.mine)) if items is None else items):
            x, src = self.staged[k]
            self.run_item(k, model_fn, x, src, into_flat=into_flat)

    # -- streaming from host memory ------------------------------------------------------------------
    def run_streaming(self, model_fn, utterances, references=None, keep_on_device=True):
        """Separate this rank's items from host waveforms.  On the GPU the copy of item k+1 is issued (pinned buffer,
        side stream) before item k's launches, so it runs under item k's compute; with keep_on_device the results go
        into the flat device buffer, otherwise each block is copied to the host through one of two pinned buffers
        while the next item is being separated and ends up in self.local."""
        nspk = self.num_spks
        with_refs = references is not None
        if not self.on_gpu:
            for k, (n, idx) in enumerate(self.mine):
                x = torch.from_numpy(np.stack([utterances[i] for i in idx], 0).astype(np.float32))
                out = self.run_item(k, model_fn, x, None, references if with_refs else None, into_flat=keep_on_device)
                if not keep_on_device:
                    blk = torch.stack(list(out), 1)
                    for j, i in enumerate(idx):
                        self.local[i] = blk[j].detach().numpy().copy()
            return
        dev = self.device
        per = 3 if with_refs else 1                                   # mixture (+ two sources) per utterance
        cur = torch.cuda.current_stream(dev)
        h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        pin_in = [torch.empty(max(per * self.max_item, 1), dtype=torch.float32).pin_memory() for _ in range(2)]
        in_free = [None, None]                                        # event: the copy out of pin_in[p] has finished
        pin_out = None if keep_on_device else [torch.empty(max(nspk * self.max_item, 1), dtype=torch.float32).pin_memory()
                                               for _ in range(2)]
        pending = [None, None]                                        # (event, item index) of the copy into pin_out[p]

        def issue(k):
            n, idx = self.mine[k]
            B, p = len(idx), k & 1
            if in_free[p] is not None:
                in_free[p].synchronize()
            host = pin_in[p][:per * B * n].view(per, B, n).numpy()
            for j, i in enumerate(idx):
                host[0, j] = utterances[i]
                if with_refs:
                    host[1, j] = references[i][0]
                    host[2, j] = references[i][1]
            with torch.cuda.stream(h2d):
                d = pin_in[p][:per * B * n].view(per, B, n).to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(h2d)
            in_free[p] = ev
            return d, ev

        def drain(p):
            if pending[p] is None:
                return
            ev, k = pending[p]
            ev.synchronize()
            n, idx = self.mine[k]
            blk = pin_out[p][:len(idx) * nspk * n].view(len(idx), nspk, n).numpy()
            for j, i in enumerate(idx):
                self.local[i] = blk[j].copy()
            pending[p] = None

        nxt = issue(0) if self.mine else None
        for k, (n, idx) in enumerate(self.mine):
            d, ev = nxt
            nxt = issue(k + 1) if k + 1 < len(self.mine) else None
            cur.wait_event(ev)
            d.record_stream(cur)
            out = self.run_item(k, model_fn, d[0], d[1:3] if with_refs else None, into_flat=keep_on_device)
            if not keep_on_device:
                p = k & 1
                drain(p)                                              # the copy that used this buffer two items ago
                blk = torch.stack(list(out), 1)                       # [B, spk, n]
                done = torch.cuda.Event()
                done.record(cur)
                d2h.wait_event(done)
                with torch.cuda.stream(d2h):
                    pin_out[p][:blk.numel()].copy_(blk.reshape(-1), non_blocking=True)
                    e2 = torch.cuda.Event()
                    e2.record(d2h)
                blk.record_stream(d2h)
                pending[p] = (e2, k)
        if not keep_on_device:
            drain(0)
            drain(1)

    # -- the exchange -----------------------------------------------------------------------------------
    def reduce_stats(self, use_dist):
        t = self.acc.clone()
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return tuple(float(v) for v in t.cpu())

    def gather_flat(self):
        """all_gather_into_tensor of every rank's flat buffer; returns the [world, size] device tensor."""
        allbuf = torch.empty(self.world * self.size, dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(allbuf, self.buf)
        return allbuf.view(self.world, self.size)

    def unpack(self, allbuf_host):
        """Per-utterance [num_spks, n] arrays from the gathered [world, size] host array."""
        outputs = [None] * len(self.lengths)
        for r in range(self.world):
            off = 0
            for n, idx in self.plan[r]:
                for i in idx:
                    m = self.num_spks * n
                    outputs[i] = allbuf_host[r, off:off + m].reshape(self.num_spks, n).copy()
                    off += m
        return outputs


def separate_sharded(model_fn, utterances, rank=0, world=1, max_batch=16, device="cpu",
                     references=None, gather=True, max_samples=LIBRARY_MAX_SAMPLES, collectives_at_one_rank=False,
                     lengths=None):
    """Separate a list of 1-D float32 utterances, sharded over ``world`` ranks.

    model_fn(x[B, T] tensor on device) -> list of num_spks tensors [B, T].
    Returns (outputs, stats): on rank 0 (or every rank if world == 1) outputs
    is a list (per utterance) of arrays [num_spks, T]; elsewhere None.
    stats = (sum, sum_sq, count) of best-permutation SI-SDR over all ranks when
    ``references`` (list of (s1, s2)) is given, else None.
    collectives_at_one_rank: with world == 1 and an initialised process group, run the reduction and the gather anyway
    (a one-GPU rehearsal of what N > 1 ranks execute).
    lengths: the utterance lengths, when ``utterances`` is an indexable that produces waveforms on demand (only this
    rank's share is then ever touched).
    """
    use_dist = world > 1 or (collectives_at_one_rank and dist.is_initialized())
    if lengths is None:
        lengths = [len(u) for u in utterances]
    sw = ShardedSweep(lengths, rank, world, max_batch, max_samples, 2, device)
    if references is not None:
        sw.reset_stats()
    if use_dist and gather:
        sw.alloc_flat()
    sw.run_streaming(model_fn, utterances, references, keep_on_device=use_dist and gather)
    stats = sw.reduce_stats(use_dist) if references is not None else None
    if not use_dist:
        return [sw.local[i] for i in range(len(lengths))], stats
    if not gather:
        return None, stats
    allbuf = sw.gather_flat()
    if rank != 0:
        return None, stats
    return sw.unpack(allbuf.cpu().numpy()), stats
