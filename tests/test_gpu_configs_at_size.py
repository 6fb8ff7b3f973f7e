"""The benchmark list's larger configurations at their own sizes, on the 24-layer model, through the C ABI:

* bf16 MFMA mode on a batch of 64 four-second mixtures (255 936 rows per activation: the largest row count and the
  widest 32-bit offsets any configuration reaches);
* the mixed-length sweep with every one of its 25 exact-length buckets (3.0 .. 15.0 s on a 0.5 s grid), through both
  forms of the sweep loop (waveforms resident on the device, as bench.py times it; waveforms streamed from the host).

Reference semantics: MossFormer.forward (models/mossformer2/mossformer2.py:712-736); utterances of a batch are
independent in the reference (no cross-utterance operation), which is what "equals the utterance alone" checks at
sizes the CPU oracle cannot reach in test time."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speech_separation_amd import _lib  # noqa: E402
from speech_separation_amd.config import FULL_8K  # noqa: E402
from speech_separation_amd.model import MossFormerHIP  # noqa: E402
from speech_separation_amd.parallel import ShardedSweep, separate_sharded  # noqa: E402
from speech_separation_amd.synth import make_state_dict, permute_si_sdr, si_sdr, synth_batch, synth_mixture  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def full_models():
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(FULL_8K, 0).items()}
    cache = {}

    def get(prec):
        if prec not in cache:
            m = MossFormerHIP(FULL_8K, compute_device=DEV, precision=prec)
            m.load_state_dict(sd)
            cache[prec] = m
        return cache[prec]
    return get


def test_bf16_batch_of_64_four_second_mixtures(full_models, golden_dir):
    """Utterance 0 of the batch is the input of the reference fixture (seed 0): its SI-SDR against the synthetic
    sources must stay within the 0.05 dB contract of the reference model's fp32 output, and the waveform within 38 dB
    of it (measured 42-43 dB).  Utterances 0, 31 and 63 must equal their separation alone to bf16 noise (>= 35 dB: alone,
    the small-grid kernel forms run, which differ from the large-grid ones by bf16 roundings)."""
    z = np.load(os.path.join(golden_dir, "fwd_full_B1_T32000.npz"))
    m = full_models(_lib.PREC_BF16)
    xb = synth_batch(64, 32000, seed0=0)
    assert np.array_equal(xb[0], z["input"][0])
    x = torch.from_numpy(xb).to(DEV)
    out = torch.stack(m(x), 0)                                   # [2, 64, 32000]
    assert out.shape == (2, 64, 32000) and torch.isfinite(out).all()
    got = out.cpu().numpy()
    ref = z["out"]
    mix, s1, s2 = synth_mixture(32000, 0)
    d = permute_si_sdr(got[0, 0], got[1, 0], s1, s2) - permute_si_sdr(ref[0, 0], ref[1, 0], s1, s2)
    fid = min(si_sdr(got[s, 0], ref[s, 0]) for s in range(2))
    print(f"bf16, 64 x 4 s: utterance 0 SI-SDR delta {d:+.4f} dB, fidelity to the fp32 reference {fid:.1f} dB")
    assert abs(d) <= 0.05
    assert fid >= 38.0
    for b in (0, 31, 63):
        alone = torch.stack(m(x[b:b + 1]), 0)[:, 0].cpu().numpy()
        worst = min(si_sdr(got[s, b], alone[s]) for s in range(2))
        print(f"bf16, 64 x 4 s: utterance {b} in the batch vs alone {worst:.1f} dB")
        assert worst >= 35.0
    # no utterance may be a copy of another or silent
    rms = np.sqrt((got.astype(np.float64) ** 2).mean(-1))
    assert (rms > 1e-6).all()


def _sweep_data(per_bucket):
    lens = [4000 * k for k in range(6, 31) for _ in range(per_bucket)]       # 3.0 .. 15.0 s, 25 buckets
    rng = np.random.default_rng(5)
    order = rng.permutation(len(lens))
    lens = [lens[i] for i in order]
    data = [synth_mixture(n, 300 + i) for i, n in enumerate(lens)]
    return lens, [d[0] for d in data], [(d[1], d[2]) for d in data]


def test_sweep_covers_every_length_bucket_fp32(full_models):
    """50 utterances, two per bucket, fp32: the resident-input loop (what bench.py --workload c5 times) and the
    host-streaming loop give identical waveforms; three of them (longest, shortest, one in between) equal the
    separation of that utterance alone; the SI-SDR sums reduced on the device equal the host formula."""
    m = full_models(_lib.PREC_FP32)
    lens, utts, refs = _sweep_data(2)
    budget = 64 * 8000
    calls = []

    def fn(x):
        calls.append(tuple(x.shape))
        return m(x)

    streamed, stats = separate_sharded(fn, utts, device=DEV, references=refs, max_batch=64, max_samples=budget)
    assert len(calls) == 25 and sorted(c[1] for c in calls) == sorted(set(lens)) and all(c[0] == 2 for c in calls)
    assert all(o.shape == (2, n) and np.isfinite(o).all() for o, n in zip(streamed, lens))

    sw = ShardedSweep(lens, 0, 1, max_batch=64, max_samples=budget, device=DEV)
    sw.alloc_flat()
    sw.stage_resident(utts, refs)
    sw.reset_stats()
    sw.run_resident(m)
    stats_res = sw.reduce_stats(False)
    resident = sw.unpack(sw.buf.cpu().numpy()[None])
    for a, b in zip(streamed, resident):
        assert np.array_equal(a, b)                               # same batches, same kernels: bit-identical
    assert stats_res[2] == len(lens) and stats[2] == len(lens)
    assert stats_res[0] == pytest.approx(stats[0], abs=1e-9)
    host = sum(permute_si_sdr(o[0], o[1], r[0], r[1]) for o, r in zip(streamed, refs))
    assert abs(stats[0] - host) <= 2e-2

    picks = [int(np.argmax(lens)), int(np.argmin(lens)), lens.index(72000)]
    for i in picks:
        alone = torch.stack(m(torch.from_numpy(utts[i][None]).to(DEV)), 0)[:, 0].cpu().numpy()
        err = np.abs(streamed[i] - alone).max() / np.abs(alone).max()
        print(f"sweep, fp32: utterance of {lens[i] / 8000:g} s in its batch vs alone: {err:.2e} of the peak")
        assert err <= 1e-5


def test_sweep_covers_every_length_bucket_bf16(full_models):
    """The same buckets in bf16 mode with the larger batch budget bench.py uses for it (256 audio-seconds per batch):
    finite everywhere, three utterances equal to their separation alone to bf16 noise."""
    m = full_models(_lib.PREC_BF16)
    lens, utts, refs = _sweep_data(3)
    outs, stats = separate_sharded(m, utts, device=DEV, references=refs, max_batch=64, max_samples=256 * 8000)
    assert stats[2] == len(lens) and np.isfinite(stats[0])
    assert all(o.shape == (2, n) and np.isfinite(o).all() for o, n in zip(outs, lens))
    for i in (int(np.argmax(lens)), int(np.argmin(lens)), lens.index(60000)):
        alone = torch.stack(m(torch.from_numpy(utts[i][None]).to(DEV)), 0)[:, 0].cpu().numpy()
        worst = min(si_sdr(outs[i][s], alone[s]) for s in range(2))
        print(f"sweep, bf16: utterance of {lens[i] / 8000:g} s in its batch vs alone: {worst:.1f} dB")
        assert worst >= 35.0
