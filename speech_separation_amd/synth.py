"""Deterministic synthetic weights and Libri2Mix-shaped synthetic mixtures.

No trained checkpoint ships with the reference (its .gitignore excludes
checkpoints), so parity is established on weights every side can regenerate
from the key name alone: one numpy Generator per key, seeded with
crc32(key) ^ base_seed.  The distributions are chosen so that every branch of
the forward pass carries signal: in particular the OffsetScale gains are
centred on 2 (the reference initialises them ~N(0, 0.02), reference:
models/mossformer2/mossformer2_block.py:71-73, which makes the quadratic
attention term vanish numerically and would leave it untested).
"""
import zlib
from collections import OrderedDict

import numpy as np

from .config import MossFormer2Config
from .schema import state_dict_schema


def _rng(key: str, base_seed: int) -> np.random.Generator:
    return np.random.default_rng((zlib.crc32(key.encode()) ^ base_seed) & 0xFFFFFFFF)


def _is_norm_gain(key: str) -> bool:
    return (key.endswith(".norm.weight") or key.endswith("intra_norm.weight")
            or key.endswith(".net.2.weight") or key.endswith(".g")
            or key.endswith("pos_enc.scale")
            # FSMN-family block: CLayerNorm / InstanceNorm gains and the LayerNorm gain of its FFConvM branches
            or key.endswith(".norm1.weight") or key.endswith(".norm2.weight") or key.endswith(".mdl.0.weight"))


def make_state_dict(cfg: MossFormer2Config, base_seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """float32 numpy arrays for every key of the schema."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in state_dict_schema(cfg).items():
        r = _rng(key, base_seed)
        if key.endswith("pos_enc.inv_freq"):
            # constructor value (reference: mossformer2_block.py:58)
            a = 1.0 / (10000.0 ** (np.arange(0, cfg.dim, 2, dtype=np.float32) / np.float32(cfg.dim)))
        elif key.endswith("rotary_pos_emb.freqs"):
            a = 1.0 / (10000.0 ** (np.arange(0, cfg.rot_dim, 2, dtype=np.float32)[: cfg.rot_dim // 2]
                                   / np.float32(cfg.rot_dim)))
        elif key.endswith("qk_offset_scale.gamma"):
            a = 2.0 + 0.1 * r.standard_normal(shape)
        elif key.endswith("qk_offset_scale.beta"):
            a = 0.1 * r.standard_normal(shape)
        elif key.endswith("prelu.weight"):
            a = np.full(shape, 0.25)
        elif key.endswith(".conv1.1.weight") or key.endswith(".prelu1.weight") or key.endswith(".prelu2.weight"):
            a = 0.25 + 0.05 * r.standard_normal(shape)      # PReLU slopes of the FSMN-family block
        elif _is_norm_gain(key):
            a = 1.0 + 0.1 * r.standard_normal(shape)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            b = 1.0 / np.sqrt(fan_in)
            a = r.uniform(-b, b, size=shape)
        else:  # biases and norm offsets
            a = 0.02 * r.standard_normal(shape)
        out[key] = np.ascontiguousarray(a, dtype=np.float32).reshape(shape)
    return out


def synth_source(n: int, seed: int, sr: int = 8000) -> np.ndarray:
    """One speech-like source: white noise through a random 2-pole resonator,
    with a ~4 Hz syllabic amplitude envelope, RMS-normalised to -25 dBFS."""
    r = np.random.default_rng(seed)
    x = r.standard_normal(n + 64)
    f0 = r.uniform(200.0, 1800.0)
    rad = r.uniform(0.90, 0.98)
    a1, a2 = -2.0 * rad * np.cos(2 * np.pi * f0 / sr), rad * rad
    try:
        from scipy.signal import lfilter
        y = lfilter([1.0], [1.0, a1, a2], x)
    except Exception:  # pragma: no cover - scipy is present in the image
        y = np.zeros_like(x)
        for i in range(2, len(x)):
            y[i] = x[i] - a1 * y[i - 1] - a2 * y[i - 2]
    y = y[64:]
    t = np.arange(n) / sr
    env = 0.55 + 0.45 * np.sin(2 * np.pi * r.uniform(3.0, 5.0) * t + r.uniform(0, 2 * np.pi))
    y = y * env
    rms = np.sqrt(np.mean(y * y)) + 1e-12
    return (y * (10 ** (-25 / 20) / rms)).astype(np.float32)


def synth_mixture(n: int, seed: int):
    """(mixture, s1, s2), float32, mixture clipped to [-1, 1]."""
    s1 = synth_source(n, 2 * seed + 1)
    s2 = synth_source(n, 2 * seed + 2)
    return np.clip(s1 + s2, -1.0, 1.0).astype(np.float32), s1, s2


def synth_batch(B: int, n: int, seed0: int = 0) -> np.ndarray:
    return np.stack([synth_mixture(n, seed0 + i)[0] for i in range(B)], 0)


def si_sdr(est: np.ndarray, ref: np.ndarray) -> float:
    """Scale-invariant SDR in dB (reference: utils/si_sdr_numpy.py:27-32)."""
    est = est.astype(np.float64) - np.mean(est)
    ref = ref.astype(np.float64) - np.mean(ref)
    tgt = np.sum(est * ref) * ref / np.sum(ref * ref)
    noise = est - tgt
    return float(10.0 * np.log10(np.sum(tgt * tgt) / np.sum(noise * noise)))


def permute_si_sdr(e1, e2, c1, c2) -> float:
    """Best-permutation mean SI-SDR of two estimates (reference: utils/si_sdr_numpy.py:35-41)."""
    a = si_sdr(e1, c1) + si_sdr(e2, c2)
    b = si_sdr(e1, c2) + si_sdr(e2, c1)
    return 0.5 * max(a, b)
