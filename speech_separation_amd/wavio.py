"""WAV file input and output for the separation drivers, without soundfile / librosa.

Mirrors what the reference does around its model:
  * reading (reference: dataloader/dataloader.py:21-27): samples as float64 in [-1, 1)
    (integer PCM divided by 2^(bits-1), which is what soundfile returns), resampled to the
    model's rate when the file's differs, first channel only;
  * writing (reference: inference.py:36-41, ``sf.write(path, float_array, rate)``): 16-bit PCM,
    sample = clip(rint(x * 32768), -32768, 32767).  The files the reference ships under
    outputs/*/output/ pin that rule: its peak-normalised waveforms contain both 32767 (x = +1.0,
    clipped) and -32768 (x = -1.0), which a 32767 scale could not produce.
Resampling parity is unpinned: the reference calls librosa.resample (soxr_hq), which is not
installed here; this module uses a polyphase Kaiser resampler (scipy.signal.resample_poly).  Every
file the reference ships is already 8 kHz, so its own data never exercises that path.
"""
import os
import struct
from math import gcd

import numpy as np


def _chunks(buf):
    pos = 12
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        yield cid, buf[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)


def read_wav(path):
    """(samples float64 [frames] or [frames, channels], sample_rate).  PCM 8/16/24/32-bit and
    IEEE float 32/64-bit RIFF files (format tags 1, 3 and their WAVE_FORMAT_EXTENSIBLE forms)."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    fmt = data = None
    for cid, body in _chunks(buf):
        if cid == b"fmt " and fmt is None:
            fmt = body
        elif cid == b"data" and data is None:
            data = body
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, nch, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:
        tag = struct.unpack("<H", fmt[24:26])[0]
    width = bits // 8
    n = len(data) // (width * nch) * nch
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(data, dtype=np.uint8, count=n).astype(np.float64) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(data, dtype="<i2", count=n).astype(np.float64) / 32768.0
        elif bits == 24:
            b = np.frombuffer(data, dtype=np.uint8, count=n * 3).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float64) / float(1 << 23)
        elif bits == 32:
            x = np.frombuffer(data, dtype="<i4", count=n).astype(np.float64) / float(1 << 31)
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3 and bits in (32, 64):
        x = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8", count=n).astype(np.float64)
    else:
        raise ValueError(f"{path}: unsupported WAV format tag {tag} / {bits} bits")
    if nch > 1:
        x = x.reshape(-1, nch)
    return x, int(rate)


def resample(x, orig_sr, target_sr):
    if orig_sr == target_sr:
        return x
    from scipy.signal import resample_poly
    g = gcd(int(orig_sr), int(target_sr))
    return resample_poly(x, int(target_sr) // g, int(orig_sr) // g, axis=0)


def audioread(path, sampling_rate):
    """Reference: dataloader/dataloader.py:21-27 (read, resample if needed, keep channel 0)."""
    data, fs = read_wav(path)
    if fs != sampling_rate:
        data = resample(data, fs, sampling_rate)
    if data.ndim > 1:
        data = data[:, 0]
    return data


def write_wav(path, samples, sampling_rate):
    """16-bit PCM mono/multi-channel, the subtype soundfile picks for a .wav path."""
    x = np.asarray(samples, dtype=np.float64)
    pcm = np.clip(np.rint(x * 32768.0), -32768, 32767).astype("<i2")
    nch = 1 if pcm.ndim == 1 else pcm.shape[1]
    body = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, nch, int(sampling_rate), int(sampling_rate) * nch * 2, nch * 2, 16) + b"data" + struct.pack("<I", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)


def find_inputs(input_path):
    """Reference: dataloader/misc.py:15-31 (decode=True): a directory of .wav files (sorted,
    recursive, as librosa.util.find_files does), a single .wav, or a list file whose first
    column holds paths.  (.flac inputs need a decoder this build does not have.)"""
    if os.path.isdir(input_path):
        found = []
        for root, _, files in os.walk(input_path):
            found += [os.path.join(root, f) for f in files if f.lower().endswith(".wav")]
        return sorted(found)
    if input_path.lower().endswith(".wav"):
        return [input_path]
    if input_path.lower().endswith(".flac"):
        raise ValueError("flac input is not supported")
    with open(input_path) as fid:
        return [line.strip().split()[0] for line in fid if line.strip()]
