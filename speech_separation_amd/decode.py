"""Decode harness with the reference's contract
(reference: utils/decode.py:18-83): ``decode_one_audio(model, device, inputs,
args) -> list of num_spks numpy arrays``, zero-padded to the reference's
window grid, optionally decoded in 1-second windows when the input is longer
than ``one_time_decode_length`` seconds, and peak-normalised over speakers.

Differences in mechanism, not in results:
  * the windows of the segmented path are independent utterances, so they go
    through the model as ONE batch (81 windows per minute of audio) instead of
    a Python loop of batch-1 calls;
  * stitching and the peak search run on the device; one copy brings the
    result to the host.
"""
import numpy as np
import torch

SUPPORTED = ("MossFormer2_SS_16K", "MossFormer2_SS_8K")


def padded_length(t: int, window: int, stride: int) -> int:
    """Length the reference pads an input of t samples to (utils/decode.py:43-51).
    Note the third rule adds ``t - floor((t-window)/stride)*stride`` samples,
    i.e. window + remainder, not the complement of the remainder."""
    if t < window:
        return window
    if t < window + stride:
        return window + stride
    rem = (t - window) % stride
    if rem != 0:
        return t + (t - (t - window) // stride * stride)
    return t


def decode_one_audio(model, device, inputs, args):
    if args.network not in SUPPORTED:
        print(f"in decode, {args.network} is not found!")
        return None
    return decode_one_audio_mossformer2_ss(model, device, inputs, args)


def decode_one_audio_mossformer2_ss(model, device, inputs, args, max_windows_per_call: int = 128):
    num_spks = int(args.num_spks)
    window = int(args.sampling_rate * args.decode_window)
    stride = int(window * 0.75)
    if torch.is_tensor(inputs):
        inputs = inputs.detach().cpu().numpy()
    inputs = np.asarray(inputs)
    b, t = inputs.shape
    segmented = t > window * args.one_time_decode_length
    if segmented:
        print("The sequence is longer than {} seconds, using segmentation decoding...".format(
            args.one_time_decode_length))
    tp = padded_length(t, window, stride)
    x = torch.zeros((b, tp), dtype=torch.float32)
    x[:, :t] = torch.from_numpy(np.float32(inputs))
    x = x.to(device)

    if not segmented:
        out_list = model(x)
        est = torch.stack([o[0, :] for o in out_list], 0)          # [spk, tp]
        peak = est.abs().max()
        return [a for a in (est / peak).cpu().numpy()]

    # segmented: windows at 0, stride, 2*stride, ... while they fit
    give = (window - stride) // 2
    starts = list(range(0, tp - window + 1, stride))
    wins = torch.stack([x[0, s:s + window] for s in starts], 0)   # only utterance 0 is decoded, as in the reference
    outs = []
    for i in range(0, len(starts), max_windows_per_call):
        o = model(wins[i:i + max_windows_per_call])
        outs.append(torch.stack(o, 0))                             # [spk, n, window]
    o = torch.cat(outs, 1).to(torch.float64)                      # [spk, windows, window]
    # Stitching as ONE gather: output sample t comes from window n(t) at offset t - n(t) * stride, where window 0 owns
    # [0, window - give) and window n > 0 owns [n * stride + give, n * stride + window - give); samples past the last
    # window's share stay zero (reference: utils/decode.py:56-71).
    t = torch.arange(tp, device=o.device)
    n = torch.clamp((t - give) // stride, min=0, max=len(starts) - 1)
    off = t - n * stride
    covered = off < window - give
    flat = n * window + torch.where(covered, off, torch.zeros_like(off))
    res = o.reshape(num_spks, -1).index_select(1, flat) * covered.to(torch.float64)
    peak = res.abs().max()
    return [a for a in (res / peak).cpu().numpy()]
