"""File-to-file separation driver: the loop of the reference's inference.py (:21-41) on top of
this package (model construction, checkpoint reload, decode harness, wav I/O).

    python -m speech_separation_amd.infer --input-path DIR|FILE|LIST --output-dir OUT \
        [--checkpoint-dir CKPT] [--precision fp32|bf16] [--one-time-decode-length 60] [--decode-window 1]

Output naming follows the reference: <name>.wav -> <name>_s1.wav, <name>_s2.wav, trimmed to the
input length, 16-bit PCM at the model's sampling rate.  Without a checkpoint directory the seeded
synthetic weights are used (the reference revision ships no trained checkpoint).
"""
import argparse
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .checkpoint import reload_for_eval
from .decode import decode_one_audio
from .model import MossFormer2_SS
from .wavio import audioread, find_inputs, write_wav


def build_args(ns):
    return SimpleNamespace(network="MossFormer2_SS_8K", sampling_rate=8000, num_spks=2, encoder_kernel_size=16,
                           encoder_embedding_dim=512, mossformer_sequence_dim=512, num_mossformer_layer=24,
                           one_time_decode_length=ns.one_time_decode_length, decode_window=ns.decode_window,
                           msf2_precision=_lib.PREC_BF16 if ns.precision == "bf16" else _lib.PREC_FP32,
                           checkpoint_dir=ns.checkpoint_dir, use_cuda=1)


def run(ns):
    args = build_args(ns)
    device = torch.device("cuda")
    model = MossFormer2_SS(args).model
    model.to(device)
    if ns.checkpoint_dir:
        reload_for_eval(model, ns.checkpoint_dir, 1)       # otherwise the module keeps its seeded initial weights
    model.eval()
    os.makedirs(ns.output_dir, exist_ok=True)
    files = find_inputs(ns.input_path)
    print("Decoding...")
    written = []
    with torch.no_grad():
        for path in files:
            wav_id = path.split("/")[-1]
            print(f"audio: {wav_id}")
            data = audioread(path, args.sampling_rate).astype(np.float32)
            outs = decode_one_audio(model, device, np.reshape(data, [1, data.shape[0]]), args)
            for spk in range(args.num_spks):
                out_path = os.path.join(ns.output_dir, wav_id.replace(".wav", "_s" + str(spk + 1) + ".wav"))
                write_wav(out_path, outs[spk][:data.shape[0]], args.sampling_rate)
                written.append(out_path)
    print("Done!")
    return written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input-path", required=True)
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--checkpoint-dir", default=None)
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32")
    ap.add_argument("--one-time-decode-length", type=int, default=60)
    ap.add_argument("--decode-window", type=int, default=1)
    run(ap.parse_args())


if __name__ == "__main__":
    main()
