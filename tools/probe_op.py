"""Launch ONE operator a few times (for rocprofv3 --pmc / --kernel-trace runs).

    python tools/probe_op.py tcn_pointwise|linear_hq|attention|dwconv|stage_a [--batch 16] [--reps 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from speech_separation_amd import _lib, ops  # noqa: E402
from speech_separation_amd.config import MossFormer2Config  # noqa: E402
from speech_separation_amd.packing import PackedWeights  # noqa: E402
from speech_separation_amd.synth import make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("op")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--samples", type=int, default=32000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--precision", type=int, default=0)
    a = ap.parse_args()
    dev = "cuda:0"
    cfg = MossFormer2Config(num_layers=1)
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()}
    pk = PackedWeights(sd, cfg, dev, a.precision)
    lib = _lib.load()
    B, S, C = a.batch, lib.msf2_num_frames(a.samples), cfg.dim
    NH = 4 * C + 128
    lw = pk.layers[0]
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, S, C, device=dev, generator=g)
    t = torch.randn(B, S, C, device=dev, generator=g)
    bstat = torch.tensor([[0.1, 1.3]] * B, device=dev)
    if a.op == "tcn_pointwise":
        fn = lambda: ops.tcn_pointwise(x, t, bstat, lw.tcn[0], a.precision)
    elif a.op == "linear_hq":
        W = torch.randn(NH, C, device=dev, generator=g) / C ** 0.5
        b = torch.randn(NH, device=dev, generator=g)
        fn = lambda: ops.linear(x, W, b, shift=True, scale_norm=True, act=1, precision=a.precision)
    elif a.op == "attention":
        hu = torch.randn(B, S, NH, device=dev, generator=g)
        gam = 2 + 0.1 * torch.randn(4, 128, device=dev, generator=g)
        bet = 0.1 * torch.randn(4, 128, device=dev, generator=g)
        fr = sd["mask_net.mdl.intra_mdl.mossformerM.layers.0.rotary_pos_emb.freqs"].to(dev)
        fn = lambda: ops.flash_attention(hu, gam, bet, fr, a.precision)
    elif a.op == "dwconv":
        hu = torch.randn(B, S, NH, device=dev, generator=g)
        dw = torch.randn(17, NH, device=dev, generator=g)
        y = torch.empty_like(hu)
        fn = lambda: ops.dwconv17(hu, dw, out=y)
    elif a.op == "gated_tcn":
        fn = lambda: ops.gated_tcn(x, lw, a.precision)
    else:
        raise SystemExit("unknown op")
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    print("done", a.op)


if __name__ == "__main__":
    main()
