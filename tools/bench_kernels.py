"""Per-operator timing at the benchmark shape (B x 4 s, dim 512) through the
C ABI, with HIP events on the launch stream.  Development aid; prints one line
per operator with the algorithmic rate."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from speech_separation_amd import _lib, ops  # noqa: E402
from speech_separation_amd.config import FULL_8K, MossFormer2Config  # noqa: E402
from speech_separation_amd.packing import PackedWeights  # noqa: E402
from speech_separation_amd.synth import make_state_dict  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--samples", type=int, default=32000)
    ap.add_argument("--precision", type=int, default=0)
    a = ap.parse_args()
    dev = "cuda:0"
    cfg = MossFormer2Config(num_layers=1)
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()}
    pk = PackedWeights(sd, cfg, dev, a.precision)
    lib = _lib.load()
    B, S, C = a.batch, lib.msf2_num_frames(a.samples), cfg.dim
    M = B * S
    E, NH = 2 * C, 4 * C + 128
    prec = a.precision
    lw = pk.layers[0]
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, S, C, device=dev, generator=g)
    t = torch.randn(B, S, C, device=dev, generator=g)
    bstat = torch.tensor([[0.1, 1.3]] * B, device=dev)
    rows = []

    ms = timeit(lambda: ops.tcn_pointwise(x, t, bstat, lw.tcn[0], prec))
    rows.append(("tcn_pointwise", ms, 2.0 * M * 2 * C * C / ms / 1e9, "TF/s"))

    Whq = torch.randn(NH, C, device=dev, generator=g) / C ** 0.5
    bhq = torch.randn(NH, device=dev, generator=g)
    ms = timeit(lambda: ops.linear(x, Whq, bhq, shift=True, scale_norm=True, act=1, precision=prec))
    rows.append(("linear hq (+rowstat)", ms, 2.0 * M * C * NH / ms / 1e9, "TF/s"))

    o = torch.randn(B, S, E, device=dev, generator=g)
    Wo = torch.randn(C, E, device=dev, generator=g) / E ** 0.5
    ms = timeit(lambda: ops.linear(o, Wo, bhq[:C].contiguous(), scale_norm=True, act=1, precision=prec))
    rows.append(("linear to_out (+rowstat)", ms, 2.0 * M * E * C / ms / 1e9, "TF/s"))

    hu = torch.randn(B, S, NH, device=dev, generator=g)
    gam = 2 + 0.1 * torch.randn(4, 128, device=dev, generator=g)
    bet = 0.1 * torch.randn(4, 128, device=dev, generator=g)
    fr = sd["mask_net.mdl.intra_mdl.mossformerM.layers.0.rotary_pos_emb.freqs"].to(dev)
    ms = timeit(lambda: ops.flash_attention(hu, gam, bet, fr, prec))
    Sp = (S + 255) // 256 * 256
    fl = 2.0 * B * (Sp * (128 * 256 + 256 * 2 * E) + S * 2 * (128 * 2 * E))
    rows.append(("flash_attention (5 kernels)", ms, fl / ms / 1e9, "TF/s"))

    dw = torch.randn(17, NH, device=dev, generator=g)
    y = torch.empty_like(hu)
    ms = timeit(lambda: ops.dwconv17(hu, dw, out=y))
    rows.append(("dwconv17 C=2176", ms, 2.0 * M * NH * 4 / ms / 1e6, "GB/s"))

    xx = x.clone()
    ms = timeit(lambda: ops.gated_tcn(xx, lw, prec), reps=3, warm=1)
    rows.append(("gated_tcn block (28 kernels)", ms, 2.0 * M * 19 * C * C / ms / 1e9, "TF/s"))

    for name, ms, rate, unit in rows:
        print(f"{name:32s} {ms:9.3f} ms  {rate:9.1f} {unit}")


if __name__ == "__main__":
    main()
