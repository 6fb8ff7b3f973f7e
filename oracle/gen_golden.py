"""Generate the golden fixtures under tests/golden/ by running the REFERENCE
model (imported from /root/reference, build container only).

TEST INFRASTRUCTURE ONLY; never imported by the product or by any GPU run.
Run:  python oracle/gen_golden.py [--full] [--out tests/golden]

Two modules the reference imports are not installed here and are provided as
in-memory stand-ins before the import:
  * ``torchinfo``  -- imported, never called (mossformer2_block.py:10).
  * ``rotary_embedding_torch`` -- ``RotaryEmbedding(dim)`` restated from the
    published algorithm of release 0.8.3 (interleaved pairs, theta 10000).
    Because this stand-in is ours, fixtures do NOT pin the rotary arithmetic
    ("rotary parity unpinned"); they pin everything else.
``librosa``/``torchaudio``/``pesq``/``joblib`` are stubbed as empty modules so
that ``utils/decode.py`` imports (it uses none of them on the live path).
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"


def install_shims():
    ti = types.ModuleType("torchinfo")
    ti.summary = lambda *a, **k: None
    sys.modules["torchinfo"] = ti

    rot = types.ModuleType("rotary_embedding_torch")

    class RotaryEmbedding(nn.Module):
        def __init__(self, dim, theta=10000):
            super().__init__()
            f = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
            self.freqs = nn.Parameter(f, requires_grad=False)

        def rotate_queries_or_keys(self, t, seq_dim=-2):
            n = t.shape[seq_dim]
            pos = torch.arange(n, device=t.device, dtype=t.dtype)
            ang = torch.einsum("i,j->ij", pos, self.freqs).repeat_interleave(2, dim=-1)
            rd = ang.shape[-1]
            mid, rest = t[..., :rd], t[..., rd:]
            pair = mid.reshape(*mid.shape[:-1], rd // 2, 2)
            half = torch.stack((-pair[..., 1], pair[..., 0]), dim=-1).reshape(mid.shape)
            return torch.cat((mid * ang.cos() + half * ang.sin(), rest), dim=-1).type(t.dtype)

    rot.RotaryEmbedding = RotaryEmbedding
    sys.modules["rotary_embedding_torch"] = rot
    for name in ("librosa", "torchaudio", "pesq", "joblib"):
        m = types.ModuleType(name)
        if name == "pesq":
            m.pesq = None
        if name == "joblib":
            m.Parallel = m.delayed = None
        sys.modules[name] = m
    sys.path.insert(0, REF)


class Args:
    def __init__(self, cfg, one_time_decode_length=60):
        self.network = "MossFormer2_SS_8K"
        self.sampling_rate = 8000
        self.decode_window = 1
        self.one_time_decode_length = one_time_decode_length
        self.num_spks = cfg.num_spks
        self.encoder_kernel_size = cfg.enc_kernel
        self.encoder_embedding_dim = cfg.dim
        self.mossformer_sequence_dim = cfg.dim
        self.num_mossformer_layer = cfg.num_layers


def build_reference(cfg, base_seed=0):
    from networks import network_wrapper  # reference
    from speech_separation_amd.schema import state_dict_schema
    from speech_separation_amd.synth import make_state_dict
    model = network_wrapper(Args(cfg)).ss_network
    if getattr(cfg, "recurrent", "tcn") == "fsmn":
        # The upstream architecture: the reference's own Gated_FSMN_Block_Dilated (mossformer2_block.py:430-463) in the
        # list that MossformerBlock_GFSMN.forward (:652-656) interleaves with the FLASH layers.  The reference revision
        # fills that list with GatedTCNBlock; both classes map [B, S, dim] -> [B, S, dim], so the forward is unchanged.
        from models.mossformer2.mossformer2_block import Gated_FSMN_Block_Dilated
        blk = model.mask_net.mdl.intra_mdl.mossformerM
        blk.fsmn = nn.ModuleList([Gated_FSMN_Block_Dilated(cfg.dim, inner_channels=cfg.fsmn_inner)
                                  for _ in range(cfg.num_layers)])
    ref_sd = model.state_dict()
    schema = state_dict_schema(cfg)
    assert list(ref_sd.keys()) == list(schema.keys()), "state_dict key order/name mismatch"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(schema[k]), (k, v.shape, schema[k])
    sd = make_state_dict(cfg, base_seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    return model, sd


def run_with_taps(model, x):
    taps = {}
    hs = []
    mn = model.mask_net
    blk = mn.mdl.intra_mdl.mossformerM
    hs.append(model.enc.register_forward_hook(lambda m, i, o: taps.__setitem__("enc", o.transpose(1, 2))))
    hs.append(mn.mdl.register_forward_pre_hook(lambda m, i: taps.__setitem__("x0", i[0].transpose(1, 2))))
    hs.append(blk.layers[0].register_forward_hook(lambda m, i, o: taps.__setitem__("flash0", o)))
    hs.append(blk.fsmn[0].register_forward_hook(lambda m, i, o: taps.__setitem__("tcn0", o)))
    hs.append(mn.mdl.register_forward_hook(lambda m, i, o: taps.__setitem__("mdl", o.transpose(1, 2))))

    def mask_hook(m, i, o):
        for s in range(o.shape[0]):
            taps[f"mask{s}"] = o[s].transpose(1, 2)
    hs.append(mn.register_forward_hook(mask_hook))
    with torch.no_grad():
        out = model(x)
    for h in hs:
        h.remove()
    return out, {k: v.contiguous().numpy() for k, v in taps.items()}


def write_fsmn_fixture(out_dir):
    # ---- FSMN-family block (dormant in the live model; mossformer2_block.py:434-463) ----
    from models.mossformer2.mossformer2_block import Gated_FSMN_Block_Dilated
    blk = Gated_FSMN_Block_Dilated(64, inner_channels=64).eval()
    r = np.random.default_rng(11)
    fs = {}
    with torch.no_grad():
        for k, v in blk.state_dict().items():
            shape = tuple(v.shape)
            if ("norm" in k or ".mdl.0." in k) and k.endswith("weight"):
                w = 1.0 + 0.1 * r.standard_normal(shape)
            elif "prelu" in k or k == "conv1.1.weight":
                w = 0.25 + 0.05 * r.standard_normal(shape)
            elif k.endswith("bias"):
                w = 0.05 * r.standard_normal(shape)
            else:
                fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
                w = r.uniform(-1, 1, shape) / np.sqrt(fan_in)
            v.copy_(torch.from_numpy(w.astype(np.float32)))
            fs["w:" + k] = v.numpy().copy()
        for name, (B, S) in (("a", (2, 301)), ("b", (1, 1000))):
            x = torch.from_numpy(r.standard_normal((B, S, 64)).astype(np.float32))
            fs["in_" + name] = x.numpy()
            fs["out_" + name] = blk(x).numpy()
    f = os.path.join(out_dir, "fsmn_block.npz")
    np.savez_compressed(f, **fs)
    print("wrote", f, {k: v.shape for k, v in fs.items() if not k.startswith("w:")})



def write_fsmn_model_fixtures(out_dir):
    """Whole forward with Gated_FSMN_Block_Dilated as the recurrent block (see build_reference)."""
    from speech_separation_amd.config import TINY_FSMN, SMALL_FSMN
    from speech_separation_amd.synth import synth_batch
    for name, cfg, B, T, want_taps in (("tinyfsmn", TINY_FSMN, 2, 8000, True), ("tinyfsmn", TINY_FSMN, 1, 32003, False),
                                       ("smallfsmn", SMALL_FSMN, 1, 16000, False)):
        model = build_reference(cfg)[0]
        x = synth_batch(B, T, seed0=0)
        out, taps = run_with_taps(model, torch.from_numpy(x))
        d = {"input": x, "out": np.stack([o.numpy() for o in out], 0)}
        if want_taps:
            d.update({f"tap_{k}": v for k, v in taps.items()})
        f = os.path.join(out_dir, f"fwd_{name}_B{B}_T{T}.npz")
        np.savez_compressed(f, **d)
        print("wrote", f, {k: v.shape for k, v in d.items()}, "out rms", float(np.sqrt((d["out"] ** 2).mean())), flush=True)


def write_longform_fixtures(out_dir, names):
    """One 60-second mixture after the decode padding rule (480000 -> 492000 samples, 61 499 frames,
    241 groups; reference: utils/decode.py:43-51) through the reference model: the long-sequence
    path of the linear-attention sum (mossformer2_block.py:318-322) and of every global GroupNorm
    (:521).  The input is regenerated from its seed by the tests; the outputs are stored every
    `stride`-th sample (a frame spans 16 samples, so every frame is still observed)."""
    from speech_separation_amd.config import TINY, SMALL, FULL_8K
    from speech_separation_amd.synth import synth_batch
    T = 492000
    for name in names:
        cfg, stride = {"tiny": (TINY, 4), "small": (SMALL, 4), "full": (FULL_8K, 8)}[name]
        model = build_reference(cfg)[0]
        x = synth_batch(1, T, seed0=3)
        with torch.no_grad():
            out = model(torch.from_numpy(x))
        o = np.stack([v.numpy() for v in out], 0)
        assert o.shape == (cfg.num_spks, 1, T)
        d = {"seed0": np.int64(3), "T": np.int64(T), "stride": np.int64(stride), "out_strided": o[:, :, ::stride].copy(),
             "out_peak": np.float32(np.abs(o).max()), "out_rms": np.float32(np.sqrt((o.astype(np.float64) ** 2).mean())),
             "input_head": x[:, :64].copy()}
        f = os.path.join(out_dir, f"long_{name}_B1_T{T}.npz")
        np.savez_compressed(f, **d)
        print("wrote", f, d["out_strided"].shape, "peak", float(d["out_peak"]), "rms", float(d["out_rms"]), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--full", action="store_true", help="also run the 24-layer 512-dim model (minutes)")
    ap.add_argument("--only-fsmn", action="store_true", help="write only the FSMN-block fixture")
    ap.add_argument("--longform", default="", help="write only the 60-second fixtures of these configurations "
                                                   "(comma separated: tiny,small,full; full takes tens of minutes)")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--fsmn-model", action="store_true", help="write only the whole-model fixtures of the FSMN form")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    install_shims()
    torch.set_num_threads(a.threads)
    if a.longform:
        write_longform_fixtures(a.out, a.longform.split(","))
        return
    if a.fsmn_model:
        write_fsmn_model_fixtures(a.out)
        return
    if a.only_fsmn:
        sys.path.insert(0, REF)
        write_fsmn_fixture(a.out)
        return
    from speech_separation_amd.config import TINY, SMALL, FULL_8K
    from speech_separation_amd.synth import synth_batch

    # ---- forward fixtures --------------------------------------------------
    cases = [("tiny", TINY, 1, 4000, True), ("tiny", TINY, 1, 8000, False), ("tiny", TINY, 3, 8000, False),
             ("tiny", TINY, 1, 32000, False), ("tiny", TINY, 1, 32003, False), ("tiny", TINY, 2, 38800, False),
             ("small", SMALL, 2, 16000, False)]
    if a.full:
        cases.append(("full", FULL_8K, 1, 32000, False))
    models = {}
    for name, cfg, B, T, want_taps in cases:
        if name not in models:
            models[name] = build_reference(cfg)[0]
        model = models[name]
        x = synth_batch(B, T, seed0=0)
        out, taps = run_with_taps(model, torch.from_numpy(x))
        d = {"input": x, "out": np.stack([o.numpy() for o in out], 0)}
        assert d["out"].shape == (cfg.num_spks, B, T), d["out"].shape
        if want_taps:
            d.update({f"tap_{k}": v for k, v in taps.items()})
        if name == "full":
            m64 = models[name].double()
            with torch.no_grad():
                o64 = m64(torch.from_numpy(x).double())
            d["out_f64"] = np.stack([o.numpy() for o in o64], 0).astype(np.float32)
            models[name].float()
        f = os.path.join(a.out, f"fwd_{name}_B{B}_T{T}.npz")
        np.savez_compressed(f, **d)
        print("wrote", f, {k: v.shape for k, v in d.items()}, "out rms", float(np.sqrt((d["out"] ** 2).mean())),
              flush=True)

    # ---- attention-term balance (quadratic and linear both alive) -----------
    # measured on layer 0 of the tiny model via the reference module itself
    model = models["tiny"]
    lay = model.mask_net.mdl.intra_mdl.mossformerM.layers[0]
    rec = {}
    orig = lay.cal_attention

    def spy(x, qq, lq, qk, lk, v, u, mask=None):
        av, au = orig(x, qq, lq, qk, lk, v, u, mask)
        av0, _ = orig(x, qq, lq * 0, qk, lk * 0, v, u, mask)
        rec["tot"] = float(av.pow(2).mean().sqrt())
        rec["quad"] = float(av0.pow(2).mean().sqrt())
        rec["lin"] = float((av - av0).pow(2).mean().sqrt())
        return av, au
    lay.cal_attention = spy
    with torch.no_grad():
        model(torch.from_numpy(synth_batch(1, 8000)))
    lay.cal_attention = orig
    print("attention balance (rms): ", rec)
    assert rec["quad"] >= 0.1 * rec["tot"] and rec["lin"] >= 0.1 * rec["tot"], rec

    # ---- decode harness fixtures (utils/decode.py) --------------------------
    from utils.decode import decode_one_audio  # reference
    dec = {}
    for T in (4000, 8000, 10000, 28400, 32000):
        x = synth_batch(1, T, seed0=7)
        with torch.no_grad():
            o = decode_one_audio(model, torch.device("cpu"), x, Args(TINY))
        dec[f"in_{T}"] = x
        dec[f"out_{T}"] = np.stack(o, 0).astype(np.float32)
    # segmented path, forced by a 2-second one-pass limit
    x = synth_batch(1, 20001, seed0=9)
    with torch.no_grad():
        o = decode_one_audio(model, torch.device("cpu"), x, Args(TINY, one_time_decode_length=2))
    dec["in_seg_20001"] = x
    dec["out_seg_20001"] = np.stack(o, 0).astype(np.float32)
    f = os.path.join(a.out, "decode_tiny.npz")
    np.savez_compressed(f, **dec)
    print("wrote", f, {k: v.shape for k, v in dec.items()})

    write_fsmn_fixture(a.out)

    # ---- SI-SNR known answers (losses/loss.py) ------------------------------
    from losses.loss import cal_si_snr  # reference; imports only torch/numpy
    r = np.random.default_rng(3)
    src = r.standard_normal((2000, 3, 2)).astype(np.float32)
    est = (src[:, :, ::-1] * 0.7 + 0.3 * r.standard_normal((2000, 3, 2))).astype(np.float32)
    val = -cal_si_snr(torch.from_numpy(src.copy()), torch.from_numpy(est.copy()))[0].numpy()
    xs = torch.Tensor([[1, 0], [123, 45], [34, 5], [2312, 421]])
    xh = xs[:, (1, 0)]
    doc = -cal_si_snr(xs.unsqueeze(-1).repeat(1, 1, 2), xh.unsqueeze(1).repeat(1, 2, 1))[0].numpy()
    f = os.path.join(a.out, "si_snr.npz")
    np.savez_compressed(f, src=src, est=est, si_snr=val, doc_example=doc)
    print("wrote", f, "docstring example:", doc)


if __name__ == "__main__":
    main()
