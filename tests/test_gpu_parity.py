"""Parity of the HIP path (through the C ABI of libmsf2.so) against the CPU
oracle and against golden fixtures written by the reference model.

Tolerances (fp32 path).  The contract is max-abs <= 1e-3 on the separated
waveforms against the reference CPU forward; the tests hold the HIP path to
tighter, scale-aware bounds: single operators to 2e-5 relative to the tensor's
peak, the tiny/small models to 1e-5 absolute on outputs of rms ~5e-3, the
24-layer model to 1e-3 of the output peak (the level at which the fp32
reference itself differs from an fp64 evaluation of the same weights).
"""
import glob
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpu_ref  # noqa: E402  (checker only)
from speech_separation_amd import _lib, ops  # noqa: E402
from speech_separation_amd.config import TINY, SMALL, FULL_8K, TINY_FSMN, SMALL_FSMN  # noqa: E402
from speech_separation_amd.model import MossFormerHIP  # noqa: E402
from speech_separation_amd.packing import PackedWeights  # noqa: E402
from speech_separation_amd.schema import MDL  # noqa: E402
from speech_separation_amd.synth import make_state_dict, synth_batch  # noqa: E402

BF = _lib.PREC_BF16
CFG = {"tiny": TINY, "small": SMALL, "full": FULL_8K, "tinyfsmn": TINY_FSMN, "smallfsmn": SMALL_FSMN}
DEV = "cuda:0"


def rel_err(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp(min=1e-30))


@pytest.fixture(scope="module")
def models():
    cache = {}

    def get(name):
        if name not in cache:
            m = MossFormerHIP(CFG[name], compute_device=DEV)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(CFG[name], 0).items()})
            cache[name] = m
        return cache[name]
    return get


@pytest.fixture(scope="module")
def small_w():
    return cpu_ref.to_torch(make_state_dict(SMALL, 0))


# ------------------------------------------------------------------ operators
@pytest.mark.parametrize("B,T,C", [(1, 4000, 64), (3, 8003, 128), (2, 16000, 512)])
def test_encoder(B, T, C):
    g = torch.Generator().manual_seed(1)
    mix = torch.randn(B, T, generator=g) * 0.1
    w = torch.randn(C, 1, 16, generator=g) * 0.25
    ref = torch.relu(torch.nn.functional.conv1d(mix[:, None], w, stride=8)).transpose(1, 2)
    enc, stats = ops.encoder(mix.to(DEV), w[:, 0, :].t().contiguous().to(DEV), eps=1e-8)
    assert rel_err(enc, ref) <= 2e-6
    mu = ref.double().mean(dim=(1, 2))
    var = ref.double().var(dim=(1, 2), unbiased=False)
    assert np.allclose(stats[:, 0].cpu().double(), mu, rtol=1e-5, atol=1e-7)
    assert np.allclose(stats[:, 1].cpu().double(), 1 / torch.sqrt(var + 1e-8), rtol=1e-5)


@pytest.mark.parametrize("B,S,C", [(1, 7, 64), (2, 499, 384), (2, 1000, 2176), (1, 33, 512)])
def test_dwconv17(B, S, C):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, S, C, generator=g)
    w = torch.randn(C, 1, 17, generator=g) * 0.2
    ref = x + cpu_ref.dwconv_time(x, w)
    wt = w[:, 0, :].t().contiguous().to(DEV)
    got = ops.dwconv17(x.to(DEV), wt)
    assert rel_err(got, ref) <= 2e-6
    base = torch.randn(B, S, C, generator=g)
    got2 = ops.dwconv17(x.to(DEV), wt, out=base.to(DEV).clone(), accumulate=True)
    assert rel_err(got2, base + ref) <= 2e-6


@pytest.mark.parametrize("B,S,K,N,shift,sn,act", [
    (1, 100, 64, 64, False, False, 0), (2, 499, 64, 384, True, True, 1), (2, 300, 512, 2176, True, True, 1),
    (1, 129, 1024, 512, False, True, 1), (3, 128, 128, 96, False, False, 2)])
def test_linear(B, S, K, N, shift, sn, act):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, S, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    xr = cpu_ref.token_shift(x) if shift else x
    if sn:
        xr = cpu_ref.scale_norm(xr, torch.ones(1))
    ref = xr.double() @ W.double().t() + b.double()
    ref = [lambda t: t, torch.nn.functional.silu, torch.sigmoid][act](ref)
    got = ops.linear(x.to(DEV), W.to(DEV), b.to(DEV), shift=shift, scale_norm=sn, act=act)
    assert rel_err(got, ref) <= 2e-5


@pytest.mark.parametrize("B,S,E", [(1, 256, 128), (2, 499, 128), (1, 1000, 256), (2, 700, 1024)])
def test_flash_attention(B, S, E):
    g = torch.Generator().manual_seed(4)
    v = torch.randn(B, S, E, generator=g)
    u = torch.randn(B, S, E, generator=g)
    qk = torch.randn(B, S, 128, generator=g)
    gam = 2.0 + 0.1 * torch.randn(4, 128, generator=g)
    bet = 0.1 * torch.randn(4, 128, generator=g)
    freqs = torch.from_numpy(make_state_dict(TINY)[f"{MDL}.layers.0.rotary_pos_emb.freqs"])
    ref = cpu_ref.attention_core(v.double(), u.double(), qk.double(), gam.double(), bet.double(), freqs.double())
    hu = torch.cat((v, u, qk), -1).contiguous()
    got = ops.flash_attention(hu.to(DEV), gam.to(DEV), bet.to(DEV), freqs.to(DEV))
    assert rel_err(got, ref) <= 2e-5


def test_gated_tcn(small_w):
    g = torch.Generator().manual_seed(5)
    B, S, C = 2, 777, SMALL.dim
    x = torch.randn(B, S, C, generator=g)
    W64 = {k: v.double() for k, v in small_w.items()}
    ref = cpu_ref.gated_tcn(x.double(), W64, f"{MDL}.fsmn.1")
    pk = PackedWeights(small_w, SMALL, DEV)
    got = ops.gated_tcn(x.to(DEV).clone(), pk.layers[1])
    assert rel_err(got, ref) <= 2e-5


def test_tcn_pointwise(small_w):
    g = torch.Generator().manual_seed(7)
    B, S, C = 2, 300, SMALL.dim
    h = torch.randn(B, S, C, generator=g)
    t = torch.nn.functional.silu(torch.randn(B, S, C, generator=g))
    p = f"{MDL}.fsmn.0.tcn.3"
    W = {k: v.double() for k, v in small_w.items() if k.startswith(p)}
    tn = cpu_ref.group_norm1(t.double(), W[f"{p}.net.2.weight"], W[f"{p}.net.2.bias"], 1e-5)
    net = tn @ W[f"{p}.net.3.weight"][:, :, 0].t() + W[f"{p}.net.3.bias"]
    gate = torch.sigmoid(h.double() @ W[f"{p}.gate.0.weight"][:, :, 0].t() + W[f"{p}.gate.0.bias"])
    ref = h.double() + gate * net
    mu = t.double().mean(dim=(1, 2))
    rstd = 1 / torch.sqrt(t.double().var(dim=(1, 2), unbiased=False) + 1e-5)
    bstat = torch.stack((mu, rstd), 1).float().contiguous()
    pk = PackedWeights(small_w, SMALL, DEV)
    got = ops.tcn_pointwise(h.to(DEV), t.to(DEV), bstat.to(DEV), pk.layers[0].tcn[3])
    assert rel_err(got, ref) <= 2e-5


@pytest.mark.parametrize("B,S,C,T", [(1, 499, 64, 4000), (2, 999, 512, 8003), (1, 100, 128, 700)])
def test_decoder(B, S, C, T):
    g = torch.Generator().manual_seed(6)
    sep = torch.randn(B, S, C, generator=g)
    w = torch.randn(C, 1, 16, generator=g) * 0.1
    est = torch.nn.functional.conv_transpose1d(sep.double().transpose(1, 2), w.double(), stride=8)[:, 0, :]
    ref = torch.zeros(B, T, dtype=torch.float64)
    n = min(T, est.shape[1])
    ref[:, :n] = est[:, :n]
    got = ops.decoder(sep.to(DEV), w[:, 0, :].contiguous().to(DEV), T)
    assert rel_err(got, ref) <= 2e-6


def test_bad_arguments_are_rejected():
    lib = _lib.load()
    cfg = _lib.Config(96, 2, 16, 2, _lib.PREC_FP32)     # dim not supported
    assert lib.msf2_workspace_bytes(cfg, 1, 8000) == 0
    assert lib.msf2_forward(cfg, None, None, None, 1, 8000, None, 0, None, None) == -1
    x = torch.zeros(1, 10, 64, device=DEV)
    with pytest.raises(_lib.Msf2Error):
        ops.linear(x[..., :48].contiguous(), torch.zeros(8, 48, device=DEV))   # K not a multiple of 32
    m = MossFormerHIP(TINY, compute_device=DEV)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 8, device=DEV))


def test_workspace_too_small_is_reported(models):
    lib = _lib.load()
    m = models("tiny")
    pk = m.packed()
    x = torch.zeros(1, 4000, device=DEV)
    out = torch.zeros(2, 1, 4000, device=DEV)
    ws = torch.empty(1024, dtype=torch.uint8, device=DEV)
    rc = lib.msf2_forward(pk.c_cfg, pk.struct, x.data_ptr(), out.data_ptr(), 1, 4000, ws.data_ptr(), 1024, None, None)
    assert rc == -2


# -------------------------------------------------------------- whole forward
def test_forward_taps_against_reference(models, golden_dir):
    z = np.load(os.path.join(golden_dir, "fwd_tiny_B1_T4000.npz"))
    taps = {}
    out = models("tiny")(torch.from_numpy(z["input"]).to(DEV), taps=taps)
    torch.cuda.synchronize()
    for name in ("enc", "x0", "flash0", "tcn0", "mdl", "mask0", "mask1"):
        ref = torch.from_numpy(z[f"tap_{name}"])
        assert rel_err(taps[name], ref) <= 3e-5, name
    got = torch.stack(out, 0).cpu().numpy()
    assert np.abs(got - z["out"]).max() <= 1e-5


def test_forward_matches_reference_fixtures(models, golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "fwd_*.npz")))
    seen = 0
    for f in files:
        name = re.match(r"fwd_(\w+?)_B", os.path.basename(f)).group(1)
        if name == "full":
            continue
        z = np.load(f)
        out = models(name)(torch.from_numpy(z["input"]).to(DEV))
        got = torch.stack(out, 0).cpu().numpy()
        assert got.shape == z["out"].shape
        assert np.abs(got - z["out"]).max() <= 1e-5, f
        seen += 1
    assert seen >= 10


def test_fsmn_form_taps_and_full_width(models, golden_dir):
    """Gated_FSMN_Block_Dilated as the recurrent block (constructor switch recurrent="fsmn"): per-stage taps of
    the tiny configuration against the reference modules composed the upstream way, then the reference's real
    widths (dim 512, inner 256; 2 layers) against the fp64 oracle, fp32 and bf16 mode."""
    from speech_separation_amd.config import MossFormer2Config
    from speech_separation_amd.synth import si_sdr
    z = np.load(os.path.join(golden_dir, "fwd_tinyfsmn_B2_T8000.npz"))
    taps = {}
    out = models("tinyfsmn")(torch.from_numpy(z["input"]).to(DEV), taps=taps)
    for name in ("enc", "x0", "flash0", "tcn0", "mdl", "mask0", "mask1"):
        assert rel_err(taps[name], torch.from_numpy(z[f"tap_{name}"])) <= 3e-5, name
    assert np.abs(torch.stack(out, 0).cpu().numpy() - z["out"]).max() <= 1e-5
    cfg = MossFormer2Config(num_layers=2, recurrent="fsmn")
    sd = make_state_dict(cfg, 0)
    x = torch.from_numpy(synth_batch(2, 16000, seed0=4))
    with torch.no_grad():
        ref = torch.stack(cpu_ref.forward(cpu_ref.to_torch(sd, torch.float64), x.double(), cfg), 0)
    for prec, tol in ((_lib.PREC_FP32, 2e-5), (BF, None)):
        m = MossFormerHIP(cfg, compute_device=DEV, precision=prec)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        got = torch.stack(m(x.to(DEV)), 0).cpu()
        if tol is not None:
            assert rel_err(got, ref) <= tol
        else:       # bf16 operands: the waveforms agree with the fp64 evaluation to better than 30 dB
            fid = [si_sdr(got[s, b].numpy(), ref[s, b].numpy()) for s in range(2) for b in range(2)]
            print("fsmn form, bf16: fidelity to the fp64 oracle", [f"{v:.1f}" for v in fid], "dB")
            assert min(fid) >= 30.0


def test_full_width_taps_against_oracle():
    """Where a difference would come from, at the reference's real widths (dim 512; 2 layers, 2 x 2 s): every tap of
    the forward (encoder, mask-net input, first FLASH block, first gated TCN block, final norms, both masks) against
    the fp64 oracle -- fp32 mode to 3e-5 relative, bf16 mode to better than 35 dB per tap."""
    from speech_separation_amd.config import MossFormer2Config
    cfg = MossFormer2Config(num_layers=2)
    sd = make_state_dict(cfg, 0)
    x = torch.from_numpy(synth_batch(2, 16000, seed0=8))
    ref_taps = {}
    with torch.no_grad():
        ref = torch.stack(cpu_ref.forward(cpu_ref.to_torch(sd, torch.float64), x.double(), cfg, taps=ref_taps), 0)
    names = ("enc", "x0", "flash0", "tcn0", "mdl", "mask0", "mask1")
    for prec in (_lib.PREC_FP32, BF):
        m = MossFormerHIP(cfg, compute_device=DEV, precision=prec)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        taps = {}
        got = torch.stack(m(x.to(DEV), taps=taps), 0).cpu()
        for name in names:
            e = rel_err(taps[name], ref_taps[name])
            print(f"full-width tap {name}, precision {prec}: max-abs error {e:.2e} of the tap's peak")
            if prec == _lib.PREC_FP32:
                assert e <= 3e-5, name
            else:
                assert e <= 10 ** (-35 / 20), (name, e)
        assert rel_err(got, ref) <= (3e-5 if prec == _lib.PREC_FP32 else 10 ** (-30 / 20))


def test_full_model_matches_reference(models, golden_dir):
    z = np.load(os.path.join(golden_dir, "fwd_full_B1_T32000.npz"))
    out = models("full")(torch.from_numpy(z["input"]).to(DEV))
    got = torch.stack(out, 0).cpu().numpy()
    peak = np.abs(z["out"]).max()
    err = np.abs(got - z["out"]).max()
    err64 = np.abs(got - z["out_f64"]).max()
    print(f"full model: max-abs vs reference fp32 {err:.3e}, vs fp64 evaluation {err64:.3e}, peak {peak:.3e}")
    assert err <= 1e-3                      # the stated contract, raw outputs
    assert err <= 1e-3 * peak               # and on the peak-normalised scale
    assert err64 <= 1e-3 * peak
    # what the kernels actually deliver (measured 1.2e-7 absolute = 9e-7 of the peak): a regression of one order of
    # magnitude fails here long before it reaches the contract
    assert err <= 5e-6 * peak


@pytest.mark.parametrize("B,T", [(1, 16), (2, 24), (1, 700), (3, 1031), (1, 2055)])
def test_very_short_inputs_match_oracle(B, T):
    """Edge cases of the frame grid: a single frame (T = 16), two frames, fewer frames than the largest TCN dilation
    (every dilated tap of the late blocks falls outside the utterance and is zero padding, reference:
    mossformer2_block.py:513-517), fewer frames than one attention group, a length that is no multiple of the hop.
    HIP forward (fp32) against the fp64 oracle on the small configuration; bf16 mode must stay finite and close."""
    from speech_separation_amd.synth import si_sdr
    sd = make_state_dict(SMALL, 0)
    x = torch.from_numpy(synth_batch(B, T, seed0=11))
    with torch.no_grad():
        ref = torch.stack(cpu_ref.forward(cpu_ref.to_torch(sd, torch.float64), x.double(), SMALL), 0)
    m = MossFormerHIP(SMALL, compute_device=DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    got = torch.stack(m(x.to(DEV)), 0).cpu()
    assert got.shape == (2, B, T)
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-9
    m16 = MossFormerHIP(SMALL, compute_device=DEV, precision=BF)
    m16.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    got16 = torch.stack(m16(x.to(DEV)), 0).cpu()
    assert torch.isfinite(got16).all()
    if T >= 700:
        assert min(si_sdr(got16[s, b].numpy(), ref[s, b].numpy()) for s in range(2) for b in range(B)) >= 30.0


def test_cpu_input_round_trip(models, golden_dir):
    """A CPU tensor in gives CPU tensors out (the Flask entry point feeds CPU tensors)."""
    z = np.load(os.path.join(golden_dir, "fwd_tiny_B1_T8000.npz"))
    out = models("tiny")(torch.from_numpy(z["input"]))
    assert all(o.device.type == "cpu" and o.shape == (1, 8000) for o in out)
    assert np.abs(torch.stack(out, 0).numpy() - z["out"]).max() <= 1e-5


# ------------------------------------------------------------ bf16 MFMA mode
# bf16 operands (8 significant bits), fp32 accumulation: single contractions agree with
# the fp64 oracle to ~1e-2 of the tensor's peak; the end-to-end gate is SI-SDR within
# 0.05 dB of the fp32 reference output.
def test_bf16_linear():
    g = torch.Generator().manual_seed(13)
    B, S, K, N = 2, 300, 512, 2176
    x = torch.randn(B, S, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    xr = cpu_ref.scale_norm(cpu_ref.token_shift(x), torch.ones(1))
    ref = torch.nn.functional.silu(xr.double() @ W.double().t() + b.double())
    got = ops.linear(x.to(DEV), W.to(DEV), b.to(DEV), shift=True, scale_norm=True, act=1, precision=BF)
    assert rel_err(got, ref) <= 1.5e-2
    # exact on bf16-representable data: the only rounding left is fp32 accumulation
    xq = x.to(torch.bfloat16).float()
    Wq = W.to(torch.bfloat16).float()
    got = ops.linear(xq.to(DEV), Wq.to(DEV), b.to(DEV), precision=BF)
    assert rel_err(got, xq.double() @ Wq.double().t() + b.double()) <= 2e-6


@pytest.mark.parametrize("B,S,E", [(2, 499, 128), (1, 700, 1024)])
def test_bf16_flash_attention(B, S, E):
    g = torch.Generator().manual_seed(14)
    v = torch.randn(B, S, E, generator=g)
    u = torch.randn(B, S, E, generator=g)
    qk = torch.randn(B, S, 128, generator=g)
    gam = 2.0 + 0.1 * torch.randn(4, 128, generator=g)
    bet = 0.1 * torch.randn(4, 128, generator=g)
    freqs = torch.from_numpy(make_state_dict(TINY)[f"{MDL}.layers.0.rotary_pos_emb.freqs"])
    ref = cpu_ref.attention_core(v.double(), u.double(), qk.double(), gam.double(), bet.double(), freqs.double())
    hu = torch.cat((v, u, qk), -1).contiguous()
    got = ops.flash_attention(hu.to(DEV), gam.to(DEV), bet.to(DEV), freqs.to(DEV), precision=BF)
    assert rel_err(got, ref) <= 3e-2


def test_bf16_gated_tcn(small_w):
    g = torch.Generator().manual_seed(15)
    B, S, C = 2, 777, SMALL.dim
    x = torch.randn(B, S, C, generator=g)
    W64 = {k: v.double() for k, v in small_w.items()}
    ref = cpu_ref.gated_tcn(x.double(), W64, f"{MDL}.fsmn.1")
    pk = PackedWeights(small_w, SMALL, DEV, BF)
    got = ops.gated_tcn(x.to(DEV).clone(), pk.layers[1], precision=BF)
    assert rel_err(got, ref) <= 3e-2


def test_bf16_full_model_si_sdr(golden_dir):
    """bf16 MFMA mode against the fp32 reference output: SI-SDR of the separated
    waveforms (against the synthetic sources) within 0.05 dB, and the waveforms
    themselves close in the SI-SDR sense (estimate vs reference output > 30 dB)."""
    from speech_separation_amd.synth import permute_si_sdr, si_sdr, synth_mixture
    z = np.load(os.path.join(golden_dir, "fwd_full_B1_T32000.npz"))
    m = MossFormerHIP(FULL_8K, compute_device=DEV, precision=BF)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(FULL_8K, 0).items()})
    got = torch.stack(m(torch.from_numpy(z["input"]).to(DEV)), 0).cpu().numpy()
    ref = z["out"]
    mix, s1, s2 = synth_mixture(32000, 0)
    assert np.array_equal(mix, z["input"][0])
    d = permute_si_sdr(got[0, 0], got[1, 0], s1, s2) - permute_si_sdr(ref[0, 0], ref[1, 0], s1, s2)
    fid = [si_sdr(got[s, 0], ref[s, 0]) for s in range(2)]
    print(f"bf16 full model: SI-SDR delta {d:+.4f} dB; fidelity to fp32 reference {fid[0]:.1f} / {fid[1]:.1f} dB; "
          f"max-abs {np.abs(got - ref).max():.3e} (peak {np.abs(ref).max():.3e})")
    assert abs(d) <= 0.05
    assert min(fid) >= 38.0                 # measured 42-43 dB


@pytest.mark.parametrize("name,B,T", [("tiny", 2, 8003), ("tiny", 3, 6408), ("small", 1, 16000), ("full", 3, 20000)])
def test_bf16_fused_attention_matches_two_kernel_form(name, B, T):
    """bf16 mode keeps the attention scores on chip (one kernel per layer: scores into LDS, both products, gating).
    The two-kernel form (scores written to HBM as bf16, then the products) rounds the same values at the same
    places; where a score lands on a rounding boundary the two forms can round it differently (the scores come out
    of the matrix cores in transposed roles), and 24 layers amplify such flips to the level of bf16 noise.  So the
    check is against the fp32 forward of the same weights: both forms are equally close to it, and no further from
    each other than from it.  Inputs include a last group that is partly padding and one (6408 samples = 800 frames)
    whose last half group lies entirely past the end of the utterance."""
    from speech_separation_amd.synth import si_sdr
    lib = _lib.load()
    m = MossFormerHIP(CFG[name], compute_device=DEV, precision=BF)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(CFG[name], 0).items()})
    x = torch.from_numpy(synth_batch(B, T, seed0=3)).to(DEV)
    outs = []
    for fused in (b"1", b"0"):
        _lib.check(lib.msf2_debug_kernel_choice(b"attn_fused", fused))
        _lib.check(lib.msf2_debug_kernel_choice(b"attn_fused_min", b"1"))     # whatever the grid size (it is a speed rule)
        try:
            outs.append(torch.stack(m(x), 0).cpu().numpy())
        finally:
            _lib.check(lib.msf2_debug_kernel_choice(None, None))
    assert np.isfinite(outs[0]).all()
    m32 = MossFormerHIP(CFG[name], compute_device=DEV)
    m32.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(CFG[name], 0).items()})
    ref = torch.stack(m32(x), 0).cpu().numpy()
    pairs = [(s, b) for s in range(2) for b in range(B)]
    mutual = min(si_sdr(outs[0][s, b], outs[1][s, b]) for s, b in pairs)
    fid = [min(si_sdr(o[s, b], ref[s, b]) for s, b in pairs) for o in outs]
    print(f"fused vs two-kernel attention, {name}: identical {np.array_equal(outs[0], outs[1])}, mutual agreement {mutual:.1f} dB, "
          f"fidelity to the fp32 forward: fused {fid[0]:.1f} dB, two-kernel {fid[1]:.1f} dB")
    # each form is as close to the fp32 forward as the other, and they differ from each other by no more than from it
    assert fid[0] >= 38.0 and abs(fid[0] - fid[1]) <= 3.0 and mutual >= min(fid) - 3.0


def test_bf16_kernel_choices_agree():
    """The bf16 projections pick their kernel by problem size (register-staged, 128 x 128 global->LDS ring with
    1 or 2 column tiles per workgroup, 256 x 256 8-wave geometry, 128 x 256 geometry, 128 x 128 with a 3- or 2-slot ring).  Every choice performs the same arithmetic
    in the same order per output element, so a gated TCN block at a size that enables all of them must
    not depend on which one ran."""
    g = torch.Generator().manual_seed(21)
    B, S, C = 32, 3999, FULL_8K.dim
    cfg1 = type(FULL_8K)(num_layers=1)
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(cfg1, 0).items()}
    pk = PackedWeights(sd, cfg1, DEV, BF)
    x = torch.randn(B, S, C, generator=g).to(DEV)
    outs = {}
    # the token-shifted N = 2176 projection (not part of the TCN block) at a size where every geometry is eligible
    xs = torch.randn(8, S, C, generator=g).to(DEV)
    W = (torch.randn(2176, C, generator=g) / C ** 0.5).to(DEV)
    bias = (torch.randn(2176, generator=g) * 0.1).to(DEV)
    outs_p = {}
    variants = (("default", {}), ("no_big", {"no_big": "1"}), ("loop1", {"no_big": "1", "ncol_loop": "1"}),
                ("loop4", {"no_big": "1", "ncol_loop": "4"}),
                ("geom2_all", {"geom2": "1", "geom3": "n", "geom4": "n"}),
                ("geom1_all", {"geom2": "0", "geom3": "n", "geom4": "n"}),
                ("geom3_all", {"geom3": "a0123", "geom4": "n"}), ("geom4_all", {"geom4": "a0123"}),
                ("row128_off", {"row128": "0"}), ("row128_all", {"row128": "2", "no_big": "1"}))
    lib = _lib.load()
    for name, choice in variants:
        _lib.check(lib.msf2_debug_kernel_choice(None, None))
        for k, v in choice.items():
            _lib.check(lib.msf2_debug_kernel_choice(k.encode(), v.encode()))
        try:
            outs[name] = ops.gated_tcn(x.clone(), pk.layers[0], precision=BF)
            outs_p[name] = ops.linear(xs, W, bias, shift=True, scale_norm=True, act=1, precision=BF)
        finally:
            _lib.check(lib.msf2_debug_kernel_choice(None, None))
    assert lib.msf2_debug_kernel_choice(b"no_such_choice", b"1") == -1
    assert torch.isfinite(outs["default"]).all() and torch.isfinite(outs_p["default"]).all()
    for name, _ in variants[1:]:
        assert torch.equal(outs[name], outs["default"]), name
        assert torch.equal(outs_p[name], outs_p["default"]), name + " (projection)"


@pytest.mark.parametrize("B,T", [(8, 32000), (11, 20011), (40, 8000)])
def test_bf16_register_resident_projection_agrees(B, T):
    """The bf16 K = 512 projections with a bf16-only output (token-shifted to_hidden | to_qk, N = 2176; the gate and proj_in
    of the gated TCN block) run on the kernel that keeps the activation rows in registers and streams only the
    weights once a call has a workgroup per CU (the third case: a batch of the segmented path's 1-s windows).  Same arithmetic in the same order per output element as the tile
    kernels it replaces: the forward must not depend on which one ran (2 layers at the real widths; the second length
    leaves a partial row tile of 3 rows)."""
    cfg2 = type(FULL_8K)(num_layers=2)
    m = MossFormerHIP(cfg2, compute_device=DEV, precision=BF)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg2, 0).items()})
    x = torch.from_numpy(synth_batch(B, T, seed0=3)).to(DEV)
    lib = _lib.load()
    outs = {}
    for val in (b"1", b"0"):
        _lib.check(lib.msf2_debug_kernel_choice(None, None))
        _lib.check(lib.msf2_debug_kernel_choice(b"astat", val))
        try:
            outs[val] = torch.stack(m(x), 0)
        finally:
            _lib.check(lib.msf2_debug_kernel_choice(None, None))
    assert torch.isfinite(outs[b"1"]).all()
    assert torch.equal(outs[b"1"], outs[b"0"])


@pytest.mark.parametrize("B,T", [(1, 32000), (3, 20011), (5, 4000), (8, 32000)])
def test_bf16_key_value_summary_kernels_agree(B, T):
    """bf16 mode computes the linear-attention key/value summary (lin_k^T [v | u] over time, reference
    mossformer2_block.py:318-322) with a kernel that copies both operands global -> LDS as they lie and transposes them in
    the LDS read; the register-staged tile kernel it replaces stays selectable.  Same bf16 products, summed by the same MFMA
    instruction in the same order over the same time chunks: the forward of a 2-layer full-width model must be
    bit-identical between the two.  The shapes cover 16 time chunks with a partial last tile, a batch with a ragged
    length, fewer frames than one chunk, and 8 chunks per utterance."""
    cfg2 = type(FULL_8K)(num_layers=2)
    m = MossFormerHIP(cfg2, compute_device=DEV, precision=BF)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg2, 0).items()})
    x = torch.from_numpy(synth_batch(B, T, seed0=5)).to(DEV)
    lib = _lib.load()
    outs = {}
    for val in (b"1", b"0"):
        _lib.check(lib.msf2_debug_kernel_choice(None, None))
        _lib.check(lib.msf2_debug_kernel_choice(b"linkv_dma", val))
        try:
            outs[val] = torch.stack(m(x), 0).cpu().numpy()
        finally:
            _lib.check(lib.msf2_debug_kernel_choice(None, None))
    assert np.isfinite(outs[b"1"]).all() and np.abs(outs[b"1"]).max() > 0
    assert np.array_equal(outs[b"1"], outs[b"0"])


def test_bf16_equivalent_kernel_forms_on_ragged_shapes():
    """The bit-identical kernel pairs of the bf16 mode (register-resident-activation projections, K = 512 and K = 1024, vs
    tile kernels; the gate and proj_in of the gated TCN block in one walk vs two; the two-product TCN contraction with h and t
    resident in registers vs the tile kernel; key/value summary global -> LDS kernel vs tile kernel) on a dozen seeded shapes with odd lengths: frame counts just below, at and
    just above multiples of the 32-frame k tile, the 128-row tile and the 256-frame group, batches 1 .. 9."""
    cfg2 = type(FULL_8K)(num_layers=2)
    m = MossFormerHIP(cfg2, compute_device=DEV, precision=BF)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg2, 0).items()})
    lib = _lib.load()
    rng = np.random.default_rng(11)
    shapes = [(1, 8 * 255 + 16), (1, 8 * 256 + 16), (2, 8 * 257 + 16), (9, 8 * 127 + 16), (3, 8 * 128 + 23), (1, 8 * 1023 + 16),
              (2, 8 * 1024 + 16), (1, 8 * 1025 + 19)]
    shapes += [(int(rng.integers(1, 10)), int(rng.integers(1600, 40000))) for _ in range(4)]
    for B, T in shapes:
        x = torch.from_numpy(synth_batch(B, T, seed0=13)).to(DEV)
        for knob in (b"astat", b"astat_k1024", b"astat_merge", b"astat_tcn", b"linkv_dma"):
            outs = {}
            for val in (b"1", b"0"):
                _lib.check(lib.msf2_debug_kernel_choice(None, None))
                _lib.check(lib.msf2_debug_kernel_choice(knob, val))
                try:
                    outs[val] = torch.stack(m(x), 0)
                finally:
                    _lib.check(lib.msf2_debug_kernel_choice(None, None))
            assert torch.isfinite(outs[b"1"]).all(), (B, T, knob)
            assert torch.equal(outs[b"1"], outs[b"0"]), (B, T, knob)


def test_bf16_batch_independence():
    """bf16 mode, full model: an utterance gives the same waveform alone and inside a batch of 12, up to
    bf16 rounding noise: the linear-attention summary is split over time in a batch-dependent number of
    chunks, the fp32 sums differ in their last bits, and in bf16 mode such bits flip roundings downstream.
    Bound: the two waveforms agree to better than 35 dB (fp32 mode: 1e-5 of the peak, see
    test_full_size_batch_properties)."""
    from speech_separation_amd.synth import si_sdr
    m = MossFormerHIP(FULL_8K, compute_device=DEV, precision=BF)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(FULL_8K, 0).items()})
    x = torch.from_numpy(synth_batch(12, 32000, seed0=0)).to(DEV)
    a = torch.stack(m(x), 0)
    single = torch.stack(m(x[7:8]), 0)
    peak = float(a.abs().max())
    assert torch.isfinite(a).all()
    assert float((a[:, 7:8] - single).abs().max()) <= 2e-2 * peak
    for spk in range(2):
        assert si_sdr(a[spk, 7].cpu().numpy(), single[spk, 0].cpu().numpy()) >= 35.0


# -------------------------------------------------- size-independent properties
def test_full_size_batch_properties(models):
    """At the benchmark size (16 x 4 s, 24 layers): utterances in a batch are
    independent, the result is deterministic, and scaling the input by a
    power of two leaves the peak-normalised output unchanged only through the
    homogeneity of the network's linear/normalised parts -- checked here as
    batch-vs-single agreement and bitwise repeatability."""
    m = models("full")
    x = torch.from_numpy(synth_batch(16, 32000, seed0=0)).to(DEV)
    a = torch.stack(m(x), 0)
    b = torch.stack(m(x), 0)
    assert torch.equal(a, b)                                     # deterministic
    assert torch.isfinite(a).all()
    single = torch.stack(m(x[5:6]), 0)
    peak = float(a.abs().max())
    assert float((a[:, 5:6] - single).abs().max()) <= 1e-5 * peak   # batch independence
    perm = torch.arange(15, -1, -1, device=DEV)
    c = torch.stack(m(x[perm]), 0)
    assert float((c[:, perm] - a).abs().max()) <= 1e-5 * peak       # permutation equivariance


@pytest.mark.parametrize("name", ["tiny", "small", "full"])
def test_long_form_matches_reference(models, golden_dir, name):
    """One 60-second mixture after the decode padding rule: 492000 samples, 61 499 frames, 241 groups, against the
    reference model's own output (stored every `stride`-th sample; a frame spans 16 samples).  This is the case
    in which the linear-attention summary is reduced over split time ranges (reference divides by the unpadded
    length, mossformer2_block.py:318-322) and every GroupNorm(1, C) reduces 61 499 x C values (:521).
    fp32: 1e-5 absolute on outputs of rms 5e-3 (tiny, small), 1e-3 of the peak for the 24-layer model;
    bf16: SI-SDR against the synthetic sources within 0.05 dB of the reference's for the 24-layer model (the
    stated contract); the 2-layer toy configurations are held to 0.1 dB (measured 0.006 / 0.047 dB: with random
    weights and 64 / 128 channels their outputs sit near -20 dB SI-SDR, where a fixed waveform error moves the
    figure more)."""
    from speech_separation_amd.synth import permute_si_sdr, synth_mixture
    f = os.path.join(golden_dir, f"long_{name}_B1_T492000.npz")
    if not os.path.exists(f):
        pytest.skip(f"{os.path.basename(f)} not generated")
    z = np.load(f)
    T, stride = int(z["T"]), int(z["stride"])
    mix, s1, s2 = synth_mixture(T, int(z["seed0"]))
    assert np.array_equal(mix[:64], z["input_head"][0])
    x = torch.from_numpy(mix[None]).to(DEV)
    ref = z["out_strided"]
    out = torch.stack(models(name)(x), 0)
    assert out.shape == (2, 1, T) and torch.isfinite(out).all()
    got = out.cpu().numpy()[:, :, ::stride]
    err = np.abs(got - ref).max()
    print(f"long form {name}: fp32 max-abs err {err:.3e} (peak {float(z['out_peak']):.3e})")
    # measured 4.1e-8 / 3.9e-8 (tiny, small) and 1.4e-7 (24 layers, peak ~ 0.1)
    assert err <= (1e-5 * float(z["out_peak"]) if name == "full" else 1e-6)
    assert err <= 1e-3                      # the stated contract
    m16 = MossFormerHIP(CFG[name], compute_device=DEV, precision=BF)
    m16.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(CFG[name], 0).items()})
    got16 = torch.stack(m16(x), 0).cpu().numpy()[:, :, ::stride]
    # SI-SDR on the stored sample grid (every stride-th sample of estimate and source alike)
    d = (permute_si_sdr(got16[0, 0], got16[1, 0], s1[::stride], s2[::stride])
         - permute_si_sdr(ref[0, 0], ref[1, 0], s1[::stride], s2[::stride]))
    print(f"long form {name}: bf16 SI-SDR delta {d:+.4f} dB")
    assert abs(d) <= (0.05 if name == "full" else 0.1)


def test_narrow_tiles_agree_with_wide_tiles(models, golden_dir):
    """One utterance per call (the callers' own shape): the fp32 projections then run on 64-column tiles so that
    the launch fills the chip.  Each output element is accumulated in the same order as on 128-column tiles; only
    the ScaleNorm partial sums are grouped differently (32- instead of 64-column slabs), so the two forms agree to
    fp32 rounding, and both match the reference fixture."""
    lib = _lib.load()
    z = np.load(os.path.join(golden_dir, "fwd_full_B1_T32000.npz"))
    x = torch.from_numpy(z["input"]).to(DEV)
    m = models("full")
    outs = {}
    for narrow in (b"1", b"0"):
        _lib.check(lib.msf2_debug_kernel_choice(b"narrow", narrow))
        try:
            outs[narrow] = torch.stack(m(x), 0).cpu().numpy()
        finally:
            _lib.check(lib.msf2_debug_kernel_choice(None, None))
    peak = np.abs(z["out"]).max()
    assert np.abs(outs[b"1"] - outs[b"0"]).max() <= 1e-5 * peak
    for o in outs.values():
        assert np.abs(o - z["out"]).max() <= 1e-3 * peak
    # a single projection is bit-identical in the two forms
    g = torch.Generator().manual_seed(41)
    xx = torch.randn(1, 3999, 512, generator=g).to(DEV)
    W = (torch.randn(512, 512, generator=g) / 512 ** 0.5).to(DEV)
    bias = (torch.randn(512, generator=g) * 0.1).to(DEV)
    ys = []
    for narrow in (b"1", b"0"):
        _lib.check(lib.msf2_debug_kernel_choice(b"narrow", narrow))
        try:
            ys.append(ops.linear(xx, W, bias, act=1))
        finally:
            _lib.check(lib.msf2_debug_kernel_choice(None, None))
    assert torch.equal(ys[0], ys[1])
    # bf16 mode: 64-row tiles of the global->LDS kernel for the same reason; same arithmetic per output element
    from speech_separation_amd.synth import si_sdr
    m16 = MossFormerHIP(FULL_8K, compute_device=DEV, precision=BF)
    m16.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(FULL_8K, 0).items()})
    o16 = []
    for narrow in (b"1", b"0"):
        _lib.check(lib.msf2_debug_kernel_choice(b"narrow", narrow))
        try:
            o16.append(torch.stack(m16(x), 0).cpu().numpy())
        finally:
            _lib.check(lib.msf2_debug_kernel_choice(None, None))
    agree = min(si_sdr(o16[0][s_, 0], o16[1][s_, 0]) for s_ in range(2))
    fid = min(si_sdr(o16[0][s_, 0], z["out"][s_, 0]) for s_ in range(2))
    print(f"bf16 narrow vs wide tiles: identical {np.array_equal(o16[0], o16[1])}, agreement {agree:.1f} dB, fidelity to the reference {fid:.1f} dB")
    assert agree >= 40.0 and fid >= 38.0


def test_small_grid_forms_reduce_their_own_statistics(models, golden_dir):
    """With one utterance per call the two-product TCN contraction (64-column fp32 tiles / 64-row bf16 tiles) reduces the
    partial sums of stage A itself instead of waiting for a finalize kernel (192 launches fewer per forward).  Same
    summation order, same folding of the statistics into the column vector: the waveforms must be bit-identical to the
    form with the finalize kernel, in both precisions."""
    lib = _lib.load()
    z = np.load(os.path.join(golden_dir, "fwd_full_B1_T32000.npz"))
    x = torch.from_numpy(z["input"]).to(DEV)
    m16 = MossFormerHIP(FULL_8K, compute_device=DEV, precision=BF)
    m16.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(FULL_8K, 0).items()})
    for name, m in (("fp32", models("full")), ("bf16", m16)):
        outs = []
        for own in (b"1", b"0"):
            _lib.check(lib.msf2_debug_kernel_choice(b"own_stats", own))
            try:
                outs.append(torch.stack(m(x), 0))
            finally:
                _lib.check(lib.msf2_debug_kernel_choice(None, None))
        assert torch.isfinite(outs[0]).all(), name
        assert torch.equal(outs[0], outs[1]), name
    # bf16, one workgroup per CU at most: the 64-row tiles take whole-line row pieces (k tiles of 64) -- same arithmetic
    outs = []
    for r128 in (b"1", b"0"):
        _lib.check(lib.msf2_debug_kernel_choice(b"row128", r128))
        try:
            outs.append(torch.stack(m16(x), 0))
        finally:
            _lib.check(lib.msf2_debug_kernel_choice(None, None))
    assert torch.equal(outs[0], outs[1])
    # a ragged shape: 3 utterances of 1.3 s (fewer than 256 tiles as well), small configuration
    ms = models("small")
    xs = torch.from_numpy(synth_batch(3, 10403, seed0=5)).to(DEV)
    outs = []
    for own in (b"1", b"0"):
        _lib.check(lib.msf2_debug_kernel_choice(b"own_stats", own))
        try:
            outs.append(torch.stack(ms(xs), 0))
        finally:
            _lib.check(lib.msf2_debug_kernel_choice(None, None))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("prec", [_lib.PREC_FP32, _lib.PREC_BF16])
def test_forward_ignores_workspace_contents(prec):
    """The caller owns the workspace and hands it over uninitialised: results must not depend on what it holds
    (zeros, NaN patterns, large finite values, or what an earlier forward of another input left there)."""
    from speech_separation_amd.config import MossFormer2Config
    lib = _lib.load()
    for cfg, B, T in ((TINY, 2, 8003), (MossFormer2Config(num_layers=2), 1, 32000), (MossFormer2Config(num_layers=2), 3, 12000)):
        m = MossFormerHIP(cfg, compute_device=DEV, precision=prec)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()})
        pk = m.packed()
        x = torch.from_numpy(synth_batch(B, T, seed0=0)).to(DEV)
        other = torch.from_numpy(synth_batch(B, T, seed0=50)).to(DEV)
        n = lib.msf2_workspace_bytes(pk.c_cfg, B, T)
        outs = []
        for fill in (0x00, 0xFF, 0x7F, None):
            ws = torch.full((n,), fill or 0, dtype=torch.uint8, device=DEV)
            out = torch.empty((2, B, T), device=DEV)
            if fill is None:        # leftovers of a forward of another input
                _lib.check(lib.msf2_forward(pk.c_cfg, pk.struct, other.data_ptr(), out.data_ptr(), B, T, ws.data_ptr(), n, None, None))
            _lib.check(lib.msf2_forward(pk.c_cfg, pk.struct, x.data_ptr(), out.data_ptr(), B, T, ws.data_ptr(), n, None, None))
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.isfinite(outs[0]).all()
        for o in outs[1:]:
            assert torch.equal(o, outs[0])


def test_batches_beyond_one_call_are_split(models):
    """A batch whose widest activation would pass 4 GiB (the kernels use 32-bit byte offsets) is separated in pieces:
    same results as the utterances alone, no error for the caller."""
    lib = _lib.load()
    m = models("full")
    B, T = 130, 32000                       # 4.16 M samples: one call takes at most about 3.9 M
    assert lib.msf2_workspace_bytes(m.packed().c_cfg, B, T) == 0
    x = torch.from_numpy(synth_batch(2, T, seed0=0)).to(DEV).repeat(B // 2, 1)
    out = torch.stack(m(x), 0)
    assert out.shape == (2, B, T) and torch.isfinite(out).all()
    two = torch.stack(m(x[:2]), 0)
    peak = float(two.abs().max())
    assert float((out[:, :2] - two).abs().max()) <= 1e-5 * peak
    assert float((out[:, 128:130] - two).abs().max()) <= 1e-5 * peak


def test_graph_replay_matches_eager():
    """use_graphs=True replays the captured launch sequence of a shape: bit-identical to the eager forward, for
    calls issued back to back without a host synchronisation and for a second shape, fp32 and bf16."""
    from speech_separation_amd.config import MossFormer2Config
    cfg = MossFormer2Config(num_layers=3)
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 0).items()}
    for prec in (_lib.PREC_FP32, BF):
        m = MossFormerHIP(cfg, compute_device=DEV, precision=prec)
        m.load_state_dict(sd)
        xs = [torch.from_numpy(synth_batch(1, 32000, seed0=s)).to(DEV) for s in range(3)]
        xs.append(torch.from_numpy(synth_batch(2, 12000, seed0=9)).to(DEV))
        want = [torch.stack(m(x), 0) for x in xs]
        m.use_graphs = True
        got = [torch.stack(m(x), 0) for x in xs + xs]          # no synchronisation in between
        torch.cuda.synchronize()
        again = [torch.stack(m(x), 0) for x in xs]
        for g, w in zip(got + again, want + want + want):
            assert torch.equal(g, w)
        assert len(m._graphs) == 2                                # one graph per input shape


def test_forward_is_reentrant_across_threads(models):
    """A threaded server shares one model object (reference: app.py:256-259, Flask's threaded dev server): four
    host threads, each on its own HIP stream, separate different inputs at the same time (one of them with the
    kernel timer running); every result equals the one computed alone."""
    import threading
    m = models("small")
    xs = [torch.from_numpy(synth_batch(2, 8000 + 800 * i, seed0=10 * i)).to(DEV) for i in range(4)]
    serial = [torch.stack(m(x), 0) for x in xs]
    torch.cuda.synchronize()
    lib = _lib.load()
    results, errors = [None] * 4, []
    start = threading.Barrier(4)

    def work(i):
        try:
            st = torch.cuda.Stream(device=DEV)
            start.wait()
            with torch.cuda.stream(st):
                if i == 0:
                    _lib.check(lib.msf2_timing_begin())
                outs = [torch.stack(m(xs[i]), 0) for _ in range(6)]
                st.synchronize()
                if i == 0:
                    import ctypes
                    tot, n = ctypes.c_double(0.0), ctypes.c_int(0)
                    _lib.check(lib.msf2_timing_end(ctypes.byref(tot), ctypes.byref(n)))
                    assert n.value == 6 * 2 * 8, n.value       # only this thread's launches: 6 forwards x 2 layers x 8 blocks
            results[i] = outs
        except Exception as e:             # noqa: BLE001
            errors.append((i, repr(e)))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for i in range(4):
        for o in results[i]:
            assert torch.equal(o, serial[i]), i


def test_file_to_file_driver(golden_dir, tmp_path):
    """inference.py's loop on a wav the reference ships: two 16-bit files of the input's length, peak normalised."""
    import shutil
    from types import SimpleNamespace
    from speech_separation_amd import infer
    from speech_separation_amd.wavio import read_wav
    src = tmp_path / "in"
    src.mkdir()
    shutil.copy(os.path.join(golden_dir, "ref_input_nam_nam_8k.wav"), src / "mix.wav")
    ns = SimpleNamespace(input_path=str(src), output_dir=str(tmp_path / "out"), checkpoint_dir=None, precision="fp32",
                         one_time_decode_length=60, decode_window=1)
    written = infer.run(ns)
    assert [os.path.basename(p) for p in written] == ["mix_s1.wav", "mix_s2.wav"]
    peaks = []
    for p in written:
        x, fs = read_wav(p)
        assert fs == 8000 and x.shape == (24360,)
        peaks.append(np.abs(x).max())
    assert 0.2 < max(peaks) <= 1.0                  # joint peak normalisation over the padded length (reference: utils/decode.py:78-83)


def test_pit_si_snr(golden_dir):
    """Device PIT SI-SNR against the oracle restatement of the reference's cal_si_snr (fp64) and against the
    values the reference itself produced for the fixture (same-index pairs)."""
    z = np.load(os.path.join(golden_dir, "si_snr.npz"))
    src = torch.from_numpy(z["src"]).permute(2, 1, 0).contiguous()        # [T, B, C] -> [C, B, T]
    est = torch.from_numpy(z["est"]).permute(2, 1, 0).contiguous()
    si, best, perm = ops.pit_si_snr(est.to(DEV), src.to(DEV))
    ref_si, ref_best, ref_perm = cpu_ref.pit_si_snr(est.double(), src.double())
    assert float((si.cpu().double() - ref_si).abs().max()) <= 1e-3
    assert float((best.cpu().double() - ref_best).abs().max()) <= 1e-3
    assert torch.equal(perm.cpu(), ref_perm)
    for b in range(3):
        for c in range(2):
            assert abs(float(si[b, c, c]) - float(z["si_snr"][b, c])) <= 1e-3
    # long, odd length, many blocks, a DC offset and swapped speakers
    g = torch.Generator().manual_seed(5)
    B, T = 5, 100003
    s = torch.randn(2, B, T, generator=g) + 0.3
    e = s.flip(0) * 0.8 + 0.1 * torch.randn(2, B, T, generator=g)
    e[:, 3] = s[:, 3] + 0.05 * torch.randn(2, T, generator=g)
    si, best, perm = ops.pit_si_snr(e.to(DEV), s.to(DEV))
    ref_si, ref_best, ref_perm = cpu_ref.pit_si_snr(e.double(), s.double())
    assert float((si.cpu().double() - ref_si).abs().max()) <= 1e-3
    assert torch.equal(perm.cpu(), ref_perm) and perm.cpu().tolist() == [1, 1, 1, 0, 1]


def test_sweep_statistic_on_device(models):
    """separate_sharded on the GPU reduces SI-SDR with the device kernel: same statistic as the host formula
    (reference: utils/si_sdr_numpy.py:27-41) up to the EPS terms of cal_si_snr."""
    from speech_separation_amd.parallel import separate_sharded
    from speech_separation_amd.synth import permute_si_sdr, synth_mixture
    m = models("tiny")
    data = [synth_mixture(n, i) for i, n in enumerate((8000, 8000, 12000))]
    utts = [d[0] for d in data]
    refs = [(d[1], d[2]) for d in data]
    outs, stats = separate_sharded(lambda x: m(x), utts, device=DEV, references=refs, max_batch=4)
    host = [permute_si_sdr(o[0], o[1], r[0], r[1]) for o, r in zip(outs, refs)]
    assert stats[2] == 3.0
    assert abs(stats[0] - sum(host)) <= 3e-3


def test_sweep_of_mixed_lengths_on_the_full_model(models):
    """A small edition of the sharded sweep (exact-length buckets, longest first, batches capped by samples) on the
    24-layer model: every utterance comes back at its own length and equals the forward of that utterance alone;
    the SI-SDR statistic reduced on the device equals the host formula."""
    from speech_separation_amd.parallel import separate_sharded
    from speech_separation_amd.synth import permute_si_sdr, synth_mixture
    m = models("full")
    lens = [24000, 32000, 24000, 40000, 32000, 24000, 28000]
    data = [synth_mixture(n, 200 + i) for i, n in enumerate(lens)]
    utts, refs = [d[0] for d in data], [(d[1], d[2]) for d in data]
    calls = []

    def fn(x):
        calls.append(tuple(x.shape))
        return m(x)
    outs, stats = separate_sharded(fn, utts, device=DEV, references=refs, max_batch=2)
    assert sorted(calls) == sorted([(1, 40000), (2, 32000), (1, 28000), (2, 24000), (1, 24000)])
    for i in (0, 3, 4):
        alone = torch.stack(m(torch.from_numpy(utts[i][None]).to(DEV)), 0)[:, 0].cpu().numpy()
        assert outs[i].shape == (2, lens[i])
        assert np.abs(outs[i] - alone).max() <= 1e-5 * np.abs(alone).max()
    host = [permute_si_sdr(o[0], o[1], r[0], r[1]) for o, r in zip(outs, refs)]
    assert stats[2] == len(lens) and abs(stats[0] - sum(host)) <= 5e-3


@pytest.mark.parametrize("name", ["tiny", "small", "tinyfsmn"])
def test_bf16_mode_on_reference_fixtures(golden_dir, name):
    """bf16 MFMA mode on the small reference fixtures: the separated waveforms stay within 30 dB (SI-SDR of the
    estimate against the reference model's own fp32 output) for every utterance and speaker."""
    from speech_separation_amd.synth import si_sdr
    m = MossFormerHIP(CFG[name], compute_device=DEV, precision=BF)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(CFG[name], 0).items()})
    worst = 1e9
    for f in sorted(glob.glob(os.path.join(golden_dir, f"fwd_{name}_B*.npz"))):
        z = np.load(f)
        got = torch.stack(m(torch.from_numpy(z["input"]).to(DEV)), 0).cpu().numpy()
        assert np.isfinite(got).all()
        for s in range(got.shape[0]):
            for b in range(got.shape[1]):
                worst = min(worst, si_sdr(got[s, b], z["out"][s, b]))
    print(f"bf16 mode, {name} fixtures: worst fidelity {worst:.1f} dB")
    assert worst >= 40.0                    # measured 45.3 - 46.9 dB


def test_gated_fsmn_block(golden_dir):
    """FSMN-family block: against the reference module's outputs (fixture, dim 64 / inner 64) and, at the
    reference's real size (dim 512, inner 256), against the fp64 oracle."""
    from speech_separation_amd.packing import pack_fsmn_block
    z = np.load(os.path.join(golden_dir, "fsmn_block.npz"))
    W = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    pk = pack_fsmn_block(W, DEV)
    for name in ("a", "b"):
        got = ops.gated_fsmn(torch.from_numpy(z["in_" + name]).to(DEV), pk)
        ref = torch.from_numpy(z["out_" + name])
        assert float((got.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 2e-5
    g = torch.Generator().manual_seed(31)
    dim, inner, B, S = 512, 256, 2, 1333
    shapes = {"conv1.0.weight": (inner, dim, 1), "conv1.0.bias": (inner,), "conv1.1.weight": (1,), "norm1.weight": (inner,),
              "norm1.bias": (inner,), "norm2.weight": (inner,), "norm2.bias": (inner,), "conv2.weight": (dim, inner, 1),
              "conv2.bias": (dim,), "gated_fsmn.fsmn.linear.weight": (inner, inner), "gated_fsmn.fsmn.linear.bias": (inner,),
              "gated_fsmn.fsmn.project.weight": (inner, inner), "gated_fsmn.fsmn.conv.conv1.weight": (inner, 1, 39, 1),
              "gated_fsmn.fsmn.conv.conv2.weight": (inner, 2, 39, 1)}
    for br in ("to_u", "to_v"):
        shapes.update({f"gated_fsmn.{br}.mdl.0.weight": (inner,), f"gated_fsmn.{br}.mdl.0.bias": (inner,),
                       f"gated_fsmn.{br}.mdl.1.weight": (inner, inner), f"gated_fsmn.{br}.mdl.1.bias": (inner,),
                       f"gated_fsmn.{br}.mdl.3.sequential.1.conv.weight": (inner, 1, 17)})
    for i in (1, 2):
        shapes.update({f"gated_fsmn.fsmn.conv.norm{i}.weight": (inner,), f"gated_fsmn.fsmn.conv.norm{i}.bias": (inner,),
                       f"gated_fsmn.fsmn.conv.prelu{i}.weight": (inner,)})
    Wb = {}
    for k, shp in shapes.items():
        if ("norm" in k or ".mdl.0." in k) and k.endswith("weight"):
            Wb[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif "prelu" in k or k == "conv1.1.weight":
            Wb[k] = 0.25 + 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            Wb[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            Wb[k] = (torch.rand(shp, generator=g) * 2 - 1) / fan ** 0.5
    x = torch.randn(B, S, dim, generator=g)
    W64 = {k: v.double() for k, v in Wb.items()}
    ref = cpu_ref.gated_fsmn_block_dilated(x.double(), W64)
    pkb = pack_fsmn_block(Wb, DEV)
    got = ops.gated_fsmn(x.to(DEV), pkb)
    assert rel_err(got, ref) <= 2e-5
    # lengths around the tiling of the 39-tap convolutions (16 frames per thread, frames `dil` apart for the dilated
    # one): shorter than a window, shorter than a tile, one past a tile, odd
    for B2, S2 in ((1, 1), (3, 7), (2, 16), (2, 33), (1, 41), (2, 130)):
        xs = torch.randn(B2, S2, dim, generator=g)
        ref = cpu_ref.gated_fsmn_block_dilated(xs.double(), W64)
        got = ops.gated_fsmn(xs.to(DEV), pkb)
        assert rel_err(got, ref) <= 2e-5, (B2, S2)
