"""CPU-side tests: the C-ABI library loads and exports what include/msf2.h
declares, and the host logic (module skeleton, checkpoint reload, decode
harness, sharding) behaves like the reference's.  No kernel is launched."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import cpu_ref
from speech_separation_amd import _lib
from speech_separation_amd.checkpoint import reload_for_eval
from speech_separation_amd.config import TINY, FULL_8K, MossFormer2Config
from speech_separation_amd.decode import decode_one_audio, padded_length
from speech_separation_amd.model import MossFormerHIP, MossFormer2_SS
from speech_separation_amd.schema import state_dict_schema
from speech_separation_amd.synth import make_state_dict, synth_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Args:
    network = "MossFormer2_SS_8K"
    sampling_rate = 8000
    decode_window = 1
    one_time_decode_length = 60
    num_spks = 2
    encoder_kernel_size = 16
    encoder_embedding_dim = 64
    mossformer_sequence_dim = 64
    num_mossformer_layer = 2


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "msf2.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(msf2_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()                       # raises if the .so or a symbol is missing
    assert lib.msf2_version() >= 1
    assert lib.msf2_num_frames(32000) == 3999
    assert lib.msf2_strerror(-2).decode() == "workspace too small"


def test_library_was_built_from_these_sources():
    """The shared library that ships with the tree (it is not tracked by git) carries a digest of the sources it
    was compiled from; a library left over from other sources fails here instead of being tested by mistake."""
    from speech_separation_amd import build
    assert os.path.exists(build.OUT)
    assert build.up_to_date(), "libmsf2.so is stale: run `python -m speech_separation_amd.build`"


def test_workspace_query_runs_without_gpu():
    lib = _lib.load()
    cfg = _lib.Config(512, 24, 16, 2, _lib.PREC_FP32)
    n1 = lib.msf2_workspace_bytes(cfg, 1, 32000)
    n16 = lib.msf2_workspace_bytes(cfg, 16, 32000)
    assert 0 < n1 < n16 < 16 * (1 << 30)
    assert lib.msf2_workspace_bytes(_lib.Config(512, 24, 16, 3, 0), 1, 32000) == 0   # 3 speakers unsupported


def test_struct_sizes_match_header_layout():
    import ctypes as C
    lib = _lib.load()
    for which, st in enumerate((_lib.Config, _lib.TcnWeights, _lib.LayerWeights, _lib.Weights, _lib.Taps)):
        assert lib.msf2_sizeof(which) == C.sizeof(st), st
    assert lib.msf2_sizeof(99) == -1


def test_state_dict_schema_and_counts():
    sch = state_dict_schema(FULL_8K)
    assert len(sch) == 2107
    numel = sum(int(np.prod(s)) for s in sch.values())
    assert numel == 162_488_010            # parameters + inv_freq buffer + 24 rotary freqs copies
    m = MossFormerHIP(TINY)
    sd = m.state_dict()
    assert list(sd.keys()) == list(state_dict_schema(TINY).keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(state_dict_schema(TINY)[k]), k
    assert "mask_net.pos_enc.inv_freq" not in dict(m.named_parameters())
    n_params = sum(p.numel() for p in m.parameters())
    # the reference shares one rotary module (16 freqs, counted once: 275 480); here each layer holds its copy
    assert n_params == 275_480 + 16


def test_wrapper_reads_reference_arg_names():
    w = MossFormer2_SS(Args())
    assert isinstance(w.model, MossFormerHIP)
    assert w.model.cfg == MossFormer2Config(dim=64, num_layers=2)
    a = Args()
    a.mossformer_sequence_dim = 128
    with pytest.raises(ValueError):
        MossFormer2_SS(a)


def test_forward_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = MossFormerHIP(TINY)
    with pytest.raises(_lib.Msf2Error):
        m(torch.zeros(1, 4000))


def test_reload_for_eval_conventions(tmp_path, capsys):
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(TINY, 5).items()}
    # checkpoint as the reference's trainer writes it: weights under 'model', DDP 'module.' prefix
    ck = {"model": {"module." + k: v for k, v in sd.items()}, "epoch": 3}
    bad = "module.mask_net.prelu.weight"
    ck["model"][bad] = torch.zeros(7)      # wrong shape -> skipped, reported
    torch.save(ck, tmp_path / "model.ckpt-3-100.pt")
    (tmp_path / "last_checkpoint").write_text("other.pt\n")
    (tmp_path / "last_best_checkpoint").write_text("model.ckpt-3-100.pt\n")
    m = MossFormerHIP(TINY)
    before = m.state_dict()["mask_net.prelu.weight"].clone()
    reload_for_eval(m, str(tmp_path), use_cuda=0)
    out = capsys.readouterr().out
    assert "mask_net.prelu.weight not loaded" in out
    got = m.state_dict()
    assert torch.equal(got["mask_net.prelu.weight"], before)
    for k in sd:
        if k != "mask_net.prelu.weight":
            assert torch.equal(got[k], sd[k]), k
    # no pointer file: warning, weights untouched
    reload_for_eval(m, str(tmp_path / "nope"), use_cuda=0)
    assert "Warning" in capsys.readouterr().out


def test_decode_harness_on_cpu_model(golden_dir):
    """Host logic of the decode harness with the oracle standing in for the model."""
    z = np.load(os.path.join(golden_dir, "decode_tiny.npz"))
    W = cpu_ref.to_torch(make_state_dict(TINY, 0))

    def model(x):
        with torch.no_grad():
            return cpu_ref.forward(W, x, TINY)
    a = Args()
    for T in (4000, 8000, 10000, 28400, 32000):
        out = decode_one_audio(model, torch.device("cpu"), z[f"in_{T}"], a)
        got = np.stack(out, 0)
        assert got.dtype == np.float32
        assert np.abs(got - z[f"out_{T}"]).max() <= 2e-4
    a.one_time_decode_length = 2
    out = decode_one_audio(model, torch.device("cpu"), z["in_seg_20001"], a)
    got = np.stack(out, 0)
    assert got.dtype == np.float64 and got.shape == (2, 28002)
    assert np.abs(got - z["out_seg_20001"]).max() <= 2e-4
    a.network = "something_else"
    assert decode_one_audio(model, torch.device("cpu"), z["in_4000"], a) is None


def test_padded_length_table():
    for t, want in ((4000, 8000), (8000, 14000), (10000, 14000), (14000, 14000), (20000, 20000),
                    (28400, 38800), (32000, 32000), (480000, 492000), (480001, 492002)):
        assert padded_length(t, 8000, 6000) == want


def test_flops_formula():
    # 1.5109 TFLOP per 4-s utterance, 377.7 GFLOP per audio-second
    assert FULL_8K.flops(32000) == pytest.approx(2 * (175_217_664 * 3999 + 13_369_344 * 4096))
    assert FULL_8K.frames(32000) == 3999 and FULL_8K.padded_frames(32000) == 4096


def test_synth_mixture_shape_and_level():
    x = synth_batch(2, 8000, seed0=0)
    assert x.shape == (2, 8000) and x.dtype == np.float32
    assert np.abs(x).max() <= 1.0
    rms = np.sqrt((x ** 2).mean())
    assert 0.03 < rms < 0.2
    assert np.array_equal(x, synth_batch(2, 8000, seed0=0))


# ------------------------------------------------------------------ wav I/O
def test_wav_read_matches_pcm16_definition(golden_dir):
    """A file the reference ships (outputs/clean/input): 16-bit mono 8 kHz, samples = int16 / 32768."""
    import wave
    from speech_separation_amd.wavio import audioread, read_wav
    path = os.path.join(golden_dir, "ref_input_nam_nam_8k.wav")
    x, fs = read_wav(path)
    w = wave.open(path, "rb")
    raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    w.close()
    assert fs == 8000 and x.shape == (24360,) and x.dtype == np.float64
    assert np.array_equal(x, raw.astype(np.float64) / 32768.0)
    assert np.array_equal(audioread(path, 8000), x)
    y = audioread(path, 16000)                       # resampling path (parity unpinned: reference uses librosa/soxr)
    assert abs(len(y) - 2 * len(x)) <= 1 and np.isfinite(y).all()


def test_wav_write_rule_is_pinned_by_a_reference_output(golden_dir, tmp_path):
    """A peak-normalised waveform the reference wrote (outputs/clean/output) holds -32768 (x = -1.0):
    scale 32768 with clipping.  Reading it and writing it back must reproduce the sample bytes."""
    import wave
    from speech_separation_amd.wavio import read_wav, write_wav
    src = os.path.join(golden_dir, "ref_output_peaknorm_8k.wav")
    x, fs = read_wav(src)
    assert x.min() == -1.0                           # the peak sample of this file is -32768
    out = str(tmp_path / "again.wav")
    write_wav(out, x, fs)
    a, b = wave.open(src, "rb"), wave.open(out, "rb")
    assert (b.getnchannels(), b.getsampwidth(), b.getframerate(), b.getnframes()) == (1, 2, 8000, a.getnframes())
    assert a.readframes(a.getnframes()) == b.readframes(b.getnframes())
    a.close(); b.close()
    write_wav(out, np.array([1.0, -1.0, 0.5, 2.0, -2.0, 1.5 / 32768.0, 2.5 / 32768.0]), 8000)
    y, _ = read_wav(out)
    assert np.array_equal(y * 32768.0, [32767, -32768, 16384, 32767, -32768, 2, 2])   # clip; round half to even


def test_find_inputs(tmp_path):
    from speech_separation_amd.wavio import find_inputs
    d = tmp_path / "in"
    (d / "sub").mkdir(parents=True)
    for n in ("b.wav", "a.wav", "sub/c.WAV", "x.txt"):
        (d / n).write_bytes(b"")
    got = find_inputs(str(d))
    assert [os.path.relpath(p, d) for p in got] == ["a.wav", "b.wav", "sub/c.WAV"]
    assert find_inputs(str(d / "a.wav")) == [str(d / "a.wav")]
    lst = tmp_path / "list.scp"
    lst.write_text(f"{d / 'a.wav'} extra\n\n{d / 'b.wav'}\n")
    assert find_inputs(str(lst)) == [str(d / "a.wav"), str(d / "b.wav")]


def test_overlay_import_paths_and_reload(tmp_path):
    """`dropin/` in front of sys.path gives the reference's import paths (reference: inference.py:4-8,
    networks.py:9-18): the factory builds the HIP-backed module, reload_for_eval fills it."""
    import sys
    saved_path = list(sys.path)
    saved_mods = set(sys.modules)
    sys.path[:0] = [os.path.join(ROOT, "dropin"), ROOT]
    try:
        from networks import network_wrapper
        from utils.decode import decode_one_audio as overlay_decode
        from utils.misc import reload_for_eval as overlay_reload
        from models.mossformer2.mossformer2 import MossFormer2_SS as OverlaySS
        assert overlay_decode is decode_one_audio and overlay_reload is reload_for_eval and OverlaySS is MossFormer2_SS
        model = network_wrapper(Args()).ss_network
        assert isinstance(model, MossFormerHIP)
        sd = {k: torch.from_numpy(v) for k, v in make_state_dict(TINY, 5).items()}
        torch.save({"model": sd}, tmp_path / "m.pt")
        (tmp_path / "last_checkpoint").write_text("m.pt\n")
        overlay_reload(model, str(tmp_path), 0)
        model.eval()
        got = model.state_dict()
        assert all(torch.equal(got[k], sd[k]) for k in sd)
        a = Args()
        a.network = "FRCRN_SE_16K"
        assert not hasattr(network_wrapper(a), "ss_network")
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved_mods and k.split(".")[0] in ("networks", "utils", "models"):
                del sys.modules[k]


class _Odd:               # not a tensor / container / number: weights_only=True refuses it
    pass


def test_checkpoint_that_needs_unpickling_is_opt_in(tmp_path, monkeypatch):
    """A checkpoint the safe loader refuses is not unpickled silently."""
    import pickle
    from speech_separation_amd.checkpoint import load_checkpoint

    torch.save({"model": {}, "extra": _Odd()}, tmp_path / "odd.pt")
    monkeypatch.delenv("MSF2_TRUST_CHECKPOINT", raising=False)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint(str(tmp_path / "odd.pt"), 0)
    ck = load_checkpoint(str(tmp_path / "odd.pt"), 0, trust=True)
    assert "model" in ck
    monkeypatch.setenv("MSF2_TRUST_CHECKPOINT", "1")
    assert "extra" in load_checkpoint(str(tmp_path / "odd.pt"), 0)
    # what the reference's trainer writes loads with the safe loader
    torch.save({"model": {"w": torch.zeros(3)}, "optimizer": {"state": {}, "param_groups": [{"lr": 1e-3}]}, "epoch": 3,
                "step": 100, "best_val_loss": 1.5, "halving": False}, tmp_path / "ok.pt")
    monkeypatch.delenv("MSF2_TRUST_CHECKPOINT")
    assert load_checkpoint(str(tmp_path / "ok.pt"), 0)["epoch"] == 3


def test_work_items_respect_the_library_batch_limit():
    from speech_separation_amd.parallel import LIBRARY_MAX_SAMPLES, make_work_items
    items = make_work_items([120000] * 100, max_batch=128)
    assert all(n * len(idx) <= LIBRARY_MAX_SAMPLES for n, idx in items)
    assert sorted(i for _, idx in items for i in idx) == list(range(100))
    lib = _lib.load()
    cfg = _lib.Config(512, 24, 16, 2, _lib.PREC_FP32)
    assert lib.msf2_workspace_bytes(cfg, 31, 120000) > 0          # 31 x 15 s = 3.72 M samples: accepted
