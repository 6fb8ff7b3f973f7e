"""The decode harness on the HIP model against what the reference's own
``decode_one_audio`` produced (reference: utils/decode.py:25-83; fixtures
written by oracle/gen_golden.py from the imported reference: 5 input lengths in
one pass, and a 20001-sample input through the segmented path, which here runs
its 1-second windows as ONE batch).  Also once through the import-path overlay
in the reference's call order (reference: inference.py:21-41).
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speech_separation_amd.config import TINY  # noqa: E402
from speech_separation_amd.decode import decode_one_audio  # noqa: E402
from speech_separation_amd.model import MossFormerHIP  # noqa: E402
from speech_separation_amd.synth import make_state_dict  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
LENGTHS = (4000, 8000, 10000, 28400, 32000)
TOL = 2e-4          # on peak-normalised waveforms (peak 1.0)


def tiny_args(**kw):
    a = SimpleNamespace(network="MossFormer2_SS_8K", sampling_rate=8000, decode_window=1, one_time_decode_length=60,
                        num_spks=2, encoder_kernel_size=16, encoder_embedding_dim=TINY.dim,
                        mossformer_sequence_dim=TINY.dim, num_mossformer_layer=TINY.num_layers)
    a.__dict__.update(kw)
    return a


@pytest.fixture(scope="module")
def tiny_model():
    m = MossFormerHIP(TINY, compute_device=DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(TINY, 0).items()})
    return m.eval()


def check_against_fixture(decode, model, device, z):
    for T in LENGTHS:
        out = decode(model, device, z[f"in_{T}"], tiny_args())
        got = np.stack(out, 0)
        ref = z[f"out_{T}"]
        assert got.dtype == np.float32 and got.shape == ref.shape, T
        assert np.abs(got).max() == pytest.approx(1.0, abs=1e-6)
        assert np.abs(got - ref).max() <= TOL, T
    # segmented path: 20001 samples with a 2-second one-pass limit -> 4 windows of 8000, padded length 28002
    out = decode(model, device, z["in_seg_20001"], tiny_args(one_time_decode_length=2))
    got = np.stack(out, 0)
    assert got.dtype == np.float64 and got.shape == (2, 28002)
    assert np.abs(got - z["out_seg_20001"]).max() <= TOL


def test_decode_harness_matches_reference_fixture(tiny_model, golden_dir):
    z = np.load(os.path.join(golden_dir, "decode_tiny.npz"))
    check_against_fixture(decode_one_audio, tiny_model, torch.device(DEV), z)
    # the Flask entry point passes device='cpu' and a torch tensor (reference: app.py:166-170, :269)
    out = decode_one_audio(tiny_model, torch.device("cpu"), torch.from_numpy(z["in_10000"]), tiny_args())
    assert np.abs(np.stack(out, 0) - z["out_10000"]).max() <= TOL


def test_segmented_windows_in_chunks_equal_one_batch(tiny_model, golden_dir):
    """The windows are independent utterances: decoding them 3 at a time or all at once gives the same result
    (no statistic of the HIP path may couple the rows of a batch)."""
    from speech_separation_amd.decode import decode_one_audio_mossformer2_ss
    z = np.load(os.path.join(golden_dir, "decode_tiny.npz"))
    a = tiny_args(one_time_decode_length=2)
    x = np.concatenate([z["in_seg_20001"], z["in_32000"]], 1)           # 52001 samples -> 9 windows
    whole = np.stack(decode_one_audio_mossformer2_ss(tiny_model, torch.device(DEV), x, a), 0)
    parts = np.stack(decode_one_audio_mossformer2_ss(tiny_model, torch.device(DEV), x, a, max_windows_per_call=3), 0)
    one = np.stack(decode_one_audio_mossformer2_ss(tiny_model, torch.device(DEV), x, a, max_windows_per_call=1), 0)
    assert whole.shape == parts.shape == one.shape
    assert np.abs(whole - one).max() <= 2e-6
    assert np.abs(parts - one).max() <= 2e-6


def test_overlay_in_reference_call_order(golden_dir, tmp_path):
    """`dropin/` in front of sys.path, then the reference's own sequence: network_wrapper(args).ss_network,
    .to(device), reload_for_eval, .eval(), decode_one_audio."""
    z = np.load(os.path.join(golden_dir, "decode_tiny.npz"))
    saved_path = list(sys.path)
    saved_mods = {k for k in sys.modules}
    sys.path[:0] = [os.path.join(ROOT, "dropin"), ROOT]
    try:
        from networks import network_wrapper
        from utils.decode import decode_one_audio as overlay_decode
        from utils.misc import reload_for_eval
        args = tiny_args(checkpoint_dir=str(tmp_path), use_cuda=1)
        # a checkpoint as the reference's trainer writes it: weights under 'model', pointer file beside it
        sd = {k: torch.from_numpy(v) for k, v in make_state_dict(TINY, 0).items()}
        torch.save({"model": sd, "epoch": 1}, tmp_path / "model.ckpt-1-10.pt")
        (tmp_path / "last_best_checkpoint").write_text("model.ckpt-1-10.pt\n")
        device = torch.device("cuda")
        model = network_wrapper(args).ss_network
        model.to(device)
        # start from other weights so that the reload is what makes the outputs match
        model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(TINY, 5).items()})
        reload_for_eval(model, args.checkpoint_dir, args.use_cuda)
        model.eval()
        with torch.no_grad():
            check_against_fixture(overlay_decode, model, device, z)
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved_mods and k.split(".")[0] in ("networks", "utils", "models"):
                del sys.modules[k]
