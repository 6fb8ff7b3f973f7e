"""``utils.decode`` at the reference's import path (reference: utils/decode.py:18-83)."""
from speech_separation_amd.decode import decode_one_audio, decode_one_audio_mossformer2_ss  # noqa: F401
