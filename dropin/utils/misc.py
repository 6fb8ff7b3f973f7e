"""``utils.misc`` at the reference's import path: the two functions the
inference entry points use (reference: utils/misc.py:45-52, :59-97)."""
from speech_separation_amd.checkpoint import load_checkpoint, reload_for_eval  # noqa: F401
