"""``models.mossformer2.mossformer2`` at the reference's import path
(reference: models/mossformer2/mossformer2.py:739-757): the wrapper class the
model factory instantiates, backed by the HIP implementation."""
from speech_separation_amd.model import MossFormer2_SS, MossFormerHIP as MossFormer  # noqa: F401
