"""Drop-in for the reference's ``functions/__init__.py``."""
from ddim_audio_amd.optim import get_optimizer, get_scheduler  # noqa: F401
