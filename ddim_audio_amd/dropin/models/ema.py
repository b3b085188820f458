"""Drop-in for the reference's ``models/ema.py``."""
from ddim_audio_amd.ema import EMAHelper  # noqa: F401
