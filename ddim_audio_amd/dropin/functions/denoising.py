"""Drop-in for the reference's ``functions/denoising.py``."""
from ddim_audio_amd.sampler import ddpm_steps, generalized_steps  # noqa: F401
