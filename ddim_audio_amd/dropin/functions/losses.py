"""Drop-in for the reference's ``functions/losses.py``."""
from ddim_audio_amd.losses import loss_registry, noise_estimation_loss  # noqa: F401
