"""ddim_audio_amd -- MI355X-native DDIM denoising hot path behind the reference's Python API.

Public names mirror the reference's modules for this path (klae01/ddim-audio):
``Model`` (models/diffusion.py), ``generalized_steps`` (functions/denoising.py),
``noise_estimation_loss`` (functions/losses.py), ``EMAHelper`` (models/ema.py), plus the schedule
helpers of runners/diffusion.py.  All tensor arithmetic runs in ``libddimx.so`` (hand-written HIP for
gfx950, C ABI in ``include/ddimx.h``); nothing here falls back to CPU or eager PyTorch.
"""
from . import configs, schedule, synth  # noqa: F401  (pure host logic, importable without the library)


def __getattr__(name):
    if name == "Model":
        from .model import Model
        return Model
    if name in ("generalized_steps", "ddpm_steps"):
        from . import sampler
        return getattr(sampler, name)
    if name in ("noise_estimation_loss", "loss_registry"):
        from . import losses
        return getattr(losses, name)
    if name == "EMAHelper":
        from .ema import EMAHelper
        return EMAHelper
    raise AttributeError(name)
