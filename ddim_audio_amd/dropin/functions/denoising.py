"""Drop-in for the reference's ``functions/denoising.py`` (``generalized_steps``; ``ddpm_steps`` is a
later scope row and raises)."""
from ddim_audio_amd.sampler import generalized_steps  # noqa: F401


def ddpm_steps(x, seq, model, b, select_index, **kwargs):
    raise NotImplementedError("ddpm_steps is not built yet (SURVEY section 8f row 2)")
