"""Drop-in for the reference's ``models/diffusion.py``: put ``ddim_audio_amd/dropin`` ahead of the
reference checkout on PYTHONPATH and ``from models.diffusion import Model`` resolves here."""
from ddim_audio_amd.model import Model  # noqa: F401
