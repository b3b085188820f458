"""Drop-in layer: the reference's module paths (``functions``, ``functions.denoising``, ``functions.losses``,
``models.diffusion``, ``models.ema``) re-exporting this build, plus the import hook that makes them win over the reference
checkout's own packages (``_ddimx_hook``).  ``python -m ddim_audio_amd.dropin main.py ...`` = ``python main.py ...`` with the
hook installed first."""
