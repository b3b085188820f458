"""Import hook of the drop-in: makes the reference's own ``import`` statements resolve to this build.

The reference imports the path's symbols as top-level modules of its checkout -- ``from functions import get_optimizer,
get_scheduler``, ``from functions.losses import loss_registry``, ``from models.diffusion import Model``, ``from models.ema
import EMAHelper`` (``runners/diffusion.py:12-15``), ``from functions.denoising import generalized_steps`` / ``ddpm_steps``
(``runners/diffusion.py:495,515``) -- and ``python main.py`` puts the checkout (``sys.path[0]``) in front of ``PYTHONPATH``, so a
shadow directory on ``PYTHONPATH`` loses against the checkout's own ``functions/`` package.  A meta-path finder in front of
the path-based one does not: ``install()`` maps exactly those six module names onto the files next to this one, whatever
``sys.path`` says.  Nothing else of the reference (``runners``, ``datasets``, ``main``, ``utils``) is touched.

Installed by ``sitecustomize.py`` in this directory (recipe A of INTEGRATION.md: this directory on ``PYTHONPATH``, then the
unchanged ``python main.py ...``) or by ``python -m ddim_audio_amd.dropin main.py ...``.
"""
import importlib.abc
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))

# reference module name -> (file, is_package)
MODULES = {
    "functions": (os.path.join(_HERE, "functions", "__init__.py"), True),
    "functions.denoising": (os.path.join(_HERE, "functions", "denoising.py"), False),
    "functions.losses": (os.path.join(_HERE, "functions", "losses.py"), False),
    "models": (os.path.join(_HERE, "models", "__init__.py"), True),
    "models.diffusion": (os.path.join(_HERE, "models", "diffusion.py"), False),
    "models.ema": (os.path.join(_HERE, "models", "ema.py"), False),
}


class DropinFinder(importlib.abc.MetaPathFinder):
    """Resolves the six module names of the hot path to this directory, ahead of ``sys.path``."""

    def find_spec(self, fullname, path=None, target=None):
        hit = MODULES.get(fullname)
        if hit is None:
            return None
        file, is_pkg = hit
        return importlib.util.spec_from_file_location(
            fullname, file, submodule_search_locations=[os.path.dirname(file)] if is_pkg else None)


def install():
    """Idempotent.  Also makes ``ddim_audio_amd`` importable when only this directory was put on ``PYTHONPATH``."""
    if not any(isinstance(f, DropinFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, DropinFinder())
    if importlib.util.find_spec("ddim_audio_amd") is None and _REPO not in sys.path:
        sys.path.append(_REPO)
    for name in MODULES:  # a module of the reference imported before the hook existed would stay in place
        mod = sys.modules.get(name)
        if mod is not None and os.path.abspath(getattr(mod, "__file__", "") or "") != MODULES[name][0]:
            del sys.modules[name]
