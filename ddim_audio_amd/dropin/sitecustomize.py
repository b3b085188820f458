"""Runs at interpreter start-up when this directory is on ``PYTHONPATH`` (CPython's ``site`` imports the first
``sitecustomize`` it finds): installs the drop-in's import hook, so that the UNCHANGED ``python main.py ...`` of the reference
resolves ``functions.*`` / ``models.*`` to this build even though the checkout precedes ``PYTHONPATH`` on ``sys.path``
(recipe A).  A ``sitecustomize`` further down the path (distributions ship one) still runs afterwards."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))


def _load(name, file):
    spec = importlib.util.spec_from_file_location(name, file)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_load("_ddimx_hook", os.path.join(_here, "_ddimx_hook.py")).install()

for _d in sys.path:  # chain to the sitecustomize this one shadows
    _f = os.path.join(_d or os.getcwd(), "sitecustomize.py")
    if os.path.isfile(_f) and os.path.abspath(os.path.dirname(_f)) != _here:
        try:
            _load("_shadowed_sitecustomize", _f)
        except Exception:  # like site.py: a failing sitecustomize must not stop the interpreter
            pass
        break
