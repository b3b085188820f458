"""``python -m ddim_audio_amd.dropin /path/to/ddim-audio/main.py <main.py's flags>``: the reference's entry script, unchanged,
with the drop-in's import hook installed first (the alternative to putting this directory on ``PYTHONPATH``)."""
import os
import runpy
import sys

from . import _ddimx_hook


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m ddim_audio_amd.dropin /path/to/ddim-audio/main.py [flags of main.py]")
    script = os.path.abspath(sys.argv[1])
    _ddimx_hook.install()
    sys.argv = [script] + sys.argv[2:]
    sys.path[0] = os.path.dirname(script)  # what `python main.py` would have put there
    runpy.run_path(script, run_name="__main__")


main()
