"""Recipe A of INTEGRATION.md resolved the way ``python main.py`` resolves imports.

A CHILD interpreter is started with the reference checkout as working directory and ``sys.path[0]`` -- in front of
``PYTHONPATH``, exactly as ``python main.py`` arranges it -- and only ``ddim_audio_amd/dropin`` on ``PYTHONPATH``.  The
drop-in's ``sitecustomize`` must have installed the import hook before the first line of user code runs; ``main.py``'s module
level is then executed (its imports pull in ``runners.diffusion``, which imports ``functions`` / ``models`` the reference's
own way, ``runners/diffusion.py:12-15``) and the seven symbols of the hot path must come from ``ddim_audio_amd``.  Test-side
stand-ins only for what the container lacks (tensorboard; the un-vendored SST / UPU submodules).  The reference never
travels: skipped when ``/root/reference`` is absent (the GPU box).
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
DROPIN = os.path.join(REPO, "ddim_audio_amd", "dropin")

_CHILD = textwrap.dedent("""
    import inspect, json, os, runpy, sys, types
    report = {"path0": sys.path[0], "hook_at_startup": any(type(f).__name__ == "DropinFinder" for f in sys.meta_path)}

    def stub(name, **attrs):
        mod = types.ModuleType(name)
        mod.__dict__.update(attrs)
        mod.__path__ = []
        sys.modules[name] = mod
        return mod

    nothing = lambda *a, **k: None
    stub("torch.utils.tensorboard", SummaryWriter=object)                  # main.py:9, not installed here
    for n in ("SST", "SST.utils", "UPU", "UPU.signal", "UPU.layers", "UPU.layers.normalize"):
        stub(n)
    stub("SST.utils.wav2img", limit_length_img=nothing, pfft2img=nothing, pfft2wav=nothing)   # runners/diffusion.py:19
    sys.modules["SST.utils"].AudioDataset = object                                             # datasets/__init__.py:9
    stub("UPU.signal.denoise", denoise_2d=nothing)                                              # runners/diffusion.py:20
    stub("UPU.layers.normalize.groupnorm", GroupNorm1D=object)

    ns = runpy.run_path("main.py", run_name="__ddimx_probe__")   # module level of main.py: every import, not main()
    rd = sys.modules["runners.diffusion"]
    from functions.denoising import generalized_steps, ddpm_steps   # runners/diffusion.py:495,515 (function-local imports)
    syms = {"Model": rd.Model, "EMAHelper": rd.EMAHelper, "get_optimizer": rd.get_optimizer, "get_scheduler": rd.get_scheduler,
            "loss_simple": rd.loss_registry["simple"], "generalized_steps": generalized_steps, "ddpm_steps": ddpm_steps}
    report["modules"] = {k: v.__module__ for k, v in syms.items()}
    report["files"] = {n: os.path.abspath(sys.modules[n].__file__) for n in
                       ("functions", "functions.losses", "functions.denoising", "models.diffusion", "models.ema", "runners.diffusion",
                        "datasets", "utils")}
    report["Diffusion_from"] = ns["Diffusion"].__module__
    # the call shapes of Diffusion.sample_image (runners/diffusion.py:497-499,517) must bind
    x, seq, model, tab = object(), [0, 10], object(), object()
    inspect.signature(generalized_steps).bind(x, seq, model, tab, eta=0.0, select_index=None)
    inspect.signature(ddpm_steps).bind(x, seq, model, tab, select_index=None)
    report["bind"] = True
    print("REPORT " + json.dumps(report))
""")


def _run_child(args, extra_env, cwd=REFERENCE):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env.update(extra_env)
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("REPORT ")][-1]
    return json.loads(line[len("REPORT "):])


def _check(rep):
    assert rep["hook_at_startup"], "sitecustomize did not install the hook before user code"
    for name, mod in rep["modules"].items():
        assert mod.startswith("ddim_audio_amd."), f"{name} resolved to {mod}: the reference's own module won"
    for n in ("functions", "functions.losses", "functions.denoising", "models.diffusion", "models.ema"):
        assert rep["files"][n].startswith(DROPIN + os.sep), (n, rep["files"][n])
    # everything that is not the hot path stays the reference's
    for n in ("runners.diffusion", "datasets", "utils"):
        assert rep["files"][n].startswith(REFERENCE + os.sep), (n, rep["files"][n])
    assert rep["Diffusion_from"] == "runners.diffusion" and rep["bind"]


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (never travels to the GPU box)")
def test_python_main_py_resolves_the_hot_path_to_this_build():
    # `python -c` with cwd = the checkout: sys.path[0] == '' == the checkout, in front of PYTHONPATH, as for `python main.py`
    rep = _run_child(["-c", _CHILD], {"PYTHONPATH": DROPIN})
    assert rep["path0"] in ("", REFERENCE)
    _check(rep)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (never travels to the GPU box)")
def test_without_the_hook_the_reference_shadows_functions():
    """The round-2 recipe (shadow directory on PYTHONPATH, no hook) is what this replaces: prove the failure mode is real, so
    the test above cannot pass vacuously.  -S skips `site`, i.e. no sitecustomize."""
    child = _CHILD
    import site
    pp = os.pathsep.join([DROPIN, REPO] + [p for p in site.getsitepackages() + [site.getusersitepackages()] if os.path.isdir(p)])
    rep = _run_child(["-S", "-c", child], {"PYTHONPATH": pp})
    assert not rep["hook_at_startup"]
    assert rep["modules"]["get_optimizer"] == "functions" and rep["modules"]["loss_simple"] == "functions.losses"
    assert rep["modules"]["Model"].startswith("ddim_audio_amd.")  # models/ has no __init__.py upstream: that one did resolve


def test_launcher_module_installs_the_hook(tmp_path):
    """`python -m ddim_audio_amd.dropin <script>`: the hook is in place and sys.path[0] is the script's directory."""
    script = tmp_path / "probe.py"
    script.write_text("import sys, json\nimport functions, models.diffusion\n"
                      "print('REPORT ' + json.dumps({'f': functions.get_optimizer.__module__, 'm': models.diffusion.Model.__module__,"
                      " 'p0': sys.path[0], 'argv': sys.argv[1:]}))\n")
    rep = _run_child(["-m", "ddim_audio_amd.dropin", str(script), "--flag", "1"], {"PYTHONPATH": REPO}, cwd=str(tmp_path))
    assert rep["f"].startswith("ddim_audio_amd.") and rep["m"].startswith("ddim_audio_amd.")
    assert rep["p0"] == str(tmp_path) and rep["argv"] == ["--flag", "1"]
