"""A/B of conv launch knobs in ONE process is impossible (the knobs are read once per process), so this runs each setting in a
child process, three rounds interleaved, and prints the median: conv_ab.py LEVEL B XF "VAR:WPS[:STAGGER]" "VAR:WPS[:STAGGER]" ..."""
import os, subprocess, sys, statistics
lvl, B, xf = sys.argv[1:4]
settings = sys.argv[4:]
res = {s: [] for s in settings}
here = os.path.dirname(os.path.abspath(__file__))
for rnd in range(3):
    for s in settings:
        var, wps, *rest = s.split(":")
        env = dict(os.environ)
        if rest and rest[0] != "-": env["DDIMX_CONV_STAGGER"] = rest[0]
        if var != "-": env["DDIMX_CONV_VAR"] = var
        if wps != "-": env["DDIMX_CONV_WPS"] = wps
        out = subprocess.run([sys.executable, os.path.join(here, "conv_time.py"), lvl, B, xf], env=env, capture_output=True, text=True).stdout
        try:
            res[s].append(float(out.split("us/launch")[1].split()[0]))
        except Exception:
            res[s].append(float("nan"))
for s in settings:
    print(f"level {lvl} B {B} xf {xf} var:wps {s:8s} median {statistics.median(res[s]):7.1f} us  all {res[s]}", flush=True)
