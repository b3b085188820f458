"""downup_ab.py {down|up} LEVEL B "VAR:WPS" ...  -- each setting in a child process, three interleaved rounds, median."""
import os, subprocess, sys, statistics
kind, lvl, B = sys.argv[1:4]
settings = sys.argv[4:]
res = {s: [] for s in settings}
here = os.path.dirname(os.path.abspath(__file__))
for rnd in range(3):
    for s in settings:
        var, wps = s.split(":")
        env = dict(os.environ)
        if var != "-": env["DDIMX_CONV_VAR"] = var
        if wps != "-": env["DDIMX_CONV_WPS"] = wps
        out = subprocess.run([sys.executable, os.path.join(here, "downup_time.py"), kind, lvl, B], env=env, capture_output=True, text=True).stdout
        try: res[s].append(float(out.split("us/launch")[1].split()[0]))
        except Exception: res[s].append(float("nan"))
for s in settings:
    print(f"{kind} level {lvl} B {B} var:wps {s:8s} median {statistics.median(res[s]):7.1f} us  all {res[s]}", flush=True)
