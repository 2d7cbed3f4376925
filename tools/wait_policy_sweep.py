#!/usr/bin/env python3
"""Dev tool: ms per step and host CPU time of the default bench under each GPU wait policy (GK_WAIT_POLICY)."""
import json, os, resource, subprocess, sys, time
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for pol in ("default", "spin", "yield", "block"):
    env = dict(os.environ)
    env.pop("GK_WAIT_POLICY", None)
    if pol != "default":
        env["GK_WAIT_POLICY"] = pol
    before = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(here, "bench.py"), "--cpu-pairs", "0", "--serial-steps", "0", "--steps", "48"],
                         env=env, capture_output=True, text=True)
    wall = time.time() - t0
    after = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (after.ru_utime - before.ru_utime) + (after.ru_stime - before.ru_stime)
    try:
        ms = json.loads(out.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception:      # noqa: BLE001
        ms = float("nan")
        print(out.stderr[-500:])
    print(f"{pol:8s} {ms:7.2f} ms per step; whole run: {cpu:6.1f} core-s over {wall:5.1f} s of wall", flush=True)
print(open("/proc/loadavg").read().strip())
