"""Interleaved A/B of library builds (BIGSI_HIP_LIB cannot change inside a process, so each arm is a subprocess of
scripts/ab_one.py and rounds alternate arms): median row-AND kernel time per arm."""
import json, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1].split(",")
thr = sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
extra = sys.argv[4:] 
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, BIGSI_HIP_LIB=os.path.join(root, "bigsi_amd", "lib%s.so" % l))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--cpu-seconds", "0", "--no-verify",
                              "--threshold", thr] + extra, env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        res[l].append((d["roofline"]["kernel_ms"], d["ms_per_step"]))
for l in libs:
    a = np.array(res[l])
    print("%-14s K2 median %.4f ms (min %.4f)   step median %.4f ms" % (l, np.median(a[:, 0]), a[:, 0].min(), np.median(a[:, 1])))
