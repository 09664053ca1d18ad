"""Scratch: same-process A/B of handle options on the LML evaluation (alternating, best-of and mean)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
n = int(sys.argv[1]); key = sys.argv[2]; vals = [float(v) for v in sys.argv[3].split(",")]; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
h = gpf.get_handle()
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
m = gpf.models.GPR(X, Y, gpf.kernels.RBF(8, lengthscales=np.sqrt(8) * np.ones(8), ARD=True), obs_var=0.1)
ts = {v: [] for v in vals}
for v in vals:
    h.set_option(key, v); m.compute_log_likelihood()
for rep in range(reps):
    for v in vals:
        h.set_option(key, v)
        t0 = time.perf_counter(); lml = m.compute_log_likelihood(); ts[v].append(1e3 * (time.perf_counter() - t0))
for v in vals:
    print("N=%d %s=%g: best %.3f ms mean %.3f ms (lml %.12g)" % (n, key, v, min(ts[v]), np.mean(ts[v]), lml), flush=True)
