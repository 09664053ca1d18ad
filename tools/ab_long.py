"""Scratch: long alternating A/B of one option at N = 32768 (prints every sample)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
opt = sys.argv[1]; vals = [float(v) for v in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
m.compute_log_likelihood()
res = {v: [] for v in vals}
for rep in range(reps):
    for v in vals:
        h.set_option(opt, v); m.compute_log_likelihood(); res[v].append(h.last_stage_ms()["potrf"])
for v in vals:
    print("%s=%g:" % (opt, v), " ".join("%.1f" % t for t in res[v]), "| min %.2f median %.2f" % (min(res[v]), float(np.median(res[v]))))
