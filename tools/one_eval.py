"""Scratch: one warm-up + N timed LML evals (for rocprofv3 traces)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = 8
X, Y, _ = orc.synthetic_gpr_data(n, d, 0)
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
for k, v in [a.split("=") for a in os.environ.get("GPS_OPTS", "").split(",") if a]:       # e.g. GPS_OPTS=potrf_rl_max=2048,potrf_rl_group=3
    gpf.get_handle().set_option(k, float(v))
for i in range(reps + 1):
    t0 = time.perf_counter(); lml = m.compute_log_likelihood(); t1 = time.perf_counter()
    print(i, lml, 1e3 * (t1 - t0), gpf.get_handle().last_stage_ms())
