"""Scratch: like one_eval.py with options: python tools/one_eval_opt.py N reps "k=v,k=v" """
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
n = int(sys.argv[1]); reps = int(sys.argv[2])
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] else []):
    k, v = kv.split("="); gpf.get_handle().set_option(k, float(v))
for i in range(reps + 1):
    t0 = time.perf_counter(); lml = m.compute_log_likelihood(); t1 = time.perf_counter()
    print(i, lml, 1e3 * (t1 - t0), gpf.get_handle().last_stage_ms())
