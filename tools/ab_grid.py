"""Scratch: grid of two options inside one process."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
o1, v1 = sys.argv[1], [float(v) for v in sys.argv[2].split(",")]
o2, v2 = sys.argv[3], [float(v) for v in sys.argv[4].split(",")]
n = 32768
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
m.compute_log_likelihood()
res = {}
for rep in range(4):
    for a in v1:
        for b in v2:
            h.set_option(o1, a); h.set_option(o2, b); m.compute_log_likelihood()
            res.setdefault((a, b), []).append(h.last_stage_ms()["potrf"])
for k, v in res.items():
    print("%s=%g %s=%g: median %.2f ms" % (o1, k[0], o2, k[1], float(np.median(v))))
