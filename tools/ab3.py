"""Scratch: compare several values of one option inside one process."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
opt = sys.argv[1]; vals = [float(v) for v in sys.argv[2].split(",")]
for n in [int(a) for a in sys.argv[3:]]:
    X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
    kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    m.compute_log_likelihood()
    res = {v: [] for v in vals}; lm = {}
    for rep in range(4):
        for v in vals:
            h.set_option(opt, v); lm[v] = m.compute_log_likelihood(); res[v].append(h.last_stage_ms()["potrf"])
    print("N=%d " % n + " | ".join("%s=%g: %.3f ms" % (opt, v, min(res[v])) for v in vals) + " | lml spread %.2e" % (max(lm.values()) - min(lm.values())))
