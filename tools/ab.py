"""Scratch: A/B an option inside one process (same box, same clocks).  usage: ab.py option v0 v1 [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
opt, v0, v1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
for n in [int(a) for a in sys.argv[4:]] or [2048, 8192, 32768]:
    X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
    kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    m.compute_log_likelihood()
    res = {v0: [], v1: []}
    vals = {}
    for rep in range(4):
        for v in (v0, v1):
            h.set_option(opt, v)
            vals[v] = m.compute_log_likelihood()
            res[v].append(h.last_stage_ms()["potrf"])
    print("N=%d %s=%g: potrf %.3f ms | %s=%g: potrf %.3f ms | lml diff %.3e" % (n, opt, v0, min(res[v0]), opt, v1, min(res[v1]), vals[v0] - vals[v1]))
