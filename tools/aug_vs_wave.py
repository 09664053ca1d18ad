"""Scratch: alpha through the augmented rows of the factorisation vs the wavefront substitution afterwards."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
for n in [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 12288]:
    d = 8
    X, Y, _ = orc.synthetic_gpr_data(n, d, 0)
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    res = {}
    for rep in range(3):
        for name, aug, wave in (("aug", 1, 1), ("wave", 0, 1), ("recursive", 0, 0)):
            h.set_option("gpr_aug_rows", aug); h.set_option("trsv_wave", wave)
            m.compute_log_likelihood()
            ts = []
            for _ in range(10):
                t0 = time.perf_counter(); m.compute_log_likelihood(); ts.append(time.perf_counter() - t0)
            res.setdefault(name, []).append(round(1e3 * float(np.median(ts)), 3))
    print(n, res, flush=True)
h.set_option("gpr_aug_rows", -1); h.set_option("trsv_wave", 1)
