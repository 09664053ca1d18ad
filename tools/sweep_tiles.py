"""Scratch: sweep the GEMM tile-selection threshold."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
for n in [int(a) for a in sys.argv[1:]] or [8192, 32768]:
    d = 8
    X, Y, _ = orc.synthetic_gpr_data(n, d, 0)
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    m.compute_log_likelihood()
    for thr in [96, 192, 384, 520, 768, 1024, 1536, 2100, 4200]:
        h.set_option("gemm_min_tiles", thr)
        best = 1e9
        for rep in range(2):
            m.compute_log_likelihood()
            best = min(best, h.last_stage_ms()["potrf"])
        print("N=%d gemm_min_tiles=%5d potrf=%.2f ms  %.2f TFLOP/s" % (n, thr, best, n ** 3 / 3 / best / 1e9))
    h.set_option("gemm_min_tiles", 192)
