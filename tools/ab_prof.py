"""Scratch: per-class profile of one evaluation for two values of an option."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
opt = sys.argv[1]; vals = [float(v) for v in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
m.compute_log_likelihood()
for rep in range(2):
    for v in vals:
        h.set_option(opt, v)
        m.compute_log_likelihood()
        h.profile_reset(); h.profile_enable(True)
        m.compute_log_likelihood()
        h.profile_enable(False)
        p = h.profile_get("gemm_f64")
        print("%s=%g: gemm class %.2f ms (%d launches) %.2f TFLOP/s; potrf stage %.2f ms" % (opt, v, p["ms"], p["launches"], p["flops"] / p["ms"] / 1e9, h.last_stage_ms()["potrf"]))
