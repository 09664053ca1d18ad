"""Scratch: one factorisation with the one-launch sweep steps, counters printed after every launch (GPS_STEP_DEBUG=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(0)
X = rng.standard_normal((n, 8)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
h = gpf.get_handle()
m = gpf.models.GPR(X, Y, gpf.kernels.RBF(8, lengthscales=np.sqrt(8.0) * np.ones(8), ARD=True), obs_var=0.1)
h.set_option("potrf_fused_step", 0)
ref = m.compute_log_likelihood()
h.set_option("potrf_fused_step", mode)
try:
    got = m.compute_log_likelihood()
    print("fused", got, "ref", ref, "rel", abs(got - ref) / abs(ref), "retries", h.profile_get("lookahead_retries"))
except Exception as e:
    print("FAILED", e)
