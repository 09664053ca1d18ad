"""LML + gradient at N = 32768 a few times (wall time; under rocprofv3 --kernel-trace --stats: kernel time beside it)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
import oracle.gp_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d = 8
X, Y, _ = orc.synthetic_gpr_data(n, d, 0)
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
for i in range(4):
    t0 = time.perf_counter(); v, g = m.compute_log_likelihood_and_gradients(); t1 = time.perf_counter()
    print("call %d: %.1f ms lml %.6f" % (i, 1e3 * (t1 - t0), v), flush=True)
