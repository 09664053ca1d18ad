"""Two likelihood evaluations of BASELINE config 4 (Matern-5/2 + Periodic, N = 16384, D = 16) for counter collection on the
kernel-matrix kernel: python tools/kmat_once.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = 16
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
k = gpf.kernels
kern = k.Matern52(d, lengthscales=4 * np.ones(d), ARD=True) + k.Periodic(d, period=2.0, lengthscales=1.0)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
for i in range(2):
    print(m.compute_log_likelihood(), gpf.get_handle().last_stage_ms())
