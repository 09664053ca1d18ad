import os, sys
ROOT = "/root/repo"
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
rng = np.random.default_rng(0)
n, d = 512, 8
X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, lengthscales=np.sqrt(d) * np.ones(d), ARD=True), obs_var=0.1)
for i in range(3):
    print(m.compute_log_likelihood())
