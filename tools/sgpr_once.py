"""SGPR / FITC bounds at config 5's shape a few times (wall time; under rocprofv3 --kernel-trace --stats: kernel time beside it)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
M, N, d = 4096, 1000000, 8
rng = np.random.default_rng(1)
X = rng.standard_normal((N, d)); Z = X[:M].copy()
Y = np.sin(X @ (rng.standard_normal((d, 1)) / np.sqrt(d))) + 0.1 * rng.standard_normal((N, 1))
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
for name, cls in (("SGPR", gpf.models.SGPR), ("GPRFITC", gpf.models.GPRFITC)):
    m = cls(X, Y, kern, Z=Z)
    for i in range(3):
        t0 = time.perf_counter(); v = m.compute_log_likelihood(); t1 = time.perf_counter()
        print("%s call %d: %.1f ms value %.6f" % (name, i, 1e3 * (t1 - t0), v), flush=True)
