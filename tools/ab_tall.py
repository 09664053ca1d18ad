"""Config 5's conditional(): 512-column panels left-looking (trsm_tall_ratio = 16) against the recursive halving (0), alternating."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
h = gpf.get_handle()
M, N, d = 4096, 1000000, 8
rng = np.random.default_rng(1)
X = rng.standard_normal((N, d)); Z = X[:M].copy()
f = rng.standard_normal((M, 1))
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
res = {}
for rep in range(4):
    for ratio in (16, 0):
        h.set_option("trsm_tall_ratio", ratio)
        t0 = time.perf_counter(); mu, var = gpf.conditionals.conditional(X, Z, kern, f, white=True); t1 = time.perf_counter()
        res.setdefault(ratio, []).append((1e3 * (t1 - t0), mu, var))
a, b = res[16][-1], res[0][-1]
print("left-looking %s ms | recursive %s ms | max |mean diff| %.2e max |var diff| %.2e" % (
    ["%.1f" % t for t, _, _ in res[16]], ["%.1f" % t for t, _, _ in res[0]], np.abs(a[1] - b[1]).max(), np.abs(a[2] - b[2]).max()))
