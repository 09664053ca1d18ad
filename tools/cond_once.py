"""Config 5's conditional() a few times in one leaf mode (for rocprofv3 --kernel-trace --stats): python tools/cond_once.py [leaf_refine]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
h = gpf.get_handle()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else -1
M, N, d = 4096, 1000000, 8
rng = np.random.default_rng(1)
X = rng.standard_normal((N, d)); Z = X[:M].copy()
f = rng.standard_normal((M, 1))
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
h.set_option("leaf_refine", mode)
for i in range(4):
    t0 = time.perf_counter(); gpf.conditionals.conditional(X, Z, kern, f, white=True); t1 = time.perf_counter()
    print("leaf_refine=%d call %d: %.1f ms" % (mode, i, 1e3 * (t1 - t0)), h.last_stage_ms() if hasattr(h, "last_stage_ms") else "")
