"""Config 5's SVGP bound a few times (for rocprofv3 --kernel-trace --stats): python tools/svgp_once.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
h = gpf.get_handle()
M, N, d = 4096, 1000000, 8
rng = np.random.default_rng(1)
X = rng.standard_normal((N, d)); Z = X[:M].copy()
Y = np.sin(X @ (rng.standard_normal((d, 1)) / np.sqrt(d))) + 0.1 * rng.standard_normal((N, 1))
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
q_mu = rng.standard_normal((M, 1)) * 0.3
q_sqrt = (np.tril(rng.standard_normal((M, M))) * (0.5 / M) + 0.5 * np.eye(M))[:, :, None]
sv = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.1), Z=Z, whiten=True)
sv._q_mu.assign(q_mu); sv._q_sqrt.assign(q_sqrt)
for i in range(4):
    t0 = time.perf_counter(); v = sv.compute_log_likelihood(); t1 = time.perf_counter()
    print("call %d: %.1f ms elbo %.6f" % (i, 1e3 * (t1 - t0), v))
if "grad" in sys.argv:
    for i in range(3):
        t0 = time.perf_counter(); v, g = sv.compute_log_likelihood_and_gradients(); t1 = time.perf_counter()
        print("bound + gradient call %d: %.1f ms" % (i, 1e3 * (t1 - t0)))
    sys.exit(0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); sv.compute_log_likelihood(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
