"""A/B of the kernel-matrix build with the feature products on the matrix pipe ("kmat_mfma" 1 / 0): kmat stage time of one LML
evaluation for the headline (RBF-ARD, N = 32768, D = 8), config 4 (Matern-5/2 + Periodic, N = 16384, D = 16) and Matern-5/2 alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf

rng = np.random.default_rng(0)
h = gpf.get_handle()
def case(name, n, d, kern):
    X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    out = {}
    for mode in (2, 0, 2, 0):
        h.set_option("kmat_mfma", mode)
        best, lml = 1e9, None
        for i in range(4):
            lml = m.compute_log_likelihood()
            best = min(best, h.last_stage_ms()["kmat"])
        out.setdefault(1 if mode else 0, []).append((best, lml))
    h.set_option("kmat_mfma", 1)
    npad = ((n + 127) // 128) * 128
    gb = 4.0 * npad * npad / 1e9
    t1 = min(t for t, _ in out[1]); t0 = min(t for t, _ in out[0])
    print("%-28s N=%d D=%d: mfma %.3f ms (%.2f TB/s) | valu %.3f ms (%.2f TB/s) | lml rel diff %.2e" % (
        name, n, d, t1, gb / t1, t0, gb / t0, abs(out[1][0][1] - out[0][0][1]) / abs(out[0][0][1])), flush=True)
k = gpf.kernels
case("rbf_ard (headline)", 32768, 8, k.RBF(8, lengthscales=np.sqrt(8) * np.ones(8), ARD=True))
case("matern52_ard", 16384, 16, k.Matern52(16, lengthscales=4 * np.ones(16), ARD=True))
case("cfg4 matern52 + periodic", 16384, 16, k.Matern52(16, lengthscales=4 * np.ones(16), ARD=True) + k.Periodic(16, period=2.0, lengthscales=1.0))
case("rbf * periodic + const", 8192, 4, k.RBF(4, lengthscales=1.5) * k.Periodic(4, period=3.0) + 0.3)
