"""Scratch: BASELINE.json configs 4 and 5 at full size (timing + size-independent sanity)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
which = sys.argv[1] if len(sys.argv) > 1 else "4"
if which == "4":
    n, d = 16384, 16
    X, Y, Xs = orc.synthetic_gpr_data(n, d, 1024)
    kern = gpf.kernels.Matern52(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True) + gpf.kernels.Periodic(d, period=2.0, variance=1.0, lengthscales=1.0)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    for i in range(3):
        t0 = time.perf_counter(); lml = m.compute_log_likelihood(); t1 = time.perf_counter()
        print("cfg4 lml", lml, "ms", 1e3 * (t1 - t0), h.last_stage_ms())
    t0 = time.perf_counter(); mu, var = m.predict_f(Xs); t1 = time.perf_counter()
    print("cfg4 predict cold ms", 1e3 * (t1 - t0), float(var.min()), float(var.max()))
else:
    M, N, d = 4096, int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000, 8
    rng = np.random.default_rng(1)
    X = rng.standard_normal((N, d)); Z = X[:M].copy()
    f = rng.standard_normal((M, 1))
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
    for white in (True, False):
        for i in range(2):
            t0 = time.perf_counter()
            mu, var = gpf.conditionals.conditional(X, Z, kern, f, white=white)
            t1 = time.perf_counter()
            print("cfg5 white=%s N=%d M=%d: %.1f ms; var range %.3g..%.3g; TFLOP/s(trsm) %.1f" % (white, N, M, 1e3 * (t1 - t0), var.min(), var.max(), M * M * N / (t1 - t0) / 1e12))
    # spot parity on a slice against the oracle
    idx = rng.choice(N, 200, replace=False)
    spec = {"type": "rbf", "variance": orc.constrained(1.0), "lengthscales": orc.constrained(np.sqrt(d) * np.ones(d)), "input_dim": d}
    rmu, rvar = orc.conditional(X[idx], Z, spec, f, white=False)
    print("cfg5 spot parity (unwhitened): mean rel %.2e var abs %.2e" % (np.abs(mu[idx] - rmu).max() / np.abs(rmu).max(), np.abs(var[idx] - rvar).max()))
    for cls in ("gemm_f64", "kmat", "reduce"):
        pass
