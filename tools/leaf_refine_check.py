"""Scratch: accuracy (vs the exact-arithmetic fixture) and cost of the refined 128-column leaves (trsm_leaf.hip)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "exact"))
import gpflowSlim as gpf
import oracle.gp_oracle as orc
import make_illcond_exact as gen
h = gpf.get_handle()
ref = np.load(os.path.join(ROOT, "tests", "golden", "exact", "illcond_conditional_exact.npz"))
for mode in (0, 1):
    h.set_option("leaf_refine", mode)
    errs = []; lap = []
    for i, (s, m, d) in enumerate(gen.CASES):
        Z, Xn, f, ls, spec = gen.inputs(s, m, d)
        kern = gpf.kernels.RBF(d, variance=1.3, lengthscales=ls, ARD=True)
        mu, _ = gpf.conditionals.conditional(Xn, Z, kern, f, white=False)
        errs.append(np.abs(mu - ref["exact%d" % i]).max()); lap.append(np.abs(ref["lapack%d" % i] - ref["exact%d" % i]).max())
        Kmm = orc.K(spec, Z) + np.eye(m) * orc.JITTER; Kmn = orc.K(spec, Z, Xn)
        mu2, _ = gpf.conditionals.base_conditional(Kmn, Kmm, orc.Kdiag(spec, Xn), f, white=False)
        errs.append(np.abs(mu2 - ref["exact%d" % i]).max()); lap.append(lap[-1])
    errs = np.array(errs); lap = np.array(lap)
    print("leaf_refine=%d: hip geomean %.2e max %.2e | lapack geomean %.2e max %.2e" % (mode, np.exp(np.log(errs).mean()), errs.max(), np.exp(np.log(lap).mean()), lap.max()), flush=True)

# cost: N = 32768 LML (GPR path), N = 8192, cfg5-like conditional
for n in (8192, 32768):
    X, Y, Xs = orc.synthetic_gpr_data(n, 8, 1024)
    kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1); m.reuse_factor = True
    for mode in (0, 1, 0, 1):
        h.set_option("leaf_refine", mode)
        ts = []
        for i in range(4):
            t0 = time.perf_counter(); lml = m.compute_log_likelihood(); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); mu, var = m.predict_f(Xs); tp = time.perf_counter() - t0
        print("N=%d leaf_refine=%d: lml %.12g  best %.2f ms  stage %s  warm predict %.2f ms" % (n, mode, lml, 1e3 * min(ts), {k: round(v, 2) for k, v in h.last_stage_ms().items()}, 1e3 * tp), flush=True)
M, N, d = 4096, 500000, 8
rng = np.random.default_rng(1)
X = rng.standard_normal((N, d)); Z = X[:M].copy(); f = rng.standard_normal((M, 1))
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
for mode in (0, 1, 0, 1):
    h.set_option("leaf_refine", mode)
    t0 = time.perf_counter(); mu, var = gpf.conditionals.conditional(X, Z, kern, f, white=True); t1 = time.perf_counter()
    print("cfg5-half leaf_refine=%d: %.1f ms" % (mode, 1e3 * (t1 - t0)), flush=True)
