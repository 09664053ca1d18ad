"""Scratch: SGPR / FITC bounds at config-5 size (M = 4096 inducing points, N points) -- time and per-class breakdown."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m_, d = 4096, 8
rng = np.random.default_rng(5)
X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1)); Z = X[:m_].copy()
h = gpf.get_handle()
out = {"n": n, "m": m_}
for name, cls in (("sgpr", gpf.models.SGPR), ("fitc", gpf.models.GPRFITC)):
    model = cls(X, Y, gpf.kernels.RBF(d, lengthscales=np.sqrt(d) * np.ones(d), ARD=True), Z=Z)
    model.likelihood._variance.assign(0.1)
    v = model.compute_log_likelihood()
    h.profile_reset(); h.profile_enable(True)
    t0 = time.perf_counter(); v = model.compute_log_likelihood(); t1 = time.perf_counter()
    h.profile_enable(False)
    cl = {k: h.profile_get(k) for k in ("gemm_f64", "potrf_base", "kmat", "trsv", "reduce", "other")}
    out[name] = {"bound": v, "ms": round(1e3 * (t1 - t0), 1), "classes_ms": {k: round(c["ms"], 1) for k, c in cl.items()},
                 "gemm_tflops": round(cl["gemm_f64"]["flops"] / max(cl["gemm_f64"]["ms"], 1e-9) / 1e9, 1)}
model = gpf.models.SGPR(X, Y, gpf.kernels.RBF(d, lengthscales=np.sqrt(d) * np.ones(d), ARD=True), Z=Z)
model.likelihood._variance.assign(0.1)
model.compute_log_likelihood_and_gradients()
t0 = time.perf_counter(); b, grads = model.compute_log_likelihood_and_gradients(); t1 = time.perf_counter()
out["sgpr_bound_plus_gradient_ms"] = round(1e3 * (t1 - t0), 1)
out["sgpr_grad_max"] = {p.name: float(np.abs(g).max()) for p, g in grads}
print(json.dumps(out))
