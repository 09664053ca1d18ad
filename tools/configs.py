"""BASELINE.json configs 2, 4 and 5 at full size: timings (one JSON line per config) + a size-independent sanity check.
    python tools/configs.py 2|4|5
Used by tools/collect_profiles.sh for the rocprofv3 kernel statistics under profiles/."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
which = sys.argv[1] if len(sys.argv) > 1 else "4"
PEAK = 78.6


def best(f, reps, stages=None):
    """(fastest wall time in ms, last value); stages: a dict that receives the handle's stage times OF THE FASTEST repetition."""
    tmin, v = None, None
    for _ in range(reps):
        t0 = time.perf_counter(); v = f(); t = 1e3 * (time.perf_counter() - t0)
        if tmin is None or t < tmin:
            tmin = t
            if stages is not None:
                stages.clear(); stages.update(h.last_stage_ms())
    return tmin, v


if which == "2":
    n, d = 8192, 8
    X, Y, Xs = orc.synthetic_gpr_data(n, d, 1024)
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True), obs_var=0.1)
    m.compute_log_likelihood()
    st = {}
    ms, lml = best(m.compute_log_likelihood, 20, st)
    m.reuse_factor = True
    pw, _ = best(lambda: m.predict_f(Xs), 5)
    m.reuse_factor = False
    pc, _ = best(lambda: m.predict_f(Xs), 5)
    print(json.dumps({"config": 2, "workload": "RBF(ARD) GPR N=8192 D=8 fp64", "lml_ms": round(ms, 3), "stage_ms": {k: round(v, 3) for k, v in st.items()},
                      "potrf_tflops": round(n ** 3 / 3 / (st["potrf"] * 1e-3) / 1e12, 2), "potrf_frac_of_peak": round(n ** 3 / 3 / (st["potrf"] * 1e-3) / 1e12 / PEAK, 4),
                      "predict_f_1024_ms": {"warm": round(pw, 3), "cold": round(pc, 3)}, "lml": lml}))
elif which == "4":
    n, d = 16384, 16
    X, Y, Xs = orc.synthetic_gpr_data(n, d, 1024)
    kern = gpf.kernels.Matern52(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True) + gpf.kernels.Periodic(d, period=2.0, variance=1.0, lengthscales=1.0)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    m.compute_log_likelihood()
    st = {}
    ms, lml = best(m.compute_log_likelihood, 5, st)
    pc, (mu, var) = best(lambda: m.predict_f(Xs), 2)
    print(json.dumps({"config": 4, "workload": "Matern-5/2(ARD) + Periodic GPR N=16384 D=16 fp64", "lml_ms": round(ms, 3),
                      "stage_ms": {k: round(v, 3) for k, v in st.items()}, "kmat_gbs": round(4.0 * n * n / (st["kmat"] * 1e-3) / 1e9, 1),
                      "potrf_tflops": round(n ** 3 / 3 / (st["potrf"] * 1e-3) / 1e12, 2), "potrf_frac_of_peak": round(n ** 3 / 3 / (st["potrf"] * 1e-3) / 1e12 / PEAK, 4),
                      "predict_f_1024_cold_ms": round(pc, 2), "var_range": [float(var.min()), float(var.max())], "lml": lml}))
else:
    M, N, d = 4096, int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000, 8
    rng = np.random.default_rng(1)
    X = rng.standard_normal((N, d)); Z = X[:M].copy()
    Y = np.sin(X @ (rng.standard_normal((d, 1)) / np.sqrt(d))) + 0.1 * rng.standard_normal((N, 1))
    f = rng.standard_normal((M, 1))
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
    out = {"config": 5, "workload": "SVGP / conditional RBF(ARD) M=4096 N=%d D=8 fp64" % N}
    for white in (True, False):
        gpf.conditionals.conditional(X[:4096], Z, kern, f, white=white)
        ms, (mu, var) = best(lambda: gpf.conditionals.conditional(X, Z, kern, f, white=white), 2)
        out["conditional_white_%s_ms" % white] = round(ms, 1)
        out["conditional_white_%s_trsm_tflops" % white] = round(float(M) * M * N / (ms * 1e-3) / 1e12, 1)
    h.set_option("leaf_refine", 0)
    ms0, _ = best(lambda: gpf.conditionals.conditional(X, Z, kern, f, white=True), 2)
    h.set_option("leaf_refine", -1)
    out["conditional_white_True_ms_plain_leaves"] = round(ms0, 1)
    q_mu = rng.standard_normal((M, 1)) * 0.3
    q_sqrt = (np.tril(rng.standard_normal((M, M))) * (0.5 / M) + 0.5 * np.eye(M))[:, :, None]
    sv = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.1), Z=Z, whiten=True)
    sv._q_mu.assign(q_mu); sv._q_sqrt.assign(q_sqrt)
    sv.compute_log_likelihood()
    ms, elbo = best(sv.compute_log_likelihood, 2)
    out["svgp_elbo_full_q_sqrt_whitened_ms"] = round(ms, 1); out["svgp_elbo"] = elbo
    if "grad" in sys.argv:
        sv.compute_log_likelihood_and_gradients()
        ms, (b2, grads) = best(sv.compute_log_likelihood_and_gradients, 2)
        out["svgp_elbo_plus_gradient_ms"] = round(ms, 1)
        out["svgp_grad_norms"] = {p.name: float(np.abs(g).max()) for p, g in grads}
        svz = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.1), Z=Z, whiten=True, train_inducing=True)
        svz._q_mu.assign(q_mu); svz._q_sqrt.assign(q_sqrt)
        svz.compute_log_likelihood_and_gradients()
        ms, (b3, gz) = best(svz.compute_log_likelihood_and_gradients, 2)
        out["svgp_elbo_plus_gradient_incl_inducing_inputs_ms"] = round(ms, 1)
        out["svgp_grad_Z_max"] = float(np.abs({id(p): g for p, g in gz}[id(svz.feature._Z)]).max())
    idx = rng.choice(N, 200, replace=False)
    spec = {"type": "rbf", "variance": orc.constrained(1.0), "lengthscales": orc.constrained(np.sqrt(d) * np.ones(d)), "input_dim": d}
    rmu, rvar = orc.conditional(X[idx], Z, spec, f, white=False)
    out["spot_parity_unwhitened"] = {"mean_rel": float(np.abs(mu[idx] - rmu).max() / np.abs(rmu).max()), "var_abs": float(np.abs(var[idx] - rvar).max())}
    print(json.dumps(out))
