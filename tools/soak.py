"""Soak: a long mixed sequence of every entry point on two handles, one line of progress per step (flushed), so that a stall
shows where it happened.  usage: soak.py [steps=400] [seed=0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
from gpflowSlim import _backend as be
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
h2 = be.Handle(0)
t_start = time.perf_counter()
worst = 0.0
for it in range(steps):
    kind = ["gpr", "gpr_grad", "predict", "cond", "svgp_grad", "sgpr_grad", "fitc_grad", "second_handle", "big"][int(rng.integers(9))]
    n = int(rng.choice([130, 300, 700, 1500, 3000, 5000, 9000])) if kind != "big" else int(rng.choice([12000, 16384]))
    d = int(rng.integers(1, 6))
    X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
    kern = gpf.kernels.Matern52(d, lengthscales=1.0 + rng.random()) + gpf.kernels.RBF(d, variance=0.5) if rng.random() < 0.3 else gpf.kernels.RBF(d, ARD=True)
    t0 = time.perf_counter()
    if kind in ("gpr", "big"):
        v = gpf.models.GPR(X, Y, kern, obs_var=0.1).compute_log_likelihood()
    elif kind == "gpr_grad":
        v, _ = gpf.models.GPR(X[:3000], Y[:3000], kern, obs_var=0.1).compute_log_likelihood_and_gradients()
    elif kind == "predict":
        mu, var = gpf.models.GPR(X, Y, kern, obs_var=0.1).predict_f(rng.standard_normal((200, d))); v = float(mu.sum() + var.sum())
    elif kind == "cond":
        m_ = min(n, 200)
        mu, var = gpf.conditionals.conditional(X, X[:m_].copy(), kern, rng.standard_normal((m_, 1)), white=bool(rng.integers(2))); v = float(mu.sum() + var.sum())
    elif kind == "svgp_grad":
        m_ = min(n // 2, 120)
        v, _ = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.2), Z=X[:m_].copy(), whiten=bool(rng.integers(2)),
                               q_diag=bool(rng.integers(2)), train_inducing=bool(rng.integers(2))).compute_log_likelihood_and_gradients()
    elif kind in ("sgpr_grad", "fitc_grad"):
        m_ = min(n // 2, 150)
        cls = gpf.models.SGPR if kind == "sgpr_grad" else gpf.models.GPRFITC
        v, _ = cls(X, Y, kern, Z=X[:m_].copy(), obs_var=0.2).compute_log_likelihood_and_gradients()
    else:
        h2.gpr_set_data(X, ("soak", it))
        v = h2.gpr_lml(kern._program(d), 0.1, Y)
    dt = time.perf_counter() - t0
    worst = max(worst, dt)
    assert np.isfinite(v), (it, kind, n, d, v)
    print("%4d %-13s n=%5d d=%d  %.1f ms" % (it, kind, n, d, 1e3 * dt), flush=True)
h = gpf.get_handle()
summary = {"steps": steps, "seconds": round(time.perf_counter() - t_start, 1), "slowest_step_s": round(worst, 3),
           "lookahead_retries": [h.profile_get("lookahead_retries")["launches"], h2.profile_get("lookahead_retries")["launches"]],
           "trsv_wave_fallbacks": [h.profile_get("trsv_wave_fallbacks")["launches"], h2.profile_get("trsv_wave_fallbacks")["launches"]],
           "small_n_fallbacks": [h.profile_get("small_n_fallbacks")["launches"], h2.profile_get("small_n_fallbacks")["launches"]],
           "what": "tools/soak.py: mixed sequence of every entry point (GPR LML / gradient / predict_f, conditional, SVGP / SGPR / FITC gradients) on two "
                   "handles, sizes 130 .. 16384 (the one-launch small-N path, the look-ahead sweeps, the wavefront substitution all take part)"}
print("done: %d steps in %.1f s, slowest step %.2f s, look-ahead retries %s, wavefront fall-backs %s, small-N fall-backs %s" % (
    steps, summary["seconds"], worst, summary["lookahead_retries"], summary["trsv_wave_fallbacks"], summary["small_n_fallbacks"]), flush=True)
import json
from bench import kernel_source_sha
summary["kernel_source_sha"] = kernel_source_sha()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(summary, open(os.path.join(ROOT, "gpurun_out", "soak.json"), "w"), indent=1)
h2.close()
