"""Generates tests/golden/sparse/*.npz: frozen inputs / outputs of the inducing-point entry points -- conditional()
(conditionals.py:24-121), gauss_kl (kullback_leiblers.py:26-105), the SVGP bound (models/svgp.py:101-130), the SGPR and
FITC bounds and predictions (models/sgpr.py:121-189, 229-318) -- from the CPU restatement oracle/gp_oracle.py (the
reference needs TensorFlow 1.x and cannot run here; the oracle itself is pinned in tests/test_oracle.py).

    python tests/golden/sparse/make_golden_sparse.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import oracle.gp_oracle as orc  # noqa: E402

c = orc.constrained
CASES = {"rbf_ard_m40": ("rbf", 40, 150, 3, 2, 11), "matern52_m130": ("matern52", 130, 300, 2, 1, 12)}


def spec_for(kind, d):
    ls = np.linspace(0.8, 1.6, d)
    return {"type": kind, "variance": c(1.2), "lengthscales": c(ls), "input_dim": d}


def inputs(name):
    kind, m, n, d, k, seed = CASES[name]
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, k))) + 0.1 * rng.standard_normal((n, k))
    Z = X[:m].copy(); Xs = rng.standard_normal((25, d))
    q_mu = rng.standard_normal((m, k)) * 0.3
    q_diag = np.abs(rng.standard_normal((m, k))) * 0.4 + 0.2
    q_full = np.tril(rng.standard_normal((k, m, m)) * (0.5 / m) + np.eye(m) * 0.5).transpose(1, 2, 0).copy()
    return dict(kind=kind, X=X, Y=Y, Z=Z, Xs=Xs, q_mu=q_mu, q_diag=q_diag, q_full=q_full)


if __name__ == "__main__":
    for name in CASES:
        g = inputs(name)
        d, m = g["X"].shape[1], g["Z"].shape[0]
        spec = spec_for(g["kind"], d)
        noise = float(c(0.2))
        Kuu = orc.K(spec, g["Z"]) + orc.JITTER * np.eye(m)
        out = dict(g, noise_var=noise, cond_Kuu=np.linalg.cond(Kuu))
        out.pop("kind")
        for white in (True, False):
            for qn in ("q_diag", "q_full"):
                tag = "%s_%s" % ("white" if white else "unwhite", qn)
                mu, var = orc.conditional(g["Xs"], g["Z"], spec, g["q_mu"], q_sqrt=g[qn], white=white)
                out["cond_mu_" + tag], out["cond_var_" + tag] = mu, var
                out["kl_" + tag] = orc.gauss_kl(g["q_mu"], g[qn], None if white else Kuu)
                out["elbo_" + tag] = orc.svgp_elbo(spec, g["X"], g["Y"], g["Z"], g["q_mu"], g[qn], noise, whiten=white, num_data=3 * g["X"].shape[0])
        out["sgpr_bound"] = orc.sgpr_bound(spec, g["X"], g["Y"], g["Z"], noise)
        out["sgpr_mu"], out["sgpr_var"] = orc.sgpr_predict(spec, g["X"], g["Y"], g["Z"], noise, g["Xs"])
        out["fitc_lml"] = orc.fitc_lml(spec, g["X"], g["Y"], g["Z"], noise)
        out["fitc_mu"], out["fitc_var"] = orc.fitc_predict(spec, g["X"], g["Y"], g["Z"], noise, g["Xs"])
        np.savez(os.path.join(HERE, name + ".npz"), **out)
        print(name, out["cond_Kuu"], out["sgpr_bound"], out["elbo_white_q_full"])
