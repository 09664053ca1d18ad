"""Generates tests/golden/*.npz.

The reference (gpflowSlim over TensorFlow 1.x) cannot be imported in the build image (no
TensorFlow), so these vectors are produced by the CPU restatement oracle/gp_oracle.py, whose own
correctness is pinned independently in tests/test_oracle.py (analytic cases, 50-digit mpmath,
scikit-learn).  They freeze the oracle (any later drift fails tests/test_oracle.py) and give the
GPU tests inputs/outputs that do not depend on running the oracle at test time.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle.gp_oracle as orc  # noqa: E402

c = orc.constrained


def specs(d):
    ls = np.linspace(0.7, 1.9, d)
    rbf = {"type": "rbf", "variance": c(1.3), "lengthscales": c(ls), "input_dim": d}
    m52 = {"type": "matern52", "variance": c(1.1), "lengthscales": c(ls * 1.5), "input_dim": d}
    m32 = {"type": "matern32", "variance": c(0.9), "lengthscales": c(1.4), "input_dim": d}
    m12 = {"type": "matern12", "variance": c(0.8), "lengthscales": c(2.0), "input_dim": d}
    per = {"type": "periodic", "variance": c(0.9), "lengthscales": c(1.2), "period": c(2.0), "input_dim": d}
    return {
        "rbf_ard": rbf, "matern52_ard": m52, "matern32_iso": m32, "matern12_iso": m12, "periodic": per,
        "m52_plus_periodic": {"type": "sum", "children": [m52, per]},
        "rbf_times_periodic_plus_const": {"type": "sum", "children": [{"type": "product", "children": [rbf, per]}, 0.5]},
    }


def case(name, spec, n, d, r, ns, seed, noise=0.1):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    Y = np.sin(X @ rng.standard_normal((d, r))) + 0.1 * rng.standard_normal((n, r))
    Xs = rng.standard_normal((ns, d))
    nv = float(c(noise))
    lml = orc.gpr_lml(spec, X, Y, nv)
    mu, var = orc.gpr_predict(spec, X, Y, nv, Xs)
    _, cov = orc.gpr_predict(spec, X, Y, nv, Xs[:16], full_cov=True)
    np.savez(os.path.join(HERE, name + ".npz"), X=X, Y=Y, Xs=Xs, noise_var=nv, lml=lml, mu=mu, var=var, cov16=cov)
    print(name, n, d, lml)


if __name__ == "__main__":
    # BASELINE.json configs[0]: examples/gpr.py-shaped RBF(ARD) GPR, N=512 D=4 (plumbing case)
    case("cfg1_rbf_ard_n512_d4", specs(4)["rbf_ard"], 512, 4, 1, 64, 20240607)
    for i, (k, s) in enumerate(specs(3).items()):
        case("n64_d3_" + k, s, 64, 3, 2, 10, 100 + i)
