"""Generates tests/golden/exact/illcond_conditional_exact.npz: conditional() means on ill-conditioned Kuu (1-D / 2-D inputs,
M up to 512, jitter 1e-6: cond(Kuu) ~ 1e8) evaluated in EXACT arithmetic (mpmath, 60 digits) from the oracle's fp64
kernel matrices, next to the oracle's own LAPACK result.  Pins how many digits any fp64 implementation of
conditionals.py:80-104 can deliver on such inputs (LAPACK: ~1e-8 absolute on O(1) means) and is the yardstick for
tests/test_gpu_parity.py::test_ill_conditioned_conditional_against_exact_arithmetic.  ~9 minutes of CPU time.

    python tests/golden/exact/make_illcond_exact.py
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import oracle.gp_oracle as orc

c = orc.constrained
CASES = [(s, m, d) for s in range(4) for (m, d) in ((257, 1), (200, 1), (512, 2))]


def inputs(seed, m, d):
    rng = np.random.default_rng(9000 + seed)
    Z = rng.standard_normal((m, d)); Xn = rng.standard_normal((5, d)); f = rng.standard_normal((m, 2))
    ls = np.linspace(0.7, 1.9, d)
    spec = {"type": "rbf", "variance": c(1.3), "lengthscales": c(ls), "input_dim": d}
    return Z, Xn, f, ls, spec


if __name__ == "__main__":
    import mpmath as mp
    mp.mp.dps = 60
    out = {}
    for i, (s, m, d) in enumerate(CASES):
        Z, Xn, f, ls, spec = inputs(s, m, d)
        Kmm = orc.K(spec, Z) + np.eye(m) * orc.JITTER
        Kmn = orc.K(spec, Z, Xn)
        A = mp.matrix(Kmm.tolist()); B = mp.matrix(Kmn.tolist()); F = mp.matrix(f.tolist())
        W = mp.matrix(m, 2)
        for j in range(2):
            col = mp.cholesky_solve(A, F[:, j])
            for r in range(m):
                W[r, j] = col[r]
        mu = B.T * W
        out["exact%d" % i] = np.array([[float(mu[a, b]) for b in range(2)] for a in range(5)])
        out["lapack%d" % i] = orc.conditional(Xn, Z, spec, f, white=False)[0]
        out["cond%d" % i] = np.linalg.cond(Kmm)
        print(i, s, m, d, "cond %.2g" % out["cond%d" % i], "lapack err %.2e" % np.abs(out["lapack%d" % i] - out["exact%d" % i]).max(), flush=True)
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "illcond_conditional_exact.npz"), **out)
