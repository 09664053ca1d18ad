"""Numerical model (numpy, no GPU) of the choice made in gpflow-slim_amd/csrc/trsm_leaf.hip: a blocked Cholesky / triangular
solve whose 128-column leaves are products with the explicit inverse of the diagonal block is about one digit behind
substitution (LAPACK, what tf.cholesky / tf.matrix_triangular_solve do at models/gpr.py:70, conditionals.py:84-100) on
the reference's 1e-6-jitter matrices; one refinement step per leaf, in BOTH the factorisation's panel solves and the
solves proper, closes the gap, refining only one of the two does not.  Yardstick: the 60-digit fixture
tests/golden/exact/illcond_conditional_exact.npz."""
import os
import sys

import numpy as np
import scipy.linalg as sl

import oracle.gp_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden", "exact"))
import make_illcond_exact as gen  # noqa: E402

T = 128


def _pad(K):
    m = K.shape[0]
    mp = -(-m // T) * T
    P = np.eye(mp)
    P[:m, :m] = K
    return P


def _leaf(B, L11, W, refine):
    X = B @ W.T
    for _ in range(refine):
        X = X + (B - X @ L11.T) @ W.T
    return X


def _potrf(A, refine):
    A = A.copy()
    n = A.shape[0]
    inv = []
    for c in range(0, n, T):
        L11 = np.linalg.cholesky(A[c:c + T, c:c + T])
        A[c:c + T, c:c + T] = L11
        inv.append(sl.solve_triangular(L11, np.eye(T), lower=True))
        if c + T < n:
            X = _leaf(A[c + T:, c:c + T], L11, inv[-1], refine)
            A[c + T:, c:c + T] = X
            A[c + T:, c + T:] -= X @ X.T
    return np.tril(A), inv


def _trsm(L, inv, Bt, refine):          # X L^T = Bt
    Bt = Bt.copy()
    n = L.shape[0]
    for c in range(0, n, T):
        X = _leaf(Bt[:, c:c + T], L[c:c + T, c:c + T], inv[c // T], refine)
        Bt[:, c:c + T] = X
        if c + T < n:
            Bt[:, c + T:] -= X @ L[c + T:, c:c + T].T
    return Bt


def test_one_refinement_step_in_every_leaf_reaches_substitution_accuracy():
    ref = np.load(os.path.join(HERE, "golden", "exact", "illcond_conditional_exact.npz"))
    err = {k: [] for k in ("lapack", "plain", "trsm_only", "potrf_only", "both")}
    for i, (s, m, d) in enumerate(gen.CASES):
        Z, Xn, f, ls, spec = gen.inputs(s, m, d)
        Kmm = orc.K(spec, Z) + np.eye(m) * orc.JITTER
        Kmn = orc.K(spec, Z, Xn)
        exact = ref["exact%d" % i]
        Kp = _pad(Kmm)
        mp = Kp.shape[0]
        Bt = np.zeros((Xn.shape[0], mp)); Bt[:, :m] = Kmn.T
        Ft = np.zeros((f.shape[1], mp)); Ft[:, :m] = f.T
        err["lapack"].append(np.abs(ref["lapack%d" % i] - exact).max())
        for name, (rp, rt) in (("plain", (0, 0)), ("trsm_only", (0, 1)), ("potrf_only", (1, 0)), ("both", (1, 1))):
            L, inv = _potrf(Kp, rp)
            mu = _trsm(L, inv, Bt, rt) @ _trsm(L, inv, Ft, rt).T          # A^T (Lm^-1 f), conditionals.py:87-103
            err[name].append(np.abs(mu - exact).max())
    gm = {k: float(np.exp(np.mean(np.log(v)))) for k, v in err.items()}
    assert gm["both"] <= 1.5 * gm["lapack"], gm                 # measured: 8e-9 vs 1.4e-8
    assert max(err["both"]) <= 2.0 * max(err["lapack"]), gm
    for k in ("plain", "trsm_only", "potrf_only"):              # measured: 7e-8 .. 9e-8
        assert gm[k] >= 3.0 * gm["both"], gm
