"""Pins for the CPU oracle (oracle/gp_oracle.py).  The reference ships no numeric test vectors and
cannot run here (no TensorFlow), so the restatement is pinned by: analytic known answers, 50-digit
mpmath evaluation of the reference formulas, scikit-learn's independent GP implementation, the
reference's own structural tests restated, and the frozen fixtures in tests/golden/."""
import glob
import os

import numpy as np
import pytest
import scipy.linalg as sl

import oracle.gp_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
c = orc.constrained


# ---- transforms ---------------------------------------------------------------------------------
def test_log1pe_roundtrip_and_floor():
    for v in [1e-5, 0.1, 1.0, 3.7, 50.0, 1e3]:
        assert abs(c(v) - v) <= 4e-16 * max(1.0, v)
    # forward is softplus + 1e-6 (transforms.py:145-146)
    assert orc.log1pe_forward(0.0) == pytest.approx(np.log(2.0) + 1e-6, rel=1e-15)
    assert orc.log1pe_forward(-800.0) == pytest.approx(1e-6, rel=1e-12)
    assert np.isfinite(orc.log1pe_backward(1e-6))


# ---- analytic known answers -------------------------------------------------------------------
def test_kat_single_point():
    v, s2, y, x = c(1.7), c(0.3), 0.9, np.array([[0.4]])
    spec = {"type": "rbf", "variance": v, "lengthscales": c(1.0), "input_dim": 1}
    lml = orc.gpr_lml(spec, x, np.array([[y]]), s2)
    assert lml == pytest.approx(-0.5 * np.log(2 * np.pi) - 0.5 * np.log(v + s2) - y * y / (2 * (v + s2)), rel=1e-14)
    mu, var = orc.gpr_predict(spec, x, np.array([[y]]), s2, x)
    assert mu[0, 0] == pytest.approx(v / (v + s2) * y, rel=1e-14)
    assert var[0, 0] == pytest.approx(v - v * v / (v + s2), rel=1e-13)


def test_kat_two_points_closed_form():
    v, s2, ell = c(1.2), c(0.1), c(0.8)
    X = np.array([[0.0], [0.5]]); Y = np.array([[0.3], [-0.2]])
    spec = {"type": "rbf", "variance": v, "lengthscales": ell, "input_dim": 1}
    k = v * np.exp(-0.5 * (0.5 / ell) ** 2)
    a = v + s2
    det = a * a - k * k
    quad = (a * Y[0, 0] ** 2 - 2 * k * Y[0, 0] * Y[1, 0] + a * Y[1, 0] ** 2) / det
    assert orc.gpr_lml(spec, X, Y, s2) == pytest.approx(-np.log(2 * np.pi) - 0.5 * np.log(det) - 0.5 * quad, rel=1e-13)


def test_kat_far_points_and_far_prediction():
    v, s2 = c(2.0), c(0.5)
    X = np.array([[0.0], [1e3], [2e3]]); Y = np.array([[1.0], [2.0], [3.0]])
    spec = {"type": "rbf", "variance": v, "lengthscales": c(1.0), "input_dim": 1}
    # K = v I  ->  independent Gaussians
    ref = -1.5 * np.log(2 * np.pi) - 1.5 * np.log(v + s2) - np.sum(Y ** 2) / (2 * (v + s2))
    assert orc.gpr_lml(spec, X, Y, s2) == pytest.approx(ref, rel=1e-14)
    mu, var = orc.gpr_predict(spec, X, Y, s2, np.array([[-5e3]]))
    assert mu[0, 0] == 0.0 and var[0, 0] == pytest.approx(v, rel=1e-15)      # prior far from the data


def test_kat_multi_output_counts_logdet_r_times():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((20, 2)); Y = rng.standard_normal((20, 3))
    spec = {"type": "rbf", "variance": c(1.0), "lengthscales": c(1.0), "input_dim": 2}
    s2 = c(0.1)
    total = orc.gpr_lml(spec, X, Y, s2)
    parts = sum(orc.gpr_lml(spec, X, Y[:, j:j + 1], s2) for j in range(3))
    assert total == pytest.approx(parts, rel=1e-13)                         # densities.py:93


def test_matern_diag_constants():
    """kernels.py:426: r = sqrt(r2 + 1e-12) => K_ii = var * exp(-1e-6) for Matern12, Kdiag = var."""
    X = np.zeros((3, 2))
    for t, f in [("matern12", np.exp(-1e-6)), ("exponential", np.exp(-0.5e-6)),
                 ("matern32", (1 + np.sqrt(3) * 1e-6) * np.exp(-np.sqrt(3) * 1e-6)),
                 ("matern52", (1 + np.sqrt(5) * 1e-6 + 5 / 3 * 1e-12) * np.exp(-np.sqrt(5) * 1e-6))]:
        spec = {"type": t, "variance": 1.5, "lengthscales": 1.0, "input_dim": 2}
        K = orc.K(spec, X)
        assert np.allclose(K, 1.5 * f, rtol=1e-15, atol=0)
        assert np.array_equal(orc.Kdiag(spec, X), np.full(3, 1.5))


def test_white_constant_sum_product_semantics():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((6, 3)); X2 = rng.standard_normal((4, 3))
    w = {"type": "white", "variance": 0.3}; k0 = {"type": "constant", "variance": 0.7}
    r = {"type": "rbf", "variance": 1.1, "lengthscales": 0.9, "input_dim": 2, "active_dims": [0, 2]}
    assert np.array_equal(orc.K(w, X), 0.3 * np.eye(6)) and np.array_equal(orc.K(w, X, X2), np.zeros((6, 4)))
    assert np.array_equal(orc.K(k0, X, X2), np.full((6, 4), 0.7))
    s = {"type": "sum", "children": [r, w, 2.0]}
    assert np.allclose(orc.K(s, X), orc.K(r, X) + 0.3 * np.eye(6) + 2.0, rtol=1e-15)
    p = {"type": "product", "children": [r, k0, 2.0]}
    assert np.allclose(orc.K(p, X, X2), orc.K(r, X, X2) * 0.7 * 2.0, rtol=1e-15)
    # active dims: column 1 must not matter
    Xp = X.copy(); Xp[:, 1] += 100.0
    assert np.array_equal(orc.K(r, X), orc.K(r, Xp))


# ---- 50-digit mpmath evaluation of the same formulas -----------------------------------------
def _mp_kernel(mp, spec, x, y, same):
    t = spec["type"]
    if t == "sum":
        return sum((_mp_kernel(mp, ch, x, y, same) if isinstance(ch, dict) else mp.mpf(ch)) for ch in spec["children"])
    if t == "product":
        out = mp.mpf(1)
        for ch in spec["children"]:
            out *= _mp_kernel(mp, ch, x, y, same) if isinstance(ch, dict) else mp.mpf(ch)
        return out
    v = mp.mpf(float(spec["variance"]))
    if t == "white":
        return v if same else mp.mpf(0)
    if t == "constant":
        return v
    ad = spec.get("active_dims") or list(range(spec["input_dim"]))
    xs = [mp.mpf(float(x[d])) for d in ad]; ys = [mp.mpf(float(y[d])) for d in ad]
    if t == "periodic":
        p, l = mp.mpf(float(spec["period"])), mp.mpf(float(spec["lengthscales"]))
        return v * mp.exp(-sum((mp.sin(mp.pi * (a - b) / p) / l) ** 2 for a, b in zip(xs, ys)) / 2)
    ls = np.broadcast_to(np.asarray(spec["lengthscales"], dtype=float), (len(ad),))
    r2 = sum(((a - b) / mp.mpf(float(l))) ** 2 for a, b, l in zip(xs, ys, ls))
    if t == "rbf":
        return v * mp.exp(-r2 / 2)
    r = mp.sqrt(r2 + mp.mpf("1e-12"))
    if t == "matern12":
        return v * mp.exp(-r)
    if t == "matern32":
        return v * (1 + mp.sqrt(3) * r) * mp.exp(-mp.sqrt(3) * r)
    if t == "matern52":
        return v * (1 + mp.sqrt(5) * r + mp.mpf(5) / 3 * r * r) * mp.exp(-mp.sqrt(5) * r)
    raise ValueError(t)


def _mp_gpr(spec, X, Y, s2, Xs):
    import mpmath as mp
    mp.mp.dps = 50
    n = X.shape[0]
    Km = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            Km[i, j] = _mp_kernel(mp, spec, X[i], X[j], i == j) + (mp.mpf(float(s2)) if i == j else 0)
    L = mp.cholesky(Km)
    y = mp.matrix([float(v) for v in Y[:, 0]])
    alpha = mp.lu_solve(L, y)
    lml = -mp.mpf(n) / 2 * mp.log(2 * mp.pi) - sum(mp.log(L[i, i]) for i in range(n)) - sum(a * a for a in alpha) / 2
    mus, vars_ = [], []
    for s in range(Xs.shape[0]):
        kx = mp.matrix([_mp_kernel(mp, spec, X[i], Xs[s], False) for i in range(n)])
        a = mp.lu_solve(L, kx)
        mus.append(float(sum(a[i] * alpha[i] for i in range(n))))
        kss = mp.mpf(float(np.ravel(orc.Kdiag(spec, Xs[s:s + 1]))[0]))   # Kdiag: exact fold of the variances
        vars_.append(float(kss - sum(a[i] * a[i] for i in range(n))))
    return float(lml), np.array(mus), np.array(vars_)


MP_SPECS = {
    "rbf": {"type": "rbf", "variance": c(1.3), "lengthscales": c(np.array([0.7, 1.1, 1.6, 2.0])), "input_dim": 4},
    "matern52": {"type": "matern52", "variance": c(0.9), "lengthscales": c(1.4), "input_dim": 4},
    "periodic": {"type": "periodic", "variance": c(1.1), "lengthscales": c(1.3), "period": c(2.5), "input_dim": 4},
    "sum": {"type": "sum", "children": [{"type": "matern32", "variance": c(0.6), "lengthscales": c(0.9), "input_dim": 2, "active_dims": [0, 3]},
                                        {"type": "periodic", "variance": c(0.8), "lengthscales": c(1.0), "period": c(3.0), "input_dim": 4}]},
    "product": {"type": "product", "children": [{"type": "rbf", "variance": c(1.2), "lengthscales": c(1.5), "input_dim": 4},
                                            {"type": "matern12", "variance": c(0.7), "lengthscales": c(2.0), "input_dim": 4}, 1.5]},
}


@pytest.mark.parametrize("name", sorted(MP_SPECS))
@pytest.mark.parametrize("n", [4, 16])
def test_oracle_matches_50_digit_evaluation(name, n):
    spec = MP_SPECS[name]
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 4)); Y = rng.standard_normal((n, 1)); Xs = rng.standard_normal((3, 4))
    s2 = c(0.1)
    lml, mu, var = _mp_gpr(spec, X, Y, s2, Xs)
    # The Matern family computes r = sqrt(r2 + 1e-12) (kernels.py:426) from a GEMM-form r2 whose
    # diagonal is "whatever rounding leaves" (O(1e-15)); the sqrt amplifies that to O(1e-9) in K_ii.
    # The 50-digit evaluation has r2_ii = 0 exactly, so Matern cases agree to ~1e-9, the rest to 1e-11.
    tol = 5e-9 if name in ("matern52", "sum", "product") else 1e-11
    assert orc.gpr_lml(spec, X, Y, s2) == pytest.approx(lml, rel=tol)
    omu, ovar = orc.gpr_predict(spec, X, Y, s2, Xs)
    assert np.abs(omu[:, 0] - mu).max() <= tol * max(1.0, np.abs(mu).max())
    assert np.abs(ovar[:, 0] - var).max() <= tol * max(1.0, np.abs(var).max())


def test_oracle_conditional_matches_committed_50_digit_fixtures():
    """tests/golden/mp/conditional.npz: conditional() at 50 digits (q_sqrt None / [M, K] / [M, M, K], whitened or not)."""
    import importlib.util
    spec_mod = importlib.util.spec_from_file_location("make_mp_golden", os.path.join(GOLD, "mp", "make_mp_golden.py"))
    mod = importlib.util.module_from_spec(spec_mod); spec_mod.loader.exec_module(mod)
    g = np.load(os.path.join(GOLD, "mp", "conditional.npz"))
    for name, white, q, fc in mod.COND_CASES:
        qs = {"none": None, "diag": g[name + "_qdiag"], "full": g[name + "_qfull"]}[q]
        mu, var = orc.conditional(g[name + "_Xn"], g[name + "_Z"], mod.SPECS[name], g[name + "_f"], full_cov=fc, q_sqrt=qs, white=white)
        tag = "%s_%s_%s_%s" % (name, "white" if white else "unwhite", q, "fullcov" if fc else "diag")
        assert np.abs(mu - g[tag + "_mu"]).max() <= 1e-8 * max(1.0, np.abs(g[tag + "_mu"]).max()), tag
        assert np.abs(var - g[tag + "_var"]).max() <= 1e-8 * max(1.0, np.abs(g[tag + "_var"]).max()), tag


def test_oracle_svgp_bound_matches_committed_50_digit_fixtures():
    """tests/golden/mp/svgp.npz: gauss_kl and the SVGP bound (Gaussian likelihood) at 50 digits."""
    import importlib.util
    spec_mod = importlib.util.spec_from_file_location("make_mp_golden", os.path.join(GOLD, "mp", "make_mp_golden.py"))
    mod = importlib.util.module_from_spec(spec_mod); spec_mod.loader.exec_module(mod)
    g = np.load(os.path.join(GOLD, "mp", "svgp.npz"))
    for name, white, q in mod.SVGP_CASES:
        X, Y, Z, q_mu = g[name + "_X"], g[name + "_Y"], g[name + "_Z"], g[name + "_qmu"]
        qs = g[name + ("_qdiag" if q == "diag" else "_qfull")]
        tag = "%s_%s_%s" % (name, "white" if white else "unwhite", q)
        Kp = None if white else orc.K(mod.SPECS[name], Z) + 1e-6 * np.eye(Z.shape[0])
        assert orc.gauss_kl(q_mu, qs, Kp) == pytest.approx(float(g[tag + "_kl"]), rel=1e-8), tag
        el = orc.svgp_elbo(mod.SPECS[name], X, Y, Z, q_mu, qs, mod.SVGP_NOISE, whiten=white, num_data=mod.SVGP_NUM_DATA)
        assert el == pytest.approx(float(g[tag + "_elbo"]), rel=1e-8), tag
    # the SGPR bound and the FITC likelihood against their dense definitions
    for name in ("rbf_ard", "matern52"):
        X, Y, Z = g[name + "_X"], g[name + "_Y"], g[name + "_Z"]
        assert orc.sgpr_bound(mod.SPECS[name], X, Y, Z, mod.SVGP_NOISE) == pytest.approx(float(g[name + "_sgpr_bound"]), rel=1e-8)
        assert orc.fitc_lml(mod.SPECS[name], X, Y, Z, mod.SVGP_NOISE) == pytest.approx(float(g[name + "_fitc_lml"]), rel=1e-8)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "mp", "*_n*.npz"))))
def test_oracle_matches_committed_50_digit_fixtures(path):
    """tests/golden/mp/*.npz (make_mp_golden.py: mpmath only, no oracle): the same fixtures the HIP path is compared with
    directly in tests/test_gpu_pins.py."""
    import importlib.util
    spec_mod = importlib.util.spec_from_file_location("make_mp_golden", os.path.join(GOLD, "mp", "make_mp_golden.py"))
    mod = importlib.util.module_from_spec(spec_mod); spec_mod.loader.exec_module(mod)
    g = np.load(path)
    spec = mod.SPECS[os.path.basename(path)[:-4].rsplit("_n", 1)[0]]
    s2 = float(g["noise_var"])
    assert orc.gpr_lml(spec, g["X"], g["Y"], s2) == pytest.approx(float(g["lml"]), rel=1e-8)
    omu, ovar = orc.gpr_predict(spec, g["X"], g["Y"], s2, g["Xs"])
    assert np.abs(omu - g["mu"]).max() <= 1e-8 * max(1.0, np.abs(g["mu"]).max())
    assert np.abs(ovar - g["var"][:, None]).max() <= 1e-8 * max(1.0, np.abs(g["var"]).max())


# ---- independent implementation: scikit-learn -------------------------------------------------
def test_sklearn_cross_check_rbf_and_matern():
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern
    rng = np.random.default_rng(5)
    n, d = 120, 3
    X = rng.standard_normal((n, d)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    Xs = rng.standard_normal((15, d))
    ls = c(np.array([0.8, 1.3, 2.1])); v = c(1.4); s2 = c(0.1)
    for t, sk, tol in [("rbf", ConstantKernel(v, "fixed") * RBF(ls, "fixed"), 1e-12),
                       ("matern52", ConstantKernel(v, "fixed") * Matern(ls, "fixed", nu=2.5), 5e-9)]:
        spec = {"type": t, "variance": v, "lengthscales": ls, "input_dim": d}
        gp = GaussianProcessRegressor(kernel=sk, alpha=float(s2), optimizer=None).fit(X, Y)
        assert orc.gpr_lml(spec, X, Y, s2) == pytest.approx(gp.log_marginal_likelihood_value_, rel=tol)
        mu, std = gp.predict(Xs, return_std=True)
        omu, ovar = orc.gpr_predict(spec, X, Y, s2, Xs)
        assert np.abs(omu[:, 0] - mu.ravel()).max() <= 1e-9
        assert np.abs(ovar[:, 0] - std ** 2).max() <= 1e-8


# ---- the reference's own structural tests, restated --------------------------------------------
def test_cholesky_logp_equals_feature_logp():
    """gpflowSlim/densities.py:159-174 (Test_multivariate_normal_feature.test_logp) in fp64."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((10, 3)); C = rng.standard_normal((10, 5)); var = 2.0
    cov = C @ C.T + var * np.eye(10)
    L = np.linalg.cholesky(cov)
    logp1 = orc.multivariate_normal(x, np.zeros_like(x), L)
    # densities.py:98-124 restated (Woodbury / matrix determinant lemma)
    n, r = x.shape
    M = np.eye(5) * var + C.T @ C
    quad = (np.sum(x * x) - np.sum((C.T @ x) * np.linalg.solve(M, C.T @ x))) / var
    logdet = np.linalg.slogdet(M)[1] + (n - 5) * np.log(var)
    logp2 = -0.5 * n * r * np.log(2 * np.pi) - 0.5 * r * logdet - 0.5 * quad
    assert logp1 == pytest.approx(logp2, rel=1e-12)


def test_cholesky_predictor_equals_woodbury_predictor():
    """gpflowSlim/models/gpr.py:135-203 (TestPredict): K = feat feat^T + 2 I."""
    rng = np.random.default_rng(10)
    feat = rng.standard_normal((20, 5)); feat_new = rng.standard_normal((10, 5))
    Y = rng.standard_normal((20, 2)); var = 2.0
    K = feat @ feat.T + var * np.eye(20)
    L = np.linalg.cholesky(K)
    A = sl.solve_triangular(L, feat @ feat_new.T, lower=True)
    V = sl.solve_triangular(L, Y, lower=True)
    mean1 = A.T @ V
    cov1 = feat_new @ feat_new.T - A.T @ A
    Minv = np.linalg.inv(feat.T @ feat + var * np.eye(5))
    W = (feat.T - feat.T @ feat @ Minv @ feat.T) / var            # models/gpr.py:97-102
    mean2 = feat_new @ (W @ Y)
    cov2 = feat_new @ feat_new.T - feat_new @ (W @ feat) @ feat_new.T
    assert np.allclose(mean1, mean2, rtol=1e-10, atol=1e-12) and np.allclose(cov1, cov2, rtol=1e-9, atol=1e-11)


def test_base_conditional_identities():
    rng = np.random.default_rng(2)
    m, n, k = 12, 7, 2
    spec = {"type": "matern52", "variance": c(1.0), "lengthscales": c(1.2), "input_dim": 2}
    Z = rng.standard_normal((m, 2)); Xn = rng.standard_normal((n, 2)); f = rng.standard_normal((m, k))
    Kmm = orc.K(spec, Z) + 1e-6 * np.eye(m)
    # unwhitened f  <->  whitened v = L^-1 f
    mu_u, var_u = orc.conditional(Xn, Z, spec, f, white=False)
    v = sl.solve_triangular(np.linalg.cholesky(Kmm), f, lower=True)
    mu_w, var_w = orc.conditional(Xn, Z, spec, v, white=True)
    assert np.allclose(mu_u, mu_w, rtol=1e-8, atol=1e-10) and np.allclose(var_u, var_w, rtol=1e-12)
    # q_sqrt = 0 equals q_sqrt = None; full_cov diagonal equals the diagonal variance
    mu0, var0 = orc.conditional(Xn, Z, spec, f, q_sqrt=np.zeros((m, k)))
    assert np.array_equal(mu0, mu_u) and np.allclose(var0, var_u, rtol=0, atol=0)
    _, cov = orc.conditional(Xn, Z, spec, f, full_cov=True)
    assert cov.shape == (n, n, k) and np.allclose(np.diagonal(cov[:, :, 0]), var_u[:, 0], rtol=1e-9, atol=1e-10)


# ---- frozen fixtures -------------------------------------------------------------------------------
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))))
def test_oracle_reproduces_golden_fixtures(path):
    import importlib.util
    spec_mod = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec_mod); spec_mod.loader.exec_module(mg)
    g = np.load(path)
    name = os.path.basename(path)[:-4]
    d = g["X"].shape[1]
    spec = mg.specs(d)["rbf_ard"] if name.startswith("cfg1") else mg.specs(d)[name[len("n64_d3_"):]]
    nv = float(g["noise_var"])
    assert orc.gpr_lml(spec, g["X"], g["Y"], nv) == pytest.approx(float(g["lml"]), rel=1e-12)
    mu, var = orc.gpr_predict(spec, g["X"], g["Y"], nv, g["Xs"])
    assert np.allclose(mu, g["mu"], rtol=1e-10, atol=1e-12) and np.allclose(var, g["var"], rtol=1e-10, atol=1e-12)


def test_gauss_kl_closed_forms():
    """KL[N(m, S) || N(0, K)] = 0.5 (tr(K^-1 S) + m^T K^-1 m - M + log det K - log det S)"""
    rng = np.random.default_rng(4)
    M, k = 9, 2
    q_mu = rng.standard_normal((M, k))
    Lq = np.tril(rng.standard_normal((k, M, M)) * 0.2 + np.eye(M))
    q_sqrt3 = np.transpose(Lq, (1, 2, 0)).copy()
    G = rng.standard_normal((M, M)); K = G @ G.T + M * np.eye(M)
    ref = 0.0
    for i in range(k):
        S = Lq[i] @ Lq[i].T
        ref += 0.5 * (np.trace(np.linalg.solve(K, S)) + q_mu[:, i] @ np.linalg.solve(K, q_mu[:, i]) - M
                      + np.linalg.slogdet(K)[1] - np.linalg.slogdet(S)[1])
    assert orc.gauss_kl(q_mu, q_sqrt3, K) == pytest.approx(ref, rel=1e-12)
    assert orc.gauss_kl(np.zeros((M, k)), np.transpose(np.array([np.eye(M)] * k), (1, 2, 0)), None) == pytest.approx(0.0, abs=1e-14)
    qd = np.abs(rng.standard_normal((M, k))) + 0.1
    refd = sum(0.5 * (np.sum(np.diag(np.linalg.inv(K)) * qd[:, i] ** 2) + q_mu[:, i] @ np.linalg.solve(K, q_mu[:, i]) - M
                      + np.linalg.slogdet(K)[1] - np.sum(np.log(qd[:, i] ** 2))) for i in range(k))
    assert orc.gauss_kl(q_mu, qd, K) == pytest.approx(refd, rel=1e-12)


def test_svgp_bound_is_tight_at_the_exact_posterior():
    """With Z = X, Gaussian likelihood and q(u) = exact posterior, the SVGP bound equals the GPR LML."""
    rng = np.random.default_rng(8)
    n, d = 25, 2
    X = rng.standard_normal((n, d)); Y = rng.standard_normal((n, 1))
    spec = {"type": "rbf", "variance": c(1.3), "lengthscales": c(0.9), "input_dim": d}
    s2 = c(0.2)
    Kff = orc.K(spec, X) + orc.JITTER * np.eye(n)
    A = np.linalg.inv(Kff + s2 * np.eye(n))
    mu_u = Kff @ A @ Y
    S_u = Kff - Kff @ A @ Kff
    Lq = np.linalg.cholesky(S_u + 1e-12 * np.eye(n))
    elbo = orc.svgp_elbo(spec, X, Y, X, mu_u, Lq[:, :, None], s2, whiten=False)
    lml_jit = orc.multivariate_normal(Y, np.zeros((n, 1)), np.linalg.cholesky(Kff + s2 * np.eye(n)))
    # equal up to the O(jitter) mismatch between Kuu (+1e-6 I) and Kuf / Kdiag (no jitter); never above
    assert elbo == pytest.approx(lml_jit, rel=2e-5) and elbo <= lml_jit + 1e-9


def test_sgpr_bound_limits():
    """Z = X: the Titsias bound equals the exact LML (up to the jitter); fewer inducing points: below it."""
    rng = np.random.default_rng(12)
    n, d = 40, 2
    X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
    spec = {"type": "rbf", "variance": c(1.1), "lengthscales": c(1.0), "input_dim": d}
    s2 = c(0.1)
    lml = orc.gpr_lml(spec, X, Y, s2)
    assert orc.sgpr_bound(spec, X, Y, X, s2) == pytest.approx(lml, rel=1e-4)
    b = orc.sgpr_bound(spec, X, Y, X[:10], s2)
    assert b < lml
    mu, var = orc.sgpr_predict(spec, X, Y, X, s2, X[:5])
    rmu, rvar = orc.gpr_predict(spec, X, Y, s2, X[:5])
    assert np.allclose(mu, rmu, atol=1e-4) and np.allclose(var, rvar, atol=1e-4)


def test_fitc_and_upper_bound_limits():
    """GPRFITC (models/sgpr.py:229-318) and the upper bound (:55-82): with Z = X both collapse onto the exact GP
    (up to the Kuu jitter); with few inducing points lower bound <= exact LML <= upper bound."""
    rng = np.random.default_rng(13)
    n, d = 40, 2
    X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
    spec = {"type": "rbf", "variance": c(1.1), "lengthscales": c(1.0), "input_dim": d}
    s2 = c(0.1)
    lml = orc.gpr_lml(spec, X, Y, s2)
    assert orc.fitc_lml(spec, X, Y, X, s2) == pytest.approx(lml, rel=1e-4)
    mu, var = orc.fitc_predict(spec, X, Y, X, s2, X[:5])
    rmu, rvar = orc.gpr_predict(spec, X, Y, s2, X[:5])
    assert np.allclose(mu, rmu, atol=1e-4) and np.allclose(var, rvar, atol=1e-4)
    # FITC's likelihood is the density of N(0, Qff + diag(nu)): check against the dense form
    Z = X[:10]
    Kuf = orc.K(spec, Z, X); Kuu = orc.K(spec, Z) + orc.JITTER * np.eye(10)
    Qff = Kuf.T @ np.linalg.solve(Kuu, Kuf)
    Kfitc = Qff + np.diag(orc.Kdiag(spec, X) - np.diag(Qff) + s2)
    dense = orc.multivariate_normal(Y, np.zeros((n, 1)), np.linalg.cholesky(Kfitc))
    assert orc.fitc_lml(spec, X, Y, Z, s2) == pytest.approx(dense, rel=1e-9)
    # full-covariance prediction: its diagonal is the marginal variance
    _, cov = orc.fitc_predict(spec, X, Y, Z, s2, X[:6], full_cov=True)
    _, var6 = orc.fitc_predict(spec, X, Y, Z, s2, X[:6])
    assert np.allclose(np.diagonal(cov[:, :, 0]), var6[:, 0], atol=1e-12)
    lower = orc.sgpr_bound(spec, X, Y, Z, s2)
    upper = orc.sgpr_upper_bound(spec, X, Y, Z, s2)
    assert lower < lml < upper
    assert orc.sgpr_upper_bound(spec, X, Y, X, s2) == pytest.approx(lml, rel=1e-4)


@pytest.mark.parametrize("name", ["rbf_ard_m40", "matern52_m130"])
def test_oracle_reproduces_sparse_golden_fixtures(name):
    import importlib.util
    """tests/golden/sparse/*.npz (conditional, gauss_kl, SVGP / SGPR / FITC bounds): any drift of the oracle shows up here."""
    here = os.path.join(GOLD, "sparse")
    spec_mod = importlib.util.spec_from_file_location("make_golden_sparse", os.path.join(here, "make_golden_sparse.py"))
    mg = importlib.util.module_from_spec(spec_mod)
    spec_mod.loader.exec_module(mg)
    g = np.load(os.path.join(here, name + ".npz"))
    kind = mg.CASES[name][0]
    spec = mg.spec_for(kind, g["X"].shape[1])
    regen = mg.inputs(name)
    for k in ("X", "Y", "Z", "Xs", "q_mu", "q_diag", "q_full"):
        assert np.array_equal(regen[k], g[k]), k
    noise = float(g["noise_var"])
    m = g["Z"].shape[0]
    Kuu = orc.K(spec, g["Z"]) + orc.JITTER * np.eye(m)
    tol = max(1e-10, 4 * np.finfo(float).eps * float(g["cond_Kuu"]))       # a re-run on another BLAS may round differently
    for white in (True, False):
        for qn in ("q_diag", "q_full"):
            tag = "%s_%s" % ("white" if white else "unwhite", qn)
            mu, var = orc.conditional(g["Xs"], g["Z"], spec, g["q_mu"], q_sqrt=g[qn], white=white)
            assert np.abs(mu - g["cond_mu_" + tag]).max() <= tol * max(1.0, np.abs(mu).max())
            assert np.abs(var - g["cond_var_" + tag]).max() <= tol * max(1.0, np.abs(var).max())
            kl = orc.gauss_kl(g["q_mu"], g[qn], None if white else Kuu)
            assert abs(kl - float(g["kl_" + tag])) <= tol * abs(kl)
            el = orc.svgp_elbo(spec, g["X"], g["Y"], g["Z"], g["q_mu"], g[qn], noise, whiten=white, num_data=3 * g["X"].shape[0])
            assert abs(el - float(g["elbo_" + tag])) <= tol * abs(el)
    assert abs(orc.sgpr_bound(spec, g["X"], g["Y"], g["Z"], noise) - float(g["sgpr_bound"])) <= tol * abs(float(g["sgpr_bound"]))
    assert abs(orc.fitc_lml(spec, g["X"], g["Y"], g["Z"], noise) - float(g["fitc_lml"])) <= tol * abs(float(g["fitc_lml"]))
