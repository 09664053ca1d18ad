"""Pins of the HIP path that do NOT go through oracle/ (round 5).

Parity is unpinned by the reference (it holds no numeric vectors and cannot run here), so the oracle must not be the only thing
between the product and the truth.  Three kinds of pins, every one compared with the HIP path directly, through the C ABI:
  (a) the reference's own two structural tests (gpflowSlim/models/gpr.py:135-203 TestPredict, gpflowSlim/densities.py:159-174
      Test_multivariate_normal_feature) restated on the product's entry points: the Cholesky forms computed by gps_potrf /
      gps_trsm_lower / densities.multivariate_normal against the Woodbury forms computed here in numpy;
  (b) 50-digit mpmath evaluations of the reference's formulas (tests/golden/mp/*.npz, generator committed beside them): the GPR
      likelihood and posterior, and conditional() with every form of q_sqrt;
  (c) analytic known answers: N = 1, N = 2, far-apart points, multi-output additivity, prediction far from the data.
Nothing in this file imports oracle."""
import glob
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
MP = os.path.join(HERE, "golden", "mp")


# ---------------------------------------------------------------- (a) the reference's structural tests on the product path
@pytest.mark.parametrize("n", [20, 300, 2000])
def test_cholesky_predictor_equals_woodbury_predictor(handle, n):
    """models/gpr.py:135-203 (TestPredict): K = feat feat^T + 2 I.  "standard one" (:183-196) through gps_potrf and
    gps_trsm_lower, "use feature" (:152-181) in numpy; the reference asserts 1e-4 in fp32, fp64 gives 1e-9."""
    rng = np.random.default_rng(10 + n)
    nn, d, variance = 10, 5, 2.0
    mX = rng.standard_normal((n, 1)); m_new = rng.standard_normal((nn, 1))
    Y = rng.standard_normal((n, 1))
    feat = rng.standard_normal((n, d)); feat_new = rng.standard_normal((nn, d))
    # ---- standard one, on the device
    Kx = feat @ feat_new.T
    K = feat @ feat.T + np.eye(n) * variance
    L = handle.potrf(K)                                         # tf.cholesky                      :186
    A = handle.trsm_lower(L, Kx)                                # tf.matrix_triangular_solve        :187
    V = handle.trsm_lower(L, Y - mX)                            #                                   :188
    standard_fmean = A.T @ V + m_new
    standard_fvar_full = feat_new @ feat_new.T - A.T @ A
    standard_fvar_diag = np.diag(feat_new @ feat_new.T) - np.sum(np.square(A), 0)
    # ---- use feature (Woodbury), in numpy
    CtC_I = feat.T @ feat + np.eye(d) * variance
    tmp = (feat.T @ feat) @ (np.linalg.inv(CtC_I) @ feat.T)
    Ct_CCT_I_inv = (feat.T - tmp) / variance
    fmean = feat_new @ (Ct_CCT_I_inv @ (Y - mX)) + m_new
    fvar_full = feat_new @ feat_new.T - feat_new @ ((Ct_CCT_I_inv @ feat) @ feat_new.T)
    fvar_diag = np.sum(feat_new ** 2, -1) - np.sum((feat_new @ (Ct_CCT_I_inv @ feat)) * feat_new, -1)
    assert np.allclose(np.tril(L) @ np.tril(L).T, K, rtol=0, atol=1e-11 * n)
    assert np.allclose(fmean, standard_fmean, rtol=1e-9, atol=1e-10)
    assert np.allclose(fvar_full, standard_fvar_full, rtol=1e-9, atol=1e-9)
    assert np.allclose(fvar_diag, standard_fvar_diag, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("n,k,r", [(10, 5, 1), (300, 7, 1), (2000, 12, 3)])
def test_cholesky_logp_equals_feature_logp(handle, n, k, r):
    """densities.py:159-174 (Test_multivariate_normal_feature.test_logp): multivariate_normal with the Cholesky factor of
    C C^T + var I -- factor by gps_potrf, density by the product's densities.multivariate_normal (gps_trsm_lower) -- against
    multivariate_normal_feature (densities.py:98-124) restated in numpy (r > 1: columns independent, :93)."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(9 + n)
    C = rng.standard_normal((n, k)); var = 2.0
    x = rng.standard_normal((n, r)); mu = np.zeros((n, r))
    L = handle.potrf(C @ C.T + var * np.eye(n))
    logp1 = gpf.densities.multivariate_normal(x, mu, np.tril(L))
    CtC = C.T @ C + var * np.eye(k)
    Lf = np.linalg.cholesky(CtC)
    logdet = 2.0 * np.sum(np.log(np.diag(Lf))) + (n - k) * np.log(var)
    logp2 = 0.0
    for q in range(r):
        Ctx = C.T @ x[:, q]
        sq = (np.sum(x[:, q] ** 2) - np.sum(np.linalg.solve(Lf, Ctx) ** 2)) / var
        logp2 += -0.5 * (sq + n * np.log(2.0 * np.pi) + logdet)
    assert logp1 == pytest.approx(logp2, rel=1e-10)


# ---------------------------------------------------------------- (b) 50-digit values, straight against the HIP path
def _build(gpf, spec):
    k = gpf.kernels
    t = spec["type"]
    if t == "sum" or t == "product":
        parts = [_build(gpf, ch) for ch in spec["children"]]
        out = parts[0]
        for p in parts[1:]:
            out = out + p if t == "sum" else out * p
        return out
    if t == "constant":
        return k.Constant(4, variance=spec["variance"])
    d = spec["input_dim"]
    ad = spec.get("active_dims")
    if t == "periodic":
        return k.Periodic(d, period=spec["period"], variance=spec["variance"], lengthscales=spec["lengthscales"], active_dims=ad)
    cls = {"rbf": k.RBF, "matern12": k.Matern12, "matern32": k.Matern32, "matern52": k.Matern52}[t]
    ls = spec["lengthscales"]
    return cls(d, variance=spec["variance"], lengthscales=np.asarray(ls, dtype=float) if np.ndim(ls) else ls, ARD=bool(np.ndim(ls)), active_dims=ad)


def _mp_specs():
    sys.path.insert(0, MP)
    try:
        import make_mp_golden
    finally:
        sys.path.remove(MP)
    return make_mp_golden.SPECS


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(MP, "*_n*.npz"))))
def test_hip_path_matches_50_digit_evaluation(handle, path):
    """LML, posterior mean and variance of the HIP path against mpmath at 50 digits (tests/golden/mp/make_mp_golden.py), 1e-8."""
    import gpflowSlim as gpf
    g = np.load(path)
    name = os.path.basename(path)[:-4].rsplit("_n", 1)[0]
    spec = _mp_specs()[name]
    m = gpf.models.GPR(g["X"], g["Y"], _build(gpf, spec), obs_var=float(g["noise_var"]))
    assert abs(float(np.squeeze(m.likelihood.variance)) - float(g["noise_var"])) <= 1e-15
    lml = m.compute_log_likelihood()
    assert abs(lml - float(g["lml"])) <= 1e-8 * abs(float(g["lml"]))
    mu, var = m.predict_f(g["Xs"])
    assert mu.shape == g["mu"].shape and var.shape == g["mu"].shape
    assert np.abs(mu - g["mu"]).max() <= 1e-8 * max(1.0, np.abs(g["mu"]).max())
    assert np.abs(var - g["var"][:, None]).max() <= 1e-8 * max(1.0, np.abs(g["var"]).max())


def _cond_cases():
    sys.path.insert(0, MP)
    try:
        import make_mp_golden
    finally:
        sys.path.remove(MP)
    return make_mp_golden.COND_CASES


@pytest.mark.parametrize("name,white,q,full_cov", _cond_cases())
def test_conditional_matches_50_digit_evaluation(handle, name, white, q, full_cov):
    """conditional() (conditionals.py:24-121: Kmm + jitter, Lm, A, the unwhitened back-solve, the q_sqrt terms) against the same
    formulas at 50 digits (tests/golden/mp/conditional.npz): q_sqrt None / [M, K] / [M, M, K], whitened or not, marginal and
    full covariance -- M = 14 inducing points in 4 dimensions, where fp64 itself is good for 1e-8."""
    import gpflowSlim as gpf
    g = np.load(os.path.join(MP, "conditional.npz"))
    kern = _build(gpf, _mp_specs()[name])
    qs = {"none": None, "diag": g[name + "_qdiag"], "full": g[name + "_qfull"]}[q]
    mu, var = gpf.conditionals.conditional(g[name + "_Xn"], g[name + "_Z"], kern, g[name + "_f"], full_cov=full_cov, q_sqrt=qs, white=white)
    tag = "%s_%s_%s_%s" % (name, "white" if white else "unwhite", q, "fullcov" if full_cov else "diag")
    rmu, rvar = g[tag + "_mu"], g[tag + "_var"]
    assert mu.shape == rmu.shape and var.shape == rvar.shape
    assert np.abs(mu - rmu).max() <= 1e-8 * max(1.0, np.abs(rmu).max())
    assert np.abs(var - rvar).max() <= 1e-8 * max(1.0, np.abs(rvar).max())


def _mp_module():
    sys.path.insert(0, MP)
    try:
        import make_mp_golden
    finally:
        sys.path.remove(MP)
    return make_mp_golden


@pytest.mark.parametrize("name,white,q", _mp_module().SVGP_CASES)
def test_svgp_bound_and_kl_match_50_digit_evaluation(handle, name, white, q):
    """gauss_kl (kullback_leiblers.py:26-105) and the SVGP bound with the Gaussian likelihood (models/svgp.py:101-125,
    likelihoods.py:186-188), both parametrisations, diagonal and full q_sqrt, two latent functions, minibatch scale 3: the
    device-resident bound and the stand-alone KL against 50-digit values (tests/golden/mp/svgp.npz)."""
    import gpflowSlim as gpf
    mod = _mp_module()
    g = np.load(os.path.join(MP, "svgp.npz"))
    kern = _build(gpf, mod.SPECS[name])
    X, Y, Z, q_mu = g[name + "_X"], g[name + "_Y"], g[name + "_Z"], g[name + "_qmu"]
    qs = g[name + ("_qdiag" if q == "diag" else "_qfull")]
    tag = "%s_%s_%s" % (name, "white" if white else "unwhite", q)
    Kuu = None if white else kern.K(Z) + 1e-6 * np.eye(Z.shape[0])
    kl = gpf.kullback_leiblers.gauss_kl(q_mu, qs, Kuu)
    assert abs(kl - float(g[tag + "_kl"])) <= 1e-8 * abs(float(g[tag + "_kl"]))
    sv = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(mod.SVGP_NOISE), Z=Z, q_diag=(q == "diag"), whiten=white, num_data=mod.SVGP_NUM_DATA)
    sv._q_mu.assign(q_mu); sv._q_sqrt.assign(qs)
    assert abs(float(np.squeeze(sv.likelihood.variance)) - mod.SVGP_NOISE) <= 1e-15
    elbo = sv.compute_log_likelihood()
    assert abs(elbo - float(g[tag + "_elbo"])) <= 1e-8 * abs(float(g[tag + "_elbo"]))


@pytest.mark.parametrize("name", ["rbf_ard", "matern52"])
def test_sgpr_bound_and_fitc_likelihood_match_their_dense_definitions(handle, name):
    """models/sgpr.py:121-155 (SGPR bound) and :229-282 (FITC likelihood) against their DEFINITIONS evaluated densely at 50 digits
    -- log N(y | 0, Qff + s2 I) - tr(Kff - Qff) / (2 s2) and log N(y | 0, Qff + diag(Kff - Qff) + s2 I), two outputs --, not
    against a restatement of the reference's Woodbury algebra (tests/golden/mp/svgp.npz)."""
    import gpflowSlim as gpf
    mod = _mp_module()
    g = np.load(os.path.join(MP, "svgp.npz"))
    X, Y, Z = g[name + "_X"], g[name + "_Y"], g[name + "_Z"]
    sg = gpf.models.SGPR(X, Y, _build(gpf, mod.SPECS[name]), Z=Z, obs_var=mod.SVGP_NOISE)
    assert abs(sg.compute_log_likelihood() - float(g[name + "_sgpr_bound"])) <= 1e-8 * abs(float(g[name + "_sgpr_bound"]))
    fi = gpf.models.GPRFITC(X, Y, _build(gpf, mod.SPECS[name]), Z=Z, obs_var=mod.SVGP_NOISE)
    assert abs(fi.compute_log_likelihood() - float(g[name + "_fitc_lml"])) <= 1e-8 * abs(float(g[name + "_fitc_lml"]))


@pytest.mark.parametrize("name", ["rbf_ard", "matern52", "periodic"])
def test_lml_gradient_matches_high_precision_differences(handle, name):
    """gps_gpr_lml_grad (what TF autodiff through tf.cholesky supplies to examples/gpr.py:53-54) against central differences of the
    60-digit likelihood with a step of 1e-25 (tests/golden/mp/gradient.npz): kernel parameters (constrained values, slot order
    of the header) and the noise variance, two outputs.  Independent of every derivative formula, the oracle's included."""
    import gpflowSlim as gpf
    mod = _mp_module()
    g = np.load(os.path.join(MP, "gradient.npz"))
    theta0, fn = mod.GRAD_SPECS[name]
    kern = _build(gpf, fn(theta0))
    X, Y = g[name + "_X"], g[name + "_Y"]
    handle.gpr_set_data(X, ("pins", name))
    lml, slots, gn, _ = handle.gpr_lml_grad(kern._program(4), mod.NOISE, Y)
    assert abs(lml - float(g[name + "_lml"])) <= 1e-8 * abs(float(g[name + "_lml"]))
    slots = np.ravel(slots)
    got = {"rbf_ard": slots[:5], "matern52": np.array([slots[0], slots[1:5].sum()]), "periodic": slots[:3]}[name]   # (isotropic: one entry per active dim, summed)
    ref = g[name + "_grad"]
    assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max()), (got, ref)
    assert abs(gn - float(g[name + "_grad_noise"])) <= 1e-8 * abs(float(g[name + "_grad_noise"]))


# ---------------------------------------------------------------- (c) analytic known answers on the HIP path itself
def _vals(m):
    return float(np.squeeze(m.kern.variance)), float(np.squeeze(m.likelihood.variance))


def test_kat_single_point(handle):
    """N = 1: lml = -1/2 log 2pi - 1/2 log(s2 + s2n) - y^2 / (2 (s2 + s2n)); predicting at the point itself."""
    import gpflowSlim as gpf
    X = np.array([[0.3, -1.2]]); Y = np.array([[0.7]])
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(2, variance=1.7, lengthscales=0.9), obs_var=0.25)
    s2, s2n = _vals(m)
    ref = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(s2 + s2n) - 0.7 ** 2 / (2 * (s2 + s2n))
    assert m.compute_log_likelihood() == pytest.approx(ref, rel=1e-13)
    mu, var = m.predict_f(X)
    assert mu[0, 0] == pytest.approx(s2 * 0.7 / (s2 + s2n), rel=1e-13)
    assert var[0, 0] == pytest.approx(s2 - s2 * s2 / (s2 + s2n), rel=1e-12)


@pytest.mark.parametrize("kind", ["rbf", "matern12", "matern32", "matern52"])
def test_kat_two_points(handle, kind):
    """N = 2 in closed form: K = [[a, b], [b, a]], a = k(x, x) + s2n, b = k(x1, x2) -- with the reference's
    r = sqrt(r2 + 1e-12) for the Matern family, whose diagonal is therefore not the variance (kernels.py:426)."""
    import gpflowSlim as gpf
    X = np.array([[0.0], [0.8]]); Y = np.array([[0.4], [-1.1]])
    cls = {"rbf": gpf.kernels.RBF, "matern12": gpf.kernels.Matern12, "matern32": gpf.kernels.Matern32, "matern52": gpf.kernels.Matern52}[kind]
    m = gpf.models.GPR(X, Y, cls(1, variance=1.4, lengthscales=0.6), obs_var=0.2)
    s2, s2n = _vals(m)
    ell = float(np.squeeze(m.kern.lengthscales))

    def k(r2):
        if kind == "rbf":
            return s2 * np.exp(-0.5 * r2)
        r = np.sqrt(r2 + 1e-12)
        if kind == "matern12":
            return s2 * np.exp(-r)
        if kind == "matern32":
            return s2 * (1 + np.sqrt(3.) * r) * np.exp(-np.sqrt(3.) * r)
        return s2 * (1 + np.sqrt(5.) * r + 5. / 3. * r * r) * np.exp(-np.sqrt(5.) * r)
    a, b = k(0.0) + s2n, k((0.8 / ell) ** 2)
    det = a * a - b * b
    quad = (a * 0.4 ** 2 + 2 * b * 0.4 * 1.1 + a * 1.1 ** 2) / det
    ref = -np.log(2 * np.pi) - 0.5 * np.log(det) - 0.5 * quad
    assert m.compute_log_likelihood() == pytest.approx(ref, rel=1e-12)


@pytest.mark.parametrize("n", [3, 200, 700])
def test_kat_far_apart_points_and_far_prediction(handle, n):
    """Points 1e3 length-scales apart: K = s2 I exactly (exp underflows), so the likelihood is n independent Gaussians of
    variance s2 + s2n; a test point far from all of them sees the prior: (mean function, s2)."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n)
    X = 1e3 * np.arange(n, dtype=float)[:, None] * np.ones((1, 2))
    Y = rng.standard_normal((n, 1))
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(2, variance=0.8, lengthscales=1.0), obs_var=0.3,
                       mean_function=gpf.mean_functions.Constant(np.array([0.25])))
    s2, s2n = _vals(m)
    res = Y - 0.25
    ref = -0.5 * n * np.log(2 * np.pi) - 0.5 * n * np.log(s2 + s2n) - 0.5 * np.sum(res ** 2) / (s2 + s2n)
    assert m.compute_log_likelihood() == pytest.approx(ref, rel=1e-13)
    mu, var = m.predict_f(np.array([[-5e4, 7e4]]))
    assert mu[0, 0] == pytest.approx(0.25, abs=1e-14) and var[0, 0] == pytest.approx(s2, rel=1e-14)
    # ... and AT a data point: the one-point posterior of that point alone
    mu, var = m.predict_f(X[1:2])
    assert mu[0, 0] == pytest.approx(0.25 + s2 * res[1, 0] / (s2 + s2n), rel=1e-12)
    assert var[0, 0] == pytest.approx(s2 - s2 * s2 / (s2 + s2n), rel=1e-12)


@pytest.mark.parametrize("n", [20, 600, 3000])
def test_kat_multi_output_counts_logdet_r_times(handle, n):
    """densities.py:93: R outputs share one factor -- the log-determinant counts R times, the quadratic forms add."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 2)); Y = rng.standard_normal((n, 3))
    kern = lambda: gpf.kernels.Matern32(2, variance=1.0, lengthscales=1.0)
    total = gpf.models.GPR(X, Y, kern(), obs_var=0.1).compute_log_likelihood()
    parts = sum(gpf.models.GPR(X, Y[:, j:j + 1], kern(), obs_var=0.1).compute_log_likelihood() for j in range(3))
    assert total == pytest.approx(parts, rel=1e-12)
