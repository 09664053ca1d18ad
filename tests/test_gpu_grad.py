"""Gradient of the log-marginal likelihood ("next" row 1) against the oracle:
1/2 tr((a a^T - R K^-1) dK/dtheta) with dK/dtheta by central differences of the oracle's K, and
against finite differences of the product's own LML through the unconstrained parameters."""
import numpy as np
import pytest

import oracle.gp_oracle as orc

pytestmark = pytest.mark.gpu
c = orc.constrained


def _cases(gpf, d):
    k = gpf.kernels
    ls = np.linspace(0.8, 1.7, d)

    def rbf_ard():
        kern = k.RBF(d, variance=1.3, lengthscales=ls, ARD=True)
        theta = np.concatenate([[c(1.3)], c(ls)])
        fn = lambda t: {"type": "rbf", "variance": t[0], "lengthscales": t[1:], "input_dim": d}
        return kern, theta, fn, [("v", 0)] + [("l", i) for i in range(d)]

    def m52_iso():
        kern = k.Matern52(d, variance=0.9, lengthscales=1.4)
        theta = np.array([c(0.9), c(1.4)])
        fn = lambda t: {"type": "matern52", "variance": t[0], "lengthscales": t[1], "input_dim": d}
        return kern, theta, fn, None

    def m32_ard():
        kern = k.Matern32(d, variance=1.1, lengthscales=ls * 1.3, ARD=True)
        theta = np.concatenate([[c(1.1)], c(ls * 1.3)])
        fn = lambda t: {"type": "matern32", "variance": t[0], "lengthscales": t[1:], "input_dim": d}
        return kern, theta, fn, None

    def m12_iso():
        kern = k.Matern12(d, variance=0.7, lengthscales=2.0)
        theta = np.array([c(0.7), c(2.0)])
        fn = lambda t: {"type": "matern12", "variance": t[0], "lengthscales": t[1], "input_dim": d}
        return kern, theta, fn, None

    def periodic():
        kern = k.Periodic(d, period=2.5, variance=0.8, lengthscales=1.2)
        theta = np.array([c(0.8), c(1.2), c(2.5)])
        fn = lambda t: {"type": "periodic", "variance": t[0], "lengthscales": t[1], "period": t[2], "input_dim": d}
        return kern, theta, fn, None

    def m52_plus_periodic():       # BASELINE config 4 kernel
        kern = k.Matern52(d, variance=1.1, lengthscales=ls * 1.5, ARD=True) + k.Periodic(d, period=2.0, variance=0.9, lengthscales=1.2)
        theta = np.concatenate([[c(1.1)], c(ls * 1.5), [c(0.9), c(1.2), c(2.0)]])
        fn = lambda t: {"type": "sum", "children": [
            {"type": "matern52", "variance": t[0], "lengthscales": t[1:1 + d], "input_dim": d},
            {"type": "periodic", "variance": t[1 + d], "lengthscales": t[2 + d], "period": t[3 + d], "input_dim": d}]}
        return kern, theta, fn, None

    def rbf_times_periodic_plus_white():
        kern = k.RBF(d, variance=1.2, lengthscales=1.6) * k.Periodic(d, period=3.0, variance=0.9, lengthscales=1.5) + k.White(d, variance=0.2)
        theta = np.array([c(1.2), c(1.6), c(0.9), c(1.5), c(3.0), c(0.2)])
        fn = lambda t: {"type": "sum", "children": [{"type": "product", "children": [
            {"type": "rbf", "variance": t[0], "lengthscales": t[1], "input_dim": d},
            {"type": "periodic", "variance": t[2], "lengthscales": t[3], "period": t[4], "input_dim": d}]},
            {"type": "white", "variance": t[5]}]}
        return kern, theta, fn, None

    return {"rbf_ard": rbf_ard, "m52_iso": m52_iso, "m32_ard": m32_ard, "m12_iso": m12_iso, "periodic": periodic,
            "m52_plus_periodic": m52_plus_periodic, "rbf_times_periodic_plus_white": rbf_times_periodic_plus_white}


def _flat_constrained_grad(model, grads):
    """d LML / d(constrained) per kernel parameter element, in kern.parameters order, from the returned
    unconstrained gradients (divide the chain factor back out)."""
    out = []
    for p, g in grads:
        if p in model.kern.parameters:
            out.append(np.atleast_1d(g / p.transform.forward_grad(p.vf_val)).ravel())
    return np.concatenate(out)


KINDS = ["rbf_ard", "m52_iso", "m32_ard", "m12_iso", "periodic", "m52_plus_periodic", "rbf_times_periodic_plus_white"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n,d,r", [(60, 2, 1), (200, 3, 2), (515, 4, 1)])
def test_lml_gradient_matches_oracle(handle, kind, n, d, r):
    import gpflowSlim as gpf
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d))
    Y = np.sin(X @ rng.standard_normal((d, r))) + 0.1 * rng.standard_normal((n, r))
    kern, theta, fn, _ = _cases(gpf, d)[kind]()
    m = gpf.models.GPR(X, Y, kern, obs_var=0.15)
    noise = c(0.15)
    lml, grads = m.compute_log_likelihood_and_gradients()
    ref_lml = orc.gpr_lml(fn(theta), X, Y, noise)
    assert abs(lml - ref_lml) <= 1e-8 * abs(ref_lml)
    g_ref, gnoise_ref, a_ref = orc.gpr_lml_grad(fn, theta, X, Y, noise)
    got = _flat_constrained_grad(m, grads)
    assert got.shape == g_ref.shape
    scale = max(1.0, np.abs(g_ref).max())
    assert np.abs(got - g_ref).max() <= 2e-6 * scale, (got, g_ref)      # oracle dK is a central difference
    gn = [g for p, g in grads if p is m.likelihood._variance][0]
    gn_c = float(gn / m.likelihood._variance.transform.forward_grad(m.likelihood._variance.vf_val))
    assert abs(gn_c - gnoise_ref) <= 1e-8 * max(1.0, abs(gnoise_ref))      # exact formula on both sides


def test_gradient_consistent_with_finite_differences_of_the_product(handle):
    """d LML / d(unconstrained) including the softplus chain rule and a Linear mean function."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(2)
    n, d = 150, 3
    X = rng.standard_normal((n, d)); Y = X @ rng.standard_normal((d, 1)) + 0.2 * rng.standard_normal((n, 1))
    kern = gpf.kernels.RBF(d, variance=1.5, lengthscales=[0.9, 1.4, 2.0], ARD=True) + gpf.kernels.Constant(d, variance=0.3)
    mf = gpf.mean_functions.Linear(rng.standard_normal((d, 1)) * 0.1, np.array([0.05]))
    m = gpf.models.GPR(X, Y, kern, mean_function=mf, obs_var=0.2)
    lml, grads = m.compute_log_likelihood_and_gradients()
    for p, g in grads:
        flat = np.atleast_1d(p.vf_val).ravel().copy()
        gflat = np.atleast_1d(g).ravel()
        for i in range(flat.size):
            h = 1e-5
            x0 = flat[i]
            flat[i] = x0 + h; p.assign_unconstrained(flat.reshape(p.vf_val.shape)); fp = m.compute_log_likelihood()
            flat[i] = x0 - h; p.assign_unconstrained(flat.reshape(p.vf_val.shape)); fm = m.compute_log_likelihood()
            flat[i] = x0; p.assign_unconstrained(flat.reshape(p.vf_val.shape))
            fd = (fp - fm) / (2 * h)
            assert abs(gflat[i] - fd) <= 1e-5 * max(1.0, abs(fd)), (p.name, i, gflat[i], fd)


def test_gradient_too_many_primitives_is_a_loud_error(handle):
    import gpflowSlim as gpf
    d = 2
    k = gpf.kernels
    kern = (k.RBF(d) + k.Matern32(d) + k.Matern52(d) + k.Periodic(d) + k.Matern12(d) + k.RBF(d, lengthscales=2.0)
            + k.Matern32(d, lengthscales=0.5) + k.White(d) + k.Constant(d))
    X = np.random.default_rng(0).standard_normal((40, d)); Y = np.ones((40, 1))
    m = gpf.models.GPR(X, Y, kern)
    with pytest.raises(RuntimeError, match="more than 8 primitive"):
        m.compute_log_likelihood_and_gradients()


def _fd_check(m, grads, tol=1e-5, h=1e-5, max_per_param=6):
    """central differences of the product's own LML through the unconstrained parameters"""
    rng = np.random.default_rng(0)
    for p, g in grads:
        flat = np.atleast_1d(p.vf_val).ravel().copy()
        gflat = np.atleast_1d(g).ravel()
        idxs = range(flat.size) if flat.size <= max_per_param else rng.choice(flat.size, max_per_param, replace=False)
        for i in idxs:
            x0 = flat[i]
            flat[i] = x0 + h; p.assign_unconstrained(flat.reshape(p.vf_val.shape)); fp = m.compute_log_likelihood()
            flat[i] = x0 - h; p.assign_unconstrained(flat.reshape(p.vf_val.shape)); fm = m.compute_log_likelihood()
            flat[i] = x0; p.assign_unconstrained(flat.reshape(p.vf_val.shape))
            fd = (fp - fm) / (2 * h)
            assert abs(gflat[i] - fd) <= tol * max(1.0, abs(fd)), (p.name, i, gflat[i], fd)


def test_gradient_six_primitive_sum_product(handle):
    """More than four primitives (csrc/grad_general.hip): oracle (central differences of the oracle's K) and finite
    differences of the product's own LML."""
    import gpflowSlim as gpf
    d = 3
    k = gpf.kernels
    rng = np.random.default_rng(8)
    n = 180
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, 1))) + 0.1 * rng.standard_normal((n, 1))
    ls = np.linspace(0.8, 1.7, d)
    kern = (k.RBF(d, variance=1.3, lengthscales=ls, ARD=True) * k.Periodic(d, period=2.5, variance=0.8, lengthscales=1.2)
            + k.Matern52(d, variance=0.9, lengthscales=1.4) * k.Matern12(1, variance=0.7, lengthscales=2.0, active_dims=[1])
            + k.Matern32(d, variance=1.1, lengthscales=1.3) + k.White(d, variance=0.2))
    theta = np.concatenate([[c(1.3)], c(ls), [c(0.8), c(1.2), c(2.5)], [c(0.9), c(1.4)], [c(0.7), c(2.0)], [c(1.1), c(1.3)], [c(0.2)]])

    def fn(t):
        return {"type": "sum", "children": [
            {"type": "product", "children": [{"type": "rbf", "variance": t[0], "lengthscales": t[1:1 + d], "input_dim": d},
                                             {"type": "periodic", "variance": t[1 + d], "lengthscales": t[2 + d], "period": t[3 + d], "input_dim": d}]},
            {"type": "product", "children": [{"type": "matern52", "variance": t[4 + d], "lengthscales": t[5 + d], "input_dim": d},
                                             {"type": "matern12", "variance": t[6 + d], "lengthscales": t[7 + d], "input_dim": 1, "active_dims": [1]}]},
            {"type": "matern32", "variance": t[8 + d], "lengthscales": t[9 + d], "input_dim": d},
            {"type": "white", "variance": t[10 + d]}]}
    m = gpf.models.GPR(X, Y, kern, obs_var=0.15)
    lml, grads = m.compute_log_likelihood_and_gradients()
    ref_lml = orc.gpr_lml(fn(theta), X, Y, c(0.15))
    assert abs(lml - ref_lml) <= 1e-8 * abs(ref_lml)
    g_ref, gnoise_ref, _ = orc.gpr_lml_grad(fn, theta, X, Y, c(0.15))
    got = _flat_constrained_grad(m, grads)
    assert got.shape == g_ref.shape
    assert np.abs(got - g_ref).max() <= 2e-6 * max(1.0, np.abs(g_ref).max()), (got, g_ref)
    _fd_check(m, grads)


def _nkn_model(gpf, d, act, n, seed):
    from test_gpu_parity import _nkn_case
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, 1))) + 0.1 * rng.standard_normal((n, 1))
    kern, spec = _nkn_case(gpf, d, act)
    return gpf.models.GPR(X, Y, kern, obs_var=0.1), spec, X, Y


@pytest.mark.parametrize("act", [False, True])
def test_nkn_gradient_matches_oracle_and_finite_differences(handle, act):
    """Neural Kernel Network (6 primitives -> Linear 6->8 -> Product(2) -> Linear 4->2 -> [exp] -> Linear 2->1): gradient
    w.r.t. every Linear weight and bias (neural_kernel_network_wrapper.py:90-120 trains them through autodiff) and every
    primitive parameter, against the oracle (1/2 tr(W dK) with dK by central differences of the oracle's own network
    forward) and against finite differences of the product's LML."""
    import copy
    import gpflowSlim as gpf
    d, n = 3, 160
    m, spec, X, Y = _nkn_model(gpf, d, act, n, 31)
    noise = c(0.1)
    lml, grads = m.compute_log_likelihood_and_gradients()
    assert abs(lml - orc.gpr_lml(spec, X, Y, noise)) <= 1e-8 * abs(lml)
    # oracle: perturb the entries of the spec in kern.parameters order (wrapper parameters first, then the primitives')
    wrapper = m.kern._nknWrapper
    lin_idx = [i for i, l in enumerate(spec["layers"]) if l[0] == "linear"]
    handles = []            # (getter / setter over a deep-copied spec) per scalar parameter, kern.parameters order
    for li in lin_idx:
        Wl, bl = spec["layers"][li][1], spec["layers"][li][2]
        handles += [("W", li, idx) for idx in np.ndindex(Wl.shape)] + [("b", li, (o,)) for o in range(bl.size)]
    for pi, ps in enumerate(spec["primitives"]):
        handles.append(("pv", pi, None))
        if "lengthscales" in ps:
            handles += [("pl", pi, q) for q in range(np.atleast_1d(ps["lengthscales"]).size)]
        if "period" in ps:
            handles.append(("pp", pi, None))
    theta = []
    for kind, a, idx in handles:
        if kind == "W": theta.append(spec["layers"][a][1][idx])
        elif kind == "b": theta.append(spec["layers"][a][2][idx])
        elif kind == "pv": theta.append(spec["primitives"][a]["variance"])
        elif kind == "pl": theta.append(np.atleast_1d(spec["primitives"][a]["lengthscales"])[idx])
        else: theta.append(spec["primitives"][a]["period"])
    theta = np.array(theta, dtype=np.float64)

    def fn(t):
        sp_ = copy.deepcopy(spec)
        layers = [list(l) for l in sp_["layers"]]
        for (kind, a, idx), v in zip(handles, t):
            if kind == "W": layers[a][1] = np.array(layers[a][1], dtype=np.float64); layers[a][1][idx] = v
            elif kind == "b": layers[a][2] = np.array(layers[a][2], dtype=np.float64); layers[a][2][idx] = v
            elif kind == "pv": sp_["primitives"][a]["variance"] = v
            elif kind == "pl":
                ls_ = np.array(np.atleast_1d(sp_["primitives"][a]["lengthscales"]), dtype=np.float64); ls_[idx] = v
                sp_["primitives"][a]["lengthscales"] = ls_ if ls_.size > 1 else float(ls_[0])
            else: sp_["primitives"][a]["period"] = v
        sp_["layers"] = [tuple(l) for l in layers]
        return sp_
    g_ref, gnoise_ref, _ = orc.gpr_lml_grad(fn, theta, X, Y, noise)
    got = _flat_constrained_grad(m, grads)
    assert got.shape == g_ref.shape, (got.shape, g_ref.shape)
    assert np.abs(got - g_ref).max() <= 2e-6 * max(1.0, np.abs(g_ref).max()), (np.abs(got - g_ref).max(), got, g_ref)
    _fd_check(m, grads, max_per_param=4)


def test_nkn_gpr_fit_lowers_the_objective(handle):
    """NeuralKernelNetwork + GPR.optimize(): the fit the reference's examples run (AdamOptimizer.minimize(objective),
    examples/gpr.py:53-54) is possible on the analytic gradients."""
    import gpflowSlim as gpf
    m, spec, X, Y = _nkn_model(gpf, 3, False, 250, 7)
    start = m.objective
    final = m.optimize(max_iter=40)
    assert final < start - 10.0, (start, final)


@pytest.mark.gpu
def test_gpr_optimize_lbfgs_and_adam():
    """GPR.optimize (models/model.py:172-196 with examples/gpr.py's model): L-BFGS-B on the analytic gradients
    reaches a stationary point of the LML (small projected gradient, objective far below the start), and the
    fitted model explains held-out data; Adam lowers the objective too."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(5)
    n, d = 300, 3
    X = rng.standard_normal((n, d)); w = np.array([[1.5], [0.0], [-0.7]])
    f = lambda A: np.sin(A @ w)
    Y = f(X) + 0.1 * rng.standard_normal((n, 1))
    Xs = rng.standard_normal((100, d))
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, ARD=True))
    start = m.objective
    final = m.optimize(max_iter=200)
    assert final < start - 50.0
    assert m.objective == pytest.approx(final, rel=1e-10)
    _, grads = m.compute_log_likelihood_and_gradients()
    gnorm = max(float(np.max(np.abs(g))) for _, g in grads)
    assert gnorm < 1e-2 * max(1.0, abs(final)), gnorm
    ls = np.asarray(m.kern.lengthscales)
    assert ls[1] > 3.0 * max(ls[0], ls[2])                   # the irrelevant input is switched off
    assert 0.005 < float(np.squeeze(m.likelihood.variance)) < 0.03      # noise variance 0.01 recovered
    mu, var = m.predict_f(Xs)
    assert np.sqrt(np.mean((mu - f(Xs)) ** 2)) < 0.1
    m2 = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, ARD=True))
    s2 = m2.objective
    f2 = m2.optimize(max_iter=150, method="adam", learning_rate=0.05)
    assert f2 < s2 - 50.0


@pytest.mark.parametrize("q_diag", [False, True])
@pytest.mark.parametrize("kind,n,m_,d,k", [("rbf_ard", 300, 40, 3, 2), ("m52_plus_periodic", 500, 150, 2, 1), ("m32_ard", 260, 130, 2, 3)])
def test_svgp_bound_gradient(handle, q_diag, kind, n, m_, d, k):
    """Gradient of the SVGP bound (whitened, Gaussian likelihood; gps_svgp_elbo_grad) w.r.t. the kernel parameters, the
    noise variance, q_mu and q_sqrt against the oracle's matrix-level reverse mode (dK by central differences of the
    oracle's own K) and against finite differences of the product's own bound through the unconstrained parameters."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n + m_ + k)
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, k))) + 0.1 * rng.standard_normal((n, k))
    Z = X[:m_].copy()
    kern, theta, fn, _ = _cases(gpf, d)[kind]()
    m = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.3), Z=Z, q_diag=q_diag, whiten=True, num_data=3 * n)
    q_mu = rng.standard_normal((m_, k)) * 0.3
    m._q_mu.assign(q_mu)
    if q_diag:
        q_sqrt = np.abs(rng.standard_normal((m_, k))) * 0.4 + 0.2
    else:
        q_sqrt = np.tril(rng.standard_normal((k, m_, m_)) * (0.5 / m_) + np.eye(m_) * 0.5).transpose(1, 2, 0).copy()
    m._q_sqrt.assign(q_sqrt)
    noise = c(0.3)
    bound, grads = m.compute_log_likelihood_and_gradients()
    assert abs(bound - m.compute_log_likelihood()) <= 1e-12 * abs(bound)
    g_ref, gn_ref, gq_ref, gs_ref, _ = orc.svgp_elbo_grad(fn, theta, X, Y, Z, q_mu, np.asarray(m.q_sqrt), noise, num_data=3 * n)
    scale = max(1.0, np.abs(g_ref).max())
    got = _flat_constrained_grad(m, grads)
    assert got.shape == g_ref.shape
    assert np.abs(got - g_ref).max() <= 5e-6 * scale, (got, g_ref)
    by = {id(p): g for p, g in grads}
    gn = float(by[id(m.likelihood._variance)] / m.likelihood._variance.transform.forward_grad(m.likelihood._variance.vf_val))
    assert abs(gn - gn_ref) <= 1e-7 * max(1.0, abs(gn_ref))
    assert np.abs(by[id(m._q_mu)] - gq_ref).max() <= 1e-7 * max(1.0, np.abs(gq_ref).max())
    if q_diag:
        gs = by[id(m._q_sqrt)] / m._q_sqrt.transform.forward_grad(m._q_sqrt.vf_val)
        assert np.abs(gs - gs_ref).max() <= 1e-7 * max(1.0, np.abs(gs_ref).max())
    else:
        rows, cols = np.tril_indices(m_, 0)
        ref_free = np.stack([gs_ref[rows, cols, q] for q in range(k)])
        assert np.abs(by[id(m._q_sqrt)] - ref_free).max() <= 1e-7 * max(1.0, np.abs(ref_free).max())
    # finite differences of the product's own bound (a few entries per parameter)
    rng2 = np.random.default_rng(1)
    for p, g in grads:
        flat = np.atleast_1d(p.vf_val).ravel().copy()
        gflat = np.atleast_1d(g).ravel()
        for i in (range(flat.size) if flat.size <= 4 else rng2.choice(flat.size, 4, replace=False)):
            h = 1e-5
            x0 = flat[i]
            flat[i] = x0 + h; p.assign_unconstrained(flat.reshape(p.vf_val.shape)); fp = m.compute_log_likelihood()
            flat[i] = x0 - h; p.assign_unconstrained(flat.reshape(p.vf_val.shape)); fm = m.compute_log_likelihood()
            flat[i] = x0; p.assign_unconstrained(flat.reshape(p.vf_val.shape))
            fd = (fp - fm) / (2 * h)
            assert abs(gflat[i] - fd) <= 2e-5 * max(1.0, abs(fd)), (p.name, i, gflat[i], fd)

@pytest.mark.parametrize("q_diag", [False, True])
@pytest.mark.parametrize("kind,n,m_,d,k", [("rbf_ard", 300, 40, 3, 2), ("m32_ard", 260, 130, 2, 1)])
def test_svgp_bound_gradient_unwhitened(handle, q_diag, kind, n, m_, d, k):
    """whiten=False (what examples/svgp.py:146 runs): the gradient of the bound -- the whitened gradient at
    (Lm^-1 q_mu, Lm^-1 L_q) pulled back, gps_svgp_elbo_grad white == 0 -- against central differences of the ORACLE's
    unwhitened bound in the constrained parameters, and of the product's own bound in the unconstrained ones."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n + m_ + k + 7)
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, k))) + 0.1 * rng.standard_normal((n, k))
    Z = X[:m_].copy()
    kern, theta, fn, _ = _cases(gpf, d)[kind]()
    m = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.3), Z=Z, q_diag=q_diag, whiten=False, num_data=3 * n)
    q_mu = rng.standard_normal((m_, k)) * 0.3
    m._q_mu.assign(q_mu)
    if q_diag:
        q_sqrt = np.abs(rng.standard_normal((m_, k))) * 0.4 + 0.2
    else:
        q_sqrt = np.tril(rng.standard_normal((k, m_, m_)) * (0.5 / m_) + np.eye(m_) * 0.5).transpose(1, 2, 0).copy()
    m._q_sqrt.assign(q_sqrt)
    q_sqrt = np.asarray(m.q_sqrt).copy()
    noise = c(0.3)
    bound, grads = m.compute_log_likelihood_and_gradients()
    ref_bound = orc.svgp_elbo(fn(theta), X, Y, Z, q_mu, q_sqrt, noise, whiten=False, num_data=3 * n)
    assert abs(bound - ref_bound) <= 1e-8 * abs(ref_bound)
    assert abs(bound - m.compute_log_likelihood()) <= 1e-9 * abs(bound)         # (forward of the gradient call: whitened form)
    by = {id(p): g for p, g in grads}

    def f(th=theta, nz=noise, qm=q_mu, qs=q_sqrt):
        return orc.svgp_elbo(fn(th), X, Y, Z, qm, qs, nz, whiten=False, num_data=3 * n)

    def cd(make, x0, hrel=1e-6):
        h = hrel * max(1.0, abs(x0))
        return (make(x0 + h) - make(x0 - h)) / (2 * h)

    # kernel parameters (constrained)
    got = _flat_constrained_grad(m, grads)
    assert got.shape == theta.shape
    for i in range(theta.size):
        def mk(v, i=i):
            th = theta.copy(); th[i] = v
            return f(th=th)
        fd = cd(mk, theta[i])
        assert abs(got[i] - fd) <= 2e-5 * max(1.0, abs(fd)), ("theta", i, got[i], fd)
    gn = float(by[id(m.likelihood._variance)] / m.likelihood._variance.transform.forward_grad(m.likelihood._variance.vf_val))
    fd = cd(lambda v: f(nz=v), noise)
    assert abs(gn - fd) <= 2e-5 * max(1.0, abs(fd))
    rng2 = np.random.default_rng(2)
    gq = by[id(m._q_mu)]
    for _ in range(4):
        a, q = int(rng2.integers(m_)), int(rng2.integers(k))
        def mk(v, a=a, q=q):
            qm = q_mu.copy(); qm[a, q] = v
            return f(qm=qm)
        fd = cd(mk, q_mu[a, q])
        assert abs(gq[a, q] - fd) <= 2e-5 * max(1.0, abs(fd)), ("q_mu", a, q, gq[a, q], fd)
    if q_diag:
        gs = by[id(m._q_sqrt)] / m._q_sqrt.transform.forward_grad(m._q_sqrt.vf_val)
        for _ in range(4):
            a, q = int(rng2.integers(m_)), int(rng2.integers(k))
            def mk(v, a=a, q=q):
                qs = q_sqrt.copy(); qs[a, q] = v
                return f(qs=qs)
            fd = cd(mk, q_sqrt[a, q])
            assert abs(gs[a, q] - fd) <= 2e-5 * max(1.0, abs(fd)), ("q_sqrt", a, q, gs[a, q], fd)
    else:
        rows, cols = np.tril_indices(m_, 0)
        gfree = by[id(m._q_sqrt)].reshape(k, -1)
        for _ in range(6):
            t, q = int(rng2.integers(rows.size)), int(rng2.integers(k))
            a, b = int(rows[t]), int(cols[t])
            def mk(v, a=a, b=b, q=q):
                qs = q_sqrt.copy(); qs[a, b, q] = v
                return f(qs=qs)
            fd = cd(mk, q_sqrt[a, b, q])
            assert abs(gfree[q, t] - fd) <= 2e-5 * max(1.0, abs(fd)), ("q_sqrt", a, b, q, gfree[q, t], fd)
    # finite differences of the product's own (unwhitened) bound through the unconstrained parameters
    for p, g in grads:
        flat = np.atleast_1d(p.vf_val).ravel().copy()
        gflat = np.atleast_1d(g).ravel()
        for i in (range(flat.size) if flat.size <= 3 else rng2.choice(flat.size, 3, replace=False)):
            h = 1e-5
            x0 = flat[i]
            flat[i] = x0 + h; p.assign_unconstrained(flat.reshape(p.vf_val.shape)); fp = m.compute_log_likelihood()
            flat[i] = x0 - h; p.assign_unconstrained(flat.reshape(p.vf_val.shape)); fm = m.compute_log_likelihood()
            flat[i] = x0; p.assign_unconstrained(flat.reshape(p.vf_val.shape))
            fd = (fp - fm) / (2 * h)
            assert abs(gflat[i] - fd) <= 5e-5 * max(1.0, abs(fd)), (p.name, i, gflat[i], fd)



def test_svgp_optimize_raises_the_bound(handle):
    """SVGP + Model.optimize(): the bound goes up on the analytic gradients (the fit of examples/svgp.py:159-161, with a
    Gaussian likelihood) and ends close to the exact GPR evidence it bounds."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(3)
    n, d, m_ = 400, 2, 60
    X = rng.standard_normal((n, d)); Y = np.sin(2.0 * X[:, :1]) * np.cos(X[:, 1:2]) + 0.1 * rng.standard_normal((n, 1))
    m = gpf.models.SVGP(X, Y, gpf.kernels.RBF(d, ARD=True), gpf.likelihoods.Gaussian(0.5), Z=X[:m_].copy(), whiten=True)
    start = m.objective
    final = m.optimize(max_iter=150)
    assert final < start - 100.0, (start, final)
    exact = gpf.models.GPR(X, Y, m.kern, obs_var=float(np.squeeze(m.likelihood.variance))).compute_log_likelihood()
    assert -final <= exact + 1e-6 * abs(exact)             # a lower bound on the evidence at the same hyper-parameters
    # the configuration of examples/svgp.py:146 (whiten=False) trains too and bounds the same evidence
    m2 = gpf.models.SVGP(X, Y, gpf.kernels.RBF(d, ARD=True), gpf.likelihoods.Gaussian(0.5), Z=X[:m_].copy(), whiten=False)
    start2 = m2.objective
    final2 = m2.optimize(max_iter=150)
    assert final2 < start2 - 100.0, (start2, final2)
    exact2 = gpf.models.GPR(X, Y, m2.kern, obs_var=float(np.squeeze(m2.likelihood.variance))).compute_log_likelihood()
    assert -final2 <= exact2 + 1e-6 * abs(exact2)


@pytest.mark.parametrize("n,m_", [(40, 300), (130, None), (257, 70)])
def test_kernel_matrix_vjp(handle, n, m_):
    """gps_kmat_vjp: sum_ij W_ij d k(X_i, X2_j) / d theta for an arbitrary cotangent W (reverse mode through kern.K) against
    central differences of the oracle's K -- rectangular and square, sum / product programs and a White kernel (which only
    contributes on the diagonal of K(X, X))."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n)
    d = 3
    for kind in ("rbf_ard", "rbf_times_periodic_plus_white", "m52_plus_periodic"):
        kern, theta, fn, _ = _cases(gpf, d)[kind]()
        X = rng.standard_normal((n, d)); X2 = None if m_ is None else rng.standard_normal((m_, d))
        W = rng.standard_normal((n, m_ or n))
        got = handle.kmat_vjp(kern._program(d), X, W, X2)
        ref = np.zeros(theta.size)
        saved, orc.SQUARE_DIST_MODE = orc.SQUARE_DIST_MODE, "diff"
        try:
            for p in range(theta.size):
                hh = 1e-6 * max(1.0, abs(theta[p])); tp, tm = theta.copy(), theta.copy(); tp[p] += hh; tm[p] -= hh
                ref[p] = np.sum(W * (orc.K(fn(tp), X, X2) - orc.K(fn(tm), X, X2))) / (2 * hh)
        finally:
            orc.SQUARE_DIST_MODE = saved
        # slots: per primitive [variance, one per active dim | lengthscale, period]; isotropic kernels sum their dim slots
        layout = kern._grad_layout(d)
        acc = {}
        order = []
        for (param, idx), g in zip(layout, got):
            if param is None:
                continue
            key = (id(param), idx)
            if key not in acc:
                acc[key] = 0.0; order.append(key)
            acc[key] += g
        flat = []
        for p in kern.parameters:
            size = np.atleast_1d(p.vf_val).size
            flat += [acc.get((id(p), None), 0.0) if size == 1 else acc.get((id(p), i), 0.0) for i in range(size)]
        flat = np.array(flat)
        assert flat.shape == ref.shape, (kind, flat.shape, ref.shape)
        assert np.abs(flat - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (kind, flat, ref)


@pytest.mark.parametrize("n,m_", [(40, 300), (130, None), (257, 70)])
def test_kernel_matrix_input_vjp(handle, n, m_):
    """gps_kmat_input_vjp: d/dX sum_ij W_ij k(X_i, X2_j) -- reverse mode through kern.K with respect to its POINTS (what a
    trainable InducingPoints.Z receives) -- against central differences of the oracle's K, entry by entry of X; rectangular
    and square (both arguments move), Matern / RBF / Periodic / White / Constant in sums and products, and an NKN."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n + 5)
    d = 3
    for kind in ("rbf_ard", "rbf_times_periodic_plus_white", "m52_plus_periodic", "m32_ard", "m12_iso"):
        kern, theta, fn, _ = _cases(gpf, d)[kind]()
        X = rng.standard_normal((n, d)); X2 = None if m_ is None else rng.standard_normal((m_, d))
        W = rng.standard_normal((n, m_ or n))
        got = handle.kmat_input_vjp(kern._program(d), X, W, X2)
        assert got.shape == (n, d)
        spec = fn(theta)
        saved, orc.SQUARE_DIST_MODE = orc.SQUARE_DIST_MODE, "diff"
        try:
            worst = 0.0
            for i in rng.choice(n, 6, replace=False):
                for k in range(d):
                    hh = 1e-6
                    Xp, Xm = X.copy(), X.copy(); Xp[i, k] += hh; Xm[i, k] -= hh
                    if X2 is None:
                        fd = np.sum(W * (orc.K(spec, Xp) - orc.K(spec, Xm))) / (2 * hh)
                    else:
                        fd = np.sum(W * (orc.K(spec, Xp, X2) - orc.K(spec, Xm, X2))) / (2 * hh)
                    worst = max(worst, abs(got[i, k] - fd) / max(1.0, abs(fd)))
            # (Matern-1/2 has a kink at r = 0: the diagonal of K(X, X) contributes nothing in the product, one-sided slopes in
            # a central difference cancel as well)
            assert worst <= 5e-6, (kind, worst)
        finally:
            orc.SQUARE_DIST_MODE = saved


@pytest.mark.parametrize("whiten,q_diag", [(True, False), (False, False), (True, True)])
def test_svgp_inducing_input_gradient_and_training(handle, whiten, q_diag):
    """train_inducing=True: Z is among the parameters (as every TF variable is for examples/svgp.py:161); its gradient
    against central differences of the oracle's bound in single entries of Z, and optimize() moves Z and ends higher than
    with Z fixed."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(17)
    n, d, m_, k = 240, 2, 24, 2
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, k)) * 1.5) + 0.1 * rng.standard_normal((n, k))
    Z = X[:m_].copy() + 0.05 * rng.standard_normal((m_, d))
    ls = np.array([0.9, 1.4])
    kern = gpf.kernels.Matern52(d, variance=1.3, lengthscales=ls, ARD=True)
    spec = {"type": "matern52", "variance": c(1.3), "lengthscales": c(ls), "input_dim": d}
    m = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.2), Z=Z, q_diag=q_diag, whiten=whiten, num_data=2 * n,
                        train_inducing=True)
    q_mu = rng.standard_normal((m_, k)) * 0.3
    m._q_mu.assign(q_mu)
    if q_diag:
        m._q_sqrt.assign(np.abs(rng.standard_normal((m_, k))) * 0.4 + 0.2)
    else:
        m._q_sqrt.assign(np.tril(rng.standard_normal((k, m_, m_)) * (0.5 / m_) + np.eye(m_) * 0.5).transpose(1, 2, 0).copy())
    q_sqrt = np.asarray(m.q_sqrt).copy()
    noise = c(0.2)
    assert any(p is m.feature._Z for p in m.parameters)
    bound, grads = m.compute_log_likelihood_and_gradients()
    gz = {id(p): g for p, g in grads}[id(m.feature._Z)]
    assert gz.shape == Z.shape
    for _ in range(8):
        a, b = int(rng.integers(m_)), int(rng.integers(d))
        hh = 1e-6
        Zp, Zm = Z.copy(), Z.copy(); Zp[a, b] += hh; Zm[a, b] -= hh
        fp = orc.svgp_elbo(spec, X, Y, Zp, q_mu, q_sqrt, noise, whiten=whiten, num_data=2 * n)
        fm = orc.svgp_elbo(spec, X, Y, Zm, q_mu, q_sqrt, noise, whiten=whiten, num_data=2 * n)
        fd = (fp - fm) / (2 * hh)
        assert abs(gz[a, b] - fd) <= 2e-5 * max(1.0, abs(fd)), (a, b, gz[a, b], fd)
    # training: with Z free the bound ends at least as high as with Z fixed, and Z has moved
    fixed = gpf.models.SVGP(X, Y, gpf.kernels.Matern52(d, variance=1.3, lengthscales=ls, ARD=True), gpf.likelihoods.Gaussian(0.2),
                            Z=Z, q_diag=q_diag, whiten=whiten, num_data=2 * n)
    f_fixed = fixed.optimize(max_iter=60)
    f_free = m.optimize(max_iter=60)
    assert np.abs(np.asarray(m.feature.Z) - Z).max() > 1e-3
    assert f_free <= f_fixed + 1e-6 * abs(f_fixed)


@pytest.mark.parametrize("act", [False, True])
def test_kernel_matrix_input_vjp_through_a_neural_kernel_network(handle, act):
    """The same input gradient through a Neural Kernel Network (the per-entry reverse pass through Linear / Product / exp
    layers yields the adjoints of the primitive values; the chain into the points is the primitives' own)."""
    import gpflowSlim as gpf
    d, n = 3, 90
    m, spec, X, _ = _nkn_model(gpf, d, act, n, 12)
    rng = np.random.default_rng(3)
    X2 = rng.standard_normal((55, d))
    W = rng.standard_normal((n, 55))
    got = handle.kmat_input_vjp(m.kern._program(d), X, W, X2)
    saved, orc.SQUARE_DIST_MODE = orc.SQUARE_DIST_MODE, "diff"
    try:
        for i in rng.choice(n, 5, replace=False):
            for k in range(d):
                hh = 1e-6
                Xp, Xm = X.copy(), X.copy(); Xp[i, k] += hh; Xm[i, k] -= hh
                fd = np.sum(W * (orc.K(spec, Xp, X2) - orc.K(spec, Xm, X2))) / (2 * hh)
                assert abs(got[i, k] - fd) <= 5e-6 * max(1.0, abs(fd)), (i, k, got[i, k], fd)
    finally:
        orc.SQUARE_DIST_MODE = saved


@pytest.mark.parametrize("kind,n,m_,d,r", [("rbf_ard", 300, 40, 3, 1), ("m52_plus_periodic", 420, 130, 2, 2), ("m32_ard", 260, 60, 2, 3)])
def test_sgpr_bound_gradient(handle, kind, n, m_, d, r):
    """Gradient of the SGPR collapsed bound (gps_sgpr_grad; models/sgpr.py:121-153 through TF autodiff in the reference) with
    respect to the kernel parameters, the noise variance, a Linear mean function and the inducing inputs, against central
    differences of the ORACLE's bound in the constrained parameters and of the product's own bound in the unconstrained
    ones."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n + m_ + r)
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, r))) + 0.1 * rng.standard_normal((n, r)) + 0.3
    Z = X[:m_].copy() + 0.05 * rng.standard_normal((m_, d))
    kern, theta, fn, _ = _cases(gpf, d)[kind]()
    Am = rng.standard_normal((d, r)) * 0.2; bm = rng.standard_normal(r) * 0.1
    mf = gpf.mean_functions.Linear(Am.copy(), bm.copy())
    m = gpf.models.SGPR(X, Y, kern, Z=Z, obs_var=0.25, mean_function=mf)
    noise = c(0.25)
    bound, grads = m.compute_log_likelihood_and_gradients()

    def f(th=theta, nz=noise, ZZ=Z, AA=Am, bb=bm):
        return orc.sgpr_bound(fn(th), X, Y, ZZ, nz, mean_X=X @ AA + bb)

    ref_bound = f()
    assert abs(bound - ref_bound) <= 1e-8 * abs(ref_bound)
    assert abs(bound - m.compute_log_likelihood()) <= 1e-12 * abs(bound)
    by = {id(p): g for p, g in grads}

    def cd(make, x0, hrel=1e-6):
        hh = hrel * max(1.0, abs(x0))
        return (make(x0 + hh) - make(x0 - hh)) / (2 * hh)

    got = _flat_constrained_grad(m, grads)
    assert got.shape == theta.shape
    for i in range(theta.size):
        def mk(v, i=i):
            th = theta.copy(); th[i] = v
            return f(th=th)
        fd = cd(mk, theta[i])
        assert abs(got[i] - fd) <= 2e-5 * max(1.0, abs(fd)), ("theta", i, got[i], fd)
    gn = float(by[id(m.likelihood._variance)] / m.likelihood._variance.transform.forward_grad(m.likelihood._variance.vf_val))
    fd = cd(lambda v: f(nz=v), noise)
    assert abs(gn - fd) <= 2e-5 * max(1.0, abs(fd)), (gn, fd)
    gz = by[id(m.feature._Z)]
    for _ in range(6):
        a, b = int(rng.integers(m_)), int(rng.integers(d))
        def mk(v, a=a, b=b):
            ZZ = Z.copy(); ZZ[a, b] = v
            return f(ZZ=ZZ)
        fd = cd(mk, Z[a, b])
        assert abs(gz[a, b] - fd) <= 2e-5 * max(1.0, abs(fd)), ("Z", a, b, gz[a, b], fd)
    gA, gb = by[id(mf.A)], by[id(mf.b)]
    for _ in range(3):
        a, b = int(rng.integers(d)), int(rng.integers(r))
        def mk(v, a=a, b=b):
            AA = Am.copy(); AA[a, b] = v
            return f(AA=AA)
        fd = cd(mk, Am[a, b])
        assert abs(np.reshape(gA, Am.shape)[a, b] - fd) <= 2e-5 * max(1.0, abs(fd)), ("A", a, b)
    fd = cd(lambda v: f(bb=np.concatenate([[v], bm[1:]])), bm[0])
    assert abs(np.ravel(gb)[0] - fd) <= 2e-5 * max(1.0, abs(fd))
    _fd_check(m, grads)


def test_sgpr_optimize_raises_the_bound(handle):
    """SGPR + Model.optimize(): hyper-parameters, noise and inducing inputs move; the bound goes up and stays below the exact
    evidence at the same hyper-parameters."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(8)
    n, d, m_ = 500, 2, 30
    X = rng.standard_normal((n, d)); Y = np.sin(2.0 * X[:, :1]) * np.cos(X[:, 1:2]) + 0.1 * rng.standard_normal((n, 1))
    Z0 = X[:m_].copy()
    m = gpf.models.SGPR(X, Y, gpf.kernels.RBF(d, ARD=True), Z=Z0, obs_var=0.5)
    start = m.objective
    final = m.optimize(max_iter=120)
    assert final < start - 100.0, (start, final)
    assert np.abs(np.asarray(m.feature.Z) - Z0).max() > 1e-3
    exact = gpf.models.GPR(X, Y, m.kern, obs_var=float(np.squeeze(m.likelihood.variance))).compute_log_likelihood()
    assert -final <= exact + 1e-6 * abs(exact)


@pytest.mark.parametrize("kind,n,m_,d,r", [("rbf_ard", 300, 40, 3, 1), ("m52_plus_periodic", 420, 130, 2, 2)])
def test_fitc_likelihood_gradient(handle, kind, n, m_, d, r):
    """Gradient of the FITC log-likelihood (gps_fitc_grad; models/sgpr.py:229-290 under TF autodiff) with respect to the kernel
    parameters, the noise variance, a Linear mean function and the inducing inputs, against central differences of the
    oracle's likelihood and of the product's own."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n + m_ + r + 1)
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, r))) + 0.1 * rng.standard_normal((n, r)) + 0.3
    Z = X[:m_].copy() + 0.05 * rng.standard_normal((m_, d))
    kern, theta, fn, _ = _cases(gpf, d)[kind]()
    Am = rng.standard_normal((d, r)) * 0.2; bm = rng.standard_normal(r) * 0.1
    mf = gpf.mean_functions.Linear(Am.copy(), bm.copy())
    m = gpf.models.GPRFITC(X, Y, kern, Z=Z, obs_var=0.25, mean_function=mf)
    noise = c(0.25)
    bound, grads = m.compute_log_likelihood_and_gradients()

    def f(th=theta, nz=noise, ZZ=Z, AA=Am, bb=bm):
        return orc.fitc_lml(fn(th), X, Y, ZZ, nz, mean_X=X @ AA + bb)

    ref_bound = f()
    assert abs(bound - ref_bound) <= 1e-8 * abs(ref_bound)
    by = {id(p): g for p, g in grads}

    def cd(make, x0, hrel=1e-6):
        hh = hrel * max(1.0, abs(x0))
        return (make(x0 + hh) - make(x0 - hh)) / (2 * hh)

    got = _flat_constrained_grad(m, grads)
    for i in range(theta.size):
        def mk(v, i=i):
            th = theta.copy(); th[i] = v
            return f(th=th)
        fd = cd(mk, theta[i])
        assert abs(got[i] - fd) <= 2e-5 * max(1.0, abs(fd)), ("theta", i, got[i], fd)
    gn = float(by[id(m.likelihood._variance)] / m.likelihood._variance.transform.forward_grad(m.likelihood._variance.vf_val))
    fd = cd(lambda v: f(nz=v), noise)
    assert abs(gn - fd) <= 2e-5 * max(1.0, abs(fd)), (gn, fd)
    gz = by[id(m.feature._Z)]
    for _ in range(6):
        a, b = int(rng.integers(m_)), int(rng.integers(d))
        def mk(v, a=a, b=b):
            ZZ = Z.copy(); ZZ[a, b] = v
            return f(ZZ=ZZ)
        fd = cd(mk, Z[a, b])
        assert abs(gz[a, b] - fd) <= 2e-5 * max(1.0, abs(fd)), ("Z", a, b, gz[a, b], fd)
    gA = by[id(mf.A)]
    a, b = 1, r - 1
    def mkA(v):
        AA = Am.copy(); AA[a, b] = v
        return f(AA=AA)
    fd = cd(mkA, Am[a, b])
    assert abs(np.reshape(gA, Am.shape)[a, b] - fd) <= 2e-5 * max(1.0, abs(fd))
    _fd_check(m, grads)
    start = m.objective
    assert m.optimize(max_iter=30) < start


def test_optimize_with_priors(handle):
    """A prior on a parameter (params.py:176-194: log p(constrained) + log-Jacobian of the transform, any object with a
    `logp`) takes part in `objective` and in the gradient optimize() descends on."""
    import gpflowSlim as gpf

    class LogNormal(object):
        def __init__(self, mu, var):
            self.mu, self.var = mu, var

        def logp(self, x):
            x = np.asarray(x, dtype=np.float64)
            return -0.5 * np.log(2 * np.pi * self.var) - np.log(x) - 0.5 * np.square(np.log(x) - self.mu) / self.var

    rng = np.random.default_rng(4)
    n, d = 150, 2
    X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, ARD=True), obs_var=0.3)
    m.kern._ls.prior = LogNormal(0.0, 0.5)
    m.likelihood._variance.prior = LogNormal(-2.0, 1.0)
    x0 = m._pack()
    f0, g0 = m._objective_and_grad(x0)
    assert abs(f0 - m.objective) <= 1e-12 * abs(f0)
    assert abs(m.objective + m.compute_log_likelihood() + m.compute_log_prior()) <= 1e-10 * abs(f0)
    for i in range(x0.size):
        hh = 1e-5
        xp, xm = x0.copy(), x0.copy(); xp[i] += hh; xm[i] -= hh
        fp, _ = m._objective_and_grad(xp); fm, _ = m._objective_and_grad(xm)
        assert abs(g0[i] - (fp - fm) / (2 * hh)) <= 1e-5 * max(1.0, abs(g0[i])), i
    m._unpack(x0)
    assert m.optimize(max_iter=40) < f0
