"""CPU oracle for the exact-GP hot path of GPflow-Slim.  TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy fp64 *restatement* of the reference algorithm, op for op.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product (``gpflow-slim_amd/``) never does and fails loudly when its HIP library is missing.

PARITY UNPINNED BY THE REFERENCE.  The reference is pure Python over TensorFlow 1.x
(``tf.cholesky`` at gpflowSlim/models/gpr.py:70, ``tensorflow.contrib.eager`` at
gpflowSlim/models/model.py:26) and TensorFlow is not installable in the build image, so the
reference cannot be imported or run here, and its only two tests
(gpflowSlim/models/gpr.py:135-203, gpflowSlim/densities.py:159-174) hold no numeric values.
The oracle is therefore pinned by (tests/test_oracle.py): analytic known-answer cases,
50-digit mpmath evaluations of the same formulas, scikit-learn's independent GP implementation,
and the reference's own structural test (Cholesky predictor == Woodbury predictor) restated.

Reference lines followed (paths relative to the reference checkout, gpflowSlim/...):
  transforms.py:145-146,177-178   Log1pe forward/backward (softplus + lower)
  kernels.py:238-245              Kernel._slice (active dims)
  kernels.py:408-421              Stationary.square_dist (GEMM form, clip at 0)
  kernels.py:424-426              Stationary.euclid_dist (sqrt(r2 + 1e-12))
  kernels.py:428-429,803-804      Kdiag
  kernels.py:436-439              RBF.K
  kernels.py:560-565,573-577,589-594,605-610   Exponential / Matern12 / Matern32 / Matern52 .K
  kernels.py:806-819              Periodic.K
  kernels.py:332-338,345-350      White.K / Constant.K
  kernels.py:1071-1084            Sum / Product (left folds)
  models/gpr.py:69-72             GPR._build_likelihood (exact branch)
  densities.py:73-95              multivariate_normal
  models/gpr.py:119-131           GPR._build_predict (exact branch)
  likelihoods.py:180-184, densities.py:24-25   predict_y / predict_density
  conditionals.py:24-66,80-121    conditional / base_conditional
  features.py:74-81               InducingPoints.Kuu / Kuf
  kullback_leiblers.py:26-105     gauss_kl
  models/svgp.py:101-130          SVGP bound (Gaussian likelihood)
  models/sgpr.py:121-189          SGPR collapsed bound and prediction
  neural_kernel_network/neural_kernel_network.py:35-47, neural_kernel_network_wrapper.py:39-44,116-117,145-148
                                  NeuralKernelNetwork.K / Kdiag and its layers
"""
from functools import reduce

import numpy as np
import scipy.linalg as sl

JITTER = 1e-6            # gpflowrc:11  numerics.jitter_level


# ---------------------------------------------------------------------------------------------
# transforms.py:117-181
def softplus(x):
    x = np.asarray(x, dtype=np.float64)
    return np.logaddexp(0.0, x)


def log1pe_forward(x, lower=1e-6):
    """transforms.py:145-146  tf.nn.softplus(x) + lower"""
    return softplus(x) + lower


def log1pe_backward(y, lower=1e-6):
    """transforms.py:177-178"""
    ys = np.maximum(np.asarray(y, dtype=np.float64) - lower, np.finfo(np.float64).eps)
    return ys + np.log(-np.expm1(-ys))


def constrained(value, lower=1e-6):
    """What the reference actually uses after Parameter(value, Log1pe(lower)).value
    (params.py:142-145,164-166): forward(backward(value))."""
    return log1pe_forward(log1pe_backward(value, lower), lower)


# ---------------------------------------------------------------------------------------------
# kernels: specs are plain dicts so that the oracle shares no code with the product.
#   {"type": "rbf"|"matern12"|"matern32"|"matern52"|"exponential", "variance": v,
#    "lengthscales": scalar or [d], "active_dims": [..] or None, "input_dim": d}
#   {"type": "periodic", "variance": v, "lengthscales": l, "period": p, "active_dims", "input_dim"}
#   {"type": "white"|"constant", "variance": v}
#   {"type": "sum"|"product", "children": [spec or float, ...]}
def _slice(spec, X, X2):
    """kernels.py:238-245"""
    ad = spec.get("active_dims")
    if ad is None:
        ad = slice(spec["input_dim"])
    else:
        ad = np.asarray(ad, dtype=int)
    X = X[:, ad]
    if X2 is not None:
        X2 = X2[:, ad]
    return X, X2


# "gemm": the reference's op order (kernels.py:408-421).  "diff": sum_d ((x_d - x'_d)/l_d)^2 -- the same
# quantity without the GEMM-form cancellation noise; used only by gpr_lml_grad, whose central differences
# of K would otherwise amplify the O(1e-16) diagonal noise through sqrt(r2 + 1e-12) (Matern family).
SQUARE_DIST_MODE = "gemm"


def square_dist(X, X2, lengthscales):
    """kernels.py:408-421, op for op"""
    if SQUARE_DIST_MODE == "diff":
        A = X / lengthscales
        B = A if X2 is None else X2 / lengthscales
        out = np.zeros((A.shape[0], B.shape[0]))
        for dd in range(A.shape[1]):
            out += np.square(A[:, dd:dd + 1] - B[None, :, dd])
        return out
    X = X / lengthscales
    Xs = np.sum(np.square(X), axis=1)
    if X2 is None:
        dist = -2 * np.matmul(X, X.T)
        dist += np.reshape(Xs, (-1, 1)) + np.reshape(Xs, (1, -1))
        return np.clip(dist, 0., np.inf)
    X2 = X2 / lengthscales
    X2s = np.sum(np.square(X2), axis=1)
    dist = -2 * np.matmul(X, X2.T)
    dist += np.reshape(Xs, (-1, 1)) + np.reshape(X2s, (1, -1))
    return np.clip(dist, 0., np.inf)


def euclid_dist(X, X2, lengthscales):
    """kernels.py:424-426"""
    return np.sqrt(square_dist(X, X2, lengthscales) + 1e-12)


def _periodic_K(X, X2, variance, lengthscales, period, chunk=256):
    """kernels.py:813-819; evaluated in row chunks (the reference's [N,M,D] temporary is 34 GB at
    N=16384, D=16) -- same arithmetic per entry."""
    if X2 is None:
        X2 = X
    out = np.empty((X.shape[0], X2.shape[0]))
    f2 = X2[None, :, :]
    for s in range(0, X.shape[0], chunk):
        f = X[s:s + chunk, None, :]
        r = np.pi * (f - f2) / period
        r = np.sum(np.square(np.sin(r) / lengthscales), 2)
        out[s:s + chunk] = variance * np.exp(-0.5 * r)
    return out


def K(spec, X, X2=None):
    """kern.K(X, X2) of the reference for a spec tree."""
    if isinstance(spec, (int, float)):
        return spec                                     # kernels.py:1060-1063 _kernel_function
    t = spec["type"]
    if t == "nkn":
        return nkn_K(spec["primitives"], spec["layers"], X, X2)
    if t in ("sum", "product"):
        vals = [K(c, X, X2) for c in spec["children"] if not isinstance(c, (int, float))]
        consts = [c for c in spec["children"] if isinstance(c, (int, float))]
        op = np.add if t == "sum" else np.multiply
        return reduce(op, vals + consts)                # kernels.py:1073 / 1081
    v = spec["variance"]
    if t == "white":                                    # kernels.py:332-338
        if X2 is None:
            return np.diag(np.full(X.shape[0], v))
        return np.zeros((X.shape[0], X2.shape[0]))
    if t == "constant":                                 # kernels.py:345-350
        return np.full((X.shape[0], X.shape[0] if X2 is None else X2.shape[0]), v)
    Xs, X2s = _slice(spec, X, X2)
    if t == "periodic":
        return _periodic_K(Xs, X2s, v, spec["lengthscales"], spec["period"])
    ls = np.asarray(spec["lengthscales"], dtype=np.float64)
    if t == "rbf":                                      # kernels.py:439
        return v * np.exp(-square_dist(Xs, X2s, ls) / 2)
    r = euclid_dist(Xs, X2s, ls)
    if t == "matern12":                                 # kernels.py:576-577
        return v * np.exp(-r)
    if t == "exponential":                              # kernels.py:564-565
        return v * np.exp(-0.5 * r)
    if t == "matern32":                                 # kernels.py:592-594
        return v * (1. + np.sqrt(3.) * r) * np.exp(-np.sqrt(3.) * r)
    if t == "matern52":                                 # kernels.py:608-610
        return v * (1.0 + np.sqrt(5.) * r + 5. / 3. * np.square(r)) * np.exp(-np.sqrt(5.) * r)
    raise ValueError("unknown kernel type %r" % t)


def nkn_forward(layers, v):
    """neural_kernel_network_wrapper.py:39-44: v [nm, k] through the layers.
    layers: [("linear", W [out,in], b [out]) | ("product", step) | ("exp",)]"""
    for ly in layers:
        if ly[0] == "linear":
            v = np.matmul(v, np.transpose(ly[1])) + ly[2]                 # :116-117
        elif ly[0] == "product":
            v = np.prod(np.reshape(v, [v.shape[0], -1, ly[1]]), -1)       # :145-148
        elif ly[0] == "exp":
            v = np.exp(v)
        else:
            raise ValueError(ly[0])
    return v


def nkn_K(prim_specs, layers, X, X2=None):
    """neural_kernel_network.py:41-47"""
    vals = [K(s, X, X2) for s in prim_specs]
    shape = vals[0].shape
    stacked = np.stack([v.reshape(-1) for v in vals], 1)
    return nkn_forward(layers, stacked).reshape(shape)


def nkn_Kdiag(prim_specs, layers, X):
    """neural_kernel_network.py:35-39"""
    stacked = np.stack([Kdiag(s, X) for s in prim_specs], 1)
    return np.squeeze(nkn_forward(layers, stacked), -1)


def Kdiag(spec, X):
    """kernels.py:428-429, 803-804, 327-328, 1075-1076, 1083-1084"""
    if isinstance(spec, (int, float)):
        return spec
    t = spec["type"]
    if t == "nkn":
        return nkn_Kdiag(spec["primitives"], spec["layers"], X)
    if t in ("sum", "product"):
        vals = [Kdiag(c, X) for c in spec["children"] if not isinstance(c, (int, float))]
        consts = [c for c in spec["children"] if isinstance(c, (int, float))]
        return reduce(np.add if t == "sum" else np.multiply, vals + consts)
    return np.ones(X.shape[0]) * spec["variance"]


# ---------------------------------------------------------------------------------------------
def multivariate_normal(x, mu, L):
    """densities.py:73-95"""
    d = x - mu
    alpha = sl.solve_triangular(L, d, lower=True)
    num_col = 1 if x.ndim == 1 else x.shape[1]
    num_dims = x.shape[0]
    ret = -0.5 * num_dims * num_col * np.log(2 * np.pi)
    ret += -num_col * np.sum(np.log(np.diag(L)))
    ret += -0.5 * np.sum(np.square(alpha))
    return ret


def gaussian_density(x, mu, var):
    """densities.py:24-25"""
    return -0.5 * (np.log(2 * np.pi) + np.log(var) + np.square(mu - x) / var)


def gpr_lml(spec, X, Y, noise_var, mean_X=None):
    """models/gpr.py:69-72.  mean_X = mean_function(X) ([N,1] or [N,R]); Zero by default."""
    Kmat = K(spec, X) + np.eye(X.shape[0]) * noise_var
    L = np.linalg.cholesky(Kmat)
    m = np.zeros((X.shape[0], 1)) if mean_X is None else mean_X
    return multivariate_normal(Y, m, L)


def gpr_predict(spec, X, Y, noise_var, Xnew, full_cov=False, mean_X=None, mean_Xnew=None):
    """models/gpr.py:119-131"""
    mX = np.zeros((X.shape[0], 1)) if mean_X is None else mean_X
    mN = np.zeros((Xnew.shape[0], 1)) if mean_Xnew is None else mean_Xnew
    Kx = K(spec, X, Xnew)
    Kmat = K(spec, X) + np.eye(X.shape[0]) * noise_var
    L = np.linalg.cholesky(Kmat)
    A = sl.solve_triangular(L, Kx, lower=True)
    V = sl.solve_triangular(L, Y - mX, lower=True)
    fmean = np.matmul(A.T, V) + mN
    if full_cov:
        fvar = K(spec, Xnew) - np.matmul(A.T, A)
        fvar = np.tile(fvar[:, :, None], [1, 1, Y.shape[1]])
    else:
        fvar = Kdiag(spec, Xnew) - np.sum(np.square(A), 0)
        fvar = np.tile(np.reshape(fvar, (-1, 1)), [1, Y.shape[1]])
    return fmean, fvar


def base_conditional(Kmn, Kmm, Knn, f, full_cov=False, q_sqrt=None, white=False):
    """conditionals.py:80-121"""
    num_func = f.shape[1]
    Lm = np.linalg.cholesky(Kmm)
    A = sl.solve_triangular(Lm, Kmn, lower=True)
    if full_cov:
        fvar = Knn - np.matmul(A.T, A)
        fvar = np.tile(fvar[None, :, :], [num_func, 1, 1])
    else:
        fvar = Knn - np.sum(np.square(A), 0)
        fvar = np.tile(fvar[None, :], [num_func, 1])
    if not white:
        A = sl.solve_triangular(Lm.T, A, lower=False)
    fmean = np.matmul(A.T, f)
    if q_sqrt is not None:
        if q_sqrt.ndim == 2:
            LTA = A * q_sqrt.T[:, :, None]                          # K x M x N
        elif q_sqrt.ndim == 3:
            Lq = np.tril(np.transpose(q_sqrt, (2, 0, 1)))           # K x M x M
            A_tiled = np.tile(A[None, :, :], [num_func, 1, 1])
            LTA = np.matmul(np.transpose(Lq, (0, 2, 1)), A_tiled)   # K x M x N
        else:
            raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.ndim))
        if full_cov:
            fvar = fvar + np.matmul(np.transpose(LTA, (0, 2, 1)), LTA)
        else:
            fvar = fvar + np.sum(np.square(LTA), 1)
    fvar = np.transpose(fvar)                                       # N x K or N x N x K
    return fmean, fvar


def conditional(Xnew, X, spec, f, full_cov=False, q_sqrt=None, white=False, jitter=JITTER):
    """conditionals.py:24-66 (also feature_conditional :69-77 with features.py:74-81)"""
    Kmm = K(spec, X) + np.eye(X.shape[0]) * jitter
    Kmn = K(spec, X, Xnew)
    Knn = K(spec, Xnew) if full_cov else Kdiag(spec, Xnew)
    return base_conditional(Kmn, Kmm, Knn, f, full_cov=full_cov, q_sqrt=q_sqrt, white=white)


def gpr_lml_grad(spec_fn, theta, X, Y, noise_var, rel_step=1e-6):
    """Gradient of gpr_lml w.r.t. the flat constrained parameter vector `theta` (kernel parameters as
    consumed by spec_fn(theta) -> spec) and w.r.t. the noise variance.  This is what TF autodiff
    differentiates in the reference (examples/gpr.py:53-54).  Semi-analytic and independent of the
    product's derivative formulas:  d LML = 1/2 tr((a a^T - R K_y^-1) dK)  with dK/dtheta_p obtained by
    central differences of the oracle's own K(spec)."""
    global SQUARE_DIST_MODE
    theta = np.asarray(theta, dtype=np.float64)
    n, R = Y.shape
    saved, SQUARE_DIST_MODE = SQUARE_DIST_MODE, "diff"
    try:
        return _gpr_lml_grad(spec_fn, theta, X, Y, noise_var, rel_step, n, R)
    finally:
        SQUARE_DIST_MODE = saved


def _gpr_lml_grad(spec_fn, theta, X, Y, noise_var, rel_step, n, R):
    Ky = K(spec_fn(theta), X) + np.eye(n) * noise_var
    Kinv = np.linalg.inv(Ky)
    a = Kinv @ Y
    W = a @ a.T - R * Kinv
    g = np.zeros_like(theta)
    for p in range(theta.size):
        h = rel_step * max(1.0, abs(theta[p]))
        tp, tm = theta.copy(), theta.copy()
        tp[p] += h; tm[p] -= h
        dK = (K(spec_fn(tp), X) - K(spec_fn(tm), X)) / (2 * h)
        g[p] = 0.5 * np.sum(W * dK)
    return g, 0.5 * np.trace(W), a


def sgpr_bound(spec, X, Y, Z, noise_var, mean_X=None, jitter=JITTER):
    """models/sgpr.py:121-153"""
    num_inducing, num_data, output_dim = Z.shape[0], Y.shape[0], Y.shape[1]
    err = Y - (0.0 if mean_X is None else mean_X)
    Kd = Kdiag(spec, X)
    Kuf = K(spec, Z, X)
    Kuu = K(spec, Z) + jitter * np.eye(num_inducing)
    L = np.linalg.cholesky(Kuu)
    sigma = np.sqrt(noise_var)
    A = sl.solve_triangular(L, Kuf, lower=True) / sigma
    AAT = np.matmul(A, A.T)
    B = AAT + np.eye(num_inducing)
    LB = np.linalg.cholesky(B)
    Aerr = np.matmul(A, err)
    c = sl.solve_triangular(LB, Aerr, lower=True) / sigma
    bound = -0.5 * num_data * output_dim * np.log(2 * np.pi)
    bound += -output_dim * np.sum(np.log(np.diag(LB)))
    bound -= 0.5 * num_data * output_dim * np.log(noise_var)
    bound += -0.5 * np.sum(np.square(err)) / noise_var
    bound += 0.5 * np.sum(np.square(c))
    bound += -0.5 * output_dim * np.sum(Kd) / noise_var
    bound += 0.5 * output_dim * np.sum(np.diag(AAT))
    return bound


def sgpr_predict(spec, X, Y, Z, noise_var, Xnew, full_cov=False, mean_X=None, mean_Xnew=None, jitter=JITTER):
    """models/sgpr.py:155-189"""
    num_inducing = Z.shape[0]
    err = Y - (0.0 if mean_X is None else mean_X)
    Kuf = K(spec, Z, X)
    Kuu = K(spec, Z) + jitter * np.eye(num_inducing)
    Kus = K(spec, Z, Xnew)
    sigma = np.sqrt(noise_var)
    L = np.linalg.cholesky(Kuu)
    A = sl.solve_triangular(L, Kuf, lower=True) / sigma
    B = np.matmul(A, A.T) + np.eye(num_inducing)
    LB = np.linalg.cholesky(B)
    Aerr = np.matmul(A, err)
    c = sl.solve_triangular(LB, Aerr, lower=True) / sigma
    tmp1 = sl.solve_triangular(L, Kus, lower=True)
    tmp2 = sl.solve_triangular(LB, tmp1, lower=True)
    mean = np.matmul(tmp2.T, c)
    if full_cov:
        var = K(spec, Xnew) + np.matmul(tmp2.T, tmp2) - np.matmul(tmp1.T, tmp1)
        var = np.tile(var[:, :, None], [1, 1, Y.shape[1]])
    else:
        var = Kdiag(spec, Xnew) + np.sum(np.square(tmp2), 0) - np.sum(np.square(tmp1), 0)
        var = np.tile(var[:, None], [1, Y.shape[1]])
    return mean + (0.0 if mean_Xnew is None else mean_Xnew), var


def _fitc_common(spec, X, Y, Z, noise_var, mean_X, jitter):
    """GPRFITC._build_common_terms, models/sgpr.py:232-250"""
    num_inducing = Z.shape[0]
    err = Y - (0.0 if mean_X is None else mean_X)
    Kd = Kdiag(spec, X)
    Kuf = K(spec, Z, X)
    Kuu = K(spec, Z) + jitter * np.eye(num_inducing)
    Luu = np.linalg.cholesky(Kuu)
    V = sl.solve_triangular(Luu, Kuf, lower=True)
    diagQff = np.sum(np.square(V), 0)
    nu = Kd - diagQff + noise_var
    B = np.eye(num_inducing) + np.matmul(V / nu, V.T)
    L = np.linalg.cholesky(B)
    beta = err / nu[:, None]
    alpha = np.matmul(V, beta)
    gamma = sl.solve_triangular(L, alpha, lower=True)
    return err, nu, Luu, L, alpha, beta, gamma


def fitc_lml(spec, X, Y, Z, noise_var, mean_X=None, jitter=JITTER):
    """GPRFITC._build_likelihood, models/sgpr.py:252-291"""
    err, nu, Luu, L, alpha, beta, gamma = _fitc_common(spec, X, Y, Z, noise_var, mean_X, jitter)
    num_data, num_latent = Y.shape[0], Y.shape[1]
    mahalanobisTerm = -0.5 * np.sum(np.square(err) / nu[:, None]) + 0.5 * np.sum(np.square(gamma))
    constantTerm = -0.5 * num_data * np.log(2.0 * np.pi)
    logDeterminantTerm = -0.5 * np.sum(np.log(nu)) - np.sum(np.log(np.diag(L)))
    return mahalanobisTerm + (constantTerm + logDeterminantTerm) * num_latent


def fitc_predict(spec, X, Y, Z, noise_var, Xnew, full_cov=False, mean_X=None, mean_Xnew=None, jitter=JITTER):
    """GPRFITC._build_predict, models/sgpr.py:293-318"""
    _, _, Luu, L, _, _, gamma = _fitc_common(spec, X, Y, Z, noise_var, mean_X, jitter)
    Kus = K(spec, Z, Xnew)
    w = sl.solve_triangular(Luu, Kus, lower=True)
    tmp = sl.solve_triangular(L.T, gamma, lower=False)
    mean = np.matmul(w.T, tmp) + (0.0 if mean_Xnew is None else mean_Xnew)
    intermediateA = sl.solve_triangular(L, w, lower=True)
    if full_cov:
        var = K(spec, Xnew) - np.matmul(w.T, w) + np.matmul(intermediateA.T, intermediateA)
        var = np.tile(var[:, :, None], [1, 1, Y.shape[1]])
    else:
        var = Kdiag(spec, Xnew) - np.sum(np.square(w), 0) + np.sum(np.square(intermediateA), 0)
        var = np.tile(var[:, None], [1, Y.shape[1]])
    return mean, var


def sgpr_upper_bound(spec, X, Y, Z, noise_var, jitter=JITTER):
    """SGPRUpperMixin.compute_upper_bound, models/sgpr.py:55-82"""
    num_data = float(Y.shape[0])
    Kd = Kdiag(spec, X)
    Kuu = K(spec, Z) + jitter * np.eye(Z.shape[0])
    Kuf = K(spec, Z, X)
    L = np.linalg.cholesky(Kuu)
    LB = np.linalg.cholesky(Kuu + noise_var ** -1.0 * np.matmul(Kuf, Kuf.T))
    LinvKuf = sl.solve_triangular(L, Kuf, lower=True)
    c = np.sum(Kd) - np.sum(LinvKuf ** 2.0)
    corrected_noise = noise_var + c
    const = -0.5 * num_data * np.log(2 * np.pi * noise_var)
    logdet = np.sum(np.log(np.diag(L))) - np.sum(np.log(np.diag(LB)))
    LC = np.linalg.cholesky(Kuu + corrected_noise ** -1.0 * np.matmul(Kuf, Kuf.T))
    v = sl.solve_triangular(LC, corrected_noise ** -1.0 * np.matmul(Kuf, Y), lower=True)
    quad = -0.5 * corrected_noise ** -1.0 * np.sum(Y ** 2.0) + 0.5 * np.sum(v ** 2.0)
    return const + logdet + quad


def gauss_kl(q_mu, q_sqrt, K=None):
    """kullback_leiblers.py:26-105"""
    if K is None:
        white, alpha = True, q_mu
    else:
        white = False
        Lp = np.linalg.cholesky(K)
        alpha = sl.solve_triangular(Lp, q_mu, lower=True)
    if q_sqrt.ndim == 2:
        diag, num_latent, NM = True, q_sqrt.shape[1], q_sqrt.size
        Lq = Lq_diag = q_sqrt
    else:
        diag, num_latent, NM = False, q_sqrt.shape[2], q_sqrt.shape[1] * q_sqrt.shape[2]
        Lq = np.tril(np.transpose(q_sqrt, (2, 0, 1)))
        Lq_diag = np.diagonal(Lq, axis1=1, axis2=2)
    mahalanobis = np.sum(np.square(alpha))
    constant = -float(NM)
    logdet_qcov = np.sum(np.log(np.square(Lq_diag)))
    if white:
        trace = np.sum(np.square(Lq))
    elif diag:
        Lp_inv = sl.solve_triangular(Lp, np.eye(Lp.shape[0]), lower=True)
        K_inv = sl.solve_triangular(Lp.T, Lp_inv, lower=False)
        trace = np.sum(np.diag(K_inv)[:, None] * np.square(q_sqrt))
    else:
        trace = sum(np.sum(np.square(sl.solve_triangular(Lp, Lq[i], lower=True))) for i in range(num_latent))
    twoKL = mahalanobis + constant - logdet_qcov + trace
    if not white:
        twoKL += num_latent * np.sum(np.log(np.square(np.diag(Lp))))
    return 0.5 * twoKL


def svgp_elbo(spec, X, Y, Z, q_mu, q_sqrt, noise_var, whiten=True, num_data=None, jitter=JITTER):
    """models/svgp.py:101-130 with likelihoods.Gaussian.variational_expectations (likelihoods.py:186-188)"""
    Kp = None if whiten else K(spec, Z) + jitter * np.eye(Z.shape[0])
    KL = gauss_kl(q_mu, q_sqrt, Kp)
    fmean, fvar = conditional(X, Z, spec, q_mu, full_cov=False, q_sqrt=q_sqrt, white=whiten, jitter=jitter)
    var_exp = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(noise_var) - 0.5 * (np.square(Y - fmean) + fvar) / noise_var
    scale = float(num_data or X.shape[0]) / float(X.shape[0])
    return np.sum(var_exp) * scale - KL


def _chol_adjoint(L, Lbar):
    """Kbar (symmetric) with <Kbar, dK> = <Lbar, dL> for L = chol(K) and symmetric dK (Murray 2016, eq. 10)."""
    P = np.tril(L.T @ np.tril(Lbar))
    P[np.diag_indices_from(P)] *= 0.5
    S = sl.solve_triangular(L, P + P.T, lower=True, trans='T')
    S = sl.solve_triangular(L, S.T, lower=True, trans='T')
    return 0.5 * S


def svgp_elbo_grad(spec_fn, theta, X, Y, Z, q_mu, q_sqrt, noise_var, num_data=None, jitter=JITTER, rel_step=1e-6):
    """Gradient of svgp_elbo (whitened parametrisation, models/svgp.py:101-130) -- what TF autodiff differentiates for
    examples/svgp.py:159-161 -- w.r.t. the flat constrained kernel parameters `theta` (spec_fn(theta) -> spec), the noise
    variance, q_mu and q_sqrt.  Reverse mode at the matrix level (A = Lm^-1 Kuf, Cholesky adjoint) with dK/dtheta by central
    differences of the oracle's own K: independent of the product's derivative formulas.  Returns
    (g_theta, g_noise, g_q_mu [M, K], g_q_sqrt like q_sqrt, dELBO/dmean(X) [N, K])."""
    theta = np.asarray(theta, dtype=np.float64)
    spec = spec_fn(theta)
    M, N, k = Z.shape[0], X.shape[0], q_mu.shape[1]
    w = float(num_data or N) / float(N)
    Kuu = K(spec, Z) + jitter * np.eye(M)
    Kuf = K(spec, Z, X)
    kd = Kdiag(spec, X)
    L = np.linalg.cholesky(Kuu)
    A = sl.solve_triangular(L, Kuf, lower=True)
    if q_sqrt.ndim == 3:
        Lq = np.tril(np.transpose(q_sqrt, (2, 0, 1)))
    else:
        Lq = np.stack([np.diag(q_sqrt[:, q]) for q in range(k)])
    mu = A.T @ q_mu
    var = np.stack([kd - np.sum(A * A, 0) + np.sum((Lq[q].T @ A) ** 2, 0) for q in range(k)], 1)
    E = w * (Y - mu) / noise_var
    sq = np.sum((Y - mu) ** 2 + var)
    g_noise = w * (-N * k / (2 * noise_var) + sq / (2 * noise_var ** 2))
    g_qmu = A @ E - q_mu
    AAT = A @ A.T
    gL = np.stack([np.tril(-(w / noise_var) * AAT @ Lq[q] - Lq[q] + np.diag(1.0 / np.diag(Lq[q]))) for q in range(k)])
    g_qsqrt = np.transpose(gL, (1, 2, 0)) if q_sqrt.ndim == 3 else np.stack([np.diag(gL[q]) for q in range(k)], 1)
    Ssum = sum(Lq[q] @ Lq[q].T for q in range(k))
    Abar = q_mu @ E.T + (w / noise_var) * (k * np.eye(M) - Ssum) @ A
    Kuf_bar = sl.solve_triangular(L, Abar, lower=True, trans='T')
    Kuu_bar = _chol_adjoint(L, -np.tril(Kuf_bar @ A.T))
    kd_bar = -w * k / (2 * noise_var)
    g = np.zeros_like(theta)
    for p in range(theta.size):
        h = rel_step * max(1.0, abs(theta[p]))
        tp, tm = theta.copy(), theta.copy()
        tp[p] += h; tm[p] -= h
        sp, sm = spec_fn(tp), spec_fn(tm)
        g[p] = (np.sum(Kuu_bar * (K(sp, Z) - K(sm, Z))) + np.sum(Kuf_bar * (K(sp, Z, X) - K(sm, Z, X)))
                + kd_bar * np.sum(Kdiag(sp, X) - Kdiag(sm, X))) / (2 * h)
    return g, g_noise, g_qmu, g_qsqrt, E


# ---------------------------------------------------------------------------------------------
# bench.py cpu_baseline leg: the same path with LAPACK dpotrf/dtrtrs (algorithm class of TF-CPU's
# Eigen LLT / triangular solve), K built the unfused way the reference's TF graph does.
def gpr_lml_timed(spec, X, Y, noise_var):
    import time
    t0 = time.perf_counter()
    Kmat = K(spec, X) + np.eye(X.shape[0]) * noise_var
    t1 = time.perf_counter()
    L = sl.cholesky(Kmat, lower=True, overwrite_a=True, check_finite=False)
    t2 = time.perf_counter()
    alpha = sl.solve_triangular(L, Y, lower=True, check_finite=False)
    n, r = Y.shape
    lml = -0.5 * n * r * np.log(2 * np.pi) - r * np.sum(np.log(np.diag(L))) - 0.5 * np.sum(np.square(alpha))
    t3 = time.perf_counter()
    return lml, {"kmat_s": t1 - t0, "potrf_s": t2 - t1, "trsv_s": t3 - t2, "total_s": t3 - t0}


def rbf_K_inplace(spec, X):
    """RBF K(X, X) with the passes of kernels.py:408-439 fused by hand (one matmul, then in-place numpy ops on the one
    [N, N] buffer): the SURVEY 8(d) "in-place" CPU variant timed beside the unfused, one-op-per-TF-op build of K()."""
    A = _slice(spec, X, None)[0] / spec["lengthscales"]
    s = np.sum(A * A, axis=1)
    Kb = A @ A.T
    Kb *= -2.0
    Kb += s[:, None]
    Kb += s[None, :]
    np.maximum(Kb, 0.0, out=Kb)
    Kb *= -0.5
    np.exp(Kb, out=Kb)
    Kb *= spec["variance"]
    return Kb


def synthetic_gpr_data(n, d, n_new=0, seed=20240607):
    """SURVEY.md section 8(d) synthetic inputs."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal((d, 1)) / np.sqrt(d)
    Y = np.sin(X @ w) + 0.1 * rng.standard_normal((n, 1))
    Xnew = rng.standard_normal((n_new, d)) if n_new else None
    return X, Y, Xnew
