"""50-digit pins of the GPR path, independent of oracle/ (imports numpy and mpmath only).

The formulas of SURVEY section 9 -- kernels.py:408-439, 557-610, 806-819, 1071-1084; models/gpr.py:69-72, 119-131;
densities.py:81-94, conditionals.py:24-121 of the reference -- evaluated with mpmath at 50 digits on small seeded inputs:
log-marginal likelihood, posterior mean and variance of GPR; mean and (co)variance of conditional() for q_sqrt None / [M, K] /
[M, M, K], whitened or not (conditional.npz); gauss_kl and the SVGP bound with the Gaussian likelihood, the SGPR bound and the FITC likelihood from their dense
definitions (svgp.npz); the gradient of the GPR likelihood by 1e-25 central differences of the 60-digit likelihood (gradient.npz).  The fp64 parameter values stored in the fixture are the ones the formulas were evaluated with;
tests/test_gpu_pins.py feeds exactly those to the HIP path and compares at 1e-8, with no oracle in between.
    python tests/golden/mp/make_mp_golden.py        # rewrites tests/golden/mp/*.npz
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# kernel descriptions: plain dicts of fp64 values (what tests/test_gpu_pins.py builds gpflowSlim kernels from)
SPECS = {
    "rbf_ard": {"type": "rbf", "variance": 1.3, "lengthscales": [0.7, 1.1, 1.6, 2.0], "input_dim": 4},
    "matern52": {"type": "matern52", "variance": 0.9, "lengthscales": 1.4, "input_dim": 4},
    "periodic": {"type": "periodic", "variance": 1.1, "lengthscales": 1.3, "period": 2.5, "input_dim": 4},
    "sum": {"type": "sum", "children": [
        {"type": "matern32", "variance": 0.6, "lengthscales": 0.9, "input_dim": 2, "active_dims": [0, 3]},
        {"type": "periodic", "variance": 0.8, "lengthscales": 1.0, "period": 3.0, "input_dim": 4}]},
    "product": {"type": "product", "children": [
        {"type": "rbf", "variance": 1.2, "lengthscales": 1.5, "input_dim": 4},
        {"type": "matern12", "variance": 0.7, "lengthscales": 2.0, "input_dim": 4},
        {"type": "constant", "variance": 1.5}]},
}
NOISE = 0.1


def _mpf(mp, v):
    return v if isinstance(v, mp.mpf) else mp.mpf(float(v))


def mp_kernel(mp, spec, x, y, same):
    """k(x, y) at working precision; same: x and y are the same data point (White, and r2 = 0 exactly).  Parameter values may be
    fp64 numbers or mpmath numbers (the gradient pins perturb them by 1e-25)."""
    t = spec["type"]
    if t == "sum":
        return sum(mp_kernel(mp, ch, x, y, same) for ch in spec["children"])
    if t == "product":
        out = mp.mpf(1)
        for ch in spec["children"]:
            out *= mp_kernel(mp, ch, x, y, same)
        return out
    v = _mpf(mp, spec["variance"])
    if t == "white":
        return v if same else mp.mpf(0)
    if t == "constant":
        return v
    ad = spec.get("active_dims") or list(range(spec["input_dim"]))
    xs = [mp.mpf(float(x[d])) for d in ad]; ys = [mp.mpf(float(y[d])) for d in ad]
    if t == "periodic":                                   # kernels.py:806-819
        p, l = _mpf(mp, spec["period"]), _mpf(mp, spec["lengthscales"])
        return v * mp.exp(-sum((mp.sin(mp.pi * (a - b) / p) / l) ** 2 for a, b in zip(xs, ys)) / 2)
    ls = spec["lengthscales"]
    ls = list(ls) if isinstance(ls, (list, tuple, np.ndarray)) else [ls] * len(ad)
    r2 = sum(((a - b) / _mpf(mp, l)) ** 2 for a, b, l in zip(xs, ys, ls))      # kernels.py:408-421
    if t == "rbf":
        return v * mp.exp(-r2 / 2)                        # :439
    r = mp.sqrt(r2 + mp.mpf("1e-12"))                     # :426
    if t == "matern12":
        return v * mp.exp(-r)
    if t == "matern32":
        return v * (1 + mp.sqrt(3) * r) * mp.exp(-mp.sqrt(3) * r)
    if t == "matern52":
        return v * (1 + mp.sqrt(5) * r + mp.mpf(5) / 3 * r * r) * mp.exp(-mp.sqrt(5) * r)
    raise ValueError(t)


def mp_kdiag(mp, spec):
    """Kdiag (kernels.py:428-429, 803-804, 1079-1084): the variances folded through Sum / Product -- not k(x, x)."""
    t = spec["type"]
    if t == "sum":
        return sum(mp_kdiag(mp, ch) for ch in spec["children"])
    if t == "product":
        out = mp.mpf(1)
        for ch in spec["children"]:
            out *= mp_kdiag(mp, ch)
        return out
    return mp.mpf(float(spec["variance"]))


def mp_gpr(spec, X, Y, s2, Xs, dps=50):
    import mpmath as mp
    mp.mp.dps = dps
    n, r = Y.shape
    Km = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            Km[i, j] = mp_kernel(mp, spec, X[i], X[j], i == j) + (mp.mpf(float(s2)) if i == j else 0)
    L = mp.cholesky(Km)
    slog = sum(mp.log(L[i, i]) for i in range(n))
    lml = -mp.mpf(n * r) / 2 * mp.log(2 * mp.pi) - r * slog                 # densities.py:92-94
    alphas = []
    for q in range(r):
        alpha = mp.lu_solve(L, mp.matrix([float(v) for v in Y[:, q]]))
        lml -= sum(a * a for a in alpha) / 2
        alphas.append(alpha)
    mu = np.zeros((Xs.shape[0], r)); var = np.zeros(Xs.shape[0])
    for s in range(Xs.shape[0]):
        kx = mp.matrix([mp_kernel(mp, spec, X[i], Xs[s], False) for i in range(n)])
        a = mp.lu_solve(L, kx)                                                 # models/gpr.py:122
        for q in range(r):
            mu[s, q] = float(sum(a[i] * alphas[q][i] for i in range(n)))       # :124
        var[s] = float(mp_kdiag(mp, spec) - sum(a[i] * a[i] for i in range(n)))    # :130
    return float(lml), mu, var


def mp_lml(mp, spec, X, Y, s2):
    """log-marginal likelihood (densities.py:81-94) as an mpmath number; spec and s2 may hold mpmath values"""
    n, r = Y.shape
    Km = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            Km[i, j] = mp_kernel(mp, spec, X[i], X[j], i == j) + (_mpf(mp, s2) if i == j else 0)
    L = mp.cholesky(Km)
    out = -mp.mpf(n * r) / 2 * mp.log(2 * mp.pi) - r * sum(mp.log(L[i, i]) for i in range(n))
    for q in range(r):
        a = mp.lu_solve(L, mp.matrix([float(v) for v in Y[:, q]]))
        out -= sum(x * x for x in a) / 2
    return out


# parameter vectors of the gradient pins: theta -> spec, in the slot order of gps_gpr_lml_grad (per primitive: variance, then the
# length-scales [, period])
GRAD_SPECS = {
    "rbf_ard": ([1.3, 0.7, 1.1, 1.6, 2.0], lambda t: {"type": "rbf", "variance": t[0], "lengthscales": list(t[1:5]), "input_dim": 4}),
    "matern52": ([0.9, 1.4], lambda t: {"type": "matern52", "variance": t[0], "lengthscales": t[1], "input_dim": 4}),
    "periodic": ([1.1, 1.3, 2.5], lambda t: {"type": "periodic", "variance": t[0], "lengthscales": t[1], "period": t[2], "input_dim": 4}),
}


def mp_lml_grad(name, X, Y, s2, dps=60):
    """d LML / d theta and d LML / d noise by central differences of the 60-digit likelihood with a step of 1e-25: exact to
    ~1e-40 -- independent of every derivative formula (what TF autodiff supplies to examples/gpr.py:53-54)."""
    import mpmath as mp
    mp.mp.dps = dps
    theta0, fn = GRAD_SPECS[name]
    h = mp.mpf("1e-25")
    grads = []
    for i in range(len(theta0)):
        tp = [mp.mpf(float(v)) for v in theta0]; tm = list(tp)
        tp[i] += h; tm[i] -= h
        grads.append(float((mp_lml(mp, fn(tp), X, Y, s2) - mp_lml(mp, fn(tm), X, Y, s2)) / (2 * h)))
    s = mp.mpf(float(s2))
    spec = fn([mp.mpf(float(v)) for v in theta0])
    gn = float((mp_lml(mp, spec, X, Y, s + h) - mp_lml(mp, spec, X, Y, s - h)) / (2 * h))
    return np.array(grads), gn, float(mp_lml(mp, spec, X, Y, s))


def mp_conditional(spec, Z, Xn, f, q_sqrt, white, full_cov, jitter=1e-6, dps=50):
    """conditionals.py:24-66 + 80-121 at working precision: Kmm = K(Z) + jitter I, Kmn = K(Z, Xnew), Knn = Kdiag(Xnew) or K(Xnew);
    q_sqrt None, [M, K] (standard deviations) or [M, M, K] (lower-triangular factors).  Returns fmean [N, K], fvar [N, K] or [N, N, K]."""
    import mpmath as mp
    mp.mp.dps = dps
    M, N, K = Z.shape[0], Xn.shape[0], f.shape[1]
    Kmm = mp.matrix(M, M)
    for i in range(M):
        for j in range(M):
            Kmm[i, j] = mp_kernel(mp, spec, Z[i], Z[j], i == j) + (mp.mpf(jitter) if i == j else 0)
    Kmn = mp.matrix(M, N)
    for i in range(M):
        for j in range(N):
            Kmn[i, j] = mp_kernel(mp, spec, Z[i], Xn[j], False)
    Lm = mp.cholesky(Kmm)                                                    # :84
    A = mp.matrix(M, N)
    for j in range(N):
        col = mp.lu_solve(Lm, Kmn[:, j])                                     # :87
        for i in range(M):
            A[i, j] = col[i]
    if full_cov:                                                              # :90-96
        base = mp.matrix(N, N)
        for a in range(N):
            for b in range(N):
                base[a, b] = mp_kernel(mp, spec, Xn[a], Xn[b], a == b) - sum(A[i, a] * A[i, b] for i in range(M))
    else:
        base = [mp_kdiag(mp, spec) - sum(A[i, j] * A[i, j] for i in range(M)) for j in range(N)]
    if not white:                                                             # :99-100
        A2 = mp.matrix(M, N)
        for j in range(N):
            col = mp.lu_solve(Lm.T, A[:, j])
            for i in range(M):
                A2[i, j] = col[i]
        A = A2
    fmean = np.zeros((N, K))
    fvar = np.zeros((N, N, K) if full_cov else (N, K))
    for k in range(K):
        for j in range(N):
            fmean[j, k] = float(sum(A[i, j] * mp.mpf(float(f[i, k])) for i in range(M)))      # :103
        if q_sqrt is None:
            LTA = None
        elif q_sqrt.ndim == 2:                                                # :106-107
            LTA = mp.matrix(M, N)
            for i in range(M):
                for j in range(N):
                    LTA[i, j] = A[i, j] * mp.mpf(float(q_sqrt[i, k]))
        else:                                                                 # :108-111 (lower band of q_sqrt[:, :, k], transposed)
            LTA = mp.matrix(M, N)
            for i in range(M):
                for j in range(N):
                    LTA[i, j] = sum(mp.mpf(float(q_sqrt[r, i, k])) * A[r, j] for r in range(i, M))
        if full_cov:
            for a in range(N):
                for b in range(N):
                    v = base[a, b] + (sum(LTA[i, a] * LTA[i, b] for i in range(M)) if LTA is not None else 0)
                    fvar[a, b, k] = float(v)
        else:
            for j in range(N):
                fvar[j, k] = float(base[j] + (sum(LTA[i, j] * LTA[i, j] for i in range(M)) if LTA is not None else 0))
    return fmean, fvar


def mp_gauss_kl(spec, Z, q_mu, q_sqrt, white, jitter=1e-6, dps=50):
    """kullback_leiblers.py:26-105: KL[N(q_mu, q_sqrt q_sqrt^T) || N(0, K)] summed over the columns; K = K(Z) + jitter I
    (models/svgp.py:101-106), or the identity when whitened.  q_sqrt [M, K] (diagonal) or [M, M, K] (lower factors)."""
    import mpmath as mp
    mp.mp.dps = dps
    M, K = q_mu.shape
    if not white:
        Kmm = mp.matrix(M, M)
        for i in range(M):
            for j in range(M):
                Kmm[i, j] = mp_kernel(mp, spec, Z[i], Z[j], i == j) + (mp.mpf(jitter) if i == j else 0)
        Lp = mp.cholesky(Kmm)
    total = mp.mpf(0)
    for k in range(K):
        mu = mp.matrix([float(v) for v in q_mu[:, k]])
        alpha = mu if white else mp.lu_solve(Lp, mu)
        maha = sum(a * a for a in alpha)
        if q_sqrt.ndim == 2:
            Lq = mp.diag([float(v) for v in q_sqrt[:, k]])
        else:
            Lq = mp.matrix(M, M)
            for i in range(M):
                for j in range(i + 1):
                    Lq[i, j] = mp.mpf(float(q_sqrt[i, j, k]))
        logdet_q = sum(mp.log(Lq[i, i] ** 2) for i in range(M))
        if white:
            trace = sum(Lq[i, j] ** 2 for i in range(M) for j in range(M))
        else:
            trace = mp.mpf(0)
            for j in range(M):
                col = mp.lu_solve(Lp, Lq[:, j])                       # Lp^-1 Lq, column by column (the diagonal form is the same sum)
                trace += sum(c * c for c in col)
        two = maha - M - logdet_q + trace
        if not white:
            two += sum(mp.log(Lp[i, i] ** 2) for i in range(M))
        total += two / 2
    return float(total)


def svgp_inputs(name):
    rng = np.random.default_rng(177 + len(name))
    N, M, K = 24, 10, 2
    X = rng.standard_normal((N, 4)); Y = rng.standard_normal((N, K)); Z = rng.standard_normal((M, 4))
    q_mu = rng.standard_normal((M, K)) * 0.5
    q_diag = np.abs(rng.standard_normal((M, K))) * 0.3 + 0.05
    q_full = np.tril(rng.standard_normal((K, M, M)) * 0.05 + np.eye(M) * 0.3).transpose(1, 2, 0).copy()
    return X, Y, Z, q_mu, q_diag, q_full


SVGP_CASES = [(name, white, q) for name in ("rbf_ard", "matern52") for white in (True, False) for q in ("diag", "full")]
SVGP_NOISE, SVGP_NUM_DATA = 0.2, 72


def mp_svgp_elbo(spec, X, Y, Z, q_mu, q_sqrt, white):
    """models/svgp.py:108-125 with the Gaussian likelihood (likelihoods.py:186-188): sum of the variational expectations, scaled by
    num_data / N, minus the KL.  (The conditional and the KL at 50 digits; the N x K expectations summed in fp64.)"""
    fmean, fvar = mp_conditional(spec, Z, X, q_mu, q_sqrt, white, False)
    ve = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(SVGP_NOISE) - 0.5 * ((Y - fmean) ** 2 + fvar) / SVGP_NOISE
    kl = mp_gauss_kl(spec, Z, q_mu, q_sqrt, white)
    return float(ve.sum()) * SVGP_NUM_DATA / X.shape[0] - kl, kl


def mp_sparse_bounds(spec, X, Y, Z, s2, jitter=1e-6, dps=50):
    """The SGPR bound (models/sgpr.py:121-155) and the FITC likelihood (:229-282) from their DEFINITIONS rather than the
    reference's Woodbury algebra: with Qff = Kfu (Kuu + jitter I)^-1 Kuf,
        SGPR  = sum_r log N(y_r | 0, Qff + s2 I) - R / (2 s2) tr(Kff - Qff)            (Titsias 2009)
        FITC  = sum_r log N(y_r | 0, Qff + diag(Kff - Qff) + s2 I)
    as dense N x N problems at working precision (Kff only through its diagonal = Kdiag, kernels.py:428-429)."""
    import mpmath as mp
    mp.mp.dps = dps
    N, R = Y.shape
    M = Z.shape[0]
    Kuu = mp.matrix(M, M)
    for i in range(M):
        for j in range(M):
            Kuu[i, j] = mp_kernel(mp, spec, Z[i], Z[j], i == j) + (mp.mpf(jitter) if i == j else 0)
    Kuf = mp.matrix(M, N)
    for i in range(M):
        for j in range(N):
            Kuf[i, j] = mp_kernel(mp, spec, Z[i], X[j], False)
    Lu = mp.cholesky(Kuu)
    V = mp.matrix(M, N)
    for j in range(N):
        col = mp.lu_solve(Lu, Kuf[:, j])
        for i in range(M):
            V[i, j] = col[i]
    Qff = V.T * V
    kd = mp_kdiag(mp, spec)
    s2m = mp.mpf(float(s2))

    def logpdf(C):
        L = mp.cholesky(C)
        ld = 2 * sum(mp.log(L[i, i]) for i in range(N))
        tot = mp.mpf(0)
        for r in range(R):
            a = mp.lu_solve(L, mp.matrix([float(v) for v in Y[:, r]]))
            tot += -mp.mpf(N) / 2 * mp.log(2 * mp.pi) - ld / 2 - sum(x * x for x in a) / 2
        return tot
    C1 = Qff.copy(); C2 = Qff.copy()
    for i in range(N):
        C1[i, i] += s2m
        C2[i, i] = kd + s2m                                   # Qff_ii + (Kff_ii - Qff_ii) + s2
    tr = sum(kd - Qff[i, i] for i in range(N))
    return float(logpdf(C1) - R * tr / (2 * s2m)), float(logpdf(C2))


COND_CASES = [(name, white, q, fc) for name in ("rbf_ard", "matern52", "sum") for white in (True, False)
              for q, fc in (("none", False), ("diag", False), ("full", False), ("full", True))]


def cond_inputs(name):
    rng = np.random.default_rng(77 + len(name))
    M, N, K = 14, 6, 2
    Z = rng.standard_normal((M, 4)); Xn = rng.standard_normal((N, 4)); f = rng.standard_normal((M, K))
    q_diag = np.abs(rng.standard_normal((M, K))) * 0.3 + 0.05
    q_full = np.tril(rng.standard_normal((K, M, M)) * 0.05 + np.eye(M) * 0.2).transpose(1, 2, 0).copy()
    return Z, Xn, f, q_diag, q_full


def main():
    out = {}
    for name in ("rbf_ard", "matern52", "sum"):
        Z, Xn, f, q_diag, q_full = cond_inputs(name)
        out.update({"%s_Z" % name: Z, "%s_Xn" % name: Xn, "%s_f" % name: f, "%s_qdiag" % name: q_diag, "%s_qfull" % name: q_full})
    for name, white, q, fc in COND_CASES:
        Z, Xn, f, q_diag, q_full = cond_inputs(name)
        mu, var = mp_conditional(SPECS[name], Z, Xn, f, {"none": None, "diag": q_diag, "full": q_full}[q], white, fc)
        tag = "%s_%s_%s_%s" % (name, "white" if white else "unwhite", q, "fullcov" if fc else "diag")
        out[tag + "_mu"] = mu; out[tag + "_var"] = var
        print("conditional", tag, float(mu.sum()), float(var.sum()))
    np.savez(os.path.join(HERE, "conditional.npz"), **out)
    out = {}
    for name in ("rbf_ard", "matern52"):
        X, Y, Z, q_mu, q_diag, q_full = svgp_inputs(name)
        out.update({"%s_X" % name: X, "%s_Y" % name: Y, "%s_Z" % name: Z, "%s_qmu" % name: q_mu, "%s_qdiag" % name: q_diag, "%s_qfull" % name: q_full})
    for name, white, q in SVGP_CASES:
        X, Y, Z, q_mu, q_diag, q_full = svgp_inputs(name)
        elbo, kl = mp_svgp_elbo(SPECS[name], X, Y, Z, q_mu, q_diag if q == "diag" else q_full, white)
        tag = "%s_%s_%s" % (name, "white" if white else "unwhite", q)
        out[tag + "_elbo"] = elbo; out[tag + "_kl"] = kl
        print("svgp", tag, elbo, kl)
    for name in ("rbf_ard", "matern52"):
        X, Y, Z, _, _, _ = svgp_inputs(name)
        sg, fi = mp_sparse_bounds(SPECS[name], X, Y, Z, SVGP_NOISE)
        out[name + "_sgpr_bound"] = sg; out[name + "_fitc_lml"] = fi
        print("sparse", name, sg, fi)
    np.savez(os.path.join(HERE, "svgp.npz"), **out)
    out = {}
    for name in sorted(GRAD_SPECS):
        rng = np.random.default_rng(31 + len(name))
        X = rng.standard_normal((16, 4)); Y = rng.standard_normal((16, 2))
        g, gn, lml = mp_lml_grad(name, X, Y, NOISE)
        out.update({name + "_X": X, name + "_Y": Y, name + "_grad": g, name + "_grad_noise": gn, name + "_lml": lml})
        print("gradient", name, lml, g, gn)
    np.savez(os.path.join(HERE, "gradient.npz"), **out)
    for name, spec in sorted(SPECS.items()):
        for n in (4, 16, 32):
            rng = np.random.default_rng(1000 + n)
            X = rng.standard_normal((n, 4)); Xs = rng.standard_normal((5, 4))
            r = 2 if n == 16 else 1
            Y = rng.standard_normal((n, r))
            lml, mu, var = mp_gpr(spec, X, Y, NOISE, Xs)
            np.savez(os.path.join(HERE, "%s_n%d.npz" % (name, n)), X=X, Y=Y, Xs=Xs, noise_var=NOISE, lml=lml, mu=mu, var=var)
            print(name, n, lml)


if __name__ == "__main__":
    main()
