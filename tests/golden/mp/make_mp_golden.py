"""50-digit pins of the GPR path, independent of oracle/ (imports numpy and mpmath only).

The formulas of SURVEY section 9 -- kernels.py:408-439, 557-610, 806-819, 1071-1084; models/gpr.py:69-72, 119-131;
densities.py:81-94, conditionals.py:24-121 of the reference -- evaluated with mpmath at 50 digits on small seeded inputs:
log-marginal likelihood, posterior mean and variance of GPR; mean and (co)variance of conditional() for q_sqrt None / [M, K] /
[M, M, K], whitened or not (conditional.npz); gauss_kl and the SVGP bound with the Gaussian likelihood (svgp.npz).  The fp64 parameter values stored in the fixture are the ones the formulas were evaluated with;
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


def mp_kernel(mp, spec, x, y, same):
    """k(x, y) at working precision; same: x and y are the same data point (White, and r2 = 0 exactly)."""
    t = spec["type"]
    if t == "sum":
        return sum(mp_kernel(mp, ch, x, y, same) for ch in spec["children"])
    if t == "product":
        out = mp.mpf(1)
        for ch in spec["children"]:
            out *= mp_kernel(mp, ch, x, y, same)
        return out
    v = mp.mpf(float(spec["variance"]))
    if t == "white":
        return v if same else mp.mpf(0)
    if t == "constant":
        return v
    ad = spec.get("active_dims") or list(range(spec["input_dim"]))
    xs = [mp.mpf(float(x[d])) for d in ad]; ys = [mp.mpf(float(y[d])) for d in ad]
    if t == "periodic":                                   # kernels.py:806-819
        p, l = mp.mpf(float(spec["period"])), mp.mpf(float(spec["lengthscales"]))
        return v * mp.exp(-sum((mp.sin(mp.pi * (a - b) / p) / l) ** 2 for a, b in zip(xs, ys)) / 2)
    ls = np.broadcast_to(np.asarray(spec["lengthscales"], dtype=float), (len(ad),))
    r2 = sum(((a - b) / mp.mpf(float(l))) ** 2 for a, b, l in zip(xs, ys, ls))      # kernels.py:408-421
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
    np.savez(os.path.join(HERE, "svgp.npz"), **out)
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
