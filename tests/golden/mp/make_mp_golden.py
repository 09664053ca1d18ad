"""50-digit pins of the GPR path, independent of oracle/ (imports numpy and mpmath only).

The formulas of SURVEY section 9 -- kernels.py:408-439, 557-610, 806-819, 1071-1084; models/gpr.py:69-72, 119-131;
densities.py:81-94 of the reference -- evaluated with mpmath at 50 digits on small seeded inputs: log-marginal likelihood,
posterior mean and variance.  The fp64 parameter values stored in the fixture are the ones the formulas were evaluated with;
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


def main():
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
