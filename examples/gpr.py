"""Counterpart of the reference's examples/gpr.py (Boston-housing GPR trained with Adam) on the
MI355X path.  `sklearn.datasets.load_boston` no longer exists, so the data are synthetic with the same
shape (N ~ 455 train / 51 test, D = 13); everything else follows the reference script: standardise,
RBF(13, ARD=True), GPR(x, y[:, None], kern), Adam(1e-3) on `objective`, predict_f every few hundred
steps, RMSE and mean test log-likelihood (examples/gpr.py:36-77).

    python examples/gpr.py [--iters 2000] [--n 506]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
import gpflowSlim as gpf  # noqa: E402


class Adam(object):
    """tf.train.AdamOptimizer(learning_rate) on the model's unconstrained parameters."""

    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps, self.t = params, lr, b1, b2, eps, 0
        self.m = [np.zeros_like(np.atleast_1d(p.vf_val)) for p in params]
        self.v = [np.zeros_like(np.atleast_1d(p.vf_val)) for p in params]

    def step(self, grads_of_objective):
        self.t += 1
        for i, (p, g) in enumerate(zip(self.params, grads_of_objective)):
            g = np.atleast_1d(g)
            self.m[i] = self.b1 * self.m[i] + (1 - self.b1) * g
            self.v[i] = self.b2 * self.v[i] + (1 - self.b2) * g * g
            mh = self.m[i] / (1 - self.b1 ** self.t)
            vh = self.v[i] / (1 - self.b2 ** self.t)
            p.assign_unconstrained(np.atleast_1d(p.vf_val) - self.lr * mh / (np.sqrt(vh) + self.eps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--n", type=int, default=506)
    ap.add_argument("--lr", type=float, default=1e-2)
    args = ap.parse_args()

    rng = np.random.default_rng(1)
    D = 13
    X = rng.standard_normal((args.n, D)) * rng.uniform(0.5, 3.0, D)
    w = rng.standard_normal(D) * (rng.uniform(size=D) < 0.5)
    y = np.tanh(X @ w / 3.0) * 9.0 + 22.0 + rng.standard_normal(args.n) * 2.0
    n_test = max(1, args.n // 10)
    perm = rng.permutation(args.n)
    x_train, y_train = X[perm[n_test:]], y[perm[n_test:]]
    x_test, y_test = X[perm[:n_test]], y[perm[:n_test]]
    mx, sx = x_train.mean(0), x_train.std(0)
    my, sy = y_train.mean(), y_train.std()
    x_train, x_test = (x_train - mx) / sx, (x_test - mx) / sx
    y_train_s, y_test_s = (y_train - my) / sy, (y_test - my) / sy

    k = gpf.kernels.RBF(D, ARD=True)                                   # examples/gpr.py:48
    m = gpf.models.GPR(x_train, y_train_s[:, None], kern=k)            # examples/gpr.py:49
    opt = Adam(m.parameters, lr=args.lr)
    t0 = time.perf_counter()
    for it in range(args.iters + 1):
        lml, grads = m.compute_log_likelihood_and_gradients()
        opt.step([-g for _, g in grads])                               # objective = -LML
        if it % max(1, args.iters // 10) == 0:
            mu, var = m.predict_f(x_test)
            mu, var = mu[:, 0], var[:, 0]
            rmse = np.sqrt(np.mean((mu - y_test_s) ** 2)) * sy
            obs = var + float(np.squeeze(m.likelihood.variance))
            ll = np.mean(-0.5 * np.log(2 * np.pi * obs) - 0.5 * (y_test_s - mu) ** 2 / obs) - np.log(sy)
            print("iter %5d  objective %.4f  test rmse %.4f  test log-lik %.4f  (%.1f s)" % (it, -lml, rmse, ll, time.perf_counter() - t0))
    print("lengthscales", np.round(np.atleast_1d(k.lengthscales), 3), "variance", float(np.squeeze(k.variance)), "noise", float(np.squeeze(m.likelihood.variance)))


if __name__ == "__main__":
    main()
