"""Counterpart of the reference's examples/svgp.py on the MI355X path, with the part of it this library covers: the
reference script trains an SVGP (whiten=False, Z initialised from the training inputs, Adam on `objective`,
examples/svgp.py:144-161) with a MultiClass likelihood on MNIST; the likelihood here is Gaussian (regression on synthetic
data -- non-Gaussian likelihoods are outside the scope of this library), everything else follows the script: the same model
construction, the same optimiser on `objective` over every parameter including the inducing inputs, predictions at
intervals.  An SGPR on the same data shows the collapsed bound the SVGP bound approaches.

    python examples/svgp.py [--iters 300] [--n 4000] [--m 64]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
import gpflowSlim as gpf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--n", type=int, default=4000)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--lr", type=float, default=2e-2)
    args = ap.parse_args()

    rng = np.random.default_rng(0)
    D = 4
    X = rng.uniform(-2.0, 2.0, (args.n, D))
    f = lambda x: np.sin(2.0 * x[:, :1]) * np.cos(x[:, 1:2]) + 0.3 * x[:, 2:3]
    Y = f(X) + 0.1 * rng.standard_normal((args.n, 1))
    Xt = rng.uniform(-2.0, 2.0, (500, D)); Yt = f(Xt) + 0.1 * rng.standard_normal((500, 1))
    Z = X[rng.choice(args.n, args.m, replace=False)].copy()            # examples/svgp.py:139-141

    kern = gpf.kernels.RBF(D, ARD=True)
    model = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.1), Z=Z, num_latent=1, whiten=False,
                            minibatch_size=None, num_data=args.n, train_inducing=True)      # :144-146

    def report(it, obj):
        if it % max(1, args.iters // 6) == 0 or it == args.iters:
            mu, var = model.predict_y(Xt)
            rmse = float(np.sqrt(np.mean((mu - Yt) ** 2)))
            ll = float(np.mean(model.predict_density(Xt, Yt)))
            print("iter %4d  objective %12.3f  test rmse %.4f  test log-density %.4f" % (it, obj, rmse, ll), flush=True)

    t0 = time.perf_counter()
    final = model.optimize(max_iter=args.iters, method="adam", learning_rate=args.lr, callback=report)
    print("SVGP: objective %.3f after %d Adam steps in %.1f s; inducing inputs moved by up to %.3f" % (
        final, args.iters, time.perf_counter() - t0, float(np.abs(np.asarray(model.feature.Z) - Z).max())))
    sgpr = gpf.models.SGPR(X, Y, gpf.kernels.RBF(D, ARD=True), Z=Z, obs_var=0.1)
    t0 = time.perf_counter()
    fs = sgpr.optimize(max_iter=min(args.iters, 100))
    mu, _ = sgpr.predict_y(Xt)
    print("SGPR: objective %.3f (L-BFGS-B, %.1f s), test rmse %.4f" % (fs, time.perf_counter() - t0,
                                                                     float(np.sqrt(np.mean((mu - Yt) ** 2)))))
    return final, fs


if __name__ == "__main__":
    main()
