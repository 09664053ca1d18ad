"""Scratch: A/B of one option on the other entry points (warm predict_f at N = 32768, conditional M = 4096 x N = 1e6, LML + gradient)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
opt = sys.argv[1]; vals = [float(v) for v in sys.argv[2].split(",")]
n, d = 32768, 8
X, Y, Xs = orc.synthetic_gpr_data(n, d, 1024)
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1); m.reuse_factor = True
m.compute_log_likelihood(); m.predict_f(Xs); m.compute_log_likelihood_and_gradients()
rng = np.random.default_rng(1)
Xc = rng.standard_normal((1000000, d)); Z = Xc[:4096].copy(); f = rng.standard_normal((4096, 1))
gpf.conditionals.conditional(Xc, Z, kern, f, white=True)
for rep in range(3):
    for v in vals:
        h.set_option(opt, v)
        m.compute_log_likelihood()
        t0 = time.perf_counter(); mu, var = m.predict_f(Xs); t1 = time.perf_counter()
        t2 = time.perf_counter(); m.compute_log_likelihood_and_gradients(); t3 = time.perf_counter()
        t4 = time.perf_counter(); cm, cv = gpf.conditionals.conditional(Xc, Z, kern, f, white=True); t5 = time.perf_counter()
        print("%s=%g: predict warm %.2f ms | lml+grad %.1f ms | conditional 1e6 x 4096 %.1f ms (checks %.12g %.12g)" % (opt, v, 1e3 * (t1 - t0), 1e3 * (t3 - t2), 1e3 * (t5 - t4), float(mu.sum()), float(cm.sum())))
