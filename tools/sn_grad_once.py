"""A few likelihood + gradient evaluations of a small problem (for rocprofv3 --kernel-trace): python tools/sn_grad_once.py [N] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(0)
d = 8
X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
kern = gpf.kernels.RBF(d, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
h = gpf.get_handle()
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
m.compute_log_likelihood()
prog = kern._program(d)
for i in range(reps):
    h.gpr_lml_grad(prog, 0.1, Y)
t0 = time.perf_counter()
for i in range(reps):
    h.gpr_lml_grad(prog, 0.1, Y)
print("N=%d lml_grad %.1f us" % (n, 1e6 * (time.perf_counter() - t0) / reps))
