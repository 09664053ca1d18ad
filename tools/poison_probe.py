"""Scratch: one handle, N = 5200, sweep-step mode from argv, 6 evaluations -- prints the likelihoods (GPS_POISON_ALLOC=1 outside)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
from gpflowSlim import _backend as be
mode = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 5200
d = 6
rng = np.random.default_rng(n)
X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
ls = np.sqrt(d) * np.linspace(0.8, 1.2, d)
prog = gpf.kernels.RBF(d, variance=1.1, lengthscales=ls, ARD=True)._program(d)
h = be.Handle(0)
h.set_option("potrf_fused_step", mode); h.set_option("potrf_two_stage_join", 1 if mode else 0)
h.gpr_set_data(X, X)
print(mode, n, [repr(h.gpr_lml(prog, 0.1, Y)) for _ in range(6)], h.profile_get("lookahead_retries")["launches"])
