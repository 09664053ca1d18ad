"""Does a look-ahead time-out + retry give the look-ahead-off result?  (injected faults; conditional at M = 4096 and GPR)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
from gpflowSlim import _backend as be

rng = np.random.default_rng(1000000)
d, k, m, n = 4, 2, 4096, 20000
Xnew = rng.standard_normal((n, d)); Z = rng.standard_normal((m, d)); f = rng.standard_normal((m, k))
prog = gpf.kernels.RBF(d, variance=1.2, lengthscales=1.4)._program(d)
h = be.Handle(0)
on = h.conditional(prog, Z, Xnew, f, 1e-6, white=True)
h.set_option("potrf_lookahead", 0)
off = h.conditional(prog, Z, Xnew, f, 1e-6, white=True)
h.set_option("potrf_lookahead", 1)
print("on == off:", np.array_equal(on[0], off[0]), np.abs(on[0] - off[0]).max())
for inj in (1, 2, 3, 5, 8, 13, 21):
    before = h.profile_get("lookahead_retries")["launches"]
    h.set_option("la_fault_inject", inj)
    try:
        got = h.conditional(prog, Z, Xnew, f, 1e-6, white=True)
        msg = "== off: %s  maxdiff %.3e" % (np.array_equal(got[0], off[0]), np.abs(got[0] - off[0]).max())
    except Exception as e:
        msg = "EXC " + str(e)[:100]
    h.set_option("la_fault_inject", 0)
    print("inject %d: retries +%d  %s" % (inj, h.profile_get("lookahead_retries")["launches"] - before, msg), flush=True)
again = h.conditional(prog, Z, Xnew, f, 1e-6, white=True)
print("afterwards == on:", np.array_equal(again[0], on[0]))
