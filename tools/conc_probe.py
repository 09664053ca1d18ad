"""Scratch: T handles in T threads, N, sweep-step mode: are the likelihoods identical step after step / across threads?"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
from gpflowSlim import _backend as be
mode = int(sys.argv[1]); n = int(sys.argv[2]); T = int(sys.argv[3]); steps = int(sys.argv[4]) if len(sys.argv) > 4 else 25
opts = dict(kv.split("=") for kv in sys.argv[5].split(",")) if len(sys.argv) > 5 and sys.argv[5] else {}
d = 6
rng = np.random.default_rng(n)
X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
ls = np.sqrt(d) * np.linspace(0.8, 1.2, d)
prog = gpf.kernels.RBF(d, variance=1.1, lengthscales=ls, ARD=True)._program(d)
out = {}
def run(t):
    h = be.Handle(0)
    h.set_option("potrf_fused_step", mode); h.set_option("potrf_two_stage_join", 1 if mode else 0)
    for k, v in opts.items(): h.set_option(k, float(v))
    h.gpr_set_data(X, X)
    out[t] = ([h.gpr_lml(prog, 0.1, Y) for _ in range(steps)], h.profile_get("lookahead_retries")["launches"])
ths = [threading.Thread(target=run, args=(t,)) for t in range(T)]
for t in ths: t.start()
for t in ths: t.join()
vals = sorted({repr(v) for t in out for v in out[t][0]})
print("mode %d N %d threads %d %s: %d distinct values %s retries %s" % (mode, n, T, opts, len(vals), vals[:4], [out[t][1] for t in out]))
