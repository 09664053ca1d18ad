"""Fresh handles in fresh threads, repeated (the virtual-rank test pattern): conditional() M = 4096 on shards.
python tools/concurrent_handles2.py [rounds] [threads]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
from gpflowSlim import _backend as be

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(1000000)
d, k, m, n = 4, 2, 4096, 400000
Xnew = rng.standard_normal((n, d)); Z = rng.standard_normal((m, d)); f = rng.standard_normal((m, k))
q_sqrt = 0.3 + rng.random((m, k))
prog = gpf.kernels.RBF(d, variance=1.2, lengthscales=1.4)._program(d)
h0 = be.Handle(0)
ref = h0.conditional(prog, Z, Xnew, f, 1e-6, q_sqrt=q_sqrt, white=True)
print("reference ok", flush=True)
bad = 0
for rnd in range(rounds):
    out = {}
    def run(t):
        h = be.Handle(0)
        lo, hi = n * t // T, n * (t + 1) // T
        try:
            fm, fv = h.conditional(prog, Z, Xnew[lo:hi], f, 1e-6, q_sqrt=q_sqrt, white=True)
            out[t] = "maxdiff %.2e" % np.abs(fm - ref[0][lo:hi]).max()
        except Exception as e:
            out[t] = "EXC " + str(e)[:90]
        out[t] += " retries=%d" % h.profile_get("lookahead_retries")["launches"]
        h.close()
    ths = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    t0 = time.time()
    for t in ths: t.start()
    for t in ths: t.join()
    line = " | ".join(out[t] for t in sorted(out))
    bad += "EXC" in line
    print("round %d (%.1f s): %s" % (rnd, time.time() - t0, line), flush=True)
print("rounds with exceptions:", bad)
