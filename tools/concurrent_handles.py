"""Several library handles driven concurrently from host threads on ONE device (what the virtual-rank tests do): every thread
runs the same conditional() on its own handle; results must be bit-identical across threads and repetitions.
python tools/concurrent_handles.py [threads] [reps] [m] [n]      env GPS_LOOKAHEAD=0 switches the look-ahead off"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
from gpflowSlim import _backend as be

def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 50000
    rng = np.random.default_rng(1000000)
    d, k = 4, 2
    Xnew = rng.standard_normal((n, d)); Z = rng.standard_normal((m, d)); f = rng.standard_normal((m, k))
    kern = gpf.kernels.RBF(d, variance=1.2, lengthscales=1.4)
    prog = kern._program(d)
    h0 = be.Handle(0)
    ref = h0.conditional(prog, Z, Xnew, f, 1e-6, white=True)
    print("reference ok", float(ref[0][0, 0]), flush=True)
    out = {}
    def run(t):
        h = be.Handle(0)
        if os.environ.get("GPS_LOOKAHEAD") == "0":
            h.set_option("potrf_lookahead", 0)
        res = []
        for i in range(reps):
            try:
                fm, fv = h.conditional(prog, Z, Xnew, f, 1e-6, white=True)
                res.append("same" if (np.array_equal(fm, ref[0]) and np.array_equal(fv, ref[1])) else "DIFF %.3e" % np.abs(fm - ref[0]).max())
            except Exception as e:
                res.append("EXC " + str(e)[:80])
        out[t] = (res, h.profile_get("lookahead_retries")["launches"])
        h.close()
    ths = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    t0 = time.time()
    for t in ths: t.start()
    for t in ths: t.join()
    for t in sorted(out):
        print(t, out[t], flush=True)
    print("elapsed %.1f s" % (time.time() - t0))

main()
