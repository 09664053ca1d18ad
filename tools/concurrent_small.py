"""Several handles in several host threads, each running small-N likelihood + gradient evaluations (the cooperative launches of
csrc/small_n.hip need their workgroups co-resident: what happens when T of them are in flight at once?).
python tools/concurrent_small.py [threads] [steps] [N]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
from gpflowSlim import _backend as be

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
d = 8
rng = np.random.default_rng(5)
X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
prog = gpf.kernels.RBF(d, variance=1.1, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)._program(d)
h0 = be.Handle(0)
h0.gpr_set_data(X, X)
ref = h0.gpr_lml_grad(prog, 0.1, Y)
out = {}

def run(t):
    h = be.Handle(0)
    h.gpr_set_data(X, X)
    worst, slow, t_max = 0.0, 0, 0.0
    for i in range(steps):
        t0 = time.perf_counter()
        if os.environ.get("GPS_CS_LML_ONLY"):
            lml, slots = h.gpr_lml(prog, 0.1, Y), ref[1]
        else:
            lml, slots, gn, kr = h.gpr_lml_grad(prog, 0.1, Y)
        dt = time.perf_counter() - t0
        t_max = max(t_max, dt); slow += dt > 0.05
        e_l, e_g = abs(lml - ref[0]) / abs(ref[0]), np.abs(slots - ref[1]).max() / max(1.0, np.abs(ref[1]).max())
        if max(e_l, e_g) > 1e-8 or dt > 0.5:
            print("  thread %d step %d: %.3f s, lml error %.2e, gradient error %.2e, fall-backs so far %d, lml %r" % (t, i, dt, e_l, e_g, h.profile_get("small_n_fallbacks")["launches"], lml), flush=True)
        worst = max(worst, e_l, e_g)
    out[t] = (worst, slow, t_max, h.profile_get("small_n_fallbacks")["launches"])
    h.close()

ths = [threading.Thread(target=run, args=(t,)) for t in range(T)]
t0 = time.time()
for t in ths: t.start()
for t in ths: t.join()
el = time.time() - t0
print("%d threads x %d likelihood + gradient evaluations at N = %d in %.2f s (%.0f us per evaluation per thread)" % (T, steps, n, el, 1e6 * el / steps))
for t in sorted(out):
    print("  thread %d: worst relative difference %.2e, steps over 50 ms: %d, slowest step %.1f ms, small-N fall-backs %d" % ((t,) + (out[t][0], out[t][1], 1e3 * out[t][2], out[t][3])))
