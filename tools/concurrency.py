"""Scratch: does a latency-bound potrf chain on one stream make progress while a big GEMM fills the GPU from another?"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
from gpflowSlim import _backend as be
import oracle.gp_oracle as orc

hA = be.Handle(0); hB = be.Handle(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
prog = kern._program(False, 8) if hasattr(kern, "_program") else be.make_program(kern._nodes(False, 8))
hB.gpr_set_data(X, object())

def chain(reps, out):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); hB.gpr_lml(prog, 0.1, Y); ts.append(1e3 * (time.perf_counter() - t0))
    out.append(ts)

def gemm(reps, out, shape):
    t0 = time.perf_counter(); ms, _ = hA.diag_gemm_timeline(0, shape[0], shape[1], shape[2], shape[3], reps=reps, cap_blocks=1 << 15); out.append((ms, 1e3 * (time.perf_counter() - t0)))

for shape in [(1, 16384, 16384, 4096), (0, 16384, 4096, 1024)]:
    o = []; chain(6, o); print("chain alone (N=%d) ms:" % n, np.round(o[0], 2).tolist())
    o = []; gemm(8, o, shape); print("gemm alone %s: %.3f ms/launch (wall %.1f)" % (shape, o[0][0], o[0][1]))
    oc, og = [], []
    tg = threading.Thread(target=gemm, args=(24, og, shape)); tc = threading.Thread(target=chain, args=(12, oc))
    tg.start(); time.sleep(0.05); tc.start(); tc.join(); tg.join()
    print("concurrent: chain ms:", np.round(oc[0], 2).tolist(), " gemm %.3f ms/launch (wall %.1f)" % og[0])
