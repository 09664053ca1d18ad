"""Scratch: small-LDS kernels (32x32-tile GEMM launches) next to a big GEMM on another stream."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
from gpflowSlim import _backend as be
hA = be.Handle(0); hB = be.Handle(0)
def small(reps, out, shape):
    t0 = time.perf_counter(); ms, _ = hB.diag_gemm_timeline(0, *shape, reps=reps, cap_blocks=1 << 12); out.append((ms, 1e3 * (time.perf_counter() - t0)))
def gemm(reps, out, shape):
    t0 = time.perf_counter(); ms, _ = hA.diag_gemm_timeline(0, *shape, reps=reps, cap_blocks=1 << 15); out.append((ms, 1e3 * (time.perf_counter() - t0)))
big = (1, 16384, 16384, 4096)
for sshape in [(0, 512, 128, 128), (0, 4096, 128, 128), (0, 2048, 512, 512)]:
    o = []; small(500, o, sshape); print("small %s alone: %.2f us/launch" % (sshape, 1e3 * o[0][0]))
    o = []; gemm(6, o, big); print("  big alone: %.3f ms/launch" % o[0][0])
    os_, og = [], []
    tg = threading.Thread(target=gemm, args=(16, og, big)); ts = threading.Thread(target=small, args=(500, os_, sshape))
    tg.start(); time.sleep(0.06); ts.start(); ts.join(); tg.join()
    print("  concurrent: small %.2f us/launch (wall %.1f ms)   big %.3f ms/launch" % (1e3 * os_[0][0], os_[0][1], og[0][0]))
