"""Scratch: where do workgroups land under a CU mask, and does a masked side stream run next to a masked GEMM?"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
from gpflowSlim import _backend as be
import oracle.gp_oracle as orc
def where(h, label):
    ms, st = h.diag_gemm_timeline(0, 0, 8192, 128, 128, reps=2, cap_blocks=1 << 12)     # 1024 small workgroups
    xcc = st[:, 3] & 0xf; cu = (st[:, 2] >> 8) & 0xf; sh = (st[:, 2] >> 12) & 1; se = (st[:, 2] >> 13) & 0x7
    ids = sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
    per = {x: sum(1 for i in ids if i[0] == x) for x in range(8)}
    print(label, "-> %d distinct CUs; per XCC %s; %.1f us" % (len(ids), per, ms * 1e3))
    return ids
hA = be.Handle(0); hB = be.Handle(0)
where(hA, "no mask")
hB.diag_set_cu_mask([0xff] + [0] * 7)
where(hB, "mask bits 0-7")
hC = be.Handle(0); hC.diag_set_cu_mask([0x01010101, 0x01010101] + [0] * 6)
where(hC, "mask bits 0,8,16,...,56")
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
# complementary masks: B gets bits 0..7, A everything else
hA.diag_set_cu_mask([0xffffff00] + [0xffffffff] * 7)
where(hA, "A: all but bits 0-7")
n = 2048
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
prog = kern._program(8)
hB.gpr_set_data(X, object())
def chain(reps, out):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); hB.gpr_lml(prog, 0.1, Y); ts.append(1e3 * (time.perf_counter() - t0))
    out.append(ts)
def gemm(reps, out):
    ms, _ = hA.diag_gemm_timeline(0, 1, 16384, 16384, 4096, reps=reps, cap_blocks=1 << 15); out.append(ms)
o = []; chain(5, o); print("chain alone on 8 CUs (N=%d) ms:" % n, np.round(o[0], 2).tolist())
o = []; gemm(6, o); print("gemm alone on 248 CUs: %.3f ms/launch" % o[0])
oc, og = [], []
tg = threading.Thread(target=gemm, args=(20, og)); tc = threading.Thread(target=chain, args=(10, oc))
tg.start(); time.sleep(0.05); tc.start(); tc.join(); tg.join()
print("concurrent: chain ms:", np.round(oc[0], 2).tolist(), " gemm %.3f ms/launch" % og[0])
