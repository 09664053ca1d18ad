"""Scratch: per-workgroup timeline of one GEMM launch (start/end stamps, CU / XCD ids)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
import gpflowSlim as gpf
h = gpf.get_handle()
shapes = [(0, 0, 16384, 1024, 1024), (0, 0, 16384, 2048, 2048), (0, 0, 8192, 2048, 2048), (0, 1, 8192, 8192, 8192),
          (0, 0, 16384, 512, 512), (0, 1, 4096, 4096, 4096)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for op, lower, m, n, k in shapes:
    ms, st = h.diag_gemm_timeline(op, lower, m, n, k, reps=5)
    t128 = (m // 128) * (m // 128 + 1) / 2 if lower == 1 else (m // 128) * (n // 128)
    fl = 2 * t128 * 128 * 128 * k
    t0 = st[:, 0].min()
    s = (st[:, 0] - t0) / 100.0; e = (st[:, 1] - t0) / 100.0        # us
    dur = e - s
    xcc = st[:, 3] & 0xf
    cu = (st[:, 2] >> 8) & 0xf; se = (st[:, 2] >> 13) & 0x7; sh = (st[:, 2] >> 12) & 1
    print("shape op=%d lower=%d M=%d N=%d K=%d: %.1f us/launch = %.1f TFLOP/s; blocks=%d" % (op, lower, m, n, k, ms * 1e3, fl / ms / 1e9, len(st)))
    print("   instrumented launch: span %.1f us; wg duration min/med/mean/max = %.1f %.1f %.1f %.1f us; first start spread %.1f us"
          % (e.max(), dur.min(), np.median(dur), dur.mean(), dur.max(), np.sort(s)[min(len(s), 512) - 1]))
    # slot occupancy: integral of (#active wgs) / span
    pro = (st[:, 4] - st[:, 0]) / 100.0; loop = (st[:, 5] - st[:, 4]) / 100.0; epi = (st[:, 1] - st[:, 5]) / 100.0
    print("   prologue %.1f  K loop %.1f  epilogue %.1f us (means); epilogue min/max %.1f %.1f" % (pro.mean(), loop.mean(), epi.mean(), epi.min(), epi.max()))
    print("   mean active workgroups %.1f ; sum(dur)/span" % (dur.sum() / e.max()))
    for x in range(8):
        mk = xcc == x
        if mk.any():
            print("   xcc %d: wgs=%4d mean dur %.1f us, last end %.1f us, distinct (se,sh,cu)=%d" % (x, mk.sum(), dur[mk].mean(), e[mk].max(), len(set(zip(se[mk], sh[mk], cu[mk])))))
    # end-time histogram in tenths of the span
    hist, _ = np.histogram(e, bins=10, range=(0, e.max()))
    print("   ends per decile of the span:", hist.tolist())
    hist, _ = np.histogram(s, bins=10, range=(0, e.max()))
    print("   starts per decile of the span:", hist.tolist())
