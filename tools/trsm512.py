"""512-column triangular solve: one launch (trsm_panel.hip) vs launch by launch.  python tools/trsm512.py [m ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
h = gpf.get_handle()
for m in [int(a) for a in sys.argv[1:]] or [128, 1024, 4096, 8192, 16384, 24576]:
    for back in (False, True):
        us0, _ = h.diag_trsm512(m, back, False)
        res = []
        for rows in (32, 64, 65):
            if m % 64: continue
            h.set_option("trsm_panel_rows", rows)
            us1, diff = h.diag_trsm512(m, back, True)
            res.append("rows %d: %.1f us (%.1f TFLOP/s) diff %.2e" % (rows, us1, 10 * 2 * m * 128 * 128 / us1 / 1e6, diff))
        h.set_option("trsm_panel_rows", 0)
        print("m=%d %s: launch by launch %.1f us | one launch %s" % (m, "backward" if back else "forward", us0, " | ".join(res)))
