"""Scratch: refined 128-column leaf, resident workgroups walking the row tiles vs one workgroup per tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
h = gpf.get_handle()
for m in (16384, 32768, 131072, 524288, 1048576):
    reps = 50 if m <= 32768 else 20
    out = {}
    for rep in range(2):
        for pers in (0, 1, 2):
            h.set_option("leaf_persistent", pers)
            us, res = h.diag_trsm_leaf(m, 1, False, reps)
            uu, resu = h.diag_trsm_leaf(m, 1, True, reps)
            out.setdefault(pers, []).append((round(us, 1), round(uu, 1), float("%.1e" % res)))
    p_us, _ = h.diag_trsm_leaf(m, 0, False, reps)
    print("m=%8d plain %.1f us | per-tile workgroups %s | persistent %s | persistent 32-row %s" % (m, p_us, out[0], out[1], out[2]), flush=True)
h.set_option("leaf_persistent", 1)
