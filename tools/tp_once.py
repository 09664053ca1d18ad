"""A few launches of the one-launch 512-column solve (for rocprofv3 --pmc): python tools/tp_once.py m rows"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
h = gpf.get_handle()
m, rows = int(sys.argv[1]), int(sys.argv[2])
h.set_option("trsm_panel_rows", rows)
us, diff = h.diag_trsm512(m, False, True, 3)
print("m=%d rows=%d: %.1f us diff %.2e" % (m, rows, us, diff))
