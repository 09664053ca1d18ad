import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
import gpflowSlim as gpf
h = gpf.get_handle()
for w in (1, 2):
    print("waves/SIMD", w, h.diag_mfma_f64(w))
