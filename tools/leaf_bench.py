"""Scratch: microseconds per launch of one 128-column solve leaf (plain product with the block inverse vs refined)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
h = gpf.get_handle()
for m in (128, 1024, 2048, 3968, 4096, 8064, 8192, 16384, 32768, 131072, 524288, 1048576):
    reps = 200 if m <= 32768 else 20
    p_us, p_res = h.diag_trsm_leaf(m, 0, False, reps)
    r_us, r_res = h.diag_trsm_leaf(m, 1, False, reps)
    u_us, u_res = h.diag_trsm_leaf(m, 1, True, reps)
    fl = 2.0 * m * 128 * 128
    print("m=%8d  plain %8.1f us (%.1f TF/s, resid %.1e) | refined %8.1f us (x%.2f, resid %.1e) | refined upper %8.1f us (resid %.1e)"
          % (m, p_us, fl / p_us / 1e6, p_res, r_us, r_us / p_us, r_res, u_us, u_res), flush=True)
