import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
import gpflowSlim as gpf
h = gpf.get_handle()
lib = h._lib
lib.gps_diag_potrf_base_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
for factor in (1, 0):
    out = np.zeros(7)
    rc = lib.gps_diag_potrf_base_stamps(h._h, factor, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    print("factor", factor, rc, "us at: start, loaded, eliminated, L stored, lvl0 done, levels done, end:", np.round(out, 2))
