"""Phase stamps of one potrf_base launch (gps_diag_potrf_base_stamps): python tools/pb_stamps.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
h = gpf.get_handle()
lib = gpf.load_library()
out = (ctypes.c_double * 7)()
for factor in (1, 0):
    rc = lib.gps_diag_potrf_base_stamps(h._h, factor, out)
    print("factor=%d rc=%d clock %.0f MHz | us since start: loaded %.2f eliminated %.2f L stored %.2f inv start %.2f inv done %.2f stored %.2f"
          % (factor, rc, out[0], out[1], out[2], out[3], out[4], out[5], out[6]))
