"""Scratch: big-GEMM throughput for the library given in GPFLOWSLIM_HIP_LIB (built with another STRIP)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpflow-slim_amd"))
import gpflowSlim as gpf
h = gpf.get_handle()
for (lower, m, n, k) in ((0, 16384, 16384, 4096), (0, 16384, 16384, 8192), (1, 16384, 16384, 16384), (1, 8192, 8192, 8192)):
    best = 0.0
    for burst in range(4):
        ms, _ = h.diag_gemm_timeline(0, lower, m, n, k, reps=6, cap_blocks=1 << 15)
        fl = (m * (m + 128.0) * k) if lower else 2.0 * m * n * k
        best = max(best, fl / ms / 1e9)
    print("%s lower=%d %dx%dx%d: %.1f TFLOP/s" % (os.environ.get("GPFLOWSLIM_HIP_LIB", "default").split("/")[-2], lower, m, n, k, best), flush=True)
