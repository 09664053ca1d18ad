import sys; sys.path.insert(0, "gpflow-slim_amd")
import gpflowSlim as gpf
h = gpf.get_handle()
for pipe in (0, 1):
    h.set_option("gemm_pipe", pipe)
    for burst in range(8):
        ms, _ = h.diag_gemm_timeline(0, 0, 16384, 16384, 4096, reps=10, cap_blocks=1 << 15)
        print("pipe=%d burst %d: %.2f ms = %.1f TFLOP/s" % (pipe, burst, ms, 2.0 * 16384 * 16384 * 4096 / ms / 1e9))
