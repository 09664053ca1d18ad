"""Scratch: time LML + gradient at several N."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
for n in [int(a) for a in sys.argv[1:]] or [2048, 8192, 32768]:
    d = 8
    X, Y, _ = orc.synthetic_gpr_data(n, d, 0)
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    for rep in range(2):
        t0 = time.perf_counter(); lml, g = m.compute_log_likelihood_and_gradients(); t1 = time.perf_counter()
        st = h.last_stage_ms()
        print("N=%d lml+grad %.1f ms (lml part %.1f, grad part %.1f)" % (n, 1e3 * (t1 - t0), st["total"], st["predict"]))
    print("   grads:", [np.round(np.atleast_1d(x), 4).tolist() for _, x in g])
    h.profile_reset(); h.profile_enable(True)
    m.compute_log_likelihood_and_gradients()
    h.profile_enable(False)
    for kc in ["gemm_f64", "potrf_base", "kmat", "trsv", "reduce", "other"]:
        p = h.profile_get(kc)
        if p["launches"]:
            print("   %-10s launches=%6d ms=%9.3f  TFLOP/s=%7.2f  GB/s=%8.1f" % (kc, p["launches"], p["ms"], p["flops"] / max(p["ms"], 1e-9) / 1e9, p["bytes"] / max(p["ms"], 1e-9) / 1e6))
