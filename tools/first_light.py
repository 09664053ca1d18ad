"""Scratch timing driver for the GPU box (not part of the product or the tests)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc

h = gpf.get_handle()
print(h.device_info())
for w in (1, 2):
    print("mfma f64 TFLOP/s, waves/SIMD", w, h.diag_mfma_f64(w))
sizes = [int(s) for s in (sys.argv[1:] or ["2048", "8192", "32768"])]
for n in sizes:
    d = 8
    X, Y, Xs = orc.synthetic_gpr_data(n, d, 1024)
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    for rep in range(3):
        t0 = time.perf_counter(); lml = m.compute_log_likelihood(); t1 = time.perf_counter()
        st = h.last_stage_ms()
        print("N=%d lml=%.10g wall=%.1f ms stages=%s  potrf TFLOP/s=%.2f" % (n, lml, 1e3 * (t1 - t0), {k: round(v, 2) for k, v in st.items()}, n ** 3 / 3 / (st["potrf"] * 1e-3) / 1e12))
    h.profile_reset(); h.profile_enable(True)
    m.compute_log_likelihood()
    h.profile_enable(False)
    for kc in ["gemm_f64", "potrf_base", "kmat", "trsv", "reduce", "other"]:
        p = h.profile_get(kc)
        if p["launches"]:
            print("   %-10s launches=%6d ms=%9.3f  TFLOP/s=%7.2f  GB/s=%8.1f" % (kc, p["launches"], p["ms"], p["flops"] / max(p["ms"], 1e-9) / 1e9, p["bytes"] / max(p["ms"], 1e-9) / 1e6))
    m.reuse_factor = False
    t0 = time.perf_counter(); mu, var = m.predict_f(Xs); t1 = time.perf_counter()
    print("   predict_f cold N*=1024: %.1f ms  stages=%s" % (1e3 * (t1 - t0), h.last_stage_ms()))
    m.reuse_factor = True
    t0 = time.perf_counter(); mu, var = m.predict_f(Xs); t1 = time.perf_counter()
    print("   predict_f warm N*=1024: %.1f ms  stages=%s" % (1e3 * (t1 - t0), h.last_stage_ms()))
