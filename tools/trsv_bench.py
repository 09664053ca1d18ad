"""Scratch: forward substitution of one N-point factor, wavefront launch vs recursive, via the LML stage timer."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
for n in [int(a) for a in sys.argv[1:]] or [4096, 16384, 32768]:
    d = 8
    X, Y, _ = orc.synthetic_gpr_data(n, d, 0)
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    h.set_option("gpr_aug_rows", 0)
    out = {}
    for wave in (0, 1, 0, 1):
        h.set_option("trsv_wave", wave)
        m.compute_log_likelihood()
        t0 = time.perf_counter(); v = m.compute_log_likelihood(); t1 = time.perf_counter()
        st = h.last_stage_ms()
        out.setdefault(wave, []).append((round(st["trsv"], 3), round(1e3 * (t1 - t0), 2), v))
    t0 = time.perf_counter(); g = m.compute_log_likelihood_and_gradients() if hasattr(m, "compute_log_likelihood_and_gradients") else None; t1 = time.perf_counter()
    print(n, "recursive (trsv ms, eval ms, lml):", out[0], " wavefront:", out[1], "gbs wave %.0f" % (4.0 * n * n / (out[1][-1][0] * 1e-3) / 1e9),
          "fallbacks", h.profile_get("trsv_wave_fallbacks")["launches"], flush=True)
    h.set_option("gpr_aug_rows", -1)
