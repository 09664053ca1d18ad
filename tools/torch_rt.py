"""Scratch: does loading torch first (its bundled libamdhip64 becomes the process's HIP runtime) change the timing?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    print("torch", torch.__version__, "hip", torch.version.hip)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
for n in (8192, 32768):
    X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
    kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    ts = []
    for i in range(5):
        m.compute_log_likelihood(); ts.append(h.last_stage_ms()["potrf"])
    print("N=%d potrf ms:" % n, " ".join("%.2f" % t for t in ts))
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l})
print(libs)
