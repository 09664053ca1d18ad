import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import torch
import gpflowSlim as gpf
from gpflowSlim.distributed import SingleComm, gpr_lml_distributed
import oracle.gp_oracle as orc
n = int(sys.argv[1]); nb = int(sys.argv[2])
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
h = gpf.get_handle()
gpr_lml_distributed(m, SingleComm(), nb=nb)
h.profile_reset(); h.profile_enable(True)
gpr_lml_distributed(m, SingleComm(), nb=nb)
h.profile_enable(False)
for kc in ["gemm_f64", "potrf_base", "kmat", "trsv", "other"]:
    p = h.profile_get(kc)
    if p["launches"]:
        print("   %-10s launches=%6d ms=%9.3f  TFLOP/s=%7.2f" % (kc, p["launches"], p["ms"], p["flops"] / max(p["ms"], 1e-9) / 1e9))
