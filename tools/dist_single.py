"""Scratch: right-looking block-column path on ONE GPU (P = 1) for several panel widths."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import torch
import gpflowSlim as gpf
from gpflowSlim.distributed import SingleComm, gpr_lml_distributed
import oracle.gp_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d = 8
X, Y, _ = orc.synthetic_gpr_data(n, d, 0)
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
ref = m.compute_log_likelihood()
print("fused", ref, gpf.get_handle().last_stage_ms())
for nb in [256, 512, 1024, 2048]:
    for la in (0, 1, 2, 3):
        gpr_lml_distributed(m, SingleComm(), nb=nb, lookahead=la)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v = gpr_lml_distributed(m, SingleComm(), nb=nb, lookahead=la)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("nb=%d lookahead=%d: %.1f ms rel err %.1e stages %s" % (nb, la, 1e3 * (t1 - t0), abs(v - ref) / abs(ref), {k: round(x, 1) for k, x in gpf.get_handle().last_stage_ms().items()}))
