"""Scratch: alternating A/B of several option SETS ("k=v,k=v;k=v,..."), potrf stage time per evaluation."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
sets = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in s.split(",") if kv) for s in sys.argv[1].split(";")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
STAGE = sys.argv[4] if len(sys.argv) > 4 else "potrf"
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
ref = m.compute_log_likelihood()
res = [[] for _ in sets]
for rep in range(reps):
    for i, st in enumerate(sets):
        for k, v in st.items():
            h.set_option(k, v)
        lml = m.compute_log_likelihood(); res[i].append(h.last_stage_ms()[STAGE])
        assert abs(lml - ref) <= 1e-9 * abs(ref), (lml, ref)
for i, st in enumerate(sets):
    print(st, " ".join("%.2f" % t for t in res[i]), "| min %.2f median %.2f" % (min(res[i]), float(np.median(res[i]))))
