"""Scratch: which option set gives a wrong likelihood?  python tools/step_combo.py N "set;set;..." reps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf
import oracle.gp_oracle as orc
n = int(sys.argv[1]); sets = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in s.split(",") if kv) for s in sys.argv[2].split(";")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
m = gpf.models.GPR(X, Y, gpf.kernels.RBF(8, lengthscales=np.sqrt(8) * np.ones(8), ARD=True), obs_var=0.1)
h = gpf.get_handle()
h.set_option("potrf_fused_step", 0)
ref = m.compute_log_likelihood()
for rep in range(reps):
    for st in sets:
        for k, v in st.items():
            h.set_option(k, v)
        try:
            lml = m.compute_log_likelihood()
            print(rep, st, "rel err %.2e" % (abs(lml - ref) / abs(ref)), "retries", h.profile_get("lookahead_retries")["launches"], flush=True)
        except Exception as e:
            print(rep, st, "EXC", e, flush=True)
