"""N evaluations of the LML at one size: spread of the stage times and the look-ahead retry counter.  python tools/loop_eval.py n reps"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
import oracle.gp_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
h = gpf.get_handle()
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
m = gpf.models.GPR(X, Y, gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True), obs_var=0.1)
m.compute_log_likelihood()
r0 = h.profile_get("lookahead_retries")["launches"]
ts, vals = [], set()
for i in range(reps):
    v = m.compute_log_likelihood(); ts.append(h.last_stage_ms()["potrf"]); vals.add(v)
ts = np.array(ts)
print("N=%d: potrf ms min %.2f median %.2f max %.2f | > median + 2 %%: %d of %d | look-ahead retries %d | distinct values %d"
      % (n, ts.min(), np.median(ts), ts.max(), int((ts > 1.02 * np.median(ts)).sum()), reps, h.profile_get("lookahead_retries")["launches"] - r0, len(vals)))
print(" ".join("%.1f" % t for t in ts))
