"""Scratch: one LML evaluation (the factor stays resident), then warm predict_f calls (for rocprofv3 traces / latency tables).
python tools/predict_once.py [N] [N*, N*, ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ns_list = [int(a) for a in sys.argv[2:]] or [1024]
d = 8
X, Y, Xs = orc.synthetic_gpr_data(n, d, max(ns_list))
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
m.reuse_factor = True
for k, v in [a.split("=") for a in os.environ.get("GPS_OPTS", "").split(",") if a]:
    gpf.get_handle().set_option(k, float(v))
lml = m.compute_log_likelihood()
print("lml", lml, gpf.get_handle().last_stage_ms())
for ns in ns_list:
    ts = []
    for i in range(4):
        t0 = time.perf_counter(); mu, var = m.predict_f(Xs[:ns]); t1 = time.perf_counter()
        ts.append(1e3 * (t1 - t0))
    print("n_new %d: warm predict_f ms %s  stage %s  mean[0] %.12g var[0] %.12g" % (ns, " ".join("%.3f" % t for t in ts), gpf.get_handle().last_stage_ms(), mu[0, 0], var[0, 0]))
