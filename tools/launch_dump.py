"""Scratch: per-launch list of one profiled LML evaluation (GPS_PROF_DUMP), binned by GEMM shape."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
path = os.path.join(ROOT, "gpurun_out", "launches_%d.txt" % n)
if os.path.exists(path):
    os.remove(path)
os.environ["GPS_PROF_DUMP"] = path
import gpflowSlim as gpf
import oracle.gp_oracle as orc
h = gpf.get_handle()
GRAD = "grad" in sys.argv[2:]
for kv in sys.argv[2:]:
    if "=" in kv:
        k, v = kv.split("="); h.set_option(k, float(v))
X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
run = m.compute_log_likelihood_and_gradients if GRAD else m.compute_log_likelihood
run(); run()
h.profile_reset(); h.profile_enable(True)
run()
h.profile_enable(False)
h.profile_get("gemm_f64")
bins = collections.OrderedDict()
tot = 0.0
for line in open(path):
    f = line.split()
    us = float(f[5]); tot += us
    if f[0] != "gemm_f64":
        key = (f[0],)
        flop = 0.0
    else:
        M, N, K, fl = int(f[1]), int(f[2]), int(f[3]), int(f[4])
        lower = (fl // 10) % 2
        tri = (fl // 20) % 5 == 1 and fl < 100
        key = ("gemm", "rowpanel" if fl >= 100 else ("lower" if lower else ("triA" if tri else "full")), "op%d" % (fl % 10), "K=%d" % K, "M<=%d" % (1 << int(np.ceil(np.log2(M)))))
        flop = (M * (M + 128) * K) if lower else (128.0 * 128.0 * (N // 128) * (M // 128) * (M + 128) if tri else 2.0 * M * N * K)
    b = bins.setdefault(key, [0, 0.0, 0.0]); b[0] += 1; b[1] += us; b[2] += flop
print("total us", tot)
for k, b in sorted(bins.items(), key=lambda kv: -kv[1][1]):
    ideal = b[2] / 70e12 * 1e6
    print("%-50s n=%5d  %9.1f us  avg %7.1f  TF %5.1f  over-ideal %8.1f us" % (" ".join(k), b[0], b[1], b[1] / b[0], b[2] / max(b[1], 1e-9) / 1e6, b[1] - ideal))
