"""Maximum-size check on one MI355X: N points in clusters that are exactly independent under the kernel
(LML(X, Y) = sum of the clusters' LMLs, each small enough for the oracle), interleaved in memory so that the
factorisation is a dense N x N one.  Prints one JSON line.  usage: big_n.py [N=131072] [cluster=2048]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import gpflowSlim as gpf
import oracle.gp_oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
nc, d = n // per, 8
rng = np.random.default_rng(n)
ls = np.sqrt(d) * np.ones(d)
Xc = [rng.standard_normal((per, d)) for _ in range(nc)]
w = rng.standard_normal((d, 1)) / np.sqrt(d)
Yc = [np.sin(x @ w) + 0.1 * rng.standard_normal((per, 1)) for x in Xc]
g = int(np.ceil(nc ** (1.0 / 3.0)))


def shift(c):
    o = np.zeros((1, d))
    o[0, 0], o[0, 1], o[0, 2] = 60.0 * ls[0] * (c % g), 60.0 * ls[1] * ((c // g) % g), 60.0 * ls[2] * (c // (g * g))
    return o


order = rng.permutation(nc * per)
X = np.concatenate([x + shift(c) for c, x in enumerate(Xc)])[order]
Y = np.concatenate(Yc)[order]
kern = gpf.kernels.RBF(d, variance=1.3, lengthscales=ls, ARD=True)
spec = {"type": "rbf", "variance": orc.constrained(1.3), "lengthscales": orc.constrained(ls), "input_dim": d}
noise = orc.constrained(0.1)
t0 = time.perf_counter()
ref = sum(orc.gpr_lml(spec, x, y, noise) for x, y in zip(Xc, Yc))
t_ref = time.perf_counter() - t0
m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
h = gpf.get_handle()
print("oracle on %d clusters of %d: %.1f s; factorising N = %d ..." % (nc, per, t_ref, n), flush=True)
t0 = time.perf_counter(); lml = m.compute_log_likelihood(); t1 = time.perf_counter()
st = h.last_stage_ms()
print("first evaluation %.2f s" % (t1 - t0), st, flush=True)
t0 = time.perf_counter(); lml2 = m.compute_log_likelihood(); t2 = time.perf_counter()
st = h.last_stage_ms()
picks = [0, nc // 3, nc - 1]
Xs = np.concatenate([rng.standard_normal((20, d)) + shift(c) for c in picks])
m.reuse_factor = True
t3 = time.perf_counter(); mu, var = m.predict_f(Xs); t4 = time.perf_counter()
perr = 0.0
for k, c in enumerate(picks):
    rmu, rvar = orc.gpr_predict(spec, Xc[c], Yc[c], noise, Xs[20 * k:20 * (k + 1)] - shift(c))
    perr = max(perr, float(np.abs(mu[20 * k:20 * (k + 1)] - rmu).max() / np.abs(rmu).max()), float(np.abs(var[20 * k:20 * (k + 1)] - rvar).max() / np.abs(rvar).max()))
import torch
print(json.dumps({"n": n, "clusters": nc, "cluster_points": per, "K_bytes": 8.0 * (n + 128) * n,
                  "lml": lml, "oracle_sum_of_cluster_lmls": ref, "rel_err": abs(lml - ref) / abs(ref),
                  "repeat_identical": lml == lml2, "eval_s": round(t2 - t0, 3), "stage_ms": {k: round(v, 2) for k, v in st.items()},
                  "potrf_tflops": round(n ** 3 / 3 / (st["potrf"] * 1e-3) / 1e12, 2),
                  "predict_f_60_points_warm_s": round(t4 - t3, 3), "predict_rel_err_vs_oracle": perr,
                  "hbm_allocated_gb": round(torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9, 1)}))
