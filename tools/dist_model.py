"""Per-panel time model of the block-column factorisation.  Measured on ONE MI355X: what rank 0 of P
ranks computes per panel at N = argv[1] (default 32768; partitioned storage: rank 0 holds 8 N^2 / P bytes, so N = 131072 fits
for every P) -- the panel factorisation (chain) and its share of the trailing update (bulk) --
with the exchange left out (other ranks' panels are garbage: kernel times do not depend on the data).  The exchange is
modelled: scatter + all-gather moves 2 S / P bytes per rank and phase over P - 1 links; assumed per-link rate below."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd")); sys.path.insert(0, ROOT)
import torch
import gpflowSlim as gpf
from gpflowSlim.distributed import HipPanelOps
import oracle.gp_oracle as orc

N, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 8
LINK_GBS = float(os.environ.get("LINK_GBS", "50"))        # assumed effective rate of one xGMI link, one direction
COLL_LAT_US = float(os.environ.get("COLL_LAT_US", "25"))   # assumed latency of one collective call
X, Y, _ = orc.synthetic_gpr_data(N, d, 0)
kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
h = gpf.get_handle()
h.gpr_set_data(X, ("model",))
prog = kern._program(d)
out = {"N": N, "link_GBs_assumed": LINK_GBS, "collective_latency_us_assumed": COLL_LAT_US, "cases": []}


def timed(f, reps=3):
    torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best


for nb in ((512, 1024) if N <= 32768 else (1024,)):
    for P in (1, 2, 4, 8):
        with HipPanelOps(h, prog, 0.1, Y, P, 0, nb, two_lanes=False) as ops:
            npan = ops.n_panels
            own = list(range(0, npan, P))
            probe = sorted(set([own[0], own[len(own) // 4], own[len(own) // 2], own[(3 * len(own)) // 4], own[-1]]))
            fac, upd = {}, {}
            for j in probe:
                fac[j] = timed(lambda: ops.panel_factor(j, j % ops.n_bufs))
            for j in sorted(set([0, npan // 4, npan // 2, (3 * npan) // 4])):
                upd[j] = timed(lambda: ops.update(j, j + 1, npan, 0))
        # linear fits in the panel's row count
        rows = lambda j: N + 128 - j * nb
        fa = np.polyfit([rows(j) for j in fac], [fac[j] for j in fac], 1)
        ua = np.polyfit([rows(j) ** 2 for j in upd], [upd[j] for j in upd], 1)
        chain = bulk = total1 = total2 = comm = 0.0
        for j in range(npan):
            S = 8.0 * (rows(j) * nb + 2 * (nb // 128) * 128 * 128 + 4)
            ex = 0.0 if P == 1 else (2.0 * S / P / (LINK_GBS * 1e9) * 1e3 + 2 * COLL_LAT_US * 1e-3)
            f = float(np.polyval(fa, rows(j)))
            u = float(max(np.polyval(ua, rows(j) ** 2), 0.0))
            chain += f + ex; bulk += u; comm += ex
            total1 += f + ex + u                 # no overlap (look-ahead 0)
            total2 += max(f + ex, u)             # chain and bulk on two lanes (look-ahead >= 2)
        case = {"nb": nb, "P": P, "n_panels": npan, "panel_factor_ms_first_last": [round(fac[probe[0]], 3), round(fac[probe[-1]], 3)],
                "update_ms_first": round(upd[0], 3), "chain_ms": round(chain, 1), "of_which_exchange_ms": round(comm, 1),
                "bulk_ms_per_rank": round(bulk, 1), "model_ms_no_overlap": round(total1, 1), "model_ms_two_lanes": round(total2, 1)}
        out["cases"].append(case)
        print(json.dumps(case), flush=True)
# the fused single-GPU evaluation of the same problem, for the "vs one GPU" column
if N <= 131072:
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    m.compute_log_likelihood()
    out["fused_one_gpu_ms"] = round(timed(m.compute_log_likelihood, 2), 1)
    print(json.dumps({"fused_one_gpu_ms": out["fused_one_gpu_ms"]}), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "dist_model_n%d.json" % N), "w"), indent=1)
