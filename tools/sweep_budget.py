"""Per-step budget of the right-looking sweep from a rocprofv3 --kernel-trace csv of tools/one_eval.py (LAST evaluation):
python tools/sweep_budget.py <kernel_trace.csv>
For every potrf_base launch: when it started, how long it ran, and what filled the time until the NEXT potrf_base started --
the kernels of the chain's own queue (panel solve, next-block-column update, join waits), the gaps between them (launch
latency / drain), and what the other queues (remainder updates, follower solve) were doing meanwhile."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "kmat_prep" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
K = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"),
      re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("gemm_nt_f64_kernel", "gemm").replace("gemm_nt_f64_pair_kernel", "gemm_pair"))) for r in rows]
pb = [k for k in K if k[3].startswith("potrf_base")]
chain_q = pb[0][2]
print("%d potrf_base launches on queue %s; evaluation span %.1f us" % (len(pb), chain_q, max(k[1] for k in K)))
print("%9s %6s | %-34s | %6s %6s | %s" % ("start", "pb us", "chain queue until the next potrf_base", "busy", "gaps", "other queues busy (us) in the step"))
tot = {"pb": 0.0, "busy": 0.0, "gap": 0.0, "step": 0.0, "n": 0}
for a, b in zip(pb[:-1], pb[1:]):
    step = b[0] - a[0]
    if step > 400:            # (a big GEMM of the recursion lies between two sweeps: not a step of a sweep)
        continue
    mid = [k for k in K if k[2] == chain_q and a[1] <= k[0] < b[0]]
    busy = sum(k[1] - k[0] for k in mid)
    gaps = step - (a[1] - a[0]) - busy
    other = {}
    for k in K:
        if k[2] != chain_q and k[0] < b[0] and k[1] > a[0]:
            other[k[2]] = other.get(k[2], 0.0) + min(k[1], b[0]) - max(k[0], a[0])
    names = " ".join("%s:%.1f" % (re.sub(r"<.*", "", k[3])[:14] + ("<%s>" % k[3].split("<")[1].split(",")[0] if "<" in k[3] else ""), k[1] - k[0]) for k in mid)
    tot["pb"] += a[1] - a[0]; tot["busy"] += busy; tot["gap"] += gaps; tot["step"] += step; tot["n"] += 1
    if tot["n"] <= 40 or tot["n"] % 8 == 0:
        print("%9.1f %6.1f | %-34s | %6.1f %6.1f | %s" % (a[0], a[1] - a[0], names[:34], busy, gaps, " ".join("q%s:%.0f" % (q, v) for q, v in sorted(other.items()))))
n = max(tot["n"], 1)
print("mean over %d steps: step %.1f us = potrf_base %.1f + other chain kernels %.1f + gaps (launch latency, drain, waits) %.1f" % (
    n, tot["step"] / n, tot["pb"] / n, tot["busy"] / n, tot["gap"] / n))
