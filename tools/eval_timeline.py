"""Timeline of the LAST evaluation in a rocprofv3 --kernel-trace csv: python tools/eval_timeline.py <kernel_trace.csv> [min_us]
Prints every kernel longer than min_us (start, duration, queue, grid, name) and the busy fraction of the evaluation."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last evaluation: from the last kmat kernel start
idx = max(i for i, r in enumerate(rows) if "kmat_prep" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("void ", "").replace("gemm_nt_f64_kernel", "gemm").replace("gemm_nt_f64_pair_kernel", "gemm_pair")
    return n[:44]
tot = 0
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1))
    if d >= min_us:
        print("%10.1f %9.1f q%-3s wg%-6d %s" % (s, d, r.get("Queue_Id", "?"), g, short(r["Kernel_Name"])))
end = max(int(r["End_Timestamp"]) for r in rows)
print("evaluation span %.1f us, %d kernels" % ((end - t0) / 1e3, len(rows)))
# per-class totals
cl = {}
for r in rows:
    k = short(r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c = cl.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += d
for k, (c, d) in sorted(cl.items(), key=lambda kv: -kv[1][1])[:14]:
    print("   %-44s %6d calls %10.1f us" % (k, c, d))
