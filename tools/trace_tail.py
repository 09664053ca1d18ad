"""Every kernel of the LAST call in a rocprofv3 --kernel-trace csv, from the last launch of <marker> (default kmat_prep) on:
python tools/trace_tail.py <kernel_trace.csv> [marker] [min_us]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "kmat_prep"
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if marker in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
def short(n):
    n = n.replace("void ", "").replace("gemm_nt_f64_kernel", "gemm").replace("gemm_nt_f64_pair_kernel", "gemm_pair")
    n = re.sub(r"\(.*", "", n)
    return n[:60]
prev_end = 0.0
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1))
    if e - s >= min_us:
        print("%10.1f %9.1f gap %6.1f q%-3s wg%-6d %s" % (s, e - s, s - prev_end, r.get("Queue_Id", "?"), g, short(r["Kernel_Name"])))
    prev_end = max(prev_end, e)
print("span %.1f us, %d kernels" % (prev_end, len(rows)))
