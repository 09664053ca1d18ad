"""Kernels of one evaluation between two times (us since the evaluation's kmat): python tools/window_dump.py <kernel_trace.csv> t_lo t_hi [max_rows]
Per queue: launches, busy time; then the kernels in start order."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lo, hi = float(sys.argv[2]), float(sys.argv[3])
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 200
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "kmat_prep" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(.*", "", n); return n.replace("void ", "").replace("gemm_nt_f64_kernel", "gemm")[:40]
sel = []
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    if lo <= s < hi:
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        sel.append((s, e - s, r["Queue_Id"], g, short(r["Kernel_Name"])))
byq = {}
for s, d, q, g, k in sel:
    a = byq.setdefault((q, k), [0, 0.0]); a[0] += 1; a[1] += d
for (q, k), (c, d) in sorted(byq.items()):
    print("q%s %-42s %5d launches %10.1f us  (%.1f each)" % (q, k, c, d, d / c))
for s, d, q, g, k in sel[:cap]:
    print("%10.1f %9.1f q%-2s wg%-6d %s" % (s, d, q, g, k))
