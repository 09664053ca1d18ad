"""Phase stamps of the one-launch 512-column solve (trsm_panel.hip): python tools/tp_stamps.py [m] [backward]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import gpflowSlim as gpf
h = gpf.get_handle()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
back = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 64
h.set_option("trsm_panel_rows", rows)
us, st = h.diag_trsm512_stamps(m, back)
st = st[st[:, 0] != 0]
if len(st) == 0:
    print("rows option %d: no stamps recorded (%.1f us per solve)" % (rows, us)); sys.exit(0)
t = st.astype(np.float64) / 100.0            # microseconds
t0 = t[:, 0].min()
span = max(t[:, 13].max(), t[:, 29].max()) - t0
print("rows option %d" % rows)
print("m=%d %s: %.1f us per solve (%.1f TFLOP/s counted as 10 products); stamped launch spans %.1f us, %d workgroups"
      % (m, "backward" if back else "forward", us, 10 * 2 * m * 128 * 128 / us / 1e6, span, len(t)))
names = ["start", "rows loaded"] + ["%s j=%d" % (n, j) for j in range(4) for n in ("inverse product", "rows stored", "updates done")]
for w, off in ((0, 0), (7, 16)):
    d = np.diff(t[:, off:off + 14], axis=1)
    print("wave %d: mean us per phase (10 / 50 / 90 %% quantiles)" % w)
    for q in range(13):
        print("   %-22s %7.2f   (%6.2f %6.2f %6.2f)" % (names[q + 1], d[:, q].mean(), *np.quantile(d[:, q], [0.1, 0.5, 0.9])))
    print("   %-22s %7.2f" % ("workgroup total", (t[:, off + 13] - t[:, off]).mean()))
# per CU: the gap between the end of one workgroup and the start of the next one on the same CU
hw, xcc = st[:, 14], st[:, 15]
cu = ((xcc & 0xF) << 16) | (hw & 0xFFF0 & ~0x30)          # HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
gaps, per_cu = [], {}
for b in range(len(t)):
    per_cu.setdefault(int(cu[b]), []).append((t[b, 0], max(t[b, 13], t[b, 29])))
busy = []
for k, v in per_cu.items():
    v.sort()
    for a, b in zip(v[:-1], v[1:]):
        gaps.append(b[0] - a[1])
    busy.append(sum(e - s for s, e in v))
gaps = np.array(gaps)
if rows == 32:
    print("(32 rows per workgroup: two workgroups share a CU, a negative gap is their overlap)")
print("CUs seen: %d, row blocks per CU %.1f; start of a row block minus end of the one before on the same CU: mean %.2f us (10 / 50 / 90 %%: %.2f %.2f %.2f); "
      "CU busy %.1f %% of the span" % (len(per_cu), len(t) / len(per_cu), gaps.mean() if len(gaps) else 0, *(np.quantile(gaps, [0.1, 0.5, 0.9]) if len(gaps) else (0, 0, 0)),
                                       100 * np.mean(busy) / span))
