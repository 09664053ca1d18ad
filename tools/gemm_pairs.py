"""Scratch: which blockIdx pairs share a CU in the first dispatch round."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
import gpflowSlim as gpf
h = gpf.get_handle()
ms, st = h.diag_gemm_timeline(0, 0, 16384, 1024, 1024, reps=2)
xcc = st[:, 3] & 0xf; cu = (st[:, 2] >> 8) & 0xf; sh = (st[:, 2] >> 12) & 1; se = (st[:, 2] >> 13) & 0x7
print("raw hwid samples:", [hex(int(x)) for x in st[:8, 2]], [hex(int(x)) for x in st[:8, 3]])
from collections import defaultdict
g = defaultdict(list)
for b in range(512):
    g[(int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b]))].append(b)
print("CUs used in round 1:", len(g))
ks = sorted(g)[:40]
for k in ks: print(k, g[k])
# second round: which block replaced which
order = np.argsort(st[:, 0])
print("start order of blocks 512..540:", order[512:540].tolist())
d = [abs(v[0] - v[1]) for v in g.values() if len(v) == 2]
print("pair index distance histogram:", np.unique(d, return_counts=True))
