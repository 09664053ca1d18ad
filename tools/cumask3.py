"""Scratch: which CUs does a CU mask word select?  GEMM timeline (XCC_ID / HW_ID per workgroup) on masked streams."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
from gpflowSlim import _backend as be
def run(label, words):
    h = be.Handle(0)
    h.diag_set_cu_mask(words)
    ms, st = h.diag_gemm_timeline(0, 0, 2048, 2048, 512, reps=2, cap_blocks=1 << 12)
    xcc = st[:, 3] & 0xf
    cu = (st[:, 2] >> 8) & 0xf; se = (st[:, 2] >> 13) & 0x7; sh = (st[:, 2] >> 12) & 1
    per = {int(x): sorted(set(zip(se[xcc == x].tolist(), sh[xcc == x].tolist(), cu[xcc == x].tolist()))) for x in sorted(set(xcc.tolist()))}
    print(label, "-> XCDs used:", sorted(per), "| CUs per XCD:", {x: len(v) for x, v in per.items()})
    print("     ", {x: v[:4] for x, v in per.items()})
run("bits 0-7 only      ", [0x000000ff, 0, 0, 0, 0, 0, 0, 0])
run("bit 0 only         ", [0x00000001, 0, 0, 0, 0, 0, 0, 0])
run("bits 0,8,16,24 only", [0x01010101, 0, 0, 0, 0, 0, 0, 0])
run("word 1 bits 0-7    ", [0, 0x000000ff, 0, 0, 0, 0, 0, 0])
