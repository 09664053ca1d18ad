import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
from gpflowSlim import _backend as be
def run(label, words):
    h = be.Handle(0)
    if words is not None:
        h.diag_set_cu_mask(words)
    for shape in [(0, 1, 16384, 16384, 4096), (0, 0, 16384, 2048, 2048)]:
        ms, st = h.diag_gemm_timeline(*shape, reps=6, cap_blocks=1 << 15)
        t128 = (shape[2] // 128) * (shape[2] // 128 + 1) / 2 if shape[1] == 1 else (shape[2] // 128) * (shape[3] // 128)
        print("%-22s %s: %.3f ms = %.1f TFLOP/s" % (label, shape[2:], ms, 2 * t128 * 128 * 128 * shape[4] / ms / 1e9))
run("no mask", None)
run("mask = all 256", [0xffffffff] * 8)
run("all but bits 0-7", [0xffffff00] + [0xffffffff] * 7)
run("all but bits 0-15", [0xffff0000] + [0xffffffff] * 7)
run("no mask again", None)
