"""Same-box A/B of several builds of the library: alternating fresh processes, LML stage times at one size.
python tools/ab_lib.py <libdir,libdir,...> [N] [rounds] [evals]   (libdirs relative to gpflow-slim_amd/, e.g. lib_r05,lib)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1].split(",")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
evals = int(sys.argv[4]) if len(sys.argv) > 4 else 3
res = {v: [] for v in libs}
for r in range(rounds):
    for v in libs:
        env = dict(os.environ, GPFLOWSLIM_HIP_LIB=os.path.join(ROOT, "gpflow-slim_amd", v, "libgpflowslim_hip.so"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "one_eval.py"), str(n), str(evals)], env=env, capture_output=True, text=True).stdout
        tot = [float(l.split("'total': ")[1].split("}")[0]) for l in out.splitlines() if "'total'" in l][1:]
        km = [float(l.split("'kmat': ")[1].split(",")[0]) for l in out.splitlines() if "'kmat'" in l][1:]
        tv = [float(l.split("'trsv': ")[1].split(",")[0]) for l in out.splitlines() if "'trsv'" in l][1:]
        res[v].append(min(tot))
        print(v, ["%.3f" % t for t in tot], "kmat", ["%.3f" % t for t in km], "trsv", ["%.3f" % t for t in tv], flush=True)
print(json.dumps({k: {"best_ms": round(min(v), 3), "all": [round(x, 3) for x in v]} for k, v in res.items()}))
