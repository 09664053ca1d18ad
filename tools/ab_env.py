"""Same-box A/B of environment settings: alternating fresh processes of tools/one_eval.py.  python tools/ab_env.py "VAR=a;VAR=b;..." N rounds evals"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sets = sys.argv[1].split(";")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
evals = int(sys.argv[4]) if len(sys.argv) > 4 else 5
res = {s: [] for s in sets}
for r in range(rounds):
    for s in sets:
        env = dict(os.environ)
        for kv in [x for x in s.split(",") if "=" in x]:
            k, v = kv.split("=", 1); env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "one_eval.py"), str(n), str(evals)], env=env, capture_output=True, text=True).stdout
        tot = [float(l.split("'total': ")[1].split("}")[0]) for l in out.splitlines() if "'total'" in l][1:]
        res[s].append(sorted(tot)[len(tot) // 2])
print("N = %d, median total ms per process: " % n + "   ".join("%s: %s" % (s or "default", " ".join("%.3f" % x for x in v)) for s, v in res.items()))
