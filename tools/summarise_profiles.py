"""Turns the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_*) into the tracked summaries profiles/<tag>_* (tag = argv[1], default r06)."""
import csv, glob, hashlib, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out"); PROF = os.path.join(ROOT, "profiles")
sys.path.insert(0, ROOT)
from bench import kernel_source_sha
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = None
sha = kernel_source_sha()


def find(d, pat):
    hits = glob.glob(os.path.join(OUT, d, "**", pat), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None      # a directory may hold an earlier collection too


def last_json(path):
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


# kernel statistics
for d, name in (("prof_bench", "bench"), ("prof_cfg2", "cfg2"), ("prof_cfg4", "cfg4"), ("prof_cfg5", "cfg5"), ("prof_small", "small_n")):
    f = find(d, "*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(PROF, "%s_%s_kernel_stats.csv" % (tag, name)))
    log = os.path.join(OUT, d + ".log")
    if os.path.exists(log) and name == "small_n":
        lines = [ln for ln in open(log).read().splitlines() if ln.startswith("N=")]
        json.dump({"lines": lines, "kernel_source_sha": sha, "commit": commit,
                   "note": "tools/small_n.py under rocprofv3 --kernel-trace --stats (tracing adds a few microseconds per launch): wall time per call of the "
                           "Python API / of the C entry point, GPU stage times, at the size of the reference's own example (examples/gpr.py: N ~ 455)"},
                  open(os.path.join(PROF, "%s_%s.json" % (tag, name)), "w"), indent=1)
        continue
    if os.path.exists(log):
        j = last_json(log)
        if j:
            j["kernel_source_sha"], j["commit"] = sha, commit
            json.dump(j, open(os.path.join(PROF, "%s_%s.json" % (tag, name)), "w"), indent=1)


f = os.path.join(OUT, "bench_unprofiled.log")
if os.path.exists(f):
    j = last_json(f)
    if j:
        j["kernel_source_sha"], j["commit"] = sha, commit
        json.dump(j, open(os.path.join(PROF, "%s_bench_unprofiled.json" % tag), "w"), indent=1)

# SQ counters of the config-4 kernel-matrix kernel (collect_profiles.sh kmatpmc)
kc = {}
for f in glob.glob(os.path.join(OUT, "kpmc_*", "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if "kmat_mfma" in r["Kernel_Name"]})
    if ids:
        for r in rows:
            if int(r["Dispatch_Id"]) == ids[-1]:
                kc[r["Counter_Name"]] = float(r["Counter_Value"])
if kc.get("SQ_INSTS_VALU") and kc.get("SQ_WAVES"):
    aw = kc["SQ_WAVES"] * 0.5039           # lower triangle of the 256 x 256 tile grid (the other workgroups return at once)
    json.dump({"what": "SQ counters of ONE launch of kmat_mfma_kernel<-1> (BASELINE config 4: Matern-5/2(ARD) + Periodic, N = 16384, D = 16; lower triangle), "
                       "rocprofv3 --pmc, five passes (tools/kmat_once.py)", "counters": kc,
               "derived": {"valu_instructions_per_active_wave": round(kc["SQ_INSTS_VALU"] / aw), "valu_instructions_per_entry": round(kc["SQ_INSTS_VALU"] / aw / 16, 1),
                           "salu_instructions_per_active_wave": round(kc.get("SQ_INSTS_SALU", 0) / aw),
                           "valu_issue_cycles_per_simd_at_4_per_instruction": round(kc["SQ_INSTS_VALU"] / 1024 * 4),
                           "mfma_busy_cycles_per_simd": round(kc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024)},
               "reading": "the build is bound by the double-precision pipe: the VALU instructions of the epilogue (two in-line exponentials, sqrt, polynomial, fold: "
                          "valu_instructions_per_entry, 4 cycles each) plus the fp64 MFMA feature products (48 per wave and 16 entries, 64 cycles each: the same rate per "
                          "multiply-add as the VALU on this part) -- at 100 % of the pipe 0.52 ms = 2.07 TB/s; see docs/LAB_NOTES.md (rounds 4 and 5)",
               "kernel_source_sha": sha, "commit": commit}, open(os.path.join(PROF, "%s_kmat_cfg4_counters.json" % tag), "w"), indent=1)

for src, dst in (("soak.json", "soak.json"), ("potrf_base_stamps.txt", "potrf_base_stamps.txt"), ("predict_timeline_1024.txt", "predict_timeline_1024.txt"),
                 ("predict_wide_build.txt", "predict_wide_build.txt"), ("predict_ab.txt", "predict_ab.txt"), ("sweep_step_budget.txt", "sweep_step_budget.txt"), ("timeline_8192.txt", "timeline_8192.txt"), ("timeline_32768.txt", "timeline_32768.txt"),
                 ("trsm512.txt", "trsm512.txt"), ("trsm512_stamps.txt", "trsm512_stamps.txt"), ("kmat_ab.txt", "kmat_ab.txt")):
    f = os.path.join(OUT, src)
    if os.path.exists(f):
        if src.endswith(".json"):
            j = json.load(open(f)); j["commit"] = commit
            json.dump(j, open(os.path.join(PROF, "%s_%s" % (tag, dst)), "w"), indent=1)
        else:
            shutil.copy(f, os.path.join(PROF, "%s_%s" % (tag, dst)))


def per_kernel(d, counters):
    """Sum of each counter per kernel name over the dispatches of the LAST evaluation of the run (the earlier ones warm up)."""
    f = find(d, "*counter_collection.csv")
    if not f:
        return None
    rows = list(csv.DictReader(open(f)))
    # dispatches in order; the evaluation boundary = second occurrence of the kernel-matrix tile kernel
    disp = {}
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    starts = [i for i in ids if disp[i]["name"].startswith("kmat_prep") or "kmat_prep_kernel" in disp[i]["name"]]
    begin = starts[-1] if starts else ids[0]
    acc = {}
    for i in ids:
        if i < begin:
            continue
        nm = re.sub(r"\(.*", "", disp[i]["name"])
        a = acc.setdefault(nm, {"launches": 0})
        a["launches"] += 1
        for c in counters:
            a[c] = a.get(c, 0.0) + disp[i].get(c, 0.0)
    return acc


fetch, write = per_kernel("prof_pmc_FETCH_SIZE", ["FETCH_SIZE"]), per_kernel("prof_pmc_WRITE_SIZE", ["WRITE_SIZE"])
if fetch and write:
    # the launch set of bench.py's roofline object: the library's KC_GEMM class = every gemm_nt_f64_kernel<*> launch and the
    # one-launch 512-column solves (trsm_panel_kernel<*> / trsm_panel_persistent_kernel<*>, trsm_panel.hip), so that traffic x launches = the total below
    in_class = lambda k: ("gemm_nt_f64_kernel" in k) or ("gemm_nt_f64_pair_kernel" in k) or ("trsm_panel_" in k)
    gem = lambda acc, c: sum(v.get(c, 0.0) for k, v in acc.items() if in_class(k))
    launches = sum(v["launches"] for k, v in fetch.items() if in_class(k))
    fb, wb = gem(fetch, "FETCH_SIZE") * 1024.0, gem(write, "WRITE_SIZE") * 1024.0          # rocprofv3 reports KB
    rec = {"command": "GPS_LOOKAHEAD=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/one_eval.py 32768 1 (last evaluation only)",
           "kernel_class": "gemm_nt_f64_kernel<*> + trsm_panel_kernel<*> / trsm_panel_persistent_kernel<*> (the library's KC_GEMM class: what bench.py's roofline.launches counts)", "launches": launches,
           "FETCH_SIZE_bytes_raw": fb, "WRITE_SIZE_bytes": wb,
           "gfx950_correction": "FETCH_SIZE reports 1/2 of the bytes of 16-byte-per-lane streaming reads (MI355X_MICROARCH.md HBM section): fetch doubled",
           "hbm_bytes_total": 2.0 * fb + wb, "hbm_bytes_per_launch": (2.0 * fb + wb) / max(launches, 1),
           "kernel_source_sha": sha, "commit": commit,
           "note": "FETCH_SIZE sits on the L2's fabric side and includes Infinity-Cache hits: L2-miss traffic, not DRAM traffic",
           "per_kernel": {k: {"launches": v["launches"], "FETCH_SIZE_KB": v.get("FETCH_SIZE"), "WRITE_SIZE_KB": write.get(k, {}).get("WRITE_SIZE")} for k, v in fetch.items()}}
    json.dump(rec, open(os.path.join(PROF, "%s_gemm_f64_hbm_bytes_per_launch.json" % tag), "w"), indent=1)
    print("hbm bytes per GEMM launch: %.3e (total %.1f GB over %d launches)" % (rec["hbm_bytes_per_launch"], rec["hbm_bytes_total"] / 1e9, launches))
mf = per_kernel("prof_pmc_mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
if mf:
    tot_b = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for v in mf.values()); tot_a = sum(v.get("GRBM_GUI_ACTIVE", 0) for v in mf.values())
    gb = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for k, v in mf.items() if "gemm_nt_f64_" in k); ga = sum(v.get("GRBM_GUI_ACTIVE", 0) for k, v in mf.items() if "gemm_nt_f64_" in k)
    # the CSV sums GRBM_GUI_ACTIVE over its 8 XCD instances: per-XCD cycles = value / 8
    rec = {"definition": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), last evaluation of tools/one_eval.py 32768 1 (GPS_LOOKAHEAD=0: counter collection serialises the dispatches)",
           "whole_evaluation": tot_b / (tot_a / 8.0 * 1024.0) if tot_a else None, "gemm_class": gb / (ga / 8.0 * 1024.0) if ga else None,
           "mfma_f64_flops_from_busy_cycles": tot_b / 64.0 * 2048.0,
           "kernel_source_sha": sha, "commit": commit,
           "per_kernel": {k: {"launches": v["launches"], "mfma_busy": (v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)) if v.get("GRBM_GUI_ACTIVE") else None,
                              "GRBM_GUI_ACTIVE": v.get("GRBM_GUI_ACTIVE")} for k, v in mf.items()}}
    json.dump(rec, open(os.path.join(PROF, "%s_pmc_mfma_summary.json" % tag), "w"), indent=1)
    print("MFMA busy: whole evaluation %.3f, GEMM class %.3f" % (rec["whole_evaluation"] or 0, rec["gemm_class"] or 0))
