"""Merge the outputs of tools/power_vs_traffic.sh into profiles/<tag>_power_vs_traffic.json."""
import csv, glob, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
strip_of = {"lib": 8, "lib_strip4": 4, "lib_strip16": 16, "lib_strip32": 32, "lib_noxcd": 8}
note_of = {"lib_noxcd": "XCD-aware workgroup -> tile remap switched off (-DGPS_GEMM_NO_XCD_REMAP): neighbouring tiles land on different XCDs"}
rows = []
for f in sorted(glob.glob(os.path.join(OUT, "pvt_power_*.json"))):
    v = os.path.basename(f)[len("pvt_power_"):-5]
    try:
        p = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    rec = {"variant": v, "STRIP": strip_of.get(v), **({"note": note_of[v]} if v in note_of else {}), **{k: p[k] for k in p if k not in ("lib", "sensors")}}
    cs = glob.glob(os.path.join(OUT, "pvt_fetch_" + v, "**", "*counter_collection.csv"), recursive=True)
    if cs:
        disp = {}
        for r in csv.DictReader(open(cs[0])):
            if r["Counter_Name"] == "FETCH_SIZE":
                disp[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
        ids = sorted(disp)
        starts = [i for i in ids if "kmat_prep" in disp[i][0]]
        begin = starts[-1] if starts else ids[0]
        kb = sum(val for i, (nm, val) in disp.items() if i >= begin)
        kb_gemm = sum(val for i, (nm, val) in disp.items() if i >= begin and "gemm_nt_f64_kernel" in nm)
        # rocprofv3 reports KB; gfx950: FETCH_SIZE counts half of the bytes of 16-byte-per-lane streaming reads (MI355X_MICROARCH.md)
        rec["fetch_gb_one_evaluation"] = round(2.0 * kb * 1024.0 / 1e9, 1)
        rec["fetch_gb_gemm_kernels"] = round(2.0 * kb_gemm * 1024.0 / 1e9, 1)
    rows.append(rec)
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
out = {"what": "VERDICT r03 item 6: does the L2-miss (fabric) traffic of the factorisation cost clock under the power cap?  STRIP = tile columns "
               "per strip of the GEMM's workgroup -> tile order (gemm_f64.hip::decode_tile); everything else identical.",
       "method": "tools/power_vs_traffic.sh: hwmon power1_average / freq1_input sampled every ~10 ms by a thread during 12 back-to-back N = 32768 "
                 "evaluations (tools/power_trace.py); FETCH_SIZE of one evaluation by rocprofv3 --pmc (GPS_LOOKAHEAD=0, separate run)",
       "commit": commit, "variants": rows,
       "conclusion": "No clock effect.  The card draws ~1.25 kW of its 1.4 kW cap during the factorisation and the shader clock stays within 0.8 % "
                     "(2.36-2.39 GHz) for every tile order; L2-miss traffic between 380 and 429 GB per evaluation (XCD remap off: +11 %) moves the "
                     "evaluation time by less than the run-to-run spread (0.15 %) -- STRIP = 16 loses 0.9 % through its tile order, not through "
                     "clock.  The factorisation is bound by MFMA issue, not by fabric traffic or power: item closed, STRIP stays 8."}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_power_vs_traffic.json" % tag), "w"), indent=1)
for r in rows:
    print(r)
