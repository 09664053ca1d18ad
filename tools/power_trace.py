"""Power / shader clock of the GPU, sampled every ~10 ms from the hwmon files of the card, during repeated N = 32768
evaluations with the library named by GPFLOWSLIM_HIP_LIB.  python tools/power_trace.py [N] [evals] -> one JSON line.
(does the fabric traffic of the factorisation cost clock under the power cap?)"""
import glob, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np


def find_sensors():
    """[(power file, sclk file)] of the card this process computes on.  The host has eight cards (others may be busy with
    other people's work): ours is found by the PCI bus id of HIP device 0, asked of the HIP runtime the library has ALREADY
    loaded (a second copy of the runtime beside it breaks device creation) -- call after the handle exists."""
    import ctypes
    pci = None
    try:
        path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
        hip = ctypes.CDLL(path)
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            pci = buf.value.decode().lower()
    except Exception:
        pass
    out = []
    for hw in sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % pci)) if pci else []:
        pw = [os.path.join(hw, nm) for nm in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, nm))]
        fq = os.path.join(hw, "freq1_input")
        if pw and os.path.exists(fq):
            out.append((pw[0], fq))
    return out, pci


def read_num(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except Exception:
        return float("nan")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    evals = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    import gpflowSlim as gpf
    import oracle.gp_oracle as orc
    X, Y, _ = orc.synthetic_gpr_data(n, 8, 0)
    kern = gpf.kernels.RBF(8, variance=1.0, lengthscales=np.sqrt(8) * np.ones(8), ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    m.compute_log_likelihood(); m.compute_log_likelihood()
    sensors, pci = find_sensors()
    if not sensors:
        print(json.dumps({"error": "no hwmon sensors for the card", "pci": pci})); return
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), [read_num(p) for p, _ in sensors], [read_num(c) for _, c in sensors]))
            time.sleep(0.01)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(0.3)                                   # idle baseline
    t_busy0 = time.perf_counter()
    ms = []
    for i in range(evals):
        t0 = time.perf_counter(); m.compute_log_likelihood(); ms.append(1e3 * (time.perf_counter() - t0))
    t_busy1 = time.perf_counter()
    time.sleep(0.2)
    stop.set(); th.join()
    h = gpf.get_handle()
    st = h.last_stage_ms()
    busy = [(p, c) for (t, p, c) in samples if t_busy0 + 0.3 <= t <= t_busy1]       # (skip the ramp of the first evaluation)
    idle = [(p, c) for (t, p, c) in samples if t < t_busy0]
    pw = np.array([p for p, _ in busy]); ck = np.array([c for _, c in busy])           # [samples][cards]
    pw_idle = np.array([p for p, _ in idle]); ck_idle = np.array([c for _, c in idle])
    card = 0
    out = {"lib": os.environ.get("GPFLOWSLIM_HIP_LIB", "default"), "n": n, "evals": evals,
           "ms_per_eval_median": round(float(np.median(ms)), 3), "ms_per_eval_min": round(float(np.min(ms)), 3),
           "potrf_ms_last": round(st["potrf"], 3), "pci": pci, "sensor": sensors[card][0], "samples_busy": len(busy),
           "sample_period_ms": round(1e3 * (samples[-1][0] - samples[0][0]) / max(1, len(samples) - 1), 2),
           "power_w_busy_mean": round(float(np.nanmean(pw[:, card])) / 1e6, 1), "power_w_busy_max": round(float(np.nanmax(pw[:, card])) / 1e6, 1),
           "power_w_busy_p10": round(float(np.nanpercentile(pw[:, card], 10)) / 1e6, 1),
           "sclk_mhz_busy_mean": round(float(np.nanmean(ck[:, card])) / 1e6, 1), "sclk_mhz_busy_min": round(float(np.nanmin(ck[:, card])) / 1e6, 1),
           "sclk_mhz_busy_max": round(float(np.nanmax(ck[:, card])) / 1e6, 1),
           "power_cap_w": read_num(os.path.join(os.path.dirname(sensors[card][0]), "power1_cap")) / 1e6,
           "power_w_idle_mean": round(float(np.nanmean(pw_idle[:, card])) / 1e6, 1), "sclk_mhz_idle_mean": round(float(np.nanmean(ck_idle[:, card])) / 1e6, 1)}
    print(json.dumps(out))


main()
