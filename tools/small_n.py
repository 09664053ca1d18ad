"""Where a small-N evaluation's time goes (the reference's own workload: examples/gpr.py, N ~ 455): Python API vs the C call
vs GPU time.  python tools/small_n.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np
import gpflowSlim as gpf

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 2048]
    rng = np.random.default_rng(0)
    h = gpf.get_handle()
    for n in sizes:
        d = 8
        X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
        kern = gpf.kernels.RBF(d, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
        m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
        m.compute_log_likelihood()
        prog = kern._program(d)
        reps = 300
        t0 = time.perf_counter()
        for i in range(reps):
            m.compute_log_likelihood()
        t_api = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for i in range(reps):
            h.gpr_lml(prog, 0.1, Y)
        t_c = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for i in range(reps):
            kern._program(d)
        t_prog = (time.perf_counter() - t0) / reps
        st = h.last_stage_ms()
        t0 = time.perf_counter()
        for i in range(reps):
            h.gpr_lml_grad(prog, 0.1, Y)
        t_g = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for i in range(reps):
            m.compute_log_likelihood_and_gradients()
        t_gapi = (time.perf_counter() - t0) / reps
        print("N=%d: API lml %.0f us | C call %.0f us | program build %.0f us | GPU stages (ms) %s | C lml_grad %.0f us | API lml+grad %.0f us"
              % (n, 1e6 * t_api, 1e6 * t_c, 1e6 * t_prog, {k: round(v, 3) for k, v in st.items()}, 1e6 * t_g, 1e6 * t_gapi))

main()
