"""Benchmark of the exact-GP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one GPR log-marginal-likelihood evaluation (K build + noise diagonal + Cholesky +
triangular solve + log-det + reductions, nothing cached between steps: the hyper-parameters change
every step) on the BASELINE.json workload: RBF(ARD) GPR, N=32768, D=8, fp64, X resident in HBM.
N>1 ranks (torch.distributed.run, one rank per GPU): every rank evaluates its own hyper-parameter
set on its own GPU (independent evaluations, no data-path collective) -> weak scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "gpflow-slim_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD public MI355X spec: 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--npoints", dest="n", type=int, default=32768)
    ap.add_argument("--dims", dest="d", type=int, default=8)
    ap.add_argument("--num-new", dest="n_new", type=int, default=1024)
    ap.add_argument("--cpu-sample-n", type=int, default=8192, help="oracle sample size for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for tests)")
    ap.add_argument("--force-device", type=int, default=-1, help="testing: put every rank on this GPU")
    ap.add_argument("--no-dist", action="store_true", help="skip the block-column distributed run (N > 1)")
    ap.add_argument("--dist-nb", type=int, default=512, help="block-column width of the distributed run")
    ap.add_argument("--dist-steps", type=int, default=2)
    ap.add_argument("--dist-timeout", type=float, default=300.0, help="watchdog (s) around the distributed run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.force_device >= 0:
        local_rank = args.force_device
    os.environ["GPFLOWSLIM_DEVICE"] = str(local_rank)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)

    import gpflowSlim as gpf

    n, d = args.n, args.d
    # synthetic inputs of SURVEY 8(d)
    rng = np.random.default_rng(20240607)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal((d, 1)) / np.sqrt(d)
    Y = np.sin(X @ w) + 0.1 * rng.standard_normal((n, 1))
    Xnew = rng.standard_normal((args.n_new, d))
    ls0 = np.sqrt(d) * np.ones(d)
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=ls0, ARD=True)
    model = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    h = gpf.get_handle()

    def set_step(i):
        # a different point of hyper-parameter space on every step and every rank
        s = 1.0 + 0.01 * ((i * world + rank) % 17)
        kern._ls.assign(ls0 * s)
        kern._variance.assign(1.0 / s)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        set_step(-1 - i)
        model.compute_log_likelihood()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        set_step(i)
        lml = model.compute_log_likelihood()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed            # whole-job evals/s

    # ---- N > 1: additionally factorise ONE N x N problem across all ranks (1-D block-cyclic columns,
    # panel broadcast over RCCL/xGMI with look-ahead).  Reported beside `value`, never instead of it.
    # Guarded by a watchdog: a hang in the collective path must not cost the benchmark line.
    dist_result = None
    printed = threading.Lock()
    state = {"done": False, "partial": None}
    if world > 1 and not args.no_dist:
        def on_timeout():
            if state["done"]:
                return
            if rank == 0 and state["partial"] is not None and printed.acquire(False):
                state["partial"]["distributed_block_column"] = {"error": "watchdog: no result within %.0f s" % args.dist_timeout}
                print(json.dumps(state["partial"]), flush=True)
            os._exit(0)
        if rank == 0:
            state["partial"] = {"metric": "GPR log-marginal-likelihood evals/sec + predict_f latency, fp64, N=%d D=%d" % (n, d),
                                "value": round(value, 4), "unit": "evals/s", "n_gpus": world, "steps": args.steps,
                                "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
                                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                                "config": {"workload": "RBF(ARD) GPR log-marginal-likelihood, N=%d D=%d R=1 fp64, X resident in HBM" % (n, d)}}
        timer = threading.Timer(args.dist_timeout, on_timeout)
        timer.daemon = True
        timer.start()
        try:
            from gpflowSlim.distributed import TorchComm, gpr_lml_distributed
            comm = TorchComm()
            kern._ls.assign(ls0); kern._variance.assign(1.0)          # identical state on every rank
            gpr_lml_distributed(model, comm, nb=args.dist_nb)          # warm-up (communicator set-up)
            sync()
            td = time.perf_counter()
            for i in range(args.dist_steps):
                lml_d = gpr_lml_distributed(model, comm, nb=args.dist_nb)
            sync()
            dt = (time.perf_counter() - td) / args.dist_steps
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            dist_result = {"ms_per_eval": round(1e3 * dt, 3), "evals_per_s": round(1.0 / dt, 4), "nb": args.dist_nb,
                           "lookahead": 1, "lml": lml_d, "scaling": "strong (one N x N factorisation over %d GPUs)" % world,
                           "speedup_vs_one_gpu_eval": round(ms_per_step / (1e3 * dt), 3),
                           "stage_ms_rank0": {k: round(v, 3) for k, v in h.last_stage_ms().items()}}
        except Exception as e:      # report, do not lose the line
            dist_result = {"error": repr(e)}
        state["done"] = True
        timer.cancel()

    out = None
    if rank == 0:
        stages = h.last_stage_ms()
        # predict_f latency (reference semantics = cold: re-factorises, models/gpr.py:119-121)
        model.reuse_factor = False
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model.predict_f(Xnew)
        torch.cuda.synchronize(); cold_ms = 1e3 * (time.perf_counter() - t1)
        model.reuse_factor = True
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model.predict_f(Xnew)
        torch.cuda.synchronize(); warm_ms = 1e3 * (time.perf_counter() - t1)

        # LML + analytic gradient (the quantity an optimiser step consumes)
        model.compute_log_likelihood_and_gradients()          # first call allocates the K^-1 work space
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model.compute_log_likelihood_and_gradients()
        torch.cuda.synchronize(); grad_ms = 1e3 * (time.perf_counter() - t1)

        roofline = None
        if not args.no_roofline:
            # one extra evaluation of the same workload with every launch bracketed by HIP events on the
            # library's stream; dominant kernel = gemm_nt_f64_kernel (fp64 MFMA trailing updates / solves)
            # (with the look-ahead off: per-launch durations of kernels that overlap on two streams do not add up to
            # wall time, and the roofline is about the kernel, not about the schedule around it)
            h.set_option("potrf_lookahead", 0)
            h.profile_reset(); h.profile_enable(True)
            set_step(args.steps)
            model.compute_log_likelihood()
            h.profile_enable(False)
            h.set_option("potrf_lookahead", 1)
            g = h.profile_get("gemm_f64")
            classes = {k: h.profile_get(k) for k in ("gemm_f64", "potrf_base", "kmat", "trsv", "reduce", "other")}
            achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "gemm_f64_hbm_bytes_per_launch.json")
            if os.path.exists(pmc) and n == 32768 and d == 8:     # measured for exactly this workload
                with open(pmc) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            peak_meas, _ = h.diag_mfma_f64(2)
            roofline = {"bound": "mfma", "kernel": "gemm_nt_f64_kernel", "achieved": round(achieved, 3),
                        "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
                        "traffic": traffic, "launches": g["launches"],
                        "avg_launch_us": round(1e3 * g["ms"] / max(g["launches"], 1), 3),
                        "flops_per_launch": g["flops"] / max(g["launches"], 1),
                        "peak_measured_issue_rate": round(peak_meas, 2),
                        "whole_eval": {"flops": n ** 3 / 3 + n * n * (2 * d + 6) + n * n,
                                       "tflops": round((n ** 3 / 3 + n * n * (2 * d + 6) + n * n) / (ms_per_step * 1e-3) / 1e12, 3),
                                       "frac": round((n ** 3 / 3 + n * n * (2 * d + 6) + n * n) / (ms_per_step * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)},
                        "per_class_ms": {k: round(v["ms"], 3) for k, v in classes.items()},
                        "note": "instrumented evaluation with the look-ahead off (launches serialised); value / ms_per_step are measured with it on"}

        cpu = None
        if not args.no_cpu_baseline and world == 1:       # the CPU stand-in is timed at N = 1 only
            import oracle.gp_oracle as orc       # the checker: cpu_baseline leg + its parity gate only
            ns = min(args.cpu_sample_n, n)
            Xc, Yc = X[:ns], Y[:ns]
            spec = {"type": "rbf", "variance": orc.constrained(1.0), "lengthscales": orc.constrained(ls0), "input_dim": d}
            ref_lml, tm = orc.gpr_lml_timed(spec, Xc, Yc, orc.constrained(0.1))
            # parity gate on the very sample that is timed
            kern._ls.assign(ls0); kern._variance.assign(1.0)
            ms = gpf.models.GPR(Xc, Yc, kern, obs_var=0.1)
            got = ms.compute_log_likelihood()
            assert abs(got - ref_lml) <= 1e-8 * abs(ref_lml), (got, ref_lml)
            try:
                import threadpoolctl
                info = threadpoolctl.threadpool_info()
                threads = max([i.get("num_threads", 1) for i in info] or [1])
                blas = ";".join(sorted({"%s %s" % (i.get("internal_api"), i.get("version")) for i in info}))
            except Exception:
                threads, blas = os.cpu_count(), "unknown"
            scale = (n / ns) ** 3
            cpu = {"value": round(1.0 / (tm["total_s"] * scale), 6), "unit": "evals/s", "cores": threads,
                   "kind": "port",
                   "sample": "oracle (numpy/scipy %s) LML at N=%d D=%d: %.2f s (kmat %.2f, dpotrf %.2f, trsv %.2f); "
                             "extrapolated to N=%d by (N/Ns)^3 = %.0fx; host has %d logical cores; stand-in for the "
                             "reference TF-CPU path, which cannot run (no TensorFlow)" % (
                                 blas, ns, d, tm["total_s"], tm["kmat_s"], tm["potrf_s"], tm["trsv_s"], n, scale,
                                 os.cpu_count()),
                   "sample_seconds": round(tm["total_s"], 3), "sample_parity_rel_err": abs(got - ref_lml) / abs(ref_lml)}

        out = {"metric": "GPR log-marginal-likelihood evals/sec + predict_f latency, fp64, N=%d D=%d" % (n, d),
               "value": round(value, 4), "unit": "evals/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "RBF(ARD) GPR log-marginal-likelihood, N=%d D=%d R=1 fp64, X resident in HBM" % (n, d),
                          "n": n, "d": d, "n_new": args.n_new,
                          "parallelism": "1 GPU" if world == 1 else "%d independent per-GPU evaluations (hyper-parameter sets), no collective" % world},
               "predict_f_latency_ms": {"cold_refactor": round(cold_ms, 2), "warm_resident_factor": round(warm_ms, 2), "n_new": args.n_new},
               "lml_plus_gradient_ms": round(grad_ms, 2),
               "stage_ms_last_step": {k: round(v, 3) for k, v in stages.items()},
               "lml_last_step": lml,
               "roofline": roofline, "cpu_baseline": cpu}
        if dist_result is not None:
            if "lml" in dist_result:
                kern._ls.assign(ls0); kern._variance.assign(1.0)
                ref1 = model.compute_log_likelihood()
                dist_result["parity_rel_err_vs_one_gpu"] = abs(dist_result["lml"] - ref1) / abs(ref1)
            out["distributed_block_column"] = dist_result
        if printed.acquire(False):
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
