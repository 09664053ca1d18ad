"""Benchmark of the exact-GP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one GPR log-marginal-likelihood evaluation (K build + noise diagonal + Cholesky +
triangular solve + log-det + reductions, nothing cached between steps: the hyper-parameters change
every step) on the BASELINE.json workload: RBF(ARD) GPR, N=32768, D=8, fp64, X resident in HBM.

N = 1:  the fused single-GPU path (gps_gpr_lml).
N > 1 (one rank per GPU): ONE N x N factorisation per step, partitioned over all ranks by 1-D block-cyclic
columns with the panel exchange over RCCL/xGMI (BASELINE.json configs[2]; gpflowSlim/distributed.py) ->
strong scaling: `value` = evaluations of the whole job per second.  The throughput of independent per-GPU
evaluations (no collective) is reported beside it.  Both launch forms run the same thing:

    python bench.py --gpus N ...                                        (this process starts the N ranks itself)
    python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...   (the ranks come from the launcher)

In the first form the parent never imports torch and never touches a GPU: it starts N fresh rank processes
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), relays rank 0's JSON line and exits non-zero
if any rank does.  Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "gpflow-slim_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD public MI355X spec: 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def kernel_source_sha():
    """Fingerprint of the device sources a PMC profile under profiles/ was taken with."""
    hsh = hashlib.sha1()
    src = os.path.join(ROOT, "gpflow-slim_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".hpp", ".inc")):
            with open(os.path.join(src, name), "rb") as f:
                hsh.update(name.encode()); hsh.update(f.read())
    return hsh.hexdigest()[:16]


ATTEMPTS = {
    # name -> (what the rank processes do differently, what it is)
    "rccl": "the library's own RCCL binding (csrc/comm_rccl.hip) and the in-library two-lane schedule (gps_dist_lml)",
    "torch": "torch.distributed issues the collectives (RCCL through PyTorch); exchange and panel width chosen by measurement",
    "torch-broadcast": "torch.distributed, one plain broadcast per panel, fixed panel width (the most conservative form)",
}


def plan_attempts(args):
    """Which configurations a multi-rank run tries, in order.  An explicit --comm means exactly that one."""
    if args.attempts:
        names = [a for a in args.attempts.split(",") if a]
    elif args.comm == "auto":
        names = ["rccl", "torch", "torch-broadcast"] if (args.backend == "nccl" and args.workload == "gpr") else ["torch"]
    else:
        names = [args.comm]
    for a in names:
        if a not in ATTEMPTS:
            raise SystemExit("bench.py: unknown attempt %r (known: %s)" % (a, ", ".join(ATTEMPTS)))
    return names


def _atomic_write(path, text):
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def supervise(local_ranks, world, argv, attempts, rdv_dir, first_stall_s, stall_s, clean_dir):
    """The GPU-free parent of the rank processes (it never imports torch and never touches a GPU: a process that has
    initialised the GPU must never be the one that starts or replaces GPU programs).

    `local_ranks`: the ranks this process is responsible for -- all of them when `python bench.py --gpus N` is its own
    launcher, one when a launcher (torch.distributed.run) started one bench.py per rank; the supervisors of one run then meet
    in `rdv_dir` (same node: files).  For every attempt, in order: FRESH rank processes (new rendezvous port), each watched
    through a progress file it appends to; an attempt ends for everybody as soon as one rank fails or stalls (the others are
    killed: they would sit in a collective), and only if every rank exited 0 is rank 0's JSON line relayed -- with a `launch`
    object that says which attempt produced it and why the earlier ones did not.  Returns the exit status."""
    import signal
    import socket
    import subprocess
    assert "torch" not in sys.modules, "the supervisor must not have imported torch"
    leader = 0 in local_ranks
    os.makedirs(rdv_dir, exist_ok=True)
    procs = {}

    def stop_ranks(signum, frame):
        for p in procs.values():
            if p.poll() is None:
                p.kill()                                              # exactly the processes started here
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, stop_ranks)

    def wait_for(path, seconds):
        t_end = time.time() + seconds
        while time.time() < t_end:
            v = _read(path)
            if v is not None:
                return v
            time.sleep(0.05)
        return None

    failed = []
    status = 3
    last_error_line = None
    for k, name in enumerate(attempts):
        tag = os.path.join(rdv_dir, "a%d" % k)
        if leader:
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            _atomic_write(tag + ".port", str(port))
        port = wait_for(tag + ".port", 180.0)
        if port is None:
            failed.append({"attempt": name, "reason": "no rendezvous port from rank 0's supervisor"})
            break
        procs.clear()
        t_start = time.time()
        for r in local_ranks:
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0",
                       GPS_BENCH_CHILD="1", GPS_BENCH_PROGRESS="%s.r%d.progress" % (tag, r))
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)             # the ranks bring their own store up on the fresh port
            env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
            _atomic_write(env["GPS_BENCH_PROGRESS"], "")
            # rank 0's stdout (the JSON line) goes to a file that is relayed only if the whole attempt succeeds
            out = open("%s.r0.out" % tag, "w") if r == 0 else sys.stderr
            procs[r] = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv) + ["--attempt-name", name],
                                        env=env, stdout=out)
        live = dict(procs)
        peer_failed_at = None
        why = {}
        while live:
            time.sleep(0.2)
            now = time.time()
            for r, p in list(live.items()):
                rc = p.poll()
                if rc is None:
                    # stalled?  (no line appended to its progress file for too long)
                    prog = env_prog = "%s.r%d.progress" % (tag, r)
                    try:
                        last = os.path.getmtime(prog)
                    except OSError:
                        last = t_start
                    text = _read(env_prog) or ""
                    limit = stall_s if text.strip() else first_stall_s
                    if now - max(last, t_start) > limit:
                        p.kill(); p.wait()
                        rc = 98
                        why[r] = "rank %d made no progress for %.0f s after '%s'" % (r, limit, (text.strip().splitlines() or ["start"])[-1])
                    elif peer_failed_at is not None and now - peer_failed_at > 10.0:
                        p.kill(); p.wait()
                        rc = 97
                    else:
                        continue
                elif rc != 0:
                    why.setdefault(r, "rank %d exited with status %d" % (r, rc))
                del live[r]
                _atomic_write("%s.r%d.rc" % (tag, r), "%d\n%s" % (rc, why.get(r, "")))
            if peer_failed_at is None:
                for r in range(world):
                    v = _read("%s.r%d.rc" % (tag, r))
                    if v is not None and v.split("\n")[0].strip() not in ("0", ""):
                        peer_failed_at = now
                        break
        # every rank's verdict (the other supervisors write theirs)
        rcs, reasons = [], []
        for r in range(world):
            v = wait_for("%s.r%d.rc" % (tag, r), stall_s + 60.0)
            code = int(v.split("\n")[0]) if v and v.split("\n")[0].strip().lstrip("-").isdigit() else 96
            rcs.append(code)
            if code not in (0, 97):
                reasons.append((v.split("\n", 1)[1].strip() if v and "\n" in v else "") or "rank %d: no verdict" % r)
        if all(c == 0 for c in rcs):
            status = 0
            if leader:
                lines = [ln for ln in (_read("%s.r0.out" % tag) or "").splitlines() if ln.startswith("{")]
                try:
                    rec = json.loads(lines[-1])
                    rec["launch"] = {"attempt": name, "what": ATTEMPTS[name], "index": k, "tried": attempts, "failed_attempts": failed,
                                     "supervisor": "bench.py (GPU-free parent; fresh rank processes per attempt)"}
                    print(json.dumps(rec), flush=True)
                except (IndexError, ValueError):
                    print("bench.py: rank 0 exited 0 without a JSON line", file=sys.stderr)
                    status = 3
            break
        reason = "; ".join(reasons) or "a rank failed"
        failed.append({"attempt": name, "reason": reason[:400]})
        print("bench.py: attempt '%s' failed (%s)%s" % (name, reason, "; trying the next one" if k + 1 < len(attempts) else ""),
              file=sys.stderr, flush=True)
        if leader:
            err = [ln for ln in (_read("%s.r0.out" % tag) or "").splitlines() if ln.startswith("{")]
            if err:
                last_error_line = err[-1]
                print("bench.py: rank 0 of the failed attempt said: %s" % err[-1][:600], file=sys.stderr, flush=True)
    if status != 0 and leader:
        # one JSON line even then: what rank 0 of the last attempt left (its `error` says why), or a generic one
        rec = None
        try:
            rec = json.loads(last_error_line) if last_error_line else None
        except ValueError:
            rec = None
        if not isinstance(rec, dict) or "error" not in rec:
            rec = {"metric": "GPR log-marginal-likelihood evals/sec + predict_f latency, fp64, N=32768 D=8", "value": 0.0,
                   "unit": "evals/s", "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "failed run"},
                   "error": "every attempt failed"}
        rec["value"] = 0.0
        rec["launch"] = {"tried": attempts, "failed_attempts": failed}
        print(json.dumps(rec), flush=True)
    if clean_dir:
        import shutil
        shutil.rmtree(rdv_dir, ignore_errors=True)
    return status


_PROGRESS = os.environ.get("GPS_BENCH_PROGRESS")


def progress(tag):
    """A rank process tells its supervisor that it is alive and where it is (one appended line)."""
    if _PROGRESS:
        with open(_PROGRESS, "a") as f:
            f.write("%s %.3f\n" % (tag, time.time()))


class alive:
    """Heartbeat for a phase of a rank process that can take longer than --stall without being hung and that cannot hang on
    the GPU or in a collective with work outstanding: the CPU stand-in timed on rank 0's host cores (44 s with 128 BLAS threads,
    several times that with the cores shared by 8 ranks), and the other ranks' wait for rank 0 in the last barrier (they have
    finished; if rank 0 fails or stalls its own supervisor says so and the attempt ends for everybody).  A progress mark every
    `period` seconds while inside; everything else stays under the supervisor's stall clock."""

    period = 15.0          # (main() lowers it to a third of --stall)

    def __init__(self, tag):
        import threading
        self.tag, self.stop = tag, threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self.stop.wait(self.period):
            progress(self.tag)

    def __enter__(self):
        progress(self.tag)
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.thread.join()
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--npoints", dest="n", type=int, default=32768)
    ap.add_argument("--dims", dest="d", type=int, default=8)
    ap.add_argument("--num-new", dest="n_new", type=int, default=1024)
    ap.add_argument("--num-new-throughput", dest="n_new_tp", type=int, default=8192)
    ap.add_argument("--cpu-sample-n", type=int, default=32768,
                    help="size the CPU stand-in (oracle) is timed at for cpu_baseline; default = the full workload, one repetition "
                         "(SURVEY 8d: ~75 s on the GPU box's host); smaller values are extrapolated stage by stage and flagged")
    ap.add_argument("--small-n", default="512,2048", help="sizes of the small-N latency table (the reference's own example is N ~ 455); '' = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for tests)")
    ap.add_argument("--force-device", type=int, default=-1, help="testing: put every rank on this GPU")
    ap.add_argument("--dist-nb", type=int, default=512, help="block-column width of the distributed run")
    ap.add_argument("--dist-lookahead", type=int, default=2, help="look-ahead depth of the distributed schedule")
    ap.add_argument("--dist-timeout", type=float, default=600.0, help="watchdog (s) around the distributed run")
    ap.add_argument("--no-dist-autotune", action="store_true", help="N > 1: keep --dist-nb / --dist-lookahead and the default exchange instead of choosing by measurement during warm-up")
    ap.add_argument("--independent-steps", type=int, default=2, help="N > 1: steps of the independent-evaluations side measurement (0 = skip)")
    ap.add_argument("--comm", default="auto", choices=["auto", "torch", "rccl"],
                    help="N > 1: who issues the collectives -- the library's own RCCL binding (csrc/comm_rccl.hip; torch.distributed "
                         "then only carries the 128-byte unique id at start-up) or torch.distributed (RCCL through PyTorch).  auto "
                         "(default, backend nccl): try rccl first and fall back -- by FRESH rank processes started from the GPU-free "
                         "supervisor -- to torch, then to torch with plain broadcasts; the line's `launch` object says which ran")
    ap.add_argument("--attempts", default="", help="N > 1: explicit comma-separated list of attempts (%s)" % ", ".join(ATTEMPTS))
    ap.add_argument("--attempt-name", default="", help=argparse.SUPPRESS)      # set by the supervisor for a rank process
    ap.add_argument("--exchange", default="auto", choices=["auto", "broadcast", "scatter_allgather"],
                    help="N > 1, torch.distributed: the panel exchange (auto: probed / chosen by measurement)")
    ap.add_argument("--first-stall", type=float, default=360.0, help="supervisor: seconds a rank process may take to its first progress mark (the first `import torch` of a fresh box takes minutes)")
    ap.add_argument("--stall", type=float, default=240.0, help="supervisor: seconds without a new progress mark after which a rank process counts as hung")
    ap.add_argument("--workload", default="gpr", choices=["gpr", "cfg5"],
                    help="gpr (default): the BASELINE headline; cfg5: side line for BASELINE configs[4] -- conditional() / SVGP bound with "
                         "M inducing points over N data points, the data points sharded over the ranks (gpflowSlim/distributed_sparse.py)")
    ap.add_argument("--cfg5-m", type=int, default=4096)
    ap.add_argument("--cfg5-n", type=int, default=1000000)
    args = ap.parse_args()
    alive.period = max(0.5, min(15.0, args.stall / 3.0))

    if args.gpus > 1 and not os.environ.get("GPS_BENCH_CHILD"):
        # Not a rank process yet: be the GPU-free supervisor of the rank processes (before torch / the library / any GPU call).
        attempts = plan_attempts(args)
        if "WORLD_SIZE" not in os.environ:
            # no launcher around us: start all N ranks
            import tempfile
            sys.exit(supervise(list(range(args.gpus)), args.gpus, sys.argv[1:], attempts, tempfile.mkdtemp(prefix="gps_bench_"),
                               args.first_stall, args.stall, True))
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            print("bench.py: --gpus %d but the launcher started %s ranks (WORLD_SIZE)" % (args.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)
            sys.exit(2)
        if len(attempts) > 1:
            # a launcher started one bench.py per rank: each becomes the supervisor of its own rank process; the supervisors of
            # the run meet in a directory named after what they share (the launcher's pid and rendezvous port; same node)
            # (+ the launcher's run id and restart count: a launcher that restarts its workers keeps pid and port, and the
            # files of the previous round -- an old port, old verdicts -- must not be found by the new one)
            rdv = os.environ.get("GPS_BENCH_RDV_DIR") or os.path.join(
                "/tmp", "gps_bench_%d_%s_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"),
                                                   "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "none") if ch.isalnum())[:24],
                                                   os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")))
            r = int(os.environ.get("RANK", "0"))
            sys.exit(supervise([r], args.gpus, sys.argv[1:], attempts, rdv, args.first_stall, args.stall, False))
        args.attempt_name = attempts[0]          # one configuration only: the launcher's rank process runs it itself
    if args.attempt_name:
        args.comm = "rccl" if args.attempt_name == "rccl" else "torch"
        if args.attempt_name == "torch-broadcast":
            args.exchange, args.no_dist_autotune = "broadcast", True
    elif args.comm == "auto":
        args.comm = "torch"
    progress("start")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and (args.gpus > 1 or os.environ.get("GPS_BENCH_CHILD")):
        print("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.force_device >= 0:
        local_rank = args.force_device
    os.environ["GPFLOWSLIM_DEVICE"] = str(local_rank)

    import numpy as np
    with alive("importing_torch"):          # (minutes on a freshly provisioned box, and nothing a GPU can hang)
        import torch
        import torch.distributed as dist
    progress("torch_imported")
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
        progress("process_group_up")

    import gpflowSlim as gpf

    if args.workload == "cfg5":
        run_cfg5(args, rank, world, torch, dist, gpf, np)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

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
    metric = "GPR log-marginal-likelihood evals/sec + predict_f latency, fp64, N=%d D=%d" % (n, d)
    workload = "RBF(ARD) GPR log-marginal-likelihood, N=%d D=%d R=1 fp64, X resident in HBM" % (n, d)
    flops_eval = n ** 3 / 3 + n * n * (2 * d + 6) + n * n            # one LML evaluation

    def set_step(i, per_rank):
        # a different point of hyper-parameter space on every step (and, for independent evaluations, every rank)
        s = 1.0 + 0.01 * (((i * world + rank) if per_rank else i) % 17)
        kern._ls.assign(ls0 * s)
        kern._variance.assign(1.0 / s)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    extra = {}
    if world == 1:
        for i in range(args.warmup):
            set_step(-1 - i, False)
            model.compute_log_likelihood()
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            set_step(i, False)
            lml = model.compute_log_likelihood()
        sync()
        elapsed = time.perf_counter() - t0
        progress("timed_steps_done")
        scaling, parallelism = "weak", "1 GPU (fused single-GPU path)"
    else:
        # ---- one factorisation over all ranks per step; a hang in the collective path must end the process, not the round
        from gpflowSlim.distributed import TorchComm, gpr_lml_distributed
        state = {"done": False}

        def on_timeout():
            if state["done"]:
                return
            if rank == 0:
                print(json.dumps({"metric": metric, "value": 0.0, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
                                  "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
                                  "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": workload},
                                  "error": "watchdog: the block-column run did not finish within %.0f s" % args.dist_timeout}), flush=True)
            os._exit(2)
        def arm():
            t = threading.Timer(args.dist_timeout, on_timeout)
            t.daemon = True
            t.start()
            return t
        timer = arm()
        if args.comm == "rccl":
            from gpflowSlim.distributed import RcclComm

            def carry(uid):
                box = [uid]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            comm = RcclComm(h, rank, world, bootstrap=carry)
        else:
            comm = TorchComm(mode=None if args.exchange == "auto" else args.exchange)
        progress("communicator_up")
        tune = None
        set_step(-1, False)
        gpr_lml_distributed(model, comm, nb=args.dist_nb, lookahead=args.dist_lookahead)      # (also sets the communicator up)
        sync()
        progress("first_distributed_evaluation_done")
        if not args.no_dist_autotune:
            # untimed: pick the panel exchange and the panel width by measurement on this node (all ranks agree through
            # max-over-ranks times); the timed steps below then run one fixed configuration
            tune = {"exchange_s": comm.autotune() if (args.backend == "nccl" and args.comm == "torch" and args.exchange == "auto") else {},
                    "candidates_ms": {}}
            progress("exchange_chosen")
            best = None
            for nb_c, la_c in ((args.dist_nb, args.dist_lookahead), (2 * args.dist_nb, args.dist_lookahead), (args.dist_nb // 2, args.dist_lookahead)):
                if nb_c < 128 or nb_c % 128 or n // nb_c < 2 * world:
                    continue
                sync(); tc = time.perf_counter()
                gpr_lml_distributed(model, comm, nb=nb_c, lookahead=la_c)
                sync()
                dt = max_over_ranks(time.perf_counter() - tc)
                tune["candidates_ms"]["nb=%d,lookahead=%d" % (nb_c, la_c)] = round(1e3 * dt, 3)
                progress("candidate_nb_%d" % nb_c)
                if best is None or dt < best[0]:
                    best = (dt, nb_c, la_c)
            if best is not None:
                args.dist_nb, args.dist_lookahead = best[1], best[2]
            tune["chosen"] = {"nb": args.dist_nb, "lookahead": args.dist_lookahead, "exchange": comm.mode}
        for i in range(args.warmup):
            set_step(-1 - i, False)
            gpr_lml_distributed(model, comm, nb=args.dist_nb, lookahead=args.dist_lookahead)
        sync()
        progress("warmup_done")
        bytes0 = comm.bytes_sent
        t0 = time.perf_counter()
        for i in range(args.steps):
            set_step(i, False)
            lml = gpr_lml_distributed(model, comm, nb=args.dist_nb, lookahead=args.dist_lookahead)
        sync()
        elapsed = time.perf_counter() - t0
        progress("timed_steps_done")
        # the collectives below (gathers, the parity evaluation, the independent evaluations, the final barrier) stay
        # under a watchdog of their own: it is only disarmed after the last barrier
        timer.cancel()
        timer = arm()
        scaling = "strong"
        parallelism = "1-D block-cyclic column Cholesky over %d GPUs (nb=%d, look-ahead %d, panel exchange: %s over %s)" % (
            world, args.dist_nb, args.dist_lookahead, comm.mode, args.backend if args.comm == "torch" else "the library's own RCCL binding")
        stage = h.last_stage_ms()
        per_rank = [None] * world
        dist.all_gather_object(per_rank, dict({k: round(v, 3) for k, v in stage.items()}, device_bytes=h.device_bytes()))
        sent = torch.tensor([float(comm.bytes_sent - bytes0)], dtype=torch.float64, device="cuda")
        dist.all_reduce(sent, op=dist.ReduceOp.SUM)
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None
        except Exception:
            rccl = None
        extra["distributed"] = {"nb": args.dist_nb, "lookahead": args.dist_lookahead, "exchange": comm.mode,
                                "backend": args.backend, "rccl_ranks": dist.get_world_size(), "rccl_version": rccl,
                                "factor_storage": "partitioned: a rank holds its own block columns only (8 N^2 / P bytes + O(N nb)); "
                                                  "three comm buffers, the updates read a panel from the buffer it arrived in",
                                "stage_ms_per_rank_last_step": per_rank,
                                "payload_bytes_per_eval_all_ranks": float(sent.item()) / args.steps,
                                "lml_last_step": lml, "autotune": tune}
    elapsed = max_over_ranks(elapsed)
    ms_per_step = 1e3 * elapsed / args.steps
    value = args.steps / elapsed                     # whole-job evaluations / s (N > 1: ONE evaluation per step)

    if world > 1:
        # parity of the block-column result with the fused single-GPU evaluation of the same hyper-parameters (rank 0's GPU)
        ref1 = model.compute_log_likelihood()
        extra["distributed"]["parity_rel_err_vs_one_gpu"] = abs(lml - ref1) / abs(ref1)
        extra["distributed"]["speedup_vs_fused_one_gpu_eval"] = None
        # side measurement: independent evaluations, one per GPU and step, no collective (weak scaling)
        if args.independent_steps > 0:
            set_step(-1, True); model.compute_log_likelihood()
            sync(); t1 = time.perf_counter()
            for i in range(args.independent_steps):
                set_step(i, True)
                model.compute_log_likelihood()
            sync()
            ti = max_over_ranks(time.perf_counter() - t1)
            progress("independent_evaluations_done")
            extra["independent_evals"] = {"evals_per_s_all_gpus": round(world * args.independent_steps / ti, 4),
                                          "ms_per_eval_per_gpu": round(1e3 * ti / args.independent_steps, 3),
                                          "steps": args.independent_steps, "scaling": "weak", "collective": None}
            extra["distributed"]["speedup_vs_fused_one_gpu_eval"] = round((1e3 * ti / args.independent_steps) / ms_per_step, 3)

    out = None
    if rank == 0:
        stages = h.last_stage_ms()
        if world > 1:
            set_step(args.steps, False); model.compute_log_likelihood(); stages = h.last_stage_ms()
        # predict_f latency (reference semantics = cold: re-factorises, models/gpr.py:119-121)
        # five calls each, median reported with min / max beside it: one slow call (a clock ramp, a page fault of a fresh
        # box) then shows up as an outlier instead of as the number
        def latency(reps=5, Xq=None):
            Xq = Xnew if Xq is None else Xq
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                model.predict_f(Xq)
                torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t1))
            ts.sort()
            return {"median": round(ts[len(ts) // 2], 2), "min": round(ts[0], 2), "max": round(ts[-1], 2), "calls": reps}
        model.reuse_factor = False
        model.predict_f(Xnew)                       # (allocates the [n_new, N] work space)
        cold = latency()
        model.reuse_factor = True
        model.predict_f(Xnew)
        warm = latency()
        cold_ms, warm_ms = cold["median"], warm["median"]
        # latency table (examples/gpr.py:56,60-68 predicts on ~51 points every 10 steps): N* = 64 / 256 / n_new on the resident
        # factor, and the FIRST such call after a new factor -- it builds the factor's 2048-column inverse blocks
        # (csrc/gps_gpr.hip: gpr_wide_inverse), every later call on that factor reuses them
        table = {}
        for ns_t in sorted(set([64, 256, args.n_new])):
            Xq = np.ascontiguousarray(Xnew[:ns_t]) if ns_t <= Xnew.shape[0] else rng.standard_normal((ns_t, d))
            model.predict_f(Xq)
            w = latency(Xq=Xq)
            set_step(args.steps + 1 + len(table), False); model.compute_log_likelihood()      # a new factor: no wide blocks yet
            torch.cuda.synchronize(); t1 = time.perf_counter()
            model.predict_f(Xq)
            torch.cuda.synchronize(); first_ms = 1e3 * (time.perf_counter() - t1)
            table["n_new=%d" % ns_t] = {"warm_ms": w["median"], "warm_min_ms": w["min"], "warm_max_ms": w["max"],
                                        "first_call_after_new_factor_ms": round(first_ms, 2),
                                        "trsm_frac_of_peak": round(float(n) * n * (64 if ns_t <= 64 else h_npad(ns_t)) / (w["median"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)}
        table["note"] = ("warm: median of 5 predict_f calls on the resident factor (wall time incl. the host round trip of mean and "
                         "variance); first_call_after_new_factor: one call right after a likelihood evaluation with new "
                         "hyper-parameters -- includes building the factor's wide inverse blocks once; trsm_frac_of_peak = N^2 x padded N* "
                         "flop (test points padded to whole 128-row tiles, to half a tile up to 64 points) / warm time / 78.6 TFLOP/s")
        progress("predict_latency_done")
        # predict_f throughput at N* = 8192 on the resident factor: trsm N^2 N* flop on the MFMA, then one
        # HBM pass over A^T for the mean and the variance
        Xtp = rng.standard_normal((args.n_new_tp, d))
        model.predict_f(Xtp)
        h.profile_reset(); h.profile_enable(True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model.predict_f(Xtp)
        torch.cuda.synchronize(); tp_s = time.perf_counter() - t1
        h.profile_enable(False)
        pg, pr = h.profile_get("gemm_f64"), h.profile_get("reduce")
        predict_tp = {"n_new": args.n_new_tp, "ms": round(1e3 * tp_s, 2), "points_per_s": round(args.n_new_tp / tp_s, 1),
                      "trsm_tflops": round(float(n) * n * args.n_new_tp / (pg["ms"] * 1e-3) / 1e12, 2),
                      "trsm_frac_of_peak": round(float(n) * n * args.n_new_tp / (pg["ms"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                      "rowdot_gbs": round(8.0 * args.n_new_tp * h_npad(n) / (pr["ms"] * 1e-3) / 1e9, 1),
                      "rowdot_frac_of_hbm_peak": round(8.0 * args.n_new_tp * h_npad(n) / (pr["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

        progress("predict_throughput_done")
        # LML + analytic gradient (the quantity an optimiser step consumes)
        model.reuse_factor = False
        model.compute_log_likelihood_and_gradients()          # first call allocates the K^-1 work space
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model.compute_log_likelihood_and_gradients()
        torch.cuda.synchronize(); grad_ms = 1e3 * (time.perf_counter() - t1)

        # per-step latency at the size of the reference's only exhibited workload (examples/gpr.py:36,60: N ~ 455, D = 13,
        # 20 000 optimiser steps, each one LML + gradient): microseconds per step and launches per step, same kernel family
        small = None
        if args.small_n:
            small = {}
            for ns_ in [int(v) for v in args.small_n.split(",") if v]:
                if ns_ > n:
                    continue
                ks = gpf.kernels.RBF(d, variance=1.0, lengthscales=ls0, ARD=True)
                msml = gpf.models.GPR(X[:ns_], Y[:ns_], ks, obs_var=0.1)
                row = {}
                for what, fn in (("lml", msml.compute_log_likelihood), ("lml_plus_gradient", msml.compute_log_likelihood_and_gradients)):
                    reps = 200 if ns_ <= 1024 else 100
                    for i in range(5):
                        ks._ls.assign(ls0 * (1.0 + 0.01 * i)); fn()
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    for i in range(reps):
                        ks._ls.assign(ls0 * (1.0 + 0.01 * (i % 17))); fn()
                    torch.cuda.synchronize(); us = 1e6 * (time.perf_counter() - t1) / reps
                    h.profile_reset(); h.profile_enable(True)
                    fn()
                    h.profile_enable(False)
                    launches = sum(h.profile_get(k)["launches"] for k in ("gemm_f64", "potrf_base", "kmat", "trsv", "reduce", "other"))
                    # the same through the C entry point alone (the drop-in boundary; no Python model layer around it)
                    prog_s, resid_s = ks._program(d), msml._resid()
                    cfn = (lambda: h.gpr_lml(prog_s, 0.1, resid_s)) if what == "lml" else (lambda: h.gpr_lml_grad(prog_s, 0.1, resid_s))
                    cfn()
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    for i in range(reps):
                        cfn()
                    torch.cuda.synchronize(); us_c = 1e6 * (time.perf_counter() - t1) / reps
                    row[what] = {"us_per_step": round(us, 1), "c_entry_point_us": round(us_c, 1), "launches_per_step": int(launches), "steps_timed": reps}
                small["n=%d" % ns_] = row
            small["note"] = ("us_per_step: wall time per call through the Python API incl. the host round trip of the result, hyper-parameters "
                             "changing every step; c_entry_point_us: gps_gpr_lml / gps_gpr_lml_grad alone; D=%d RBF(ARD)" % d)

        progress("gradient_and_small_n_done")
        roofline = None
        hbm_bound = None
        if not args.no_roofline:
            # one extra evaluation of the same workload with every launch bracketed by HIP events on the
            # library's stream; dominant kernel = gemm_nt_f64_kernel (fp64 MFMA trailing updates / solves)
            # (with the look-ahead off: per-launch durations of kernels that overlap on two streams do not add up to
            # wall time, and the roofline is about the kernel, not about the schedule around it)
            h.set_option("potrf_lookahead", 0)
            h.profile_reset(); h.profile_enable(True)
            set_step(args.steps, False)
            model.compute_log_likelihood()
            h.profile_enable(False)
            h.set_option("potrf_lookahead", 1)
            g = h.profile_get("gemm_f64")
            classes = {k: h.profile_get(k) for k in ("gemm_f64", "potrf_base", "kmat", "trsv", "reduce", "other")}
            # the algorithmic work of the class is the potrf's N^3/3 (all of it lands in this kernel)
            alg = n ** 3 / 3.0
            if not g["ms"] > 0:            # (N <= 2048: the whole factorisation is one cooperative launch of another class)
                g = dict(classes["potrf_base"])
            achieved = alg / (g["ms"] * 1e-3) / 1e12
            traffic, traffic_src, traffic_total, traffic_launches = None, None, None, None
            import glob
            for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_f64_hbm_bytes_per_launch.json")), reverse=True):
                if not (n == 32768 and d == 8):
                    break
                with open(pmc) as f:
                    rec = json.load(f)
                # measured on exactly these device sources, over exactly this launch set (the KC_GEMM class: gemm_nt_f64_kernel<*>
                # and trsm_panel_*kernel<*>): traffic x launches == traffic_total_bytes
                if rec.get("kernel_source_sha") == kernel_source_sha() and int(rec.get("launches", -1)) == int(g["launches"]):
                    traffic_total, traffic_launches = rec.get("hbm_bytes_total"), int(rec["launches"])
                    traffic, traffic_src = traffic_total / traffic_launches, "profiles/" + os.path.basename(pmc)
                    break
            peak_meas, _ = h.diag_mfma_f64(2)
            roofline = {"bound": "mfma", "kernel": "gemm_nt_f64_kernel (+ trsm_panel_kernel / trsm_panel_persistent_kernel, the 512-column solve built from the same MFMA products: the library's GEMM class)",
                        "achieved": round(achieved, 3),
                        "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
                        "traffic": traffic, "traffic_total_bytes": traffic_total, "traffic_launches": traffic_launches,
                        "traffic_source": traffic_src, "launches": g["launches"],
                        "avg_launch_us": round(1e3 * g["ms"] / max(g["launches"], 1), 3),
                        "algorithmic_flops_per_launch": alg / max(g["launches"], 1),
                        "launch_counted": {"flops": g["flops"], "tflops": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 3),
                                           "note": "whole diagonal tiles of lower-triangular updates, the block-inverse products of the leaves"},
                        "peak_measured_issue_rate": round(peak_meas, 2),
                        "whole_eval": {"flops": flops_eval,
                                       "tflops": round(flops_eval / (stages["total"] * 1e-3) / 1e12, 3),
                                       "frac": round(flops_eval / (stages["total"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                                       "ms": round(stages["total"], 3)},
                        "per_class_ms": dict({k: round(v["ms"], 3) for k, v in classes.items()},
                                             schedule="look-ahead OFF: the instrumented pass (every launch bracketed by events, launches serialised)"),
                        "stage_ms": dict({k: round(v, 3) for k, v in stages.items()},
                                         schedule="look-ahead ON: stage events of a timed evaluation (what value / ms_per_step measure)"),
                        "kernel_source_sha": kernel_source_sha(),
                        "note": "instrumented evaluation with the look-ahead off (launches serialised); value / ms_per_step are measured with it on"}
            # the HBM-bound kernels of the evaluation: algorithmic bytes / class time
            npad = h_npad(n)
            kb, tb = 4.0 * npad * npad, 4.0 * npad * npad
            # (forward substitution: wall time between the stage events of the timed evaluation -- the instrumented pass adds
            # ~5 us of event handling to each of its ~500 launches)
            tv = dict(classes["trsv"])
            if stages.get("trsv", 0.0) > 0.2 and tv["ms"] > 0:
                tv["ms"] = stages["trsv"]
            km = classes["kmat"]["ms"]          # (0 at sizes where the one-launch small-N path generates K + noise itself)
            hbm_bound = {"kmat": {"bytes": kb, "ms": round(km, 3),
                                  "gbs": round(kb / (km * 1e-3) / 1e9, 1) if km > 0 else None,
                                  "frac_of_hbm_peak": round(kb / (km * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if km > 0 else None},
                         "trsv": ({"bytes": tb, "ms": round(tv["ms"], 3), "launches": tv["launches"],
                                   "gbs": round(tb / (tv["ms"] * 1e-3) / 1e9, 1),
                                   "frac_of_hbm_peak": round(tb / (tv["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "forward substitution as one wavefront launch over the 128-row blocks of L (trsv_wave.hip): one pass over the lower triangle, bounded by the block-to-block hand-over chain (N / 128 fabric round trips)"}
                                  if tv["ms"] > 0 else
                                  {"bytes": 0.0, "ms": 0.0, "launches": 0, "gbs": None, "frac_of_hbm_peak": None,
                                   "note": "no forward-substitution pass at this size: (Y - m)^T rides through the factorisation as augmented rows"}),
                         "rowdot_predict_f": {"gbs": predict_tp["rowdot_gbs"], "frac_of_hbm_peak": predict_tp["rowdot_frac_of_hbm_peak"]},
                         "peak_gbs": HBM_PEAK_GBS}

        progress("roofline_done")
        if world > 1 and roofline is not None:
            # the line of a multi-GPU run is about the whole job: algorithmic flop of one evaluation over the time of the
            # block-column step on P GPUs against P times the peak (= the per-GPU fraction); the per-kernel evidence of the
            # single-GPU instrumented pass stays beside it
            one = roofline
            ach = flops_eval / (ms_per_step * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "gemm_nt_f64_kernel (panel solves + trailing updates of the block-column schedule, all ranks)",
                        "achieved": round(ach, 3), "peak": round(FP64_MFMA_PEAK_TFLOPS * world, 1), "unit": "TFLOP/s",
                        "frac": round(ach / (FP64_MFMA_PEAK_TFLOPS * world), 4), "per_gpu_frac": round(ach / (FP64_MFMA_PEAK_TFLOPS * world), 4),
                        "per_gpu_tflops": round(ach / world, 3), "traffic": None,
                        "algorithmic_flops_per_step": flops_eval, "ms_per_step": round(ms_per_step, 3),
                        "note": "whole job: SURVEY 8(d) flop of ONE evaluation / (the measured step time x P GPUs x 78.6 TFLOP/s); "
                                "`one_gpu_kernel` = the instrumented fused evaluation on rank 0's GPU",
                        "one_gpu_kernel": one}
        cpu = None
        if not args.no_cpu_baseline:       # rank 0's host cores, whatever the number of GPUs (the other ranks wait at the last barrier)
            import oracle.gp_oracle as orc       # the checker: cpu_baseline leg + its parity gate only
            ns = min(args.cpu_sample_n, n)
            Xc, Yc = X[:ns], Y[:ns]
            spec = {"type": "rbf", "variance": orc.constrained(1.0), "lengthscales": orc.constrained(ls0), "input_dim": d}
            with alive("cpu_baseline_running"):
                # (testing: GPS_BENCH_TEST_SLOW_TAIL = seconds rank 0 spends here on top, as if its host cores were shared)
                time.sleep(float(os.environ.get("GPS_BENCH_TEST_SLOW_TAIL", "0")))
                ref_lml, tm = orc.gpr_lml_timed(spec, Xc, Yc, orc.constrained(0.1))
                t1 = time.perf_counter(); orc.rbf_K_inplace(spec, Xc); kin = time.perf_counter() - t1
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
            # the full workload, one repetition -- or, for a smaller --cpu-sample-n, a stage-wise extrapolation
            # (K build and the triangular solve are O(N^2), dpotrf O(N^3)) that the line flags as such
            q = float(n) / ns
            full_s = tm["kmat_s"] * q ** 2 + tm["potrf_s"] * q ** 3 + tm["trsv_s"] * q ** 2
            how = ("the full workload, timed once, nothing extrapolated" if ns == n else
                   "extrapolated stage by stage to N=%d: K build and solve x%.0f, dpotrf x%.0f -> %.1f s per evaluation" % (
                       n, q ** 2, q ** 3, full_s))
            cpu = {"value": round(1.0 / full_s, 6), "unit": "evals/s", "cores": threads,
                   "kind": "port", "extrapolated": ns != n,
                   "sample": "oracle (numpy/scipy %s, %d BLAS threads of %d logical cores) LML at N=%d D=%d: %.2f s = K build "
                             "unfused like the reference's TF graph %.2f (in place: %.2f) + dpotrf %.2f + dtrtrs / reductions %.2f; "
                             "%s; stand-in for the reference TF-CPU path, which cannot run (no TensorFlow)" % (
                                 blas, threads, os.cpu_count(), ns, d, tm["total_s"], tm["kmat_s"], kin, tm["potrf_s"], tm["trsv_s"], how),
                   "sample_seconds": round(tm["total_s"], 3), "kmat_inplace_seconds": round(kin, 3),
                   "sample_parity_rel_err": abs(got - ref_lml) / abs(ref_lml)}

        out = {"metric": metric, "value": round(value, 4), "unit": "evals/s",
               "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
               "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": workload, "n": n, "d": d, "n_new": args.n_new, "parallelism": parallelism},
               "predict_f_latency_ms": {"cold_refactor": round(cold_ms, 2), "warm_resident_factor": round(warm_ms, 2), "n_new": args.n_new,
                                        "statistic": "median of 5 calls", "cold_calls": cold, "warm_calls": warm},
               "predict_f_latency_table": table,
               "fallback_counters": {"lookahead_retries": int(h.profile_get("lookahead_retries")["launches"]),
                                     "trsv_wave_fallbacks": int(h.profile_get("trsv_wave_fallbacks")["launches"]),
                                     "small_n_fallbacks": int(h.profile_get("small_n_fallbacks")["launches"]),
                                     "note": "evaluations re-run without look-ahead after a missed hand-over / wavefront substitutions "
                                             "that gave up / one-launch small-N factorisations that gave up, over the whole life of "
                                             "this process' handle (0 = the fast paths ran)"},
               "predict_f_throughput": predict_tp,
               "lml_plus_gradient_ms": round(grad_ms, 2),
               "small_n_latency": small,
               "stage_ms_one_gpu_eval": {k: round(v, 3) for k, v in stages.items()},
               "lml_last_step": lml,
               "roofline": roofline, "hbm_bound_kernels": hbm_bound, "cpu_baseline": cpu}
        out.update(extra)
        progress("cpu_baseline_done")
        print(json.dumps(out), flush=True)
    if world > 1:
        with alive("last_barrier"):          # (ranks != 0 wait here while rank 0 runs its single-GPU tail and the CPU stand-in)
            dist.barrier()
        state["done"] = True
        timer.cancel()
        dist.destroy_process_group()


def run_cfg5(args, rank, world, torch, dist, gpf, np):
    """Side line for BASELINE configs[4] (examples/svgp.py shape: RBF, M inducing points, N data points): one step = one
    conditional() (Kuu potrf + Kuf build + trsm + one-pass mean / variance; conditionals.py:24-119) over all N points, the
    points sharded over the ranks (no data-path collective: only the gather of the [N, 1] outputs) -> weak in M, strong in
    N: `value` = points / s of the whole job.  The SVGP bound (one scalar reduction) is timed beside it."""
    from gpflowSlim.distributed import TorchComm, SingleComm
    from gpflowSlim.distributed_sparse import conditional_distributed, svgp_bound_distributed, sparse_bound_distributed
    m, n, d = args.cfg5_m, args.cfg5_n, 8
    rng = np.random.default_rng(20240607)
    X = rng.standard_normal((n, d))
    Y = np.sin(X @ (rng.standard_normal((d, 1)) / np.sqrt(d))) + 0.1 * rng.standard_normal((n, 1))
    Z = X[:m].copy()
    f = 0.1 * rng.standard_normal((m, 1))
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=np.sqrt(d) * np.ones(d), ARD=True)
    comm = TorchComm() if world > 1 else SingleComm()
    # a rank that leaves a sharded evaluation early (an error on its shard) would leave the others in the collective: the
    # run is under a watchdog like the block-column one
    done = {"v": False}

    def on_timeout():
        if done["v"]:
            return
        if rank == 0:
            print(json.dumps({"metric": "conditional() test points/sec, RBF, M=%d inducing points, N=%d points, fp64" % (m, n), "value": 0.0,
                              "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": "BASELINE configs[4]"},
                              "error": "watchdog: the sharded run did not finish within %.0f s" % args.dist_timeout}), flush=True)
        os._exit(2)
    watchdog = threading.Timer(args.dist_timeout, on_timeout)
    watchdog.daemon = True
    if world > 1:
        watchdog.start()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        sync(); t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / steps, out
    t_cond, (fm, fv) = timed(lambda: conditional_distributed(X, Z, kern, f, comm=comm, white=True), args.steps, args.warmup)
    progress("conditional_done")
    mod = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.1), Z=Z, q_diag=True, whiten=True)
    t_svgp, elbo = timed(lambda: svgp_bound_distributed(mod, comm), max(1, args.steps // 2), 1)
    sg = gpf.models.SGPR(X, Y, kern, Z=Z, obs_var=0.1)
    t_sgpr, sbound = timed(lambda: sparse_bound_distributed(sg, comm), max(1, args.steps // 2), 1)
    if rank == 0:
        mp = ((m + 127) // 128) * 128
        flops = float(mp) * mp * n + mp ** 3 / 3.0                       # trsm M^2 N + Kuu potrf
        print(json.dumps({
            "metric": "conditional() test points/sec, RBF, M=%d inducing points, N=%d points, fp64" % (m, n),
            "value": round(n / t_cond, 1), "unit": "points/s", "n_gpus": dist.get_world_size() if world > 1 else 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * t_cond, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: conditional (white) with M=%d inducing points over N=%d points, D=%d, "
                                   "data points sharded over the ranks" % (m, n, d),
                       "parallelism": "%d rank(s), contiguous shards of the data points, Kuu factored by every rank, no data-path "
                                      "collective (outputs gathered)" % world},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_f64_kernel (trsm_rec)", "achieved": round(flops / t_cond / 1e12, 3),
                         "peak": FP64_MFMA_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                         "frac": round(flops / t_cond / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world), 4), "traffic": None,
                         "note": "whole-step rate incl. the 8 M N-byte Kuf build, the mean / variance pass and the host gather"},
            "cpu_baseline": None,
            "svgp_bound": {"ms": round(1e3 * t_svgp, 3), "elbo": elbo, "q_diag": True,
                           "collective": "one gathered scalar per rank, added in rank order"},
            "sgpr_bound": {"ms": round(1e3 * t_sgpr, 3), "bound": sbound,
                           "collective": "one in-place device all-reduce of [A A^T | A err | diag | scalars] (%d doubles) inside gps_sgpr" % (
                               mp * mp + mp * 1 + mp + 4)},
            "checksum": {"fmean_sum": float(fm.sum()), "fvar_min": float(fv.min())}}), flush=True)
    if world > 1:
        dist.barrier()
    done["v"] = True
    watchdog.cancel()


def h_npad(n):
    return ((n + 127) // 128) * 128


def _main_reporting_failures():
    """A rank of a multi-rank run that fails says so in the contract's own format before it dies: rank 0 prints one JSON line
    with `value` 0 and an `error` field (the driver's record then carries the reason, not an empty stdout), every rank exits
    non-zero at once -- without running the interpreter's clean-up, which could wait on a broken process group -- so that the
    launcher (torch.distributed.run, or launch_ranks above) stops the other ranks instead of leaving them in a collective."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or "RANK" not in os.environ:
        return main()
    try:
        return main()
    except SystemExit:
        raise
    except BaseException as e:       # noqa: BLE001 -- reported, then the process ends
        import traceback
        traceback.print_exc()
        try:                             # a native communicator is given up at once: the peers' collectives then fail instead of waiting
            import gpflowSlim
            gpflowSlim.get_handle().comm_abort()
        except BaseException:            # noqa: BLE001
            pass
        if os.environ.get("RANK", "0") == "0":
            print(json.dumps({"metric": "GPR log-marginal-likelihood evals/sec + predict_f latency, fp64, N=32768 D=8", "value": 0.0,
                              "unit": "evals/s", "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "failed run"},
                              "error": "%s: %s" % (type(e).__name__, str(e)[:500])}), flush=True)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(3)


if __name__ == "__main__":
    _main_reporting_failures()
