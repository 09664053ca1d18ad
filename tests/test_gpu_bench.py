"""bench.py end to end on the GPU box: the N = 1 line, and the N > 1 branch -- one block-column factorisation over all
ranks per step -- with two processes sharing the one GPU of the box over gloo (RCCL refuses two ranks on one device;
everything else is the production path: torch.distributed.run launch, TorchComm, HipPanelOps, the two-lane schedule)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
          "dtype", "data", "config", "roofline", "cpu_baseline"}


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_one_gpu_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--npoints", "4096",
                        "--cpu-sample-n", "1024", "--num-new-throughput", "1024"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    out = _last_json(p.stdout)
    assert COMMON <= set(out), sorted(COMMON - set(out))
    assert out["n_gpus"] == 1 and out["unit"] == "evals/s" and out["dtype"] == "f64" and out["value"] > 0
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None or rf["kernel_source_sha"]          # never a constant from other sources
    assert rf["launch_counted"]["flops"] >= 4096 ** 3 / 3
    for k in ("kmat", "rowdot_predict_f"):
        assert 0 < out["hbm_bound_kernels"][k]["gbs"] < 8000
    tv = out["hbm_bound_kernels"]["trsv"]              # (N = 4096: alpha comes out of the factorisation, no trsv pass)
    assert tv["gbs"] is None or 0 < tv["gbs"] < 8000
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample_parity_rel_err"] <= 1e-8
    assert out["predict_f_throughput"]["points_per_s"] > 0
    tab = out["predict_f_latency_table"]                # N* = 64 / 256 / n_new on the resident factor + the first call after a new factor
    for k in ("n_new=64", "n_new=256", "n_new=%d" % out["config"]["n_new"]):
        assert 0 < tab[k]["warm_ms"] <= tab[k]["warm_max_ms"] and tab[k]["first_call_after_new_factor_ms"] > 0 and 0 < tab[k]["trsm_frac_of_peak"] < 1


def test_bench_two_ranks_block_column_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--force-device", "0", "--npoints", "4096", "--dist-nb", "256", "--num-new-throughput", "512",
           "--dist-timeout", "200", "--no-dist-autotune", "--cpu-sample-n", "1024"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    out = _last_json(p.stdout)
    assert COMMON <= set(out), sorted(COMMON - set(out))
    assert out["distributed"]["autotune"] is None
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 3 and out["value"] > 0
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) <= 1e-3 * out["value"]          # ONE evaluation of the whole job per step
    dd = out["distributed"]
    assert dd["rccl_ranks"] == 2 and dd["nb"] == 256 and dd["parity_rel_err_vs_one_gpu"] <= 1e-9
    assert len(dd["stage_ms_per_rank_last_step"]) == 2
    n_panels = 4096 // 256
    rows = lambda j: 4096 + 128 - j * 256
    payload = sum(-(-(rows(j) * 256 + 2 * 2 * 128 * 128 + 4) // 2) * 2 for j in range(n_panels)) * 8   # every panel to the one other rank
    assert abs(dd["payload_bytes_per_eval_all_ranks"] - payload) <= 1e-6 * payload
    assert "block-cyclic" in out["config"]["parallelism"]
    assert out["independent_evals"]["scaling"] == "weak" and out["independent_evals"]["evals_per_s_all_gpus"] > 0
    # N > 1 lines carry the whole-job roofline (per-GPU fraction) and the CPU stand-in timed on rank 0's host cores
    rf = out["roofline"]
    assert rf["peak"] == pytest.approx(2 * 78.6) and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["one_gpu_kernel"]["kernel"].startswith("gemm_nt_f64_kernel")
    assert abs(rf["achieved"] - rf["algorithmic_flops_per_step"] / (out["ms_per_step"] * 1e-3) / 1e12) <= 1e-2 * rf["achieved"]
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert out["predict_f_latency_ms"]["warm_calls"]["calls"] == 5 and out["fallback_counters"]["lookahead_retries"] == 0


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command form): bench.py starts the two rank
    processes itself and relays rank 0's line -- the block-column path, not the 1-GPU path timed twice."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
           "--force-device", "0", "--npoints", "4096", "--dist-nb", "256", "--num-new-throughput", "512", "--dist-timeout", "200",
           "--no-dist-autotune", "--independent-steps", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                                   # ONE JSON line, rank 0's
    out = json.loads(lines[0])
    assert COMMON <= set(out), sorted(COMMON - set(out))
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 2 and out["value"] > 0
    dd = out["distributed"]
    assert dd["rccl_ranks"] == 2 and dd["parity_rel_err_vs_one_gpu"] <= 1e-9
    assert "block-cyclic" in out["config"]["parallelism"]
    assert out["launch"]["attempt"] == "torch" and out["launch"]["failed_attempts"] == []          # (backend gloo: the one attempt)
    # a --gpus that disagrees with the launcher's rank count is an error, not a silently different run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True,
                         text=True, timeout=120, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode == 2 and "--gpus 4" in bad.stderr


def test_bench_slow_rank0_tail_does_not_trip_the_stall_supervisor():
    """ranks != 0 finish early and wait in the last barrier while rank 0 runs its single-GPU tail and the CPU
    stand-in; with a stall limit shorter than that tail the supervisor used to kill the waiting (healthy) ranks.  They now
    send a heartbeat from the barrier, and rank 0 one from the CPU stand-in: --stall 6 against a 15-s tail must succeed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GPS_BENCH_TEST_SLOW_TAIL"] = "15"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--backend", "gloo",
           "--force-device", "0", "--npoints", "4096", "--dist-nb", "256", "--num-new-throughput", "256", "--dist-timeout", "200",
           "--no-dist-autotune", "--independent-steps", "0", "--cpu-sample-n", "512", "--small-n", "", "--stall", "6"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["launch"]["failed_attempts"] == []
    assert out["cpu_baseline"]["value"] > 0


def test_bench_two_ranks_autotuned_panel_width():
    """The default N > 1 run picks the panel width (and, under RCCL, the exchange) by measurement during warm-up: the
    timed steps run the chosen configuration and the line says which."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--force-device", "0", "--npoints", "4096", "--dist-nb", "256", "--num-new-throughput", "512",
           "--dist-timeout", "200", "--independent-steps", "0", "--no-roofline", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    out = _last_json(p.stdout)
    dd = out["distributed"]
    tune = dd["autotune"]
    assert set(tune["candidates_ms"]) == {"nb=256,lookahead=2", "nb=512,lookahead=2", "nb=128,lookahead=2"}
    best = min(tune["candidates_ms"], key=tune["candidates_ms"].get)
    assert "nb=%d," % dd["nb"] in best and tune["chosen"]["nb"] == dd["nb"] and tune["exchange_s"] == {}
    assert dd["parity_rel_err_vs_one_gpu"] <= 1e-9 and out["value"] > 0
    assert "nb=%d" % dd["nb"] in out["config"]["parallelism"]


def test_bench_cfg5_side_line_two_ranks():
    """`bench.py --workload cfg5 --gpus 2`: BASELINE configs[4] with the data points sharded over the ranks (started by
    bench.py itself, gloo, both ranks on the one GPU) -- same outputs as the one-rank run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    outs = []
    for gpus in (1, 2):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg5", "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
               "--backend", "gloo", "--force-device", "0", "--cfg5-m", "512", "--cfg5-n", "60000"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
        assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
        outs.append(_last_json(p.stdout))
    one, two = outs
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["unit"] == "points/s" and two["value"] > 0
    assert "configs[4]" in two["config"]["workload"] and two["scaling"] == "strong"
    assert abs(two["svgp_bound"]["elbo"] - one["svgp_bound"]["elbo"]) <= 1e-11 * abs(one["svgp_bound"]["elbo"])
    assert abs(two["sgpr_bound"]["bound"] - one["sgpr_bound"]["bound"]) <= 1e-11 * abs(one["sgpr_bound"]["bound"])      # (gloo reduces the device buffer)
    assert abs(two["checksum"]["fmean_sum"] - one["checksum"]["fmean_sum"]) <= 1e-9 * max(1.0, abs(one["checksum"]["fmean_sum"]))


def test_bench_two_ranks_native_communicator_stand_in_transport():
    """`bench.py --gpus 2 --comm rccl`: the block-column run with the collectives issued by the library's own RCCL binding
    (csrc/comm_rccl.hip) instead of torch.distributed -- two real processes on the one GPU, the RCCL *transport* replaced by
    the shared-memory stand-in of tests/fake_rccl ($GPFLOWSLIM_RCCL_LIB); torch.distributed (gloo) only carries the unique id."""
    src = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib, src, "-lrt", "-Wl,-Bsymbolic"])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GPFLOWSLIM_RCCL_LIB"] = lib
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "rccl", "--steps", "2", "--warmup", "1", "--backend", "gloo",
           "--force-device", "0", "--npoints", "4096", "--dist-nb", "256", "--num-new-throughput", "512", "--dist-timeout", "200",
           "--no-dist-autotune", "--independent-steps", "0", "--no-roofline", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["launch"]["attempt"] == "rccl" and out["launch"]["tried"] == ["rccl"]
    dd = out["distributed"]
    assert dd["parity_rel_err_vs_one_gpu"] <= 1e-9 and dd["exchange"] == "scatter_allgather"
    assert "own RCCL binding" in out["config"]["parallelism"]
    n_panels = 4096 // 256
    rows = lambda j: 4096 + 128 - j * 256
    payload = sum(-(-(rows(j) * 256 + 2 * 2 * 128 * 128 + 4) // 2) * 2 for j in range(n_panels)) * 8
    assert abs(dd["payload_bytes_per_eval_all_ranks"] - 1.5 * payload) <= 1e-6 * payload      # scatter + all-gather: (P + 1) / P of a broadcast's bytes


def _fake_rccl_env():
    src = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib, src, "-lrt", "-Wl,-Bsymbolic"])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GPFLOWSLIM_RCCL_LIB"] = lib
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


SMALL = ["--steps", "2", "--warmup", "1", "--backend", "gloo", "--force-device", "0", "--npoints", "4096", "--dist-nb", "256",
         "--num-new-throughput", "512", "--dist-timeout", "200", "--no-dist-autotune", "--independent-steps", "0", "--no-roofline",
         "--no-cpu-baseline", "--small-n", ""]


@pytest.mark.parametrize("mode", ["error", "stall"])
def test_bench_falls_back_to_fresh_ranks_when_the_native_communicator_fails(mode):
    """The first attempt of a multi-GPU run is the library's own RCCL binding.  Here its transport (the stand-in) fails on
    rank 1 in the middle of the schedule -- an ncclSend that returns an error inside an open group, or one that never
    returns --: the GPU-free supervisor ends that attempt (rank processes killed), starts FRESH rank processes with
    torch.distributed as the carrier, and rank 0's one JSON line says so."""
    env = dict(_fake_rccl_env(), FAKE_RCCL_FAIL_RANK="1", FAKE_RCCL_FAIL_AFTER="5", FAKE_RCCL_FAIL_MODE=mode, FAKE_RCCL_TIMEOUT_S="8")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--attempts", "rccl,torch", "--stall", "25"] + SMALL
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    la = out["launch"]
    assert la["attempt"] == "torch" and la["index"] == 1 and [f["attempt"] for f in la["failed_attempts"]] == ["rccl"], la
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["distributed"]["parity_rel_err_vs_one_gpu"] <= 1e-9
    assert "attempt 'rccl' failed" in p.stderr


def test_bench_fallback_under_a_launcher():
    """The same with the ranks started by torch.distributed.run (the driver's command form): every rank's bench.py becomes
    the supervisor of its own rank process, the supervisors agree through the rendezvous directory, the second attempt runs."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(_fake_rccl_env(), FAKE_RCCL_FAIL_RANK="1", FAKE_RCCL_FAIL_AFTER="5", FAKE_RCCL_TIMEOUT_S="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--attempts", "rccl,torch", "--stall", "25"] + SMALL
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["launch"]["attempt"] == "torch" and [f["attempt"] for f in out["launch"]["failed_attempts"]] == ["rccl"]
    assert out["n_gpus"] == 2 and out["value"] > 0


def test_example_gpr_distributed_two_ranks():
    """examples/gpr_distributed.py under torch.distributed.run with two ranks (gloo, both on the one GPU): partitioned
    factorisation + streamed predictions, the printed numbers equal to the one-rank run's."""
    import re
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    vals = []
    for nproc in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "examples", "gpr_distributed.py"), "--npoints", "3000", "--dims", "4", "--num-test", "101",
               "--block", "256", "--backend", "gloo", "--force-device", "0"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
        m = re.search(r"log marginal likelihood (-?[\d.]+) .* rmse vs noise-free truth ([\d.]+)", p.stdout)
        assert m, p.stdout[-2000:]
        vals.append((float(m.group(1)), float(m.group(2))))
    assert abs(vals[0][0] - vals[1][0]) <= 1e-6 * abs(vals[0][0]) and abs(vals[0][1] - vals[1][1]) <= 1e-4
    assert vals[0][1] < 0.1                     # the GP has learnt the function
