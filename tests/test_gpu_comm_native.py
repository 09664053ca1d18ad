"""The native RCCL communicator (csrc/comm_rccl.hip behind gpflowSlim.distributed.RcclComm) on the GPU box.  The box has ONE
GPU and RCCL refuses two ranks on one device, so what can run here is world size 1: library loading, communicator set-up,
every collective entry point and the stream / event plumbing around them (an exchange is enqueued on the communicator's own
stream between two events) -- and the whole block-column factorisation and the data-sharded sparse bound driven through it."""
import numpy as np
import pytest

import oracle.gp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def comm_handle():
    from gpflowSlim import _backend as be
    from gpflowSlim.distributed import RcclComm
    h = be.Handle(0)
    comm = RcclComm(h, 0, 1)
    yield h, comm
    comm.close()
    h.close()


def test_collectives_world_one(comm_handle):
    import torch
    from gpflowSlim import _backend as be
    h, comm = comm_handle
    assert be.comm_version() >= 20000 and comm.world == 1 and comm.rank == 0
    ref = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    for mode in ("broadcast", "scatter_allgather"):
        comm.mode = mode
        t = ref.clone()
        torch.cuda.synchronize()
        w = comm.exchange(t, 0)
        assert w.wait() is True
        torch.cuda.synchronize()
        assert torch.equal(t, ref)
    t = ref.clone()
    torch.cuda.synchronize()
    comm.all_reduce_sum(t)
    assert torch.equal(t, ref)                       # one rank: the sum is the operand
    rows = comm.all_gather_rows(np.arange(6.0).reshape(3, 2), [3])
    assert np.array_equal(rows, np.arange(6.0).reshape(3, 2))
    with pytest.raises(RuntimeError, match="already has a communicator"):
        h.comm_init(0, 1, be.comm_unique_id())


@pytest.mark.parametrize("native_schedule", [True, False])
@pytest.mark.parametrize("partitioned", [True, False])
def test_block_column_factorisation_through_the_native_communicator(comm_handle, partitioned, native_schedule):
    """gpr_lml_distributed + predict_f_distributed with RcclComm as the communicator (exchange = gps_comm_exchange on the
    communicator's stream, wait = an event wait of the chain lane): LML and predictions against the oracle."""
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    from gpflowSlim.distributed import gpr_lml_distributed, predict_f_distributed
    h, comm = comm_handle
    comm.native_schedule = native_schedule        # True: gps_dist_lml / gps_dist_predict (the schedule inside the library); False: the Python schedule
    n, d = 2500, 4
    X, Y, Xs = orc.synthetic_gpr_data(n, d, 40, seed=3)
    ls = np.linspace(0.9, 1.7, d)
    spec = {"type": "rbf", "variance": orc.constrained(1.2), "lengthscales": orc.constrained(ls), "input_dim": d}
    noise = orc.constrained(0.1)
    saved = be.get_handle()
    be.set_handle(h)                                  # the model API uses the process' default handle
    try:
        m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, variance=1.2, lengthscales=ls, ARD=True), obs_var=0.1)
        lml = gpr_lml_distributed(m, comm, nb=256, lookahead=2, partitioned=partitioned)
        ref = orc.gpr_lml(spec, X, Y, noise)
        assert abs(lml - ref) <= 1e-8 * abs(ref)
        mu, var = predict_f_distributed(m, Xs, comm)
        rmu, rvar = orc.gpr_predict(spec, X, Y, noise, Xs)
        assert np.abs(mu - rmu).max() <= 1e-8 * np.abs(rmu).max() and np.abs(var - rvar).max() <= 1e-8 * np.abs(rvar).max()
        assert comm.exchanges >= n // 256
    finally:
        be.set_handle(saved)


def test_sparse_bound_reduces_through_the_native_allreduce(comm_handle):
    """gps_sgpr / gps_fitc on a data shard with gps_comm_install_allreduce as the collective (a C callback into ncclAllReduce:
    no host language in the loop); one rank -> the bound of the whole data set."""
    import torch
    import gpflowSlim as gpf
    h, comm = comm_handle
    rng = np.random.default_rng(8)
    n, m, d, r = 3000, 200, 3, 2
    X = rng.standard_normal((n, d)); Y = np.cos(X[:, :1]) @ np.ones((1, r)) + 0.1 * rng.standard_normal((n, r))
    Z = X[:m].copy()
    prog = gpf.kernels.RBF(d, variance=1.3, lengthscales=1.1)._program(d)
    for fitc in (False, True):
        plain = h.sgpr(prog, Z, X, Y, 1e-6, 0.15, fitc=fitc)[0]
        buf = torch.zeros(h.allreduce_doubles(m, r), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()                       # (the fill runs on torch's stream, the library on its own)
        h.comm_install_allreduce(buf.data_ptr(), buf.numel())
        try:
            hooked = h.sgpr(prog, Z, X, Y, 1e-6, 0.15, fitc=fitc)[0]
        finally:
            h.set_allreduce(None, 0, 0)
        assert abs(hooked - plain) <= 1e-12 * abs(plain)
        assert float(buf[-4 + 2]) == n                 # the reduced data-point count travelled through the buffer


def test_two_ranks_through_a_stand_in_transport(tmp_path):
    """World size 2 for the native communicator on the one GPU of the box: two real processes, each with its own handle and
    communicator, and tests/fake_rccl -- a shared-memory stand-in for the RCCL *transport* behind the same API -- loaded through
    gps_comm_load(path).  Everything on our side of that API is the production code: gps_comm_exchange's chunking (scatter
    of P chunks + in-place all-gather, ragged end, plain broadcast), roots, slots, the communicator's stream and events,
    gps_comm_allreduce as the sparse models' collective, RcclComm.  Checked against the oracle and the one-GPU results."""
    import json
    import os
    import subprocess
    import sys
    import gpflowSlim as gpf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "fake_rccl", "fake_rccl.cpp")
    lib = os.path.join(root, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib, src, "-lrt", "-Wl,-Bsymbolic"])
    uid = str(tmp_path / "uid.bin")
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(2)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "_native_comm_worker.py"), str(r), "2", uid, lib, outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    res = [json.load(open(o)) for o in outs]
    # references
    n, d = 3000, 3
    X, Y, Xs = orc.synthetic_gpr_data(n, d, 45, seed=13)
    ls = np.linspace(0.9, 1.6, d)
    spec = {"type": "rbf", "variance": orc.constrained(1.2), "lengthscales": orc.constrained(ls), "input_dim": d}
    noise = orc.constrained(0.1)
    ref = orc.gpr_lml(spec, X, Y, noise)
    rmu, rvar = orc.gpr_predict(spec, X, Y, noise, Xs)
    for r in res:
        assert abs(r["lml_part_sag"] - ref) <= 1e-8 * abs(ref) and abs(r["lml_repl_bcast"] - ref) <= 1e-8 * abs(ref)
        assert np.abs(np.array(r["mu"]) - rmu).max() <= 1e-8 * np.abs(rmu).max()
        assert np.abs(np.array(r["var"]) - rvar).max() <= 1e-8 * np.abs(rvar).max()
        assert r["exchanges"] > 20 and r["bytes_sent"] > 0
    assert res[0]["lml_part_sag"] == res[1]["lml_part_sag"] and res[0]["lml_repl_bcast"] == res[1]["lml_repl_bcast"]
    # the schedule inside the library (gps_dist_lml) and the Python schedule issue the same launches in the same order
    assert all(r["lml_part_py"] == r["lml_part_sag"] and r["mu_py"] == r["mu"] for r in res)
    assert res[0]["mu"] == res[1]["mu"] and res[0]["cond_mean"] == res[1]["cond_mean"]
    Z = X[:150].copy()
    k2 = gpf.kernels.RBF(d, variance=1.3, lengthscales=1.1)
    sg = gpf.models.SGPR(X, Y, k2, Z=Z, obs_var=0.15)
    want = sg.compute_log_likelihood()
    assert all(abs(r["sgpr"] - want) <= 1e-12 * abs(want) for r in res) and res[0]["sgpr"] == res[1]["sgpr"]
    fm, _ = gpf.conditionals.conditional(Xs, Z, k2, np.sin(Z[:, :2]))
    assert np.abs(np.array(res[0]["cond_mean"]) - fm).max() <= 1e-9 * max(1.0, np.abs(fm).max())
    sv = gpf.models.SVGP(X, Y, k2, gpf.likelihoods.Gaussian(0.2), Z=Z, q_diag=True)
    wsv = sv.compute_log_likelihood()
    assert all(abs(r["svgp"] - wsv) <= 1e-12 * abs(wsv) for r in res) and res[0]["svgp"] == res[1]["svgp"]


def _fake_lib():
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "fake_rccl", "fake_rccl.cpp")
    lib = os.path.join(root, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib, src, "-lrt", "-Wl,-Bsymbolic"])
    return root, lib


def test_a_failing_send_aborts_the_communicator_and_leaves_the_handle_usable(tmp_path):
    """Rank 1's ncclSend fails in the middle of the schedule, between ncclGroupStart and ncclGroupEnd (stand-in transport with
    fault injection).  The group is closed all the same, the communicator aborted, gps_dist_lml leaves through its clean-up
    (the handle gets its own stream back): every rank reports an error instead of hanging, has no communicator afterwards,
    and its handle evaluates the ordinary single-GPU path correctly."""
    import json
    import os
    import subprocess
    import sys
    root, lib = _fake_lib()
    uid = str(tmp_path / "uid.bin")
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(2)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPS_WORKER_MODE="fault", FAKE_RCCL_FAIL_RANK="1", FAKE_RCCL_FAIL_AFTER="2",
               FAKE_RCCL_TIMEOUT_S="5")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "_native_comm_worker.py"), str(r), "2", uid, lib, outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    res = [json.load(open(o)) for o in outs]
    for r in res:
        assert r["error"] and "communicator aborted" in r["error"], r
        assert "no communicator" in r["after"], r
        assert abs(r["lml_after"] - r["lml_ref"]) <= 1e-8 * abs(r["lml_ref"])
    assert "gps_comm_exchange: nccl" in res[1]["error"]          # (the failing call itself, or the group end that reports it)
