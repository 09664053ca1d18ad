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


@pytest.mark.parametrize("partitioned", [True, False])
def test_block_column_factorisation_through_the_native_communicator(comm_handle, partitioned):
    """gpr_lml_distributed + predict_f_distributed with RcclComm as the communicator (exchange = gps_comm_exchange on the
    communicator's stream, wait = an event wait of the chain lane): LML and predictions against the oracle."""
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    from gpflowSlim.distributed import gpr_lml_distributed, predict_f_distributed
    h, comm = comm_handle
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
        h.comm_install_allreduce(buf.data_ptr(), buf.numel())
        try:
            hooked = h.sgpr(prog, Z, X, Y, 1e-6, 0.15, fitc=fitc)[0]
        finally:
            h.set_allreduce(None, 0, 0)
        assert abs(hooked - plain) <= 1e-12 * abs(plain)
        assert float(buf[-4 + 2]) == n                 # the reduced data-point count travelled through the buffer
