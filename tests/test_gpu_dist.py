"""Block-column distributed factorisation on the GPU box.  The box has ONE GPU, so P > 1 is simulated:
P host threads = P virtual ranks, each with its own library handle (its own full set of device
buffers) on the same device, exchanging panels through a thread barrier + device copy instead of
RCCL.  Everything except the collective itself is the production code path (C ABI gps_dist_*,
gpflowSlim.distributed.block_column_schedule, look-ahead)."""
import threading

import numpy as np
import pytest

import oracle.gp_oracle as orc

pytestmark = pytest.mark.gpu


class _Done(object):
    def wait(self):
        return True


class ThreadComm(object):
    def __init__(self, rank, world, shared):
        self.rank, self.world, self.shared = rank, world, shared

    def broadcast(self, tensor, src, async_op):
        import torch
        sh = self.shared
        torch.cuda.synchronize()
        if self.rank == src:
            sh["slot"] = tensor
        sh["barrier"].wait()
        if self.rank != src:
            assert sh["slot"].numel() == tensor.numel(), "message size differs between ranks"
            tensor.copy_(sh["slot"])
            torch.cuda.synchronize()
        sh["barrier"].wait()
        return _Done()


def _data(n, d, seed=0):
    X, Y, _ = orc.synthetic_gpr_data(n, d, 0, seed=seed)
    ls = np.linspace(0.9, 2.2, d)
    spec = {"type": "rbf", "variance": orc.constrained(1.1), "lengthscales": orc.constrained(ls), "input_dim": d}
    return X, Y, ls, spec


@pytest.mark.parametrize("n,nb", [(300, 128), (1000, 256), (2048, 512), (1500, 512)])
@pytest.mark.parametrize("lookahead", [True, False])
def test_single_rank_distributed_equals_fused_path(handle, n, nb, lookahead):
    import gpflowSlim as gpf
    from gpflowSlim.distributed import SingleComm, gpr_lml_distributed
    X, Y, ls, spec = _data(n, 4)
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(4, variance=1.1, lengthscales=ls, ARD=True), obs_var=0.1)
    ref = orc.gpr_lml(spec, X, Y, orc.constrained(0.1))
    got = gpr_lml_distributed(m, SingleComm(), nb=nb, lookahead=lookahead)
    assert abs(got - ref) <= 1e-8 * abs(ref)
    # the replicated factor serves predict_f (warm) afterwards
    m.reuse_factor = True
    Xs = np.random.default_rng(1).standard_normal((33, 4))
    mu, var = m.predict_f(Xs)
    rmu, rvar = orc.gpr_predict(spec, X, Y, orc.constrained(0.1), Xs)
    assert np.abs(mu - rmu).max() <= 1e-8 * np.abs(rmu).max() and np.abs(var - rvar).max() <= 1e-8 * np.abs(rvar).max()
    from gpflowSlim.distributed import predict_f_distributed
    mu2, var2 = predict_f_distributed(m, Xs, SingleComm())
    assert np.array_equal(mu2, mu) and np.array_equal(var2, var)
    # and the ordinary fused path still works on the same handle afterwards
    assert abs(m.compute_log_likelihood() - ref) <= 1e-8 * abs(ref)


@pytest.mark.parametrize("world,n,nb", [(2, 1024, 128), (3, 1400, 256), (4, 2048, 256), (8, 4096, 256)])
def test_virtual_ranks_on_one_gpu(world, n, nb):
    import torch
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    from gpflowSlim.distributed import HipPanelOps, block_column_schedule
    X, Y, ls, spec = _data(n, 5, seed=3)
    kern = gpf.kernels.Matern52(5, variance=1.1, lengthscales=ls if False else np.linspace(0.9, 2.2, 5), ARD=True)
    spec = {"type": "matern52", "variance": orc.constrained(1.1), "lengthscales": orc.constrained(np.linspace(0.9, 2.2, 5)), "input_dim": 5}
    prog = kern._program(5)
    noise = float(orc.constrained(0.1))
    ref = orc.gpr_lml(spec, X, Y, noise)
    shared = {"barrier": threading.Barrier(world), "slot": None}
    out, errs = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            h = be.Handle(0)
            h.gpr_set_data(X, ("sim", rank))
            ops = HipPanelOps(h, prog, noise, Y, world, rank, nb)
            block_column_schedule(ops, ThreadComm(rank, world, shared), ops.n_panels, lookahead=True)
            out[rank] = ops.finish()
            h.close()
        except Exception as e:        # pragma: no cover
            errs.append((rank, repr(e)))
            shared["barrier"].abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errs, errs
    for rank in range(world):
        assert out[rank] is not None and abs(out[rank] - ref) <= 1e-8 * abs(ref), (rank, out[rank], ref)
    assert len(set(out)) == 1, "every rank must hold bitwise the same factor / result"
