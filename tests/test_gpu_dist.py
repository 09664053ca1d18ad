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

    bytes_sent = 0
    exchanges = 0

    def exchange(self, tensor, src):
        return self.broadcast(tensor, src, True)

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
@pytest.mark.parametrize("lookahead", [0, 1, 2, 3])
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


def _run_virtual_ranks(world, X, Y, prog, noise, nb, lookahead=2, r_out=None):
    """P host threads = P virtual ranks on one GPU; returns (per-rank results or exceptions)."""
    import torch
    from gpflowSlim import _backend as be
    from gpflowSlim.distributed import HipPanelOps, block_column_schedule
    shared = {"barrier": threading.Barrier(world), "slot": None}
    out, errs = [None] * world, [None] * world

    def run(rank):
        h = None
        try:
            torch.cuda.set_device(0)
            h = be.Handle(0)
            h.gpr_set_data(X, ("sim", rank))
            with HipPanelOps(h, prog, noise, Y, world, rank, nb) as ops:
                block_column_schedule(ops, ThreadComm(rank, world, shared), ops.n_panels, lookahead=lookahead)
                out[rank] = ops.finish()
        except be.NotPositiveDefiniteError as e:
            errs[rank] = e
        except Exception as e:        # pragma: no cover
            errs[rank] = e
            shared["barrier"].abort()
        finally:
            if h is not None:
                h.close()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    return out, errs


@pytest.mark.parametrize("world,n,nb,lookahead", [(2, 1024, 128, 2), (3, 1400, 256, 1), (4, 2048, 256, 3), (8, 4096, 256, 2),
                                                  (2, 1024, 512, 0)])
def test_virtual_ranks_on_one_gpu(world, n, nb, lookahead):
    import gpflowSlim as gpf
    X, Y, ls, spec = _data(n, 5, seed=3)
    kern = gpf.kernels.Matern52(5, variance=1.1, lengthscales=np.linspace(0.9, 2.2, 5), ARD=True)
    spec = {"type": "matern52", "variance": orc.constrained(1.1), "lengthscales": orc.constrained(np.linspace(0.9, 2.2, 5)), "input_dim": 5}
    prog = kern._program(5)
    noise = float(orc.constrained(0.1))
    ref = orc.gpr_lml(spec, X, Y, noise)
    out, errs = _run_virtual_ranks(world, X, Y, prog, noise, nb, lookahead)
    assert not any(errs), errs
    for rank in range(world):
        assert out[rank] is not None and abs(out[rank] - ref) <= 1e-8 * abs(ref), (rank, out[rank], ref)
    assert len(set(out)) == 1, "every rank must hold bitwise the same factor / result"


def test_virtual_ranks_two_outputs_and_warm_predict():
    """R = 2 outputs through the augmented rows; afterwards the replicated factor and alpha serve predict_f."""
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    rng = np.random.default_rng(4)
    n, d = 900, 3
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, 2))) + 0.1 * rng.standard_normal((n, 2))
    kern, spec = gpf.kernels.RBF(d, variance=1.2, lengthscales=1.4), {"type": "rbf", "variance": orc.constrained(1.2), "lengthscales": orc.constrained(1.4), "input_dim": d}
    noise = float(orc.constrained(0.2))
    ref = orc.gpr_lml(spec, X, Y, noise)
    out, errs = _run_virtual_ranks(3, X, Y, kern._program(d), noise, 128, 2)
    assert not any(errs), errs
    assert len(set(out)) == 1 and abs(out[0] - ref) <= 1e-8 * abs(ref)


def test_not_positive_definite_is_collective():
    """A failing pivot in the LAST panel: the info word travels in the tail of every panel message, so every rank --
    not just the owner of that panel -- raises the same NotPositiveDefiniteError (and none keeps a resident factor)."""
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    n, d, nb, world = 1024, 6, 128, 3
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, d))                   # far apart against the length-scale: K is close to the identity ...
    X[-1] = X[-2]                                     # ... except for a duplicate point in the last panel
    Y = rng.standard_normal((n, 1))
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=0.3)
    out, errs = _run_virtual_ranks(world, X, Y, kern._program(d), -1e-9, nb, 2)       # "noise" that makes the last pivot negative
    assert all(isinstance(e, be.NotPositiveDefiniteError) for e in errs), errs
    assert len({str(e) for e in errs}) == 1, [str(e) for e in errs]
    assert "1024" in str(errs[0])


@pytest.mark.parametrize("nb", [512, 1024])
def test_full_size_config3_virtual_ranks(nb):
    """BASELINE configs[2] shape -- N = 32768, D = 8, block-column Cholesky over 8 ranks -- with the 8 ranks as threads
    on the one GPU of the box (69 GB of the 288 GB HBM): every rank returns bit for bit the same LML, equal to the
    fused single-GPU evaluation to 1e-9."""
    import gpflowSlim as gpf
    n, d, world = 32768, 8, 8
    X, Y, _ = orc.synthetic_gpr_data(n, d, 0)
    ls = np.sqrt(d) * np.ones(d)
    kern = gpf.kernels.RBF(d, variance=1.0, lengthscales=ls, ARD=True)
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    ref = m.compute_log_likelihood()
    noise = float(np.squeeze(m.likelihood.variance))
    out, errs = _run_virtual_ranks(world, X, Y, kern._program(d), noise, nb, 2)
    assert not any(errs), errs
    assert len(set(out)) == 1, out
    assert abs(out[0] - ref) <= 1e-9 * abs(ref), (out[0], ref)
