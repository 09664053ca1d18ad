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
@pytest.mark.parametrize("partitioned", [True, False])
def test_single_rank_distributed_equals_fused_path(handle, n, nb, lookahead, partitioned):
    import gpflowSlim as gpf
    from gpflowSlim.distributed import SingleComm, gpr_lml_distributed, predict_f_distributed
    X, Y, ls, spec = _data(n, 4)
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(4, variance=1.1, lengthscales=ls, ARD=True), obs_var=0.1)
    ref = orc.gpr_lml(spec, X, Y, orc.constrained(0.1))
    got = gpr_lml_distributed(m, SingleComm(), nb=nb, lookahead=lookahead, partitioned=partitioned)
    assert abs(got - ref) <= 1e-8 * abs(ref)
    Xs = np.random.default_rng(1).standard_normal((33, 4))
    rmu, rvar = orc.gpr_predict(spec, X, Y, orc.constrained(0.1), Xs)
    # predictions from the distributed factor: streamed panels (partitioned) or the replicated factor (warm predict_f)
    mu2, var2 = predict_f_distributed(m, Xs, SingleComm())
    assert np.abs(mu2 - rmu).max() <= 1e-8 * np.abs(rmu).max() and np.abs(var2 - rvar).max() <= 1e-8 * np.abs(rvar).max()
    if not partitioned:
        m.reuse_factor = True
        mu, var = m.predict_f(Xs)
        assert np.array_equal(mu2, mu) and np.array_equal(var2, var)
        m.reuse_factor = False
    # and the ordinary fused path still works on the same handle afterwards
    assert abs(m.compute_log_likelihood() - ref) <= 1e-8 * abs(ref)
    if partitioned:
        # ... and has overwritten the distributed factor.  The claim "a partitioned factor is resident" lives on the handle
        # and went with it: the prediction neither reads stale panels nor refuses, it factorises again
        assert handle.dist_state is None
        mu3, var3 = predict_f_distributed(m, Xs, SingleComm())
        assert np.abs(mu3 - rmu).max() <= 1e-8 * np.abs(rmu).max() and np.abs(var3 - rvar).max() <= 1e-8 * np.abs(rvar).max()


def test_partitioned_factor_belongs_to_the_handle_not_the_model(handle):
    """Two models that share X and the handle: B's distributed evaluation leaves B's partitioned factor behind; a streamed
    prediction for A must not use it (it factorises again), and B's own prediction still streams."""
    import gpflowSlim as gpf
    from gpflowSlim.distributed import SingleComm, gpr_lml_distributed, predict_f_distributed
    X, Y, ls, spec = _data(700, 4)
    YB = 2.0 * Y + 0.3
    a = gpf.models.GPR(X, Y, gpf.kernels.RBF(4, variance=1.1, lengthscales=ls, ARD=True), obs_var=0.1)
    b = gpf.models.GPR(X, YB, gpf.kernels.RBF(4, variance=1.1, lengthscales=ls, ARD=True), obs_var=0.1)
    Xs = np.random.default_rng(2).standard_normal((21, 4))
    gpr_lml_distributed(a, SingleComm(), nb=256)
    gpr_lml_distributed(b, SingleComm(), nb=256)
    assert handle.dist_state is not None and handle.dist_state["key"] == b._state_key()
    ra, rva = orc.gpr_predict(spec, X, Y, orc.constrained(0.1), Xs)
    rb, rvb = orc.gpr_predict(spec, X, YB, orc.constrained(0.1), Xs)
    mb, vb = predict_f_distributed(b, Xs, SingleComm())
    assert np.abs(mb - rb).max() <= 1e-8 * np.abs(rb).max() and np.abs(vb - rvb).max() <= 1e-8 * np.abs(rvb).max()
    ma, va = predict_f_distributed(a, Xs, SingleComm())
    assert np.abs(ma - ra).max() <= 1e-8 * np.abs(ra).max() and np.abs(va - rva).max() <= 1e-8 * np.abs(rva).max()


def _run_virtual_ranks(world, X, Y, prog, noise, nb, lookahead=2, r_out=None, partitioned=None, Xnew=None, extra=None):
    """P host threads = P virtual ranks on one GPU; returns (per-rank results or exceptions).  `extra` (a dict) receives
    per-rank device bytes and, with Xnew, each rank's share of the streamed prediction."""
    import torch
    from gpflowSlim import _backend as be
    from gpflowSlim.distributed import HipPanelOps, block_column_schedule, predict_streamed
    shared = {"barrier": threading.Barrier(world), "slot": None}
    out, errs = [None] * world, [None] * world

    def run(rank):
        h = None
        try:
            torch.cuda.set_device(0)
            h = be.Handle(0)
            h.gpr_set_data(X, ("sim", rank))
            comm = ThreadComm(rank, world, shared)
            with HipPanelOps(h, prog, noise, Y, world, rank, nb, partitioned=partitioned) as ops:
                block_column_schedule(ops, comm, ops.n_panels, lookahead=lookahead)
                out[rank] = ops.finish()
                if extra is not None:
                    extra.setdefault("bytes", {})[rank] = h.device_bytes()
                    extra.setdefault("n_bufs", {})[rank] = ops.n_bufs
                if Xnew is not None:
                    lo, hi = Xnew.shape[0] * rank // world, Xnew.shape[0] * (rank + 1) // world
                    extra.setdefault("pred", {})[rank] = predict_streamed(h, prog, Xnew[lo:hi], comm, ops.n_panels, ops.bufs,
                                                                          ops.nparts, Y.shape[1])
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


@pytest.mark.parametrize("partitioned", [True, False])
def test_failure_in_the_middle_of_the_schedule_leaves_the_handle_usable(partitioned):
    """An exchange that raises half-way through the schedule (look-ahead 2: trailing updates are in flight on the BULK lane):
    leaving HipPanelOps drains BOTH lanes before the handle gets its own stream back, so that the next evaluation on the
    handle -- which rebuilds K in the same buffer -- cannot race with them: it returns the ordinary result, repeatedly."""
    import torch
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    from gpflowSlim.distributed import HipPanelOps, SingleComm, block_column_schedule
    n, d, nb = 6144, 4, 256
    X, Y, ls, spec = _data(n, d, seed=9)
    prog = gpf.kernels.RBF(d, variance=1.1, lengthscales=ls, ARD=True)._program(d)
    noise = float(orc.constrained(0.1))

    class Breaks(SingleComm):
        calls = 0

        def exchange(self, tensor, src):
            Breaks.calls += 1
            if Breaks.calls == 9:
                raise RuntimeError("link down (injected)")
            return SingleComm.exchange(self, tensor, src)
    h = be.Handle(0)
    try:
        h.gpr_set_data(X, ("midfail", partitioned))
        ref = h.gpr_lml(prog, noise, Y)
        for attempt in range(3):
            Breaks.calls = 0
            with pytest.raises(RuntimeError, match="injected"):
                with HipPanelOps(h, prog, noise, Y, 1, 0, nb, partitioned=partitioned) as ops:
                    block_column_schedule(ops, Breaks(), ops.n_panels, lookahead=2)
            assert h.gpr_lml(prog, noise, Y) == ref            # bit for bit: nothing of the aborted run was still writing
        torch.cuda.synchronize()
    finally:
        h.close()


@pytest.mark.parametrize("world,n,nb", [(2, 1024, 128), (3, 1400, 256), (4, 3000, 256)])
def test_partitioned_factor_streams_predictions(world, n, nb):
    """Partitioned storage (the default): after the factorisation no rank holds another rank's block columns, and
    predict_f streams the panels once more (gps_dist_solve_*: test points sharded over the ranks, alpha picked up from the
    augmented rows of each panel) -- mean and variance against the oracle (models/gpr.py:119-131)."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n)
    d, r = 4, 2
    X = rng.standard_normal((n, d)); Y = np.sin(X @ rng.standard_normal((d, r))) + 0.1 * rng.standard_normal((n, r))
    Xs = rng.standard_normal((world * 40 + 3, d))
    ls = np.linspace(0.9, 1.8, d)
    kern = gpf.kernels.RBF(d, variance=1.2, lengthscales=ls, ARD=True)
    spec = {"type": "rbf", "variance": orc.constrained(1.2), "lengthscales": orc.constrained(ls), "input_dim": d}
    noise = float(orc.constrained(0.15))
    extra = {}
    out, errs = _run_virtual_ranks(world, X, Y, kern._program(d), noise, nb, 2, partitioned=True, Xnew=Xs, extra=extra)
    assert not any(errs), errs
    ref = orc.gpr_lml(spec, X, Y, noise)
    assert len(set(out)) == 1 and abs(out[0] - ref) <= 1e-8 * abs(ref)
    assert set(extra["n_bufs"].values()) == {3}
    mu = np.concatenate([extra["pred"][k][0] for k in range(world)], axis=0)
    var = np.concatenate([extra["pred"][k][1] for k in range(world)], axis=0)
    rmu, rvar = orc.gpr_predict(spec, X, Y, noise, Xs)
    assert np.abs(mu - rmu).max() <= 1e-8 * np.abs(rmu).max()
    assert np.abs(var - rvar[:, 0]).max() <= 1e-8 * np.abs(rvar).max()


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
    extra = {}
    out, errs = _run_virtual_ranks(world, X, Y, kern._program(d), noise, nb, 2, extra=extra)
    assert not any(errs), errs
    assert len(set(out)) == 1, out
    assert abs(out[0] - ref) <= 1e-9 * abs(ref), (out[0], ref)
    # partitioned storage (the default): a rank holds its own block columns, not the matrix -- SURVEY 8e "1.07 GB of K/L
    # per GPU": 8 N^2 / P plus O(N nb) of block inverses, features and scalars (the comm buffers, 3 x 8 (N + 128) nb, are
    # the caller's)
    bound = 8 * n * n / world + 64 * n * nb
    assert max(extra["bytes"].values()) <= bound, (extra["bytes"], bound)
    if nb == 512:
        # the replicated mode is still there on request, same result bit for bit
        out_r, errs_r = _run_virtual_ranks(world, X, Y, kern._program(d), noise, nb, 2, partitioned=False)
        assert not any(errs_r) and set(out_r) == set(out)
