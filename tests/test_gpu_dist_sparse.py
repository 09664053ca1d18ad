"""BASELINE config 5 with the data points sharded over ranks (gpflowSlim/distributed_sparse.py; SURVEY 8e "independent over
RHS columns").  The box has ONE GPU, so P > 1 is simulated as in tests/test_gpu_dist.py: P host threads = P virtual ranks,
each with its own library handle on the same device; the collectives are thread barriers + host / device sums in rank
order.  Everything except the collective itself is the production path (C ABI incl. gps_set_allreduce and the
"svgp_kl_weight" option).  Oracle for the sharded forms = the single-GPU result, to 1e-12 relative."""
import threading

import numpy as np
import pytest

import oracle.gp_oracle as orc

pytestmark = pytest.mark.gpu


class ThreadComm(object):
    bytes_sent = 0

    def __init__(self, rank, world, shared):
        self.rank, self.world, self.sh = rank, world, shared

    def all_gather_rows(self, local, counts):
        sh = self.sh
        sh["rows"][self.rank] = np.array(local, copy=True)
        sh["barrier"].wait()
        out = np.concatenate([sh["rows"][p][: counts[p]] for p in range(self.world)], axis=0)
        sh["barrier"].wait()
        return out

    def all_reduce_sum(self, tensor):
        import torch
        sh = self.sh
        torch.cuda.synchronize()
        sh["tens"][self.rank] = tensor
        sh["barrier"].wait()
        tot = sh["tens"][0].clone()
        for p in range(1, self.world):
            tot += sh["tens"][p]
        torch.cuda.synchronize()
        sh["barrier"].wait()                         # everyone has read every operand
        tensor.copy_(tot)
        torch.cuda.synchronize()
        sh["barrier"].wait()
        return tensor


def _virtual(world, fn):
    """Run fn(comm, handle) on `world` virtual ranks; returns the per-rank results (raises the first error)."""
    import torch
    from gpflowSlim import _backend as be
    shared = {"barrier": threading.Barrier(world), "rows": [None] * world, "tens": [None] * world}
    out, errs = [None] * world, [None] * world

    def run(rank):
        h = None
        try:
            torch.cuda.set_device(0)
            h = be.Handle(0)
            out[rank] = fn(ThreadComm(rank, world, shared), h)
        except BaseException as e:
            errs[rank] = e
            shared["barrier"].abort()
        finally:
            if h is not None:
                h.close()
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    assert all(e is None for e in errs), errs
    return out


def _rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("world,n,m,white", [(2, 3001, 200, True), (3, 5000, 333, False), (4, 1_000_000, 4096, True)])
def test_conditional_sharded_over_test_points(handle, world, n, m, white):
    """conditionals.py:24-119 with the N test points split over P ranks (config 5 at its stated size for P = 4): equal to the
    one-GPU call; a slice of it equal to the oracle."""
    import gpflowSlim as gpf
    from gpflowSlim.distributed_sparse import conditional_distributed
    rng = np.random.default_rng(n)
    d, k = 4, 2
    Xnew = rng.standard_normal((n, d)); Z = rng.standard_normal((m, d)); f = rng.standard_normal((m, k))
    q_sqrt = 0.3 + rng.random((m, k))
    kern = gpf.kernels.RBF(d, variance=1.2, lengthscales=1.4)
    ref = handle.conditional(kern._program(d), Z, Xnew, f, 1e-6, q_sqrt=q_sqrt, white=white)
    res = _virtual(world, lambda comm, h: conditional_distributed(Xnew, Z, kern, f, comm=comm, handle=h, q_sqrt=q_sqrt, white=white))
    # a shard is a different launch shape (other tile sizes, other summation order): rounding differences of the solves
    # against the jittered Kuu scale with its conditioning -- two backward-stable solves agree to ~2 eps cond (solve_tol of
    # tests/test_gpu_parity.py), never asked tighter than 1e-12
    spec = {"type": "rbf", "variance": orc.constrained(1.2), "lengthscales": orc.constrained(1.4), "input_dim": d}
    ev = np.linalg.eigvalsh(orc.K(spec, Z) + 1e-6 * np.eye(m))
    cond = ev[-1] / ev[0]
    same = max(1e-12, 2 * np.finfo(float).eps * cond)
    for fm, fv in res:
        assert fm.shape == (n, k) and fv.shape == (n, k)
        assert _rel(fm, ref[0]) <= same and _rel(fv, ref[1]) <= same, (cond, _rel(fm, ref[0]), _rel(fv, ref[1]))
        assert np.array_equal(fm, res[0][0]) and np.array_equal(fv, res[0][1])
    if m <= 400:
        ofm, ofv = orc.conditional(Xnew[:500], Z, spec, f, q_sqrt=q_sqrt, white=white)
        tol = max(1e-8, 2 * np.finfo(float).eps * cond)
        assert _rel(res[0][0][:500], ofm) <= tol and _rel(res[0][1][:500], ofv) <= tol


@pytest.mark.parametrize("world,n,m,whiten,q_diag", [(2, 4001, 150, True, False), (3, 6000, 260, False, True),
                                                     (4, 200_000, 1024, True, False)])
def test_svgp_bound_and_gradient_sharded_over_data(handle, world, n, m, whiten, q_diag):
    """models/svgp.py:108-125 with the data points split over P ranks: rank p evaluates scale * sum_shard var_exp - KL / P
    (option "svgp_kl_weight"), the ranks add up -- bound and every gradient (kernel, noise, q_mu, q_sqrt, inducing inputs)
    equal to the one-GPU evaluation, identical on all ranks."""
    import gpflowSlim as gpf
    from gpflowSlim.distributed_sparse import svgp_bound_distributed, svgp_bound_and_gradients_distributed
    rng = np.random.default_rng(m)
    d, k = 3, 2
    X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) @ np.ones((1, k)) + 0.1 * rng.standard_normal((n, k))
    Z = X[rng.choice(n, m, replace=False)].copy()

    def make():
        kern = gpf.kernels.Matern52(d, variance=1.1, lengthscales=np.linspace(0.9, 1.5, d), ARD=True)
        mod = gpf.models.SVGP(X, Y, kern, gpf.likelihoods.Gaussian(0.2), Z=Z, q_diag=q_diag, whiten=whiten, num_data=3 * n,
                              train_inducing=True)
        r2 = np.random.default_rng(5)
        mod._q_mu.assign(0.2 * r2.standard_normal((m, k)))
        if q_diag:
            mod._q_sqrt.assign(0.4 + r2.random((m, k)))
        else:
            mod._q_sqrt.assign(np.stack([np.tril(0.05 * r2.standard_normal((m, m))) + 0.7 * np.eye(m) for _ in range(k)], axis=2))
        return mod
    mod = make()
    ref_b = mod.compute_log_likelihood()
    ref_b2, ref_g = mod.compute_log_likelihood_and_gradients()
    assert abs(ref_b2 - ref_b) <= 1e-12 * abs(ref_b)

    def run(comm, h):
        mine = make()                                             # (every virtual rank its own model objects)
        b = svgp_bound_distributed(mine, comm, h)
        b2, g = svgp_bound_and_gradients_distributed(mine, comm, h)
        return b, b2, [np.asarray(x) for _, x in g]
    res = _virtual(world, run)
    for b, b2, g in res:
        assert abs(b - ref_b) <= 1e-12 * abs(ref_b) and abs(b2 - ref_b) <= 1e-12 * abs(ref_b)
        assert b == res[0][0] and b2 == res[0][1]
        assert len(g) == len(ref_g)
        for got, (_, want), g0 in zip(g, ref_g, res[0][2]):
            assert got.shape == np.asarray(want).shape
            assert _rel(got, want) <= 1e-10, _rel(got, want)
            assert np.array_equal(got, g0)
    assert handle.svgp_elbo(mod.kern._program(d), Z, X[:100], Y[:100], mod.q_mu, mod.q_sqrt, 1e-6, 0.2)[0] != 0.0   # weight restored: plain call still works


@pytest.mark.parametrize("world,n,m,fitc", [(2, 3000, 128, False), (3, 4000, 200, True), (4, 100_000, 512, False),
                                            (4, 100_000, 512, True)])
def test_sgpr_and_fitc_sharded_over_data(handle, world, n, m, fitc):
    """models/sgpr.py:121-153 / 252-291 with the data points split over P ranks and ONE device all-reduce of
    [A A^T | A err | diag | scalars] inside gps_sgpr / gps_fitc (gps_set_allreduce): bound and predictions equal to the
    one-GPU model's, identical on all ranks; the gradient entry points refuse to run on shards."""
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    from gpflowSlim.distributed_sparse import sparse_bound_distributed, sparse_predict_distributed, DeviceAllReduce
    rng = np.random.default_rng(n + m)
    d, r = 3, 2
    X = rng.standard_normal((n, d)); Y = np.cos(X[:, :1]) @ np.ones((1, r)) + 0.1 * rng.standard_normal((n, r))
    Z = X[rng.choice(n, m, replace=False)].copy()
    Xs = rng.standard_normal((77, d))
    cls = gpf.models.GPRFITC if fitc else gpf.models.SGPR

    def make():
        return cls(X, Y, gpf.kernels.RBF(d, variance=1.3, lengthscales=1.1), Z=Z, obs_var=0.15)
    mod = make()
    ref_b = mod.compute_log_likelihood()
    ref_mu, ref_var = mod.predict_f(Xs)
    if n <= 5000:
        spec = {"type": "rbf", "variance": orc.constrained(1.3), "lengthscales": orc.constrained(1.1), "input_dim": d}
        ob = (orc.fitc_lml if fitc else orc.sgpr_bound)(spec, X, Y, Z, orc.constrained(0.15))
        assert abs(ref_b - ob) <= 1e-7 * abs(ob)

    def run(comm, h):
        mine = make()
        b = sparse_bound_distributed(mine, comm, h)
        mu, var = sparse_predict_distributed(mine, Xs, comm, h)
        refused = False
        with DeviceAllReduce(h, comm, m, r):
            try:
                h.sgpr_grad(mine.kern._program(d), Z, X[:500], Y[:500], 1e-6, 0.15, fitc=fitc)
            except RuntimeError as e:
                refused = "sharded" in str(e)
        # the collective is gone afterwards: the plain call on this handle is the whole-data model again
        b_plain = mine._call(handle=h)[0] if n <= 5000 else None
        return b, mu, var, refused, b_plain
    res = _virtual(world, run)
    for b, mu, var, refused, b_plain in res:
        assert abs(b - ref_b) <= 1e-12 * abs(ref_b), (b, ref_b)
        assert b == res[0][0]
        assert mu.shape == ref_mu.shape and var.shape == ref_var.shape
        assert _rel(mu, ref_mu) <= 1e-10 and _rel(var, ref_var) <= 1e-10
        assert np.array_equal(mu, res[0][1]) and np.array_equal(var, res[0][2])
        assert refused
        assert b_plain is None or abs(b_plain - ref_b) <= 1e-12 * abs(ref_b)
