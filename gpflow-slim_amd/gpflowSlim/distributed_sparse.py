"""The inducing-point models with the DATA POINTS sharded over ranks (BASELINE config 5: M = 4096, N = 10^6).

There is no reference counterpart (GPflow-Slim is single-device); the oracle for this module is the
single-GPU result.  SURVEY 8(e): "trsm with many RHS (predict, cfg5): independent over RHS columns" -- every O(M^2 N)
piece of conditionals.py:87-103, models/svgp.py:108-125 and models/sgpr.py:138-153 / 241-290 is a map over data points
followed by a sum, so rank p works on rows [lo_p, hi_p) of the data, Kuu (M x M: 7 ms at M = 4096) is factored by every
rank, and the ranks exchange

  conditional      nothing but the final gather of the [N*, K] outputs            (conditionals.py:87-119)
  SVGP bound       one vector: scale * sum_shard var_exp - KL / P and, for the gradient, every parameter's slot --
                   the bound and all its gradients are linear in the per-rank terms  (models/svgp.py:108-125)
  SGPR / FITC      one in-place device all-reduce of [A A^T | A err | diag | sum err^2, sum log nu, n] in the middle of
                   gps_sgpr / gps_fitc (C ABI gps_set_allreduce)                    (models/sgpr.py:138-153, 241-290)

Vector sums are made bit-identical on every rank by gathering the per-rank vectors and adding them in rank order.
`comm` is a gpflowSlim.distributed communicator (TorchComm: RCCL or gloo; SingleComm) -- needs rank, world,
all_gather_rows(local, counts) and, for SGPR / FITC, all_reduce_sum(device_tensor).
"""
import numpy as np

from . import _backend as be
from ._settings import settings
from .distributed import _default_comm


def shard_bounds(n, world):
    """Row ranges of the P shards of n data points: [b[p], b[p + 1]) -- contiguous, sizes differ by at most one."""
    return [(int(n) * p) // int(world) for p in range(int(world) + 1)]


def ordered_sum(comm, vec):
    """Sum of the ranks' vectors, added in rank order on every rank (bit-identical everywhere, whatever the collective
    library does inside an all-reduce)."""
    vec = np.ascontiguousarray(vec, dtype=np.float64).reshape(1, -1)
    if comm.world == 1:
        return vec[0].copy()
    rows = comm.all_gather_rows(vec, [1] * comm.world)
    out = rows[0].copy()
    for p in range(1, comm.world):
        out += rows[p]
    return out


def conditional_distributed(Xnew, X, kern, f, *, comm=None, handle=None, q_sqrt=None, white=False):
    """conditionals.conditional (conditionals.py:24-66, full_cov=False) with the test points sharded over the ranks:
    every rank passes the same arguments and gets the full (fmean [N*, K], fvar [N*, K]) back."""
    comm = comm or _default_comm()
    h = handle or be.get_handle()
    Xnew = np.ascontiguousarray(Xnew, dtype=settings.float_type)
    X = np.ascontiguousarray(X, dtype=settings.float_type)
    b = shard_bounds(Xnew.shape[0], comm.world)
    lo, hi = b[comm.rank], b[comm.rank + 1]
    prog = kern._program(X.shape[1])
    fm, fv = h.conditional(prog, X, Xnew[lo:hi], f, settings.numerics.jitter_level, q_sqrt=q_sqrt, white=white, full_cov=False)
    k = fm.shape[1]
    both = comm.all_gather_rows(np.concatenate([fm, fv], axis=1), [b[p + 1] - b[p] for p in range(comm.world)])
    return np.ascontiguousarray(both[:, :k]), np.ascontiguousarray(both[:, k:])


class _KLShare(object):
    """Option "svgp_kl_weight" = 1 / P on the handle for the duration of a sharded SVGP evaluation."""

    def __init__(self, handle, world):
        self.h, self.w = handle, 1.0 / float(world)

    def __enter__(self):
        self.h.set_option("svgp_kl_weight", self.w)
        return self

    def __exit__(self, *exc):
        self.h.set_option("svgp_kl_weight", 1.0)
        return False


def _shard(model, comm):
    b = shard_bounds(model.X.shape[0], comm.world)
    lo, hi = b[comm.rank], b[comm.rank + 1]
    if hi <= lo:
        raise ValueError("fewer data points than ranks")
    return model.X[lo:hi], model.Y[lo:hi]


def svgp_bound_distributed(model, comm=None, handle=None):
    """SVGP.compute_log_likelihood (models/svgp.py:108-125, Gaussian likelihood) with the data points sharded: rank p
    evaluates scale * sum_shard var_exp - KL / P on its rows, the ranks add the P numbers up in rank order."""
    comm = comm or _default_comm()
    h = handle or be.get_handle()
    Xs, Ys = _shard(model, comm)
    scale = float(model.num_data) / float(model.X.shape[0])
    with _KLShare(h, comm.world):
        part = model._bound_on(Xs, Ys, scale, handle=h)
    return float(ordered_sum(comm, [part])[0])


def svgp_bound_and_gradients_distributed(model, comm=None, handle=None):
    """SVGP.compute_log_likelihood_and_gradients with the data points sharded; returns (bound, [(Parameter, gradient), ...])
    identical (bit for bit) on every rank."""
    comm = comm or _default_comm()
    h = handle or be.get_handle()
    Xs, Ys = _shard(model, comm)
    scale = float(model.num_data) / float(model.X.shape[0])
    with _KLShare(h, comm.world):
        part, grads = model._bound_and_gradients_on(Xs, Ys, scale, handle=h)
    flat = np.concatenate([[part]] + [np.asarray(g, dtype=np.float64).reshape(-1) for _, g in grads])
    tot = ordered_sum(comm, flat)
    out, at = [], 1
    for p, g in grads:
        g = np.asarray(g)
        out.append((p, tot[at:at + g.size].reshape(g.shape)))
        at += g.size
    return float(tot[0]), out


class DeviceAllReduce(object):
    """Installs the communicator's device all-reduce as the handle's collective (gps_set_allreduce) for a `with` block.
    The buffer is a torch tensor (device memory the collective library knows); the callback reduces a slice of it."""

    def __init__(self, handle, comm, m, r):
        import torch
        self.h, self.comm = handle, comm
        self.cap = handle.allreduce_doubles(m, r)
        # (empty, not zeros: the library writes every double it reduces, and a fill kernel on torch's stream would not be
        # ordered against the library's own non-blocking stream)
        self.buf = torch.empty(self.cap, dtype=torch.float64, device=torch.device("cuda", handle.device))
        torch.cuda.synchronize(self.buf.device)
        self.error = None
        base = self.buf.data_ptr()

        def cb(ctx, ptr, count):
            try:
                off = (int(ptr) - base) // 8
                comm.all_reduce_sum(self.buf[off:off + int(count)])
                return 0
            except BaseException as e:          # never let an exception cross the C frame
                self.error = e
                return 1
        self._cb = be.ALLREDUCE_FN(cb)

    def __enter__(self):
        self.h.set_allreduce(self._cb, self.buf.data_ptr(), self.cap)
        return self

    def __exit__(self, *exc):
        self.h.set_allreduce(None, 0, 0)
        return False


def sparse_bound_distributed(model, comm=None, handle=None):
    """SGPR / GPRFITC .compute_log_likelihood (models/sgpr.py:121-153 / 252-291) with the data points sharded: the partial
    A A^T [M, M], A err, diag and scalars of all ranks are added by ONE device all-reduce inside gps_sgpr / gps_fitc, the
    second factorisation and the bound are finished redundantly -- the same value on every rank."""
    comm = comm or _default_comm()
    h = handle or be.get_handle()
    Xs, Ys = _shard(model, comm)
    if comm.world == 1:
        return model._call(X=Xs, Y=Ys, handle=h)[0]
    red = DeviceAllReduce(h, comm, len(model.feature), model.Y.shape[1])
    try:
        with red:
            return model._call(X=Xs, Y=Ys, handle=h)[0]
    except RuntimeError:
        if red.error is not None:
            raise red.error
        raise


def sparse_predict_distributed(model, Xnew, comm=None, handle=None):
    """predict_f of SGPR / GPRFITC (models/sgpr.py:155-189 / 293-318) from data points sharded over the ranks: after the
    all-reduce every rank holds the whole posterior, so each predicts its share of Xnew and the outputs are gathered."""
    comm = comm or _default_comm()
    h = handle or be.get_handle()
    Xs, Ys = _shard(model, comm)
    Xnew = np.ascontiguousarray(Xnew, dtype=settings.float_type)
    b = shard_bounds(Xnew.shape[0], comm.world)
    lo, hi = b[comm.rank], b[comm.rank + 1]
    red = DeviceAllReduce(h, comm, len(model.feature), model.Y.shape[1]) if comm.world > 1 else None
    R = model.Y.shape[1]
    Xmine = Xnew[lo:hi] if hi > lo else Xnew[:1]          # (every rank must take part in the all-reduce)
    try:
        if red is not None:
            with red:
                _, mean, var = model._call(Xnew=Xmine, want_bound=False, X=Xs, Y=Ys, handle=h)
        else:
            _, mean, var = model._call(Xnew=Xmine, want_bound=False, X=Xs, Y=Ys, handle=h)
    except RuntimeError:
        if red is not None and red.error is not None:
            raise red.error
        raise
    mean = mean + model.mean_function(Xmine)
    both = np.concatenate([mean, np.tile(var[:, None], [1, R])], axis=1)[: max(hi - lo, 0)]
    both = comm.all_gather_rows(both, [b[p + 1] - b[p] for p in range(comm.world)])
    return np.ascontiguousarray(both[:, :R]), np.ascontiguousarray(both[:, R:])
