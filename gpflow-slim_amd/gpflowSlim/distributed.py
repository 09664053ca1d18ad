"""Multi-GPU exact-GP factorisation: 1-D block-cyclic column Cholesky, one process per GPU.

There is no reference counterpart (GPflow-Slim is single-device); the oracle for this
module is the single-GPU result.  Partitioning: rank g owns block columns c with
c % P == g (width nb); per panel j the owner factors it (diagonal block + rows below), the panel is
broadcast (root = owner; RCCL over xGMI through torch.distributed, backend "nccl"), and every rank
applies it to the block columns it owns.  Look-ahead of one panel: the owner of panel j+1 updates and
factors that column first and starts its broadcast while everybody (itself included) is still
applying panel j, so panel work and the exchange hide under the trailing update.  Every rank keeps
each received panel, so L ends up replicated and alpha / predictions need no further exchange.

`block_column_schedule` is written against two small interfaces so that exactly the same schedule
runs (a) on GPUs: `HipPanelOps` (C ABI gps_dist_*) + `TorchComm`, and (b) in the CPU tests:
an emulation of the per-step pieces with numpy + the gloo backend (tests/test_dist_cpu.py).
"""
import numpy as np


class _Done(object):
    def wait(self):
        return True


def block_column_schedule(ops, comm, n_panels, lookahead=True):
    """Run the factorisation.  `ops`: panel_factor(j, buf), message(j, buf) -> buffer object for comm,
    unpack(j, buf), update(j, c_lo, c_hi).  `comm`: rank, world, broadcast(buffer, src, async_op) ->
    object with wait()."""
    P, rank = comm.world, comm.rank
    owner = lambda j: j % P

    if rank == owner(0):
        ops.panel_factor(0, 0)
    comm.broadcast(ops.message(0, 0), owner(0), False).wait()
    if rank != owner(0):
        ops.unpack(0, 0)

    for j in range(n_panels):
        nxt = j + 1
        if nxt >= n_panels:
            break
        buf = nxt % 2
        if lookahead:
            if rank == owner(nxt):
                ops.update(j, nxt, nxt + 1)            # the next panel's column first ...
                ops.panel_factor(nxt, buf)              # ... factor it ...
            work = comm.broadcast(ops.message(nxt, buf), owner(nxt), True)   # ... and ship it while
            ops.update(j, nxt + 1, n_panels)            # everybody applies panel j to the rest
            work.wait()
        else:
            ops.update(j, nxt, n_panels)
            if rank == owner(nxt):
                ops.panel_factor(nxt, buf)
            comm.broadcast(ops.message(nxt, buf), owner(nxt), False).wait()
        if rank != owner(nxt):
            ops.unpack(nxt, buf)


# ---- GPU side -------------------------------------------------------------------------------------
class TorchComm(object):
    """torch.distributed (backend "nccl" = RCCL on ROCm, or "gloo") behind the tiny comm interface."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def broadcast(self, tensor, src, async_op):
        if self.world == 1:
            return _Done()
        w = self._dist.broadcast(tensor, src=src, group=self.group, async_op=async_op)
        return w if w is not None else _Done()


    def all_gather_rows(self, local, counts):
        """Concatenate per-rank row blocks `local` [counts[rank], c] (numpy, host) on every rank."""
        if self.world == 1:
            return local
        import torch
        dev = "cuda" if self._dist.get_backend(self.group) == "nccl" else "cpu"
        c = local.shape[1]
        mx = max(counts)
        pad = np.zeros((mx, c))
        pad[: local.shape[0]] = local
        mine = torch.from_numpy(pad).to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self._dist.all_gather(parts, mine, group=self.group)
        return np.concatenate([p.cpu().numpy()[:k] for p, k in zip(parts, counts)], axis=0)


class SingleComm(object):
    rank, world = 0, 1

    def broadcast(self, tensor, src, async_op):
        return _Done()

    def all_gather_rows(self, local, counts):
        return local


class HipPanelOps(object):
    """Per-step pieces on one GPU through the C ABI; comm buffers are torch tensors (device memory
    plumbing only), library kernels run on torch's current stream so that they are ordered with the
    collectives."""

    def __init__(self, handle, prog, noise_var, resid, nparts, part, nb):
        import torch
        self.h = handle
        self.torch = torch
        handle.set_stream(torch.cuda.current_stream().cuda_stream, True)
        self.n_panels, mx = handle.dist_begin(prog, noise_var, resid, nparts, part, nb)
        self.bufs = [torch.empty(mx, dtype=torch.float64, device="cuda") for _ in range(2)]
        handle.dist_set_comm(self.bufs[0].data_ptr(), self.bufs[1].data_ptr())

    def panel_factor(self, j, buf):
        self.h.dist_panel_factor(j, buf)

    def message(self, j, buf):
        return self.bufs[buf][: self.h.dist_msg_doubles(j)]

    def unpack(self, j, buf):
        self.h.dist_unpack(j, buf)

    def update(self, j, c_lo, c_hi):
        self.h.dist_update(j, c_lo, c_hi)

    def finish(self):
        try:
            return self.h.dist_finish()
        finally:
            self.h.set_stream(0, False)


def gpr_lml_distributed(model, comm=None, nb=512, lookahead=True):
    """Log-marginal likelihood of a gpflowSlim.models.GPR with the covariance factorised across the
    ranks of `comm` (default: the default torch.distributed group, or a single rank).  Every rank must
    call this with the same model state; every rank returns the same value."""
    from . import _backend as be
    if comm is None:
        try:
            import torch.distributed as dist
            comm = TorchComm() if dist.is_available() and dist.is_initialized() else SingleComm()
        except ImportError:
            comm = SingleComm()
    h = model._handle()
    prog = model.kern._program(model.X.shape[1])
    model._factor_key = None
    ops = HipPanelOps(h, prog, float(np.squeeze(model.likelihood.variance)), model._resid(), comm.world,
                      comm.rank, nb)
    try:
        block_column_schedule(ops, comm, ops.n_panels, lookahead=lookahead)
    except Exception:
        h.set_stream(0, False)
        raise
    lml = ops.finish()
    model._factor_key = model._state_key()      # L and alpha are resident (replicated) on every rank
    return lml


def predict_f_distributed(model, Xnew, comm=None):
    """predict_f with the test points sharded over the ranks (the multi-RHS solve is
    independent over right-hand sides; L is replicated after gpr_lml_distributed, so there is no
    exchange in the solve -- only the final gather of the [N*, R] outputs).  Every rank passes the same
    Xnew and gets the full (mean, var) back.  Requires a resident factor (call gpr_lml_distributed or
    compute_log_likelihood first); models/gpr.py:119-131 per shard."""
    if comm is None:
        try:
            import torch.distributed as dist
            comm = TorchComm() if dist.is_available() and dist.is_initialized() else SingleComm()
        except ImportError:
            comm = SingleComm()
    Xnew = np.ascontiguousarray(Xnew, dtype=np.float64)
    n_new = Xnew.shape[0]
    bounds = [(n_new * r) // comm.world for r in range(comm.world + 1)]
    counts = [bounds[r + 1] - bounds[r] for r in range(comm.world)]
    lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
    saved = model.reuse_factor
    model.reuse_factor = True
    try:
        if hi > lo:
            mu, var = model.predict_f(Xnew[lo:hi])
        else:
            R = model.Y.shape[1]
            mu, var = np.zeros((0, R)), np.zeros((0, R))
    finally:
        model.reuse_factor = saved
    both = comm.all_gather_rows(np.concatenate([mu, var], axis=1), counts)
    R = mu.shape[1]
    return both[:, :R], both[:, R:]
