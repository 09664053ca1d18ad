"""Multi-GPU exact-GP factorisation: 1-D block-cyclic column Cholesky, one process per GPU.

There is no reference counterpart (GPflow-Slim is single-device); the oracle for this
module is the single-GPU result.  Partitioning: rank g owns block columns c with
c % P == g (width nb); per panel t the owner factors it (diagonal block + rows below, with (Y - m)^T
riding along as augmented rows so that alpha = L^-1 (Y - m) falls out block by block), the panel is
exchanged (root = owner), and every rank applies it to the block columns it owns.

Schedule (`block_column_schedule`), two lanes per rank:

* CHAIN lane -- the critical path.  For panel p, as soon as it is in place: update the next block column
  with it, factor that column if this rank owns it and start its exchange; then update the other "urgent"
  columns p+2 .. p+D; then receive panel p+1.
* BULK lane -- everything else of the trailing update of panel p (columns > p+D), one launch per panel.

Column c therefore takes panels 0 .. c-D-1 on the BULK lane and c-D .. c-1 on the CHAIN lane; the first
CHAIN update of a column waits for the last BULK update that touches it (panel c-D-1), BULK(p) waits
for panel p to be in place.  With depth D >= 2 the chain never waits for the bulk update of the panel it
has just received, so a step costs max(chain, bulk) instead of their sum.  D = 1 is the classic
one-panel look-ahead on two lanes; D = 0 is the plain right-looking order on one lane.

Exchange (`TorchComm.exchange`): RCCL over xGMI through torch.distributed.  xGMI is a full mesh of
point-to-point links, so a ring / tree broadcast of the whole panel is bound by one link; instead the
root scatters the message in P equal chunks (all of its links carry 1/P of the panel at once) and an
all-gather completes it (every link again carries 1/P): 2 S / (P b) per panel instead of S / b.
gloo (CPU tests; CUDA tensors only support broadcast there) falls back to one broadcast.

`block_column_schedule` is written against small interfaces so that exactly the same schedule runs
(a) on GPUs: `HipPanelOps` (C ABI gps_dist_*, torch streams / events) + `TorchComm`, and (b) in the CPU
tests: a numpy emulation with a vector-clock race detector over the two lanes + gloo
(tests/test_dist_cpu.py).
"""
import numpy as np

CHAIN, BULK = 0, 1


class _Done(object):
    def wait(self):
        return True


def block_column_schedule(ops, comm, n_panels, lookahead=2):
    """Run the factorisation.

    `ops`: panel_factor(t, buf), message(t, buf) -> buffer object for comm, unpack(t, buf),
    update(p, c_lo, c_hi, lane), record(lane) -> token, wait(lane, token), and the context manager
    comm_lane() under which collectives are issued.  `comm`: rank, world, exchange(buffer, src) ->
    object with wait().  `lookahead`: depth D (bool accepted: True = 2, False = 0)."""
    P, rank = comm.world, comm.rank
    D = 2 if lookahead is True else (0 if lookahead is False else int(lookahead))
    # comm buffers: panel t travels through (and, with partitioned storage, is read by the updates from) buffer t % nbufs.
    # Two suffice when a received panel is copied into place at once; three when the BULK update of panel p - 1 may still
    # be reading its buffer while panel p + 1 arrives (its last reader, BULK(p - 1), is joined by the CHAIN lane at step p
    # -- before the exchange of panel p + 2 = (p - 1) + 3 starts at step p + 1).
    nbufs = int(getattr(ops, "n_bufs", 2))
    owner = lambda t: t % P
    two_lanes = D >= 1
    bulk_lane = BULK if two_lanes else CHAIN
    bulk_done = {}                      # panel -> token recorded after its BULK update

    def exchange(t, buf):
        with ops.comm_lane():
            return comm.exchange(ops.message(t, buf), owner(t))

    def receive(t, buf, handle):
        with ops.comm_lane():
            handle.wait()
        if rank != owner(t):
            ops.unpack(t, buf)

    if rank == owner(0):
        ops.panel_factor(0, 0)
    receive(0, 0, exchange(0, 0))

    for p in range(n_panels - 1):
        # panel p is in place (CHAIN lane)
        nxt, buf = p + 1, (p + 1) % nbufs
        if D == 0:
            ops.update(p, nxt, n_panels, CHAIN)
            if rank == owner(nxt):
                ops.panel_factor(nxt, buf)
            receive(nxt, buf, exchange(nxt, buf))
            continue
        in_place = ops.record(CHAIN)
        last_urgent = min(p + D, n_panels - 1)

        def urgent(c):
            # first CHAIN update of column c = p + D: the BULK updates of panels <= p - 1 may still be running on it
            if c == p + D and (p - 1) in bulk_done:
                ops.wait(CHAIN, bulk_done.pop(p - 1))
            ops.update(p, c, c + 1, CHAIN)

        urgent(nxt)
        if rank == owner(nxt):
            ops.panel_factor(nxt, buf)
        handle = exchange(nxt, buf)                       # in flight while ...
        for c in range(nxt + 1, last_urgent + 1):         # ... the other urgent columns
            urgent(c)
        if last_urgent + 1 < n_panels:                     # ... and the bulk of the update run
            ops.wait(BULK, in_place)
            ops.update(p, last_urgent + 1, n_panels, BULK)
            bulk_done[p] = ops.record(BULK)
        receive(nxt, buf, handle)
    for tok in bulk_done.values():                         # (nothing is left to do there; join for the caller)
        ops.wait(CHAIN, tok)


# ---- GPU side -------------------------------------------------------------------------------------
_SIDE_GROUPS = {}


def _is_unsupported(exc):
    """An exception a collective raises synchronously, on every rank alike, because the backend lacks the operation for
    these tensors (gloo + device tensors: "... does not support ..." / "not implemented")."""
    if isinstance(exc, NotImplementedError):
        return True
    text = str(exc).lower()
    return any(k in text for k in ("not support", "unsupported", "not implemented", "no backend type"))


class TorchComm(object):
    """torch.distributed (backend "nccl" = RCCL on ROCm, or "gloo") behind the tiny comm interface."""

    def __init__(self, group=None, mode=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        # "scatter_allgather" needs scatter / all_gather_into_tensor on the tensors' device: RCCL, or gloo on CPU tensors
        self.mode = mode or ("scatter_allgather" if self.backend == "nccl" else "broadcast")
        self.bytes_sent = 0            # payload bytes this rank put on the wire (diagnostics for bench.py)
        self.exchanges = 0
        self._scratch = None
        # Votes about a collective that may have FAILED are not taken on the communicator that failed: a small gloo group
        # with a time-out beside the RCCL one (a rank stuck inside a broken collective then shows up as a time-out of the
        # vote -- fatal, with a message -- instead of a hang).  One per process group, kept for the process' lifetime.
        self._side = None
        if self.backend == "nccl" and self.world > 1:
            key = id(group) if group is not None else 0
            if key not in _SIDE_GROUPS:
                import datetime
                ranks = dist.get_process_group_ranks(group) if group is not None else None
                _SIDE_GROUPS[key] = dist.new_group(ranks=ranks, backend="gloo", timeout=datetime.timedelta(seconds=120))
            self._side = _SIDE_GROUPS[key]
        if self.mode == "scatter_allgather" and self.world > 1 and mode is None:
            self._probe()

    def _agree(self, ok, value=0.0):
        """(min over ranks of ok, max over ranks of value).  With the gloo side group any local failure may be voted on;
        without one (gloo itself is the backend) only failures that are raised synchronously on every rank alike."""
        import torch
        dist = self._dist
        grp = self._side if self._side is not None else self.group
        a = torch.tensor([float(ok)], dtype=torch.float64)
        b = torch.tensor([float(value)], dtype=torch.float64)
        try:
            dist.all_reduce(a, op=dist.ReduceOp.MIN, group=grp)
            dist.all_reduce(b, op=dist.ReduceOp.MAX, group=grp)
        except Exception as e:
            raise RuntimeError("the ranks could not agree on the outcome of a collective probe (a rank is stuck or gone): %s" % e)
        return float(a.item()), float(b.item())

    def _votable(self, exc):
        return self._side is not None or _is_unsupported(exc)

    def _probe(self):
        """One tiny scatter + all-gather before the first panel: a collective library that lacks one of the two
        operations for device tensors raises on every rank alike, and all ranks then agree (all-reduce MIN of a flag) to
        exchange panels by plain broadcast instead of failing in the middle of a factorisation."""
        import torch
        dist, P = self._dist, self.world
        dev = "cuda" if self.backend == "nccl" else "cpu"
        ok = 1
        try:
            t = torch.arange(2 * P, dtype=torch.float64, device=dev) if self.rank == 0 else torch.zeros(2 * P, dtype=torch.float64, device=dev)
            self._scatter_allgather(t, 0, 2).wait()
            if dev == "cuda":
                torch.cuda.synchronize()
            if not bool((t == torch.arange(2 * P, dtype=torch.float64, device=dev)).all()):
                ok = 0
        except Exception as e:
            # Without a side group only "this backend has no such operation for these tensors" may be voted on: raised at call
            # time, before anything is enqueued, on every rank alike.  Anything else (an RCCL error on one rank, an aborted
            # communicator) leaves the other ranks inside the collective -- an agreement all-reduce on the same group would
            # hang.  With the gloo side group (RCCL runs) every failure is voted on there, under a time-out.
            if not self._votable(e):
                raise
            ok = 0
        if self._agree(ok)[0] < 0.5:
            self.mode = "broadcast"
        self.bytes_sent, self.exchanges = 0, 0

    def autotune(self, doubles=1 << 22, reps=3):
        """Pick the panel exchange by measurement: time `reps` exchanges of a `doubles`-long message (default 32 MB, the
        order of a panel) as one broadcast and as scatter + all-gather, take the maximum over ranks of each, keep the
        faster.  Collective; every rank ends with the same mode.  Returns {mode: seconds per exchange}."""
        if self.world == 1:
            return {}
        import time
        import torch
        dist, P = self._dist, self.world
        dev = "cuda" if self.backend == "nccl" else "cpu"
        n = (int(doubles) // P) * P
        buf = torch.zeros(n, dtype=torch.float64, device=dev)
        sync = torch.cuda.synchronize if dev == "cuda" else (lambda: None)
        times = {}
        keep = self.mode
        for mode in ("broadcast", "scatter_allgather"):
            ok, dt = 1, float("inf")
            try:
                self.mode = mode
                self.exchange(buf, 0).wait()                     # (first use of an operation sets it up)
                sync(); dist.barrier(group=self.group); sync()
                t0 = time.perf_counter()
                for i in range(reps):
                    self.exchange(buf, i % P).wait()
                sync()
                dt = (time.perf_counter() - t0) / reps
            except Exception as e:
                if not self._votable(e):            # (see _probe)
                    raise
                ok = 0
            all_ok, worst = self._agree(ok, dt if ok else 1e30)
            if all_ok > 0.5:
                times[mode] = worst
        self.mode = min(times, key=times.get) if times else keep
        self.bytes_sent, self.exchanges = 0, 0
        return times

    def exchange(self, tensor, src):
        """Make `tensor` (complete on rank `src`) complete on every rank; returns an object with wait()."""
        if self.world == 1:
            return _Done()
        dist, P = self._dist, self.world
        self.exchanges += 1
        n = tensor.numel()
        if self.mode == "broadcast" or n < 2 * P:
            if self.rank == src:
                self.bytes_sent += 8 * n * (P - 1)
            w = dist.broadcast(tensor, src=src, group=self.group, async_op=True)
            return w if w is not None else _Done()
        return self._scatter_allgather(tensor, src, n // P)

    def _scatter_allgather(self, tensor, src, chunk):
        # scatter + all-gather on P equal chunks (the ragged end of the message goes by a small broadcast)
        dist, P, n = self._dist, self.world, tensor.numel()
        body = tensor[: chunk * P]
        mine = body[self.rank * chunk:(self.rank + 1) * chunk]
        if self.rank == src:
            # the root's own chunk is already where it belongs: let the scatter deliver it into a scratch chunk rather
            # than onto itself
            if self._scratch is None or self._scratch.numel() < chunk or self._scratch.device != tensor.device:
                self._scratch = tensor.new_empty(chunk)
            parts, landing = [body[i * chunk:(i + 1) * chunk] for i in range(P)], self._scratch[:chunk]
        else:
            parts, landing = None, mine
        self.bytes_sent += 8 * chunk * (P - 1) * (2 if self.rank == src else 1)
        works = [dist.scatter(landing, scatter_list=parts, src=src, group=self.group, async_op=True)]
        if self.backend != "nccl":
            works[-1].wait()          # gloo runs queued work on a thread pool: keep the two phases ordered
        works.append(dist.all_gather_into_tensor(body, mine, group=self.group, async_op=True))
        if chunk * P < n:
            works.append(dist.broadcast(tensor[chunk * P:], src=src, group=self.group, async_op=True))
        return _Works(works)

    # kept for callers of the round-1 interface
    def broadcast(self, tensor, src, async_op):
        if self.world == 1:
            return _Done()
        w = self._dist.broadcast(tensor, src=src, group=self.group, async_op=async_op)
        return w if w is not None else _Done()

    def all_reduce_sum(self, tensor):
        """In-place sum over ranks of a device tensor; returns once the result is visible to the device (the collective of
        gps_set_allreduce: gpflowSlim/distributed_sparse.py).  RCCL; gloo reduces device tensors through the host."""
        if self.world == 1:
            return tensor
        self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM, group=self.group)
        if tensor.is_cuda:
            import torch
            torch.cuda.synchronize(tensor.device)
        self.bytes_sent += 8 * tensor.numel()
        return tensor

    def all_gather_rows(self, local, counts):
        """Concatenate per-rank row blocks `local` [counts[rank], c] (numpy, host) on every rank."""
        if self.world == 1:
            return local
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        c = local.shape[1]
        mx = max(counts)
        pad = np.zeros((mx, c))
        pad[: local.shape[0]] = local
        mine = torch.from_numpy(pad).to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self._dist.all_gather(parts, mine, group=self.group)
        return np.concatenate([p.cpu().numpy()[:k] for p, k in zip(parts, counts)], axis=0)


class _NativeWork(object):
    def __init__(self, handle, slot):
        self.h, self.slot = handle, slot

    def wait(self):
        self.h.comm_wait(self.slot)          # the handle's stream waits; the host does not
        return True


class RcclComm(object):
    """RCCL called by the library itself (csrc/comm_rccl.hip: gps_comm_*) behind the same small comm interface -- no
    PyTorch in the data path; torch (or anything else) is only the side channel that carries the 128-byte unique id from
    rank 0 to the others at start-up.  One communicator per library handle.

    `bootstrap(payload_or_None) -> payload`: called once with rank 0's unique id (None on the other ranks); must return
    rank 0's bytes on every rank (e.g. torch.distributed.broadcast_object_list over gloo, an MPI bcast, a shared file).
    Verified on hardware with the real RCCL at world size 1 and, through the shared-memory stand-in transport of tests/fake_rccl,
    with two processes on one GPU -- the build loop has one GPU; TorchComm stays the default of bench.py."""

    def __init__(self, handle, rank, world, bootstrap=None, mode="scatter_allgather"):
        from . import _backend as be
        self.h, self.rank, self.world, self.mode = handle, int(rank), int(world), mode
        self.backend = "rccl-native"
        self.native_schedule = True        # gpr_lml_distributed / predict_f_distributed run gps_dist_lml / gps_dist_predict (False: the Python schedule)
        self.bytes_sent = 0
        self.exchanges = 0
        self._slot = 0
        be.comm_load()                        # every rank binds the same librccl ($GPFLOWSLIM_RCCL_LIB / PyTorch's / the system's)
        uid = be.comm_unique_id() if self.rank == 0 else None
        if self.world > 1:
            if bootstrap is None:
                raise ValueError("RcclComm: world > 1 needs a bootstrap callable that carries rank 0's unique id to every rank")
            uid = bootstrap(uid)
        handle.comm_init(self.rank, self.world, uid)

    def close(self):
        self.h.comm_destroy()

    def exchange(self, tensor, src):
        """`tensor`: anything with data_ptr() / numel() holding doubles on the handle's device (a torch tensor)."""
        n = tensor.numel()
        P = self.world
        mode = 1 if (self.mode == "scatter_allgather" and n >= 2 * P) else 0
        slot = self._slot
        self._slot = (self._slot + 1) % 8
        self.h.comm_exchange(tensor.data_ptr(), n, src, mode, slot)
        self.exchanges += 1
        if P > 1:
            chunk = n // P
            self.bytes_sent += (8 * chunk * (P - 1) * (2 if self.rank == src else 1)) if mode == 1 else (8 * n * (P - 1) if self.rank == src else 0)
        return _NativeWork(self.h, slot)

    def all_reduce_sum(self, tensor):
        self.h.comm_allreduce(tensor.data_ptr(), tensor.numel())
        self.bytes_sent += 8 * tensor.numel()
        return tensor

    def all_gather_rows(self, local, counts):
        """Ragged row blocks of host arrays: every rank writes its rows into a zeroed [sum(counts), c] device buffer, one
        all-reduce adds the ranks' contributions up (x + 0 = x: exact)."""
        if self.world == 1:
            return local
        import torch
        c = local.shape[1]
        tot = int(sum(counts))
        off = int(sum(counts[: self.rank]))
        buf = torch.zeros((max(tot, 1), c), dtype=torch.float64, device=torch.device("cuda", self.h.device))
        if local.shape[0]:
            buf[off:off + local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(buf.device)
        torch.cuda.synchronize(buf.device)
        self.h.comm_allreduce(buf.data_ptr(), buf.numel())
        return buf[:tot].cpu().numpy()


class _Works(object):
    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()
        return True


class SingleComm(object):
    rank, world = 0, 1
    bytes_sent = 0
    exchanges = 0

    def exchange(self, tensor, src):
        return _Done()

    def broadcast(self, tensor, src, async_op):
        return _Done()

    def all_gather_rows(self, local, counts):
        return local

    def all_reduce_sum(self, tensor):
        return tensor


class HipPanelOps(object):
    """Per-step pieces on one GPU through the C ABI.  Comm buffers are torch tensors and the two lanes are torch
    streams (device-memory / stream plumbing only): the CHAIN lane -- a high-priority stream on which the library's
    kernels and the collectives are ordered -- and the BULK lane for the trailing updates.  Use as a context manager:
    leaving it always gives the handle its own stream back."""

    def __init__(self, handle, prog, noise_var, resid, nparts, part, nb, two_lanes=True, partitioned=None):
        import torch
        self.h = handle
        self.torch = torch
        self.nparts = max(int(nparts), 1)
        if partitioned is not None:
            handle.set_option("dist_partitioned", 1 if partitioned else 0)
        self.n_bufs = handle.dist_comm_bufs_needed()          # 3: partitioned storage (panels are read from the buffers); 2: replicated
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        self.chain = torch.cuda.Stream(priority=hi)
        self.bulk = torch.cuda.Stream(priority=lo) if two_lanes else self.chain
        self._installed = False
        try:
            self.chain.wait_stream(torch.cuda.current_stream())
            handle.set_stream(self.chain.cuda_stream, True)
            self._installed = True
            handle.dist_set_bulk_stream(self.bulk.cuda_stream if two_lanes else 0)
            self.n_panels, mx = handle.dist_begin(prog, noise_var, resid, nparts, part, nb)
            mx = -(-mx // max(nparts, 1)) * max(nparts, 1)          # room for equal chunks
            with torch.cuda.stream(self.chain):
                self.bufs = [torch.empty(mx, dtype=torch.float64, device="cuda") for _ in range(self.n_bufs)]
            handle.dist_set_comm_bufs([b.data_ptr() for b in self.bufs])
        except Exception:
            self.close()
            raise

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self._installed:
            self._installed = False
            # both lanes are drained on every exit path: when the schedule, an exchange or the finish raised (a
            # not-positive-definite panel raises on every rank), trailing updates may still be running on the BULK lane
            # and the caller's next launch on the handle's own stream would rebuild K under them
            if self.bulk is not self.chain:
                self.chain.wait_stream(self.bulk)
            self.torch.cuda.current_stream().wait_stream(self.chain)
            try:
                self.h.dist_set_bulk_stream(0)
            finally:
                self.h.set_stream(0, False)

    def comm_lane(self):
        return self.torch.cuda.stream(self.chain)

    def panel_factor(self, t, buf):
        self.h.dist_panel_factor(t, buf)

    def message(self, t, buf):
        n = self.h.dist_msg_doubles(t)
        return self.bufs[buf][: -(-n // self.nparts) * self.nparts]      # whole chunks for the scatter + all-gather

    def unpack(self, t, buf):
        self.h.dist_unpack(t, buf)

    def update(self, p, c_lo, c_hi, lane):
        self.h.dist_update(p, c_lo, c_hi, lane)

    def record(self, lane):
        ev = self.torch.cuda.Event()
        ev.record(self.chain if lane == CHAIN else self.bulk)
        return ev

    def wait(self, lane, token):
        (self.chain if lane == CHAIN else self.bulk).wait_event(token)

    def finish(self):
        return self.h.dist_finish()


def _default_comm():
    try:
        import torch.distributed as dist
        return TorchComm() if dist.is_available() and dist.is_initialized() else SingleComm()
    except ImportError:
        return SingleComm()


def gpr_lml_distributed(model, comm=None, nb=512, lookahead=2, partitioned=True):
    """Log-marginal likelihood of a gpflowSlim.models.GPR with the covariance factorised across the
    ranks of `comm` (default: the default torch.distributed group, or a single rank).  Every rank must
    call this with the same model state; every rank returns the same value (bit for bit), or every rank
    raises NotPositiveDefiniteError.

    partitioned=True (default): a rank stores only the block columns it owns -- 8 N^2 / P bytes -- and the
    factor stays distributed (predict_f_distributed streams the panels once more); False: every rank keeps every
    panel (8 N^2 bytes per rank), after which the ordinary warm predict_f works on every rank."""
    comm = comm or _default_comm()
    h = model._handle()
    prog = model.kern._program(model.X.shape[1])
    model._factor_key = None               # (the handle's setter also drops any partitioned-factor claim)
    if isinstance(comm, RcclComm) and comm.native_schedule and comm.h is h:
        # the library's own communicator: the whole schedule runs inside the library too (gps_dist_lml) -- no Python per panel
        h.set_option("dist_partitioned", 1 if partitioned else 0)
        mode = 1 if comm.mode == "scatter_allgather" else 0
        lml = h.dist_lml(prog, float(np.squeeze(model.likelihood.variance)), model._resid(), nb, 2 if lookahead is True else int(lookahead), mode)
        n_panels = -(-model.X.shape[0] // nb)
        comm.exchanges += n_panels
        for j in range(n_panels):              # the payload the library put on the wire (what RcclComm.exchange would have counted)
            cnt = -(-h.dist_msg_doubles(j) // comm.world) * comm.world
            root = (j % comm.world) == comm.rank
            if comm.world > 1:
                comm.bytes_sent += (8 * (cnt // comm.world) * (comm.world - 1) * (2 if root else 1)) if mode == 1 else (8 * cnt * (comm.world - 1) if root else 0)
        if partitioned:
            h.dist_state = {"key": model._state_key(), "native": True, "world": comm.world}
        else:
            model._factor_key = model._state_key()
        return lml
    with HipPanelOps(h, prog, float(np.squeeze(model.likelihood.variance)), model._resid(), comm.world, comm.rank,
                     nb, two_lanes=bool(lookahead), partitioned=partitioned) as ops:
        block_column_schedule(ops, comm, ops.n_panels, lookahead=lookahead)
        lml = ops.finish()
        if partitioned:
            # what predict_f_distributed needs to stream the panels again (the comm buffers stay alive with it)
            h.dist_state = {"key": model._state_key(), "n_panels": ops.n_panels, "bufs": ops.bufs, "nparts": ops.nparts,
                            "world": comm.world}
    if not partitioned:
        model._factor_key = model._state_key()      # L and alpha are resident (replicated) on every rank
    return lml


def predict_f_distributed(model, Xnew, comm=None):
    """predict_f with the test points sharded over the ranks (the multi-RHS solve is
    independent over right-hand sides; L is replicated after gpr_lml_distributed, so there is no
    exchange in the solve -- only the final gather of the [N*, R] outputs).  Every rank passes the same
    Xnew and gets the full (mean, var) back.  Requires a resident factor (call gpr_lml_distributed or
    compute_log_likelihood first); models/gpr.py:119-131 per shard."""
    comm = comm or _default_comm()
    Xnew = np.ascontiguousarray(Xnew, dtype=np.float64)
    n_new = Xnew.shape[0]
    bounds = [(n_new * r) // comm.world for r in range(comm.world + 1)]
    counts = [bounds[r + 1] - bounds[r] for r in range(comm.world)]
    lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
    # the claim "a partitioned factor of THIS model state is resident" is the handle's (several models may share a handle and
    # even an X array; any other evaluation on the handle clears it): no match -> the ordinary path below re-factorises
    st = model._handle().dist_state
    if st is not None and st.get("native") and st["key"] == model._state_key() and st["world"] == comm.world:
        R = model.Y.shape[1]
        Xmine = Xnew[lo:hi]
        mu, var = model._handle().dist_predict(model.kern._program(model.X.shape[1]), Xmine, R, 1 if comm.mode == "scatter_allgather" else 0)
        mu = mu + model.mean_function(Xmine) if Xmine.shape[0] else mu
        both = comm.all_gather_rows(np.concatenate([mu, np.tile(var[:, None], [1, R])], axis=1), counts)
        return both[:, :R], both[:, R:]
    if st is not None and st["key"] == model._state_key() and st["world"] == comm.world:
        R = model.Y.shape[1]
        Xmine = Xnew[lo:hi]
        mu, var = predict_streamed(model._handle(), model.kern._program(model.X.shape[1]), Xmine, comm, st["n_panels"],
                                   st["bufs"], st["nparts"], R)
        mu = mu + model.mean_function(Xmine) if Xmine.shape[0] else mu
        both = comm.all_gather_rows(np.concatenate([mu, np.tile(var[:, None], [1, R])], axis=1), counts)     # models/gpr.py:131
        return both[:, :R], both[:, R:]
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


def predict_streamed(h, prog, Xmine, comm, n_panels, bufs, nparts, R):
    """predict_f for this rank's test points from a PARTITIONED factor on handle `h` (gps_dist_solve_*): the owner of panel
    j packs it again, the panel is exchanged like during the factorisation, every rank applies it to its right-hand sides;
    the exchange of panel j + 1 is in flight while panel j is applied.  Returns (A^T alpha [n*, R] -- the caller adds the
    mean function --, fvar [n*]); models/gpr.py:119-131.  `bufs`: the comm buffers of the factorisation (>= 2)."""
    import torch
    P, rank = comm.world, comm.rank
    mine = Xmine.shape[0] > 0
    lane = torch.cuda.Stream()
    lane.wait_stream(torch.cuda.current_stream())
    h.set_stream(lane.cuda_stream, True)
    try:
        h.dist_set_comm_bufs([b.data_ptr() for b in bufs])
        if mine:
            h.dist_solve_begin(prog, Xmine)

        def send(j):
            buf = j % 2
            if rank == j % P:
                h.dist_solve_pack(j, buf)
            n = h.dist_msg_doubles(j)
            with torch.cuda.stream(lane):
                return comm.exchange(bufs[buf][: -(-n // nparts) * nparts], j % P)
        pending = send(0)
        for j in range(n_panels):
            with torch.cuda.stream(lane):
                pending.wait()
            if j + 1 < n_panels:
                pending = send(j + 1)          # (stream-ordered after apply(j - 1), the last reader of that buffer) in flight ...
            if mine:
                h.dist_solve_apply(j, j % 2)   # ... while panel j is applied
        if mine:
            mu, var = h.dist_solve_finish(prog, Xmine.shape[0], R)
        else:
            mu, var = np.zeros((0, R)), np.zeros((0,))
    finally:
        torch.cuda.current_stream().wait_stream(lane)
        h.set_stream(0, False)
    return mu, var
