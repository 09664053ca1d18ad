"""world_size > 1 on CPU (gloo): the block-column schedule of gpflowSlim/distributed.py -- the SAME
function the GPU path runs -- with the per-step pieces emulated in numpy/scipy (test infrastructure
only) and the panel exchange (scatter + all-gather, or broadcast) carried by torch.distributed/gloo.
Checks ownership, message sizes, the event protocol between the two lanes (vector clocks), look-ahead
depths and that every rank ends with the full factor."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyPanelOps(object):
    """Emulates gps_dist_* on the host: full [np, np] buffer per rank, only owned block columns are
    initialised (everything else NaN, so any use of data that was never received shows up).

    The two lanes of the schedule are emulated sequentially, but every access is checked with vector
    clocks: an operation on lane A that touches a region last written on lane B must have waited for a
    token recorded on B after that write (ops.record / ops.wait) -- on the GPU the lanes are concurrent
    streams, so an unordered pair is a data race even though it cannot misbehave here."""

    def __init__(self, A, nb, nparts, part):
        import torch
        self.torch = torch
        self.np_, self.nb, self.P, self.rank = A.shape[0], nb, nparts, part
        self.n_panels = self.np_ // nb
        self.M = np.full_like(A, np.nan)
        for c in range(part, self.n_panels, nparts):
            self.M[c * nb:, c * nb:(c + 1) * nb] = A[c * nb:, c * nb:(c + 1) * nb]
        mx = -(-(self.np_ * nb) // nparts) * nparts
        self.bufs = [torch.full((mx,), float("nan"), dtype=torch.float64) for _ in range(2)]
        self.log = []
        self.clock = [[0, 0], [0, 0]]            # clock[lane] = [events of lane 0 seen, events of lane 1 seen]
        self.last_write = {}                     # region -> (lane, count)
        self.last_reads = {}                     # region -> list of (lane, count) since the last write
        self.max_lag = 0                         # how many panels the BULK lane was behind the CHAIN lane at most
        self._bulk_panel = -1

    # ---- lanes
    def _tick(self, lane):
        self.clock[lane][lane] += 1
        return (lane, self.clock[lane][lane])

    def _ordered(self, lane, ev):
        return ev is None or ev[0] == lane or self.clock[lane][ev[0]] >= ev[1]

    def _read(self, lane, region):
        me = self._tick(lane)
        assert self._ordered(lane, self.last_write.get(region)), ("read of %s on lane %d races with its writer" % (region, lane))
        self.last_reads.setdefault(region, []).append(me)

    def _write(self, lane, region):
        me = self._tick(lane)
        assert self._ordered(lane, self.last_write.get(region)), ("write of %s on lane %d races with the previous writer" % (region, lane))
        for ev in self.last_reads.get(region, []):
            assert self._ordered(lane, ev), ("write of %s on lane %d races with a reader" % (region, lane))
        self.last_write[region] = me
        self.last_reads[region] = []

    def record(self, lane):
        return (lane, self.clock[lane][lane], list(self.clock[lane]))

    def wait(self, lane, token):
        src, _, vc = token
        for i in (0, 1):
            self.clock[lane][i] = max(self.clock[lane][i], vc[i])

    class _Nop(object):
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def comm_lane(self):
        return self._Nop()

    # ---- per-step pieces
    def panel_factor(self, j, buf):
        nb, s = self.nb, j * self.nb
        assert j % self.P == self.rank, "only the owner factors a panel"
        self._write(0, ("col", j)); self._write(0, ("buf", buf))
        D = self.M[s:s + nb, s:s + nb]
        assert not np.isnan(np.tril(D)).any(), "panel %d factored before all of its updates" % j
        L = np.linalg.cholesky(np.tril(D) + np.tril(D, -1).T)
        self.M[s:s + nb, s:s + nb] = L
        if s + nb < self.np_:
            self.M[s + nb:, s:s + nb] = sl.solve_triangular(L, self.M[s + nb:, s:s + nb].T, lower=True).T
        rows = self.np_ - s
        self.bufs[buf][: rows * nb] = self.torch.from_numpy(np.ascontiguousarray(self.M[s:, s:s + nb]).ravel())
        self.log.append(("factor", j))

    def message(self, j, buf):
        return self.bufs[buf][: (self.np_ - j * self.nb) * self.nb]

    def unpack(self, j, buf):
        nb, s = self.nb, j * self.nb
        assert j % self.P != self.rank
        self._read(0, ("buf", buf)); self._write(0, ("col", j))
        rows = self.np_ - s
        self.M[s:, s:s + nb] = self.bufs[buf][: rows * nb].numpy().reshape(rows, nb)
        self.log.append(("unpack", j))

    def update(self, j, c_lo, c_hi, lane=0):
        nb = self.nb
        if lane == 1:
            self._bulk_panel = j
        for c in range(max(c_lo, j + 1), min(c_hi, self.n_panels)):
            if c % self.P != self.rank:
                continue
            self._read(lane, ("col", j)); self._write(lane, ("col", c))
            Lc = self.M[c * nb:, j * nb:(j + 1) * nb]
            assert not np.isnan(Lc).any(), "panel %d used before it was received" % j
            self.M[c * nb:, c * nb:(c + 1) * nb] -= Lc @ Lc[:nb].T
            self.log.append(("update", j, c, lane))


class PartitionedNumpyOps(NumpyPanelOps):
    """The PARTITIONED storage mode of gps_dist_* (option "dist_partitioned"): a rank only ever holds its own block columns;
    a received panel stays in the comm buffer it arrived in (slot = panel % n_bufs) and the trailing updates read it
    there -- so a buffer may only be overwritten by a later panel once every update that reads it is ordered before the
    write.  The vector clocks check exactly that; with two buffers and a look-ahead depth >= 2 it must fire."""

    def __init__(self, A, nb, nparts, part, n_bufs=3):
        NumpyPanelOps.__init__(self, A, nb, nparts, part)
        self.n_bufs = n_bufs
        mx = -(-(self.np_ * nb) // nparts) * nparts
        self.bufs = [self.torch.full((mx,), float("nan"), dtype=self.torch.float64) for _ in range(n_bufs)]

    def panel_factor(self, j, buf):
        assert buf == j % self.n_bufs, "panel j travels through buffer j % n_bufs"
        NumpyPanelOps.panel_factor(self, j, buf)

    def message(self, j, buf):
        assert buf == j % self.n_bufs
        if j % self.P != self.rank:
            self._write(0, ("buf", buf))           # the collective writes the receive buffer, ordered on the CHAIN lane
        return NumpyPanelOps.message(self, j, buf)

    def unpack(self, j, buf):
        assert j % self.P != self.rank and buf == j % self.n_bufs
        self._read(0, ("buf", buf))                # (only the panel's scalars are kept)
        self.log.append(("unpack", j))

    def update(self, j, c_lo, c_hi, lane=0):
        nb, buf = self.nb, j % self.n_bufs
        rows = self.np_ - j * nb
        for c in range(max(c_lo, j + 1), min(c_hi, self.n_panels)):
            if c % self.P != self.rank:
                continue
            self._read(lane, ("buf", buf)); self._write(lane, ("col", c))
            panel = self.bufs[buf][: rows * nb].numpy().reshape(rows, nb)
            Lc = panel[(c - j) * nb:]
            assert not np.isnan(Lc).any(), "panel %d read from a buffer that does not hold it" % j
            self.M[c * nb:, c * nb:(c + 1) * nb] -= Lc @ Lc[:nb].T
            self.log.append(("update", j, c, lane))


def _part_worker(rank, world, port, n, nb, lookahead, n_bufs, q):
    sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
    import torch.distributed as dist
    from gpflowSlim.distributed import TorchComm, block_column_schedule
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(321)
        G = rng.standard_normal((n, n))
        A = G @ G.T + n * np.eye(n)
        ops = PartitionedNumpyOps(A, nb, world, rank, n_bufs)
        comm = TorchComm(mode="broadcast")
        raced = None
        try:
            block_column_schedule(ops, comm, ops.n_panels, lookahead=lookahead)
        except AssertionError as e:
            raced = str(e)
        L = np.linalg.cholesky(A)
        own = [c for c in range(ops.n_panels) if c % world == rank]
        err = max(float(np.abs(np.tril(ops.M[:, c * nb:(c + 1) * nb] - L[:, c * nb:(c + 1) * nb])[c * nb:]).max()) for c in own) if raced is None else None
        others_untouched = all(np.isnan(ops.M[c * nb:, c * nb:(c + 1) * nb]).all() for c in range(ops.n_panels) if c % world != rank)
        q.put((rank, err, others_untouched, raced))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nb,lookahead", [(2, 128, 16, 2), (3, 176, 16, 3), (2, 96, 16, 1), (3, 112, 16, 0), (2, 160, 16, 5)])
def test_partitioned_storage_schedule_gloo(world, n, nb, lookahead):
    """Partitioned storage under gloo: every rank ends with exactly its own block columns of the factor, never touches the
    others, and three comm buffers are enough for every look-ahead depth (vector-clock race detector)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_part_worker, args=(r, world, port, n, nb, lookahead, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, err, untouched, raced in res:
        assert raced is None, raced
        assert err <= 1e-10 and untouched


def test_partitioned_storage_needs_three_buffers():
    """With only two comm buffers the BULK update of panel p - 1 can still be reading its buffer when panel p + 1 is
    received into it (look-ahead depth >= 2): the race detector must say so -- this is why gps_dist_comm_bufs_needed() is 3."""
    sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
    from gpflowSlim.distributed import block_column_schedule

    class FakeTwoRanks(object):                    # rank 0 of 2, nothing really exchanged: every panel counts as received
        rank, world = 0, 2

        def exchange(self, tensor, src):
            class _W(object):
                def wait(self):
                    return True
            return _W()
    rng = np.random.default_rng(5)
    n, nb = 160, 16
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
    L = np.linalg.cholesky(A)

    class Fed(PartitionedNumpyOps):                # panels of the other rank appear in the buffer as if received
        def message(self, j, buf):
            out = PartitionedNumpyOps.message(self, j, buf)
            if j % 2 != 0:
                rows = self.np_ - j * nb
                self.bufs[buf][: rows * nb] = self.torch.from_numpy(np.ascontiguousarray(L[j * nb:, j * nb:(j + 1) * nb]).ravel())
            return out
    ok = Fed(A, nb, 2, 0, n_bufs=3)
    block_column_schedule(ok, FakeTwoRanks(), ok.n_panels, lookahead=2)
    bad = Fed(A, nb, 2, 0, n_bufs=2)
    with pytest.raises(AssertionError, match="races"):
        block_column_schedule(bad, FakeTwoRanks(), bad.n_panels, lookahead=2)


def _worker(rank, world, port, n, nb, lookahead, mode, q):
    sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
    import torch.distributed as dist
    from gpflowSlim.distributed import TorchComm, block_column_schedule
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)
        G = rng.standard_normal((n, n))
        A = G @ G.T + n * np.eye(n)
        ops = NumpyPanelOps(A, nb, world, rank)
        comm = TorchComm(mode=mode)
        if mode == "scatter_allgather":
            # the start-up probe the RCCL path runs (one tiny scatter + all-gather, all ranks agree on the outcome)
            comm._probe()
            assert comm.mode == "scatter_allgather" and comm.bytes_sent == 0
            real = comm._scatter_allgather
            # a backend that lacks the operation raises synchronously on every rank alike ...
            comm._scatter_allgather = lambda *a: (_ for _ in ()).throw(RuntimeError("ProcessGroup: scatter is not supported for these tensors"))
            comm._probe()
            assert comm.mode == "broadcast"                     # every rank fell back together
            # ... anything else is not voted on (the other ranks may be inside the failed collective): it propagates
            comm.mode = "scatter_allgather"
            comm._scatter_allgather = lambda *a: (_ for _ in ()).throw(RuntimeError("RCCL error: unhandled system error"))
            try:
                comm._probe()
                raise AssertionError("a rank-local collective failure must not be swallowed")
            except RuntimeError as e:
                assert "unhandled system error" in str(e)
            # ... unless the vote has a channel of its own (what RCCL runs get: a gloo side group with a time-out): then any
            # local failure is voted on there and every rank falls back together
            import datetime
            comm._side = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=60))
            comm.mode = "scatter_allgather"
            comm._probe()
            assert comm.mode == "broadcast"
            comm._side = None
            comm._scatter_allgather, comm.mode = real, "scatter_allgather"
            times = comm.autotune(doubles=4096, reps=2)         # measured choice of the exchange: both modes work here
            assert set(times) == {"broadcast", "scatter_allgather"} and comm.mode in times and comm.bytes_sent == 0
            comm.mode = "scatter_allgather"
        block_column_schedule(ops, comm, ops.n_panels, lookahead=lookahead)
        L = np.linalg.cholesky(A)
        err = float(np.abs(np.tril(ops.M) - L).max())
        factored = [e[1] for e in ops.log if e[0] == "factor"]
        per_col = {}
        for e in ops.log:
            if e[0] == "update":
                per_col.setdefault(e[2], []).append(e[1])
        q.put((rank, err, factored, per_col, comm.bytes_sent))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,n,nb,lookahead,mode", [
    (2, 96, 16, 2, "scatter_allgather"), (2, 96, 16, 0, "broadcast"), (2, 96, 16, 1, "broadcast"),
    (3, 112, 16, 2, "scatter_allgather"), (3, 176, 16, 3, "scatter_allgather"), (2, 48, 48, True, "scatter_allgather"),
    (3, 112, 16, 4, "broadcast")])
def test_block_column_schedule_gloo(world, n, nb, lookahead, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, lookahead, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_panels = n // nb
    total_bytes = 0
    for rank, err, factored, per_col, sent in res:
        assert err <= 1e-10, (rank, err)                       # every rank holds the whole factor
        assert factored == list(range(rank, n_panels, world))  # block-cyclic ownership
        for c, panels in per_col.items():                      # every owned column took every earlier panel exactly once, in order
            assert c % world == rank and panels == list(range(c)), (rank, c, panels)
        total_bytes += sent
    # payload on the wire: a broadcast delivers every panel to the P - 1 other ranks once; scatter + all-gather moves
    # (P + 1) / P of that in total, but 1 / P of a panel per link and phase instead of a whole panel over one link
    expect = sum((n - j * nb) * nb for j in range(n_panels)) * 8 * (world - 1)
    if mode == "scatter_allgather":
        expect = expect * (world + 1) / world
    assert abs(total_bytes - expect) <= 8 * 2 * world * world * n_panels, (total_bytes, expect)


@pytest.mark.parametrize("lookahead", [0, 1, 2, 3, 7, True, False])
def test_schedule_single_rank_matches_lapack(lookahead):
    sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
    from gpflowSlim.distributed import SingleComm, block_column_schedule
    rng = np.random.default_rng(5)
    n, nb = 96, 16
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
    ops = NumpyPanelOps(A, nb, 1, 0)
    block_column_schedule(ops, SingleComm(), ops.n_panels, lookahead=lookahead)
    assert np.abs(np.tril(ops.M) - np.linalg.cholesky(A)).max() <= 1e-10
    lanes = {e[3] for e in ops.log if e[0] == "update"}
    depth = 2 if lookahead is True else int(lookahead)
    assert lanes == ({0} if depth == 0 or depth >= ops.n_panels - 1 else {0, 1})   # the bulk of the updates really is on the second lane


def test_race_detector_catches_a_missing_wait():
    """The vector-clock check of NumpyPanelOps is what validates the schedule's event protocol: drop the wait of the
    first CHAIN update of a column on the last BULK update that touched it, and it must fire."""
    sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
    from gpflowSlim.distributed import SingleComm, block_column_schedule
    rng = np.random.default_rng(5)
    n, nb = 96, 16
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)

    class Forgetful(NumpyPanelOps):
        def wait(self, lane, token):
            if lane == 0:
                return                                   # the CHAIN lane never waits for the BULK lane
            NumpyPanelOps.wait(self, lane, token)

    ops = Forgetful(A, nb, 1, 0)
    with pytest.raises(AssertionError, match="races"):
        block_column_schedule(ops, SingleComm(), ops.n_panels, lookahead=2)


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
    import torch.distributed as dist
    from gpflowSlim.distributed import TorchComm
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        counts = [3, 0, 5][:world]
        local = np.full((counts[rank], 2), float(rank)) + np.arange(counts[rank])[:, None]
        out = TorchComm().all_gather_rows(local, counts)
        q.put((rank, out.tolist()))
    finally:
        dist.destroy_process_group()


def test_all_gather_rows_ragged_gloo():
    """The gather behind predict_f_distributed: ragged (even empty) row blocks, same result on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [[0.0, 0.0], [1.0, 1.0], [2.0, 2.0]] + [[2.0 + i, 2.0 + i] for i in range(5)]
    for r in range(world):
        assert res[r] == expect


# ---- config 5 sharded over ranks: the reduction logic of gpflowSlim/distributed_sparse.py under gloo ----------------------
def _sparse_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle.gp_oracle as orc
    from gpflowSlim.distributed import TorchComm
    from gpflowSlim.distributed_sparse import shard_bounds, ordered_sum
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        comm = TorchComm(mode="broadcast")
        rng = np.random.default_rng(77)
        n, m, d, k = 211, 17, 2, 2
        X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) @ np.ones((1, k)) + 0.1 * rng.standard_normal((n, k))
        Z = X[:m].copy(); q_mu = 0.3 * rng.standard_normal((m, k)); q_sqrt = 0.5 + rng.random((m, k))
        spec = {"type": "rbf", "variance": 1.3, "lengthscales": 0.9, "input_dim": d}
        noise, num_data = 0.2, 3 * n
        b = shard_bounds(n, world)
        assert b[0] == 0 and b[-1] == n and all(0 <= (b[p + 1] - b[p]) - n // world <= 1 for p in range(world))
        lo, hi = b[rank], b[rank + 1]
        # what rank p evaluates: scale * sum_shard var_exp - KL / P   (models/svgp.py:108-125 is linear in the per-point terms)
        s = float(num_data) / n
        KL = orc.gauss_kl(q_mu, q_sqrt, None)
        part = orc.svgp_elbo(spec, X[lo:hi], Y[lo:hi], Z, q_mu, q_sqrt, noise, whiten=True, num_data=s * (hi - lo)) + KL * (1.0 - 1.0 / world)
        tot = ordered_sum(comm, [part, float(rank + 1)])
        ref = orc.svgp_elbo(spec, X, Y, Z, q_mu, q_sqrt, noise, whiten=True, num_data=num_data)
        # the SGPR partial sums: A A^T, A err, err^T err over shards add up to the whole (models/sgpr.py:138-143)
        import scipy.linalg as sl_
        L = np.linalg.cholesky(orc.K(spec, Z) + 1e-6 * np.eye(m))
        A = sl_.solve_triangular(L, orc.K(spec, Z, X[lo:hi]), lower=True)
        t = torch.from_numpy(np.concatenate([(A @ A.T).reshape(-1), (A @ Y[lo:hi]).reshape(-1), [np.sum(Y[lo:hi] ** 2), hi - lo]]))
        comm.all_reduce_sum(t)
        Aall = sl_.solve_triangular(L, orc.K(spec, Z, X), lower=True)
        want = np.concatenate([(Aall @ Aall.T).reshape(-1), (Aall @ Y).reshape(-1), [np.sum(Y ** 2), n]])
        # ragged gather incl. an empty block (more ranks than rows)
        rows = comm.all_gather_rows(np.full((1 if rank == 0 else 0, 3), 5.0), [1] + [0] * (world - 1))
        q.put((rank, float(tot[0]), float(tot[1]), float(ref), float(np.abs(t.numpy() - want).max() / np.abs(want).max()), rows.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sparse_reductions_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert len({r[1] for r in res}) == 1                       # bit-identical on every rank
    for rank, tot, ranks_sum, ref, err, shape in res:
        assert abs(tot - ref) <= 1e-12 * abs(ref)
        assert ranks_sum == world * (world + 1) / 2
        assert err <= 1e-13 and shape == (1, 3)
