"""world_size > 1 on CPU (gloo): the block-column schedule of gpflowSlim/distributed.py -- the SAME
function the GPU path runs -- with the per-step pieces emulated in numpy/scipy (test infrastructure
only) and the panel broadcast carried by torch.distributed/gloo.  Checks ownership, message sizes,
look-ahead ordering and that every rank ends with the full factor."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyPanelOps(object):
    """Emulates gps_dist_* on the host: full [np, np] buffer per rank, only owned block columns are
    initialised (everything else NaN, so any use of data that was never received shows up)."""

    def __init__(self, A, nb, nparts, part):
        import torch
        self.torch = torch
        self.np_, self.nb, self.P, self.rank = A.shape[0], nb, nparts, part
        self.n_panels = self.np_ // nb
        self.M = np.full_like(A, np.nan)
        for c in range(part, self.n_panels, nparts):
            self.M[c * nb:, c * nb:(c + 1) * nb] = A[c * nb:, c * nb:(c + 1) * nb]
        mx = self.np_ * nb
        self.bufs = [torch.full((mx,), float("nan"), dtype=torch.float64) for _ in range(2)]
        self.log = []

    def panel_factor(self, j, buf):
        nb, s = self.nb, j * self.nb
        assert j % self.P == self.rank, "only the owner factors a panel"
        D = self.M[s:s + nb, s:s + nb]
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
        rows = self.np_ - s
        self.M[s:, s:s + nb] = self.bufs[buf][: rows * nb].numpy().reshape(rows, nb)
        self.log.append(("unpack", j))

    def update(self, j, c_lo, c_hi):
        nb = self.nb
        for c in range(max(c_lo, j + 1), min(c_hi, self.n_panels)):
            if c % self.P != self.rank:
                continue
            Lc = self.M[c * nb:, j * nb:(j + 1) * nb]
            assert not np.isnan(Lc).any(), "panel %d used before it was received" % j
            self.M[c * nb:, c * nb:(c + 1) * nb] -= Lc @ Lc[:nb].T
            self.log.append(("update", j, c))


def _worker(rank, world, port, n, nb, lookahead, q):
    sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
    import torch.distributed as dist
    from gpflowSlim.distributed import TorchComm, block_column_schedule
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)
        G = rng.standard_normal((n, n))
        A = G @ G.T + n * np.eye(n)
        ops = NumpyPanelOps(A, nb, world, rank)
        block_column_schedule(ops, TorchComm(), ops.n_panels, lookahead=lookahead)
        L = np.linalg.cholesky(A)
        err = float(np.abs(np.tril(ops.M) - L).max())
        factored = [e[1] for e in ops.log if e[0] == "factor"]
        q.put((rank, err, factored))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,n,nb,lookahead", [(2, 96, 16, True), (2, 96, 16, False), (3, 112, 16, True), (2, 48, 48, True)])
def test_block_column_schedule_gloo(world, n, nb, lookahead):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, lookahead, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_panels = n // nb
    for rank, err, factored in res:
        assert err <= 1e-10, (rank, err)                       # every rank holds the whole factor
        assert factored == list(range(rank, n_panels, world))  # block-cyclic ownership


def test_schedule_single_rank_matches_lapack():
    sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))
    from gpflowSlim.distributed import SingleComm, block_column_schedule
    rng = np.random.default_rng(5)
    n, nb = 80, 16
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
    ops = NumpyPanelOps(A, nb, 1, 0)
    block_column_schedule(ops, SingleComm(), ops.n_panels)
    assert np.abs(np.tril(ops.M) - np.linalg.cholesky(A)).max() <= 1e-10


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
