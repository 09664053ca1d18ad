"""GPU unit tests of the individual HIP kernels, through the C ABI (include/gpflowslim_hip.h)."""
import numpy as np
import pytest
import scipy.linalg as sl

pytestmark = pytest.mark.gpu


def test_mfma_f64_layout_and_rate(handle):
    tf1, ok = handle.diag_mfma_f64(1)
    assert ok, "v_mfma_f64_16x16x4_f64 lane map differs from the one gemm_f64.hip assumes"
    tf2, _ = handle.diag_mfma_f64(2)
    print("fp64 MFMA issue rate: %.1f TFLOP/s (1 wave/SIMD), %.1f (2 waves/SIMD)" % (tf1, tf2))
    assert tf1 > 10.0


@pytest.mark.parametrize("m,n,k", [(128, 128, 16), (256, 384, 128), (384, 256, 400), (1024, 1024, 1024),
                                   (2048, 2304, 256)])
@pytest.mark.parametrize("op", [0, 1])
@pytest.mark.parametrize("tile", [0, 128, 64, 32])
def test_gemm_nt_full(handle, m, n, k, op, tile):
    handle.set_option("gemm_force_tile", tile)
    try:
        _gemm_nt_full(handle, m, n, k, op)
    finally:
        handle.set_option("gemm_force_tile", 0)


def _gemm_nt_full(handle, m, n, k, op):
    k = (k // 16) * 16
    rng = np.random.default_rng(m + n + k + op)
    A = rng.standard_normal((m, k)); B = rng.standard_normal((n, k)); C = rng.standard_normal((m, n))
    out = handle.diag_gemm_nt(op, False, A, B, C)
    ref = (C - A @ B.T) if op == 0 else A @ B.T
    # fp64: only summation order differs
    assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("m,n", [(128, 128), (384, 256), (1024, 640), (2048, 128)])
@pytest.mark.parametrize("op", [0, 1, 2, 3])
@pytest.mark.parametrize("tile", [0, 128, 64, 32])
def test_gemm_nt_triangular_a(handle, m, n, op, tile):
    """lower == 2: A [m, m] upper triangular, the kernel skips k < (first row of the tile); whatever sits below
    A's diagonal blocks must not be read (NaN there)."""
    rng = np.random.default_rng(m + n + op)
    A = np.triu(rng.standard_normal((m, m)))
    Apoison = A.copy()
    for ti in range(m // 128):
        Apoison[ti * 128:(ti + 1) * 128, :ti * 128] = np.nan
    B = rng.standard_normal((n, m)); C = rng.standard_normal((m, n))
    handle.set_option("gemm_force_tile", tile)
    try:
        out = handle.diag_gemm_nt(op, 2, Apoison, B, C)
    finally:
        handle.set_option("gemm_force_tile", 0)
    ref = {0: C - A @ B.T, 1: A @ B.T, 2: C + A @ B.T, 3: -(A @ B.T)}[op]
    assert np.isfinite(out).all()
    assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("m,n", [(128, 256), (512, 384), (1024, 1024), (2048, 128)])
@pytest.mark.parametrize("op", [1, 3])
@pytest.mark.parametrize("tile", [0, 128, 64, 32])
def test_gemm_nt_lower_triangular_operands(handle, m, n, op, tile):
    """Round 6 (the wide inverse blocks of predict_f): lower == 3: A [m, m] lower triangular, the kernel stops at the last
    column of the tile's rows; lower == 4: B [n, n] lower triangular (C [m, n] = A B^T with k <= column).  Whatever sits in
    the zero part of the triangular operand beyond its diagonal blocks must not be read (NaN there)."""
    rng = np.random.default_rng(m + n + op + tile)
    handle.set_option("gemm_force_tile", tile)
    try:
        A = np.tril(rng.standard_normal((m, m))); Ap = A.copy()
        for ti in range(m // 128):
            Ap[ti * 128:(ti + 1) * 128, (ti + 1) * 128:] = np.nan
        B = rng.standard_normal((n, m)); C = rng.standard_normal((m, n))
        out = handle.diag_gemm_nt(op, 3, Ap, B, C)
        ref = A @ B.T if op == 1 else -(A @ B.T)
        assert np.isfinite(out).all() and np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        Bt = np.tril(rng.standard_normal((n, n))); Bp = Bt.copy()
        for ti in range(n // 128):
            Bp[ti * 128:(ti + 1) * 128, (ti + 1) * 128:] = np.nan
        A2 = rng.standard_normal((m, n)); C2 = rng.standard_normal((m, n))
        out = handle.diag_gemm_nt(op, 4, A2, Bp, C2)
        ref = A2 @ Bt.T if op == 1 else -(A2 @ Bt.T)
        assert np.isfinite(out).all() and np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    finally:
        handle.set_option("gemm_force_tile", 0)


@pytest.mark.parametrize("m,n,k,tri", [(64, 128, 128, 0), (64, 128, 128, 4), (64, 2048, 2048, 0), (64, 2048, 2048, 4), (192, 384, 4096, 0), (64, 16384, 256, 0)])
@pytest.mark.parametrize("op", [0, 1])
def test_gemm_nt_half_tile_row(handle, m, n, k, tri, op):
    """M a multiple of 64 only (predict_f on at most 64 test points pads its right-hand sides to half a tile row): the 64 x 64 /
    32 x 32 tiles, with and without a lower-triangular B."""
    rng = np.random.default_rng(m + n + k + op + tri)
    A = rng.standard_normal((m, k)); B = rng.standard_normal((n, k)); C = rng.standard_normal((m, n))
    if tri == 4: B = np.tril(B)
    out = handle.diag_gemm_nt(op, tri, A, B, C)
    ref = C - A @ B.T if op == 0 else A @ B.T
    assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("batch,m,n,k,tri", [(1, 128, 128, 128, 0), (5, 128, 256, 128, 0), (16, 256, 256, 256, 1), (7, 512, 512, 512, 2),
                                            (3, 1024, 1024, 1024, 3), (64, 128, 128, 128, 3), (9, 384, 128, 384, 2)])
@pytest.mark.parametrize("op", [0, 1, 3])
def test_gemm_nt_batched(handle, batch, m, n, k, tri, op):
    """One launch over a batch of equal problems (what builds the 2048-column inverse blocks level by level), with and without a
    triangular operand: every problem equals its own numpy product."""
    rng = np.random.default_rng(batch + m + n + k + tri + op)
    A = rng.standard_normal((batch, m, k)); B = rng.standard_normal((batch, n, k)); C = rng.standard_normal((batch, m, n))
    if tri == 1: A = np.triu(A)
    if tri == 2: A = np.tril(A)
    if tri == 3: B = np.tril(B)
    out = handle.diag_gemm_nt_batched(op, tri, A, B, C)
    prod = np.einsum("pmk,pnk->pmn", A, B)
    ref = {0: C - prod, 1: prod, 3: -prod}[op]
    assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("op,lower,m,n,k", [(0, 0, 8192, 1152, 1024), (1, 0, 8192, 1152, 512), (2, 0, 9216, 1024, 256),
                                             (3, 0, 8192, 1152, 384), (0, 1, 4224, 4224, 256), (2, 1, 4480, 4480, 1024)])
def test_gemm_nt_tail_split(handle, op, lower, m, n, k):
    """Launches whose 128x128 tile count is a little above a multiple of the 512 resident workgroup slots: the
    tiles of the partial last round are cut into K-slices whose partial sums are added in a fixed order.  Same
    answer as one workgroup per tile (up to summation order), bit-identical from run to run."""
    rng = np.random.default_rng(m + n + k + op)
    A = rng.standard_normal((m, k)); B = A if lower else rng.standard_normal((n, k)); C = rng.standard_normal((m, n))
    ref = {0: C - A @ B.T, 1: A @ B.T, 2: C + A @ B.T, 3: -(A @ B.T)}[op]
    outs = {}
    handle.set_option("gemm_force_tile", 128)          # (the tile heuristic would pick 64x64 below 768 tiles)
    try:
        for split in (1, 0, 1):
            handle.set_option("gemm_tail_split", split)
            outs.setdefault(split, []).append(handle.diag_gemm_nt(op, lower, A, B, C))
    finally:
        handle.set_option("gemm_tail_split", 1)
        handle.set_option("gemm_force_tile", 0)
    mask = np.tril(np.ones((m, n), dtype=bool)) if lower else np.ones((m, n), dtype=bool)
    if lower:          # whole diagonal 128-blocks are computed; only their lower triangles are defined
        for out in outs[0] + outs[1]:
            out[~mask] = ref[~mask]
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(outs[0][0] - ref).max() <= 1e-11 * scale
    assert np.abs(outs[1][0] - ref).max() <= 1e-11 * scale
    assert np.array_equal(outs[1][0], outs[1][1])
    assert not np.array_equal(outs[1][0], outs[0][0]) or k <= 128       # the split really ran


@pytest.mark.parametrize("n,k", [(128, 64), (640, 128), (1152, 256), (2176, 512)])
@pytest.mark.parametrize("tile", [0, 128, 64, 32])
def test_gemm_nt_lower(handle, n, k, tile):
    handle.set_option("gemm_force_tile", tile)
    try:
        _gemm_nt_lower(handle, n, k)
    finally:
        handle.set_option("gemm_force_tile", 0)


def _gemm_nt_lower(handle, n, k):
    rng = np.random.default_rng(n + k)
    A = rng.standard_normal((n, k)); C = rng.standard_normal((n, n))
    out = handle.diag_gemm_nt(0, True, A, A, C)
    ref = C - A @ A.T
    T = 128
    for ti in range(n // T):
        for tj in range(n // T):
            blk = (slice(ti * T, (ti + 1) * T), slice(tj * T, (tj + 1) * T))
            if tj < ti:
                assert np.abs(out[blk] - ref[blk]).max() <= 1e-11 * np.abs(ref).max(), (ti, tj)
            elif tj == ti:       # diagonal block: only its lower triangle is defined
                assert np.abs(np.tril(out[blk] - ref[blk])).max() <= 1e-11 * np.abs(ref).max(), (ti, tj)
            else:
                assert np.array_equal(out[blk], C[blk]), "tile above the diagonal must stay untouched"


@pytest.mark.parametrize("n", [1, 2, 5, 64, 127, 128, 129, 300, 512, 1000, 2048])
def test_potrf_matches_lapack(handle, n):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n + 3))
    A = G @ G.T + 0.5 * np.eye(n)
    L = handle.potrf(A)
    ref = np.linalg.cholesky(A)
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L)), "upper triangle is zero-filled like tf.cholesky"
    assert np.abs(L - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(L @ L.T - A).max() <= 1e-12 * np.abs(A).max() * n


def test_potrf_reports_not_positive_definite(handle):
    import gpflowSlim
    n = 300
    rng = np.random.default_rng(0)
    G = rng.standard_normal((n, n))
    A = G @ G.T + np.eye(n)
    A[200, 200] = -1.0
    with pytest.raises(gpflowSlim.NotPositiveDefiniteError) as e:
        handle.potrf(A)
    assert "201" in str(e.value)          # LAPACK-style: first failing leading minor


@pytest.mark.parametrize("pos", [0, 1, 15, 16, 17, 31, 47, 48, 63, 64, 100, 111, 112, 127, 128, 129, 255, 256, 383])
def test_potrf_first_failing_minor_at_every_panel_position(handle, pos):
    """The base kernel factors 16 columns per step with several waves doing the diagonal block redundantly and the rows
    below in spare lanes; only one of them reports.  The first non-positive leading minor must come back whatever
    step, wave or lane it falls in (positions inside each of the 16-column steps and across 128-blocks)."""
    import gpflowSlim
    n = 384
    rng = np.random.default_rng(pos)
    G = rng.standard_normal((n, n + 5))
    A = G @ G.T + np.eye(n)
    # make the leading minor of order pos+1 the first singular/indefinite one: Schur complement pivot -> negative
    L = np.linalg.cholesky(A)
    A[pos, pos] -= 1.5 * L[pos, pos] ** 2
    with pytest.raises(gpflowSlim.NotPositiveDefiniteError) as e:
        handle.potrf(A)
    assert "order %d " % (pos + 1) in str(e.value), str(e.value)


@pytest.mark.parametrize("n", [640, 2048, 4224])
def test_potrf_sweep_and_recursion_agree(handle, n):
    """Diagonal blocks up to 4096 columns are factored by the right-looking sweep (groups of 1, 2 or 3 panels per
    remainder update), larger ones by the recursion; "potrf_rl_max" = 0 forces the recursion everywhere.  Same
    factor to rounding, whatever the schedule."""
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n + 3))
    A = G @ G.T + 0.5 * np.eye(n)
    ref = np.linalg.cholesky(A)
    try:
        for opts in ({"potrf_rl_max": 0}, {"potrf_rl_max": 4096, "potrf_rl_group": 1}, {"potrf_rl_max": 4096, "potrf_rl_group": 2},
                     {"potrf_rl_max": 1024, "potrf_rl_group": 3}):
            for k, v in opts.items():
                handle.set_option(k, v)
            L = handle.potrf(A)
            assert np.abs(L - ref).max() <= 1e-10 * np.abs(ref).max(), opts
    finally:
        handle.set_option("potrf_rl_max", 4096); handle.set_option("potrf_rl_group", 2)


@pytest.mark.parametrize("n,nrhs", [(1, 1), (7, 3), (128, 1), (200, 5), (513, 130), (1024, 1000)])
@pytest.mark.parametrize("trans", [False, True])
def test_trsm_lower(handle, n, nrhs, trans):
    rng = np.random.default_rng(n * 7 + nrhs)
    G = rng.standard_normal((n, n))
    L = np.linalg.cholesky(G @ G.T + n * np.eye(n))
    B = rng.standard_normal((n, nrhs))
    X = handle.trsm_lower(L, B, trans=trans)
    ref = sl.solve_triangular(L, B, lower=True, trans='T' if trans else 'N')
    assert np.abs(X - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n,nrhs", [(512, 64), (512, 300), (1024, 1000), (1536, 130), (2048, 64)])
@pytest.mark.parametrize("trans", [False, True])
@pytest.mark.parametrize("rows", [0, 32, 64, 65])
def test_trsm_512_column_solve_in_one_launch(handle, n, nrhs, trans, rows):
    """csrc/trsm_panel.hip: every 512-column node of the blocked solve (blocked.hpp::trsm_rec / trsm_rn_rec) as one launch,
    for every form of the launcher (32 rows per workgroup, 64 rows persistent, 64 rows), against scipy; and against the
    launch-by-launch form of the same solve."""
    rng = np.random.default_rng(n + nrhs)
    G = rng.standard_normal((n, n))
    L = np.linalg.cholesky(G @ G.T + n * np.eye(n))
    B = rng.standard_normal((n, nrhs))
    ref = sl.solve_triangular(L, B, lower=True, trans='T' if trans else 'N')
    try:
        handle.set_option("leaf_refine", 0)           # (refined leaves -- the default of this entry point -- stay launch by launch)
        handle.set_option("trsm_panel_rows", rows)
        before = handle.profile_get("gemm_f64")["launches"]
        X = handle.trsm_lower(L, B, trans=trans)
        mid = handle.profile_get("gemm_f64")["launches"]
        handle.set_option("trsm_panel", 0)
        X0 = handle.trsm_lower(L, B, trans=trans)
        after = handle.profile_get("gemm_f64")["launches"]
    finally:
        handle.set_option("trsm_panel", 1); handle.set_option("trsm_panel_rows", 0); handle.set_option("leaf_refine", -1)
    assert np.abs(X - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(X - X0).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    # six launches fewer per 512 columns
    assert (after - mid) - (mid - before) == 6 * (n // 512), (before, mid, after)


@pytest.mark.parametrize("trans", [False, True])
def test_trsm_tall_right_hand_sides_go_panel_by_panel(handle, trans):
    """blocked.hpp::tall_panels: 16 times more right-hand sides than columns (conditionals.py:87 at config 5's shape) -- the solve
    goes over 512-column panels left to right (right to left for L^T), one long-K update and one launch each; same result as the
    recursive halving and as scipy, fewer launches."""
    n, nrhs = 1024, 16384
    rng = np.random.default_rng(5)
    G = rng.standard_normal((n, n))
    L = np.linalg.cholesky(G @ G.T + n * np.eye(n))
    B = rng.standard_normal((n, nrhs))
    ref = sl.solve_triangular(L, B, lower=True, trans='T' if trans else 'N')
    try:
        handle.set_option("leaf_refine", 0)
        l0 = handle.profile_get("gemm_f64")["launches"]
        X = handle.trsm_lower(L, B, trans=trans)
        l1 = handle.profile_get("gemm_f64")["launches"]
        handle.set_option("trsm_tall_ratio", 0)
        X0 = handle.trsm_lower(L, B, trans=trans)
        l2 = handle.profile_get("gemm_f64")["launches"]
    finally:
        handle.set_option("trsm_tall_ratio", 16); handle.set_option("leaf_refine", -1)
    assert np.abs(X - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(X - X0).max() <= 1e-12 * np.abs(ref).max()
    assert (l1 - l0, l2 - l1) == (3, 3)          # two panels: one update + two solves either way at 1024 columns


def test_trsm_512_column_solve_many_rows(handle):
    """The same launch on 128 .. 64000 rows (every form, both directions; 16384 rows: one row block per workgroup of the
    persistent form, 24576: one or two, 64000: three or four): identical to rounding with the launch-by-launch solve of the
    same synthetic block (gps_diag_trsm512)."""
    for m in (128, 4096, 16384, 24576, 64000):
        for back in (False, True):
            for rows in (32, 64, 65):
                handle.set_option("trsm_panel_rows", rows)
                try:
                    _, diff = handle.diag_trsm512(m, back, True, reps=1)
                finally:
                    handle.set_option("trsm_panel_rows", 0)
                assert diff <= 1e-14, (m, back, rows, diff)


def test_trsm_512_column_solve_phase_stamps(handle):
    """gps_diag_trsm512_stamps: every row block leaves the times of its phases, in order, for both launch shapes."""
    for m, rows in ((4096, 32), (32768, 64)):
        handle.set_option("trsm_panel_rows", rows)
        try:
            us, st = handle.diag_trsm512_stamps(m, False, reps=1)
        finally:
            handle.set_option("trsm_panel_rows", 0)
        st = st[: m // rows]
        assert us > 0 and (st[:, 0] > 0).all()
        for off in (0, 16):
            assert (np.diff(st[:, off:off + 14], axis=1) >= 0).all()


def test_trsm_leaf_refined_residual(handle):
    """csrc/trsm_leaf.hip: the refined 128-column leaf (three triangular MFMA products in one launch) solves
    X L11^T = B and X L11 = B to a residual at rounding level for every row-tile size the launcher picks."""
    for m in (128, 4096, 8192, 16384):
        for upper in (False, True):
            _, res = handle.diag_trsm_leaf(m, 1, upper, reps=2)
            assert res <= 5e-16, (m, upper, res)
    _, res0 = handle.diag_trsm_leaf(1024, 0, False, reps=2)
    assert res0 <= 5e-15


def test_lookahead_timeout_is_retried(handle):
    """A missed hand-over of the look-ahead (here: injected -- the k-th join waits for a ticket that never comes and gives
    up after its 1 s bound) must not surface as an error: the evaluation is re-run once without look-ahead on the same
    handle, the result is the ordinary one, the event is counted, and the next evaluation uses the look-ahead again."""
    import os
    import gpflowSlim as gpf
    import oracle.gp_oracle as orc
    if os.environ.get("GPS_LOOKAHEAD") == "0":
        pytest.skip("the look-ahead is switched off by the environment")
    n, d = 8192, 4
    X, Y, _ = orc.synthetic_gpr_data(n, d, 0, seed=11)
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, lengthscales=1.5), obs_var=0.1)
    ref = m.compute_log_likelihood()
    handle.set_option("potrf_lookahead", 0)
    try:
        ref_plain = m.compute_log_likelihood()             # (other launch shapes: equal to rounding, not bit for bit)
    finally:
        handle.set_option("potrf_lookahead", 1)
    assert abs(ref_plain - ref) <= 1e-12 * abs(ref)
    before = handle.profile_get("lookahead_retries")["launches"]
    handle.set_option("la_fault_inject", 3)
    try:
        got = m.compute_log_likelihood()
    finally:
        handle.set_option("la_fault_inject", 0)
    assert got == ref_plain                                # what came back IS the evaluation without look-ahead
    assert handle.profile_get("lookahead_retries")["launches"] == before + 1
    assert m.compute_log_likelihood() == ref               # and the look-ahead is back on afterwards
    assert handle.profile_get("lookahead_retries")["launches"] == before + 1


def test_lookahead_stress_mixed_sizes_two_handles():
    """1000 evaluations of mixed sizes, shuffled, alternating between two handles on the one device: not a single
    hand-over time-out (every evaluation equals the first one of its size bit for bit, the retry counter stays 0)."""
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    rng = np.random.default_rng(5)
    sizes = [300, 1100, 2048, 2500, 4096, 5000]
    hs = [be.Handle(0), be.Handle(0)]
    data, ref = {}, {}
    kern = gpf.kernels.Matern32(3, variance=1.2, lengthscales=1.1)
    prog = kern._program(3)
    for n in sizes:
        X = rng.standard_normal((n, 3))
        data[n] = (X, np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1)))
    order = rng.choice(sizes, 1000)
    try:
        for i, n in enumerate(order):
            h = hs[i % 2]
            X, Y = data[int(n)]
            h.gpr_set_data(X, ("stress", int(n), i % 2))
            v = h.gpr_lml(prog, 0.1, Y)
            assert ref.setdefault(int(n), v) == v, (i, n, v, ref[int(n)])
        assert all(h.profile_get("lookahead_retries")["launches"] == 0 for h in hs)
    finally:
        for h in hs:
            h.close()


@pytest.mark.parametrize("n,r", [(256, 1), (300, 2), (1000, 1), (2500, 3), (4096, 1), (6000, 5)])
def test_trsv_wavefront_equals_recursive_substitution(handle, n, r):
    """L a = y and L^T a = y as one wavefront launch (trsv_wave.hip) against the recursive substitution of blocked.hpp
    (same factor, same block inverses: the two differ only in summation order) and against the oracle."""
    import gpflowSlim as gpf
    import oracle.gp_oracle as orc
    rng = np.random.default_rng(n + r)
    d = 3
    X = rng.standard_normal((n, d)); Y = rng.standard_normal((n, r))
    kern = gpf.kernels.Matern32(d, variance=1.3, lengthscales=0.9)
    spec = {"type": "matern32", "variance": orc.constrained(1.3), "lengthscales": orc.constrained(0.9), "input_dim": d}
    prog = kern._program(d)
    handle.gpr_set_data(X, ("wave", n, r))
    res = {}
    try:
        handle.set_option("gpr_aug_rows", 0)
        for wave in (1, 0):
            handle.set_option("trsv_wave", wave)
            res[wave] = handle.gpr_lml_grad(prog, 0.2, Y)
    finally:
        handle.set_option("trsv_wave", 1); handle.set_option("gpr_aug_rows", -1)
    ref = orc.gpr_lml(spec, X, Y, 0.2)
    assert abs(res[1][0] - ref) <= 1e-9 * abs(ref)
    assert abs(res[1][0] - res[0][0]) <= 1e-12 * abs(ref)
    kinv = np.linalg.solve(orc.K(spec, X) + 0.2 * np.eye(n), Y)
    assert np.abs(res[1][3] - kinv).max() <= 1e-9 * np.abs(kinv).max()                 # backward wavefront
    assert np.abs(res[1][3] - res[0][3]).max() <= 1e-12 * np.abs(kinv).max()
    assert np.abs(res[1][1] - res[0][1]).max() <= 1e-10 * max(1.0, np.abs(res[0][1]).max())
    assert handle.profile_get("trsv_wave_fallbacks")["launches"] == 0


@pytest.mark.parametrize("which", [1, 2])
def test_wavefront_give_up_is_retried_by_the_gradient(which):
    """A wavefront substitution that gives up (injected: the forward one of the likelihood, or the backward one of the
    gradient's K^-1 y) must not leave NaN gradients behind with status OK, nor a sticky counter for the next entry point:
    the call that caused it re-runs through the recursive substitution, the event is counted, the handle falls back."""
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    import oracle.gp_oracle as orc
    n, d = 6400, 3                                         # (above the augmented-row limit: both solves are wavefronts)
    X, Y, _ = orc.synthetic_gpr_data(n, d, 0, seed=21)
    prog = gpf.kernels.RBF(d, variance=1.1, lengthscales=1.4)._program(d)
    h = be.Handle(0)
    try:
        h.gpr_set_data(X, ("wavefault", which))
        ref = h.gpr_lml_grad(prog, 0.1, Y)
        assert h.profile_get("trsv_wave_fallbacks")["launches"] == 0
        h.set_option("wave_fault_inject", which)
        got = h.gpr_lml_grad(prog, 0.1, Y)
        assert h.profile_get("trsv_wave_fallbacks")["launches"] == 1
        for a, b in zip(got[:4], ref[:4]):
            a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
            assert np.isfinite(a).all() and np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
        # the next, unrelated entry point neither fails nor retries
        before = h.profile_get("lookahead_retries")["launches"]
        assert abs(h.gpr_lml(prog, 0.1, Y) - ref[0]) <= 1e-12 * abs(ref[0])
        assert h.profile_get("lookahead_retries")["launches"] == before
    finally:
        h.close()


def test_release_buffers_gives_memory_back_and_the_handle_stays_usable(handle):
    import torch
    import gpflowSlim as gpf
    import oracle.gp_oracle as orc
    n, d = 6000, 3
    X, Y, Xs = orc.synthetic_gpr_data(n, d, 10, seed=5)
    kern = gpf.kernels.RBF(d, variance=1.1, lengthscales=1.3)
    spec = {"type": "rbf", "variance": orc.constrained(1.1), "lengthscales": orc.constrained(1.3), "input_dim": d}
    m = gpf.models.GPR(X, Y, kern, obs_var=0.1)
    ref = orc.gpr_lml(spec, X, Y, orc.constrained(0.1))
    assert abs(m.compute_log_likelihood() - ref) <= 1e-8 * abs(ref)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    handle.release_buffers()
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 - free0 >= 6016 * 6016 * 8                    # at least K / L came back
    # the model notices that its data set is no longer resident and uploads it again; warm predict_f is refused a stale factor
    m.reuse_factor = True
    mu, var = m.predict_f(Xs)
    rmu, rvar = orc.gpr_predict(spec, X, Y, orc.constrained(0.1), Xs)
    assert np.abs(mu - rmu).max() <= 1e-8 * np.abs(rmu).max() and np.abs(var - rvar).max() <= 1e-8 * np.abs(rvar).max()
    assert abs(m.compute_log_likelihood() - ref) <= 1e-8 * abs(ref)


@pytest.mark.parametrize("n,m", [(70, None), (300, 190), (1000, None)])
def test_kernel_matrix_chain_kernel_equals_interpreter(handle, n, m):
    """Sum / Product of primitives folded left to right (kmat_chain_kernel: one accumulator, in-line exponential) against
    the stack interpreter on the same program: equal to rounding of the two exponentials."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(n)
    d = 5
    X = rng.standard_normal((n, d)); X2 = None if m is None else rng.standard_normal((m, d))
    ks = gpf.kernels
    cases = [ks.Matern52(d, variance=1.2, lengthscales=np.linspace(0.8, 1.7, d), ARD=True) + ks.Periodic(d, variance=0.7, lengthscales=1.3, period=2.1),
             ks.RBF(d, variance=0.9, lengthscales=1.1) * ks.Periodic(d, variance=1.1, lengthscales=0.9, period=1.7) + ks.White(d, variance=0.3) + ks.Constant(d, variance=0.25),
             ks.Matern32(d, variance=1.0, lengthscales=0.7) * ks.Matern12(d, variance=2.0, lengthscales=1.9) * ks.Exponential(d, variance=0.5, lengthscales=1.2)]
    for kern in cases:
        out = {}
        try:
            for fast in (1, 0):
                handle.set_option("kmat_fast", fast)
                out[fast] = kern.K(X) if X2 is None else kern.K(X, X2)
        finally:
            handle.set_option("kmat_fast", 1)
        assert out[1].shape == out[0].shape
        assert np.abs(out[1] - out[0]).max() <= 4e-15 * np.abs(out[0]).max()


def test_nan_inputs_propagate_through_the_kernel_matrix(handle):
    """tf.exp propagates NaN (kernels.py:439, 576-610): a NaN coordinate must poison its row and column of K on every build
    path (the exponent-injection exp returned 0 for a NaN argument), and the likelihood must not come back finite."""
    import gpflowSlim as gpf
    rng = np.random.default_rng(3)
    X = rng.standard_normal((300, 3)); X[17, 1] = np.nan
    for kern in (gpf.kernels.RBF(3, lengthscales=1.1), gpf.kernels.Matern52(3, lengthscales=0.9),
                 gpf.kernels.Matern32(3) + gpf.kernels.Periodic(3, period=2.0)):
        K = kern.K(X)
        bad = np.isnan(K)
        assert bad[17, :].all() and bad[:, 17].all() and not np.delete(np.delete(bad, 17, 0), 17, 1).any()
    Y = rng.standard_normal((300, 1))
    for small in (1, 0):
        handle.set_option("small_n", small)
        try:
            m = gpf.models.GPR(X, Y, gpf.kernels.RBF(3, lengthscales=1.1), obs_var=0.1)
            try:
                v = m.compute_log_likelihood()
            except gpf.NotPositiveDefiniteError:
                v = np.nan
            assert not np.isfinite(v)
        finally:
            handle.set_option("small_n", 1)


def test_overflowed_distances_give_zero_not_nan(handle):
    """tf.exp(-inf) = 0 (kernels.py:439): points so far apart that their squared distance overflows have covariance 0 under RBF
    on every build path -- the one-primitive kernel, the chains on the matrix pipe (its lean exponential gave
    NaN there), the chain kernel and the interpreter.  (Matern-3/2 and -5/2 give NaN there in the reference too:
    (1 + sqrt(3) inf) * exp(-inf) = inf * 0, kernels.py:592-594, 608-610.)"""
    import gpflowSlim as gpf
    rng = np.random.default_rng(5)
    # (|x|^2 = 6e307 .. 7e307 is finite and so is the sum of two of them; the squared distance between the two far points, 4 |x|^2,
    # overflows: r^2 = +inf, exp(-inf) = 0.  Larger coordinates would make the DIAGONAL inf - inf = NaN in the reference's formula too.)
    X = rng.standard_normal((200, 3)); X[11] = 4.9e153; X[12] = -4.9e153
    for opt, kern in ((None, gpf.kernels.RBF(3, lengthscales=1.1)), (None, gpf.kernels.RBF(3, lengthscales=0.9) + gpf.kernels.RBF(3, variance=0.5)),
                      (("kmat_mfma", 0), gpf.kernels.RBF(3, lengthscales=0.9) + gpf.kernels.RBF(3, variance=0.5)),
                      (("kmat_fast", 0), gpf.kernels.RBF(3, lengthscales=0.9) * gpf.kernels.RBF(3, variance=0.5))):
        if opt: handle.set_option(opt[0], opt[1])
        try:
            K = kern.K(X)
        finally:
            handle.set_option("kmat_mfma", 1); handle.set_option("kmat_fast", 1)
        far = np.zeros(200, dtype=bool); far[[11, 12]] = True
        assert np.isfinite(K).all(), (opt, np.argwhere(~np.isfinite(K))[:5])
        assert (K[far][:, ~far] == 0).all() and K[11, 12] == 0 and (np.diag(K) > 0).all()


@pytest.mark.parametrize("n,r", [(1, 1), (2, 1), (50, 1), (128, 2), (129, 1), (300, 3), (455, 1), (512, 1), (640, 2), (768, 1), (896, 1), (1000, 1), (1300, 2), (2048, 1), (2100, 1)])
def test_one_launch_factorisation_of_small_problems(handle, n, r):
    """Problems of up to 2048 padded rows (the reference's own example is N ~ 455, examples/gpr.py:36) are factored by ONE
    cooperative launch (csrc/small_n.hip): chain workgroup + slab workgroups, hand-overs through counters.  Same likelihood,
    predictions and gradient as the launch-by-launch path and as the oracle; three launches per evaluation; no fall-back."""
    import gpflowSlim as gpf
    import oracle.gp_oracle as orc
    d = 3
    X, Y, Xs = orc.synthetic_gpr_data(n, d, 17, seed=100 + n)
    if r > 1:
        Y = np.concatenate([Y * (q + 1) + 0.1 * q for q in range(r)], axis=1)
    ls = np.linspace(0.8, 1.5, d)
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, variance=1.3, lengthscales=ls, ARD=True), obs_var=0.1)
    spec = {"type": "rbf", "variance": orc.constrained(1.3), "lengthscales": orc.constrained(ls), "input_dim": d}
    ref = orc.gpr_lml(spec, X, Y, orc.constrained(0.1))
    before = handle.profile_get("small_n_fallbacks")["launches"]
    res = {}
    for small in (1, 0):
        handle.set_option("small_n", small)
        handle.profile_reset()
        lml = m.compute_log_likelihood()
        launches = sum(handle.profile_get(k)["launches"] for k in ("gemm_f64", "potrf_base", "kmat", "trsv", "reduce", "other"))
        mu, var = m.predict_f(Xs)                            # cold: factors again, then needs the transposed inverses
        lg, grads = m.compute_log_likelihood_and_gradients()
        res[small] = (lml, mu, var, lg, np.concatenate([np.ravel(g) for _, g in grads]), launches)
    handle.set_option("small_n", 1)
    one, many = res[1], res[0]
    assert abs(one[0] - ref) <= 1e-8 * abs(ref) and abs(one[0] - many[0]) <= 1e-11 * abs(ref)
    if n <= 2048:
        assert one[5] <= 3 and (many[5] > one[5] or n <= 128), (one[5], many[5])          # the factorisation launch (+ kmat prep + kmat for programs it does not generate itself)
    rmu, rvar = orc.gpr_predict(spec, X, Y, orc.constrained(0.1), Xs)
    assert np.abs(one[1] - rmu).max() <= 1e-8 * max(1.0, np.abs(rmu).max()) and np.abs(one[2] - rvar).max() <= 1e-8 * np.abs(rvar).max()
    assert np.abs(one[1] - many[1]).max() <= 1e-10 * max(1.0, np.abs(rmu).max())
    assert abs(one[3] - many[3]) <= 1e-11 * abs(ref) and np.abs(one[4] - many[4]).max() <= 1e-9 * max(1.0, np.abs(many[4]).max())
    assert handle.profile_get("small_n_fallbacks")["launches"] == before


@pytest.mark.parametrize("n", [300, 512, 768, 1500])
def test_small_launch_that_gives_up_is_redone_launch_by_launch(handle, n):
    """A bounded wait of a cooperative small-N launch that runs out (injected: "small_fault_inject" = k makes the k-th such launch
    start with its abort word set) must not surface: the evaluation comes back through the launch-by-launch path with the same
    values, the give-up is counted, and the next evaluation takes the one-launch path again."""
    import gpflowSlim as gpf
    import oracle.gp_oracle as orc
    d = 3
    X, Y, _ = orc.synthetic_gpr_data(n, d, 0, seed=7 + n)
    ls = np.linspace(0.9, 1.4, d)
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, variance=1.1, lengthscales=ls, ARD=True), obs_var=0.1)
    spec = {"type": "rbf", "variance": orc.constrained(1.1), "lengthscales": orc.constrained(ls), "input_dim": d}
    ref = orc.gpr_lml(spec, X, Y, orc.constrained(0.1))
    lml0 = m.compute_log_likelihood()
    lg0, grads0 = m.compute_log_likelihood_and_gradients()
    g0 = np.concatenate([np.ravel(g) for _, g in grads0])
    count = lambda: handle.profile_get("small_n_fallbacks")["launches"]
    base = count()
    # the likelihood alone: its one cooperative launch gives up
    handle.set_option("small_fault_inject", 1)
    assert abs(m.compute_log_likelihood() - ref) <= 1e-8 * abs(ref) and count() == base + 1
    # likelihood + gradient: the factorisation launch gives up / the inverse launch gives up
    for k in (1, 2):
        handle.set_option("small_fault_inject", k)
        lg, grads = m.compute_log_likelihood_and_gradients()
        g = np.concatenate([np.ravel(x) for _, x in grads])
        assert abs(lg - lg0) <= 1e-11 * abs(ref) and np.abs(g - g0).max() <= 1e-9 * max(1.0, np.abs(g0).max()), k
    assert count() == base + 3
    # and the fast path is back
    handle.profile_reset()
    assert abs(m.compute_log_likelihood() - lml0) <= 1e-13 * abs(ref)
    assert sum(handle.profile_get(k)["launches"] for k in ("gemm_f64", "potrf_base", "kmat", "trsv", "reduce", "other")) <= 3
    assert count() == base + 3


def test_one_launch_factorisation_reports_not_positive_definite(handle):
    import gpflowSlim as gpf
    X = np.zeros((300, 2)); X[:, 0] = np.arange(300) % 7; Y = np.ones((300, 1))
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(2), obs_var=1e-30, min_var=0.0)
    m.likelihood._variance.transform._lower = -1.0     # a negative "variance" on the diagonal: duplicate points -> singular
    handle.set_option("small_n", 1)
    with pytest.raises(gpf.NotPositiveDefiniteError):
        m.compute_log_likelihood()
    # ... and the handle evaluates a proper problem right afterwards (counters back in order)
    import oracle.gp_oracle as orc
    Xg, Yg, _ = orc.synthetic_gpr_data(400, 2, 0, seed=5)
    mg = gpf.models.GPR(Xg, Yg, gpf.kernels.RBF(2, lengthscales=1.2), obs_var=0.1)
    spec = {"type": "rbf", "variance": orc.constrained(1.0), "lengthscales": orc.constrained(1.2), "input_dim": 2}
    ref = orc.gpr_lml(spec, Xg, Yg, orc.constrained(0.1))
    assert abs(mg.compute_log_likelihood() - ref) <= 1e-8 * abs(ref)


_CONC_REF = {}


def _concurrent_reference(n, d):
    """Inputs + LML, dLML/d(variance, lengthscales) and K_y^-1 r of an RBF(ARD) GPR from the oracle's kernel matrix and LAPACK."""
    if n in _CONC_REF:
        return _CONC_REF[n]
    import oracle.gp_oracle as orc
    rng = np.random.default_rng(5 + n)
    X = rng.standard_normal((n, d)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((n, 1))
    var, ls, noise = 1.1, np.sqrt(d) * np.linspace(0.8, 1.2, d), 0.1
    spec = {"type": "rbf", "variance": var, "lengthscales": ls, "input_dim": d}
    K = orc.K(spec, X)
    Ky = K + noise * np.eye(n)
    c = sl.cho_factor(Ky, lower=True)
    a = sl.cho_solve(c, Y)
    Kinv = sl.cho_solve(c, np.eye(n))
    lml = orc.gpr_lml(spec, X, Y, noise)
    G = 0.5 * (a @ a.T - Kinv)
    g = [np.sum(G * K) / var]
    for q in range(d):
        dq = (X[:, q:q + 1] - X[:, q:q + 1].T) ** 2
        g.append(np.sum(G * K * dq) / ls[q] ** 3)
    _CONC_REF[n] = (X, Y, var, ls, noise, lml, np.array(g), float(np.trace(G)), a)
    return _CONC_REF[n]


@pytest.mark.parametrize("n", [300, 455, 512, 896, 1300, 2000])
@pytest.mark.parametrize("threads", [4, 8])
def test_concurrent_small_launches(n, threads):
    """The cooperative launches of csrc/small_n.hip (every workgroup draws its task from a counter; hand-overs through counters
    in HBM) with 4 / 8 of them in flight at once: as many handles in as many host threads, each running >= 200 likelihood +
    gradient evaluations (the reference's loop: examples/gpr.py:48-61).  Round 4's first queued form stalled for its full
    bounded wait in this regime and once returned a wrong gradient without reporting a give-up; only a tool looked for that.
    Every thread: bit-identical results step after step, equal to the oracle / LAPACK to 1e-8, no fall-back, no stalled step."""
    import threading
    import time
    import gpflowSlim as gpf
    from gpflowSlim import _backend as be
    d, steps = 8, 200
    X, Y, var, ls, noise, lml_ref, g_ref, gn_ref, a_ref = _concurrent_reference(n, d)
    prog = gpf.kernels.RBF(d, variance=var, lengthscales=ls, ARD=True)._program(d)
    out, errors = {}, []

    def run(t):
        try:
            h = be.Handle(0)
            h.gpr_set_data(X, X)
            first, t_max, slow, same = None, 0.0, 0, True
            for i in range(steps):
                t0 = time.perf_counter()
                lml, slots, gn, kr = h.gpr_lml_grad(prog, noise, Y)
                dt = time.perf_counter() - t0
                if i >= 3:
                    t_max = max(t_max, dt); slow += dt > 0.05
                cur = (lml, np.array(slots, copy=True), gn, np.array(kr, copy=True))
                if first is None:
                    first = cur
                else:
                    same = same and cur[0] == first[0] and cur[2] == first[2] and np.array_equal(cur[1], first[1]) and np.array_equal(cur[3], first[3])
            out[t] = (first, t_max, slow, same, h.profile_get("small_n_fallbacks")["launches"], h.profile_get("small_n_cooldown")["launches"])
            h.close()
        except Exception as e:       # (surfaced by the main thread)
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=run, args=(t,)) for t in range(threads)]
    for t in ths: t.start()
    for t in ths: t.join()
    assert not errors, errors
    assert sorted(out) == list(range(threads))
    for t in range(threads):
        (lml, slots, gn, kr), t_max, slow, same, fallbacks, cooldown = out[t]
        assert same, "thread %d: results changed between steps" % t
        assert abs(lml - lml_ref) <= 1e-8 * abs(lml_ref)
        assert np.abs(np.ravel(slots)[:1 + d] - g_ref).max() <= 1e-8 * max(1.0, np.abs(g_ref).max())
        assert abs(gn - gn_ref) <= 1e-8 * max(1.0, abs(gn_ref))
        assert np.abs(kr - a_ref).max() <= 1e-8 * np.abs(a_ref).max()
        assert fallbacks == 0 and cooldown == 0, (t, fallbacks, cooldown)
        assert t_max < 0.25 and slow <= 2, "thread %d: slowest step %.1f ms, %d steps over 50 ms" % (t, 1e3 * t_max, slow)
        # every thread computes the same bits as every other
        assert out[t][0][0] == out[0][0][0] and np.array_equal(out[t][0][1], out[0][0][1])


@pytest.mark.parametrize("n", [8192, 9000, 12288])
def test_follower_solve_launch_caps_agree(handle, n):
    """The solve that follows a sweep on the second stream (blocked.hpp::potrf_rl_groups, right-looking): its updates whole or in
    launches of at most 256 / 64 tiles (option "follower_max_wgs"), and the sweep without any look-ahead: same likelihood and
    predictions (the pieces are summed in another order: 1e-11), no evaluation re-run without look-ahead -- sizes with one (8192),
    with a ragged (9000) and with two (12288) followed sweeps."""
    import gpflowSlim as gpf
    import oracle.gp_oracle as orc
    X, Y, Xs = orc.synthetic_gpr_data(n, 6, 64, seed=n)
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(6, lengthscales=2.0) + gpf.kernels.Matern32(6, variance=0.3, lengthscales=1.1), obs_var=0.07)
    res = {}
    try:
        for la, cap in ((0, 0), (1, 0), (1, 256), (1, 64)):
            handle.set_option("potrf_lookahead", la); handle.set_option("follower_max_wgs", cap)
            before = handle.profile_get("lookahead_retries")["launches"]
            lml = m.compute_log_likelihood()
            mu, var = m.predict_f(Xs)
            res[(la, cap)] = (lml, mu, var)
            assert handle.profile_get("lookahead_retries")["launches"] == before
    finally:
        handle.set_option("potrf_lookahead", 1); handle.set_option("follower_max_wgs", 256)
    a = res[(0, 0)]
    for key, b in res.items():
        assert abs(a[0] - b[0]) <= 1e-11 * abs(a[0]), key
        assert np.abs(a[1] - b[1]).max() <= 1e-10 * max(1.0, np.abs(a[1]).max()) and np.abs(a[2] - b[2]).max() <= 1e-10 * np.abs(a[2]).max(), key
