"""The recursive blocked driver (gpflow-slim_amd/csrc/blocked.hpp) instantiated with naive host
loops (tests/cpu_blocked/emul.cpp, test infrastructure only) and checked against LAPACK: validates
the block / index arithmetic of potrf_rec, trsm_rec, trsm_rn_rec and trsv_rec without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(HERE, "cpu_blocked", "libemul.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "cpu_blocked", "emul.cpp")])
    lib = ctypes.CDLL(so)
    return lib


@pytest.mark.parametrize("rl_max,group", [(0, 1), (256, 1), (1024, 1), (256, 2), (1024, 2), (1024, 3), (1024, 4), (1024, -2), (1024, -3), (256, -2), (384, -2), (512, -3), (512, -12), (256, -12)])
@pytest.mark.parametrize("n,m,r", [(128, 128, 1), (256, 128, 2), (384, 256, 3), (640, 128, 1), (896, 128, 2)])
def test_blocked_recursion_matches_lapack(emul, n, m, r, rl_max, group):
    emul.emul_set_rl_max(ctypes.c_int64(rl_max))      # diagonal blocks up to rl_max: right-looking sweep (potrf_rl)
    # negative group: with the look-ahead split of the remainder update and the follower solve (-12: in 512-column pieces)
    emul.emul_set_lookahead(2 if group == -12 else (1 if group < 0 else 0))
    group = 2 if group == -12 else abs(group)
    emul.emul_set_rl_group(ctypes.c_int64(group))     # ... updating the remainder once per group of panels (potrf_rl_groups)
    rng = np.random.default_rng(n + m)
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
    B = rng.standard_normal((m, n)); B2 = B.copy(); y = rng.standard_normal((r, n))
    A0, B0, y0 = A.copy(), B.copy(), y.copy()
    info = ctypes.c_int(0)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    rc = emul.emul_all(p(A), ctypes.c_int64(n), p(B), p(B2), ctypes.c_int64(m), p(y), ctypes.c_int64(r), ctypes.byref(info))
    assert rc == 0 and info.value == 0
    L = sl.cholesky(A0, lower=True)
    assert np.abs(np.tril(A) - L).max() <= 1e-12 * np.abs(L).max()
    assert np.abs(B - sl.solve_triangular(L, B0.T, lower=True).T).max() <= 1e-12
    assert np.abs(B2 - sl.solve_triangular(L, B0.T, lower=True, trans='T').T).max() <= 1e-12
    assert np.abs(y - sl.solve_triangular(L, y0.T, lower=True).T).max() <= 1e-12


@pytest.mark.parametrize("n,m", [(512, 128), (640, 128), (1024, 256), (1536, 128), (1920, 256), (2048, 128)])
@pytest.mark.parametrize("rl_max,la,tall", [(0, 0, 0), (512, 2, 0), (1024, 2, 0), (1024, 2, 1)])
def test_512_column_nodes_of_the_solves(emul, n, m, rl_max, la, tall):
    """blocked.hpp::split_solve / Ops::trsm_leaf512: with the 512-column node as one operation (emulated the way
    csrc/trsm_panel.hip does it) every solve of the driver still matches LAPACK, whatever the number of tiles; tall: the
    solves of a whole number of 512-column panels go panel by panel, left-looking (blocked.hpp::tall_panels), both directions."""
    emul.emul_set_rl_max(ctypes.c_int64(rl_max))
    emul.emul_set_lookahead(la)
    emul.emul_set_rl_group(ctypes.c_int64(2))
    emul.emul_set_leaf512(1)
    emul.emul_set_tall(tall)
    try:
        rng = np.random.default_rng(n + m)
        G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
        B = rng.standard_normal((m, n)); B2 = B.copy(); y = rng.standard_normal((1, n))
        A0, B0, y0 = A.copy(), B.copy(), y.copy()
        info = ctypes.c_int(0)
        p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        rc = emul.emul_all(p(A), ctypes.c_int64(n), p(B), p(B2), ctypes.c_int64(m), p(y), ctypes.c_int64(1), ctypes.byref(info))
    finally:
        emul.emul_set_leaf512(0); emul.emul_set_lookahead(0); emul.emul_set_tall(0)
    assert rc == 0 and info.value == 0
    L = sl.cholesky(A0, lower=True)
    assert np.abs(np.tril(A) - L).max() <= 1e-12 * np.abs(L).max()
    assert np.abs(B - sl.solve_triangular(L, B0.T, lower=True).T).max() <= 1e-12
    assert np.abs(B2 - sl.solve_triangular(L, B0.T, lower=True, trans='T').T).max() <= 1e-12


@pytest.mark.parametrize("rl_max,group", [(0, 1), (512, 1), (512, 2), (512, 3)])
def test_blocked_recursion_reports_first_bad_pivot(emul, rl_max, group):
    emul.emul_set_rl_max(ctypes.c_int64(rl_max))
    emul.emul_set_lookahead(0)
    emul.emul_set_rl_group(ctypes.c_int64(group))
    n = 384
    rng = np.random.default_rng(1)
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
    A[300, 300] = -5.0
    B = np.zeros((128, n)); B2 = B.copy(); y = np.zeros((1, n))
    info = ctypes.c_int(0)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    emul.emul_all(p(A), ctypes.c_int64(n), p(B), p(B2), ctypes.c_int64(128), p(y), ctypes.c_int64(1), ctypes.byref(info))
    assert info.value == 301


@pytest.mark.parametrize("n,r", [(128, 1), (256, 2), (384, 1), (640, 3)])
def test_gradient_pieces_match_lapack(emul, n, r):
    """trsv_t_rec (L^T a = y), inv_t_rec (Y = L^-T) and lauum_rec (K^-1 = Y Y^T)."""
    emul.emul_set_rl_max(ctypes.c_int64(0))
    rng = np.random.default_rng(n + r)
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
    A0 = A.copy()
    yt = rng.standard_normal((r, n)); y0 = yt.copy()
    Y = np.zeros((n, n)); Kinv = np.zeros((n, n))
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    rc = emul.emul_grad_pieces(p(A), ctypes.c_int64(n), p(yt), ctypes.c_int64(r), p(Y), p(Kinv))
    assert rc == 0
    L = sl.cholesky(A0, lower=True)
    assert np.abs(yt - sl.solve_triangular(L, y0.T, lower=True, trans='T').T).max() <= 1e-12
    Linv = np.linalg.inv(L)
    assert np.abs(Y - Linv.T).max() <= 1e-12 * np.abs(Linv).max()
    ref = np.linalg.inv(A0)
    assert np.abs(np.tril(Kinv) - np.tril(ref)).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("n,nb,rl_max", [(896, 256, 0), (896, 384, 256), (640, 128, 0)])
def test_panel_sweep_matches_lapack(emul, n, nb, rl_max):
    """potrf_rl with panels wider than one block (factored by potrf_rec, rows below solved by trsm_rec)."""
    emul.emul_set_rl_max(ctypes.c_int64(rl_max))
    rng = np.random.default_rng(n + nb)
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n); A0 = A.copy()
    info = ctypes.c_int(0)
    rc = emul.emul_potrf_rl(A.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_int64(n), ctypes.c_int64(nb), ctypes.byref(info))
    assert rc == 0 and info.value == 0
    L = sl.cholesky(A0, lower=True)
    assert np.abs(np.tril(A) - L).max() <= 1e-12 * np.abs(L).max()


@pytest.mark.parametrize("n,rl_max,la", [(1024, 512, 1), (1536, 512, 2), (2048, 1024, 1), (2048, 1024, 2), (2560, 1024, 2), (2048, 512, 2)])
def test_follower_solve_right_looking(emul, n, rl_max, la):
    """blocked.hpp::potrf_rl_groups: the solve of the block below a swept diagonal block follows the sweep piece by piece,
    right-looking (a solved piece is applied to all the columns after it).  Same factor as LAPACK's (the block below IS part of it)."""
    emul.emul_set_rl_max(ctypes.c_int64(rl_max))
    emul.emul_set_lookahead(la)
    emul.emul_set_rl_group(ctypes.c_int64(2))
    try:
        rng = np.random.default_rng(n + 2)
        G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
        B = rng.standard_normal((128, n)); B2 = B.copy(); y = rng.standard_normal((1, n))
        A0 = A.copy()
        info = ctypes.c_int(0)
        p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        rc = emul.emul_all(p(A), ctypes.c_int64(n), p(B), p(B2), ctypes.c_int64(128), p(y), ctypes.c_int64(1), ctypes.byref(info))
    finally:
        emul.emul_set_lookahead(0)
    assert rc == 0 and info.value == 0
    L = sl.cholesky(A0, lower=True)
    assert np.abs(np.tril(A) - L).max() <= 1e-12 * np.abs(L).max()


@pytest.mark.parametrize("n,rl_max", [(1536, 384), (1280, 256), (2048, 512)])
def test_deferred_piece_of_the_parent_solve(emul, n, rl_max):
    """Nodes whose first half is itself a node with a swept first half hand the first part of their panel solve down
    (blocked.hpp: Deferred): the child issues it once those columns are final, the parent skips it.  Index check with
    the look-ahead hooks on (the emulation's hooks verify that every open / close / join pairs up)."""
    emul.emul_set_rl_max(ctypes.c_int64(rl_max))
    emul.emul_set_lookahead(1)
    emul.emul_set_rl_group(ctypes.c_int64(2))
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
    B = rng.standard_normal((128, n)); B2 = B.copy(); y = rng.standard_normal((1, n))
    A0, B0, y0 = A.copy(), B.copy(), y.copy()
    info = ctypes.c_int(0)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    rc = emul.emul_all(p(A), ctypes.c_int64(n), p(B), p(B2), ctypes.c_int64(128), p(y), ctypes.c_int64(1), ctypes.byref(info))
    assert rc == 0 and info.value == 0
    L = sl.cholesky(A0, lower=True)
    assert np.abs(np.tril(A) - L).max() <= 1e-12 * np.abs(L).max()
    assert np.abs(y - sl.solve_triangular(L, y0.T, lower=True).T).max() <= 1e-12
    emul.emul_set_lookahead(0)


@pytest.mark.parametrize("rl_max,group,la", [(0, 1, 0), (256, 1, 0), (512, 2, 0), (512, 2, 1), (256, 2, 2), (1024, 3, 0), (384, 2, 1)])
@pytest.mark.parametrize("n,e", [(128, 128), (256, 128), (640, 128), (1152, 256), (1536, 128)])
def test_augmented_rows_ride_through_the_factorisation(emul, n, e, rl_max, group, la):
    """potrf_rec with e augmented rows stored under the matrix (the residual^T of gps_gpr_lml): on return the rows hold
    E L^-T, i.e. the forward substitution came out of the panel solves and (trapezoidal) trailing updates -- in every
    variant of the recursion (plain, sweeps, groups, look-ahead / follower / deferred hooks)."""
    emul.emul_set_rl_max(ctypes.c_int64(rl_max))
    emul.emul_set_lookahead(la)
    emul.emul_set_rl_group(ctypes.c_int64(group))
    rng = np.random.default_rng(n + e + rl_max)
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
    E = rng.standard_normal((e, n))
    M = np.ascontiguousarray(np.vstack([A, E]))
    info = ctypes.c_int(0)
    rc = emul.emul_potrf_aug(M.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_int64(n), ctypes.c_int64(e), ctypes.byref(info))
    emul.emul_set_lookahead(0)
    assert rc == 0 and info.value == 0
    L = sl.cholesky(A, lower=True)
    assert np.abs(np.tril(M[:n]) - L).max() <= 1e-12 * np.abs(L).max()
    ref = sl.solve_triangular(L, E.T, lower=True).T                     # E L^-T
    assert np.abs(M[n:] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n,m,e", [(256, 128, 0), (640, 128, 128), (1024, 256, 0), (1408, 128, 128)])
@pytest.mark.parametrize("rl_max,group,la", [(256, 2, 1), (512, 2, 1), (1024, 2, 2), (1024, 3, 1), (2048, 4, 2)])
def test_sweep_look_ahead_as_index_logic(emul, n, m, e, rl_max, group, la):
    """blocked.hpp::potrf_rl_groups with the look-ahead on: which rows, which earlier panels of the group, where the fork ticket
    and the joins go -- against LAPACK, with and without augmented rows, followers on; the emulation's race detector checks that
    the chain touches nothing a side section wrote (or writes what it read) before the matching join."""
    emul.emul_set_rl_max(ctypes.c_int64(rl_max))
    emul.emul_set_lookahead(la)
    emul.emul_set_rl_group(ctypes.c_int64(group))
    emul.emul_side_bad(1)
    try:
        rng = np.random.default_rng(n + m + e)
        G = rng.standard_normal((n, n)); K = G @ G.T + n * np.eye(n)
        p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        info = ctypes.c_int(0)
        if e:
            E = rng.standard_normal((e, n))
            A = np.ascontiguousarray(np.vstack([K, E]))
            rc = emul.emul_potrf_aug(p(A), ctypes.c_int64(n), ctypes.c_int64(e), ctypes.byref(info))
            assert rc == 0 and info.value == 0
            L = sl.cholesky(K, lower=True)
            assert np.abs(np.tril(A[:n]) - L).max() <= 1e-12 * np.abs(L).max()
            assert np.abs(A[n:] - sl.solve_triangular(L, E.T, lower=True).T).max() <= 1e-11
        else:
            A = K.copy(); B = rng.standard_normal((m, n)); B2 = B.copy(); y = rng.standard_normal((1, n))
            B0, y0 = B.copy(), y.copy()
            rc = emul.emul_all(p(A), ctypes.c_int64(n), p(B), p(B2), ctypes.c_int64(m), p(y), ctypes.c_int64(1), ctypes.byref(info))
            assert rc == 0 and info.value == 0
            L = sl.cholesky(K, lower=True)
            assert np.abs(np.tril(A) - L).max() <= 1e-12 * np.abs(L).max()
            assert np.abs(B - sl.solve_triangular(L, B0.T, lower=True).T).max() <= 1e-12
        assert emul.emul_side_bad(0) == 0
    finally:
        emul.emul_set_lookahead(0)


def test_side_stream_race_detector_detects(emul):
    """The detector behind the test above, firing: with the lookahead_min threshold of the emulation every group forks a side
    section; an Ops that "forgets" the joins (emul_set_forget_join) must be flagged."""
    emul.emul_set_rl_max(ctypes.c_int64(1024)); emul.emul_set_lookahead(1); emul.emul_set_rl_group(ctypes.c_int64(2))
    emul.emul_set_forget_join(1)
    emul.emul_side_bad(1)
    try:
        n = 1024
        rng = np.random.default_rng(n)
        G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n)
        B = rng.standard_normal((128, n)); B2 = B.copy(); y = rng.standard_normal((1, n))
        info = ctypes.c_int(0)
        p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        emul.emul_all(p(A), ctypes.c_int64(n), p(B), p(B2), ctypes.c_int64(128), p(y), ctypes.c_int64(1), ctypes.byref(info))
        assert emul.emul_side_bad(0) > 0
    finally:
        emul.emul_set_forget_join(0); emul.emul_set_lookahead(0)


@pytest.mark.parametrize("n,wb,m", [(256, 256, 128), (512, 256, 64), (1024, 512, 128), (1280, 512, 192), (2048, 1024, 128), (1536, 256, 128)])
def test_wide_inverse_blocks_as_index_logic(emul, n, wb, m):
    """blocked.hpp::wide_inverse / trsm_wide_rec (round 6; predict_f on few test points, models/gpr.py:122): the inverses of the
    factor's wb-column diagonal blocks, built level by level from the 128-column ones by batched products whose operands step
    along diagonals (row / column strides, column offsets modulo the block width), then X L^T = B as one product per wb-column
    node.  Against LAPACK: every wide block equals inv(L_cc) (its strict upper part untouched: still NaN beyond the diagonal
    128-blocks), and X equals the triangular solve for the whole blocks' columns."""
    emul.emul_set_rl_max(ctypes.c_int64(256)); emul.emul_set_lookahead(0); emul.emul_set_rl_group(ctypes.c_int64(2))
    rng = np.random.default_rng(n + wb + m)
    G = rng.standard_normal((n, n)); A = G @ G.T + n * np.eye(n); A0 = A.copy()
    nf = (n // wb) * wb
    W = np.full((nf, wb), np.nan); Wt = np.full((nf, wb), np.nan); Tm = np.full((max(nf // 2, 1), wb // 2), np.nan)
    B = rng.standard_normal((m, n)); B0 = B.copy(); X = np.full((m, n), np.nan)
    info = ctypes.c_int(0)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    rc = emul.emul_wide(p(A), ctypes.c_int64(n), ctypes.c_int64(wb), p(W), p(Wt), p(Tm), p(B), p(X), ctypes.c_int64(m), ctypes.byref(info))
    assert rc == 0 and info.value == 0
    L = sl.cholesky(A0, lower=True)
    low128 = np.tril(np.ones((128, 128), dtype=bool))
    for c in range(nf // wb):
        blk = L[c * wb:(c + 1) * wb, c * wb:(c + 1) * wb]
        ref = sl.solve_triangular(blk, np.eye(wb), lower=True)
        got = W[c * wb:(c + 1) * wb]
        low = np.tril(np.ones((wb, wb), dtype=bool))
        assert np.abs(got[low] - ref[low]).max() <= 1e-12 * np.abs(ref).max()
        # above the diagonal: zeros inside the diagonal 128-blocks (the 128-column inverses are stored whole), untouched beyond
        for i in range(wb // 128):
            assert np.all(got[i * 128:(i + 1) * 128, i * 128:(i + 1) * 128][~low128] == 0.0)
            assert np.isnan(got[i * 128:(i + 1) * 128, (i + 1) * 128:]).all()
    ref = sl.solve_triangular(L[:nf, :nf], B0[:, :nf].T, lower=True).T
    assert np.abs(X[:, :nf] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
