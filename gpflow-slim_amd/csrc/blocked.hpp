// Recursive blocked factorisation / triangular solves, written once over an "Ops" policy.
//
// The product instantiates it with HipOps (gps_ops.hpp: every op is a HIP kernel launch on the
// handle's stream).  tests/cpu_blocked/ instantiates the same template with naive host loops so
// the index arithmetic of the recursion can be checked without a GPU -- that emulation is test
// infrastructure only and is never linked into libgpflowslim_hip.so.
//
// All matrices are row-major with dimensions that are multiples of 128 (callers pad with an
// identity block, whose Cholesky factor is the identity).  Only lower triangles are defined.
//
//   potrf_rec : A = L L^T                        tf.cholesky               models/gpr.py:70
//   trsm_rec  : X L^T = B   (B [m,n] in place)    tf.matrix_triangular_solve(L, Kx) computed on
//               the transposed right-hand side    models/gpr.py:122, conditionals.py:87
//   trsm_rn_rec: X L = B with U = L^T stored      conditionals.py:100 (unwhitened back-solve)
//   trsv_rec  : L a = y     (y [r][ld] in place)  densities.py:82, models/gpr.py:123
//   trsv_t_rec / inv_t_rec / lauum_rec : L^T a = y, Y = L^-T, K^-1 = Y Y^T -- the pieces of the analytic
//               gradient of the log-marginal likelihood (what TF autodiff through tf.cholesky supplies to
//               examples/gpr.py:53-54)
//
// Every flop of the recursion lands in Ops::gemm (C -= A B^T / C = A B^T / C += A B^T / C = -A B^T) with K = half the
// current block, i.e. long-K MFMA GEMMs; the 128x128 leaves use the explicit block inverses
// produced by Ops::potrf_base -- as the solve itself, or (Ops decides; trsm_leaf.hip) as the preconditioner of one
// refinement step against the diagonal block D of the factor, which every leaf call therefore also receives.
#pragma once
#include <cstdint>

#ifndef GPS_TILE
#define GPS_TILE 128
#endif

// A batch of equal NT products whose operands step along diagonals (Ops::gemm_ex): problem p reads / writes its operands at
// base + (p * rs) * ld + (p * cs) % cm  (cm == 0: no wrap) -- blocks along the diagonal of the factor, or inside a buffer that
// stacks wide diagonal blocks (column offset modulo the width).
struct GemmBatch { int64_t batch = 1; int64_t a_rs = 0, a_cs = 0, a_cm = 0, b_rs = 0, b_cs = 0, b_cm = 0, c_rs = 0, c_cs = 0, c_cm = 0; };

template <class Ops>
struct Blocked {
  Ops& ops;
  explicit Blocked(Ops& o) : ops(o) {}
  typedef int64_t i64;

  static i64 split(i64 n) { return ((n / GPS_TILE) / 2) * GPS_TILE; }   // n >= 256 -> 128 <= n1 < n
  // the triangular solves: first parts that are multiples of 512 columns, so that their recursion ends in 512-column nodes
  // (one launch each: Ops::trsm_leaf512) whatever the number of tiles
  static i64 split_solve(i64 n) {
    const i64 P = 4 * GPS_TILE;
    if (n <= P) return split(n);
    const i64 h = ((n / P) / 2) * P;
    return h > P ? h : P;
  }

  // A piece of the PARENT's panel solve handed down to a child node: solve  X L^T = B  for the first dn columns of the
  // child's matrix (B [dm, dn]: the parent's rows below, same columns).  The child issues it on the deferred stream as
  // soon as those columns are final -- its own first half is done --, where it runs beside the child's second-half
  // sweep, during which the GPU is otherwise mostly idle.  The parent then skips that part of its trsm_rec.
  struct Deferred { double* B; i64 ldb, dm, dn; bool issued; };

  // blk0: index of the first 128-block of this sub-matrix in the block-inverse array;
  // row0: global row of A's first row (for info reporting)
  // e: AUGMENTED rows -- e more rows stored directly below A (same leading dimension, a multiple of 128) that are not
  //    factored but solved along:  on return they hold  E L^-T.  With E = (Y - m)^T this is alpha^T = (L^-1 (Y - m))^T
  //    (densities.py:82): the forward substitution of the likelihood costs no pass of its own -- the rows ride through
  //    the panel solves and trailing updates of the right spine of the recursion (where "the rows below" are contiguous
  //    with them), as the e extra rows of a few GEMMs.
  // (Measured and removed, rounds 3 - 5; code in the history, tables in docs/LAB_NOTES.md: the substitution following the
  //  factorisation block by block on a stream of its own; cross-level look-ahead -- the rest of a trailing update / the first
  //  rows of a panel solve beside the sweeps --; the trailing update applied piece by piece behind a sweep.)
  int potrf_rec(double* A, i64 lda, i64 n, i64 blk0, i64 row0, Deferred* dj = nullptr, i64 e = 0) {
    if (n <= 0) return 0;
    if (n == GPS_TILE) {
      int rc = ops.potrf_base(A, lda, blk0, row0);
      if (rc) return rc;
      return e > 0 ? ops.trsm_base(blk0, 0, A + GPS_TILE * lda, lda, e, A, lda) : 0;
    }
    if (n <= ops.rl_max())
      return ops.rl_group() > 1 ? potrf_rl_groups(A, lda, n, ops.rl_group(), blk0, row0, nullptr, 0, 0, e) : potrf_rl(A, lda, n, GPS_TILE, blk0, row0, e);
    const i64 n1 = split(n), n2 = n - n1;
    const i64 m2 = n2 + e;                          // rows below A11: A21 and, under it, the augmented rows
    double* A21 = A + n1 * lda;
    double* A22 = A21 + n1;
    int rc;
    if (n1 > GPS_TILE && n1 <= ops.rl_max() && ops.rl_group() > 1 && ops.follower()) {
      // A11 is factored by the sweep: the solve of A21 against it follows the sweep on the follower stream
      rc = potrf_rl_groups(A, lda, n1, ops.rl_group(), blk0, row0, A21, lda, m2, 0);
      if (rc) return rc;
      if (dj && dj->dn == n1) {
        // the first n1 columns are final: the parent's rows below can be solved against them from now on
        rc = ops.deferred_open();
        if (rc) return rc;
        rc = trsm_rec(A, lda, n1, blk0, dj->B, dj->ldb, dj->dm);
        const int rc2 = ops.deferred_close();
        if (rc || rc2) return rc ? rc : rc2;
        dj->issued = true;
      }
    } else {
      // a child whose own first half is a sweep can take the first part of this node's panel solve with it
      const i64 n1a = split(n1);
      Deferred job{A21, lda, m2, n1a, false};
      const bool hand_down = n1 > ops.rl_max() && n1a > GPS_TILE && n1a <= ops.rl_max() && ops.rl_group() > 1 && ops.follower() && ops.deferred();
      rc = potrf_rec(A, lda, n1, blk0, row0, hand_down ? &job : nullptr, 0);
      if (rc) return rc;
      if (hand_down && job.issued) {
        // trsm_rec(A, n1) = trsm_rec(first n1a columns) [done on the deferred stream] ; update ; trsm_rec(the others)
        rc = ops.deferred_join();
        if (rc) return rc;
        rc = ops.gemm(0, 0, m2, n1 - n1a, n1a, A21, lda, A + n1a * lda, lda, A21 + n1a, lda);
        if (rc) return rc;
        rc = trsm_rec(A + n1a * lda + n1a, lda, n1 - n1a, blk0 + n1a / GPS_TILE, A21 + n1a, lda, m2);
      } else {
        rc = trsm_rec(A, lda, n1, blk0, A21, lda, m2);
      }
      if (rc) return rc;
    }
    rc = ops.gemm(/*op sub*/ 0, /*lower (trapezoid when e > 0)*/ 1, m2, n2, n1, A21, lda, A21, lda, A22, lda);
    if (rc) return rc;
    return potrf_rec(A22, lda, n2, blk0 + n1 / GPS_TILE, row0 + n1, nullptr, e);
  }

  // Right-looking sweep over nbp-column panels.
  //  nbp = 128 for a small diagonal block (n <= Ops::rl_max()): at this size every launch of the recursion sits at
  //  the launch-latency floor, so what counts is the number of launches between two consecutive potrf_base calls:
  //  here always two (one solve of all rows below, one lower-triangular K = 128 update of the whole remainder)
  //  instead of up to 2 log2(n/128) + 1.  The K = 128 update re-reads and re-writes the remainder once per panel,
  //  so the sweep loses against the recursion once that traffic costs more than the launches saved.
  //  nbp > 128: the same sweep with panels factored by potrf_rec and solved by trsm_rec (measured on MI355X for
  //  blocks of 4096 .. 32768 columns in panels of 256 .. 4096: always behind the recursion; not used by potrf_rec).
  int potrf_rl(double* A, i64 lda, i64 n, i64 nbp, i64 blk0, i64 row0, i64 e = 0) {
    for (i64 c = 0; c < n; c += nbp) {
      const i64 w = (n - c < nbp) ? n - c : nbp;
      double* Ajj = A + c * lda + c;
      const i64 blk = blk0 + c / GPS_TILE;
      int rc = (w == GPS_TILE) ? ops.potrf_base(Ajj, lda, blk, row0 + c) : potrf_rec(Ajj, lda, w, blk, row0 + c);
      if (rc) return rc;
      const i64 sq = n - c - w, m = sq + e;           // square remainder; rows below incl. the augmented ones
      if (m == 0) break;
      double* B = Ajj + w * lda;                      // rows below the diagonal block
      rc = trsm_rec(Ajj, lda, w, blk, B, lda, m);
      if (rc) return rc;
      if (sq == 0) break;
      rc = ops.gemm(0, 1, m, sq, w, B, lda, B, lda, B + w, lda);
      if (rc) return rc;
    }
    return 0;
  }

  // The 128-column sweep with the remainder updated once per GROUP of g panels: inside a group, after panel i only
  // the next block column is updated, with all i+1 panels of the group so far (m x 128, K = 128 (i+1): they are
  // adjacent columns of the same rows); after the last panel one K = 128 g update of the remainder.  Same number of
  // launches per 128 columns as potrf_rl, but the remainder -- whose read-modify-write is what bounds the K = 128
  // update above ~2000 rows -- crosses HBM once per group.
  // Look-ahead (Ops::lookahead): the chain updates only the next block column, the rest of the remainder runs on the side
  // stream beside the next group's first potrf_base / panel solve; its join is awaited before the next group's first
  // block-column update, the first launch that touches what the side stream writes.
  // Follower (FB != nullptr, look-ahead on): FB [fm, n] is the block below this diagonal block in the parent's
  // recursion (its A21), which the parent would solve against L afterwards (trsm_rec).  Column block [c0, c0 + 128 g)
  // of that solve needs nothing but the columns of L up to c0 + 128 g, which are final as soon as the group's last
  // panel is solved -- so the solve follows the sweep on the follower stream, right-looking: a solved piece is applied to all
  // the columns after it at once (K = its width: the updates SHRINK towards the end of the sweep, and what is left for the
  // chain is one 512-column solve), where it fills the GPU the latency-bound chain leaves idle.  What the follower has not
  // reached when the sweep ends is finished on the chain.
  // e: augmented rows directly below A (see potrf_rec): every "rows below" of the sweep simply has e rows more.
  // (Measured and removed, round 5: one launch per 128 columns -- sweep_step_kernel --, joins carried by GEMMs or by the step
  //  before, the two-stage join, the left-looking follower: docs/LAB_NOTES.md.)
  int potrf_rl_groups(double* A, i64 lda, i64 n, i64 g, i64 blk0, i64 row0, double* FB = nullptr, i64 ldfb = 0, i64 fm = 0,
                      i64 e = 0) {
    const i64 T = GPS_TILE;
    const bool la = (g >= 2) && ops.lookahead();
    const bool fol = la && FB != nullptr && fm > 0;
    unsigned long long pending = 0;              // join value of a remainder update still running on the side stream
    bool forked = false, fforked = false;        // a hand-over to the side / follower stream has already happened in this sweep
    i64 fdone = 0;                               // columns of the follower solve issued so far
    auto follower_piece = [&](i64 c_lo, i64 c_hi) -> int {          // columns [c_lo, c_hi) of  X L^T = FB (all earlier columns applied)
      int rc = trsm_rec(A + c_lo * lda + c_lo, lda, c_hi - c_lo, blk0 + c_lo / T, FB + c_lo, ldfb, fm);
      if (!rc && c_hi < n)                                          // FB[:, c_hi:n] -= X[:, c_lo:c_hi] L[c_hi:n, c_lo:c_hi]^T
        rc = ops.gemm(0, 0, fm, n - c_hi, c_hi - c_lo, FB + c_lo, ldfb, A + c_hi * lda + c_lo, lda, FB + c_hi, ldfb);
      return rc;
    };
    auto finish = [&]() -> int {
      int rc = pending ? ops.chain_join(pending) : 0;
      pending = 0;
      if (rc || FB == nullptr) return rc;
      if (fol && fdone > 0) { rc = ops.follower_join(); if (rc) return rc; }
      return fdone < n ? follower_piece(fdone, n) : 0;               // the rest (all of it without look-ahead) on the chain
    };
    for (i64 c0 = 0; c0 < n; c0 += g * T) {
      for (i64 i = 0; i < g; ++i) {
        const i64 c = c0 + i * T;
        if (c >= n) return finish();
        double* Acc = A + c * lda + c;
        int rc = ops.potrf_base(Acc, lda, blk0 + c / T, row0 + c);
        if (rc) return rc;
        const i64 sq = n - c - T, m = sq + e;          // columns right of this block ; rows below it
        if (m == 0) return finish();
        rc = ops.trsm_base(blk0 + c / T, 0, Acc + T * lda, lda, m, Acc, lda);
        if (rc) return rc;
        if (sq == 0) return finish();
        double* P = A + (c + T) * lda + c0;            // the group's panels so far, rows below this block: [m, (i+1) 128]
        double* Cn = A + (c + T) * lda + (c + T);      // the next block column / the remainder
        if (i + 1 < g) {
          // (the next block column was last written by the remainder update of the previous group)
          if (pending) { rc = ops.chain_join(pending); pending = 0; if (rc) return rc; }
          rc = ops.gemm(0, 0, m, T, (i + 1) * T, P, lda, P, lda, Cn, lda);      // next block column
          if (rc) return rc;
          continue;
        }
        // ---- last panel of the group
        const bool split = la && sq - T >= ops.lookahead_min_rows();
        if (!split && !fol) {
          rc = ops.gemm(0, 1, m, sq, g * T, P, lda, P, lda, Cn, lda);           // remainder
          if (rc) return rc;
          continue;
        }
        // look-ahead.  The chain's next GEMM publishes the fork ticket when it starts (= the panel solve before it has
        // completed).
        const unsigned long long t = ops.la_fork();      // (also when only the follower needs it)
        if (split) rc = ops.gemm(0, 0, m, T, g * T, P, lda, P, lda, Cn, lda);
        else rc = ops.gemm(0, 1, m, sq, g * T, P, lda, P, lda, Cn, lda);
        if (rc) return rc;
        if (split) {
          rc = ops.side_open(t, !forked);
          forked = true;
          if (rc) return rc;
          rc = ops.gemm(0, 1, m - T, sq - T, g * T, P + T * lda, lda, P + T * lda, lda, Cn + T * lda + T, lda);
          if (!rc) rc = ops.side_publish_join(t);
          pending = t;
          const int rc2 = ops.side_close();
          if (rc || rc2) return rc ? rc : rc2;
        }
        // the follower's pieces are long: on a stream of their own, or the next group's remainder update -- which the chain
        // waits for -- would queue behind them
        if (fol && c0 + g * T - fdone >= ops.follower_cols()) {
          rc = ops.follower_open(t, !fforked);
          fforked = true;
          if (rc) return rc;
          rc = follower_piece(fdone, c0 + g * T); fdone = c0 + g * T;
          if (!rc) rc = ops.follower_publish();
          const int rc2 = ops.follower_close();
          if (rc || rc2) return rc ? rc : rc2;
        }
      }
    }
    return finish();
  }

  // Many more rows than columns (conditionals.py:87: the [N, M] cross matrix against the M x M factor): 512-column panels left
  // to right, each updated ONCE with everything solved before it (K = 512 j: long K loops, its columns read and written once)
  // instead of the recursive halving, whose K = 512 and K = 1024 updates run 7 % / 3 % below the K = 2048 ones (config 5:
  // 8.1 / 30.7 ms where the long-K rate gives 7.5 / 29.9).  Only when every panel can take the one-launch solve.
  bool tall_panels(i64 m, i64 n, int transposed, i64 blk0) const {
    const i64 P = 4 * GPS_TILE;
    if (!ops.trsm_left_looking(m, n) || n <= P || n % P) return false;
    for (i64 c = 0; c < n; c += P)
      if (!ops.leaf512(m, transposed, blk0 + c / GPS_TILE)) return false;
    return true;
  }

  // solve X L^T = B in place; L [n,n] lower at (L, ldl); B [m,n] at (B, ldb)
  int trsm_rec(const double* L, i64 ldl, i64 n, i64 blk0, double* B, i64 ldb, i64 m) {
    if (n <= 0 || m <= 0) return 0;
    if (n == GPS_TILE) return ops.trsm_base(blk0, /*transposed inverse*/ 0, B, ldb, m, L, ldl);
    if (n == 4 * GPS_TILE && ops.leaf512(m, 0, blk0)) return ops.trsm_leaf512(blk0, 0, B, ldb, m, L, ldl);
    if (tall_panels(m, n, 0, blk0)) {
      const i64 P = 4 * GPS_TILE;
      for (i64 c = 0; c < n; c += P) {
        int rc = c ? ops.gemm(0, 0, m, P, c, B, ldb, L + c * ldl, ldl, B + c, ldb) : 0;          // B_c -= X[:, 0:c] L[c:c+P, 0:c]^T
        if (!rc) rc = ops.trsm_leaf512(blk0 + c / GPS_TILE, 0, B + c, ldb, m, L + c * ldl + c, ldl);
        if (rc) return rc;
      }
      return 0;
    }
    const i64 n1 = split_solve(n), n2 = n - n1;
    int rc = trsm_rec(L, ldl, n1, blk0, B, ldb, m);
    if (rc) return rc;
    rc = ops.gemm(0, 0, m, n2, n1, B, ldb, L + n1 * ldl, ldl, B + n1, ldb);   // B2 -= X1 L21^T
    if (rc) return rc;
    return trsm_rec(L + n1 * ldl + n1, ldl, n2, blk0 + n1 / GPS_TILE, B + n1, ldb, m);
  }

  // solve X L = B in place, given U = L^T (upper, row-major) and the transposed block inverses
  int trsm_rn_rec(const double* U, i64 ldu, i64 n, i64 blk0, double* B, i64 ldb, i64 m) {
    if (n <= 0 || m <= 0) return 0;
    if (n == GPS_TILE) return ops.trsm_base(blk0, /*transposed inverse*/ 1, B, ldb, m, U, ldu);
    if (n == 4 * GPS_TILE && ops.leaf512(m, 1, blk0)) return ops.trsm_leaf512(blk0, 1, B, ldb, m, U, ldu);
    if (tall_panels(m, n, 1, blk0)) {                                                              // (right to left)
      const i64 P = 4 * GPS_TILE;
      for (i64 c = n - P; c >= 0; c -= P) {
        // B_c -= X[:, c+P:n] L[c+P:n, c:c+P] = X[:, c+P:n] (U[c:c+P, c+P:n])^T
        int rc = (c + P < n) ? ops.gemm(0, 0, m, P, n - c - P, B + c + P, ldb, U + c * ldu + c + P, ldu, B + c, ldb) : 0;
        if (!rc) rc = ops.trsm_leaf512(blk0 + c / GPS_TILE, 1, B + c, ldb, m, U + c * ldu + c, ldu);
        if (rc) return rc;
      }
      return 0;
    }
    const i64 n1 = split_solve(n), n2 = n - n1;
    int rc = trsm_rn_rec(U + n1 * ldu + n1, ldu, n2, blk0 + n1 / GPS_TILE, B + n1, ldb, m);
    if (rc) return rc;
    // B1 -= X2 L21 = X2 (U12)^T ; U12 = U[0:n1, n1:n]
    rc = ops.gemm(0, 0, m, n1, n2, B + n1, ldb, U + n1, ldu, B, ldb);
    if (rc) return rc;
    return trsm_rn_rec(U, ldu, n1, blk0, B, ldb, m);
  }

  // solve L^T a = y in place (backward substitution), right-hand sides stored as rows y[q*ldy + i]
  int trsv_t_rec(const double* L, i64 ldl, i64 n, i64 blk0, double* y, i64 ldy, i64 r) {
    if (n <= 0 || r <= 0) return 0;
    if (n == GPS_TILE) return ops.trsv_t_base(blk0, y, ldy, r, L, ldl);
    const i64 n1 = split(n), n2 = n - n1;
    int rc = trsv_t_rec(L + n1 * ldl + n1, ldl, n2, blk0 + n1 / GPS_TILE, y + n1, ldy, r);
    if (rc) return rc;
    rc = ops.gemv_t_sub(L + n1 * ldl, ldl, n2, n1, y + n1, y, ldy, r);        // y1 -= L21^T a2
    if (rc) return rc;
    return trsv_t_rec(L, ldl, n1, blk0, y, ldy, r);
  }

  // Y = L^-T (upper triangular, full storage with explicit zeros below the diagonal)
  //   L^-T = [[Y11, -Y11 L21^T L22^-T], [0, Y22]]
  int inv_t_rec(const double* L, i64 ldl, i64 n, i64 blk0, double* Y, i64 ldy) {
    if (n <= 0) return 0;
    if (n == GPS_TILE) return ops.copy_linvT(blk0, Y, ldy);
    const i64 n1 = split(n), n2 = n - n1;
    int rc = inv_t_rec(L, ldl, n1, blk0, Y, ldy);
    if (rc) return rc;
    rc = inv_t_rec(L + n1 * ldl + n1, ldl, n2, blk0 + n1 / GPS_TILE, Y + n1 * ldy + n1, ldy);
    if (rc) return rc;
    double* Y12 = Y + n1;
    if (ops.fill_zeros()) {                   // only for a caller that reads Y as a full matrix: lauum_rec never
      rc = ops.zero_block(Y + n1 * ldy, ldy, n2, n1);      // touches the blocks below the diagonal
      if (rc) return rc;
    }
    rc = ops.gemm(/*C = -A B^T*/ 3, /*A upper triangular*/ 2, n1, n2, n1, Y, ldy, L + n1 * ldl, ldl, Y12, ldy);   // Y12 = -Y11 L21^T
    if (rc) return rc;
    return trsm_rec(L + n1 * ldl + n1, ldl, n2, blk0 + n1 / GPS_TILE, Y12, ldy, n1);   // ... L22^-T
  }

  // Kv (lower triangle) = Y Y^T for upper-triangular Y  (= (L L^T)^-1 when Y = L^-T)
  int lauum_rec(const double* Y, i64 ldy, i64 n, double* Kv, i64 ldk) {
    if (n <= 0) return 0;
    if (n == GPS_TILE) return ops.gemm(1, 1, n, n, n, Y, ldy, Y, ldy, Kv, ldk);
    const i64 n1 = split(n), n2 = n - n1;
    int rc = lauum_rec(Y, ldy, n1, Kv, ldk);
    if (rc) return rc;
    const double* Y12 = Y + n1;
    const double* Y22 = Y + n1 * ldy + n1;
    rc = ops.gemm(2, 1, n1, n1, n2, Y12, ldy, Y12, ldy, Kv, ldk);                // K11 += Y12 Y12^T
    if (rc) return rc;
    rc = ops.gemm(1, /*A upper triangular*/ 2, n2, n1, n2, Y22, ldy, Y12, ldy, Kv + n1 * ldk, ldk);   // K21 = Y22 Y12^T
    if (rc) return rc;
    return lauum_rec(Y22, ldy, n2, Kv + n1 * ldk + n1, ldk);
  }

  // solve L a = y in place for r right-hand sides stored as rows y[q*ldy + i]
  int trsv_rec(const double* L, i64 ldl, i64 n, i64 blk0, double* y, i64 ldy, i64 r) {
    if (n <= 0 || r <= 0) return 0;
    if (n == GPS_TILE) return ops.trsv_base(blk0, y, ldy, r, L, ldl);
    const i64 n1 = split(n), n2 = n - n1;
    int rc = trsv_rec(L, ldl, n1, blk0, y, ldy, r);
    if (rc) return rc;
    rc = ops.gemv_sub(L + n1 * ldl, ldl, n2, n1, y, y + n1, ldy, r);          // y2 -= L21 a1
    if (rc) return rc;
    return trsv_rec(L + n1 * ldl + n1, ldl, n2, blk0 + n1 / GPS_TILE, y + n1, ldy, r);
  }

  // ---- wide inverse blocks of a factor (round 6; predict_f on few test points, models/gpr.py:122) --------------------------------
  // W_c = inv(L_cc) for every wb-column diagonal block c of the first nf columns of L (nf a multiple of wb, wb = 128 * 2^k), built
  // level by level from the 128-column inverses (linv / linvT: [nf / 128][128 x 128], potrf_base's) for ALL diagonal blocks at once:
  //   [[L11, 0], [L21, L22]]^-1 = [[W11, 0], [-W22 L21 W11, W22]]   as three batched NT products per level b -> 2 b
  //   T^T = W11^T L21^T  (A upper triangular) ;  W21 = -W22 (T^T)^T  (A lower triangular) ;  (W^T)12 = -T^T W22^T  (B lower triangular)
  // W, Wt: [nf, wb] -- the wide blocks (lower triangular) / their transposes, stacked: block c at rows c wb, level-b block k at rows
  // k b, columns (k b) % wb.  T: scratch [nf / 2, wb / 2].  The transposes are kept because the NT form needs W11^T as a left operand;
  // the last level does not produce them.
  int wide_inverse(const double* L, i64 ldl, i64 nf, i64 wb, const double* linv, const double* linvT, double* W, double* Wt, double* Tm) {
    if (nf <= 0) return 0;
    const i64 T = GPS_TILE, ldt = wb / 2;
    int rc = ops.blocks_to_diag(linv, W, nf / T, wb);
    if (!rc) rc = ops.blocks_to_diag(linvT, Wt, nf / T, wb);
    for (i64 b = T; b < wb && !rc; b *= 2) {
      GemmBatch bt;
      bt.batch = nf / (2 * b);
      // T^T [b, b] of pair p (rows p b of the scratch) = W11^T L21^T
      bt.a_rs = 2 * b; bt.a_cs = 2 * b; bt.a_cm = wb; bt.b_rs = 2 * b; bt.b_cs = 2 * b; bt.b_cm = 0; bt.c_rs = b; bt.c_cs = 0; bt.c_cm = 0;
      rc = ops.gemm_ex(/*C = A B^T*/ 1, /*A upper*/ 1, b, b, b, Wt, wb, L + b * ldl, ldl, Tm, ldt, &bt);
      if (rc) break;
      // W21 = -W22 (T^T)^T
      bt.a_rs = 2 * b; bt.a_cs = 2 * b; bt.a_cm = wb; bt.b_rs = b; bt.b_cs = 0; bt.b_cm = 0; bt.c_rs = 2 * b; bt.c_cs = 2 * b; bt.c_cm = wb;
      rc = ops.gemm_ex(/*C = -A B^T*/ 3, /*A lower*/ 2, b, b, b, W + b * wb + b, wb, Tm, ldt, W + b * wb, wb, &bt);
      if (rc || 2 * b >= wb) break;                       // (the transposes only feed the next level)
      // (W^T)12 = -T^T W22^T
      bt.a_rs = b; bt.a_cs = 0; bt.a_cm = 0; bt.b_rs = 2 * b; bt.b_cs = 2 * b; bt.b_cm = wb; bt.c_rs = 2 * b; bt.c_cs = 2 * b; bt.c_cm = wb;
      rc = ops.gemm_ex(3, /*B lower*/ 3, b, b, b, Tm, ldt, W + b * wb + b, wb, Wt + b, wb, &bt);
    }
    return rc;
  }

  // X L^T = B for columns [c0, c0 + n) (n a multiple of wb) against the wide inverse blocks W: X [m, .] <- solution, B destroyed.
  // n == wb: X_c = B_c W_c^T, ONE product (B lower triangular: half the K range per tile on average); else halves with the
  // update B2 -= X1 L21^T between them (K >= wb: no launch below the wb-column level).
  int trsm_wide_rec(const double* L, i64 ldl, i64 c0, i64 n, i64 wb, const double* W, double* B, double* X, i64 ld, i64 m) {
    if (n <= 0 || m <= 0) return 0;
    if (n == wb) return ops.gemm_ex(1, /*B lower*/ 3, m, wb, wb, B + c0, ld, W + c0 * wb, wb, X + c0, ld, nullptr);
    const i64 n1 = ((n / wb) / 2) * wb, n2 = n - n1;
    int rc = trsm_wide_rec(L, ldl, c0, n1, wb, W, B, X, ld, m);
    if (rc) return rc;
    rc = ops.gemm(0, 0, m, n2, n1, X + c0, ld, L + (c0 + n1) * ldl + c0, ldl, B + c0 + n1, ld);       // B2 -= X1 L21^T
    if (rc) return rc;
    return trsm_wide_rec(L, ldl, c0 + n1, n2, wb, W, B, X, ld, m);
  }
};
