// Recursive blocked factorisation / triangular solves, written once over an "Ops" policy.
//
// The product instantiates it with HipOps (gps_api.hip: every op is a HIP kernel launch on the
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

  // The forward substitution of the likelihood (alpha = L^-1 (Y - m), densities.py:82) FOLLOWS the factorisation: as
  // soon as a diagonal block is final its trsv_rec is issued on a stream of its own (Ops::y_*), as soon as the block
  // below it is solved the gemv that carries alpha on into the rows below -- ~500 launch-latency-bound kernels that
  // then run beside the GEMMs of the rest of the factorisation instead of after it.  y: right-hand sides of the
  // current column range, [r][ldy], in place.
  struct YFollow { double* y; i64 ldy, r; };

  // CROSS-LEVEL look-ahead (round 5).  The sweeps are latency chains that leave most of the GPU idle (a 4096-column sweep
  // holds 0.3 ms of GEMM work and takes 1.6 - 1.9 ms); the big GEMMs of the levels above have slack.  Two kinds of BULK work
  // are therefore taken out of the calling stream's order and issued on a stream of their own (Ops::bulk_open / bulk_close,
  // joined by Ops::bulk_join before the first launch that needs their result; one piece in flight at a time: Ops::bulk()):
  //  (a) the REST of a trailing update.  A22 -= A21 A21^T is split at A22's own split point: its leading n2a columns (all
  //      rows) on the calling stream, the lower-right block -- which the child potrf_rec(A22) does not touch before its own
  //      trailing update -- on the bulk stream, beside the child's first half (sweeps, panel solve).  The child is told by
  //      `pend` and joins before that update.
  //  (b) the first rows of a node's panel solve, beside the LAST sweep of the node's first half (the one with nothing of
  //      its own to run beside): rows [0, rows) of  X L^T = A21  against the first n1a columns -- final since the first
  //      half of A11 was factored -- are issued by the leaf that is about to start that sweep (`last` is handed down the
  //      right spine); the node then solves the other rows itself, joins and carries on as after a handed-down piece.
  struct Last { const double* L; i64 ldl, n, blk0; double* B; i64 ldb, rows; bool issued; };
  int issue_last(Last* last) {
    if (!last || last->issued || last->rows <= 0 || !ops.bulk()) return 0;
    int rc = ops.bulk_open();
    if (rc) return rc;
    rc = trsm_rec(last->L, last->ldl, last->n, last->blk0, last->B, last->ldb, last->rows);
    const int rc2 = ops.bulk_close();
    if (rc || rc2) return rc ? rc : rc2;
    last->issued = true;
    return 0;
  }

  // blk0: index of the first 128-block of this sub-matrix in the block-inverse array;
  // row0: global row of A's first row (for info reporting)
  // e: AUGMENTED rows -- e more rows stored directly below A (same leading dimension, a multiple of 128) that are not
  //    factored but solved along:  on return they hold  E L^-T.  With E = (Y - m)^T this is alpha^T = (L^-1 (Y - m))^T
  //    (densities.py:82): the forward substitution of the likelihood costs no pass of its own -- the rows ride through
  //    the panel solves and trailing updates of the right spine of the recursion (where "the rows below" are contiguous
  //    with them), as the e extra rows of a few GEMMs.
  // issue  y[0:n] <- L^-1 y[0:n]  of a block that the calling stream has just finished
  int y_block(const double* L, i64 ldl, i64 n, i64 blk0, const YFollow* yf) {
    if (!yf) return 0;
    int rc = ops.y_open();
    if (rc) return rc;
    rc = ops.y_prepare(blk0, n / GPS_TILE);          // (whatever the leaves of the substitution need of these blocks)
    if (!rc) rc = trsv_rec(L, ldl, n, blk0, yf->y, yf->ldy, yf->r);
    const int rc2 = ops.y_close();
    return rc ? rc : rc2;
  }

  // pend: a bulk piece (a) is in flight that writes this node's A22 -- join before the trailing update
  // last: a bulk piece (b) to issue right before the last sweep of this sub-matrix
  int potrf_rec(double* A, i64 lda, i64 n, i64 blk0, i64 row0, Deferred* dj = nullptr, i64 e = 0, const YFollow* yf = nullptr,
                bool pend = false, Last* last = nullptr) {
    if (n <= 0) return 0;
    if (n == GPS_TILE) {
      int rc = ops.potrf_base(A, lda, blk0, row0);
      if (rc) return rc;
      if (e > 0) { rc = ops.trsm_base(blk0, 0, A + GPS_TILE * lda, lda, e, A, lda); if (rc) return rc; }
      return y_block(A, lda, n, blk0, yf);
    }
    if (n <= ops.rl_max()) {
      int rc = issue_last(last);
      if (rc) return rc;
      rc = ops.rl_group() > 1 ? potrf_rl_groups(A, lda, n, ops.rl_group(), blk0, row0, nullptr, 0, 0, e) : potrf_rl(A, lda, n, GPS_TILE, blk0, row0, e);
      if (rc) return rc;
      return y_block(A, lda, n, blk0, yf);
    }
    const i64 n1 = split(n), n2 = n - n1;
    const i64 m2 = n2 + e;                          // rows below A11: A21 and, under it, the augmented rows
    double* A21 = A + n1 * lda;
    double* A22 = A21 + n1;
    int rc;
    bool trailing_done = false;
    if (n1 > GPS_TILE && n1 <= ops.rl_max() && ops.rl_group() > 1 && ops.follower()) {
      // A11 is factored by the sweep: the solve of A21 against it follows the sweep on the side stream
      // ... and so does the trailing update of A22: every solved column block of A21 is applied to A22 at once, behind
      // the sweep, instead of as one update after it (below: skipped)
      if (pend && ops.trail_follows()) { rc = ops.bulk_join(); if (rc) return rc; pend = false; }     // (the follower writes A22)
      rc = potrf_rl_groups(A, lda, n1, ops.rl_group(), blk0, row0, A21, lda, m2, 0, ops.trail_follows() ? A22 : nullptr, lda, n2);
      if (rc) return rc;
      trailing_done = ops.trail_follows();
      rc = y_block(A, lda, n1, blk0, yf);
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
      // (b): a first half of at least four sweeps has a last sweep with nothing to run beside -- the first rows of this
      // node's panel solve against its first n1a columns go there
      Last mine{A, lda, n1a, blk0, A21, lda, 0, false};
      if (!hand_down && n1a > ops.rl_max()) mine.rows = ops.bulk_rows(n1a, m2);
      rc = potrf_rec(A, lda, n1, blk0, row0, hand_down ? &job : nullptr, 0, yf, false, mine.rows > 0 ? &mine : nullptr);
      if (rc) return rc;
      if (mine.issued) {
        // rows [0, rows) of the first n1a columns are on the bulk stream: the other rows here, then as after a handed-down piece
        rc = trsm_rec(A, lda, n1a, blk0, A21 + mine.rows * lda, lda, m2 - mine.rows);
        if (rc) return rc;
        rc = ops.bulk_join();
        if (rc) return rc;
        rc = ops.gemm(0, 0, m2, n1 - n1a, n1a, A21, lda, A + n1a * lda, lda, A21 + n1a, lda);
        if (rc) return rc;
        rc = trsm_rec(A + n1a * lda + n1a, lda, n1 - n1a, blk0 + n1a / GPS_TILE, A21 + n1a, lda, m2);
      } else if (hand_down && job.issued) {
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
    YFollow y2{nullptr, 0, 0};
    if (yf) {
      // L21 is final (the calling stream's order): carry alpha_1 on into the rows below,  y2 -= L21 alpha_1
      y2 = YFollow{yf->y + n1, yf->ldy, yf->r};
      rc = ops.y_open();
      if (rc) return rc;
      rc = ops.gemv_sub(A21, lda, n2, n1, yf->y, y2.y, yf->ldy, yf->r);
      const int rc2 = ops.y_close();
      if (rc || rc2) return rc ? rc : rc2;
    }
    if (pend) { rc = ops.bulk_join(); if (rc) return rc; }         // the rest of the parent's update of this A22 has landed
    bool pend2 = false;
    if (!trailing_done) {
      const i64 n2a = split(n2);
      if (n2 > ops.rl_max() && n2a >= GPS_TILE && ops.bulk_rest()) {
        // (a): the columns the child factors first on this stream, the block it does not touch before its own trailing
        // update on the bulk stream
        rc = ops.gemm(0, 1, m2, n2a, n1, A21, lda, A21, lda, A22, lda);
        if (rc) return rc;
        rc = ops.bulk_open();
        if (rc) return rc;
        // (in K chunks: workgroups that live tens of microseconds instead of milliseconds, so that the kernels of the
        // chain find slots as they arrive -- Ops::bulk_chunk)
        double* R = A21 + n2a * lda;
        const i64 kc = ops.bulk_chunk(n1);
        for (i64 k0 = 0; k0 < n1 && !rc; k0 += kc)
          rc = ops.gemm(0, 1, m2 - n2a, n2 - n2a, (n1 - k0 < kc) ? n1 - k0 : kc, R + k0, lda, R + k0, lda, A22 + n2a * lda + n2a, lda);
        const int rc2 = ops.bulk_close();
        if (rc || rc2) return rc ? rc : rc2;
        pend2 = true;
      } else {
        rc = ops.gemm(/*op sub*/ 0, /*lower (trapezoid when e > 0)*/ 1, m2, n2, n1, A21, lda, A21, lda, A22, lda);
        if (rc) return rc;
      }
    }
    return potrf_rec(A22, lda, n2, blk0 + n1 / GPS_TILE, row0 + n1, nullptr, e, yf ? &y2 : nullptr, pend2, last);
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
  // Follower (FB != nullptr, look-ahead on): FB [fm, n] is the block below this diagonal block in the parent's
  // recursion (its A21), which the parent would solve against L afterwards (trsm_rec).  Column block [c0, c0 + 128 g)
  // of that solve needs nothing but the columns of L up to c0 + 128 g, which are final as soon as the group's last
  // panel is solved -- so the solve follows the sweep group by group on the side stream (left-looking: one update with
  // all previous columns, K = c0, then the 128 g-column solve), where it fills the GPU the latency-bound chain leaves
  // idle.  What the side stream has not reached when the sweep ends is finished on the chain.
  // e: augmented rows directly below A (see potrf_rec): every "rows below" of the sweep simply has e rows more.
  // TR (with FB): the parent's trailing matrix [fm, tn] (its A22, lower trapezoid); every solved column block X of FB is
  // applied to it right away, TR -= X X^T, on the stream the block was solved on -- the parent's syrk in pieces, in the
  // shadow of the sweep.
  int potrf_rl_groups(double* A, i64 lda, i64 n, i64 g, i64 blk0, i64 row0, double* FB = nullptr, i64 ldfb = 0, i64 fm = 0,
                      i64 e = 0, double* TR = nullptr, i64 ldt = 0, i64 tn = 0) {
    const i64 T = GPS_TILE;
    const bool la = (g >= 2) && ops.lookahead();
    const bool fol = la && FB != nullptr && fm > 0;
    // join values of a remainder update still running on the side stream (they grow: 2 t after its FIRST block column -- the
    // one the next group's first block-column update writes --, 2 t + 1 after the rest, which nothing touches before the next
    // group's second update: one more step of the chain for the side stream to finish in)
    unsigned long long pending = 0, pending2 = 0;
    bool forked = false, fforked = false;        // a hand-over to the side / follower stream has already happened in this sweep
    i64 fdone = 0;                               // columns of the follower solve issued so far
    // How a solved piece meets the columns after it (Ops::follower_tail): 0 -- left-looking, every piece first takes one update
    // with ALL columns before it (K = c_lo: the longest update comes last, and the chain waits for it when the sweep ends);
    // 1 -- the LAST follower_cols() columns [ts, n) are kept up to date by every earlier piece as soon as it is solved;
    // 2 -- right-looking throughout: a solved piece is applied to all columns after it at once (K = its width: the updates
    // SHRINK towards the end of the sweep, and what is left for the chain is one 512-column solve).
    const int fmode = fol ? ops.follower_tail() : 0;
    const i64 ts = fmode == 2 ? 0 : ((fmode == 1 && n >= 2 * ops.follower_cols()) ? n - ops.follower_cols() : -1);
    auto follower_piece = [&](i64 c_lo, i64 c_hi) -> int {          // columns [c_lo, c_hi) of  X L^T = FB
      int rc = 0;
      while (c_lo < c_hi && !rc) {
        const i64 hi = (ts > c_lo && ts < c_hi) ? ts : c_hi;        // (a piece across the start of the last block: in two)
        const i64 k0 = fmode == 2 ? c_lo : ((ts >= 0 && c_lo >= ts) ? ts : 0);     // columns [0, k0) have been applied to it already
        if (c_lo > k0) rc = ops.gemm(0, 0, fm, hi - c_lo, c_lo - k0, FB + k0, ldfb, A + c_lo * lda + k0, lda, FB + c_lo, ldfb);
        if (!rc) rc = trsm_rec(A + c_lo * lda + c_lo, lda, hi - c_lo, blk0 + c_lo / T, FB + c_lo, ldfb, fm);
        if (!rc && TR != nullptr) rc = ops.gemm(0, 1, fm, tn, hi - c_lo, FB + c_lo, ldfb, FB + c_lo, ldfb, TR, ldt);
        const i64 u0 = fmode == 2 ? hi : ((ts >= 0 && hi <= ts) ? ts : n);          // first column it is applied to right away
        if (!rc && u0 < n)                                          // FB[:, u0:n] -= X[:, c_lo:hi] L[u0:n, c_lo:hi]^T
          rc = ops.gemm(0, 0, fm, n - u0, hi - c_lo, FB + c_lo, ldfb, A + u0 * lda + c_lo, lda, FB + u0, ldfb);
        c_lo = hi;
      }
      return rc;
    };
    auto finish = [&]() -> int {
      if (pending2) pending = pending2;          // (values grow: the later one covers the earlier)
      int rc = pending ? ops.chain_join(pending) : 0;
      pending = 0; pending2 = 0;
      if (rc || FB == nullptr) return rc;
      if (fol && fdone > 0) { rc = ops.follower_join(); if (rc) return rc; }
      return fdone < n ? follower_piece(fdone, n) : 0;               // the rest (all of it without look-ahead) on the chain
    };
    // One launch per 128 columns (Ops::step, round 5): the panel solve, the update of the next block column and the NEXT
    // block's potrf_base as one launch (the next diagonal block only needs the top tile of the first two).  have_diag: the
    // block this iteration starts with was factored by the previous iteration's launch.  The launch publishes the fork ticket
    // when the panel is solved (what the start of the next-block-column GEMM used to signal) and awaits a pending join inside.
    const bool fused = la && ops.fused_step();
    bool have_diag = false;
    for (i64 c0 = 0; c0 < n; c0 += g * T) {
      for (i64 i = 0; i < g; ++i) {
        const i64 c = c0 + i * T;
        if (c >= n) return finish();
        double* Acc = A + c * lda + c;
        int rc = 0;
        if (!have_diag) rc = ops.potrf_base(Acc, lda, blk0 + c / T, row0 + c);
        have_diag = false;
        if (rc) return rc;
        const i64 sq = n - c - T, m = sq + e;          // columns right of this block ; rows below it
        if (m == 0) return finish();
        const bool fuse = fused && sq > 0;
        if (!fuse) {
          rc = ops.trsm_base(blk0 + c / T, 0, Acc + T * lda, lda, m, Acc, lda);
          if (rc) return rc;
          if (sq == 0) return finish();
        }
        double* P = A + (c + T) * lda + c0;            // the group's panels so far, rows below this block: [m, (i+1) 128]
        double* Cn = A + (c + T) * lda + (c + T);      // the next block column / the remainder
        // (second panel of a group: its update writes the first column of the REST of the previous group's remainder update)
        if (i == 1 && pending2) { rc = fuse ? ops.step_join(pending2) : ops.chain_join(pending2); pending2 = 0; if (rc) return rc; }
        // Joins CARRIED by a step (Ops::step_exit_join): the step launched for this panel awaits, before it ends, the join
        // value the NEXT panel's step needs -- first panel of a group: the rest of the previous remainder update (pending2);
        // last panel: the first block column of the remainder update forked right behind this step (its value is known now).
        const bool carry = fuse && ops.step_exit_join() && sq > T;       // (there is a next step)
        if (i + 1 < g) {
          // (the next block column was last written by the remainder update of the previous group)
          if (fuse) {
            // (Ops::step_join: awaited inside the launch, or -- its workgroups would hold their CUs while they wait for an
            // update that needs CUs -- by a launch of its own in front of it)
            if (pending) { rc = ops.step_join(pending); pending = 0; if (rc) return rc; }
            if (carry && i == 0 && pending2) { rc = ops.step_carry_join(pending2); pending2 = 0; if (rc) return rc; }
            rc = ops.step(blk0 + c / T, Acc + T * lda, lda, m, i * T, row0 + c + T);
            if (rc) return rc;
            have_diag = true;
            continue;
          }
          if (pending) { rc = ops.chain_join_next_gemm(pending); pending = 0; if (rc) return rc; }
          rc = ops.gemm(0, 0, m, T, (i + 1) * T, P, lda, P, lda, Cn, lda);      // next block column
          if (rc) return rc;
          continue;
        }
        // ---- last panel of the group
        const bool split = la && sq - T >= ops.lookahead_min_rows();
        if (!split && !fol) {
          if (fuse) {
            rc = ops.step(blk0 + c / T, Acc + T * lda, lda, m, i * T, row0 + c + T);
            if (rc) return rc;
            have_diag = true;
            if (sq > T) rc = ops.gemm(0, 1, m - T, sq - T, g * T, P + T * lda, lda, P + T * lda, lda, Cn + T * lda + T, lda);   // the rest of the remainder
          } else {
            rc = ops.gemm(0, 1, m, sq, g * T, P, lda, P, lda, Cn, lda);         // remainder
          }
          if (rc) return rc;
          continue;
        }
        // look-ahead.  The chain's next GEMM publishes the fork ticket when it starts (= the panel solve before it has
        // completed).  split: the chain updates only the next block column, the rest of the remainder runs on the side
        // stream beside the next group's first potrf_base / panel solve; its join is awaited before the next group's
        // first block-column update, the first launch that touches what the side stream writes.
        const unsigned long long t = ops.la_fork();      // (also when only the follower needs it)
        const bool two = split && ops.two_stage_join() && sq - T >= 2 * T;
        bool carried = false;
        if (fuse) {
          // (not with the sweep's FIRST hand-over: the side stream is then parked behind an event recorded after this launch
          // -- Ops::side_open --, i.e. it starts when this launch has ENDED: the launch would wait for itself)
          if (carry && split && forked) { rc = ops.step_carry_join(two ? 2 * t : 2 * t + 1); carried = true; if (rc) return rc; }
          rc = ops.step(blk0 + c / T, Acc + T * lda, lda, m, i * T, row0 + c + T);       // (carries the fork ticket)
          if (rc) return rc;
          have_diag = true;
          if (!split && sq > T) rc = ops.gemm(0, 1, m - T, sq - T, g * T, P + T * lda, lda, P + T * lda, lda, Cn + T * lda + T, lda);
        } else if (split) rc = ops.gemm(0, 0, m, T, g * T, P, lda, P, lda, Cn, lda);
        else rc = ops.gemm(0, 1, m, sq, g * T, P, lda, P, lda, Cn, lda);
        if (rc) return rc;
        if (split) {
          rc = ops.side_open(t, !forked);
          forked = true;
          if (rc) return rc;
          if (two) {
            // the remainder's first block column, which the next group's first update writes, then the rest -- which nothing
            // touches before the next group's second update
            rc = ops.gemm(0, 0, m - T, T, g * T, P + T * lda, lda, P + T * lda, lda, Cn + T * lda + T, lda);
            if (!rc) rc = ops.side_publish_join(2 * t);
            if (!rc) rc = ops.gemm(0, 1, m - 2 * T, sq - 2 * T, g * T, P + 2 * T * lda, lda, P + 2 * T * lda, lda, Cn + 2 * T * lda + 2 * T, lda);
            if (!rc) rc = ops.side_publish_join(2 * t + 1);
            pending = carried ? 0 : 2 * t; pending2 = 2 * t + 1;
          } else {
            rc = ops.gemm(0, 1, m - T, sq - T, g * T, P + T * lda, lda, P + T * lda, lda, Cn + T * lda + T, lda);
            if (!rc) rc = ops.side_publish_join(2 * t + 1);
            pending = carried ? 0 : 2 * t + 1;
          }
          const int rc2 = ops.side_close();
          if (rc || rc2) return rc ? rc : rc2;
        }
        // the follower's pieces are long (a K = c0 update of all its rows): on a stream of their own, or the next
        // group's remainder update -- which the chain waits for -- would queue behind them
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
};
