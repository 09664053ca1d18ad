// Host orchestration shared by the entry-point families of libgpflowslim_hip.so (gps_handle.hip, gps_gpr.hip, gps_cond.hip,
// gps_dist.hip, gps_sparse.hip): the HIP "Ops" policy of blocked.hpp and the helpers around a factorisation.  Every arithmetic
// step is a HIP kernel from the sibling .hip files.
#pragma once
#include "gps_common.hpp"
#include "blocked.hpp"
#include <climits>
#include <cmath>
#include <algorithm>

// ---- the HIP "Ops" policy for blocked.hpp ----------------------------------------------------
struct HipOps {
  gps_handle_t h;
  double* linv;     // [nblk][128*128]
  double* linvT;    // optional transposed inverses (same layout) or nullptr
  int* d_info;
  int factor = 1;   // 0: matrix already holds L, only build the inverses
  bool store_T = true;   // potrf_base also stores the transposed inverse (false: produced later by transpose_blocks)

  int potrf_base(double* A, i64 lda, i64 blk, i64 row0) {
    if (h->plain_linv == linv) h->plain_linv = nullptr;         // (these block inverses are being produced anew: classify_blocks again)
    return gps_launch_potrf_base(h, A, lda, linv + blk * GPS_TILE * GPS_TILE,
                                 (linvT && store_T) ? linvT + blk * GPS_TILE * GPS_TILE : nullptr, d_info, row0,
                                 factor);
  }
  // B[m,128] = B * Linv[blk]^T  (transposed == 0)   or   B * Linv[blk]  (transposed == 1)
  // D: the diagonal block the leaf solves against (lower block of L, or the upper block of U = L^T when transposed)
  int trsm_base(i64 blk, int transposed, double* B, i64 ldb, i64 m, const double* D, i64 ldd) {
    const double* W = (transposed ? linvT : linv) + blk * GPS_TILE * GPS_TILE;
    if (h->refine_now && !plain_ok(blk, 1)) { h->leaves_refined++; return gps_launch_trsm_leaf_refine(h, B, ldb, m, W, D, ldd, transposed); }
    if (h->refine_now) h->leaves_plain++;
    return gps_launch_gemm_nt(h, /*op set*/ 1, 0, m, GPS_TILE, GPS_TILE, B, ldb, W, GPS_TILE, B, ldb);
  }
  // refine mode: blocks blk .. blk + cnt - 1 of THIS factor have been classified well conditioned (classify_blocks): plain leaves
  bool plain_ok(i64 blk, i64 cnt) const {
    if (h->plain_linv != linv || blk < 0 || (size_t)(blk + cnt) > h->plain_flags.size()) return false;
    for (i64 b = blk; b < blk + cnt; ++b) if (!h->plain_flags[(size_t)b]) return false;
    return true;
  }
  // many rows against few columns: 512-column panels left to right (blocked.hpp::tall_panels)
  bool trsm_left_looking(i64 m, i64 n) const { return h->trsm_tall_ratio > 0 && m >= (i64)h->trsm_tall_ratio * n; }
  // four leaves and the updates between them as one launch (trsm_panel.hip); not for refined leaves
  bool leaf512(i64 m, int transposed, i64 blk) const {
    return h->trsm_panel > 0 && (!h->refine_now || plain_ok(blk, 4)) && m >= 64 && m % 64 == 0 && (transposed ? linvT != nullptr : true);
  }
  int trsm_leaf512(i64 blk, int transposed, double* B, i64 ldb, i64 m, const double* D, i64 ldd) {
    if (h->refine_now) h->leaves_plain += 4;
    const double* W = (transposed ? linvT : linv) + blk * GPS_TILE * GPS_TILE;
    return gps_launch_trsm_panel(h, B, ldb, m, D, ldd, W, transposed);
  }
  int gemm(int op, int lower, i64 M, i64 N, i64 K, const double* A, i64 lda, const double* B,
           i64 ldb, double* C, i64 ldc) {
    // Beside a latency chain (the follower / deferred stream) a rectangular update goes out in launches of at most
    // follower_max_wgs 128 x 128 tiles: a launch with more workgroups than the GPU holds keeps handing freed slots to its own
    // waiting workgroups, and the chain's launches -- a few hundred short workgroups each -- queue behind them (measured,
    // N = 8192: chain steps of 49 - 218 us beside the follower, 41 - 58 us without one).  With every workgroup of a launch
    // resident at once the CUs drain towards its end and the chain gets them.
    if (h->follower_max_wgs > 0 && op == 0 && !lower && h->def_stream && h->stream == h->def_stream && M >= 128 && N > 128) {
      const i64 rows = (M + 127) / 128;
      const i64 w = std::max<i64>(1, h->follower_max_wgs / rows) * 128;
      if (w < N) {
        for (i64 u = 0; u < N; u += w) {
          const int rc = gps_launch_gemm_nt(h, op, lower, M, std::min(w, N - u), K, A, lda, B + u * ldb, ldb, C + u, ldc);
          if (rc) return rc;
        }
        return GPS_OK;
      }
    }
    return gps_launch_gemm_nt(h, op, lower, M, N, K, A, lda, B, ldb, C, ldc);
  }
  // a product with a triangular operand (tri: 1 A upper, 2 A lower, 3 B lower) and / or a batch of them (blocked.hpp: wide_inverse)
  int gemm_ex(int op, int tri, i64 M, i64 N, i64 K, const double* A, i64 lda, const double* B, i64 ldb, double* C, i64 ldc, const GemmBatch* bt) {
    return gps_launch_gemm_nt_ex(h, op, 0, tri, M, N, K, A, lda, B, ldb, C, ldc, bt);
  }
  int blocks_to_diag(const double* src, double* dst, i64 nblk, i64 wb) { return gps_launch_blocks_to_diag(h, src, dst, nblk, wb); }
  int trsv_base(i64 blk, double* y, i64 ldy, i64 r, const double* D, i64 ldd) {
    if (!linvT) return gps_fail(h, GPS_ERR_STATE, "trsv needs the transposed block inverses");
    if (h->refine_now) return gps_launch_trsv_leaf_refine(h, linvT + blk * GPS_TILE * GPS_TILE, D, ldd, y, ldy, r, 0);
    return gps_launch_trsv_base(h, linvT + blk * GPS_TILE * GPS_TILE, y, ldy, r);
  }
  int gemv_sub(const double* L21, i64 ldl, i64 n2, i64 n1, const double* y1, double* y2, i64 ldy,
               i64 r) {
    return gps_launch_gemv_sub(h, L21, ldl, n2, n1, y1, y2, ldy, r);
  }
  // ---- pieces of the gradient path
  int trsv_t_base(i64 blk, double* y, i64 ldy, i64 r, const double* D, i64 ldd) {       // y = Linv^T y : the kernel wants M[c][i] = Linv[c][i]
    if (h->refine_now) return gps_launch_trsv_leaf_refine(h, linv + blk * GPS_TILE * GPS_TILE, D, ldd, y, ldy, r, 1);
    return gps_launch_trsv_base(h, linv + blk * GPS_TILE * GPS_TILE, y, ldy, r);
  }
  int gemv_t_sub(const double* L21, i64 ldl, i64 n2, i64 n1, const double* y2, double* y1, i64 ldy, i64 r) {
    return gps_launch_gemv_t_sub(h, L21, ldl, n2, n1, y2, y1, ldy, r);
  }
  int copy_linvT(i64 blk, double* Y, i64 ldy) {
    if (!linvT) return gps_fail(h, GPS_ERR_STATE, "inverse needs the transposed block inverses");
    GPS_HIP(h, hipMemcpy2DAsync(Y, (size_t)ldy * 8, linvT + blk * GPS_TILE * GPS_TILE, (size_t)GPS_TILE * 8,
                                (size_t)GPS_TILE * 8, GPS_TILE, hipMemcpyDeviceToDevice, h->stream));
    return GPS_OK;
  }
  i64 rl_max() const { return h->potrf_rl_max; }   // diagonal blocks up to this size: right-looking panel sweep
  i64 rl_group() const { return h->potrf_rl_group; }
  // ---- look-ahead of the sweep (see blocked.hpp::potrf_rl_groups)
  i64 lookahead_min_rows() const { return 1024; }   // rows of the remainder from which the hand-over pays
  bool lookahead() {
    // never on an external stream (gps_set_stream): that may be the legacy default stream, which synchronises
    // implicitly with a blocking side stream -- the hand-over would wait on itself
    if (!h->potrf_lookahead || h->ext_stream) return false;
    if (!h->side_stream) {
      // The side stream leaves some CUs alone (la_mask_word0; mask bit i = CU i/8 of XCD i%8): potrf_base needs a whole
      // CU's LDS, and a GEMM that keeps refilling every CU with small workgroups would starve it until its own tail.
      hipError_t e;
      if (h->prop.multiProcessorCount == 256) {
        const uint32_t mask[8] = {GPS_LA_MASK_WORD0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        e = hipExtStreamCreateWithCUMask(&h->side_stream, 8, mask);
      } else {
        e = hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking);
      }
      if (e != hipSuccess) { h->side_stream = nullptr; return false; }
      // all or nothing: a half-initialised look-ahead (stream without flags / event) must not be used by the next call
      // The flags are cleared ON THE HANDLE'S STREAM and the clear is complete before anything uses them.  (A plain
      // hipMemset goes to the legacy default stream and is asynchronous to the host: it runs whenever every blocking
      // stream of the process -- other handles' CU-masked side streams -- has drained, which with several handles in
      // one process could be after this handle's first tickets, or its time-out count, had been written: found with
      // four handles in four host threads, a round-3 probe; tests/test_gpu_kernels.py::test_concurrent_small_launches runs that regime now.)
      if (h->dLaFlags.ensure(64) != hipSuccess || hipMemsetAsync(h->dLaFlags.p, 0, 64, h->stream) != hipSuccess ||
          hipStreamSynchronize(h->stream) != hipSuccess ||
          (!h->ev_la && hipEventCreateWithFlags(&h->ev_la, hipEventDisableTiming) != hipSuccess)) {
        (void)hipStreamDestroy(h->side_stream);
        h->side_stream = nullptr;
        return false;
      }
      h->la_ticket = 0; h->fol_ticket = 0;
    }
    return true;
  }
  unsigned long long* la_flags() const { return (unsigned long long*)h->dLaFlags.p; }
  // fork: the NEXT GEMM launched on the chain publishes the ticket when it starts; the side stream waits for it
  unsigned long long la_fork() {
    const unsigned long long t = ++h->la_ticket;
    h->next_sig_ptr = la_flags(); h->next_sig_val = t;
    return t;
  }
  hipStream_t saved_stream = nullptr;
  hipStream_t saved_stream_d = nullptr;
  bool follower() { return lookahead() && aux_stream(); }
  i64 follower_cols() const { return 512; }         // the follower solve goes out in pieces of at least this many columns
  // `first`: first hand-over of a sweep.  The side stream is then idle and its wait kernel would start at once and spin
  // until the chain gets here -- through whole big GEMMs of the level above, where one extra resident wave costs a CU
  // its second GEMM workgroup (measured: every big launch 4-10 % slower).  An event keeps the queue parked instead;
  // inside a sweep the waits are tens of microseconds and stay in-kernel.
  int side_open(unsigned long long t, bool first) {
    if (first) {
      GPS_HIP(h, hipEventRecord(h->ev_la, h->stream));
      GPS_HIP(h, hipStreamWaitEvent(h->side_stream, h->ev_la, 0));
    }
    int rc = gps_launch_la_wait(h, h->side_stream, nullptr, 0, la_flags(), t, la_flags() + 2);
    if (rc) return rc;
    saved_stream = h->stream; h->stream = h->side_stream;
    return GPS_OK;
  }
  int side_publish_join(unsigned long long t) {       // after the remainder update: the join ticket
    return gps_launch_la_wait(h, h->side_stream, la_flags() + 1, t, nullptr, 0, la_flags() + 2);
  }
  int side_close() { h->stream = saved_stream; saved_stream = nullptr; return GPS_OK; }
  // follower solve: on the deferred stream (created by follower()); it waits for the same fork ticket as the side
  // stream, publishes a ticket of its own after each piece, and the chain waits for the last one
  int follower_open(unsigned long long t, bool first) {
    if (first) {
      GPS_HIP(h, hipEventRecord(h->ev_def_fork, h->stream));
      GPS_HIP(h, hipStreamWaitEvent(h->def_stream, h->ev_def_fork, 0));
    }
    int rc = gps_launch_la_wait(h, h->def_stream, nullptr, 0, la_flags(), t, la_flags() + 2);
    if (rc) return rc;
    saved_stream_d = h->stream; h->stream = h->def_stream;
    return GPS_OK;
  }
  int follower_close() { h->stream = saved_stream_d; saved_stream_d = nullptr; return GPS_OK; }
  int follower_publish() {
    return gps_launch_la_wait(h, h->def_stream, la_flags() + 3, ++h->fol_ticket, nullptr, 0, la_flags() + 2);
  }
  int follower_join() {
    return gps_launch_la_wait(h, h->stream, nullptr, 0, la_flags() + 3, h->fol_ticket, la_flags() + 2);
  }
  // ---- deferred stream: big pieces of a parent's panel solve that run beside a child's sweep (coarse: events)
  bool deferred() { return lookahead() && aux_stream(); }
  bool aux_stream() {
    if (!h->def_stream) {
      hipError_t e;
      if (h->prop.multiProcessorCount == 256) {
        const uint32_t mask[8] = {GPS_LA_MASK_WORD0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        e = hipExtStreamCreateWithCUMask(&h->def_stream, 8, mask);
      } else {
        e = hipStreamCreateWithFlags(&h->def_stream, hipStreamNonBlocking);
      }
      if (e != hipSuccess) { h->def_stream = nullptr; return false; }
      // (the events outlive the stream: it is re-created after every external-stream episode, they are not)
      if ((!h->ev_def_fork && hipEventCreateWithFlags(&h->ev_def_fork, hipEventDisableTiming) != hipSuccess) ||
          (!h->ev_def_join && hipEventCreateWithFlags(&h->ev_def_join, hipEventDisableTiming) != hipSuccess)) {
        (void)hipStreamDestroy(h->def_stream);
        h->def_stream = nullptr;
        return false;
      }
    }
    return true;
  }
  int deferred_open() {
    GPS_HIP(h, hipEventRecord(h->ev_def_fork, h->stream));
    GPS_HIP(h, hipStreamWaitEvent(h->def_stream, h->ev_def_fork, 0));
    saved_stream_d = h->stream; h->stream = h->def_stream;
    return GPS_OK;
  }
  int deferred_close() {
    hipError_t e = hipEventRecord(h->ev_def_join, h->def_stream);
    h->stream = saved_stream_d; saved_stream_d = nullptr;
    GPS_HIP(h, e);
    return GPS_OK;
  }
  int deferred_join() {
    GPS_HIP(h, hipStreamWaitEvent(h->stream, h->ev_def_join, 0));
    return GPS_OK;
  }
  int chain_join(unsigned long long t) {
    // diagnostics ("la_fault_inject" = k): the k-th join from now waits for a ticket that never comes, i.e. takes the
    // time-out path of a missed hand-over (tests/test_gpu_kernels.py::test_lookahead_timeout_is_retried)
    if (h->la_fault_inject > 0 && --h->la_fault_inject == 0) t = ~0ull;
    return gps_launch_la_wait(h, h->stream, nullptr, 0, la_flags() + 1, t, la_flags() + 2);
  }
  bool fill_zeros() const { return false; }     // nothing on the device path reads L^-T below its diagonal blocks
  int zero_block(double* Y, i64 ldy, i64 rows, i64 cols) {
    GPS_HIP(h, hipMemset2DAsync(Y, (size_t)ldy * 8, 0, (size_t)cols * 8, (size_t)rows, h->stream));
    return GPS_OK;
  }
};

// L a = y / L^T a = y for r right-hand sides (rows of y): one wavefront launch (trsv_wave.hip) -- with one refinement step
// per diagonal block inside the wavefront where the leaves are to be refined (jittered / low-noise factors; round 4: the
// recursive substitution with refined leaves, 4 N / 128 launches, stays as the option "trsv_wave_refine" = 0).
static int trsv_forward(gps_handle_t h, HipOps& ops, const double* L, i64 ldl, i64 n, double* y, i64 ldy, i64 r) {
  if (h->trsv_wave && (!h->refine_now || h->trsv_wave_refine) && ops.linvT && n >= 2 * GPS_TILE)
    return gps_launch_trsv_wave(h, L, ldl, n, ops.linvT, y, ldy, r, 0, h->refine_now ? 1 : 0);
  Blocked<HipOps> bl(ops);
  return bl.trsv_rec(L, ldl, n, 0, y, ldy, r);
}
static int trsv_backward(gps_handle_t h, HipOps& ops, const double* L, i64 ldl, i64 n, double* y, i64 ldy, i64 r) {
  if (h->trsv_wave && (!h->refine_now || h->trsv_wave_refine) && n >= 2 * GPS_TILE)
    return gps_launch_trsv_wave(h, L, ldl, n, ops.linv, y, ldy, r, 1, h->refine_now ? 1 : 0);
  Blocked<HipOps> bl(ops);
  return bl.trsv_t_rec(L, ldl, n, 0, y, ldy, r);
}

static int read_info(gps_handle_t h, int* d_info, int* info) {
  int v = 0;
  unsigned long long la_timeouts = 0;
  GPS_HIP(h, hipMemcpyAsync(&v, d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (h->dLaFlags.p) GPS_HIP(h, hipMemcpyAsync(&la_timeouts, (unsigned long long*)h->dLaFlags.p + 2, 8, hipMemcpyDeviceToHost, h->stream));
  unsigned wave_gave_up = 0;
  if (h->dWaveCtl.p) GPS_HIP(h, hipMemcpyAsync(&wave_gave_up, (unsigned*)h->dWaveCtl.p + 1, 4, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  if (info) *info = (v == INT_MAX) ? 0 : v;
  if (wave_gave_up) {
    // a bounded wait of the trsv wavefront gave up (never seen; the bound is there so that a scheduling surprise is an
    // error, not a hung GPU): the result is poisoned -- switch the wavefront off for this handle and have the entry
    // point run the evaluation again through the recursive substitution (with_la_retry)
    (void)hipMemsetAsync((unsigned*)h->dWaveCtl.p + 1, 0, 4, h->stream);
    h->trsv_wave = 0;
    h->wave_fallbacks++;
    h->la_timed_out = true;
    return gps_fail(h, GPS_ERR_STATE, "trsv wavefront timed out (result invalid)");
  }
  if (la_timeouts) {
    // not sticky: the counter is cleared (stream-ordered) so that the handle is usable again; the entry point re-runs
    // the evaluation once without look-ahead (with_la_retry)
    (void)hipMemsetAsync((unsigned long long*)h->dLaFlags.p + 2, 0, 8, h->stream);
    h->la_timed_out = true;
    return gps_fail(h, GPS_ERR_STATE, "look-ahead hand-over timed out (result invalid)");
  }
  return GPS_OK;
}

// A missed hand-over of the look-ahead (a bounded wait of la_wait_kernel that gave up) invalidates the evaluation, not
// the handle: run the entry point's body again, once, with the look-ahead off -- same process, same handle, same
// (host-owned, unchanged) inputs -- and count it.
template <class F>
static int with_la_retry(gps_handle_t h, F&& body) {
  if (h) h->la_timed_out = false;
  int rc = body();
  if (h && rc == GPS_ERR_STATE && h->la_timed_out) {
    h->la_timed_out = false;
    h->la_retries++;
    (void)hipDeviceSynchronize();
    // (kernels of the side streams that were still waiting for a hand-over of the failed attempt ran into their own bounds
    // after the counter was cleared: clear it again now that everything has drained)
    if (h->dLaFlags.p) { (void)hipMemsetAsync((unsigned long long*)h->dLaFlags.p + 2, 0, 8, h->stream); (void)hipStreamSynchronize(h->stream); }
    const int saved = h->potrf_lookahead;
    h->potrf_lookahead = 0;
    rc = body();
    h->potrf_lookahead = saved;
  }
  return rc;
}

// Refine mode, after a factorisation: kappa_1 of every diagonal block from the factor and its block inverses (one small launch,
// one read-back); blocks at or below "leaf_plain_kappa" are solved against by the plain product from now on (HipOps::plain_ok).
static int classify_blocks(gps_handle_t h, const HipOps& ops, const double* L, i64 ldl, i64 n) {
  h->plain_linv = nullptr;
  if (!h->refine_now || !(h->leaf_plain_kappa > 0.0) || n < GPS_TILE) return GPS_OK;
  const i64 nblk = n / GPS_TILE;
  GPS_HIP(h, h->dBlkCond.ensure((size_t)nblk * 8));
  int rc = gps_launch_block_cond(h, L, ldl, ops.linv, nblk, h->dBlkCond.d());
  if (rc) return rc;
  std::vector<double> k((size_t)nblk);
  GPS_HIP(h, hipMemcpyAsync(k.data(), h->dBlkCond.p, (size_t)nblk * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  h->plain_flags.assign((size_t)nblk, 0);
  for (i64 b = 0; b < nblk; ++b) h->plain_flags[(size_t)b] = (k[(size_t)b] == k[(size_t)b] && k[(size_t)b] <= h->leaf_plain_kappa) ? 1 : 0;   // (NaN: not positive definite -- refine)
  h->plain_linv = ops.linv;
  return GPS_OK;
}

// a cooperative launch of the small-N path gave up: counted; the fourth in a row sends the handle's next 256 evaluations of
// that size launch by launch (a GPU shared with something that holds its CUs must not cost a bounded wait per optimiser step;
// gps_profile_get "small_n_cooldown" reads what is left of the back-off)
static void small_gave_up(gps_handle_t h) {
  h->small_fallbacks++;
  if (++h->small_consec >= 4) { h->small_cooldown = 256; h->small_consec = 0; }
}
static int stage_time(gps_handle_t h, int a, int b, double* out) {
  float ms = 0.f;
  GPS_HIP(h, hipEventElapsedTime(&ms, h->ev[a], h->ev[b]));
  *out = ms;
  return GPS_OK;
}

// Cholesky adjoint on the device (Murray 2016, eq. 10):  for L = chol(K) and a lower-triangular cotangent Lbar, the symmetric
// Kbar with <Kbar, dK> = <Lbar, dL> is  L^-T (Phi(P) + Phi(P)^T) L^-1 / 2,  P = L^T Lbar, Phi = lower triangle with halved
// diagonal.  U = L^T (upper, row-major; `bl` carries L's block inverses).  out <- 2 Kbar (the callers fold the 1/2); P: scratch.
static int chol_adjoint2(gps_handle_t h, Blocked<HipOps>& bl, const double* U, const double* Lbar, double* out, double* P, i64 mp) {
  int rc = gps_launch_transpose(h, Lbar, mp, mp, mp, out, mp);
  if (rc) return rc;
  rc = gps_launch_gemm_nt(h, 1, 0, mp, mp, mp, U, mp, out, mp, P, mp);                 // P[i][j] = sum_k L[k][i] Lbar[k][j]
  if (rc) return rc;
  rc = gps_launch_tri_map(h, P, mp, mp, 0);               // Phi(P) + Phi(P)^T = the lower triangle of P mirrored
  if (rc) return rc;
  rc = bl.trsm_rn_rec(U, mp, mp, 0, P, mp, mp);                                        // Y = Psym L^-1
  if (rc) return rc;
  rc = gps_launch_transpose(h, P, mp, mp, mp, out, mp);                                // Y^T
  if (rc) return rc;
  return bl.trsm_rn_rec(U, mp, mp, 0, out, mp, mp);                                    // Y^T L^-1 = (L^-T Y)^T  (symmetric)
}
