// C ABI of libgpflowslim_hip.so (declared in include/gpflowslim_hip.h).
// Host orchestration only: every arithmetic step is a HIP kernel from the sibling .hip files.
#include "gps_common.hpp"
#include "blocked.hpp"
#include <climits>
#include <cmath>

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

static int gpr_lml_finish(gps_handle_t h, i64 r, double* lml);
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

// ---- life cycle ----------------------------------------------------------------------------------
extern "C" int gps_create(int device_id, gps_handle_t* out) {
  if (!out) return GPS_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GPS_ERR_HIP;
  if (device_id < 0 || device_id >= count) return GPS_ERR_ARG;
  gps_handle_t h = new gps_handle_s();
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&h->prop, device_id) != hipSuccess) {
    delete h;
    return GPS_ERR_HIP;
  }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return GPS_ERR_HIP; }
  for (int i = 0; i < 8; ++i) {
    if (hipEventCreate(&h->ev[i]) != hipSuccess) { delete h; return GPS_ERR_HIP; }
  }
  if (const char* la = getenv("GPS_LOOKAHEAD")) h->potrf_lookahead = atoi(la);     // diagnostics: counter collection serialises the dispatches (tools/collect_profiles.sh)
  if (h->dInfo.ensure(64) != hipSuccess || h->dScal.ensure(4096) != hipSuccess) { delete h; return GPS_ERR_HIP; }
  *out = h;
  return GPS_OK;
}

// every growable device buffer of the handle (the small fixed ones -- info word, look-ahead flags, pinned ring -- stay)
static void release_work_buffers(gps_handle_t h, bool all) {
  DevBuf* bufs[] = {&h->dX, &h->dK, &h->dLinv, &h->dAlpha, &h->dFeat, &h->dFeat2, &h->dProg,
                    &h->dXnew, &h->dB, &h->dMean, &h->dVar, &h->dTmp, &h->dTmp2, &h->dTmp3, &h->dA, &h->dY,
                    &h->dKinv, &h->dNkn, &h->dS1, &h->dS2, &h->dS3, &h->dS4, &h->dGemvWs, &h->dGemvCnt, &h->dGemmWs, &h->dGemmCnt, &h->dBlkCond, &h->dStage,
                    &h->dDistScal, &h->dGradSums, &h->dSmallOut, &h->dFeatG, &h->dG1, &h->dG2, &h->dG3, &h->dG4, &h->dWave, &h->dDistComm[0], &h->dDistComm[1], &h->dDistComm[2]};
  for (DevBuf* b : bufs) b->release();
  if (all) h->dSmallSync.release();
  if (all) { h->dInfo.release(); h->dScal.release(); h->dWaveCtl.release(); }      // (allocated by gps_create; every reduction writes there)
}

// Hand the handle's device memory back to the allocator (K / L of a large problem is N^2 x 8 bytes and stays allocated
// for re-use otherwise).  The resident data set and factor are gone afterwards: gps_gpr_set_data again before the next
// GPR call.  Streams, events and options are kept.
extern "C" int gps_release_buffers(gps_handle_t h) {
  if (!h) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  if (h->side_stream) GPS_HIP(h, hipStreamSynchronize(h->side_stream));
  if (h->def_stream) GPS_HIP(h, hipStreamSynchronize(h->def_stream));
  gps_profile_collect(h);
  release_work_buffers(h, false);
  h->have_factor = false; h->dist_have_part_factor = false; h->n = 0; h->npad = 0; h->r = 0;
  h->dist_np = 0; h->dist_nb = 0;          // (a distributed factorisation must start over with gps_dist_begin)
  return GPS_OK;
}

extern "C" int gps_destroy(gps_handle_t h) {
  if (!h) return GPS_OK;
  (void)hipSetDevice(h->device);
  (void)gps_comm_destroy(h);
  (void)hipStreamSynchronize(h->stream);
  gps_profile_collect(h);
  for (auto e : h->evt_pool) (void)hipEventDestroy(e);
  for (int i = 0; i < 8; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  release_work_buffers(h, true);
  (void)hipStreamDestroy(h->ext_stream ? h->own_stream : h->stream);
  if (h->side_stream) { (void)hipStreamSynchronize(h->side_stream); (void)hipStreamDestroy(h->side_stream); }
  if (h->def_stream) { (void)hipStreamSynchronize(h->def_stream); (void)hipStreamDestroy(h->def_stream); }
  if (h->dist_chain) { (void)hipStreamSynchronize(h->dist_chain); (void)hipStreamDestroy(h->dist_chain); }
  if (h->dist_bulk_own) { (void)hipStreamSynchronize(h->dist_bulk_own); (void)hipStreamDestroy(h->dist_bulk_own); }
  for (auto e : h->dist_events) (void)hipEventDestroy(e);
  if (h->ev_def_fork) (void)hipEventDestroy(h->ev_def_fork);
  if (h->ev_def_join) (void)hipEventDestroy(h->ev_def_join);
  h->dLaFlags.release();
  h->ring.release();
  if (h->hRes) (void)hipHostFree(h->hRes);
  if (h->ev_la) (void)hipEventDestroy(h->ev_la);
  delete h;
  return GPS_OK;
}

extern "C" const char* gps_last_error(gps_handle_t h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int gps_device_info(gps_handle_t h, char* name, int name_len, int* n_cu, int64_t* hbm_bytes,
                               char* arch, int arch_len) {
  if (!h) return GPS_ERR_ARG;
  if (name && name_len > 0) { strncpy(name, h->prop.name, name_len - 1); name[name_len - 1] = 0; }
  if (arch && arch_len > 0) { strncpy(arch, h->prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
  if (n_cu) *n_cu = h->prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)h->prop.totalGlobalMem;
  return GPS_OK;
}

// ---- measurement ---------------------------------------------------------------------------------
extern "C" int gps_profile_enable(gps_handle_t h, int on) {
  if (!h) return GPS_ERR_ARG;
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  gps_profile_collect(h);
  h->prof_on = on != 0;
  return GPS_OK;
}
extern "C" int gps_profile_reset(gps_handle_t h) {
  if (!h) return GPS_ERR_ARG;
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  gps_profile_collect(h);
  for (int i = 0; i < KC_COUNT; ++i) h->stat[i] = KClassStat();
  return GPS_OK;
}
extern "C" int gps_profile_get(gps_handle_t h, const char* klass, int64_t* launches, double* ms,
                               double* flops, double* bytes) {
  if (!h || !klass) return GPS_ERR_ARG;
  if (strcmp(klass, "lookahead_retries") == 0) {      // evaluations re-run without look-ahead after a missed hand-over
    if (launches) *launches = h->la_retries;
    if (ms) *ms = 0.0; if (flops) *flops = 0.0; if (bytes) *bytes = 0.0;
    return GPS_OK;
  }
  if (strcmp(klass, "factor_refined") == 0) {         // 1: the resident GPR factor was built (and is solved) with refined leaves
    if (launches) *launches = h->factor_refine ? 1 : 0;
    if (ms) *ms = 0.0; if (flops) *flops = 0.0; if (bytes) *bytes = 0.0;
    return GPS_OK;
  }
  if (strcmp(klass, "small_n_fallbacks") == 0) {      // one-launch factorisations of small problems that gave up and were redone launch by launch
    if (launches) *launches = (int64_t)h->small_fallbacks;
    if (ms) *ms = 0.0; if (flops) *flops = 0.0; if (bytes) *bytes = 0.0;
    return GPS_OK;
  }
  if (strcmp(klass, "leaves_plain") == 0 || strcmp(klass, "leaves_refined") == 0) {     // leaf launches in refine mode, by kind
    if (launches) *launches = (int64_t)(klass[7] == 'p' ? h->leaves_plain : h->leaves_refined);
    if (ms) *ms = 0.0; if (flops) *flops = 0.0; if (bytes) *bytes = 0.0;
    return GPS_OK;
  }
  if (strcmp(klass, "small_n_cooldown") == 0) {       // evaluations the small-N back-off (small_gave_up) still sends launch by launch
    if (launches) *launches = (int64_t)h->small_cooldown;
    if (ms) *ms = 0.0; if (flops) *flops = 0.0; if (bytes) *bytes = 0.0;
    return GPS_OK;
  }
  if (strcmp(klass, "trsv_wave_fallbacks") == 0) {    // wavefront substitutions that gave up (handle fell back to the recursive one)
    if (launches) *launches = (int64_t)h->wave_fallbacks;
    if (ms) *ms = 0.0; if (flops) *flops = 0.0; if (bytes) *bytes = 0.0;
    return GPS_OK;
  }
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  gps_profile_collect(h);
  for (int i = 0; i < KC_COUNT; ++i) {
    if (strcmp(klass, kc_names[i]) == 0) {
      if (launches) *launches = h->stat[i].launches;
      if (ms) *ms = h->stat[i].ms;
      if (flops) *flops = h->stat[i].flops;
      if (bytes) *bytes = h->stat[i].bytes;
      return GPS_OK;
    }
  }
  return gps_fail(h, GPS_ERR_ARG, "unknown kernel class");
}
extern "C" int gps_last_stage_ms(gps_handle_t h, double* out5) {
  if (!h || !out5) return GPS_ERR_ARG;
  for (int i = 0; i < 5; ++i) out5[i] = h->stage_ms[i];
  return GPS_OK;
}

extern "C" int gps_set_option(gps_handle_t h, const char* key, double value) {
  if (!h || !key) return GPS_ERR_ARG;
  if (strcmp(key, "gemm_force_tile") == 0) { h->gemm_force_tb = (int)value; return GPS_OK; }
  if (strcmp(key, "gemm_tail_split") == 0) { h->gemm_tail_split = (int)value; return GPS_OK; }
  if (strcmp(key, "kmat_fast") == 0) { h->kmat_fast = (int)value; return GPS_OK; }
  if (strcmp(key, "kmat_mfma") == 0) { h->kmat_mfma = (int)value; return GPS_OK; }
  if (strcmp(key, "trsv_wave") == 0) { h->trsv_wave = (int)value; return GPS_OK; }
  if (strcmp(key, "gpr_aug_rows") == 0) { h->gpr_aug_rows = (int)value; return GPS_OK; }
  if (strcmp(key, "leaf_refine") == 0) { h->leaf_refine = (int)value; return GPS_OK; }
  if (strcmp(key, "potrf_rl_max") == 0) { h->potrf_rl_max = (int)value; return GPS_OK; }
  if (strcmp(key, "potrf_lookahead") == 0) { h->potrf_lookahead = (int)value; return GPS_OK; }
  if (strcmp(key, "la_fault_inject") == 0) { h->la_fault_inject = (int)value; return GPS_OK; }
  if (strcmp(key, "wave_fault_inject") == 0) { h->wave_fault_inject = (int)value; return GPS_OK; }
  if (strcmp(key, "small_fault_inject") == 0) { h->small_fault_inject = (int)value; return GPS_OK; }
  if (strcmp(key, "svgp_kl_weight") == 0) { h->svgp_kl_weight = value; return GPS_OK; }
  if (strcmp(key, "dist_partitioned") == 0) { h->dist_partitioned = (int)value; return GPS_OK; }
  if (strcmp(key, "leaf_plain_kappa") == 0) { h->leaf_plain_kappa = value; h->plain_linv = nullptr; return GPS_OK; }
  if (strcmp(key, "follower_max_wgs") == 0) { h->follower_max_wgs = (int)value; return GPS_OK; }
  if (strcmp(key, "potrf_rl_group") == 0) { h->potrf_rl_group = (int)value < 1 ? 1 : (int)value; return GPS_OK; }
  if (strcmp(key, "small_n") == 0) { h->small_n = (int)value; return GPS_OK; }
  if (strcmp(key, "trsm_panel") == 0) { h->trsm_panel = (int)value; return GPS_OK; }
  if (strcmp(key, "trsm_tall_ratio") == 0) { h->trsm_tall_ratio = (int)value; return GPS_OK; }
  if (strcmp(key, "trsm_panel_rows") == 0) {
    if (value != 0 && value != 32 && value != 64 && value != 65) return gps_fail(h, GPS_ERR_ARG, "trsm_panel_rows: 0, 32, 64 or 65");
    h->trsm_panel_rows = (int)value; return GPS_OK;
  }
  if (strcmp(key, "trsv_wave_refine") == 0) { h->trsv_wave_refine = (int)value; return GPS_OK; }
  return gps_fail(h, GPS_ERR_ARG, "unknown option");
}

// ---- diagnostics ---------------------------------------------------------------------------------
extern "C" int gps_diag_mfma_f64(gps_handle_t h, int waves_per_simd, double* tflops, int* layout_ok) {
  if (!h) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  return gps_run_mfma_diag(h, waves_per_simd, tflops, layout_ok);
}

extern "C" int gps_diag_gemm_nt(gps_handle_t h, int op, int lower, int64_t m, int64_t n, int64_t k,
                                const double* A, const double* B, double* C) {
  if (!h || !A || !B || !C) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  const size_t ab = (size_t)m * k * 8, bb = (size_t)n * k * 8, cb = (size_t)m * n * 8;
  GPS_HIP(h, h->dTmp.ensure(ab)); GPS_HIP(h, h->dTmp2.ensure(bb)); GPS_HIP(h, h->dTmp3.ensure(cb));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp.p, A, ab, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, B, bb, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp3.p, C, cb, hipMemcpyHostToDevice, h->stream));
  int rc = gps_launch_gemm_nt(h, op, lower, m, n, k, h->dTmp.d(), k, h->dTmp2.d(), k, h->dTmp3.d(), n);
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(C, h->dTmp3.p, cb, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

extern "C" int gps_diag_gemm_timeline(gps_handle_t h, int op, int lower, int64_t m, int64_t n, int64_t k, int reps,
                                      long long* stamps_out, int64_t cap_blocks, int64_t* nblocks, double* ms_per_launch) {
  if (!h || !stamps_out || cap_blocks <= 0 || reps <= 0) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  return gps_run_gemm_timeline(h, op, lower, m, n, k, reps, stamps_out, cap_blocks, nblocks, ms_per_launch);
}

// phase stamps (100 MHz ticks) of one potrf_base launch on a random SPD block: load, eliminate,
// scale + L store, (gap), inverse level 0, inverse levels, stores
extern "C" int gps_diag_potrf_base_stamps(gps_handle_t h, int factor, double* us_out7) {
  if (!h || !us_out7) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  const size_t bb = (size_t)GPS_TILE * GPS_TILE * 8;
  GPS_HIP(h, h->dTmp.ensure(4 * bb + 2048));
  std::vector<double> A((size_t)GPS_TILE * GPS_TILE, 0.0);
  for (int i = 0; i < GPS_TILE; ++i) for (int j = 0; j <= i; ++j) A[(size_t)i * GPS_TILE + j] = (i == j) ? 2.0 + 0.01 * i : 0.3 / (1.0 + i - j);
  double* dA = h->dTmp.d();
  long long* dS = (long long*)(dA + 3 * GPS_TILE * GPS_TILE);
  long long hs[96] = {0};
  const bool per_wave = getenv("GPS_PB_WAVE_STAMPS") != nullptr;      // per-wave phase-A stamps perturb the timing they measure
  for (int rep = 0; rep < 3; ++rep) {
    GPS_HIP(h, hipMemsetAsync(dS, 0, 96 * sizeof(long long), h->stream));
    if (per_wave) { const long long one = 1; GPS_HIP(h, hipMemcpyAsync(dS + 31, &one, sizeof(one), hipMemcpyHostToDevice, h->stream)); }
    GPS_HIP(h, hipMemcpyAsync(dA, A.data(), bb, hipMemcpyHostToDevice, h->stream));
    int rc = gps_launch_fill_info(h, (int*)h->dInfo.p, INT_MAX);
    if (rc) return rc;
    // (GPS_PB_NO_T: without the transposed inverse, as the GPR path runs it -- the transposes come from one batched launch)
    rc = gps_launch_potrf_base(h, dA, GPS_TILE, dA + GPS_TILE * GPS_TILE, getenv("GPS_PB_NO_T") ? nullptr : dA + 2 * GPS_TILE * GPS_TILE,
                               (int*)h->dInfo.p, 0, factor, dS);
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(hs, dS, sizeof(hs), hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
  }
  for (int q = 0; q < 7; ++q) us_out7[q] = (double)(hs[q] - hs[0]) * 0.01;
  // shader clock (MHz) held during the elimination phase
  us_out7[0] = (double)(hs[8 + 2] - hs[8 + 1]) / ((double)(hs[2] - hs[1]) * 0.01);
  if (factor) {
    fprintf(stderr, "potrf_base phases (us): A (panel | update + inverse in its shadow) %.2f  B (next block column) %.2f\n", hs[16] * 0.01, hs[17] * 0.01);
    for (int g = 0; g < 8 && per_wave; ++g) {
      fprintf(stderr, "  step %d: per-wave end of phase A (us):", g);
      for (int w = 0; w < 8; ++w) fprintf(stderr, " %.2f", hs[32 + 8 * g + w] * 0.01);
      fprintf(stderr, "\n");
    }
  }
  return GPS_OK;
}

// one 128-column leaf  X L11^T = B  (upper: X L11 = B through U = L^T) on m rows, timed over `reps` launches:
// Diagnostics: the 512-column triangular solve of m rows, launch by launch (panel = 0) or as one launch (panel = 1,
// trsm_panel.hip); backward: X L = B instead of X L^T = B.  maxdiff_out: largest |difference| between the two on the same input.
static int diag_trsm512_impl(gps_handle_t h, int64_t m, int backward, int panel, int reps, double* us_per_solve,
                             double* maxdiff_out, long long* stamps_out, int64_t cap_blocks) {
  if (!h || m <= 0 || m % GPS_TILE || reps <= 0 || !us_per_solve) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  const i64 T = GPS_TILE, n = 4 * T;
  std::vector<double> L((size_t)n * n, 0.0), B((size_t)T * n);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  for (i64 i = 0; i < n; ++i) for (i64 j = 0; j <= i; ++j) L[i * n + j] = (i == j) ? 2.0 + 0.5 * (rnd() + 0.5) : 0.1 * rnd();
  for (auto& v : B) v = rnd();
  GPS_HIP(h, h->dTmp.ensure((size_t)2 * n * n * 8));
  GPS_HIP(h, h->dTmp3.ensure((size_t)8 * T * T * 8));
  GPS_HIP(h, h->dB.ensure((size_t)m * n * 8 * 3));
  double* dL = h->dTmp.d(); double* dU = dL + n * n; double* dInv = h->dTmp3.d(); double* dInvT = dInv + 4 * T * T;
  double* dBm = h->dB.d(); double* dB0 = dBm + m * n; double* dB1 = dB0 + m * n;
  GPS_HIP(h, hipMemcpyAsync(dL, L.data(), (size_t)n * n * 8, hipMemcpyHostToDevice, h->stream));
  for (i64 q = 0; q < m / T; ++q) GPS_HIP(h, hipMemcpyAsync(dB0 + q * T * n, B.data(), (size_t)T * n * 8, hipMemcpyHostToDevice, h->stream));
  int rc = gps_launch_transpose(h, dL, n, n, n, dU, n);
  if (rc) return rc;
  rc = gps_launch_fill_info(h, (int*)h->dInfo.p, INT_MAX);
  if (rc) return rc;
  HipOps ops{h, dInv, dInvT, (int*)h->dInfo.p};
  ops.factor = 0;
  for (i64 b = 0; b < 4 && !rc; ++b) rc = ops.potrf_base(dL + b * T * n + b * T, n, b, b * T);
  if (rc) return rc;
  Blocked<HipOps> bl(ops);
  const bool saved_ref = h->refine_now; const int saved_panel = h->trsm_panel;
  h->refine_now = false;
  auto solve = [&](double* X) -> int { return backward ? bl.trsm_rn_rec(dU, n, n, 0, X, n, m) : bl.trsm_rec(dL, n, n, 0, X, n, m); };
  hipEvent_t e0, e1;
  GPS_HIP(h, hipEventCreate(&e0)); GPS_HIP(h, hipEventCreate(&e1));
  float ms = 0.f;
  h->trsm_panel = panel;
  for (int pass = 0; pass < 2 && !rc; ++pass) {            // pass 0 warms up
    GPS_HIP(h, hipEventRecord(e0, h->stream));
    for (int it = 0; it < reps && !rc; ++it) {
      if (it == 0 || it == reps - 1) GPS_HIP(h, hipMemcpyAsync(dBm, dB0, (size_t)m * n * 8, hipMemcpyDeviceToDevice, h->stream));
      rc = solve(dBm);
    }
    GPS_HIP(h, hipEventRecord(e1, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    GPS_HIP(h, hipEventElapsedTime(&ms, e0, e1));
  }
  if (!rc && stamps_out) {                                  // one more launch, every workgroup leaving its phase stamps
    const i64 nb = std::min<i64>(m / 32, cap_blocks);
    GPS_HIP(h, h->dGemvWs.ensure((size_t)(m / 32) * 32 * sizeof(long long)));
    GPS_HIP(h, hipMemsetAsync(h->dGemvWs.p, 0, (size_t)(m / 32) * 32 * sizeof(long long), h->stream));
    GPS_HIP(h, hipMemcpyAsync(dBm, dB0, (size_t)m * n * 8, hipMemcpyDeviceToDevice, h->stream));
    h->tp_stamps = (long long*)h->dGemvWs.p;
    rc = solve(dBm);
    h->tp_stamps = nullptr;
    GPS_HIP(h, hipMemcpyAsync(stamps_out, h->dGemvWs.p, (size_t)nb * 32 * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
  }
  if (!rc && maxdiff_out) {
    h->trsm_panel = panel ? 0 : 1;
    GPS_HIP(h, hipMemcpyAsync(dB1, dB0, (size_t)m * n * 8, hipMemcpyDeviceToDevice, h->stream));
    rc = solve(dB1);
    if (!rc) {
      std::vector<double> x0((size_t)m * n), x1((size_t)m * n);
      GPS_HIP(h, hipMemcpy(x0.data(), dBm, (size_t)m * n * 8, hipMemcpyDeviceToHost));
      GPS_HIP(h, hipMemcpy(x1.data(), dB1, (size_t)m * n * 8, hipMemcpyDeviceToHost));
      double w = 0.0;
      for (size_t i = 0; i < x0.size(); ++i) { const double d = fabs(x0[i] - x1[i]); if (!(d <= w)) w = d; }
      *maxdiff_out = w;
    }
  }
  h->refine_now = saved_ref; h->trsm_panel = saved_panel;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc) return rc;
  *us_per_solve = 1e3 * ms / reps;
  return GPS_OK;
}
extern "C" int gps_diag_trsm512(gps_handle_t h, int64_t m, int backward, int panel, int reps, double* us_per_solve,
                                double* maxdiff_out) {
  return diag_trsm512_impl(h, m, backward, panel, reps, us_per_solve, maxdiff_out, nullptr, 0);
}
// the one-launch form once more with phase stamps: stamps_out [min(m / 64, cap_blocks)][32] (trsm_panel.hip: TP_STAMP; m / 64 >= the
// number of CUs, so that the launch takes 64 rows per workgroup)
extern "C" int gps_diag_trsm512_stamps(gps_handle_t h, int64_t m, int backward, int reps, double* us_per_solve, long long* stamps_out,
                                       int64_t cap_blocks) {
  if (!stamps_out || cap_blocks <= 0) return GPS_ERR_ARG;
  return diag_trsm512_impl(h, m, backward, 1, reps, us_per_solve, nullptr, stamps_out, cap_blocks);
}

// mode 0 = product with the block inverse, 1 = refined (trsm_leaf.hip); resid_out = max |X T - B| / (|X| |T|)_max
// of the last launch's first 128 rows (T = L11^T or L11), checked on the host
extern "C" int gps_diag_trsm_leaf(gps_handle_t h, int64_t m, int mode, int upper, int reps, double* us_per_launch,
                                  double* resid_out) {
  if (!h || m <= 0 || m % GPS_TILE || reps <= 0 || !us_per_launch) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  const i64 T = GPS_TILE;
  std::vector<double> L((size_t)T * T, 0.0), B((size_t)T * T), X((size_t)T * T);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  for (i64 i = 0; i < T; ++i) for (i64 j = 0; j <= i; ++j) L[i * T + j] = (i == j) ? 1.0 + 0.5 * (rnd() + 0.5) : 0.4 * rnd();
  for (auto& v : B) v = rnd();
  GPS_HIP(h, h->dTmp.ensure((size_t)4 * T * T * 8));
  GPS_HIP(h, h->dB.ensure((size_t)m * T * 8 * 2));
  double* dL = h->dTmp.d(); double* dU = dL + T * T; double* dInv = dU + T * T; double* dInvT = dInv + T * T;
  double* dBm = h->dB.d(); double* dB0 = dBm + m * T;
  GPS_HIP(h, hipMemcpyAsync(dL, L.data(), (size_t)T * T * 8, hipMemcpyHostToDevice, h->stream));
  for (i64 q = 0; q < m / T; ++q) GPS_HIP(h, hipMemcpyAsync(dB0 + q * T * T, B.data(), (size_t)T * T * 8, hipMemcpyHostToDevice, h->stream));
  int rc = gps_launch_transpose(h, dL, T, T, T, dU, T);
  if (rc) return rc;
  rc = gps_launch_fill_info(h, (int*)h->dInfo.p, INT_MAX);
  if (rc) return rc;
  rc = gps_launch_potrf_base(h, dL, T, dInv, dInvT, (int*)h->dInfo.p, 0, /*factor*/ 0);
  if (rc) return rc;
  HipOps ops{h, dInv, dInvT, (int*)h->dInfo.p};
  const bool saved = h->refine_now;
  h->refine_now = (mode != 0);
  hipEvent_t e0, e1;
  GPS_HIP(h, hipEventCreate(&e0)); GPS_HIP(h, hipEventCreate(&e1));
  float ms = 0.f;
  for (int pass = 0; pass < 2 && !rc; ++pass) {            // pass 0 warms up
    GPS_HIP(h, hipEventRecord(e0, h->stream));
    for (int it = 0; it < reps && !rc; ++it) {
      if (it == 0 || it == reps - 1) GPS_HIP(h, hipMemcpyAsync(dBm, dB0, (size_t)m * T * 8, hipMemcpyDeviceToDevice, h->stream));
      rc = ops.trsm_base(0, upper, dBm, T, m, upper ? dU : dL, T);
    }
    GPS_HIP(h, hipEventRecord(e1, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    GPS_HIP(h, hipEventElapsedTime(&ms, e0, e1));
  }
  if (!rc && mode != 0 && getenv("GPS_LEAF_STAMPS")) {      // one more launch with phase stamps of workgroup 0
    long long* dS = (long long*)(dB0 + m * T) - 16;         // tail of the spare copy of B
    long long hs[8] = {0};
    GPS_HIP(h, hipMemsetAsync(dS, 0, sizeof(hs), h->stream));
    h->leaf_stamps = dS;
    rc = ops.trsm_base(0, upper, dBm, T, m, upper ? dU : dL, T);
    h->leaf_stamps = nullptr;
    GPS_HIP(h, hipMemcpyAsync(hs, dS, sizeof(hs), hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    fprintf(stderr, "leaf m=%lld phases (us): load+stage %.2f | product1 %.2f | stage2 %.2f | product2 %.2f | stage3 %.2f | product3 %.2f | store %.2f\n",
            (long long)m, (hs[1] - hs[0]) * 0.01, (hs[2] - hs[1]) * 0.01, (hs[3] - hs[2]) * 0.01, (hs[4] - hs[3]) * 0.01,
            (hs[5] - hs[4]) * 0.01, (hs[6] - hs[5]) * 0.01, (hs[7] - hs[6]) * 0.01);
  }
  h->refine_now = saved;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc) return rc;
  *us_per_launch = 1e3 * ms / reps;
  if (resid_out) {
    GPS_HIP(h, hipMemcpy(X.data(), dBm + (m - T) * T, (size_t)T * T * 8, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    for (i64 i = 0; i < T; ++i) for (i64 j = 0; j < T; ++j) {
      double s = 0.0, a = 0.0;
      for (i64 k = 0; k < T; ++k) {
        const double tkj = upper ? ((k >= j) ? L[k * T + j] : 0.0) : ((k <= j) ? L[j * T + k] : 0.0);   // T[k][j]
        s += X[i * T + k] * tkj; a += fabs(X[i * T + k] * tkj);
      }
      worst = fmax(worst, fabs(s - B[i * T + j])); scale = fmax(scale, a);
    }
    *resid_out = worst / scale;
  }
  return GPS_OK;
}

// ---- kernels.K ---------------------------------------------------------------------------------------
extern "C" int gps_kmat(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* X,
                        int64_t n, const double* X2, int64_t m, int64_t d_all, double diag_add,
                        double* K_out) {
  if (!h || !X || !K_out || n < 0 || d_all <= 0) return gps_fail(h, GPS_ERR_ARG, "gps_kmat: bad argument");
  GPS_HIP(h, hipSetDevice(h->device));
  const bool sym = (X2 == nullptr);
  if (sym) m = n;
  if (n == 0 || m == 0) return GPS_OK;
  const i64 prow = ((n + 63) / 64) * 64;
  const i64 pcol = sym ? prow : ((m + 63) / 64) * 64;
  GPS_HIP(h, h->dXnew.ensure((size_t)n * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dXnew.p, X, (size_t)n * d_all * 8, hipMemcpyHostToDevice, h->stream));
  const double* dX2 = nullptr;
  if (!sym) {
    GPS_HIP(h, h->dTmp3.ensure((size_t)m * d_all * 8));
    GPS_HIP(h, hipMemcpyAsync(h->dTmp3.p, X2, (size_t)m * d_all * 8, hipMemcpyHostToDevice, h->stream));
    dX2 = h->dTmp3.d();
  }
  GPS_HIP(h, h->dTmp.ensure((size_t)prow * pcol * 8));
  int rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n, dX2, m, d_all, diag_add, h->dTmp.d(), pcol,
                           prow, pcol, /*lower_only*/ 0, /*identity_pad*/ 0);
  if (rc) return rc;
  if (prow == n && pcol == m) {
    GPS_HIP(h, hipMemcpyAsync(K_out, h->dTmp.p, (size_t)n * m * 8, hipMemcpyDeviceToHost, h->stream));
  } else {
    GPS_HIP(h, h->dTmp2.ensure((size_t)n * m * 8));
    rc = gps_launch_extract(h, h->dTmp.d(), pcol, n, m, h->dTmp2.d(), m, 0);
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(K_out, h->dTmp2.p, (size_t)n * m * 8, hipMemcpyDeviceToHost, h->stream));
  }
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

// ---- tf.cholesky on a host matrix ----------------------------------------------------------------------
extern "C" int gps_potrf(gps_handle_t h, const double* A, int64_t n, double* L_out, int* info) {
  return with_la_retry(h, [&]() -> int {
  if (!h || !A || !L_out || n < 0) return gps_fail(h, GPS_ERR_ARG, "gps_potrf: bad argument");
  if (info) *info = 0;
  if (n == 0) return GPS_OK;
  GPS_HIP(h, hipSetDevice(h->device));
  h->refine_now = (h->leaf_refine != 0);
  const i64 np = gps_pad(n);
  GPS_HIP(h, h->dTmp2.ensure((size_t)n * n * 8));
  GPS_HIP(h, h->dTmp.ensure((size_t)np * np * 8));
  GPS_HIP(h, h->dTmp3.ensure((size_t)(np / GPS_TILE) * GPS_TILE * GPS_TILE * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, A, (size_t)n * n * 8, hipMemcpyHostToDevice, h->stream));
  int rc = gps_launch_pad_copy(h, h->dTmp2.d(), n, n, n, h->dTmp.d(), np, np, np, 1, 0.0);
  if (rc) return rc;
  int* d_info = (int*)h->dInfo.p;
  rc = gps_launch_fill_info(h, d_info, INT_MAX);
  if (rc) return rc;
  HipOps ops{h, h->dTmp3.d(), nullptr, d_info};
  Blocked<HipOps> bl(ops);
  rc = bl.potrf_rec(h->dTmp.d(), np, np, 0, 0);
  if (rc) return rc;
  rc = gps_launch_extract(h, h->dTmp.d(), np, n, n, h->dTmp2.d(), n, 1);
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(L_out, h->dTmp2.p, (size_t)n * n * 8, hipMemcpyDeviceToHost, h->stream));
  return read_info(h, d_info, info);
  });
}

// ---- tf.matrix_triangular_solve on host matrices ---------------------------------------------------------
extern "C" int gps_trsm_lower(gps_handle_t h, const double* L, int64_t n, double* B, int64_t nrhs,
                              int trans) {
  return with_la_retry(h, [&]() -> int {
  if (!h || !L || !B || n < 0 || nrhs < 0) return gps_fail(h, GPS_ERR_ARG, "gps_trsm_lower: bad argument");
  if (n == 0 || nrhs == 0) return GPS_OK;
  GPS_HIP(h, hipSetDevice(h->device));
  h->refine_now = (h->leaf_refine != 0);
  const i64 np = gps_pad(n), mp = gps_pad(nrhs);
  const size_t blk_bytes = (size_t)(np / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  // dTmp: L padded (and, for trans, U = L^T) ; dTmp2: staging ; dTmp3: inverses ; dB: B^T padded
  GPS_HIP(h, h->dTmp2.ensure((size_t)(n * n > n * nrhs ? n * n : n * nrhs) * 8));
  GPS_HIP(h, h->dTmp.ensure((size_t)np * np * 8 * (trans ? 2 : 1)));
  GPS_HIP(h, h->dTmp3.ensure(2 * blk_bytes));
  GPS_HIP(h, h->dB.ensure((size_t)mp * np * 8));
  double* dL = h->dTmp.d();
  GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, L, (size_t)n * n * 8, hipMemcpyHostToDevice, h->stream));
  int rc = gps_launch_pad_copy(h, h->dTmp2.d(), n, n, n, dL, np, np, np, 1, 0.0);
  if (rc) return rc;
  int* d_info = (int*)h->dInfo.p;
  HipOps ops{h, h->dTmp3.d(), h->dTmp3.d() + blk_bytes / 8, d_info};
  ops.factor = 0;
  for (i64 b = 0; b < np / GPS_TILE; ++b) {
    rc = ops.potrf_base(dL + b * GPS_TILE * np + b * GPS_TILE, np, b, b * GPS_TILE);
    if (rc) return rc;
  }
  // B [n, nrhs] -> Bt [mp, np]
  GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, B, (size_t)n * nrhs * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemsetAsync(h->dB.p, 0, (size_t)mp * np * 8, h->stream));
  rc = gps_launch_transpose(h, h->dTmp2.d(), nrhs, n, nrhs, h->dB.d(), np);
  if (rc) return rc;
  Blocked<HipOps> bl(ops);
  if (!trans) {
    rc = bl.trsm_rec(dL, np, np, 0, h->dB.d(), np, mp);        // X^T L^T = B^T  <=>  L X = B
  } else {
    double* dU = dL + np * np;
    rc = gps_launch_transpose(h, dL, np, np, np, dU, np);
    if (rc) return rc;
    rc = bl.trsm_rn_rec(dU, np, np, 0, h->dB.d(), np, mp);     // X^T L = B^T    <=>  L^T X = B
  }
  if (rc) return rc;
  rc = gps_launch_transpose(h, h->dB.d(), np, nrhs, n, h->dTmp2.d(), nrhs);
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(B, h->dTmp2.p, (size_t)n * nrhs * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
  });
}

// ---- GPR ------------------------------------------------------------------------------------------------
extern "C" int gps_gpr_set_data(gps_handle_t h, const double* X, int64_t n, int64_t d_all) {
  if (!h || !X || n <= 0 || d_all <= 0) return gps_fail(h, GPS_ERR_ARG, "gps_gpr_set_data: bad argument");
  GPS_HIP(h, hipSetDevice(h->device));
  h->have_factor = false; h->dist_have_part_factor = false;
  h->n = n; h->d_all = d_all; h->npad = gps_pad(n);
  GPS_HIP(h, h->dX.ensure((size_t)n * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dX.p, X, (size_t)n * d_all * 8, hipMemcpyHostToDevice, h->stream));
  // (K itself is allocated by whoever factors it: gpr_factor the whole [N, N], a rank of the block-column path only its
  // own block columns -- 8 N^2 / P bytes)
  GPS_HIP(h, h->dLinv.ensure(2 * (size_t)(h->npad / GPS_TILE) * GPS_TILE * GPS_TILE * 8));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

// pinned host landing area of the small read-backs (a copy into pageable memory goes through a staging buffer of the
// runtime and blocks the host for tens of microseconds)
static int ensure_hres(gps_handle_t h) {
  if (!h->hRes) GPS_HIP(h, hipHostMalloc(&h->hRes, GPS_HRES_BYTES, hipHostMallocDefault));
  return GPS_OK;
}

// the transposed block inverses of the resident GPR factor, if the factorisation left them out (the one-launch small path)
static int gpr_ensure_linvT(gps_handle_t h) {
  if (!h->gpr_linvT_stale) return GPS_OK;
  const i64 nb = h->npad / GPS_TILE;
  int rc = gps_launch_transpose_blocks(h, h->dLinv.d(), h->dLinv.d() + nb * GPS_TILE * GPS_TILE, nb);
  if (rc == GPS_OK) h->gpr_linvT_stale = false;
  return rc;
}

// K + noise I -> L, alpha.  Records ev[0..3].
static int gpr_factor(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double noise_var,
                      const double* resid, i64 r, int* info) {
  if (h->n <= 0) return gps_fail(h, GPS_ERR_STATE, "gps_gpr_set_data has not been called");
  if (r < 0 || (r > 0 && !resid)) return gps_fail(h, GPS_ERR_ARG, "resid missing");
  const i64 n = h->n, np = h->npad;
  h->have_factor = false; h->dist_have_part_factor = false;
  {
    // leaves refined or not: from the bound cond(K + noise I) <= (N Kdiag + noise) / noise (gps_gpr_needs_refine)
    double kd = 0.0;
    int rck = gps_launch_kdiag(h, prog, n_nodes, &kd);
    if (rck) return rck;
    h->refine_now = gps_gpr_needs_refine(h, noise_var, kd, h->n);
    h->factor_refine = h->refine_now;
  }
  GPS_HIP(h, hipEventRecord(h->ev[0], h->stream));
  // Augmented rows (blocked.hpp::potrf_rec; option "gpr_aug_rows"): (Y - m)^T stored as 128 more rows under K rides
  // through the factorisation, which leaves alpha^T = (L^-1 (Y - m))^T there (densities.py:82) -- no forward-substitution
  // pass (~4 N / 128 launch-latency-bound kernels).  Same-process A/B on MI355X: N = 2048 / 4096 / 8192 / 12288:
  // -9 / -10 / -4.4 / -3.8 %; from N = 16384 on it loses (+0.9 %, N = 32768 +1.6 %): the extra tile row breaks the
  // power-of-two tile counts of the big launches, whose whole rounds of 512 workgroup slots matter more than the 3 ms
  // of trsv.  Against the one-launch wavefront substitution (trsv_wave.hip: 0.28 ms at N = 8192, 1.2 ms at 32768, where the
  // recursive one took 0.62 / 3.2 ms) the augmented rows still win up to N = 4096 (-3 %), lose from 8192 on (+1.5 %) and
  // tie at 12288.  Hence automatic (-1): on below 6200 points.  (The block-column multi-GPU path always uses it.)
  // (The recursive substitution issued block by block behind the factorisation on a stream of its own was measured far worse
  // still -- round 2, docs/LAB_NOTES.md -- and is gone.)
  const bool aug = r > 0 && r <= GPS_TILE && (h->gpr_aug_rows > 0 || (h->gpr_aug_rows < 0 && np < 6200));
  GPS_HIP(h, h->dK.ensure((size_t)(np + GPS_TILE) * np * 8));
  double* const dAug = h->dK.d() + np * np;
  // Small problems (the reference's own size: examples/gpr.py, N ~ 455): the whole factorisation, alpha and the two
  // reductions of the likelihood as ONE cooperative launch (small_n.hip) -- three launches per evaluation with the two of the
  // kernel-matrix build, no memset, no transposition, one 32-byte read-back.  Not for refined leaves (ill-conditioned K).
  h->small_valid = false;
  // (up to small_n_max padded points; above seven blocks the launch draws all its work from a queue)
  bool small = h->small_n > 0 && aug && np <= h->small_n_max && np <= 4096 && r <= 16 && !h->refine_now &&
               h->prop.multiProcessorCount >= 160;
  if (small && h->small_cooldown > 0) { --h->small_cooldown; small = false; }      // (back-off after give-ups in a row: small_gave_up)
  // residual, transposed to [r][np] and zero padded
  if (r > 0) {
    GPS_HIP(h, h->dAlpha.ensure((size_t)r * np * 8));
    GPS_HIP(h, h->dTmp2.ensure((size_t)n * r * 8));
    // (through a pinned slot when small: a copy from pageable memory blocks the host for its staging)
    if ((size_t)n * r * 8 <= (size_t)h->resid_ring_max) GPS_HIP(h, h->ring.upload(h->dTmp2.p, resid, (size_t)n * r * 8, h->stream));
    else GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, resid, (size_t)n * r * 8, hipMemcpyHostToDevice, h->stream));
    if (!small) {
      double* dst = aug ? dAug : h->dAlpha.d();
      GPS_HIP(h, hipMemsetAsync(dst, 0, (size_t)(aug ? GPS_TILE : r) * np * 8, h->stream));
      int rc0 = gps_launch_transpose(h, h->dTmp2.d(), r, n, r, dst, np);
      if (rc0) return rc0;
    }
  }
  // (small path, one stationary primitive: the cooperative launch generates K itself -- no kernel-matrix launches at all)
  SmallKgen kg;
  const int op0 = n_nodes == 1 ? prog[0].op : -1;
  if (small && h->small_n >= 1 && (op0 == GPS_K_RBF || op0 == GPS_K_MATERN12 || op0 == GPS_K_MATERN32 || op0 == GPS_K_MATERN52 ||
      op0 == GPS_K_EXPONENTIAL) && prog[0].n_dims >= 1 && prog[0].n_dims <= 16 && prog[0].variance > 0.0) {
    kg.on = 1; kg.op = op0; kg.X = h->dX.d(); kg.d_all = (int)h->d_all; kg.nd = prog[0].n_dims; kg.variance = prog[0].variance; kg.noise = noise_var;
    for (int d = 0; d < 16; ++d) { kg.dims[d] = 0; kg.inv_ls[d] = 0.0; }
    for (int d = 0; d < kg.nd; ++d) {
      kg.dims[d] = prog[0].active_dims[d]; kg.inv_ls[d] = 1.0 / prog[0].lengthscales[d];
      if (kg.dims[d] < 0 || kg.dims[d] >= kg.d_all || !(prog[0].lengthscales[d] > 0.0)) kg.on = 0;       // (left to the ordinary build and its error text)
    }
  }
  int rc = GPS_OK;
  if (!kg.on) {
    rc = gps_launch_kmat(h, prog, n_nodes, h->dX.d(), n, nullptr, n, h->d_all, noise_var, h->dK.d(), np,
                         np, np, /*lower_only*/ 1, /*identity_pad*/ 1);
    if (rc) return rc;
  }
  GPS_HIP(h, hipEventRecord(h->ev[1], h->stream));
  int* d_info = (int*)h->dInfo.p;
  if (small) {
    double* d_res = h->dScal.d() + 256;
    if (h->small_defer) {                 // (gps_gpr_lml_grad: with everything else it reads back, in one buffer)
      GPS_HIP(h, h->dSmallOut.ensure((size_t)(5 + GPS_GRAD_SUMS + n * r) * 8));
      d_res = h->dSmallOut.d();
    }
    double* linv = h->dLinv.d();
    // (the transposed block inverses are not on the path of the likelihood: whoever needs them afterwards -- the gradient,
    // a prediction from this factor -- has them produced by one batched launch then: gpr_ensure_linvT)
    if (h->plain_linv == linv) h->plain_linv = nullptr;        // (these block inverses are produced anew, not through HipOps::potrf_base)
    rc = gps_launch_small_factor(h, h->dK.d(), np, linv, nullptr, h->dTmp2.d(), n, r, d_info, d_res, h->dAlpha.d(), np, r, &kg);
    h->gpr_linvT_stale = true;
    if (rc == GPS_OK) {
      GPS_HIP(h, hipEventRecord(h->ev[2], h->stream));
      h->r = r;
      if (h->small_defer) { h->small_pending = true; return GPS_OK; }      // (gps_gpr_lml_grad reads the results back itself, later)
      int rch = ensure_hres(h);
      if (rch) return rch;
      double* res = (double*)h->hRes;
      res[3] = 1.0;
      GPS_HIP(h, hipMemcpyAsync(res, d_res, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      GPS_HIP(h, hipStreamSynchronize(h->stream));
      if (res[3] == 0.0) {
        const int v = (int)res[2];
        if (info) *info = (v == INT_MAX) ? 0 : v;
        h->r = r;
        h->small_valid = true; h->small_slog = res[0]; h->small_ssq = res[1];
        h->have_factor = (info == nullptr) || (*info == 0);
        h->small_consec = 0;
        return GPS_OK;
      }
      // a bounded wait of the launch ran out (never seen; e.g. several such launches of one process interleaved on the GPU
      // so that none was fully resident): counters back to zero, this evaluation again launch by launch
      small_gave_up(h);
      rc = gps_small_factor_reset(h);
      if (rc) return rc;
      const int saved = h->small_n;
      h->small_n = 0;
      rc = gpr_factor(h, prog, n_nodes, noise_var, resid, r, info);
      h->small_n = saved;
      return rc;
    }
    if (rc != GPS_ERR_UNSUPPORTED) return rc;
    // (not a shape for that path after all: K, if the launch was to generate it, and the residual still have to go where the
    // launch-by-launch path expects them)
    if (kg.on) {
      rc = gps_launch_kmat(h, prog, n_nodes, h->dX.d(), n, nullptr, n, h->d_all, noise_var, h->dK.d(), np, np, np, 1, 1);
      if (rc) return rc;
    }
    if (r > 0) {
      GPS_HIP(h, hipMemsetAsync(dAug, 0, (size_t)GPS_TILE * np * 8, h->stream));
      rc = gps_launch_transpose(h, h->dTmp2.d(), r, n, r, dAug, np);
      if (rc) return rc;
    }
  }
  h->gpr_linvT_stale = false;
  rc = gps_launch_fill_info(h, d_info, INT_MAX);
  if (rc) return rc;
  HipOps ops{h, h->dLinv.d(), h->dLinv.d() + (np / GPS_TILE) * GPS_TILE * GPS_TILE, d_info};
  {
    // the factorisation itself only needs the block inverses; their transposes (for the vector solves) are produced
    // by batched launches off the critical path rather than by 128 KB of extra stores in every potrf_base.
    HipOps fops = ops;
    fops.store_T = false;
    Blocked<HipOps> fbl(fops);
    rc = fbl.potrf_rec(h->dK.d(), np, np, 0, 0, nullptr, aug ? (i64)GPS_TILE : 0);
    if (rc) return rc;
    rc = gps_launch_transpose_blocks(h, ops.linv, ops.linvT, np / GPS_TILE);
    if (rc) return rc;
    if (h->refine_now) { rc = classify_blocks(h, ops, h->dK.d(), np, np); if (rc) return rc; }      // (low noise: the predictions' solves)
  }
  Blocked<HipOps> bl(ops);
  GPS_HIP(h, hipEventRecord(h->ev[2], h->stream));
  if (aug) {
    GPS_HIP(h, hipMemcpyAsync(h->dAlpha.p, dAug, (size_t)r * np * 8, hipMemcpyDeviceToDevice, h->stream));
  } else if (r > 0) {
    rc = trsv_forward(h, ops, h->dK.d(), np, np, h->dAlpha.d(), np, r);
    if (rc) return rc;
  }
  h->r = r;
  rc = read_info(h, d_info, info);
  if (rc) return rc;
  h->have_factor = (info == nullptr) || (*info == 0);
  return GPS_OK;
}

extern "C" int gps_gpr_lml(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double noise_var,
                           const double* resid, int64_t r, double* lml, int* info) {
  return with_la_retry(h, [&]() -> int {
  if (!h || !lml) return gps_fail(h, GPS_ERR_ARG, "gps_gpr_lml: bad argument");
  GPS_HIP(h, hipSetDevice(h->device));
  int linfo = 0;
  int rc = gpr_factor(h, prog, n_nodes, noise_var, resid, r, &linfo);
  if (info) *info = linfo;
  if (rc) return rc;
  return gpr_lml_finish(h, r, lml);
  });
}

// the likelihood from the resident factor and alpha (densities.py:92-94); stage times of the evaluation
static int gpr_lml_finish(gps_handle_t h, i64 r, double* lml) {
  int rc;
  const i64 n = h->n, np = h->npad;
  double slog = 0.0, ssq = 0.0;
  if (h->small_valid) {
    // (the one-launch factorisation of a small problem has reduced both sums itself and they are on the host already)
    slog = h->small_slog; ssq = h->small_ssq;
    h->ev3_is_ev2 = true;
  } else {
    h->ev3_is_ev2 = false;
    double* part = h->dScal.d();
    rc = gps_launch_lml_reduce(h, h->dK.d(), np, n, h->dAlpha.d(), np, r, part);
    if (rc) return rc;
    GPS_HIP(h, hipEventRecord(h->ev[3], h->stream));
    double hp[2 * 64];
    GPS_HIP(h, hipMemcpyAsync(hp, part, sizeof(hp), hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < 64; ++b) { slog += hp[2 * b]; ssq += hp[2 * b + 1]; }
  }
  // densities.py:92-94
  *lml = -0.5 * (double)n * (double)r * log(2.0 * M_PI) - (double)r * slog - 0.5 * ssq;
  stage_time(h, 0, 1, &h->stage_ms[0]);
  stage_time(h, 1, 2, &h->stage_ms[1]);
  if (h->ev3_is_ev2) h->stage_ms[2] = 0.0; else stage_time(h, 2, 3, &h->stage_ms[2]);
  h->stage_ms[3] = 0.0;
  stage_time(h, 0, h->ev3_is_ev2 ? 2 : 3, &h->stage_ms[4]);
  return GPS_OK;
}

// What follows the (not yet read back) one-launch factorisation of a small problem in gps_gpr_lml_grad.  *done = false: a
// bounded wait gave up, nothing of the outputs is valid.
static int gpr_small_grad_tail(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, i64 r, double* lml, double* grad_slots,
                               double* grad_noise, double* kinv_resid, int* info, bool* done) {
  const i64 n = h->n, np = h->npad;
  *done = false;
  int rc = ensure_hres(h);
  if (rc) return rc;
  GPS_HIP(h, hipEventRecord(h->ev[5], h->stream));
  GPS_HIP(h, h->dA.ensure((size_t)r * np * 8));
  GPS_HIP(h, h->dY.ensure((size_t)np * np * 8));
  GPS_HIP(h, h->dKinv.ensure((size_t)np * np * 8));
  // dSmallOut: [0..3] the factorisation's results (already on their way), [4] the inverse launch's abort word, [5 ..] the
  // gradient sums, then K^-1 resid as [n][r] -- one copy brings all of it back
  double* d_res = h->dSmallOut.d();
  double* d_kr = d_res + 5 + GPS_GRAD_SUMS;
  rc = gps_launch_small_inverse(h, h->dK.d(), np, h->dLinv.d(), h->dAlpha.d(), r, h->dY.d(), h->dKinv.d(), h->dA.d(), d_res + 4,
                                kinv_resid ? d_kr : nullptr, n);
  if (rc) return rc;                   // (the factorisation took this shape: so does the inverse)
  // (launching the gradient kernel's features in front of the factorisation instead -- gps_grad_prepare -- was measured: no gain)
  GradPost post;
  rc = gps_grad_enqueue(h, prog, n_nodes, h->dX.d(), n, h->d_all, np, h->dKinv.d(), np, h->dA.d(), np, r, d_res + 5, &post);
  if (rc) return rc;
  double* res = (double*)h->hRes;
  double* kr = res + 5 + GPS_GRAD_SUMS;
  res[3] = 1.0; res[4] = 1.0;
  GPS_HIP(h, hipMemcpyAsync(res, d_res, (size_t)(5 + GPS_GRAD_SUMS + (kinv_resid ? n * r : 0)) * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipEventRecord(h->ev[6], h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  if (res[3] != 0.0 || res[4] != 0.0) return GPS_OK;
  *done = true;
  h->small_consec = 0;
  const int v = (int)res[2];
  *info = (v == INT_MAX) ? 0 : v;
  h->have_factor = (*info == 0);
  h->small_valid = true; h->small_slog = res[0]; h->small_ssq = res[1];
  if (*info) return GPS_OK;            // not positive definite: outputs undefined
  rc = gpr_lml_finish(h, r, lml);
  if (rc) return rc;
  stage_time(h, 5, 6, &h->stage_ms[3]);
  gps_grad_finish(post, res + 5, grad_slots, grad_noise);
  if (kinv_resid) memcpy(kinv_resid, kr, (size_t)n * r * 8);
  return GPS_OK;
}

// LML and its gradient: d/d(kernel parameter slots), d/d(noise variance), d/d(resid) = -K_y^-1 resid ... see header
extern "C" int gps_gpr_lml_grad(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double noise_var,
                                const double* resid, int64_t r, double* lml, double* grad_slots,
                                int n_slots_cap, int* n_slots_out, double* grad_noise, double* kinv_resid,
                                int* info) {
  if (!h || !lml || !grad_slots || !grad_noise || r <= 0)
    return gps_fail(h, GPS_ERR_ARG, "gps_gpr_lml_grad: bad argument");
  // (a hand-over of the look-ahead or of the backward wavefront substitution that gives up invalidates this call, not the
  // next one: the body runs again, once, through the recursive forms -- with_la_retry)
  return with_la_retry(h, [&]() -> int {
  GPS_HIP(h, hipSetDevice(h->device));
  int ns = 0;
  int rc = gps_grad_slots(h, prog, n_nodes, &ns);
  if (rc) return rc;
  if (n_slots_out) *n_slots_out = ns;
  if (ns > n_slots_cap) return gps_fail(h, GPS_ERR_ARG, "gps_gpr_lml_grad: grad_slots too small");
  int linfo = 0;
  // Small problems (the reference's own size: examples/gpr.py): factorisation, inverse and gradient sums are enqueued
  // back to back -- six launches -- and everything the host needs comes back in one pinned copy behind ONE synchronisation.
  h->small_defer = h->small_n > 0 && r <= 16 && h->npad <= 2048 && h->npad <= h->small_n_max && gps_grad_is_simple(prog, n_nodes) && (!kinv_resid || (size_t)(5 + GPS_GRAD_SUMS + h->n * r) * 8 <= GPS_HRES_BYTES);
  h->small_pending = false;
  rc = gpr_factor(h, prog, n_nodes, noise_var, resid, r, &linfo);
  h->small_defer = false;
  if (rc) return rc;
  if (h->small_pending) {
    h->small_pending = false;
    bool done = false;
    rc = gpr_small_grad_tail(h, prog, n_nodes, r, lml, grad_slots, grad_noise, kinv_resid, &linfo, &done);
    if (rc) return rc;
    if (done) { if (info) *info = linfo; return GPS_OK; }
    // a bounded wait of one of the two cooperative launches ran out (never seen): this evaluation again, launch by launch
    small_gave_up(h);
    rc = gps_small_factor_reset(h);
    if (rc) return rc;
    const int saved = h->small_n;
    h->small_n = 0;
    rc = gpr_factor(h, prog, n_nodes, noise_var, resid, r, &linfo);
    h->small_n = saved;
    if (rc) return rc;
  }
  if (info) *info = linfo;
  if (linfo) return GPS_OK;
  rc = gpr_lml_finish(h, r, lml);
  if (rc) return rc;
  const i64 n = h->n, np = h->npad;
  GPS_HIP(h, hipEventRecord(h->ev[5], h->stream));
  GPS_HIP(h, h->dA.ensure((size_t)r * np * 8));
  GPS_HIP(h, h->dY.ensure((size_t)np * np * 8));
  GPS_HIP(h, h->dKinv.ensure((size_t)np * np * 8));
  // K_y^-1 (lower triangle) and A = K_y^-1 resid.  After the one-launch factorisation of a small problem: one more
  // cooperative launch (small_n.hip) instead of ~25 (the two recursions and the backward substitution below)
  bool small_inv = h->small_valid && h->small_n > 0;
  double* d_res1 = h->dScal.d() + 260;
  if (small_inv) {
    rc = gps_launch_small_inverse(h, h->dK.d(), np, h->dLinv.d(), h->dAlpha.d(), r, h->dY.d(), h->dKinv.d(), h->dA.d(), d_res1);
    if (rc == GPS_ERR_UNSUPPORTED) small_inv = false;
    else if (rc) return rc;
  }
  for (int pass = 0; pass < 2; ++pass) {
    if (!small_inv) {
      rc = gpr_ensure_linvT(h);
      if (rc) return rc;
      double* linv = h->dLinv.d();
      HipOps ops{h, linv, linv + (np / GPS_TILE) * GPS_TILE * GPS_TILE, (int*)h->dInfo.p};
      Blocked<HipOps> bl(ops);
      // A = K_y^-1 resid = L^-T (L^-1 resid)
      GPS_HIP(h, hipMemcpyAsync(h->dA.p, h->dAlpha.p, (size_t)r * np * 8, hipMemcpyDeviceToDevice, h->stream));
      rc = trsv_backward(h, ops, h->dK.d(), np, np, h->dA.d(), np, r);
      if (rc) return rc;
      // K_y^-1 = L^-T L^-1
      rc = bl.inv_t_rec(h->dK.d(), np, np, 0, h->dY.d(), np);
      if (rc) return rc;
      rc = bl.lauum_rec(h->dY.d(), np, np, h->dKinv.d(), np);
      if (rc) return rc;
    }
    rc = gps_launch_grad(h, prog, n_nodes, h->dX.d(), n, h->d_all, np, h->dKinv.d(), np, h->dA.d(), np, r, grad_slots,
                         grad_noise);
    if (rc) return rc;
    if (kinv_resid) {
      GPS_HIP(h, h->dTmp2.ensure((size_t)n * r * 8));
      rc = gps_launch_transpose(h, h->dA.d(), np, r, n, h->dTmp2.d(), r);
      if (rc) return rc;
      GPS_HIP(h, hipMemcpyAsync(kinv_resid, h->dTmp2.p, (size_t)n * r * 8, hipMemcpyDeviceToHost, h->stream));
    }
    if (!small_inv) break;
    // (the launch-by-launch path touches neither the info word nor a hand-over: the only thing to read back is whether a
    // bounded wait of the cooperative launch ran out -- never seen -- and then the same again launch by launch)
    double ab = 1.0;
    GPS_HIP(h, hipMemcpyAsync(&ab, d_res1, 8, hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    if (ab == 0.0) {
      GPS_HIP(h, hipEventRecord(h->ev[6], h->stream));
      GPS_HIP(h, hipEventSynchronize(h->ev[6]));
      stage_time(h, 5, 6, &h->stage_ms[3]);
      return GPS_OK;
    }
    small_gave_up(h);
    rc = gps_small_factor_reset(h);
    if (rc) return rc;
    small_inv = false;
  }
  GPS_HIP(h, hipEventRecord(h->ev[6], h->stream));
  // synchronises, and surfaces a backward wavefront substitution that gave up (its result would poison dA and every
  // gradient slot) as GPS_ERR_STATE for the retry above instead of leaving the counter for the next entry point
  rc = read_info(h, (int*)h->dInfo.p, nullptr);
  if (rc) return rc;
  stage_time(h, 5, 6, &h->stage_ms[3]);
  return GPS_OK;
  });
}

extern "C" int gps_gpr_predict(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                               double noise_var, const double* resid, int64_t r, const double* Xnew,
                               int64_t n_new, int full_cov, int refactor, double* mean_out,
                               double* var_out, int* info) {
  return with_la_retry(h, [&]() -> int {
  if (!h || !Xnew || n_new <= 0 || !var_out || (r > 0 && !mean_out))
    return gps_fail(h, GPS_ERR_ARG, "gps_gpr_predict: bad argument");
  GPS_HIP(h, hipSetDevice(h->device));
  if (info) *info = 0;
  int rc;
  if (refactor) {
    int linfo = 0;
    rc = gpr_factor(h, prog, n_nodes, noise_var, resid, r, &linfo);
    if (info) *info = linfo;
    if (rc) return rc;
    if (linfo) return GPS_OK;             // not positive definite: outputs undefined
    rc = gpr_ensure_linvT(h);
    if (rc) return rc;
  } else {
    if (!h->have_factor) return gps_fail(h, GPS_ERR_STATE, "no resident factor: call gps_gpr_lml first or pass refactor=1");
    if (r != h->r) return gps_fail(h, GPS_ERR_STATE, "resident alpha has a different number of outputs");
    h->refine_now = h->factor_refine;
    rc = gpr_ensure_linvT(h);
    if (rc) return rc;
    GPS_HIP(h, hipEventRecord(h->ev[0], h->stream));
    GPS_HIP(h, hipEventRecord(h->ev[1], h->stream));
    GPS_HIP(h, hipEventRecord(h->ev[2], h->stream));
  }
  const i64 n = h->n, np = h->npad, d = h->d_all;
  const i64 nsp = gps_pad(n_new);
  GPS_HIP(h, hipEventRecord(h->ev[3], h->stream));
  GPS_HIP(h, h->dXnew.ensure((size_t)n_new * d * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dXnew.p, Xnew, (size_t)n_new * d * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dB.ensure((size_t)nsp * np * 8));
  // Kx^T = K(Xnew, X)  [nsp, np]                                 models/gpr.py:119
  rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n_new, h->dX.d(), n, d, 0.0, h->dB.d(), np, nsp, np, 0, 0);
  if (rc) return rc;
  // A^T = Kx^T L^-T                                              models/gpr.py:122
  HipOps ops{h, h->dLinv.d(), h->dLinv.d() + (np / GPS_TILE) * GPS_TILE * GPS_TILE, (int*)h->dInfo.p};
  Blocked<HipOps> bl(ops);
  rc = bl.trsm_rec(h->dK.d(), np, np, 0, h->dB.d(), np, nsp);
  if (rc) return rc;
  // fmean = A^T V ; sumsq = colsum(A*A)                          models/gpr.py:124,130
  GPS_HIP(h, h->dMean.ensure((size_t)(n_new * (r > 0 ? r : 1) + n_new) * 8));
  double* dmean = h->dMean.d();
  double* dss = dmean + n_new * (r > 0 ? r : 1);
  rc = gps_launch_rowdot(h, h->dB.d(), np, n_new, np, h->dAlpha.d(), np, r, dmean, dss);
  if (rc) return rc;
  double kd = 0.0;
  rc = gps_launch_kdiag(h, prog, n_nodes, &kd);
  if (rc) return rc;
  if (!full_cov) {
    GPS_HIP(h, h->dVar.ensure((size_t)n_new * 8));
    rc = gps_launch_var_finish(h, h->dVar.d(), nullptr, kd, dss, n_new);
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(var_out, h->dVar.p, (size_t)n_new * 8, hipMemcpyDeviceToHost, h->stream));
  } else {
    // K(Xnew) - A^T A                                             models/gpr.py:126
    GPS_HIP(h, h->dVar.ensure((size_t)nsp * nsp * 8));
    rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n_new, nullptr, n_new, d, 0.0, h->dVar.d(), nsp, nsp,
                         nsp, 0, 0);
    if (rc) return rc;
    rc = gps_launch_gemm_nt(h, 0, 0, nsp, nsp, np, h->dB.d(), np, h->dB.d(), np, h->dVar.d(), nsp);
    if (rc) return rc;
    GPS_HIP(h, h->dTmp2.ensure((size_t)n_new * n_new * 8));
    rc = gps_launch_extract(h, h->dVar.d(), nsp, n_new, n_new, h->dTmp2.d(), n_new, 0);
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(var_out, h->dTmp2.p, (size_t)n_new * n_new * 8, hipMemcpyDeviceToHost, h->stream));
  }
  if (r > 0)
    GPS_HIP(h, hipMemcpyAsync(mean_out, dmean, (size_t)n_new * r * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipEventRecord(h->ev[4], h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  stage_time(h, 0, 1, &h->stage_ms[0]);
  stage_time(h, 1, 2, &h->stage_ms[1]);
  stage_time(h, 2, 3, &h->stage_ms[2]);
  stage_time(h, 3, 4, &h->stage_ms[3]);
  stage_time(h, 0, 4, &h->stage_ms[4]);
  return GPS_OK;
  });
}

// ---- conditionals -------------------------------------------------------------------------------------
// Shared tail of conditional / base_conditional.  On entry:
//   Kmm  [mp, mp]  device, padded with identity (jitter already added), lower triangle valid
//   Bt   [nsp, mp] device = Kmn^T zero padded
//   knn_const / dKnnDiag / dKnnFull describe Knn.
struct CondIn {
  i64 m, mp, n_new, nsp, k;
  double* Kmm; double* Bt; double* linv; double* linvT;
  const double* dKnnDiag; double knn_const; double* dKnnFull /* [nsp,nsp], overwritten */;
};

// SVGP mode of conditional_tail (gps_svgp_elbo): instead of returning fmean / fvar, reduce them on the device to the
// Gaussian variational expectations (likelihoods.py:186-188) and evaluate KL[q(u) || p(u)] (kullback_leiblers.py:26-105)
// from the SAME factor Lm = chol(Kuu + jitter I) the conditional has just built (the reference factors it twice:
// conditionals.py:84 and kullback_leiblers.py:51).
struct SvgpAcc {
  const double* yres;      // host [n, k] = Y - mean_function(X)
  double noise_var;
  double sq_sum = 0.0;     // sum_{i,q} (yres - fmean)^2 + fvar
  double kl = 0.0;
};

// tr(Sigma_p^-1 Sigma_q) pieces of the KL for p = N(0, L L^T)                      kullback_leiblers.py:83-94
//   diag q_sqrt [m, k]:   sum_j diag(K^-1)_j sum_q q_sqrt[j][q]^2 ,  diag(K^-1)_j = sum_i (L^-1)[i][j]^2 : L^-T by one
//                         triangular solve against the identity (dWork [mp, mp]), row sums of squares, no K^-1 formed
//   full  q_sqrt [m,m,k]: sum (L^-1 L_q)^2 per latent: dLqT [mp, mp] holds L_q^T on entry (overwritten)
static int kl_diag_kinv(gps_handle_t h, Blocked<HipOps>& bl, const double* L, i64 mp, i64 m, double* dWork,
                        std::vector<double>& kinv_diag) {
  int rc = gps_launch_pad_copy(h, dWork, mp, 0, 0, dWork, mp, mp, mp, /*identity*/ 1, 0.0);
  if (rc) return rc;
  rc = bl.trsm_rec(L, mp, mp, 0, dWork, mp, mp);                  // X L^T = I  ->  X = L^-T (upper triangular)
  if (rc) return rc;
  GPS_HIP(h, h->dTmp3.ensure((size_t)mp * 8));
  rc = gps_launch_rowdot(h, dWork, mp, m, mp, nullptr, mp, 0, nullptr, h->dTmp3.d());
  if (rc) return rc;
  kinv_diag.resize((size_t)m);
  GPS_HIP(h, hipMemcpyAsync(kinv_diag.data(), h->dTmp3.p, (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}
static int kl_full_one(gps_handle_t h, Blocked<HipOps>& bl, const double* L, i64 mp, i64 m, double* dLqT, double* out) {
  int rc = bl.trsm_rec(L, mp, mp, 0, dLqT, mp, mp);               // X L^T = L_q^T  ->  X = (L^-1 L_q)^T
  if (rc) return rc;
  GPS_HIP(h, h->dScal.ensure(4096 + (size_t)mp * 8));
  double* dss = h->dScal.d() + 512;
  rc = gps_launch_rowdot(h, dLqT, mp, m, mp, nullptr, mp, 0, nullptr, dss);
  if (rc) return rc;
  std::vector<double> ss((size_t)m);
  GPS_HIP(h, hipMemcpyAsync(ss.data(), dss, (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  double t = 0.0;
  for (i64 i = 0; i < m; ++i) t += ss[i];
  *out = t;
  return GPS_OK;
}
// everything of the KL that only needs the host copies of q_mu / q_sqrt                 kullback_leiblers.py:68-82
static void kl_host_terms(const double* q_sqrt, int ndim, i64 m, i64 k, double* logdet_qcov, double* trace_white) {
  double ld = 0.0, tw = 0.0;
  if (ndim == 2) {
    for (i64 i = 0; i < m * k; ++i) { ld += log(q_sqrt[i] * q_sqrt[i]); tw += q_sqrt[i] * q_sqrt[i]; }
  } else {
    for (i64 q = 0; q < k; ++q)                                    // C-ABI layout [k][m][m]
      for (i64 a = 0; a < m; ++a)
        for (i64 b = 0; b <= a; ++b) {                             // lower triangle only (tf.matrix_band_part, :64)
          const double v = q_sqrt[((size_t)q * m + a) * m + b];
          tw += v * v;
          if (a == b) ld += log(v * v);
        }
  }
  *logdet_qcov = ld; *trace_white = tw;
}

// dst [mp, mp] (device) = scale * tril(Lq [m, m] host, row-major), transposed or not, zero elsewhere: through the staging buffer
static int upload_tril(gps_handle_t h, const double* Lq, i64 m, double* dst, i64 mp, double scale, int transpose) {
  GPS_HIP(h, h->dStage.ensure((size_t)mp * mp * 8));      // (padded size: the callers that bring a result back through it need that much)
  GPS_HIP(h, hipMemcpyAsync(h->dStage.p, Lq, (size_t)m * m * 8, hipMemcpyHostToDevice, h->stream));
  return gps_launch_tril_pad(h, h->dStage.d(), m, dst, mp, scale, transpose);
}

static int conditional_tail(gps_handle_t h, CondIn& c, const double* f, const double* q_sqrt,
                            int q_sqrt_ndim, int white, int full_cov, double* fmean_out,
                            double* fvar_out, int* info, SvgpAcc* sv = nullptr) {
  const i64 m = c.m, mp = c.mp, n_new = c.n_new, nsp = c.nsp, k = c.k;
  int* d_info = (int*)h->dInfo.p;
  int rc = gps_launch_fill_info(h, d_info, INT_MAX);
  if (rc) return rc;
  const bool need_back = (!white) && (q_sqrt != nullptr);
  HipOps ops{h, c.linv, c.linvT, d_info};
  Blocked<HipOps> bl(ops);
  rc = bl.potrf_rec(c.Kmm, mp, mp, 0, 0);                          // Lm   conditionals.py:84
  if (rc) return rc;
  rc = classify_blocks(h, ops, c.Kmm, mp, mp);                     // (refined leaves only against ill-conditioned diagonal blocks)
  if (rc) return rc;
  rc = bl.trsm_rec(c.Kmm, mp, mp, 0, c.Bt, mp, nsp);               // A^T  conditionals.py:87
  if (rc) return rc;

  // f -> [k][mp]; white: fmean = A^T f ; else fmean = A^T (Lm^-1 f)   conditionals.py:99-103
  GPS_HIP(h, h->dAlpha.ensure((size_t)k * mp * 8));
  GPS_HIP(h, h->dTmp2.ensure((size_t)((full_cov && n_new * n_new > m * k) ? n_new * n_new : m * k) * 8 + 64));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, f, (size_t)m * k * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemsetAsync(h->dAlpha.p, 0, (size_t)k * mp * 8, h->stream));
  rc = gps_launch_transpose(h, h->dTmp2.d(), k, m, k, h->dAlpha.d(), mp);
  if (rc) return rc;
  if (!white) {
    rc = bl.trsv_rec(c.Kmm, mp, mp, 0, h->dAlpha.d(), mp, k);
    if (rc) return rc;
  }
  double kl_hp[2 * 64];
  if (sv) {
    // sum log diag Lm and the Mahalanobis term sum (Lm^-1 q_mu)^2 (white: sum q_mu^2)    kullback_leiblers.py:68-69,98-103
    rc = gps_launch_lml_reduce(h, c.Kmm, mp, m, h->dAlpha.d(), mp, k, h->dScal.d());
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(kl_hp, h->dScal.p, sizeof(kl_hp), hipMemcpyDeviceToHost, h->stream));
  }
  GPS_HIP(h, h->dMean.ensure((size_t)(n_new * k + n_new) * 8));
  double* dmean = h->dMean.d();
  double* dss = dmean + n_new * k;
  rc = gps_launch_rowdot(h, c.Bt, mp, n_new, mp, h->dAlpha.d(), mp, k, dmean, dss);
  if (rc) return rc;
  if (!sv) GPS_HIP(h, hipMemcpyAsync(fmean_out, dmean, (size_t)n_new * k * 8, hipMemcpyDeviceToHost, h->stream));

  // base variance (shared by all k)                                  conditionals.py:90-96
  if (!full_cov) {
    GPS_HIP(h, h->dVar.ensure((size_t)n_new * 8 * (k + 1)));
    rc = gps_launch_var_finish(h, h->dVar.d(), c.dKnnDiag, c.knn_const, dss, n_new);
    if (rc) return rc;
  } else {
    rc = gps_launch_gemm_nt(h, 0, 0, nsp, nsp, mp, c.Bt, mp, c.Bt, mp, c.dKnnFull, nsp);
    if (rc) return rc;
  }

  std::vector<double> base;        // host copies for the final assembly
  double* dYres = nullptr;         // SVGP mode: [n_new, k] on the device, after the k + 1 variance vectors
  if (sv) {
    if (full_cov || !q_sqrt) return gps_fail(h, GPS_ERR_ARG, "svgp: needs q_sqrt and marginal variances");
    GPS_HIP(h, h->dS1.ensure((size_t)n_new * k * 8 + 64 * 8));
    dYres = h->dS1.d();
    GPS_HIP(h, hipMemcpyAsync(dYres, sv->yres, (size_t)n_new * k * 8, hipMemcpyHostToDevice, h->stream));
    GPS_HIP(h, hipMemsetAsync(dYres + (size_t)n_new * k, 0, 64 * 8, h->stream));
  } else if (!full_cov) {
    base.resize(n_new);
    GPS_HIP(h, hipMemcpyAsync(base.data(), h->dVar.p, (size_t)n_new * 8, hipMemcpyDeviceToHost, h->stream));
  } else {
    base.resize((size_t)n_new * n_new);
    rc = gps_launch_extract(h, c.dKnnFull, nsp, n_new, n_new, h->dTmp2.d(), n_new, 0);
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(base.data(), h->dTmp2.p, (size_t)n_new * n_new * 8, hipMemcpyDeviceToHost, h->stream));
  }
  GPS_HIP(h, hipStreamSynchronize(h->stream));

  const size_t per = full_cov ? (size_t)n_new * n_new : (size_t)n_new;
  if (!q_sqrt) {
    if (!full_cov) {
      // fvar [n_new, k]: tile                                          conditionals.py:96,119
      for (i64 i = 0; i < n_new; ++i) for (i64 q = 0; q < k; ++q) fvar_out[i * k + q] = base[i];
    } else {
      for (i64 q = 0; q < k; ++q) memcpy(fvar_out + q * per, base.data(), per * 8);
    }
    return read_info(h, d_info, info);
  }

  // ---- q_sqrt terms                                                  conditionals.py:105-118
  if (need_back) {
    // A^T <- A^T Lm^-1  (A = Lm^-T A)                                  conditionals.py:100
    GPS_HIP(h, h->dTmp.ensure((size_t)mp * mp * 8));
    rc = gps_launch_transpose(h, c.Kmm, mp, mp, mp, h->dTmp.d(), mp);
    if (rc) return rc;
    rc = bl.trsm_rn_rec(h->dTmp.d(), mp, mp, 0, c.Bt, mp, nsp);
    if (rc) return rc;
  }
  std::vector<double> extra(sv ? 0 : per);
  GPS_HIP(h, h->dTmp3.ensure((size_t)std::max(nsp, q_sqrt_ndim == 3 ? m : (i64)0) * mp * 8));   // LTA^T [nsp, mp] (and, before it, the raw L_q [m, m])
  double* dLTA = h->dTmp3.d();
  bool lta_transposed = false;
  for (i64 q = 0; q < k; ++q) {
    if (q_sqrt_ndim == 2) {
      // LTA^T[i][j] = A^T[i][j] * q_sqrt[j][q] : one column-scaling pass          conditionals.py:107
      std::vector<double> sv(mp, 0.0);
      for (i64 j = 0; j < m; ++j) sv[j] = q_sqrt[j * k + q];
      GPS_HIP(h, h->dTmp2.ensure((size_t)mp * 8));
      GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, sv.data(), (size_t)mp * 8, hipMemcpyHostToDevice, h->stream));
      GPS_HIP(h, hipStreamSynchronize(h->stream));
      rc = gps_launch_scale_cols(h, c.Bt, mp, nsp, mp, h->dTmp2.d(), dLTA, mp);
      if (rc) return rc;
    } else {
      // LTA^T = A^T L_q ; as C = A B^T with B = L_q^T (upper) -> upload tril(L_q) transposed
      // (the user's row-major L_q goes up as it is -- into the front of dLTA, which the product below overwrites -- and is
      // transposed, masked and padded on the device)
      GPS_HIP(h, h->dTmp2.ensure((size_t)mp * mp * 8));
      const double* Lq = q_sqrt + (size_t)q * m * m;
      GPS_HIP(h, hipMemcpyAsync(dLTA, Lq, (size_t)m * m * 8, hipMemcpyHostToDevice, h->stream));
      rc = gps_launch_tril_pad(h, dLTA, m, h->dTmp2.d(), mp, 1.0, 1);
      if (rc) return rc;
      if (!full_cov) {
        // only the column sums of squares of L_q^T A are needed: form it as (L_q^T) A^T-transposed, [mp, nsp], with the
        // upper-triangular L_q^T as the A operand -- the GEMM skips its zero half (half the flop of the product below)
        lta_transposed = true;
        rc = gps_launch_gemm_nt(h, 1, /*A upper triangular*/ 2, mp, nsp, mp, h->dTmp2.d(), mp, c.Bt, mp, dLTA, nsp);
      } else {
        rc = gps_launch_gemm_nt(h, 1, 0, nsp, mp, mp, c.Bt, mp, h->dTmp2.d(), mp, dLTA, mp);
      }
      if (rc) return rc;
    }
    if (sv) {
      rc = lta_transposed ? gps_launch_colsumsq(h, dLTA, nsp, mp, n_new, dss)
                          : gps_launch_rowdot(h, dLTA, mp, n_new, mp, nullptr, mp, 0, nullptr, dss);
      if (rc) return rc;
      // sum_i (yres - fmean)^2 + fvar for this latent, fvar = base + extra           likelihoods.py:186-188
      rc = gps_launch_varexp(h, dmean, dYres, k, (int)q, h->dVar.d(), dss, n_new, dYres + (size_t)n_new * k);
      if (rc) return rc;
      if (!white && q_sqrt_ndim == 3) {
        // tr(Kuu^-1 S_q) from the L_q^T that is already on the device (dLTA no longer needs it)
        double t = 0.0;
        rc = kl_full_one(h, bl, c.Kmm, mp, m, h->dTmp2.d(), &t);
        if (rc) return rc;
        sv->kl += t;                                                    // (trace term, completed below)
      }
    } else if (!full_cov) {
      rc = lta_transposed ? gps_launch_colsumsq(h, dLTA, nsp, mp, n_new, dss)
                          : gps_launch_rowdot(h, dLTA, mp, n_new, mp, nullptr, mp, 0, nullptr, dss);
      if (rc) return rc;
      GPS_HIP(h, hipMemcpyAsync(extra.data(), dss, (size_t)n_new * 8, hipMemcpyDeviceToHost, h->stream));
      GPS_HIP(h, hipStreamSynchronize(h->stream));
      for (i64 i = 0; i < n_new; ++i) fvar_out[i * k + q] = base[i] + extra[i];
    } else {
      GPS_HIP(h, h->dVar.ensure((size_t)nsp * nsp * 8));
      rc = gps_launch_gemm_nt(h, 1, 0, nsp, nsp, mp, dLTA, mp, dLTA, mp, h->dVar.d(), nsp);
      if (rc) return rc;
      GPS_HIP(h, h->dTmp2.ensure((size_t)n_new * n_new * 8));
      rc = gps_launch_extract(h, h->dVar.d(), nsp, n_new, n_new, h->dTmp2.d(), n_new, 0);
      if (rc) return rc;
      GPS_HIP(h, hipMemcpyAsync(extra.data(), h->dTmp2.p, per * 8, hipMemcpyDeviceToHost, h->stream));
      GPS_HIP(h, hipStreamSynchronize(h->stream));
      double* o = fvar_out + q * per;
      for (size_t e = 0; e < per; ++e) o[e] = base[e] + extra[e];
    }
  }
  if (sv) {
    double part[64];
    GPS_HIP(h, hipMemcpyAsync(part, dYres + (size_t)n_new * k, sizeof(part), hipMemcpyDeviceToHost, h->stream));
    rc = read_info(h, d_info, info);
    if (rc) return rc;
    for (int b = 0; b < 64; ++b) sv->sq_sum += part[b];
    // KL[q || p]                                                           kullback_leiblers.py:68-105
    double slog = 0.0, mahal = 0.0, logdet_q = 0.0, trace = 0.0;
    for (int b = 0; b < 64; ++b) { slog += kl_hp[2 * b]; mahal += kl_hp[2 * b + 1]; }
    kl_host_terms(q_sqrt, q_sqrt_ndim, m, k, &logdet_q, &trace);
    if (!white) {
      if (q_sqrt_ndim == 2) {
        std::vector<double> kd;
        GPS_HIP(h, h->dTmp.ensure((size_t)mp * mp * 8));
        rc = kl_diag_kinv(h, bl, c.Kmm, mp, m, h->dTmp.d(), kd);
        if (rc) return rc;
        trace = 0.0;
        for (i64 j = 0; j < m; ++j) { double sq = 0.0; for (i64 q = 0; q < k; ++q) sq += q_sqrt[j * k + q] * q_sqrt[j * k + q]; trace += kd[j] * sq; }
      } else {
        trace = sv->kl;                                                    // accumulated in the loop above
      }
    }
    double twoKL = mahal - (double)(m * k) - logdet_q + trace;
    if (!white) twoKL += (double)k * 2.0 * slog;
    sv->kl = 0.5 * twoKL;
    return GPS_OK;
  }
  return read_info(h, d_info, info);
}

extern "C" int gps_conditional(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z,
                               int64_t m, int64_t d_all, double jitter, const double* Xnew,
                               int64_t n_new, const double* f, int64_t k, const double* q_sqrt,
                               int q_sqrt_ndim, int white, int full_cov, double* fmean_out,
                               double* fvar_out, int* info) {
  return with_la_retry(h, [&]() -> int {
  if (!h || !Z || !Xnew || !f || !fmean_out || !fvar_out || m <= 0 || n_new <= 0 || k <= 0 || d_all <= 0)
    return gps_fail(h, GPS_ERR_ARG, "gps_conditional: bad argument");
  if (q_sqrt && q_sqrt_ndim != 2 && q_sqrt_ndim != 3)
    return gps_fail(h, GPS_ERR_ARG, "gps_conditional: q_sqrt_ndim must be 2 or 3");
  GPS_HIP(h, hipSetDevice(h->device));
  h->have_factor = false; h->dist_have_part_factor = false;          // dK / dLinv / dAlpha are reused below
  h->refine_now = (h->leaf_refine != 0);
  if (info) *info = 0;
  CondIn c;
  c.m = m; c.mp = gps_pad(m); c.n_new = n_new; c.nsp = gps_pad(n_new); c.k = k;
  const size_t blk_bytes = (size_t)(c.mp / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  GPS_HIP(h, h->dX.ensure((size_t)m * d_all * 8));
  h->n = 0;                        // resident GPR data is gone
  GPS_HIP(h, hipMemcpyAsync(h->dX.p, Z, (size_t)m * d_all * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dXnew.ensure((size_t)n_new * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dXnew.p, Xnew, (size_t)n_new * d_all * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dK.ensure((size_t)c.mp * c.mp * 8));
  GPS_HIP(h, h->dLinv.ensure(2 * blk_bytes));
  GPS_HIP(h, h->dB.ensure((size_t)c.nsp * c.mp * 8));
  c.Kmm = h->dK.d(); c.Bt = h->dB.d(); c.linv = h->dLinv.d(); c.linvT = h->dLinv.d() + blk_bytes / 8;
  // Kmm = K(Z) + jitter I  (features.py:74-77 / conditionals.py:60) ; Kmn^T = K(Xnew, Z)
  int rc = gps_launch_kmat(h, prog, n_nodes, h->dX.d(), m, nullptr, m, d_all, jitter, c.Kmm, c.mp, c.mp, c.mp, 1, 1);
  if (rc) return rc;
  rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n_new, h->dX.d(), m, d_all, 0.0, c.Bt, c.mp, c.nsp, c.mp, 0, 0);
  if (rc) return rc;
  c.dKnnDiag = nullptr; c.dKnnFull = nullptr; c.knn_const = 0.0;
  if (!full_cov) {
    rc = gps_launch_kdiag(h, prog, n_nodes, &c.knn_const);
    if (rc) return rc;
  } else {
    GPS_HIP(h, h->dTmp.ensure((size_t)c.nsp * c.nsp * 8 + (size_t)c.mp * c.mp * 8));
    c.dKnnFull = h->dTmp.d() + (size_t)c.mp * c.mp;     // keep the first mp*mp for U = Lm^T
    rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n_new, nullptr, n_new, d_all, 0.0, c.dKnnFull, c.nsp,
                         c.nsp, c.nsp, 0, 0);
    if (rc) return rc;
  }
  return conditional_tail(h, c, f, q_sqrt, q_sqrt_ndim, white, full_cov, fmean_out, fvar_out, info);
  });
}

extern "C" int gps_base_conditional(gps_handle_t h, const double* Kmn, const double* Kmm,
                                    const double* Knn, int64_t m, int64_t n_new, const double* f,
                                    int64_t k, const double* q_sqrt, int q_sqrt_ndim, int white,
                                    int full_cov, double* fmean_out, double* fvar_out, int* info) {
  return with_la_retry(h, [&]() -> int {
  if (!h || !Kmn || !Kmm || !Knn || !f || !fmean_out || !fvar_out || m <= 0 || n_new <= 0 || k <= 0)
    return gps_fail(h, GPS_ERR_ARG, "gps_base_conditional: bad argument");
  if (q_sqrt && q_sqrt_ndim != 2 && q_sqrt_ndim != 3)
    return gps_fail(h, GPS_ERR_ARG, "gps_base_conditional: q_sqrt_ndim must be 2 or 3");
  GPS_HIP(h, hipSetDevice(h->device));
  h->have_factor = false; h->dist_have_part_factor = false;
  h->n = 0;
  h->refine_now = (h->leaf_refine != 0);
  if (info) *info = 0;
  CondIn c;
  c.m = m; c.mp = gps_pad(m); c.n_new = n_new; c.nsp = gps_pad(n_new); c.k = k;
  const size_t blk_bytes = (size_t)(c.mp / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  GPS_HIP(h, h->dK.ensure((size_t)c.mp * c.mp * 8));
  GPS_HIP(h, h->dLinv.ensure(2 * blk_bytes));
  GPS_HIP(h, h->dB.ensure((size_t)c.nsp * c.mp * 8));
  c.Kmm = h->dK.d(); c.Bt = h->dB.d(); c.linv = h->dLinv.d(); c.linvT = h->dLinv.d() + blk_bytes / 8;
  // staging buffer big enough for Kmm, Kmn and Knn
  size_t stage = (size_t)m * m;
  if ((size_t)m * n_new > stage) stage = (size_t)m * n_new;
  if (full_cov && (size_t)n_new * n_new > stage) stage = (size_t)n_new * n_new;
  GPS_HIP(h, h->dTmp3.ensure(stage * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp3.p, Kmm, (size_t)m * m * 8, hipMemcpyHostToDevice, h->stream));
  int rc = gps_launch_pad_copy(h, h->dTmp3.d(), m, m, m, c.Kmm, c.mp, c.mp, c.mp, 1, 0.0);
  if (rc) return rc;
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp3.p, Kmn, (size_t)m * n_new * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemsetAsync(c.Bt, 0, (size_t)c.nsp * c.mp * 8, h->stream));
  rc = gps_launch_transpose(h, h->dTmp3.d(), n_new, m, n_new, c.Bt, c.mp);
  if (rc) return rc;
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  c.dKnnDiag = nullptr; c.dKnnFull = nullptr; c.knn_const = 0.0;
  GPS_HIP(h, h->dTmp.ensure((size_t)c.nsp * c.nsp * 8 * (full_cov ? 1 : 0) + (size_t)c.mp * c.mp * 8 + (size_t)n_new * 8));
  if (!full_cov) {
    double* dk = h->dTmp.d() + (size_t)c.mp * c.mp;
    GPS_HIP(h, hipMemcpyAsync(dk, Knn, (size_t)n_new * 8, hipMemcpyHostToDevice, h->stream));
    c.dKnnDiag = dk;
  } else {
    c.dKnnFull = h->dTmp.d() + (size_t)c.mp * c.mp;
    GPS_HIP(h, hipMemcpyAsync(h->dTmp3.p, Knn, (size_t)n_new * n_new * 8, hipMemcpyHostToDevice, h->stream));
    rc = gps_launch_pad_copy(h, h->dTmp3.d(), n_new, n_new, n_new, c.dKnnFull, c.nsp, c.nsp, c.nsp, 0, 0.0);
    if (rc) return rc;
    GPS_HIP(h, hipStreamSynchronize(h->stream));
  }
  return conditional_tail(h, c, f, q_sqrt, q_sqrt_ndim, white, full_cov, fmean_out, fvar_out, info);
  });
}

// ---- SVGP bound: models/svgp.py:108-125 for the Gaussian likelihood ----------------------------------------------
// elbo = scale * sum_{i,q} E_q[log N(y | f, sigma^2)] - KL[q(u) || p(u)] ; Kuu / Kuf and everything O(M^2 N) stay in HBM
extern "C" int gps_svgp_elbo(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                             int64_t d_all, double jitter, const double* X, int64_t n, const double* yres,
                             const double* q_mu, int64_t k, const double* q_sqrt, int q_sqrt_ndim, int white,
                             double noise_var, double scale, double* elbo, double* kl_out, double* var_exp_sum,
                             int* info) {
  return with_la_retry(h, [&]() -> int {
  if (!h || !Z || !X || !yres || !q_mu || !q_sqrt || !elbo || m <= 0 || n <= 0 || k <= 0 || d_all <= 0 || !(noise_var > 0.0))
    return gps_fail(h, GPS_ERR_ARG, "gps_svgp_elbo: bad argument");
  if (q_sqrt_ndim != 2 && q_sqrt_ndim != 3) return gps_fail(h, GPS_ERR_ARG, "gps_svgp_elbo: q_sqrt_ndim must be 2 or 3");
  GPS_HIP(h, hipSetDevice(h->device));
  h->have_factor = false; h->dist_have_part_factor = false; h->n = 0;
  h->refine_now = (h->leaf_refine != 0);
  if (info) *info = 0;
  CondIn c;
  c.m = m; c.mp = gps_pad(m); c.n_new = n; c.nsp = gps_pad(n); c.k = k;
  const size_t blk_bytes = (size_t)(c.mp / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  GPS_HIP(h, h->dX.ensure((size_t)m * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dX.p, Z, (size_t)m * d_all * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dXnew.ensure((size_t)n * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dXnew.p, X, (size_t)n * d_all * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dK.ensure((size_t)c.mp * c.mp * 8));
  GPS_HIP(h, h->dLinv.ensure(2 * blk_bytes));
  GPS_HIP(h, h->dB.ensure((size_t)c.nsp * c.mp * 8));
  c.Kmm = h->dK.d(); c.Bt = h->dB.d(); c.linv = h->dLinv.d(); c.linvT = h->dLinv.d() + blk_bytes / 8;
  int rc = gps_launch_kmat(h, prog, n_nodes, h->dX.d(), m, nullptr, m, d_all, jitter, c.Kmm, c.mp, c.mp, c.mp, 1, 1);
  if (rc) return rc;
  rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n, h->dX.d(), m, d_all, 0.0, c.Bt, c.mp, c.nsp, c.mp, 0, 0);
  if (rc) return rc;
  c.dKnnDiag = nullptr; c.dKnnFull = nullptr; c.knn_const = 0.0;
  rc = gps_launch_kdiag(h, prog, n_nodes, &c.knn_const);
  if (rc) return rc;
  SvgpAcc sv; sv.yres = yres; sv.noise_var = noise_var;
  int linfo = 0;
  rc = conditional_tail(h, c, q_mu, q_sqrt, q_sqrt_ndim, white, 0, nullptr, nullptr, &linfo, &sv);
  if (info) *info = linfo;
  if (rc || linfo) return rc;
  // likelihoods.py:186-188 summed over all points and latents
  const double ve = (double)n * (double)k * (-0.5 * log(2.0 * M_PI) - 0.5 * log(noise_var)) - 0.5 * sv.sq_sum / noise_var;
  if (var_exp_sum) *var_exp_sum = ve;
  if (kl_out) *kl_out = sv.kl;
  *elbo = ve * scale - h->svgp_kl_weight * sv.kl;       // (weight 1 / P when the data points are sharded over P ranks)
  return GPS_OK;
  });
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

// ---- gradient of the SVGP bound (whitened parametrisation, Gaussian likelihood) ------------------------------------------
// What TF autodiff gives the reference's optimiser for models/svgp.py:108-125 (examples/svgp.py:159-161 minimises
// `objective`): reverse mode at the matrix level, every O(M^2 N) product on the fp64 MFMA and resident in HBM.
//   forward (gps_svgp_elbo): Lm = chol(Kuu + jitter I), A = Lm^-1 Kuf, mu = A^T q_mu,
//                            var_q = Kdiag - colsum(A^2) + colsum((L_q^T A)^2)
//   E  = scale (Y - mu) / s2                                             d ELBO / d mu
//   g(q_mu) = A E - q_mu ;  g(L_q) = tril(-(scale/s2) (A A^T) L_q - L_q + diag(1 / L_q,ii))   (diagonal q_sqrt: elementwise)
//   Abar = q_mu E^T + (scale/s2) (k I - sum_q L_q L_q^T) A                 d ELBO / d A
//   Kuf_bar = Lm^-T Abar ;  Lm_bar = -tril(Kuf_bar A^T) ;  Kuu_bar = Lm^-T (Phi(Lm^T Lm_bar) + Phi(.)^T) Lm^-1 / 2   (Phi: tril, diagonal halved)
//   d/d theta = <Kuf_bar, dKuf> + <Kuu_bar, dKuu> + kbar dKdiag            (gps_launch_kmat_vjp: the kernel-matrix VJP)
// (Checked in tests/test_gpu_grad.py against a CPU restatement and finite differences.)  The inducing inputs Z are held
// fixed unless the caller asks for grad_Z (gps_launch_kmat_input_vjp: the kernel-matrix build differentiated in its points).
// Unwhitened parametrisation (white == 0; examples/svgp.py:146 runs with whiten=False): the bound is the whitened one at
//   m_w = Lm^-1 q_mu,  L_w,q = Lm^-1 L_q          (same predictive moments, KL invariant under the linear map),
// so the whitened gradient (g_w, G_w) is computed at (m_w, L_w) and pulled back:
//   g(q_mu) = Lm^-T g_w ;  g(L_q) = tril(Lm^-T G_w,q) ;  Lm_bar += -tril(g(q_mu) m_w^T + sum_q (Lm^-T G_w,q) L_w,q^T)
// (the last term is the dependence of m_w, L_w on Lm; it joins Lm_bar before the Cholesky adjoint).
static int svgp_whiten(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, i64 m, i64 d_all,
                       double jitter, const double* q_mu, i64 k, const double* q_sqrt, int q_sqrt_ndim,
                       std::vector<double>& mw, std::vector<double>& Lw, int* info) {
  GPS_HIP(h, hipSetDevice(h->device));
  h->have_factor = false; h->dist_have_part_factor = false; h->n = 0;
  h->refine_now = (h->leaf_refine != 0);
  const i64 mp = gps_pad(m);
  const size_t blk_bytes = (size_t)(mp / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  GPS_HIP(h, h->dX.ensure((size_t)m * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dX.p, Z, (size_t)m * d_all * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dK.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dLinv.ensure(2 * blk_bytes));
  int rc = gps_launch_kmat(h, prog, n_nodes, h->dX.d(), m, nullptr, m, d_all, jitter, h->dK.d(), mp, mp, mp, 1, 1);
  if (rc) return rc;
  int* d_info = (int*)h->dInfo.p;
  rc = gps_launch_fill_info(h, d_info, INT_MAX);
  if (rc) return rc;
  HipOps ops{h, h->dLinv.d(), h->dLinv.d() + blk_bytes / 8, d_info};
  Blocked<HipOps> bl(ops);
  rc = bl.potrf_rec(h->dK.d(), mp, mp, 0, 0);
  if (rc) return rc;
  rc = read_info(h, d_info, info);
  if (rc || (info && *info)) return rc;
  rc = classify_blocks(h, ops, h->dK.d(), mp, mp);
  if (rc) return rc;
  // m_w^T = (Lm^-1 q_mu)^T : right-hand sides as rows
  std::vector<double> buf((size_t)GPS_TILE * mp, 0.0);
  for (i64 j = 0; j < m; ++j) for (i64 q = 0; q < k; ++q) buf[(size_t)q * mp + j] = q_mu[j * k + q];
  GPS_HIP(h, h->dG3.ensure(buf.size() * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dG3.p, buf.data(), buf.size() * 8, hipMemcpyHostToDevice, h->stream));
  rc = bl.trsm_rec(h->dK.d(), mp, mp, 0, h->dG3.d(), mp, GPS_TILE);
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(buf.data(), h->dG3.p, buf.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  mw.assign((size_t)m * k, 0.0);
  for (i64 j = 0; j < m; ++j) for (i64 q = 0; q < k; ++q) mw[j * k + q] = buf[(size_t)q * mp + j];
  // L_w,q^T = (Lm^-1 L_q)^T
  Lw.assign((size_t)k * m * m, 0.0);
  std::vector<double> LT((size_t)mp * mp);
  GPS_HIP(h, h->dTmp2.ensure((size_t)mp * mp * 8));
  for (i64 q = 0; q < k; ++q) {
    if (q_sqrt_ndim == 2) {
      std::fill(LT.begin(), LT.end(), 0.0);
      for (i64 a = 0; a < m; ++a) LT[(size_t)a * mp + a] = q_sqrt[a * k + q];
      GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, LT.data(), LT.size() * 8, hipMemcpyHostToDevice, h->stream));
    } else {
      rc = upload_tril(h, q_sqrt + (size_t)q * m * m, m, h->dTmp2.d(), mp, 1.0, 1);      // (transposed and padded on the device)
      if (rc) return rc;
    }
    rc = bl.trsm_rec(h->dK.d(), mp, mp, 0, h->dTmp2.d(), mp, mp);
    if (rc) return rc;
    // back as rows of L_w,q: transposed on the device, read back in one sequential pass
    GPS_HIP(h, h->dStage.ensure((size_t)mp * mp * 8));
    rc = gps_launch_transpose(h, h->dTmp2.d(), mp, mp, mp, h->dStage.d(), mp);
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(LT.data(), h->dStage.p, LT.size() * 8, hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    double* out = Lw.data() + (size_t)q * m * m;
    for (i64 a = 0; a < m; ++a) for (i64 b = 0; b <= a; ++b) out[a * m + b] = LT[(size_t)a * mp + b];
  }
  return GPS_OK;
}

static int svgp_elbo_grad_body(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                                  int64_t d_all, double jitter, const double* X, int64_t n, const double* yres,
                                  const double* q_mu, int64_t k, const double* q_sqrt, int q_sqrt_ndim, int white,
                                  double noise_var, double scale, double* elbo, double* grad_slots, int n_slots_cap,
                                  int* n_slots_out, double* grad_noise, double* grad_q_mu, double* grad_q_sqrt,
                                  double* grad_mean, double* grad_Z, int* info);
// (wrapped like every factorising entry point: a missed look-ahead hand-over re-runs the body once, with_la_retry)
extern "C" int gps_svgp_elbo_grad(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                                  int64_t d_all, double jitter, const double* X, int64_t n, const double* yres,
                                  const double* q_mu, int64_t k, const double* q_sqrt, int q_sqrt_ndim, int white,
                                  double noise_var, double scale, double* elbo, double* grad_slots, int n_slots_cap,
                                  int* n_slots_out, double* grad_noise, double* grad_q_mu, double* grad_q_sqrt,
                                  double* grad_mean, double* grad_Z, int* info) {
  return with_la_retry(h, [&]() -> int { return svgp_elbo_grad_body(h, prog, n_nodes, Z, m, d_all, jitter, X, n, yres, q_mu, k, q_sqrt, q_sqrt_ndim, white, noise_var, scale, elbo, grad_slots, n_slots_cap, n_slots_out, grad_noise, grad_q_mu, grad_q_sqrt, grad_mean, grad_Z, info); });
}
static int svgp_elbo_grad_body(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                                  int64_t d_all, double jitter, const double* X, int64_t n, const double* yres,
                                  const double* q_mu, int64_t k, const double* q_sqrt, int q_sqrt_ndim, int white,
                                  double noise_var, double scale, double* elbo, double* grad_slots, int n_slots_cap,
                                  int* n_slots_out, double* grad_noise, double* grad_q_mu, double* grad_q_sqrt,
                                  double* grad_mean, double* grad_Z, int* info) {
  if (!h || !elbo || !grad_slots || !grad_noise || !grad_q_mu || !grad_q_sqrt)
    return gps_fail(h, GPS_ERR_ARG, "gps_svgp_elbo_grad: bad argument");
  if (k > GPS_TILE) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gps_svgp_elbo_grad: at most 128 latent functions");
  if (!Z || !q_mu || !q_sqrt || m <= 0 || k <= 0 || (q_sqrt_ndim != 2 && q_sqrt_ndim != 3))
    return gps_fail(h, GPS_ERR_ARG, "gps_svgp_elbo_grad: bad argument");
  int ns = 0;
  int rc = gps_grad_general_slots(h, prog, n_nodes, &ns);
  if (rc) return rc;
  if (n_slots_out) *n_slots_out = ns;
  if (ns > n_slots_cap) return gps_fail(h, GPS_ERR_ARG, "gps_svgp_elbo_grad: grad_slots too small");
  double kl = 0.0, ve = 0.0;
  int linfo = 0;
  // unwhitened: differentiate the whitened bound at (m_w, L_w) and pull the result back (see above)
  const bool unwhite = !white;
  const double* const q_mu_in = q_mu; const double* const q_sqrt_in = q_sqrt; const int ndim_in = q_sqrt_ndim;
  double* const grad_q_sqrt_out = grad_q_sqrt;
  std::vector<double> mw_h, Lw_h, gw_tmp;
  if (unwhite) {
    rc = svgp_whiten(h, prog, n_nodes, Z, m, d_all, jitter, q_mu, k, q_sqrt, q_sqrt_ndim, mw_h, Lw_h, &linfo);
    if (info) *info = linfo;
    if (rc || linfo) return rc;
    q_mu = mw_h.data(); q_sqrt = Lw_h.data(); q_sqrt_ndim = 3;
    if (ndim_in == 2) { gw_tmp.assign((size_t)k * m * m, 0.0); grad_q_sqrt = gw_tmp.data(); }
  }
  (void)q_mu_in; (void)q_sqrt_in;
  rc = gps_svgp_elbo(h, prog, n_nodes, Z, m, d_all, jitter, X, n, yres, q_mu, k, q_sqrt, q_sqrt_ndim, 1, noise_var, scale,
                     elbo, &kl, &ve, &linfo);
  if (info) *info = linfo;
  if (rc || linfo) return rc;
  // what the forward pass left on the device: dK = Lm [mp, mp], dLinv (+T), dB = A^T [nsp, mp], dX = Z, dXnew = X,
  // dMean = fmean [n, k], dS1 = yres [n, k]
  const i64 mp = gps_pad(m), nsp = gps_pad(n);
  const double w = scale, s2 = noise_var;
  const size_t blk_bytes = (size_t)(mp / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  HipOps ops{h, h->dLinv.d(), h->dLinv.d() + blk_bytes / 8, (int*)h->dInfo.p};
  Blocked<HipOps> bl(ops);
  double* Lm = h->dK.d();
  double* Bt = h->dB.d();
  // sum ((y - mu)^2 + var) back out of the variational expectations (likelihoods.py:186-188)
  const double c0 = -0.5 * log(2.0 * M_PI) - 0.5 * log(s2);
  const double sq_sum = ((double)n * (double)k * c0 - ve) * 2.0 * s2;
  *grad_noise = w * (-(double)n * (double)k / (2.0 * s2) + sq_sum / (2.0 * s2 * s2));

  // E^T [k][nsp]
  GPS_HIP(h, h->dA.ensure((size_t)k * nsp * 8));
  double* Et = h->dA.d();
  rc = gps_launch_svgp_et(h, h->dS1.d(), h->dMean.d(), k, n, nsp, w / s2, Et);
  if (rc) return rc;
  if (grad_mean) {                                       // d ELBO / d mean_function(X) = E   [n, k]
    GPS_HIP(h, h->dTmp2.ensure((size_t)n * k * 8));
    rc = gps_launch_transpose(h, Et, nsp, k, n, h->dTmp2.d(), k);
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(grad_mean, h->dTmp2.p, (size_t)n * k * 8, hipMemcpyDeviceToHost, h->stream));
  }
  // A = (A^T)^T [mp, nsp] ;  A E [m, k] and diag(A A^T) in one pass over A
  GPS_HIP(h, h->dS2.ensure((size_t)mp * nsp * 8));
  double* Am = h->dS2.d();
  rc = gps_launch_transpose(h, Bt, mp, nsp, mp, Am, nsp);
  if (rc) return rc;
  GPS_HIP(h, h->dG4.ensure((size_t)(mp * k + 2 * mp) * 8));
  double* dAE = h->dG4.d();
  double* dDiag = dAE + (size_t)mp * k;
  double* dCoef = dDiag + mp;
  rc = gps_launch_rowdot(h, Am, nsp, m, nsp, Et, nsp, k, dAE, dDiag);
  if (rc) return rc;
  std::vector<double> hAE((size_t)m * k), hDiag((size_t)m);
  GPS_HIP(h, hipMemcpyAsync(hAE.data(), dAE, hAE.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipMemcpyAsync(hDiag.data(), dDiag, hDiag.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  const double klw = h->svgp_kl_weight;          // the KL's share of this rank (its three gradient terms below)
  for (i64 i = 0; i < m * k; ++i) grad_q_mu[i] = hAE[i] - klw * q_mu[i];                    // (KL white: q_mu)

  // Abar^T [nsp, mp] = coef (.) A^T + E q_mu^T  (- (w/s2) sum_q (A^T L_q) L_q^T for a full q_sqrt, below)
  std::vector<double> coef((size_t)mp, 0.0), qmp((size_t)mp * k, 0.0);
  for (i64 j = 0; j < m; ++j) {
    double c = (double)k;
    if (q_sqrt_ndim == 2) for (i64 q = 0; q < k; ++q) c -= q_sqrt[j * k + q] * q_sqrt[j * k + q];
    coef[j] = (w / s2) * c;
    for (i64 q = 0; q < k; ++q) qmp[j * k + q] = q_mu[j * k + q];
  }
  GPS_HIP(h, h->dG3.ensure((size_t)mp * k * 8 + 64));
  GPS_HIP(h, h->ring.upload(dCoef, coef.data(), (size_t)mp * 8, h->stream));
  GPS_HIP(h, hipMemcpyAsync(h->dG3.p, qmp.data(), (size_t)mp * k * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  GPS_HIP(h, h->dY.ensure((size_t)nsp * mp * 8));
  double* Abar = h->dY.d();
  rc = gps_launch_svgp_abar(h, Bt, mp, nsp, mp, dCoef, Et, nsp, h->dG3.d(), k, Abar);
  if (rc) return rc;

  if (q_sqrt_ndim == 2) {
    for (i64 j = 0; j < m; ++j)
      for (i64 q = 0; q < k; ++q) {
        const double sv = q_sqrt[j * k + q];
        grad_q_sqrt[j * k + q] = -(w / s2) * hDiag[j] * sv + klw * (-sv + 1.0 / sv);
      }
  } else {
    // A A^T (lower by one long-K GEMM, mirrored) ; per latent: S += (w/s2) L_q L_q^T and (A A^T) L_q (both M^3) ; then ONE
    // M^2 N product for all latents:  Abar^T -= A^T S   (S symmetric; instead of (A^T L_q) L_q^T per latent: 2k -> 1 products)
    GPS_HIP(h, h->dS3.ensure((size_t)mp * mp * 8));
    double* AAT = h->dS3.d();
    rc = gps_launch_gemm_nt(h, 1, 1, mp, mp, nsp, Am, nsp, Am, nsp, AAT, mp);
    if (rc) return rc;
    rc = gps_launch_tri_map(h, AAT, mp, mp, 0);
    if (rc) return rc;
    GPS_HIP(h, h->dTmp.ensure((size_t)mp * mp * 8));
    GPS_HIP(h, h->dTmp2.ensure((size_t)mp * mp * 8));
    GPS_HIP(h, h->dG1.ensure((size_t)mp * mp * 8));
    GPS_HIP(h, h->dG2.ensure((size_t)mp * mp * 8));
    double* Ssum = h->dG2.d();
    const double rs = sqrt(w / s2);
    std::vector<double> G((size_t)mp * mp);
    for (i64 q = 0; q < k; ++q) {
      const double* Lq = q_sqrt + (size_t)q * m * m;                  // C-ABI layout [k][m][m]
      // L_q^T and sqrt(w / s2) L_q, masked and padded on the device from one upload
      rc = upload_tril(h, Lq, m, h->dTmp2.d(), mp, 1.0, 1);
      if (!rc) rc = gps_launch_tril_pad(h, h->dStage.d(), m, h->dTmp.d(), mp, rs, 0);
      if (rc) return rc;
      rc = gps_launch_gemm_nt(h, q == 0 ? 1 : 2, 0, mp, mp, mp, h->dTmp.d(), mp, h->dTmp.d(), mp, Ssum, mp);    // S (+)= (w/s2) L_q L_q^T
      if (rc) return rc;
      rc = gps_launch_gemm_nt(h, 1, 0, mp, mp, mp, AAT, mp, h->dTmp2.d(), mp, h->dG1.d(), mp);           // (A A^T) L_q
      if (rc) return rc;
      GPS_HIP(h, hipMemcpyAsync(G.data(), h->dG1.p, (size_t)mp * mp * 8, hipMemcpyDeviceToHost, h->stream));
      GPS_HIP(h, hipStreamSynchronize(h->stream));
      double* gq = grad_q_sqrt + (size_t)q * m * m;
      for (i64 a = 0; a < m; ++a)
        for (i64 b = 0; b < m; ++b)
          gq[a * m + b] = (b > a) ? 0.0 : (-(w / s2) * G[(size_t)a * mp + b] + klw * (-Lq[a * m + b] + (a == b ? 1.0 / Lq[a * m + a] : 0.0)));
    }
    rc = gps_launch_gemm_nt(h, 0, 0, nsp, mp, mp, Bt, mp, Ssum, mp, Abar, mp);                           // Abar^T -= A^T S
    if (rc) return rc;
  }
  // Kuf_bar^T = Abar^T Lm^-1  (X Lm = Abar^T through U = Lm^T), then Kuf_bar [mp, nsp]
  GPS_HIP(h, h->dTmp.ensure((size_t)mp * mp * 8));
  double* U = h->dTmp.d();
  rc = gps_launch_transpose(h, Lm, mp, mp, mp, U, mp);
  if (rc) return rc;
  rc = gps_launch_tri_map(h, U, mp, mp, 3);            // (above the diagonal blocks the factor's buffer was never written)
  if (rc) return rc;
  rc = bl.trsm_rn_rec(U, mp, mp, 0, Abar, mp, nsp);
  if (rc) return rc;
  GPS_HIP(h, h->dS1.ensure((size_t)mp * nsp * 8));
  double* KufBar = h->dS1.d();
  rc = gps_launch_transpose(h, Abar, mp, nsp, mp, KufBar, nsp);
  if (rc) return rc;
  // Lm_bar = -tril(Kuf_bar A^T)
  GPS_HIP(h, h->dS3.ensure((size_t)mp * mp * 8));
  double* LmBar = h->dS3.d();
  rc = gps_launch_gemm_nt(h, 1, 1, mp, mp, nsp, KufBar, nsp, Am, nsp, LmBar, mp);
  if (rc) return rc;
  if (unwhite) {
    // pull-back of (g_w, G_w) through m_w = Lm^-1 q_mu, L_w = Lm^-1 L_q; their dependence on Lm joins Lm_bar (still
    // un-negated here: Lm_bar = -tril(Kuf_bar A^T + g(q_mu) m_w^T + sum_q (Lm^-T G_w,q) L_w,q^T))
    std::vector<double> buf((size_t)GPS_TILE * mp, 0.0);
    for (i64 j = 0; j < m; ++j) for (i64 q = 0; q < k; ++q) buf[(size_t)q * mp + j] = grad_q_mu[j * k + q];
    GPS_HIP(h, h->dG3.ensure(buf.size() * 8));
    GPS_HIP(h, hipMemcpyAsync(h->dG3.p, buf.data(), buf.size() * 8, hipMemcpyHostToDevice, h->stream));
    rc = bl.trsm_rn_rec(U, mp, mp, 0, h->dG3.d(), mp, GPS_TILE);                       // rows: g_w^T Lm^-1 = (Lm^-T g_w)^T
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(buf.data(), h->dG3.p, buf.size() * 8, hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    std::vector<double> ga((size_t)mp * GPS_TILE, 0.0), mb((size_t)mp * GPS_TILE, 0.0);
    for (i64 j = 0; j < m; ++j)
      for (i64 q = 0; q < k; ++q) {
        const double g = buf[(size_t)q * mp + j];
        grad_q_mu[j * k + q] = g;
        ga[(size_t)j * GPS_TILE + q] = g;
        mb[(size_t)j * GPS_TILE + q] = q_mu[j * k + q];
      }
    GPS_HIP(h, h->dG1.ensure((size_t)mp * mp * 8));
    GPS_HIP(h, h->dG2.ensure((size_t)mp * mp * 8));
    GPS_HIP(h, h->dTmp2.ensure((size_t)mp * mp * 8));
    GPS_HIP(h, hipMemcpyAsync(h->dG1.p, ga.data(), ga.size() * 8, hipMemcpyHostToDevice, h->stream));
    GPS_HIP(h, hipMemcpyAsync(h->dG2.p, mb.data(), mb.size() * 8, hipMemcpyHostToDevice, h->stream));
    rc = gps_launch_gemm_nt(h, 2, 1, mp, mp, GPS_TILE, h->dG1.d(), GPS_TILE, h->dG2.d(), GPS_TILE, LmBar, mp);   // += g(q_mu) m_w^T
    if (rc) return rc;
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    std::vector<double> T((size_t)mp * mp);
    for (i64 q = 0; q < k; ++q) {
      const double* Gw = grad_q_sqrt + (size_t)q * m * m;              // whitened gradient, lower triangular [m][m]
      const double* Lwq = q_sqrt + (size_t)q * m * m;
      std::fill(T.begin(), T.end(), 0.0);
      for (i64 a = 0; a < m; ++a) for (i64 b = 0; b <= a; ++b) T[(size_t)b * mp + a] = Gw[a * m + b];      // G_w^T
      GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, T.data(), T.size() * 8, hipMemcpyHostToDevice, h->stream));
      rc = bl.trsm_rn_rec(U, mp, mp, 0, h->dTmp2.d(), mp, mp);                          // (Lm^-T G_w)^T
      if (rc) return rc;
      rc = gps_launch_transpose(h, h->dTmp2.d(), mp, mp, mp, h->dG1.d(), mp);          // Lm^-T G_w
      if (rc) return rc;
      GPS_HIP(h, hipMemcpyAsync(T.data(), h->dG1.p, T.size() * 8, hipMemcpyDeviceToHost, h->stream));
      GPS_HIP(h, hipStreamSynchronize(h->stream));
      if (ndim_in == 2) {
        for (i64 a = 0; a < m; ++a) grad_q_sqrt_out[a * k + q] = T[(size_t)a * mp + a];
      } else {
        double* gq = grad_q_sqrt_out + (size_t)q * m * m;
        for (i64 a = 0; a < m; ++a) for (i64 b = 0; b < m; ++b) gq[a * m + b] = (b <= a) ? T[(size_t)a * mp + b] : 0.0;
      }
      std::fill(T.begin(), T.end(), 0.0);
      for (i64 a = 0; a < m; ++a) for (i64 b = 0; b <= a; ++b) T[(size_t)a * mp + b] = Lwq[a * m + b];
      GPS_HIP(h, hipMemcpyAsync(h->dG2.p, T.data(), T.size() * 8, hipMemcpyHostToDevice, h->stream));
      rc = gps_launch_gemm_nt(h, 2, 1, mp, mp, mp, h->dG1.d(), mp, h->dG2.d(), mp, LmBar, mp);               // += (Lm^-T G_w) L_w^T
      if (rc) return rc;
      GPS_HIP(h, hipStreamSynchronize(h->stream));
    }
  }
  rc = gps_launch_tri_map(h, LmBar, mp, mp, 1);
  if (rc) return rc;
  // Cholesky adjoint: Kuu_bar = Lm^-T (Phi(P) + Phi(P)^T) Lm^-1 / 2, P = Lm^T Lm_bar   (chol_adjoint2 leaves twice that)
  GPS_HIP(h, h->dG1.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dG2.ensure((size_t)mp * mp * 8));
  double* LmBarT = h->dG1.d();
  rc = chol_adjoint2(h, bl, U, LmBar, LmBarT, h->dG2.d(), mp);
  if (rc) return rc;
  // contractions with the kernel derivatives
  for (int sI = 0; sI < ns; ++sI) grad_slots[sI] = 0.0;
  rc = gps_launch_kmat_vjp(h, prog, n_nodes, h->dX.d(), m, h->dXnew.d(), n, d_all, KufBar, nsp, 0, grad_slots);
  if (rc) return rc;
  {
    std::vector<double> uu((size_t)ns, 0.0);
    rc = gps_launch_kmat_vjp(h, prog, n_nodes, h->dX.d(), m, nullptr, 0, d_all, LmBarT, mp, 0, uu.data());
    if (rc) return rc;
    for (int sI = 0; sI < ns; ++sI) grad_slots[sI] += 0.5 * uu[sI];
  }
  rc = gps_kdiag_vjp(h, prog, n_nodes, d_all, -w * (double)k * (double)n / (2.0 * s2), grad_slots);
  if (rc) return rc;
  if (grad_Z) {
    // inducing inputs: Z enters through Kuf = k(Z, X) (cotangent Kuf_bar) and Kuu = k(Z, Z) (cotangent Kuu_bar / 2 on the full
    // symmetric matrix: both arguments move, which doubles the first-argument gradient); Kdiag and the jitter do not depend on Z
    for (i64 i = 0; i < m * d_all; ++i) grad_Z[i] = 0.0;
    rc = gps_launch_kmat_input_vjp(h, prog, n_nodes, h->dX.d(), m, h->dXnew.d(), n, d_all, KufBar, nsp, 1.0, grad_Z);
    if (rc) return rc;
    rc = gps_launch_kmat_input_vjp(h, prog, n_nodes, h->dX.d(), m, nullptr, 0, d_all, LmBarT, mp, 1.0, grad_Z);
    if (rc) return rc;
  }
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

// ---- vector-Jacobian product of kernels.K: grad_slots = sum_ij W[i][j] d k(X_i, X2_j) / d theta ---------------------------
// (reverse-mode autodiff through kern.K(X, X2), kernels.py:408-439 / 1071-1084 / neural_kernel_network.py:41-47, for a
// caller-supplied cotangent W host [n, m]; X2 == NULL: K(X, X), W [n, n] taken as given -- no symmetrisation.)
extern "C" int gps_kmat_vjp(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* X, int64_t n,
                            const double* X2, int64_t m, int64_t d_all, const double* W, double* grad_slots,
                            int n_slots_cap, int* n_slots_out) {
  if (!h || !X || !W || !grad_slots || n <= 0 || d_all <= 0) return gps_fail(h, GPS_ERR_ARG, "gps_kmat_vjp: bad argument");
  GPS_HIP(h, hipSetDevice(h->device));
  if (!X2) m = n;
  int ns = 0;
  int rc = gps_grad_general_slots(h, prog, n_nodes, &ns);
  if (rc) return rc;
  if (n_slots_out) *n_slots_out = ns;
  if (ns > n_slots_cap) return gps_fail(h, GPS_ERR_ARG, "gps_kmat_vjp: grad_slots too small");
  GPS_HIP(h, h->dXnew.ensure((size_t)n * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dXnew.p, X, (size_t)n * d_all * 8, hipMemcpyHostToDevice, h->stream));
  const double* dX2 = nullptr;
  if (X2) {
    GPS_HIP(h, h->dTmp3.ensure((size_t)m * d_all * 8));
    GPS_HIP(h, hipMemcpyAsync(h->dTmp3.p, X2, (size_t)m * d_all * 8, hipMemcpyHostToDevice, h->stream));
    dX2 = h->dTmp3.d();
  }
  GPS_HIP(h, h->dTmp.ensure((size_t)n * m * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp.p, W, (size_t)n * m * 8, hipMemcpyHostToDevice, h->stream));
  rc = gps_launch_kmat_vjp(h, prog, n_nodes, h->dXnew.d(), n, dX2, m, d_all, h->dTmp.d(), m, 0, grad_slots);
  if (rc) return rc;
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

// gradient of sum_ij W[i][j] k(X_i, X2_j) with respect to the points X (first argument): grad_X host [n, d_all].
// X2 == NULL: k(X_i, X_j), BOTH arguments move (W [n, n] as given): grad = first-argument gradient of (W + W^T).
extern "C" int gps_kmat_input_vjp(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* X, int64_t n,
                                  const double* X2, int64_t m, int64_t d_all, const double* W, double* grad_X) {
  if (!h || !X || !W || !grad_X || n <= 0 || d_all <= 0) return gps_fail(h, GPS_ERR_ARG, "gps_kmat_input_vjp: bad argument");
  GPS_HIP(h, hipSetDevice(h->device));
  if (!X2) m = n;
  GPS_HIP(h, h->dXnew.ensure((size_t)n * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dXnew.p, X, (size_t)n * d_all * 8, hipMemcpyHostToDevice, h->stream));
  const double* dX2 = nullptr;
  if (X2) {
    GPS_HIP(h, h->dTmp3.ensure((size_t)m * d_all * 8));
    GPS_HIP(h, hipMemcpyAsync(h->dTmp3.p, X2, (size_t)m * d_all * 8, hipMemcpyHostToDevice, h->stream));
    dX2 = h->dTmp3.d();
  }
  GPS_HIP(h, h->dTmp.ensure((size_t)n * m * 8));
  std::vector<double> Ws;
  const double* Wsrc = W;
  if (!X2) {                                   // symmetrise: d/dx_i of sum_ij W_ij k(x_i, x_j) = sum_j (W_ij + W_ji) d1 k(x_i, x_j)
    Ws.resize((size_t)n * n);
    for (int64_t i = 0; i < n; ++i) for (int64_t j = 0; j < n; ++j) Ws[(size_t)i * n + j] = W[(size_t)i * n + j] + W[(size_t)j * n + i];
    Wsrc = Ws.data();
  }
  GPS_HIP(h, hipMemcpyAsync(h->dTmp.p, Wsrc, (size_t)n * m * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  for (int64_t i = 0; i < n * d_all; ++i) grad_X[i] = 0.0;
  int rc = gps_launch_kmat_input_vjp(h, prog, n_nodes, h->dXnew.d(), n, dX2, m, d_all, h->dTmp.d(), m, 1.0, grad_X);
  if (rc) return rc;
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

// ---- KL[q || p], q = N(q_mu, q_sqrt q_sqrt^T), p = N(0, K) or N(0, I): kullback_leiblers.py:26-105 -------------------
// K host [m, m] or NULL; q_mu host [m, k]; q_sqrt host [m, k] (ndim 2) or [k, m, m] (ndim 3, as gps_conditional).
extern "C" int gps_gauss_kl(gps_handle_t h, const double* K, int64_t m, const double* q_mu, int64_t k,
                            const double* q_sqrt, int q_sqrt_ndim, double* kl_out, int* info) {
  return with_la_retry(h, [&]() -> int {
  if (!h || !q_mu || !q_sqrt || !kl_out || m <= 0 || k <= 0) return gps_fail(h, GPS_ERR_ARG, "gps_gauss_kl: bad argument");
  if (q_sqrt_ndim != 2 && q_sqrt_ndim != 3) return gps_fail(h, GPS_ERR_ARG, "gps_gauss_kl: q_sqrt_ndim must be 2 or 3");
  if (info) *info = 0;
  double logdet_q = 0.0, trace = 0.0, mahal = 0.0, slog = 0.0;
  kl_host_terms(q_sqrt, q_sqrt_ndim, m, k, &logdet_q, &trace);
  if (!K) {                                            // p = N(0, I): nothing to factor
    for (i64 i = 0; i < m * k; ++i) mahal += q_mu[i] * q_mu[i];
    *kl_out = 0.5 * (mahal - (double)(m * k) - logdet_q + trace);
    return GPS_OK;
  }
  GPS_HIP(h, hipSetDevice(h->device));
  h->refine_now = (h->leaf_refine != 0);
  const i64 mp = gps_pad(m);
  const size_t blk_bytes = (size_t)(mp / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  // dS2: Lp ; dS4: block inverses ; dS3: alpha ; dTmp / dTmp2: work
  GPS_HIP(h, h->dS2.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dS4.ensure(2 * blk_bytes));
  GPS_HIP(h, h->dS3.ensure((size_t)k * mp * 8));
  GPS_HIP(h, h->dTmp.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dTmp2.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp.p, K, (size_t)m * m * 8, hipMemcpyHostToDevice, h->stream));
  int rc = gps_launch_pad_copy(h, h->dTmp.d(), m, m, m, h->dS2.d(), mp, mp, mp, 1, 0.0);
  if (rc) return rc;
  int* d_info = (int*)h->dInfo.p;
  rc = gps_launch_fill_info(h, d_info, INT_MAX);
  if (rc) return rc;
  HipOps ops{h, h->dS4.d(), h->dS4.d() + blk_bytes / 8, d_info};
  Blocked<HipOps> bl(ops);
  rc = bl.potrf_rec(h->dS2.d(), mp, mp, 0, 0);                                       // Lp          :51
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, q_mu, (size_t)m * k * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemsetAsync(h->dS3.p, 0, (size_t)k * mp * 8, h->stream));
  rc = gps_launch_transpose(h, h->dTmp2.d(), k, m, k, h->dS3.d(), mp);
  if (rc) return rc;
  rc = bl.trsv_rec(h->dS2.d(), mp, mp, 0, h->dS3.d(), mp, k);                          // alpha       :52
  if (rc) return rc;
  rc = gps_launch_lml_reduce(h, h->dS2.d(), mp, m, h->dS3.d(), mp, k, h->dScal.d());
  if (rc) return rc;
  double hp[2 * 64];
  GPS_HIP(h, hipMemcpyAsync(hp, h->dScal.p, sizeof(hp), hipMemcpyDeviceToHost, h->stream));
  int linfo = 0;
  rc = read_info(h, d_info, &linfo);
  if (rc) return rc;
  if (info) *info = linfo;
  if (linfo) return GPS_OK;
  for (int b = 0; b < 64; ++b) { slog += hp[2 * b]; mahal += hp[2 * b + 1]; }
  if (q_sqrt_ndim == 2) {
    std::vector<double> kd;
    rc = kl_diag_kinv(h, bl, h->dS2.d(), mp, m, h->dTmp.d(), kd);
    if (rc) return rc;
    trace = 0.0;
    for (i64 j = 0; j < m; ++j) { double sq = 0.0; for (i64 q = 0; q < k; ++q) sq += q_sqrt[j * k + q] * q_sqrt[j * k + q]; trace += kd[j] * sq; }
  } else {
    trace = 0.0;
    for (i64 q = 0; q < k; ++q) {
      rc = upload_tril(h, q_sqrt + (size_t)q * m * m, m, h->dTmp2.d(), mp, 1.0, 1);
      if (rc) return rc;
      double t = 0.0;
      rc = kl_full_one(h, bl, h->dS2.d(), mp, m, h->dTmp2.d(), &t);
      if (rc) return rc;
      trace += t;
    }
  }
  *kl_out = 0.5 * (mahal - (double)(m * k) - logdet_q + trace + (double)k * 2.0 * slog);
  return GPS_OK;
  });
}


// ---- block-column distributed factorisation ------------------------------------------------------
// 1-D block-cyclic columns over P ranks.  Two storage modes (option "dist_partitioned"):
//   1 (default) PARTITIONED: a rank stores only the block columns it owns (c % P == rank), side by side in an
//     [np + 128, ncl * nb] buffer -- 8 N^2 / P bytes per rank (N = 32768, P = 8: 1.07 GB) -- and the trailing
//     updates read a received panel straight from the comm buffer it arrived in (>= 3 of them, slot = panel % count:
//     the bulk lane may still be reading panel p - 1 while panel p + 1 arrives).  The factor stays distributed:
//     predict_f streams the panels once more (gps_dist_solve_*), or the caller asks for the replicated mode.
//   0 REPLICATED: every rank holds an [np + 128, np] buffer and keeps every received panel in place, so that L ends
//     up on every rank and warm predict_f needs no further exchange (8 N^2 bytes per rank).
//
// Augmented rows ("alpha distributed"): rows np .. np+127 of the buffer hold (Y - m)^T (r real rows).  The
// panel solve  X L_jj^T = B  and the trailing update treat them like any other rows below the diagonal block, which
// is exactly the forward substitution: after panel j the augmented rows of block column j are alpha_j^T
// (alpha = L^-1 (Y - m), densities.py:82).  So there is no forward-substitution pass over a replicated factor at the
// end: the owner reduces  sum log L_ii  and  sum alpha^2  of its panel and ships them -- with its not-positive-definite
// info word -- in the tail of the panel message; every rank adds the tails in panel order, so LML and info are
// bit-identical on all ranks without a further collective.
//
// The library only provides the per-step pieces; the exchange itself (RCCL through torch.distributed, or gloo in the
// CPU tests) and the two-lane schedule (chain: receive / urgent columns / factor / send; bulk: the rest of each
// trailing update) are driven by gpflowSlim/distributed.py.
#define DIST_TAIL 4          // doubles at the end of a panel message: sum log L_ii, sum alpha^2, info, (spare)
static inline i64 dist_msg_doubles(gps_handle_t h, i64 j) {
  const i64 rows = h->dist_np + GPS_TILE - j * h->dist_nb;
  return rows * h->dist_nb + 2 * (h->dist_nb / GPS_TILE) * GPS_TILE * GPS_TILE + DIST_TAIL;
}

extern "C" int gps_set_stream(gps_handle_t h, void* hip_stream, int external) {
  if (!h) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  gps_profile_collect(h);
  if (external) {                       // hip_stream may be NULL: the legacy default stream
    if (!h->ext_stream) { h->own_stream = h->stream; h->ext_stream = true; }
    h->stream = (hipStream_t)hip_stream;
    // The look-ahead side stream comes from hipExtStreamCreateWithCUMask, which has no "non-blocking" flag: while it
    // exists, every launch on the legacy default stream is ordered against it (measured: the block-column run on the
    // default stream 2.6x slower).  It is not used with an external stream, so it goes; lookahead() re-creates it.
    if (h->side_stream) {
      (void)hipStreamSynchronize(h->side_stream);
      (void)hipStreamDestroy(h->side_stream);
      h->side_stream = nullptr;
    }
    if (h->def_stream) {
      (void)hipStreamSynchronize(h->def_stream);
      (void)hipStreamDestroy(h->def_stream);
      h->def_stream = nullptr;
    }
  } else if (h->ext_stream) {
    h->stream = h->own_stream; h->ext_stream = false;
  }
  return GPS_OK;
}

// diagnostics: replace the handle's own stream by one restricted to the CUs of `mask` (hipExtStreamCreateWithCUMask)
extern "C" int gps_diag_set_cu_mask(gps_handle_t h, const uint32_t* mask, int n_words) {
  if (!h || !mask || n_words <= 0 || h->ext_stream) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  gps_profile_collect(h);
  hipStream_t s = nullptr;
  GPS_HIP(h, hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask));
  (void)hipStreamDestroy(h->stream);
  h->stream = s;
  return GPS_OK;
}

extern "C" int gps_dist_begin(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double noise_var,
                              const double* resid, int64_t r, int nparts, int part, int64_t nb,
                              int64_t* n_panels, int64_t* msg_doubles_max) {
  if (!h || nparts <= 0 || part < 0 || part >= nparts || nb <= 0 || nb % GPS_TILE || r < 0 || r > GPS_TILE || (r > 0 && !resid))
    return gps_fail(h, GPS_ERR_ARG, "gps_dist_begin: bad argument (at most 128 outputs)");
  if (h->n <= 0) return gps_fail(h, GPS_ERR_STATE, "gps_gpr_set_data has not been called");
  GPS_HIP(h, hipSetDevice(h->device));
  const i64 n = h->n;
  const i64 np = ((n + nb - 1) / nb) * nb;
  const i64 nblk = np / nb;
  h->have_factor = false; h->dist_have_part_factor = false;
  h->npad = np; h->dist_np = np; h->dist_nb = nb; h->dist_P = nparts; h->dist_rank = part; h->dist_r = r;
  h->dist_part = h->dist_partitioned != 0;
  h->dist_ncl = part < nblk ? (nblk - 1 - part) / nparts + 1 : 0;            // owned block columns
  const i64 ld = h->dist_part ? (h->dist_ncl > 0 ? h->dist_ncl : 1) * nb : np;
  h->dist_ld = ld;
  {
    double kd = 0.0;
    int rck = gps_launch_kdiag(h, prog, n_nodes, &kd);
    if (rck) return rck;
    h->factor_refine = gps_gpr_needs_refine(h, noise_var, kd, h->n);
  }
  GPS_HIP(h, h->dK.ensure((size_t)(np + GPS_TILE) * ld * 8));
  GPS_HIP(h, h->dLinv.ensure(2 * (size_t)(np / GPS_TILE) * GPS_TILE * GPS_TILE * 8));
  GPS_HIP(h, h->dDistScal.ensure((size_t)nblk * DIST_TAIL * 8));
  GPS_HIP(h, hipEventRecord(h->ev[0], h->stream));
  // augmented rows: (Y - m)^T, zero padded to 128 rows (replicated mode: all columns, owned or not -- the bytes are few;
  // partitioned mode: the owned block columns, gathered from a transposed copy of the residual)
  double* aug = h->dK.d() + np * ld;
  GPS_HIP(h, hipMemsetAsync(aug, 0, (size_t)GPS_TILE * ld * 8, h->stream));
  if (r > 0) {
    GPS_HIP(h, h->dTmp2.ensure((size_t)n * r * 8));
    GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, resid, (size_t)n * r * 8, hipMemcpyHostToDevice, h->stream));
    if (!h->dist_part) {
      int rc0 = gps_launch_transpose(h, h->dTmp2.d(), r, n, r, aug, np);
      if (rc0) return rc0;
    } else {
      GPS_HIP(h, h->dAlpha.ensure((size_t)r * np * 8));
      GPS_HIP(h, hipMemsetAsync(h->dAlpha.p, 0, (size_t)r * np * 8, h->stream));
      int rc0 = gps_launch_transpose(h, h->dTmp2.d(), r, n, r, h->dAlpha.d(), np);
      if (rc0) return rc0;
      for (i64 lc = 0; lc < h->dist_ncl; ++lc)
        GPS_HIP(h, hipMemcpy2DAsync(aug + lc * nb, (size_t)ld * 8, h->dAlpha.d() + (lc * nparts + part) * nb, (size_t)np * 8,
                                    (size_t)nb * 8, (size_t)r, hipMemcpyDeviceToDevice, h->stream));
    }
  }
  int prep = 1;
  for (i64 c = part; c < nblk; c += nparts) {
    double* blk = h->dist_part ? h->dK.d() + c * nb * ld + (c / nparts) * nb : h->dK.d() + c * nb * np + c * nb;
    int rc = gps_launch_kmat_block(h, prog, n_nodes, h->dX.d(), n, h->d_all, np, noise_var, blk, ld, c * nb,
                                   np - c * nb, c * nb, nb, prep);
    if (rc) return rc;
    prep = 0;
  }
  GPS_HIP(h, hipEventRecord(h->ev[1], h->stream));
  int rc = gps_launch_fill_info(h, (int*)h->dInfo.p, INT_MAX);
  if (rc) return rc;
  if (n_panels) *n_panels = nblk;
  if (msg_doubles_max) *msg_doubles_max = dist_msg_doubles(h, 0);
  return GPS_OK;
}

extern "C" int gps_dist_msg_doubles(gps_handle_t h, int64_t j, int64_t* out) {
  if (!h || !out || h->dist_nb <= 0 || j < 0 || j * h->dist_nb >= h->dist_np) return gps_fail(h, GPS_ERR_ARG, "gps_dist_msg_doubles: bad argument");
  *out = dist_msg_doubles(h, j);
  return GPS_OK;
}

extern "C" int gps_dist_set_comm(gps_handle_t h, void* dev_buf0, void* dev_buf1) {
  void* bufs[2] = {dev_buf0, dev_buf1};
  return gps_dist_set_comm_bufs(h, bufs, 2);
}

extern "C" int gps_dist_set_comm_bufs(gps_handle_t h, void* const* dev_bufs, int count) {
  if (!h || !dev_bufs || count < 2 || count > 8) return gps_fail(h, GPS_ERR_ARG, "gps_dist_set_comm_bufs: 2 .. 8 buffers");
  for (int i = 0; i < count; ++i) if (!dev_bufs[i]) return gps_fail(h, GPS_ERR_ARG, "gps_dist_set_comm_bufs: null buffer");
  for (int i = 0; i < 8; ++i) h->dist_comm[i] = i < count ? (double*)dev_bufs[i] : nullptr;
  h->dist_ncomm = count;
  return GPS_OK;
}

extern "C" int gps_dist_comm_bufs_needed(gps_handle_t h, int* count) {
  if (!h || !count) return GPS_ERR_ARG;
  *count = h->dist_partitioned ? 3 : 2;
  return GPS_OK;
}

// second lane: gps_dist_update(..., lane = 1) launches on this stream instead of the handle's (no synchronisation
// here: the caller orders the lanes with events); NULL: one lane
extern "C" int gps_dist_set_bulk_stream(gps_handle_t h, void* hip_stream) {
  if (!h) return GPS_ERR_ARG;
  h->dist_bulk_stream = (hipStream_t)hip_stream;
  h->dist_bulk_set = (hip_stream != nullptr);
  return GPS_OK;
}

#define DIST_CHECK(h, j)                                                                              \
  if (!h || h->dist_nb <= 0 || j < 0 || j * h->dist_nb >= h->dist_np)                                 \
    return gps_fail(h, GPS_ERR_ARG, "gps_dist_*: bad panel index or gps_dist_begin not called");      \
  GPS_HIP(h, hipSetDevice(h->device));                                                                \
  h->refine_now = h->factor_refine;                                                                   \
  const i64 np = h->dist_np, nb = h->dist_nb, ld = h->dist_ld;                                        \
  const i64 rows = np + GPS_TILE - j * nb;              /* panel rows incl. the augmented ones */     \
  const i64 blk0 = j * nb / GPS_TILE, nbb = nb / GPS_TILE;                                            \
  /* the panel in this rank's storage (partitioned: only meaningful on the owner) */                  \
  double* const panel = h->dK.d() + j * nb * ld + (h->dist_part ? (j / h->dist_P) * nb : j * nb);      \
  double* const linv = h->dLinv.d();                                                                  \
  double* const linvT = linv + (np / GPS_TILE) * GPS_TILE * GPS_TILE;                                 \
  (void)rows; (void)blk0; (void)nbb; (void)panel; (void)linvT; (void)ld;

// owner of panel j: factor it in place (diagonal nb x nb block + rows below, augmented rows included), reduce its
// share of log-det / sum alpha^2, and pack the message
extern "C" int gps_dist_panel_factor(gps_handle_t h, int64_t j, int buf) {
  DIST_CHECK(h, j)
  if (buf < 0 || buf >= h->dist_ncomm || !h->dist_comm[buf]) return gps_fail(h, GPS_ERR_STATE, "gps_dist_set_comm has not been called");
  if (h->dist_part && (j % h->dist_P != h->dist_rank || buf != (int)(j % h->dist_ncomm)))
    return gps_fail(h, GPS_ERR_ARG, "gps_dist_panel_factor: partitioned storage -- not the owner, or not the panel's comm slot (panel % count)");
  HipOps ops{h, linv, linvT, (int*)h->dInfo.p};
  Blocked<HipOps> bl(ops);
  int rc = bl.potrf_rec(panel, ld, nb, blk0, j * nb);
  if (rc) return rc;
  rc = bl.trsm_rec(panel, ld, nb, blk0, panel + nb * ld, ld, rows - nb);
  if (rc) return rc;
  double* msg = h->dist_comm[buf];
  rc = gps_launch_extract(h, panel, ld, rows, nb, msg, nb, 0);
  if (rc) return rc;
  const size_t ib = (size_t)nbb * GPS_TILE * GPS_TILE * 8;
  double* tail = msg + rows * nb + 2 * nbb * GPS_TILE * GPS_TILE;
  GPS_HIP(h, hipMemcpyAsync(msg + rows * nb, linv + blk0 * GPS_TILE * GPS_TILE, ib, hipMemcpyDeviceToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(msg + rows * nb + nbb * GPS_TILE * GPS_TILE, linvT + blk0 * GPS_TILE * GPS_TILE, ib,
                            hipMemcpyDeviceToDevice, h->stream));
  // this panel's share of  sum log L_ii  and  sum alpha^2  (the augmented rows of this block column are alpha^T now),
  // folded with the info word into the message tail -- and into this rank's own per-panel table
  const double* aug = h->dK.d() + np * ld + (h->dist_part ? (j / h->dist_P) * nb : j * nb);
  rc = gps_launch_lml_reduce(h, panel, ld, nb, aug, ld, h->dist_r, h->dScal.d());
  if (rc) return rc;
  rc = gps_launch_dist_tail(h, h->dScal.d(), (const int*)h->dInfo.p, tail, h->dDistScal.d() + j * DIST_TAIL);
  if (rc) return rc;
  return GPS_OK;
}

// every other rank: copy the received panel (and its block inverses, and its scalars) into place
extern "C" int gps_dist_unpack(gps_handle_t h, int64_t j, int buf) {
  DIST_CHECK(h, j)
  if (buf < 0 || buf >= h->dist_ncomm || !h->dist_comm[buf]) return gps_fail(h, GPS_ERR_STATE, "gps_dist_set_comm has not been called");
  const double* msg = h->dist_comm[buf];
  if (h->dist_part) {
    // partitioned storage: the panel stays in its comm slot (the updates read it there); only its scalars are kept
    if (buf != (int)(j % h->dist_ncomm)) return gps_fail(h, GPS_ERR_ARG, "gps_dist_unpack: partitioned storage -- panel j lives in comm slot j % count");
    GPS_HIP(h, hipMemcpyAsync(h->dDistScal.d() + j * DIST_TAIL, msg + rows * nb + 2 * nbb * GPS_TILE * GPS_TILE, DIST_TAIL * 8,
                              hipMemcpyDeviceToDevice, h->stream));
    return GPS_OK;
  }
  int rc = gps_launch_extract(h, msg, nb, rows, nb, panel, np, 0);
  if (rc) return rc;
  const size_t ib = (size_t)nbb * GPS_TILE * GPS_TILE * 8;
  GPS_HIP(h, hipMemcpyAsync(linv + blk0 * GPS_TILE * GPS_TILE, msg + rows * nb, ib, hipMemcpyDeviceToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(linvT + blk0 * GPS_TILE * GPS_TILE, msg + rows * nb + nbb * GPS_TILE * GPS_TILE, ib,
                            hipMemcpyDeviceToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(h->dDistScal.d() + j * DIST_TAIL, msg + rows * nb + 2 * nbb * GPS_TILE * GPS_TILE, DIST_TAIL * 8,
                            hipMemcpyDeviceToDevice, h->stream));
  return GPS_OK;
}

// apply panel j to the owned block columns c in [c_lo, c_hi), c > j:  A[c*nb:, c] -= L[c*nb:, j] L[c, j]^T
// (rows down to and including the augmented ones).  lane 1: on the bulk stream (gps_dist_set_bulk_stream).
extern "C" int gps_dist_update(gps_handle_t h, int64_t j, int64_t c_lo, int64_t c_hi, int lane) {
  DIST_CHECK(h, j)
  const i64 nblk = np / nb;
  if (c_lo <= j) c_lo = j + 1;
  if (c_hi > nblk) c_hi = nblk;
  // owned column blocks in [c_lo, c_hi): first, first + P, ...  -> one lower-trapezoidal launch
  i64 first = c_lo + ((h->dist_rank - c_lo % h->dist_P) + h->dist_P) % h->dist_P;
  if (first >= c_hi) return GPS_OK;
  const i64 count = (c_hi - 1 - first) / h->dist_P + 1;
  hipStream_t saved = h->stream;
  if (lane == 1 && h->dist_bulk_set) h->stream = h->dist_bulk_stream;
  int rc;
  if (h->dist_part) {
    // panel j as it arrived (or was packed by its owner): [rows of panel j][nb] in comm slot j % count
    if (h->dist_ncomm < 3) { h->stream = saved; return gps_fail(h, GPS_ERR_STATE, "partitioned storage needs >= 3 comm buffers (gps_dist_set_comm_bufs)"); }
    const double* Lc = h->dist_comm[j % h->dist_ncomm] + (first - j) * nb * nb;
    double* C = h->dK.d() + first * nb * ld + (first / h->dist_P) * nb;
    rc = gps_launch_gemm_nt_cyclic(h, np + GPS_TILE - first * nb, count, nb, (i64)h->dist_P * nb, nb, Lc, nb, C, ld, 1);
  } else {
    const double* Lc = h->dK.d() + first * nb * np + j * nb;          // rows first*nb.. of panel j
    double* C = h->dK.d() + first * nb * np + first * nb;
    rc = gps_launch_gemm_nt_cyclic(h, np + GPS_TILE - first * nb, count, nb, (i64)h->dist_P * nb, nb, Lc, np, C, np);
  }
  h->stream = saved;
  return rc;
}

// after the last panel: add the per-panel scalars in panel order (identical on every rank), info = first failing pivot
extern "C" int gps_dist_finish(gps_handle_t h, double* lml, int* info) {
  if (!h || !lml || h->dist_nb <= 0) return gps_fail(h, GPS_ERR_ARG, "gps_dist_finish: bad argument");
  GPS_HIP(h, hipSetDevice(h->device));
  h->refine_now = h->factor_refine;
  const i64 n = h->n, np = h->dist_np, r = h->dist_r, nblk = np / h->dist_nb;
  GPS_HIP(h, hipEventRecord(h->ev[2], h->stream));
  // alpha [r][np] for warm predict_f: the augmented rows of the (replicated) factor
  if (r > 0 && !h->dist_part) {
    GPS_HIP(h, h->dAlpha.ensure((size_t)r * np * 8));
    GPS_HIP(h, hipMemcpyAsync(h->dAlpha.p, h->dK.d() + np * np, (size_t)r * np * 8, hipMemcpyDeviceToDevice, h->stream));
  }
  std::vector<double> tails((size_t)nblk * DIST_TAIL);
  GPS_HIP(h, hipMemcpyAsync(tails.data(), h->dDistScal.p, tails.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipEventRecord(h->ev[3], h->stream));
  int own = 0;
  int rc = read_info(h, (int*)h->dInfo.p, &own);          // (also surfaces look-ahead time-outs of this rank's panels)
  if (rc) return rc;
  double slog = 0.0, ssq = 0.0;
  int linfo = 0;
  for (i64 j = 0; j < nblk; ++j) {
    slog += tails[j * DIST_TAIL]; ssq += tails[j * DIST_TAIL + 1];
    const int pj = (int)tails[j * DIST_TAIL + 2];
    if (pj > 0 && (linfo == 0 || pj < linfo)) linfo = pj;
  }
  if (info) *info = linfo;
  *lml = -0.5 * (double)n * (double)r * log(2.0 * M_PI) - (double)r * slog - 0.5 * ssq;
  h->r = r;
  h->have_factor = (linfo == 0) && !h->dist_part;          // a partitioned factor serves gps_dist_solve_* only
  h->dist_have_part_factor = (linfo == 0) && h->dist_part;
  stage_time(h, 0, 1, &h->stage_ms[0]);
  stage_time(h, 1, 2, &h->stage_ms[1]);
  stage_time(h, 2, 3, &h->stage_ms[2]);
  h->stage_ms[3] = 0.0;
  stage_time(h, 0, 3, &h->stage_ms[4]);
  return GPS_OK;
}

// ---- the whole block-column factorisation driven from here (no host language in the panel loop) ------------------------
// gpflowSlim/distributed.py::block_column_schedule, statement for statement (that Python function stays the specification:
// it is what the vector-clock race detector of tests/test_dist_cpu.py validates), with the per-step pieces above, the
// handle's native communicator (comm_rccl.hip) for the exchange and HIP streams / events for the two lanes:
//   CHAIN lane: urgent updates, panel factorisations, exchanges (a high-priority stream installed as the handle's stream)
//   BULK  lane: the rest of every trailing update (a low-priority stream)
// lookahead = D: panel p's update of columns p+1 .. p+D runs on the CHAIN lane, the rest on the BULK lane; D = 0: one lane.
extern "C" int gps_dist_lml(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double noise_var, const double* resid,
                            int64_t r, int64_t nb, int lookahead, int exchange_mode, double* lml, int* info) {
  if (!h || !lml) return gps_fail(h, GPS_ERR_ARG, "gps_dist_lml: bad argument");
  if (!h->comm) return gps_fail(h, GPS_ERR_STATE, "gps_dist_lml: the handle has no communicator (gps_comm_init)");
  if (h->ext_stream) return gps_fail(h, GPS_ERR_STATE, "gps_dist_lml: an external stream is installed (gps_set_stream)");
  GPS_HIP(h, hipSetDevice(h->device));
  const int P = h->comm_world, rank = h->comm_rank, D = lookahead < 0 ? 0 : lookahead;
  // the two lanes and the events of the schedule belong to the HANDLE: created on the first call, reused by every later one
  // (an evaluation of a fit creates no stream and no event), destroyed by gps_destroy
  if (!h->dist_chain) {
    int lo = 0, hi = 0;
    GPS_HIP(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t c = nullptr, b = nullptr;
    GPS_HIP(h, hipStreamCreateWithPriority(&c, hipStreamNonBlocking, hi));
    hipError_t e = hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo);
    if (e != hipSuccess) { (void)hipStreamDestroy(c); return gps_fail(h, GPS_ERR_HIP, std::string("gps_dist_lml: bulk lane: ") + hipGetErrorString(e)); }
    h->dist_chain = c; h->dist_bulk_own = b;
  }
  hipStream_t chain = h->dist_chain, bulk = D >= 1 ? h->dist_bulk_own : nullptr;
  h->dist_event_next = 0;
  auto new_event = [&]() -> hipEvent_t {
    if (h->dist_event_next < h->dist_events.size()) return h->dist_events[h->dist_event_next++];
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    h->dist_events.push_back(e); h->dist_event_next++;
    return e;
  };
  // (inside this function every failure leaves through cleanup(): the handle must get its own stream back)
#define DL_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return dl_fail(std::string(#call) + ": " + hipGetErrorString(e__)); } while (0)
  int rc = gps_set_stream(h, chain, 1);
  auto dl_fail = [&](const std::string& msg) -> int { return gps_fail(h, GPS_ERR_HIP, msg); };
  auto cleanup = [&](int code) -> int {
    // both lanes drained on every exit path, the handle's own stream back
    (void)hipStreamSynchronize(chain);
    if (bulk) (void)hipStreamSynchronize(bulk);
    if (h->comm && h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    (void)gps_dist_set_bulk_stream(h, nullptr);
    (void)gps_set_stream(h, nullptr, 0);
    return code;
  };
  if (rc) return cleanup(rc);
  rc = gps_dist_set_bulk_stream(h, bulk);
  if (rc) return cleanup(rc);
  int64_t n_panels = 0, mx = 0;
  rc = gps_dist_begin(h, prog, n_nodes, noise_var, resid, r, P, rank, nb, &n_panels, &mx);
  if (rc) return cleanup(rc);
  const int nbufs = h->dist_partitioned ? 3 : 2;
  const i64 cap = ((mx + P - 1) / P) * P;                       // whole chunks for the scatter + all-gather
  void* bufs[3] = {nullptr, nullptr, nullptr};
  for (int b = 0; b < nbufs; ++b) {
    hipError_t e = h->dDistComm[b].ensure((size_t)cap * 8);
    if (e != hipSuccess) return cleanup(gps_fail(h, GPS_ERR_HIP, "gps_dist_lml: comm buffer allocation failed"));
    bufs[b] = h->dDistComm[b].p;
  }
  rc = gps_dist_set_comm_bufs(h, bufs, nbufs);
  if (rc) return cleanup(rc);
  auto owner = [&](i64 t) { return (int)(t % P); };
  auto exchange = [&](i64 t, int buf) -> int {                  // returns rc; the slot is t % 8
    const i64 n = dist_msg_doubles(h, t);
    return gps_comm_exchange(h, bufs[buf], ((n + P - 1) / P) * P, owner(t), exchange_mode, (int)(t % 8));
  };
  auto receive = [&](i64 t, int buf) -> int {
    int rcc = gps_comm_wait(h, (int)(t % 8));
    if (rcc) return rcc;
    return rank != owner(t) ? gps_dist_unpack(h, t, buf) : GPS_OK;
  };
#define GPS_TRY(call) do { rc = (call); if (rc) return cleanup(rc); } while (0)
  if (rank == owner(0)) GPS_TRY(gps_dist_panel_factor(h, 0, 0));
  GPS_TRY(exchange(0, 0));
  GPS_TRY(receive(0, 0));
  std::vector<hipEvent_t> bulk_done((size_t)n_panels, nullptr);
  for (i64 p = 0; p + 1 < n_panels; ++p) {
    const i64 nxt = p + 1; const int buf = (int)(nxt % nbufs);
    if (D == 0) {
      GPS_TRY(gps_dist_update(h, p, nxt, n_panels, 0));
      if (rank == owner(nxt)) GPS_TRY(gps_dist_panel_factor(h, nxt, buf));
      GPS_TRY(exchange(nxt, buf));
      GPS_TRY(receive(nxt, buf));
      continue;
    }
    hipEvent_t in_place = new_event();
    if (!in_place) return cleanup(gps_fail(h, GPS_ERR_HIP, "gps_dist_lml: hipEventCreate failed"));
    GPS_TRY([&]() -> int { DL_HIP(hipEventRecord(in_place, chain)); return GPS_OK; }());
    const i64 last_urgent = (p + D < n_panels - 1) ? p + D : n_panels - 1;
    auto urgent = [&](i64 c) -> int {
      // first CHAIN update of column c = p + D: the BULK updates of panels <= p - 1 may still be running on it
      if (c == p + D && p >= 1 && bulk_done[p - 1]) { DL_HIP(hipStreamWaitEvent(chain, bulk_done[p - 1], 0)); bulk_done[p - 1] = nullptr; }
      return gps_dist_update(h, p, c, c + 1, 0);
    };
    GPS_TRY(urgent(nxt));
    if (rank == owner(nxt)) GPS_TRY(gps_dist_panel_factor(h, nxt, buf));
    GPS_TRY(exchange(nxt, buf));                                  // in flight while ...
    for (i64 c = nxt + 1; c <= last_urgent; ++c) GPS_TRY(urgent(c));   // ... the other urgent columns
    if (last_urgent + 1 < n_panels) {                            // ... and the bulk of the update run
      GPS_TRY([&]() -> int { DL_HIP(hipStreamWaitEvent(bulk, in_place, 0)); return GPS_OK; }());
      GPS_TRY(gps_dist_update(h, p, last_urgent + 1, n_panels, 1));
      bulk_done[p] = new_event();
      if (!bulk_done[p]) return cleanup(gps_fail(h, GPS_ERR_HIP, "gps_dist_lml: hipEventCreate failed"));
      GPS_TRY([&]() -> int { DL_HIP(hipEventRecord(bulk_done[p], bulk)); return GPS_OK; }());
    }
    GPS_TRY(receive(nxt, buf));
  }
  for (hipEvent_t e : bulk_done) if (e) GPS_TRY([&]() -> int { DL_HIP(hipStreamWaitEvent(chain, e, 0)); return GPS_OK; }());
  int linfo = 0;
  rc = gps_dist_finish(h, lml, &linfo);
  if (info) *info = linfo;
#undef GPS_TRY
#undef DL_HIP
  return cleanup(rc);
}

// ---- predict_f from a PARTITIONED factor: the panels are streamed once more (models/gpr.py:119-131) -------------------
// Every rank holds a shard of the test points and solves  A^T = Kx^T L^-T  for it panel by panel as the panels come by
// (forward substitution at panel granularity: block column j of A^T is final after panel j, the columns to its right
// take its update); the augmented rows of each panel message are alpha_j^T, so alpha assembles itself on every rank.
//   gps_dist_solve_begin(Xnew shard)     Kx^T = K(Xnew, X) [n*, np], alpha <- 0
//   for j in panels:  owner: gps_dist_solve_pack(j, buf) ; exchange (same message as the factorisation's) ;
//                     all:   gps_dist_solve_apply(j, buf)
//   gps_dist_solve_finish(mean, var)     fmean = A^T alpha ; fvar = Kdiag - rowsum((A^T)^2)   (full_cov == 0)
extern "C" int gps_dist_solve_begin(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Xnew, int64_t n_new) {
  if (!h || !Xnew || n_new <= 0) return gps_fail(h, GPS_ERR_ARG, "gps_dist_solve_begin: bad argument");
  if (!h->dist_have_part_factor || h->dist_nb <= 0) return gps_fail(h, GPS_ERR_STATE, "gps_dist_solve_begin: no partitioned factor (run the distributed factorisation first)");
  GPS_HIP(h, hipSetDevice(h->device));
  h->refine_now = h->factor_refine;
  const i64 n = h->n, np = h->dist_np, d = h->d_all, r = h->dist_r;
  const i64 nsp = gps_pad(n_new);
  h->dist_solve_n = n_new;
  GPS_HIP(h, h->dXnew.ensure((size_t)n_new * d * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dXnew.p, Xnew, (size_t)n_new * d * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dB.ensure((size_t)nsp * np * 8));
  int rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n_new, h->dX.d(), n, d, 0.0, h->dB.d(), np, nsp, np, 0, 0);
  if (rc) return rc;
  GPS_HIP(h, h->dAlpha.ensure((size_t)(r > 0 ? r : 1) * np * 8));
  GPS_HIP(h, hipMemsetAsync(h->dAlpha.p, 0, (size_t)(r > 0 ? r : 1) * np * 8, h->stream));
  return GPS_OK;
}

extern "C" int gps_dist_solve_pack(gps_handle_t h, int64_t j, int buf) {
  DIST_CHECK(h, j)
  if (!h->dist_have_part_factor) return gps_fail(h, GPS_ERR_STATE, "gps_dist_solve_pack: no partitioned factor");
  if (buf < 0 || buf >= h->dist_ncomm || !h->dist_comm[buf]) return gps_fail(h, GPS_ERR_STATE, "gps_dist_set_comm has not been called");
  if (j % h->dist_P != h->dist_rank) return gps_fail(h, GPS_ERR_ARG, "gps_dist_solve_pack: not the owner of this panel");
  double* msg = h->dist_comm[buf];
  int rc = gps_launch_extract(h, panel, ld, rows, nb, msg, nb, 0);
  if (rc) return rc;
  const size_t ib = (size_t)nbb * GPS_TILE * GPS_TILE * 8;
  GPS_HIP(h, hipMemcpyAsync(msg + rows * nb, linv + blk0 * GPS_TILE * GPS_TILE, ib, hipMemcpyDeviceToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(msg + rows * nb + nbb * GPS_TILE * GPS_TILE, linvT + blk0 * GPS_TILE * GPS_TILE, ib,
                            hipMemcpyDeviceToDevice, h->stream));
  GPS_HIP(h, hipMemsetAsync(msg + rows * nb + 2 * nbb * GPS_TILE * GPS_TILE, 0, DIST_TAIL * 8, h->stream));
  return GPS_OK;
}

extern "C" int gps_dist_solve_apply(gps_handle_t h, int64_t j, int buf) {
  DIST_CHECK(h, j)
  if (!h->dist_have_part_factor || h->dist_solve_n <= 0) return gps_fail(h, GPS_ERR_STATE, "gps_dist_solve_apply: gps_dist_solve_begin has not been called");
  if (buf < 0 || buf >= h->dist_ncomm || !h->dist_comm[buf]) return gps_fail(h, GPS_ERR_STATE, "gps_dist_set_comm has not been called");
  double* msg = h->dist_comm[buf];
  const i64 nsp = gps_pad(h->dist_solve_n);
  double* Bj = h->dB.d() + j * nb;                                   // [nsp, nb] block column j of Kx^T / A^T, ld np
  // B_j <- B_j L_jj^-T  (the panel's own block inverses travel with it)
  HipOps ops{h, msg + rows * nb, msg + rows * nb + nbb * GPS_TILE * GPS_TILE, (int*)h->dInfo.p};
  Blocked<HipOps> bl(ops);
  int rc = bl.trsm_rec(msg, nb, nb, 0, Bj, np, nsp);
  if (rc) return rc;
  // B_{>j} -= B_j L[>j, j]^T
  const i64 below = np - (j + 1) * nb;
  if (below > 0) {
    rc = gps_launch_gemm_nt(h, 0, 0, nsp, below, nb, Bj, np, msg + nb * nb, nb, Bj + nb, np);
    if (rc) return rc;
  }
  // alpha_j^T: the augmented rows of the panel
  if (h->dist_r > 0)
    GPS_HIP(h, hipMemcpy2DAsync(h->dAlpha.d() + j * nb, (size_t)np * 8, msg + (rows - GPS_TILE) * nb, (size_t)nb * 8, (size_t)nb * 8,
                                (size_t)h->dist_r, hipMemcpyDeviceToDevice, h->stream));
  return GPS_OK;
}

extern "C" int gps_dist_solve_finish(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double* mean_out, double* var_out) {
  if (!h || !var_out || h->dist_solve_n <= 0 || !h->dist_have_part_factor) return gps_fail(h, GPS_ERR_STATE, "gps_dist_solve_finish: nothing to finish");
  if (h->dist_r > 0 && !mean_out) return gps_fail(h, GPS_ERR_ARG, "gps_dist_solve_finish: mean_out missing");
  GPS_HIP(h, hipSetDevice(h->device));
  const i64 np = h->dist_np, r = h->dist_r, n_new = h->dist_solve_n;
  GPS_HIP(h, h->dMean.ensure((size_t)(n_new * (r > 0 ? r : 1) + n_new) * 8));
  double* dmean = h->dMean.d();
  double* dss = dmean + n_new * (r > 0 ? r : 1);
  int rc = gps_launch_rowdot(h, h->dB.d(), np, n_new, np, h->dAlpha.d(), np, r, dmean, dss);
  if (rc) return rc;
  double kd = 0.0;
  rc = gps_launch_kdiag(h, prog, n_nodes, &kd);
  if (rc) return rc;
  GPS_HIP(h, h->dVar.ensure((size_t)n_new * 8));
  rc = gps_launch_var_finish(h, h->dVar.d(), nullptr, kd, dss, n_new);
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(var_out, h->dVar.p, (size_t)n_new * 8, hipMemcpyDeviceToHost, h->stream));
  if (r > 0) GPS_HIP(h, hipMemcpyAsync(mean_out, dmean, (size_t)n_new * r * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  h->dist_solve_n = 0;
  return GPS_OK;
}

// predict_f for this rank's shard of the test points from the partitioned factor gps_dist_lml left behind: gps_dist_solve_*
// with the native communicator, the exchange of panel j + 1 in flight while panel j is applied (gpflowSlim/distributed.py::
// predict_streamed, statement for statement).  n_new may be 0 (the rank still takes part in the exchanges).
extern "C" int gps_dist_predict(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Xnew, int64_t n_new,
                                int exchange_mode, double* mean_out, double* var_out) {
  if (!h || n_new < 0 || (n_new > 0 && (!Xnew || !var_out))) return gps_fail(h, GPS_ERR_ARG, "gps_dist_predict: bad argument");
  if (!h->comm) return gps_fail(h, GPS_ERR_STATE, "gps_dist_predict: the handle has no communicator (gps_comm_init)");
  if (!h->dist_have_part_factor || h->dist_nb <= 0) return gps_fail(h, GPS_ERR_STATE, "gps_dist_predict: no partitioned factor (gps_dist_lml first)");
  GPS_HIP(h, hipSetDevice(h->device));
  const int P = h->comm_world, rank = h->comm_rank;
  const i64 n_panels = h->dist_np / h->dist_nb;
  const i64 cap = ((dist_msg_doubles(h, 0) + P - 1) / P) * P;
  void* bufs[2];
  for (int b = 0; b < 2; ++b) { GPS_HIP(h, h->dDistComm[b].ensure((size_t)cap * 8)); bufs[b] = h->dDistComm[b].p; }
  int rc = gps_dist_set_comm_bufs(h, bufs, 2);
  if (rc) return rc;
  if (n_new > 0) { rc = gps_dist_solve_begin(h, prog, n_nodes, Xnew, n_new); if (rc) return rc; }
  auto send = [&](i64 j) -> int {
    const int buf = (int)(j % 2);
    if (rank == (int)(j % P)) { int rcc = gps_dist_solve_pack(h, j, buf); if (rcc) return rcc; }
    const i64 n = dist_msg_doubles(h, j);
    return gps_comm_exchange(h, bufs[buf], ((n + P - 1) / P) * P, (int)(j % P), exchange_mode, (int)(j % 8));
  };
  rc = send(0);
  for (i64 j = 0; j < n_panels && !rc; ++j) {
    rc = gps_comm_wait(h, (int)(j % 8));
    if (!rc && j + 1 < n_panels) rc = send(j + 1);            // (stream-ordered after apply(j - 1), the last reader of that buffer)
    if (!rc && n_new > 0) rc = gps_dist_solve_apply(h, j, (int)(j % 2));
  }
  if (rc) { (void)hipStreamSynchronize(h->stream); if (h->comm && h->comm_stream) (void)hipStreamSynchronize(h->comm_stream); return rc; }
  if (n_new > 0) return gps_dist_solve_finish(h, prog, n_nodes, mean_out, var_out);
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

// device memory the handle holds right now: every growable buffer, the comm buffers of the all-native driver (gps_dist_lml /
// gps_dist_predict allocate them on the handle) included; comm buffers a caller brings (gps_dist_set_comm_bufs) are its own
extern "C" int gps_device_bytes(gps_handle_t h, int64_t* bytes) {
  if (!h || !bytes) return GPS_ERR_ARG;
  DevBuf* bufs[] = {&h->dX, &h->dK, &h->dLinv, &h->dAlpha, &h->dFeat, &h->dFeat2, &h->dProg,
                    &h->dXnew, &h->dB, &h->dMean, &h->dVar, &h->dTmp, &h->dTmp2, &h->dTmp3, &h->dA, &h->dY,
                    &h->dKinv, &h->dNkn, &h->dS1, &h->dS2, &h->dS3, &h->dS4, &h->dGemvWs, &h->dGemvCnt, &h->dGemmWs, &h->dGemmCnt,
                    &h->dDistScal, &h->dGradSums, &h->dSmallOut, &h->dFeatG, &h->dG1, &h->dG2, &h->dG3, &h->dG4, &h->dWave, &h->dInfo, &h->dScal, &h->dWaveCtl, &h->dLaFlags,
                    &h->dDistComm[0], &h->dDistComm[1], &h->dDistComm[2]};
  int64_t tot = 0;
  for (DevBuf* b : bufs) tot += (int64_t)b->cap;
  *bytes = tot;
  return GPS_OK;
}


// ---- SGPR (Titsias 2009): bound and prediction ---------------------------------------------------------
// models/sgpr.py:121-153 (_build_likelihood) and :155-189 (_build_predict).  Everything O(M^2 N) runs on the
// device: Kuu potrf, (L^-1 Kuf)^T by trsm_rec, A A^T as one long-K NT GEMM, second potrf, solves.
// Shared by gps_sgpr (fitc == 0: every data point weighs 1/sigma^2) and gps_fitc (fitc == 1: point i weighs
// 1/nu_i, nu_i = Kdiag_i - Qff_ii + sigma^2, sgpr.py:232-250).  With W = rows of (L^-1 Kuf)^T scaled by sqrt(weight)
// both are  B = I + W^T W,  c = LB^-1 W^T (err * sqrt(weight)).
static int sparse_gpr_impl(gps_handle_t h, int fitc, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                           const double* X, int64_t n, int64_t d_all, double jitter, double noise_var,
                           const double* resid, int64_t r, const double* Xnew, int64_t n_new, int full_cov,
                           double* bound_out, double* mean_out, double* var_out, int* info) {
  if (!h || !Z || !X || !resid || m <= 0 || n <= 0 || d_all <= 0 || r <= 0 || !(noise_var > 0.0))
    return gps_fail(h, GPS_ERR_ARG, "gps_sgpr: bad argument");
  if (n_new > 0 && (!Xnew || !mean_out || !var_out)) return gps_fail(h, GPS_ERR_ARG, "gps_sgpr: prediction outputs missing");
  GPS_HIP(h, hipSetDevice(h->device));
  if (info) *info = 0;
  h->have_factor = false; h->dist_have_part_factor = false; h->n = 0;                       // GPR resident buffers are reused below
  h->refine_now = (h->leaf_refine != 0);
  const i64 mp = gps_pad(m), np = gps_pad(n);
  const size_t blk_bytes = (size_t)(mp / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  const double sigma2 = noise_var;
  // buffers: dX <- Z ; dXnew <- X (then Xnew) ; dK <- Kuu/L ; dLinv (2 sets for L) ; dS1 <- At [np, mp] ;
  //          dS2 <- A [mp, np] ; dS3 <- B / LB [mp, mp] ; dS4 <- inverses of LB (2 sets)
  GPS_HIP(h, h->dX.ensure((size_t)m * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dX.p, Z, (size_t)m * d_all * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dXnew.ensure((size_t)(n > n_new ? n : n_new) * d_all * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dXnew.p, X, (size_t)n * d_all * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dK.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dLinv.ensure(2 * blk_bytes));
  GPS_HIP(h, h->dS1.ensure((size_t)np * mp * 8));
  GPS_HIP(h, h->dS2.ensure((size_t)mp * np * 8));
  GPS_HIP(h, h->dS3.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dS4.ensure(2 * blk_bytes));
  int* d_info = (int*)h->dInfo.p;
  int rc = gps_launch_fill_info(h, d_info, INT_MAX);
  if (rc) return rc;
  // Kuu + jitter I -> L                                                 (features.py:74-77, sgpr.py:133-135)
  rc = gps_launch_kmat(h, prog, n_nodes, h->dX.d(), m, nullptr, m, d_all, jitter, h->dK.d(), mp, mp, mp, 1, 1);
  if (rc) return rc;
  HipOps opsL{h, h->dLinv.d(), h->dLinv.d() + blk_bytes / 8, d_info};
  Blocked<HipOps> blL(opsL);
  rc = blL.potrf_rec(h->dK.d(), mp, mp, 0, 0);
  if (rc) return rc;
  rc = classify_blocks(h, opsL, h->dK.d(), mp, mp);
  if (rc) return rc;
  // At = K(X, Z) L^-T  = (L^-1 Kuf)^T   [np, mp]                         (sgpr.py:139, without the 1/sigma)
  rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n, h->dX.d(), m, d_all, 0.0, h->dS1.d(), mp, np, mp, 0, 0);
  if (rc) return rc;
  rc = blL.trsm_rec(h->dK.d(), mp, mp, 0, h->dS1.d(), mp, np);
  if (rc) return rc;
  double kdiag = 0.0;
  rc = gps_launch_kdiag(h, prog, n_nodes, &kdiag);
  if (rc) return rc;
  std::vector<double> wsq;                                       // FITC: 1/sqrt(nu_i)
  double sum_log_nu = 0.0, shard_err = 0.0;
  if (fitc) {
    // diag Qff = rowsumsq((L^-1 Kuf)^T) ; nu = Kdiag - diag Qff + sigma^2          (sgpr.py:241-242)
    GPS_HIP(h, h->dTmp3.ensure((size_t)np * 8));
    rc = gps_launch_rowdot(h, h->dS1.d(), mp, n, mp, nullptr, mp, 0, nullptr, h->dTmp3.d());
    if (rc) return rc;
    wsq.assign((size_t)np, 0.0);
    GPS_HIP(h, hipMemcpyAsync(wsq.data(), h->dTmp3.p, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    for (i64 i = 0; i < n; ++i) {
      double nu = kdiag - wsq[i] + sigma2;
      // a failure that depends on this rank's data must not leave before the collective below (the peers would wait in it
      // for ever): carry on with a harmless value, send the flag along, fail on EVERY rank after the reduction
      if (!(nu > 0.0)) { shard_err = 1.0; nu = 1.0; }
      sum_log_nu += log(nu);
      wsq[i] = 1.0 / sqrt(nu);
    }
    if (shard_err != 0.0 && !h->allreduce) return gps_fail(h, GPS_ERR_ARG, "gps_fitc: non-positive FITC variance nu");
    GPS_HIP(h, hipMemcpyAsync(h->dTmp3.p, wsq.data(), (size_t)np * 8, hipMemcpyHostToDevice, h->stream));
    rc = gps_launch_scale_rows(h, h->dS1.d(), mp, n, mp, h->dTmp3.d());
    if (rc) return rc;
  }
  const double wgt = fitc ? 1.0 : 1.0 / sigma2;                  // what multiplies A A^T and the c terms
  // A = At^T [mp, np] ; A A^T (lower)                                         (sgpr.py:140-141, 244)
  rc = gps_launch_transpose(h, h->dS1.d(), mp, np, mp, h->dS2.d(), np);
  if (rc) return rc;
  rc = gps_launch_gemm_nt(h, 1, 1, mp, mp, np, h->dS2.d(), np, h->dS2.d(), np, h->dS3.d(), mp);
  if (rc) return rc;
  // Aerr*sigma = (L^-1 Kuf) err [m, r] and rowsumsq(A*sigma) = sigma^2 diag(AAT)   (sgpr.py:143, 152)
  GPS_HIP(h, h->dAlpha.ensure((size_t)r * (np > mp ? np : mp) * 8 * 2));
  double* dErrT = h->dAlpha.d();                                 // [r][np]
  double* dC = dErrT + (size_t)r * np;                           // [r][mp]
  GPS_HIP(h, h->dTmp2.ensure((size_t)n * r * 8));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, resid, (size_t)n * r * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemsetAsync(dErrT, 0, (size_t)r * np * 8, h->stream));
  rc = gps_launch_transpose(h, h->dTmp2.d(), r, n, r, dErrT, np);
  if (rc) return rc;
  if (fitc) {                                                    // beta * sqrt(nu) = err / sqrt(nu)
    rc = gps_launch_scale_cols(h, dErrT, np, r, n, h->dTmp3.d(), dErrT, np);
    if (rc) return rc;
  }
  GPS_HIP(h, h->dMean.ensure((size_t)(mp * r + mp) * 8 + (size_t)(n_new > 0 ? (n_new * r + 2 * n_new) * 8 : 0)));
  double* dAerr = h->dMean.d();                                  // [m][r]
  double* dDiag = dAerr + (size_t)mp * r;                        // [m]
  rc = gps_launch_rowdot(h, h->dS2.d(), np, m, np, dErrT, np, r, dAerr, dDiag);
  if (rc) return rc;
  // sum err^2 (FITC: err^2 / nu) over the data points of this call
  double serr2 = 0.0;
  if (fitc) {
    for (i64 i = 0; i < n; ++i) for (i64 q = 0; q < r; ++q) { const double e = resid[i * r + q] * wsq[i]; serr2 += e * e; }
  } else {
    for (i64 i = 0; i < n * r; ++i) serr2 += resid[i] * resid[i];
  }
  double n_total = (double)n;
  if (h->allreduce) {
    // X / resid were this rank's shard: everything above that sums over data points is a partial sum.  Pack
    // [A A^T (lower; the rest of the square is never read) | A err | diag | sum err^2, sum log nu, n], one all-reduce, unpack.
    const i64 cnt = mp * mp + mp * r + mp + 4;
    if (cnt > h->red_cap) return gps_fail(h, GPS_ERR_ARG, "gps_set_allreduce: the device buffer is too small for this m, r");
    double* rb = h->red_buf;
    const double sc[4] = {serr2, sum_log_nu, (double)n, shard_err};        // [3]: data-dependent failures of the shards, summed
    GPS_HIP(h, hipMemcpyAsync(rb, h->dS3.p, (size_t)mp * mp * 8, hipMemcpyDeviceToDevice, h->stream));
    GPS_HIP(h, hipMemcpyAsync(rb + mp * mp, dAerr, (size_t)(mp * r + mp) * 8, hipMemcpyDeviceToDevice, h->stream));
    GPS_HIP(h, hipMemcpyAsync(rb + mp * mp + mp * r + mp, sc, sizeof(sc), hipMemcpyHostToDevice, h->stream));
    rc = gps_launch_tri_map(h, rb, mp, mp, 0);          // mirror: what the lower-triangular GEMM left untouched would be summed as stale bytes
    if (rc) return rc;
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    if (h->allreduce(h->allreduce_ctx, rb, cnt) != 0) return gps_fail(h, GPS_ERR_STATE, "the all-reduce callback failed");
    double sc_out[4];
    GPS_HIP(h, hipMemcpyAsync(h->dS3.p, rb, (size_t)mp * mp * 8, hipMemcpyDeviceToDevice, h->stream));
    GPS_HIP(h, hipMemcpyAsync(dAerr, rb + mp * mp, (size_t)(mp * r + mp) * 8, hipMemcpyDeviceToDevice, h->stream));
    GPS_HIP(h, hipMemcpyAsync(sc_out, rb + mp * mp + mp * r + mp, sizeof(sc_out), hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    serr2 = sc_out[0]; sum_log_nu = sc_out[1]; n_total = sc_out[2];
    if (sc_out[3] != 0.0) return gps_fail(h, GPS_ERR_ARG, "gps_fitc: non-positive FITC variance nu (on at least one shard)");
  }
  // B = A A^T * weight + I ; LB = chol(B)                                     (sgpr.py:141-142, 244-245)
  rc = gps_launch_scale_add_eye(h, h->dS3.d(), mp, mp, m, wgt);
  if (rc) return rc;
  HipOps opsB{h, h->dS4.d(), h->dS4.d() + blk_bytes / 8, d_info};
  Blocked<HipOps> blB(opsB);
  rc = blB.potrf_rec(h->dS3.d(), mp, mp, 0, 0);
  if (rc) return rc;
  // c = LB^-1 Aerr / sigma : c*sigma^2 = LB^-1 (Aerr*sigma)              (sgpr.py:144)
  GPS_HIP(h, hipMemsetAsync(dC, 0, (size_t)r * mp * 8, h->stream));
  rc = gps_launch_transpose(h, dAerr, r, m, r, dC, mp);
  if (rc) return rc;
  rc = blB.trsv_rec(h->dS3.d(), mp, mp, 0, dC, mp, r);
  if (rc) return rc;
  // reductions: sum log diag LB, sum (c sigma^2)^2
  double* part = h->dScal.d();
  rc = gps_launch_lml_reduce(h, h->dS3.d(), mp, m, dC, mp, r, part);
  if (rc) return rc;
  double hp[2 * 64];
  GPS_HIP(h, hipMemcpyAsync(hp, part, sizeof(hp), hipMemcpyDeviceToHost, h->stream));
  std::vector<double> hdiag(m);
  GPS_HIP(h, hipMemcpyAsync(hdiag.data(), dDiag, (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
  int linfo = 0;
  rc = read_info(h, d_info, &linfo);
  if (rc) return rc;
  if (info) *info = linfo;
  if (linfo) return GPS_OK;
  double slogLB = 0.0, sc2 = 0.0, trAAT = 0.0;
  for (int b = 0; b < 64; ++b) { slogLB += hp[2 * b]; sc2 += hp[2 * b + 1]; }
  for (i64 i = 0; i < m; ++i) trAAT += hdiag[i];
  h->sparse_terms[0] = slogLB; h->sparse_terms[1] = trAAT; h->sparse_terms[2] = sc2 * wgt * wgt;
  h->sparse_terms[3] = kdiag; h->sparse_terms[4] = sum_log_nu;
  trAAT *= wgt;
  sc2 *= wgt * wgt;                                              // SGPR: c = (c sigma^2) / sigma^2
  if (bound_out && fitc) {
    const double N = n_total, R = (double)r;                     // sgpr.py:256-290
    *bound_out = -0.5 * serr2 + 0.5 * sc2 + R * (-0.5 * N * log(2.0 * M_PI) - 0.5 * sum_log_nu - slogLB);
  } else if (bound_out) {
    const double N = n_total, R = (double)r;
    double bound = -0.5 * N * R * log(2.0 * M_PI);               // sgpr.py:147-153
    bound += -R * slogLB;
    bound -= 0.5 * N * R * log(sigma2);
    bound += -0.5 * serr2 / sigma2;
    bound += 0.5 * sc2;
    bound += -0.5 * R * (N * kdiag) / sigma2;
    bound += 0.5 * R * trAAT;
    *bound_out = bound;
  }
  if (n_new <= 0) return GPS_OK;
  // ---- prediction                                                      (sgpr.py:155-189)
  const i64 nsp = gps_pad(n_new);
  GPS_HIP(h, hipMemcpyAsync(h->dXnew.p, Xnew, (size_t)n_new * d_all * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, h->dB.ensure((size_t)nsp * mp * 8 * 2));
  double* T1 = h->dB.d();                                         // tmp1^T [nsp, mp]
  double* T2 = T1 + (size_t)nsp * mp;                             // tmp2^T
  rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n_new, h->dX.d(), m, d_all, 0.0, T1, mp, nsp, mp, 0, 0);
  if (rc) return rc;
  rc = blL.trsm_rec(h->dK.d(), mp, mp, 0, T1, mp, nsp);
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(T2, T1, (size_t)nsp * mp * 8, hipMemcpyDeviceToDevice, h->stream));
  rc = blB.trsm_rec(h->dS3.d(), mp, mp, 0, T2, mp, nsp);
  if (rc) return rc;
  double* dmean = dDiag + mp;                                     // [n_new][r]
  double* dss2 = dmean + (size_t)n_new * r;
  double* dss1 = dss2 + n_new;
  rc = gps_launch_rowdot(h, T2, mp, n_new, mp, dC, mp, r, dmean, dss2);     // tmp2^T (c sigma^2)
  if (rc) return rc;
  rc = gps_launch_rowdot(h, T1, mp, n_new, mp, nullptr, mp, 0, nullptr, dss1);
  if (rc) return rc;
  std::vector<double> hm((size_t)n_new * r), h2(n_new), h1(n_new);
  GPS_HIP(h, hipMemcpyAsync(hm.data(), dmean, hm.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipMemcpyAsync(h2.data(), dss2, (size_t)n_new * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipMemcpyAsync(h1.data(), dss1, (size_t)n_new * 8, hipMemcpyDeviceToHost, h->stream));
  if (full_cov) {
    GPS_HIP(h, h->dVar.ensure((size_t)nsp * nsp * 8));
    rc = gps_launch_kmat(h, prog, n_nodes, h->dXnew.d(), n_new, nullptr, n_new, d_all, 0.0, h->dVar.d(), nsp, nsp, nsp, 0, 0);
    if (rc) return rc;
    rc = gps_launch_gemm_nt(h, 2, 0, nsp, nsp, mp, T2, mp, T2, mp, h->dVar.d(), nsp);
    if (rc) return rc;
    rc = gps_launch_gemm_nt(h, 0, 0, nsp, nsp, mp, T1, mp, T1, mp, h->dVar.d(), nsp);
    if (rc) return rc;
    GPS_HIP(h, h->dTmp2.ensure((size_t)n_new * n_new * 8));
    rc = gps_launch_extract(h, h->dVar.d(), nsp, n_new, n_new, h->dTmp2.d(), n_new, 0);
    if (rc) return rc;
    GPS_HIP(h, hipMemcpyAsync(var_out, h->dTmp2.p, (size_t)n_new * n_new * 8, hipMemcpyDeviceToHost, h->stream));
  }
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < hm.size(); ++i) mean_out[i] = hm[i] * wgt;          // SGPR: c = (c sigma^2)/sigma^2
  if (!full_cov)
    for (i64 i = 0; i < n_new; ++i) var_out[i] = kdiag + h2[i] - h1[i];
  return GPS_OK;
}

extern "C" int gps_sgpr(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                        const double* X, int64_t n, int64_t d_all, double jitter, double noise_var,
                        const double* resid, int64_t r, const double* Xnew, int64_t n_new, int full_cov,
                        double* bound_out, double* mean_out, double* var_out, int* info) {
  return with_la_retry(h, [&]() -> int {
  return sparse_gpr_impl(h, 0, prog, n_nodes, Z, m, X, n, d_all, jitter, noise_var, resid, r, Xnew, n_new, full_cov,
                         bound_out, mean_out, var_out, info);
  });
}

extern "C" int gps_fitc(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                        const double* X, int64_t n, int64_t d_all, double jitter, double noise_var,
                        const double* resid, int64_t r, const double* Xnew, int64_t n_new, int full_cov,
                        double* bound_out, double* mean_out, double* var_out, int* info) {
  return with_la_retry(h, [&]() -> int {
  return sparse_gpr_impl(h, 1, prog, n_nodes, Z, m, X, n, d_all, jitter, noise_var, resid, r, Xnew, n_new, full_cov,
                         bound_out, mean_out, var_out, info);
  });
}

// common tail of the SGPR / FITC gradients: from A_bar^T [np, mp] (cotangent of A = L^-1 Kuf, transposed) to the kernel
// parameters and the inducing inputs.  On the device: dK = L (blL: its block inverses), A [mp, np], dX = Z, dXnew = X.
//   Kuf_bar = L^-T A_bar ; L_bar = -tril(Kuf_bar A^T) ; Kuu_bar = adjoint(L, L_bar) ; kernel-matrix VJPs ; Kdiag's share kbar
static int sparse_grad_tail(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, i64 m, i64 n, i64 d_all,
                            Blocked<HipOps>& blL, double* AbarT, const double* A, double* U, double kdiag_bar, int ns,
                            double* grad_slots, double* grad_Z) {
  const i64 mp = gps_pad(m), np = gps_pad(n);
  double* L = h->dK.d();
  int rc;
  // Kuf_bar^T = A_bar^T L^-1 ; Kuf_bar [mp, np]
  rc = gps_launch_transpose(h, L, mp, mp, mp, U, mp);
  if (rc) return rc;
  rc = gps_launch_tri_map(h, U, mp, mp, 3);
  if (rc) return rc;
  rc = blL.trsm_rn_rec(U, mp, mp, 0, AbarT, mp, np);
  if (rc) return rc;
  GPS_HIP(h, h->dB.ensure((size_t)mp * np * 8));
  double* KufBar = h->dB.d();
  rc = gps_launch_transpose(h, AbarT, mp, np, mp, KufBar, np);
  if (rc) return rc;
  // L_bar = -tril(Kuf_bar A^T) ; Kuu_bar
  double* Lbar = h->dG1.d();
  rc = gps_launch_gemm_nt(h, 1, 1, mp, mp, np, KufBar, np, A, np, Lbar, mp);
  if (rc) return rc;
  rc = gps_launch_tri_map(h, Lbar, mp, mp, 1);
  if (rc) return rc;
  double* K2 = h->dG2.d();                                        // 2 Kuu_bar
  rc = chol_adjoint2(h, blL, U, Lbar, K2, h->dTmp2.d(), mp);
  if (rc) return rc;
  for (int sI = 0; sI < ns; ++sI) grad_slots[sI] = 0.0;
  rc = gps_launch_kmat_vjp(h, prog, n_nodes, h->dX.d(), m, h->dXnew.d(), n, d_all, KufBar, np, 0, grad_slots);
  if (rc) return rc;
  {
    std::vector<double> uu((size_t)ns, 0.0);
    rc = gps_launch_kmat_vjp(h, prog, n_nodes, h->dX.d(), m, nullptr, 0, d_all, K2, mp, 0, uu.data());
    if (rc) return rc;
    for (int sI = 0; sI < ns; ++sI) grad_slots[sI] += 0.5 * uu[sI];
  }
  rc = gps_kdiag_vjp(h, prog, n_nodes, d_all, kdiag_bar, grad_slots);
  if (rc) return rc;
  if (grad_Z) {
    for (i64 i = 0; i < m * d_all; ++i) grad_Z[i] = 0.0;
    rc = gps_launch_kmat_input_vjp(h, prog, n_nodes, h->dX.d(), m, h->dXnew.d(), n, d_all, KufBar, np, 1.0, grad_Z);
    if (rc) return rc;
    rc = gps_launch_kmat_input_vjp(h, prog, n_nodes, h->dX.d(), m, nullptr, 0, d_all, K2, mp, 1.0, grad_Z);
    if (rc) return rc;
  }
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

// ---- gradient of the SGPR bound ----------------------------------------------------------------------------------
// What TF autodiff through models/sgpr.py:121-153 supplies to the reference's optimiser (Z is a Parameter, features.py:65,
// and moves like every other variable).  Reverse mode at the matrix level over what gps_sgpr leaves on the device, s = noise variance, R outputs:
//   forward   L = chol(Kuu + jitter I), A = L^-1 Kuf, G = A A^T, B = I + G / s, LB = chol(B), v = A err, u = LB^-1 v, c = u / s
//             F = const - R sum log diag LB - N R / 2 log s - |err|^2 / (2 s) + |c|^2 / 2 - R sum Kdiag / (2 s) + R tr(G) / (2 s)
//   ubar = u / s^2 ; vbar = LB^-T ubar ; LB_bar = -tril(vbar u^T + R diag(1 / LB_ii)) ; B_bar = adjoint(LB, LB_bar)
//   G_bar = B_bar / s + R / (2 s) I ; A_bar = 2 G_bar A + vbar err^T ; Kuf_bar = L^-T A_bar ; L_bar = -tril(Kuf_bar A^T)
//   Kuu_bar = adjoint(L, L_bar) ; d/d theta = <Kuf_bar, dKuf> + <Kuu_bar, dKuu> - R N / (2 s) dKdiag   (kernel-matrix VJPs)
//   d/d s = -|u|^2 / s^3 - <B_bar, G> / s^2 - R tr(G) / (2 s^2) - N R / (2 s) + |err|^2 / (2 s^2) + R N Kdiag / (2 s^2),
//           <B_bar, G> = s (<LB_bar, LB> / 2 - tr B_bar)     (B = LB LB^T scales like LB^2; no second copy of G is kept)
//   d/d mean(X) = err / s - A^T vbar ; d/d Z through k(Z, X) and k(Z, Z) (gps_launch_kmat_input_vjp).
static int sgpr_grad_body(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                             const double* X, int64_t n, int64_t d_all, double jitter, double noise_var,
                             const double* resid, int64_t r, double* bound, double* grad_slots, int n_slots_cap,
                             int* n_slots_out, double* grad_noise, double* grad_mean, double* grad_Z, int* info);
// (wrapped like every factorising entry point: a missed look-ahead hand-over re-runs the body once, with_la_retry)
extern "C" int gps_sgpr_grad(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                             const double* X, int64_t n, int64_t d_all, double jitter, double noise_var,
                             const double* resid, int64_t r, double* bound, double* grad_slots, int n_slots_cap,
                             int* n_slots_out, double* grad_noise, double* grad_mean, double* grad_Z, int* info) {
  return with_la_retry(h, [&]() -> int { return sgpr_grad_body(h, prog, n_nodes, Z, m, X, n, d_all, jitter, noise_var, resid, r, bound, grad_slots, n_slots_cap, n_slots_out, grad_noise, grad_mean, grad_Z, info); });
}
static int sgpr_grad_body(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                             const double* X, int64_t n, int64_t d_all, double jitter, double noise_var,
                             const double* resid, int64_t r, double* bound, double* grad_slots, int n_slots_cap,
                             int* n_slots_out, double* grad_noise, double* grad_mean, double* grad_Z, int* info) {
  if (h && h->allreduce) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradients of the sparse bounds are not available with the data sharded over ranks");
  if (!h || !bound || !grad_slots || !grad_noise) return gps_fail(h, GPS_ERR_ARG, "gps_sgpr_grad: bad argument");
  if (r > GPS_TILE) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gps_sgpr_grad: at most 128 outputs");
  int ns = 0;
  int rc = gps_grad_general_slots(h, prog, n_nodes, &ns);
  if (rc) return rc;
  if (n_slots_out) *n_slots_out = ns;
  if (ns > n_slots_cap) return gps_fail(h, GPS_ERR_ARG, "gps_sgpr_grad: grad_slots too small");
  int linfo = 0;
  rc = gps_sgpr(h, prog, n_nodes, Z, m, X, n, d_all, jitter, noise_var, resid, r, nullptr, 0, 0, bound, nullptr, nullptr, &linfo);
  if (info) *info = linfo;
  if (rc || linfo) return rc;
  // on the device: dK = L, dLinv ; dS1 = A^T [np, mp] ; dS2 = A [mp, np] ; dS3 = LB, dS4 its block inverses ;
  // dAlpha = err^T [r][np], then u^T [r][mp] ; dX = Z ; dXnew = X
  const i64 mp = gps_pad(m), np = gps_pad(n);
  const double s = noise_var, R = (double)r, N = (double)n;
  const size_t blk_bytes = (size_t)(mp / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  HipOps opsL{h, h->dLinv.d(), h->dLinv.d() + blk_bytes / 8, (int*)h->dInfo.p};
  HipOps opsB{h, h->dS4.d(), h->dS4.d() + blk_bytes / 8, (int*)h->dInfo.p};
  Blocked<HipOps> blL(opsL), blB(opsB);
  double* At = h->dS1.d(); double* A = h->dS2.d(); double* LB = h->dS3.d();
  double* dErrT = h->dAlpha.d(); double* dUT = dErrT + (size_t)r * np;
  const double kdiag = h->sparse_terms[3], trG = h->sparse_terms[1];
  // u, then vbar^T = (LB^-T u / s^2)^T as rows
  std::vector<double> hu((size_t)r * mp), hv((size_t)r * mp);
  GPS_HIP(h, hipMemcpyAsync(hu.data(), dUT, hu.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  double u2 = 0.0;
  for (size_t i = 0; i < hu.size(); ++i) { u2 += hu[i] * hu[i]; hv[i] = hu[i] / (s * s); }
  GPS_HIP(h, h->dG3.ensure((size_t)(GPS_TILE + r) * mp * 8));
  double* dVT = h->dG3.d();                                       // [r][mp]
  GPS_HIP(h, hipMemcpyAsync(dVT, hv.data(), hv.size() * 8, hipMemcpyHostToDevice, h->stream));
  rc = blB.trsv_t_rec(LB, mp, mp, 0, dVT, mp, r);
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(hv.data(), dVT, hv.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  // LB_bar = -tril(vbar u^T + R diag(1 / LB_ii))
  GPS_HIP(h, h->dG1.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dG2.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dTmp.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dTmp2.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dTmp3.ensure((size_t)2 * mp * GPS_TILE * 8));
  std::vector<double> va((size_t)mp * GPS_TILE, 0.0), ub((size_t)mp * GPS_TILE, 0.0);
  for (i64 j = 0; j < m; ++j) for (i64 q = 0; q < r; ++q) { va[(size_t)j * GPS_TILE + q] = hv[(size_t)q * mp + j]; ub[(size_t)j * GPS_TILE + q] = hu[(size_t)q * mp + j]; }
  double* dVa = h->dTmp3.d(); double* dUb = dVa + (size_t)mp * GPS_TILE;
  GPS_HIP(h, hipMemcpyAsync(dVa, va.data(), va.size() * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(dUb, ub.data(), ub.size() * 8, hipMemcpyHostToDevice, h->stream));
  double* LBbar = h->dG1.d();
  rc = gps_launch_gemm_nt(h, 1, 1, mp, mp, GPS_TILE, dVa, GPS_TILE, dUb, GPS_TILE, LBbar, mp);
  if (rc) return rc;
  rc = gps_launch_diag_recip_add(h, LBbar, mp, LB, mp, m, R);
  if (rc) return rc;
  rc = gps_launch_tri_map(h, LBbar, mp, mp, 1);
  if (rc) return rc;
  double dots[2];
  rc = gps_tri_dot(h, LBbar, mp, LB, mp, m, dots);                 // <LB_bar, LB> over the lower triangle
  if (rc) return rc;
  const double lbar_dot_lb = dots[0];
  // B_bar: U_B = LB^T, adjoint
  double* U = h->dTmp.d();
  rc = gps_launch_transpose(h, LB, mp, mp, mp, U, mp);
  if (rc) return rc;
  rc = gps_launch_tri_map(h, U, mp, mp, 3);
  if (rc) return rc;
  double* B2 = h->dG2.d();                                        // 2 B_bar
  rc = chol_adjoint2(h, blB, U, LBbar, B2, h->dTmp2.d(), mp);
  if (rc) return rc;
  rc = gps_tri_dot(h, B2, mp, B2, mp, m, dots);
  if (rc) return rc;
  const double trBbar = 0.5 * dots[1];
  const double Bbar_dot_G = s * (0.5 * lbar_dot_lb - trBbar);
  *grad_noise = -u2 / (s * s * s) - Bbar_dot_G / (s * s) - 0.5 * R * trG / (s * s) - 0.5 * N * R / s + 0.5 * R * N * kdiag / (s * s);
  {
    double serr2 = 0.0;
    for (i64 i = 0; i < n * r; ++i) serr2 += resid[i] * resid[i];
    *grad_noise += 0.5 * serr2 / (s * s);
  }
  // 2 G_bar = (2 B_bar) / s + (R / s) I
  rc = gps_launch_axpby_eye(h, B2, mp, mp, m, 1.0 / s, R / s);
  if (rc) return rc;
  // A_bar^T [np, mp] = err vbar^T + A^T (2 G_bar)
  GPS_HIP(h, h->dY.ensure((size_t)np * mp * 8));
  double* AbarT = h->dY.d();
  std::vector<double> zero((size_t)mp, 0.0), vmk((size_t)mp * r, 0.0);
  for (i64 j = 0; j < m; ++j) for (i64 q = 0; q < r; ++q) vmk[(size_t)j * r + q] = hv[(size_t)q * mp + j];
  GPS_HIP(h, h->dG4.ensure((size_t)(mp + mp * r) * 8));
  double* dZero = h->dG4.d(); double* dVmk = dZero + mp;
  GPS_HIP(h, hipMemcpyAsync(dZero, zero.data(), (size_t)mp * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(dVmk, vmk.data(), vmk.size() * 8, hipMemcpyHostToDevice, h->stream));
  rc = gps_launch_svgp_abar(h, At, mp, np, mp, dZero, dErrT, np, dVmk, r, AbarT);
  if (rc) return rc;
  rc = gps_launch_gemm_nt(h, 2, 0, np, mp, mp, At, mp, B2, mp, AbarT, mp);
  if (rc) return rc;
  if (grad_mean) {                                                // err / s - A^T vbar   [n, r]
    GPS_HIP(h, h->dMean.ensure((size_t)(n * r + n) * 8));
    rc = gps_launch_rowdot(h, At, mp, n, mp, dVT, mp, r, h->dMean.d(), h->dMean.d() + (size_t)n * r);
    if (rc) return rc;
    std::vector<double> av((size_t)n * r);
    GPS_HIP(h, hipMemcpyAsync(av.data(), h->dMean.p, av.size() * 8, hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    for (i64 i = 0; i < n * r; ++i) grad_mean[i] = resid[i] / s - av[i];
  }
  GPS_HIP(h, hipStreamSynchronize(h->stream));                    // (host vectors above are read by the copies)
  return sparse_grad_tail(h, prog, n_nodes, m, n, d_all, blL, AbarT, A, U, -0.5 * R * N / s, ns, grad_slots, grad_Z);
}

// ---- gradient of the FITC log-likelihood (models/sgpr.py:229-290 under TF autodiff) ---------------------------------------
//   forward   A = L^-1 Kuf, q_i = |a_i|^2, nu_i = Kdiag - q_i + s, w_i = nu_i^-1/2, Ah = A diag(w), B = I + Ah Ah^T, LB = chol(B),
//             beta = err . w (rows), v = Ah beta, u = LB^-1 v,
//             F = -|beta|^2 / 2 + |u|^2 / 2 - R (N / 2 log 2 pi + sum log nu / 2 + sum log diag LB)
//   vbar = LB^-T u ; LB_bar = -tril(vbar u^T + R diag(1 / LB_ii)) ; B_bar = adjoint(LB, LB_bar)
//   Ah_bar = 2 B_bar Ah + vbar beta^T ; beta_bar = Ah^T vbar - beta
//   wbar_i = <Ah_bar[:, i], A[:, i]> + <beta_bar_i, err_i> ; nubar_i = -wbar_i nu_i^-3/2 / 2 - R / (2 nu_i)
//   sbar = sum nubar ; Kdiag_bar = sum nubar ; A_bar[:, i] = w_i Ah_bar[:, i] - 2 nubar_i A[:, i] ; then the common tail.
static int fitc_grad_body(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                             const double* X, int64_t n, int64_t d_all, double jitter, double noise_var,
                             const double* resid, int64_t r, double* bound, double* grad_slots, int n_slots_cap,
                             int* n_slots_out, double* grad_noise, double* grad_mean, double* grad_Z, int* info);
// (wrapped like every factorising entry point: a missed look-ahead hand-over re-runs the body once, with_la_retry)
extern "C" int gps_fitc_grad(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                             const double* X, int64_t n, int64_t d_all, double jitter, double noise_var,
                             const double* resid, int64_t r, double* bound, double* grad_slots, int n_slots_cap,
                             int* n_slots_out, double* grad_noise, double* grad_mean, double* grad_Z, int* info) {
  return with_la_retry(h, [&]() -> int { return fitc_grad_body(h, prog, n_nodes, Z, m, X, n, d_all, jitter, noise_var, resid, r, bound, grad_slots, n_slots_cap, n_slots_out, grad_noise, grad_mean, grad_Z, info); });
}
static int fitc_grad_body(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Z, int64_t m,
                             const double* X, int64_t n, int64_t d_all, double jitter, double noise_var,
                             const double* resid, int64_t r, double* bound, double* grad_slots, int n_slots_cap,
                             int* n_slots_out, double* grad_noise, double* grad_mean, double* grad_Z, int* info) {
  if (h && h->allreduce) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradients of the sparse bounds are not available with the data sharded over ranks");
  if (!h || !bound || !grad_slots || !grad_noise) return gps_fail(h, GPS_ERR_ARG, "gps_fitc_grad: bad argument");
  if (r > GPS_TILE) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gps_fitc_grad: at most 128 outputs");
  int ns = 0;
  int rc = gps_grad_general_slots(h, prog, n_nodes, &ns);
  if (rc) return rc;
  if (n_slots_out) *n_slots_out = ns;
  if (ns > n_slots_cap) return gps_fail(h, GPS_ERR_ARG, "gps_fitc_grad: grad_slots too small");
  int linfo = 0;
  rc = gps_fitc(h, prog, n_nodes, Z, m, X, n, d_all, jitter, noise_var, resid, r, nullptr, 0, 0, bound, nullptr, nullptr, &linfo);
  if (info) *info = linfo;
  if (rc || linfo) return rc;
  // on the device: dK = L ; dS1 = Ah^T [np, mp] (rows scaled by w) ; dS2 = Ah [mp, np] ; dS3 = LB ; dS4 its block inverses ;
  // dAlpha = beta^T [r][np], then u^T [r][mp] ; dTmp3 = w [np]
  const i64 mp = gps_pad(m), np = gps_pad(n);
  const double R = (double)r;
  const size_t blk_bytes = (size_t)(mp / GPS_TILE) * GPS_TILE * GPS_TILE * 8;
  HipOps opsL{h, h->dLinv.d(), h->dLinv.d() + blk_bytes / 8, (int*)h->dInfo.p};
  HipOps opsB{h, h->dS4.d(), h->dS4.d() + blk_bytes / 8, (int*)h->dInfo.p};
  Blocked<HipOps> blL(opsL), blB(opsB);
  double* Aht = h->dS1.d(); double* Ah = h->dS2.d(); double* LB = h->dS3.d();
  double* dBetaT = h->dAlpha.d(); double* dUT = dBetaT + (size_t)r * np;
  std::vector<double> w((size_t)np, 0.0), hu((size_t)r * mp), hv((size_t)r * mp);
  GPS_HIP(h, hipMemcpyAsync(w.data(), h->dTmp3.p, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipMemcpyAsync(hu.data(), dUT, hu.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  // the vector of weights moves to a buffer of its own (dTmp3 is scratch below)
  GPS_HIP(h, h->dG4.ensure((size_t)(3 * np + mp + mp * r) * 8));
  double* dW = h->dG4.d(); double* dCa = dW + np; double* dCb = dCa + np; double* dZero = dCb + np; double* dVmk = dZero + mp;
  GPS_HIP(h, hipMemcpyAsync(dW, w.data(), (size_t)np * 8, hipMemcpyHostToDevice, h->stream));
  // vbar^T = (LB^-T u)^T
  GPS_HIP(h, h->dG3.ensure((size_t)(GPS_TILE + r) * mp * 8));
  double* dVT = h->dG3.d();
  GPS_HIP(h, hipMemcpyAsync(dVT, hu.data(), hu.size() * 8, hipMemcpyHostToDevice, h->stream));
  rc = blB.trsv_t_rec(LB, mp, mp, 0, dVT, mp, r);
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(hv.data(), dVT, hv.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  // LB_bar, B_bar
  GPS_HIP(h, h->dG1.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dG2.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dTmp.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dTmp2.ensure((size_t)mp * mp * 8));
  GPS_HIP(h, h->dTmp3.ensure((size_t)2 * mp * GPS_TILE * 8));
  std::vector<double> va((size_t)mp * GPS_TILE, 0.0), ub((size_t)mp * GPS_TILE, 0.0);
  for (i64 j = 0; j < m; ++j) for (i64 q = 0; q < r; ++q) { va[(size_t)j * GPS_TILE + q] = hv[(size_t)q * mp + j]; ub[(size_t)j * GPS_TILE + q] = hu[(size_t)q * mp + j]; }
  double* dVa = h->dTmp3.d(); double* dUb = dVa + (size_t)mp * GPS_TILE;
  GPS_HIP(h, hipMemcpyAsync(dVa, va.data(), va.size() * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(dUb, ub.data(), ub.size() * 8, hipMemcpyHostToDevice, h->stream));
  double* LBbar = h->dG1.d();
  rc = gps_launch_gemm_nt(h, 1, 1, mp, mp, GPS_TILE, dVa, GPS_TILE, dUb, GPS_TILE, LBbar, mp);
  if (rc) return rc;
  rc = gps_launch_diag_recip_add(h, LBbar, mp, LB, mp, m, R);
  if (rc) return rc;
  rc = gps_launch_tri_map(h, LBbar, mp, mp, 1);
  if (rc) return rc;
  double* U = h->dTmp.d();
  rc = gps_launch_transpose(h, LB, mp, mp, mp, U, mp);
  if (rc) return rc;
  rc = gps_launch_tri_map(h, U, mp, mp, 3);
  if (rc) return rc;
  double* B2 = h->dG2.d();                                        // 2 B_bar
  rc = chol_adjoint2(h, blB, U, LBbar, B2, h->dTmp2.d(), mp);
  if (rc) return rc;
  // Ah_bar^T [np, mp] = beta vbar^T + Ah^T (2 B_bar)
  GPS_HIP(h, h->dY.ensure((size_t)np * mp * 8));
  double* AbarT = h->dY.d();
  std::vector<double> zero((size_t)mp, 0.0), vmk((size_t)mp * r, 0.0);
  for (i64 j = 0; j < m; ++j) for (i64 q = 0; q < r; ++q) vmk[(size_t)j * r + q] = hv[(size_t)q * mp + j];
  GPS_HIP(h, hipMemcpyAsync(dZero, zero.data(), (size_t)mp * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(dVmk, vmk.data(), vmk.size() * 8, hipMemcpyHostToDevice, h->stream));
  rc = gps_launch_svgp_abar(h, Aht, mp, np, mp, dZero, dBetaT, np, dVmk, r, AbarT);
  if (rc) return rc;
  rc = gps_launch_gemm_nt(h, 2, 0, np, mp, mp, Aht, mp, B2, mp, AbarT, mp);
  if (rc) return rc;
  // beta_bar = Ah^T vbar - beta  [n, r] ; row dots <Ah_bar^T[i], Ah^T[i]> (= w_i <Ah_bar[:, i], A[:, i]>)
  GPS_HIP(h, h->dMean.ensure((size_t)(n * r + 2 * np) * 8));
  double* dAv = h->dMean.d(); double* dRd = dAv + (size_t)n * r;
  rc = gps_launch_rowdot(h, Aht, mp, n, mp, dVT, mp, r, dAv, dRd + np);
  if (rc) return rc;
  rc = gps_launch_rowdot2(h, AbarT, mp, Aht, mp, n, mp, dRd);
  if (rc) return rc;
  std::vector<double> av((size_t)n * r), rd((size_t)n);
  GPS_HIP(h, hipMemcpyAsync(av.data(), dAv, av.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipMemcpyAsync(rd.data(), dRd, rd.size() * 8, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<double> ca((size_t)np, 0.0), cb((size_t)np, 0.0), iw((size_t)np, 0.0);
  double nubar_sum = 0.0;
  for (i64 i = 0; i < n; ++i) {
    const double wi = w[i], nu = 1.0 / (wi * wi);
    double wbar = rd[i] / wi;                                     // <Ah_bar[:, i], A[:, i]>, A[:, i] = Ah[:, i] / w_i
    for (i64 q = 0; q < r; ++q) {
      const double beta = resid[i * r + q] * wi;
      const double bbar = av[i * r + q] - beta;
      wbar += bbar * resid[i * r + q];
      if (grad_mean) grad_mean[i * r + q] = -bbar * wi;           // err = Y - mean(X)
    }
    const double nubar = -0.5 * wbar * wi * wi * wi - 0.5 * R / nu;
    nubar_sum += nubar;
    ca[i] = wi; cb[i] = -2.0 * nubar / wi;                        // A_bar^T[i] = w_i Ah_bar^T[i] - 2 nubar_i A^T[i], A^T[i] = Ah^T[i] / w_i
    iw[i] = 1.0 / wi;
  }
  *grad_noise = nubar_sum;
  GPS_HIP(h, hipMemcpyAsync(dCa, ca.data(), (size_t)np * 8, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(dCb, cb.data(), (size_t)np * 8, hipMemcpyHostToDevice, h->stream));
  rc = gps_launch_rows_axpby(h, AbarT, mp, Aht, mp, n, mp, dCa, dCb);
  if (rc) return rc;
  // A = Ah diag(1 / w)  (columns of [mp, np]) for the tail
  GPS_HIP(h, hipMemcpyAsync(dW, iw.data(), (size_t)np * 8, hipMemcpyHostToDevice, h->stream));
  rc = gps_launch_scale_cols(h, Ah, np, mp, n, dW, Ah, np);
  if (rc) return rc;
  GPS_HIP(h, hipStreamSynchronize(h->stream));                    // (host vectors above are read by the copies)
  return sparse_grad_tail(h, prog, n_nodes, m, n, d_all, blL, AbarT, Ah, U, nubar_sum, ns, grad_slots, grad_Z);
}

extern "C" int gps_set_allreduce(gps_handle_t h, gps_allreduce_fn fn, void* ctx, void* dev_buf, int64_t capacity_doubles) {
  if (!h) return GPS_ERR_ARG;
  if (fn && (!dev_buf || capacity_doubles <= 0)) return gps_fail(h, GPS_ERR_ARG, "gps_set_allreduce: a device buffer is required");
  h->allreduce = fn; h->allreduce_ctx = fn ? ctx : nullptr;
  h->red_buf = fn ? (double*)dev_buf : nullptr; h->red_cap = fn ? capacity_doubles : 0;
  return GPS_OK;
}

extern "C" int gps_allreduce_doubles(int64_t m, int64_t r, int64_t* out) {
  if (!out || m <= 0 || r < 0) return GPS_ERR_ARG;
  const i64 mp = gps_pad(m);
  *out = mp * mp + mp * r + mp + 4;
  return GPS_OK;
}

extern "C" int gps_sparse_last_terms(gps_handle_t h, double* out5) {
  if (!h || !out5) return GPS_ERR_ARG;
  for (int i = 0; i < 5; ++i) out5[i] = h->sparse_terms[i];
  return GPS_OK;
}
