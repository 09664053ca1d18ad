// C ABI of libgpflowslim_hip.so (include/gpflowslim_hip.h): external streams and the block-column distributed factorisation.
#include "gps_ops.hpp"

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
  h->have_factor = false; h->factor_gen++; h->dist_have_part_factor = false;
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
                    &h->dBlkCond, &h->dStage, &h->dWbig, &h->dWtbig, &h->dBigT, &h->dB2, &h->dSmallSync,
                    &h->dDistComm[0], &h->dDistComm[1], &h->dDistComm[2]};
  int64_t tot = 0;
  for (DevBuf* b : bufs) tot += (int64_t)b->cap;
  *bytes = tot;
  return GPS_OK;
}


