// C ABI of libgpflowslim_hip.so (include/gpflowslim_hip.h): kernels.K, tf.cholesky / tf.matrix_triangular_solve on host matrices,
// GPR._build_likelihood / _build_predict (models/gpr.py:69-72, 119-131) and the likelihood's gradient.
#include "gps_ops.hpp"

static int gpr_lml_finish(gps_handle_t h, i64 r, double* lml);
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
  h->have_factor = false; h->factor_gen++; h->dist_have_part_factor = false;
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
  h->have_factor = false; h->factor_gen++; h->dist_have_part_factor = false;
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

// ---- predict_f on few test points: wide inverse blocks of the resident factor ------------------------------------------------
// A^T = Kx^T L^-T for few rows (models/gpr.py:122; used up to GPS_WIDE_MAX_ROWS = 8192 test points) is a chain of launches that cannot fill the GPU below the 4096-column
// level of trsm_rec: measured at N = 32768, m = 1024 (rocprofv3, round 6): the 64 one-launch 512-column leaves 43 us each on
// 32 workgroups, the K = 512 / 1024 updates at 32 / 40 TFLOP/s -- 4.2 of the call's 19.5 ms for 5 % of its flop.  With the
// explicit inverses W_c of the GPS_WB = 2048-column diagonal blocks of L the 2048-column node is ONE product
// X_c = B_c W_c^T (B lower triangular: half the K range per tile on average), and only the K >= 2048 updates remain.
// W is built level by level from the 128-column inverses potrf_base leaves behind, for all diagonal blocks at once
// (blocked.hpp: wide_inverse -- three batched NT products with a triangular operand per level; trsm_wide_rec; both checked as index
// logic on the CPU, tests/test_blocked_cpu.py), 11 launches, ~N * 3 * sum b^2 flop = 1.4e11 dense
// at N = 32768, cut by the triangular forms; cached until the factor changes (factor_gen).  Not where leaves are refined.
// cond(L_cc) <= sqrt(cond(K + s I)): the products stay within the same 7 u cond bound as the 128-column ones (gps_common.hpp).
static int gpr_wide_inverse(gps_handle_t h) {
  const i64 np = h->npad, WB = GPS_WB, nf = (np / WB) * WB;
  if (h->big_inv_gen == h->factor_gen && h->big_inv_nf == nf) return GPS_OK;
  h->big_inv_gen = ~0ull;
  GPS_HIP(h, h->dWbig.ensure((size_t)nf * WB * 8));
  GPS_HIP(h, h->dWtbig.ensure((size_t)nf * WB * 8));
  GPS_HIP(h, h->dBigT.ensure((size_t)(nf / 2) * (WB / 2) * 8));
  int rc = gpr_ensure_linvT(h);
  if (rc) return rc;
  const i64 nblk = np / GPS_TILE;
  HipOps ops{h, h->dLinv.d(), h->dLinv.d() + nblk * GPS_TILE * GPS_TILE, (int*)h->dInfo.p};
  Blocked<HipOps> bl(ops);
  rc = bl.wide_inverse(h->dK.d(), np, nf, WB, ops.linv, ops.linvT, h->dWbig.d(), h->dWtbig.d(), h->dBigT.d());
  if (rc) return rc;
  h->big_inv_gen = h->factor_gen; h->big_inv_nf = nf;
  return GPS_OK;
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
  // (up to 64 test points, wide path: half a tile row of right-hand sides -- examples/gpr.py predicts on ~51 points)
  const bool wide = h->predict_inv_blocks && !h->refine_now && gps_pad(n_new) <= GPS_WIDE_MAX_ROWS && (np / GPS_WB) * GPS_WB >= 2 * GPS_WB;
  const i64 nsp = (wide && !full_cov && n_new <= 64) ? 64 : gps_pad(n_new);
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
  const double* dAt = h->dB.d();                  // where A^T ends up
  const i64 nf = (np / GPS_WB) * GPS_WB;
  if (wide) {
    // up to 8192 test points: 2048-column nodes as one product with the wide inverse blocks (the choice depends on the shapes
    // only: a call that re-factors and a call on the resident factor give the same bits).  What it buys shrinks with the rows:
    // N = 32768: N* = 64 -48 %, 1024 -15 %, 4096 -2.6 %, 8192 -0.6 %
    rc = gpr_wide_inverse(h);
    if (rc) return rc;
    GPS_HIP(h, h->dB2.ensure((size_t)nsp * np * 8));
    rc = bl.trsm_wide_rec(h->dK.d(), np, 0, nf, GPS_WB, h->dWbig.d(), h->dB.d(), h->dB2.d(), np, nsp);
    if (rc) return rc;
    if (nf < np) {
      // the columns behind the last whole block: one update, the recursive solve in place, then beside the others
      rc = gps_launch_gemm_nt(h, 0, 0, nsp, np - nf, nf, h->dB2.d(), np, h->dK.d() + nf * np, np, h->dB.d() + nf, np);
      if (!rc) rc = bl.trsm_rec(h->dK.d() + nf * np + nf, np, np - nf, nf / GPS_TILE, h->dB.d() + nf, np, nsp);
      if (rc) return rc;
      GPS_HIP(h, hipMemcpy2DAsync(h->dB2.d() + nf, (size_t)np * 8, h->dB.d() + nf, (size_t)np * 8, (size_t)(np - nf) * 8, (size_t)nsp,
                                  hipMemcpyDeviceToDevice, h->stream));
    }
    dAt = h->dB2.d();
  } else {
    rc = bl.trsm_rec(h->dK.d(), np, np, 0, h->dB.d(), np, nsp);
    if (rc) return rc;
  }
  // fmean = A^T V ; sumsq = colsum(A*A)                          models/gpr.py:124,130
  GPS_HIP(h, h->dMean.ensure((size_t)(n_new * (r > 0 ? r : 1) + n_new) * 8));
  double* dmean = h->dMean.d();
  double* dss = dmean + n_new * (r > 0 ? r : 1);
  rc = gps_launch_rowdot(h, dAt, np, n_new, np, h->dAlpha.d(), np, r, dmean, dss);
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
    rc = gps_launch_gemm_nt(h, 0, 0, nsp, nsp, np, dAt, np, dAt, np, h->dVar.d(), nsp);
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

