// C ABI of libgpflowslim_hip.so (include/gpflowslim_hip.h): handle life cycle, measurement, options, diagnostics.
#include "gps_ops.hpp"

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
  if (const char* v = getenv("GPS_GEMM_PAIR")) h->gemm_pair = atoi(v);               // diagnostics: A/B of the paired triangular tiles
  if (const char* la = getenv("GPS_LOOKAHEAD")) h->potrf_lookahead = atoi(la);     // diagnostics: counter collection serialises the dispatches (tools/collect_profiles.sh)
  if (h->dInfo.ensure(64) != hipSuccess || h->dScal.ensure(4096) != hipSuccess) { delete h; return GPS_ERR_HIP; }
  *out = h;
  return GPS_OK;
}

// every growable device buffer of the handle (the small fixed ones -- info word, look-ahead flags, pinned ring -- stay)
static void release_work_buffers(gps_handle_t h, bool all) {
  DevBuf* bufs[] = {&h->dX, &h->dK, &h->dLinv, &h->dAlpha, &h->dFeat, &h->dFeat2, &h->dProg,
                    &h->dXnew, &h->dB, &h->dMean, &h->dVar, &h->dTmp, &h->dTmp2, &h->dTmp3, &h->dA, &h->dY,
                    &h->dKinv, &h->dNkn, &h->dS1, &h->dS2, &h->dS3, &h->dS4, &h->dGemvWs, &h->dGemvCnt, &h->dGemmWs, &h->dGemmCnt, &h->dBlkCond, &h->dStage, &h->dWbig, &h->dWtbig, &h->dBigT, &h->dB2,
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
  h->have_factor = false; h->factor_gen++; h->dist_have_part_factor = false; h->n = 0; h->npad = 0; h->r = 0;
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
  if (strcmp(key, "predict_inverse_blocks") == 0) { h->predict_inv_blocks = (int)value; return GPS_OK; }
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
  // (lower 3 / 4: A lower triangular / B lower triangular -- the forms behind the wide inverse blocks of predict_f)
  int rc = lower >= 3 ? gps_launch_gemm_nt_ex(h, op, 0, lower - 1, m, n, k, h->dTmp.d(), k, h->dTmp2.d(), k, h->dTmp3.d(), n, nullptr)
                      : gps_launch_gemm_nt(h, op, lower, m, n, k, h->dTmp.d(), k, h->dTmp2.d(), k, h->dTmp3.d(), n);
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(C, h->dTmp3.p, cb, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

// the same for a batch of `batch` equal problems stacked row-wise: A [batch * m, k], B [batch * n, k], C [batch * m, n];
// tri: 0 none, 1 A upper, 2 A lower, 3 B lower triangular
extern "C" int gps_diag_gemm_nt_batched(gps_handle_t h, int op, int tri, int64_t batch, int64_t m, int64_t n, int64_t k,
                                        const double* A, const double* B, double* C) {
  if (!h || !A || !B || !C || batch < 1) return GPS_ERR_ARG;
  GPS_HIP(h, hipSetDevice(h->device));
  const size_t ab = (size_t)batch * m * k * 8, bb = (size_t)batch * n * k * 8, cb = (size_t)batch * m * n * 8;
  GPS_HIP(h, h->dTmp.ensure(ab)); GPS_HIP(h, h->dTmp2.ensure(bb)); GPS_HIP(h, h->dTmp3.ensure(cb));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp.p, A, ab, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp2.p, B, bb, hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(h->dTmp3.p, C, cb, hipMemcpyHostToDevice, h->stream));
  GemmBatch bt; bt.batch = batch; bt.a_rs = m; bt.b_rs = n; bt.c_rs = m;
  int rc = gps_launch_gemm_nt_ex(h, op, 0, tri, m, n, k, h->dTmp.d(), k, h->dTmp2.d(), k, h->dTmp3.d(), n, &bt);
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

