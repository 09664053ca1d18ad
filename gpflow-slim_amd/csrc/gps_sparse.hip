// C ABI of libgpflowslim_hip.so (include/gpflowslim_hip.h): SGPR / GPRFITC bounds, predictions and gradients (models/sgpr.py).
#include "gps_ops.hpp"

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
  h->have_factor = false; h->factor_gen++; h->dist_have_part_factor = false; h->n = 0;                       // GPR resident buffers are reused below
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

