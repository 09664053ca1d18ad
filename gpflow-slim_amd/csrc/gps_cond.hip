// C ABI of libgpflowslim_hip.so (include/gpflowslim_hip.h): conditionals.py, the SVGP bound and its gradient, kernels.K's
// vector-Jacobian products, kullback_leiblers.py.
#include "gps_ops.hpp"

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
  h->have_factor = false; h->factor_gen++; h->dist_have_part_factor = false;          // dK / dLinv / dAlpha are reused below
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
  h->have_factor = false; h->factor_gen++; h->dist_have_part_factor = false;
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
  h->have_factor = false; h->factor_gen++; h->dist_have_part_factor = false; h->n = 0;
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
  h->have_factor = false; h->factor_gen++; h->dist_have_part_factor = false; h->n = 0;
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


