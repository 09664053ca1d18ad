// HBM-bound vector / reduction kernels of the path:
//   - forward substitution pieces of  alpha = L^-1 (Y - m)     (densities.py:81-82,
//     models/gpr.py:123): 128-block solve through the stored block inverse and the
//     gemv update  y2 -= L21 y1  of the recursive trsv;
//   - sum(log diag L), sum(alpha^2)                             (densities.py:92-94);
//   - fmean = A^T V and colsum(A*A) in ONE pass over A^T        (models/gpr.py:124,130);
//   - layout helpers (pad / extract / transpose) for the host-matrix entry points.
// All of them stream each matrix element exactly once with 16-byte loads; wave
// reductions use 64-wide shuffles.
#include "gps_common.hpp"

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// y_blk[r][0..127] = Linv * y_blk[r], read through the TRANSPOSED inverse so that consecutive
// threads read consecutive addresses: out[i] = sum_c LinvT[c][i] * y[c]      (grid.x = r)
__global__ __launch_bounds__(256) void trsv_base_kernel(const double* __restrict__ LinvT,
                                                        double* __restrict__ y, i64 ldy) {
  __shared__ double ys[128];
  __shared__ double part[128];
  double* yr = y + (i64)blockIdx.x * ldy;
  const int tid = threadIdx.x, i = tid & 127, half = tid >> 7;
  if (tid < 128) ys[tid] = yr[tid];
  __syncthreads();
  const double* Lc = LinvT + (half * 64) * 128 + i;
  double s = 0.0;
#pragma unroll 16
  for (int c = 0; c < 64; ++c) s += Lc[c * 128] * ys[half * 64 + c];
  if (half) part[i] = s;
  __syncthreads();
  if (!half) yr[i] = s + part[i];
}

// y2[r][i] -= sum_k L21[i][k] * y1[r][k],  i < n2, k < n1 (n1 multiple of 128).
// One wave per row, 16-byte loads along the row; RC right-hand sides per pass.
template <int RC>
__global__ __launch_bounds__(256) void gemv_sub_kernel(const double* __restrict__ L21, i64 ldl,
                                                       i64 n2, i64 n1,
                                                       const double* __restrict__ y1,
                                                       double* __restrict__ y2, i64 ldy, int r0,
                                                       int rcount) {
  const int lane = threadIdx.x & 63;
  const i64 wave_global = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
  const i64 nwaves = (i64)gridDim.x * 4;
  for (i64 row = wave_global; row < n2; row += nwaves) {
    const double* Lr = L21 + row * ldl;
    double acc[RC];
#pragma unroll
    for (int q = 0; q < RC; ++q) acc[q] = 0.0;
    for (i64 k = 2 * lane; k < n1; k += 128) {
      const v2d l = *reinterpret_cast<const v2d*>(Lr + k);
#pragma unroll
      for (int q = 0; q < RC; ++q) {
        if (q < rcount) {
          const v2d v = *reinterpret_cast<const v2d*>(y1 + (i64)(r0 + q) * ldy + k);
          acc[q] += l.x * v.x + l.y * v.y;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RC; ++q) {
      const double s = wave_sum(acc[q]);
      if (lane == 0 && q < rcount) y2[(i64)(r0 + q) * ldy + row] -= s;
    }
  }
}

// y1[q][k] -= sum_i L21[i][k] * y2[q][i]   (transposed panel: backward substitution L^T a = y).
// Workgroup (bx, s) owns 128 columns k and the s-th slice of the n2 rows: a single column block has too little
// memory parallelism to stream a tall panel (one CU draws ~20-50 GB/s), so the rows are split over gridDim.y
// workgroups.  Their partial sums go to `ws`; the workgroup that arrives last at the column block's counter
// adds them in slice order, so the result does not depend on arrival order (no floating-point atomics).
template <int RC>
__global__ __launch_bounds__(256) void gemv_t_sub_kernel(const double* __restrict__ L21, i64 ldl,
                                                         i64 n2, i64 n1,
                                                         const double* __restrict__ y2,
                                                         double* __restrict__ y1, i64 ldy, int r0,
                                                         int rcount, double* __restrict__ ws,
                                                         unsigned* __restrict__ counters) {
  __shared__ double part[4][RC][128];
  __shared__ unsigned ticket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = (int)gridDim.y, sl = (int)blockIdx.y;
  const i64 k = (i64)blockIdx.x * 128 + 2 * lane;
  const i64 rows_per = ((n2 + S - 1) / S + 3) & ~(i64)3;
  const i64 i0 = (i64)sl * rows_per, i1 = min(n2, i0 + rows_per);
  double acc[RC][2];
#pragma unroll
  for (int q = 0; q < RC; ++q) acc[q][0] = acc[q][1] = 0.0;
  // wave w takes rows i0 + w, i0 + w + 4, ...; 8 row loads in flight per wave
  i64 i = i0 + wave;
  for (; i + 28 < i1; i += 32) {
    v2d l[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) l[u] = *reinterpret_cast<const v2d*>(L21 + (i + 4 * u) * ldl + k);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < RC; ++q)
        if (q < rcount) {
          const double v = y2[(i64)(r0 + q) * ldy + i + 4 * u];
          acc[q][0] = fma(l[u].x, v, acc[q][0]);
          acc[q][1] = fma(l[u].y, v, acc[q][1]);
        }
  }
  for (; i < i1; i += 4) {
    const v2d l = *reinterpret_cast<const v2d*>(L21 + i * ldl + k);
#pragma unroll
    for (int q = 0; q < RC; ++q)
      if (q < rcount) {
        const double v = y2[(i64)(r0 + q) * ldy + i];
        acc[q][0] = fma(l.x, v, acc[q][0]);
        acc[q][1] = fma(l.y, v, acc[q][1]);
      }
  }
#pragma unroll
  for (int q = 0; q < RC; ++q) { part[wave][q][2 * lane] = acc[q][0]; part[wave][q][2 * lane + 1] = acc[q][1]; }
  __syncthreads();
  const int kk = threadIdx.x & 127, qh = threadIdx.x >> 7;     // 2 right-hand sides per pass
  double* wsb = ws + ((i64)blockIdx.x * S) * (RC * 128);
  for (int q = qh; q < RC; q += 2) {
    const double t = (part[0][q][kk] + part[1][q][kk]) + (part[2][q][kk] + part[3][q][kk]);
    if (S == 1) {
      if (q < rcount) y1[(i64)(r0 + q) * ldy + (i64)blockIdx.x * 128 + kk] -= t;
    } else {
      wsb[(i64)sl * (RC * 128) + q * 128 + kk] = t;
    }
  }
  if (S == 1) return;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) ticket = atomicAdd(&counters[blockIdx.x], 1u);
  __syncthreads();
  if (ticket != (unsigned)(S - 1)) return;
  __threadfence();
  for (int q = qh; q < RC; q += 2) {
    if (q >= rcount) continue;
    double t = 0.0;
    for (int s2 = 0; s2 < S; ++s2) t += __builtin_nontemporal_load(&wsb[(i64)s2 * (RC * 128) + q * 128 + kk]);
    y1[(i64)(r0 + q) * ldy + (i64)blockIdx.x * 128 + kk] -= t;
  }
  if (threadIdx.x == 0) counters[blockIdx.x] = 0;     // ready for the next launch on this stream
}

// partial[b][0] = sum_i log L_ii ; partial[b][1] = sum alpha^2   (fixed-order, host adds
// the gridDim.x partials so the result is bitwise reproducible)
__global__ __launch_bounds__(256) void lml_reduce_kernel(const double* __restrict__ L, i64 ldl,
                                                         i64 n, const double* __restrict__ alpha,
                                                         i64 ldy, i64 r,
                                                         double* __restrict__ partial) {
  __shared__ double sh[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    s0 += log(L[i * ldl + i]);
    for (i64 q = 0; q < r; ++q) {
      const double a = alpha[q * ldy + i];
      s1 += a * a;
    }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if (lane == 0) { sh[0][wave] = s0; sh[1][wave] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x + 0] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    partial[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}

// One workgroup per row of A^T [n_new, npad]:
//   mean[i][q] = sum_n At[i][n] * alpha[q][n] ;  sumsq[i] = sum_n At[i][n]^2
__global__ __launch_bounds__(256) void rowdot_kernel(const double* __restrict__ At, i64 ldat,
                                                     i64 npad, const double* __restrict__ alpha,
                                                     i64 ldy, int r, double* __restrict__ mean,
                                                     double* __restrict__ sumsq) {
  __shared__ double sh[4];
  const i64 row = blockIdx.x;
  const double* Ar = At + row * ldat;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // sum of squares + first output together, remaining outputs re-read the row (L2-hot)
  for (int q = 0; q < (r > 0 ? r : 1); ++q) {
    double s = 0.0, ss = 0.0;
    for (i64 k = 2 * (i64)threadIdx.x; k < npad; k += 512) {
      const v2d a = *reinterpret_cast<const v2d*>(Ar + k);
      if (r > 0) {
        const v2d v = *reinterpret_cast<const v2d*>(alpha + (i64)q * ldy + k);
        s += a.x * v.x + a.y * v.y;
      }
      if (q == 0) ss += a.x * a.x + a.y * a.y;
    }
    s = wave_sum(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0 && r > 0) mean[row * r + q] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    if (q == 0) {
      ss = wave_sum(ss);
      if (lane == 0) sh[wave] = ss;
      __syncthreads();
      if (threadIdx.x == 0) sumsq[row] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
      __syncthreads();
    }
  }
}

// SVGP, Gaussian likelihood (likelihoods.py:186-188): partial[b] += sum over this block's points of
//   (yres[i][q] - fmean[i][q])^2 + base[i] + extra[i]      (fvar = base + extra, conditionals.py:96,107-118)
// One writer per slot and launches of one stream are ordered, so the k per-latent launches add up reproducibly.
__global__ __launch_bounds__(256) void varexp_kernel(const double* __restrict__ fmean, const double* __restrict__ yres,
                                                     int k, int q, const double* __restrict__ base,
                                                     const double* __restrict__ extra, i64 n,
                                                     double* __restrict__ partial) {
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0.0;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double e = yres[i * k + q] - fmean[i * k + q];
    s += e * e + base[i] + (extra ? extra[i] : 0.0);
  }
  s = wave_sum(s);
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] += (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- SVGP gradient helpers (gps_svgp_elbo_grad) -----------------------------------------------------------------------
// Et[q][i] = scale (yres[i][q] - fmean[i][q]) / noise  (i < n; 0 in the padding): d ELBO / d fmean, transposed
__global__ __launch_bounds__(256) void svgp_et_kernel(const double* __restrict__ yres, const double* __restrict__ fmean, int k,
                                                      i64 n, i64 npad, double coef, double* __restrict__ Et) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  for (int q = 0; q < k; ++q) Et[(i64)q * npad + i] = (i < n) ? coef * (yres[i * k + q] - fmean[i * k + q]) : 0.0;
}
// Abar[i][m] = coef[m] Bt[i][m] + sum_q Et[q][i] qmu[m][q]     ([rows, cols], cols = inducing points)
__global__ __launch_bounds__(256) void svgp_abar_kernel(const double* __restrict__ Bt, i64 ld, i64 rows, i64 cols,
                                                        const double* __restrict__ coef, const double* __restrict__ Et, i64 lde,
                                                        const double* __restrict__ qmu, int k, double* __restrict__ Abar) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const double cf = coef[c];
  double qm[8];
  for (int q = 0; q < k && q < 8; ++q) qm[q] = qmu[c * k + q];
  for (i64 rr = blockIdx.y; rr < rows; rr += gridDim.y) {
    double v = cf * Bt[rr * ld + c];
    for (int q = 0; q < k; ++q) v = fma(Et[(i64)q * lde + rr], (q < 8) ? qm[q] : qmu[c * k + q], v);
    Abar[rr * ld + c] = v;
  }
}
// in-place maps on an [n, n] matrix (every element is written by one thread and nobody reads what another one writes).
// mode 0: mirror the lower triangle into the upper one (also Phi(A) + Phi(A)^T of the Cholesky adjoint, Murray 2016: Phi =
// lower triangle with halved diagonal) ; 1: A <- -tril(A) ; 3: A <- triu(A)
__global__ __launch_bounds__(256) void tri_map_kernel(double* __restrict__ A, i64 ld, i64 n, int mode) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  for (i64 rr = blockIdx.y; rr < n; rr += gridDim.y) {
    if (mode == 0) { if (c > rr) A[rr * ld + c] = A[c * ld + rr]; }
    else if (mode == 1) { A[rr * ld + c] = (c <= rr) ? -A[rr * ld + c] : 0.0; }
    else if (mode == 3) { if (c < rr) A[rr * ld + c] = 0.0; }                    // keep the upper triangle only

  }
}

__global__ void fill_info_kernel(int* p, int v) { *p = v; }

// block-column distributed factorisation: fold the 64 fixed-order partials of lml_reduce_kernel and the info word of
// the owner into the tail of the panel message (and the owner's own per-panel table)
__global__ void dist_tail_kernel(const double* __restrict__ part, const int* __restrict__ info, double* __restrict__ t0,
                                 double* __restrict__ t1) {
  double a = 0.0, b = 0.0;
  for (int i = 0; i < 64; ++i) { a += part[2 * i]; b += part[2 * i + 1]; }
  const int v = *info;
  const double iv = (v == 0x7fffffff) ? 0.0 : (double)v;
  t0[0] = a; t0[1] = b; t0[2] = iv; t0[3] = 0.0;
  t1[0] = a; t1[1] = b; t1[2] = iv; t1[3] = 0.0;
}

// var[i] = kdiag - sumsq[i]
__global__ void var_finish_kernel(double* __restrict__ var, const double* __restrict__ kdiag,
                                  double kconst, const double* __restrict__ sumsq, i64 n) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) var[i] = (kdiag ? kdiag[i] : kconst) - sumsq[i];
}

// B <- scale * B + I on an [n, n] matrix (lower triangle is what matters); rows/cols >= n_real: identity
__global__ __launch_bounds__(256) void scale_add_eye_kernel(double* __restrict__ B, i64 ldb, i64 n, i64 n_real,
                                                            double scale) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  for (i64 rr = blockIdx.y; rr < n; rr += gridDim.y) {
    double v;
    if (rr < n_real && c < n_real) v = scale * B[rr * ldb + c] + (rr == c ? 1.0 : 0.0);
    else v = (rr == c) ? 1.0 : 0.0;
    B[rr * ldb + c] = v;
  }
}

// dst[r][c] = src[r][c] * s[c]   (column scaling; conditionals.py:107, the 2-D q_sqrt term)
__global__ __launch_bounds__(256) void scale_cols_kernel(const double* __restrict__ src, i64 lds_, i64 rows, i64 cols,
                                                         const double* __restrict__ sc, double* __restrict__ dst, i64 ldd) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const double f = sc[c];
  for (i64 rr = blockIdx.y; rr < rows; rr += gridDim.y) dst[rr * ldd + c] = src[rr * lds_ + c] * f;
}

// A[r][c] *= s[r]   (row scaling in place: the FITC weights 1/sqrt(nu_i), sgpr.py:244)
__global__ __launch_bounds__(256) void scale_rows_kernel(double* __restrict__ A, i64 lda, i64 rows, i64 cols,
                                                         const double* __restrict__ sc) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (i64 rr = blockIdx.y; rr < rows; rr += gridDim.y) A[rr * lda + c] *= sc[rr];
}

// dst[c][r] = src[r][c]  (32x32 LDS tiles)
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ src, i64 lds_,
                                                        i64 rows, i64 cols,
                                                        double* __restrict__ dst, i64 ldd) {
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const i64 r0 = (i64)blockIdx.y * 32, c0 = (i64)blockIdx.x * 32;
  for (int j = ty; j < 32; j += 8) {
    const i64 rr = r0 + j, cc = c0 + tx;
    t[j][tx] = (rr < rows && cc < cols) ? src[rr * lds_ + cc] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const i64 cc = c0 + j, rr = r0 + tx;      // dst row = src col
    if (cc < cols && rr < rows) dst[cc * ldd + rr] = t[tx][j];
  }
}

// dst [np][np] = scale * tril(src [n][n]) (T = false) or its transpose (T = true), zero elsewhere (padding included): the
// upper-triangular L_q^T the q_sqrt product of the conditional wants (conditionals.py:112-113: tf.matrix_band_part(q_sqrt, -1, 0)),
// from the user's row-major q_sqrt as uploaded.  (The host loop that did this -- a strided walk over 134 MB at M = 4096 -- was
// 0.13 s of config 5's 0.70 s bound.)
template <bool T>
__global__ __launch_bounds__(256) void tril_pad_kernel(const double* __restrict__ src, i64 n, double* __restrict__ dst, i64 np, double scale) {
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const i64 r0 = (i64)blockIdx.y * 32, c0 = (i64)blockIdx.x * 32;      // tile of dst: rows r0.., columns c0..
  if (!T) {
    for (int j = ty; j < 32; j += 8) {
      const i64 a = r0 + j, b = c0 + tx;
      if (a < np && b < np) dst[a * np + b] = (a < n && b <= a) ? scale * src[a * n + b] : 0.0;
    }
    return;
  }
  for (int j = ty; j < 32; j += 8) {
    const i64 a = c0 + j, b = r0 + tx;                       // dst[b][a] = src[a][b] for b <= a < n
    t[j][tx] = (a < n && b <= a) ? scale * src[a * n + b] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const i64 b = r0 + j, a = c0 + tx;
    if (b < np && a < np) dst[b * np + a] = t[tx][j];
  }
}
int gps_launch_tril_pad(gps_handle_t h, const double* src, i64 n, double* dst, i64 np, double scale, int transpose) {
  if (np <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, 0.0, 8.0 * (0.5 * n * n + (double)np * np));
  dim3 grid((unsigned)((np + 31) / 32), (unsigned)((np + 31) / 32));
  if (transpose) hipLaunchKernelGGL(tril_pad_kernel<true>, grid, dim3(256), 0, h->stream, src, n, dst, np, scale);
  else hipLaunchKernelGGL(tril_pad_kernel<false>, grid, dim3(256), 0, h->stream, src, n, dst, np, scale);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// dst[b][c][r] = src[b][r][c] for a batch of 128x128 blocks (grid = 4 x 4 x batch): all block inverses of a factor
// are transposed by one launch after the factorisation instead of 128 KB of extra stores inside every potrf_base
__global__ __launch_bounds__(256) void transpose_blocks_kernel(const double* __restrict__ src, double* __restrict__ dst) {
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const i64 base = (i64)blockIdx.z * 128 * 128;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int j = ty; j < 32; j += 8) t[j][tx] = src[base + (i64)(r0 + j) * 128 + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8) dst[base + (i64)(c0 + j) * 128 + r0 + tx] = t[tx][j];
}

// The 128 x 128 blocks src[b] laid out along the diagonals of wide blocks: dst is [nblk * 128, wb] (diagonal blocks of wb columns
// stacked), block b goes to rows 128 b .., columns (128 b) % wb ..  (level 0 of the wide inverse blocks: gps_gpr.hip)
__global__ __launch_bounds__(256) void blocks_to_diag_kernel(const double* __restrict__ src, double* __restrict__ dst, i64 wb) {
  const i64 b = blockIdx.y;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), c = (threadIdx.x & 63) * 2;
  const double2 v = *reinterpret_cast<const double2*>(src + b * 128 * 128 + (i64)r * 128 + c);
  *reinterpret_cast<double2*>(dst + (b * 128 + r) * wb + (b * 128) % wb + c) = v;
}

// dst [prow, pcol] <- src [rows, cols] zero padded; identity_pad: dst[i][i] = 1 for i >= rows;
// diag_add added on the real diagonal.
__global__ __launch_bounds__(256) void pad_copy_kernel(const double* __restrict__ src, i64 lds_,
                                                       i64 rows, i64 cols,
                                                       double* __restrict__ dst, i64 ldd, i64 prow,
                                                       i64 pcol, int identity_pad,
                                                       double diag_add) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= pcol) return;
  for (i64 rr = blockIdx.y; rr < prow; rr += gridDim.y) {
    double v = 0.0;
    if (rr < rows && c < cols) {
      v = src[rr * lds_ + c];
      if (rr == c) v += diag_add;
    } else if (identity_pad && rr == c) {
      v = 1.0;
    }
    dst[rr * ldd + c] = v;
  }
}

__global__ __launch_bounds__(256) void extract_kernel(const double* __restrict__ src, i64 lds_,
                                                      i64 rows, i64 cols,
                                                      double* __restrict__ dst, i64 ldd,
                                                      int lower_only) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (i64 rr = blockIdx.y; rr < rows; rr += gridDim.y)
    dst[rr * ldd + c] = (lower_only && c > rr) ? 0.0 : src[rr * lds_ + c];
}

// Look-ahead hand-over: one wave.  Publishes *sig = sval (if sig), then waits until *flag >= val (if flag) -- bounded, so a
// broken schedule shows up as a counted time-out (and a wrong result the caller rejects), never as a hung GPU.
__global__ __launch_bounds__(64) void la_wait_kernel(unsigned long long* sig, unsigned long long sval,
                                                     const unsigned long long* flag, unsigned long long val,
                                                     unsigned long long* timeouts) {
  if (threadIdx.x != 0) return;
  if (sig) __hip_atomic_store(sig, sval, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (!flag) return;
  // bounded by the 100 MHz wall clock: 1 s is four orders of magnitude beyond any legitimate hand-over wait (tens of
  // microseconds), still not a hang; the evaluation is then re-run without look-ahead (gps_ops.hpp: with_la_retry)
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= val) return;
    if (wall_clock64() - t0 > 100000000ull) break;
    __builtin_amdgcn_s_sleep(4);
  }
  atomicAdd(timeouts, 1ull);
}

int gps_launch_la_wait(gps_handle_t h, hipStream_t st, unsigned long long* sig, unsigned long long sval,
                       const unsigned long long* flag, unsigned long long val, unsigned long long* timeouts) {
  hipLaunchKernelGGL(la_wait_kernel, dim3(1), dim3(64), 0, st, sig, sval, flag, val, timeouts);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// ---------------------------------------------------------------------------------
int gps_launch_trsv_base(gps_handle_t h, const double* LinvT_blk, double* y, i64 ldy, i64 r) {
  LaunchScope ls(h, KC_TRSV, 2.0 * 128 * 128 * r, 128.0 * 128 * 8);
  hipLaunchKernelGGL(trsv_base_kernel, dim3((unsigned)r), dim3(256), 0, h->stream, LinvT_blk, y, ldy);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_gemv_sub(gps_handle_t h, const double* L21, i64 ldl, i64 n2, i64 n1,
                        const double* y1, double* y2, i64 ldy, i64 r) {
  if (n2 <= 0 || n1 <= 0) return GPS_OK;
  i64 blocks = (n2 + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  for (i64 r0 = 0; r0 < r; r0 += 4) {
    const int rc = (int)((r - r0) < 4 ? (r - r0) : 4);
    LaunchScope ls(h, KC_TRSV, 2.0 * n2 * n1 * rc, (double)n2 * n1 * 8.0);
    hipLaunchKernelGGL(gemv_sub_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, h->stream, L21,
                       ldl, n2, n1, y1, y2, ldy, (int)r0, rc);
    GPS_HIP(h, hipGetLastError());
  }
  return GPS_OK;
}

int gps_launch_gemv_t_sub(gps_handle_t h, const double* L21, i64 ldl, i64 n2, i64 n1,
                          const double* y2, double* y1, i64 ldy, i64 r) {
  if (n2 <= 0 || n1 <= 0) return GPS_OK;
  const i64 cb = n1 / 128;
  i64 S = (1024 + cb - 1) / cb;                 // aim at ~1024 workgroups ...
  if (S > n2 / 128) S = n2 / 128;               // ... of at least 128 rows each
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  const size_t ws_bytes = (size_t)cb * S * 4 * 128 * sizeof(double), cnt_bytes = (size_t)cb * sizeof(unsigned);
  GPS_HIP(h, h->dGemvWs.ensure(ws_bytes));
  if (h->dGemvCnt.cap < cnt_bytes) {          // the counters persist (every launch leaves them at zero)
    GPS_HIP(h, h->dGemvCnt.ensure(cnt_bytes));
    GPS_HIP(h, hipMemsetAsync(h->dGemvCnt.p, 0, h->dGemvCnt.cap, h->stream));
  }
  for (i64 r0 = 0; r0 < r; r0 += 4) {
    const int rc = (int)((r - r0) < 4 ? (r - r0) : 4);
    LaunchScope ls(h, KC_TRSV, 2.0 * n2 * n1 * rc, (double)n2 * n1 * 8.0);
    hipLaunchKernelGGL(gemv_t_sub_kernel<4>, dim3((unsigned)cb, (unsigned)S), dim3(256), 0, h->stream, L21, ldl,
                       n2, n1, y2, y1, ldy, (int)r0, rc, h->dGemvWs.d(), (unsigned*)h->dGemvCnt.p);
    GPS_HIP(h, hipGetLastError());
  }
  return GPS_OK;
}

#define LML_BLOCKS 64
int gps_launch_lml_reduce(gps_handle_t h, const double* L, i64 ldl, i64 n, const double* alpha,
                          i64 ldy, i64 r, double* out_partials) {
  LaunchScope ls(h, KC_REDUCE, 0.0, (double)n * (64.0 + 8.0 * r));
  hipLaunchKernelGGL(lml_reduce_kernel, dim3(LML_BLOCKS), dim3(256), 0, h->stream, L, ldl, n, alpha,
                     ldy, r, out_partials);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_rowdot(gps_handle_t h, const double* At, i64 ldat, i64 n_new, i64 npad,
                      const double* alpha, i64 ldy, i64 r, double* mean, double* sumsq) {
  if (n_new <= 0) return GPS_OK;
  LaunchScope ls(h, KC_REDUCE, 2.0 * n_new * npad * (r + 1), (double)n_new * npad * 8.0);
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)n_new), dim3(256), 0, h->stream, At, ldat, npad,
                     alpha, ldy, (int)r, mean, sumsq);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// out[j] = sum_i A[i][j]^2 over the rows of A [rows, ld], j < cols (cols even up to padding: ld >= cols rounded up to 2).
// Thread = two adjacent columns (16-byte loads along the row), workgroup = 512 columns, blockIdx.y = a slice of the
// rows; slices > 1: partial sums [slices][cols] are folded in slice order by colsum_fold_kernel (no atomics).
__global__ __launch_bounds__(256) void colsumsq_kernel(const double* __restrict__ A, i64 ld, i64 rows, i64 cols,
                                                       double* __restrict__ out, i64 out_stride) {
  const i64 j = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
  if (j >= cols) return;
  const i64 per = (rows + gridDim.y - 1) / gridDim.y;
  const i64 i0 = (i64)blockIdx.y * per, i1 = min(rows, i0 + per);
  double s0 = 0.0, s1 = 0.0;
  const double* p = A + i0 * ld + j;
  i64 i = i0;
  for (; i + 8 <= i1; i += 8) {
    v2d v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const v2d*>(p + (i64)u * ld);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 = fma(v[u].x, v[u].x, s0); s1 = fma(v[u].y, v[u].y, s1); }
    p += 8 * ld;
  }
  for (; i < i1; ++i) {
    const v2d v = *reinterpret_cast<const v2d*>(p);
    s0 = fma(v.x, v.x, s0); s1 = fma(v.y, v.y, s1);
    p += ld;
  }
  double* o = out + (i64)blockIdx.y * out_stride + j;
  o[0] = s0;
  if (j + 1 < cols) o[1] = s1;
}
__global__ __launch_bounds__(256) void colsum_fold_kernel(const double* __restrict__ part, i64 stride, int slices, i64 cols,
                                                          double* __restrict__ out) {
  const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  double s = 0.0;
  for (int q = 0; q < slices; ++q) s += part[(i64)q * stride + j];
  out[j] = s;
}

// sumsq[j] = sum_i A[i][j]^2 for A [rows, ld] row-major (the column form of rowdot's sum of squares: conditionals.py:109
// on (L_q^T A) stored [M, N] instead of [N, M])
int gps_launch_colsumsq(gps_handle_t h, const double* A, i64 ld, i64 rows, i64 cols, double* sumsq) {
  if (rows <= 0 || cols <= 0) return GPS_OK;
  if ((ld & 1) || ((uintptr_t)A & 15)) return gps_fail(h, GPS_ERR_ARG, "colsumsq: operand must be 16-byte aligned with even leading dimension");
  const i64 gx = (cols + 511) / 512;
  i64 slices = (2048 + gx - 1) / gx;                    // aim at ~2048 workgroups
  if (slices > rows / 64) slices = rows / 64;
  if (slices > 64) slices = 64;
  if (slices < 1) slices = 1;
  LaunchScope ls(h, KC_REDUCE, 2.0 * rows * cols, (double)rows * cols * 8.0);
  if (slices == 1) {
    hipLaunchKernelGGL(colsumsq_kernel, dim3((unsigned)gx, 1), dim3(256), 0, h->stream, A, ld, rows, cols, sumsq, (i64)0);
    GPS_HIP(h, hipGetLastError());
    return GPS_OK;
  }
  GPS_HIP(h, h->dGemvWs.ensure((size_t)slices * cols * 8));
  hipLaunchKernelGGL(colsumsq_kernel, dim3((unsigned)gx, (unsigned)slices), dim3(256), 0, h->stream, A, ld, rows, cols,
                     h->dGemvWs.d(), cols);
  GPS_HIP(h, hipGetLastError());
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, h->stream, h->dGemvWs.d(), cols,
                     (int)slices, cols, sumsq);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_varexp(gps_handle_t h, const double* fmean, const double* yres, i64 k, int q, const double* base,
                      const double* extra, i64 n, double* partial64) {
  if (n <= 0) return GPS_OK;
  LaunchScope ls(h, KC_REDUCE, 4.0 * n, 32.0 * n);
  hipLaunchKernelGGL(varexp_kernel, dim3(64), dim3(256), 0, h->stream, fmean, yres, (int)k, q, base, extra, n, partial64);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_dist_tail(gps_handle_t h, const double* partials64x2, const int* d_info, double* tail_msg, double* tail_own) {
  hipLaunchKernelGGL(dist_tail_kernel, dim3(1), dim3(1), 0, h->stream, partials64x2, d_info, tail_msg, tail_own);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_svgp_et(gps_handle_t h, const double* yres, const double* fmean, i64 k, i64 n, i64 npad, double coef, double* Et) {
  LaunchScope ls(h, KC_OTHER, 2.0 * n * k, 24.0 * n * k);
  hipLaunchKernelGGL(svgp_et_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, yres, fmean, (int)k, n, npad, coef, Et);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
int gps_launch_svgp_abar(gps_handle_t h, const double* Bt, i64 ld, i64 rows, i64 cols, const double* coef, const double* Et, i64 lde,
                         const double* qmu, i64 k, double* Abar) {
  LaunchScope ls(h, KC_OTHER, 2.0 * rows * cols * (k + 1), 16.0 * rows * cols);
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
  hipLaunchKernelGGL(svgp_abar_kernel, grid, dim3(256), 0, h->stream, Bt, ld, rows, cols, coef, Et, lde, qmu, (int)k, Abar);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
// A <- alpha A + beta I on the leading [n_real, n_real] block of an [n, n] matrix, zero elsewhere
__global__ __launch_bounds__(256) void axpby_eye_kernel(double* __restrict__ A, i64 ld, i64 n, i64 n_real, double alpha, double beta) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  for (i64 rr = blockIdx.y; rr < n; rr += gridDim.y)
    A[rr * ld + c] = (rr < n_real && c < n_real) ? alpha * A[rr * ld + c] + (rr == c ? beta : 0.0) : 0.0;
}
// A[i][i] += coef / L[i][i], i < n
__global__ __launch_bounds__(256) void diag_recip_add_kernel(double* __restrict__ A, i64 lda, const double* __restrict__ L, i64 ldl,
                                                             i64 n, double coef) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[i * lda + i] += coef / L[i * ldl + i];
}
// part[2 b] = sum over the rows i = b (mod 64) of sum_{j <= i} A[i][j] B[i][j] ; part[2 b + 1] = their share of trace(A)
__global__ __launch_bounds__(256) void tri_dot_kernel(const double* __restrict__ A, i64 lda, const double* __restrict__ B, i64 ldb,
                                                      i64 n, double* __restrict__ part) {
  __shared__ double red[2][4];
  double s = 0.0, t = 0.0;
  for (i64 i = blockIdx.x; i < n; i += gridDim.x) {
    for (i64 j = threadIdx.x; j <= i; j += 256) s = fma(A[i * lda + j], B[i * ldb + j], s);
    if (threadIdx.x == 0) t += A[i * lda + i];
  }
  s = wave_sum(s); t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// out[i] = sum_j A[i][j] B[i][j]   (one wave per row, 16-byte loads; cols a multiple of 2)
__global__ __launch_bounds__(256) void rowdot2_kernel(const double* __restrict__ A, i64 lda, const double* __restrict__ B, i64 ldb,
                                                      i64 rows, i64 cols, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const i64 row = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const double* a = A + row * lda; const double* b = B + row * ldb;
  double s = 0.0;
  for (i64 k = 2 * lane; k < cols; k += 128) {
    const v2d x = *reinterpret_cast<const v2d*>(a + k), y = *reinterpret_cast<const v2d*>(b + k);
    s = fma(x.x, y.x, fma(x.y, y.y, s));
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}
// X[i][:] = a[i] X[i][:] + b[i] Y[i][:]
__global__ __launch_bounds__(256) void rows_axpby_kernel(double* __restrict__ X, i64 ldx, const double* __restrict__ Y, i64 ldy,
                                                         i64 rows, i64 cols, const double* __restrict__ a, const double* __restrict__ b) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (i64 rr = blockIdx.y; rr < rows; rr += gridDim.y) X[rr * ldx + c] = a[rr] * X[rr * ldx + c] + b[rr] * Y[rr * ldy + c];
}
int gps_launch_rowdot2(gps_handle_t h, const double* A, i64 lda, const double* B, i64 ldb, i64 rows, i64 cols, double* out) {
  if (rows <= 0 || cols <= 0) return GPS_OK;
  LaunchScope ls(h, KC_REDUCE, 2.0 * rows * cols, 16.0 * rows * cols);
  hipLaunchKernelGGL(rowdot2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->stream, A, lda, B, ldb, rows, cols, out);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
int gps_launch_rows_axpby(gps_handle_t h, double* X, i64 ldx, const double* Y, i64 ldy, i64 rows, i64 cols, const double* a,
                          const double* b) {
  if (rows <= 0 || cols <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, 3.0 * rows * cols, 24.0 * rows * cols);
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
  hipLaunchKernelGGL(rows_axpby_kernel, grid, dim3(256), 0, h->stream, X, ldx, Y, ldy, rows, cols, a, b);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
int gps_launch_axpby_eye(gps_handle_t h, double* A, i64 ld, i64 n, i64 n_real, double alpha, double beta) {
  if (n <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, (double)n * n, 16.0 * n * n);
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)(n < 32768 ? n : 32768));
  hipLaunchKernelGGL(axpby_eye_kernel, grid, dim3(256), 0, h->stream, A, ld, n, n_real, alpha, beta);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
int gps_launch_diag_recip_add(gps_handle_t h, double* A, i64 lda, const double* L, i64 ldl, i64 n, double coef) {
  if (n <= 0) return GPS_OK;
  hipLaunchKernelGGL(diag_recip_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, A, lda, L, ldl, n, coef);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
// out2[0] = sum_{j <= i < n} A[i][j] B[i][j], out2[1] = trace(A)   (host values; synchronises the stream)
int gps_tri_dot(gps_handle_t h, const double* A, i64 lda, const double* B, i64 ldb, i64 n, double* out2) {
  out2[0] = out2[1] = 0.0;
  if (n <= 0) return GPS_OK;
  double* part = h->dScal.d();
  {
    LaunchScope ls(h, KC_REDUCE, (double)n * n, 8.0 * n * n);
    hipLaunchKernelGGL(tri_dot_kernel, dim3(64), dim3(256), 0, h->stream, A, lda, B, ldb, n, part);
    GPS_HIP(h, hipGetLastError());
  }
  double hp[128];
  GPS_HIP(h, hipMemcpyAsync(hp, part, sizeof(hp), hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  for (int b = 0; b < 64; ++b) { out2[0] += hp[2 * b]; out2[1] += hp[2 * b + 1]; }
  return GPS_OK;
}

int gps_launch_tri_map(gps_handle_t h, double* A, i64 ld, i64 n, int mode) {
  if (n <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, (double)n * n, 16.0 * n * n);
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)(n < 32768 ? n : 32768));
  hipLaunchKernelGGL(tri_map_kernel, grid, dim3(256), 0, h->stream, A, ld, n, mode);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_fill_info(gps_handle_t h, int* d_info, int value) {
  hipLaunchKernelGGL(fill_info_kernel, dim3(1), dim3(1), 0, h->stream, d_info, value);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_var_finish(gps_handle_t h, double* var, const double* kdiag_or_null,
                          double kdiag_const, const double* sumsq, i64 n) {
  if (n <= 0) return GPS_OK;
  LaunchScope ls(h, KC_REDUCE, (double)n, 16.0 * n);
  hipLaunchKernelGGL(var_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream,
                     var, kdiag_or_null, kdiag_const, sumsq, n);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_transpose(gps_handle_t h, const double* src, i64 lds_, i64 rows, i64 cols,
                         double* dst, i64 ldd) {
  if (rows <= 0 || cols <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, 0.0, 16.0 * rows * cols);
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, h->stream, src, lds_, rows, cols, dst, ldd);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// kappa_2 of every 128 x 128 diagonal block of a factor, from the block and its explicit inverse W_j: ||L_jj||_2 ||W_j||_2 with both
// norms by 16 steps of power iteration on A^T A (it converges from below, hence the factor 1.5; the start vector mixes a constant,
// an alternating and a pseudo-random part -- round 6, ADVICE: from the all-ones vector alone the iteration starts nearly orthogonal
// to the oscillatory dominant singular vectors of W_j = L_jj^-1 of a smooth Gram block and could under-estimate a small spectral gap),
// capped by the rigorous bound sqrt(||A||_1 ||A||_inf) of each norm (the norm bounds alone are 6 - 50 x too pessimistic on the
// blocks of an RBF Gram matrix).  One workgroup per block, thread t: row t / column t.  What decides, per leaf, whether the
// product with the explicit inverse IS the solve or only its preconditioner (gps_ops.hpp: classify_blocks).
// (the block sits in LDS, row stride 129: thread t walks row t -- stride 129 across the lanes, conflict-free -- for A x, and
// column t -- consecutive lanes, consecutive words -- for A^T y.  From global memory, where a row walk is a 1 KB stride across
// the lanes, the launch took 0.63 ms; round 5)
#define BC_LS 129
__device__ __forceinline__ double bc_norm2(const double* __restrict__ A, i64 lda, double* As, double* x, double* y, double* red, int t) {
  // A lower triangular [128][128]; returns min(1.5 * power-iteration estimate of ||A||_2, sqrt(||A||_1 ||A||_inf))
  __syncthreads();                                        // (the image of the matrix before)
  for (int r0 = 0; r0 < 128; r0 += 32) {                  // 32 rows in flight (one at a time: 128 memory latencies)
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = A[(i64)(r0 + u) * lda + t];
#pragma unroll
    for (int u = 0; u < 32; ++u) As[(r0 + u) * BC_LS + t] = (t <= r0 + u) ? v[u] : 0.0;
  }
  __syncthreads();
  // sums / maxima over the 128 threads: inside the two waves by lane shuffles, across them through red[0 .. 1]
  auto wave_sum = [](double v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64); return v; };
  auto wave_max = [](double v) { for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64)); return v; };
  double cs = 0.0, rs = 0.0;
  // (whole rows and columns of the zero-filled image, unrolled: with the triangular trip counts every LDS read waited for the one before)
#pragma unroll 16
  for (int r = 0; r < 128; ++r) cs += fabs(As[r * BC_LS + t]);
#pragma unroll 16
  for (int q = 0; q < 128; ++q) rs += fabs(As[t * BC_LS + q]);
  cs = wave_max(cs); rs = wave_max(rs);
  if ((t & 63) == 0) { red[t >> 6] = cs; red[2 + (t >> 6)] = rs; }
  // start vector: constant + alternating + pseudo-random (a fixed integer hash of t): components along smooth AND oscillatory directions
  const double x0 = 1.0 + ((t & 1) ? -0.75 : 0.75) + 0.25 * ((double)((((unsigned)t * 2654435761u) >> 16) & 0xffu) / 127.5 - 1.0);
  x[t] = x0;
  const double xx = wave_sum(x0 * x0);
  if ((t & 63) == 0) red[8 + (t >> 6)] = xx;
  __syncthreads();
  const double bound = sqrt(fmax(red[0], red[1]) * fmax(red[2], red[3]));
  double est = 0.0, nx = sqrt(red[8] + red[9]);                                   // (||x|| = 1 from the second step on)
  for (int it = 0; it < 16; ++it) {
    double s = 0.0;
#pragma unroll 16
    for (int q = 0; q < 128; ++q) s = fma(As[t * BC_LS + q], x[q], s);           // y = A x
    y[t] = s;
    __syncthreads();
    double z = 0.0;
#pragma unroll 16
    for (int r = 0; r < 128; ++r) z = fma(As[r * BC_LS + t], y[r], z);           // z = A^T y
    const double zz = wave_sum(z * z);
    if ((t & 63) == 0) red[4 + (it & 1) * 2 + (t >> 6)] = zz;                     // (two slots: the next step's write does not race this step's reads)
    __syncthreads();
    const double nz = sqrt(red[4 + (it & 1) * 2] + red[5 + (it & 1) * 2]);
    est = sqrt(nz / nx);                                                          // ||A^T A x|| / ||x|| -> sigma_max^2
    nx = 1.0;
    x[t] = z / nz;
    __syncthreads();
  }
  return fmin(1.5 * est, bound);
}

__global__ __launch_bounds__(128) void block_cond_kernel(const double* __restrict__ L, i64 ldl, const double* __restrict__ W, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char bc_smem[];
  double* As = reinterpret_cast<double*>(bc_smem);        // [128][BC_LS]
  double* x = As + 128 * BC_LS; double* y = x + 128; double* red = y + 128;
  const int j = blockIdx.x, t = threadIdx.x;
  const double nl = bc_norm2(L + (i64)j * 128 * ldl + (i64)j * 128, ldl, As, x, y, red, t);
  const double nw = bc_norm2(W + (i64)j * 128 * 128, 128, As, x, y, red, t);
  if (t == 0) out[j] = nl * nw;
}

int gps_launch_block_cond(gps_handle_t h, const double* L, i64 ldl, const double* W, i64 nblk, double* d_out) {
  if (nblk <= 0) return GPS_OK;
  const size_t lds = (size_t)(128 * BC_LS + 512) * sizeof(double);
  int rc = gps_dyn_lds(h, reinterpret_cast<const void*>(&block_cond_kernel), (int)lds);
  if (rc) return rc;
  LaunchScope ls(h, KC_OTHER, 0.0, 2.0 * 8.0 * 128 * 128 * nblk);
  hipLaunchKernelGGL(block_cond_kernel, dim3((unsigned)nblk), dim3(128), lds, h->stream, L, ldl, W, d_out);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_blocks_to_diag(gps_handle_t h, const double* src, double* dst, i64 nblk, i64 wb) {
  if (nblk <= 0) return GPS_OK;
  if (wb % 128 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return gps_fail(h, GPS_ERR_ARG, "blocks_to_diag: bad layout");
  LaunchScope ls(h, KC_OTHER, 0.0, (double)nblk * 128 * 128 * 16);
  hipLaunchKernelGGL(blocks_to_diag_kernel, dim3(32, (unsigned)nblk), dim3(256), 0, h->stream, src, dst, wb);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_transpose_blocks(gps_handle_t h, const double* src, double* dst, i64 nblk) {
  if (nblk <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, 0.0, 16.0 * 128 * 128 * nblk);
  hipLaunchKernelGGL(transpose_blocks_kernel, dim3(4, 4, (unsigned)nblk), dim3(256), 0, h->stream, src, dst);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_pad_copy(gps_handle_t h, const double* src, i64 lds_, i64 rows, i64 cols,
                        double* dst, i64 ldd, i64 prow, i64 pcol, int identity_pad,
                        double diag_add) {
  if (prow <= 0 || pcol <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, 0.0, 8.0 * (rows * cols + prow * pcol));
  dim3 grid((unsigned)((pcol + 255) / 256), (unsigned)(prow < 32768 ? prow : 32768));
  hipLaunchKernelGGL(pad_copy_kernel, grid, dim3(256), 0, h->stream, src, lds_, rows, cols, dst, ldd,
                     prow, pcol, identity_pad, diag_add);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_extract(gps_handle_t h, const double* src, i64 lds_, i64 rows, i64 cols,
                       double* dst, i64 ldd, int lower_only) {
  if (rows <= 0 || cols <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, 0.0, 16.0 * rows * cols);
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
  hipLaunchKernelGGL(extract_kernel, grid, dim3(256), 0, h->stream, src, lds_, rows, cols, dst, ldd,
                     lower_only);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_scale_add_eye(gps_handle_t h, double* B, i64 ldb, i64 n, i64 n_real, double scale) {
  if (n <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, (double)n * n, 16.0 * n * n);
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)(n < 32768 ? n : 32768));
  hipLaunchKernelGGL(scale_add_eye_kernel, grid, dim3(256), 0, h->stream, B, ldb, n, n_real, scale);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_scale_rows(gps_handle_t h, double* A, i64 lda, i64 rows, i64 cols, const double* sc) {
  if (rows <= 0 || cols <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, (double)rows * cols, 16.0 * rows * cols);
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
  hipLaunchKernelGGL(scale_rows_kernel, grid, dim3(256), 0, h->stream, A, lda, rows, cols, sc);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_scale_cols(gps_handle_t h, const double* src, i64 lds_, i64 rows, i64 cols, const double* sc,
                          double* dst, i64 ldd) {
  if (rows <= 0 || cols <= 0) return GPS_OK;
  LaunchScope ls(h, KC_OTHER, (double)rows * cols, 16.0 * rows * cols);
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
  hipLaunchKernelGGL(scale_cols_kernel, grid, dim3(256), 0, h->stream, src, lds_, rows, cols, sc, dst, ldd);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
