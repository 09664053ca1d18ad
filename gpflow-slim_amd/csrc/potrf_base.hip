// Base case of the blocked Cholesky: factor one 128x128 diagonal block inside a
// single workgroup and invert the factor.
//
// Replaces the innermost part of tf.cholesky (models/gpr.py:70,121;
// conditionals.py:84).  The explicit 128x128 inverse turns every panel solve
// X L11^T = B into an MFMA GEMM (gemm_f64.hip) and every vector solve into a
// 128x128 gemv, which is what makes the recursive trsm / trsv GEMM-only.
//
// Factorisation: the block lives in REGISTERS, 2-D cyclic over a 16x16 thread grid
// (thread (ty,tx) owns A[ty+16a][tx+16b], an 8x8 patch).  Right-looking elimination with
// deferred column scaling, one barrier per column: the owners of column j publish it (and
// 1/pivot) into a double-buffered LDS vector, everybody updates its patch with
// a_ik -= a_ij a_kj / a_jj.  The 16-column groups are unrolled at compile time so that the
// patch is indexed statically (no scratch) and finished groups / the upper triangle cost
// nothing.
//
// Inverse: LDS image [128][129]; the lower triangle holds L, the strictly upper triangle
// receives inv(L)^T as it is built (inv(L) is lower triangular, so its transpose fits exactly
// there), 1/L_ii sits in dinv[].  16x16 diagonal blocks by substitution, then three
// doubling levels X21 = -X22 L21 X11.  Stride 129 makes row and column walks conflict free.
#include "gps_common.hpp"

#define PB 128
#define PS 129
#define NT 256

typedef double v4d __attribute__((ext_vector_type(4)));
#define CSTRIDE (PB + 8)

// publish column j (held in patch column B, rows a >= B) and its pivot into the LDS vector cb
template <int B>
__device__ __forceinline__ void publish_column(const double (&v)[8][8], double* cb, double* piv,
                                               int* info, int row0, int j, int ty, bool diag_owner) {
#pragma unroll
  for (int a = B; a < 8; ++a) cb[ty + 16 * a] = v[a][B];
  if (diag_owner) {
    const double p = v[B][B];
    if (!(p > 0.0)) atomicMin(info, row0 + j + 1);
    cb[PB] = 1.0 / p;
    piv[j] = p;
  }
}

// one 16-column group of the elimination (BJ compile-time so v[][] is statically indexed).
// Column j+1 is updated and published FIRST in every step, so that its LDS write and the
// barrier overlap with the bulk of the rank-1 update.
template <int BJ>
__device__ __forceinline__ void eliminate_group(double (&v)[8][8], double* col, double* piv,
                                                int* info, int row0, int tx, int ty) {
#pragma unroll 1
  for (int tj = 0; tj < 16; ++tj) {
    const int j = BJ * 16 + tj;
    double* cb = col + (j & 1) * CSTRIDE;
    double* nb = col + ((j + 1) & 1) * CSTRIDE;
    __syncthreads();                               // column j is visible
    const double inv_p = cb[PB];
    double ci[8], ck[8];
#pragma unroll
    for (int a = BJ; a < 8; ++a) {
      ci[a] = cb[ty + 16 * a] * inv_p;
      ck[a] = cb[tx + 16 * a];
    }
    if (tj < 15) {
      if (tx > tj) {
#pragma unroll
        for (int a = BJ; a < 8; ++a) v[a][BJ] -= ci[a] * ck[BJ];
        if (tx == tj + 1) publish_column<BJ>(v, nb, piv, info, row0, j + 1, ty, ty == tj + 1);
      }
#pragma unroll
      for (int b = BJ + 1; b < 8; ++b)
#pragma unroll
        for (int a = b; a < 8; ++a) v[a][b] -= ci[a] * ck[b];
    } else {
      if constexpr (BJ < 7) {
#pragma unroll
        for (int a = BJ + 1; a < 8; ++a) v[a][BJ + 1] -= ci[a] * ck[BJ + 1];
        if (tx == 0) publish_column<BJ + 1>(v, nb, piv, info, row0, j + 1, ty, ty == 0);
#pragma unroll
        for (int b = BJ + 2; b < 8; ++b)
#pragma unroll
          for (int a = b; a < 8; ++a) v[a][b] -= ci[a] * ck[b];
      }
    }
  }
}

// ---- MFMA helpers for the inverse levels --------------------------------------------------------
// X (= inv L, lower) is stored transposed in the strict upper triangle of the image, diag in dinv.
__device__ __forceinline__ double x_elem(const double* a, const double* dinv, int r, int c) {
  return (r > c) ? a[c * PS + r] : ((r == c) ? dinv[r] : 0.0);
}

// factor != 0: A holds the SPD block, L is written back.  factor == 0: A already holds L
// (caller-supplied factor); only the inverse is produced.  LinvT (optional) receives inv(L)^T.
__global__ __launch_bounds__(NT) void potrf_base_kernel(double* __restrict__ A, i64 lda,
                                                        double* __restrict__ Linv,
                                                        double* __restrict__ LinvT,
                                                        int* __restrict__ info, int row0,
                                                        int factor, long long* __restrict__ stamps) {
#define STAMP(q) do { if (stamps && threadIdx.x == 0) { stamps[q] = (long long)wall_clock64(); stamps[8 + q] = (long long)clock64(); } } while (0)
  STAMP(0);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* a = reinterpret_cast<double*>(smem_raw);   // [PB][PS]
  double* dinv = a + PB * PS;                        // [PB]
  double* piv = dinv + PB;                           // [PB]
  double* col = piv + PB;                            // [2][PB + 8]

  const int tid = threadIdx.x;

  if (factor) {
    const int tx = tid & 15, ty = tid >> 4;
    double v[8][8];
#pragma unroll
    for (int a_ = 0; a_ < 8; ++a_)
#pragma unroll
      for (int b_ = 0; b_ < 8; ++b_) {
        const int i = ty + 16 * a_, k = tx + 16 * b_;
        v[a_][b_] = (k <= i) ? A[(i64)i * lda + k] : 0.0;
      }
    for (int q = tid; q < 2 * CSTRIDE; q += NT) col[q] = 0.0;
    __syncthreads();
    if (tx == 0) publish_column<0>(v, col, piv, info, row0, 0, ty, ty == 0);
    STAMP(1);
    eliminate_group<0>(v, col, piv, info, row0, tx, ty);
    eliminate_group<1>(v, col, piv, info, row0, tx, ty);
    eliminate_group<2>(v, col, piv, info, row0, tx, ty);
    eliminate_group<3>(v, col, piv, info, row0, tx, ty);
    eliminate_group<4>(v, col, piv, info, row0, tx, ty);
    eliminate_group<5>(v, col, piv, info, row0, tx, ty);
    eliminate_group<6>(v, col, piv, info, row0, tx, ty);
    eliminate_group<7>(v, col, piv, info, row0, tx, ty);
    __syncthreads();
    STAMP(2);
    // scale: L_jj = sqrt(a_jj), L_ij = a_ij / L_jj ; image to LDS
    if (tid < PB) {
      const double d = sqrt(piv[tid]);
      col[tid] = d;                 // reuse: col[0..127] = L_jj
      dinv[tid] = 1.0 / d;
    }
    __syncthreads();
#pragma unroll
    for (int a_ = 0; a_ < 8; ++a_)
#pragma unroll
      for (int b_ = 0; b_ <= a_; ++b_) {
        const int i = ty + 16 * a_, k = tx + 16 * b_;
        if (k < i) a[i * PS + k] = v[a_][b_] / col[k];
        else if (k == i) a[i * PS + k] = col[k];
      }
    __syncthreads();
    // L back to HBM (upper triangle of the diagonal block zero-filled, like tf.cholesky)
    for (int idx = tid; idx < PB * PB; idx += NT) {
      const int i = idx >> 7, j = idx & 127;
      A[(i64)i * lda + j] = (j <= i) ? a[i * PS + j] : 0.0;
    }
  } else {
    for (int idx = tid; idx < PB * PB; idx += NT) {
      const int i = idx >> 7, j = idx & 127;
      if (j <= i) a[i * PS + j] = A[(i64)i * lda + j];
    }
    __syncthreads();
    if (tid < PB) dinv[tid] = 1.0 / a[tid * PS + tid];
    __syncthreads();
  }

  STAMP(3);
  // ---- inverse, level 0: the eight 16x16 diagonal blocks, one column per thread.
  // X[i][c] is stored at a[c][i] (i > c); X[c][c] = dinv[c].
  if (tid < PB) {
    const int c = tid;
    const int e = (c | 15);                    // last row of this 16-block
    for (int i = c + 1; i <= e; ++i) {
      double s = a[i * PS + c] * dinv[c];      // k = c term
      for (int k = c + 1; k < i; ++k) s += a[i * PS + k] * a[c * PS + k];
      a[c * PS + i] = -s * dinv[i];
    }
  }
  __syncthreads();

  STAMP(4);
  // ---- levels s = 16, 32, 64: X21 = -X22 * (L21 * X11) for each pair of s-blocks, on the
  // fp64 MFMA (16x16x4).  Output tiles are dealt round-robin to the 4 waves.
  {
    const int lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    for (int s = 16; s < PB; s <<= 1) {
      const int tps = s >> 4;                          // 16-tiles per side of an s-block
      const int tiles_pair = tps * tps;
      const int ntile = (PB / (2 * s)) * tiles_pair;   // 4, 8, 16
      // step A: W = L21 * X11 ; W[i][c] -> a[c][i]
      for (int t = wave; t < ntile; t += 4) {
        const int pr = t / tiles_pair, w = t - pr * tiles_pair;
        const int o = pr * 2 * s;
        const int i0 = o + s + (w / tps) * 16, c0 = o + (w % tps) * 16;
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int k0 = c0; k0 < o + s; k0 += 4) {       // X11[k][c] = 0 for k < c
          const double av = a[(i0 + fr) * PS + k0 + fk];                 // L21[i][k]
          const double bv = x_elem(a, dinv, k0 + fk, c0 + fr);           // X11[k][c]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) a[(c0 + fr) * PS + i0 + fk + 4 * rg] = acc[rg];
      }
      __syncthreads();
      // step B: Z = -X22 * W, kept in registers until every W has been consumed
      v4d z[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        z[q] = (v4d){0.0, 0.0, 0.0, 0.0};
        const int t = wave + 4 * q;
        if (t < ntile) {
          const int pr = t / tiles_pair, w = t - pr * tiles_pair;
          const int o = pr * 2 * s;
          const int i0 = o + s + (w / tps) * 16, c0 = o + (w % tps) * 16;
          for (int k0 = o + s; k0 < i0 + 16; k0 += 4) {  // X22[i][k] = 0 for k > i
            const double av = x_elem(a, dinv, i0 + fr, k0 + fk);         // X22[i][k]
            const double bv = a[(c0 + fr) * PS + k0 + fk];               // W[k][c]
            z[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, z[q], 0, 0, 0);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = wave + 4 * q;
        if (t < ntile) {
          const int pr = t / tiles_pair, w = t - pr * tiles_pair;
          const int o = pr * 2 * s;
          const int i0 = o + s + (w / tps) * 16, c0 = o + (w % tps) * 16;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) a[(c0 + fr) * PS + i0 + fk + 4 * rg] = -z[q][rg];
        }
      }
      __syncthreads();
    }
  }

  STAMP(5);
  // inverse to HBM, full block with zero upper triangle
  for (int idx = tid; idx < PB * PB; idx += NT) {
    const int i = idx >> 7, c = idx & 127;
    double v = 0.0;
    if (c < i) v = a[c * PS + i];
    else if (c == i) v = dinv[i];
    Linv[idx] = v;
  }
  if (LinvT) {
    for (int idx = tid; idx < PB * PB; idx += NT) {
      const int c = idx >> 7, i = idx & 127;        // LinvT[c][i] = Linv[i][c]
      double v = 0.0;
      if (c < i) v = a[c * PS + i];
      else if (c == i) v = dinv[i];
      LinvT[idx] = v;
    }
  }
  __syncthreads();
  STAMP(6);
#undef STAMP
}

int gps_launch_potrf_base(gps_handle_t h, double* A, i64 lda, double* Linv_blk,
                          double* LinvT_blk, int* d_info, i64 row0, int factor, long long* d_stamps) {
  static bool attr_set = false;
  const size_t lds = (size_t)(PB * PS + 2 * PB + 2 * CSTRIDE) * sizeof(double);
  if (!attr_set) {
    GPS_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_base_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  // potrf n^3/3 + trtri n^3/3
  LaunchScope ls(h, KC_POTRF_BASE, 2.0 * PB * PB * PB / 3.0, 3.0 * PB * PB * 8.0);
  hipLaunchKernelGGL(potrf_base_kernel, dim3(1), dim3(NT), lds, h->stream, A, lda, Linv_blk,
                     LinvT_blk, d_info, (int)row0, factor, d_stamps);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
