// Base case of the blocked Cholesky: factor one 128x128 diagonal block inside a
// single workgroup (LDS-resident) and invert the factor.
//
// Replaces the innermost part of tf.cholesky (models/gpr.py:70,121;
// conditionals.py:84).  The explicit 128x128 inverse turns every panel solve
// X L11^T = B into an MFMA GEMM (gemm_f64.hip) and every vector solve into a
// 128x128 gemv, which is what makes the recursive trsm / trsv GEMM-only.
//
// Layout in LDS: one [128][129] fp64 image.  The lower triangle holds A then L;
// the strictly upper triangle holds inv(L)^T as it is built (inv(L) is lower
// triangular, so its transpose fits exactly there); 1/L_ii sits in dinv[].
// Stride 129 makes both row walks and column walks bank-conflict free.
#include "gps_common.hpp"

#define PB 128
#define PS 129

// factor != 0: A holds the SPD block, L is written back.  factor == 0: A already holds L
// (caller-supplied factor); only the inverse is produced.  LinvT (optional) receives inv(L)^T.
__global__ __launch_bounds__(1024) void potrf_base_kernel(double* __restrict__ A, i64 lda,
                                                          double* __restrict__ Linv,
                                                          double* __restrict__ LinvT,
                                                          int* __restrict__ info, int row0,
                                                          int factor) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* a = reinterpret_cast<double*>(smem_raw);   // [PB][PS]
  double* dinv = a + PB * PS;                        // [PB]

  const int tid = threadIdx.x;
  const int tx = tid & 31, ty = tid >> 5;

  for (int idx = tid; idx < PB * PB; idx += 1024) {
    const int i = idx >> 7, j = idx & 127;
    a[i * PS + j] = (j <= i) ? A[(i64)i * lda + j] : 0.0;
  }
  __syncthreads();

  if (factor) {
  // right-looking elimination with deferred column scaling: column j is final
  // (unscaled) before step j; step j only writes columns > j, so one barrier per
  // column is enough.   a_ik -= a_ij * a_kj / a_jj
  for (int j = 0; j < PB; ++j) {
    const double p = a[j * PS + j];
    if (tid == 0 && !(p > 0.0)) atomicMin(info, row0 + j + 1);
    const double inv_p = 1.0 / p;
    for (int i = j + 1 + ty; i < PB; i += 32) {
      const double lij = a[i * PS + j] * inv_p;
      for (int k = j + 1 + tx; k <= i; k += 32) a[i * PS + k] -= lij * a[k * PS + j];
    }
    __syncthreads();
  }

  // scale: L_jj = sqrt(a_jj), L_ij = a_ij / L_jj
  if (tid < PB) dinv[tid] = sqrt(a[tid * PS + tid]);
  __syncthreads();
  for (int idx = tid; idx < PB * PB; idx += 1024) {
    const int i = idx >> 7, j = idx & 127;
    if (j < i) a[i * PS + j] = a[i * PS + j] / dinv[j];
  }
  __syncthreads();
  if (tid < PB) {
    const double d = dinv[tid];
    a[tid * PS + tid] = d;
    dinv[tid] = 1.0 / d;
  }
  __syncthreads();

  // L back to HBM (upper triangle of the diagonal block zero-filled, like tf.cholesky)
  for (int idx = tid; idx < PB * PB; idx += 1024) {
    const int i = idx >> 7, j = idx & 127;
    A[(i64)i * lda + j] = (j <= i) ? a[i * PS + j] : 0.0;
  }
  } else {
    if (tid < PB) dinv[tid] = 1.0 / a[tid * PS + tid];
    __syncthreads();
  }

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

  // ---- levels s = 16, 32, 64: X21 = -X22 * L21 * X11 for each pair of s-blocks
  for (int s = 16; s < PB; s <<= 1) {
    const int per_pair = s * s;
    const int total = (PB / (2 * s)) * per_pair;      // outputs this level (1024, 2048, 4096)
    // step A: W = L21 * X11, W[i][c] -> a[c][i]
    for (int e = tid; e < total; e += 1024) {
      const int pr = e / per_pair, w = e - pr * per_pair;
      const int il = w / s, cl = w - il * s;
      const int o = pr * 2 * s;
      const int i = o + s + il, c = o + cl;
      double acc = a[i * PS + c] * dinv[c];           // k = c
      for (int k = c + 1; k < o + s; ++k) acc += a[i * PS + k] * a[c * PS + k];
      a[c * PS + i] = acc;
    }
    __syncthreads();
    // step B: Z = -X22 * W, held in registers until every W has been consumed
    double z[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      z[q] = 0.0;
      const int e = tid + q * 1024;
      if (e < total) {
        const int pr = e / per_pair, w = e - pr * per_pair;
        const int il = w / s, cl = w - il * s;
        const int o = pr * 2 * s;
        const int i = o + s + il, c = o + cl;
        double acc = dinv[i] * a[c * PS + i];         // k = i
        for (int k = o + s; k < i; ++k) acc += a[k * PS + i] * a[c * PS + k];
        z[q] = -acc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + q * 1024;
      if (e < total) {
        const int pr = e / per_pair, w = e - pr * per_pair;
        const int il = w / s, cl = w - il * s;
        const int o = pr * 2 * s;
        a[(o + cl) * PS + (o + s + il)] = z[q];
      }
    }
    __syncthreads();
  }

  // inverse to HBM, full block with zero upper triangle
  for (int idx = tid; idx < PB * PB; idx += 1024) {
    const int i = idx >> 7, c = idx & 127;
    double v = 0.0;
    if (c < i) v = a[c * PS + i];
    else if (c == i) v = dinv[i];
    Linv[idx] = v;
    if (LinvT) LinvT[c * PB + i] = v;
  }
}

int gps_launch_potrf_base(gps_handle_t h, double* A, i64 lda, double* Linv_blk,
                          double* LinvT_blk, int* d_info, i64 row0, int factor) {
  static bool attr_set = false;
  const size_t lds = (size_t)(PB * PS + PB) * sizeof(double);
  if (!attr_set) {
    GPS_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_base_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  // potrf n^3/3 + trtri n^3/3
  LaunchScope ls(h, KC_POTRF_BASE, 2.0 * PB * PB * PB / 3.0, 3.0 * PB * PB * 8.0);
  hipLaunchKernelGGL(potrf_base_kernel, dim3(1), dim3(1024), lds, h->stream, A, lda, Linv_blk,
                     LinvT_blk, d_info, (int)row0, factor);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
