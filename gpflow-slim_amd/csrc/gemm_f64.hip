// fp64 MFMA GEMM for gfx950:  C (op)= A[M,K] * B[N,K]^T, row-major.
//
// This is the dense-contraction workhorse behind every O(N^3) step of the path:
//   - the trailing update  A22 -= L21 L21^T       of tf.cholesky   (models/gpr.py:70)
//   - the panel solves     X L^T = B               of tf.matrix_triangular_solve
//                                                  (models/gpr.py:122, conditionals.py:87)
//   - K(X*) - A^T A                                (models/gpr.py:126)
// Both operands are read "row, k" with k contiguous, so one kernel serves all of
// them (see DESIGN.md "one GEMM form").
//
// Tiling (MI355X first): 128x128 output tile per 256-thread workgroup (4 waves in
// a 2x2 grid, 64x64 per wave = 4x4 v_mfma_f64_16x16x4_f64 accumulators), BK = 16
// staged through LDS with a register prefetch of the next K-slab.  Row stride in
// LDS is 18 doubles: the 16 rows x 2 k of one ds_read_b64 half-wave then cover all
// 64 banks exactly once.  2 workgroups per CU (147 KB LDS, <=256 VGPR) so that one
// workgroup's C read-modify-write epilogue hides under the other's MFMA stream.
// blockIdx -> tile mapping is XCD-aware: each XCD walks a contiguous range of
// 8-tile-wide column strips, so the 64 tiles resident on one XCD share 16 operand
// panels in its private L2.
#include "gps_common.hpp"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define BM 128
#define BN 128
#define BK 16
#define LS 18          // LDS row stride in doubles (BK + 2)
#define STRIP 8        // tile columns per strip

struct GemmArgs {
  const double* A; const double* B; double* C;
  i64 lda, ldb, ldc;
  int Tm, Tn;       // tiles
  int K;
  int ntiles;
};

// bijective XCD remap: blocks b, b+8, b+16 ... (same XCD) get consecutive ids
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int x = b & 7, idx = b >> 3;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + idx;
}

template <bool LOWER>
__device__ __forceinline__ void decode_tile(int id, int Tm, int Tn, int& tm, int& tn) {
  if (!LOWER) {
    // strips of STRIP tile-columns, rows inner-major: id = strip*(Tm*w) + tm*w + c
    const int full = Tn / STRIP;                 // number of full strips
    const int per = Tm * STRIP;
    int s = id / per;
    if (s >= full) {                             // ragged last strip
      const int w = Tn - full * STRIP;
      const int rem = id - full * per;
      tm = rem / w; tn = full * STRIP + rem % w;
    } else {
      const int rem = id - s * per;
      tm = rem / STRIP; tn = s * STRIP + rem % STRIP;
    }
  } else {
    // lower triangle of a T x T tile grid, strip s covers columns [8s, 8s+w),
    // rows 8s .. T-1; inside the diagonal super-block only tn <= tm.
    const int T = Tm;
    int s = 0, rem = id;
    for (;;) {
      const int c0 = s * STRIP;
      const int w = min(STRIP, T - c0);
      const int tri = w * (w + 1) / 2;           // diagonal super-block
      const int cnt = tri + (T - c0 - w) * w;
      if (rem < cnt) {
        if (rem < tri) {
          // row rr (0..w-1) of the triangle has rr+1 tiles
          int rr = 0;
          while (rem >= rr + 1) { rem -= rr + 1; ++rr; }
          tm = c0 + rr; tn = c0 + rem;
        } else {
          rem -= tri;
          tm = c0 + w + rem / w; tn = c0 + rem % w;
        }
        return;
      }
      rem -= cnt; ++s;
    }
  }
}

template <bool LOWER, int OP>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* As = reinterpret_cast<double*>(smem_raw);       // [2][BM][LS]
  double* Bs = As + 2 * BM * LS;                          // [2][BN][LS]

  int tm, tn;
  decode_tile<LOWER>(xcd_remap((int)blockIdx.x, g.ntiles), g.Tm, g.Tn, tm, tn);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;

  // global -> register staging map: 8 threads cover one 16-double (128 B) row slab
  const int lrow = tid >> 3;          // 0..31
  const int lk = (tid & 7) * 2;       // 0,2,..,14
  const double* Ag = g.A + (i64)(tm * BM + lrow) * g.lda + lk;
  const double* Bg = g.B + (i64)(tn * BN + lrow) * g.ldb + lk;

  v2d ra[4], rb[4];
  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  const int nk = g.K / BK;

  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const v2d*>(Ag + (i64)(i * 32) * g.lda + (i64)kt * BK);
      rb[i] = *reinterpret_cast<const v2d*>(Bg + (i64)(i * 32) * g.ldb + (i64)kt * BK);
    }
  };
  auto lstore = [&](int buf) {
    double* a = As + buf * BM * LS + lrow * LS + lk;
    double* b = Bs + buf * BN * LS + lrow * LS + lk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<v2d*>(a + i * 32 * LS) = ra[i];
      *reinterpret_cast<v2d*>(b + i * 32 * LS) = rb[i];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);

    const double* a_base = As + buf * BM * LS + (wr * 64 + fr) * LS + fk;
    const double* b_base = Bs + buf * BN * LS + (wc * 64 + fr) * LS + fk;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = a_base[i * 16 * LS + kk * 4];
        b[i] = b_base[i * 16 * LS + kk * 4];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }

    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue.  f64 accumulator map (differs from every other dtype on gfx950):
  //   col = lane & 15, row = (lane >> 4) + 4 * reg.
  const i64 row0 = (i64)tm * BM + wr * 64 + (lane >> 4);
  const i64 col0 = (i64)tn * BN + wc * 64 + (lane & 15);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        double* cp = g.C + (row0 + i * 16 + 4 * rg) * g.ldc + col0 + j * 16;
        if (OP == 0) *cp = *cp - acc[i][j][rg];
        else *cp = acc[i][j][rg];
      }
    }
  }
}

static const size_t kGemmLds = (size_t)(2 * BM * LS + 2 * BN * LS) * sizeof(double);

template <bool LOWER, int OP>
static int launch_variant(gps_handle_t h, const GemmArgs& g) {
  static bool attr_set = false;
  if (!attr_set) {
    GPS_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f64_kernel<LOWER, OP>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_f64_kernel<LOWER, OP>), dim3(g.ntiles), dim3(256), kGemmLds, h->stream, g);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_gemm_nt(gps_handle_t h, int op, int lower, i64 M, i64 N, i64 K,
                       const double* A, i64 lda, const double* B, i64 ldb,
                       double* C, i64 ldc) {
  if (M <= 0 || N <= 0 || K <= 0) return GPS_OK;
  if (M % BM || N % BN || K % BK || (lower && M != N))
    return gps_fail(h, GPS_ERR_ARG, "gemm_nt: M,N must be multiples of 128 and K of 16");
  if ((lda & 1) || (ldb & 1) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    return gps_fail(h, GPS_ERR_ARG, "gemm_nt: operands must be 16-byte aligned with even leading dimension");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.Tm = (int)(M / BM); g.Tn = (int)(N / BN); g.K = (int)K;
  const i64 nt = lower ? (i64)g.Tm * (g.Tm + 1) / 2 : (i64)g.Tm * g.Tn;
  if (nt > 0x7fffffff) return gps_fail(h, GPS_ERR_ARG, "gemm_nt: too many tiles");
  g.ntiles = (int)nt;
  const double flops = 2.0 * (double)nt * BM * BN * (double)K;
  const double bytes = (double)nt * ((op == 0 ? 2.0 : 1.0) * BM * BN * 8.0) +
                       8.0 * (double)K * (double)(M + N);   // compulsory traffic: C rmw + each panel once
  LaunchScope ls(h, KC_GEMM, flops, bytes);
  if (lower) return op == 0 ? launch_variant<true, 0>(h, g) : launch_variant<true, 1>(h, g);
  return op == 0 ? launch_variant<false, 0>(h, g) : launch_variant<false, 1>(h, g);
}
