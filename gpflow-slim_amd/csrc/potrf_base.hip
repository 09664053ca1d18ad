// Base case of the blocked Cholesky: factor one 128x128 diagonal block inside a
// single workgroup and invert the factor.
//
// Replaces the innermost part of tf.cholesky (models/gpr.py:70,121;
// conditionals.py:84).  The explicit 128x128 inverse turns every panel solve
// X L11^T = B into an MFMA GEMM (gemm_f64.hip) and every vector solve into a
// 128x128 gemv, which is what makes the recursive trsm / trsv GEMM-only.
//
// Factorisation: LDS image [128][129], blocked right-looking with panel width 16: the 16x16
// diagonal block is factored by ONE wave in registers (v_readlane broadcasts, hardware rsqrt + two
// Newton steps, no barrier), the rows below by one thread each (forward substitution against that
// block), and the trailing update C_IJ -= P_I P_J^T runs on the fp64 MFMA over 16x16 tiles --
// 3 barriers per panel (24 in all) instead of one per column.
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

// wave-uniform broadcast of lane `lane`'s fp64 value (lane is a compile-time constant at every call)
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(p): hardware estimate + two Newton steps (full double accuracy; keeps the 16 serial pivots of a
// diagonal block off the IEEE sqrt/divide sequences)
__device__ __forceinline__ double rsqrt_nr(double p) {
  double y = __builtin_amdgcn_rsq(p);
  y = y * (1.5 - 0.5 * p * y * y);
  y = y * (1.5 - 0.5 * p * y * y);
  return y;
}

// whole 128x128 block -> LDS image; 16-byte loads, all 32 of a thread in flight at once (one memory round trip:
// a lone workgroup draws ~40 GB/s, so the tile load is latency, not bandwidth); the upper triangle is loaded too:
// it is never read before being overwritten
__device__ __forceinline__ void load_image(double* a, const double* __restrict__ A, i64 lda, int tid) {
  double2 v[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int idx = u * NT + tid;
    const int i = idx >> 6, j = (idx & 63) * 2;
    v[u] = *reinterpret_cast<const double2*>(A + (i64)i * lda + j);
  }
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int idx = u * NT + tid;
    const int i = idx >> 6, j = (idx & 63) * 2;
    a[i * PS + j] = v[u].x;
    a[i * PS + j + 1] = v[u].y;
  }
}

// Blocked right-looking factorisation of the 128x128 LDS image, panel width 16:
//   phase 1 (wave 0, registers + v_readlane, no barrier): Cholesky of the 16x16 diagonal block
//   phase 2 (one thread per row below): forward substitution against that block
//   phase 3 (all waves, fp64 MFMA): trailing update C_IJ -= P_I P_J^T on 16x16 tiles
__device__ __forceinline__ void factor_image(double* a, double* dinv, int* info, int row0, int tid, long long* stamps) {
  long long t_p1 = 0, t_p2 = 0, t_p3 = 0, t0 = 0;
#define PH_BEGIN() do { if (stamps && tid == 0) t0 = (long long)wall_clock64(); } while (0)
#define PH_END(acc) do { if (stamps && tid == 0) acc += (long long)wall_clock64() - t0; } while (0)
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  for (int g = 0; g < 8; ++g) {
    const int o = 16 * g;
    // ---- phase 1
    PH_BEGIN();
    if (wave == 0) {
      const int i = lane & 15;
      double r[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) r[c] = a[(o + i) * PS + o + c];
      // branch-free body (one basic block): the scheduler can overlap column j's trailing updates with
      // the rsqrt / Newton chain of column j+1
      int bad = 0x7fffffff;
      double my_y = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double p = readlane_f64(r[j], j);
        bad = (!(p > 0.0) && bad == 0x7fffffff) ? j : bad;
        const double y = rsqrt_nr(p);
        r[j] = (i == j) ? p * y : r[j] * y;
        my_y = (i == j) ? y : my_y;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) {
          const double lkj = readlane_f64(r[j], k);
          r[k] -= r[j] * lkj;
        }
      }
      if (lane == 0 && bad != 0x7fffffff) atomicMin(info, row0 + o + bad + 1);
      if (lane < 16) {
        dinv[o + i] = my_y;
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c <= i) a[(o + i) * PS + o + c] = r[c];
      }
    }
    __syncthreads();
    PH_END(t_p1);
    if (g == 7) break;
    // ---- phase 2
    PH_BEGIN();
    const int nrows = PB - o - 16;
    if (tid < nrows) {
      const int i = o + 16 + tid;
      double x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = a[i * PS + o + c];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        double sacc = x[j];
#pragma unroll
        for (int k = 0; k < j; ++k) sacc -= x[k] * a[(o + j) * PS + o + k];
        x[j] = sacc * dinv[o + j];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) a[i * PS + o + c] = x[c];
    }
    __syncthreads();
    PH_END(t_p2);
    // ---- phase 3
    PH_BEGIN();
    const int nt = 7 - g;
    const int ntile = nt * (nt + 1) / 2;
    for (int t = wave; t < ntile; t += 4) {
      int Ip = 0, rem = t;
      while (rem >= Ip + 1) { rem -= Ip + 1; ++Ip; }
      const int I = g + 1 + Ip, J = g + 1 + rem;
      v4d acc;
      double av[4], bv[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        av[s4] = -a[(16 * I + fr) * PS + o + 4 * s4 + fk];
        bv[s4] = a[(16 * J + fr) * PS + o + 4 * s4 + fk];
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) acc[rg] = a[(16 * I + fk + 4 * rg) * PS + 16 * J + fr];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) a[(16 * I + fk + 4 * rg) * PS + 16 * J + fr] = acc[rg];
    }
    __syncthreads();
    PH_END(t_p3);
  }
  if (stamps && tid == 0) { stamps[16] = t_p1; stamps[17] = t_p2; stamps[18] = t_p3; }
#undef PH_BEGIN
#undef PH_END
}

// ---- MFMA helpers for the inverse levels --------------------------------------------------------
// X (= inv L, lower) is stored transposed in the strict upper triangle of the image, diag in dinv.
__device__ __forceinline__ double x_elem(const double* a, const double* dinv, int r, int c) {
  // two UNCONDITIONAL LDS reads + selects.  The empty asm pins both loaded values: without it the compiler
  // sinks the reads back under divergent branches and every MFMA of the inverse levels waits for its own
  // pair of serialised LDS round trips.
  double off = a[(r > c) ? (c * PS + r) : 0];
  double dia = dinv[r & (PB - 1)];
  asm volatile("" : "+v"(off), "+v"(dia));
  return (r > c) ? off : ((r == c) ? dia : 0.0);
}

// one doubling level of the block inverse (S = size of the already inverted diagonal blocks).
// Every wave walks its TPW tiles together (KSPL partial sums per tile) so that the LDS operand reads of
// several products are in flight at once; zero operands outside the triangular ranges (x_elem) make
// every product unconditional.
template <int S>
__device__ __forceinline__ void level_step(double* a, const double* dinv, int wave, int fr, int fk) {
  constexpr int KSTEPS = S / 4;
  constexpr int tps = S / 16;
  constexpr int tiles_pair = tps * tps;
  constexpr int ntile = (PB / (2 * S)) * tiles_pair;     // 4, 8, 16
  constexpr int TPW = ntile / 4;                         // tiles per wave: 1, 2, 4
  constexpr int KSPL = 4 / TPW;                          // partial sums per tile: 4, 2, 1
  int i0[TPW], c0[TPW], ob[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int t = wave + 4 * q;
    const int pr = t / tiles_pair, w = t - pr * tiles_pair;
    ob[q] = pr * 2 * S;
    i0[q] = ob[q] + S + (w / tps) * 16;
    c0[q] = ob[q] + (w % tps) * 16;
  }
  // step A: W = L21 * X11 ; W[i][c] -> a[c][i]
  {
    v4d acc[TPW][KSPL];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
      for (int u = 0; u < KSPL; ++u) acc[q][u] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int k = ob[q] + 4 * ks + fk;                       // k runs over the TL block
        const double av = a[(i0[q] + fr) * PS + k];              // L21[i][k]
        const double bv = x_elem(a, dinv, k, c0[q] + fr);        // X11[k][c] (0 for k < c)
        acc[q][ks % KSPL] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[q][ks % KSPL], 0, 0, 0);
      }
    }
    __syncthreads();          // every L21 / X11 operand has been read before W overwrites the slot
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
#pragma unroll
      for (int u = 1; u < KSPL; ++u) acc[q][0] += acc[q][u];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) a[(c0[q] + fr) * PS + i0[q] + fk + 4 * rg] = acc[q][0][rg];
    }
  }
  __syncthreads();
  // step B: Z = -X22 * W, kept in registers until every W has been consumed
  {
    v4d z[TPW][KSPL];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
      for (int u = 0; u < KSPL; ++u) z[q][u] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int k = ob[q] + S + 4 * ks + fk;                   // k runs over the BR block
        const double av = x_elem(a, dinv, i0[q] + fr, k);        // X22[i][k] (0 for k > i)
        const double bv = a[(c0[q] + fr) * PS + k];              // W[k][c]
        z[q][ks % KSPL] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, z[q][ks % KSPL], 0, 0, 0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
#pragma unroll
      for (int u = 1; u < KSPL; ++u) z[q][0] += z[q][u];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) a[(c0[q] + fr) * PS + i0[q] + fk + 4 * rg] = -z[q][0][rg];
    }
  }
  __syncthreads();
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

  const int tid = threadIdx.x;

  if (factor) {
    load_image(a, A, lda, tid);
    __syncthreads();
    STAMP(1);
    factor_image(a, dinv, info, row0, tid, stamps);
    STAMP(2);
    // L back to HBM (upper triangle of the diagonal block zero-filled, like tf.cholesky)
    for (int idx = tid; idx < PB * PB / 2; idx += NT) {
      const int i = idx >> 6, j = (idx & 63) * 2;
      double2 v;
      v.x = (j <= i) ? a[i * PS + j] : 0.0;
      v.y = (j + 1 <= i) ? a[i * PS + j + 1] : 0.0;
      *reinterpret_cast<double2*>(A + (i64)i * lda + j) = v;
    }
  } else {
    load_image(a, A, lda, tid);
    __syncthreads();
    if (tid < PB) dinv[tid] = 1.0 / a[tid * PS + tid];
    __syncthreads();
  }

  STAMP(3);
  // ---- inverse, level 0: the eight 16x16 diagonal blocks, one column per thread.
  // X[i][c] is stored at a[c][i] (i > c); X[c][c] = dinv[c].
  if (tid < PB) {
    // the column stays in registers: reading back the X entries this thread has just written would make every
    // step wait for an LDS store -> load round trip (15 dependent ones for the first column of a block: 4.4 us)
    const int c = tid, rem = 15 - (c & 15);    // rows below c inside its 16-block
    double x[16];
    x[0] = dinv[c];
#pragma unroll
    for (int d = 1; d < 16; ++d) {
      if (d <= rem) {
        const int i = c + d;
        double s = a[i * PS + c] * x[0];
#pragma unroll
        for (int t = 1; t < d; ++t) s += a[i * PS + c + t] * x[t];
        x[d] = -s * dinv[i];
      }
    }
#pragma unroll
    for (int d = 1; d < 16; ++d)
      if (d <= rem) a[c * PS + c + d] = x[d];
  }
  __syncthreads();

  STAMP(4);
  // ---- levels s = 16, 32, 64: X21 = -X22 * (L21 * X11) for each pair of s-blocks, on the
  // fp64 MFMA (16x16x4).  Output tiles are dealt round-robin to the 4 waves; all operands of a tile are
  // fetched from LDS first (KS k-steps, zero outside the triangular range), then the MFMA chain runs.
  {
    const int lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    level_step<16>(a, dinv, wave, fr, fk);
    level_step<32>(a, dinv, wave, fr, fk);
    level_step<64>(a, dinv, wave, fr, fk);
  }

  STAMP(5);
  // inverse (and its transpose) to HBM, full blocks with explicit zeros, 16-byte stores
  for (int idx = tid; idx < PB * PB / 2; idx += NT) {
    const int i = idx >> 6, c = (idx & 63) * 2;
    double2 v;
    v.x = x_elem(a, dinv, i, c);
    v.y = x_elem(a, dinv, i, c + 1);
    *reinterpret_cast<double2*>(Linv + i * PB + c) = v;
  }
  if (LinvT) {
    for (int idx = tid; idx < PB * PB / 2; idx += NT) {
      const int c = idx >> 6, i = (idx & 63) * 2;        // LinvT[c][i] = X[i][c]
      double2 v;
      v.x = x_elem(a, dinv, i, c);
      v.y = x_elem(a, dinv, i + 1, c);
      *reinterpret_cast<double2*>(LinvT + c * PB + i) = v;
    }
  }
  __syncthreads();
  STAMP(6);
#undef STAMP
}

int gps_launch_potrf_base(gps_handle_t h, double* A, i64 lda, double* Linv_blk,
                          double* LinvT_blk, int* d_info, i64 row0, int factor, long long* d_stamps) {
  const size_t lds = (size_t)(PB * PS + PB) * sizeof(double);
  int rc0 = gps_dyn_lds(h, reinterpret_cast<const void*>(&potrf_base_kernel), (int)lds);
  if (rc0) return rc0;
  // potrf n^3/3 + trtri n^3/3
  LaunchScope ls(h, KC_POTRF_BASE, 2.0 * PB * PB * PB / 3.0, 3.0 * PB * PB * 8.0);
  hipLaunchKernelGGL(potrf_base_kernel, dim3(1), dim3(NT), lds, h->stream, A, lda, Linv_blk,
                     LinvT_blk, d_info, (int)row0, factor, d_stamps);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
