// Backward-stable 128-column leaves of the triangular solves.
//
// The recursion of blocked.hpp turns every 128-wide solve  X L11^T = B  into a product with the explicit inverse
// W = inv(L11) (gemm_f64.hip).  That is one GEMM, but not backward stable: the residual  B - X L11^T  is
// O(eps cond(L11)) |B| instead of O(eps) |X| |L11|, and on the matrices the reference solves with its 1e-6 jitter
// (conditionals.py:60, features.py:76: cond(Kuu) ~ 1e8) the result is about one digit behind
// tf.matrix_triangular_solve (conditionals.py:87,100; densities.py:82), which substitutes.
//
// The leaves here use W as a PRECONDITIONER instead: one step of iterative refinement in working precision,
//
//     X0 = B W^T ;   R = B - X0 L11^T ;   X = X0 + R W^T ,
//
// after which the residual is O(eps) |X| |L11| + O(eps^2 cond^2) -- the componentwise backward error of substitution
// (numpy experiment: tests/test_leaf_refine_model.py; exact-arithmetic fixture: tests/golden/exact/).  All three
// products stay on the fp64 MFMA and in one launch: a workgroup owns BM complete rows of B, keeps the current A
// operand (B, then X0, then R) and the current triangular operand (W, L11, W) whole in LDS, and holds X0 in
// accumulators from the first product to the last.  L11 is read straight from the factor;
// entries on the far side of its diagonal are masked while staging, so whatever the caller's matrix holds there
// (tf.matrix_triangular_solve ignores it as well) never reaches the arithmetic.
//
// `upper` selects the right-sided form  X L11 = B  of the unwhitened conditional (conditionals.py:100), which works
// on U = L^T: W = inv(L11)^T is then read as stored (W[j][k] = inv(L11)[k][j]) and the diagonal block of U is
// upper triangular (kept: k >= j).
#include "gps_common.hpp"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define LF_N 128                 // columns of a leaf
#define LF_LS (LF_N + 2)         // row stride of the A-operand tile: 16 rows x 2 k of a half-wave cover all 64 banks

// One workgroup = BM complete rows.  LDS: Pa [BM][130] (the A operand: B, then X0, then R, whole K) and Wp, the
// current triangular operand with only the non-zero part of every row: block row q (16 rows) keeps 16 (q + 1) columns
// (+2 padding), 74 KB in all -- the dense [128][130] image would not fit beside Pa.  Each product is one barrier-free
// run of k-steps; the next operand is prefetched into registers meanwhile (one L2 round trip per product instead of
// one per K slab: a slab-staged first version spent 24 dependent round trips per workgroup, 26 us at 16 rows).
// All three operands are triangular the same way (lower: W[j][k] = D[j][k] = 0 for k > j), so output column block c
// (16 wide) only needs the k blocks <= c (>= c for `upper`); waves own the mirrored column blocks {q, 7 - q}, which
// gives every wave 9/16 of the dense work.  The product is compiled once per (wave position, upper) with every
// block test folded (a first version with run-time block loops moved its accumulators through a drained MFMA pipe
// on every iteration: 326 cycles per MFMA) and called from a three-pass loop, so the code stays small.
__host__ __device__ constexpr int lf_q(int jb, bool upper) { return upper ? 7 - jb : jb; }                 // storage order of block row jb
__host__ __device__ constexpr int lf_len(int jb, bool upper) { return 16 * (lf_q(jb, upper) + 1); }        // columns kept
__host__ __device__ constexpr int lf_base(int jb, bool upper) { return 128 * lf_q(jb, upper) * (lf_q(jb, upper) + 1) + 32 * lf_q(jb, upper); }
__host__ __device__ constexpr int lf_c0(int jb, bool upper) { return upper ? 16 * jb : 0; }                // first column kept
#define LF_WP_DOUBLES (128 * 8 * 9 + 32 * 8)

#define LF_ACTIVE(C, kb) (UPPER ? ((kb) >= (C)) : ((kb) <= (C)))
#define LF_LOADB(J, C)                                                                                          \
  if (LF_ACTIVE(C, hb >> 1)) {                                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                            \
      fb[buf][kk][J] = Wp[lf_base(C, UPPER) + fr * (lf_len(C, UPPER) + 2) + hb * 8 + kk * 4 + fk - lf_c0(C, UPPER)]; \
  }
#define LF_MMA(J, C)                                                                                            \
  if (LF_ACTIVE(C, hb >> 1)) {                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                              \
      acc[i][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[buf][kk][i], fb[buf][kk][J], acc[i][J], 0, 0, 0);    \
  }

// acc[:, J] += Pa[rows of this wave][k] * Wp[column block C_J][k] over the k blocks that are not structurally zero;
// column blocks of the wave: {Q, 7 - Q}.  Software pipeline over half blocks (8 k = 2 MFMA k-steps): the fragments of
// half block hb + 1 are requested before the MFMAs of hb issue (two register sets; sched_barrier keeps that order --
// left alone, the scheduler hoists every LDS read of the unrolled product to the top and needs > 512 registers).
template <int MI, int NI, int Q, bool UPPER>
__device__ __forceinline__ void leaf_product(v4d (&acc)[MI][NI], const double* __restrict__ a_base,
                                             const double* __restrict__ Wp, int fr, int fk) {
  constexpr int HB0 = UPPER ? 2 * Q : 0, HB1 = UPPER ? 16 : 16 - 2 * Q;      // half blocks with an active column block
  double fa[2][2][MI], fb[2][2][NI];
  auto load = [&](const int buf, const int hb) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[buf][kk][i] = a_base[i * 16 * LF_LS + hb * 8 + kk * 4];
    LF_LOADB(0, Q)
    LF_LOADB(1, 7 - Q)
  };
  load(0, HB0);
#pragma unroll
  for (int hb = HB0; hb < HB1; ++hb) {
    const int buf = (hb - HB0) & 1;
    if (hb + 1 < HB1) load(buf ^ 1, hb + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      LF_MMA(0, Q)
      LF_MMA(1, 7 - Q)
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// NT threads: 4 waves side by side own the mirrored column-block pairs; 64 rows use 8 waves (two wave rows), which
// keeps every wave at <= 256 registers and gives each SIMD a second wave to cover LDS latency and the stage phases.
// Barrier for data that is exchanged through LDS only: __syncthreads() also waits for every outstanding GLOBAL access
// of the wave (vmcnt(0)) -- here that would be the operand prefetch that is meant to stay in flight behind the product,
// and the parked X0 store.
__device__ __forceinline__ void lf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BM, int NT, bool UPPER, bool WALK>
__global__ __launch_bounds__(NT) void trsm_leaf_refine_kernel(double* Bm, i64 ldb, i64 m,
                                                              const double* __restrict__ W,
                                                              const double* __restrict__ D, i64 ldd,
                                                              long long* __restrict__ stamps) {
  constexpr int NWV = NT / 64, WGN = 4, WGM = NWV / WGN;
  constexpr int WTM = BM / WGM, WTN = LF_N / WGN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int NPRE = 128 / NWV;                            // operand rows (16-byte loads) per thread
  static_assert(NI == 2 && MI >= 1, "column blocks are owned in mirrored pairs");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* Pa = reinterpret_cast<double*>(smem_raw);          // [BM][LF_LS]
  double* Wp = Pa + BM * LF_LS;                              // packed triangular operand

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WGN, wc = wave % WGN;
  const int fr = lane & 15, fk = lane >> 4;
  // Persistent over row tiles: workgroup b takes tiles b, b + gridDim.x, ...  From the second tile on the operand W is
  // still staged from the previous tile's last product and the tile's rows were fetched behind that product: a tile
  // costs two operand stagings instead of three, no exposed load of its rows and no workgroup launch.
  const i64 ntiles = m / BM;
  if ((i64)blockIdx.x >= ntiles) return;
  double* Bg = Bm + (i64)blockIdx.x * BM * ldb;
  // diagnostics (gps_diag_trsm_leaf): phase stamps of workgroup 0 in 100 MHz ticks
#define LF_STAMP(q) do { if (stamps && blockIdx.x == 0 && tid == 0) stamps[q] = (long long)wall_clock64(); } while (0)
  LF_STAMP(0);

  // column blocks of this wave
  const int cb[NI] = {wc, 7 - wc};

  // triangular 128x128 operand: up to 32 16-byte loads per thread (row j = 4u + wave per wave-load; lanes outside the
  // kept part of the row do not load).  The values are only touched again in stage(), so the loads stay in flight
  // behind the product that runs meanwhile.
  v2d pre[NPRE];
  int lane_v = lane, tid_v = tid;                            // re-laundered per tile (see the tile loop)
  auto fetch = [&](const double* __restrict__ S, i64 lds_) {
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int j = NWV * u + wave, jb = (NWV * u) >> 4;
      const bool keep = UPPER ? (2 * lane_v >= 16 * jb) : (2 * lane_v < 16 * (jb + 1));
      pre[u] = keep ? *reinterpret_cast<const v2d*>(S + (i64)j * lds_ + 2 * lane_v) : (v2d){0.0, 0.0};
    }
  };
  // registers -> packed rows; entries beyond the diagonal are zeroed here (whatever the caller's matrix holds on
  // the far side of the diagonal of D never reaches the arithmetic)
  auto stage = [&]() {
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int j = NWV * u + wave, jb = (NWV * u) >> 4, k = 2 * lane_v;
      v2d v = pre[u];
      if (UPPER) { if (k < j) v.x = 0.0; if (k + 1 < j) v.y = 0.0; }
      else { if (k > j) v.x = 0.0; if (k + 1 > j) v.y = 0.0; }
      const bool keep = UPPER ? (k >= 16 * jb) : (k < 16 * (jb + 1));
      if (keep) *reinterpret_cast<v2d*>(Wp + lf_base(jb, UPPER) + (j & 15) * (lf_len(jb, UPPER) + 2) + k - lf_c0(jb, UPPER)) = v;
    }
  };
  constexpr int NLB = BM * LF_N / 2 / NT;                    // 16-byte loads per thread of the B tile
  v2d rows[NLB];                                             // a tile's rows on their way from HBM to Pa
  auto fetch_rows = [&](const double* __restrict__ G) {
#pragma unroll
    for (int u = 0; u < NLB; ++u) {
      const int idx = u * NT + tid_v;
      rows[u] = *reinterpret_cast<const v2d*>(G + (i64)(idx >> 6) * ldb + (idx & 63) * 2);
    }
  };
  auto stage_rows = [&]() {
#pragma unroll
    for (int u = 0; u < NLB; ++u) {
      const int idx = u * NT + tid_v;
      *reinterpret_cast<v2d*>(Pa + (idx >> 6) * LF_LS + (idx & 63) * 2) = rows[u];
    }
  };
  fetch_rows(Bg);
  stage_rows();

  // accumulator map of v_mfma_f64_16x16x4: col = lane & 15, row = (lane >> 4) + 4 reg
  int crow = wr * WTM + (lane >> 4), ccol = lane & 15;       // (laundered per tile with the other thread indices)
  double bc[MI][NI][4];                                      // B in accumulator layout
  const double* a_base = Pa + (wr * WTM + fr) * LF_LS + fk;
  v4d acc[MI][NI], x0[MI][NI];
  // pass 0: X0 = B W^T (D on its way) ; pass 1: R = B - X0 D^T (W on its way again: L2-hot) ; pass 2: X = X0 + R W^T
  // (pass -1 only stages W)
#pragma unroll 1
  for (i64 tile = blockIdx.x; tile < ntiles; tile += (WALK ? (i64)gridDim.x : ntiles)) {       // WALK == false: one tile per workgroup
  const bool first = (tile == (i64)blockIdx.x);
  Bg = Bm + tile * BM * ldb;
  // (pointers and thread indices pass an opaque barrier per tile: hoisted out of the tile loop, the per-thread fetch /
  // stage / store addresses cost ~100 registers for the whole loop, and the 64-row variant spills)
  const double* Wt = W; const double* Dt = D;
  asm volatile("" : "+s"(Wt), "+s"(Dt));
  asm volatile("" : "+v"(lane_v), "+v"(tid_v), "+v"(crow), "+v"(ccol));
  if (!first) {
    // this tile's rows were fetched behind the previous tile's last product; W is still staged in Wp from it
    lf_lds_barrier();                                        // every wave is done with Pa
    stage_rows();
    lf_lds_barrier();
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) bc[i][j][rg] = Pa[(crow + i * 16 + 4 * rg) * LF_LS + cb[j] * 16 + ccol];
  }
#pragma unroll 1
  for (int pass = first ? -1 : 0; pass < 2; ++pass) {
    fetch((pass == 0) ? Dt : Wt, (pass == 0) ? ldd : (i64)LF_N);
    if (pass >= 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
      switch (wc) {
        case 0: leaf_product<MI, NI, 0, UPPER>(acc, a_base, Wp, fr, fk); break;
        case 1: leaf_product<MI, NI, 1, UPPER>(acc, a_base, Wp, fr, fk); break;
        case 2: leaf_product<MI, NI, 2, UPPER>(acc, a_base, Wp, fr, fk); break;
        default: leaf_product<MI, NI, 3, UPPER>(acc, a_base, Wp, fr, fk); break;
      }
      LF_STAMP(2 + 2 * pass);
      lf_lds_barrier();               // every wave is done with Pa and Wp
      // X0 is parked in the output rows (B itself lives on in `bc`) and comes back as the accumulator of pass 2 (its
      // own lanes wrote it: no fence needed).  Keeping it in registers instead costs the 64-row tile spills.
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (pass == 0) {
            x0[i][j] = acc[i][j];
          } else {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[i][j][rg] = bc[i][j][rg] - acc[i][j][rg];
          }
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) Pa[(crow + i * 16 + 4 * rg) * LF_LS + cb[j] * 16 + ccol] = acc[i][j][rg];
        }
      if (pass == 1) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[i][j][rg] = x0[i][j][rg];
      }
    }
    stage();
    lf_lds_barrier();
    LF_STAMP(3 + 2 * pass);
    if (pass < 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) bc[i][j][rg] = Pa[(crow + i * 16 + 4 * rg) * LF_LS + cb[j] * 16 + ccol];
    }
  }
  // pass 2 (outside the loop: the rows fetched here stay in registers until the next tile stages them, and inside the
  // loop the allocator would keep those registers reserved through every pass): X = X0 + R W^T
  // (unconditional, so that the registers are dead from the staging at the top of a tile to here: the last tile of a
  // workgroup fetches its own rows once more and drops them)
  if (WALK) fetch_rows(Bm + ((tile + (i64)gridDim.x < ntiles) ? tile + (i64)gridDim.x : tile) * BM * ldb);     // behind the last product
  switch (wc) {
    case 0: leaf_product<MI, NI, 0, UPPER>(acc, a_base, Wp, fr, fk); break;
    case 1: leaf_product<MI, NI, 1, UPPER>(acc, a_base, Wp, fr, fk); break;
    case 2: leaf_product<MI, NI, 2, UPPER>(acc, a_base, Wp, fr, fk); break;
    default: leaf_product<MI, NI, 3, UPPER>(acc, a_base, Wp, fr, fk); break;
  }
  LF_STAMP(6);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) Bg[(i64)(crow + i * 16 + 4 * rg) * ldb + cb[j] * 16 + ccol] = acc[i][j][rg];
  LF_STAMP(7);
  }
}

template <int BM, int NT, bool UPPER>
static int launch_leaf_u(gps_handle_t h, double* B, i64 ldb, i64 m, const double* W, const double* D, i64 ldd) {
  long long* stamps = h->leaf_stamps;
  const size_t lds = (size_t)(BM * LF_LS + LF_WP_DOUBLES) * sizeof(double);
  // one workgroup per CU fits (140 KB of LDS at BM = 64): more tiles than CUs are walked by the resident workgroups
  const i64 ntiles = m / BM;
  const i64 cus = h->prop.multiProcessorCount > 0 ? h->prop.multiProcessorCount : 256;
  if (ntiles > cus) {
    int rc = gps_dyn_lds(h, reinterpret_cast<const void*>(&trsm_leaf_refine_kernel<BM, NT, UPPER, true>), (int)lds);
    if (rc) return rc;
    hipLaunchKernelGGL((trsm_leaf_refine_kernel<BM, NT, UPPER, true>), dim3((unsigned)cus), dim3(NT), lds, h->stream, B, ldb, m, W, D, ldd, stamps);
  } else {
    int rc = gps_dyn_lds(h, reinterpret_cast<const void*>(&trsm_leaf_refine_kernel<BM, NT, UPPER, false>), (int)lds);
    if (rc) return rc;
    hipLaunchKernelGGL((trsm_leaf_refine_kernel<BM, NT, UPPER, false>), dim3((unsigned)ntiles), dim3(NT), lds, h->stream, B, ldb, m, W, D, ldd, stamps);
  }
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
template <int BM, int NT>
static int launch_leaf(gps_handle_t h, double* B, i64 ldb, i64 m, const double* W, const double* D, i64 ldd, int upper) {
  return upper ? launch_leaf_u<BM, NT, true>(h, B, ldb, m, W, D, ldd) : launch_leaf_u<BM, NT, false>(h, B, ldb, m, W, D, ldd);
}

// B [m, 128] <- solution X of  X D^T = B (upper == 0, D lower triangular, W = inv(D))  or  X D = B given U = D^T
// (upper == 1: D points at the upper-triangular diagonal block of U, W = inv(L11)^T as stored), refined once.
int gps_launch_trsm_leaf_refine(gps_handle_t h, double* B, i64 ldb, i64 m, const double* W, const double* D, i64 ldd,
                                int upper) {
  if (m <= 0) return GPS_OK;
  if (m % 128 || (ldb & 1) || (ldd & 1) || ((uintptr_t)B & 15) || ((uintptr_t)W & 15) || ((uintptr_t)D & 15))
    return gps_fail(h, GPS_ERR_ARG, "trsm_leaf_refine: m must be a multiple of 128, operands 16-byte aligned with even leading dimension");
  // three triangular products: 3 x 9/16 of the dense 2 m 128^2
  LaunchScope ls(h, KC_GEMM, 3.0 * (9.0 / 16.0) * 2.0 * (double)m * 128.0 * 128.0, 16.0 * (double)m * 128.0 + 3.0 * 128.0 * 128.0 * 8.0);
  ls.tag[0] = m; ls.tag[1] = 128; ls.tag[2] = 128; ls.tag[3] = 1000 + upper;
  // latency-bound up to one workgroup per CU: the smallest tile that still covers the rows in one round of workgroups;
  // beyond, the 64-row tile has the densest MFMA stream (and its resident workgroups walk the tiles)
  const i64 cus = h->prop.multiProcessorCount > 0 ? h->prop.multiProcessorCount : 256;
  if (m / 16 <= cus) return launch_leaf<16, 256>(h, B, ldb, m, W, D, ldd, upper);
  if (m / 32 <= cus) return launch_leaf<32, 256>(h, B, ldb, m, W, D, ldd, upper);
  return launch_leaf<64, 512>(h, B, ldb, m, W, D, ldd, upper);
}

// ---- vector leaf:  y <- solution of  L11 a = y  (one workgroup per right-hand side), refined once
//   a0 = W y ;  r = y - L11 a0 ;  a = a0 + W r         W = inv(L11), read through its transpose for coalescing.
// upper == 1: the backward substitution leaf  L11^T a = y  of the gradient path (W^T products through W itself,
// L11^T read by columns = rows of L11 walked by the whole workgroup).
__global__ __launch_bounds__(256) void trsv_leaf_refine_kernel(const double* __restrict__ Wt, const double* __restrict__ D,
                                                               i64 ldd, double* __restrict__ y, i64 ldy, int upper) {
  __shared__ double ys[128], a0[128], rs[128];
  __shared__ double part[2][128];
  double* yr = y + (i64)blockIdx.x * ldy;
  const int tid = threadIdx.x, i = tid & 127, half = tid >> 7;
  if (tid < 128) ys[tid] = yr[tid];
  __syncthreads();
  // out[i] = sum_c Wt[c][i] v[c]   (two halves of c, fixed-order sum)
  auto apply_w = [&](const double* v) -> double {
    const double* Lc = Wt + (half * 64) * 128 + i;
    double s = 0.0;
#pragma unroll 16
    for (int c = 0; c < 64; ++c) s += Lc[c * 128] * v[half * 64 + c];
    part[half][i] = s;
    __syncthreads();
    const double o = part[0][i] + part[1][i];
    __syncthreads();
    return o;
  };
  const double v0 = apply_w(ys);
  if (!half) a0[i] = v0;
  __syncthreads();
  // r = y - T a0 with T = L11 (upper == 0: T[i][c] = D[i][c], c <= i) or L11^T (upper == 1: T[i][c] = D[c][i], c >= i)
  double s = 0.0;
  if (!upper) {
    // wave w takes rows w, w+4, ...: a row of the block is read by one wave with 16-byte loads
    const int lane = tid & 63, wave = tid >> 6;
    for (int row = wave; row < 128; row += 4) {
      const v2d l = *reinterpret_cast<const v2d*>(D + (i64)row * ldd + 2 * lane);
      double p = 0.0;
      if (2 * lane <= row) p += l.x * a0[2 * lane];
      if (2 * lane + 1 <= row) p += l.y * a0[2 * lane + 1];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
      if (lane == 0) rs[row] = ys[row] - p;
    }
    __syncthreads();
  } else {
    const double* Dc = D + (i64)(half * 64) * ldd + i;
    for (int c = 0; c < 64; ++c) {
      const int cc = half * 64 + c;
      if (cc >= i) s += Dc[(i64)c * ldd] * a0[cc];
    }
    part[half][i] = s;
    __syncthreads();
    if (!half) rs[i] = ys[i] - (part[0][i] + part[1][i]);
    __syncthreads();
  }
  const double v1 = apply_w(rs);
  if (!half) yr[i] = a0[i] + v1;
}

// Wt: the TRANSPOSE of the matrix applied (lower: inv(L11)^T block; upper: inv(L11) block)
int gps_launch_trsv_leaf_refine(gps_handle_t h, const double* Wt, const double* D, i64 ldd, double* y, i64 ldy, i64 r,
                                int upper) {
  if (r <= 0) return GPS_OK;
  LaunchScope ls(h, KC_TRSV, 6.0 * 128 * 128 * r, 2.5 * 128.0 * 128 * 8);
  hipLaunchKernelGGL(trsv_leaf_refine_kernel, dim3((unsigned)r), dim3(256), 0, h->stream, Wt, D, ldd, y, ldy, upper);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
