// Triangular solve of many rows against a 512-column diagonal block of the factor, as ONE launch.
//
//   forward  (RN = false):  X L^T = B     L [512, 512] lower (models/gpr.py:70-71: the rows below a factored block), W_j = inv(L_jj)
//   backward (RN = true) :  X L   = B     given U = L^T (upper, row-major) and the transposed block inverses W_j = inv(L_jj)^T
//
// in place in B [m, 512].  Launch by launch (blocked.hpp::trsm_rec down to 128 columns) this is 4 leaf products, two K = 128
// and one K = 256 update: seven launches that each stream the whole [m, 128 .. 256] panel through HBM for 2 m 128^2 flops --
// measured at m = 16384: 30 + 16 us per leaf, 63 us per K = 128 update, ~0.41 ms per 512 columns (8.6 GFLOP: 21 TFLOP/s), a
// sixth of the time of the N = 32768 factorisation spent in launches that cannot fill the MFMA pipes.
//
// Here a workgroup owns R = 16 RT rows of B and keeps all four 128-column blocks of them in MFMA accumulators for the whole
// substitution (RT = 4: 128 VGPRs):
//     for j: X_j = B_j W_j^T ;  B_i -= X_j L_ij^T  (i > j)             (backward: j = 3 .. 0, i < j, with U_ij)
// B is read once and X written once.  Each product is [R x 128] x [128 x 128]^T on v_mfma_f64_16x16x4_f64, one column
// tile of the 128 per wave (so a wave needs only ITS 16 rows of the right-hand operand), RT row tiles each:
//   * left operand (the current B_j, then X_j): accumulators -> LDS image As [R][130] (the accumulator layout is not the
//     operand layout), read back by conflict-free ds_read_b64 -- shared by the eight waves;
//   * right operand (rows 16 w .. of W_j / L_ij): streamed through a wave-private LDS region in two chunks of 64 k, the next
//     chunk loaded into registers (coalesced 16-byte loads, two 512-byte runs per instruction) while the current one feeds
//     the MFMAs -- private to the wave, so no workgroup barrier in the k loop.  Every workgroup reads the same ten 128 x 128
//     blocks: L2 hits.
// W_j is triangular: the k-steps that multiply structural zeros are skipped (the wave's k range is uniform).
// LDS: As 66.5 KB + 8 x 8.25 KB = 134 KB (RT = 4), one workgroup per CU, two waves per SIMD.
#include "gps_common.hpp"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));      // (HIP's d2 is a struct around a union: arrays of it passed by reference stay in scratch)

#define TP_NT 512
#define TP_LSA 130                 // As row stride (doubles)
// the right operand goes through the wave's region in chunks of KC = 64 or 32 k; row stride of the region KC + 2 doubles
// (132 / 68 dwords = 4 mod 64, as 130 is); 16 (KC + 2) doubles per wave

struct TrsmPanelArgs {
  double* B; i64 ldb;              // [m][512] in place
  const double* L; i64 ldl;        // the 512 x 512 diagonal block (lower; or U = L^T upper when backward)
  const double* W;                 // 4 block inverses [128][128] (transposed ones when backward)
  long long* stamps;               // diagnostics (gps_diag_trsm512_stamps): [workgroup][32] phase stamps of waves 0 and 7 (100 MHz ticks); else null
};

template <int KC>
__device__ __forceinline__ void tp_load_chunk(d2 (&v)[KC / 8], const double* rows, i64 ld, int kc, int lane) {
  // this wave's 16 rows x KC k: 128 / KC rows per instruction
  constexpr int LPR = KC / 2, RPI = 64 / LPR;          // lanes per row, rows per instruction
#pragma unroll
  for (int u = 0; u < KC / 8; ++u) v[u] = *reinterpret_cast<const d2*>(rows + (i64)(RPI * u + lane / LPR) * ld + kc * KC + 2 * (lane % LPR));
}
template <int KC>
__device__ __forceinline__ void tp_put_chunk(double* Bsw, const d2 (&v)[KC / 8], int lane) {
  constexpr int LPR = KC / 2, RPI = 64 / LPR;
#pragma unroll
  for (int u = 0; u < KC / 8; ++u) *reinterpret_cast<d2*>(Bsw + (RPI * u + lane / LPR) * (KC + 2) + 2 * (lane % LPR)) = v[u];
}

// acc[t] += As[16 t .. ][k] * rows[16 w + .][k]   for the k-steps [ks_lo, ks_hi) of 4 out of 32 (both multiples of 4; FULL: all 32).
// On entry v holds the first chunk of `rows` (loaded during the previous product); on exit the first chunk of `next_rows`:
// every load has the k-steps of a chunk to arrive in, across the barriers between the products too (the right operands do
// not depend on anything this kernel computes).
//
// The fragment reads run D k-steps ahead of the MFMAs that use them, in a ring of D + 1 register sets, and the order
// reads(s + D) -> MFMAs(s) is pinned (left alone the compiler emits read, s_waitcnt lgkmcnt(0), MFMA one by one: the LDS
// latency of every fragment exposed).  FULL: one pipeline over the 32 k-steps -- the next chunk goes into the wave's region
// when the reads of the current one have all been ISSUED (LDS operations of one wave execute in order).  Otherwise (the
// triangular products with the block inverses): groups of four k-steps under a wave-uniform test, a pipeline per group.
template <int RT, int KC, bool FULL>
__device__ __forceinline__ void tp_product(v4d (&acc)[RT], const double* As, double* Bsw, d2 (&v)[KC / 8], const double* rows, i64 ld,
                                           const double* next_rows, i64 next_ld, int ks_lo, int ks_hi, int lane, int fr, int fk) {
  constexpr int NC = 128 / KC, SPC = KC / 4;           // chunks per product, k-steps per chunk
  constexpr int D = (RT >= 4) ? 1 : 2;                 // (a k-step is RT MFMAs = RT * 64 cycles of the pipe)
  const double* pa = As + fr * TP_LSA + fk;
  const double* pb = Bsw + fr * (KC + 2) + fk;
  double fa[D + 1][RT], fb[D + 1];
  auto rd = [&](int set, int ks) {
    fb[set] = pb[4 * (ks % SPC)];
#pragma unroll
    for (int t = 0; t < RT; ++t) fa[set][t] = pa[16 * t * TP_LSA + 4 * ks];
  };
  auto mm = [&](int set) {
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][t], fb[set], acc[t], 0, 0, 0);
  };
  auto stage = [&](int c) {                            // chunk c into the wave's region, the one after it into v
    tp_put_chunk<KC>(Bsw, v, lane);
    if (c + 1 < NC) tp_load_chunk<KC>(v, rows, ld, c + 1, lane);
    else tp_load_chunk<KC>(v, next_rows, next_ld, 0, lane);
  };
  if (FULL) {
    stage(0);
#pragma unroll
    for (int d = 0; d < D; ++d) rd(d, d);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      const int nk = ks + D;
      if (nk < 32) {
        if (nk % SPC == 0) stage(nk / SPC);
        rd(nk % (D + 1), nk);
      }
      __builtin_amdgcn_sched_barrier(0);
      mm(ks % (D + 1));
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      stage(c);
#pragma unroll
      for (int q = 0; q < SPC / 4; ++q) {
        const int k0 = SPC * c + 4 * q;
        if (k0 >= ks_lo && k0 < ks_hi) {
#pragma unroll
          for (int d = 0; d < D; ++d) rd(d, k0 + d);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (s + D < 4) rd((s + D) % (D + 1), k0 + s + D);
            __builtin_amdgcn_sched_barrier(0);
            mm(s % (D + 1));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
}

// accumulator tiles -> As (rows 16 t + fk + 4 rg, columns 16 w + fr); NEG: the accumulators of the blocks still to be solved hold
// -B_i (the updates ADD X_j L_ij^T: no operand is negated in the k loops), the left operand of the next product is B_i
template <int RT, bool NEG>
__device__ __forceinline__ void tp_to_lds(double* As, const v4d (&x)[RT], int ct, int fr, int fk) {
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) As[(16 * t + fk + 4 * rg) * TP_LSA + 16 * ct + fr] = NEG ? -x[t][rg] : x[t][rg];
}

// KC = 64: one workgroup per CU (two waves per SIMD, up to 256 registers); KC = 32 (with RT = 2): 68 KB of LDS and 128 registers,
// TWO workgroups per CU -- one's loads, stores, barriers and unbalanced inverse products under the other's updates
template <int RT, bool RN, bool ST = false, int KC = 64>        // ST: the diagnostics build that leaves phase stamps (a few more registers)
__global__ __launch_bounds__(TP_NT) __attribute__((amdgpu_waves_per_eu(128 / KC, 128 / KC))) void trsm_panel_kernel(TrsmPanelArgs g) {
  extern __shared__ __attribute__((aligned(16))) char tp_smem[];
  constexpr int R = 16 * RT;
  double* As = reinterpret_cast<double*>(tp_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  double* Bsw = As + R * TP_LSA + wave * (16 * (KC + 2));
  // column tile of this wave: waves w and w + 4 share a SIMD, so they take the tiles w and 7 - w -- the triangular products skip
  // 4 (7 - ct) (forward) / 4 ct (backward) of the 32 k-steps of tile ct, and every SIMD then carries the same 36
  const int ct = (wave < 4) ? wave : 11 - wave;
  double* Brow = g.B + (i64)blockIdx.x * R * g.ldb;
#define TP_STAMP(q) do { if (ST && lane == 0 && (wave == 0 || wave == 7)) g.stamps[(i64)blockIdx.x * 32 + (wave ? 16 : 0) + (q)] = (long long)wall_clock64(); } while (0)
  if (KC == 32) __builtin_amdgcn_s_setprio(2);
  TP_STAMP(0);
  if (ST && tid == 0) {
    g.stamps[(i64)blockIdx.x * 32 + 14] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
    g.stamps[(i64)blockIdx.x * 32 + 15] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
  }

  // the right operands in the order they are used: W_j, then the blocks L_ij still to be solved
  auto w_rows = [&](int j) { return g.W + (i64)j * 128 * 128 + (i64)16 * ct * 128; };
  // forward: L_ij[c][k] = L[128 i + c][128 j + k] ; backward: U_ij[c][k] = U[128 i + c][128 j + k]  (i < j: above the diagonal)
  auto l_rows = [&](int i, int j) { return g.L + (i64)(128 * i + 16 * ct) * g.ldl + 128 * j; };
  d2 v[KC / 8];
  tp_load_chunk<KC>(v, w_rows(RN ? 3 : 0), 128, 0, lane);

  v4d acc[4][RT];
  {
    auto load_rows = [&](d2 (&t)[2 * RT], int i) {
#pragma unroll
      for (int u = 0; u < 2 * RT; ++u) t[u] = *reinterpret_cast<const d2*>(Brow + (i64)(2 * RT * wave + u) * g.ldb + 128 * i + 2 * lane);
    };
    auto put_rows = [&](const d2 (&t)[2 * RT]) {
#pragma unroll
      for (int u = 0; u < 2 * RT; ++u) *reinterpret_cast<d2*>(As + (2 * RT * wave + u) * TP_LSA + 2 * lane) = t[u];
    };
    // the block solved first: whole rows (1 KB per instruction) into As, where the solve wants it.  The others straight into
    // the accumulator layout (128-byte runs): they are not needed before the first update, one product away
    constexpr int o3 = RN ? 3 : 0;
    d2 t0[2 * RT];
    load_rows(t0, o3);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i != o3) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) acc[i][t][rg] = -Brow[(i64)(16 * t + fk + 4 * rg) * g.ldb + 128 * i + 16 * ct + fr];
      }
    put_rows(t0);
    __syncthreads();
  }
  TP_STAMP(1);

#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = RN ? 3 - jj : jj;
    // B_j (updated) as left operand
    if (jj > 0) {
      tp_to_lds<RT, true>(As, acc[j], ct, fr, fk);
      __syncthreads();
    }
    // X_j = B_j W_j^T.  forward: W_j lower, W[c][k] = 0 for k > c: column tile w needs k <= 16 w + 15.  backward: upper, k >= 16 w.
    v4d x[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) x[t] = v4d{0.0, 0.0, 0.0, 0.0};
    {
      // what comes after this product: the first block still to be solved, or (last block) nothing -- any valid address
      const int i1 = RN ? j - 1 : j + 1;
      const double* nx = (jj < 3) ? l_rows(i1, j) : w_rows(j);
      tp_product<RT, KC, false>(x, As, Bsw, v, w_rows(j), 128, nx, (jj < 3) ? g.ldl : 128, RN ? 4 * ct : 0, RN ? 32 : 4 * (ct + 1), lane, fr, fk);
    }
    TP_STAMP(2 + 3 * jj);
    __syncthreads();                       // everybody has read B_j
    tp_to_lds<RT, false>(As, x, ct, fr, fk);
    __syncthreads();
    // X_j is final: whole rows to HBM (wave w: rows 2 RT w ..), 1 KB per instruction
#pragma unroll
    for (int u = 0; u < 2 * RT; ++u) {
      const int row = 2 * RT * wave + u;
      const d2 val = *reinterpret_cast<const d2*>(As + row * TP_LSA + 2 * lane);
      *reinterpret_cast<d2*>(Brow + (i64)row * g.ldb + 128 * j + 2 * lane) = val;
    }
    TP_STAMP(3 + 3 * jj);
    // the blocks still to be solved (constant loop bounds: the accumulator array must keep compile-time indices)
    // two workgroups per CU: the one between its updates (unbalanced product, barriers, stores) goes first on the SIMD -- the
    // other's updates fill what it leaves
    if (KC == 32) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      if (q < 4 - jj) {
        const int i = RN ? j - q : j + q;
        const bool last = (q == 3 - jj);
        const int jn = RN ? j - 1 : j + 1;
        const double* nx = last ? w_rows(jn) : l_rows(RN ? i - 1 : i + 1, j);
        tp_product<RT, KC, true>(acc[i], As, Bsw, v, l_rows(i, j), g.ldl, nx, last ? 128 : g.ldl, 0, 32, lane, fr, fk);
      }
    }
    if (KC == 32) __builtin_amdgcn_s_setprio(2);
    __syncthreads();                       // As is rewritten by the next block
    TP_STAMP(4 + 3 * jj);
  }
#undef TP_STAMP
}

template <int RT, bool RN, bool ST = false, int KC = 64>
static int tp_launch(gps_handle_t h, const TrsmPanelArgs& a, i64 m) {
  constexpr int R = 16 * RT;
  const size_t lds = (size_t)(R * TP_LSA + 8 * 16 * (KC + 2)) * 8;
  int rc = gps_dyn_lds(h, reinterpret_cast<const void*>(&trsm_panel_kernel<RT, RN, ST, KC>), (int)lds);
  if (rc) return rc;
  LaunchScope ls(h, KC_GEMM, 10.0 * 2.0 * (double)m * 128.0 * 128.0, 2.0 * 8.0 * (double)m * 512.0);
  ls.tag[0] = m; ls.tag[1] = 512; ls.tag[2] = 512; ls.tag[3] = 1000 + (RN ? 1 : 0);
  hipLaunchKernelGGL((trsm_panel_kernel<RT, RN, ST, KC>), dim3((unsigned)(m / R)), dim3(TP_NT), lds, h->stream, a);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// B [m, 512] <- B L^-T (backward == 0: L lower at (L, ldl), W = the four block inverses) or B L^-1 (backward == 1: L is U = L^T
// upper, W = the four transposed block inverses).  m a multiple of 64.
int gps_launch_trsm_panel(gps_handle_t h, double* B, i64 ldb, i64 m, const double* L, i64 ldl, const double* W, int backward) {
  if (m <= 0) return GPS_OK;
  if (m % 64) return gps_fail(h, GPS_ERR_ARG, "trsm_panel: rows must be a multiple of 64");
  TrsmPanelArgs a{B, ldb, L, ldl, W, h->tp_stamps};
  // rows per workgroup: 64 unless that leaves CUs idle (one workgroup per CU: LDS)
  if (h->trsm_panel_rows == 33 && a.stamps) return backward ? tp_launch<2, true, true, 32>(h, a, m) : tp_launch<2, false, true, 32>(h, a, m);
  if (h->trsm_panel_rows == 33) return backward ? tp_launch<2, true, false, 32>(h, a, m) : tp_launch<2, false, false, 32>(h, a, m);
  const bool r64 = h->trsm_panel_rows == 64 || (h->trsm_panel_rows == 0 && m / 64 >= (i64)h->prop.multiProcessorCount);
  if (r64 && a.stamps) return backward ? tp_launch<4, true, true>(h, a, m) : tp_launch<4, false, true>(h, a, m);
  if (r64) return backward ? tp_launch<4, true>(h, a, m) : tp_launch<4, false>(h, a, m);
  return backward ? tp_launch<2, true>(h, a, m) : tp_launch<2, false>(h, a, m);
}
