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
//   * left operand (the current B_j, then X_j): accumulators -> LDS image [R][130] (the accumulator layout is not the
//     operand layout), read back by conflict-free ds_read_b128 -- shared by the eight waves;
//   * right operand (rows 16 w .. of W_j / L_ij): straight from L2 into MFMA operand registers.  The k index of a product is
//     summed over, so any bijection between (k-step, lane quarter fk) and k will do as long as both operands use the same
//     one: with k = 16 q + 4 fk + h for k-step 4 q + h, the four values a lane needs of a QUAD q of k-steps are 32 contiguous
//     bytes of its row (two 16-byte loads; the 16 rows of a wave: half a cache line each) and two ds_read_b128 of its row of
//     the LDS image.  The prefetch depth is a register ring of TP_NB quads (3 ahead of the MFMAs), carried ACROSS the
//     products and the barriers between them (the right operands do not depend on anything this kernel computes).  Every
//     workgroup reads the same ten 128 x 128 blocks: L2 hits.
// W_j is triangular: the quads that multiply structural zeros are skipped (the wave's range is uniform).
//
// Round 5 (tools/tp_stamps.py: phase stamps of every workgroup; docs/LAB_NOTES.md).  The first form staged the right operand
// through a wave-private LDS region (global -> registers -> LDS -> registers, 134 KB of LDS, one workgroup per CU) and left the
// fragment reads to the compiler: ds_read, s_waitcnt lgkmcnt(0), one MFMA, again -- 86.9 us per 64 rows at m = 2^20 of which
// 57.8 are MFMA time (10 % rows not yet loaded, 27 % the four triangular products at two thirds of the pipe, 7 % barriers and
// stores, updates at 87 %).  Now: fragment reads pinned one k-step pair ahead of their MFMAs (sched_barrier), no LDS staging,
// and two launch shapes --
//   * trsm_panel_kernel<RT = 2>: 32 rows per workgroup, 33 KB of LDS and 128 registers: TWO workgroups per CU, one's loads,
//     barriers, stores and unbalanced triangular products under the other's updates (m < 64 rows x the number of CUs);
//   * trsm_panel_persistent_kernel (RT = 4): one workgroup per CU walks over its row blocks with the NEXT block's rows
//     arriving in the accumulator registers of every column block as soon as that block has been solved (larger m).
// m = 2^20: 5.69 -> 4.80 ms (-16 %); m = 4096: 54 -> 44 us.
#include "gps_common.hpp"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));      // (HIP's d2 is a struct around a union: arrays of it passed by reference stay in scratch)

#define TP_NT 512
#define TP_LSA 130                 // row stride of an LDS image (doubles)

struct TrsmPanelArgs {
  double* B; i64 ldb;              // [m][512] in place
  const double* L; i64 ldl;        // the 512 x 512 diagonal block (lower; or U = L^T upper when backward)
  const double* W;                 // 4 block inverses [128][128] (transposed ones when backward)
  long long* stamps;               // diagnostics (gps_diag_trsm512_stamps): [row block][32] phase stamps of waves 0 and 7 (100 MHz ticks); else null
};

// accumulator tiles -> an LDS image (rows 16 t + fk + 4 rg, columns 16 w + fr), negated or not
template <int RT, bool NEG>
__device__ __forceinline__ void tp_to_lds(double* As, const v4d (&x)[RT], int ct, int fr, int fk) {
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) As[(16 * t + fk + 4 * rg) * TP_LSA + 16 * ct + fr] = NEG ? -x[t][rg] : x[t][rg];
}

// quads of the right operand in registers (a ring; must divide 8)
#define TP_NB 4

// per-lane pointer to the wave's rows of a right operand + the quads [lo, hi) of it that are not structurally zero
// (a wave-uniform base and a 32-bit byte offset of the lane: scalar base + vector offset addressing, no 64-bit pointer per operand)
struct TpOpnd { const double* base; unsigned off; int lo, hi; };
__device__ __forceinline__ const double* tp_at(const double* base, unsigned byte_off) {
  return reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}

template <int NB>
__device__ __forceinline__ void tp_ldq(d2 (&bq)[NB][2], int slot, const TpOpnd& o, int q) {
  const double* p = tp_at(o.base + 16 * q, o.off);
  bq[slot][0] = *reinterpret_cast<const d2*>(p);
  bq[slot][1] = *reinterpret_cast<const d2*>(p + 2);
}

// acc[t] += As[16 t .. ][k] * rows[16 w + .][k] over the quads [cur.lo, cur.hi) (FULL: all eight).  On entry the ring holds
// the quads 0 .. TP_NB - 1 of `cur` (those in range), on exit those of `nxt`.
template <int RT, bool FULL, int NB>
__device__ __forceinline__ void tp_product(v4d (&acc)[RT], const double* pa, d2 (&bq)[NB][2], const TpOpnd cur, const TpOpnd nxt) {
  d2 fa[2][RT];
  auto rd = [&](int P) {                          // fragments of the k-step pair P (k-steps 2 P, 2 P + 1) into set P & 1
#pragma unroll
    for (int t = 0; t < RT; ++t) fa[P & 1][t] = *reinterpret_cast<const d2*>(pa + 16 * t * TP_LSA + 16 * (P >> 1) + 2 * (P & 1));
  };
  auto mm = [&](int P) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int t = 0; t < RT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[P & 1][t][hh], bq[(P >> 1) % NB][P & 1][hh], acc[t], 0, 0, 0);
  };
  auto refill = [&](int q) {                      // quad q is used up: its slot takes the quad NB further on
    const int n = q + NB;
    if (n < 8) { if (FULL || (n >= cur.lo && n < cur.hi)) tp_ldq(bq, q % NB, cur, n); }
    else if (n - 8 >= nxt.lo && n - 8 < nxt.hi) tp_ldq(bq, q % NB, nxt, n - 8);
  };
  if (FULL) {
    rd(0);
#pragma unroll
    for (int P = 0; P < 16; ++P) {
      if (P + 1 < 16) rd(P + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(P);
      __builtin_amdgcn_sched_barrier(0);
      if (P & 1) refill(P >> 1);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q >= cur.lo && q < cur.hi) {
        if (q == 0 || q == cur.lo) rd(2 * q);     // (otherwise read during the quad before)
        rd(2 * q + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(2 * q);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < 8 && q + 1 < cur.hi) rd(2 * q + 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(2 * q + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      refill(q);
    }
  }
}

// One row block of R = 16 RT rows per workgroup.  RT = 2: 33 KB of LDS, 128 registers, TWO workgroups per CU -- one's loads,
// stores, barriers and unbalanced triangular products under the other's updates; RT = 4: one per CU (diagnostics).  The
// accumulators of the blocks still to be solved hold -B_i (the updates ADD X_j L_ij^T: no operand is negated in the k loops).
template <int RT, bool RN, bool ST = false>
__global__ __launch_bounds__(TP_NT) __attribute__((amdgpu_waves_per_eu(8 / RT, 8 / RT))) void trsm_panel_kernel(TrsmPanelArgs g) {
  extern __shared__ __attribute__((aligned(16))) char tp_smem[];
  constexpr int R = 16 * RT;
  double* As = reinterpret_cast<double*>(tp_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int ct = (wave < 4) ? wave : 11 - wave;    // (as in the first form: every SIMD carries the same 36 k-steps of a triangular product)
  double* Brow = g.B + (i64)blockIdx.x * R * g.ldb;
#define TP_STAMP(q) do { if (ST && lane == 0 && (wave == 0 || wave == 7)) g.stamps[(i64)blockIdx.x * 32 + (wave ? 16 : 0) + (q)] = (long long)wall_clock64(); } while (0)
  TP_STAMP(0);
  if (ST && tid == 0) {
    g.stamps[(i64)blockIdx.x * 32 + 14] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
    g.stamps[(i64)blockIdx.x * 32 + 15] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
  }
  // the right operands: W_j (triangular: forward the quads [0, ct + 1), backward [ct, 8)) and the blocks L_ij still to be solved
  const unsigned w_off = (unsigned)(((16 * ct + fr) * 128 + 4 * fk) * 8), l_off = (unsigned)(((i64)(16 * ct + fr) * g.ldl + 4 * fk) * 8);
  auto w_op = [&](int j) { return TpOpnd{g.W + (i64)j * 128 * 128, w_off, RN ? ct : 0, RN ? 8 : ct + 1}; };
  auto l_op = [&](int i, int j) { return TpOpnd{g.L + (i64)(128 * i) * g.ldl + 128 * j, l_off, 0, 8}; };
  const TpOpnd none{g.W, 0u, 0, 0};
  d2 bq[TP_NB][2];
  {
    const TpOpnd first = w_op(RN ? 3 : 0);
#pragma unroll
    for (int q = 0; q < TP_NB; ++q)
      if (q >= first.lo && q < first.hi) tp_ldq(bq, q, first, q);
  }

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
  const double* pa = As + fr * TP_LSA + 4 * fk;

#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = RN ? 3 - jj : jj;
    if (jj > 0) {
      tp_to_lds<RT, true>(As, acc[j], ct, fr, fk);
      __syncthreads();
    }
    v4d x[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) x[t] = v4d{0.0, 0.0, 0.0, 0.0};
    tp_product<RT, false, TP_NB>(x, pa, bq, w_op(j), (jj < 3) ? l_op(RN ? j - 1 : j + 1, j) : none);
    TP_STAMP(2 + 3 * jj);
    __syncthreads();                       // everybody has read B_j
    tp_to_lds<RT, false>(As, x, ct, fr, fk);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2 * RT; ++u) {
      const int row = 2 * RT * wave + u;
      const d2 val = *reinterpret_cast<const d2*>(As + row * TP_LSA + 2 * lane);
      *reinterpret_cast<d2*>(Brow + (i64)row * g.ldb + 128 * j + 2 * lane) = val;
    }
    TP_STAMP(3 + 3 * jj);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      if (q < 4 - jj) {
        const int i = RN ? j - q : j + q;
        const bool last = (q == 3 - jj);
        tp_product<RT, true, TP_NB>(acc[i], pa, bq, l_op(i, j), last ? w_op(RN ? j - 1 : j + 1) : l_op(RN ? i - 1 : i + 1, j));
      }
    }
    __syncthreads();                       // As is rewritten by the next block
    TP_STAMP(4 + 3 * jj);
  }
#undef TP_STAMP
}

template <int RT, bool RN, bool ST = false>
static int tp_launch(gps_handle_t h, const TrsmPanelArgs& a, i64 m) {
  constexpr int R = 16 * RT;
  const size_t lds = (size_t)(R * TP_LSA) * 8;
  int rc = gps_dyn_lds(h, reinterpret_cast<const void*>(&trsm_panel_kernel<RT, RN, ST>), (int)lds);
  if (rc) return rc;
  LaunchScope ls(h, KC_GEMM, 10.0 * 2.0 * (double)m * 128.0 * 128.0, 2.0 * 8.0 * (double)m * 512.0);
  ls.tag[0] = m; ls.tag[1] = 512; ls.tag[2] = 512; ls.tag[3] = 1000 + (RN ? 1 : 0);
  hipLaunchKernelGGL((trsm_panel_kernel<RT, RN, ST>), dim3((unsigned)(m / R)), dim3(TP_NT), lds, h->stream, a);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// ---- the same, persistent -------------------------------------------------------------------------------------------------
// One workgroup per CU walks over row blocks b, b + gridDim.x, ...  LDS holds TWO images [64][130]: A -- the column block being
// solved (B_j, the left operand of the product with W_j) and X -- the solved one (-X_j, the left operand of the updates, which
// then ADD: the accumulators hold +B_i and nothing is negated in the k loops).  X_j goes into the other image, so there is no
// barrier between "everybody has read B_j" and writing it.  The NEXT row block arrives while this one is being solved: the
// accumulators of a column block are reloaded (plain loads into the registers the block lives in, nothing depends on them
// until the next row block) as soon as the block has been solved -- the 10 us a workgroup of the kernel above spends waiting
// for its rows (13 %) are gone, and no extra register is held.  What is left of them: a wave's loads return in order, so the
// first wait for a right-operand quad issued after these HBM loads waits for them too (~1.5 us per column block).  A start
// stagger of the workgroups did not change it; issuing them at different times in the two waves of a SIMD, so that the matrix
// pipe keeps one of them, spilled registers inside the loop (256 are in use) and cost 40 %.
template <int RT, bool RN, bool ST = false>
__global__ __launch_bounds__(TP_NT) __attribute__((amdgpu_waves_per_eu(8 / RT, 8 / RT))) void trsm_panel_persistent_kernel(TrsmPanelArgs g, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) char tp_smem[];
  constexpr int R = 16 * RT, NB = (RT == 2) ? 2 : TP_NB;       // (two workgroups per CU: 128 registers -- a ring of two quads)
  double* Aimg = reinterpret_cast<double*>(tp_smem);
  double* Ximg = Aimg + R * TP_LSA;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int ct = (wave < 4) ? wave : 11 - wave;
  unsigned w_off = (unsigned)(((16 * ct + fr) * 128 + 4 * fk) * 8), l_off = (unsigned)(((i64)(16 * ct + fr) * g.ldl + 4 * fk) * 8);
  auto w_op = [&](int j) { return TpOpnd{g.W + (i64)j * 128 * 128, w_off, RN ? ct : 0, RN ? 8 : ct + 1}; };
  auto l_op = [&](int i, int j) { return TpOpnd{g.L + (i64)(128 * i) * g.ldl + 128 * j, l_off, 0, 8}; };
  constexpr int o3 = RN ? 3 : 0;             // the column block solved first
  d2 bq[NB][2];
  {
    const TpOpnd first = w_op(o3);
#pragma unroll
    for (int q = 0; q < NB; ++q)
      if (q >= first.lo && q < first.hi) tp_ldq(bq, q, first, q);
  }
  unsigned row_off = (unsigned)(2 * lane * 8);                                            // whole rows: 16 bytes per lane
  unsigned acc_off = (unsigned)(((i64)fk * g.ldb + 16 * ct + fr) * 8);                    // accumulator layout: 128-byte runs
  auto load_acc = [&](v4d (&a)[RT], const double* Brow, int i) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) a[t][rg] = *tp_at(Brow + (i64)(16 * t + 4 * rg) * g.ldb + 128 * i, acc_off);
  };
  const double* pA = Aimg + fr * TP_LSA + 4 * fk;
  const double* pX = Ximg + fr * TP_LSA + 4 * fk;

  v4d acc[4][RT];
  int b = blockIdx.x;
  {
    const double* Brow = g.B + (i64)b * R * g.ldb;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) load_acc(acc[RN ? 3 - jj : jj], Brow, RN ? 3 - jj : jj);
  }
  for (; b < nblocks; b += gridDim.x) {
    double* Brow = g.B + (i64)b * R * g.ldb;
    const int nb = b + gridDim.x;
    const bool has_next = nb < nblocks;
    const double* Bnext = g.B + (i64)(has_next ? nb : b) * R * g.ldb;
    // (the lane offsets are loop invariants: left alone, every base + offset sum of the body is hoisted out of the loop as a
    // 64-bit register pair per lane -- 13 of them spilled -- instead of scalar base + 32-bit offset addressing at the use)
    asm volatile("" : "+v"(w_off), "+v"(l_off), "+v"(row_off), "+v"(acc_off));
#define TP_STAMP(q) do { if (ST && lane == 0 && (wave == 0 || wave == 7)) g.stamps[(i64)b * 32 + (wave ? 16 : 0) + (q)] = (long long)wall_clock64(); } while (0)
    TP_STAMP(0);
    if (ST && tid == 0) {
      g.stamps[(i64)b * 32 + 14] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
      g.stamps[(i64)b * 32 + 15] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = RN ? 3 - jj : jj;
      // B_j (updated) as the left operand.  (Everybody is through with the updates of the block before when the barrier
      // falls, and so with X; with A since the barrier after the product before.)
      tp_to_lds<RT, false>(Aimg, acc[j], ct, fr, fk);
      __syncthreads();
      if (jj == 0) TP_STAMP(1);
      // X_j = B_j W_j^T in the registers of the block's own accumulators
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[j][t] = v4d{0.0, 0.0, 0.0, 0.0};
      tp_product<RT, false, NB>(acc[j], pA, bq, w_op(j), (jj < 3) ? l_op(RN ? j - 1 : j + 1, j) : w_op(o3));
      TP_STAMP(2 + 3 * jj);
      tp_to_lds<RT, true>(Ximg, acc[j], ct, fr, fk);
      // the same column block of the next row block
      if (has_next) load_acc(acc[j], Bnext, j);
      __syncthreads();
      // X_j is final: whole rows to HBM (wave w: rows 8 w ..), 1 KB per instruction
#pragma unroll
      for (int u = 0; u < 2 * RT; ++u) {
        const int row = 2 * RT * wave + u;
        const d2 val = *reinterpret_cast<const d2*>(Ximg + row * TP_LSA + 2 * lane);
        *reinterpret_cast<d2*>(const_cast<double*>(tp_at(Brow + (i64)row * g.ldb + 128 * j, row_off))) = -val;
      }
      TP_STAMP(3 + 3 * jj);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        if (q < 4 - jj) {
          const int i = RN ? j - q : j + q;
          const bool last = (q == 3 - jj);
          tp_product<RT, true, NB>(acc[i], pX, bq, l_op(i, j), last ? w_op(RN ? j - 1 : j + 1) : l_op(RN ? i - 1 : i + 1, j));
        }
      }
      TP_STAMP(4 + 3 * jj);
    }
#undef TP_STAMP
  }
}

template <int RT, bool RN, bool ST = false>
static int tp_launch_persistent(gps_handle_t h, const TrsmPanelArgs& a, i64 m) {
  constexpr int R = 16 * RT;
  const size_t lds = (size_t)(2 * R * TP_LSA) * 8;
  int rc = gps_dyn_lds(h, reinterpret_cast<const void*>(&trsm_panel_persistent_kernel<RT, RN, ST>), (int)lds);
  if (rc) return rc;
  LaunchScope ls(h, KC_GEMM, 10.0 * 2.0 * (double)m * 128.0 * 128.0, 2.0 * 8.0 * (double)m * 512.0);
  ls.tag[0] = m; ls.tag[1] = 512; ls.tag[2] = 512; ls.tag[3] = 1000 + (RN ? 1 : 0);
  const int nblocks = (int)(m / R);
  const int grid = std::min(nblocks, h->prop.multiProcessorCount * (4 / RT));
  hipLaunchKernelGGL((trsm_panel_persistent_kernel<RT, RN, ST>), dim3((unsigned)grid), dim3(TP_NT), lds, h->stream, a, nblocks);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// B [m, 512] <- B L^-T (backward == 0: L lower at (L, ldl), W = the four block inverses) or B L^-1 (backward == 1: L is U = L^T
// upper, W = the four transposed block inverses).  m a multiple of 64.
int gps_launch_trsm_panel(gps_handle_t h, double* B, i64 ldb, i64 m, const double* L, i64 ldl, const double* W, int backward) {
  if (m <= 0) return GPS_OK;
  if (m % 64) return gps_fail(h, GPS_ERR_ARG, "trsm_panel: rows must be a multiple of 64");
  TrsmPanelArgs a{B, ldb, L, ldl, W, h->tp_stamps};
  // the persistent form when every CU gets a block of 64 rows, else 32 rows per workgroup and two workgroups per CU
  // (trsm_panel_rows: 64 / 32 force one of them, 65 = 64 rows per workgroup, not persistent: diagnostics)
  const int form = h->trsm_panel_rows ? h->trsm_panel_rows : (m / 64 >= (i64)h->prop.multiProcessorCount ? 64 : 32);
  if (form == 64) {
    if (a.stamps) return backward ? tp_launch_persistent<4, true, true>(h, a, m) : tp_launch_persistent<4, false, true>(h, a, m);
    return backward ? tp_launch_persistent<4, true>(h, a, m) : tp_launch_persistent<4, false>(h, a, m);
  }
  if (form == 65) {
    if (a.stamps) return backward ? tp_launch<4, true, true>(h, a, m) : tp_launch<4, false, true>(h, a, m);
    return backward ? tp_launch<4, true>(h, a, m) : tp_launch<4, false>(h, a, m);
  }
  if (a.stamps) return backward ? tp_launch<2, true, true>(h, a, m) : tp_launch<2, false, true>(h, a, m);
  return backward ? tp_launch<2, true>(h, a, m) : tp_launch<2, false>(h, a, m);
}
