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
// The 128x128 C -= A B^T kernel has a second instantiation whose steady-state K-loop body is branch-free and
// ordered with __builtin_amdgcn_sched_group_barrier (ablation: with one workgroup pair per CU the MFMA + LDS-read
// stream alone runs at the MFMA peak; 9 % of the loop went to the global->LDS staging at the slab boundary and 3.5 %
// to the barrier): the fragment reads are interleaved two MFMAs apart, the next slab's ds_write_b128 with the MFMAs
// before the barrier, and the slab's last 8 MFMAs are deferred across the barrier (they only need registers) so that
// they cover the next slab's first fragment reads.  With that the K loop runs at the rate of the staging-free
// ablation (847 vs 845 us per K = 4096 tile).
//
// Tiling (MI355X first): BM x BN output tile per 256-thread workgroup (4 waves, each owning
// MI x NI v_mfma_f64_16x16x4_f64 accumulators: 4x4 at 128x128), BK = 16 staged through LDS
// with a register prefetch of the next K-slab.  Row stride in LDS is 18 doubles: the
// 16 rows x 2 k of one ds_read_b64 half-wave then cover all 64 banks exactly once.
//
// fp64 MFMA is slow *per CU* (128 FLOP/clk: a 128x128x128 tile is 15 us of one CU), so
// what matters for the many small launches on the factorisation's critical path is to
// spread a launch over all 256 CUs: the launcher picks 128x128, 64x64 or 32x32 tiles (and
// 64/32/16 x 128 row panels for the in-place B <- B W^T leaves) so that the grid has
// enough workgroups.  128x128: 2 workgroups per CU (147 KB LDS, <=256 VGPR) so that one
// workgroup's C read-modify-write epilogue hides under the other's MFMA stream.
// blockIdx -> tile mapping is XCD-aware: each XCD walks a contiguous range of
// 8-tile-wide column strips, so the tiles resident on one XCD share operand panels in
// its private L2.
#include "gps_common.hpp"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define BK_MIN 16      // K must be a multiple of this (the slab depth of the square tiles)
#ifndef STRIP
#define STRIP 8        // tile columns per strip
#endif

struct GemmArgs {
  const double* A; const double* B; double* C;
  i64 lda, ldb, ldc;
  int Tm, Tn;       // tiles
  int K;
  int ntiles;
  // Triangular operands (the zero part is never read; K loop cut to the non-zero range of the tile):
  //   1: A upper triangular (M == K; A[i][k] == 0 for k < i): tile row tm starts at k = tm*BM
  //   2: A lower triangular (M == K; k > i zero): tile row tm ends at k = (tm+1)*BM
  //   3: B lower triangular (N == K; B[n][k] == 0 for k > n): tile column tn ends at k = (tn+1)*BN
  int triA;
  // Batch (batch > 1; round 6: the 2048-column inverse blocks of a factor are built by batched products over all diagonal
  // blocks at once): problem p reads / writes its operands at  base + (p * rs) * ld + (p * cs) % cm  (cm == 0: no wrap) --
  // blocks that step along a diagonal, possibly inside a buffer that stacks wide diagonal blocks (column offset modulo the width).
  int batch, tpb;   // problems ; tiles per problem (ntiles = batch * tpb)
  int pair;         // workgroups take two mirror tiles (kernel template PAIR): 1 tile columns, 2 tile rows; ntiles / tpb count the PAIRS
  i64 a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, a_cm, b_cm, c_cm;
  // Block-cyclic columns (cb_tiles > 0; the multi-GPU trailing update): tile column tn lies in owned block
  // tn / cb_tiles, whose rows of B and columns of C start cb_stride elements after those of the previous owned
  // block; tiles entirely above a block's first row are skipped (the update is lower-trapezoidal).
  int cb_tiles;
  i64 cb_stride;
  int cb_cpack = 0;   // C holds the owned blocks side by side (column b * nb, not b * cb_stride): partitioned storage
  // Tail split (nsplit > 1, grid = nfull + (ntiles - nfull) * nsplit): see the comment at the kernel.
  int nfull, nsplit;
  double* ws;         // [(ntiles - nfull) * nsplit][BM * BN] slice partials
  unsigned* cnt;      // [ntiles - nfull] arrival counters (left at 0)
  // Look-ahead hand-over (blocked.hpp::potrf_rl_groups): the first workgroup publishes sig_val at sig_ptr when the
  // kernel STARTS -- on an in-order stream that means "everything launched before this kernel has completed".
  unsigned long long* sig_ptr; unsigned long long sig_val;
  long long* stamps;  // diagnostics (gps_diag_gemm_timeline): [blockIdx][6] = start, end, HW_ID, XCC_ID, K loop start, K loop end (100 MHz ticks); else null
};

// bijective XCD remap: blocks b, b+8, b+16 ... (same XCD) get consecutive ids
__device__ __forceinline__ int xcd_remap(int b, int nb) {
#ifdef GPS_GEMM_NO_XCD_REMAP      // experiment (tools/power_vs_traffic.sh): neighbouring tiles on different XCDs -- every L2 fetches every panel
  return b;
#endif
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
    // lower trapezoid of a Tm x Tn tile grid (Tm >= Tn; Tm == Tn: the lower triangle), strip s covers columns
    // [8s, 8s+w), rows 8s .. Tm-1; inside the diagonal super-block only tn <= tm.
    int s = 0, rem = id;
    for (;;) {
      const int c0 = s * STRIP;
      const int w = min(STRIP, Tn - c0);
      const int tri = w * (w + 1) / 2;           // diagonal super-block
      const int cnt = tri + (Tm - c0 - w) * w;
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

// BM x BN output tile, 4 waves arranged WGM x (4/WGM); every wave owns a
// (BM/WGM) x (BN/WGN) sub-tile = MI x NI accumulators of 16x16.
// BKT: K-slab depth.  The per-slab cost of the staging and the barrier (~0.3 us) is fixed, so the small tiles, whose
// slab holds only 4-16 MFMAs per wave, use deeper slabs (fewer of them); the square MFMA-bound tiles keep 16.
#define GEMM_KERNEL_NAME gemm_nt_f64_kernel
#define GEMM_PAIR_TPARAM
#include "gemm_f64_kernel.inc"
#undef GEMM_KERNEL_NAME
#undef GEMM_PAIR_TPARAM
#define GEMM_PAIRED
#define GEMM_KERNEL_NAME gemm_nt_f64_pair_kernel
#define GEMM_PAIR_TPARAM , int PAIR = 1
#include "gemm_f64_kernel.inc"
#undef GEMM_PAIRED
#undef GEMM_KERNEL_NAME
#undef GEMM_PAIR_TPARAM

template <int BM, int BN, int WGM, bool LOWER, int OP, int PIPE = 0, int BKT = 16>
static int launch_variant(gps_handle_t h, const GemmArgs& g) {
  const size_t lds = (size_t)(2 * (BM + BN) * (BKT + 2)) * sizeof(double);
  // paired tiles: the square scheduled tiles with a triangular operand, plain C = +- A B^T (what the wide inverse blocks use)
  if (g.pair && BM == BN && PIPE == 2 && !LOWER && (OP == 1 || OP == 3)) {
    const int grid = g.ntiles;          // (already halved: launch_cfg)
    if (g.pair == 1) {
      int rc = gps_dyn_lds(h, reinterpret_cast<const void*>(&gemm_nt_f64_pair_kernel<BM, BN, WGM, false, OP, PIPE, BKT, 1>), (int)lds);
      if (rc) return rc;
      hipLaunchKernelGGL((gemm_nt_f64_pair_kernel<BM, BN, WGM, false, OP, PIPE, BKT, 1>), dim3(grid), dim3(256), lds, h->stream, g);
    } else {
      int rc = gps_dyn_lds(h, reinterpret_cast<const void*>(&gemm_nt_f64_pair_kernel<BM, BN, WGM, false, OP, PIPE, BKT, 2>), (int)lds);
      if (rc) return rc;
      hipLaunchKernelGGL((gemm_nt_f64_pair_kernel<BM, BN, WGM, false, OP, PIPE, BKT, 2>), dim3(grid), dim3(256), lds, h->stream, g);
    }
    GPS_HIP(h, hipGetLastError());
    return GPS_OK;
  }
  int rc = gps_dyn_lds(h, reinterpret_cast<const void*>(&gemm_nt_f64_kernel<BM, BN, WGM, LOWER, OP, PIPE, BKT>), (int)lds);
  if (rc) return rc;
  const int grid = g.nsplit > 1 ? g.nfull + (g.ntiles - g.nfull) * g.nsplit : g.ntiles;
  hipLaunchKernelGGL((gemm_nt_f64_kernel<BM, BN, WGM, LOWER, OP, PIPE, BKT>), dim3(grid), dim3(256), lds, h->stream, g);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

template <int BM, int BN, int WGM, int PIPE, int BKT = 16>
static int dispatch_ops(gps_handle_t h, int op, int lower, const GemmArgs& g) {
  if (BM == BN && lower) {      // (M > N: lower trapezoid -- the square's lower triangle plus all rows below it)
    if (op == 0) return launch_variant<BM, BN, WGM, true, 0, PIPE, BKT>(h, g);
    if (op == 2) return launch_variant<BM, BN, WGM, true, 2, PIPE, BKT>(h, g);
    if (op == 3) return gps_fail(h, GPS_ERR_ARG, "gemm_nt: op 3 has no lower-triangle form");
    return launch_variant<BM, BN, WGM, true, 1, PIPE, BKT>(h, g);
  }
  if (op == 0) return launch_variant<BM, BN, WGM, false, 0, PIPE, BKT>(h, g);
  if (op == 2) return launch_variant<BM, BN, WGM, false, 2, PIPE, BKT>(h, g);
  if (op == 3) return launch_variant<BM, BN, WGM, false, 3, PIPE, BKT>(h, g);
  return launch_variant<BM, BN, WGM, false, 1, PIPE, BKT>(h, g);
}

template <int BM, int BN, int WGM>
static int launch_cfg(gps_handle_t h, int op, int lower, GemmArgs& g, i64 M, i64 N) {
  g.Tm = (int)(M / BM); g.Tn = (int)(N / BN);
  i64 nt = lower ? (i64)g.Tn * (g.Tn + 1) / 2 + (i64)(g.Tm - g.Tn) * g.Tn : (i64)g.Tm * g.Tn;
  if (g.cb_tiles > 0) {                  // needed tiles of the lower-trapezoidal block-cyclic update
    const i64 q = g.cb_stride / BM, nblocks = g.Tn / g.cb_tiles;
    nt = 0;
    for (i64 ob = 0; ob < nblocks; ++ob) nt += ((i64)g.Tm - ob * q) * g.cb_tiles;
  }
  // a triangular operand on the square scheduled tiles (C = +- A B^T): mirror tiles in pairs when the tile count allows it
  g.pair = 0;
  if (g.triA && !lower && g.cb_tiles == 0 && BM == BN && (BM == 128 || BM == 64) && (op == 1 || op == 3) && g.C != g.A && h->gemm_pair) {
    if (g.triA == 3 && g.Tn % 2 == 0) g.pair = 1;
    if (g.triA != 3 && g.Tm % 2 == 0) g.pair = 2;
    if (g.pair) nt /= 2;
  }
  g.tpb = (int)nt;
  if (g.batch > 1) nt *= g.batch;
  if (nt > 0x7fffffff) return gps_fail(h, GPS_ERR_ARG, "gemm_nt: too many tiles");
  g.ntiles = (int)nt;
  // tail split: 128x128 tiles only (512 resident slots on 256 CUs), never for the in-place or triangular-A forms
  g.nfull = g.ntiles; g.nsplit = 1; g.ws = nullptr; g.cnt = nullptr;
  // (not on the look-ahead side stream either: the slice work space is shared by all launches of the handle)
  if (BM == 128 && BN == 128 && h->gemm_tail_split && !g.triA && g.batch <= 1 && g.cb_tiles == 0 && g.C != g.A && h->prop.multiProcessorCount == 256 &&
      (h->side_stream == nullptr || h->stream != h->side_stream) && (h->def_stream == nullptr || h->stream != h->def_stream)) {
    const int slots = 512;
    const int nfull = (g.ntiles / slots) * slots, r = g.ntiles - nfull;
    int ns = r > 0 ? slots / r : 1;
    if (ns > h->gemm_tail_max_slices) ns = h->gemm_tail_max_slices;
    while (ns > 1 && (g.K / BK_MIN) / ns < 8) --ns;           // at least 8 slabs (K = 128) per slice
    if (nfull >= slots && ns > 1) {
      DevBuf& ws = h->dGemmWs;
      DevBuf& cnt = h->dGemmCnt;
      GPS_HIP(h, ws.ensure((size_t)r * ns * BM * BN * sizeof(double)));
      if (cnt.cap == 0) {
        GPS_HIP(h, cnt.ensure(512 * sizeof(unsigned)));
        GPS_HIP(h, hipMemsetAsync(cnt.p, 0, 512 * sizeof(unsigned), h->stream));
      }
      g.nfull = nfull; g.nsplit = ns; g.ws = ws.d(); g.cnt = (unsigned*)cnt.p;
    }
  }
  // scheduled K loop (see the kernel) for the square 128x128 and 64x64 tiles
  constexpr int P = ((BM == 128 && BN == 128) || (BM == 64 && BN == 64)) ? 2 : 0;
  if (P) return dispatch_ops<BM, BN, WGM, P>(h, op, lower, g);
  // deeper K slabs for the latency-bound small tiles (when K allows it): 32x32 tiles 64-deep, the 16/32 x 128 row panels 32-deep
  constexpr int DEEP = (BM == 32 && BN == 32) ? 64 : ((BN == 128 && BM <= 32) ? 32 : 16);
  if (DEEP > 16 && g.K % DEEP == 0) return dispatch_ops<BM, BN, WGM, 0, DEEP>(h, op, lower, g);
  return dispatch_ops<BM, BN, WGM, 0>(h, op, lower, g);
}

// rowpanel != 0: C may alias A (in-place B <- B W^T with N == K == 128): every workgroup then owns
// complete rows (BN = N = 128), so it has consumed all of its A rows before it stores.
// lower: 0 = all of C, 1 = lower triangle of a square C only -- or, with M > N, the lower trapezoid: that triangle plus
// every row below it (the rows of right-hand sides that ride along under a factorisation) --, 2 = all of C with A (M == K) upper triangular:
// the zero part of A is skipped (half the flops; used by the L^-T / K^-1 recursions of the gradient).
int gps_launch_gemm_nt(gps_handle_t h, int op, int lower, i64 M, i64 N, i64 K,
                       const double* A, i64 lda, const double* B, i64 ldb,
                       double* C, i64 ldc) {
  return gps_launch_gemm_nt_ex(h, op, lower == 2 ? 0 : lower, lower == 2 ? 1 : 0, M, N, K, A, lda, B, ldb, C, ldc, nullptr);
}

// tri: 0 none, 1 A upper triangular, 2 A lower triangular (both M == K), 3 B lower triangular (N == K): GemmArgs::triA.
// bt: batch of equal problems whose operands step along diagonals (GemmBatch, gps_common.hpp) or nullptr.
int gps_launch_gemm_nt_ex(gps_handle_t h, int op, int lower, int tri, i64 M, i64 N, i64 K,
                          const double* A, i64 lda, const double* B, i64 ldb,
                          double* C, i64 ldc, const GemmBatch* bt) {
  if (M <= 0 || N <= 0 || K <= 0) return GPS_OK;
  const int triA = tri;
  if (triA) {
    if (triA < 0 || triA > 3 || lower) return gps_fail(h, GPS_ERR_ARG, "gemm_nt: bad triangular mode");
    if ((triA == 3 ? N : M) != K) return gps_fail(h, GPS_ERR_ARG, "gemm_nt: a triangular operand must be square");
  }
  const i64 nbatch = (bt && bt->batch > 1) ? bt->batch : 1;
  // (M a multiple of 64 only: half a tile row of right-hand sides -- predict_f on at most 64 test points, round 6 --, with the
  // 64 x 64 / 32 x 32 tiles or the 16-row in-place panels; not for the lower-triangle form or a triangular A)
  const bool half_rows = (M % 128 == 64) && !lower && triA != 1 && triA != 2 && nbatch == 1;
  if ((M % 128 && !half_rows) || N % 128 || K % BK_MIN || (lower && M < N))
    return gps_fail(h, GPS_ERR_ARG, "gemm_nt: M,N must be multiples of 128 and K of 16 (lower: M >= N)");
  if ((lda & 1) || (ldb & 1) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    return gps_fail(h, GPS_ERR_ARG, "gemm_nt: operands must be 16-byte aligned with even leading dimension");
  const bool rowpanel = (C == A);
  if (rowpanel && (N != 128 || lower || triA || nbatch > 1))
    return gps_fail(h, GPS_ERR_ARG, "gemm_nt: in-place form needs N == 128 (no batch, no triangular operand)");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.K = (int)K; g.triA = triA;
  g.batch = (int)nbatch; g.tpb = 0;
  g.a_rs = g.a_cs = g.b_rs = g.b_cs = g.c_rs = g.c_cs = g.a_cm = g.b_cm = g.c_cm = 0;
  if (nbatch > 1) {
    if (lower) return gps_fail(h, GPS_ERR_ARG, "gemm_nt: no batched lower-triangle form");
    g.a_rs = bt->a_rs; g.a_cs = bt->a_cs; g.a_cm = bt->a_cm; g.b_rs = bt->b_rs; g.b_cs = bt->b_cs; g.b_cm = bt->b_cm;
    g.c_rs = bt->c_rs; g.c_cs = bt->c_cs; g.c_cm = bt->c_cm;
    if ((g.a_cs & 1) || (g.b_cs & 1) || (g.a_cm & 1) || (g.b_cm & 1)) return gps_fail(h, GPS_ERR_ARG, "gemm_nt: batch column steps must be even");
  }
  g.stamps = h->gemm_stamps;
  g.sig_ptr = h->next_sig_ptr; g.sig_val = h->next_sig_val; h->next_sig_ptr = nullptr;      // consumed by this launch
  g.cb_tiles = 0; g.cb_stride = 0;
  // lower: the triangle is counted at the granularity of the chosen tile; with a tile < 128 the
  // part of a diagonal 128-block above the diagonal tiles is simply not touched (never read).
  const double t128 = (double)nbatch * (lower ? 0.5 * (double)(N / 128) * (double)(N / 128 + 1) + (double)((M - N) / 128) * (double)(N / 128)
                                             : ((double)M / 128.0) * (double)(N / 128));
  // (triangular operand: sum over the tiles of 2 x their non-zero K range, counted in 128-blocks)
  const double flops = triA == 3 ? (double)nbatch * 128.0 * 128.0 * ((double)M / 128.0) * (double)(N / 128) * (double)(N + 128)
                     : triA      ? (double)nbatch * 128.0 * 128.0 * (double)(N / 128) * (double)(M / 128) * (double)(M + 128)
                                 : 2.0 * t128 * 128.0 * 128.0 * (double)K;
  const double bytes = t128 * (((op == 0 || op == 2) ? 2.0 : 1.0) * 128.0 * 128.0 * 8.0) +
                       8.0 * (double)K * (double)(M + N);   // compulsory traffic: C rmw + each panel once
  LaunchScope ls(h, KC_GEMM, flops, bytes);
  ls.tag[0] = M; ls.tag[1] = N; ls.tag[2] = K; ls.tag[3] = op + 10 * lower + 20 * triA + 100 * (rowpanel ? 1 : 0) + 1000 * (nbatch > 1 ? nbatch : 0);
  const double target = (double)h->gemm_min_tiles;   // workgroups wanted before a larger tile is used
  const int force = h->gemm_force_tb;
  if (rowpanel) {
    // leaves are latency-bound: aim at ~256 workgroups (one per CU), never fewer rows than 16
    int bm = 16;
    if (force) bm = (half_rows && force > 64) ? 64 : force;
    else if (M / 128 >= 256 && !half_rows) bm = 128;
    else if (M / 64 >= 256) bm = 64;
    else if (M / 32 >= 256) bm = 32;
    if (bm == 128) return launch_cfg<128, 128, 2>(h, op, 0, g, M, N);
    if (bm == 64) return launch_cfg<64, 128, 2>(h, op, 0, g, M, N);
    if (bm == 32) return launch_cfg<32, 128, 1>(h, op, 0, g, M, N);
    return launch_cfg<16, 128, 1>(h, op, 0, g, M, N);
  }
  int tb = 32;
  if (force) tb = force < 32 ? 32 : force;
  else if (t128 >= target && !half_rows) tb = 128;
  else if (4.0 * t128 >= target) tb = 64;
  if (half_rows && tb > 64) tb = 64;
  if (tb == 128) return launch_cfg<128, 128, 2>(h, op, lower, g, M, N);
  if (tb == 64) return launch_cfg<64, 64, 2>(h, op, lower, g, M, N);
  return launch_cfg<32, 32, 2>(h, op, lower, g, M, N);
}

// Trailing update of the block-column (multi-GPU) factorisation in one launch:  for every owned column block
// b = 0 .. nblocks-1 (nb columns each, block b starts `stride` rows / columns after block b-1):
//     C[r][b*stride + c] -= sum_k A[r][k] * A[b*stride + c][k]     for rows r >= b*stride, c < nb
// A: [M, K] panel rows from the first owned block on; C points at the same first row / first owned column.
// c_packed: the owned blocks of C lie side by side (block b at column b * nb of C): every rank stores only its own block
// columns (partitioned factor); the rows of A that form the B operand still step by `stride`.
int gps_launch_gemm_nt_cyclic(gps_handle_t h, i64 M, i64 nblocks, i64 nb, i64 stride, i64 K, const double* A, i64 lda,
                              double* C, i64 ldc, int c_packed) {
  if (M <= 0 || nblocks <= 0) return GPS_OK;
  if (M % 128 || nb % 128 || stride % 128 || K % BK_MIN || stride < nb || (nblocks - 1) * stride + nb > M)
    return gps_fail(h, GPS_ERR_ARG, "gemm_nt_cyclic: sizes must be multiples of 128 and the blocks must lie inside the panel");
  GemmArgs g;
  g.A = A; g.B = A; g.C = C; g.lda = lda; g.ldb = lda; g.ldc = ldc; g.K = (int)K; g.triA = 0;
  g.batch = 1; g.tpb = 0; g.pair = 0; g.a_rs = g.a_cs = g.b_rs = g.b_cs = g.c_rs = g.c_cs = g.a_cm = g.b_cm = g.c_cm = 0;
  g.sig_ptr = nullptr; g.sig_val = 0;
  g.stamps = h->gemm_stamps;
  // needed 128x128 tiles: block b uses rows >= b*stride
  double t128 = 0.0;
  for (i64 b = 0; b < nblocks; ++b) { const i64 rows = M - b * stride; if (rows > 0) t128 += (double)(rows / 128) * (double)(nb / 128); }
  const double flops = 2.0 * t128 * 128.0 * 128.0 * (double)K;
  LaunchScope ls(h, KC_GEMM, flops, t128 * 2.0 * 128.0 * 128.0 * 8.0 + 8.0 * (double)K * (double)(M + nblocks * nb));
  const double target = (double)h->gemm_min_tiles;
  g.cb_stride = stride;
  g.cb_cpack = c_packed ? 1 : 0;
  if (t128 >= target) { g.cb_tiles = (int)(nb / 128); return launch_cfg<128, 128, 2>(h, 0, 0, g, M, nblocks * nb); }
  if (4.0 * t128 >= target) { g.cb_tiles = (int)(nb / 64); return launch_cfg<64, 64, 2>(h, 0, 0, g, M, nblocks * nb); }
  g.cb_tiles = (int)(nb / 32);
  return launch_cfg<32, 32, 2>(h, 0, 0, g, M, nblocks * nb);
}
