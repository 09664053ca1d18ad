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
template <int BM, int BN, int WGM, bool LOWER, int OP, int PIPE = 0, int BKT = 16>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(GemmArgs g) {
  constexpr int BK = BKT, LS = BKT + 2;           // (shadow the file-level defaults)
  constexpr int TPR = BKT / 2, RPP = 256 / TPR;   // threads per staged row (16 bytes each), rows per pass
  constexpr int WGN = 4 / WGM;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;        // wave tile
  constexpr int MI = WTM / 16, NI = WTN / 16;          // 16x16 MFMA tiles per wave
  constexpr int LPA = (BM + RPP - 1) / RPP, LPB = (BN + RPP - 1) / RPP;   // 16-byte loads per thread per K-slab
  static_assert(MI >= 1 && NI >= 1, "wave tile must hold at least one MFMA tile");
  // Small tiles split K over KS accumulator sets (k-step kk of a slab goes to set kk % KS), summed in the
  // epilogue.  (Measured: a dependent v_mfma_f64_16x16x4_f64 chain already issues every ~67 cycles, so this
  // is neutral for throughput; it only shortens the dependency chain seen by the scheduler.)
  constexpr int KS = (MI * NI >= 4) ? 1 : (MI * NI == 2 ? 2 : 4);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* As = reinterpret_cast<double*>(smem_raw);       // [2][BM][LS]
  double* Bs = As + 2 * BM * LS;                          // [2][BN][LS]

  // Tail split.  Tiles of one launch are equal, so a launch whose tile count is slightly above a multiple of the
  // resident workgroup slots (512 at 128x128: e.g. the 2080 tiles of a lower-triangular 8192 update) ends with a
  // round in which a few tiles run for a whole tile time while the rest of the GPU idles.  The first
  // nfull = multiple-of-512 tiles run as usual; each of the remaining tiles is cut into nsplit K-slices handled by
  // different workgroups, whose partial sums go to `ws`; the slice that arrives last at the tile's counter adds
  // them in slice order and applies the result to C (deterministic: no floating-point atomics, and the split is
  // a function of the launch shape only).
  int tile_id, slice = -1, tail = 0;
  if ((int)blockIdx.x < g.nfull || g.nsplit <= 1) {
    tile_id = xcd_remap((int)blockIdx.x, g.nsplit > 1 ? g.nfull : g.ntiles);
  } else {
    const int q = (int)blockIdx.x - g.nfull;
    tail = q / g.nsplit;
    slice = q - tail * g.nsplit;
    tile_id = g.nfull + tail;
  }
  int tm, tn;
  i64 bcol;                              // first row of B / first column of C of this tile
  const double* gA = g.A; const double* gB = g.B; double* gC = g.C;
  if (g.batch > 1) {
    const int p = tile_id / g.tpb;
    tile_id -= p * g.tpb;
    gA += (i64)p * g.a_rs * g.lda + (g.a_cm ? ((i64)p * g.a_cs) % g.a_cm : (i64)p * g.a_cs);
    gB += (i64)p * g.b_rs * g.ldb + (g.b_cm ? ((i64)p * g.b_cs) % g.b_cm : (i64)p * g.b_cs);
    gC += (i64)p * g.c_rs * g.ldc + (g.c_cm ? ((i64)p * g.c_cs) % g.c_cm : (i64)p * g.c_cs);
  }
  if (g.cb_tiles > 0) {
    // only the needed tiles are enumerated (so that the XCD chunks carry equal work): owned block ob holds
    // (Tm - ob * q) tile rows x cb_tiles tile columns, q = cb_stride / BM
    const int q = (int)(g.cb_stride / BM);
    int ob = 0, rem = tile_id;
    for (;;) {
      const int cnt = (g.Tm - ob * q) * g.cb_tiles;
      if (rem < cnt) break;
      rem -= cnt; ++ob;
    }
    tm = ob * q + rem / g.cb_tiles;
    const int wi = rem % g.cb_tiles;
    tn = ob * g.cb_tiles + wi;
    bcol = (i64)ob * g.cb_stride + (i64)wi * BN;
  } else {
    decode_tile<LOWER>(tile_id, g.Tm, g.Tn, tm, tn);
    bcol = (i64)tn * BN;
  }
  if (g.sig_ptr && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(g.sig_ptr, g.sig_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (g.stamps && threadIdx.x == 0) {
    long long* st = g.stamps + 6 * (long long)blockIdx.x;
    st[0] = (long long)wall_clock64();
    st[2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
    st[3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave / WGN, wc = wave % WGN;
  const int fr = lane & 15, fk = lane >> 4;

  // global -> register staging map: 8 threads cover one 16-double (128 B) row slab
  const int lrow = tid / TPR;         // 0..RPP-1
  const int lk = (tid % TPR) * 2;     // 0,2,..,BKT-2
  const double* Ag = gA + (i64)(tm * BM + lrow) * g.lda + lk;
  const double* Bg = gB + (bcol + lrow) * g.ldb + lk;
  const bool a_ld = (BM >= RPP) || (lrow < BM);         // BM < RPP: only some of the threads stage A

  v2d ra[LPA], rb[LPB];
  v4d acc[KS][MI][NI];
#pragma unroll
  for (int q = 0; q < KS; ++q)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[q][i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  int nk = g.K / BK;
  int kt0 = g.triA == 1 ? (tm * BM) / BK : 0;
  // (rounded up to whole slabs: a slab deeper than the tile -- 32 x 32 tiles, 64-deep -- ends inside the same diagonal 128-block,
  // whose part beyond the diagonal is stored as zeros)
  if (g.triA == 2) nk = ((tm + 1) * BM + BK - 1) / BK;
  if (g.triA == 3) nk = ((tn + 1) * BN + BK - 1) / BK;
  if (slice >= 0) {                      // slabs [kt0, nk) of this slice: an even split of K / BK, remainder to the first slices
    const int per = nk / g.nsplit, rem = nk - per * g.nsplit;
    kt0 = slice * per + min(slice, rem);
    nk = kt0 + per + (slice < rem ? 1 : 0);
  }

  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < LPA; ++i)
      if (a_ld) ra[i] = *reinterpret_cast<const v2d*>(Ag + (i64)(i * RPP) * g.lda + (i64)kt * BK);
#pragma unroll
    for (int i = 0; i < LPB; ++i)
      rb[i] = *reinterpret_cast<const v2d*>(Bg + (i64)(i * RPP) * g.ldb + (i64)kt * BK);
  };
  auto lstore = [&](int buf) {
    double* a = As + buf * BM * LS + lrow * LS + lk;
    double* b = Bs + buf * BN * LS + lrow * LS + lk;
#pragma unroll
    for (int i = 0; i < LPA; ++i)
      if (a_ld) *reinterpret_cast<v2d*>(a + i * RPP * LS) = ra[i];
#pragma unroll
    for (int i = 0; i < LPB; ++i)
      *reinterpret_cast<v2d*>(b + i * RPP * LS) = rb[i];
  };

  gload(kt0);
  lstore(kt0 & 1);
  __syncthreads();

  if (g.stamps && threadIdx.x == 0) g.stamps[6 * (long long)blockIdx.x + 4] = (long long)wall_clock64();
  if (PIPE == 2) {
    // Scheduled K loop with the slab's last 8 MFMAs deferred across the barrier: they only need registers, so they
    // run while the next slab's first fragments are being read.  Two fragment sets F0 / F1 (one k-step each).
    double fa[2][MI], fb[2][NI];
    auto rd = [&](int set, int buf, int kk) {
      const double* a_base = As + buf * BM * LS + (wr * WTM + fr) * LS + fk + kk * 4;
      const double* b_base = Bs + buf * BN * LS + (wc * WTN + fr) * LS + fk + kk * 4;
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[set][i] = a_base[i * 16 * LS];
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[set][j] = b_base[j * 16 * LS];
    };
    auto mm = [&](int set, int i0, int i1) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
        if (i >= i0 && i < i1) {
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i], fb[set][j], acc[0][i][j], 0, 0, 0);
        }
    };
    int kt = kt0;
    if (kt + 1 < nk) {
      // first slab: nothing deferred yet
      gload(kt + 1);
      rd(0, kt & 1, 0);
      rd(1, kt & 1, 1); mm(0, 0, MI);
      rd(0, kt & 1, 2); mm(1, 0, MI);
      rd(1, kt & 1, 3); mm(0, 0, MI);
      mm(1, 0, MI / 2);
      lstore((kt & 1) ^ 1);
      __syncthreads();
      ++kt;
      for (; kt + 1 < nk; ++kt) {
        const int buf = kt & 1;
        gload(kt + 1);
        rd(0, buf, 0);
        mm(1, MI / 2, MI);               // deferred from the previous slab (F1 still holds its k-step 3)
        rd(1, buf, 1); mm(0, 0, MI);
        rd(0, buf, 2); mm(1, 0, MI);
        rd(1, buf, 3); mm(0, 0, MI);
        mm(1, 0, MI / 2);
        lstore(buf ^ 1);
        // order of the block (counts for a BM x BN tile: NLD 16-byte global loads / LDS writes per thread, MI + NI
        // fragment reads per k-step pair -- the compiler pairs k-steps into ds_read2_b64 --, MI * NI MFMAs per k-step)
        constexpr int NLD = LPA + LPB, NFR = MI + NI, NMM = MI * NI;
        __builtin_amdgcn_sched_group_barrier(0x020, NLD, 0);   // global loads of the next slab
        __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);   // fragments of k-steps 0, 1
        __builtin_amdgcn_sched_group_barrier(0x008, NMM / 2, 0);   // deferred MFMAs of the previous slab
#pragma unroll
        for (int q = 0; q < 3 * NMM / 2; ++q) {                // k-steps 0..2 with the remaining fragment reads
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {                        // first half of k-step 3 + the next slab's LDS writes
          __builtin_amdgcn_sched_group_barrier(0x008, (NMM / 2 + NLD - 1) / NLD, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __syncthreads();
      }
      // last slab (in LDS since the barrier), then everything that is still deferred
      rd(0, kt & 1, 0);
      mm(1, MI / 2, MI);
      rd(1, kt & 1, 1); mm(0, 0, MI);
      rd(0, kt & 1, 2); mm(1, 0, MI);
      rd(1, kt & 1, 3); mm(0, 0, MI);
      mm(1, 0, MI);
    } else {
      rd(0, kt & 1, 0);
      rd(1, kt & 1, 1); mm(0, 0, MI);
      rd(0, kt & 1, 2); mm(1, 0, MI);
      rd(1, kt & 1, 3); mm(0, 0, MI);
      mm(1, 0, MI);
    }
    __syncthreads();
  } else
  for (int kt = kt0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);

    const double* a_base = As + buf * BM * LS + (wr * WTM + fr) * LS + fk;
    const double* b_base = Bs + buf * BN * LS + (wc * WTN + fr) * LS + fk;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = a_base[i * 16 * LS + kk * 4];
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = b_base[j * 16 * LS + kk * 4];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[kk % KS][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[kk % KS][i][j], 0, 0, 0);
    }

    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  if (g.stamps && threadIdx.x == 0) g.stamps[6 * (long long)blockIdx.x + 5] = (long long)wall_clock64();
  // epilogue.  f64 accumulator map (differs from every other dtype on gfx950):
  //   col = lane & 15, row = (lane >> 4) + 4 * reg.
#pragma unroll
  for (int q = 1; q < KS; ++q)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[0][i][j] += acc[q][i][j];
  if (slice >= 0) {
    __shared__ unsigned s_ticket;
    double* mine = g.ws + ((i64)tail * g.nsplit + slice) * (BM * BN) + tid;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) mine[((i * NI + j) * 4 + rg) * 256] = acc[0][i][j][rg];
    __threadfence();
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(&g.cnt[tail], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != (unsigned)(g.nsplit - 1)) {
      if (g.stamps && tid == 0) g.stamps[6 * (long long)blockIdx.x + 1] = (long long)wall_clock64();
      return;
    }
    __threadfence();
    // slice order outermost: all MI*NI*4 loads of one slice are in flight together (element by element, each sum
    // was a chain of nsplit dependent loads: 365 us for 16 slices); per element the order of the additions is the same
    const double* all = g.ws + (i64)tail * g.nsplit * (BM * BN) + tid;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[0][i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int sl = 0; sl < g.nsplit; ++sl) {
      const double* src = all + (i64)sl * (BM * BN);
      double v[MI][NI][4];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) v[i][j][rg] = __builtin_nontemporal_load(src + ((i * NI + j) * 4 + rg) * 256);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) acc[0][i][j][rg] += v[i][j][rg];
    }
    if (tid == 0) g.cnt[tail] = 0;       // ready for the next launch on this stream
  }
  const i64 row0 = (i64)tm * BM + wr * WTM + (lane >> 4);
  const i64 col0 = (g.cb_cpack ? (i64)tn * BN : bcol) + wc * WTN + (lane & 15);
  // C read-modify-write in batches: all loads of a batch are issued before the first store.  (Written as
  // `*cp = *cp - acc` per element the compiler has to assume that a store may alias the next load -- ldc is a
  // run-time value -- and emits load / wait / store one element at a time: 64 memory round trips per thread,
  // 96 us per tile when every workgroup of a round does it at the same moment.)
  constexpr int IB = (MI >= 2) ? 2 : 1;                 // tile rows of 16 per batch
#pragma unroll
  for (int i0 = 0; i0 < MI; i0 += IB) {
    double cv[IB][NI][4];
    if (OP == 0 || OP == 2) {
#pragma unroll
      for (int ii = 0; ii < IB; ++ii)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
            cv[ii][j][rg] = gC[(row0 + (i0 + ii) * 16 + 4 * rg) * g.ldc + col0 + j * 16];
    }
#pragma unroll
    for (int ii = 0; ii < IB; ++ii)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          double* cp = gC + (row0 + (i0 + ii) * 16 + 4 * rg) * g.ldc + col0 + j * 16;
          const double av = acc[0][i0 + ii][j][rg];
          if (OP == 0) *cp = cv[ii][j][rg] - av;
          else if (OP == 2) *cp = cv[ii][j][rg] + av;
          else if (OP == 3) *cp = -av;
          else *cp = av;
        }
  }
  if (g.stamps && threadIdx.x == 0) g.stamps[6 * (long long)blockIdx.x + 1] = (long long)wall_clock64();
}

template <int BM, int BN, int WGM, bool LOWER, int OP, int PIPE = 0, int BKT = 16>
static int launch_variant(gps_handle_t h, const GemmArgs& g) {
  const size_t lds = (size_t)(2 * (BM + BN) * (BKT + 2)) * sizeof(double);
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
  if (M % 128 || N % 128 || K % BK_MIN || (lower && M < N))
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
                                             : (double)(M / 128) * (double)(N / 128));
  // (triangular operand: sum over the tiles of 2 x their non-zero K range, counted in 128-blocks)
  const double flops = triA == 3 ? (double)nbatch * 128.0 * 128.0 * (double)(M / 128) * (double)(N / 128) * (double)(N + 128)
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
    if (force) bm = force;
    else if (M / 128 >= 256) bm = 128;
    else if (M / 64 >= 256) bm = 64;
    else if (M / 32 >= 256) bm = 32;
    if (bm == 128) return launch_cfg<128, 128, 2>(h, op, 0, g, M, N);
    if (bm == 64) return launch_cfg<64, 128, 2>(h, op, 0, g, M, N);
    if (bm == 32) return launch_cfg<32, 128, 1>(h, op, 0, g, M, N);
    return launch_cfg<16, 128, 1>(h, op, 0, g, M, N);
  }
  int tb = 32;
  if (force) tb = force < 32 ? 32 : force;
  else if (t128 >= target) tb = 128;
  else if (4.0 * t128 >= target) tb = 64;
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
  g.batch = 1; g.tpb = 0; g.a_rs = g.a_cs = g.b_rs = g.b_cs = g.c_rs = g.c_cs = g.a_cm = g.b_cm = g.c_cm = 0;
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
