// Exact-GP factorisation of a SMALL covariance (up to SN_MAXB = 32 blocks of 128 padded rows; the entry points use it up to
// "small_n_max" = 2048) as ONE launch.
//
// The reference's own workload is this size (examples/gpr.py:36,48-61: N ~ 455, 20 000 optimiser steps; each step
// tf.cholesky + tf.matrix_triangular_solve, models/gpr.py:70, densities.py:82).  Launch by launch (blocked.hpp) such a
// factorisation is a chain of potrf_base / panel solve / update launches, each at the launch-latency floor: 14 launches at
// N = 512.  Here the whole of
//     L = chol(K),  inv of the diagonal blocks,  alpha^T = (Y - m)^T L^-T  (augmented rows),  sum log L_ii,  sum alpha^2
// runs in one persistent launch of 1 + P tasks, synchronised by monotone counters in HBM.  Every workgroup DRAWS its task from
// one counter (sn_draw) -- nothing is derived from blockIdx, so nothing is assumed about which workgroups are resident:
//
//   task 0, the CHAIN: for every 128-column block j: wait until the updates of the panels before it have reached the
//     diagonal block, factor + invert it (potrf_base_body: the kernel of potrf_base.hip as a device function), publish F[j].
//   task 1 + p, a PAIR (row slab s of SH rows, block column k): owns the SH x 128 piece of the matrix at (s, k) for the
//     whole launch -- applies panel j = 0 .. k-1 to it as soon as the slabs it needs are solved (U: C -= X_s,j X_k,j^T),
//     then, once F[k] is there, solves it against the block's inverse (S: X <- X W_k^T) and publishes that.  The pieces
//     of a diagonal block only take the updates (and count them for the chain); the augmented rows (Y - m)^T are one more
//     block row of slabs that is never factored.  Every product is a [SH x 128] x [128 x 128]^T MFMA product with both
//     operands staged through LDS.  Order of the pairs: column by column, the diagonal pairs of a column first (sn_decode) --
//     in that order every task waits only for tasks drawn before it.
//
// Hand-overs (cdna_hip_programming.md, Guideline 16): producer -- every storing wave drains its stores, workgroup barrier, one
// lane's agent-scope release fence, relaxed agent-scope add on the counter; consumer -- one lane polls (relaxed, s_sleep,
// bounded by the wall clock), agent-scope acquire, barrier, plain loads.  A wait that runs out sets the abort word: every
// workgroup leaves, the host falls back to the launch-by-launch path (and counts it).  The counters are zeroed at the very end
// by the chain (as many workgroups as tasks) or by the workgroup that leaves last (queued form), so an evaluation costs no memset.
#include <climits>
#include <vector>
#define GPS_PB_DEVICE_ONLY
#define GPS_PB_WT 1            // the chain's results leave with write-through stores
#include "potrf_base.hip"

typedef unsigned int u32;

// sync words (u32), each counter on a 64-byte line of its own
#define SN_LINE 16
#define SN_MAXB 32                                           // block rows / columns the counters are laid out for (4096 padded points)
#define SN_ABORT 0                                           // (first: sn_wait looks for it at this offset of whichever region it is given)
#define SN_ADONE (1 * SN_LINE)                               // solves of the augmented rows
#define SN_QUEUE (2 * SN_LINE)                               // next task (queued form)
#define SN_DONE (3 * SN_LINE)                                // workgroups that have left (queued form)
#define SN_F(j) ((5 + (j)) * SN_LINE)                        // [32]  block j factored and inverted
#define SN_D(j) ((5 + SN_MAXB + (j)) * SN_LINE)              // [32]  update pieces applied to diagonal block j
#define SN_XB(i, j) ((5 + 2 * SN_MAXB + (i) * SN_MAXB + (j)) * SN_LINE)   // [33][32] slabs of block row i (nblk = the augmented rows) solved against panel j
#define SN_WORDS ((5 + 2 * SN_MAXB + (SN_MAXB + 1) * SN_MAXB) * SN_LINE)
#define SN_USED_WORDS(nblk) ((5 + 2 * SN_MAXB + ((nblk) + 1) * SN_MAXB) * SN_LINE)   // (the augmented rows are block row nblk)
#define SN_INV_WORDS ((3 + 8 * 16 + 16 * 16) * SN_LINE)      // the inverse launch's counters (SI_*, below)

struct SmallArgs {
  double* K; i64 ld;             // [np + aug rows][ld]: K + noise (lower, identity padded), then the augmented rows
  double* Linv; double* LinvT;   // [nblk][128 * 128] each (LinvT may be null)
  const double* resid;           // [n][r] on the device
  int n, r, nblk;
  int* info;
  u32* sync;                     // SN_WORDS words, zero on entry
  double* res;                   // [0] sum log L_ii  [1] sum alpha^2  [2] info  [3] abort
  double* alpha; i64 ld_alpha; int alpha_rows;   // optional second home of alpha^T: [alpha_rows][ld_alpha]
  SmallKgen kg;                  // kg.on: K is not in memory yet -- every pair generates its tile first (and block row 0 gets pairs for that)
  long long* stamps;             // diagnostics (GPS_SMALL_STAMPS): 100 MHz wall-clock stamps of the chain [0..63] and of the pairs of the first slab of each block row
};

__device__ __forceinline__ u32 sn_load(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The handed-off data was stored write-through (sc1) by whoever stored it: every storing wave drains its stores, the workgroup
// meets, one lane adds to the counter -- no release fence (Guideline 16, R1).
__device__ __forceinline__ void sn_publish(u32* counter, u32* counter2) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (counter2) __hip_atomic_fetch_add(counter2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void sn_store(double* p, double v) {          // 8-byte write-through store
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// wait until *c1 >= t1 and (c2 ? *c2 >= t2 : true); false: aborted (this wait ran out, or another one did)
__device__ __forceinline__ bool sn_wait(u32* sync, const u32* c1, u32 t1, const u32* c2, u32 t2, int* s_flag) {
  if (threadIdx.x == 0) {
    int ok = 1;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
      if (sn_load(c1) >= t1 && (!c2 || sn_load(c2) >= t2)) break;
      if (sn_load(sync + SN_ABORT) != 0u) { ok = 0; break; }
      if (wall_clock64() - t0 > 200000000ull) {                    // 2 s at 100 MHz: a scheduling surprise is an error, not a hang
        __hip_atomic_store(sync + SN_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0; break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *s_flag = ok;
  }
  __syncthreads();
  const int ok = *s_flag;
  __syncthreads();
  return ok != 0;
}

// [SH x 128] x [128 x 128]^T product of one pair, wave w = column tile w (columns 16 w .. 16 w + 15):
//   UPDATE: C[SH][128 @ ldc] -= A[SH][K = 128 @ lda] * Bm[128][K @ ldb]^T
//   SOLVE : C <- A * Bm^T in place (A == C), Bm lower triangular: k-steps beyond the tile's last column are skipped
// Both operands go through LDS: whole rows by coalesced 16-byte loads (a wave stages the 16 rows of Bm its column tile needs
// into a region of its own, the workgroup the SH rows of A together), then the MFMA fragments by conflict-free ds_read_b64
// (row stride 130 doubles).  Fetching the fragments straight from L2 (8 bytes per lane, 32-byte runs) took 7.5 us per
// product (stamps: profiles/r04_small_n_stamps.txt) -- the operands have just been written by other CUs and come from HBM.
// What the pair OWNS is fetched before it waits for anybody: the C tile (UPDATE) or its rows of A (SOLVE) -- sn_own.
#define SN_LS 130                                     // LDS row stride (doubles)
#define SN_LDS_BYTES ((8 * 16 * SN_LS + 16 * SN_LS) * 8)   // 8 wave regions of Bm rows + the 16 rows of A (150 KB: slabs of 32 rows would not fit)

template <int SH>
__device__ __forceinline__ void sn_stage_rows(double* lds, const double* g, i64 ldg, int row_lo, int row_hi, int lane) {
  // rows [row_lo, row_hi) of a [.. x 128] global block -> lds[row][SN_LS], one 1-KB row per wave instruction
  double2 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) if (row_lo + u < row_hi) v[u] = *reinterpret_cast<const double2*>(g + (i64)(row_lo + u) * ldg + 2 * lane);
#pragma unroll
  for (int u = 0; u < 16; ++u) if (row_lo + u < row_hi) *reinterpret_cast<double2*>(lds + (row_lo + u) * SN_LS + 2 * lane) = v[u];
}

template <int SH, bool SOLVE>
struct SnOwn { v4d acc[SH / 16]; };

// before the wait: the accumulators (UPDATE: the C tile; SOLVE: zero) and, for SOLVE, this pair's rows of A into LDS
template <int SH, bool SOLVE>
__device__ __forceinline__ void sn_own(SnOwn<SH, SOLVE>& o, char* smem, const double* A, i64 lda, const double* C, i64 ldc,
                                       int wave, int lane, int fr, int fk) {
  constexpr int RT = SH / 16;
  double* As = reinterpret_cast<double*>(smem) + 8 * 16 * SN_LS;
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) o.acc[t][rg] = SOLVE ? 0.0 : C[(i64)(16 * t + fk + 4 * rg) * ldc + 16 * wave + fr];
  if (SOLVE) sn_stage_rows<SH>(As, A, lda, wave * (SH / 8), (wave + 1) * (SH / 8), lane);
}

// the product's two halves: sn_mma -- operands through LDS, the MFMA chain into o.acc -- and sn_store (below); sn_product = both
template <int SH, bool SOLVE>
__device__ __forceinline__ void sn_mma(SnOwn<SH, SOLVE>& o, char* smem, const double* A, i64 lda, const double* __restrict__ Bm,
                                       i64 ldb, int wave, int lane, int fr, int fk, long long* st = nullptr) {
  constexpr int RT = SH / 16;
  const int ks = SOLVE ? 4 * (wave + 1) : 32;                    // k-steps of 4
  double* Bs = reinterpret_cast<double*>(smem) + wave * 16 * SN_LS;
  double* As = reinterpret_cast<double*>(smem) + 8 * 16 * SN_LS;
  sn_stage_rows<SH>(Bs - 16 * wave * SN_LS, Bm - 0, ldb, 16 * wave, 16 * wave + 16, lane);      // rows 16 w .. of Bm -> this wave's region (indexed by absolute row)
  if (!SOLVE) sn_stage_rows<SH>(As, A, lda, wave * (SH / 8), (wave + 1) * (SH / 8), lane);
  __syncthreads();                                               // A rows of every wave are in LDS (and, SOLVE: everybody has read the rows about to be overwritten)
  if (st && threadIdx.x == 0) st[0] = (long long)wall_clock64();
  const double* pb = Bs + fr * SN_LS + fk;
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    if (s < ks) {
      const double b = pb[4 * s];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const double a = As[(16 * t + fr) * SN_LS + 4 * s + fk];
        o.acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(SOLVE ? a : -a, b, o.acc[t], 0, 0, 0);
      }
    }
  }
  if (st && threadIdx.x == 0) st[1] = (long long)wall_clock64();
}

// The tile leaves as whole 128-byte row segments: transposed through this wave's LDS region (its fragments have been read),
// lane l then stores columns 2 (l & 7), +1 of row l >> 3 (and of row 8 + (l >> 3)) with one 16-byte write-through store --
// full lines for the consumers on other CUs, instead of 8-byte stores scattered over four rows.
template <int SH, bool SOLVE>
__device__ __forceinline__ void sn_store_tile(SnOwn<SH, SOLVE>& o, char* smem, double* C, i64 ldc, double* C2, i64 ldc2, int c2_rows,
                                              int wave, int lane, int fr, int fk) {
  constexpr int RT = SH / 16;
  double* Bs = reinterpret_cast<double*>(smem) + wave * 16 * SN_LS;
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) Bs[(fk + 4 * rg) * SN_LS + fr] = o.acc[t][rg];
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int row = 8 * hlf + (lane >> 3), col = 2 * (lane & 7);
      const double2 v = *reinterpret_cast<const double2*>(Bs + row * SN_LS + col);
      pb_store16(&C[(i64)(16 * t + row) * ldc + 16 * wave + col], v);
      if (C2 && 16 * t + row < c2_rows) *reinterpret_cast<double2*>(&C2[(i64)(16 * t + row) * ldc2 + 16 * wave + col]) = v;   // (alpha^T also where the other entry points expect it)
    }
  }
  __syncthreads();                                               // (the LDS regions are free again)
}

template <int SH, bool SOLVE>
__device__ __forceinline__ void sn_product(SnOwn<SH, SOLVE>& o, char* smem, const double* A, i64 lda, const double* __restrict__ Bm,
                                           i64 ldb, double* C, i64 ldc, double* C2, i64 ldc2, int c2_rows, int wave, int lane,
                                           int fr, int fk, long long* st = nullptr) {
  sn_mma<SH, SOLVE>(o, smem, A, lda, Bm, ldb, wave, lane, fr, fk, st);
  sn_store_tile<SH, SOLVE>(o, smem, C, ldc, C2, ldc2, c2_rows, wave, lane, fr, fk);
}

// One pair = one 16-row slab of block row bi (bi == nblk: the augmented rows) x block column k: generate (K inside the launch), update
// with the columns j < k as they appear, solve against block k.
template <int SH>
__device__ __forceinline__ void sn_pair_task(const SmallArgs& g, char* smem_raw, int* s_flag_p, int bi, int sl, int k) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int nblk = g.nblk, spb = 128 / SH, aug_slabs = (g.r + SH - 1) / SH;
  const i64 ld = g.ld, np = (i64)nblk * 128;
  u32* sync = g.sync;
  const bool aug = bi == nblk;
  const i64 r0 = aug ? np + (i64)sl * SH : (i64)bi * 128 + (i64)sl * SH;
  double* C = g.K + r0 * ld + (i64)k * 128;
  if (aug) {
    // this pair's piece of the augmented rows: (Y - m)^T, zero beyond the r outputs / n points
    for (int idx = tid; idx < SH * 128; idx += NT) {
      const int q = idx >> 7, c = idx & 127;
      const i64 pt = (i64)k * 128 + c, out = (i64)sl * SH + q;
      C[(i64)q * ld + c] = (out < g.r && pt < g.n) ? g.resid[pt * g.r + out] : 0.0;
    }
    __syncthreads();
  }
  if (!aug && g.kg.on) {
    // this pair's tile of K + noise I (identity padded), one stationary primitive with r2 = sum_d ((x_id - x_jd) / l_d)^2 (exactly 0 on
    // the diagonal); thread: one column, four rows.  Block (0, 0) goes to the chain on another CU: write-through stores.
    const SmallKgen& kg = g.kg;
    const int c = tid & 127;
    const i64 pj = (i64)k * 128 + c;
    double xj[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) xj[d] = (d < kg.nd && pj < g.n) ? kg.X[pj * kg.d_all + kg.dims[d]] * kg.inv_ls[d] : 0.0;
    for (int q = 0; q < SH * 128 / NT; ++q) {
      const int row = (tid >> 7) + (NT >> 7) * q;
      const i64 pi = r0 + row;
      double v;
      if (pi >= g.n || pj >= g.n) v = (pi == pj) ? 1.0 : 0.0;
      else {
        double r2 = 0.0;
#pragma unroll
        for (int d = 0; d < 16; ++d)
          if (d < kg.nd) { const double dl = kg.X[pi * kg.d_all + kg.dims[d]] * kg.inv_ls[d] - xj[d]; r2 = fma(dl, dl, r2); }
        // kernels.py:436-439 (RBF), 557-610 (Matern family, Exponential: r = sqrt(r2 + 1e-12), also on the diagonal)
        const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
        if (pi == pj) r2 *= 0.0;                   // (exactly 0 on the diagonal; NaN stays NaN)
        if (kg.op == GPS_K_RBF) v = kg.variance * exp(-0.5 * r2);
        else {
          const double rr = sqrt(r2 + 1e-12);
          if (kg.op == GPS_K_MATERN12) v = kg.variance * exp(-rr);
          else if (kg.op == GPS_K_EXPONENTIAL) v = kg.variance * exp(-0.5 * rr);
          else if (kg.op == GPS_K_MATERN32) v = kg.variance * (1.0 + sq3 * rr) * exp(-sq3 * rr);
          else v = kg.variance * (1.0 + sq5 * rr + 5.0 / 3.0 * (rr * rr)) * exp(-sq5 * rr);
        }
        if (pi == pj) v += kg.noise;
      }
      sn_store(&C[(i64)row * ld + c], v);
    }
    if (bi == 0) { sn_publish(sync + SN_D(0), nullptr); return; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const u32 my_slabs = (u32)(aug ? aug_slabs : spb);
  long long* ps = (g.stamps && sl == 0 && nblk <= 8) ? g.stamps + 64 + ((aug ? 8 : bi) * 8 + k) * 24 : nullptr;
#define SN_PSTAMP(q) do { if (ps && tid == 0) ps[q] = (long long)wall_clock64(); } while (0)
  SN_PSTAMP(0);
  for (int j = 0; j < k; ++j) {
    SnOwn<SH, false> own;
    sn_own<SH, false>(own, smem_raw, nullptr, 0, C, ld, wave, lane, fr, fk);
    if (!sn_wait(sync, sync + SN_XB(bi, j), my_slabs, sync + SN_XB(k, j), (u32)spb, s_flag_p)) return;
    SN_PSTAMP(1 + 2 * j);
    sn_product<SH, false>(own, smem_raw, g.K + r0 * ld + (i64)j * 128, ld, g.K + (i64)k * 128 * ld + (i64)j * 128, ld, C, ld, nullptr, 0, 0,
                          wave, lane, fr, fk, (ps && j == 0) ? ps + 12 : nullptr);
    if (ps && j == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (tid == 0) ps[14] = (long long)wall_clock64(); }
    if (bi == k) sn_publish(sync + SN_D(k), nullptr);
    SN_PSTAMP(2 + 2 * j);
  }
  if (bi != k) {
    SnOwn<SH, true> own;
    sn_own<SH, true>(own, smem_raw, C, ld, C, ld, wave, lane, fr, fk);
    if (!sn_wait(sync, sync + SN_F(k), 1u, nullptr, 0u, s_flag_p)) return;
    SN_PSTAMP(20);
    double* mirror = (aug && g.alpha) ? g.alpha + (i64)sl * SH * g.ld_alpha + (i64)k * 128 : nullptr;      // (that buffer has r rows only)
    sn_product<SH, true>(own, smem_raw, C, ld, g.Linv + (i64)k * 128 * 128, 128, C, ld, mirror, g.ld_alpha, g.alpha_rows - sl * SH,
                         wave, lane, fr, fk, ps ? ps + 16 : nullptr);
    SN_PSTAMP(21);
    sn_publish(sync + SN_XB(bi, k), aug ? sync + SN_ADONE : nullptr);
    SN_PSTAMP(22);
  }
}

// Draw the next task of the launch: one lane takes a ticket and reads the abort word, everybody gets both through LDS (so that
// all waves of the workgroup take the same branch whatever another workgroup does to the abort word in between).
__device__ __forceinline__ int sn_draw(u32* sync, int queue_word, int* s_flag_p, bool* aborted) {
  __syncthreads();                                                 // (the previous task's use of the LDS words is over)
  if (threadIdx.x == 0) {
    s_flag_p[1] = (int)__hip_atomic_fetch_add(sync + queue_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_flag_p[3] = (sn_load(sync + SN_ABORT) != 0u) ? 1 : 0;
  }
  __syncthreads();
  *aborted = s_flag_p[3] != 0;
  return s_flag_p[1];
}

// The CHAIN: block after block -- wait for its updates, factor + invert, publish --, then the two sums of the likelihood.
__device__ __forceinline__ bool sn_chain(const SmallArgs& g, char* smem_raw, int* s_flag_p, double* s_red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = g.nblk, spb = 128 / 16, aug_slabs = (g.r + 15) / 16;
  const i64 ld = g.ld, np = (i64)nblk * 128;
  u32* sync = g.sync;
  if (tid == 0) *g.info = 0x7fffffff;
  double slog = 0.0;
  bool ok = true;
#define SN_STAMP(q) do { if (g.stamps && tid == 0) g.stamps[q] = (long long)wall_clock64(); } while (0)
  SN_STAMP(0);
  for (int j = 0; j < nblk && ok; ++j) {
    // (with the kernel matrix generated in this launch block 0 is stored by its eight pairs first)
    if (j > 0 || g.kg.on) ok = sn_wait(sync, sync + SN_D(j), (u32)(j > 0 ? spb * j : spb), nullptr, 0u, s_flag_p);
    if (!ok) break;
    SN_STAMP(1 + 3 * j);
    potrf_base_body(smem_raw, g.K + (i64)j * 128 * ld + (i64)j * 128, ld, g.Linv + (i64)j * 128 * 128,
                    g.LinvT ? g.LinvT + (i64)j * 128 * 128 : nullptr, g.info, j * 128, 1, nullptr);
    SN_STAMP(2 + 3 * j);
    sn_publish(sync + SN_F(j), nullptr);
    SN_STAMP(3 + 3 * j);
    // log of the diagonal from the image the body leaves in LDS (identity padding: log 1 = 0)
    if (tid < PB) { const double* a = reinterpret_cast<const double*>(smem_raw); slog += log(a[tid * PS + tid]); }
  }
  // sum alpha^2 once every slab of the augmented rows has been solved against every block
  if (ok) ok = sn_wait(sync, sync + SN_ADONE, (u32)(aug_slabs * nblk), nullptr, 0u, s_flag_p);
  SN_STAMP(30);
  if (!ok) return false;
  double ssq = 0.0;
  for (i64 idx = tid; idx < (i64)g.r * np; idx += NT) {
    const i64 q = idx / np, i = idx - q * np;
    const double v = g.K[(np + q) * ld + i];
    ssq = fma(v, v, ssq);
  }
  // fixed-order reductions: lanes by shuffle, waves through LDS
  for (int which = 0; which < 2; ++which) {
    double v = which ? ssq : slog;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < NT / 64; ++w) t += s_red[w]; g.res[which] = t; }
  }
  if (tid == 0) g.res[2] = (double)*g.info;
  return true;
}

// Task p >= 1 of the launch -> (block row bi, slab sl, block column k): column by column, the DIAGONAL pairs of a column first
// (without the in-launch kernel matrix block (0, 0) has none), then the slabs of the rows below, then the augmented rows.
__device__ __forceinline__ bool sn_decode(int p, int nblk, int spb, int aug_slabs, bool kgen, int* bi, int* sl, int* k) {
  p -= 1;
  for (int c = 0; c < nblk; ++c) {
    const int nd = (c == 0 && !kgen) ? 0 : spb;
    const int nrow = spb * (nblk - 1 - c);
    const int cnt = nd + nrow + aug_slabs;
    if (p < cnt) {
      *k = c;
      if (p < nd) { *bi = c; *sl = p; }
      else if (p - nd < nrow) { *bi = c + 1 + (p - nd) / spb; *sl = (p - nd) % spb; }
      else { *bi = nblk; *sl = p - nd - nrow; }
      return true;
    }
    p -= cnt;
  }
  return false;                                            // the queue is empty
}

// Every workgroup DRAWS its task (a ticket from one counter) instead of deriving it from blockIdx: task 0 is the chain, the
// others are the pairs in sn_decode's order.  In that order a pair waits only for pairs drawn before it and for the chain's
// block k, and the chain waits only for the diagonal pairs of column k, which are drawn before every pair that waits for block
// k -- and whoever has drawn a task is running.  So whatever part of the launch is resident makes progress: no assumption about
// which workgroups of a launch become resident first, or together (round 4 gave the chain to workgroup 0 and the pairs to
// workgroup 1 + p, block row by block row: with several such launches of one process in flight a launch whose workgroup 0 was
// not resident stalled for the full bounded wait, docs/LAB_NOTES.md).
//   QUEUED = false: as many workgroups as tasks (up to seven blocks: one per CU), one draw each; the chain, which finishes
//     last by construction, reports and leaves the counters zero.
//   QUEUED = true : more tasks than CUs -- a workgroup keeps drawing until the queue is empty (the chain is run straight-line in
//     front of the loop: inside it the compiler spilled 44 VGPRs and 459 SGPRs); the workgroup that leaves last reports and zeroes.
template <int SH, bool QUEUED>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void small_factor_kernel(SmallArgs g) {
  // (no static __shared__: it would sit in front of the dynamic region and push the image off its 16-byte alignment)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int LDS_MAIN = (PB_LDS_BYTES > SN_LDS_BYTES) ? PB_LDS_BYTES : SN_LDS_BYTES;       // the chain's image / a pair's operand regions
  int* const s_flag_p = reinterpret_cast<int*>(smem_raw + LDS_MAIN);        // [0] sn_wait  [1] ticket  [2] last to leave  [3] aborted
  double* const s_red = reinterpret_cast<double*>(smem_raw + LDS_MAIN + 16);
  const int tid = threadIdx.x;
  const int nblk = g.nblk, spb = 128 / SH, aug_slabs = (g.r + SH - 1) / SH;
  u32* sync = g.sync;
  const bool kgen = g.kg.on != 0;
  bool aborted = false;
  int p = sn_draw(sync, SN_QUEUE, s_flag_p, &aborted);
  if (p == 0) {
    const bool ok = !aborted && sn_chain(g, smem_raw, s_flag_p, s_red);
    if (!QUEUED) {
      if (tid == 0) g.res[3] = ok ? (double)sn_load(sync + SN_ABORT) : 1.0;
      // nobody reads the counters any more (every pair is needed by something the chain has waited for, so every ticket has
      // been drawn and the last waits above were the last of the launch): leave them zero for the next call
      __syncthreads();
      if (ok) for (int i = tid; i < SN_USED_WORDS(nblk); i += NT) sync[i] = 0u;
      if (g.stamps && tid == 0) g.stamps[31] = (long long)wall_clock64();
      return;
    }
    if (ok) p = sn_draw(sync, SN_QUEUE, s_flag_p, &aborted); else aborted = true;
  }
  while (!aborted) {
    int bi = 0, sl = 0, k = 0;
    if (!sn_decode(p, nblk, spb, aug_slabs, kgen, &bi, &sl, &k)) break;
    sn_pair_task<SH>(g, smem_raw, s_flag_p, bi, sl, k);
    if (!QUEUED) return;
    p = sn_draw(sync, SN_QUEUE, s_flag_p, &aborted);
  }
  if (!QUEUED) return;
  // ---- the workgroup that leaves last reports and leaves the counters zero for the next call (not the chain: others may still
  // be drawing from the queue when it has finished)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const u32 done = __hip_atomic_fetch_add(sync + SN_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int last = 0;
    if (done + 1 == gridDim.x) {
      const u32 ab = sn_load(sync + SN_ABORT);
      g.res[3] = (double)ab;
      last = (ab == 0u) ? 1 : 0;
    }
    s_flag_p[2] = last;
  }
  __syncthreads();
  if (s_flag_p[2]) for (int w = tid; w < SN_USED_WORDS(nblk); w += NT) sync[w] = 0u;
}

// K (lower, + noise, identity padded) is in dK [np + 128][np]; resid [n][r] on the device.  On success dK holds L and the
// augmented rows alpha^T, linv / linvT the block inverses, res4 (host) = {sum log L_ii, sum alpha^2, info, abort}.
// GPS_ERR_UNSUPPORTED: not a shape for this path (the caller takes the launch-by-launch one).
int gps_launch_small_factor(gps_handle_t h, double* dK, i64 np, double* linv, double* linvT, const double* d_resid, i64 n, i64 r,
                            int* d_info, double* d_res4, double* d_alpha, i64 ld_alpha, i64 alpha_rows, const SmallKgen* kgen) {
  // (up to 512 padded rows in slabs of 16: 93 workgroups at most, several such launches fit the GPU side by side; up to 896: 232,
  // the whole GPU -- the workgroup count is checked against the CU count below)
  if (np % 128 || np < 128 || np > 128 * SN_MAXB || r < 1 || r > 16) return GPS_ERR_UNSUPPORTED;
  if (h->prop.multiProcessorCount < 160) return GPS_ERR_UNSUPPORTED;           // every workgroup must be resident (one per CU)
  const int nblk = (int)(np / 128);
  const int SH = 16;
  const int spb = 128 / SH, aug_slabs = (int)((r + SH - 1) / SH);
  int pairs = 0;
  const bool kg_on = kgen && kgen->on;
  for (int i = kg_on ? 0 : 1; i < nblk; ++i) pairs += spb * (i + 1);
  pairs += aug_slabs * nblk;
  // as many workgroups as pairs (+ the chain) while they all fit, one per CU (LDS) -- up to seven blocks --; above, every
  // piece of the work is a task of one queue
  const bool queued = 1 + pairs > h->prop.multiProcessorCount;
  const int ntasks = 1 + pairs;                        // the chain + the pairs
  const int slots = h->prop.multiProcessorCount - 8;
  const int grid = queued ? (ntasks < slots ? ntasks : slots) : 1 + pairs;
  if (!h->dSmallSync.p) {
    GPS_HIP(h, h->dSmallSync.ensure((size_t)(SN_WORDS + SN_INV_WORDS) * 4));    // behind the first SN_WORDS: the inverse launch (below)
    GPS_HIP(h, hipMemsetAsync(h->dSmallSync.p, 0, (size_t)(SN_WORDS + SN_INV_WORDS) * 4, h->stream));
  }
  SmallArgs a;
  a.K = dK; a.ld = np; a.Linv = linv; a.LinvT = linvT; a.resid = d_resid; a.n = (int)n; a.r = (int)r; a.nblk = nblk;
  a.info = d_info; a.sync = (u32*)h->dSmallSync.p; a.res = d_res4;
  a.alpha = d_alpha; a.ld_alpha = ld_alpha; a.alpha_rows = (int)alpha_rows;
  if (kg_on) a.kg = *kgen; else a.kg.on = 0;
  a.stamps = nullptr;
  static const bool want_stamps = getenv("GPS_SMALL_STAMPS") != nullptr;
  const size_t stamp_words = 64 + 9 * 8 * 24;
  if (want_stamps && nblk <= 8) {
    GPS_HIP(h, h->dTmp3.ensure(stamp_words * 8));
    GPS_HIP(h, hipMemsetAsync(h->dTmp3.p, 0, stamp_words * 8, h->stream));
    a.stamps = (long long*)h->dTmp3.p;
  }
  // diagnostics ("small_fault_inject" = k): the k-th cooperative launch from now starts with its abort word set, as if one of
  // its bounded waits had run out -- the evaluation must come back right through the launch-by-launch path
  if (h->small_fault_inject > 0 && --h->small_fault_inject == 0) {
    const u32 one = 1u;
    GPS_HIP(h, hipMemcpyAsync((u32*)h->dSmallSync.p + SN_ABORT, &one, 4, hipMemcpyHostToDevice, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
  }
  const size_t lds = (size_t)((PB_LDS_BYTES > SN_LDS_BYTES) ? PB_LDS_BYTES : SN_LDS_BYTES) + 128;
  int rc0 = gps_dyn_lds(h, reinterpret_cast<const void*>(&small_factor_kernel<16, false>), (int)lds);
  if (!rc0) rc0 = gps_dyn_lds(h, reinterpret_cast<const void*>(&small_factor_kernel<16, true>), (int)lds);
  if (rc0) return rc0;
  LaunchScope ls(h, KC_POTRF_BASE, (double)np * np * np / 3.0, 8.0 * np * np);
  if (queued) hipLaunchKernelGGL((small_factor_kernel<16, true>), dim3(grid), dim3(NT), lds, h->stream, a);
  else hipLaunchKernelGGL((small_factor_kernel<16, false>), dim3(grid), dim3(NT), lds, h->stream, a);
  GPS_HIP(h, hipGetLastError());
  if (want_stamps && nblk <= 8) {
    std::vector<long long> st(stamp_words);
    GPS_HIP(h, hipMemcpyAsync(st.data(), h->dTmp3.p, stamp_words * 8, hipMemcpyDeviceToHost, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
    const long long t0 = st[0];
    fprintf(stderr, "small_n chain (us since start):");
    for (int j = 0; j < nblk; ++j) fprintf(stderr, "  blk%d: D ready %.1f factored %.1f published %.1f |", j, (st[1 + 3 * j] - t0) * 0.01, (st[2 + 3 * j] - t0) * 0.01, (st[3 + 3 * j] - t0) * 0.01);
    fprintf(stderr, "  alpha done %.1f end %.1f\n", (st[30] - t0) * 0.01, (st[31] - t0) * 0.01);
    for (int bi = 1; bi <= 8; ++bi) for (int k = 0; k < 8; ++k) {
      const long long* ps = st.data() + 64 + (bi * 8 + k) * 24;
      if (!ps[0]) continue;
      fprintf(stderr, "  pair (row %d%s, col %d) start %.1f:", bi, bi == 8 ? " = aug" : "", k, (ps[0] - t0) * 0.01);
      for (int j = 0; j < k; ++j) fprintf(stderr, " U%d wait-end %.1f done %.1f;", j, (ps[1 + 2 * j] - t0) * 0.01, (ps[2 + 2 * j] - t0) * 0.01);
      if (k > 0) fprintf(stderr, " [U0: operands in %.1f, mfma done %.1f, stores drained %.1f]", (ps[12] - t0) * 0.01, (ps[13] - t0) * 0.01, (ps[14] - t0) * 0.01);
      if (ps[20]) fprintf(stderr, " S wait-end %.1f [operands in %.1f, mfma done %.1f] stored %.1f published %.1f", (ps[20] - t0) * 0.01, (ps[16] - t0) * 0.01, (ps[17] - t0) * 0.01, (ps[21] - t0) * 0.01, (ps[22] - t0) * 0.01);
      fprintf(stderr, "\n");
    }
  }
  return GPS_OK;
}

// after an aborted launch the counters are in an unknown state
int gps_small_factor_reset(gps_handle_t h) {
  if (h->dSmallSync.p) GPS_HIP(h, hipMemsetAsync(h->dSmallSync.p, 0, (size_t)(SN_WORDS + SN_INV_WORDS) * 4, h->stream));
  return GPS_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// K_y^-1 (lower triangle) and K_y^-1 (Y - m) of a small problem from its factor, as ONE launch: what the gradient of the
// likelihood needs (examples/gpr.py:53-54: tf.gradients through tf.cholesky), launch by launch ~25 launches at N = 512
// (L^-T by recursion, K^-1 = L^-T L^-1 by recursion, the backward substitution).  With Y = L^-1 (lower, blocks Y_ij):
//     Y_ii = W_i (the block inverses of the factorisation),   Y_ij = - sum_{k=j}^{i-1} (W_i L_ik) Y_kj   (i > j),
//     K^-1_ab = sum_{i >= a} Y_ia^T Y_ib   (a >= b),          A^T = Y^T alpha.
// Every product is one 16-row slab x one 128 x 128 block, as in the factorisation launch (operands through LDS, one column
// tile per wave); the slabs are spread over the workgroups as a list of tasks with counters between them (below).  M_ik =
// W_i L_ik is parked in the unused upper block (k, i) of the Y buffer.
#define SI_LSA 16                       // the 128 x 16 strip is read (4 s + fk) * 16 + fr: conflict-free; the per-wave 16 x 16 store staging too
#define SI_AS_DOUBLES 2304                // >= 16 * SN_LS (16 rows of A) and >= 128 * SI_LSA (the strip)
#define SI_LDS_BYTES ((128 * SN_LS + SI_AS_DOUBLES) * 8)      // a whole 128 x 128 B operand + the A operand (16 rows, or a 128 x 16 column strip)

struct SmallInvArgs {
  const double* L; i64 ld;        // factor (lower blocks), [np][ld]
  const double* W;                // [nblk][128 * 128] block inverses
  const double* alpha; i64 ld_alpha; int r;      // alpha^T [r][ld_alpha]
  double* Y; i64 ldy;             // [np][ldy] work: Y (lower blocks), M (upper blocks)
  double* Kinv; i64 ldk;          // out: lower blocks of K^-1
  double* At; i64 lda;            // out: (K^-1 resid)^T [r][lda]
  double* AtT; int n;             // optional second home of the same, as [n][r] (what the host wants back)
  int nblk;
  u32* sync;                      // zero on entry; zeroed again at the end
  double* res;                    // [0] abort
};

// acc += A * B  with A [16 x 128] (rows in LDS, or AT: staged as its transpose's strip [128][16]) and B [128 x 128] given as B[k][n]
template <bool AT>
__device__ __forceinline__ void si_mma(v4d& acc, const double* As, const double* Bs, int wave, int fr, int fk) {
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const double a = AT ? As[(4 * s + fk) * SI_LSA + fr] : As[fr * SN_LS + 4 * s + fk];
    const double b = Bs[(4 * s + fk) * SN_LS + 16 * wave + fr];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
}
// whole 128 x 128 block (rows of the source) -> Bs[128][SN_LS]: wave w its rows 16 w ..
__device__ __forceinline__ void si_stage_B(double* Bs, const double* g, i64 ldg, int wave, int lane) {
  sn_stage_rows<16>(Bs, g, ldg, 16 * wave, 16 * wave + 16, lane);
}
// 16 rows of A -> As[16][SN_LS] (wave w: rows 2 w, 2 w + 1)
__device__ __forceinline__ void si_stage_A(double* As, const double* g, i64 ldg, int wave, int lane) {
  sn_stage_rows<16>(As, g, ldg, 2 * wave, 2 * wave + 2, lane);
}
// column strip [128 rows][16 columns] -> As[128][SI_LSA]: 8 lanes per row, 16 bytes each
__device__ __forceinline__ void si_stage_At(double* As, const double* g, i64 ldg, int wave, int lane) {
  double2 v[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) v[u] = *reinterpret_cast<const double2*>(g + (i64)(16 * wave + 8 * u + (lane >> 3)) * ldg + 2 * (lane & 7));
#pragma unroll
  for (int u = 0; u < 2; ++u) *reinterpret_cast<double2*>(As + (16 * wave + 8 * u + (lane >> 3)) * SI_LSA + 2 * (lane & 7)) = v[u];
}
// the wave's 16 x 16 tile (accumulator layout) -> rows of C, whole 128-byte segments, write-through
__device__ __forceinline__ void si_store_tile(double* scratch, const v4d& acc, double sign, double* C, i64 ldc, int wave, int lane, int fr, int fk) {
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) scratch[(fk + 4 * rg) * SI_LSA + fr] = sign * acc[rg];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int hlf = 0; hlf < 2; ++hlf) {
    const int row = 8 * hlf + (lane >> 3), col = 2 * (lane & 7);
    const double2 v = *reinterpret_cast<const double2*>(scratch + row * SI_LSA + col);
    pb_store16(&C[(i64)row * ldc + 16 * wave + col], v);
  }
}

// One workgroup = one task at a time (ticket order = the order below; a task only waits for tasks before it):
//   M (i, k, sl), k < i      : M_ik[slab] = W_i[slab] L_ik                                    -> counter SI_M(i, sl)  (i of them)
//   Y (i, j, sl), j < i      : Y_ij[slab] = - sum_{k=j}^{i-1} M_ik[slab] Y_kj  (Y_jj = W_j)  -> counter SI_Y(i, j)   (8 slabs)
//                              ordered by i - j: the blocks next to the diagonal need no other Y block
//   K (a, b, sl), b <= a     : K^-1_ab[slab] = sum_{i >= a} Y_ia[:, slab]^T Y_ib
//   A (t)                    : block t of (Y^T alpha)^T
// Chain of dependent products at four blocks: M -> Y_10 -> Y_20 -> Y_30 -> K_00: five, against the nine plus four of a
// workgroup that walks a whole slab of Y by itself.
// The tasks are drawn from a queue in that order by however many workgroups are resident (a task only waits for tasks drawn
// before it, and whoever has drawn a task is running: no co-residency assumption at all) -- also when there are as many
// workgroups as tasks (round 5: that form took its task from blockIdx before).
#define SI_QUEUE (1 * SN_LINE)                             // (line 0 is sn_wait's abort word: SN_ABORT)
#define SI_DONE (2 * SN_LINE)
#define SI_MAXB 16                                         // blocks the counters are laid out for (2048 padded points)
#define SI_M(i, sl) ((3 + (i) * 8 + (sl)) * SN_LINE)       // [16][8]
#define SI_Y(i, j) ((3 + 8 * SI_MAXB + (i) * SI_MAXB + (j)) * SN_LINE)        // [16][16]
#define SI_WORDS ((3 + 8 * SI_MAXB + SI_MAXB * SI_MAXB) * SN_LINE)
template <bool QUEUED>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void small_inverse_kernel(SmallInvArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* Bs = reinterpret_cast<double*>(smem_raw);
  double* As = Bs + 128 * SN_LS;
  int* const s_flag_p = reinterpret_cast<int*>(smem_raw + SI_LDS_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int nblk = g.nblk, npair = nblk * (nblk - 1) / 2;
  u32* sync = g.sync;                       // (sn_wait looks for the abort word at sync + SN_ABORT)
  double* scratch = As + 16 * SI_LSA * wave;          // per-wave 16 x 16 staging for the stores: inside the A region, used only between products
  bool ok = true;
  const int ntasks = 16 * npair + 8 * (npair + nblk) + nblk;
  // (QUEUED = false: as many workgroups as tasks -- up to four blocks --, one draw each.  A template parameter: as a run-time
  // flag the loop around the tasks cost the straight-line form 11 us of its 38)
  constexpr bool queued = QUEUED;
  for (int round = 0;; ++round) {
  // the task is DRAWN (both forms; QUEUED = false: exactly once), never derived from blockIdx: a task only waits for tasks
  // drawn before it, and whoever has drawn a task is running -- no co-residency assumption at all
  if (!queued && round > 0) break;
  bool aborted = false;
  const int t = sn_draw(sync, SI_QUEUE, s_flag_p, &aborted);
  if (aborted) break;
  if (t >= ntasks || !ok) break;
  const int sl = t % 8;

  if (t < 8 * npair) {
    // ---- M (i, k, sl)
    int i = 1, rem = t / 8;
    while (rem >= i) { rem -= i; ++i; }
    const int k = rem;
    si_stage_A(As, g.W + (i64)i * 128 * 128 + (i64)16 * sl * 128, 128, wave, lane);
    si_stage_B(Bs, g.L + (i64)i * 128 * g.ld + (i64)k * 128, g.ld, wave, lane);
    __syncthreads();
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    si_mma<false>(acc, As, Bs, wave, fr, fk);
    __syncthreads();
    // parked in the unused upper block (k, i) of the Y buffer
    si_store_tile(scratch, acc, 1.0, g.Y + ((i64)k * 128 + 16 * sl) * g.ldy + (i64)i * 128, g.ldy, wave, lane, fr, fk);
    sn_publish(sync + SI_M(i, sl), nullptr);
  } else if (t < 16 * npair) {
    // ---- Y (i, j, sl): pairs by distance d = i - j, then by j
    int d = 1, rem = (t - 8 * npair) / 8;
    while (rem >= nblk - d) { rem -= nblk - d; ++d; }
    const int j = rem, i = j + d;
    ok = sn_wait(sync, sync + SI_M(i, sl), (u32)i, nullptr, 0u, s_flag_p);
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int k = j; k < i && ok; ++k) {
      si_stage_A(As, g.Y + ((i64)k * 128 + 16 * sl) * g.ldy + (i64)i * 128, g.ldy, wave, lane);          // M_ik[slab]
      if (k == j) {
        si_stage_B(Bs, g.W + (i64)j * 128 * 128, 128, wave, lane);                                          // Y_jj = W_j
      } else {
        ok = sn_wait(sync, sync + SI_Y(k, j), 8u, nullptr, 0u, s_flag_p);
        if (!ok) break;
        si_stage_B(Bs, g.Y + (i64)k * 128 * g.ldy + (i64)j * 128, g.ldy, wave, lane);                      // Y_kj
      }
      __syncthreads();
      si_mma<false>(acc, As, Bs, wave, fr, fk);
      __syncthreads();
    }
    if (ok) {
      si_store_tile(scratch, acc, -1.0, g.Y + ((i64)i * 128 + 16 * sl) * g.ldy + (i64)j * 128, g.ldy, wave, lane, fr, fk);
      sn_publish(sync + SI_Y(i, j), nullptr);
    }
  } else if (t < 16 * npair + 8 * (npair + nblk)) {
    // ---- K (a, b, sl)
    int a = 0, rem = (t - 16 * npair) / 8;
    while (rem >= a + 1) { rem -= a + 1; ++a; }
    const int b = rem;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int i = a; i < nblk && ok; ++i) {
      if (i > a) ok = sn_wait(sync, sync + SI_Y(i, a), 8u, (i > b && b != a) ? sync + SI_Y(i, b) : nullptr, 8u, s_flag_p);
      else if (b < a) ok = sn_wait(sync, sync + SI_Y(a, b), 8u, nullptr, 0u, s_flag_p);
      if (!ok) break;
      const double* Ya = (i == a) ? g.W + (i64)a * 128 * 128 + 16 * sl : g.Y + (i64)i * 128 * g.ldy + (i64)a * 128 + 16 * sl;
      const double* Yb = (i == b) ? g.W + (i64)b * 128 * 128 : g.Y + (i64)i * 128 * g.ldy + (i64)b * 128;
      si_stage_At(As, Ya, (i == a) ? 128 : g.ldy, wave, lane);
      si_stage_B(Bs, Yb, (i == b) ? 128 : g.ldy, wave, lane);
      __syncthreads();
      si_mma<true>(acc, As, Bs, wave, fr, fk);
      __syncthreads();
    }
    if (ok) si_store_tile(scratch, acc, 1.0, g.Kinv + ((i64)a * 128 + 16 * sl) * g.ldk + (i64)b * 128, g.ldk, wave, lane, fr, fk);
  } else {
    // ---- A^T = Y^T alpha, block tb: thread (c, part) sums a quarter of the rows of column c, LDS adds the quarters
    const int tb = t - (16 * npair + 8 * (npair + nblk));
    for (int i = tb + 1; i < nblk && ok; ++i) ok = sn_wait(sync, sync + SI_Y(i, tb), 8u, nullptr, 0u, s_flag_p);
    if (ok) {
      const int c = tid & 127, part = tid >> 7;
      for (int q = 0; q < g.r; ++q) {
        double sum = 0.0;
        for (int i = tb; i < nblk; ++i) {
          const double* Yc = (i == tb) ? g.W + (i64)tb * 128 * 128 + c : g.Y + (i64)i * 128 * g.ldy + (i64)tb * 128 + c;
          const i64 ldc = (i == tb) ? 128 : g.ldy;
          const double* al = g.alpha + (i64)q * g.ld_alpha + (i64)i * 128;
#pragma unroll 8
          for (int m = 32 * part; m < 32 * part + 32; ++m) sum = fma(Yc[(i64)m * ldc], al[m], sum);
        }
        Bs[part * 128 + c] = sum;
        __syncthreads();
        if (part == 0) {
          const double v = (Bs[c] + Bs[128 + c]) + (Bs[256 + c] + Bs[384 + c]);
          g.At[(i64)q * g.lda + (i64)tb * 128 + c] = v;
          if (g.AtT && tb * 128 + c < g.n) g.AtT[(i64)(tb * 128 + c) * g.r + q] = v;
        }
        __syncthreads();
      }
    }
  }
  }   // task loop
  // ---- the last workgroup to finish leaves the counters zero for the next call and reports
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const u32 done = __hip_atomic_fetch_add(sync + SI_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int last = 0;
    if (done + 1 == gridDim.x) {
      const u32 ab = sn_load(sync + SN_ABORT);
      g.res[0] = (double)ab;
      last = (ab == 0u) ? 1 : 0;
    }
    s_flag_p[2] = last;
  }
  __syncthreads();
  if (s_flag_p[2]) for (int w = tid; w < SI_WORDS; w += NT) sync[w] = 0u;
}

// dK: the factor of gps_launch_small_factor; linv: its block inverses; d_alpha [r][np].  Fills dY (work), dKinv (lower blocks of
// K^-1) and dA ([r][np]: K^-1 resid; dAT, if given: the same as [n][r]); res1 (device): abort flag.  GPS_ERR_UNSUPPORTED: not a shape for this path.
int gps_launch_small_inverse(gps_handle_t h, const double* dK, i64 np, const double* linv, const double* d_alpha, i64 r,
                             double* dY, double* dKinv, double* dA, double* d_res1, double* dAT, i64 n) {
  if (np % 128 || np < 128 || np > 128 * 16 || r < 1 || !h->dSmallSync.p) return GPS_ERR_UNSUPPORTED;
  const int nblk = (int)(np / 128);
  SmallInvArgs a;
  a.L = dK; a.ld = np; a.W = linv; a.alpha = d_alpha; a.ld_alpha = np; a.r = (int)r;
  a.Y = dY; a.ldy = np; a.Kinv = dKinv; a.ldk = np; a.At = dA; a.lda = np; a.nblk = nblk; a.AtT = dAT; a.n = (int)n;
  a.sync = (u32*)h->dSmallSync.p + SN_WORDS; a.res = d_res1;
  const size_t lds = (size_t)SI_LDS_BYTES + 64;
  int rc0 = gps_dyn_lds(h, reinterpret_cast<const void*>(&small_inverse_kernel<false>), (int)lds);
  if (!rc0) rc0 = gps_dyn_lds(h, reinterpret_cast<const void*>(&small_inverse_kernel<true>), (int)lds);
  if (rc0) return rc0;
  LaunchScope ls(h, KC_GEMM, 2.0 * (double)np * np * np / 3.0, 16.0 * np * np);
  if (h->small_fault_inject > 0 && --h->small_fault_inject == 0) {
    const u32 one = 1u;
    GPS_HIP(h, hipMemcpyAsync((u32*)h->dSmallSync.p + SN_WORDS + SN_ABORT, &one, 4, hipMemcpyHostToDevice, h->stream));
    GPS_HIP(h, hipStreamSynchronize(h->stream));
  }
  const int npair = nblk * (nblk - 1) / 2;
  const int ntasks = 16 * npair + 8 * (npair + nblk) + nblk;          // M, Y, K and A tasks: 180 at four blocks, 567 at seven
  const int slots = h->prop.multiProcessorCount - 8;                  // one workgroup per CU (LDS); a few CUs left to whatever else runs
  const int grid = ntasks < slots ? ntasks : slots;
  if (grid == ntasks) hipLaunchKernelGGL(small_inverse_kernel<false>, dim3(grid), dim3(NT), lds, h->stream, a);
  else hipLaunchKernelGGL(small_inverse_kernel<true>, dim3(grid), dim3(NT), lds, h->stream, a);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
