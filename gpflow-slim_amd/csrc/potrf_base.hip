// Base case of the blocked Cholesky: factor one 128x128 diagonal block inside a
// single workgroup and invert the factor.
//
// Replaces the innermost part of tf.cholesky (models/gpr.py:70,121;
// conditionals.py:84).  The explicit 128x128 inverse turns every panel solve
// X L11^T = B into an MFMA GEMM (gemm_f64.hip) and every vector solve into a
// 128x128 gemv, which is what makes the recursive trsm / trsv GEMM-only.
//
// One workgroup of 8 waves, LDS image [128][129], 8 steps of 16 columns.  The kernel is one long dependent chain
// (128 pivots), so the waves are specialised and everything that is not on the chain runs in its shadow:
//
//   phase A(g)  PANEL waves (wave 0; wave 1 while more than 32 rows lie below the block; wave 2 at step 0): every ROW of 16
//               lanes holds a copy of the 16 rows of diagonal block g and factors it in registers (hardware rsqrt + one
//               third-order correction, no barrier; broadcasts by DPP row_newbcast -- panel_step_dpp below); every lane
//               carries a second vector, a row below the block, that takes the very same updates -- which for it IS the forward
//               substitution x L_gg^T = b -- so the whole 16-column panel is finished when the diagonal block is.  Lanes
//               0..15 of wave 0 carry the rows of the identity instead (they end as the columns of inv(L_gg): the 16x16
//               triangular inverse costs no instruction of its own), lanes 16..31 the rows of the block itself (they end as
//               the rows of L_gg); one loop of 16 LDS stores per lane puts all of it back into the image.
//               The OTHER waves, meanwhile, pull items from two work queues (LDS counters, longest items first): MFMA items --
//               block row g-1 of the inverse (below) and the LEFT-LOOKING update of block column g+1 with the panels 0..g-1
//               (one accumulator chain of 4 g MFMAs per 16x16 tile, operands fetched one panel ahead) --, then store items:
//               the rows of block g-1 of L leave for HBM here instead of in one burst at the end.
//   phase B(g)  block column g+1 takes panel g (the only update the next panel waits for): one 4-MFMA tile per wave.
//   tail        block row 7 of the inverse; beside it (queue) the rows of inv(L) that are final and the last rows of L.
//
// Inverse X = inv(L), built by block rows in the shadow of the factorisation: the lower triangle of the image holds
// L, the strictly upper triangle receives X^T (X is lower triangular, so its transpose fits exactly there), 1/L_ii
// sits in dinv[].  X_gc = -X_gg * sum_{c <= k < g} L_gk X_kc for c < g: two MFMA chains per 16x16 tile, the first
// product handed to the second in registers (its accumulator layout is the B-operand layout).  Block row g trails the
// factorisation by one step; after the last panel only block row 7 is left.  Stride 129 makes row and column walks
// conflict free.
#include "gps_common.hpp"

#define PB 128
#define PS 129
#define NT 512
// Measured and not kept (tools/pb_stamps.py, tools/dpp_probe.hip; round 4):
//  * no fp64 MFMA on a panel wave's SIMD.  The fp64 vector FMAs of a wave drop from one per 6.4 cycles to one per 22 while
//    ANOTHER wave of the same SIMD issues fp64 MFMAs (the two share the DP pipe and the MFMA wins), so with the partners of
//    the panel waves idle the pivot chain runs at 1.55 instead of 1.9 us per 16 columns -- but the other waves' work then sits
//    on two SIMDs: 29.9 us per call against 29.3 with every non-panel wave pulling from both queues.
//  * s_setprio for the panel waves: no effect either way.
//  * rows of inv(L) stored in the panels' shadow as well: +2 us (the waves that would have the issue slots for it do not
//    have the time); they leave beside block row 7 of the inverse instead.
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d_ __attribute__((ext_vector_type(2)));

// 16-byte global store of a result another WORKGROUP may read inside the same launch (small_n.hip): write-through (sc1), so
// that the consumer needs no release fence on this side -- only the drained stores and a flag (cdna_hip_programming.md,
// Guideline 16 R1).  GPS_PB_WT 0 (the stand-alone kernel): a plain store.
#ifndef GPS_PB_WT
#define GPS_PB_WT 0
#endif
__device__ __forceinline__ void pb_store16(double* p, double2 v) {
#if GPS_PB_WT
  v2d_ w; w.x = v.x; w.y = v.y;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory");
#else
  *reinterpret_cast<double2*>(p) = v;
#endif
}
typedef __attribute__((address_space(3))) int lds_int;

// whole 128x128 block -> LDS image; 16-byte loads, all 16 of a thread in flight at once (one memory round trip:
// a lone workgroup draws ~40 GB/s, so the tile load is latency, not bandwidth); the upper triangle is loaded too:
// it is never read before being overwritten
__device__ __forceinline__ void load_image(double* a, const double* __restrict__ A, i64 lda, int tid) {
  double2 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int idx = u * NT + tid;
    const int i = idx >> 6, j = (idx & 63) * 2;
    v[u] = *reinterpret_cast<const double2*>(A + (i64)i * lda + j);
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int idx = u * NT + tid;
    const int i = idx >> 6, j = (idx & 63) * 2;
    a[i * PS + j] = v[u].x;
    a[i * PS + j + 1] = v[u].y;
  }
}

// X (= inv L, lower) is stored transposed in the strict upper triangle of the image, diag in dinv.
__device__ __forceinline__ double x_elem(const double* a, const double* dinv, int r, int c) {
  // two UNCONDITIONAL LDS reads + selects.  The empty asm pins both loaded values: without it the compiler
  // sinks the reads back under divergent branches and every MFMA of the inverse waits for its own
  // pair of serialised LDS round trips.
  double off = a[(r > c) ? (c * PS + r) : 0];
  double dia = dinv[r & (PB - 1)];
  asm volatile("" : "+v"(off), "+v"(dia));
  return (r > c) ? off : ((r == c) ? dia : 0.0);
}

// ---- phase A, panel wave: 16 columns o..o+15, broadcasts by DPP (round 4; rounds 1-3 used v_readlane + an LDS scratch).
// gfx950 has two fp64 DPP forms, v_fmac_f64_dpp and v_mov_b64_dpp, and for the DP ALU only the control row_newbcast:k
// (lane k of each ROW of 16 lanes to every lane of that row).  Measured (tools/dpp_probe.hip, one wave alone on its SIMD):
// a broadcast update by two v_readlane + v_fma_f64 issues every 23 cycles, v_fmac_f64_dpp every 6.4 -- the price of a
// plain v_fma_f64.  A row broadcast cannot reach another row of 16 lanes, so every row of the wave keeps its OWN copy of
// the 16 rows of the diagonal block (r[], factored four times over by the same instructions) and every lane carries a
// second vector x[] -- a row below the block, or (owner, lanes 0..15) a row of the identity, which ends as a column of
// inv(L_gg) -- that takes the same updates with the multiplier broadcast from the lane's own copy of the block:
//     r[k] -= r[j] * bcast(r[j], lane k)        x[k] -= x[j] * bcast(r[j], lane k)
// 480 instructions per 16 columns instead of ~620, none of them a v_readlane, no LDS scratch, 64 rows below per wave
// (owner: 48) so two panel waves cover what took three.  Same arithmetic per element as the v_readlane form (one fma
// per update, in the same column order): the factor is bit-identical.
// Hazards (the compiler pads nothing inside or around an asm statement): a VGPR written by a VALU instruction must not
// be read through DPP for two wait states -- the scaling of column j ends with s_nop 1, and every other DPP source was
// written at least two instructions earlier (order of the volatile statements below).
#define PB_FMAC_DPP(d, a, b, k) asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(a), "v"(b), "n"(k))
#define PB_MOV_DPP(d, a, k) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(a), "n"(k))
#define PB_SCALE2(a, b, y) asm volatile("v_mul_f64 %0, %0, %2\n\tv_mul_f64 %1, %1, %2\n\ts_nop 1" : "+v"(a), "+v"(b) : "v"(y))
// updates of column K by column J (K > J): the diagonal copy and the lane's own row
#define PB_UPD2(J, K) do { PB_FMAC_DPP(r[K], r[J], r[J], K); PB_FMAC_DPP(x[K], r[J], x[J], K); } while (0)

template <int J>
__device__ __forceinline__ void pb_pivot(double (&r)[16], double (&x)[16], double& p, double& y, double& sd, double& dd, int& bad) {
  // ---- chain: l = s y_j, p_{j+1} = d - l^2 (the value lane j+1 of the copy ends up with), its reciprocal square root.
  // One chain operation per group of updates, the order pinned with sched_barrier: left alone the compiler issues the
  // five dependent operations of the rsqrt refinement back to back (10 cycles each instead of an issue slot of 6.4).
#define PB_FENCE() __builtin_amdgcn_sched_barrier(0)
  const double lj = sd * y;
  PB_SCALE2(r[J], x[J], y);                      // column j: L_kj (rows of the block) / x_j (rows below)
  PB_FENCE();
  const double pn = __builtin_fma(-lj, lj, dd);
  // the two columns the next pivot reads first, then its s and d (two other instructions between a write and its DPP read)
  if constexpr (J + 1 < 16) PB_UPD2(J, J + 1);
  PB_FENCE();
  double yn = __builtin_amdgcn_rsq(pn);
  bad = (!(pn > 0.0) && bad == 0x7fffffff) ? J + 1 : bad;
  if constexpr (J + 2 < 16) PB_UPD2(J, J + 2);
  if constexpr (J + 3 < 16) PB_UPD2(J, J + 3);
  PB_FENCE();
  const double t = pn * yn;
  if constexpr (J + 2 < 16) { PB_MOV_DPP(sd, r[J + 1], J + 2); PB_MOV_DPP(dd, r[J + 2], J + 2); }
  if constexpr (J + 4 < 16) PB_UPD2(J, J + 4);
  PB_FENCE();
  const double e = __builtin_fma(-t, yn, 1.0);
  if constexpr (J + 5 < 16) PB_UPD2(J, J + 5);
  if constexpr (J + 6 < 16) PB_UPD2(J, J + 6);
  if constexpr (J + 7 < 16) PB_UPD2(J, J + 7);
  PB_FENCE();
  const double q = __builtin_fma(0.375, e, 0.5);
  const double ye = yn * e;
  if constexpr (J + 8 < 16) PB_UPD2(J, J + 8);
  if constexpr (J + 9 < 16) PB_UPD2(J, J + 9);
  if constexpr (J + 10 < 16) PB_UPD2(J, J + 10);
  if constexpr (J + 11 < 16) PB_UPD2(J, J + 11);
  PB_FENCE();
  yn = __builtin_fma(ye, q, yn);
  if constexpr (J + 12 < 16) PB_UPD2(J, J + 12);
  if constexpr (J + 13 < 16) PB_UPD2(J, J + 13);
  if constexpr (J + 14 < 16) PB_UPD2(J, J + 14);
  if constexpr (J + 15 < 16) PB_UPD2(J, J + 15);
  PB_FENCE();
#undef PB_FENCE
  y = yn; p = pn;
}

// owner (wave 0): lanes 0..15 carry the rows of the identity, lanes 16..63 rows o+16 .. o+63; the other panel wave:
// rows xrow0 .. xrow0+63.  On return r[] (any row of lanes) holds the factored diagonal block, x[] of the owner's lanes
// 0..15 the columns of its inverse; rows below have gone back to the image.
__device__ __forceinline__ void panel_step_dpp(double* a, int* info, int row0, int o, int lane, int pw, int n_pw,
                                               lds_int* pflag, int flag_target) {
  // What the lane's own vector x is (pw: index of this panel wave; the r copy of the block is in every lane):
  //   owner (pw 0)  lanes  0..15  row c of the identity      -> column c of inv(L_gg): X[k][c], k > c, into the strict upper
  //                               triangle of the image (transposed), X[c][c] into dinv; zeros (exactly) for k < c
  //                 lanes 16..31  row c of the block itself   -> row c of L_gg (A_gg L_gg^-T = L_gg: the same recurrence in the
  //                               same order as the copy r, the same bits), k <= c into the lower triangle
  //                 lanes 32..63  rows o+16 .. o+47 below the block
  //   pw 1: rows o+48 .. o+111;  pw 2 (only at o = 0, where every other wave is idle anyway): rows o+112 ..
  // so that ONE loop of 16 LDS stores with a per-lane address (column range [klo, khi] of the row at `base`, anything else to a
  // dummy slot, the diagonal of the inverse to dinv) puts everything back -- the factored block included, at the end of this
  // phase rather than in a phase of its own.
  const int c = lane & 15;
  const bool owner = pw == 0;
  const bool is_inv = owner && lane < 16, is_diag = owner && lane >= 16 && lane < 32;
  const int row = owner ? (lane < 32 ? o + c : o + lane - 16) : o + 48 + 64 * (pw - 1) + lane;
  const bool valid = row < PB;
  const int rr = valid ? row : PB - 1;
  double r[16], x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    r[k] = a[(o + c) * PS + o + k];
    const double v = a[rr * PS + o + k];
    x[k] = is_inv ? ((k == c) ? 1.0 : 0.0) : v;
  }
  // The other panel waves have their copies of the block in registers: from then on the owner may overwrite it (below).  (The
  // LDS executes a wave's instructions in issue order: once the counter add has executed, so have the loads above.  The counter is
  // accessed through an explicit LDS pointer with relaxed atomics: an ordered access through a GENERIC pointer may touch
  // private memory as far as the compiler knows, and r[] / x[] would live in scratch instead of registers.)
  if (!owner && lane == 0) __hip_atomic_fetch_add(pflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  int bad = 0x7fffffff;
  double p, y, sd, dd;
  PB_MOV_DPP(p, r[0], 0);                        // pivot of the current column (the same in every lane)
  bad = !(p > 0.0) ? 0 : bad;
  {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double t = p * y0;
    const double e = __builtin_fma(-t, y0, 1.0);
    y = __builtin_fma(y0 * e, __builtin_fma(0.375, e, 0.5), y0);
  }
  PB_MOV_DPP(sd, r[0], 1); PB_MOV_DPP(dd, r[1], 1);
  pb_pivot<0>(r, x, p, y, sd, dd, bad);  pb_pivot<1>(r, x, p, y, sd, dd, bad);  pb_pivot<2>(r, x, p, y, sd, dd, bad);
  pb_pivot<3>(r, x, p, y, sd, dd, bad);  pb_pivot<4>(r, x, p, y, sd, dd, bad);  pb_pivot<5>(r, x, p, y, sd, dd, bad);
  pb_pivot<6>(r, x, p, y, sd, dd, bad);  pb_pivot<7>(r, x, p, y, sd, dd, bad);  pb_pivot<8>(r, x, p, y, sd, dd, bad);
  pb_pivot<9>(r, x, p, y, sd, dd, bad);  pb_pivot<10>(r, x, p, y, sd, dd, bad); pb_pivot<11>(r, x, p, y, sd, dd, bad);
  pb_pivot<12>(r, x, p, y, sd, dd, bad); pb_pivot<13>(r, x, p, y, sd, dd, bad); pb_pivot<14>(r, x, p, y, sd, dd, bad);
  PB_SCALE2(r[15], x[15], y);
  if (owner && lane == 0 && bad != 0x7fffffff) atomicMin(info, row0 + o + bad + 1);
  if (owner && n_pw > 1) { while (__hip_atomic_load(pflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < flag_target) __builtin_amdgcn_s_sleep(1); }
  const int base = rr * PS + o;
  const int klo = is_inv ? c + 1 : (valid ? 0 : 16);
  const int khi = is_diag ? c : 15;
  const int kd = is_inv ? c : -1;                 // the column whose value is the inverse's diagonal entry
  const int dummy = PB * PS + PB + lane, dslot = PB * PS + o + c;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    int addr = (k >= klo && k <= khi) ? base + k : dummy;
    addr = (k == kd) ? dslot : addr;
    a[addr] = x[k];
  }
}

// trailing-update tile: C_IJ -= P_I P_J^T, P = columns o..o+15 (one wave, fp64 MFMA 16x16x4).  I, J, o are wave-uniform:
// every address is a scalar base + the lane's constant fr * PS + fk (operands) / fk * PS + fr (accumulator) + an immediate.
__device__ __forceinline__ void update_tile(double* a, int I, int J, int o, int fr, int fk) {
  v4d acc;
  double av[4], bv[4];
  const double* pa = a + (16 * I * PS + o) + (fr * PS + fk);
  const double* pb = a + (16 * J * PS + o) + (fr * PS + fk);
  double* pc = a + (16 * I * PS + 16 * J) + (fk * PS + fr);
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) { av[s4] = -pa[4 * s4]; bv[s4] = pb[4 * s4]; }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) acc[rg] = pc[4 * rg * PS];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) pc[4 * rg * PS] = acc[rg];
}

// tile t (row-major over the lower triangle I >= J) of the blocks j0..7
__device__ __forceinline__ void tri_index(int t, int j0, int& I, int& J) {
  int Ip = 0, rem = t;
  while (rem >= Ip + 1) { rem -= Ip + 1; ++Ip; }
  I = j0 + Ip; J = j0 + rem;
}

// Left-looking tile: C_IJ -= sum over the panels p = 0 .. np-1 (columns 0 .. 16 np - 1) of P_I P_J^T -- what the right-looking
// form applied panel by panel (one LDS round trip of the accumulator and one 4-MFMA tile per panel) as ONE accumulator chain,
// the operands of panel p+1 fetched while the MFMAs of panel p run.  Same operands in the same order: the same bits.
// (The fetch is unconditional -- the last panel is fetched twice -- so that no control-flow merge sits between the loads and
// the MFMAs: at a merge the compiler's wait is lgkmcnt(0), i.e. no overlap.)
__device__ __forceinline__ void lookleft_tile(double* a, int I, int J, int np, int fr, int fk) {
  const double* pa = a + 16 * I * PS + (fr * PS + fk);
  const double* pb = a + 16 * J * PS + (fr * PS + fk);
  double* pc = a + (16 * I * PS + 16 * J) + (fk * PS + fr);
  v4d acc;
  double av[4], bv[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) { av[s4] = pa[4 * s4]; bv[s4] = pb[4 * s4]; }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) acc[rg] = pc[4 * rg * PS];
  for (int p = 0; p < np; ++p) {
    const int q = (p + 1 < np) ? p + 1 : p;
    double an[4], bn[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { an[s4] = pa[16 * q + 4 * s4]; bn[s4] = pb[16 * q + 4 * s4]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { av[s4] = an[s4]; bv[s4] = bn[s4]; }
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) pc[4 * rg * PS] = acc[rg];
}

// next item of a work queue in LDS (one counter per queue and step, zeroed at kernel start): wave-uniform
__device__ __forceinline__ int q_pull(int* counter, int lane) {
  int t = 0;
  if (lane == 0) t = atomicAdd(counter, 1);
  return __builtin_amdgcn_readfirstlane(t);
}

// ---- finished pieces go to HBM in the shadow of later panels (one wave per item; branch-free: unconditional LDS reads +
// selects, 16-byte stores at 32-bit offsets from a wave-uniform base)
// rows i0 .. i0+NR-1 of L (upper triangle zero-filled, like tf.cholesky)
template <int NR>
__device__ __forceinline__ void store_L_rows(const double* a, double* __restrict__ A, i64 lda, int i0, int lane) {
  const int j = 2 * lane;
  char* base = reinterpret_cast<char*>(A + (i64)i0 * lda) + 16 * lane;
  const unsigned step = (unsigned)lda * 8u;
  double2 v[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) { v[u].x = a[(i0 + u) * PS + j]; v[u].y = a[(i0 + u) * PS + j + 1]; }
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int i = i0 + u;
    v[u].x = (j <= i) ? v[u].x : 0.0;
    v[u].y = (j + 1 <= i) ? v[u].y : 0.0;
    pb_store16(reinterpret_cast<double*>(base + (size_t)(u * step)), v[u]);
  }
}
// X[r][c] for r > c sits at a[c][r]; the diagonal in dinv; zeros above
// rows i0 .. i0+NR-1 of inv(L)
template <int NR>
__device__ __forceinline__ void store_inv_rows(const double* a, const double* dinv, double* __restrict__ Linv, int i0, int lane) {
  const int c = 2 * lane;
  double2 v[NR];
  double d[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) { v[u].x = a[c * PS + i0 + u]; v[u].y = a[(c + 1) * PS + i0 + u]; d[u] = dinv[i0 + u]; }
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int i = i0 + u;
    v[u].x = (i > c) ? v[u].x : ((i == c) ? d[u] : 0.0);
    v[u].y = (i > c + 1) ? v[u].y : ((i == c + 1) ? d[u] : 0.0);
    pb_store16(Linv + i * PB + c, v[u]);
  }
}
// 16x16 triangular inverse of diagonal block gb, one column per lane (lanes 0..15 of one wave).
// X[i][c] is stored at a[c][i] (i > c); X[c][c] = dinv[c].  The column stays in registers: reading back the X
// entries this lane has just written would make every step wait for an LDS store -> load round trip.
__device__ __forceinline__ void inv_diag16(double* a, const double* dinv, int c) {
  const int rem = 15 - (c & 15);    // rows below c inside its 16-block
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

// block (gp, c), c < gp, of the inverse:  X_gp,c = -X_gp,gp * sum_{c <= kb < gp} L_gp,kb X_kb,c   (one wave).
// The kb = c term reads the triangular X_cc through x_elem (zeros above its diagonal); for kb > c both operands are
// plain tiles of the image, fetched one iteration ahead of the MFMAs that consume them.
__device__ __forceinline__ void inv_row_tile(double* a, const double* dinv, int gp, int c, int fr, int fk) {
  v4d t0 = (v4d){0.0, 0.0, 0.0, 0.0}, t1 = (v4d){0.0, 0.0, 0.0, 0.0};
  const double* la = a + (16 * gp + fr) * PS + fk;          // L[16 gp + fr][k]     at la[k - fk]
  const double* xb = a + (16 * c + fr) * PS + fk;           // X[k][16 c + fr]      at xb[k - fk]   (k >= 16 (c + 1))
  double av[4], bv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    av[s] = la[16 * c + 4 * s];
    bv[s] = x_elem(a, dinv, 16 * c + 4 * s + fk, 16 * c + fr);
  }
  for (int kb = c + 1; kb < gp; ++kb) {
    double an[4], bn[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { an[s] = la[16 * kb + 4 * s]; bn[s] = xb[16 * kb + 4 * s]; }
    t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], t0, 0, 0, 0);
    t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], t1, 0, 0, 0);
    t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], t0, 0, 0, 0);
    t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], t1, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) { av[s] = an[s]; bv[s] = bn[s]; }
  }
  double xv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) xv[s] = x_elem(a, dinv, 16 * gp + fr, 16 * gp + 4 * s + fk);   // X_gp,gp[fr][4 s + fk]
  t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], t0, 0, 0, 0);
  t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], t1, 0, 0, 0);
  t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], t0, 0, 0, 0);
  t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], t1, 0, 0, 0);
  t0 += t1;      // T[4 s + fk][fr] = t0[s]: exactly the B operand of k-step s of the second product
  v4d z = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 4; ++s) z = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[s], t0[s], z, 0, 0, 0);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) a[(16 * c + fr) * PS + 16 * gp + fk + 4 * rg] = -z[rg];   // X[16 gp + fk + 4 rg][16 c + fr]
}

// factor != 0: A holds the SPD block, L is written back.  factor == 0: A already holds L
// (caller-supplied factor); only the inverse is produced.  LinvT (optional) receives inv(L)^T.
// One 128x128 block, by the calling workgroup (NT threads, `smem_raw`: PB_LDS_BYTES of LDS); every thread returns after the
// block's last global store has been ISSUED (callers that hand the results to other workgroups fence / drain themselves).
__device__ __forceinline__ void potrf_base_body(char* smem_raw, double* __restrict__ A, i64 lda, double* __restrict__ Linv,
                                                double* __restrict__ LinvT, int* __restrict__ info, int row0, int factor,
                                                long long* __restrict__ stamps) {
#define STAMP(q) do { if (stamps && threadIdx.x == 0) { stamps[q] = (long long)wall_clock64(); stamps[8 + q] = (long long)clock64(); } } while (0)
  STAMP(0);
  double* a = reinterpret_cast<double*>(smem_raw);   // [PB][PS]
  double* dinv = a + PB * PS;                        // [PB], then 3 x 64 dummy store targets

  const int tid = threadIdx.x;
  // (readfirstlane: the compiler cannot prove tid >> 6 wave-uniform and would otherwise run every role branch, item loop and
  // tile index below as divergent control flow with per-lane copies of the scalars)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;

  int* q_mfma = reinterpret_cast<int*>(dinv + PB + 192);           // [9] next item of the MFMA queue of step g; [8]: the queue of the tail
  int* q_store = q_mfma + 9;                                        // [8] ... of the store queue
  lds_int* pflag = (lds_int*)(q_store + 8);                         // panel waves (other than the owner) whose copy of the diagonal block is in registers, all steps so far
  __syncthreads();                                                  // (a caller's previous use of the LDS is over)
  if (tid < 18) q_mfma[tid] = 0;
  load_image(a, A, lda, tid);
  __syncthreads();
  STAMP(1);

  int tail_row0 = 0;         // first row of inv(L) that the common tail still has to store
  int flag_sum = 0;          // panel waves other than the owner that have announced their copy of the block, all steps so far
  if (factor) {
    long long t_a = 0, t_b = 0, t0 = 0;
    for (int g = 0; g < 8; ++g) {
      const int o = 16 * g;
      const int nrows = PB - o - 16;                       // rows below the diagonal block
      // panel waves: the owner takes 32 of the rows below, wave 1 up to 64 more, wave 2 the last 16 of step 0
      const int npanel = nrows > 96 ? 3 : (nrows > 32 ? 2 : 1);
      if (stamps && (tid == 0 || (lane == 0 && stamps[31]))) t0 = (long long)wall_clock64();
      // ---- phase A
      if (wave < npanel) {
        flag_sum += npanel - 1;
        panel_step_dpp(a, info, row0, o, lane, wave, npanel, pflag, flag_sum);
      } else if (g >= 1) {
        {
          const int n_inv = g - 1, n_ll = (g <= 6) ? 7 - g : 0, n_items = n_inv + n_ll;
          for (int it = q_pull(q_mfma + g, lane); it < n_items; it = q_pull(q_mfma + g, lane)) {
            // order: inverse tile 0 | left-looking tiles | inverse tiles 1 ..
            if (n_inv > 0 && it == 0) inv_row_tile(a, dinv, g - 1, 0, fr, fk);
            else if (it - (n_inv > 0 ? 1 : 0) < n_ll) lookleft_tile(a, g + 1 + it - (n_inv > 0 ? 1 : 0), g + 1, g, fr, fk);
            else inv_row_tile(a, dinv, g - 1, it - n_ll, fr, fk);
          }
        }
        for (int it = q_pull(q_store + g, lane); it < 4; it = q_pull(q_store + g, lane))
          store_L_rows<4>(a, A, lda, 16 * (g - 1) + 4 * it, lane);
      }
      if (stamps && lane == 0 && stamps[31]) stamps[32 + 8 * g + wave] = (long long)wall_clock64() - t0;   // diagnostics (stamps[31] != 0): per-wave end of phase A
      __syncthreads();
      if (stamps && tid == 0) { const long long t1 = (long long)wall_clock64(); t_a += t1 - t0; t0 = t1; }
      // ---- phase B: block column g+1 takes panel g (the only update the next panel waits for), one tile per wave
      if (wave == 0) { }
      else {
        const int I = g + wave;                                  // waves 1..7: tiles (g+1 .. 7, g+1)
        if (I < 8) update_tile(a, I, g + 1, o, fr, fk);
      }
      __syncthreads();
      if (stamps && tid == 0) t_b += (long long)wall_clock64() - t0;
    }
    if (stamps && tid == 0) { stamps[16] = t_a; stamps[17] = t_b; stamps[18] = 0; }
    STAMP(2);
    // ---- what is left: block row 7 of the inverse (tile c = 0 is the longest chain: wave 0 takes it, wave 6 the shortest), and
    // beside it the stores that are possible already -- rows 112..127 of L, and the rows of inv(L) of the block rows that are
    // final (0..6): pulled in 4-row pieces from a queue by wave 7 at once and by every other wave as soon as its tile is done
    STAMP(3);
    STAMP(4);
    if (wave < 7) inv_row_tile(a, dinv, 7, wave, fr, fk);
    {
      for (int it = q_pull(q_mfma + 8, lane); it < 4 + 28; it = q_pull(q_mfma + 8, lane)) {
        if (it < 4) store_L_rows<4>(a, A, lda, 112 + 4 * it, lane);
        else store_inv_rows<4>(a, dinv, Linv, 4 * (it - 4), lane);
      }
    }
    tail_row0 = 112;
    __syncthreads();
  } else {
    if (tid < PB) dinv[tid] = 1.0 / a[tid * PS + tid];
    __syncthreads();
    STAMP(2); STAMP(3);
    if (tid < PB) inv_diag16(a, dinv, tid);
    __syncthreads();
    STAMP(4);
    for (int gp = 1; gp < 8; ++gp) {
      if (wave < gp) inv_row_tile(a, dinv, gp, wave, fr, fk);
      __syncthreads();
    }
  }

  STAMP(5);
  // inverse (and its transpose) to HBM, explicit zeros above the diagonal, 16-byte stores.  With factor != 0 the rows 0..111
  // of the inverse left beside the computation of its block row 7: only that block row remains
  {
    const int i0 = 0, w2 = PB / 2;                     // (the whole transposed inverse: nothing of it left early)
    for (int idx = tail_row0 * 64 + tid; idx < PB * PB / 2; idx += NT) {
      const int i = idx >> 6, c = (idx & 63) * 2;
      double2 v;
      v.x = x_elem(a, dinv, i, c);
      v.y = x_elem(a, dinv, i, c + 1);
      pb_store16(Linv + i * PB + c, v);
    }
    if (LinvT) {
      for (int idx = tid; idx < PB * w2; idx += NT) {
        const int c = idx / w2, i = i0 + 2 * (idx - c * w2);        // LinvT[c][i] = X[i][c]
        double2 v;
        v.x = x_elem(a, dinv, i, c);
        v.y = x_elem(a, dinv, i + 1, c);
        pb_store16(LinvT + c * PB + i, v);
      }
    }
  }
  __syncthreads();
  STAMP(6);
#undef STAMP
}


#define PB_LDS_BYTES ((PB * PS + PB + 192 + 10) * 8)      // image, dinv, 3 x 64 dummy slots, queue counters + flag (18 ints)

#ifndef GPS_PB_DEVICE_ONLY
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void potrf_base_kernel(double* __restrict__ A, i64 lda,
                                                        double* __restrict__ Linv,
                                                        double* __restrict__ LinvT,
                                                        int* __restrict__ info, int row0,
                                                        int factor, long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  potrf_base_body(smem_raw, A, lda, Linv, LinvT, info, row0, factor, stamps);
}

int gps_launch_potrf_base(gps_handle_t h, double* A, i64 lda, double* Linv_blk,
                          double* LinvT_blk, int* d_info, i64 row0, int factor, long long* d_stamps) {
  const size_t lds = (size_t)PB_LDS_BYTES;
  int rc0 = gps_dyn_lds(h, reinterpret_cast<const void*>(&potrf_base_kernel), (int)lds);
  if (rc0) return rc0;
  // potrf n^3/3 + trtri n^3/3
  LaunchScope ls(h, KC_POTRF_BASE, 2.0 * PB * PB * PB / 3.0, 3.0 * PB * PB * 8.0);
  hipLaunchKernelGGL(potrf_base_kernel, dim3(1), dim3(NT), lds, h->stream, A, lda, Linv_blk,
                     LinvT_blk, d_info, (int)row0, factor, d_stamps);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
#endif   // GPS_PB_DEVICE_ONLY
