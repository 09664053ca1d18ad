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
//   phase A(g)  PANEL waves (wave 0, and wave 1 while more than 48 rows lie below the block): every ROW of 16 lanes holds a
//               copy of the 16 rows of diagonal block g and factors it in registers (hardware rsqrt + one third-order
//               correction, no barrier; broadcasts by DPP row_newbcast -- panel_step_dpp below); every lane carries a second
//               vector, a row below the block, that takes the very same updates -- which for it IS the forward substitution
//               x L_gg^T = b -- so the whole 16-column panel is finished when the diagonal block is.  Lanes 0..15 of
//               wave 0 carry the rows of the identity instead: what they end with is e_c^T L_gg^-T, i.e. column c of
//               inv(L_gg) -- the 16x16 triangular inverse costs no instruction of its own.
//               UPDATE waves (all others), meanwhile: the rest of the trailing update of step g-1 (16x16 tiles
//               C_IJ -= P_I P_J^T on the fp64 MFMA, J >= g+1; operands fetched one tile ahead), block row g-1 of the
//               inverse (below), and the stores of what is final: rows of block g-1 of L, rows of block g-2 of inv(L)
//               and the matching strip of its transpose leave for HBM here instead of in one burst at the end.
//   phase B(g)  waves 1..7: the tiles of block column g+1 of the trailing update of step g (the only ones the next
//               panel needs), one per wave; wave 0 puts the factored block and its inverse back into the image.
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

typedef double v4d __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ void panel_step_dpp(double* a, int* info, int row0, int o, int lane, int xrow0, bool owner,
                                               double (&r)[16], double (&x)[16]) {
  const int c = lane & 15;
  const bool is_inv = owner && lane < 16;
  const int row = owner ? o + lane : xrow0 + lane;       // (owner: lanes 16.. are rows o+16..)
  const bool valid = !is_inv && row < PB;
  const int rr = (row < PB) ? row : PB - 1;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    r[k] = a[(o + c) * PS + o + k];
    const double v = a[rr * PS + o + k];
    x[k] = is_inv ? ((k == c) ? 1.0 : 0.0) : v;
  }
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
  if (valid) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a[row * PS + o + k] = x[k];
  }
}

// trailing-update tile: C_IJ -= P_I P_J^T, P = columns o..o+15 (one wave, fp64 MFMA 16x16x4)
__device__ __forceinline__ void update_tile(double* a, int I, int J, int o, int fr, int fk) {
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

// tile t (row-major over the lower triangle I >= J) of the blocks j0..7
__device__ __forceinline__ void tri_index(int t, int j0, int& I, int& J) {
  int Ip = 0, rem = t;
  while (rem >= Ip + 1) { rem -= Ip + 1; ++Ip; }
  I = j0 + Ip; J = j0 + rem;
}

// The same tile in three pieces, so that a wave with several tiles to do fetches the operands of the next one before the
// four dependent MFMAs (64 cycles each) of the current one: a tile costs its MFMAs instead of MFMAs + an LDS round trip.
struct UTile { double av[4], bv[4]; v4d acc; int I, J; };
__device__ __forceinline__ void ut_load(UTile& t, const double* a, int I, int J, int o, int fr, int fk) {
  t.I = I; t.J = J;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    t.av[s4] = -a[(16 * I + fr) * PS + o + 4 * s4 + fk];
    t.bv[s4] = a[(16 * J + fr) * PS + o + 4 * s4 + fk];
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) t.acc[rg] = a[(16 * I + fk + 4 * rg) * PS + 16 * J + fr];
}
__device__ __forceinline__ void ut_run_store(UTile& t, double* a, int fr, int fk) {
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) t.acc = __builtin_amdgcn_mfma_f64_16x16x4f64(t.av[s4], t.bv[s4], t.acc, 0, 0, 0);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) a[(16 * t.I + fk + 4 * rg) * PS + 16 * t.J + fr] = t.acc[rg];
}
// tiles first, first + stride, ... < ntile of the trailing update (blocks j0..7, panel columns o..o+15); returns the first
// item index >= ntile of this wave's round-robin sequence
__device__ __forceinline__ int update_tiles_pipelined(double* a, int first, int stride, int ntile, int j0, int o, int fr, int fk) {
  int item = first;
  if (item >= ntile) return item;
  UTile ta, tb;
  int I, J;
  tri_index(item, j0, I, J);
  ut_load(ta, a, I, J, o, fr, fk);
  for (;;) {
    bool more = item + stride < ntile;
    if (more) { tri_index(item + stride, j0, I, J); ut_load(tb, a, I, J, o, fr, fk); }
    __builtin_amdgcn_sched_barrier(0);
    ut_run_store(ta, a, fr, fk);
    item += stride;
    if (!more) break;
    more = item + stride < ntile;
    if (more) { tri_index(item + stride, j0, I, J); ut_load(ta, a, I, J, o, fr, fk); }
    __builtin_amdgcn_sched_barrier(0);
    ut_run_store(tb, a, fr, fk);
    item += stride;
    if (!more) break;
  }
  return item;
}

// ---- finished pieces go to HBM in the shadow of later panels (one wave each; 16-byte stores)
// rows i0 .. i0+7 of L (upper triangle zero-filled, like tf.cholesky)
__device__ __forceinline__ void store_L_rows8(const double* a, double* __restrict__ A, i64 lda, int i0, int lane) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = i0 + u, j = 2 * lane;
    double2 v;
    v.x = (j <= i) ? a[i * PS + j] : 0.0;
    v.y = (j + 1 <= i) ? a[i * PS + j + 1] : 0.0;
    *reinterpret_cast<double2*>(A + (i64)i * lda + j) = v;
  }
}
// rows i0 .. i0+7 of inv(L)
__device__ __forceinline__ void store_inv_rows8(const double* a, const double* dinv, double* __restrict__ Linv, int i0, int lane) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = i0 + u, c = 2 * lane;
    double2 v;
    v.x = x_elem(a, dinv, i, c);
    v.y = x_elem(a, dinv, i, c + 1);
    *reinterpret_cast<double2*>(Linv + i * PB + c) = v;
  }
}
// columns 16 gp .. 16 gp + 15 of inv(L)^T for rows c0 .. c0+63:  LinvT[c][i] = X[i][c]
__device__ __forceinline__ void store_invT_strip64(const double* a, const double* dinv, double* __restrict__ LinvT, int gp, int c0, int lane) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = c0 + 8 * u + (lane >> 3), i = 16 * gp + 2 * (lane & 7);
    double2 v;
    v.x = x_elem(a, dinv, i, c);
    v.y = x_elem(a, dinv, i + 1, c);
    *reinterpret_cast<double2*>(LinvT + c * PB + i) = v;
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
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void potrf_base_kernel(double* __restrict__ A, i64 lda,
                                                        double* __restrict__ Linv,
                                                        double* __restrict__ LinvT,
                                                        int* __restrict__ info, int row0,
                                                        int factor, long long* __restrict__ stamps) {
#define STAMP(q) do { if (stamps && threadIdx.x == 0) { stamps[q] = (long long)wall_clock64(); stamps[8 + q] = (long long)clock64(); } } while (0)
  STAMP(0);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* a = reinterpret_cast<double*>(smem_raw);   // [PB][PS]
  double* dinv = a + PB * PS;                        // [PB], then 3 x 64 slots: panel-wave scratch (phase A) / dummy store targets (phase B)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;

  load_image(a, A, lda, tid);
  __syncthreads();
  STAMP(1);

  if (factor) {
    long long t_a = 0, t_b = 0, t0 = 0;
    for (int g = 0; g < 8; ++g) {
      const int o = 16 * g;
      const int nrows = PB - o - 16;                       // rows below the diagonal block
      const int npanel = nrows > 48 ? 2 : 1;               // wave 0 takes 48 of them, wave 1 the other (up to) 64
      if (stamps && (tid == 0 || (lane == 0 && stamps[31]))) t0 = (long long)wall_clock64();
      // ---- phase A
      double r[16], x[16];
      if (wave < npanel) {
        panel_step_dpp(a, info, row0, o, lane, o + 64 * wave, wave == 0, r, x);
      } else if (g >= 1) {
        // update waves, one round-robin list of items:
        //   the rest of the trailing update of step g-1 (tiles (I, J), g+1 <= J <= I <= 7), operands fetched one tile ahead
        //   block row g-1 of the inverse
        //   what is final goes back to HBM: rows of block g-1 of L, rows of block g-2 of inv(L) (completed in phase A(g-1)) and
        //   the matching column strip of inv(L)^T -- spread over the panels instead of one burst behind the last one
        const int nuw = 8 - npanel;
        const int nt = 7 - g, ntile = nt * (nt + 1) / 2;
        int item = update_tiles_pipelined(a, wave - npanel, nuw, ntile, g + 1, o - 16, fr, fk) - ntile;
        for (; item < g - 1; item += nuw) inv_row_tile(a, dinv, g - 1, item, fr, fk);
        item -= g - 1;
        for (; item < 2; item += nuw) store_L_rows8(a, A, lda, 16 * (g - 1) + 8 * item, lane);
        item -= 2;
        if (g >= 2) {
          for (; item < 2; item += nuw) store_inv_rows8(a, dinv, Linv, 16 * (g - 2) + 8 * item, lane);
          item -= 2;
          if (LinvT) for (; item < 2; item += nuw) store_invT_strip64(a, dinv, LinvT, g - 2, 64 * item, lane);
        }
      }
      if (stamps && lane == 0 && stamps[31]) stamps[32 + 8 * g + wave] = (long long)wall_clock64() - t0;   // diagnostics (stamps[31] != 0): per-wave end of phase A
      __syncthreads();
      if (stamps && tid == 0) { const long long t1 = (long long)wall_clock64(); t_a += t1 - t0; t0 = t1; }
      // ---- phase B: the factored diagonal block and its inverse go back to the image (wave 0), block column g+1
      // of the trailing update of step g (neither reads the other's data)
      if (wave == 0) {
        // row o+c of the image = [ L[c][0..c] (lane 16+c: its copy of the block) | X[c+1..15][c] (lane c: column c of
        // inv(L_gg), transposed) ]; masked-out elements go to a per-lane dummy slot instead of a divergent branch per element
        const int c = lane & 15;
        const bool inv_l = lane < 16, act = lane < 32;
        double dc = x[0];                      // X[c][c] = 1 / L_cc (the pivot's refined reciprocal square root)
#pragma unroll
        for (int k = 1; k < 16; ++k) dc = (k == c) ? x[k] : dc;
        if (inv_l) dinv[o + c] = dc;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const bool w = act && (inv_l ? (k > c) : (k <= c));
          a[w ? (o + c) * PS + o + k : PB * PS + PB + lane] = inv_l ? x[k] : r[k];
        }
      }
      else {
        const int I = g + wave;                                  // waves 1..7: tiles (g+1 .. 7, g+1)
        if (I < 8) update_tile(a, I, g + 1, o, fr, fk);
      }
      __syncthreads();
      if (stamps && tid == 0) t_b += (long long)wall_clock64() - t0;
    }
    if (stamps && tid == 0) { stamps[16] = t_a; stamps[17] = t_b; stamps[18] = 0; }
    STAMP(2);
    // rows 112..127 of L
    for (int idx = 112 * 64 + tid; idx < PB * PB / 2; idx += NT) {
      const int i = idx >> 6, j = (idx & 63) * 2;
      double2 v;
      v.x = (j <= i) ? a[i * PS + j] : 0.0;
      v.y = (j + 1 <= i) ? a[i * PS + j + 1] : 0.0;
      *reinterpret_cast<double2*>(A + (i64)i * lda + j) = v;
    }
    STAMP(3);
    // ---- what is left of the inverse: block row 7
    STAMP(4);
    if (wave < 7) inv_row_tile(a, dinv, 7, wave, fr, fk);
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
  // inverse (and its transpose) to HBM, explicit zeros above the diagonal, 16-byte stores.  With factor != 0 the block rows
  // 0..5 (and the matching column strips of the transpose) left in the panels' shadow: only block rows 6 and 7 remain
  {
    const int b0 = factor ? 6 : 0, i0 = 16 * b0, w2 = (PB - i0) / 2;      // w2: double2 per row of the transposed strip
    for (int idx = i0 * 64 + tid; idx < PB * PB / 2; idx += NT) {
      const int i = idx >> 6, c = (idx & 63) * 2;
      double2 v;
      v.x = x_elem(a, dinv, i, c);
      v.y = x_elem(a, dinv, i, c + 1);
      *reinterpret_cast<double2*>(Linv + i * PB + c) = v;
    }
    if (LinvT) {
      for (int idx = tid; idx < PB * w2; idx += NT) {
        const int c = idx / w2, i = i0 + 2 * (idx - c * w2);        // LinvT[c][i] = X[i][c]
        double2 v;
        v.x = x_elem(a, dinv, i, c);
        v.y = x_elem(a, dinv, i + 1, c);
        *reinterpret_cast<double2*>(LinvT + c * PB + i) = v;
      }
    }
  }
  __syncthreads();
  STAMP(6);
#undef STAMP
}

int gps_launch_potrf_base(gps_handle_t h, double* A, i64 lda, double* Linv_blk,
                          double* LinvT_blk, int* d_info, i64 row0, int factor, long long* d_stamps) {
  const size_t lds = (size_t)(PB * PS + PB + 192) * sizeof(double);   // image, dinv, 3 x 64 scratch / dummy slots
  int rc0 = gps_dyn_lds(h, reinterpret_cast<const void*>(&potrf_base_kernel), (int)lds);
  if (rc0) return rc0;
  // potrf n^3/3 + trtri n^3/3
  LaunchScope ls(h, KC_POTRF_BASE, 2.0 * PB * PB * PB / 3.0, 3.0 * PB * PB * 8.0);
  hipLaunchKernelGGL(potrf_base_kernel, dim3(1), dim3(NT), lds, h->stream, A, lda, Linv_blk,
                     LinvT_blk, d_info, (int)row0, factor, d_stamps);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}
