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
//   phase A(g)  PANEL waves: each holds the 16 rows of the diagonal block g in lanes 0..15 (every panel wave factors
//               it, redundantly, in registers: v_readlane broadcasts, hardware rsqrt + one third-order correction, no barrier)
//               and rows below it in its other lanes -- those lanes execute the very same instruction stream, which
//               for them IS the forward substitution x L_gg^T = b against the block, so the whole 16-column panel is
//               finished when the diagonal block is.  Lanes 16..31 of wave 0 start from the rows of the identity
//               instead: what they end with is e_c^T L_gg^-T, i.e. column c of inv(L_gg) -- the 16x16 triangular
//               inverse costs no instruction of its own.  (Wave 0: 32 rows below the block, the others 48 each.)
//               UPDATE waves (all others), meanwhile: the rest of the trailing update of step g-1 (16x16 tiles
//               C_IJ -= P_I P_J^T on the fp64 MFMA, J >= g+1) and block row g-1 of the inverse (below).
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

// wave-uniform broadcast of lane `lane`'s fp64 value (lane is a compile-time constant at every call)
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

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

// ---- phase A, panel wave: 16 columns o..o+15.  Lanes 0..15: rows of the diagonal block.  Owner (wave 0): lanes
// 16..31 rows of the identity (-> columns of the block's inverse), lanes 32..63 rows o+16 .. o+47; other panel waves:
// lanes 16..63 rows xrow0 .. xrow0+47 (rows >= 128 are idle lanes).  Branch-free body (one basic block): the scheduler
// can overlap column j's trailing updates with the rsqrt chain of column j+1.
// The rows below go back to the image here; the diagonal block's rows and the inverse's columns stay in r[] (lanes
// 0..31 of the owner) and are written by the caller AFTER the barrier that ends the phase: the other panel waves read
// the same block.
__device__ __forceinline__ void panel_step(double* a, double* scr, int* info, int row0, int o, int lane, int xrow0,
                                           bool owner, double (&r)[16]) {
  const bool is_diag = lane < 16;
  const bool is_inv = owner && lane >= 16 && lane < 32;
  const int row = is_diag ? o + lane : (owner ? o + lane - 16 : xrow0 + lane - 16);
  const bool valid = row < PB;
  const int rr = valid ? row : PB - 1;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const double v = a[rr * PS + o + c];
    r[c] = is_inv ? ((c == lane - 16) ? 1.0 : 0.0) : v;
  }
  // Column j: pivot p_j = L_jj^2, y_j = 1/sqrt(p_j), scale the column, then r[k] -= r[j] * L_kj for k > j (L_kj
  // broadcast from diagonal lane k).  The 16 pivots are one dependent chain; everything else is kept off it:
  //  * The next pivot does not wait for the vector update of column j+1.  With s = r[j] and d = r[j+1] of diagonal
  //    lane j+1 read (v_readlane) BEFORE column j is scaled -- both are complete one column earlier --
  //    p_{j+1} = d - (s y_j)^2 is two uniform operations after y_j (the same mul / fma the lanes execute, so it is
  //    the value lane j+1 ends up with).
  //  * 1/sqrt: hardware estimate y0 (2^-24) and ONE third-order step, e = 1 - p y0^2, y = y0 + y0 e (1/2 + 3/8 e)
  //    (error 5/16 e^3 ~ 2^-70): four dependent operations instead of the six of two Newton steps.
  //  * Only the updates of columns j+1 and j+2 take the v_readlane path.  For k >= j+3 the diagonal lanes publish the
  //    scaled column in an LDS scratch of this wave and every lane reads it back as broadcast ds_read_b128 (two values
  //    per read, no SGPR traffic); those updates are applied one column later.  (Updates of one register commute;
  //    r[k] is complete before it is read for pivot k: its LDS-path updates come from columns <= k-3.)
  //  The vector work is placed in the latency gaps of the pivot chain and the order pinned with sched_barrier: left
  //  alone, the compiler finishes every update of column j before it starts the next pivot.
#define PB_UPD(kk) do { if ((kk) < 16) { const double l_ = readlane_f64(r[j], (kk)); r[(kk)] -= r[j] * l_; } } while (0)
#define PB_LUPD(kk) do { if (j >= 1 && (kk) < 16) r[(kk)] -= r[j - 1] * lq[(kk)]; } while (0)
#define PB_FENCE() __builtin_amdgcn_sched_barrier(0)
  int bad = 0x7fffffff;
  double p = readlane_f64(r[0], 0);            // pivot of the current column (uniform)
  bad = !(p > 0.0) ? 0 : bad;
  double y;                                    // its reciprocal square root
  {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double t = p * y0;
    const double e = __builtin_fma(-t, y0, 1.0);
    y = __builtin_fma(y0 * e, __builtin_fma(0.375, e, 0.5), y0);
  }
  double sd = readlane_f64(r[0], 1), dd = readlane_f64(r[1], 1);     // s, d of the next pivot
  double lq[16];                                  // L_k,j-1 (k >= j+2) of the column whose LDS-path updates are pending
#pragma unroll
  for (int k = 0; k < 16; ++k) lq[k] = 0.0;
#pragma unroll
  for (int j = 0; j < 15; ++j) {
    double ln[16];
    // ---- chain: l = s y_j
    const double lj = sd * y;
    r[j] = (lane == j) ? p * y : r[j] * y;       // rows below: x_j = (a_ij - sum_k<j x_k l_jk) / l_jj
    PB_FENCE();
    // ---- chain: p_{j+1}
    const double pn = __builtin_fma(-lj, lj, dd);
    if (j + 3 < 16) {
      scr[lane] = r[j];          // slots 16..63 are never read: an unconditional store keeps the column loop one basic block
#pragma unroll
      for (int m = (j + 3) / 2; m < 8; ++m) {
        const double2 v = *reinterpret_cast<const double2*>(scr + 2 * m);
        ln[2 * m] = v.x; ln[2 * m + 1] = v.y;
      }
    }
    PB_FENCE();
    // ---- chain: estimate
    double yn = __builtin_amdgcn_rsq(pn);
    PB_LUPD(j + 2);
    PB_UPD(j + 1);
    PB_FENCE();
    // ---- chain: t = p y0
    const double t = pn * yn;
    bad = (!(pn > 0.0) && bad == 0x7fffffff) ? j + 1 : bad;
    PB_UPD(j + 2);
    PB_FENCE();
    // ---- chain: e = 1 - t y0
    const double e = __builtin_fma(-t, yn, 1.0);
    if (j + 2 < 16) { sd = readlane_f64(r[j + 1], j + 2); dd = readlane_f64(r[j + 2], j + 2); }
    PB_LUPD(j + 3); PB_LUPD(j + 4); PB_LUPD(j + 5);
    PB_FENCE();
    // ---- chain: q, y0 e
    const double q = __builtin_fma(0.375, e, 0.5);
    const double ye = yn * e;
    PB_LUPD(j + 6); PB_LUPD(j + 7); PB_LUPD(j + 8); PB_LUPD(j + 9); PB_LUPD(j + 10);
    PB_FENCE();
    // ---- chain: y_{j+1}
    yn = __builtin_fma(ye, q, yn);
    PB_LUPD(j + 11); PB_LUPD(j + 12); PB_LUPD(j + 13); PB_LUPD(j + 14); PB_LUPD(j + 15);
    PB_FENCE();
    y = yn; p = pn;
#pragma unroll
    for (int k = j + 3; k < 16; ++k) lq[k] = ln[k];
  }
  r[15] = (lane == 15) ? p * y : r[15] * y;
#undef PB_UPD
#undef PB_LUPD
#undef PB_FENCE
  if (owner && lane == 0 && bad != 0x7fffffff) atomicMin(info, row0 + o + bad + 1);
  if (!is_diag && !is_inv && valid) {
#pragma unroll
    for (int c = 0; c < 16; ++c) a[row * PS + o + c] = r[c];
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
      const int npanel = nrows > 32 ? 1 + (nrows - 32 + 47) / 48 : 1;   // wave 0 takes 32 of them, every further panel wave 48
      if (stamps && (tid == 0 || (lane == 0 && stamps[31]))) t0 = (long long)wall_clock64();
      // ---- phase A
      double r[16];
      if (wave < npanel) {
        panel_step(a, dinv + PB + 64 * wave, info, row0, o, lane, o + 48 * wave, wave == 0, r);
      } else if (g >= 1) {
        const int nuw = 8 - npanel;
        int item = wave - npanel;
        // rest of the trailing update of step g-1: tiles (I, J), g+1 <= J <= I <= 7
        const int nt = 7 - g, ntile = nt * (nt + 1) / 2;
        for (; item < ntile; item += nuw) {
          int I, J;
          tri_index(item, g + 1, I, J);
          update_tile(a, I, J, o - 16, fr, fk);
        }
        item -= ntile;
        // block row g-1 of the inverse
        for (; item < g - 1; item += nuw) inv_row_tile(a, dinv, g - 1, item, fr, fk);
        // last step: rows 0..111 of L are final and go back to HBM in the last panel's shadow (upper triangle
        // zero-filled, like tf.cholesky), 16 rows per update wave
        if (g == 7) {
          const int r0 = 16 * (wave - 1), r1 = r0 + 16;
#pragma unroll 4
          for (int i = r0; i < r1; ++i) {
            const int j = 2 * lane;
            double2 v;
            v.x = (j <= i) ? a[i * PS + j] : 0.0;
            v.y = (j + 1 <= i) ? a[i * PS + j + 1] : 0.0;
            *reinterpret_cast<double2*>(A + (i64)i * lda + j) = v;
          }
        }
      }
      if (stamps && lane == 0 && stamps[31]) stamps[32 + 8 * g + wave] = (long long)wall_clock64() - t0;   // diagnostics (stamps[31] != 0): per-wave end of phase A
      __syncthreads();
      if (stamps && tid == 0) { const long long t1 = (long long)wall_clock64(); t_a += t1 - t0; t0 = t1; }
      // ---- phase B: the factored diagonal block and its inverse go back to the image (wave 0), block column g+1
      // of the trailing update of step g (neither reads the other's data)
      if (wave == 0) {
        // row o+c of the image = [ L[c][0..c] (lane c) | X[c+1..15][c] (lane 16+c: column c of inv(L_gg), transposed) ];
        // masked-out elements go to a per-lane dummy slot instead of a divergent branch per element
        const int c = lane & 15;
        const bool lo = lane < 16, act = lane < 32;
        double dc = r[0];                      // X[c][c] = 1 / L_cc (the pivot's refined reciprocal square root)
#pragma unroll
        for (int k = 1; k < 16; ++k) dc = (k == c) ? r[k] : dc;
        if (act && !lo) dinv[o + c] = dc;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const bool w = act && (lo ? (k <= c) : (k > c));
          a[w ? (o + c) * PS + o + k : PB * PS + PB + lane] = r[k];
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
