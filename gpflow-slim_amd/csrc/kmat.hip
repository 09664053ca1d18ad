// Kernel-matrix build: K[i][j] = k(x_i, x'_j) for a whole kernel *program* in one pass.
//
// Replaces, fused into one HBM write of K:
//   Kernel._slice            kernels.py:217-253   (active-dims gather, folded into the prep)
//   Stationary.square_dist   kernels.py:408-421   (x/l ; -2 a.b + |a|^2 + |b|^2 ; clip >= 0)
//   Stationary.euclid_dist   kernels.py:424-426   (sqrt(r2 + 1e-12))
//   RBF / Matern12/32/52 / Exponential .K         kernels.py:436-439, 560-610
//   Periodic.K               kernels.py:806-819   (via u = (cos, sin) features: the docstring's
//                                                  own mapping, kernels.py:776 -- avoids the
//                                                  reference's [N,M,D] temporary and D*N*M sin())
//   White.K / Constant.K     kernels.py:332-350
//   Sum.K / Product.K        kernels.py:1071-1084 (RPN program, left folds)
//   "+ eye(N) * variance"    models/gpr.py:69,120 ; features.py:76 (diag_add)
//
// Two kernels: a tiny prep pass turns X[n, d_all] into feature-major rows
// Ft[feature][point] (scaled inputs x_d / l_d, or cos/sin(2 pi x_d / p)) plus squared norms;
// the tile pass gives every 256-thread workgroup a 64x64 tile, stages the two 64-point feature
// slabs of one primitive at a time in LDS, accumulates a 4x4 patch of dot products per thread
// and pushes k() onto a register-resident evaluation stack.  HBM traffic = one coalesced write
// of K (512-byte row segments per wave) + O((n+m) F) feature reads.
#include "gps_common.hpp"
#include <cmath>

#define KT 64          // tile edge
#define KLS 64         // LDS slab stride (doubles) per feature row
#define KMAXF 64       // features per primitive (32 dims x {cos, sin})

struct PrepFeat { int dim; int kind; double param; };   // kind 0: x/param ; 1: cos(2 pi x/param) ; 2: sin(...)
struct PrepNorm { int f0; int nf; };

struct KNodeDev {
  int op; int f0; int nf; int norm_row;   // norm_row: row index in Ft holding |a|^2 (or -1)
  double variance; double c0;             // c0: periodic lengthscale
  double c1;                              // periodic: -1 / (4 c0^2)
};
struct KProgDev {
  int n_nodes;
  KNodeDev nodes[GPS_MAX_NODES];
};

struct KmatArgs {
  const double* Fr; const double* Fc;     // feature-major [rows][ldf_r / ldf_c]
  i64 ldfr, ldfc;
  double* K; i64 ldk;
  i64 n, m;                               // real extents
  int sym; int lower_only; int identity_pad; int maxnf;
  int tri_grid = 0;   // (kmat_single_kernel) 1-D grid over the tiles of the lower triangle only
  i64 row_off, col_off;                   // global index of local row / column 0 (sub-block builds)
  double diag_add;
};

// ---- prep: one thread per point ------------------------------------------------------------
// The feature / norm tables either as device arrays (uploaded through the pinned ring) or, when they are small, by value in
// the kernel arguments: no copy command in front of the launch (two of them were 14 us of a 200 us small-N evaluation).
#define PREP_SMALL_F 24
#define PREP_SMALL_N 8
struct PrepTabPtr { const PrepFeat* f; const PrepNorm* n; };
struct PrepTabVal { PrepFeat f[PREP_SMALL_F]; PrepNorm n[PREP_SMALL_N]; };

template <class Tab>
__device__ __forceinline__ void kmat_prep_body(const double* __restrict__ X, i64 n, i64 d_all, i64 npts_pad, const Tab& tab, int nfeat,
                                               int nnorm, double* __restrict__ Ft, i64 ldf) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts_pad) return;
  if (i >= n) {
    for (int f = 0; f < nfeat + nnorm; ++f) Ft[(i64)f * ldf + i] = 0.0;
    return;
  }
  const double* x = X + i * d_all;
  for (int f = 0; f < nfeat; ++f) {
    const PrepFeat pf = tab.f[f];
    const double xv = x[pf.dim];
    double v;
    if (pf.kind == 0) v = xv / pf.param;
    else {
      const double ang = 2.0 * M_PI * xv / pf.param;
      v = (pf.kind == 1) ? cos(ang) : sin(ang);
    }
    Ft[(i64)f * ldf + i] = v;
  }
  for (int q = 0; q < nnorm; ++q) {
    const PrepNorm pn = tab.n[q];
    double s = 0.0;
    for (int f = pn.f0; f < pn.f0 + pn.nf; ++f) {
      const double v = Ft[(i64)f * ldf + i];
      s += v * v;
    }
    Ft[(i64)(nfeat + q) * ldf + i] = s;
  }
}
__global__ __launch_bounds__(256) void kmat_prep_kernel(const double* __restrict__ X, i64 n,
                                                        i64 d_all, i64 npts_pad,
                                                        const PrepFeat* __restrict__ feats,
                                                        int nfeat,
                                                        const PrepNorm* __restrict__ norms,
                                                        int nnorm, double* __restrict__ Ft,
                                                        i64 ldf) {
  const PrepTabPtr tab{feats, norms};
  kmat_prep_body(X, n, d_all, npts_pad, tab, nfeat, nnorm, Ft, ldf);
}
__global__ __launch_bounds__(256) void kmat_prep_args_kernel(const double* __restrict__ X, i64 n, i64 d_all, i64 npts_pad, PrepTabVal tab,
                                                             int nfeat, int nnorm, double* __restrict__ Ft, i64 ldf) {
  kmat_prep_body(X, n, d_all, npts_pad, tab, nfeat, nnorm, Ft, ldf);
}

// ---- exp(x) for x <= 0 (every kernel-matrix formula of kernels.py:436-439, 560-610, 813-819 has a non-positive argument up
// to rounding): rint reduction, degree-13 polynomial, v_ldexp_f64 -- no overflow branch, <= 1 ulp-class error.  One shared
// path for all four tile kernels.
// The coefficients live in constant memory: the loads are uniform, so they sit in SGPRs and every Horner step is ONE
// v_fma_f64 with a scalar operand.  (As 64-bit literals the compiler materialised each of them with two v_mov_b32 in front of
// a v_fmac -- 330 of the ~1000 VALU instructions a thread issued for its 16 entries; the kernels are VALU-issue-bound.)
__constant__ double gps_exp_tab[16] = {
    1.6059043836821613e-10,   // 1/13!
    2.08767569878681e-09,     // 1/12!
    2.505210838544172e-08,    // 1/11!
    2.755731922398589e-07,    // 1/10!
    2.7557319223985893e-06,   // 1/9!
    2.48015873015873e-05,     // 1/8!
    1.984126984126984e-04,    // 1/7!
    1.3888888888888889e-03,   // 1/6!
    8.333333333333333e-03,    // 1/5!
    4.1666666666666664e-02,   // 1/4!
    1.6666666666666666e-01,   // 1/3!
    0.5, 1.0,
    1.4426950408889634,            // [13] log2(e)
    6.93147180369123816490e-01,    // [14] ln2 high
    1.90821492927058770002e-10};   // [15] ln2 low
// max(r2, 0) of kernels.py:421 the way tf.maximum does it: a NaN argument stays NaN (fmax would return the 0)
__device__ __forceinline__ double gps_clamp0(double r2) { return (r2 < 0.0) ? 0.0 : r2; }
struct ExpTab { double c[16]; };
__device__ __forceinline__ ExpTab gps_exp_load() {
  ExpTab t;
#pragma unroll
  for (int i = 0; i < 16; ++i) t.c[i] = gps_exp_tab[i];
  return t;
}
__device__ __forceinline__ double gps_exp_nonpos(double x, const ExpTab& t) {        // exp(x) for x <= 0, <= 1 ulp-class error
#ifndef GPS_EXP_LDEXP
  // k = round(x log2 e) by adding 1.5 * 2^52: the sum's low mantissa bits ARE the integer k (two's complement), and 2^k is
  // applied by adding k to the exponent field of the polynomial's value -- no v_rndne_f64 / v_cvt_i32_f64 / v_ldexp_f64 (each
  // a multi-cycle fp64 instruction; the kernel-matrix kernels are VALU-issue-bound).  Arguments below -708 (results below
  // 2.2e-308, where exp() would go through the denormals) give 0.
  const double magic = 6755399441055744.0;
  const double kd = fma(fmax(x, -1100.0), t.c[13], magic);
  const int ki = __double2loint(kd);
  const double k = kd - magic;
#else
  const double k = rint(x * t.c[13]);
#endif
  double r = fma(-k, t.c[14], x);
  r = fma(-k, t.c[15], r);
  // Taylor to degree 13 on |r| <= ln2 / 2 (remainder 4e-18), Horner
  double p = t.c[0];
#pragma unroll
  for (int i = 1; i <= 10; ++i) p = fma(p, r, t.c[i]);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
#ifndef GPS_EXP_LDEXP
  const int hi = __double2hiint(p) + (ki << 20);         // p in [0.70, 1.42]: the exponent field cannot wrap for k >= -1021
  const double v = __hiloint2double(hi, __double2loint(p));
  // (a NaN argument -- NaN in X or in a hyper-parameter -- is clamped to -1100 above: hand it through, as tf.exp and the
  // small-N path's exp() do, instead of returning 0 and a finite kernel matrix)
  return ki < -1021 ? (x == x ? 0.0 : x) : v;
#else
  return ldexp(p, (int)k);                          // k >= -1075: gradual underflow to 0 like exp()
#endif
}
// sqrt(x) for x in [1e-12, huge): v_rsq_f64 and one coupled Newton step + a final correction (the library sqrt adds range
// scaling and special cases around the same core)
__device__ __forceinline__ double gps_sqrt_pos(double x) {
#ifndef GPS_SQRT_LIB
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, hh = 0.5 * y;
  const double e = fma(-hh, g, 0.5);
  g = fma(g, e, g); hh = fma(hh, e, hh);
  const double d = fma(-g, g, x);
  g = fma(d, hh, g);
  const double d2 = fma(-g, g, x);
  return fma(d2, hh, g);
#else
  return sqrt(x);
#endif
}
// The same for the epilogue of kmat_mfma_kernel, where every VALU instruction counts (round 5).  A NaN argument is handed
// through by a compare and a select on the result (round 6: the round-5 form, one FMA x * 0 + result, also turned an
// argument of -infinity -- a squared distance that overflowed -- into NaN where tf.exp and the other kernel-matrix kernels
// give 0: tests/test_gpu_parity.py::test_overflowed_distances_give_zero_not_nan).
__device__ __forceinline__ double gps_exp_nonpos_lean(double x, const ExpTab& t) {
  const double magic = 6755399441055744.0;
  const double kd = fma(fmax(x, -1100.0), t.c[13], magic);
  const int ki = __double2loint(kd);
  const double k = kd - magic;
  double r = fma(-k, t.c[14], x);
  r = fma(-k, t.c[15], r);
  double p = t.c[0];
#pragma unroll
  for (int i = 1; i <= 10; ++i) p = fma(p, r, t.c[i]);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const int hi = __double2hiint(p) + (ki << 20);
  const double v = __hiloint2double(hi, __double2loint(p));
  const double z = ki < -1021 ? 0.0 : v;
  return x != x ? x : z;
}
__device__ __forceinline__ double gps_exp_nonpos(double x) {        // (kernels that call it a few times only)
  const ExpTab t = gps_exp_load();
  return gps_exp_nonpos(x, t);
}

// ---- tile pass --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kmat_tile_kernel(KmatArgs a, KProgDev P) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (a.lower_only && ((a.col_off >> 6) + tj) >> 1 > ((a.row_off >> 6) + ti) >> 1) return;   // 128-granular: diagonal blocks stay full

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* nr_s = reinterpret_cast<double*>(smem_raw);      // [KT]
  double* nc_s = nr_s + KT;                                // [KT]
  double* Fr_s = nc_s + KT;                                // [a.maxnf][KLS]
  double* Fc_s = Fr_s + a.maxnf * KLS;                     // [a.maxnf][KLS]

  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const i64 gi0 = (i64)ti * KT, gj0 = (i64)tj * KT;

  double st[GPS_MAX_STACK][16];
#pragma unroll
  for (int s = 0; s < GPS_MAX_STACK; ++s)
#pragma unroll
    for (int e = 0; e < 16; ++e) st[s][e] = 0.0;

  for (int nd = 0; nd < P.n_nodes; ++nd) {
    const KNodeDev node = P.nodes[nd];
    if (node.op == GPS_K_ADD || node.op == GPS_K_MUL) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        st[0][e] = (node.op == GPS_K_ADD) ? (st[1][e] + st[0][e]) : (st[1][e] * st[0][e]);
#pragma unroll
        for (int s = 1; s < GPS_MAX_STACK - 1; ++s) st[s][e] = st[s + 1][e];
      }
      continue;
    }
    // ---- primitive: value v[16] ----
    double v[16];
    if (node.op == GPS_K_CONSTANT) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = node.variance;
    } else if (node.op == GPS_K_WHITE) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const i64 gi = a.row_off + gi0 + ty * 4 + (e >> 2), gj = a.col_off + gj0 + tx * 4 + (e & 3);
        v[e] = (a.sym && gi == gj) ? node.variance : 0.0;
      }
    } else {
      // stage this primitive's feature slabs
      __syncthreads();
      for (int idx = tid; idx < node.nf * KT; idx += 256) {
        const int f = idx >> 6, p = idx & 63;
        Fr_s[f * KLS + p] = a.Fr[(i64)(node.f0 + f) * a.ldfr + gi0 + p];
        Fc_s[f * KLS + p] = a.Fc[(i64)(node.f0 + f) * a.ldfc + gj0 + p];
      }
      if (node.norm_row >= 0 && tid < KT) {
        nr_s[tid] = a.Fr[(i64)node.norm_row * a.ldfr + gi0 + tid];
        nc_s[tid] = a.Fc[(i64)node.norm_row * a.ldfc + gj0 + tid];
      }
      __syncthreads();
      double dot[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) dot[e] = 0.0;
      for (int f = 0; f < node.nf; ++f) {
        double fr[4], fc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          fr[q] = Fr_s[f * KLS + ty * 4 + q];
          fc[q] = Fc_s[f * KLS + tx * 4 + q];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) dot[e] = fma(fr[e >> 2], fc[e & 3], dot[e]);
      }
      if (node.op == GPS_K_PERIODIC) {
        const double half_d = 0.5 * (double)(node.nf / 2);
        const double l2 = node.c0 * node.c0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          // sum_d sin^2(pi (x - x')/p) / l^2 = (D - sum_d cos(a_d - b_d)) / (2 l^2)
          const double rs = (half_d - 0.5 * dot[e]) / l2;
          v[e] = node.variance * gps_exp_nonpos(-0.5 * rs);
        }
      } else {
        const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const double ni = nr_s[ty * 4 + (e >> 2)], nj = nc_s[tx * 4 + (e & 3)];
          double r2 = -2.0 * dot[e] + (ni + nj);
          r2 = gps_clamp0(r2);
          double val;
          if (node.op == GPS_K_RBF) {
            val = node.variance * gps_exp_nonpos(-r2 / 2.0);
          } else if (node.op == GPS_K_SQDIST) {
            val = node.variance * r2;                                    // kernels.py:408-421 as a callable
          } else {
            const double r = gps_sqrt_pos(r2 + 1e-12);
            if (node.op == GPS_K_MATERN12) val = node.variance * gps_exp_nonpos(-r);
            else if (node.op == GPS_K_EUCLID) val = node.variance * r;    // kernels.py:424-426
            else if (node.op == GPS_K_EXPONENTIAL) val = node.variance * gps_exp_nonpos(-0.5 * r);
            else if (node.op == GPS_K_MATERN32) val = node.variance * (1.0 + sq3 * r) * gps_exp_nonpos(-sq3 * r);
            else val = node.variance * (1.0 + sq5 * r + 5.0 / 3.0 * (r * r)) * gps_exp_nonpos(-sq5 * r);
          }
          v[e] = val;
        }
      }
    }
    // push
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int s = GPS_MAX_STACK - 1; s > 0; --s) st[s][e] = st[s - 1][e];
      st[0][e] = v[e];
    }
  }

  // ---- write the 4x4 patch ----
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const i64 li = gi0 + ty * 4 + q;
    const i64 gi = a.row_off + li;
    double o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const i64 gj = a.col_off + gj0 + tx * 4 + c;
      double val = st[0][q * 4 + c];
      if (gi >= a.n || gj >= a.m) val = (a.identity_pad && gi == gj) ? 1.0 : 0.0;
      else if (a.sym && gi == gj) val += a.diag_add;
      o[c] = val;
    }
    double* dst = a.K + li * a.ldk + gj0 + tx * 4;
    *reinterpret_cast<double2*>(dst) = make_double2(o[0], o[1]);
    *reinterpret_cast<double2*>(dst + 2) = make_double2(o[2], o[3]);
  }
}


// ---- single stationary primitive: the program of RBF / Matern GPR (BASELINE configs 1-3, 5) -----------------------
// The interpreter above keeps a four-deep register stack of 16 entries per thread (251 VGPRs, two waves per SIMD) and
// walks the program node by node; for the one-node programs that carry the headline workloads that is all overhead, and
// the kernel was VALU-latency-bound at 2.4 TB/s of stores (a store-only kernel with the same access pattern reaches
// 5.2 TB/s, tools/store_bw.hip).  Same tile, same 4x4 patch, same epilogue -- no stack, no node loop, and the
// exponential in line (arguments are <= 0: no overflow branch; rint reduction, degree-13 polynomial, v_ldexp_f64),
// which brings the kernel under 128 VGPRs (four waves per SIMD).
template <int OP>
__global__ __launch_bounds__(256, 4) void kmat_single_kernel(KmatArgs a, KNodeDev node) {
  int ti, tj;
  if (a.tri_grid) {
    // lower triangle of a square matrix: only the needed tiles are launched (round 6; the rectangular grid started as many
    // workgroups again that returned at once).  128-granular (diagonal blocks stay full): block q of the row-major lower
    // triangle of 128-blocks, 64 x 64 tile s of it.
    const unsigned q = blockIdx.x >> 2, sub = blockIdx.x & 3;
    unsigned bi = (unsigned)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= q) ++bi;
    while (bi * (bi + 1) / 2 > q) --bi;
    const unsigned bj = q - bi * (bi + 1) / 2;
    ti = (int)(2 * bi + (sub >> 1)); tj = (int)(2 * bj + (sub & 1));
  } else {
    ti = blockIdx.y; tj = blockIdx.x;
    if (a.lower_only && ((a.col_off >> 6) + tj) >> 1 > ((a.row_off >> 6) + ti) >> 1) return;   // 128-granular: diagonal blocks stay full
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* nr_s = reinterpret_cast<double*>(smem_raw);      // [KT]
  double* nc_s = nr_s + KT;                                // [KT]
  double* Fr_s = nc_s + KT;                                // [nf][KLS]
  double* Fc_s = Fr_s + node.nf * KLS;                     // [nf][KLS]
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const i64 gi0 = (i64)ti * KT, gj0 = (i64)tj * KT;
  for (int idx = tid; idx < node.nf * KT; idx += 256) {
    const int f = idx >> 6, p = idx & 63;
    Fr_s[f * KLS + p] = a.Fr[(i64)(node.f0 + f) * a.ldfr + gi0 + p];
    Fc_s[f * KLS + p] = a.Fc[(i64)(node.f0 + f) * a.ldfc + gj0 + p];
  }
  if (tid < KT) {
    nr_s[tid] = a.Fr[(i64)node.norm_row * a.ldfr + gi0 + tid];
    nc_s[tid] = a.Fc[(i64)node.norm_row * a.ldfc + gj0 + tid];
  }
  __syncthreads();
  // the thread's 4 x 4 patch: rows 4 ty .. 4 ty + 3, columns {2 tx, 2 tx + 1, 32 + 2 tx, 33 + 2 tx} -- the sixteen lanes of a
  // row then store 256 contiguous bytes per instruction (whole 128-byte lines; columns 4 tx .. 4 tx + 3 made every store
  // instruction write every other 16 bytes of a row: round 6)
  const int c_lo = 2 * tx, c_hi = 32 + 2 * tx;
  double dot[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) dot[e] = 0.0;
  for (int f = 0; f < node.nf; ++f) {
    const double2 r01 = *reinterpret_cast<const double2*>(Fr_s + f * KLS + ty * 4), r23 = *reinterpret_cast<const double2*>(Fr_s + f * KLS + ty * 4 + 2);
    const double2 c01 = *reinterpret_cast<const double2*>(Fc_s + f * KLS + c_lo), c23 = *reinterpret_cast<const double2*>(Fc_s + f * KLS + c_hi);
    const double fr[4] = {r01.x, r01.y, r23.x, r23.y}, fc[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
    for (int e = 0; e < 16; ++e) dot[e] = fma(fr[e >> 2], fc[e & 3], dot[e]);
  }
  const double2 n01 = *reinterpret_cast<const double2*>(nc_s + c_lo), n23 = *reinterpret_cast<const double2*>(nc_s + c_hi);
  const double ncol[4] = {n01.x, n01.y, n23.x, n23.y};
  const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const i64 li = gi0 + ty * 4 + q;
    const i64 gi = a.row_off + li;
    const double ni = nr_s[ty * 4 + q];
    double o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const i64 gj = a.col_off + gj0 + (c < 2 ? c_lo + c : c_hi + c - 2);
      double r2 = -2.0 * dot[q * 4 + c] + (ni + ncol[c]);        // kernels.py:409-421, same op order as the interpreter
      r2 = gps_clamp0(r2);
      double val;
      if (OP == GPS_K_RBF) {
        val = node.variance * gps_exp_nonpos(-r2 / 2.0);
      } else {
        const double r = gps_sqrt_pos(r2 + 1e-12);
        if (OP == GPS_K_MATERN12) val = node.variance * gps_exp_nonpos(-r);
        else if (OP == GPS_K_EXPONENTIAL) val = node.variance * gps_exp_nonpos(-0.5 * r);
        else if (OP == GPS_K_MATERN32) val = node.variance * (1.0 + sq3 * r) * gps_exp_nonpos(-sq3 * r);
        else val = node.variance * (1.0 + sq5 * r + 5.0 / 3.0 * (r * r)) * gps_exp_nonpos(-sq5 * r);
      }
      if (gi >= a.n || gj >= a.m) val = (a.identity_pad && gi == gj) ? 1.0 : 0.0;
      else if (a.sym && gi == gj) val += a.diag_add;
      o[c] = val;
    }
    double* dst = a.K + li * a.ldk + gj0;
    *reinterpret_cast<double2*>(dst + c_lo) = make_double2(o[0], o[1]);
    *reinterpret_cast<double2*>(dst + c_hi) = make_double2(o[2], o[3]);
  }
}

template <int OP>
static int launch_single(gps_handle_t h, const KmatArgs& a, const KNodeDev& node, i64 prow, i64 pcol, double tiles) {
  const size_t lds = (size_t)(2 * KT + 2 * node.nf * KLS) * sizeof(double);
  int rcl = gps_dyn_lds(h, reinterpret_cast<const void*>(&kmat_single_kernel<OP>), (int)((2 * KT + 2 * KMAXF * KLS) * sizeof(double)));
  if (rcl) return rcl;
  LaunchScope ls(h, KC_KMAT, tiles * KT * KT * (2.0 * node.nf + 30.0), tiles * KT * KT * 8.0);
  KmatArgs at = a;
  at.tri_grid = (a.lower_only && a.row_off == 0 && a.col_off == 0 && prow == pcol && prow % 128 == 0 && prow / 128 < 40000) ? 1 : 0;
  if (at.tri_grid) {
    const i64 nb = prow / 128;
    hipLaunchKernelGGL(kmat_single_kernel<OP>, dim3((unsigned)(2 * nb * (nb + 1))), dim3(256), lds, h->stream, at, node);
  } else {
    hipLaunchKernelGGL(kmat_single_kernel<OP>, dim3((unsigned)(pcol / KT), (unsigned)(prow / KT)), dim3(256), lds, h->stream, at, node);
  }
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// ---- left-deep chains: p0 (p_i op_i)*  -- Sum / Product of a few primitives (BASELINE config 4: Matern-5/2 + Periodic) --
// kernels.py:1071-1084 folds its operands left to right, so the RPN program of a Sum / Product of primitives is
// "p0 p1 op p2 op ...": at no point more than one intermediate value is alive.  The interpreter nevertheless carries its
// four-deep stack with run-time pushes and pops (251 VGPRs, two waves per SIMD, 0.85 TB/s on config 4); here the one
// running value is an accumulator, the primitives use the in-line exponential of the single-primitive kernel (every
// argument is <= 0 up to rounding), and the kernel fits four waves per SIMD.  Same tile, staging, formulas and operand
// order (older op newer) as kmat_tile_kernel.
__global__ __launch_bounds__(256, 2) void kmat_chain_kernel(KmatArgs a, KProgDev P) {      // (2: no scratch; with 3 the kernel spilled 20 VGPRs -- it is the fall-back behind kmat_mfma_kernel)
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (a.lower_only && ((a.col_off >> 6) + tj) >> 1 > ((a.row_off >> 6) + ti) >> 1) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* nr_s = reinterpret_cast<double*>(smem_raw);      // [KT]
  double* nc_s = nr_s + KT;                                // [KT]
  double* Fr_s = nc_s + KT;                                // [a.maxnf][KLS]
  double* Fc_s = Fr_s + a.maxnf * KLS;                     // [a.maxnf][KLS]
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const i64 gi0 = (i64)ti * KT, gj0 = (i64)tj * KT;
  double acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0;
  for (int nd = 0; nd < P.n_nodes; nd += (nd == 0 ? 1 : 2)) {
    const KNodeDev node = P.nodes[nd];
    const int op = (nd == 0) ? -1 : P.nodes[nd + 1].op;
    double v[16];
    if (node.op == GPS_K_CONSTANT) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = node.variance;
    } else if (node.op == GPS_K_WHITE) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const i64 gi = a.row_off + gi0 + ty * 4 + (e >> 2), gj = a.col_off + gj0 + tx * 4 + (e & 3);
        v[e] = (a.sym && gi == gj) ? node.variance : 0.0;
      }
    } else {
      __syncthreads();
      for (int idx = tid; idx < node.nf * KT; idx += 256) {
        const int f = idx >> 6, p = idx & 63;
        Fr_s[f * KLS + p] = a.Fr[(i64)(node.f0 + f) * a.ldfr + gi0 + p];
        Fc_s[f * KLS + p] = a.Fc[(i64)(node.f0 + f) * a.ldfc + gj0 + p];
      }
      if (node.norm_row >= 0 && tid < KT) {
        nr_s[tid] = a.Fr[(i64)node.norm_row * a.ldfr + gi0 + tid];
        nc_s[tid] = a.Fc[(i64)node.norm_row * a.ldfc + gj0 + tid];
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = 0.0;               // the dot products
      for (int f = 0; f < node.nf; ++f) {
        const double2 r01 = *reinterpret_cast<const double2*>(Fr_s + f * KLS + ty * 4), r23 = *reinterpret_cast<const double2*>(Fr_s + f * KLS + ty * 4 + 2);
        const double2 c01 = *reinterpret_cast<const double2*>(Fc_s + f * KLS + tx * 4), c23 = *reinterpret_cast<const double2*>(Fc_s + f * KLS + tx * 4 + 2);
        const double fr[4] = {r01.x, r01.y, r23.x, r23.y}, fc[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = fma(fr[e >> 2], fc[e & 3], v[e]);
      }
      if (node.op == GPS_K_PERIODIC) {
        const double half_d = 0.5 * (double)(node.nf / 2);
        const double l2 = node.c0 * node.c0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const double rs = (half_d - 0.5 * v[e]) / l2;        // kernels.py:605-610 in cos / sin feature form
          v[e] = node.variance * gps_exp_nonpos(-0.5 * rs);
        }
      } else {
        const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const double ni = nr_s[ty * 4 + (e >> 2)], nj = nc_s[tx * 4 + (e & 3)];
          double r2 = -2.0 * v[e] + (ni + nj);
          r2 = gps_clamp0(r2);
          double val;
          if (node.op == GPS_K_RBF) {
            val = node.variance * gps_exp_nonpos(-r2 / 2.0);
          } else {
            const double r = gps_sqrt_pos(r2 + 1e-12);
            if (node.op == GPS_K_MATERN12) val = node.variance * gps_exp_nonpos(-r);
            else if (node.op == GPS_K_EXPONENTIAL) val = node.variance * gps_exp_nonpos(-0.5 * r);
            else if (node.op == GPS_K_MATERN32) val = node.variance * (1.0 + sq3 * r) * gps_exp_nonpos(-sq3 * r);
            else val = node.variance * (1.0 + sq5 * r + 5.0 / 3.0 * (r * r)) * gps_exp_nonpos(-sq5 * r);
          }
          v[e] = val;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = (op < 0) ? v[e] : ((op == GPS_K_ADD) ? (acc[e] + v[e]) : (acc[e] * v[e]));
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const i64 li = gi0 + ty * 4 + q;
    const i64 gi = a.row_off + li;
    double o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const i64 gj = a.col_off + gj0 + tx * 4 + c;
      double val = acc[q * 4 + c];
      if (gi >= a.n || gj >= a.m) val = (a.identity_pad && gi == gj) ? 1.0 : 0.0;
      else if (a.sym && gi == gj) val += a.diag_add;
      o[c] = val;
    }
    double* dst = a.K + li * a.ldk + gj0 + tx * 4;
    *reinterpret_cast<double2*>(dst) = make_double2(o[0], o[1]);
    *reinterpret_cast<double2*>(dst + 2) = make_double2(o[2], o[3]);
  }
}


// ---- the dot products on the matrix pipe ------------------------------------------------------------------------------
// kernels.py:408-421 is r2 = |a|^2 + |b|^2 - 2 a.b with a.b a [64 x F] x [F x 64] product per tile -- a dense contraction, and
// for Sum / Product programs (config 4: Matern-5/2 + Periodic, 16 + 32 features) the 48 multiply-adds per entry were what
// the fp64 VALU spent most of its time on beside the exponentials (kmat_chain_kernel: 0.95 TB/s of stores).  Here the
// features stay feature-major in LDS ([F][KLM]: exactly the K-major operand layout of v_mfma_f64_16x16x4), wave w of the four
// owns rows 16w .. 16w+15 of the tile and all 64 columns (four 16 x 16 accumulators), and the VALU is left with the
// per-entry epilogue (norms, clamp, sqrt, the in-line exponential) while another wave's products run on the matrix pipe.
// Row stride KLM = 80 doubles: lanes 16-31 of a fragment read (feature k0 + 1) land on the other half of the banks.
// The exact diagonal of a symmetric build is r2 = 0 by construction (x_i == x_j; the VALU kernels got that from using the
// same FMA chain for norms and products, the matrix pipe sums in another order) -- kernels.py:409-421 leaves "whatever
// rounding leaves, clamped at 0" there.
#define KLM 80
typedef double v4d_k __attribute__((ext_vector_type(4)));

// (no diagonal test per entry: the accumulators of the diagonal entries of a diagonal tile are patched before the epilogue
// to the dot product that makes r2 -- or the Periodic argument -- exactly 0; see the kernel)
// The value as TWO factors, pre (variance x polynomial) and e (the exponential): the fold into the running result is then an
// explicit fma(pre, e, run) / run * (pre * e) -- one instruction less per entry, and no mul + add pair left for the compiler to
// contract at some unrolled positions and not at others (K[i][j] and K[j][i] are computed at different positions of different
// lanes: with the contraction left to the compiler a Sum kernel came out 1 ulp asymmetric).
template <int OP>
__device__ __forceinline__ void kmat_prim_from_dot(const KNodeDev& node, int op, double dot, double nsum, const ExpTab& et, double& pre, double& e) {
  const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
  const int o = (OP >= 0) ? OP : op;      // (always a compile-time constant at the call sites)
  pre = node.variance;
  if (o == GPS_K_PERIODIC) {
    // kernels.py:813-819 in cos / sin feature form: sum_d sin^2(pi (x - x') / p) / l^2 = (D - sum_d cos(a_d - b_d)) / (2 l^2);
    // node.c1 = -1 / (4 l^2), folded on the host (a division per entry is ~12 VALU instructions)
    e = gps_exp_nonpos_lean(((double)(node.nf / 2) - dot) * node.c1, et);       // (no clamp: kernels.py:817-819)
    return;
  }
  const double r2 = gps_clamp0(fma(-2.0, dot, nsum));                                    // kernels.py:409-421 (NaN stays NaN)
  if (o == GPS_K_RBF) { e = gps_exp_nonpos_lean(-0.5 * r2, et); return; }
  const double r = gps_sqrt_pos(r2 + 1e-12);
  if (o == GPS_K_MATERN12) { e = gps_exp_nonpos_lean(-r, et); return; }
  if (o == GPS_K_EXPONENTIAL) { e = gps_exp_nonpos_lean(-0.5 * r, et); return; }
  // variance (1 + sqrt(3) r) and variance (1 + sqrt(5) r + 5/3 r^2) in Horner form on wave-uniform coefficients
  if (o == GPS_K_MATERN32) { pre = fma(node.variance * sq3, r, node.variance); e = gps_exp_nonpos_lean(-sq3 * r, et); return; }
  pre = fma(fma(node.variance * (5.0 / 3.0), r, node.variance * sq5), r, node.variance);
  e = gps_exp_nonpos_lean(-sq5 * r, et);
}

// OP >= 0: the program is that one stationary primitive (compile-time formulas); OP < 0: a left-deep chain p0 (p_i op_i)*
//
// Operand roles are swapped and the column operand is permuted so that the accumulator layout suits the STORES: the
// instruction computes D = A B^T with D[(lane >> 4) + 4 r][lane & 15] in acc[r].  With A = the tile's COLUMN features, read
// through pi(q) = 4 (q & 3) + (q >> 2), and B = the ROW features, lane (l15, kq) ends up with acc[j][r] = entry (row 16 w + l15,
// column 16 j + 4 kq + r): four consecutive doubles of one row per 16-column block -- two 16-byte stores, and the four kq
// lanes of a row cover 128 contiguous bytes (the VALU kernels' store shape; the natural layout -- four ROWS per lane --
// needs 8-byte stores and measured 8 % slower than the VALU kernel on the store-bound RBF build).
template <int OP>
__global__ __launch_bounds__(256, 3) void kmat_mfma_kernel(KmatArgs a, KProgDev P) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (a.lower_only && ((a.col_off >> 6) + tj) >> 1 > ((a.row_off >> 6) + ti) >> 1) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* nr_s = reinterpret_cast<double*>(smem_raw);      // [KT]
  double* nc_s = nr_s + KT;                                // [KT]
  double* Fr_s = nc_s + KT;                                // [maxnf4][KLM]
  double* Fc_s = Fr_s + a.maxnf * KLM;                     // [maxnf4][KLM]   (a.maxnf: already a multiple of 4 here)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int pi = 4 * (l15 & 3) + (l15 >> 2);
  const int row = 16 * w + l15;                            // this lane's tile row; its columns: 16 j + 4 kq + r
  const i64 gi0 = (i64)ti * KT, gj0 = (i64)tj * KT;
  const bool diag_tile = a.sym && (a.row_off + gi0 == a.col_off + gj0);
  const ExpTab et = gps_exp_load();
  double run[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) run[e] = 0.0;
  const int n_nodes = (OP >= 0) ? 1 : P.n_nodes;
  for (int nd = 0; nd < n_nodes; nd += (nd == 0 ? 1 : 2)) {
    const KNodeDev node = P.nodes[nd];
    // the lane's entry on the diagonal of a diagonal tile, as an index into its 16 (row == 16 j + 4 kq + r: j == w, kq == l15 >> 2,
    // r == l15 & 3), or -1.  (Opaque to the optimiser per node: sixteen hoisted lane masks cost 32 scalar registers.)
    int dce = (diag_tile && kq == (l15 >> 2)) ? 4 * w + (l15 & 3) : -1;
    asm volatile("" : "+v"(dce));
    // the first value is ADDED to the zeros the running result starts from: two ways to fold, not three
    const bool mul = (nd > 0) && P.nodes[nd + 1].op == GPS_K_MUL;
    // every value is folded into the running result at once (no 16-entry temporary: the kernel sits at three waves per
    // SIMD because of its registers)
#define KMAT_FOLD(e, val) run[e] = mul ? (run[e] * (val)) : (run[e] + (val))
    if (OP < 0 && node.op == GPS_K_CONSTANT) {
#pragma unroll
      for (int e = 0; e < 16; ++e) KMAT_FOLD(e, node.variance);
    } else if (OP < 0 && node.op == GPS_K_WHITE) {
#pragma unroll
      for (int e = 0; e < 16; ++e) KMAT_FOLD(e, (e == dce) ? node.variance : 0.0);
    } else {
      const int nf4 = (node.nf + 3) & ~3;
      if (nd > 0) __syncthreads();
      for (int idx = tid; idx < nf4 * KT; idx += 256) {
        const int f = idx >> 6, p = idx & 63;
        const bool real = f < node.nf;
        Fr_s[f * KLM + p] = real ? a.Fr[(i64)(node.f0 + f) * a.ldfr + gi0 + p] : 0.0;
        Fc_s[f * KLM + p] = real ? a.Fc[(i64)(node.f0 + f) * a.ldfc + gj0 + p] : 0.0;
      }
      if (node.norm_row >= 0 && tid < KT) {
        nr_s[tid] = a.Fr[(i64)node.norm_row * a.ldfr + gi0 + tid];
        nc_s[tid] = a.Fc[(i64)node.norm_row * a.ldfc + gj0 + tid];
      }
      __syncthreads();
      v4d_k acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = (v4d_k){0.0, 0.0, 0.0, 0.0};
      const double* rp = Fr_s + kq * KLM + row;            // B operand: the wave's 16 rows
      const double* cp = Fc_s + kq * KLM + pi;             // A operand: 16 columns of block j, permuted
      for (int k0 = 0; k0 < nf4; k0 += 4) {
        const double bv = rp[k0 * KLM];
        const double a0 = cp[k0 * KLM], a1 = cp[k0 * KLM + 16], a2 = cp[k0 * KLM + 32], a3 = cp[k0 * KLM + 48];
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bv, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, bv, acc[3], 0, 0, 0);
      }
      const bool has_norm = node.norm_row >= 0;
      const double ni = has_norm ? nr_s[row] : 0.0;
      if (diag_tile) {
        // the exact diagonal (x_i == x_j: r2 = 0, Periodic argument 0 -- NaN / infinite coordinates stay NaN).  Stationary:
        // dot = (n_i + n_i) / 2 makes -2 dot + (n_i + n_j) exactly 0; Periodic: dot = D.  Here, once per node of a diagonal
        // tile, instead of a test and two selects per entry.
        const double dv = has_norm ? ni : (double)(node.nf / 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * j + r == dce) acc[j][r] = fma(acc[j][r], 0.0, dv);
      }
      // the primitive's formula and the fold are chosen ONCE per node, outside the 16 entries: every inner loop is
      // branch-free (two independent chains at a time for the scheduler to interleave)
#define KMAT_EVAL2(OPC, MUL)                                                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
        double ns[4] = {0.0, 0.0, 0.0, 0.0};                                                                    \
        if (has_norm) {                                                                                         \
          const double2 n01 = *reinterpret_cast<const double2*>(nc_s + 16 * j + 4 * kq);                        \
          const double2 n23 = *reinterpret_cast<const double2*>(nc_s + 16 * j + 4 * kq + 2);                    \
          ns[0] = ni + n01.x; ns[1] = ni + n01.y; ns[2] = ni + n23.x; ns[3] = ni + n23.y;                       \
        }                                                                                                       \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                         \
          double pre, ex;                                                                                       \
          kmat_prim_from_dot<OPC>(node, OPC, acc[j][r], ns[r], et, pre, ex);                                    \
          run[j * 4 + r] = MUL ? run[j * 4 + r] * (pre * ex) : fma(pre, ex, run[j * 4 + r]);                    \
          if (r & 1) __builtin_amdgcn_sched_barrier(0);   /* two chains at a time: four or more interleaved spill (128 registers) */ \
        }                                                                                                       \
      }
#define KMAT_EVAL(OPC) if (mul) { KMAT_EVAL2(OPC, true) } else { KMAT_EVAL2(OPC, false) }
      if (OP >= 0) { KMAT_EVAL2(OP, false) }
      else switch (node.op) {
        case GPS_K_RBF: KMAT_EVAL(GPS_K_RBF) break;
        case GPS_K_MATERN12: KMAT_EVAL(GPS_K_MATERN12) break;
        case GPS_K_MATERN32: KMAT_EVAL(GPS_K_MATERN32) break;
        case GPS_K_MATERN52: KMAT_EVAL(GPS_K_MATERN52) break;
        case GPS_K_EXPONENTIAL: KMAT_EVAL(GPS_K_EXPONENTIAL) break;
        default: KMAT_EVAL(GPS_K_PERIODIC) break;
      }
#undef KMAT_EVAL
#undef KMAT_EVAL2
    }
#undef KMAT_FOLD
  }
  // ---- store: 4 consecutive doubles per (lane, j)
  const i64 li = gi0 + row;
  double* dst = a.K + li * a.ldk + gj0 + 4 * kq;
  if (!diag_tile && a.row_off + gi0 + KT <= a.n && a.col_off + gj0 + KT <= a.m) {
    // interior tile (almost all of them): no padding, no diagonal -- no per-entry tests
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<double2*>(dst + 16 * j) = make_double2(run[j * 4], run[j * 4 + 1]);
      *reinterpret_cast<double2*>(dst + 16 * j + 2) = make_double2(run[j * 4 + 2], run[j * 4 + 3]);
    }
    return;
  }
  const i64 gi = a.row_off + li;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const i64 gj = a.col_off + gj0 + 16 * j + 4 * kq + r;
      double val = run[j * 4 + r];
      if (gi >= a.n || gj >= a.m) val = (a.identity_pad && gi == gj) ? 1.0 : 0.0;
      else if (a.sym && gi == gj) val += a.diag_add;
      o[r] = val;
    }
    *reinterpret_cast<double2*>(dst + 16 * j) = make_double2(o[0], o[1]);
    *reinterpret_cast<double2*>(dst + 16 * j + 2) = make_double2(o[2], o[3]);
  }
}

template <int OP>
static int launch_mfma(gps_handle_t h, KmatArgs a, const KProgDev& P, i64 prow, i64 pcol, double tiles, int nfeat_total) {
  int maxnf = 4;
  for (int i = 0; i < P.n_nodes; ++i) if (((P.nodes[i].nf + 3) & ~3) > maxnf) maxnf = (P.nodes[i].nf + 3) & ~3;
  a.maxnf = maxnf;
  const size_t lds = (size_t)(2 * KT + 2 * maxnf * KLM) * sizeof(double);
  int rcl = gps_dyn_lds(h, reinterpret_cast<const void*>(&kmat_mfma_kernel<OP>), (int)((2 * KT + 2 * KMAXF * KLM) * sizeof(double)));
  if (rcl) return rcl;
  LaunchScope ls(h, KC_KMAT, tiles * KT * KT * (2.0 * nfeat_total + 30.0 * ((P.n_nodes + 1) / 2)), tiles * KT * KT * 8.0);
  hipLaunchKernelGGL(kmat_mfma_kernel<OP>, dim3((unsigned)(pcol / KT), (unsigned)(prow / KT)), dim3(256), lds, h->stream, a, P);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// p0 (p_i op_i)* with op in {ADD, MUL}
static bool is_left_deep_chain(const KProgDev& P) {
  if (P.n_nodes < 3 || (P.n_nodes & 1) == 0) return false;
  // (the distance ops are evaluated by the interpreter only: they are helpers, not kernels of a model)
  auto prim = [](int op) { return op != GPS_K_ADD && op != GPS_K_MUL && op != GPS_K_SQDIST && op != GPS_K_EUCLID && op < GPS_K_NKN_LINROW; };
  if (!prim(P.nodes[0].op)) return false;
  for (int i = 1; i < P.n_nodes; i += 2) {
    if (!prim(P.nodes[i].op)) return false;
    if (P.nodes[i + 1].op != GPS_K_ADD && P.nodes[i + 1].op != GPS_K_MUL) return false;
  }
  return true;
}

// ---- Neural Kernel Network epilogue ---------------------------------------------------------------
// neural_kernel_network/neural_kernel_network.py:41-47 stacks the primitive kernel values of every
// (i, j) entry into a vector and pushes it through Linear (positive weights) / Product / Activation
// layers (neural_kernel_network_wrapper.py:90-173): a pure per-entry epilogue.  Here: all primitives'
// feature slabs of a 64x64 tile sit in LDS at once, every thread walks its 4x4 patch one entry at a
// time, evaluates up to 8 primitives and runs the network in registers (layer width <= 16, weights
// zero-padded to 16x16 in LDS so the inner loops are branch-free).
#define NKN_MAXP 8
#define NKN_W 16
#define NKN_MAXL 8
struct NknLayer { int type; int in_dim; int out_dim; int step; int act; int w_off; };   // type 0 linear, 1 product, 2 act
struct NknNet { int n_layers; int n_prims; int ftot; int nnorm; NknLayer layers[NKN_MAXL]; };

__device__ __forceinline__ double prim_value(const KNodeDev& node, double dot, double ni, double nj) {
  if (node.op == GPS_K_PERIODIC) {
    const double l2 = node.c0 * node.c0;
    const double rs = (0.5 * (double)(node.nf / 2) - 0.5 * dot) / l2;
    return node.variance * gps_exp_nonpos(-0.5 * rs);
  }
  const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
  double r2 = -2.0 * dot + (ni + nj);
  r2 = gps_clamp0(r2);
  if (node.op == GPS_K_RBF) return node.variance * gps_exp_nonpos(-r2 / 2.0);
  const double r = gps_sqrt_pos(r2 + 1e-12);
  if (node.op == GPS_K_MATERN12) return node.variance * gps_exp_nonpos(-r);
  if (node.op == GPS_K_EXPONENTIAL) return node.variance * gps_exp_nonpos(-0.5 * r);
  if (node.op == GPS_K_MATERN32) return node.variance * (1.0 + sq3 * r) * gps_exp_nonpos(-sq3 * r);
  return node.variance * (1.0 + sq5 * r + 5.0 / 3.0 * (r * r)) * gps_exp_nonpos(-sq5 * r);
}

__global__ __launch_bounds__(256) void nkn_tile_kernel(KmatArgs a, KProgDev P, NknNet net,
                                                       const double* __restrict__ Wg) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (a.lower_only && ((a.col_off >> 6) + tj) >> 1 > ((a.row_off >> 6) + ti) >> 1) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int rows_f = net.ftot + net.nnorm;                 // feature rows incl. the norm rows
  double* Fr_s = reinterpret_cast<double*>(smem_raw);      // [rows_f][64]
  double* Fc_s = Fr_s + rows_f * KT;                       // [rows_f][64]
  double* W_s = Fc_s + rows_f * KT;                        // [n_layers][NKN_W][NKN_W + 1] (last col = bias)
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const i64 gi0 = (i64)ti * KT, gj0 = (i64)tj * KT;
  for (int idx = tid; idx < rows_f * KT; idx += 256) {
    const int f = idx >> 6, p = idx & 63;
    Fr_s[idx] = a.Fr[(i64)f * a.ldfr + gi0 + p];
    Fc_s[idx] = a.Fc[(i64)f * a.ldfc + gj0 + p];
  }
  for (int idx = tid; idx < net.n_layers * NKN_W * (NKN_W + 1); idx += 256) W_s[idx] = Wg[idx];
  __syncthreads();

  for (int e = 0; e < 16; ++e) {
    const int li = ty * 4 + (e >> 2), lj = tx * 4 + (e & 3);
    const i64 gi = a.row_off + gi0 + li, gj = a.col_off + gj0 + lj;
    double vin[NKN_W], vout[NKN_W];
#pragma unroll
    for (int q = 0; q < NKN_W; ++q) vin[q] = 0.0;
    // primitives (program order = input order of the first layer)
#pragma unroll
    for (int p = 0; p < NKN_MAXP; ++p) {
      if (p < net.n_prims) {
        const KNodeDev node = P.nodes[p];
        double val;
        if (node.op == GPS_K_CONSTANT) val = node.variance;
        else if (node.op == GPS_K_WHITE) val = (a.sym && gi == gj) ? node.variance : 0.0;
        else {
          double dot = 0.0;
          for (int f = 0; f < node.nf; ++f) dot = fma(Fr_s[(node.f0 + f) * KT + li], Fc_s[(node.f0 + f) * KT + lj], dot);
          double ni = 0.0, nj = 0.0;
          if (node.norm_row >= 0) { ni = Fr_s[node.norm_row * KT + li]; nj = Fc_s[node.norm_row * KT + lj]; }
          val = prim_value(node, dot, ni, nj);
        }
        vin[p] = val;
      }
    }
    for (int L = 0; L < net.n_layers; ++L) {
      const NknLayer ly = net.layers[L];
      if (ly.type == 0) {
        const double* Wl = W_s + L * NKN_W * (NKN_W + 1);
#pragma unroll
        for (int o = 0; o < NKN_W; ++o) {
          double acc = Wl[o * (NKN_W + 1) + NKN_W];
#pragma unroll
          for (int j = 0; j < NKN_W; ++j) acc = fma(Wl[o * (NKN_W + 1) + j], vin[j], acc);
          vout[o] = acc;
        }
      } else if (ly.type == 1) {
#pragma unroll
        for (int o = 0; o < NKN_W; ++o) vout[o] = 0.0;
        if (ly.step == 2) {
#pragma unroll
          for (int o = 0; o < NKN_W / 2; ++o) vout[o] = vin[2 * o] * vin[2 * o + 1];
        } else if (ly.step == 3) {
#pragma unroll
          for (int o = 0; o < NKN_W / 3; ++o) vout[o] = vin[3 * o] * vin[3 * o + 1] * vin[3 * o + 2];
        } else {
#pragma unroll
          for (int o = 0; o < NKN_W / 4; ++o) vout[o] = (vin[4 * o] * vin[4 * o + 1]) * (vin[4 * o + 2] * vin[4 * o + 3]);
        }
      } else {
#pragma unroll
        for (int o = 0; o < NKN_W; ++o) vout[o] = exp(vin[o]);
      }
#pragma unroll
      for (int q = 0; q < NKN_W; ++q) vin[q] = vout[q];
    }
    double val = vin[0];
    if (gi >= a.n || gj >= a.m) val = (a.identity_pad && gi == gj) ? 1.0 : 0.0;
    else if (a.sym && gi == gj) val += a.diag_add;
    a.K[(gi0 + li) * a.ldk + gj0 + lj] = val;
  }
}

// ---- host side ---------------------------------------------------------------------------------
struct KCompiled {
  KProgDev prog;
  std::vector<PrepFeat> feats;
  std::vector<PrepNorm> norms;
  bool has_nkn = false;
  NknNet net;
  std::vector<double> W;       // [n_layers][16][17]
};

static int compile_prog(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, i64 d_all,
                        KCompiled& out) {
  if (!prog || n_nodes <= 0 || n_nodes > GPS_MAX_NODES)
    return gps_fail(h, GPS_ERR_ARG, "kernel program: 1..GPS_MAX_NODES nodes expected");
  int depth = 0;
  out.prog.n_nodes = n_nodes;
  for (int i = 0; i < n_nodes; ++i)
    if (prog[i].op >= GPS_K_NKN_LINROW) out.has_nkn = true;
  int n_prim_nodes = n_nodes;
  if (out.has_nkn) {
    n_prim_nodes = 0;
    while (n_prim_nodes < n_nodes && prog[n_prim_nodes].op < GPS_K_ADD) ++n_prim_nodes;
    if (n_prim_nodes == 0 || n_prim_nodes > NKN_MAXP)
      return gps_fail(h, GPS_ERR_UNSUPPORTED, "NKN program: 1..8 primitive kernels first");
    for (int i = 0; i < n_prim_nodes; ++i)
      if (prog[i].op == GPS_K_SQDIST || prog[i].op == GPS_K_EUCLID)
        return gps_fail(h, GPS_ERR_UNSUPPORTED, "NKN program: the distance ops are not kernels");
    for (int i = n_prim_nodes; i < n_nodes; ++i)
      if (prog[i].op < GPS_K_NKN_LINROW) return gps_fail(h, GPS_ERR_ARG, "NKN program: layers must follow the primitives");
    out.prog.n_nodes = n_prim_nodes;
  }
  for (int i = 0; i < n_prim_nodes; ++i) {
    const gps_kern_node_t& nd = prog[i];
    KNodeDev& kd = out.prog.nodes[i];
    kd.op = nd.op; kd.f0 = 0; kd.nf = 0; kd.norm_row = -1; kd.variance = nd.variance; kd.c0 = 0.0; kd.c1 = 0.0;
    switch (nd.op) {
      case GPS_K_ADD: case GPS_K_MUL:
        if (depth < 2) return gps_fail(h, GPS_ERR_ARG, "kernel program: stack underflow");
        depth -= 1;
        break;
      case GPS_K_WHITE: case GPS_K_CONSTANT:
        depth += 1;
        break;
      case GPS_K_RBF: case GPS_K_MATERN12: case GPS_K_MATERN32: case GPS_K_MATERN52:
      case GPS_K_EXPONENTIAL: case GPS_K_PERIODIC: case GPS_K_SQDIST: case GPS_K_EUCLID: {
        if (nd.n_dims <= 0 || nd.n_dims > GPS_MAX_DIMS)
          return gps_fail(h, GPS_ERR_ARG, "kernel program: n_dims out of range");
        kd.f0 = (int)out.feats.size();
        for (int d = 0; d < nd.n_dims; ++d) {
          if (nd.active_dims[d] < 0 || nd.active_dims[d] >= d_all)
            return gps_fail(h, GPS_ERR_ARG, "kernel program: active dim outside X");
        }
        if (nd.op == GPS_K_PERIODIC) {
          if (!(nd.period > 0.0) || !(nd.lengthscales[0] > 0.0))
            return gps_fail(h, GPS_ERR_ARG, "kernel program: period / lengthscale must be positive");
          for (int d = 0; d < nd.n_dims; ++d) {
            out.feats.push_back({nd.active_dims[d], 1, nd.period});
            out.feats.push_back({nd.active_dims[d], 2, nd.period});
          }
          kd.nf = 2 * nd.n_dims;
          kd.c0 = nd.lengthscales[0];
          kd.c1 = -1.0 / (4.0 * kd.c0 * kd.c0);
        } else {
          for (int d = 0; d < nd.n_dims; ++d) {
            if (!(nd.lengthscales[d] > 0.0))
              return gps_fail(h, GPS_ERR_ARG, "kernel program: lengthscale must be positive");
            out.feats.push_back({nd.active_dims[d], 0, nd.lengthscales[d]});
          }
          kd.nf = nd.n_dims;
          kd.norm_row = (int)out.norms.size();     // fixed up below (offset by total features)
          out.norms.push_back({kd.f0, kd.nf});
        }
        depth += 1;
        break;
      }
      default:
        return gps_fail(h, GPS_ERR_UNSUPPORTED, "kernel program: unknown op");
    }
    if (!out.has_nkn && depth > GPS_MAX_STACK)
      return gps_fail(h, GPS_ERR_UNSUPPORTED, "kernel program: expression deeper than GPS_MAX_STACK");
  }
  if (!out.has_nkn && depth != 1) return gps_fail(h, GPS_ERR_ARG, "kernel program: must leave exactly one value");
  const int nfeat = (int)out.feats.size();
  for (int i = 0; i < n_prim_nodes; ++i)
    if (out.prog.nodes[i].norm_row >= 0) out.prog.nodes[i].norm_row += nfeat;
  if (!out.has_nkn) return GPS_OK;
  // ---- layers
  NknNet& net = out.net;
  net.n_layers = 0; net.n_prims = n_prim_nodes; net.ftot = nfeat; net.nnorm = (int)out.norms.size();
  int width = n_prim_nodes;
  int i = n_prim_nodes;
  while (i < n_nodes) {
    if (net.n_layers >= NKN_MAXL) return gps_fail(h, GPS_ERR_UNSUPPORTED, "NKN program: more than 8 layers");
    NknLayer& ly = net.layers[net.n_layers];
    const int lid = prog[i].active_dims[0];
    out.W.resize((size_t)(net.n_layers + 1) * NKN_W * (NKN_W + 1), 0.0);
    double* Wl = out.W.data() + (size_t)net.n_layers * NKN_W * (NKN_W + 1);
    ly.w_off = net.n_layers * NKN_W * (NKN_W + 1); ly.step = 0; ly.act = 0;
    if (prog[i].op == GPS_K_NKN_LINROW) {
      int o = 0;
      while (i < n_nodes && prog[i].op == GPS_K_NKN_LINROW && prog[i].active_dims[0] == lid) {
        if (prog[i].n_dims != width || o >= NKN_W) return gps_fail(h, GPS_ERR_ARG, "NKN program: Linear layer width mismatch (max 16)");
        for (int j = 0; j < width; ++j) Wl[o * (NKN_W + 1) + j] = prog[i].lengthscales[j];
        Wl[o * (NKN_W + 1) + NKN_W] = prog[i].variance;
        ++o; ++i;
      }
      ly.type = 0; ly.in_dim = width; ly.out_dim = o; width = o;
    } else if (prog[i].op == GPS_K_NKN_PRODUCT) {
      const int step = prog[i].n_dims;
      if (step < 2 || step > 4 || width % step) return gps_fail(h, GPS_ERR_UNSUPPORTED, "NKN program: Product step must be 2, 3 or 4 and divide the width");
      ly.type = 1; ly.in_dim = width; ly.step = step; ly.out_dim = width / step; width = ly.out_dim; ++i;
    } else if (prog[i].op == GPS_K_NKN_ACT) {
      if (prog[i].period != 1.0) return gps_fail(h, GPS_ERR_UNSUPPORTED, "NKN program: only the exp activation is available");
      ly.type = 2; ly.in_dim = width; ly.out_dim = width; ly.act = 1; ++i;
    } else return gps_fail(h, GPS_ERR_ARG, "NKN program: unknown layer op");
    ++net.n_layers;
  }
  if (width != 1) return gps_fail(h, GPS_ERR_ARG, "NKN program: the network must end with one output");
  if (nfeat + (int)out.norms.size() > 2 * KMAXF) return gps_fail(h, GPS_ERR_UNSUPPORTED, "NKN program: too many features");
  return GPS_OK;
}

// launch the tile pass (plain program or NKN) for prepared features
static int launch_tiles(gps_handle_t h, const KCompiled& kc, KmatArgs& a, i64 prow, i64 pcol, int n_nodes,
                        double tiles) {
  const int nfeat_total = (int)kc.feats.size();
  if (kc.has_nkn) {
    const size_t wbytes = kc.W.size() * 8;
    GPS_HIP(h, h->dNkn.ensure(wbytes + 64));
    GPS_HIP(h, h->ring.upload(h->dNkn.p, kc.W.data(), wbytes, h->stream));
    const int rows_f = kc.net.ftot + kc.net.nnorm;
    const size_t lds = ((size_t)2 * rows_f * KT + (size_t)kc.net.n_layers * NKN_W * (NKN_W + 1)) * 8;
    int rcl = gps_dyn_lds(h, reinterpret_cast<const void*>(&nkn_tile_kernel), 160 * 1024);
    if (rcl) return rcl;
    if (lds > 160 * 1024) return gps_fail(h, GPS_ERR_UNSUPPORTED, "NKN program: feature slabs exceed LDS");
    LaunchScope ls(h, KC_KMAT, tiles * KT * KT * (2.0 * nfeat_total + 600.0 * kc.net.n_layers), tiles * KT * KT * 8.0);
    hipLaunchKernelGGL(nkn_tile_kernel, dim3((unsigned)(pcol / KT), (unsigned)(prow / KT)), dim3(256), lds, h->stream,
                       a, kc.prog, kc.net, (const double*)h->dNkn.p);
    GPS_HIP(h, hipGetLastError());
    return GPS_OK;
  }
  // One stationary primitive: the VALU kernel is store-bound already (RBF, N = 32768: 0.99 ms = 4.3 TB/s either way; the
  // matrix-pipe version measured 0-4 % slower), so the matrix-pipe kernel is used for it only on request ("kmat_mfma" = 2).
  if (kc.prog.n_nodes == 1 && kc.prog.nodes[0].norm_row >= 0 && h->kmat_fast && h->kmat_mfma >= 2) {
    switch (kc.prog.nodes[0].op) {
      case GPS_K_RBF: return launch_mfma<GPS_K_RBF>(h, a, kc.prog, prow, pcol, tiles, nfeat_total);
      case GPS_K_MATERN12: return launch_mfma<GPS_K_MATERN12>(h, a, kc.prog, prow, pcol, tiles, nfeat_total);
      case GPS_K_MATERN32: return launch_mfma<GPS_K_MATERN32>(h, a, kc.prog, prow, pcol, tiles, nfeat_total);
      case GPS_K_MATERN52: return launch_mfma<GPS_K_MATERN52>(h, a, kc.prog, prow, pcol, tiles, nfeat_total);
      case GPS_K_EXPONENTIAL: return launch_mfma<GPS_K_EXPONENTIAL>(h, a, kc.prog, prow, pcol, tiles, nfeat_total);
      default: break;
    }
  }
  if (h->kmat_fast && h->kmat_mfma && is_left_deep_chain(kc.prog))                     // Sum / Product of primitives, same
    return launch_mfma<-1>(h, a, kc.prog, prow, pcol, tiles, nfeat_total);
  if (kc.prog.n_nodes == 1 && kc.prog.nodes[0].norm_row >= 0 && h->kmat_fast) {      // one stationary primitive
    const KNodeDev& nd = kc.prog.nodes[0];
    switch (nd.op) {
      case GPS_K_RBF: return launch_single<GPS_K_RBF>(h, a, nd, prow, pcol, tiles);
      case GPS_K_MATERN12: return launch_single<GPS_K_MATERN12>(h, a, nd, prow, pcol, tiles);
      case GPS_K_MATERN32: return launch_single<GPS_K_MATERN32>(h, a, nd, prow, pcol, tiles);
      case GPS_K_MATERN52: return launch_single<GPS_K_MATERN52>(h, a, nd, prow, pcol, tiles);
      case GPS_K_EXPONENTIAL: return launch_single<GPS_K_EXPONENTIAL>(h, a, nd, prow, pcol, tiles);
      default: break;
    }
  }
  int maxnf = 1;
  for (int i = 0; i < kc.prog.n_nodes; ++i) if (kc.prog.nodes[i].nf > maxnf) maxnf = kc.prog.nodes[i].nf;
  a.maxnf = maxnf;
  const size_t lds = (size_t)(2 * KT + 2 * maxnf * KLS) * sizeof(double);
  if (h->kmat_fast && is_left_deep_chain(kc.prog)) {                                   // Sum / Product of primitives
    int rcc = gps_dyn_lds(h, reinterpret_cast<const void*>(&kmat_chain_kernel), (int)((2 * KT + 2 * KMAXF * KLS) * sizeof(double)));
    if (rcc) return rcc;
    LaunchScope ls(h, KC_KMAT, tiles * KT * KT * (2.0 * nfeat_total + 30.0 * ((kc.prog.n_nodes + 1) / 2)), tiles * KT * KT * 8.0);
    hipLaunchKernelGGL(kmat_chain_kernel, dim3((unsigned)(pcol / KT), (unsigned)(prow / KT)), dim3(256), lds, h->stream, a, kc.prog);
    GPS_HIP(h, hipGetLastError());
    return GPS_OK;
  }
  int rcl = gps_dyn_lds(h, reinterpret_cast<const void*>(&kmat_tile_kernel), (int)((2 * KT + 2 * KMAXF * KLS) * sizeof(double)));
  if (rcl) return rcl;
  LaunchScope ls(h, KC_KMAT, tiles * KT * KT * (2.0 * nfeat_total + 30.0), tiles * KT * KT * 8.0);
  hipLaunchKernelGGL(kmat_tile_kernel, dim3((unsigned)(pcol / KT), (unsigned)(prow / KT)), dim3(256), lds,
                     h->stream, a, kc.prog);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_kdiag(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double* kdiag_const) {
  // Kdiag of every primitive is its variance (kernels.py:428-429, 803-804, 327-328);
  // Sum.Kdiag / Product.Kdiag fold them (kernels.py:1075-1076, 1083-1084).
  bool nkn = false;
  for (int i = 0; i < n_nodes; ++i) if (prog[i].op >= GPS_K_NKN_LINROW) nkn = true;
  if (nkn) {
    // NeuralKernelNetwork.Kdiag (neural_kernel_network.py:35-39): the primitives' Kdiag through the layers
    KCompiled kc;
    int rc = compile_prog(h, prog, n_nodes, 1 << 20, kc);
    if (rc) return rc;
    double vin[NKN_W] = {0}, vout[NKN_W];
    for (int p = 0; p < kc.net.n_prims; ++p) vin[p] = prog[p].variance;
    for (int L = 0; L < kc.net.n_layers; ++L) {
      const NknLayer& ly = kc.net.layers[L];
      const double* Wl = kc.W.data() + (size_t)L * NKN_W * (NKN_W + 1);
      for (int o = 0; o < NKN_W; ++o) vout[o] = 0.0;
      if (ly.type == 0) {
        for (int o = 0; o < NKN_W; ++o) { double acc = Wl[o * (NKN_W + 1) + NKN_W]; for (int j = 0; j < NKN_W; ++j) acc = fma(Wl[o * (NKN_W + 1) + j], vin[j], acc); vout[o] = acc; }
      } else if (ly.type == 1) {
        for (int o = 0; o < NKN_W / ly.step; ++o) { double pr = 1.0; for (int q = 0; q < ly.step; ++q) pr *= vin[o * ly.step + q]; vout[o] = pr; }
      } else {
        for (int o = 0; o < NKN_W; ++o) vout[o] = exp(vin[o]);
      }
      for (int o = 0; o < NKN_W; ++o) vin[o] = vout[o];
    }
    *kdiag_const = vin[0];
    return GPS_OK;
  }
  double st[GPS_MAX_NODES]; int sp = 0;
  for (int i = 0; i < n_nodes; ++i) {
    if (prog[i].op == GPS_K_ADD || prog[i].op == GPS_K_MUL) {
      if (sp < 2) return gps_fail(h, GPS_ERR_ARG, "kernel program: stack underflow");
      const double b = st[--sp], a = st[--sp];
      st[sp++] = (prog[i].op == GPS_K_ADD) ? a + b : a * b;
    } else st[sp++] = prog[i].variance;
  }
  if (sp != 1) return gps_fail(h, GPS_ERR_ARG, "kernel program: must leave exactly one value");
  *kdiag_const = st[0];
  return GPS_OK;
}

static int run_prep(gps_handle_t h, const KCompiled& kc, const double* dX, i64 n, i64 d_all,
                    i64 npad, DevBuf& feat, DevBuf& tables, i64* ldf_out) {
  const int nfeat = (int)kc.feats.size(), nnorm = (int)kc.norms.size();
  const i64 rows = nfeat + nnorm;
  *ldf_out = npad;
  if (rows == 0) return GPS_OK;
  GPS_HIP(h, feat.ensure((size_t)rows * npad * sizeof(double)));
  if (nfeat <= PREP_SMALL_F && nnorm <= PREP_SMALL_N) {
    PrepTabVal tab;
    memset(&tab, 0, sizeof(tab));
    for (int f = 0; f < nfeat; ++f) tab.f[f] = kc.feats[f];
    for (int q = 0; q < nnorm; ++q) tab.n[q] = kc.norms[q];
    LaunchScope ls(h, KC_KMAT, 0.0, 8.0 * (double)npad * (double)(rows + d_all));
    hipLaunchKernelGGL(kmat_prep_args_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, dX, n, d_all, npad, tab,
                       nfeat, nnorm, feat.d(), npad);
    GPS_HIP(h, hipGetLastError());
    return GPS_OK;
  }
  const size_t fb = (size_t)nfeat * sizeof(PrepFeat), nb = (size_t)nnorm * sizeof(PrepNorm);
  GPS_HIP(h, tables.ensure(fb + nb + 64));
  // (through pinned slots: the host vectors die after return, and the stream is not synchronised)
  if (fb) GPS_HIP(h, h->ring.upload(tables.p, kc.feats.data(), fb, h->stream));
  if (nb) GPS_HIP(h, h->ring.upload((char*)tables.p + fb, kc.norms.data(), nb, h->stream));
  LaunchScope ls(h, KC_KMAT, 0.0, 8.0 * (double)npad * (double)(rows + d_all));
  hipLaunchKernelGGL(kmat_prep_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream,
                     dX, n, d_all, npad, (const PrepFeat*)tables.p,
                     nfeat, (const PrepNorm*)((char*)tables.p + fb), nnorm, feat.d(), npad);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

int gps_launch_kmat(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                    const double* dX, i64 n, const double* dX2, i64 m, i64 d_all,
                    double diag_add, double* dK, i64 ldk, i64 prow, i64 pcol,
                    int lower_only, int identity_pad) {
  KCompiled kc;
  int rc = compile_prog(h, prog, n_nodes, d_all, kc);
  if (rc) return rc;
  const int sym = (dX2 == nullptr);
  if (sym) m = n;
  if (prow % KT || pcol % KT || prow < n || pcol < m)
    return gps_fail(h, GPS_ERR_ARG, "kmat: padded extents must be multiples of 64 covering n, m");
  if (prow / KT > 65535) return gps_fail(h, GPS_ERR_UNSUPPORTED, "kmat: more than 65535*64 rows");
  i64 ldfr = 0, ldfc = 0;
  rc = run_prep(h, kc, dX, n, d_all, prow, h->dFeat, h->dProg, &ldfr);
  if (rc) return rc;
  const double* Fc = h->dFeat.d();
  ldfc = ldfr;
  if (!sym) {
    rc = run_prep(h, kc, dX2, m, d_all, pcol, h->dFeat2, h->dProg, &ldfc);
    if (rc) return rc;
    Fc = h->dFeat2.d();
  } else if (pcol != prow) {
    return gps_fail(h, GPS_ERR_ARG, "kmat: symmetric build needs square padding");
  }
  KmatArgs a;
  a.Fr = h->dFeat.d(); a.Fc = Fc; a.ldfr = ldfr; a.ldfc = ldfc;
  a.K = dK; a.ldk = ldk; a.n = n; a.m = m; a.sym = sym;
  a.lower_only = (sym && lower_only) ? 1 : 0; a.identity_pad = identity_pad; a.diag_add = diag_add;
  a.row_off = 0; a.col_off = 0;
  double tiles = (double)(prow / KT) * (double)(pcol / KT);
  if (a.lower_only) tiles = 0.5 * tiles + 0.5 * (double)(prow / KT);
  a.maxnf = 1;
  return launch_tiles(h, kc, a, prow, pcol, n_nodes, tiles);
}


// Sub-block of the symmetric matrix K(X, X) + diag_add I (padded with identity): rows [r0, r0+nrows),
// columns [c0, c0+ncols) written to dKb (leading dimension ldk).  Used by the block-column
// distributed factorisation, where every rank builds only the column blocks it owns.  Features of the
// whole (padded) point set are prepared once per evaluation (prep != 0) and kept in h->dFeat.
int gps_launch_kmat_block(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* dX,
                          i64 n, i64 d_all, i64 npad, double diag_add, double* dKb, i64 ldk, i64 r0,
                          i64 nrows, i64 c0, i64 ncols, int prep) {
  KCompiled kc;
  int rc = compile_prog(h, prog, n_nodes, d_all, kc);
  if (rc) return rc;
  if (r0 % KT || c0 % KT || nrows % KT || ncols % KT || r0 + nrows > npad || c0 + ncols > npad || nrows <= 0 || ncols <= 0)
    return gps_fail(h, GPS_ERR_ARG, "kmat_block: extents must be multiples of 64 inside the padded matrix");
  if (nrows / KT > 65535) return gps_fail(h, GPS_ERR_UNSUPPORTED, "kmat_block: too many rows");
  i64 ldf = npad;
  if (prep) {
    rc = run_prep(h, kc, dX, n, d_all, npad, h->dFeat, h->dProg, &ldf);
    if (rc) return rc;
  }
  KmatArgs a;
  a.Fr = h->dFeat.d() + r0; a.Fc = h->dFeat.d() + c0; a.ldfr = ldf; a.ldfc = ldf;
  a.K = dKb; a.ldk = ldk; a.n = n; a.m = n; a.sym = 1; a.lower_only = 0; a.identity_pad = 1;
  a.diag_add = diag_add; a.row_off = r0; a.col_off = c0;
  a.maxnf = 1;
  const double tiles = (double)(nrows / KT) * (double)(ncols / KT);
  return launch_tiles(h, kc, a, nrows, ncols, n_nodes, tiles);
}
