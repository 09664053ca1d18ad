// Gradient of the GPR log-marginal likelihood for the kernel programs csrc/grad.hip does not take: more than four
// primitive kernels, and Neural-Kernel-Network programs (Linear (positive weights, bias) / Product / exp-Activation
// layers over the stacked primitive values; neural_kernel_network_wrapper.py:90-173, neural_kernel_network.py:41-47).
// The reference gets these gradients from TensorFlow autodiff through tf.cholesky and the network
// (examples/gpr.py:53-54); here it is the same fused pass as grad.hip --
//
//     d LML / d theta = sum_{i >= j} c_ij W_ij d k(x_i, x_j) / d theta ,   W = A A^T - R K_y^-1 ,  c = 1 (1/2 on the diagonal)
//
// over the lower triangle of K_y^-1 -- with, per matrix entry, a forward pass through the network that keeps every
// layer's input, a reverse pass that yields  d k / d (layer weights, biases)  and the adjoints of the primitive values,
// and then the primitives' own parameter derivatives exactly as in grad.hip.  One workgroup walks 32x32 tiles, a thread
// owns a 2x2 patch; per-parameter sums are reduced with wave shuffles into LDS slots, 2048 fixed-order partials are
// added on the host (bit-reproducible, no floating-point atomics).
//
// Slot layout (host and device agree on it): primitives first, as in grad.hip ([variance], then one slot per active
// dim, or [lengthscale, period] for Periodic); then, for every Linear layer in network order and every output o:
// [W[o][0 .. in-1], bias[o]].
#include "gps_common.hpp"
#include <cmath>

#define GG_T 32            // tile edge
#define GG_E 4             // elements per thread (2 x 2)
#define GG_MAXP 8          // primitives
#define GG_MAXL 8          // network layers
#define GG_W 16            // layer width
#define GG_MAXSLOT 640
#define GG_MAX_NODES 32
#define GG_MAXF 64         // feature rows of one primitive

struct GGFeat { int dim; int kind; double param; };   // 0: x/param ; 1: cos(2pi x/param) ; 2: sin ; 3: 2pi x/param
struct GGNode { int op; int prim; int f0; int nf; int slot0; int ndims; double variance; double ls0; double period; };
struct GGLayer { int type; int in_dim; int out_dim; int step; int slot0; };       // 0 linear, 1 product, 2 exp
struct GGProg {
  int n_nodes, n_prims, n_slots, nkn, n_layers;
  GGNode nodes[GG_MAX_NODES];
  GGLayer layers[GG_MAXL];
};
struct GGArgs {
  const double* Ft; i64 ldf;            // features of the row points
  const double* Ftc; i64 ldfc;          // features of the column points (== Ft for K(X, X))
  const double* Kinv; i64 ldk;
  const double* A; i64 lda; int r;
  const double* Wnet;                   // [n_layers][GG_W][GG_W + 1] (last column = bias)
  i64 n, npad;
  double* partial;                      // [gridDim.x][GG_MAXSLOT + 1]  (last = noise)
  int tiles;                            // tile rows
  // rect != 0: the vector-Jacobian product of the kernel-matrix build for an arbitrary cotangent:
  //   slots += sum_{i < nr, j < nc} Wd[i][j] d k(xr_i, xc_j) / d theta     (all tiles, weight 1 everywhere)
  // same_points: rows and columns index the same point set (K(Z, Z)): White contributes on i == j
  int rect, tiles_c, same_points;
  const double* Wd; i64 ldw; i64 nr, nc;
};

__device__ __forceinline__ double gg_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void gg_prep_kernel(const double* __restrict__ X, i64 n, i64 d_all, i64 npad,
                                                      const GGFeat* __restrict__ feats, int nfeat,
                                                      double* __restrict__ Ft, i64 ldf) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  for (int f = 0; f < nfeat; ++f) {
    double v = 0.0;
    if (i < n) {
      const GGFeat pf = feats[f];
      const double xv = X[i * d_all + pf.dim];
      if (pf.kind == 0) v = xv / pf.param;
      else {
        const double ang = 2.0 * M_PI * xv / pf.param;
        v = (pf.kind == 1) ? cos(ang) : (pf.kind == 2 ? sin(ang) : ang);
      }
    }
    Ft[(i64)f * ldf + i] = v;
  }
}

// stage the feature rows [f0, f0 + nf) of the tile's rows / columns
__device__ __forceinline__ void gg_stage(const GGArgs& a, int f0, int nf, i64 gi0, i64 gj0, double* Fr_s, double* Fc_s, int tid) {
  __syncthreads();
  for (int idx = tid; idx < nf * GG_T; idx += 256) {
    const int f = idx >> 5, pp = idx & 31;
    Fr_s[f * GG_T + pp] = a.Ft[(i64)(f0 + f) * a.ldf + gi0 + pp];
    Fc_s[f * GG_T + pp] = a.Ftc[(i64)(f0 + f) * a.ldfc + gj0 + pp];
  }
  __syncthreads();
}

// value and (squared distance | periodic sum) of one primitive at the thread's four entries
__device__ __forceinline__ void gg_prim(const GGNode& node, const double* Fr_s, const double* Fc_s, int ty, int tx, i64 gi0,
                                        i64 gj0, bool same_points, double (&val)[GG_E], double (&rr)[GG_E]) {
  if (node.op == GPS_K_CONSTANT) {
#pragma unroll
    for (int e = 0; e < GG_E; ++e) { val[e] = node.variance; rr[e] = 0.0; }
    return;
  }
  if (node.op == GPS_K_WHITE) {
#pragma unroll
    for (int e = 0; e < GG_E; ++e) {
      const i64 i = gi0 + ty * 2 + (e >> 1), j = gj0 + tx * 2 + (e & 1);
      val[e] = (same_points && i == j) ? node.variance : 0.0; rr[e] = 0.0;
    }
    return;
  }
  double acc[GG_E];
#pragma unroll
  for (int e = 0; e < GG_E; ++e) acc[e] = 0.0;
  if (node.op == GPS_K_PERIODIC) {
    for (int f = 0; f < 2 * node.ndims; ++f) {
      const double r0 = Fr_s[f * GG_T + ty * 2], r1 = Fr_s[f * GG_T + ty * 2 + 1];
      const double c0 = Fc_s[f * GG_T + tx * 2], c1 = Fc_s[f * GG_T + tx * 2 + 1];
      acc[0] = fma(r0, c0, acc[0]); acc[1] = fma(r0, c1, acc[1]); acc[2] = fma(r1, c0, acc[2]); acc[3] = fma(r1, c1, acc[3]);
    }
    const double l2 = node.ls0 * node.ls0;
#pragma unroll
    for (int e = 0; e < GG_E; ++e) {
      const double S = 0.5 * ((double)node.ndims - acc[e]);       // sum_d sin^2(pi D_d / p)
      rr[e] = S;
      val[e] = node.variance * exp(-0.5 * S / l2);
    }
    return;
  }
  for (int f = 0; f < node.ndims; ++f) {
    const double r0 = Fr_s[f * GG_T + ty * 2], r1 = Fr_s[f * GG_T + ty * 2 + 1];
    const double c0 = Fc_s[f * GG_T + tx * 2], c1 = Fc_s[f * GG_T + tx * 2 + 1];
    double d;
    d = r0 - c0; acc[0] = fma(d, d, acc[0]); d = r0 - c1; acc[1] = fma(d, d, acc[1]);
    d = r1 - c0; acc[2] = fma(d, d, acc[2]); d = r1 - c1; acc[3] = fma(d, d, acc[3]);
  }
  const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
#pragma unroll
  for (int e = 0; e < GG_E; ++e) {
    const double q2 = acc[e];
    rr[e] = q2;
    double v_;
    if (node.op == GPS_K_RBF) v_ = node.variance * exp(-q2 / 2.0);
    else {
      const double rad = sqrt(q2 + 1e-12);
      if (node.op == GPS_K_MATERN12) v_ = node.variance * exp(-rad);
      else if (node.op == GPS_K_EXPONENTIAL) v_ = node.variance * exp(-0.5 * rad);
      else if (node.op == GPS_K_MATERN32) v_ = node.variance * (1.0 + sq3 * rad) * exp(-sq3 * rad);
      else v_ = node.variance * (1.0 + sq5 * rad + 5.0 / 3.0 * (rad * rad)) * exp(-sq5 * rad);
    }
    val[e] = v_;
  }
}

// plain (Sum / Product) program: d out / d prim_p by forward mode
__device__ __forceinline__ double gg_prog_tangent(const GGProg& P, const double (&pv)[GG_MAXP], int p) {
  double sv[GPS_MAX_STACK], st[GPS_MAX_STACK];
#pragma unroll
  for (int s = 0; s < GPS_MAX_STACK; ++s) { sv[s] = 0.0; st[s] = 0.0; }
  for (int nd = 0; nd < P.n_nodes; ++nd) {
    const int op = P.nodes[nd].op;
    if (op == GPS_K_ADD || op == GPS_K_MUL) {
      const double a = sv[1], ta = st[1], b = sv[0], tb = st[0];
      sv[0] = (op == GPS_K_ADD) ? a + b : a * b;
      st[0] = (op == GPS_K_ADD) ? ta + tb : ta * b + a * tb;
#pragma unroll
      for (int s = 1; s < GPS_MAX_STACK - 1; ++s) { sv[s] = sv[s + 1]; st[s] = st[s + 1]; }
    } else {
      const int q = P.nodes[nd].prim;
      double val = pv[0];
#pragma unroll
      for (int u = 1; u < GG_MAXP; ++u) val = (q == u) ? pv[u] : val;
#pragma unroll
      for (int s = GPS_MAX_STACK - 1; s > 0; --s) { sv[s] = sv[s - 1]; st[s] = st[s - 1]; }
      sv[0] = val; st[0] = (q == p) ? 1.0 : 0.0;
    }
  }
  return st[0];
}

// ---- pass 1: values of all primitives at the thread's four entries of tile (gi0, gj0)
__device__ __forceinline__ void gg_values(const GGArgs& a, const GGProg& P, i64 gi0, i64 gj0, bool same_points, double* Fr_s,
                                          double* Fc_s, int tid, int ty, int tx, double (&pv)[GG_MAXP][GG_E]) {
#pragma unroll
  for (int p = 0; p < GG_MAXP; ++p)
#pragma unroll
    for (int e = 0; e < GG_E; ++e) pv[p][e] = 0.0;
  for (int nd = 0; nd < P.n_nodes; ++nd) {
    const GGNode node = P.nodes[nd];
    if (node.prim < 0) continue;
    if (node.nf > 0) gg_stage(a, node.f0, (node.op == GPS_K_PERIODIC) ? 2 * node.ndims : node.ndims, gi0, gj0, Fr_s, Fc_s, tid);
    double val[GG_E], rr[GG_E];
    gg_prim(node, Fr_s, Fc_s, ty, tx, gi0, gj0, same_points, val, rr);
#pragma unroll
    for (int p = 0; p < GG_MAXP; ++p)
      if (node.prim == p) {
#pragma unroll
        for (int e = 0; e < GG_E; ++e) pv[p][e] = val[e];
      }
  }
}

// ---- pass 2: adjoints  fp[p][e] = w_e d k / d prim_p ; ACCW: the network's weight / bias slots are accumulated on the way
template <bool ACCW>
__device__ __forceinline__ void gg_adjoints(const GGProg& P, const double* W_s, const double (&w)[GG_E],
                                            const double (&pv)[GG_MAXP][GG_E], double (&fp)[GG_MAXP][GG_E],
                                            double (*acc_s)[GG_MAXSLOT + 1], int lane, int wave) {
    if (!P.nkn) {
#pragma unroll
      for (int e = 0; e < GG_E; ++e) {
        double pve[GG_MAXP];
#pragma unroll
        for (int u = 0; u < GG_MAXP; ++u) pve[u] = pv[u][e];
#pragma unroll
        for (int p = 0; p < GG_MAXP; ++p) fp[p][e] = (p < P.n_prims) ? w[e] * gg_prog_tangent(P, pve, p) : 0.0;
      }
    } else {
#pragma unroll 1
      for (int e = 0; e < GG_E; ++e) {
        double act[GG_MAXL + 1][GG_W];                      // act[l] = input of layer l ; act[n_layers] = output
        double we = 0.0;
#pragma unroll
        for (int u = 0; u < GG_E; ++u) we = (u == e) ? w[u] : we;
#pragma unroll
        for (int q = 0; q < GG_W; ++q) act[0][q] = 0.0;
#pragma unroll
        for (int p = 0; p < GG_MAXP; ++p) {
          double v = 0.0;
#pragma unroll
          for (int u = 0; u < GG_E; ++u) v = (u == e) ? pv[p][u] : v;
          act[0][p] = v;
        }
        for (int L = 0; L < P.n_layers; ++L) {
          const GGLayer ly = P.layers[L];
          for (int o = 0; o < GG_W; ++o) act[L + 1][o] = 0.0;
          if (ly.type == 0) {
            const double* Wl = W_s + L * GG_W * (GG_W + 1);
            for (int o = 0; o < ly.out_dim; ++o) {
              double s = Wl[o * (GG_W + 1) + GG_W];
              for (int j = 0; j < ly.in_dim; ++j) s = fma(Wl[o * (GG_W + 1) + j], act[L][j], s);
              act[L + 1][o] = s;
            }
          } else if (ly.type == 1) {
            for (int o = 0; o < ly.out_dim; ++o) {
              double pr = 1.0;
              for (int q = 0; q < ly.step; ++q) pr *= act[L][o * ly.step + q];
              act[L + 1][o] = pr;
            }
          } else {
            for (int o = 0; o < ly.out_dim; ++o) act[L + 1][o] = exp(act[L][o]);
          }
        }
        // reverse: adj = c W d k / d act[L]
        double adj[GG_W], nxt[GG_W];
        for (int q = 0; q < GG_W; ++q) adj[q] = 0.0;
        adj[0] = we;
        for (int L = P.n_layers - 1; L >= 0; --L) {
          const GGLayer ly = P.layers[L];
          for (int q = 0; q < GG_W; ++q) nxt[q] = 0.0;
          if (ly.type == 0) {
            const double* Wl = W_s + L * GG_W * (GG_W + 1);
            for (int o = 0; o < ly.out_dim; ++o) {
              const double ao = adj[o];
              for (int j = 0; j < ly.in_dim; ++j) {
                if (ACCW) {
                  double g = gg_wave_sum(ao * act[L][j]);             // d / d W[o][j]
                  if (lane == 0) acc_s[wave][ly.slot0 + o * (ly.in_dim + 1) + j] += g;
                }
                nxt[j] = fma(Wl[o * (GG_W + 1) + j], ao, nxt[j]);
              }
              if (ACCW) {
                double gb = gg_wave_sum(ao);                          // d / d bias[o]
                if (lane == 0) acc_s[wave][ly.slot0 + o * (ly.in_dim + 1) + ly.in_dim] += gb;
              }
            }
          } else if (ly.type == 1) {
            for (int o = 0; o < ly.out_dim; ++o)
              for (int q = 0; q < ly.step; ++q) {
                double pr = adj[o];
                for (int q2 = 0; q2 < ly.step; ++q2) if (q2 != q) pr *= act[L][o * ly.step + q2];
                nxt[o * ly.step + q] = pr;
              }
          } else {
            for (int o = 0; o < ly.out_dim; ++o) nxt[o] = adj[o] * act[L + 1][o];
          }
          for (int q = 0; q < GG_W; ++q) adj[q] = nxt[q];
        }
#pragma unroll
        for (int p = 0; p < GG_MAXP; ++p)
#pragma unroll
          for (int u = 0; u < GG_E; ++u) if (u == e) fp[p][u] = adj[p];
      }
    }
}

__global__ __launch_bounds__(256) void gg_kernel(GGArgs a, GGProg P) {
  __shared__ double Fr_s[GG_MAXF * GG_T];
  __shared__ double Fc_s[GG_MAXF * GG_T];
  __shared__ double acc_s[4][GG_MAXSLOT + 1];
  __shared__ double W_s[GG_MAXL * GG_W * (GG_W + 1)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = tid & 15, ty = tid >> 4;       // 16 x 16 threads ; 2 rows x 2 cols each
  for (int s = tid; s < 4 * (GG_MAXSLOT + 1); s += 256) (&acc_s[0][0])[s] = 0.0;
  if (P.nkn) for (int s = tid; s < P.n_layers * GG_W * (GG_W + 1); s += 256) W_s[s] = a.Wnet[s];
  __syncthreads();

  const bool same_points = a.rect ? (a.same_points != 0) : true;
  const i64 ntiles = (i64)a.tiles * a.tiles_c;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int ti = (int)(t / a.tiles_c), tj = (int)(t % a.tiles_c);
    if (!a.rect && tj > ti) continue;
    const i64 gi0 = (i64)ti * GG_T, gj0 = (i64)tj * GG_T;
    // ---- weights c_e W_e
    double w[GG_E];
    double dsum = 0.0;
#pragma unroll
    for (int e = 0; e < GG_E; ++e) {
      const i64 i = gi0 + ty * 2 + (e >> 1), j = gj0 + tx * 2 + (e & 1);
      double val = 0.0;
      if (a.rect) {
        if (i < a.nr && j < a.nc) val = a.Wd[i * a.ldw + j];
      } else if (i < a.n && j <= i) {
        double s = 0.0;
        for (int q = 0; q < a.r; ++q) s += a.A[(i64)q * a.lda + i] * a.A[(i64)q * a.lda + j];
        val = s - (double)a.r * a.Kinv[i * a.ldk + j];
        if (i == j) { val *= 0.5; dsum += val; }
      }
      w[e] = val;
    }
    if (!a.rect && ti == tj) {                             // noise: d K_y / d sigma^2 = I
      dsum = gg_wave_sum(dsum);
      if (lane == 0) acc_s[wave][GG_MAXSLOT] += dsum;
    }
    // ---- pass 1: primitive values ; pass 2: adjoints  fp[p][e] = c W d k / d prim_p (network weights on the way)
    double pv[GG_MAXP][GG_E], fp[GG_MAXP][GG_E];
    gg_values(a, P, gi0, gj0, same_points, Fr_s, Fc_s, tid, ty, tx, pv);
    gg_adjoints<true>(P, W_s, w, pv, fp, acc_s, lane, wave);
    // ---- pass 3: the primitives' own parameters
    for (int nd = 0; nd < P.n_nodes; ++nd) {
      const GGNode node = P.nodes[nd];
      if (node.prim < 0) continue;
      double f4[GG_E], k4[GG_E];
#pragma unroll
      for (int e = 0; e < GG_E; ++e) { f4[e] = 0.0; k4[e] = 0.0; }
#pragma unroll
      for (int p = 0; p < GG_MAXP; ++p)
        if (node.prim == p) {
#pragma unroll
          for (int e = 0; e < GG_E; ++e) { f4[e] = fp[p][e]; k4[e] = pv[p][e]; }
        }
      {                                                    // variance: d prim / d v = prim / v
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < GG_E; ++e) s += f4[e] * k4[e];
        s = gg_wave_sum(s) / node.variance;
        if (lane == 0) acc_s[wave][node.slot0] += s;
      }
      if (node.op == GPS_K_WHITE || node.op == GPS_K_CONSTANT) continue;
      if (node.op == GPS_K_PERIODIC) {
        gg_stage(a, node.f0, 3 * node.ndims, gi0, gj0, Fr_s, Fc_s, tid);
        double val[GG_E], S4[GG_E];
        gg_prim(node, Fr_s, Fc_s, ty, tx, gi0, gj0, same_points, val, S4);
        const double l = node.ls0, l2 = l * l;
        double s = 0.0;                                    // d k / d l = k S / l^3
#pragma unroll
        for (int e = 0; e < GG_E; ++e) s += f4[e] * k4[e] * S4[e];
        s = gg_wave_sum(s) / (l2 * l);
        if (lane == 0) acc_s[wave][node.slot0 + 1] += s;
        // d k / d p = k / (2 l^2) sum_d sin(a_i - a_j) (a_i - a_j) / (2 p),  a = 2 pi x / p
        double sp = 0.0;
        for (int d = 0; d < node.ndims; ++d) {
#pragma unroll
          for (int e = 0; e < GG_E; ++e) {
            const int ri = ty * 2 + (e >> 1), cj = tx * 2 + (e & 1);
            const double ci = Fr_s[(2 * d) * GG_T + ri], si = Fr_s[(2 * d + 1) * GG_T + ri];
            const double cj_ = Fc_s[(2 * d) * GG_T + cj], sj = Fc_s[(2 * d + 1) * GG_T + cj];
            const double da = Fr_s[(2 * node.ndims + d) * GG_T + ri] - Fc_s[(2 * node.ndims + d) * GG_T + cj];
            sp += f4[e] * k4[e] * (si * cj_ - ci * sj) * da;
          }
        }
        sp = gg_wave_sum(sp) / (2.0 * l2) / (2.0 * node.period);
        if (lane == 0) acc_s[wave][node.slot0 + 2] += sp;
        continue;
      }
      // stationary: Q_e = f dk/d(r2) ; d k / d l_d = Q * (-2 delta_d^2 / l_d)
      gg_stage(a, node.f0, node.ndims, gi0, gj0, Fr_s, Fc_s, tid);
      double val[GG_E], q4[GG_E], Q[GG_E];
      gg_prim(node, Fr_s, Fc_s, ty, tx, gi0, gj0, same_points, val, q4);
      const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
#pragma unroll
      for (int e = 0; e < GG_E; ++e) {
        const double k = k4[e];
        double dk;
        if (node.op == GPS_K_RBF) dk = -0.5 * k;
        else {
          const double rad = sqrt(q4[e] + 1e-12);
          if (node.op == GPS_K_MATERN12) dk = -k / (2.0 * rad);
          else if (node.op == GPS_K_EXPONENTIAL) dk = -k / (4.0 * rad);
          else if (node.op == GPS_K_MATERN32) dk = -1.5 * node.variance * exp(-sq3 * rad);
          else dk = -(5.0 / 6.0) * node.variance * (1.0 + sq5 * rad) * exp(-sq5 * rad);
        }
        Q[e] = f4[e] * dk;
      }
      for (int d = 0; d < node.ndims; ++d) {
        const double r0 = Fr_s[d * GG_T + ty * 2], r1 = Fr_s[d * GG_T + ty * 2 + 1];
        const double c0 = Fc_s[d * GG_T + tx * 2], c1 = Fc_s[d * GG_T + tx * 2 + 1];
        double s = Q[0] * (r0 - c0) * (r0 - c0) + Q[1] * (r0 - c1) * (r0 - c1) + Q[2] * (r1 - c0) * (r1 - c0) + Q[3] * (r1 - c1) * (r1 - c1);
        s = gg_wave_sum(s);
        if (lane == 0) acc_s[wave][node.slot0 + 1 + d] += -2.0 * s;     // the host divides by l_d
      }
    }
  }
  __syncthreads();
  for (int s = tid; s < GG_MAXSLOT + 1; s += 256)
    a.partial[(i64)blockIdx.x * (GG_MAXSLOT + 1) + s] = (acc_s[0][s] + acc_s[1][s]) + (acc_s[2][s] + acc_s[3][s]);
}

// ---- gradient with respect to the INPUT points of the first argument ------------------------------------------------
//   G[i][d] = sum_j Wd[i][j] d k(xr_i, xc_j) / d xr_i[d]
// -- what reverse-mode autodiff through kern.K(Z, X) hands to a trainable Z (features.py:65: the inducing inputs are a
// Parameter; examples/svgp.py:161 minimises over every variable of the graph).  Same per-entry forward / reverse pass
// through the program as gg_kernel; then, per primitive, the chain through its argument:
//   stationary  r2 = sum_d (F_id - F_jd)^2, F = x / l :  d k / d x_id = (d k / d r2) 2 (F_id - F_jd) / l_d
//   Periodic    S = sum_d sin^2((a_id - a_jd) / 2), a = 2 pi x / p :  d k / d x_id = -k / (2 l^2) (1/2) sin(a_id - a_jd) 2 pi / p
// Workgroup (ti, slice) owns 32 rows and a slice of the column tiles; row sums over the 16 threads of a patch row by
// shuffles, over the tiles in LDS; the slices' partial sums [slices][rows][d_all] are added in order on the host.
struct GIArgs { const GGFeat* feats; double* part; int d_all; int slices; i64 rows_pad; };

__global__ __launch_bounds__(256) void gg_input_kernel(GGArgs a, GGProg P, GIArgs gi) {
  __shared__ double Fr_s[GG_MAXF * GG_T];
  __shared__ double Fc_s[GG_MAXF * GG_T];
  __shared__ double W_s[GG_MAXL * GG_W * (GG_W + 1)];
  __shared__ double rowacc[GG_T][GPS_MAX_DIMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = tid & 15, ty = tid >> 4;
  for (int s = tid; s < GG_T * GPS_MAX_DIMS; s += 256) (&rowacc[0][0])[s] = 0.0;
  if (P.nkn) for (int s = tid; s < P.n_layers * GG_W * (GG_W + 1); s += 256) W_s[s] = a.Wnet[s];
  __syncthreads();
  const bool same_points = a.same_points != 0;
  const int ti = blockIdx.x, slice = blockIdx.y;
  const int per = (a.tiles_c + gi.slices - 1) / gi.slices;
  const int tj0 = slice * per, tj1 = min(a.tiles_c, tj0 + per);
  const i64 gi0 = (i64)ti * GG_T;
  for (int tj = tj0; tj < tj1; ++tj) {
    const i64 gj0 = (i64)tj * GG_T;
    double w[GG_E];
#pragma unroll
    for (int e = 0; e < GG_E; ++e) {
      const i64 i = gi0 + ty * 2 + (e >> 1), j = gj0 + tx * 2 + (e & 1);
      w[e] = (i < a.nr && j < a.nc) ? a.Wd[i * a.ldw + j] : 0.0;
    }
    double pv[GG_MAXP][GG_E], fp[GG_MAXP][GG_E];
    gg_values(a, P, gi0, gj0, same_points, Fr_s, Fc_s, tid, ty, tx, pv);
    gg_adjoints<false>(P, W_s, w, pv, fp, nullptr, lane, wave);
    for (int nd = 0; nd < P.n_nodes; ++nd) {
      const GGNode node = P.nodes[nd];
      if (node.prim < 0 || node.op == GPS_K_WHITE || node.op == GPS_K_CONSTANT) continue;
      double f4[GG_E], k4[GG_E];
#pragma unroll
      for (int e = 0; e < GG_E; ++e) { f4[e] = 0.0; k4[e] = 0.0; }
#pragma unroll
      for (int p = 0; p < GG_MAXP; ++p)
        if (node.prim == p) {
#pragma unroll
          for (int e = 0; e < GG_E; ++e) { f4[e] = fp[p][e]; k4[e] = pv[p][e]; }
        }
      double Q[GG_E];
      if (node.op == GPS_K_PERIODIC) {
        gg_stage(a, node.f0, 2 * node.ndims, gi0, gj0, Fr_s, Fc_s, tid);
        const double coef = -0.25 / (node.ls0 * node.ls0) * (2.0 * M_PI / node.period);
#pragma unroll
        for (int e = 0; e < GG_E; ++e) Q[e] = f4[e] * k4[e] * coef;
        for (int d = 0; d < node.ndims; ++d) {
          const double ci0 = Fr_s[(2 * d) * GG_T + ty * 2], si0 = Fr_s[(2 * d + 1) * GG_T + ty * 2];
          const double ci1 = Fr_s[(2 * d) * GG_T + ty * 2 + 1], si1 = Fr_s[(2 * d + 1) * GG_T + ty * 2 + 1];
          const double cj0 = Fc_s[(2 * d) * GG_T + tx * 2], sj0 = Fc_s[(2 * d + 1) * GG_T + tx * 2];
          const double cj1 = Fc_s[(2 * d) * GG_T + tx * 2 + 1], sj1 = Fc_s[(2 * d + 1) * GG_T + tx * 2 + 1];
          double s0 = Q[0] * (si0 * cj0 - ci0 * sj0) + Q[1] * (si0 * cj1 - ci0 * sj1);
          double s1 = Q[2] * (si1 * cj0 - ci1 * sj0) + Q[3] * (si1 * cj1 - ci1 * sj1);
#pragma unroll
          for (int off = 1; off < 16; off <<= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
          if (tx == 0) {
            const int dim = gi.feats[node.f0 + 2 * d].dim;
            rowacc[ty * 2][dim] += s0; rowacc[ty * 2 + 1][dim] += s1;
          }
        }
        continue;
      }
      gg_stage(a, node.f0, node.ndims, gi0, gj0, Fr_s, Fc_s, tid);
      double val[GG_E], q4[GG_E];
      gg_prim(node, Fr_s, Fc_s, ty, tx, gi0, gj0, same_points, val, q4);
      const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
#pragma unroll
      for (int e = 0; e < GG_E; ++e) {
        const double k = k4[e];
        double dk;
        if (node.op == GPS_K_RBF) dk = -0.5 * k;
        else {
          const double rad = sqrt(q4[e] + 1e-12);
          if (node.op == GPS_K_MATERN12) dk = -k / (2.0 * rad);
          else if (node.op == GPS_K_EXPONENTIAL) dk = -k / (4.0 * rad);
          else if (node.op == GPS_K_MATERN32) dk = -1.5 * node.variance * exp(-sq3 * rad);
          else dk = -(5.0 / 6.0) * node.variance * (1.0 + sq5 * rad) * exp(-sq5 * rad);
        }
        Q[e] = 2.0 * f4[e] * dk;
      }
      for (int d = 0; d < node.ndims; ++d) {
        const double r0 = Fr_s[d * GG_T + ty * 2], r1 = Fr_s[d * GG_T + ty * 2 + 1];
        const double c0 = Fc_s[d * GG_T + tx * 2], c1 = Fc_s[d * GG_T + tx * 2 + 1];
        double s0 = Q[0] * (r0 - c0) + Q[1] * (r0 - c1);
        double s1 = Q[2] * (r1 - c0) + Q[3] * (r1 - c1);
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
        if (tx == 0) {
          const GGFeat ft = gi.feats[node.f0 + d];
          rowacc[ty * 2][ft.dim] += s0 / ft.param; rowacc[ty * 2 + 1][ft.dim] += s1 / ft.param;
        }
      }
    }
  }
  __syncthreads();
  for (int s = tid; s < GG_T * gi.d_all; s += 256) {
    const int r = s / gi.d_all, d = s - r * gi.d_all;
    gi.part[((i64)slice * gi.rows_pad + gi0 + r) * gi.d_all + d] = rowacc[r][d];
  }
}

// ---- host side ------------------------------------------------------------------------------------
#define GG_BLOCKS 2048

static bool gg_is_prim(int op) {
  return op == GPS_K_RBF || op == GPS_K_MATERN12 || op == GPS_K_MATERN32 || op == GPS_K_MATERN52 || op == GPS_K_PERIODIC ||
         op == GPS_K_WHITE || op == GPS_K_CONSTANT || op == GPS_K_EXPONENTIAL;
}

int gps_grad_general_slots(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, int* n_slots) {
  int s = 0;
  for (int i = 0; i < n_nodes; ++i) {
    const int op = prog[i].op;
    if (op == GPS_K_ADD || op == GPS_K_MUL || op == GPS_K_NKN_PRODUCT || op == GPS_K_NKN_ACT) continue;
    if (op == GPS_K_WHITE || op == GPS_K_CONSTANT) s += 1;
    else if (op == GPS_K_PERIODIC) s += 3;
    else if (op == GPS_K_NKN_LINROW) s += prog[i].n_dims + 1;
    else if (gg_is_prim(op)) s += 1 + prog[i].n_dims;
    else return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: unknown op");
  }
  *n_slots = s;
  return GPS_OK;
}

// compiled form of a kernel program for gg_kernel
struct GGBuilt {
  GGProg P;
  std::vector<GGFeat> feats;
  std::vector<double> ls_of_slot;
  std::vector<double> W;
};

static int gg_build(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, i64 d_all, GGBuilt& B) {
  GGProg& P = B.P;
  std::vector<GGFeat>& feats = B.feats;
  std::vector<double>& ls_of_slot = B.ls_of_slot;
  std::vector<double>& W = B.W;
  W.assign((size_t)GG_MAXL * GG_W * (GG_W + 1), 0.0);
  memset(&P, 0, sizeof(P));
  int n_prim_nodes = 0;
  for (int i = 0; i < n_nodes; ++i) if (prog[i].op < GPS_K_NKN_LINROW) n_prim_nodes = i + 1;
  P.nkn = (n_prim_nodes < n_nodes) ? 1 : 0;
  if (n_prim_nodes <= 0 || n_prim_nodes > GG_MAX_NODES) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: program too long");
  P.n_nodes = n_prim_nodes;
  int depth = 0;
  for (int i = 0; i < n_prim_nodes; ++i) {
    const gps_kern_node_t& nd = prog[i];
    GGNode& g = P.nodes[i];
    g.op = nd.op; g.prim = -1; g.variance = nd.variance; g.period = nd.period;
    if (nd.op == GPS_K_ADD || nd.op == GPS_K_MUL) {
      if (P.nkn) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: NKN primitives must be primitive kernels");
      if (depth < 2) return gps_fail(h, GPS_ERR_ARG, "gradient: stack underflow");
      depth -= 1; continue;
    }
    if (!gg_is_prim(nd.op)) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: unknown op");
    if (P.n_prims >= GG_MAXP) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: kernel programs with more than 8 primitive nodes are not supported");
    g.prim = P.n_prims++;
    g.slot0 = P.n_slots;
    depth += 1;
    if (!P.nkn && depth > GPS_MAX_STACK) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: expression too deep");
    if (!(nd.variance > 0.0)) return gps_fail(h, GPS_ERR_ARG, "gradient: variance must be positive");
    ls_of_slot.push_back(0.0);
    if (nd.op == GPS_K_WHITE || nd.op == GPS_K_CONSTANT) { P.n_slots += 1; continue; }
    if (nd.n_dims <= 0 || nd.n_dims > GPS_MAX_DIMS) return gps_fail(h, GPS_ERR_ARG, "gradient: n_dims out of range");
    for (int d = 0; d < nd.n_dims; ++d)
      if (nd.active_dims[d] < 0 || nd.active_dims[d] >= d_all) return gps_fail(h, GPS_ERR_ARG, "gradient: active dim outside X");
    g.ndims = nd.n_dims;
    g.f0 = (int)feats.size();
    if (nd.op == GPS_K_PERIODIC) {
      for (int d = 0; d < nd.n_dims; ++d) { feats.push_back({nd.active_dims[d], 1, nd.period}); feats.push_back({nd.active_dims[d], 2, nd.period}); }
      for (int d = 0; d < nd.n_dims; ++d) feats.push_back({nd.active_dims[d], 3, nd.period});
      g.nf = 3 * nd.n_dims; g.ls0 = nd.lengthscales[0];
      P.n_slots += 3; ls_of_slot.push_back(0.0); ls_of_slot.push_back(0.0);
    } else {
      for (int d = 0; d < nd.n_dims; ++d) { feats.push_back({nd.active_dims[d], 0, nd.lengthscales[d]}); ls_of_slot.push_back(nd.lengthscales[d]); }
      g.nf = nd.n_dims;
      P.n_slots += 1 + nd.n_dims;
    }
    if (g.nf > GG_MAXF) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: too many active dims");
  }
  if (!P.nkn && depth != 1) return gps_fail(h, GPS_ERR_ARG, "gradient: program must leave exactly one value");
  // ---- network layers (kernel-program encoding of gpflowSlim/neural_kernel_network: see kmat.hip compile_prog)
  if (P.nkn) {
    int width = P.n_prims, i = n_prim_nodes;
    while (i < n_nodes) {
      if (P.n_layers >= GG_MAXL) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: more than 8 network layers");
      GGLayer& ly = P.layers[P.n_layers];
      if (prog[i].op == GPS_K_NKN_LINROW) {
        const int layer_id = prog[i].active_dims[0];
        double* Wl = W.data() + (size_t)P.n_layers * GG_W * (GG_W + 1);
        int o = 0;
        ly.type = 0; ly.in_dim = width; ly.slot0 = P.n_slots;
        while (i < n_nodes && prog[i].op == GPS_K_NKN_LINROW && prog[i].active_dims[0] == layer_id) {
          if (prog[i].n_dims != width || o >= GG_W || width > GG_W) return gps_fail(h, GPS_ERR_ARG, "gradient: Linear layer width mismatch (max 16)");
          for (int j = 0; j < width; ++j) Wl[o * (GG_W + 1) + j] = prog[i].lengthscales[j];
          Wl[o * (GG_W + 1) + GG_W] = prog[i].variance;
          for (int j = 0; j <= width; ++j) ls_of_slot.push_back(0.0);
          P.n_slots += width + 1;
          ++o; ++i;
        }
        ly.out_dim = o; width = o;
      } else if (prog[i].op == GPS_K_NKN_PRODUCT) {
        const int step = prog[i].n_dims;
        if (step < 2 || step > 4 || width % step) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: Product step must be 2, 3 or 4 and divide the width");
        ly.type = 1; ly.in_dim = width; ly.step = step; ly.out_dim = width / step; width = ly.out_dim; ++i;
      } else if (prog[i].op == GPS_K_NKN_ACT) {
        if (prog[i].period != 1.0) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: only the exp activation is available");
        ly.type = 2; ly.in_dim = width; ly.out_dim = width; ++i;
      } else return gps_fail(h, GPS_ERR_ARG, "gradient: unknown layer op");
      ++P.n_layers;
    }
    if (width != 1) return gps_fail(h, GPS_ERR_ARG, "gradient: the network must end with one output");
  }
  if (P.n_slots > GG_MAXSLOT) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: too many parameters");
  return GPS_OK;
}

// features of `n` points (padded to npad) into `feat`
static int gg_features(gps_handle_t h, const GGBuilt& B, const double* dX, i64 n, i64 d_all, i64 npad, DevBuf& feat) {
  const int nfeat = (int)B.feats.size();
  GPS_HIP(h, feat.ensure((size_t)(nfeat > 0 ? nfeat : 1) * npad * 8));
  if (nfeat == 0) return GPS_OK;
  GPS_HIP(h, h->dProg.ensure((size_t)nfeat * sizeof(GGFeat) + 64));
  GPS_HIP(h, h->ring.upload(h->dProg.p, B.feats.data(), (size_t)nfeat * sizeof(GGFeat), h->stream));
  LaunchScope ls(h, KC_KMAT, 0.0, 8.0 * (double)npad * nfeat);
  hipLaunchKernelGGL(gg_prep_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, dX, n, d_all, npad,
                     (const GGFeat*)h->dProg.p, nfeat, feat.d(), npad);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// launch + fold the fixed-order partials; slots: set (accumulate == 0) or added to
static int gg_run(gps_handle_t h, const GGBuilt& B, GGArgs& a, double flops, double bytes, int accumulate,
                  double* grad_slots_host, double* grad_noise_host) {
  const GGProg& P = B.P;
  const size_t wbytes = B.W.size() * 8;
  GPS_HIP(h, h->dNkn.ensure(wbytes + 64));
  GPS_HIP(h, h->ring.upload(h->dNkn.p, B.W.data(), wbytes, h->stream));
  a.Wnet = (const double*)h->dNkn.p;
  const size_t pbytes = (size_t)GG_BLOCKS * (GG_MAXSLOT + 1) * 8;
  GPS_HIP(h, h->dTmp2.ensure(pbytes));
  a.partial = h->dTmp2.d();
  {
    LaunchScope ls(h, KC_REDUCE, flops, bytes);
    hipLaunchKernelGGL(gg_kernel, dim3(GG_BLOCKS), dim3(256), 0, h->stream, a, P);
    GPS_HIP(h, hipGetLastError());
  }
  std::vector<double> part((size_t)GG_BLOCKS * (GG_MAXSLOT + 1));
  GPS_HIP(h, hipMemcpyAsync(part.data(), a.partial, pbytes, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  for (int s = 0; s <= GG_MAXSLOT; ++s) {
    if (s >= P.n_slots && s != GG_MAXSLOT) continue;
    double tot = 0.0;
    for (int b = 0; b < GG_BLOCKS; ++b) tot += part[(size_t)b * (GG_MAXSLOT + 1) + s];
    if (s == GG_MAXSLOT) { if (grad_noise_host) *grad_noise_host = tot; }
    else {
      if (B.ls_of_slot[s] > 0.0) tot /= B.ls_of_slot[s];       // -2 delta^2 / l_d : delta is already x/l
      if (accumulate) grad_slots_host[s] += tot; else grad_slots_host[s] = tot;
    }
  }
  return GPS_OK;
}

int gps_launch_grad_general(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* dX, i64 n, i64 d_all,
                            i64 npad, const double* dKinv, i64 ldk, const double* dA, i64 lda, i64 r,
                            double* grad_slots_host, double* grad_noise_host) {
  GGBuilt B;
  int rc = gg_build(h, prog, n_nodes, d_all, B);
  if (rc) return rc;
  rc = gg_features(h, B, dX, n, d_all, npad, h->dFeat);
  if (rc) return rc;
  GGArgs a;
  memset(&a, 0, sizeof(a));
  a.Ft = h->dFeat.d(); a.ldf = npad; a.Ftc = a.Ft; a.ldfc = npad; a.Kinv = dKinv; a.ldk = ldk; a.A = dA; a.lda = lda; a.r = (int)r;
  a.n = n; a.npad = npad; a.tiles = (int)(npad / GG_T); a.tiles_c = a.tiles; a.rect = 0; a.same_points = 1;
  return gg_run(h, B, a, 0.5 * (double)npad * npad * (60.0 + 4.0 * B.feats.size() + 40.0 * B.P.n_layers), 4.0 * (double)npad * npad,
                0, grad_slots_host, grad_noise_host);
}

// Vector-Jacobian product of the kernel-matrix build:  slots (+)= sum_{i < nr, j < nc} Wd[i][j] d k(xr_i, xc_j) / d theta
// for a device-resident cotangent Wd [nr, nc] (leading dimension ldw) -- what reverse-mode autodiff through kern.K(X, X2)
// (kernels.py:408-439, 1071-1084; neural_kernel_network.py:41-47) delivers.  dXc == nullptr: K(Xr, Xr) (White on i == j).
int gps_launch_kmat_vjp(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* dXr, i64 nr, const double* dXc,
                        i64 nc, i64 d_all, const double* Wd, i64 ldw, int accumulate, double* grad_slots_host) {
  GGBuilt B;
  int rc = gg_build(h, prog, n_nodes, d_all, B);
  if (rc) return rc;
  const bool same = (dXc == nullptr);
  if (same) { dXc = dXr; nc = nr; }
  const i64 nrp = gps_pad(nr), ncp = gps_pad(nc);
  rc = gg_features(h, B, dXr, nr, d_all, nrp, h->dFeat);
  if (rc) return rc;
  if (!same) { rc = gg_features(h, B, dXc, nc, d_all, ncp, h->dFeat2); if (rc) return rc; }
  GGArgs a;
  memset(&a, 0, sizeof(a));
  a.Ft = h->dFeat.d(); a.ldf = nrp;
  a.Ftc = same ? h->dFeat.d() : h->dFeat2.d(); a.ldfc = same ? nrp : ncp;
  a.n = nr; a.npad = nrp; a.tiles = (int)(nrp / GG_T); a.tiles_c = (int)(ncp / GG_T);
  a.rect = 1; a.same_points = same ? 1 : 0; a.Wd = Wd; a.ldw = ldw; a.nr = nr; a.nc = nc;
  return gg_run(h, B, a, (double)nrp * ncp * (60.0 + 4.0 * B.feats.size() + 40.0 * B.P.n_layers), 8.0 * (double)nrp * ncp,
                accumulate, grad_slots_host, nullptr);
}

// d (sum_i kbar_i Kdiag_i) / d theta for the constant Kdiag of these programs (every primitive's Kdiag is its variance,
// kernels.py:428-429, 803-804, 327-328; Sum / Product fold them, :1075-1076, 1083-1084): kbar = sum_i kbar_i; only the
// variance slots receive anything.  Forward mode over the fold, one primitive at a time.
int gps_kdiag_vjp(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, i64 d_all, double kbar, double* grad_slots_host) {
  GGBuilt B;
  int rc = gg_build(h, prog, n_nodes, d_all, B);
  if (rc) return rc;
  if (B.P.nkn) return gps_fail(h, GPS_ERR_UNSUPPORTED, "Kdiag gradient: neural-kernel-network programs are not supported here");
  for (int p = 0; p < B.P.n_prims; ++p) {
    double sv[GPS_MAX_STACK + 1], st[GPS_MAX_STACK + 1]; int sp = 0;
    int slot = -1;
    for (int nd = 0; nd < B.P.n_nodes; ++nd) {
      const GGNode& g = B.P.nodes[nd];
      if (g.op == GPS_K_ADD || g.op == GPS_K_MUL) {
        const double b = sv[--sp], tb = st[sp], a = sv[--sp], ta = st[sp];
        sv[sp] = (g.op == GPS_K_ADD) ? a + b : a * b;
        st[sp] = (g.op == GPS_K_ADD) ? ta + tb : ta * b + a * tb;
        ++sp;
      } else {
        sv[sp] = g.variance; st[sp] = (g.prim == p) ? 1.0 : 0.0; ++sp;
        if (g.prim == p) slot = g.slot0;
      }
    }
    if (slot >= 0) grad_slots_host[slot] += kbar * st[0];
  }
  return GPS_OK;
}

// grad_X_host [nr, d_all] += factor * sum_j Wd[i][j] d k(xr_i, xc_j) / d xr_i  for a device-resident cotangent Wd [nr, nc];
// dXc == nullptr: k(xr_i, xr_j) differentiated in its FIRST argument only (for a symmetric Wd the full gradient of
// sum_ij Wd_ij k(x_i, x_j) with respect to x is twice that).
int gps_launch_kmat_input_vjp(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* dXr, i64 nr,
                              const double* dXc, i64 nc, i64 d_all, const double* Wd, i64 ldw, double factor,
                              double* grad_X_host) {
  if (d_all > GPS_MAX_DIMS) return gps_fail(h, GPS_ERR_UNSUPPORTED, "input gradient: more input dimensions than GPS_MAX_DIMS");
  GGBuilt B;
  int rc = gg_build(h, prog, n_nodes, d_all, B);
  if (rc) return rc;
  const bool same = (dXc == nullptr);
  if (same) { dXc = dXr; nc = nr; }
  const i64 nrp = gps_pad(nr), ncp = gps_pad(nc);
  if (!same) { rc = gg_features(h, B, dXc, nc, d_all, ncp, h->dFeat2); if (rc) return rc; }
  rc = gg_features(h, B, dXr, nr, d_all, nrp, h->dFeat);            // (last: leaves the feature table in dProg)
  if (rc) return rc;
  GGArgs a;
  memset(&a, 0, sizeof(a));
  a.Ft = h->dFeat.d(); a.ldf = nrp;
  a.Ftc = same ? h->dFeat.d() : h->dFeat2.d(); a.ldfc = same ? nrp : ncp;
  a.n = nr; a.npad = nrp; a.tiles = (int)(nrp / GG_T); a.tiles_c = (int)(ncp / GG_T);
  a.rect = 1; a.same_points = same ? 1 : 0; a.Wd = Wd; a.ldw = ldw; a.nr = nr; a.nc = nc;
  const size_t wbytes = B.W.size() * 8;
  GPS_HIP(h, h->dNkn.ensure(wbytes + 64));
  GPS_HIP(h, h->ring.upload(h->dNkn.p, B.W.data(), wbytes, h->stream));
  a.Wnet = (const double*)h->dNkn.p;
  GIArgs gi;
  gi.feats = (const GGFeat*)h->dProg.p; gi.d_all = (int)d_all; gi.rows_pad = nrp;
  int slices = 2048 / a.tiles; if (slices > a.tiles_c) slices = a.tiles_c; if (slices < 1) slices = 1; if (slices > 64) slices = 64;
  gi.slices = slices;
  const size_t pbytes = (size_t)slices * nrp * d_all * 8;
  GPS_HIP(h, h->dTmp2.ensure(pbytes));
  gi.part = h->dTmp2.d();
  {
    LaunchScope ls(h, KC_REDUCE, (double)nrp * ncp * (60.0 + 6.0 * B.feats.size() + 40.0 * B.P.n_layers), 8.0 * (double)nrp * ncp);
    hipLaunchKernelGGL(gg_input_kernel, dim3((unsigned)a.tiles, (unsigned)slices), dim3(256), 0, h->stream, a, B.P, gi);
    GPS_HIP(h, hipGetLastError());
  }
  std::vector<double> part((size_t)slices * nrp * d_all);
  GPS_HIP(h, hipMemcpyAsync(part.data(), gi.part, pbytes, hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  for (i64 i = 0; i < nr; ++i)
    for (i64 d = 0; d < d_all; ++d) {
      double tot = 0.0;
      for (int q = 0; q < slices; ++q) tot += part[((size_t)q * nrp + i) * d_all + d];
      grad_X_host[i * d_all + d] += factor * tot;
    }
  return GPS_OK;
}
