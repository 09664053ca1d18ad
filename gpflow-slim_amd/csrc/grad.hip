// Gradient of the GPR log-marginal likelihood w.r.t. the (constrained) kernel hyper-parameters and
// the noise variance:   d LML / d theta = 1/2 sum_ij W_ij dK_ij/dtheta ,  W = A A^T - R K_y^-1 ,
// A = K_y^-1 (Y - m).  The reference obtains this from TensorFlow autodiff through
// tf.cholesky / tf.matrix_triangular_solve / tf.exp (examples/gpr.py:53-54,
// `AdamOptimizer.minimize(objective)`; models/model.py:172-187 for L-BFGS); here it is one fused pass
// over the lower triangle of K_y^-1: every workgroup walks 64x32 tiles, recomputes k(x_i, x_j) and its
// parameter derivatives from the same feature slabs the kernel-matrix build uses, and reduces
// c_ij W_ij dk_ij/dtheta (c = 1 below the diagonal, 1/2 on it) with wave shuffles into per-slot sums.
// HBM traffic: one read of the lower triangle of K_y^-1.
//
// Slot layout (host and device agree on it): for every primitive node of the kernel program, in
// program order: [variance] then, stationary kernels: one slot per active dim (d k / d lengthscale_d;
// an isotropic kernel's gradient is the sum of its slots), Periodic: [lengthscale, period],
// White / Constant: nothing more.  The noise variance has its own output.
#include "gps_common.hpp"
#include <cmath>

#define GT_R 64            // tile rows
#define GT_C 32            // tile cols
#define G_MAXP 4           // primitives per program supported by the gradient kernel
#define G_MAXSLOT 160
#define GRAD_MAX_NODES 32
#define G_MAXF 64          // features per primitive: periodic = 3 per dim (<= 21 dims)

struct GPrepFeat { int dim; int kind; double param; };   // 0: x/param ; 1: cos(2pi x/param) ; 2: sin ; 3: 2pi x/param
struct GNode {
  int op; int prim; int f0; int nf; int norm_row; int slot0; int ndims;
  double variance; double ls0; double period;
};
struct GProg {
  int n_nodes; int n_prims; int n_slots;
  GNode nodes[GRAD_MAX_NODES];
};
struct GArgs {
  const double* Ft; i64 ldf;            // feature-major [rows][ldf]
  const double* Kinv; i64 ldk;          // lower triangle valid
  const double* A; i64 lda; int r;      // [r][lda]
  i64 n, npad;
  double* partial;                      // [gridDim.x][G_MAXSLOT + 1]  (last = noise)
  int tiles_r, tiles_c;
};

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

#define GPREP_SMALL_F 24
struct GPrepTabPtr { const GPrepFeat* f; };
struct GPrepTabVal { GPrepFeat f[GPREP_SMALL_F]; };
template <class Tab>
__device__ __forceinline__ void gprep_body(const double* __restrict__ X, i64 n, i64 d_all, i64 npad, const Tab& tab, int nfeat,
                                           double* __restrict__ Ft, i64 ldf) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  for (int f = 0; f < nfeat; ++f) {
    double v = 0.0;
    if (i < n) {
      const GPrepFeat pf = tab.f[f];
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
__global__ __launch_bounds__(256) void gprep_kernel(const double* __restrict__ X, i64 n, i64 d_all,
                                                    i64 npad, const GPrepFeat* __restrict__ feats,
                                                    int nfeat, double* __restrict__ Ft, i64 ldf) {
  const GPrepTabPtr tab{feats};
  gprep_body(X, n, d_all, npad, tab, nfeat, Ft, ldf);
}
// (a small table travels in the kernel arguments: no copy command in front of the launch)
__global__ __launch_bounds__(256) void gprep_args_kernel(const double* __restrict__ X, i64 n, i64 d_all, i64 npad, GPrepTabVal tab,
                                                         int nfeat, double* __restrict__ Ft, i64 ldf) {
  gprep_body(X, n, d_all, npad, tab, nfeat, Ft, ldf);
}

// value of the program with d(out)/d(prim p) by forward-mode: returns tangent
__device__ __forceinline__ double prog_tangent(const GProg& P, const double (&pv)[G_MAXP], int p) {
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
      for (int u = 1; u < G_MAXP; ++u) val = (q == u) ? pv[u] : val;
#pragma unroll
      for (int s = GPS_MAX_STACK - 1; s > 0; --s) { sv[s] = sv[s - 1]; st[s] = st[s - 1]; }
      sv[0] = val; st[0] = (q == p) ? 1.0 : 0.0;
    }
  }
  return st[0];
}

__global__ __launch_bounds__(256) void grad_kernel(GArgs a, GProg P) {
  __shared__ double Fr_s[G_MAXF * GT_R];
  __shared__ double Fc_s[G_MAXF * GT_C];
  __shared__ double acc_s[4][G_MAXSLOT + 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = tid & 15, ty = tid >> 4;       // 16 x 16 threads ; 4 rows x 2 cols each
  for (int s = tid; s < 4 * (G_MAXSLOT + 1); s += 256) (&acc_s[0][0])[s] = 0.0;
  __syncthreads();

  const i64 ntiles = (i64)a.tiles_r * a.tiles_c;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int ti = (int)(t / a.tiles_c), tj = (int)(t % a.tiles_c);
    const i64 gi0 = (i64)ti * GT_R, gj0 = (i64)tj * GT_C;
    if (gj0 > gi0 + GT_R - 1) continue;                     // strictly above the diagonal
    // ---- weights c_e * W_e for the thread's 8 elements
    double w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const i64 i = gi0 + ty * 4 + (e >> 1), j = gj0 + tx * 2 + (e & 1);
      double val = 0.0;
      if (i < a.n && j <= i) {
        double s = 0.0;
        for (int q = 0; q < a.r; ++q) s += a.A[(i64)q * a.lda + i] * a.A[(i64)q * a.lda + j];
        val = s - (double)a.r * a.Kinv[i * a.ldk + j];
        if (i == j) val *= 0.5;
      }
      w[e] = val;
    }
    // noise: d K_y / d sigma2 = I
    {
      double s = 0.0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const i64 i = gi0 + ty * 4 + (e >> 1), j = gj0 + tx * 2 + (e & 1);
        s += (i == j) ? w[e] : 0.0;
      }
      if (gj0 + GT_C > gi0) {                               // only tiles touching the diagonal
        s = wave_sum64(s);
        if (lane == 0) acc_s[wave][G_MAXSLOT] += s;
      }
    }
    // ---- pass 1: primitive values pv[p][e] and the squared distance / periodic sum r2[p][e]
    double pv[G_MAXP][8], r2[G_MAXP][8];
#pragma unroll
    for (int p = 0; p < G_MAXP; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) { pv[p][e] = 0.0; r2[p][e] = 0.0; }
    for (int nd = 0; nd < P.n_nodes; ++nd) {
      const GNode node = P.nodes[nd];
      if (node.op == GPS_K_ADD || node.op == GPS_K_MUL) continue;
      double val[8], rr[8];
      if (node.op == GPS_K_CONSTANT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { val[e] = node.variance; rr[e] = 0.0; }
      } else if (node.op == GPS_K_WHITE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const i64 i = gi0 + ty * 4 + (e >> 1), j = gj0 + tx * 2 + (e & 1);
          val[e] = (i == j) ? node.variance : 0.0; rr[e] = 0.0;
        }
      } else {
        __syncthreads();
        const int nfd = (node.op == GPS_K_PERIODIC) ? 2 * node.ndims : node.ndims;   // cos/sin or scaled x
        for (int idx = tid; idx < nfd * GT_R; idx += 256) {
          const int f = idx >> 6, pp = idx & 63;
          Fr_s[f * GT_R + pp] = a.Ft[(i64)(node.f0 + f) * a.ldf + gi0 + pp];
        }
        for (int idx = tid; idx < nfd * GT_C; idx += 256) {
          const int f = idx >> 5, pp = idx & 31;
          Fc_s[f * GT_C + pp] = a.Ft[(i64)(node.f0 + f) * a.ldf + gj0 + pp];
        }
        __syncthreads();
        double acc8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc8[e] = 0.0;
        if (node.op == GPS_K_PERIODIC) {
          for (int f = 0; f < nfd; ++f) {
            double fr[4], fc[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) fr[q] = Fr_s[f * GT_R + ty * 4 + q];
#pragma unroll
            for (int q = 0; q < 2; ++q) fc[q] = Fc_s[f * GT_C + tx * 2 + q];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc8[e] = fma(fr[e >> 1], fc[e & 1], acc8[e]);
          }
          const double l2 = node.ls0 * node.ls0;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const double S = 0.5 * ((double)node.ndims - acc8[e]);     // sum_d sin^2(pi D_d / p)
            rr[e] = S;
            val[e] = node.variance * exp(-0.5 * S / l2);
          }
        } else {
          for (int f = 0; f < nfd; ++f) {
            double fr[4], fc[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) fr[q] = Fr_s[f * GT_R + ty * 4 + q];
#pragma unroll
            for (int q = 0; q < 2; ++q) fc[q] = Fc_s[f * GT_C + tx * 2 + q];
#pragma unroll
            for (int e = 0; e < 8; ++e) { const double dlt = fr[e >> 1] - fc[e & 1]; acc8[e] = fma(dlt, dlt, acc8[e]); }
          }
          const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const double q2 = acc8[e];
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
      }
#pragma unroll
      for (int p = 0; p < G_MAXP; ++p)
        if (node.prim == p) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { pv[p][e] = val[e]; r2[p][e] = rr[e]; }
        }
    }
    // ---- pass 2: per primitive, adjoint and parameter contributions
#pragma unroll
    for (int p = 0; p < G_MAXP; ++p) {
      if (p >= P.n_prims) continue;
      // locate the node of primitive p
      GNode node = P.nodes[0];
      for (int nd = 0; nd < P.n_nodes; ++nd)
        if (P.nodes[nd].op != GPS_K_ADD && P.nodes[nd].op != GPS_K_MUL && P.nodes[nd].prim == p) node = P.nodes[nd];
      double f8[8];                       // c W d out / d prim_p
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        double pve[G_MAXP];
#pragma unroll
        for (int u = 0; u < G_MAXP; ++u) pve[u] = pv[u][e];
        f8[e] = w[e] * prog_tangent(P, pve, p);
      }
      // variance: d prim / d v = prim / v
      {
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f8[e] * pv[p][e];
        s = wave_sum64(s) / node.variance;
        if (lane == 0) acc_s[wave][node.slot0] += s;
      }
      if (node.op == GPS_K_WHITE || node.op == GPS_K_CONSTANT) continue;
      if (node.op == GPS_K_PERIODIC) {
        const double l = node.ls0, l2 = l * l;
        // d k / d l = k S / l^3
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f8[e] * pv[p][e] * r2[p][e];
        s = wave_sum64(s) / (l2 * l);
        if (lane == 0) acc_s[wave][node.slot0 + 1] += s;
        // d k / d p = k / (2 l^2) sum_d sin(a_i - a_j) (a_i - a_j) / (2 p),  a = 2 pi x / p
        __syncthreads();
        const int nfd = 3 * node.ndims;
        for (int idx = tid; idx < nfd * GT_R; idx += 256) {
          const int f = idx >> 6, pp = idx & 63;
          Fr_s[f * GT_R + pp] = a.Ft[(i64)(node.f0 + f) * a.ldf + gi0 + pp];
        }
        for (int idx = tid; idx < nfd * GT_C; idx += 256) {
          const int f = idx >> 5, pp = idx & 31;
          Fc_s[f * GT_C + pp] = a.Ft[(i64)(node.f0 + f) * a.ldf + gj0 + pp];
        }
        __syncthreads();
        double sp = 0.0;
        for (int d = 0; d < node.ndims; ++d) {
          // features: [cos_d, sin_d] pairs first (2*ndims), then the angles (ndims)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ri = ty * 4 + (e >> 1), cj = tx * 2 + (e & 1);
            const double ci = Fr_s[(2 * d) * GT_R + ri], si = Fr_s[(2 * d + 1) * GT_R + ri];
            const double cj_ = Fc_s[(2 * d) * GT_C + cj], sj = Fc_s[(2 * d + 1) * GT_C + cj];
            const double da = Fr_s[(2 * node.ndims + d) * GT_R + ri] - Fc_s[(2 * node.ndims + d) * GT_C + cj];
            sp += f8[e] * pv[p][e] * (si * cj_ - ci * sj) * da;
          }
        }
        sp = wave_sum64(sp) / (2.0 * l2) / (2.0 * node.period);
        if (lane == 0) acc_s[wave][node.slot0 + 2] += sp;
        continue;
      }
      // stationary: Q_e = c W adj * dk/d(r2) ; d k / d l_d = Q * (-2 delta_d^2 / l_d)
      double Q[8];
      {
        const double sq3 = 1.7320508075688772, sq5 = 2.23606797749979;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const double k = pv[p][e];
          double dk;
          if (node.op == GPS_K_RBF) dk = -0.5 * k;
          else {
            const double rad = sqrt(r2[p][e] + 1e-12);
            if (node.op == GPS_K_MATERN12) dk = -k / (2.0 * rad);
            else if (node.op == GPS_K_EXPONENTIAL) dk = -k / (4.0 * rad);
            else if (node.op == GPS_K_MATERN32) dk = -1.5 * node.variance * exp(-sq3 * rad);
            else dk = -(5.0 / 6.0) * node.variance * (1.0 + sq5 * rad) * exp(-sq5 * rad);
          }
          Q[e] = f8[e] * dk;
        }
      }
      __syncthreads();
      for (int idx = tid; idx < node.ndims * GT_R; idx += 256) {
        const int f = idx >> 6, pp = idx & 63;
        Fr_s[f * GT_R + pp] = a.Ft[(i64)(node.f0 + f) * a.ldf + gi0 + pp];
      }
      for (int idx = tid; idx < node.ndims * GT_C; idx += 256) {
        const int f = idx >> 5, pp = idx & 31;
        Fc_s[f * GT_C + pp] = a.Ft[(i64)(node.f0 + f) * a.ldf + gj0 + pp];
      }
      __syncthreads();
      for (int d = 0; d < node.ndims; ++d) {
        double fr[4], fc[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) fr[q] = Fr_s[d * GT_R + ty * 4 + q];
#pragma unroll
        for (int q = 0; q < 2; ++q) fc[q] = Fc_s[d * GT_C + tx * 2 + q];
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const double dlt = fr[e >> 1] - fc[e & 1]; s += Q[e] * dlt * dlt; }
        s = wave_sum64(s);
        // lengthscale of dim d travels in the feature table's param; the host divides: see launcher
        if (lane == 0) acc_s[wave][node.slot0 + 1 + d] += -2.0 * s;
      }
    }
  }
  __syncthreads();
  for (int s = tid; s < G_MAXSLOT + 1; s += 256)
    a.partial[(i64)blockIdx.x * (G_MAXSLOT + 1) + s] = (acc_s[0][s] + acc_s[1][s]) + (acc_s[2][s] + acc_s[3][s]);
}

// ---- host side ------------------------------------------------------------------------------------
#define GRAD_BLOCKS 2048

// the per-workgroup partial sums of one slot -> one number, in a fixed order (block s of the launch = slot s; the last = noise).
// (Folding this into grad_kernel -- the workgroup that arrives last reduces -- was measured slower: one workgroup walking
// 161 x nblocks partial sums takes longer than this launch costs: N = 512: +19 us, N = 2048: +150 us.)
__global__ __launch_bounds__(256) void grad_reduce_kernel(const double* __restrict__ partial, int nblocks, int n_slots, double* __restrict__ out) {
  __shared__ double red[256];
  const int s = ((int)blockIdx.x < n_slots) ? (int)blockIdx.x : G_MAXSLOT;
  double t = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) t += partial[(i64)b * (G_MAXSLOT + 1) + s];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[s] = red[0];
}

// programs this file's kernel does not take go to grad_general.hip
static bool grad_needs_general(const gps_kern_node_t* prog, int n_nodes) {
  int prims = 0;
  for (int i = 0; i < n_nodes; ++i) {
    if (prog[i].op >= GPS_K_NKN_LINROW) return true;
    if (prog[i].op != GPS_K_ADD && prog[i].op != GPS_K_MUL) ++prims;
  }
  return prims > G_MAXP;
}

int gps_grad_slots(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, int* n_slots) {
  if (grad_needs_general(prog, n_nodes)) return gps_grad_general_slots(h, prog, n_nodes, n_slots);
  int s = 0;
  for (int i = 0; i < n_nodes; ++i) {
    switch (prog[i].op) {
      case GPS_K_ADD: case GPS_K_MUL: break;
      case GPS_K_WHITE: case GPS_K_CONSTANT: s += 1; break;
      case GPS_K_PERIODIC: s += 3; break;
      case GPS_K_RBF: case GPS_K_MATERN12: case GPS_K_MATERN32: case GPS_K_MATERN52: case GPS_K_EXPONENTIAL:
        s += 1 + prog[i].n_dims; break;
      default: return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: unknown op");
    }
  }
  *n_slots = s;
  return GPS_OK;
}

bool gps_grad_is_simple(const gps_kern_node_t* prog, int n_nodes) { return !grad_needs_general(prog, n_nodes); }

// post: the analysed program, and what the host still has to do with the sums once it has them (gps_grad_finish).
// The gradient in two halves for callers that want the features early (gps_gpr_lml_grad at small N launches gps_grad_prepare in
// front of the factorisation, where it is off the chain): prepare = program analysis + feature launch (into dFeatG), run = the
// tile sums and their reduction into d_sums[GPS_GRAD_SUMS] (slot s at [s], noise at [G_MAXSLOT]).  No synchronisation in either.
int gps_grad_prepare(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* dX, i64 n, i64 d_all, i64 npad, GradPost* post) {
  static_assert(GPS_GRAD_SUMS == G_MAXSLOT + 1, "GPS_GRAD_SUMS");
  post->blob.resize(sizeof(GProg));
  GProg& P = *reinterpret_cast<GProg*>(post->blob.data());
  std::vector<GPrepFeat> feats;
  std::vector<double>& ls_of_slot = post->ls_of_slot;          // lengthscale that divides a per-dim slot
  ls_of_slot.clear();
  P.n_nodes = n_nodes; P.n_prims = 0; P.n_slots = 0;
  if (n_nodes <= 0 || n_nodes > GRAD_MAX_NODES) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: program too long");
  for (int i = 0; i < n_nodes; ++i)
    if (prog[i].op >= GPS_K_NKN_LINROW) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: neural-kernel-network programs are not supported yet");
  int depth = 0;
  for (int i = 0; i < n_nodes; ++i) {
    const gps_kern_node_t& nd = prog[i];
    GNode& g = P.nodes[i];
    g.op = nd.op; g.prim = -1; g.f0 = 0; g.nf = 0; g.norm_row = -1; g.slot0 = 0; g.ndims = 0;
    g.variance = nd.variance; g.ls0 = 0.0; g.period = nd.period;
    if (nd.op == GPS_K_ADD || nd.op == GPS_K_MUL) { if (depth < 2) return gps_fail(h, GPS_ERR_ARG, "gradient: stack underflow"); depth -= 1; continue; }
    if (P.n_prims >= G_MAXP)
      return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: kernel programs with more than 4 primitive nodes are not supported yet");
    g.prim = P.n_prims++;
    g.slot0 = P.n_slots;
    depth += 1;
    if (depth > GPS_MAX_STACK) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: expression too deep");
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
    if (g.nf > G_MAXF) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: too many active dims");
  }
  if (depth != 1) return gps_fail(h, GPS_ERR_ARG, "gradient: program must leave exactly one value");
  if (P.n_slots > G_MAXSLOT) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gradient: too many parameters");

  const int nfeat = (int)feats.size();
  if (nfeat > 0) {
    GPS_HIP(h, h->dFeatG.ensure((size_t)nfeat * npad * 8));
    LaunchScope ls(h, KC_KMAT, 0.0, 8.0 * (double)npad * nfeat);
    if (nfeat <= GPREP_SMALL_F) {
      GPrepTabVal tab;
      memset(&tab, 0, sizeof(tab));
      for (int f = 0; f < nfeat; ++f) tab.f[f] = feats[f];
      hipLaunchKernelGGL(gprep_args_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, dX, n, d_all, npad, tab,
                         nfeat, h->dFeatG.d(), npad);
    } else {
      GPS_HIP(h, h->dProg.ensure((size_t)nfeat * sizeof(GPrepFeat) + 64));
      GPS_HIP(h, h->ring.upload(h->dProg.p, feats.data(), (size_t)nfeat * sizeof(GPrepFeat), h->stream));
      hipLaunchKernelGGL(gprep_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, dX, n, d_all, npad,
                         (const GPrepFeat*)h->dProg.p, nfeat, h->dFeatG.d(), npad);
    }
    GPS_HIP(h, hipGetLastError());
  }
  post->n_slots = P.n_slots; post->nfeat = nfeat;
  return GPS_OK;
}

int gps_grad_run(gps_handle_t h, const GradPost& post, i64 n, i64 npad, const double* dKinv, i64 ldk, const double* dA, i64 lda, i64 r,
                 double* d_sums) {
  const GProg& P = *reinterpret_cast<const GProg*>(post.blob.data());
  const int nfeat = post.nfeat;
  GArgs a;
  a.Ft = h->dFeatG.d(); a.ldf = npad; a.Kinv = dKinv; a.ldk = ldk; a.A = dA; a.lda = lda; a.r = (int)r;
  a.n = n; a.npad = npad; a.tiles_r = (int)(npad / GT_R); a.tiles_c = (int)(npad / GT_C);
  const i64 ntiles = (i64)a.tiles_r * a.tiles_c;
  const int nblocks = (int)(ntiles < GRAD_BLOCKS ? ntiles : GRAD_BLOCKS);
  const size_t pbytes = (size_t)nblocks * (G_MAXSLOT + 1) * 8;
  GPS_HIP(h, h->dTmp2.ensure(pbytes));
  a.partial = h->dTmp2.d();
  {
    LaunchScope ls(h, KC_REDUCE, 0.5 * (double)npad * npad * (60.0 + 4.0 * nfeat), 4.0 * (double)npad * npad);
    hipLaunchKernelGGL(grad_kernel, dim3(nblocks), dim3(256), 0, h->stream, a, P);
    GPS_HIP(h, hipGetLastError());
  }
  {
    // (2048 x 161 partial sums used to travel to the host, 2.6 MB per gradient: 48 us of copy at N = 512)
    LaunchScope ls(h, KC_REDUCE, 0.0, (double)pbytes);
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(P.n_slots + 1), dim3(256), 0, h->stream, a.partial, nblocks, P.n_slots, d_sums);
    GPS_HIP(h, hipGetLastError());
  }
  return GPS_OK;
}

int gps_grad_enqueue(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* dX, i64 n,
                     i64 d_all, i64 npad, const double* dKinv, i64 ldk, const double* dA, i64 lda, i64 r,
                     double* d_sums, GradPost* post) {
  int rc = gps_grad_prepare(h, prog, n_nodes, dX, n, d_all, npad, post);
  if (rc) return rc;
  return gps_grad_run(h, *post, n, npad, dKinv, ldk, dA, lda, r, d_sums);
}

void gps_grad_finish(const GradPost& post, const double* sums, double* grad_slots_host, double* grad_noise_host) {
  if (grad_noise_host) *grad_noise_host = sums[G_MAXSLOT];
  for (int s = 0; s < post.n_slots; ++s) {
    // the kernel accumulated the FULL symmetric sum / 2 through c_ij (1 below, 1/2 on the diagonal):
    // 1/2 sum_ij W dK = sum_{i>j} W dK + 1/2 sum_i W_ii dK_ii
    double tot = sums[s];
    if (post.ls_of_slot[s] > 0.0) tot /= post.ls_of_slot[s];       // -2 delta^2 / l_d : delta is already x/l
    grad_slots_host[s] = tot;
  }
}

int gps_launch_grad(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* dX, i64 n,
                    i64 d_all, i64 npad, const double* dKinv, i64 ldk, const double* dA, i64 lda, i64 r,
                    double* grad_slots_host, double* grad_noise_host) {
  if (grad_needs_general(prog, n_nodes))
    return gps_launch_grad_general(h, prog, n_nodes, dX, n, d_all, npad, dKinv, ldk, dA, lda, r, grad_slots_host, grad_noise_host);
  GradPost post;
  GPS_HIP(h, h->dGradSums.ensure((size_t)GPS_GRAD_SUMS * 8));
  int rc = gps_grad_enqueue(h, prog, n_nodes, dX, n, d_all, npad, dKinv, ldk, dA, lda, r, h->dGradSums.d(), &post);
  if (rc) return rc;
  double sums[GPS_GRAD_SUMS];
  GPS_HIP(h, hipMemcpyAsync(sums, h->dGradSums.p, sizeof(sums), hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  gps_grad_finish(post, sums, grad_slots_host, grad_noise_host);
  return GPS_OK;
}
