// Forward / backward substitution  L a = y  and  L^T a = y  (densities.py:81-82, models/gpr.py:123; the gradient's
// K^-1 y) as ONE launch: a wavefront over the 128-row blocks of L.
//
// The recursive trsv (blocked.hpp::trsv_rec) is 4 N / 128 launch-latency-bound kernels (511 at N = 32768, 1.3 TB/s).
// Here workgroup t takes block row i = t of L (i = nblk-1-t for L^T) in the order of a ticket counter, streams that
// block row once (HBM-bound part, all workgroups at the same time) and consumes the solution blocks a_j of the rows
// before it as they appear.  The hand-over is the data itself: the exchange buffer starts as a NaN pattern no result
// can have (TW_EMPTY); the 128 threads that need a_j poll its 128 words with device-scope loads, the producer stores
// them with device-scope stores -- one fabric round trip per block on the critical chain, no flag, no fence.
// A workgroup only ever waits for tickets drawn before its own, so the launch cannot deadlock whatever the number of
// resident workgroups; the wait is bounded by the wall clock all the same (TW_POISON travels down the chain and the
// host falls back to the recursive path).
//
// Per workgroup: 512 threads = 8 waves.  L block (128 x 128, row-major) of block column j:
//   forward : wave w owns rows 16w..16w+15, lane l columns 2l, 2l+1; 16 partial sums per lane stay in registers over
//             the whole block row and are reduced across lanes once (reduce-scatter butterfly: 17 shuffles);
//   backward: wave w owns rows 16w..16w+15 of block j, lane l OUTPUT columns 2l, 2l+1; two sums per lane, reduced
//             across the 8 waves once through LDS.
// The diagonal block is applied through its explicit inverse (potrf_base's W = L_ii^-1, or W^T), parked in LDS at the
// start (128 KB: this also pins one workgroup per CU, which spreads the block rows over all CUs).
#include "gps_common.hpp"

typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

static constexpr u64 TW_EMPTY = 0x7ff8dead00000001ull;    // not written yet
static constexpr u64 TW_POISON = 0x7ff8dead00000002ull;   // a producer gave up: give up too

__device__ __forceinline__ u64 tw_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tw_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ctl[0] = ticket, ctl[1] = number of workgroups that gave up (sticky until the host reads it)
__global__ __launch_bounds__(256) void tw_init_kernel(u64* __restrict__ xch, i64 count, unsigned* __restrict__ ctl) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i < count) xch[i] = TW_EMPTY;
  if (i == 0) ctl[0] = 0u;
}

// REFINE: the diagonal solve a_i = W t is followed by one step of iterative refinement against the diagonal block of the factor
// itself,  a_i += W (t - L_ii a_i)  (L_ii^T for the transposed solve) -- what trsv_leaf_refine_kernel does for the leaves of the
// recursive substitution (trsm_leaf.hip), here inside the wavefront: jittered / low-noise factors no longer fall back
// to the 4 N / 128 launches of the recursion.  The 32 entries of L_ii a thread needs sit in registers from the start.
template <int RC, bool TRANS, bool REFINE>
__global__ __launch_bounds__(512) void trsv_wave_kernel(const double* __restrict__ L, i64 ldl, int nblk,
                                                        const double* __restrict__ Wall, double* __restrict__ y,
                                                        i64 ldy, int r0, u64* __restrict__ xch,
                                                        unsigned* __restrict__ ctl) {
  extern __shared__ double smem[];
  double* Ws = smem;                          // [128][128]   Ws[c][o]: out[o] = sum_c Ws[c][o] t[c]
  double* ys = Ws + 128 * 128;                // [2][RC][128] the solution block being consumed (double-buffered)
  double* tv = ys + 2 * RC * 128;             // [RC][128]    right-hand side of the diagonal solve
  double* red = tv + RC * 128;                // [8][RC][128] cross-wave partial sums (backward) / [4][RC][128] W product
  double* a0s = red + 8 * RC * 128;           // [RC][128]    (REFINE) first solution / residual
  __shared__ int s_blk, s_bad;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) { s_blk = (int)atomicAdd(ctl, 1u); s_bad = 0; }
  __syncthreads();
  const int t = s_blk;
  const int i = TRANS ? nblk - 1 - t : t;     // block row (forward) / block column (backward) of this workgroup
  const i64 n = (i64)nblk * 128;
  const int oc = tid & 127, part = tid >> 7;
  // diagonal-block inverse -> registers now, LDS once the first panel loads are in flight
  double wr[32];
  {
    const double* Wp = Wall + (i64)i * 128 * 128 + (i64)(32 * part) * 128 + oc;
#pragma unroll
    for (int k = 0; k < 32; ++k) wr[k] = Wp[k * 128];
  }
  // (REFINE) this thread's 32 entries of the diagonal block: row oc of L_ii (forward) / column oc (backward), k = 32 part ..
  // (fetched right BEHIND the panel loop -- the first product with W and two barriers lie between the loads and their use --:
  // held from the start they cost the forward two-right-hand-side form 124 bytes of scratch; round 6)
  double lr[REFINE ? 32 : 1];
  // first block of the panel:  forward j = 0..i-1 (block column j of block row i);  backward j = nblk-1..i+1 (block row j
  // of block column i).  Step s of the loop is block jj(s).
  const int steps = t;
  const double* Lp;          // element (row 16w of the block, column 2 lane) of block jj(0)
  i64 jstride;               // from block jj(s) to jj(s+1)
  if (!TRANS) { Lp = L + ((i64)i * 128 + 16 * w) * ldl + 2 * lane; jstride = 128; }
  else        { Lp = L + ((i64)(nblk - 1) * 128 + 16 * w) * ldl + (i64)i * 128 + 2 * lane; jstride = -128 * ldl; }
  v2d cur[16], nxt[16];
  if (steps > 0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) cur[u] = *reinterpret_cast<const v2d*>(Lp + (i64)u * ldl);
  }
  const bool poller = tid < 128 * RC;
  const int pq = tid >> 7;                                      // right-hand side this poller serves
  const u64* xp = xch + (i64)pq * n + (tid & 127);              // + 128 * block
  u64 yv = TW_EMPTY;
  if (poller && steps > 0) yv = tw_load(xp + (i64)(TRANS ? nblk - 1 : 0) * 128);
#pragma unroll
  for (int k = 0; k < 32; ++k) Ws[(32 * part + k) * 128 + oc] = wr[k];

  // this block's right-hand side: fetched now, needed after the panel (a load there would sit on the chain)
  const int brow = TRANS ? (tid & 127) : 16 * w + ((lane & 32) ? 8 : 0) + ((lane & 16) ? 4 : 0) + ((lane & 8) ? 2 : 0) + ((lane & 4) ? 1 : 0);
  double bval[RC];
#pragma unroll
  for (int q = 0; q < RC; ++q) bval[q] = 0.0;
  if (TRANS) { if (poller) bval[0] = y[(i64)(r0 + pq) * ldy + (i64)i * 128 + brow]; }
  else if ((lane & 3) == 0) {
#pragma unroll
    for (int q = 0; q < RC; ++q) bval[q] = y[(i64)(r0 + q) * ldy + (i64)i * 128 + brow];
  }
  double acc[RC][TRANS ? 2 : 16];
#pragma unroll
  for (int q = 0; q < RC; ++q)
#pragma unroll
    for (int u = 0; u < (TRANS ? 2 : 16); ++u) acc[q][u] = 0.0;

#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    const int j = TRANS ? nblk - 1 - s : s;
    if (s + 1 < steps) {
      const double* Ln = Lp + (i64)(s + 1) * jstride;
#pragma unroll
      for (int u = 0; u < 16; ++u) nxt[u] = *reinterpret_cast<const v2d*>(Ln + (i64)u * ldl);
    }
    if (poller) {
      if (yv == TW_EMPTY) {
        const u64 t0 = wall_clock64();
        for (;;) {
          yv = tw_load(xp + (i64)j * 128);
          if (yv != TW_EMPTY) break;
          if (wall_clock64() - t0 > 200000000ull) { yv = TW_POISON; break; }     // 2 s at 100 MHz
        }
      }
      if (yv == TW_POISON) s_bad = 1;
      ys[((s & 1) * RC + pq) * 128 + (tid & 127)] = __longlong_as_double((long long)yv);
      yv = TW_EMPTY;
      if (s + 1 < steps) yv = tw_load(xp + (i64)(TRANS ? j - 1 : j + 1) * 128);   // usually there already
    }
    __syncthreads();
    const double* yb = ys + (s & 1) * RC * 128;
    if (!TRANS) {
#pragma unroll
      for (int q = 0; q < RC; ++q) {
        const v2d v = *reinterpret_cast<const v2d*>(yb + q * 128 + 2 * lane);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[q][u] = fma(cur[u].x, v.x, fma(cur[u].y, v.y, acc[q][u]));
      }
    } else {
#pragma unroll
      for (int q = 0; q < RC; ++q)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const double v = yb[q * 128 + 16 * w + u];
          acc[q][0] = fma(cur[u].x, v, acc[q][0]);
          acc[q][1] = fma(cur[u].y, v, acc[q][1]);
        }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
  }

  if (REFINE) {
    const double* Dp = L + (i64)i * 128 * ldl + (i64)i * 128;
#pragma unroll
    for (int k = 0; k < 32; ++k) lr[k] = TRANS ? Dp[(i64)(32 * part + k) * ldl + oc] : Dp[(i64)oc * ldl + 32 * part + k];
  }
  // ---- t = y_i - (sum over the panel)
  if (!TRANS) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int q = 0; q < RC; ++q) {
      double a8[8], a4[4], a2[2], a1;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double keep = b5 ? acc[q][u + 8] : acc[q][u], send = b5 ? acc[q][u] : acc[q][u + 8];
        a8[u] = keep + __shfl_xor(send, 32, 64);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double keep = b4 ? a8[u + 4] : a8[u], send = b4 ? a8[u] : a8[u + 4];
        a4[u] = keep + __shfl_xor(send, 16, 64);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const double keep = b3 ? a4[u + 2] : a4[u], send = b3 ? a4[u] : a4[u + 2];
        a2[u] = keep + __shfl_xor(send, 8, 64);
      }
      {
        const double keep = b2 ? a2[1] : a2[0], send = b2 ? a2[0] : a2[1];
        a1 = keep + __shfl_xor(send, 4, 64);
      }
      a1 += __shfl_xor(a1, 2, 64);
      a1 += __shfl_xor(a1, 1, 64);
      if ((lane & 3) == 0) tv[q * 128 + brow] = bval[q] - a1;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int q = 0; q < RC; ++q) {
      red[(w * RC + q) * 128 + 2 * lane] = acc[q][0];
      red[(w * RC + q) * 128 + 2 * lane + 1] = acc[q][1];
    }
    __syncthreads();
    if (poller) {
      double sum = 0.0;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) sum += red[(ww * RC + pq) * 128 + (tid & 127)];
      tv[pq * 128 + brow] = bval[0] - sum;
    }
    __syncthreads();
  }
  // ---- a_i = W t : thread (oc, part) takes 32 of the 128 terms
  double ps[RC];
#pragma unroll
  for (int q = 0; q < RC; ++q) ps[q] = 0.0;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const double wv = Ws[(32 * part + k) * 128 + oc];
#pragma unroll
    for (int q = 0; q < RC; ++q) ps[q] = fma(wv, tv[q * 128 + 32 * part + k], ps[q]);
  }
#pragma unroll
  for (int q = 0; q < RC; ++q) red[(part * RC + q) * 128 + oc] = ps[q];
  __syncthreads();
  double a = 0.0;
  const int o = tid & 127;
  if (poller) a = (red[(0 * RC + pq) * 128 + o] + red[(1 * RC + pq) * 128 + o]) + (red[(2 * RC + pq) * 128 + o] + red[(3 * RC + pq) * 128 + o]);
  if (REFINE) {
    // residual  rr = t - L_ii a  (fixed order of additions), then  a += W rr
    if (poller) a0s[pq * 128 + o] = a;
    __syncthreads();
    double pr[RC];
#pragma unroll
    for (int q = 0; q < RC; ++q) pr[q] = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
#pragma unroll
      for (int q = 0; q < RC; ++q) pr[q] = fma(lr[k], a0s[q * 128 + 32 * part + k], pr[q]);
    }
    __syncthreads();                                                   // (everybody has read the first product's partial sums)
#pragma unroll
    for (int q = 0; q < RC; ++q) red[(part * RC + q) * 128 + oc] = pr[q];
    __syncthreads();
    if (poller) {
      const double la = (red[(0 * RC + pq) * 128 + o] + red[(1 * RC + pq) * 128 + o]) + (red[(2 * RC + pq) * 128 + o] + red[(3 * RC + pq) * 128 + o]);
      a0s[pq * 128 + o] = tv[pq * 128 + o] - la;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RC; ++q) ps[q] = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const double wv = Ws[(32 * part + k) * 128 + oc];
#pragma unroll
      for (int q = 0; q < RC; ++q) ps[q] = fma(wv, a0s[q * 128 + 32 * part + k], ps[q]);
    }
#pragma unroll
    for (int q = 0; q < RC; ++q) red[(part * RC + q) * 128 + oc] = ps[q];
    __syncthreads();
    if (poller) a += (red[(0 * RC + pq) * 128 + o] + red[(1 * RC + pq) * 128 + o]) + (red[(2 * RC + pq) * 128 + o] + red[(3 * RC + pq) * 128 + o]);
  }
  if (poller) {
    u64 bits = (u64)__double_as_longlong(a);
    if (bits == TW_EMPTY || bits == TW_POISON) bits = 0x7ff8000000000000ull;      // cannot come out of arithmetic; be safe
    if (s_bad) { bits = TW_POISON; a = __longlong_as_double((long long)0x7ff8000000000000ull); }
    tw_store(xch + (i64)pq * n + (i64)i * 128 + o, bits);
    y[(i64)(r0 + pq) * ldy + (i64)i * 128 + o] = a;
  }
  if (tid == 0 && s_bad) atomicAdd(ctl + 1, 1u);
}

template <int RC, bool TRANS, bool REFINE>
static int tw_launch(gps_handle_t h, const double* L, i64 ldl, int nblk, const double* W, double* y, i64 ldy, int r0,
                     u64* xch, unsigned* ctl) {
  const size_t lds = (size_t)(128 * 128 + 2 * RC * 128 + RC * 128 + 8 * RC * 128 + RC * 128) * 8;
  int rca = gps_dyn_lds(h, reinterpret_cast<const void*>(&trsv_wave_kernel<RC, TRANS, REFINE>), (int)lds);      // (per handle = per device)
  if (rca) return rca;
  const i64 n = (i64)nblk * 128;
  hipLaunchKernelGGL(tw_init_kernel, dim3((unsigned)((n * RC + 255) / 256)), dim3(256), 0, h->stream, xch, n * RC, ctl);
  GPS_HIP(h, hipGetLastError());
  LaunchScope ls(h, KC_TRSV, 2.0 * n * n / 2 * RC, (double)n * n / 2 * 8.0);
  hipLaunchKernelGGL((trsv_wave_kernel<RC, TRANS, REFINE>), dim3((unsigned)nblk), dim3(512), lds, h->stream, L, ldl, nblk, W, y,
                     ldy, r0, xch, ctl);
  GPS_HIP(h, hipGetLastError());
  return GPS_OK;
}

// L a = y (trans 0, W = transposed block inverses) or L^T a = y (trans 1, W = block inverses) in place, r right-hand
// sides as rows y[q * ldy + .]; n = 128 nblk.  Two right-hand sides share one pass over L.
template <bool REFINE>
static int tw_pick(gps_handle_t h, const double* L, i64 ldl, int nblk, const double* W, double* y, i64 ldy, int r0, u64* xch,
                   unsigned* ctl, bool two, int trans) {
  if (two) return trans ? tw_launch<2, true, REFINE>(h, L, ldl, nblk, W, y, ldy, r0, xch, ctl) : tw_launch<2, false, REFINE>(h, L, ldl, nblk, W, y, ldy, r0, xch, ctl);
  return trans ? tw_launch<1, true, REFINE>(h, L, ldl, nblk, W, y, ldy, r0, xch, ctl) : tw_launch<1, false, REFINE>(h, L, ldl, nblk, W, y, ldy, r0, xch, ctl);
}

int gps_launch_trsv_wave(gps_handle_t h, const double* L, i64 ldl, i64 n, const double* W, double* y, i64 ldy, i64 r,
                         int trans, int refine) {
  if (n <= 0 || r <= 0) return GPS_OK;
  if (n % 128) return gps_fail(h, GPS_ERR_ARG, "trsv wavefront: n must be a multiple of 128");
  const int nblk = (int)(n / 128);
  GPS_HIP(h, h->dWave.ensure((size_t)2 * n * 8));
  if (!h->dWaveCtl.p) {                                 // control words: allocated and cleared once, persistent afterwards
    GPS_HIP(h, h->dWaveCtl.ensure(256));
    GPS_HIP(h, hipMemsetAsync(h->dWaveCtl.p, 0, 256, h->stream));
  }
  unsigned* ctl = (unsigned*)h->dWaveCtl.p;
  u64* xch = (u64*)h->dWave.p;
  for (i64 r0 = 0; r0 < r; r0 += 2) {
    const int rc = refine ? tw_pick<true>(h, L, ldl, nblk, W, y, ldy, (int)r0, xch, ctl, r - r0 >= 2, trans)
                          : tw_pick<false>(h, L, ldl, nblk, W, y, ldy, (int)r0, xch, ctl, r - r0 >= 2, trans);
    if (rc) return rc;
  }
  // diagnostics ("wave_fault_inject" = k): the k-th substitution from now leaves the give-up counter set, as a workgroup
  // whose bounded wait ran out would (tests/test_gpu_kernels.py::test_wavefront_give_up_is_retried_by_the_gradient)
  if (h->wave_fault_inject > 0 && --h->wave_fault_inject == 0)
    GPS_HIP(h, hipMemsetAsync((unsigned char*)(ctl + 1), 1, 1, h->stream));
  return GPS_OK;
}
