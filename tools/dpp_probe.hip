// Probe (gfx950): issue cost of the fp64 DPP forms (v_fmac_f64_dpp / v_mov_b64_dpp with row_newbcast) against the
// v_readlane + v_fma_f64 broadcast they would replace in the pivot chain of potrf_base, for one wave alone on its SIMD
// and for two waves sharing it.  Build: hipcc -O3 --offload-arch=gfx950 tools/dpp_probe.hip -o tools/bin/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

#define FMAC_DPP(d, a, b, k) asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(a), "v"(b))
#define MOV_DPP(d, a, k) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(a))

// mode 0: 15 independent readlane-broadcast updates per iteration; 1: 15 independent DPP updates; 2: 15 plain fma (no broadcast)
// 3: 15 DEPENDENT dpp fmacs (latency); 4: 15 dependent plain fma; 5: 15 mov_dpp
template <int MODE>
__global__ void probe(double* out, long long* cyc, int iters) {
  double r[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) r[k] = 1.0 + 1e-3 * (threadIdx.x & 15) + 1e-4 * k;
  double s = 1e-9;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 1; k < 16; ++k) { const double l = readlane_f64(r[0], k); r[k] -= s * l; }
    } else if (MODE == 1) {
      FMAC_DPP(r[1], r[0], s, 1); FMAC_DPP(r[2], r[0], s, 2); FMAC_DPP(r[3], r[0], s, 3); FMAC_DPP(r[4], r[0], s, 4);
      FMAC_DPP(r[5], r[0], s, 5); FMAC_DPP(r[6], r[0], s, 6); FMAC_DPP(r[7], r[0], s, 7); FMAC_DPP(r[8], r[0], s, 8);
      FMAC_DPP(r[9], r[0], s, 9); FMAC_DPP(r[10], r[0], s, 10); FMAC_DPP(r[11], r[0], s, 11); FMAC_DPP(r[12], r[0], s, 12);
      FMAC_DPP(r[13], r[0], s, 13); FMAC_DPP(r[14], r[0], s, 14); FMAC_DPP(r[15], r[0], s, 15);
    } else if (MODE == 2) {
#pragma unroll
      for (int k = 1; k < 16; ++k) { r[k] = __builtin_fma(-s, r[0], r[k]); asm volatile("" : "+v"(r[k])); }
    } else if (MODE == 3) {
      FMAC_DPP(r[1], r[1], s, 1); FMAC_DPP(r[1], r[1], s, 2); FMAC_DPP(r[1], r[1], s, 3); FMAC_DPP(r[1], r[1], s, 4);
      FMAC_DPP(r[1], r[1], s, 5); FMAC_DPP(r[1], r[1], s, 6); FMAC_DPP(r[1], r[1], s, 7); FMAC_DPP(r[1], r[1], s, 8);
      FMAC_DPP(r[1], r[1], s, 9); FMAC_DPP(r[1], r[1], s, 10); FMAC_DPP(r[1], r[1], s, 11); FMAC_DPP(r[1], r[1], s, 12);
      FMAC_DPP(r[1], r[1], s, 13); FMAC_DPP(r[1], r[1], s, 14); FMAC_DPP(r[1], r[1], s, 15);
    } else if (MODE == 4) {
#pragma unroll
      for (int k = 1; k < 16; ++k) { r[1] = __builtin_fma(-s, r[1], r[1]); asm volatile("" : "+v"(r[1])); }
    } else {
      MOV_DPP(r[1], r[0], 1); MOV_DPP(r[2], r[0], 2); MOV_DPP(r[3], r[0], 3); MOV_DPP(r[4], r[0], 4);
      MOV_DPP(r[5], r[0], 5); MOV_DPP(r[6], r[0], 6); MOV_DPP(r[7], r[0], 7); MOV_DPP(r[8], r[0], 8);
      MOV_DPP(r[9], r[0], 9); MOV_DPP(r[10], r[0], 10); MOV_DPP(r[11], r[0], 11); MOV_DPP(r[12], r[0], 12);
      MOV_DPP(r[13], r[0], 13); MOV_DPP(r[14], r[0], 14); MOV_DPP(r[15], r[0], 15);
    }
  }
  const long long t1 = clock64();
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc += r[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// correctness of the broadcast: lane L of row q must receive lane (16 q + k)'s value
__global__ void check(double* out) {
  double v = 100.0 * (threadIdx.x >> 4) + (threadIdx.x & 15), d = 0.0, one = -1.0;
  FMAC_DPP(d, v, one, 5);          // d += bcast(v, 5) * -(-1)
  double m;
  MOV_DPP(m, v, 9);
  out[threadIdx.x] = d; out[64 + threadIdx.x] = m;
}

// Do the fp64 vector FMAs of one wave slow the fp64 MFMAs of ANOTHER wave on the same SIMD?  512 threads: wave `valu_wave`
// (or none, -1) runs the DPP FMA stream, every other wave a chain of v_mfma_f64_16x16x4; cycles per MFMA and the SIMD id
// (HW_REG_HW_ID bits 5:4) per wave.
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void contend(double* out, long long* cyc, int* simd, int iters, int valu_wave, int valu_wave2) {
  const int wave = threadIdx.x >> 6;
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if ((threadIdx.x & 63) == 0) simd[wave] = (hwid >> 4) & 3;
  double r[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) r[k] = 1.0 + 1e-3 * (threadIdx.x & 15) + 1e-4 * k;
  double s = 1e-9;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  __syncthreads();
  const long long t0 = clock64();
  if (wave == valu_wave || wave == valu_wave2) {
    for (int it = 0; it < iters; ++it) {
      FMAC_DPP(r[1], r[0], s, 1); FMAC_DPP(r[2], r[0], s, 2); FMAC_DPP(r[3], r[0], s, 3); FMAC_DPP(r[4], r[0], s, 4);
      FMAC_DPP(r[5], r[0], s, 5); FMAC_DPP(r[6], r[0], s, 6); FMAC_DPP(r[7], r[0], s, 7); FMAC_DPP(r[8], r[0], s, 8);
      FMAC_DPP(r[9], r[0], s, 9); FMAC_DPP(r[10], r[0], s, 10); FMAC_DPP(r[11], r[0], s, 11); FMAC_DPP(r[12], r[0], s, 12);
      FMAC_DPP(r[13], r[0], s, 13); FMAC_DPP(r[14], r[0], s, 14); FMAC_DPP(r[15], r[0], s, 15); FMAC_DPP(r[1], r[0], s, 3);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(r[u], r[4 + u], acc, 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  double a2 = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
  for (int k = 0; k < 16; ++k) a2 += r[k];
  out[threadIdx.x] = a2;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

static void run_contend(int valu_wave, int valu_wave2) {
  double* out; long long* cyc; int* simd;
  hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64); hipMalloc(&simd, 32);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(contend, dim3(1), dim3(512), 0, 0, out, cyc, simd, iters, valu_wave, valu_wave2);
  hipDeviceSynchronize();
  long long h[8]; int sd[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost); hipMemcpy(sd, simd, 32, hipMemcpyDeviceToHost);
  printf("VALU waves %d,%d:", valu_wave, valu_wave2);
  for (int w = 0; w < 8; ++w) {
    const bool v = (w == valu_wave || w == valu_wave2);
    printf("  w%d(simd %d) %s %.1f", w, sd[w], v ? "cyc/fma" : "cyc/mfma", (double)h[w] / iters / (v ? 16.0 : 4.0));
  }
  printf("\n");
  hipFree(out); hipFree(cyc); hipFree(simd);
}

template <int MODE>
static void run(const char* name, int threads) {
  double* out; long long* cyc;
  hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-34s threads %4d: %.2f cycles per instruction-group of 15 -> %.2f per op (wave 0)\n", name, threads, (double)h[0] / iters,
         (double)h[0] / iters / 15.0);
  hipFree(out); hipFree(cyc);
}

int main() {
  double* o; hipMalloc(&o, 1024);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, o);
  std::vector<double> h(128);
  hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) { const double e5 = 100.0 * (t >> 4) + 5, e9 = 100.0 * (t >> 4) + 9; if (h[t] != e5 || h[64 + t] != e9) ++bad; }
  printf("row_newbcast semantics: %s (lane 20: fmac %.0f mov %.0f)\n", bad ? "MISMATCH" : "ok", h[20], h[84]);
  run_contend(-1, -1); run_contend(0, -1); run_contend(0, 1); run_contend(0, 4);
  for (int threads : {64}) {
    run<0>("readlane x2 + fma (independent)", threads);
    run<1>("v_fmac_f64_dpp (independent)", threads);
    run<2>("v_fma_f64 plain (independent)", threads);
    run<3>("v_fmac_f64_dpp (dependent)", threads);
    run<4>("v_fma_f64 plain (dependent)", threads);
    run<5>("v_mov_b64_dpp (independent)", threads);
  }
  return 0;
}
