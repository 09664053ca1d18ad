// Diagnostics: fp64-MFMA issue-rate microbenchmark (calibrates the roofline denominator on
// the device at hand) and an exact-integer check of the v_mfma_f64_16x16x4_f64 lane maps the
// GEMM kernel relies on (A: [i = lane&15][k = lane>>4], B: [k = lane>>4][j = lane&15],
// C/D: col = lane&15, row = (lane>>4) + 4*reg).
#include "gps_common.hpp"

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void mfma_f64_rate_kernel(double* out, int iters) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + 1e-3 * lane, b = 1.0 - 1e-3 * lane;
  v4d acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;   // keep the chain live
}

// same issue loop with NCH independent accumulator chains per wave: measures the dependent-issue interval
template <int NCH>
__global__ __launch_bounds__(256, 2) void mfma_f64_chain_kernel(double* out, int iters) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + 1e-3 * lane, b = 1.0 - 1e-3 * lane;
  v4d acc[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}

// D = A[16x4] * B[4x16] with asymmetric integer data; D written through the assumed map
__global__ void mfma_f64_layout_kernel(const double* A, const double* B, double* D) {
  const int lane = threadIdx.x & 63;
  const double a = A[(lane & 15) * 4 + (lane >> 4)];
  const double b = B[(lane >> 4) * 16 + (lane & 15)];
  v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}

int gps_run_mfma_diag(gps_handle_t h, int waves_per_simd, double* tflops, int* layout_ok) {
  if (waves_per_simd < 1) waves_per_simd = 1;
  if (waves_per_simd > 2) waves_per_simd = 2;
  // ---- layout ----
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = (double)(i * 4 + k + 1);
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (double)((k + 1) * 100 + j * 3 + 7);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j];
    ref[i * 16 + j] = s;
  }
  GPS_HIP(h, h->dTmp.ensure(4096 * sizeof(double)));
  double* d = h->dTmp.d();
  GPS_HIP(h, hipMemcpyAsync(d, hA, sizeof(hA), hipMemcpyHostToDevice, h->stream));
  GPS_HIP(h, hipMemcpyAsync(d + 64, hB, sizeof(hB), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(mfma_f64_layout_kernel, dim3(1), dim3(64), 0, h->stream, d, d + 64, d + 128);
  GPS_HIP(h, hipGetLastError());
  GPS_HIP(h, hipMemcpyAsync(hD, d + 128, sizeof(hD), hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  int ok = 1;
  for (int i = 0; i < 256; ++i) if (hD[i] != ref[i]) ok = 0;
  if (layout_ok) *layout_ok = ok;

  // ---- rate ----
  const int iters = 4096;
  const int blocks = h->prop.multiProcessorCount * waves_per_simd;
  hipEvent_t e0, e1;
  GPS_HIP(h, hipEventCreate(&e0)); GPS_HIP(h, hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, h->stream, d, 64);   // warm
  GPS_HIP(h, hipEventRecord(e0, h->stream));
  hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, h->stream, d, iters);
  GPS_HIP(h, hipEventRecord(e1, h->stream));
  GPS_HIP(h, hipEventSynchronize(e1));
  float ms = 0.f;
  GPS_HIP(h, hipEventElapsedTime(&ms, e0, e1));
  const double flop = (double)blocks * 4.0 * iters * 8.0 * (2.0 * 16 * 16 * 4);
  if (tflops) *tflops = flop / (ms * 1e-3) / 1e12;
  if (getenv("GPS_DIAG_CHAINS")) {
    // cycles per MFMA for 1, 2, 4 chains in ONE wave on ONE CU (and with every CU busy)
    for (int nch = 1; nch <= 4; nch *= 2) {
      for (int nb = 1; nb <= h->prop.multiProcessorCount; nb *= h->prop.multiProcessorCount) {
        GPS_HIP(h, hipEventRecord(e0, h->stream));
        if (nch == 1) hipLaunchKernelGGL(mfma_f64_chain_kernel<1>, dim3(nb), dim3(64), 0, h->stream, d, iters);
        else if (nch == 2) hipLaunchKernelGGL(mfma_f64_chain_kernel<2>, dim3(nb), dim3(64), 0, h->stream, d, iters);
        else hipLaunchKernelGGL(mfma_f64_chain_kernel<4>, dim3(nb), dim3(64), 0, h->stream, d, iters);
        GPS_HIP(h, hipEventRecord(e1, h->stream));
        GPS_HIP(h, hipEventSynchronize(e1));
        float t = 0.f;
        GPS_HIP(h, hipEventElapsedTime(&t, e0, e1));
        fprintf(stderr, "chains=%d blocks=%d : %.1f ns per MFMA per wave\n", nch, nb, t * 1e6 / ((double)iters * nch));
      }
    }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return GPS_OK;
}

// ---- GEMM timeline: device-resident pseudo-random operands, one launch, per-workgroup stamps ----
__global__ void diag_fill_kernel(double* p, long long n, unsigned seed) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((double)(x & 0xffffff) - 8388608.0) * (1.0 / 8388608.0);
  }
}

int gps_run_gemm_timeline(gps_handle_t h, int op, int lower, i64 m, i64 n, i64 k, int reps, long long* stamps_out,
                          i64 cap_blocks, i64* nblocks, double* ms_out) {
  const size_t ab = (size_t)m * k * 8, bb = (size_t)n * k * 8, cb = (size_t)m * n * 8;
  GPS_HIP(h, h->dTmp.ensure(ab)); GPS_HIP(h, h->dTmp2.ensure(bb)); GPS_HIP(h, h->dTmp3.ensure(cb));
  GPS_HIP(h, h->dGemvWs.ensure((size_t)cap_blocks * 6 * sizeof(long long)));
  hipLaunchKernelGGL(diag_fill_kernel, dim3(2048), dim3(256), 0, h->stream, h->dTmp.d(), (long long)(m * k), 1u);
  hipLaunchKernelGGL(diag_fill_kernel, dim3(2048), dim3(256), 0, h->stream, h->dTmp2.d(), (long long)(n * k), 2u);
  hipLaunchKernelGGL(diag_fill_kernel, dim3(2048), dim3(256), 0, h->stream, h->dTmp3.d(), (long long)(m * n), 3u);
  GPS_HIP(h, hipMemsetAsync(h->dGemvWs.p, 0, (size_t)cap_blocks * 6 * sizeof(long long), h->stream));
  const double* B = (lower == 1) ? h->dTmp.d() : h->dTmp2.d();       // syrk form: B = A
  int rc = GPS_OK;
  for (int i = 0; i < 2 && !rc; ++i) rc = gps_launch_gemm_nt(h, op, lower, m, n, k, h->dTmp.d(), k, B, k, h->dTmp3.d(), n);   // warm
  if (rc) return rc;
  hipEvent_t e0, e1;
  GPS_HIP(h, hipEventCreate(&e0)); GPS_HIP(h, hipEventCreate(&e1));
  GPS_HIP(h, hipEventRecord(e0, h->stream));
  for (int i = 0; i < reps && !rc; ++i) rc = gps_launch_gemm_nt(h, op, lower, m, n, k, h->dTmp.d(), k, B, k, h->dTmp3.d(), n);
  GPS_HIP(h, hipEventRecord(e1, h->stream));
  if (rc) return rc;
  h->gemm_stamps = (long long*)h->dGemvWs.p;
  rc = gps_launch_gemm_nt(h, op, lower, m, n, k, h->dTmp.d(), k, B, k, h->dTmp3.d(), n);
  h->gemm_stamps = nullptr;
  if (rc) return rc;
  GPS_HIP(h, hipMemcpyAsync(stamps_out, h->dGemvWs.p, (size_t)cap_blocks * 6 * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  GPS_HIP(h, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (ms_out) *ms_out = ms / reps;
  i64 nb = 0;
  for (i64 b = 0; b < cap_blocks; ++b) if (stamps_out[6 * b + 1] != 0) nb = b + 1;
  if (nblocks) *nblocks = nb;
  return GPS_OK;
}
