// Probe (gfx950): what a lone workgroup's waves pay for streaming 16-byte-per-lane global stores (1 KB per wave instruction):
// cycles per store instruction with 1, 2, 4, 8 storing waves of one 512-thread workgroup, data from registers.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void st(double2* out, long long* cyc, int iters, int nstoring, int stride_rows) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double2 v; v.x = threadIdx.x; v.y = 1.0;
  __syncthreads();
  const long long t0 = clock64();
  if (wave < nstoring) {
    double2* p = out + (size_t)wave * iters * 64 * stride_rows + lane;
    for (int it = 0; it < iters; ++it) { p[(size_t)it * 64 * stride_rows] = v; v.x += 1.0; }
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[wave] = t1 - t0;
}
int main() {
  double2* out; long long* cyc;
  hipMalloc(&out, (size_t)1 << 30); hipMalloc(&cyc, 64);
  for (int stride : {1, 16}) for (int ns : {1, 2, 4, 8}) {
    const int iters = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(st, dim3(1), dim3(512), 0, 0, out, cyc, iters, ns, stride);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("row stride %2d KB, %d storing waves: %.0f cycles per 1-KB store instruction (wave 0), %.1f B/cycle for the CU\n", stride, ns,
           (double)h[0] / iters, 1024.0 * ns * iters / (double)h[0]);
  }
  return 0;
}
