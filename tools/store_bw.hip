// Scratch: store-bandwidth ceilings for the kernel-matrix build (N x N fp64, lower triangle in 128-blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef long long i64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// 64x64 tile per WG, 4x4 patch per thread (the product's pattern)
__global__ __launch_bounds__(256) void fill_4x4(double* K, i64 ld, int lower) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (lower && (tj >> 1) > (ti >> 1)) return;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  for (int q = 0; q < 4; ++q) {
    double* dst = K + ((i64)ti * 64 + ty * 4 + q) * ld + (i64)tj * 64 + tx * 4;
    *reinterpret_cast<double2*>(dst) = make_double2(1.0, 2.0);
    *reinterpret_cast<double2*>(dst + 2) = make_double2(3.0, 4.0);
  }
}
// 64 rows x 128 cols per WG: a wave writes one full 1 KB row segment per store (16 B per lane), 16 rows per wave
__global__ __launch_bounds__(256) void fill_rows(double* K, i64 ld, int lower) {
  const int ti = blockIdx.y, tj = blockIdx.x;          // 64-row x 128-col tiles
  if (lower && tj > (ti >> 1)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = 0; q < 16; ++q) {
    double* dst = K + ((i64)ti * 64 + wave * 16 + q) * ld + (i64)tj * 128 + lane * 2;
    *reinterpret_cast<double2*>(dst) = make_double2(1.0, 2.0);
  }
}
// same with non-temporal stores
__global__ __launch_bounds__(256) void fill_rows_nt(double* K, i64 ld, int lower) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (lower && tj > (ti >> 1)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = 0; q < 16; ++q) {
    double* dst = K + ((i64)ti * 64 + wave * 16 + q) * ld + (i64)tj * 128 + lane * 2;
    __builtin_nontemporal_store(1.0, dst);
    __builtin_nontemporal_store(2.0, dst + 1);
  }
}
// persistent: 2048 WGs walk the 128x128 blocks of the lower triangle, each wave writes 1 KB rows
__global__ __launch_bounds__(256) void fill_persist(double* K, i64 ld, int nb) {
  const i64 nblk = (i64)nb * (nb + 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (i64 b = blockIdx.x; b < nblk; b += gridDim.x) {
    int bi = 0; i64 rem = b;
    while (rem >= bi + 1) { rem -= bi + 1; ++bi; }
    const int bj = (int)rem;
    for (int q = 0; q < 32; ++q) {
      double* dst = K + ((i64)bi * 128 + wave * 32 + q) * ld + (i64)bj * 128 + lane * 2;
      *reinterpret_cast<double2*>(dst) = make_double2(1.0, 2.0);
    }
  }
}

int main() {
  const i64 n = 32768;
  double* K; CK(hipMalloc(&K, n * n * 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms;
  auto time = [&](const char* name, double bytes, auto&& f) {
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s %.3f ms  %.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
  };
  const double full = (double)n * n * 8, low = full / 2 + (double)n * 128 * 8 / 2;
  time("memset full", full, [&] { CK(hipMemsetAsync(K, 0, n * n * 8)); });
  time("fill_4x4 full", full, [&] { hipLaunchKernelGGL(fill_4x4, dim3(512, 512), dim3(256), 0, 0, K, n, 0); });
  time("fill_4x4 lower", low, [&] { hipLaunchKernelGGL(fill_4x4, dim3(512, 512), dim3(256), 0, 0, K, n, 1); });
  time("fill_rows full", full, [&] { hipLaunchKernelGGL(fill_rows, dim3(256, 512), dim3(256), 0, 0, K, n, 0); });
  time("fill_rows lower", low, [&] { hipLaunchKernelGGL(fill_rows, dim3(256, 512), dim3(256), 0, 0, K, n, 1); });
  time("fill_rows_nt lower", low, [&] { hipLaunchKernelGGL(fill_rows_nt, dim3(256, 512), dim3(256), 0, 0, K, n, 1); });
  time("fill_persist lower", low, [&] { hipLaunchKernelGGL(fill_persist, dim3(2048), dim3(256), 0, 0, K, n, 256); });
  time("fill_persist lower 4096wg", low, [&] { hipLaunchKernelGGL(fill_persist, dim3(4096), dim3(256), 0, 0, K, n, 256); });
  return 0;
}
