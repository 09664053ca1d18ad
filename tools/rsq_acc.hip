// Scratch: accuracy of v_rsq_f64 and of one / two Newton steps on top of it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p = x[i];
  double y = __builtin_amdgcn_rsq(p);
  y0[i] = y;
  y = y * (1.5 - 0.5 * p * y * y);
  y1[i] = y;
  y = y * (1.5 - 0.5 * p * y * y);
  y2[i] = y;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-20.0, 20.0);
  for (auto& v : x) v = std::exp2(u(g));
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    long double r = 1.0L / sqrtl((long double)x[i]);
    e0 = fmax(e0, (double)fabsl((a[i] - r) / r));
    e1 = fmax(e1, (double)fabsl((b[i] - r) / r));
    e2 = fmax(e2, (double)fabsl((c[i] - r) / r));
  }
  printf("max rel err: v_rsq_f64 %.3e (2^%.1f)  +1 Newton %.3e  +2 Newton %.3e\n", e0, log2(e0), e1, e2);
  return 0;
}
