"""Scratch: vendor-library fp64 GEMM rate (torch.mm -> rocBLAS / hipBLASLt) for comparison with gemm_nt_f64_kernel."""
import time, torch
torch.manual_seed(0)
for (m, n, k) in [(16384, 4096, 4096), (16384, 16384, 4096), (8192, 8192, 8192), (16384, 1024, 1024), (4096, 4096, 4096)]:
    A = torch.randn(m, k, dtype=torch.float64, device="cuda"); B = torch.randn(n, k, dtype=torch.float64, device="cuda")
    C = torch.randn(m, n, dtype=torch.float64, device="cuda")
    for _ in range(2):
        C.addmm_(A, B.t(), beta=1.0, alpha=-1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        C.addmm_(A, B.t(), beta=1.0, alpha=-1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print("torch addmm fp64 C -= A B^T  M=%d N=%d K=%d: %.3f ms = %.1f TFLOP/s" % (m, n, k, dt * 1e3, 2.0 * m * n * k / dt / 1e12))
