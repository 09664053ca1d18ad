"""Scratch: vendor fp64 GEMM rate over a sustained run (thermal / power behaviour) vs short bursts."""
import time, torch
m, n, k = 16384, 16384, 4096
A = torch.randn(m, k, dtype=torch.float64, device="cuda"); B = torch.randn(n, k, dtype=torch.float64, device="cuda")
C = torch.randn(m, n, dtype=torch.float64, device="cuda")
for _ in range(2): C.addmm_(A, B.t(), beta=1.0, alpha=-1.0)
torch.cuda.synchronize()
for burst in range(8):
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps): C.addmm_(A, B.t(), beta=1.0, alpha=-1e-9)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print("burst %d: %.2f ms = %.1f TFLOP/s" % (burst, dt * 1e3, 2.0 * m * n * k / dt / 1e12))
