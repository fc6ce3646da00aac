"""What the vendor library (rocBLAS / hipBLASLt through torch.mm, float64) sustains on the shapes of the Cholesky trailing update
(C[M,M] -= P[M,K] P[M,K]'), as a practical yardstick beside the 78.6 TFLOP/s pipe rate.  Development probe; GPU box.
A full M x M GEMM does twice the flops of the triangular update; rates below are for the flops each call actually executes."""
import time, torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
for M, K in ((17408, 1024), (9216, 1024), (4096, 1024), (17408, 256), (8192, 8192)):
    A = torch.randn(M, K, dtype=torch.float64, device=dev)
    C = torch.randn(M, M, dtype=torch.float64, device=dev)
    for _ in range(2): C.addmm_(A, A.T, beta=1.0, alpha=-1.0)
    torch.cuda.synchronize()
    reps = 6
    t0 = time.perf_counter()
    for _ in range(reps): C.addmm_(A, A.T, beta=1.0, alpha=-1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("M %6d K %5d  %8.3f ms  %6.2f TFLOP/s" % (M, K, dt * 1e3, 2.0 * M * M * K / dt / 1e12), flush=True)
    del A, C
