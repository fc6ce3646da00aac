// Round trip "small kernel -> 17 doubles on the host" three ways (development probe; GPU box):
//   A  hipMemcpyAsync D2H into pinned memory + hipStreamSynchronize          (read_scal as it is)
//   B  the kernel stores the block into host-mapped pinned memory, hipStreamSynchronize
//   C  as B plus a sequence word stored last (after a system-scope fence); the host spins on the word
// build: hipcc --offload-arch=gfx950 -O3 -o sync_latency sync_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_work(double* d, int it) { if (threadIdx.x < 17) d[threadIdx.x] = it + threadIdx.x; }
__global__ void k_work_host(double* d, double* hm, int it) {
    if (threadIdx.x < 17) { double v = it + threadIdx.x; d[threadIdx.x] = v; hm[threadIdx.x] = v; }
}
__global__ void k_work_seq(double* d, double* hm, volatile unsigned* seq, unsigned it) {
    if (threadIdx.x < 17) { double v = it + threadIdx.x; d[threadIdx.x] = v; hm[threadIdx.x] = v; }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store((unsigned*)seq, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main() {
    hipStream_t st; CHK(hipStreamCreate(&st));
    double *d, *hp, *hm, *hm_dev; unsigned *sq, *sq_dev;
    CHK(hipMalloc(&d, 64 * sizeof(double)));
    CHK(hipHostMalloc((void**)&hp, 64 * sizeof(double)));
    CHK(hipHostMalloc((void**)&hm, 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    CHK(hipHostMalloc((void**)&sq, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CHK(hipHostGetDevicePointer((void**)&hm_dev, hm, 0));
    CHK(hipHostGetDevicePointer((void**)&sq_dev, sq, 0));
    *sq = 0;
    const int N = 2000;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 1; i <= N; ++i) {
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, i);
            CHK(hipMemcpyAsync(hp, d, 17 * sizeof(double), hipMemcpyDeviceToHost, st));
            CHK(hipStreamSynchronize(st));
            if (hp[0] != i) { std::printf("A mismatch\n"); return 1; }
        }
        double ta = (now() - t0) / N;
        t0 = now();
        for (int i = 1; i <= N; ++i) {
            hipLaunchKernelGGL(k_work_host, dim3(1), dim3(64), 0, st, d, hm_dev, i);
            CHK(hipStreamSynchronize(st));
            if (hm[0] != i) { std::printf("B mismatch\n"); return 1; }
        }
        double tb = (now() - t0) / N;
        t0 = now();
        for (int i = 1; i <= N; ++i) {
            unsigned want = (unsigned)(rep * N + i);
            hipLaunchKernelGGL(k_work_seq, dim3(1), dim3(64), 0, st, d, hm_dev, sq_dev, want);
            long spins = 0;
            while (__atomic_load_n(sq, __ATOMIC_ACQUIRE) != want) { if (++spins > 2000000000L) { std::printf("C timeout\n"); return 1; } }
            if (hm[0] != want) { std::printf("C mismatch %f %u\n", hm[0], want); return 1; }
        }
        double tc = (now() - t0) / N;
        std::printf("round trip per kernel + read-back: A memcpy+sync %.2f us   B mapped store+sync %.2f us   C mapped store + host spin %.2f us\n", ta, tb, tc);
    }
    return 0;
}
