// Scratch: how many small dependent-kernel chains does one MI355X run side by side?  T host threads, one non-blocking stream each,
// N tiny kernels per thread with a stream synchronisation every SYNC launches.  Prints the aggregate launch rate.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_tiny(double* x, int n, int work) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double v = x[i];
        for (int k = 0; k < work; ++k) v = v * 1.0000001 + 1e-9;
        x[i] = v;
    }
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 20000, SYNC = argc > 2 ? atoi(argv[2]) : 25, wgs = argc > 3 ? atoi(argv[3]) : 1, work = argc > 4 ? atoi(argv[4]) : 64;
    for (int T : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(T);
        std::vector<double*> buf(T);
        for (int t = 0; t < T; ++t) {
            hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking);
            hipMalloc(&buf[t], sizeof(double) * 256 * wgs);
            hipMemset(buf[t], 0, sizeof(double) * 256 * wgs);
        }
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                for (int i = 0; i < N; ++i) {
                    k_tiny<<<wgs, 256, 0, st[t]>>>(buf[t], 256 * wgs, work);
                    if ((i + 1) % SYNC == 0) hipStreamSynchronize(st[t]);
                }
                hipStreamSynchronize(st[t]);
            });
        for (auto& x : th) x.join();
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %d  wgs %d  sync every %d: %.0f kernels/s aggregate, %.2f us per kernel per stream\n", T, wgs, SYNC, T * N / s, 1e6 * s / N);
        for (int t = 0; t < T; ++t) { hipStreamDestroy(st[t]); hipFree(buf[t]); }
    }
    return 0;
}
