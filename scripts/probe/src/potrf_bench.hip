// Development probe (GPU box): cycles and accuracy of the 64 x 64 diagonal step of the Cholesky (potrf64_body: factor + explicit inverse)
// in isolation, and the accuracy of v_rsq_f64.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I activesetmethods_amd/csrc -o potrf_bench potrf_bench.hip
#include "asm_kernels.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k_bench(double* S, int64_t ldS, int nb, const double* diag0, double thr, double* Linv, long long* cyc) {
    __shared__ double D[ASM_NB * ASM_DP];
    __shared__ double W[ASM_NB * ASM_DP];
    __shared__ potrf_T_t T[4];
    __shared__ double d0[ASM_NB], dinv[ASM_NB];
    double* Sb = S + (int64_t)blockIdx.x * ASM_NB * ldS;
    double* Lb = Linv + (int64_t)blockIdx.x * ASM_NB * ASM_NB;
    __syncthreads();
    const long long t0 = clock64();
    if (V == 0) potrf64_body<true>(D, W, T, d0, dinv, Sb, ldS, 0, nb, diag0, thr, Lb);
    else potrf64_body<false>(D, W, T, d0, dinv, Sb, ldS, 0, nb, diag0, thr, Lb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_rsq(const double* x, double* y0, double* y1, double* y2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double inv = __builtin_amdgcn_rsq(d);
    y0[i] = inv;
    inv = inv * fma(-0.5 * d * inv, inv, 1.5);
    y1[i] = inv;
    inv = inv * fma(-0.5 * d * inv, inv, 1.5);
    y2[i] = inv;
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 64, reps = 64;
    const int64_t ld = 64;
    std::vector<double> S(64 * 64), Bm(64 * 80), diag(64);
    srand(7);
    for (auto& v : Bm) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
            double s = 0;
            for (int k = 0; k < 80; ++k) s += Bm[i * 80 + k] * Bm[j * 80 + k];
            S[i * 64 + j] = s + (i == j ? 1.0 : 0.0);
        }
    for (int i = 0; i < 64; ++i) diag[i] = S[i * 64 + i];
    // host reference (long double)
    std::vector<long double> L(64 * 64, 0.0L), Wi(64 * 64, 0.0L);
    for (int j = 0; j < nb; ++j) {
        long double d = S[j * 64 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 64 + k] * L[j * 64 + k];
        L[j * 64 + j] = sqrtl(d);
        for (int i = j + 1; i < nb; ++i) {
            long double s = S[i * 64 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 64 + k] * L[j * 64 + k];
            L[i * 64 + j] = s / L[j * 64 + j];
        }
    }
    for (int c = 0; c < nb; ++c)
        for (int r = c; r < nb; ++r) {
            long double s = (r == c) ? 1.0L : 0.0L;
            for (int k = c; k < r; ++k) s -= L[r * 64 + k] * Wi[k * 64 + c];
            Wi[r * 64 + c] = s / L[r * 64 + r];
        }
    double *dS, *dLinv, *ddiag;
    long long* dcyc;
    CHK(hipMalloc(&dS, reps * 64 * 64 * sizeof(double)));
    CHK(hipMalloc(&dLinv, reps * 64 * 64 * sizeof(double)));
    CHK(hipMalloc(&ddiag, 64 * sizeof(double)));
    CHK(hipMalloc(&dcyc, reps * sizeof(long long)));
    CHK(hipMemcpy(ddiag, diag.data(), 64 * sizeof(double), hipMemcpyHostToDevice));
    for (int variant = 0; variant < 2; ++variant) {
        long long best = 1LL << 60;
        double eL = 0, eW = 0;
        for (int rep = 0; rep < 6; ++rep) {
            for (int r = 0; r < reps; ++r) CHK(hipMemcpy(dS + r * 64 * 64, S.data(), 64 * 64 * sizeof(double), hipMemcpyHostToDevice));
            CHK(hipMemset(dLinv, 0, reps * 64 * 64 * sizeof(double)));
            // one workgroup per launch (grid 1): the step as it runs on the critical path; several launches for the minimum
            for (int r = 0; r < reps; ++r) {
                if (variant == 0) hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(256), 0, 0, dS + r * 64 * 64, ld, nb, (const double*)ddiag, 1e-14, dLinv + r * 64 * 64, dcyc + r);
                else hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(256), 0, 0, dS + r * 64 * 64, ld, nb, (const double*)ddiag, 1e-14, dLinv + r * 64 * 64, dcyc + r);
            }
            CHK(hipDeviceSynchronize());
            std::vector<long long> cyc(reps);
            CHK(hipMemcpy(cyc.data(), dcyc, reps * sizeof(long long), hipMemcpyDeviceToHost));
            for (auto c : cyc) best = c < best ? c : best;
        }
        std::vector<double> Lg(64 * 64), Wg(64 * 64);
        CHK(hipMemcpy(Lg.data(), dS, 64 * 64 * sizeof(double), hipMemcpyDeviceToHost));
        CHK(hipMemcpy(Wg.data(), dLinv, 64 * 64 * sizeof(double), hipMemcpyDeviceToHost));
        double mL = 0, mW = 0;
        for (int i = 0; i < nb; ++i)
            for (int j = 0; j <= i; ++j) {
                eL = fmax(eL, fabs((double)(Lg[i * 64 + j] - L[i * 64 + j])));
                eW = fmax(eW, fabs((double)(Wg[i * 64 + j] - Wi[i * 64 + j])));
                mL = fmax(mL, fabs((double)L[i * 64 + j]));
                mW = fmax(mW, fabs((double)Wi[i * 64 + j]));
            }
        // upper triangle of the inverse must be exact zeros, rows past nb the identity
        int bad = 0;
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                const double want = (i >= nb || j >= nb) ? (i == j ? 1.0 : 0.0) : (j > i ? 0.0 : (double)Wi[i * 64 + j]);
                if ((j > i || i >= nb || j >= nb) && Wg[i * 64 + j] != want) ++bad;
            }
        std::printf("variant %d (0 = round 3, %d = round 4): nb %d  min cycles %lld  |L - ref| / max %.2e  |W - ref| / max %.2e  structural mismatches %d\n", variant, variant, nb, best, eL / mL, eW / mW, bad);
    }
    // v_rsq_f64
    const int n = 1 << 16;
    std::vector<double> x(n), y0(n), y1(n), y2(n);
    for (int i = 0; i < n; ++i) x[i] = exp((rand() / (double)RAND_MAX - 0.5) * 40.0);
    double *dx, *d0, *d1, *d2;
    CHK(hipMalloc(&dx, n * 8)); CHK(hipMalloc(&d0, n * 8)); CHK(hipMalloc(&d1, n * 8)); CHK(hipMalloc(&d2, n * 8));
    CHK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rsq, dim3(n / 256), dim3(256), 0, 0, (const double*)dx, d0, d1, d2, n);
    CHK(hipMemcpy(y0.data(), d0, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(y1.data(), d1, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(y2.data(), d2, n * 8, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double r = 1.0L / sqrtl((long double)x[i]);
        e0 = fmax(e0, fabs((double)((y0[i] - r) / r))); e1 = fmax(e1, fabs((double)((y1[i] - r) / r))); e2 = fmax(e2, fabs((double)((y2[i] - r) / r)));
    }
    std::printf("v_rsq_f64 max relative error: raw %.3e, after one Newton step %.3e, after two %.3e\n", e0, e1, e2);
    return 0;
}
