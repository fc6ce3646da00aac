// Development probe (GPU box): what one wavefront alone pays per instruction on gfx950 - FP64 FMA (dependent / independent), v_readlane_b32 -> SGPR -> FP64
// FMA, wave-uniform ds_read_b128 + FMA, v_rsq_f64, DPP row broadcast.   build: hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__global__ __launch_bounds__(64) void k_cost(double* out, long long* cyc, int which) {
    __shared__ double lds[64];
    const int lane = threadIdx.x;
    lds[lane] = 1.0 + lane * 1e-3;
    __syncthreads();
    double a0 = out[lane], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double m = 1.0000001, l = a0 * 0.5;
    const long long t0 = clock64();
    if (which == 0) {            // dependent FP64 FMA chain
        REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a0) : "v"(m));)
    } else if (which == 1) {     // 8 independent FP64 FMAs
        REP16(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                           "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (which == 2) {     // lane read (2 x v_readlane_b32) -> FMA with the scalar pair, independent accumulators (the chain's update)
        REP16(asm volatile("v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 3\n s_nop 1\n v_fma_f64 %0, -%10, s[20:21], %0\n"
                           "v_readlane_b32 s22, %8, 4\n v_readlane_b32 s23, %9, 4\n s_nop 1\n v_fma_f64 %1, -%10, s[22:23], %1\n"
                           "v_readlane_b32 s20, %8, 5\n v_readlane_b32 s21, %9, 5\n s_nop 1\n v_fma_f64 %2, -%10, s[20:21], %2\n"
                           "v_readlane_b32 s22, %8, 6\n v_readlane_b32 s23, %9, 6\n s_nop 1\n v_fma_f64 %3, -%10, s[22:23], %3"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                           : "v"(__double2loint(l)), "v"(__double2hiint(l)), "v"(l) : "s20", "s21", "s22", "s23");)
    } else if (which == 3) {     // lane reads only
        REP64(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3" : : "v"(__double2loint(l)), "v"(__double2hiint(l)) : "s20", "s21");)
    } else if (which == 4) {     // wave-uniform LDS read of two doubles + two FMAs
        const unsigned addr = 0;
        double2 v;
        REP16(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
              asm volatile("v_fma_f64 %0, -%2, %3, %0\n v_fma_f64 %1, -%2, %4, %1" : "+v"(a0), "+v"(a1) : "v"(l), "v"(v.x), "v"(v.y));)
    } else if (which == 5) {     // LDS reads issued back to back, one wait
        const unsigned addr = 0;
        double2 v0, v1, v2, v3;
        REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                           : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory");
              asm volatile("v_fma_f64 %0, -%2, %3, %0\n v_fma_f64 %1, -%2, %4, %1" : "+v"(a0), "+v"(a1) : "v"(l), "v"(v0.x), "v"(v1.y));
              asm volatile("v_fma_f64 %0, -%2, %3, %0\n v_fma_f64 %1, -%2, %4, %1" : "+v"(a2), "+v"(a3) : "v"(l), "v"(v2.x), "v"(v3.y));)
    } else if (which == 6) {     // dependent v_rsq_f64
        REP64(asm volatile("v_rsq_f64 %0, %0" : "+v"(a0));)
    } else if (which == 7) {     // the pivot chain of one column: rsq, mul, fma, mul + fma, fma, mul, fma (all dependent)
        REP16(asm volatile("v_rsq_f64 %1, %0\n v_mul_f64 %2, %0, %1\n v_fma_f64 %2, -%2, %1, 1.0\n v_mul_f64 %3, %1, %2\n v_fma_f64 %2, %2, %4, 0.5\n v_fma_f64 %1, %3, %2, %1\n"
                           "v_mul_f64 %3, %1, %0\n v_fma_f64 %0, -%3, %3, %0"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));)
    } else if (which == 8) {     // 32-bit VALU, independent
        float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
        REP64(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0001f));)
        a0 += f0 + f1 + f2 + f3;
    } else if (which == 9) {     // FP64 FMA with a constant scalar-pair operand, independent accumulators
        REP16(asm volatile("v_fma_f64 %0, -%8, s[20:21], %0\n v_fma_f64 %1, -%8, s[20:21], %1\n v_fma_f64 %2, -%8, s[20:21], %2\n v_fma_f64 %3, -%8, s[20:21], %3\n"
                           "v_fma_f64 %4, -%8, s[20:21], %4\n v_fma_f64 %5, -%8, s[20:21], %5\n v_fma_f64 %6, -%8, s[20:21], %6\n v_fma_f64 %7, -%8, s[20:21], %7"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(l) : "s20", "s21");)
    }
    const long long t1 = clock64();
    out[lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + l;
    if (lane == 0) cyc[which] = t1 - t0;
}

int main() {
    double* d; long long* c;
    CHK(hipMalloc(&d, 64 * 8)); CHK(hipMalloc(&c, 16 * 8));
    CHK(hipMemset(d, 0, 64 * 8));
    const char* name[10] = {"dependent v_fma_f64 (64)", "independent v_fma_f64 (128)", "2 v_readlane + s_nop 1 + v_fma_f64 (64 groups)", "v_readlane_b32 (128)",
                            "ds_read_b128 + wait + 2 v_fma_f64 (16 groups)", "4 ds_read_b128 + wait + 4 v_fma_f64 (16 groups)", "dependent v_rsq_f64 (64)",
                            "pivot chain of one column, 8 dependent ops (16 columns)", "independent v_fma_f32 (256)", "v_fma_f64 with scalar-pair operand (128)"};
    const double per[10] = {64, 128, 64, 128, 16, 16, 64, 16, 256, 128};
    for (int rep = 0; rep < 2; ++rep)
        for (int w = 0; w < 10; ++w) {
            hipLaunchKernelGGL(k_cost, dim3(1), dim3(64), 0, 0, d, c, w);
            CHK(hipDeviceSynchronize());
            long long cy;
            CHK(hipMemcpy(&cy, c + w, 8, hipMemcpyDeviceToHost));
            if (rep) std::printf("%-62s %7lld cycles = %6.1f per unit\n", name[w], cy, cy / per[w]);
        }
    return 0;
}
