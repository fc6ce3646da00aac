// Scenario batches (SURVEY.md section 8 rows b / e: "batch variants with a leading scenario dimension", "batched kernel grid"):
// ONE launch of any kernel of this library can serve several scenarios.  The reference has no batching at all (one Optimizer <->
// one Model <-> one SLP object, src/MOI_wrapper.jl:1093-1152); this is the build's own design for BASELINE.json's scenario batch.
//
// Every __global__ kernel takes a leading `AsmBt bt` and starts with ASM_BARGS(bt, <its parameters>):
//   bt.tab == nullptr   an ordinary launch: the parameters are the kernel arguments, nothing changes;
//   bt.tab != nullptr   a merged launch of nb scenarios: gridDim.z = bt.gz * nb, the scenario of a workgroup is blockIdx.z / bt.gz, and
//                       its parameters are read from entry `scenario` of an argument table in HBM (entries of bt.stride bytes, each
//                       parameter at its natural alignment in declaration order - asm_batch.hip.h writes them with the same rule).
// gridDim.x / gridDim.y are the same for every scenario of a merged launch (the launcher only merges launches with equal grids), so
// kernels may keep using them; kernels that use blockIdx.z read asm_bz(bt) instead.
// The table is read through the constant address space: uniform address + constant memory = scalar loads (s_load_dword*), so the
// parameters stay in SGPRs exactly like kernel arguments.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct AsmBt {
    const void* tab;        // nullptr: ordinary launch
    unsigned stride;        // bytes per table entry
    unsigned gz;            // gridDim.z of ONE scenario (>= 1)
};

#define ASM_CONSTANT_AS __attribute__((address_space(4)))

#define ASM_GLOBAL_AS __attribute__((address_space(1)))
// A table entry can only hold pointers into global memory (never LDS / scratch).  Plain pointer parameters are told so explicitly;
// pointers inside parameter structs are promoted by the backend (loads of pointers through a kernel argument from unclobbered
// memory, AMDGPUPromoteKernelArguments) - either way the kernels keep global_load / global_store and their vmcnt-only waits instead
// of flat instructions.
template <class T>
__device__ __forceinline__ void asm_bget(const char* e, unsigned& off, T& out) {
    static_assert(sizeof(T) % 4 == 0 && alignof(T) <= 8 && alignof(T) >= 4, "kernel parameters are 4- or 8-byte aligned plain data");
    off = (off + (unsigned)alignof(T) - 1u) & ~((unsigned)alignof(T) - 1u);
    using U = __remove_restrict(T);
    if constexpr (__is_pointer(U)) {
        const U p = *reinterpret_cast<const U*>(e + off);
        using E = __remove_pointer(U);
        out = (U)(ASM_GLOBAL_AS E*)p;
    } else {
        __builtin_memcpy((void*)&out, __builtin_assume_aligned(e + off, alignof(T)), sizeof(T));
    }
    off += (unsigned)sizeof(T);
}
template <class... T>
__device__ __forceinline__ void asm_bload(const AsmBt& bt, T&... a) {
    const unsigned sc = blockIdx.z / bt.gz;
    // through the constant address space: uniform address + constant memory = scalar loads
    const char* e = (const char*)((const char ASM_CONSTANT_AS*)bt.tab + (size_t)sc * bt.stride);
    unsigned off = 0;
    (asm_bget(e, off, a), ...);
}

#define ASM_BARGS(bt, ...)                          \
    do {                                            \
        if ((bt).tab) asm_bload((bt), __VA_ARGS__); \
    } while (0)

// blockIdx.z of the scenario's own grid
__device__ __forceinline__ unsigned asm_bz(const AsmBt& bt) { return blockIdx.z % bt.gz; }
