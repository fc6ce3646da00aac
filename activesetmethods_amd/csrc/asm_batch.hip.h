// Host side of the scenario batch (device side: asm_bt.hip.h).  SURVEY.md section 8 rows b / e; the reference has one model per
// optimizer and no batching (src/MOI_wrapper.jl:1093-1152), so this is the build's own design for BASELINE.json's scenario batch.
//
// B scenarios are solved by B *fibers* (user-level contexts) of ONE host thread.  Each fiber runs the ordinary single-scenario solver
// code on its own handle; what changes is what a launch does:
//   * outside a fiber     hipLaunchKernelGGL / hipMemcpyAsync / ... go to the stream as before (bt.tab = nullptr);
//   * inside a fiber      the operation is RECORDED (kernel, grid, packed arguments; copies become copy-kernel operations whose
//                         payload travels in the round's blob), and the fiber yields when it needs a result on the host
//                         (hipStreamSynchronize, read-backs) or reaches an alignment point (barrier).
// When every fiber is blocked the scheduler MERGES the recorded lists position by position: operations of different fibers that are
// the same kernel with the same grid become one launch with the scenario index in gridDim.z and an argument table in HBM (one H2D
// copy of all tables + payloads per round), on one stream; a completion word in host-mapped memory ends the round.  Scenarios that
// leave the common path simply stop matching: their operations are launched on their own, in order.  Results are bit-identical to
// the per-scenario path by construction (same kernels, same arguments, same grids).
// Alignment: fibers wait at barriers (tags in program order); the scheduler releases the group with the smallest (cycle, tag) first
// once nobody can run - min-PC-first reconvergence, as a SIMT machine treats a divergent loop.
#pragma once
#include <sys/mman.h>
#include <ucontext.h>
#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "asm_bt.hip.h"

// copy / fill operations of recorded streams (every hipMemcpyAsync / hipMemsetAsync inside a fiber becomes one of these)
__global__ __launch_bounds__(256) void k_bcopy(AsmBt bt, char* dst, const char* src, int64_t bytes) {
    ASM_BARGS(bt, dst, src, bytes);
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const uintptr_t al = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
    if ((al & 15) == 0) {
        for (int64_t i = t0; i < bytes / 16; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    } else if ((al & 7) == 0) {
        for (int64_t i = t0; i < bytes / 8; i += stride) reinterpret_cast<uint64_t*>(dst)[i] = reinterpret_cast<const uint64_t*>(src)[i];
    } else if ((al & 3) == 0) {
        for (int64_t i = t0; i < bytes / 4; i += stride) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
    } else {
        for (int64_t i = t0; i < bytes; i += stride) dst[i] = src[i];
    }
}
__global__ __launch_bounds__(256) void k_bfill(AsmBt bt, char* dst, int value, int64_t bytes) {
    ASM_BARGS(bt, dst, value, bytes);
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const unsigned v = (unsigned)(value & 0xff) * 0x01010101u;
    const uintptr_t al = (uintptr_t)dst | (uintptr_t)bytes;
    if ((al & 15) == 0) {
        for (int64_t i = t0; i < bytes / 16; i += stride) reinterpret_cast<uint4*>(dst)[i] = make_uint4(v, v, v, v);
    } else if ((al & 3) == 0) {
        for (int64_t i = t0; i < bytes / 4; i += stride) reinterpret_cast<uint32_t*>(dst)[i] = v;
    } else {
        for (int64_t i = t0; i < bytes; i += stride) dst[i] = (char)value;
    }
}
// last operation of a round: everything before it on the stream has finished when the word arrives on the host
__global__ void k_bsignal(AsmBt bt, unsigned* word, unsigned value) {
    ASM_BARGS(bt, word, value);
    __threadfence_system();
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

namespace asmb {

struct BatchError : std::runtime_error {
    explicit BatchError(const std::string& s) : std::runtime_error(s) {}
};

// ---- argument packing: the layout rule of asm_bget (natural alignment, declaration order, entry rounded up to 16 bytes)
template <class T>
constexpr size_t pk_align(size_t off) { return (off + alignof(T) - 1) & ~(alignof(T) - 1); }
template <class... KA>
constexpr size_t pk_size() {
    size_t off = 0;
    ((off = pk_align<KA>(off) + sizeof(KA)), ...);
    return (off + 15) & ~(size_t)15;
}
template <class... KA>
inline void pk_write(char* dst, const KA&... a) {
    size_t off = 0;
    ((off = pk_align<KA>(off), std::memcpy(dst + off, &a, sizeof(KA)), off += sizeof(KA)), ...);
}
template <class T>
inline T conv(T v) { return v; }

enum { OP_RESIDENT = 1 };      // every workgroup of the launch must be resident at once (dataflow panel kernels)

struct Op {
    const void* kfn;
    void (*thunk)(const void* kfn, dim3 g, dim3 b, unsigned sh, hipStream_t s, AsmBt bt);
    unsigned gx, gy, gz, bx, by, bz, shmem, flags;
    uint32_t arg_off, arg_size;            // packed arguments in the fiber's argument arena
    int32_t src_payload;                   // >= 0: H2D copy - argument `src` (offset 8) is patched to the payload's place in the round's blob
    uint32_t payload_bytes;
    void* h_dst;                           // != nullptr: D2H copy - argument `dst` (offset 0) is patched to a place in the out-staging buffer
    uint32_t out_bytes;
};

struct Sched;

struct Fiber {
    enum State { RUNNABLE, WAIT_FLUSH, WAIT_BARRIER, DONE };
    ucontext_t ctx;
    void* stack = nullptr;
    size_t stack_size = 0;
    State state = RUNNABLE;
    long cycle = 0;
    int tag = 0;
    int index = 0;
    Sched* sched = nullptr;
    std::vector<Op> ops;
    std::vector<char> args, payload;
    std::function<void()> body;
    std::exception_ptr err;
    // merge cursor / per-round scratch
    size_t cursor = 0;
    std::vector<uint32_t> out_off;
};

extern thread_local Fiber* cur;
thread_local Fiber* cur = nullptr;

inline bool in_fiber() { return cur != nullptr; }

struct Sched {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<Fiber*> fibers;
    ucontext_t main_ctx;
    char *h_blob = nullptr, *d_blob = nullptr;
    size_t blob_cap = 0;
    char *h_out = nullptr, *d_out = nullptr;
    size_t out_cap = 0;
    unsigned *h_sig = nullptr, *d_sig = nullptr;
    unsigned round = 0;
    int panel_wgs = 480;
    bool use_barriers = true;
    // statistics
    uint64_t n_rounds = 0, n_ops = 0, n_launches = 0, n_releases = 0, blob_bytes = 0;
    double t_emit_ms = 0, t_wait_ms = 0, t_host_ms = 0;
    bool verbose = false;                                   // ASM_BATCH_VERBOSE=1: per-kernel merge statistics at release
    // HIP events around every merged launch of the dataflow panel kernels (OP_RESIDENT) on this group's stream: the family bench.py's roofline names
    bool time_resident = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> res_events;
    std::vector<hipEvent_t> res_pool;
    double res_ms = 0.0;
    uint64_t res_launches = 0, res_ops = 0;
    hipEvent_t res_event() {
        if (!res_pool.empty()) { hipEvent_t e = res_pool.back(); res_pool.pop_back(); return e; }
        hipEvent_t e;
        chk(hipEventCreate(&e), "hipEventCreate");
        return e;
    }
    void resolve_resident() {                               // (the stream is idle: every round ends with a completion wait)
        for (auto& pr : res_events) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) res_ms += ms;
            res_pool.push_back(pr.first); res_pool.push_back(pr.second);
        }
        res_events.clear();
    }
    std::map<const void*, std::pair<uint64_t, uint64_t>> per_kernel;      // kernel -> (operations recorded, launches made)

    static void chk(hipError_t e, const char* what) {
        if (e != hipSuccess) throw BatchError(std::string(what) + ": " + hipGetErrorString(e));
    }
    void reserve_blob(size_t need) {
        if (need <= blob_cap) return;
        if (h_blob) (void)hipHostFree(h_blob);
        if (d_blob) (void)hipFree(d_blob);
        blob_cap = std::max<size_t>(need * 2, (size_t)4 << 20);
        chk(hipHostMalloc((void**)&h_blob, blob_cap), "hipHostMalloc(blob)");
        chk(hipMalloc((void**)&d_blob, blob_cap), "hipMalloc(blob)");
    }
    void reserve_out(size_t need) {
        if (need <= out_cap) return;
        if (h_out) (void)hipHostFree(h_out);
        out_cap = std::max<size_t>(need * 2, (size_t)1 << 20);
        chk(hipHostMalloc((void**)&h_out, out_cap, hipHostMallocMapped), "hipHostMalloc(out)");
        chk(hipHostGetDevicePointer((void**)&d_out, h_out, 0), "hipHostGetDevicePointer(out)");
    }
    void init(int dev, hipStream_t s, int pwgs) {
        device = dev; stream = s; panel_wgs = pwgs;
        if (const char* v = std::getenv("ASM_BATCH_VERBOSE")) verbose = v[0] == '1';
        if (const char* v = std::getenv("ASM_HIP_TIMING")) time_resident = v[0] != '0'; else time_resident = true;      // as for a handle: the panel family is timed by default
        chk(hipHostMalloc((void**)&h_sig, 64, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc(sig)");
        chk(hipHostGetDevicePointer((void**)&d_sig, h_sig, 0), "hipHostGetDevicePointer(sig)");
        *h_sig = 0;
        reserve_blob((size_t)4 << 20);
        reserve_out((size_t)1 << 20);
    }
    void release() {
        if (verbose && !per_kernel.empty()) {
            std::vector<std::pair<uint64_t, const void*>> v;
            for (auto& kv : per_kernel) v.push_back({kv.second.second, kv.first});
            std::sort(v.rbegin(), v.rend());
            std::fprintf(stderr, "[asm batch] rounds %llu, launches by kernel (launches, operations, operations per launch):\n", (unsigned long long)n_rounds);
            for (size_t i = 0; i < v.size() && i < 40; ++i) {
                const auto& pk = per_kernel[v[i].second];
                const char* nm = hipKernelNameRefByPtr(v[i].second, stream);
                std::fprintf(stderr, "[asm batch]   %8llu %9llu %6.1f  %.60s\n", (unsigned long long)pk.second, (unsigned long long)pk.first, (double)pk.first / (double)pk.second, nm ? nm : "?");
            }
        }
        for (Fiber* f : fibers) {
            if (f->stack) munmap(f->stack, f->stack_size);
            delete f;
        }
        fibers.clear();
        if (h_blob) (void)hipHostFree(h_blob);
        if (d_blob) (void)hipFree(d_blob);
        if (h_out) (void)hipHostFree(h_out);
        if (h_sig) (void)hipHostFree(h_sig);
        h_blob = d_blob = h_out = d_out = nullptr;
        h_sig = d_sig = nullptr;
        blob_cap = out_cap = 0;
    }

    static void trampoline(unsigned lo, unsigned hi) {
        Fiber* f = reinterpret_cast<Fiber*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
        try {
            f->body();
        } catch (...) {
            f->err = std::current_exception();
        }
        f->state = Fiber::DONE;
        cur = nullptr;
        swapcontext(&f->ctx, &f->sched->main_ctx);      // never resumed
    }
    Fiber* add_fiber(std::function<void()> body) {
        Fiber* f = new Fiber();
        f->sched = this;
        f->index = (int)fibers.size();
        f->body = std::move(body);
        f->stack_size = (size_t)2 << 20;
        f->stack = mmap(nullptr, f->stack_size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (f->stack == MAP_FAILED) { f->stack = nullptr; delete f; throw BatchError("mmap of a fiber stack failed"); }
        getcontext(&f->ctx);
        f->ctx.uc_stack.ss_sp = f->stack;
        f->ctx.uc_stack.ss_size = f->stack_size;
        f->ctx.uc_link = nullptr;
        const uintptr_t p = (uintptr_t)f;
        makecontext(&f->ctx, (void (*)())trampoline, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
        fibers.push_back(f);
        return f;
    }
    void resume(Fiber* f) {
        cur = f;
        f->state = Fiber::RUNNABLE;
        swapcontext(&main_ctx, &f->ctx);
        cur = nullptr;
    }
    static double now_ms() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    }

    static bool same(const Op& a, const Op& b) {
        return a.kfn == b.kfn && a.gx == b.gx && a.gy == b.gy && a.gz == b.gz && a.bx == b.bx && a.by == b.by && a.bz == b.bz && a.shmem == b.shmem &&
               a.arg_size == b.arg_size;
    }

    // merge and launch the recorded operations of `fl`, wait for the round to finish, hand the read-backs to their destinations
    void emit(std::vector<Fiber*>& fl) {
        const double t0 = now_ms();
        // ---- sizes: tables + payloads (blob), read-back staging (out)
        size_t need_blob = 0, need_out = 0;
        for (Fiber* f : fl) {
            for (const Op& o : f->ops) {
                need_blob += o.arg_size + 16;
                if (o.src_payload >= 0) need_blob += ((size_t)o.payload_bytes + 15) & ~(size_t)15;
                if (o.h_dst) need_out += ((size_t)o.out_bytes + 15) & ~(size_t)15;
            }
            f->cursor = 0;
            f->out_off.assign(f->ops.size(), 0);
        }
        reserve_blob(need_blob + 64);
        reserve_out(need_out + 64);
        // ---- payloads first, then one table per merged launch
        size_t bo = 0, oo = 0;
        for (Fiber* f : fl)
            for (size_t i = 0; i < f->ops.size(); ++i) {
                Op& o = f->ops[i];
                if (o.src_payload >= 0) {
                    std::memcpy(h_blob + bo, f->payload.data() + o.src_payload, o.payload_bytes);
                    const char* src = d_blob + bo;
                    std::memcpy(f->args.data() + o.arg_off + 8, &src, 8);
                    bo += ((size_t)o.payload_bytes + 15) & ~(size_t)15;
                }
                if (o.h_dst) {
                    char* dst = d_out + oo;
                    std::memcpy(f->args.data() + o.arg_off, &dst, 8);
                    f->out_off[i] = (uint32_t)oo;
                    oo += ((size_t)o.out_bytes + 15) & ~(size_t)15;
                }
            }
        struct Launch { const Op* op; size_t tab; unsigned nb; };
        std::vector<Launch> launches;
        std::vector<Fiber*> group;
        for (;;) {
            Fiber* lead = nullptr;                       // the least advanced fiber leads (keeps the cursors together)
            for (Fiber* f : fl)
                if (f->cursor < f->ops.size() && (!lead || f->cursor < lead->cursor)) lead = f;
            if (!lead) break;
            const Op& X = lead->ops[lead->cursor];
            size_t cap = fl.size();
            if (X.flags & OP_RESIDENT) cap = std::max<size_t>(1, (size_t)panel_wgs / std::max(1u, X.gx * X.gy * X.gz));
            cap = std::min<size_t>(cap, 65535u / std::max(1u, X.gz));
            group.clear();
            for (Fiber* f : fl)
                if (group.size() < cap && f->cursor < f->ops.size() && (f == lead || same(f->ops[f->cursor], X))) group.push_back(f);
            bo = (bo + 15) & ~(size_t)15;
            launches.push_back({&X, bo, (unsigned)group.size()});
            if (verbose) { auto& pk = per_kernel[X.kfn]; pk.first += group.size(); pk.second += 1; }
            for (Fiber* f : group) {
                std::memcpy(h_blob + bo, f->args.data() + f->ops[f->cursor].arg_off, X.arg_size);
                bo += X.arg_size;
                f->cursor += 1;
            }
        }
        // ---- one copy of the blob, then the launches, then the completion word
        if (bo > 0) chk(hipMemcpyAsync(d_blob, h_blob, bo, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(blob)");
        for (const Launch& L : launches) {
            const Op& X = *L.op;
            AsmBt bt{(const void*)(d_blob + L.tab), X.arg_size, X.gz};
            const bool timed = time_resident && (X.flags & OP_RESIDENT);
            hipEvent_t ea = nullptr, eb = nullptr;
            if (timed) { ea = res_event(); eb = res_event(); chk(hipEventRecord(ea, stream), "hipEventRecord"); }
            X.thunk(X.kfn, dim3(X.gx, X.gy, X.gz * L.nb), dim3(X.bx, X.by, X.bz), X.shmem, stream, bt);
            if (timed) { chk(hipEventRecord(eb, stream), "hipEventRecord"); res_events.push_back({ea, eb}); res_launches += 1; res_ops += L.nb; }
        }
        round += 1;
        if (round == 0) round = 1;
        k_bsignal<<<dim3(1), dim3(1), 0, stream>>>(AsmBt{nullptr, 0, 1}, d_sig, round);
        chk(hipGetLastError(), "launch");
        const double t1 = now_ms();
        wait_round();
        const double t2 = now_ms();
        if (!res_events.empty()) resolve_resident();
        for (Fiber* f : fl) {
            for (size_t i = 0; i < f->ops.size(); ++i)
                if (f->ops[i].h_dst) std::memcpy(f->ops[i].h_dst, h_out + f->out_off[i], f->ops[i].out_bytes);
            n_ops += f->ops.size();
            f->ops.clear(); f->args.clear(); f->payload.clear();
        }
        n_rounds += 1;
        n_launches += launches.size() + 1;
        blob_bytes += bo;
        t_emit_ms += t1 - t0;
        t_wait_ms += t2 - t1;
    }
    void wait_round() {
        const double t0 = now_ms();
        for (unsigned long spins = 1;; ++spins) {
            if (__atomic_load_n(h_sig, __ATOMIC_ACQUIRE) == round) return;
            __builtin_ia32_pause();
            if ((spins & 0xfffff) == 0 && now_ms() - t0 > 60000.0) {
                chk(hipStreamSynchronize(stream), "hipStreamSynchronize");      // a device fault surfaces here
                if (__atomic_load_n(h_sig, __ATOMIC_ACQUIRE) == round) return;
                throw BatchError("batch round: the stream finished without its completion word");
            }
        }
    }

    // run every fiber to completion
    void run() {
        std::vector<Fiber*> fl;
        for (;;) {
            bool any_live = false, progressed = false;
            for (Fiber* f : fibers) {
                if (f->state == Fiber::RUNNABLE) {
                    const double t0 = now_ms();
                    resume(f);
                    t_host_ms += now_ms() - t0;
                    progressed = true;
                }
                any_live = any_live || f->state != Fiber::DONE;
            }
            if (!any_live) break;
            fl.clear();
            for (Fiber* f : fibers)
                if (f->state == Fiber::WAIT_FLUSH) fl.push_back(f);
            if (!fl.empty()) {
                emit(fl);
                for (Fiber* f : fl) f->state = Fiber::RUNNABLE;
                continue;
            }
            // everybody waits at a barrier (or is done): release the group with the smallest (cycle, tag)
            Fiber* mn = nullptr;
            for (Fiber* f : fibers)
                if (f->state == Fiber::WAIT_BARRIER && (!mn || f->cycle < mn->cycle || (f->cycle == mn->cycle && f->tag < mn->tag))) mn = f;
            if (!mn) {
                if (!progressed) throw BatchError("batch scheduler: no fiber can run");
                continue;
            }
            for (Fiber* f : fibers)
                if (f->state == Fiber::WAIT_BARRIER && f->cycle == mn->cycle && f->tag == mn->tag) f->state = Fiber::RUNNABLE;
            n_releases += 1;
        }
        // a fiber that ended with recorded operations still pending (it should not): launch them so that the stream is complete
        fl.clear();
        for (Fiber* f : fibers)
            if (!f->ops.empty()) fl.push_back(f);
        if (!fl.empty()) emit(fl);
        for (Fiber* f : fibers)
            if (f->err) std::rethrow_exception(f->err);
    }
};

// ---------------------------------------------------------------------------------------------- what the solver code calls
inline void yield_to_scheduler(Fiber* f) {
    cur = nullptr;
    swapcontext(&f->ctx, &f->sched->main_ctx);
}
// the fiber needs the results of everything it has recorded
inline void flush_wait() {
    Fiber* f = cur;
    if (!f) return;
    f->state = Fiber::WAIT_FLUSH;
    yield_to_scheduler(f);
}
// alignment point `tag` (tags grow in program order inside one cycle); no-op outside a batch
inline void barrier(int tag) {
    Fiber* f = cur;
    if (!f || !f->sched->use_barriers) return;
    f->tag = tag;
    f->state = Fiber::WAIT_BARRIER;
    yield_to_scheduler(f);
}
inline void next_cycle() {
    if (cur) cur->cycle += 1;
}

template <class... KA>
void thunk(const void* kfn, dim3 g, dim3 b, unsigned sh, hipStream_t s, AsmBt bt) {
    auto k = reinterpret_cast<void (*)(AsmBt, KA...)>(const_cast<void*>(kfn));
    k<<<g, b, sh, s>>>(bt, KA{}...);
}

template <class... KA>
inline Op& record(Fiber* f, void (*kern)(AsmBt, KA...), dim3 g, dim3 b, unsigned sh, unsigned flags, const KA&... a) {
    constexpr size_t sz = pk_size<KA...>();
    const size_t off = f->args.size();
    f->args.resize(off + sz);
    pk_write<KA...>(f->args.data() + off, a...);
    Op o;
    o.kfn = (const void*)kern;
    o.thunk = &thunk<KA...>;
    o.gx = g.x; o.gy = g.y; o.gz = g.z; o.bx = b.x; o.by = b.y; o.bz = b.z; o.shmem = sh; o.flags = flags;
    o.arg_off = (uint32_t)off; o.arg_size = (uint32_t)sz;
    o.src_payload = -1; o.payload_bytes = 0; o.h_dst = nullptr; o.out_bytes = 0;
    f->ops.push_back(o);
    return f->ops.back();
}

template <class... KA, class... A>
inline void launch(void (*kern)(AsmBt, KA...), dim3 g, dim3 b, unsigned sh, hipStream_t st, A&&... a) {
    static_assert(sizeof...(KA) == sizeof...(A), "kernel launch: argument count does not match the kernel's parameter list");
    if (g.x == 0 || g.y == 0 || g.z == 0) return;
    Fiber* f = cur;
    if (!f) {
        kern<<<g, b, sh, st>>>(AsmBt{nullptr, 0, g.z}, conv<KA>(a)...);
        return;
    }
    record<KA...>(f, kern, g, b, sh, 0u, conv<KA>(a)...);
}
template <class... KA, class... A>
inline void launch_resident(void (*kern)(AsmBt, KA...), dim3 g, dim3 b, unsigned sh, hipStream_t st, A&&... a) {
    static_assert(sizeof...(KA) == sizeof...(A), "kernel launch: argument count does not match the kernel's parameter list");
    Fiber* f = cur;
    if (!f) {
        kern<<<g, b, sh, st>>>(AsmBt{nullptr, 0, g.z}, conv<KA>(a)...);
        return;
    }
    record<KA...>(f, kern, g, b, sh, (unsigned)OP_RESIDENT, conv<KA>(a)...);
}

inline unsigned copy_grid(size_t bytes) { return (unsigned)std::min<size_t>(64, std::max<size_t>(1, (bytes + 16383) / 16384)); }

inline hipError_t memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    Fiber* f = cur;
    if (!f) {
        // outside a batch: device-to-device copies of the solver's vectors (4 - 150 KB, ~60 per LP) as a copy kernel of this library - the
        // runtime's blit path brackets each copy with its own barrier / signal packets
        static const bool own = [] { const char* v = std::getenv("ASM_HIP_OWN_COPIES"); return !(v && v[0] == '0'); }();
        if (own && kind == hipMemcpyDeviceToDevice && bytes > 0 && bytes <= ((size_t)64 << 20)) {
            k_bcopy<<<dim3((unsigned)std::min<size_t>(256, std::max<size_t>(1, (bytes + 16383) / 16384))), dim3(256), 0, st>>>(AsmBt{nullptr, 0, 1}, (char*)dst, (const char*)src, (int64_t)bytes);
            return hipGetLastError();
        }
        return ::hipMemcpyAsync(dst, src, bytes, kind, st);
    }
    if (bytes == 0) return hipSuccess;
    if (kind == hipMemcpyHostToDevice) {
        const size_t po = (f->payload.size() + 15) & ~(size_t)15;
        f->payload.resize(po + bytes);
        std::memcpy(f->payload.data() + po, src, bytes);
        Op& o = record<char*, const char*, int64_t>(f, k_bcopy, dim3(copy_grid(bytes)), dim3(256), 0, 0u, (char*)dst, (const char*)nullptr, (int64_t)bytes);
        o.src_payload = (int32_t)po;
        o.payload_bytes = (uint32_t)bytes;
    } else if (kind == hipMemcpyDeviceToHost) {
        Op& o = record<char*, const char*, int64_t>(f, k_bcopy, dim3(copy_grid(bytes)), dim3(256), 0, 0u, (char*)nullptr, (const char*)src, (int64_t)bytes);
        o.h_dst = dst;
        o.out_bytes = (uint32_t)bytes;
    } else if (kind == hipMemcpyDeviceToDevice) {
        record<char*, const char*, int64_t>(f, k_bcopy, dim3(copy_grid(bytes)), dim3(256), 0, 0u, (char*)dst, (const char*)src, (int64_t)bytes);
    } else {
        return hipErrorInvalidValue;
    }
    return hipSuccess;
}
inline hipError_t memset_async(void* dst, int value, size_t bytes, hipStream_t st) {
    Fiber* f = cur;
    if (!f) {
        static const bool own = [] { const char* v = std::getenv("ASM_HIP_OWN_COPIES"); return !(v && v[0] == '0'); }();
        if (own && bytes > 0 && bytes <= ((size_t)64 << 20)) {
            k_bfill<<<dim3((unsigned)std::min<size_t>(256, std::max<size_t>(1, (bytes + 16383) / 16384))), dim3(256), 0, st>>>(AsmBt{nullptr, 0, 1}, (char*)dst, value, (int64_t)bytes);
            return hipGetLastError();
        }
        return ::hipMemsetAsync(dst, value, bytes, st);
    }
    if (bytes == 0) return hipSuccess;
    record<char*, int, int64_t>(f, k_bfill, dim3(copy_grid(bytes)), dim3(256), 0, 0u, (char*)dst, value, (int64_t)bytes);
    return hipSuccess;
}
inline hipError_t stream_synchronize(hipStream_t st) {
    if (!cur) return ::hipStreamSynchronize(st);
    flush_wait();
    return hipSuccess;
}
inline hipError_t memcpy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    Fiber* f = cur;
    if (!f) return ::hipMemcpy(dst, src, bytes, kind);
    hipError_t e = memcpy_async(dst, src, bytes, kind, nullptr);
    flush_wait();
    return e;
}
inline hipError_t memset_sync(void* dst, int value, size_t bytes) {
    Fiber* f = cur;
    if (!f) return ::hipMemset(dst, value, bytes);
    hipError_t e = memset_async(dst, value, bytes, nullptr);
    flush_wait();
    return e;
}
inline hipError_t free_sync(void* p) {
    if (cur) flush_wait();      // recorded operations may still use the buffer
    return ::hipFree(p);
}

}  // namespace asmb

// From here on the solver code's stream operations go through the recorder (a no-op layer outside a fiber)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) ::asmb::launch(kernel, grid, block, (unsigned)(shmem), stream, ##__VA_ARGS__)
#define hipMemcpyAsync(...) ::asmb::memcpy_async(__VA_ARGS__)
#define hipMemsetAsync(...) ::asmb::memset_async(__VA_ARGS__)
#define hipStreamSynchronize(...) ::asmb::stream_synchronize(__VA_ARGS__)
#define hipMemcpy(...) ::asmb::memcpy_sync(__VA_ARGS__)
#define hipMemset(...) ::asmb::memset_sync(__VA_ARGS__)
#define hipFree(...) ::asmb::free_sync(__VA_ARGS__)
