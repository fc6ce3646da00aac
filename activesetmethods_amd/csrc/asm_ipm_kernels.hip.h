// Device-resident interior-point iteration (oracle/lp_solver.py: class IPM).  All O(M+n) state lives in HBM; the
// host only sequences launches and reads back a handful of scalars per iteration.  Reductions run in a single
// 1024-thread workgroup with a fixed tree, so every scalar is deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct IpmPtrs {
    // problem data (scaled LP)
    const double *q, *lb, *ub, *r, *w, *slo, *scoef;
    const int *rtype, *srow, *rs0, *rs1;      // rs0/rs1: the (up to two) slack columns of a row, -1 if none
    // iterate
    double *p, *s, *g, *y, *tL, *tU, *muL, *muU, *ts, *mus, *pi;
    // residuals / work
    double *act, *aty, *rp, *rdp, *rds, *thp_inv, *ths_inv, *dS, *hp, *hs, *tmpn, *t1, *rhs, *res;
    double *rcL, *rcU, *rcs, *rcg;
    double* scal;                             // device scalars, see enum below
    double* hscal;                            // the same block in host-mapped pinned memory (written by scal_publish, read by the host)
    unsigned* hseq;                           // its sequence word (host-mapped): the host spins on it instead of copy + stream synchronise
    double* rpart;                            // partial results of the multi-workgroup reductions (IPM_RED_SLOTS per workgroup)
    unsigned* rcnt;                           // their arrival counter (0 between launches)
    int64_t n, M, ns, ncomp;
    double scale_q;
};
struct IpmDir {
    double *dp, *ds, *dg, *dy, *dmuL, *dmuU, *dmus, *dpi;
};
enum { SC_PINF = 0, SC_DINF, SC_MU, SC_YMAX, SC_AP, SC_AD, SC_SM, SC_EMAX, SC_RMAX, SC_RZ, SC_RPMAX, SC_RZ0, SC_STOP, SC_NSERR, SC_SPEC, SC_COUNT };   // SC_SPEC: a solve whose residual check was deferred (k_ipm_res, spec != 0) missed its tolerance;   // SC_NSERR: dual-equation error of a null-space Newton step

// Hand the scalar block to the host without a copy command or a stream synchronisation: the single workgroup that has just written
// P.scal stores the block into host-mapped memory, fences at system scope and sets the sequence word the host is spinning on
// (round trip kernel -> host 6.7 us instead of 14.5 us with hipMemcpyAsync + hipStreamSynchronize; scripts/probe/src/sync_latency.hip).
// pub == 0: nothing to publish (the host does not read after this launch).  Called by every thread of the workgroup.
__device__ __forceinline__ void scal_publish(const IpmPtrs& P, unsigned pub) {
    if (pub == 0) return;
    __syncthreads();                                            // the writers of P.scal are done (same workgroup)
    if (threadIdx.x < SC_COUNT) P.hscal[threadIdx.x] = P.scal[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(P.hseq, pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ double blk_reduce_max(double v, double* sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double out = 0.0;
    if (threadIdx.x == 0) {
        out = sh[0];
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) out = fmax(out, sh[k]);
        sh[0] = out;
    }
    __syncthreads();
    out = sh[0];
    __syncthreads();
    return out;
}
__device__ __forceinline__ double blk_reduce_min(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double out = 0.0;
    if (threadIdx.x == 0) {
        out = sh[0];
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) out = fmin(out, sh[k]);
        sh[0] = out;
    }
    __syncthreads();
    out = sh[0];
    __syncthreads();
    return out;
}

// Reductions over O(M + n) elements that the host (or the next kernel) needs as scalars: several workgroups reduce their share, store
// IPM_RED_SLOTS partial values each (agent-scope stores: visible across the XCDs' L2s) and count themselves in; the last one to arrive
// combines the partials IN WORKGROUP ORDER (sums stay deterministic), writes the scalars and publishes them.  One workgroup: as before.
#define IPM_RED_SLOTS 8
#define IPM_RED_MAXWG 64
__device__ __forceinline__ void red_store(const IpmPtrs& P, int slot, double v) {
    __hip_atomic_store(P.rpart + (int64_t)blockIdx.x * IPM_RED_SLOTS + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double red_load(const IpmPtrs& P, int wg, int slot) {
    return __hip_atomic_load(P.rpart + (int64_t)wg * IPM_RED_SLOTS + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true in every thread of the workgroup that arrives last (all others are done with this launch); called by all threads
__device__ __forceinline__ bool red_last_arrival(const IpmPtrs& P, bool* sh_flag) {
    if (gridDim.x == 1) return true;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned t = __hip_atomic_fetch_add(P.rcnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        *sh_flag = (t == gridDim.x - 1);
        if (*sh_flag) __hip_atomic_store(P.rcnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return *sh_flag;
}
__device__ __forceinline__ double blk_reduce_sum(double v, double* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double out = 0.0;
    if (threadIdx.x == 0) {
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) out += sh[k];
        sh[0] = out;
    }
    __syncthreads();
    out = sh[0];
    __syncthreads();
    return out;
}

__device__ __forceinline__ double slack_sum(const IpmPtrs& P, int64_t i, const double* v) {
    double a = 0.0;
    int k0 = P.rs0[i], k1 = P.rs1[i];
    if (k0 >= 0) a += P.scoef[k0] * v[k0];
    if (k1 >= 0) a += P.scoef[k1] * v[k1];
    return a;
}

// residuals + convergence measures; act = Ah p and aty = Ah' y were produced by the gemv kernels
__global__ __launch_bounds__(1024) void k_ipm_measures(AsmBt abt, IpmPtrs P, unsigned pub) {
    ASM_BARGS(abt, P, pub);
    __shared__ double sh[16];
    __shared__ bool last;
    double pinf = 0.0, dinf = 0.0, mu = 0.0, ymax = 0.0, rpmax = 0.0;
    const int64_t t0 = (int64_t)blockIdx.x * 1024 + threadIdx.x, stride = (int64_t)gridDim.x * 1024;
    // (loads unconditional, the tests select afterwards - see k_ipm_steps; loops and the order of each thread's sum are kept)
    for (int64_t i = t0; i < P.M; i += stride) {
        const int rt = P.rtype[i];
        const double act = P.act[i], ri = P.r[i], gi = P.g[i], pii = P.pi[i], yi = P.y[i];
        bool ineq = rt != 0;
        double sg = (double)rt;
        double a = act + (P.ns ? slack_sum(P, i, P.s) : 0.0);
        double rp = a - (ri + sg * (ineq ? gi : 0.0));
        P.rp[i] = rp;
        pinf = fmax(pinf, fabs(rp) / (1.0 + fabs(ri)));
        rpmax = fmax(rpmax, fabs(rp));
        if (ineq) mu += gi * pii;
        ymax = fmax(ymax, fabs(yi));
    }
    for (int64_t j = t0; j < P.n; j += stride) {
        const double ubj = P.ub[j], lbj = P.lb[j], qj = P.q[j], at = P.aty[j], mL = P.muL[j], mU = P.muU[j], tL = P.tL[j], tU = P.tU[j];
        bool fr = ubj > lbj;
        double rd = fr ? qj - at - mL + mU : 0.0;
        P.rdp[j] = rd;
        dinf = fmax(dinf, fabs(rd));
        if (fr) mu += tL * mL + tU * mU;
    }
    for (int64_t k = t0; k < P.ns; k += stride) {
        double rd = P.w[k] - P.scoef[k] * P.y[P.srow[k]] - P.mus[k];
        P.rds[k] = rd;
        dinf = fmax(dinf, fabs(rd));
        mu += P.ts[k] * P.mus[k];
    }
    pinf = blk_reduce_max(pinf, sh);
    dinf = blk_reduce_max(dinf, sh);
    ymax = blk_reduce_max(ymax, sh);
    rpmax = blk_reduce_max(rpmax, sh);
    mu = blk_reduce_sum(mu, sh);
    if (gridDim.x > 1) {
        if (threadIdx.x == 0) { red_store(P, 0, pinf); red_store(P, 1, dinf); red_store(P, 2, ymax); red_store(P, 3, rpmax); red_store(P, 4, mu); }
        if (!red_last_arrival(P, &last)) return;
        if (threadIdx.x == 0) {
            pinf = dinf = ymax = rpmax = mu = 0.0;
            for (int w = 0; w < (int)gridDim.x; ++w) {
                pinf = fmax(pinf, red_load(P, w, 0)); dinf = fmax(dinf, red_load(P, w, 1)); ymax = fmax(ymax, red_load(P, w, 2));
                rpmax = fmax(rpmax, red_load(P, w, 3)); mu += red_load(P, w, 4);
            }
        }
    }
    if (threadIdx.x == 0) {
        P.scal[SC_PINF] = pinf;
        P.scal[SC_DINF] = dinf / P.scale_q;
        P.scal[SC_MU] = mu / (double)P.ncomp;
        P.scal[SC_YMAX] = ymax;
        P.scal[SC_RPMAX] = rpmax;
    }
    scal_publish(P, pub);
}

__global__ __launch_bounds__(256) void k_ipm_theta(AsmBt abt, IpmPtrs P, double rho_p) {
    ASM_BARGS(abt, P, rho_p);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < P.n) {
        bool fr = P.ub[t] > P.lb[t];
        P.thp_inv[t] = fr ? 1.0 / (P.muL[t] / P.tL[t] + P.muU[t] / P.tU[t] + rho_p) : 0.0;
    }
    if (t < P.ns) P.ths_inv[t] = P.ts[t] / P.mus[t];
    if (t < P.M) {
        double d = P.rtype[t] != 0 ? P.g[t] / P.pi[t] : 0.0;
        int k0 = P.rs0[t], k1 = P.rs1[t];
        if (P.ns) {
            if (k0 >= 0) d += P.ts[k0] / P.mus[k0];
            if (k1 >= 0) d += P.ts[k1] / P.mus[k1];
        }
        P.dS[t] = d;
    }
}

// complementarity right-hand sides and hp, hs, theta*hp.
//   mode 0: affine (predictor);  mode 1: Mehrotra corrector with the affine direction A;
//   mode 2: Gondzio centrality corrector for the direction A at the trial step (tp, td): the products of the trial
//           point are projected onto [lo, hi], the residual terms rp / rdp / rds are dropped (oracle: IPM.run, corr()).
__device__ __forceinline__ double mcc_term(double x, double dx, double z, double dz, double lo, double hi) {
    double v = (x + dx) * (z + dz);
    return fmax(fmin(fmax(v, lo), hi) - v, -hi);
}
__global__ __launch_bounds__(256) void k_ipm_rhs1(AsmBt abt, IpmPtrs P, IpmDir A, int mode, double tp, double td, double bmin, double bmax) {
    ASM_BARGS(abt, P, A, mode, tp, td, bmin, bmax);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    const double sm = mode ? P.scal[SC_SM] : 0.0;
    const double lo = bmin * sm, hi = bmax * sm;
    const double res = mode == 2 ? 0.0 : 1.0;
    if (t < P.n) {
        bool fr = P.ub[t] > P.lb[t];
        double rcL, rcU;
        if (mode == 2) {
            rcL = fr ? mcc_term(P.tL[t], tp * A.dp[t], P.muL[t], td * A.dmuL[t], lo, hi) : 0.0;
            rcU = fr ? mcc_term(P.tU[t], tp * -A.dp[t], P.muU[t], td * A.dmuU[t], lo, hi) : 0.0;
        } else {
            rcL = sm - P.tL[t] * P.muL[t];
            rcU = sm - P.tU[t] * P.muU[t];
            if (mode) {
                rcL -= A.dp[t] * A.dmuL[t];
                rcU += A.dp[t] * A.dmuU[t];
            }
        }
        P.rcL[t] = rcL;
        P.rcU[t] = rcU;
        double hp = fr ? -res * P.rdp[t] + rcL / P.tL[t] - rcU / P.tU[t] : 0.0;
        P.hp[t] = hp;
        P.tmpn[t] = P.thp_inv[t] * hp;
    }
    if (t < P.ns) {
        double rcs;
        if (mode == 2) rcs = mcc_term(P.ts[t], tp * A.ds[t], P.mus[t], td * A.dmus[t], lo, hi);
        else {
            rcs = sm - P.ts[t] * P.mus[t];
            if (mode) rcs -= A.ds[t] * A.dmus[t];
        }
        P.rcs[t] = rcs;
        P.hs[t] = -res * P.rds[t] + rcs / P.ts[t];
    }
    if (t < P.M) {
        double rcg;
        if (mode == 2) rcg = P.rtype[t] != 0 ? mcc_term(P.g[t], tp * A.dg[t], P.pi[t], td * A.dpi[t], lo, hi) : 0.0;
        else {
            rcg = sm - P.g[t] * P.pi[t];
            if (mode) rcg -= A.dg[t] * A.dpi[t];
        }
        P.rcg[t] = rcg;
    }
}

// rhs = -rp - Ah(theta hp) + sg rcg/pi - E(ths hs)
__global__ __launch_bounds__(256) void k_ipm_rhs2(AsmBt abt, IpmPtrs P, double res) {
    ASM_BARGS(abt, P, res);
    int64_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.M) return;
    bool ineq = P.rtype[i] != 0;
    double v = -res * P.rp[i] - P.t1[i] + (ineq ? (double)P.rtype[i] * P.rcg[i] / P.pi[i] : 0.0);
    if (P.ns) {
        int k0 = P.rs0[i], k1 = P.rs1[i];
        if (k0 >= 0) v -= P.scoef[k0] * P.ths_inv[k0] * P.hs[k0];
        if (k1 >= 0) v -= P.scoef[k1] * P.ths_inv[k1] * P.hs[k1];
    }
    P.rhs[i] = v;
}

__global__ __launch_bounds__(256) void k_vec_mul(AsmBt abt, double* __restrict__ x, const double* __restrict__ a, int64_t len) {
    ASM_BARGS(abt, x, a, len);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < len) x[t] *= a[t];
}

// res = rhs - (sres + dS dy) ; scal[EMAX] = max|res| ; scal[RMAX] = max(1, max|rhs|)
// spec != 0: the host does not read this check before it goes on (the factor of S itself is the preconditioner: the residual of the first
// solve is at rounding level); the verdict "emax > max(crel rmax, floor)" is kept in scal[SC_SPEC] instead - set by the first solve of an
// iteration (spec == 1), or-ed by the later ones (spec == 2) - and arrives with the next block the iteration reads anyway.
__global__ __launch_bounds__(1024) void k_ipm_res(AsmBt abt, IpmPtrs P, const double* __restrict__ sres, const double* __restrict__ dy, unsigned pub, int spec, double crel, double floor_) {
    ASM_BARGS(abt, P, sres, dy, pub, spec, crel, floor_);
    __shared__ double sh[16];
    double emax = 0.0, rmax = 1.0;
    for (int64_t i = threadIdx.x; i < P.M; i += 1024) {
        double v = P.rhs[i] - (sres[i] + P.dS[i] * dy[i]);
        P.res[i] = v;
        emax = fmax(emax, fabs(v));
        rmax = fmax(rmax, fabs(P.rhs[i]));
    }
    emax = blk_reduce_max(emax, sh);
    rmax = blk_reduce_max(rmax, sh);
    if (threadIdx.x == 0) {
        P.scal[SC_EMAX] = emax;
        P.scal[SC_RMAX] = rmax;
        if (spec) {
            const double bad = emax > fmax(crel * rmax, floor_) ? 1.0 : 0.0;
            P.scal[SC_SPEC] = spec == 1 ? bad : fmax(P.scal[SC_SPEC], bad);
        }
    }
    scal_publish(P, pub);
}

// ---- preconditioned conjugate gradients on  S dy = rhs,  S = Ah Th^-1 Ah' + dS,  preconditioner = the Cholesky factor.
// p = z ; rz = r'z
__global__ __launch_bounds__(1024) void k_pcg_start(AsmBt abt, IpmPtrs P, const double* __restrict__ z, double* __restrict__ p) {
    ASM_BARGS(abt, P, z, p);
    __shared__ double sh[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < P.M; i += 1024) {
        double zi = z[i];
        p[i] = zi;
        acc += P.res[i] * zi;
    }
    acc = blk_reduce_sum(acc, sh);
    if (threadIdx.x == 0) {
        P.scal[SC_RZ] = acc;
        P.scal[SC_RZ0] = acc;
        P.scal[SC_STOP] = 0.0;
    }
}
// Sp = sres + dS p ; alpha = rz / p'Sp ; x += alpha p ; r -= alpha Sp ; scal[EMAX] = max|r|
__global__ __launch_bounds__(1024) void k_pcg_step1(AsmBt abt, IpmPtrs P, const double* __restrict__ sres, const double* __restrict__ p, double* __restrict__ x, unsigned pub) {
    ASM_BARGS(abt, P, sres, p, x, pub);
    __shared__ double sh[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < P.M; i += 1024) acc += p[i] * (sres[i] + P.dS[i] * p[i]);
    acc = blk_reduce_sum(acc, sh);
    const double rz = P.scal[SC_RZ];
    if (!(acc > 0.0 && rz > 1e-30 * P.scal[SC_RZ0] && rz < 1e12 * acc)) {   // breakdown: the rest of the residual is outside range(S)
        if (threadIdx.x == 0) P.scal[SC_STOP] = 1.0;
        scal_publish(P, pub);                      // (the condition is uniform: every thread takes this branch)
        return;
    }
    const double alpha = rz / acc;
    double emax = 0.0;
    for (int64_t i = threadIdx.x; i < P.M; i += 1024) {
        double pi = p[i];
        x[i] += alpha * pi;
        double r = P.res[i] - alpha * (sres[i] + P.dS[i] * pi);
        P.res[i] = r;
        emax = fmax(emax, fabs(r));
    }
    emax = blk_reduce_max(emax, sh);
    if (threadIdx.x == 0) P.scal[SC_EMAX] = emax;
    scal_publish(P, pub);
}
// beta = r'z / rz_old ; p = z + beta p ; rz = r'z
__global__ __launch_bounds__(1024) void k_pcg_step2(AsmBt abt, IpmPtrs P, const double* __restrict__ z, double* __restrict__ p) {
    ASM_BARGS(abt, P, z, p);
    __shared__ double sh[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < P.M; i += 1024) acc += P.res[i] * z[i];
    acc = blk_reduce_sum(acc, sh);
    const double beta = acc / P.scal[SC_RZ];
    for (int64_t i = threadIdx.x; i < P.M; i += 1024) p[i] = z[i] + beta * p[i];
    __syncthreads();
    if (threadIdx.x == 0) P.scal[SC_RZ] = acc;
}

// Newton direction from dy and aty2 = Ah' dy (in P.aty is NOT touched; tN holds Ah' dy)
__global__ __launch_bounds__(256) void k_ipm_dir(AsmBt abt, IpmPtrs P, IpmDir D, const double* __restrict__ tN) {
    ASM_BARGS(abt, P, D, tN);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < P.n) {
        bool fr = P.ub[t] > P.lb[t];
        double dp = P.thp_inv[t] * (P.hp[t] + tN[t]);
        D.dp[t] = dp;
        D.dmuL[t] = fr ? (P.rcL[t] - P.muL[t] * dp) / P.tL[t] : 0.0;
        D.dmuU[t] = fr ? (P.rcU[t] + P.muU[t] * dp) / P.tU[t] : 0.0;
    }
    if (t < P.ns) {
        double ds = P.ths_inv[t] * (P.hs[t] + P.scoef[t] * D.dy[P.srow[t]]);
        D.ds[t] = ds;
        D.dmus[t] = (P.rcs[t] - P.mus[t] * ds) / P.ts[t];
    }
    if (t < P.M) {
        bool ineq = P.rtype[t] != 0;
        double dpi = ineq ? (double)P.rtype[t] * D.dy[t] : 0.0;
        D.dpi[t] = dpi;
        D.dg[t] = ineq ? (P.rcg[t] - P.g[t] * dpi) / P.pi[t] : 0.0;
    }
}

__device__ __forceinline__ double ratio(double x, double dx) { return dx < 0.0 ? -x / dx : 1e300; }

// step lengths to the boundary (ap primal, ad dual), each capped at 1
__global__ __launch_bounds__(1024) void k_ipm_steps(AsmBt abt, IpmPtrs P, IpmDir D, unsigned pub) {
    ASM_BARGS(abt, P, D, pub);
    __shared__ double sh[16];
    __shared__ bool last;
    double ap = 1e300, ad = 1e300;
    const int64_t t0 = (int64_t)blockIdx.x * 1024 + threadIdx.x, stride = (int64_t)gridDim.x * 1024;
    // ONE loop over the three index ranges with every load unconditional (clamped index, selected afterwards): a load behind a test of another
    // load is a second global round trip, and three loops of three to four iterations were fourteen of them in a row (17 us for 30 k
    // entries).  Minima only: any order gives the same result.
    const int64_t top = max(max(P.n, P.M), (int64_t)P.ns);
    for (int64_t t = t0; t < top; t += stride) {
        const int64_t j = max(min(t, (int64_t)P.n - 1), (int64_t)0), i = max(min(t, (int64_t)P.M - 1), (int64_t)0), k = max(min(t, (int64_t)P.ns - 1), (int64_t)0);
        const double ubj = P.ub[j], lbj = P.lb[j], tL = P.tL[j], tU = P.tU[j], dp = D.dp[j], mL = P.muL[j], dmL = D.dmuL[j], mU = P.muU[j], dmU = D.dmuU[j];
        const int rt = P.rtype[i];
        const double g = P.g[i], dg = D.dg[i], pi = P.pi[i], dpi = D.dpi[i];
        double ts = 1.0, dsk = 0.0, ms = 1.0, dms = 0.0;
        if (P.ns) { ts = P.ts[k]; dsk = D.ds[k]; ms = P.mus[k]; dms = D.dmus[k]; }      // (uniform)
        if ((t < P.n) & (ubj > lbj)) {
            ap = fmin(ap, fmin(ratio(tL, dp), ratio(tU, -dp)));
            ad = fmin(ad, fmin(ratio(mL, dmL), ratio(mU, dmU)));
        }
        if (t < P.ns) {
            ap = fmin(ap, ratio(ts, dsk));
            ad = fmin(ad, ratio(ms, dms));
        }
        if ((t < P.M) & (rt != 0)) {
            ap = fmin(ap, ratio(g, dg));
            ad = fmin(ad, ratio(pi, dpi));
        }
    }
    ap = blk_reduce_min(ap, sh);
    ad = blk_reduce_min(ad, sh);
    if (gridDim.x > 1) {
        if (threadIdx.x == 0) { red_store(P, 0, ap); red_store(P, 1, ad); }
        if (!red_last_arrival(P, &last)) return;
        if (threadIdx.x == 0)
            for (int w = 0; w < (int)gridDim.x; ++w) { ap = fmin(ap, red_load(P, w, 0)); ad = fmin(ad, red_load(P, w, 1)); }
    }
    if (threadIdx.x == 0) {
        P.scal[SC_AP] = fmin(1.0, ap);
        P.scal[SC_AD] = fmin(1.0, ad);
    }
    scal_publish(P, pub);
}

// mu_aff -> sigma = (mu_aff/mu)^3 -> sm = sigma mu
__global__ __launch_bounds__(1024) void k_ipm_muaff(AsmBt abt, IpmPtrs P, IpmDir A, int sexp) {
    ASM_BARGS(abt, P, A, sexp);
    __shared__ double sh[16];
    __shared__ bool last;
    const double ap = P.scal[SC_AP], ad = P.scal[SC_AD];
    double acc = 0.0;
    const int64_t t0 = (int64_t)blockIdx.x * 1024 + threadIdx.x, stride = (int64_t)gridDim.x * 1024;
    // (loads unconditional, the tests select afterwards - see k_ipm_steps; the three loops and the order of each thread's sum are kept)
    for (int64_t j = t0; j < P.n; j += stride) {
        const double ubj = P.ub[j], lbj = P.lb[j], tL = P.tL[j], tU = P.tU[j], dp = A.dp[j], mL = P.muL[j], dmL = A.dmuL[j], mU = P.muU[j], dmU = A.dmuU[j];
        if (!(ubj > lbj)) continue;
        acc += (tL + ap * dp) * (mL + ad * dmL) + (tU - ap * dp) * (mU + ad * dmU);
    }
    for (int64_t k = t0; k < P.ns; k += stride) acc += (P.ts[k] + ap * A.ds[k]) * (P.mus[k] + ad * A.dmus[k]);
    for (int64_t i = t0; i < P.M; i += stride) {
        const int rt = P.rtype[i];
        const double g = P.g[i], dg = A.dg[i], pi = P.pi[i], dpi = A.dpi[i];
        if (rt != 0) acc += (g + ap * dg) * (pi + ad * dpi);
    }
    acc = blk_reduce_sum(acc, sh);
    if (gridDim.x > 1) {
        if (threadIdx.x == 0) red_store(P, 0, acc);
        if (!red_last_arrival(P, &last)) return;
        if (threadIdx.x == 0) {
            acc = 0.0;
            for (int w = 0; w < (int)gridDim.x; ++w) acc += red_load(P, w, 0);
        }
    }
    if (threadIdx.x == 0) {
        double mu = P.scal[SC_MU];
        double mu_aff = acc / (double)P.ncomp;
        double r = mu > 0.0 ? mu_aff / mu : 0.0;
        P.scal[SC_SM] = (sexp == 2 ? r * r : (sexp == 4 ? r * r * r * r : r * r * r)) * mu;      // Mehrotra's centring (r^3; 2 / 4: measurement knob)
    }
}

// D += E (candidate direction of a centrality corrector)
__global__ __launch_bounds__(256) void k_ipm_diradd(AsmBt abt, IpmPtrs P, IpmDir D, IpmDir E) {
    ASM_BARGS(abt, P, D, E);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < P.n) {
        D.dp[t] += E.dp[t];
        D.dmuL[t] += E.dmuL[t];
        D.dmuU[t] += E.dmuU[t];
    }
    if (t < P.ns) {
        D.ds[t] += E.ds[t];
        D.dmus[t] += E.dmus[t];
    }
    if (t < P.M) {
        D.dg[t] += E.dg[t];
        D.dy[t] += E.dy[t];
        D.dpi[t] += E.dpi[t];
    }
}

// iterate += (al primal, be dual) * direction
__global__ __launch_bounds__(256) void k_ipm_update(AsmBt abt, IpmPtrs P, IpmDir C, double al, double be) {
    ASM_BARGS(abt, P, C, al, be);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < P.n) {
        bool fr = P.ub[t] > P.lb[t];
        P.p[t] += al * C.dp[t];
        P.tL[t] = fr ? P.tL[t] + al * C.dp[t] : 1.0;
        P.tU[t] = fr ? P.tU[t] - al * C.dp[t] : 1.0;
        P.muL[t] += be * C.dmuL[t];
        P.muU[t] += be * C.dmuU[t];
    }
    if (t < P.ns) {
        P.s[t] += al * C.ds[t];
        P.ts[t] += al * C.ds[t];
        P.mus[t] += be * C.dmus[t];
    }
    if (t < P.M) {
        bool ineq = P.rtype[t] != 0;
        P.g[t] = ineq ? P.g[t] + al * C.dg[t] : 1.0;
        double pi = P.pi[t] + be * C.dpi[t];
        P.pi[t] = pi;
        P.y[t] = ineq ? (double)P.rtype[t] * pi : P.y[t] + be * C.dy[t];
    }
}

// best-iterate safeguard (oracle: IPM.snapshot / restore): the iterate - p, tL, tU, muL, muU | g, y, pi | s, ts, mus | e (null-space form) -
// saved to (dir = 0) or brought back from (dir = 1) `snap` (6 ldn + 3 Mp + 3 nsp doubles)
__global__ __launch_bounds__(256) void k_ipm_snapshot(AsmBt abt, IpmPtrs P, double* __restrict__ snap, double* __restrict__ e, int64_t ldn, int64_t Mp, int64_t nsp, int dir) {
    ASM_BARGS(abt, P, snap, e, ldn, Mp, nsp, dir);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double* sn = snap;
    double* vn[5] = {P.p, P.tL, P.tU, P.muL, P.muU};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (t < P.n) { if (dir) vn[k][t] = sn[t]; else sn[t] = vn[k][t]; }
        sn += ldn;
    }
    if (e) { if (t < ldn) { if (dir) e[t] = sn[t]; else sn[t] = e[t]; } }
    sn += ldn;
    double* vm[3] = {P.g, P.y, P.pi};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (t < P.M) { if (dir) vm[k][t] = sn[t]; else sn[t] = vm[k][t]; }
        sn += Mp;
    }
    double* vs[3] = {P.s, P.ts, P.mus};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (t < P.ns) { if (dir) vs[k][t] = sn[t]; else sn[t] = vs[k][t]; }
        sn += nsp;
    }
}

// starting point (oracle: IPM.__init__); act = Ah p0 must already be in P.act
// start point (oracle: IPM.__init__): the box's midpoint; normal phase (no slack columns): the origin moved into the middle half of the box
__global__ __launch_bounds__(256) void k_ipm_init_p(AsmBt abt, IpmPtrs P, int origin) {
    ASM_BARGS(abt, P, origin);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < P.n) {
        const double lb = P.lb[t], ub = P.ub[t];
        double v = 0.5 * (lb + ub);
        if (origin) {
            const double w4 = 0.25 * (ub - lb);
            v = fmin(fmax(0.0, lb + w4), ub - w4);
        }
        P.p[t] = v;
    }
    if (t < P.ns) P.s[t] = P.slo[t] + 1.0;
}
__global__ __launch_bounds__(256) void k_ipm_init_rest(AsmBt abt, IpmPtrs P, double mu_factor) {
    ASM_BARGS(abt, P, mu_factor);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    const double mu0 = mu_factor * P.scale_q;
    if (t < P.n) {
        bool fr = P.ub[t] > P.lb[t];
        double tl = fr ? P.p[t] - P.lb[t] : 1.0, tu = fr ? P.ub[t] - P.p[t] : 1.0;
        P.tL[t] = tl;
        P.tU[t] = tu;
        P.muL[t] = fr ? mu0 / tl : 0.0;
        P.muU[t] = fr ? mu0 / tu : 0.0;
    }
    if (t < P.ns) {
        double ts = P.s[t] - P.slo[t];
        P.ts[t] = ts;
        P.mus[t] = mu0 / ts;
    }
    if (t < P.M) {
        bool ineq = P.rtype[t] != 0;
        double sg = (double)P.rtype[t];
        double a = P.act[t] + (P.ns ? slack_sum(P, t, P.s) : 0.0);
        double g = ineq ? fmax(sg * (a - P.r[t]), 1.0) : 1.0;
        P.g[t] = g;
        double pi = ineq ? mu0 / g : 0.0;
        P.pi[t] = pi;
        P.y[t] = sg * pi;
    }
}

// ---- column form of the Newton system (restoration LPs): K = Th + Ah' D^-1 Ah, preconditioner by Sherman-Morrison-Woodbury
// dinv = 1/dS (rows; the padding up to the k-chunk multiple stays 0), th = Theta + rho_p for free columns, `fixed` else
__global__ __launch_bounds__(256) void k_ipm_col_prep(AsmBt abt, IpmPtrs P, double rho_p, double fixed, double* __restrict__ dinv, double* __restrict__ th) {
    ASM_BARGS(abt, P, rho_p, fixed, dinv, th);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < P.M) dinv[t] = 1.0 / P.dS[t];
    if (t < P.n) {
        bool fr = P.ub[t] > P.lb[t];
        th[t] = fr ? P.muL[t] / P.tL[t] + P.muU[t] / P.tU[t] + rho_p : fixed;
    }
}
// u = dinv .* r
__global__ __launch_bounds__(256) void k_col_scale(AsmBt abt, const double* __restrict__ dinv, const double* __restrict__ r, double* __restrict__ u, int64_t M) {
    ASM_BARGS(abt, dinv, r, u, M);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < M) u[t] = dinv[t] * r[t];
}
// out = u - dinv .* w
__global__ __launch_bounds__(256) void k_col_finish(AsmBt abt, const double* __restrict__ dinv, const double* __restrict__ u, const double* __restrict__ w, double* __restrict__ out, int64_t M) {
    ASM_BARGS(abt, dinv, u, w, out, M);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < M) out[t] = u[t] - dinv[t] * w[t];
}

// ---- reduced row form (normal phase of large sparse problems): s_ii = sum_j Ah_ij^2 / Th_j over the CSR copy of the pattern,
// gather / scatter between the full row space and the factored subset E, diagonal preconditioner on the dropped rows I
__global__ __launch_bounds__(256) void k_ipm_sdiag_csr(AsmBt abt, const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ vals, const double* __restrict__ thinv, double* __restrict__ out, int64_t M) {
    ASM_BARGS(abt, ptr, col, vals, thinv, out, M);
    int64_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    double acc = 0.0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) {
        double a = vals[k];
        acc = fma(a * a, thinv[col[k]], acc);
    }
    out[i] = acc;
}
// ce[a] = r[E[a]]
__global__ __launch_bounds__(256) void k_red_gather(AsmBt abt, const int* __restrict__ E, int nE, const double* __restrict__ r, double* __restrict__ ce) {
    ASM_BARGS(abt, E, nE, r, ce);
    int a = blockIdx.x * 256 + threadIdx.x;
    if (a < nE) ce[a] = r[E[a]];
}
// z[E[a]] = ze[a] ; z[I[b]] = r[I[b]] / dI[b]
__global__ __launch_bounds__(256) void k_red_scatter(AsmBt abt, const int* __restrict__ E, int nE, const double* __restrict__ ze, const int* __restrict__ I, int nI, const double* __restrict__ dI, const double* __restrict__ r, double* __restrict__ z) {
    ASM_BARGS(abt, E, nE, ze, I, nI, dI, r, z);
    int a = blockIdx.x * 256 + threadIdx.x;
    if (a < nE) z[E[a]] = ze[a];
    if (a < nI) z[I[a]] = r[I[a]] / dI[a];
}
