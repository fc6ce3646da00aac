// Device-resident active-set machinery (oracle/lp_solver.py: identify, eqp, kkt_measures, correct, face_primal, face_dual).
// The working sets, the index lists derived from them and every O(M+n) vector of the equality-constrained solves live in
// HBM; the host sequences launches and reads back one small block of counters / scalars per solve (the size of the
// gathered Schur system, which fixes the launch grids of the SYRK and the factorisation, and the optimality measures).
// Reductions and compactions run in one 1024-thread workgroup with a fixed order, so every decision is deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// integer counters / double scalars handed back to the host
enum { AC_NH = 0, AC_NF, AC_ANYSOFT, AC_NCHG, AC_NDIFF, AC_NVIOL, AC_NREL, AC_COUNT };
enum { AS_PR = 0, AS_DU, AS_EQRES, AS_HARDRES, AS_COUNT };

struct AsSets {
    int *rowst, *bst, *sst;      // M / n / ns : row active, bound state (-1 lower, +1 upper, 0 free), slack basic
};

struct AsPtrs {
    // scaled LP (the arrays of the interior-point arena)
    const double *q, *lb, *ub, *r, *w, *slo, *scoef;
    const int *rtype, *srow, *rs0, *rs1;
    int64_t n, M, ns;
    double scale_q;
    // derived from the current working set
    int *ksoft;                  // M : first basic slack of the row, -1 if none
    int *Hidx, *hpos;            // hard rows (active, no basic slack) in the order of `rperm` (ascending when null) / inverse map (-1)
    const int* rperm;            // M : the rows in reverse Cuthill-McKee order of their coupling graph (banded Gram matrices), or null
    int *Fidx, *fpos;            // free variables in ascending order / inverse map (-1)
    double *Fmask, *Hmask;       // 1.0 / 0.0 over the (padded) columns / rows: theta operands of the two Gram builds
    double *sl;                  // M : sum_k scoef_k slo_k of the row's slack columns (all slacks at their bound)
    // solution of the last solve
    double *p, *s, *y, *act, *z;
    // work vectors: n-sized ...
    double *pB, *pF, *cF, *rd, *tN, *xfull, *nu;
    // ... and M-sized (compact vectors are indexed by position in Hidx / Fidx; both fit max(M, n))
    double *t, *bH, *v, *u, *yH, *yfull, *uacc, *ax;
    int* cnt;
    double* scal;
};

// ---- block-wide ordered compaction helper: returns the output position of a set flag, advances `base` by the chunk's count
__device__ __forceinline__ int blk_compact_pos(bool flag, int& base, int* sh_cnt) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned long long m = __ballot(flag);
    if (lane == 0) sh_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < nw; ++k) {
        int c = sh_cnt[k];
        if (k < wv) off += c;
        tot += c;
    }
    int pos = base + off + __popcll(m & ((1ull << lane) - 1ull));
    base += tot;
    __syncthreads();
    return pos;
}

// optimal-partition guess from the interior-point iterate (oracle: identify)
__global__ __launch_bounds__(256) void k_as_identify(AsmBt abt, IpmPtrs P, AsSets S) {
    ASM_BARGS(abt, P, S);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    const double sq = P.scale_q;
    if (t < P.n) {
        int b = 0;
        if (!(P.ub[t] > P.lb[t])) b = -1;
        else {
            double width = P.ub[t] - P.lb[t];
            if ((P.tL[t] / width) < (P.muL[t] / sq)) b = -1;
            if ((P.tU[t] / width) < (P.muU[t] / sq)) b = 1;
        }
        S.bst[t] = b;
    }
    if (t < P.ns) S.sst[t] = ((P.ts[t] / (1.0 + fabs(P.slo[t]))) >= (P.mus[t] / sq)) ? 1 : 0;
    if (t < P.M) {
        int a = P.rtype[t] == 0 ? 1 : (((P.g[t] / (1.0 + fabs(P.r[t]))) < (P.pi[t] / sq)) ? 1 : 0);
        if (P.ns) {
            int k0 = P.rs0[t], k1 = P.rs1[t];
            if (k0 >= 0 && (P.ts[k0] / (1.0 + fabs(P.slo[k0]))) >= (P.mus[k0] / sq)) a = 1;
            if (k1 >= 0 && (P.ts[k1] / (1.0 + fabs(P.slo[k1]))) >= (P.mus[k1] / sq)) a = 1;
        }
        S.rowst[t] = a;
    }
}

// out = src (0 when src is null) clipped into [lb, ub]: reference points of the two projections (oracle: zero_p, np.clip(ip.p))
__global__ __launch_bounds__(256) void k_as_clip0(AsmBt abt, const double* __restrict__ lb, const double* __restrict__ ub, const double* __restrict__ src, double* __restrict__ out, int64_t n) {
    ASM_BARGS(abt, lb, ub, src, out, n);
    int64_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) out[j] = fmin(fmax(src ? src[j] : 0.0, lb[j]), ub[j]);
}

// sl[i] = sum of scoef*slo over the row's slack columns (once per LP)
__global__ __launch_bounds__(256) void k_as_sl(AsmBt abt, AsPtrs A) {
    ASM_BARGS(abt, A);
    int64_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.M) return;
    double a = 0.0;
    if (A.ns) {
        int k0 = A.rs0[i], k1 = A.rs1[i];
        if (k0 >= 0) a += A.scoef[k0] * A.slo[k0];
        if (k1 >= 0) a += A.scoef[k1] * A.slo[k1];
    }
    A.sl[i] = a;
}

// last-resort answer of a restoration LP (oracle: solve_scaled, 'ipm-conv'): every slack keeps its own value s (no basic slack is recomputed
// from its row): sl[i] = sum of scoef * s over the row's slack columns, ksoft := -1.  (k_as_sl restores sl at the start of the next LP.)
__global__ __launch_bounds__(256) void k_as_sl_values(AsmBt abt, AsPtrs A) {
    ASM_BARGS(abt, A);
    int64_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.M) return;
    double a = 0.0;
    if (A.ns) {
        int k0 = A.rs0[i], k1 = A.rs1[i];
        if (k0 >= 0) a += A.scoef[k0] * A.s[k0];
        if (k1 >= 0) a += A.scoef[k1] * A.s[k1];
    }
    A.sl[i] = a;
    A.ksoft[i] = -1;
}

// s = max(src, slo)
__global__ __launch_bounds__(256) void k_as_smax(AsmBt abt, const double* __restrict__ src, const double* __restrict__ slo, double* __restrict__ dst, int64_t ns) {
    ASM_BARGS(abt, src, slo, dst, ns);
    int64_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < ns) dst[k] = fmax(src[k], slo[k]);
}

// Everything an equality-constrained solve derives from its working set: soft rows and their known multiplier, the ordered
// lists of hard rows and free variables with their inverse maps and masks, the bound-active part of p.  One workgroup.
__global__ __launch_bounds__(1024) void k_as_setup(AsmBt abt, AsPtrs A, AsSets S, const double* __restrict__ p_ref, int64_t ldn, int64_t ldT) {
    ASM_BARGS(abt, A, S, p_ref, ldn, ldT);
    __shared__ int sh_cnt[16];
    int baseH = 0, baseF = 0, any_soft = 0;
    for (int64_t i0 = 0; i0 < A.M; i0 += 1024) {
        const int64_t qpos = i0 + threadIdx.x;                              // position in the row order of the factorisations
        const int64_t i = qpos < A.M ? (A.rperm ? (int64_t)A.rperm[qpos] : qpos) : qpos;
        bool hard = false;
        if (i < A.M) {
            int ks = -1;
            if (A.ns) {
                int k0 = A.rs0[i], k1 = A.rs1[i];
                if (k0 >= 0 && S.sst[k0] == 1) ks = k0;
                else if (k1 >= 0 && S.sst[k1] == 1) ks = k1;
            }
            A.ksoft[i] = ks;
            A.y[i] = ks >= 0 ? A.w[ks] * A.scoef[ks] : 0.0;
            hard = S.rowst[i] == 1 && ks < 0;
            A.Hmask[i] = hard ? 1.0 : 0.0;
            if (ks >= 0) any_soft = 1;
        }
        int pos = blk_compact_pos(hard, baseH, sh_cnt);
        if (i < A.M) {
            A.hpos[i] = hard ? pos : -1;
            if (hard) A.Hidx[pos] = (int)i;
        }
    }
    for (int64_t i = A.M + threadIdx.x; i < ldT; i += 1024) A.Hmask[i] = 0.0;
    for (int64_t j0 = 0; j0 < A.n; j0 += 1024) {
        int64_t j = j0 + threadIdx.x;
        bool fr = false;
        if (j < A.n) {
            int b = S.bst[j];
            fr = b == 0;
            A.Fmask[j] = fr ? 1.0 : 0.0;
            double ref = p_ref ? p_ref[j] : 0.0;
            double pj = b < 0 ? A.lb[j] : (b > 0 ? A.ub[j] : ref);
            A.p[j] = pj;
            A.pB[j] = fr ? 0.0 : pj;
            A.pF[j] = fr ? ref : 0.0;
        }
        int pos = blk_compact_pos(fr, baseF, sh_cnt);
        if (j < A.n) {
            A.fpos[j] = fr ? pos : -1;
            if (fr) A.Fidx[pos] = (int)j;
        }
    }
    for (int64_t j = A.n + threadIdx.x; j < ldn; j += 1024) { A.Fmask[j] = 0.0; A.pB[j] = 0.0; A.pF[j] = 0.0; A.p[j] = 0.0; }
    for (int64_t k = threadIdx.x; k < A.ns; k += 1024) A.s[k] = A.slo[k];
    sh_cnt[0] = 0;
    __syncthreads();
    if (any_soft) sh_cnt[0] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        A.cnt[AC_NH] = baseH;
        A.cnt[AC_NF] = baseF;
        A.cnt[AC_ANYSOFT] = sh_cnt[0];
    }
}

// right-hand sides of the two projections:  bH = r_H - (Ah pB)_H - sl_H ;  cF = q_F - (Ah' y_soft)_F ;  yH = y_ref_H
// t = Ah pB and (when any_soft) tN = Ah' y_soft were produced by the matrix-vector kernels.
__global__ __launch_bounds__(256) void k_as_rhs(AsmBt abt, AsPtrs A, const double* __restrict__ y_ref) {
    ASM_BARGS(abt, A, y_ref);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    const int nH = A.cnt[AC_NH];
    const bool soft = A.cnt[AC_ANYSOFT] != 0;
    if (t < nH) {
        int i = A.Hidx[t];
        A.bH[t] = A.r[i] - A.t[i] - A.sl[i];
        A.yH[t] = y_ref ? y_ref[i] : 0.0;
        A.uacc[t] = 0.0;
    }
    if (t < A.n) A.cF[t] = A.Fmask[t] != 0.0 ? A.q[t] - (soft ? A.tN[t] : 0.0) : 0.0;
}
// v[a] = bH[a] - t[H[a]]
__global__ __launch_bounds__(256) void k_as_res_p(AsmBt abt, AsPtrs A) {
    ASM_BARGS(abt, A);
    int64_t a = blockIdx.x * 256 + threadIdx.x;
    if (a < A.cnt[AC_NH]) A.v[a] = A.bH[a] - A.t[A.Hidx[a]];
}
// yfull = scatter of a compact H-vector (zero elsewhere); optionally accumulate it into uacc
__global__ __launch_bounds__(256) void k_as_scatter_h(AsmBt abt, AsPtrs A, const double* __restrict__ src, int accumulate) {
    ASM_BARGS(abt, A, src, accumulate);
    int64_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.M) return;
    int pos = A.hpos[i];
    A.yfull[i] = pos >= 0 ? src[pos] : 0.0;
    if (accumulate && pos >= 0) A.uacc[pos] += src[pos];
}
// pF += Fmask .* tN
__global__ __launch_bounds__(256) void k_as_add_f(AsmBt abt, AsPtrs A) {
    ASM_BARGS(abt, A);
    int64_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < A.n) A.pF[j] += A.Fmask[j] * A.tN[j];
}
// rd = Fmask .* (cF - tN)
__global__ __launch_bounds__(256) void k_as_rd(AsmBt abt, AsPtrs A) {
    ASM_BARGS(abt, A);
    int64_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < A.n) A.rd[j] = A.Fmask[j] * (A.cF[j] - A.tN[j]);
}
// v[a] = t[H[a]]
__global__ __launch_bounds__(256) void k_as_gather_h(AsmBt abt, AsPtrs A) {
    ASM_BARGS(abt, A);
    int64_t a = blockIdx.x * 256 + threadIdx.x;
    if (a < A.cnt[AC_NH]) A.v[a] = A.t[A.Hidx[a]];
}
// yH += u
__global__ __launch_bounds__(256) void k_as_add_yh(AsmBt abt, AsPtrs A) {
    ASM_BARGS(abt, A);
    int64_t a = blockIdx.x * 256 + threadIdx.x;
    if (a < A.cnt[AC_NH]) A.yH[a] += A.u[a];
}
// p_F <- pF ; y_H <- yH (soft rows keep their known multiplier, inactive rows 0)
__global__ __launch_bounds__(256) void k_as_merge(AsmBt abt, AsPtrs A, int with_y) {
    ASM_BARGS(abt, A, with_y);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < A.n && A.Fmask[t] != 0.0) A.p[t] = A.pF[t];
    if (with_y && t < A.M) {
        int pos = A.hpos[t];
        if (pos >= 0) A.y[t] = A.yH[pos];
    }
}

// Final stage of an equality-constrained solve (oracle: eqp tail, kkt_measures, correct).  Inputs: t = Ah p, tN = Ah' y.
// Computes the basic slack values, act = Ah p + E s, z = q - Ah' y, the LP optimality measures of (p, s, y) on the working
// set `cur`, the corrected working set `nx` with the number of changes, and the number of positions where `nx` differs
// from `prev` (cycle detection).  One workgroup.
__global__ __launch_bounds__(1024) void k_as_finish(AsmBt abt, AsPtrs A, AsSets cur, AsSets nx, AsSets prev, int have_prev, double tol_p, double tol_d) {
    ASM_BARGS(abt, A, cur, nx, prev, have_prev, tol_p, tol_d);
    __shared__ double sh[16];
    double pr = 0.0, du = 0.0, nchg = 0.0, ndiff = 0.0;
    const double td = tol_d * A.scale_q;
    for (int64_t i = threadIdx.x; i < A.M; i += 1024) {
        double a = A.t[i] + A.sl[i];
        int ks = A.ksoft[i];
        if (ks >= 0) {
            double snew = A.slo[ks] + (A.r[i] - a) / A.scoef[ks];
            a += A.scoef[ks] * (snew - A.slo[ks]);
            A.s[ks] = snew;
        }
        A.act[i] = a;
        const int rt = A.rtype[i];
        const double den = 1.0 + fabs(A.r[i]);
        double viol = rt == 0 ? fabs(a - A.r[i]) : fmax(0.0, rt * (A.r[i] - a));
        pr = fmax(pr, viol / den);
        const double y = A.y[i];
        const int st = cur.rowst[i];
        double dr = st == 0 ? fabs(y) : (rt == 1 ? fmax(-y, 0.0) : (rt == -1 ? fmax(y, 0.0) : 0.0));
        du = fmax(du, dr);
        int ns_ = st;
        if (rt != 0) {
            double v2 = rt * (A.r[i] - a) / den;
            if (st == 1 && rt * y < -td) { ns_ = 0; nchg += 1.0; }
            else if (st == 0 && v2 > tol_p) { ns_ = 1; nchg += 1.0; }
        }
        nx.rowst[i] = ns_;
    }
    for (int64_t j = threadIdx.x; j < A.n; j += 1024) {
        const double z = A.q[j] - A.tN[j];
        A.z[j] = z;
        const double pj = A.p[j], lb = A.lb[j], ub = A.ub[j];
        pr = fmax(pr, fmax(lb - pj, 0.0));
        pr = fmax(pr, fmax(pj - ub, 0.0));
        const int b = cur.bst[j];
        const bool fixed = ub <= lb;
        if (!fixed) {
            double dz = b < 0 ? fmax(-z, 0.0) : (b > 0 ? fmax(z, 0.0) : fabs(z));
            du = fmax(du, dz);
        }
        int nb = b;
        if (!fixed && ((b < 0 && z < -td) || (b > 0 && z > td))) { nb = 0; nchg += 1.0; }
        else if (b == 0 && pj < lb - tol_p) { nb = -1; nchg += 1.0; }
        else if (b == 0 && pj > ub + tol_p) { nb = 1; nchg += 1.0; }
        nx.bst[j] = nb;
    }
    __syncthreads();                                   // s[] of the basic slacks is complete
    for (int64_t k = threadIdx.x; k < A.ns; k += 1024) {
        const double slo = A.slo[k], sk = A.s[k];
        pr = fmax(pr, fmax(slo - sk, 0.0) / (1.0 + fabs(slo)));
        const double zs = A.w[k] - A.scoef[k] * A.y[A.srow[k]];
        const int st = cur.sst[k];
        du = fmax(du, st == 0 ? fmax(-zs, 0.0) : fabs(zs));
        int ns_ = st;
        if (st == 0 && zs < -td) { ns_ = 1; nchg += 1.0; }
        else if (st == 1 && sk < slo - tol_p * (1.0 + fabs(slo))) { ns_ = 0; nchg += 1.0; }
        nx.sst[k] = ns_;
    }
    __syncthreads();
    for (int64_t k = threadIdx.x; k < A.ns; k += 1024)
        if (nx.sst[k] == 1) nx.rowst[A.srow[k]] = 1;
    __syncthreads();
    if (have_prev) {
        for (int64_t i = threadIdx.x; i < A.M; i += 1024) ndiff += nx.rowst[i] != prev.rowst[i];
        for (int64_t j = threadIdx.x; j < A.n; j += 1024) ndiff += nx.bst[j] != prev.bst[j];
        for (int64_t k = threadIdx.x; k < A.ns; k += 1024) ndiff += nx.sst[k] != prev.sst[k];
    }
    pr = blk_reduce_max(pr, sh);
    du = blk_reduce_max(du, sh);
    nchg = blk_reduce_sum(nchg, sh);
    ndiff = blk_reduce_sum(ndiff, sh);
    if (threadIdx.x == 0) {
        A.scal[AS_PR] = pr;
        A.scal[AS_DU] = du / A.scale_q;
        A.cnt[AC_NCHG] = (int)nchg;
        A.cnt[AC_NDIFF] = have_prev ? (int)ndiff : -1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Canonical pair of a non-unique optimum (oracle: face_primal / face_dual).
// Primal round tail.  Inputs: t = Ah p, tN = Ah' u_full (u = multipliers of the least-norm problem on the hard rows).
// Violated inequalities of the face join the working set W (in place); when none is violated, non-mandatory members of W
// with a wrong-sign multiplier leave; when nothing changes the hard rows must hold (AS_HARDRES).
__global__ __launch_bounds__(1024) void k_face_primal_finish(AsmBt abt, AsPtrs A, AsSets W, AsSets part, double tol_p, double tol_m, int check_only) {
    ASM_BARGS(abt, A, W, part, tol_p, tol_m, check_only);
    __shared__ double sh[16];
    __shared__ int s_viol;
    double nviol = 0.0, nrel = 0.0, hres = 0.0;
    // ---- pass 1: slack values, activities, violations
    for (int64_t i = threadIdx.x; i < A.M; i += 1024) {
        double a = A.t[i] + A.sl[i];
        int ks = A.ksoft[i];
        if (ks >= 0) {
            double snew = A.slo[ks] + (A.r[i] - a) / A.scoef[ks];
            a += A.scoef[ks] * (snew - A.slo[ks]);
            A.s[ks] = snew;
        }
        A.act[i] = a;
        const int rt = A.rtype[i];
        const double den = 1.0 + fabs(A.r[i]);
        if (rt != 0 && W.rowst[i] == 0 && rt * (A.r[i] - a) / den > tol_p) nviol += 1.0;
        if (A.hpos[i] >= 0) hres = fmax(hres, fabs(a - A.r[i]) / den);
    }
    for (int64_t j = threadIdx.x; j < A.n; j += 1024) {
        const double pj = A.p[j];
        A.nu[j] = pj - A.tN[j];
        if (W.bst[j] == 0 && (pj < A.lb[j] - tol_p || pj > A.ub[j] + tol_p)) nviol += 1.0;
    }
    __syncthreads();
    for (int64_t k = threadIdx.x; k < A.ns; k += 1024)
        if (W.sst[k] == 1 && A.s[k] < A.slo[k] - tol_p * (1.0 + fabs(A.slo[k]))) nviol += 1.0;
    nviol = blk_reduce_sum(nviol, sh);
    hres = blk_reduce_max(hres, sh);
    if (threadIdx.x == 0) s_viol = nviol > 0.0;
    __syncthreads();
    if (check_only) {
        if (threadIdx.x == 0) {
            A.cnt[AC_NVIOL] = (int)nviol;
            A.cnt[AC_NREL] = 0;
            A.scal[AS_HARDRES] = hres;
        }
        return;
    }
    if (s_viol) {
        // ---- grow
        for (int64_t i = threadIdx.x; i < A.M; i += 1024) {
            const int rt = A.rtype[i];
            if (rt != 0 && W.rowst[i] == 0 && rt * (A.r[i] - A.act[i]) / (1.0 + fabs(A.r[i])) > tol_p) W.rowst[i] = 1;
        }
        for (int64_t j = threadIdx.x; j < A.n; j += 1024) {
            if (W.bst[j] != 0) continue;
            const double pj = A.p[j];
            if (pj < A.lb[j] - tol_p) W.bst[j] = -1;
            else if (pj > A.ub[j] + tol_p) W.bst[j] = 1;
        }
        for (int64_t k = threadIdx.x; k < A.ns; k += 1024)
            if (W.sst[k] == 1 && A.s[k] < A.slo[k] - tol_p * (1.0 + fabs(A.slo[k]))) W.sst[k] = 0;
    } else {
        // ---- release non-mandatory members with a wrong-sign multiplier of the least-norm problem
        for (int64_t i = threadIdx.x; i < A.M; i += 1024) {
            const int rt = A.rtype[i];
            const int pos = A.hpos[i];
            if (pos >= 0 && part.rowst[i] != 1 && rt != 0 && rt * A.uacc[pos] < -tol_m) { W.rowst[i] = 0; nrel += 1.0; }
        }
        for (int64_t k = threadIdx.x; k < A.ns; k += 1024) {
            const int i = A.srow[k];
            const int pos = A.hpos[i];
            if (W.sst[k] == 0 && part.sst[k] != 0 && pos >= 0 && A.scoef[k] * A.uacc[pos] > tol_m) { W.sst[k] = 1; nrel += 1.0; }
        }
        for (int64_t j = threadIdx.x; j < A.n; j += 1024) {
            const int b = W.bst[j];
            if (b == 0 || part.bst[j] != 0 || A.ub[j] <= A.lb[j]) continue;
            const double nu = A.nu[j];
            if ((b < 0 && nu < -tol_m) || (b > 0 && nu > tol_m)) { W.bst[j] = 0; nrel += 1.0; }
        }
        __syncthreads();
        for (int64_t k = threadIdx.x; k < A.ns; k += 1024)
            if (W.sst[k] == 1) W.rowst[A.srow[k]] = 1;
    }
    nrel = blk_reduce_sum(nrel, sh);
    if (threadIdx.x == 0) {
        A.cnt[AC_NVIOL] = (int)nviol;
        A.cnt[AC_NREL] = (int)nrel;
        A.scal[AS_HARDRES] = hres;
    }
}

// ---- anchored method in the null space of the mandatory set (oracle: _face_primal_anchored) -------------------------------
// p = p0 + sum_c u_c z_c   (Zbuf row c = z_c, an n-vector supported on the partition's free columns)
__global__ __launch_bounds__(256) void k_face_ns_combine(AsmBt abt, const double* __restrict__ p0, const double* __restrict__ Zbuf, int64_t ldz, const double* __restrict__ u, int k, double* __restrict__ p, int64_t n) {
    ASM_BARGS(abt, p0, Zbuf, ldz, u, k, p, n);
    int64_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double acc = p0[j];
    for (int c = 0; c < k; ++c) acc = fma(u[c], Zbuf[(int64_t)c * ldz + j], acc);
    p[j] = acc;
}
// One step at the candidate p (A.p, with t = Ah p): basic slacks and activities on the working set W, feasibility; if some
// inequality of the face is violated, the anchor (pa, sa, acta) moves towards the candidate until the first one blocks and
// that one is marked in W and reported (AC_NCHG = family: 0 row, 1 slack bound, 2 lower, 3 upper; AC_NDIFF = its index;
// ties: family order, then lowest index).  AC_NVIOL = number of violated inequalities (0: the candidate is feasible).
__device__ __forceinline__ double as_ratio(double g0, double g1) {
    const double a = fmax(g0, 0.0);
    return a / (a - g1);
}
__global__ __launch_bounds__(1024) void k_face_ns_step(AsmBt abt, AsPtrs A, AsSets W, double* __restrict__ pa, double* __restrict__ sa, double* __restrict__ acta, double tol_p) {
    ASM_BARGS(abt, A, W, pa, sa, acta, tol_p);
    __shared__ double sh[16];
    const int64_t M = A.M, n = A.n, ns = A.ns;
    double nviol = 0.0, alpha = 2.0, hres = 0.0;
    for (int64_t i = threadIdx.x; i < M; i += 1024) {
        int ks = -1;
        if (ns) {
            const int k0 = A.rs0[i], k1 = A.rs1[i];
            if (k0 >= 0 && W.sst[k0] == 1) ks = k0;
            else if (k1 >= 0 && W.sst[k1] == 1) ks = k1;
        }
        A.ksoft[i] = ks;
        double a = A.t[i] + A.sl[i];
        if (ks >= 0) {
            const double snew = A.slo[ks] + (A.r[i] - a) / A.scoef[ks];
            a += A.scoef[ks] * (snew - A.slo[ks]);
            A.s[ks] = snew;
        }
        A.act[i] = a;
        const int rt = A.rtype[i];
        const double den = 1.0 + fabs(A.r[i]);
        if (W.rowst[i] == 1 && ks < 0) hres = fmax(hres, fabs(a - A.r[i]) / den);
        if (rt != 0 && W.rowst[i] == 0) {
            const double g1 = rt * (a - A.r[i]) / den;
            if (g1 < -tol_p) { nviol += 1.0; alpha = fmin(alpha, as_ratio(rt * (acta[i] - A.r[i]) / den, g1)); }
        }
    }
    for (int64_t j = threadIdx.x; j < n; j += 1024) {
        if (W.bst[j] != 0) continue;
        double g1 = A.p[j] - A.lb[j];
        if (g1 < -tol_p) { nviol += 1.0; alpha = fmin(alpha, as_ratio(pa[j] - A.lb[j], g1)); }
        g1 = A.ub[j] - A.p[j];
        if (g1 < -tol_p) { nviol += 1.0; alpha = fmin(alpha, as_ratio(A.ub[j] - pa[j], g1)); }
    }
    __syncthreads();                                  // s[] of the basic slacks is complete
    for (int64_t k = threadIdx.x; k < ns; k += 1024) {
        if (W.sst[k] != 1) { A.s[k] = A.slo[k]; continue; }
        const double den = 1.0 + fabs(A.slo[k]);
        const double g1 = (A.s[k] - A.slo[k]) / den;
        if (g1 < -tol_p) { nviol += 1.0; alpha = fmin(alpha, as_ratio((sa[k] - A.slo[k]) / den, g1)); }
    }
    nviol = blk_reduce_sum(nviol, sh);
    alpha = blk_reduce_min(alpha, sh);
    hres = blk_reduce_max(hres, sh);
    if (nviol == 0.0) {
        if (threadIdx.x == 0) { A.cnt[AC_NVIOL] = 0; A.scal[AS_HARDRES] = hres; }
        return;
    }
    // the blocking inequality: smallest ratio, then family order (rows, slacks, lower, upper), then lowest index
    double code = 1e300;                              // family * 2^40 + index, exact in a double
    const double F40 = 1099511627776.0;
    for (int64_t i = threadIdx.x; i < M; i += 1024) {
        const int rt = A.rtype[i];
        if (rt == 0 || W.rowst[i] != 0) continue;
        const double den = 1.0 + fabs(A.r[i]);
        const double g1 = rt * (A.act[i] - A.r[i]) / den;
        if (g1 < -tol_p && as_ratio(rt * (acta[i] - A.r[i]) / den, g1) == alpha) code = fmin(code, (double)i);
    }
    for (int64_t k = threadIdx.x; k < ns; k += 1024) {
        if (W.sst[k] != 1) continue;
        const double den = 1.0 + fabs(A.slo[k]);
        const double g1 = (A.s[k] - A.slo[k]) / den;
        if (g1 < -tol_p && as_ratio((sa[k] - A.slo[k]) / den, g1) == alpha) code = fmin(code, F40 + (double)k);
    }
    for (int64_t j = threadIdx.x; j < n; j += 1024) {
        if (W.bst[j] != 0) continue;
        double g1 = A.p[j] - A.lb[j];
        if (g1 < -tol_p && as_ratio(pa[j] - A.lb[j], g1) == alpha) code = fmin(code, 2.0 * F40 + (double)j);
        g1 = A.ub[j] - A.p[j];
        if (g1 < -tol_p && as_ratio(A.ub[j] - pa[j], g1) == alpha) code = fmin(code, 3.0 * F40 + (double)j);
    }
    code = blk_reduce_min(code, sh);
    const int fam = (int)(code / F40);
    const int64_t e = (int64_t)(code - (double)fam * F40);
    const double al = fmin(alpha, 1.0);
    for (int64_t i = threadIdx.x; i < M; i += 1024) acta[i] += al * (A.act[i] - acta[i]);
    for (int64_t j = threadIdx.x; j < n; j += 1024) pa[j] += al * (A.p[j] - pa[j]);
    for (int64_t k = threadIdx.x; k < ns; k += 1024) sa[k] += al * (A.s[k] - sa[k]);
    if (threadIdx.x == 0) {
        if (fam == 0) W.rowst[e] = 1;
        else if (fam == 1) W.sst[e] = 0;
        else if (fam == 2) W.bst[e] = -1;
        else W.bst[e] = 1;
        A.cnt[AC_NVIOL] = (int)nviol;
        A.cnt[AC_NCHG] = fam;
        A.cnt[AC_NDIFF] = (int)e;
        A.scal[AS_HARDRES] = hres;
    }
}
// The blocking constraint as  c'p (>= | <=) b  on the free columns of the partition: rd = c, AS_EQRES = g = b - c'(p0 - pfix),
// AS_PR = c'c.   t0 = Ah p0.
__global__ __launch_bounds__(1024) void k_face_ns_col(AsmBt abt, AsPtrs A, const double* __restrict__ Ah, int64_t ld, int fam, int64_t e, const double* __restrict__ p0, const double* __restrict__ t0) {
    ASM_BARGS(abt, A, Ah, ld, fam, e, p0, t0);
    __shared__ double sh[16];
    double cc = 0.0;
    const int64_t row = fam == 0 ? e : (fam == 1 ? (int64_t)A.srow[e] : -1);
    for (int64_t j = threadIdx.x; j < A.n; j += 1024) {
        double c = 0.0;
        if (row >= 0) c = A.Fmask[j] * Ah[row * ld + j];
        else if (j == e) c = 1.0;
        A.rd[j] = c;
        cc += c * c;
    }
    cc = blk_reduce_sum(cc, sh);
    if (threadIdx.x == 0) {
        double g;
        if (row >= 0) g = A.r[row] - A.sl[row] - t0[row];
        else g = (fam == 2 ? A.lb[e] : A.ub[e]) - p0[e];
        A.scal[AS_EQRES] = g;
        A.scal[AS_PR] = cc;
    }
}
// z = rd - Fmask .* tN   (tN = N0' S0^-1 N0 c)
__global__ __launch_bounds__(256) void k_face_ns_z(AsmBt abt, AsPtrs A, double* __restrict__ z) {
    ASM_BARGS(abt, A, z);
    int64_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < A.n) z[j] = A.rd[j] - A.Fmask[j] * A.tN[j];
}
// undo a mark of the working set (release of an added constraint)
__global__ void k_face_ns_unmark(AsmBt abt, AsPtrs A, AsSets W, int fam, int64_t e) {
    ASM_BARGS(abt, A, W, fam, e);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (fam == 0) W.rowst[e] = 0;
    else if (fam == 1) { W.sst[e] = 1; W.rowst[A.srow[e]] = 1; }
    else W.bst[e] = 0;
}

// Dual round tail (oracle: face_dual).  Input: tN = Ah' y.  z = q - Ah'y; sign conditions of the LP dual that y violates
// become active: the dual working set D shrinks in place (row leaves H / variable joins F / slack becomes basic).
__global__ __launch_bounds__(1024) void k_face_dual_finish(AsmBt abt, AsPtrs A, AsSets D, double tol_m) {
    ASM_BARGS(abt, A, D, tol_m);
    __shared__ double sh[16];
    const double td = tol_m * A.scale_q;
    double nviol = 0.0;
    for (int64_t j = threadIdx.x; j < A.n; j += 1024) {
        const double z = A.q[j] - A.tN[j];
        A.z[j] = z;
        const int b = D.bst[j];
        if (b != 0 && A.ub[j] > A.lb[j] && ((b < 0 && z < -td) || (b > 0 && z > td))) { D.bst[j] = 0; nviol += 1.0; }
    }
    for (int64_t i = threadIdx.x; i < A.M; i += 1024) {
        const int rt = A.rtype[i];
        if (A.hpos[i] >= 0 && rt != 0 && rt * A.y[i] < -td) { D.rowst[i] = 0; nviol += 1.0; }
    }
    __syncthreads();
    for (int64_t k = threadIdx.x; k < A.ns; k += 1024) {
        const double zs = A.w[k] - A.scoef[k] * A.y[A.srow[k]];
        if (D.sst[k] == 0 && zs < -td) { D.sst[k] = 1; nviol += 1.0; }
    }
    __syncthreads();
    for (int64_t k = threadIdx.x; k < A.ns; k += 1024)
        if (D.sst[k] == 1) D.rowst[A.srow[k]] = 1;
    nviol = blk_reduce_sum(nviol, sh);
    if (threadIdx.x == 0) A.cnt[AC_NVIOL] = (int)nviol;
}

// LP optimality measures of a (p, s, y) triple with separate primal / dual working sets (oracle: face_polish tail).
// Inputs: act (with slacks), z, y, s are current.  pr uses only feasibility; du is measured on the dual working set.
__global__ __launch_bounds__(1024) void k_face_kkt(AsmBt abt, AsPtrs A, AsSets D) {
    ASM_BARGS(abt, A, D);
    __shared__ double sh[16];
    double pr = 0.0, du = 0.0;
    for (int64_t i = threadIdx.x; i < A.M; i += 1024) {
        const int rt = A.rtype[i];
        const double a = A.act[i];
        double viol = rt == 0 ? fabs(a - A.r[i]) : fmax(0.0, rt * (A.r[i] - a));
        pr = fmax(pr, viol / (1.0 + fabs(A.r[i])));
        const double y = A.y[i];
        double dr = D.rowst[i] == 0 ? fabs(y) : (rt == 1 ? fmax(-y, 0.0) : (rt == -1 ? fmax(y, 0.0) : 0.0));
        du = fmax(du, dr);
    }
    for (int64_t j = threadIdx.x; j < A.n; j += 1024) {
        const double pj = A.p[j], z = A.z[j];
        pr = fmax(pr, fmax(A.lb[j] - pj, 0.0));
        pr = fmax(pr, fmax(pj - A.ub[j], 0.0));
        if (A.ub[j] <= A.lb[j]) continue;
        const int b = D.bst[j];
        du = fmax(du, b < 0 ? fmax(-z, 0.0) : (b > 0 ? fmax(z, 0.0) : fabs(z)));
    }
    for (int64_t k = threadIdx.x; k < A.ns; k += 1024) {
        pr = fmax(pr, fmax(A.slo[k] - A.s[k], 0.0) / (1.0 + fabs(A.slo[k])));
        const double zs = A.w[k] - A.scoef[k] * A.y[A.srow[k]];
        du = fmax(du, D.sst[k] == 0 ? fmax(-zs, 0.0) : fabs(zs));
    }
    pr = blk_reduce_max(pr, sh);
    du = blk_reduce_max(du, sh);
    if (threadIdx.x == 0) {
        A.scal[AS_PR] = pr;
        A.scal[AS_DU] = du / A.scale_q;
    }
}

// the answer of an LP packed for ONE device-to-host copy (Solver::as_download): p | z | y | act | s as doubles, then the working set's
// three state vectors as 32-bit integers behind them
__global__ __launch_bounds__(256) void k_as_pack(AsmBt abt, const double* __restrict__ p, const double* __restrict__ z, const double* __restrict__ y, const double* __restrict__ act,
                                                 const double* __restrict__ s, AsSets S, int64_t n, int64_t M, int64_t ns, double* __restrict__ dst) {
    ASM_BARGS(abt, p, z, y, act, s, S, n, M, ns, dst);
    const int64_t t = blockIdx.x * 256 + threadIdx.x;
    int* di = reinterpret_cast<int*>(dst + 2 * n + 2 * M + ns);
    if (t < n) { dst[t] = p[t]; dst[n + t] = z[t]; di[M + t] = S.bst[t]; }
    if (t < M) { dst[2 * n + t] = y[t]; dst[2 * n + M + t] = act[t]; di[t] = S.rowst[t]; }
    if (t < ns) { dst[2 * n + 2 * M + t] = s[t]; di[M + n + t] = S.sst[t]; }
}

// copy of a working set
__global__ __launch_bounds__(256) void k_as_copy_sets(AsmBt abt, AsSets dst, AsSets src, int64_t M, int64_t n, int64_t ns) {
    ASM_BARGS(abt, dst, src, M, n, ns);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < M) dst.rowst[t] = src.rowst[t];
    if (t < n) dst.bst[t] = src.bst[t];
    if (t < ns) dst.sst[t] = src.sst[t];
}
