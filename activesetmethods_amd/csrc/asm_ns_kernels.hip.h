// Null-space (equality-elimination) form of the interior-point Newton system - oracle/lp_solver.py: class NullSpace and the
// `use_ns` branch of IPM.run.  Normal-phase LPs of the ACOPF configurations carry ~n hard equality rows E (case1354pegase-sized:
// 10.7k of them for n = 11.2k variables): Newton steps live in  p = pbar + Z u  with Z an orthonormal basis of null(A_EF)
// (dimension k ~ 500), so the matrix factored in every interior-point iteration is k x k instead of M x M.  Per LP:
//   S0 = A_EF A_EF'  (existing Schur-build + Cholesky kernels), W = S0^-1 A_EF[:, J] for the k retained basis columns J
//   (multi-right-hand-side triangular solves = k_gemm_nt on the matrix cores against the explicit inverses of the wide
//   diagonal blocks), P[J, :] = E_J' - W' A_EF (k_ns_pj), L_J L_J' = P[J, J], Zt = L_J^-1 P[J, :] (k_ns_ortho), GI' = (A_I Z)'.
// Vector layouts: Zt and GI' share one buffer Gt (k rows, pitch ldg = ldn + nIp) so that N = Gt diag(theta~) Gt' is one
// rank-K launch of the existing k_syrk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------------------------------------------
// C[a,b] = (C0 ? C0[a,b] : 0) -/+ sum_k A[a,k] B[b,k]      a < Ma, b < Mb, K a multiple of 32 (operands zero-padded along k)
//   mode 0:  C  = A B'          mode 1:  C = C0 - A B'   (C0 may alias C: every tile is read and written by one workgroup)
// 64 x 64 tile per 256-thread workgroup (four wavefronts as 2 x 2, each 32 x 32 = 2 x 2 MFMA tiles of 16 x 16), 32-deep k-chunks
// double-buffered in LDS through registers - the generic k_syrk<2, 4, 32, 1> with two operand matrices and a full rectangle.
// Used by the multi-right-hand-side triangular solves (rows of A = right-hand sides, rows of B = rows of the factor / of the
// explicit inverse of a wide diagonal block).
// TA = rows of A per workgroup: 64, or 32 when 64 x 64 tiles would leave CUs idle (519 right-hand sides against one wide block: 9 x 16 = 144
// workgroups on 256 CUs; with 32-row tiles 17 x 16 = 272) - every entry of C is the same sum in the same order either way.
// TB = columns of C (rows of B) per workgroup: 64, or 96 when that brings the grid down to one workgroup per CU (519 right-hand sides against
// one wide block: 17 x 16 = 272 workgroups - sixteen CUs get a second one and the launch takes two workgroup times; 17 x 11 = 187 take 1.5).
template <int TA, int TB = 64>
__device__ __forceinline__ void gemm_nt_body(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb, const double* C0, int64_t ldc0, double* C, int64_t ldc, int Ma, int Mb, int K, int mode) {
    constexpr int TS = TB, KC = 32, PITCH = KC + 2, NI = TA / 32, HA = TA / 2, NJ = TB / 32, HB = TB / 2;
    __shared__ __attribute__((aligned(16))) double As[2][TA * PITCH];
    __shared__ __attribute__((aligned(16))) double Bs[2][TS * PITCH];
    const int bi = blockIdx.y, bj = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    v4f64 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
            if (mode != 0 && C0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = bi * TA + wr * HA + i * 16 + (lane >> 4) + 4 * r;
                    const int col = bj * TS + wc * HB + j * 16 + (lane & 15);
                    const double v = C0[(int64_t)min(row, Ma - 1) * ldc0 + min(col, Mb - 1)];
                    acc[i][j][r] = (row < Ma && col < Mb) ? v : 0.0;
                }
            }
        }
    const double asign = mode != 0 ? -1.0 : 1.0;
    // staging: TA / 64 rows x 32 doubles per operand; thread moves 2 doubles per pass, 16 threads per row, 16 rows per pass
    const int lr = tid >> 4, lk = (tid & 15) * 2;
    const double* arow[TA / 16];
    const double* brow[TB / 16];
#pragma unroll
    for (int ps = 0; ps < TA / 16; ++ps) {
        const int ga = bi * TA + ps * 16 + lr;
        arow[ps] = ga < Ma ? A + (int64_t)ga * lda : nullptr;
    }
#pragma unroll
    for (int ps = 0; ps < TB / 16; ++ps) {
        const int gb = bj * TS + ps * 16 + lr;
        brow[ps] = gb < Mb ? B + (int64_t)gb * ldb : nullptr;
    }
    double2 ra[TA / 16], rb[TB / 16];
    auto gload = [&](int k0) {
#pragma unroll
        for (int ps = 0; ps < TA / 16; ++ps) ra[ps] = arow[ps] ? *reinterpret_cast<const double2*>(arow[ps] + k0 + lk) : make_double2(0.0, 0.0);
#pragma unroll
        for (int ps = 0; ps < TB / 16; ++ps) rb[ps] = brow[ps] ? *reinterpret_cast<const double2*>(brow[ps] + k0 + lk) : make_double2(0.0, 0.0);
    };
    auto lstore = [&](int st) {
#pragma unroll
        for (int ps = 0; ps < TA / 16; ++ps) *reinterpret_cast<double2*>(&As[st][(ps * 16 + lr) * PITCH + lk]) = make_double2(asign * ra[ps].x, asign * ra[ps].y);
#pragma unroll
        for (int ps = 0; ps < TB / 16; ++ps) *reinterpret_cast<double2*>(&Bs[st][(ps * 16 + lr) * PITCH + lk]) = rb[ps];
    };
    const int nchunks = K / KC;
    if (nchunks > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int st = c & 1;
        if (c + 1 < nchunks) gload((c + 1) * KC);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            double af[NI], bf[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) af[i] = As[st][(wr * HA + i * 16 + (lane & 15)) * PITCH + kk + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = Bs[st][(wc * HB + j * 16 + (lane & 15)) * PITCH + kk + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < nchunks) lstore(st ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bi * TA + wr * HA + i * 16 + (lane >> 4) + 4 * r;
                const int col = bj * TS + wc * HB + j * 16 + (lane & 15);
                if (row < Ma && col < Mb) C[(int64_t)row * ldc + col] = acc[i][j][r];
            }
}
__global__ __launch_bounds__(256) void k_gemm_nt(AsmBt abt, const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb, const double* C0, int64_t ldc0, double* C, int64_t ldc, int Ma, int Mb, int K, int mode) {
    ASM_BARGS(abt, A, lda, B, ldb, C0, ldc0, C, ldc, Ma, Mb, K, mode);
    gemm_nt_body<64>(A, lda, B, ldb, C0, ldc0, C, ldc, Ma, Mb, K, mode);
}
__global__ __launch_bounds__(256) void k_gemm_nt32(AsmBt abt, const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb, const double* C0, int64_t ldc0, double* C, int64_t ldc, int Ma, int Mb, int K, int mode) {
    ASM_BARGS(abt, A, lda, B, ldb, C0, ldc0, C, ldc, Ma, Mb, K, mode);
    gemm_nt_body<32>(A, lda, B, ldb, C0, ldc0, C, ldc, Ma, Mb, K, mode);
}
__global__ __launch_bounds__(256) void k_gemm_nt32w(AsmBt abt, const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb, const double* C0, int64_t ldc0, double* C, int64_t ldc, int Ma, int Mb, int K, int mode) {
    ASM_BARGS(abt, A, lda, B, ldb, C0, ldc0, C, ldc, Ma, Mb, K, mode);
    gemm_nt_body<32, 96>(A, lda, B, ldb, C0, ldc0, C, ldc, Ma, Mb, K, mode);
}

// ---------------------------------------------------------------------------------------------------
// index lists of the hard equality rows E / inequality rows I (static per LP skeleton): epos[i] = position in E or -1,
// ipos[i] = position in I or -1
struct NsIdx {
    const int *Eidx, *Epos, *Iidx, *Ipos;
    int nE, nI;
};

// R[c, :] = column J[c] of A_EF on the rows E (dense row of length ldr, zero elsewhere).  One workgroup per right-hand side.
__global__ __launch_bounds__(256) void k_ns_rhs_cols(AsmBt abt, const int* __restrict__ cptr, const int* __restrict__ crow, const int* __restrict__ cpos, const double* __restrict__ vals, NsIdx X, const int* __restrict__ J, const double* __restrict__ Fm, double* __restrict__ R, int64_t ldr) {
    ASM_BARGS(abt, cptr, crow, cpos, vals, X, J, Fm, R, ldr);
    const int c = blockIdx.x;
    double* row = R + (int64_t)c * ldr;
    for (int64_t e = threadIdx.x; e < ldr; e += 256) row[e] = 0.0;
    __syncthreads();
    const int j = J ? J[c] : c;
    if (Fm[j] == 0.0) return;
    for (int k = cptr[j] + threadIdx.x; k < cptr[j + 1]; k += 256) {
        const int ep = X.Epos[crow[k]];
        if (ep >= 0) row[ep] = vals[cpos[k]];
    }
}

// PJ[c, j] = Fm_j (delta(j, J[c]) - sum_{i in E, A_ij != 0} W[c, epos(i)] A_ij)      (rows of the projector P)
__global__ __launch_bounds__(256) void k_ns_pj(AsmBt abt, const int* __restrict__ cptr, const int* __restrict__ crow, const int* __restrict__ cpos, const double* __restrict__ vals, NsIdx X, const int* __restrict__ J, const double* __restrict__ Fm, const double* __restrict__ W, int64_t ldw, double* PJ, int64_t ldp, int64_t n, int64_t ldn, int second_pass) {
    ASM_BARGS(abt, cptr, crow, cpos, vals, X, J, Fm, W, ldw, PJ, ldp, n, ldn, second_pass);
    const int c = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldn) return;
    double v = 0.0;
    if (j < n && Fm[j] != 0.0) {
        const double* w = W + (int64_t)c * ldw;
        double acc = 0.0;
        for (int k = cptr[j]; k < cptr[j + 1]; ++k) {
            const int ep = X.Epos[crow[k]];
            if (ep >= 0) acc += w[ep] * vals[cpos[k]];
        }
        // first pass: row J[c] of the projector; second pass: the row already in PJ projected once more (z - A_EF' S0^-1 A_EF z)
        v = (second_pass ? PJ[(int64_t)c * ldp + j] : (j == J[c] ? 1.0 : 0.0)) - acc;
    }
    PJ[(int64_t)c * ldp + j] = v;
}

// R[c, e] = sum_j A[Eidx[e], j] Zt[c, j]   (the equality rows applied to the basis rows: right-hand sides of the second projection pass)
__global__ __launch_bounds__(256) void k_ns_rows_e(AsmBt abt, const int* __restrict__ rptr, const int* __restrict__ rcol, const double* __restrict__ vals, NsIdx X, const double* __restrict__ Fm, const double* __restrict__ Zt, int64_t ldz, double* __restrict__ R, int64_t ldr) {
    ASM_BARGS(abt, rptr, rcol, vals, X, Fm, Zt, ldz, R, ldr);
    const int c = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ldr) return;
    double acc = 0.0;
    if (t < X.nE) {
        const int i = X.Eidx[t];
        const double* z = Zt + (int64_t)c * ldz;
        for (int k = rptr[i]; k < rptr[i + 1]; ++k) acc += vals[k] * Fm[rcol[k]] * z[rcol[k]];      // A_EF: fixed columns do not count
    }
    R[(int64_t)c * ldr + t] = acc;
}

// T[c, d] = PJ[c, J[d]]  (d <= c; the lower triangle of P[J, J]) into the factor buffer of the small system
__global__ __launch_bounds__(256) void k_ns_gather_t(AsmBt abt, const double* __restrict__ PJ, int64_t ldp, const int* __restrict__ J, int k, double* __restrict__ T, int64_t ldt) {
    ASM_BARGS(abt, PJ, ldp, J, k, T, ldt);
    const int c = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d <= c && d < k) T[(int64_t)c * ldt + d] = PJ[(int64_t)c * ldp + J[d]];
}

// v[i] = val for i < len
__global__ __launch_bounds__(256) void k_ns_fill(AsmBt abt, double* __restrict__ v, double val, int64_t len) {
    ASM_BARGS(abt, v, val, len);
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < len) v[t] = val;
}

// d[i] = S[i, i]
__global__ __launch_bounds__(256) void k_ns_diag(AsmBt abt, const double* __restrict__ S, int64_t ld, int N, double* __restrict__ d) {
    ASM_BARGS(abt, S, ld, N, d);
    int t = blockIdx.x * 256 + threadIdx.x;
    if (t < N) d[t] = S[(int64_t)t * ld + t];
}

// lower triangle of S[0:N, 0:N] := diag(dvec) (strictly lower part zero); dvec == nullptr: zero diagonal as well
__global__ __launch_bounds__(256) void k_ns_set_diag(AsmBt abt, double* __restrict__ S, int64_t ld, int N, const double* __restrict__ dvec) {
    ASM_BARGS(abt, S, ld, N, dvec);
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j <= i && j < N) S[(int64_t)i * ld + j] = (j == i && dvec) ? dvec[i] : 0.0;
}

// cnt[0] = number of guarded pivots (diagonal entries of the factor >= big): one workgroup
__global__ __launch_bounds__(1024) void k_ns_count_big(AsmBt abt, const double* __restrict__ S, int64_t ld, int N, double big, int* __restrict__ cnt) {
    ASM_BARGS(abt, S, ld, N, big, cnt);
    __shared__ int sh[16];
    int c = 0;
    for (int i = threadIdx.x; i < N; i += 1024) c += S[(int64_t)i * ld + i] >= big ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) t += sh[k];
        cnt[0] = t;
    }
}

// In place  Zt = L^-1 PJ  (forward substitution down the k rows, one thread per column; L = k x k lower factor, pitch ldl).
// The rows already finished are read back from Zt itself (coalesced across the workgroup, L2 resident); the row of L is
// staged in LDS 64 entries at a time.
__global__ __launch_bounds__(256) void k_ns_ortho(AsmBt abt, const double* __restrict__ L, int64_t ldl, int k, double* __restrict__ Zt, int64_t ldz, int64_t ncols) {
    ASM_BARGS(abt, L, ldl, k, Zt, ldz, ncols);
    __shared__ double lrow[64];
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = j < ncols;
    for (int c = 0; c < k; ++c) {
        double acc = live ? Zt[(int64_t)c * ldz + j] : 0.0;
        for (int d0 = 0; d0 < c; d0 += 64) {
            const int nd = min(64, c - d0);
            __syncthreads();
            if ((int)threadIdx.x < nd) lrow[threadIdx.x] = L[(int64_t)c * ldl + d0 + threadIdx.x];
            __syncthreads();
            if (live)
                for (int d = 0; d < nd; ++d) acc -= lrow[d] * Zt[(int64_t)(d0 + d) * ldz + j];
        }
        if (live) Zt[(int64_t)c * ldz + j] = acc / L[(int64_t)c * ldl + c];
        __threadfence_block();
    }
}

// GIt[c, ipos] = sum_{j in row i} A_ij Zt[c, j]   for the inequality rows i = Iidx[ipos]   (Zt is zero in the fixed columns)
__global__ __launch_bounds__(256) void k_ns_gi(AsmBt abt, const int* __restrict__ rptr, const int* __restrict__ rcol, const double* __restrict__ vals, NsIdx X, const double* __restrict__ Zt, int64_t ldz, double* __restrict__ GIt, int nIp) {
    ASM_BARGS(abt, rptr, rcol, vals, X, Zt, ldz, GIt, nIp);
    const int c = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nIp) return;
    double acc = 0.0;
    if (t < X.nI) {
        const int i = X.Iidx[t];
        const double* z = Zt + (int64_t)c * ldz;
        for (int k = rptr[i]; k < rptr[i + 1]; ++k) acc += vals[k] * z[rcol[k]];
    }
    GIt[(int64_t)c * ldz + t] = acc;
}

// theta~ = [ (muL/tL + muU/tU + rho) Fm  (n, zero padded to ldn) | 1/dS on the inequality rows (nI, zero padded to nIp) ]
__global__ __launch_bounds__(256) void k_ns_theta(AsmBt abt, IpmPtrs P, NsIdx X, double rho_p, double* __restrict__ th, int64_t ldn, int nIp) {
    ASM_BARGS(abt, P, X, rho_p, th, ldn, nIp);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < ldn) {
        double v = 0.0;
        if (t < P.n && P.ub[t] > P.lb[t]) v = P.muL[t] / P.tL[t] + P.muU[t] / P.tU[t] + rho_p;
        th[t] = v;
    }
    if (t < nIp) th[ldn + t] = t < X.nI ? 1.0 / P.dS[X.Iidx[t]] : 0.0;
}

// k_ipm_theta + k_ns_theta in one launch (the reciprocal of dS on the inequality rows is formed from the row's own terms: another thread of
// this launch writes P.dS)
__global__ __launch_bounds__(256) void k_ipm_theta_ns(AsmBt abt, IpmPtrs P, double rho_p, NsIdx X, double* __restrict__ th, int64_t ldn, int nIp) {
    ASM_BARGS(abt, P, rho_p, X, th, ldn, nIp);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    auto dS_of = [&](int64_t i) {
        double d = P.rtype[i] != 0 ? P.g[i] / P.pi[i] : 0.0;
        if (P.ns) {
            const int k0 = P.rs0[i], k1 = P.rs1[i];
            if (k0 >= 0) d += P.ts[k0] / P.mus[k0];
            if (k1 >= 0) d += P.ts[k1] / P.mus[k1];
        }
        return d;
    };
    if (t < P.n) {
        const bool fr = P.ub[t] > P.lb[t];
        P.thp_inv[t] = fr ? 1.0 / (P.muL[t] / P.tL[t] + P.muU[t] / P.tU[t] + rho_p) : 0.0;
    }
    if (t < P.ns) P.ths_inv[t] = P.ts[t] / P.mus[t];
    if (t < P.M) P.dS[t] = dS_of(t);
    if (t < ldn) {
        double v = 0.0;
        if (t < P.n && P.ub[t] > P.lb[t]) v = P.muL[t] / P.tL[t] + P.muU[t] / P.tU[t] + rho_p;
        th[t] = v;
    }
    if (t < nIp) th[ldn + t] = t < X.nI ? 1.0 / dS_of(X.Iidx[t]) : 0.0;
}

// ---- vector kernels of the null-space Newton solve (oracle: IPM.run, solve_ns).  thI = theta~ + ldn = D_I^-1 by position in I.
// out[e] = scale * r[Eidx[e]]
__global__ __launch_bounds__(256) void k_ns_gather_e(AsmBt abt, NsIdx X, const double* __restrict__ r, double scale, double* __restrict__ out) {
    ASM_BARGS(abt, X, r, scale, out);
    int t = blockIdx.x * 256 + threadIdx.x;
    if (t < X.nE) out[t] = scale * r[X.Eidx[t]];
}
// yM[i] = tE[epos] on the equality rows, zero on the inequality rows
__global__ __launch_bounds__(256) void k_ns_rowvec_e(AsmBt abt, NsIdx X, const double* __restrict__ tE, double* __restrict__ yM, int64_t M) {
    ASM_BARGS(abt, X, tE, yM, M);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int ep = X.Epos[i];
    yM[i] = ep >= 0 ? tE[ep] : 0.0;
}
// inequality rows: bI = -res rp + sg rcg / pi,  yM = D_I^-1 bI ; equality rows: both zero
__global__ __launch_bounds__(256) void k_ns_bi(AsmBt abt, IpmPtrs P, NsIdx X, const double* __restrict__ thI, double res, double* __restrict__ bI, double* __restrict__ yM) {
    ASM_BARGS(abt, P, X, thI, res, bI, yM);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P.M) return;
    const int ip = X.Ipos[i];
    double b = 0.0, y = 0.0;
    if (ip >= 0) {
        b = -res * P.rp[i] + (double)P.rtype[i] * P.rcg[i] / P.pi[i];
        y = thI[ip] * b;
    }
    bI[i] = b;
    yM[i] = y;
}
// wM[i] = D_I^-1 aM[i] on the inequality rows, zero on the equality rows
__global__ __launch_bounds__(256) void k_ns_wm(AsmBt abt, NsIdx X, const double* __restrict__ thI, const double* __restrict__ aM, double* __restrict__ wM, int64_t M) {
    ASM_BARGS(abt, X, thI, aM, wM, M);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int ip = X.Ipos[i];
    wM[i] = ip >= 0 ? thI[ip] * aM[i] : 0.0;
}
// free columns (th != 0):  out = th x + atw - (h ? h : 0) ; fixed / padded columns: 0       (K x - h,  K = Th + A_I' D_I^-1 A_I)
__global__ __launch_bounds__(256) void k_ns_kx(AsmBt abt, const double* __restrict__ th, const double* __restrict__ x, const double* __restrict__ atw, const double* __restrict__ h, double* __restrict__ out, int64_t ldn) {
    ASM_BARGS(abt, th, x, atw, h, out, ldn);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldn) return;
    const double t = th[j];
    out[j] = t != 0.0 ? t * x[j] + atw[j] - (h ? h[j] : 0.0) : 0.0;
}
// x[j] = 0 on the fixed / padded columns (th == 0)
__global__ __launch_bounds__(256) void k_ns_mask(AsmBt abt, double* __restrict__ x, const double* __restrict__ th, int64_t ldn) {
    ASM_BARGS(abt, x, th, ldn);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < ldn && th[j] == 0.0) x[j] = 0.0;
}
// h~ = hp + A_I' D_I^-1 bI on the free columns (zero elsewhere) ;  v = h~ - res K dpbar
__global__ __launch_bounds__(256) void k_ns_ht(AsmBt abt, const double* __restrict__ th, const double* __restrict__ hp, const double* __restrict__ atw, const double* __restrict__ kdpb, double res, double* __restrict__ ht, double* __restrict__ v, int64_t n, int64_t ldn) {
    ASM_BARGS(abt, th, hp, atw, kdpb, res, ht, v, n, ldn);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldn) return;
    double h = 0.0, o = 0.0;
    if (j < n && th[j] != 0.0) {
        h = hp[j] + atw[j];
        o = h - res * kdpb[j];
    }
    ht[j] = h;
    v[j] = o;
}
// out = rhs - N0 x  with N0 symmetric, lower triangle stored (pitch ld): one wavefront per row
__global__ __launch_bounds__(256) void k_ns_symv_res(AsmBt abt, const double* __restrict__ N0, int64_t ld, int k, const double* __restrict__ x, const double* __restrict__ rhs, double* __restrict__ out) {
    ASM_BARGS(abt, N0, ld, k, x, rhs, out);
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= k) return;
    double acc = 0.0;
    for (int j = lane; j <= i; j += 64) acc = fma(N0[(int64_t)i * ld + j], x[j], acc);
    for (int j = i + 1 + lane; j < k; j += 64) acc = fma(N0[(int64_t)j * ld + i], x[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[i] = rhs[i] - acc;
}
// copy of the lower triangle  dst[i, j] = src[i, j], j <= i < k;  full: mirrored into the upper triangle as well
__global__ __launch_bounds__(256) void k_ns_copy_lower(AsmBt abt, const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd, int k, int full) {
    ASM_BARGS(abt, src, lds_, dst, ldd, k, full);
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j <= i && j < k) {
        const double v = src[(int64_t)i * lds_ + j];
        dst[(int64_t)i * ldd + j] = v;
        if (full) dst[(int64_t)j * ldd + i] = v;      // small systems (k_ns_reduced_solve): the copy is stored in full, its products read rows, never columns
    }
}
// lower triangle of the sum of the split-K slices (fixed order), written to the factor buffer and to the unregularised copy
__global__ __launch_bounds__(256) void k_ns_reduce_lower(AsmBt abt, const double* __restrict__ parts, int nsplit, int64_t pstride, int64_t ld, double* __restrict__ S, double* __restrict__ N0, int k, int full, double* __restrict__ diag0, double rel, double absv) {
    ASM_BARGS(abt, parts, nsplit, pstride, ld, S, N0, k, full, diag0, rel, absv);
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j <= i && j < k) {
        const int64_t o = (int64_t)i * ld + j;
        double v = parts[o];
        for (int s_ = 1; s_ < nsplit; ++s_) v += parts[(int64_t)s_ * pstride + o];
        // diag0 != null: k_diag_prepare (mode 0) on the way - diag0[i] = S_ii, S_ii += rel S_ii + absv (the copy keeps the unregularised value)
        S[o] = (diag0 && j == i) ? v + (rel * v + absv) : v;
        if (diag0 && j == i) diag0[i] = v;
        N0[o] = v;
        if (full) N0[(int64_t)j * ld + i] = v;          // (see k_ns_copy_lower)
    }
}
// x = a + b
__global__ __launch_bounds__(256) void k_ns_add(AsmBt abt, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ x, int64_t len) {
    ASM_BARGS(abt, a, b, x, len);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < len) x[j] = a[j] + b[j];
}
// dp = res dpbar + Z du on the free columns, with the bound multipliers' directions (k_ipm_dir's column part)
__global__ __launch_bounds__(256) void k_ns_dp(AsmBt abt, IpmPtrs P, IpmDir D, const double* __restrict__ th, const double* __restrict__ dpb, double res, const double* __restrict__ zu, int64_t ldn) {
    ASM_BARGS(abt, P, D, th, dpb, res, zu, ldn);
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ldn) return;
    double dp = 0.0, dL = 0.0, dU = 0.0;
    if (t < P.n && th[t] != 0.0) {
        dp = res * dpb[t] + zu[t];
        dL = (P.rcL[t] - P.muL[t] * dp) / P.tL[t];
        dU = (P.rcU[t] + P.muU[t] * dp) / P.tU[t];
    }
    D.dp[t] = dp;
    if (t < P.n) { D.dmuL[t] = dL; D.dmuU[t] = dU; }
}
// inequality rows: dy = D_I^-1 (bI - aM), dpi = sg dy, dg = (rcg - g dpi)/pi, wM = D_I^-1 aM ; equality rows: dy = dpi = dg = wM = 0
__global__ __launch_bounds__(256) void k_ns_rows(AsmBt abt, IpmPtrs P, IpmDir D, NsIdx X, const double* __restrict__ thI, const double* __restrict__ bI, const double* __restrict__ aM, double* __restrict__ wM) {
    ASM_BARGS(abt, P, D, X, thI, bI, aM, wM);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P.M) return;
    const int ip = X.Ipos[i];
    double dy = 0.0, dpi = 0.0, dg = 0.0, w = 0.0;
    if (ip >= 0) {
        dy = thI[ip] * (bI[i] - aM[i]);
        dpi = (double)P.rtype[i] * dy;
        dg = (P.rcg[i] - P.g[i] * dpi) / P.pi[i];
        w = thI[ip] * aM[i];
    }
    D.dy[i] = dy;
    D.dpi[i] = dpi;
    D.dg[i] = dg;
    wM[i] = w;
}
// ---- fused forms of the null-space Newton solve (round 4: a null-space iteration is a chain of ~45 small launches; every pair
// "sparse product, then a row- or column-wise kernel on its result" and every pair of adjacent element-wise kernels is one launch here).
// The arithmetic of each element is the separate kernels' own, statement by statement.
// k_ipm_rhs1 (complementarity right-hand sides) + k_ns_bi (its inequality-row part: row i needs only rcg[i])
__global__ __launch_bounds__(256) void k_ns_rhs1_bi(AsmBt abt, IpmPtrs P, IpmDir A, int mode, NsIdx X, const double* __restrict__ thI, double res_bi, double* __restrict__ bI, double* __restrict__ yM) {
    ASM_BARGS(abt, P, A, mode, X, thI, res_bi, bI, yM);
    int64_t t = blockIdx.x * 256 + threadIdx.x;
    const double sm = mode ? P.scal[SC_SM] : 0.0;
    const double res = 1.0;
    if (t < P.n) {
        bool fr = P.ub[t] > P.lb[t];
        double rcL = sm - P.tL[t] * P.muL[t];
        double rcU = sm - P.tU[t] * P.muU[t];
        if (mode) {
            rcL -= A.dp[t] * A.dmuL[t];
            rcU += A.dp[t] * A.dmuU[t];
        }
        P.rcL[t] = rcL;
        P.rcU[t] = rcU;
        double hp = fr ? -res * P.rdp[t] + rcL / P.tL[t] - rcU / P.tU[t] : 0.0;
        P.hp[t] = hp;
        P.tmpn[t] = P.thp_inv[t] * hp;
    }
    if (t < P.ns) {
        double rcs = sm - P.ts[t] * P.mus[t];
        if (mode) rcs -= A.ds[t] * A.dmus[t];
        P.rcs[t] = rcs;
        P.hs[t] = -res * P.rds[t] + rcs / P.ts[t];
    }
    if (t < P.M) {
        double rcg = sm - P.g[t] * P.pi[t];
        if (mode) rcg -= A.dg[t] * A.dpi[t];
        P.rcg[t] = rcg;
        const int ip = X.Ipos[t];
        double b = 0.0, y = 0.0;
        if (ip >= 0) {
            b = -res_bi * P.rp[t] + (double)P.rtype[t] * rcg / P.pi[t];
            y = thI[ip] * b;
        }
        bI[t] = b;
        yM[t] = y;
    }
}
// k_spmv_t (atw = Ah' yM, eight lanes per column) + k_ns_ht
__global__ __launch_bounds__(256) void k_ns_spmvt_ht(AsmBt abt, const int* __restrict__ cptr, const int* __restrict__ row, const int* __restrict__ pos, const double* __restrict__ vals, const double* __restrict__ y,
                                                     const double* __restrict__ th, const double* __restrict__ hp, const double* __restrict__ kdpb, double res, double* __restrict__ ht, double* __restrict__ v, int64_t n, int64_t ldn) {
    ASM_BARGS(abt, cptr, row, pos, vals, y, th, hp, kdpb, res, ht, v, n, ldn);
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    double acc = 0.0;
    if (j < n)
        for (int k = cptr[j] + sub; k < cptr[j + 1]; k += 8) acc += vals[pos[k]] * y[row[k]];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (sub == 0 && j < ldn) {
        double h = 0.0, o = 0.0;
        if (j < n && th[j] != 0.0) {
            h = hp[j] + acc;
            o = h - res * kdpb[j];
        }
        ht[j] = h;
        v[j] = o;
    }
}
// k_spmv_t (atw = Ah' wM) + k_ns_kx (h == nullptr form)
__global__ __launch_bounds__(256) void k_ns_spmvt_kx(AsmBt abt, const int* __restrict__ cptr, const int* __restrict__ row, const int* __restrict__ pos, const double* __restrict__ vals, const double* __restrict__ y,
                                                     const double* __restrict__ th, const double* __restrict__ x, double* __restrict__ out, int64_t n, int64_t ldn) {
    ASM_BARGS(abt, cptr, row, pos, vals, y, th, x, out, n, ldn);
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    double acc = 0.0;
    if (j < n)
        for (int k = cptr[j] + sub; k < cptr[j + 1]; k += 8) acc += vals[pos[k]] * y[row[k]];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (sub == 0 && j < ldn) {
        const double t = th[j];
        out[j] = t != 0.0 ? t * x[j] + acc : 0.0;
    }
}
// k_spmv_n (aM = Ah dpbar) + k_ns_wm
__global__ __launch_bounds__(256) void k_ns_spmvn_wm(AsmBt abt, const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ vals, const double* __restrict__ x, NsIdx X,
                                                     const double* __restrict__ thI, double* __restrict__ wM, int64_t M) {
    ASM_BARGS(abt, ptr, col, vals, x, X, thI, wM, M);
    int64_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    double acc = 0.0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) acc += vals[k] * x[col[k]];
    const int ip = X.Ipos[i];
    wM[i] = ip >= 0 ? thI[ip] * acc : 0.0;
}
// k_ns_neg_clear + k_ns_spmvn_wm: dpbar = -e is formed on the way (stored by the first len threads), the products read -e directly
__global__ __launch_bounds__(256) void k_ns_spmvn_wm_neg(AsmBt abt, const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ vals, const double* __restrict__ e,
                                                         double* __restrict__ dpb, int64_t len, double* __restrict__ clear, NsIdx X, const double* __restrict__ thI, double* __restrict__ wM, int64_t M) {
    ASM_BARGS(abt, ptr, col, vals, e, dpb, len, clear, X, thI, wM, M);
    int64_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < len) dpb[i] = -e[i];
    if (i == 0) *clear = 0.0;
    if (i >= M) return;
    double acc = 0.0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) acc += vals[k] * (-e[col[k]]);
    const int ip = X.Ipos[i];
    wM[i] = ip >= 0 ? thI[ip] * acc : 0.0;
}
// k_spmv_n (aM = Ah dp) + k_ns_rows
__global__ __launch_bounds__(256) void k_ns_spmvn_rows(AsmBt abt, const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ vals, IpmPtrs P, IpmDir D, NsIdx X,
                                                       const double* __restrict__ thI, const double* __restrict__ bI, double* __restrict__ wM) {
    ASM_BARGS(abt, ptr, col, vals, P, D, X, thI, bI, wM);
    int64_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.M) return;
    double acc = 0.0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) acc += vals[k] * D.dp[col[k]];
    const int ip = X.Ipos[i];
    double dy = 0.0, dpi = 0.0, dg = 0.0, w = 0.0;
    if (ip >= 0) {
        dy = thI[ip] * (bI[i] - acc);
        dpi = (double)P.rtype[i] * dy;
        dg = (P.rcg[i] - P.g[i] * dpi) / P.pi[i];
        w = thI[ip] * acc;
    }
    D.dy[i] = dy;
    D.dpi[i] = dpi;
    D.dg[i] = dg;
    wM[i] = w;
}
// k_ipm_update + k_ns_scale of e (the component of the iterate outside pbar + null(A_EF) shrinks by 1 - a)
__global__ __launch_bounds__(256) void k_ns_update(AsmBt abt, IpmPtrs P, IpmDir C, double al, double be, double* __restrict__ e, double es, int64_t ldn) {
    ASM_BARGS(abt, P, C, al, be, e, es, ldn);
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
    if (t < ldn) e[t] *= es;
}
// The same update with the step lengths taken from the scalar block on the device (k_ipm_steps left them there): the host does not wait
// for them - one read-back per null-space iteration (the measures) instead of two.  al = min(1, eta ap), be = min(1, eta ad) as the host
// formed them.  An iteration whose reduced solves lost their accuracy (SC_NSERR > rerr) leaves the iterate alone: the host sees the same
// number with the next measures and redoes the iteration in row form (Solver::ipm_run).
__global__ __launch_bounds__(256) void k_ns_update_dev(AsmBt abt, IpmPtrs P, IpmDir C, double eta, double* __restrict__ e, int64_t ldn, double rerr) {
    ASM_BARGS(abt, P, C, eta, e, ldn, rerr);
    if (P.scal[SC_NSERR] > rerr) return;
    const double al = fmin(1.0, eta * P.scal[SC_AP]), be = fmin(1.0, eta * P.scal[SC_AD]), es = 1.0 - al;
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
    if (t < ldn) e[t] *= es;
}
// k_ns_neg (dpbar = -e) + clearing the accumulated residual measure of the iteration's reduced solves
__global__ __launch_bounds__(256) void k_ns_neg_clear(AsmBt abt, const double* __restrict__ x, double* __restrict__ out, int64_t len, double* __restrict__ clear) {
    ASM_BARGS(abt, x, out, len, clear);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < len) out[j] = -x[j];
    if (j == 0) *clear = 0.0;
}

// out[Eidx[e]] = tE[e]   (add != 0: += )
__global__ __launch_bounds__(256) void k_ns_scatter_e(AsmBt abt, NsIdx X, const double* __restrict__ tE, double* __restrict__ out, int add) {
    ASM_BARGS(abt, X, tE, out, add);
    int t = blockIdx.x * 256 + threadIdx.x;
    if (t < X.nE) out[X.Eidx[t]] = (add ? out[X.Eidx[t]] : 0.0) + tE[t];
}
// slot[0] = max_j |th_j dp_j - aty_j - hp_j| over the free columns: the dual-equation error of the step (one workgroup)
__global__ __launch_bounds__(1024) void k_ns_err(AsmBt abt, const double* __restrict__ th, const double* __restrict__ dp, const double* __restrict__ aty, const double* __restrict__ hp, int64_t n, double* __restrict__ slot) {
    ASM_BARGS(abt, th, dp, aty, hp, n, slot);
    __shared__ double sh[16];
    double m = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 1024)
        if (th[j] != 0.0) m = fmax(m, fabs(th[j] * dp[j] - aty[j] - hp[j]));
    m = blk_reduce_max(m, sh);
    if (threadIdx.x == 0) slot[0] = m;
}

// ---------------------------------------------------------------------------------------------------
// Active-set (equality-constrained) solves in the reduced coordinates - oracle/lp_solver.py: eqp_ns.
// Working set of a normal-phase LP = the equality rows (always) + bound-active non-fixed variables B + active inequality rows Ia:
// nact constraints C u = d on p = pbar + Z u, rows of C = columns sel[] of Gt (bounds: column j; rows: column ldn + ipos).
struct NsEq {
    int *sel, *bpos, *rpos, *cnt;     // constraint -> column of Gt ; variable -> constraint (-1) ; position in I -> constraint (-1) ; {nB, nact}
    double *Csel, *d, *u, *lam, *v, *w, *pbar, *tbar, *u0, *qh;
    int64_t ldc;                      // pitch of Csel (>= k, multiple of 32)
};
// ordered compaction of the working set (one workgroup): bounds by variable index, then inequality rows by index; ksoft := -1
__global__ __launch_bounds__(1024) void k_nseq_setup(AsmBt abt, AsPtrs A, AsSets S, NsIdx X, NsEq Q, int64_t ldn) {
    ASM_BARGS(abt, A, S, X, Q, ldn);
    __shared__ int sh_cnt[16];
    int base = 0;
    for (int64_t j0 = 0; j0 < A.n; j0 += 1024) {
        const int64_t j = j0 + threadIdx.x;
        const bool f = j < A.n && S.bst[j] != 0 && A.ub[j] > A.lb[j];
        const int pos = blk_compact_pos(f, base, sh_cnt);
        if (j < A.n) Q.bpos[j] = f ? pos : -1;
        if (f) Q.sel[pos] = (int)j;
    }
    const int nB = base;
    for (int t0 = 0; t0 < X.nI; t0 += 1024) {
        const int t = t0 + threadIdx.x;
        const bool f = t < X.nI && S.rowst[X.Iidx[t]] == 1;
        const int pos = blk_compact_pos(f, base, sh_cnt);
        if (t < X.nI) Q.rpos[t] = f ? pos : -1;
        if (f) Q.sel[pos] = (int)ldn + t;
    }
    for (int64_t i = threadIdx.x; i < A.M; i += 1024) A.ksoft[i] = -1;
    if (threadIdx.x == 0) { Q.cnt[0] = nB; Q.cnt[1] = base; }
}
// Csel[a, c] = Gt[c, sel[a]]  (c < k; zero in the padding columns) and the right-hand side d[a]
__global__ __launch_bounds__(256) void k_nseq_gather(AsmBt abt, AsPtrs A, AsSets S, NsIdx X, NsEq Q, const double* __restrict__ Gt, int64_t ldg, int k, int64_t ldn) {
    ASM_BARGS(abt, A, S, X, Q, Gt, ldg, k, ldn);
    const int a = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int col = Q.sel[a];
    if (c < Q.ldc) Q.Csel[(int64_t)a * Q.ldc + c] = c < k ? Gt[(int64_t)c * ldg + col] : 0.0;
    if (c == 0) {
        double dv;
        if (col < ldn) dv = (S.bst[col] < 0 ? A.lb[col] : A.ub[col]) - Q.pbar[col];
        else { const int i = X.Iidx[col - ldn]; dv = A.r[i] - Q.tbar[i]; }
        Q.d[a] = dv;
    }
}
// pfix: fixed columns at their value, zero elsewhere
__global__ __launch_bounds__(256) void k_nseq_pfix(AsmBt abt, AsPtrs A, double* __restrict__ pfix, int64_t ldn) {
    ASM_BARGS(abt, A, pfix, ldn);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < ldn) pfix[j] = (j < A.n && !(A.ub[j] > A.lb[j])) ? A.lb[j] : 0.0;
}
// out[e] = r[Eidx[e]] - aM[Eidx[e]]
__global__ __launch_bounds__(256) void k_nseq_be(AsmBt abt, AsPtrs A, NsIdx X, const double* __restrict__ aM, double* __restrict__ out) {
    ASM_BARGS(abt, A, X, aM, out);
    int t = blockIdx.x * 256 + threadIdx.x;
    if (t < X.nE) out[t] = A.r[X.Eidx[t]] - aM[X.Eidx[t]];
}
// pbar = pfix + (masked) x ;  vz = Fm (clip0 - pbar)   (input of u0 = Zt vz)
__global__ __launch_bounds__(256) void k_nseq_pbar(AsmBt abt, AsPtrs A, const double* __restrict__ pfix, const double* __restrict__ x, const double* __restrict__ zero, double* __restrict__ pbar, double* __restrict__ vz, int64_t ldn) {
    ASM_BARGS(abt, A, pfix, x, zero, pbar, vz, ldn);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldn) return;
    const bool fr = j < A.n && A.ub[j] > A.lb[j];
    const double pb = pfix[j] + (fr ? x[j] : 0.0);
    pbar[j] = pb;
    vz[j] = fr ? zero[j] - pb : 0.0;
}
// out = a - b  (len)
__global__ __launch_bounds__(256) void k_nseq_sub(AsmBt abt, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int64_t len) {
    ASM_BARGS(abt, a, b, out, len);
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < len) out[t] = a[t] - b[t];
}
// p = pbar + Z u on the free columns, bound-active variables exactly on their bound, fixed columns at their value
__global__ __launch_bounds__(256) void k_nseq_p(AsmBt abt, AsPtrs A, AsSets S, NsEq Q, const double* __restrict__ zu, int64_t ldn) {
    ASM_BARGS(abt, A, S, Q, zu, ldn);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= A.n) return;
    double v;
    if (!(A.ub[j] > A.lb[j])) v = A.lb[j];
    else if (Q.bpos[j] >= 0) v = S.bst[j] < 0 ? A.lb[j] : A.ub[j];
    else v = Q.pbar[j] + zu[j];
    A.p[j] = v;
}
// y on the inequality rows: the multiplier of the row's constraint (0 when inactive); zero on the equality rows
__global__ __launch_bounds__(256) void k_nseq_yi(AsmBt abt, AsPtrs A, NsIdx X, NsEq Q, double* __restrict__ yM) {
    ASM_BARGS(abt, A, X, Q, yM);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.M) return;
    const int ip = X.Ipos[i];
    double v = 0.0;
    if (ip >= 0 && Q.rpos[ip] >= 0) v = Q.lam[Q.rpos[ip]];
    yM[i] = v;
}
// w = Fm (q - A_Ia' y_Ia) - z_B
__global__ __launch_bounds__(256) void k_nseq_w(AsmBt abt, AsPtrs A, NsEq Q, const double* __restrict__ atw, double* __restrict__ w, int64_t ldn) {
    ASM_BARGS(abt, A, Q, atw, w, ldn);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldn) return;
    double v = 0.0;
    if (j < A.n && A.ub[j] > A.lb[j]) {
        v = A.q[j] - atw[j];
        if (Q.bpos[j] >= 0) v -= Q.lam[Q.bpos[j]];
    }
    w[j] = v;
}
// y = yM with the equality rows from tE
__global__ __launch_bounds__(256) void k_nseq_y(AsmBt abt, AsPtrs A, NsIdx X, const double* __restrict__ yM, const double* __restrict__ tE) {
    ASM_BARGS(abt, A, X, yM, tE);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.M) return;
    const int ep = X.Epos[i];
    A.y[i] = ep >= 0 ? tE[ep] : yM[i];
}

// ---- null-space iteration without solves against S0 (oracle: IPM.run use_ns / IPM.measures / IPM.ns_finish_y)
// d0 = Fm (p - pbar)
__global__ __launch_bounds__(256) void k_ns_e0(AsmBt abt, IpmPtrs P, const double* __restrict__ pbar, double* __restrict__ d0, int64_t ldn) {
    ASM_BARGS(abt, P, pbar, d0, ldn);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldn) return;
    d0[j] = (j < P.n && P.ub[j] > P.lb[j]) ? P.p[j] - pbar[j] : 0.0;
}
// e = d0 - zz on the free columns ; dpb = -e
__global__ __launch_bounds__(256) void k_ns_e1(AsmBt abt, IpmPtrs P, const double* __restrict__ d0, const double* __restrict__ zz, double* __restrict__ e, int64_t ldn) {
    ASM_BARGS(abt, P, d0, zz, e, ldn);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldn) return;
    e[j] = (j < P.n && P.ub[j] > P.lb[j]) ? d0[j] - zz[j] : 0.0;
}
// x *= a
__global__ __launch_bounds__(256) void k_ns_scale(AsmBt abt, double* __restrict__ x, double a, int64_t len) {
    ASM_BARGS(abt, x, a, len);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < len) x[j] *= a;
}
__global__ __launch_bounds__(256) void k_ns_neg(AsmBt abt, const double* __restrict__ x, double* __restrict__ out, int64_t len) {
    ASM_BARGS(abt, x, out, len);
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < len) out[j] = -x[j];
}
// scal[SC_DINF] = max_c |zr_c| / scale_q  (the dual residual inside the null space of the equality rows), then the scalar block goes to the host
__global__ __launch_bounds__(1024) void k_ns_dinf(AsmBt abt, IpmPtrs P, const double* __restrict__ zr, int k, unsigned pub) {
    ASM_BARGS(abt, P, zr, k, pub);
    __shared__ double sh[16];
    double m = 0.0;
    for (int c = threadIdx.x; c < k; c += 1024) m = fmax(m, fabs(zr[c]));
    m = blk_reduce_max(m, sh);
    if (threadIdx.x == 0) P.scal[SC_DINF] = m / P.scale_q;
    scal_publish(P, pub);
}
// slot[0] = max(slot[0], max|r| / max(1, max|rhs|))   (relative residual of a reduced solve; one workgroup)
__global__ __launch_bounds__(1024) void k_ns_relres(AsmBt abt, const double* __restrict__ r, const double* __restrict__ rhs, int k, double* __restrict__ slot) {
    ASM_BARGS(abt, r, rhs, k, slot);
    __shared__ double sh[16];
    double a = 0.0, b = 1.0;
    for (int c = threadIdx.x; c < k; c += 1024) { a = fmax(a, fabs(r[c])); b = fmax(b, fabs(rhs[c])); }
    a = blk_reduce_max(a, sh);
    b = blk_reduce_max(b, sh);
    if (threadIdx.x == 0) slot[0] = fmax(slot[0], a / b);
}

// ---------------------------------------------------------------------------------------------------
// Small systems (order k <= 1024) in ONE workgroup: the k x k matrices of the null-space form are factored in every interior-point
// iteration and solved against a handful of times; with the explicit block inverses + wide-block substitution of the large
// factorisations that is ~10 launches per factorisation and 4 per solve - more launch overhead than arithmetic.  Here the solve uses
// the factor L (lower, pitch ld) and the inverses of its 64 x 64 diagonal blocks (Linv, written by the factorisation), vectors in LDS.
#define ASM_SMALL_MAX 1024      // LDS capacity of the one-workgroup kernels
#define ASM_SMALL_USE 256       // ... used up to this order: beyond it one workgroup is slower than the launches it saves (k = 519: 86 vs 66 ms per C4 LP)
// x (LDS, length k) <- (L L')^-1 x.   t: LDS scratch (k), part: LDS scratch (16 * 64).  Called by all 1024 threads.
__device__ __forceinline__ void small_chol_solve(const double* __restrict__ L, int64_t ld, const double* __restrict__ Linv, int k, double* x, double* t, double* part) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nb = (k + 63) >> 6;
    // forward: L z = x   (z overwrites x block by block)
    for (int b = 0; b < nb; ++b) {
        const int b0 = b << 6;
        {                                                          // t_b = x_b - L[b, 0:b0] z[0:b0]: one wavefront per row, lanes along the row;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};                  // the wavefront's four rows together (their loads in one round trip, each row's sum as before)
            for (int j = lane; j < b0; j += 64) {
                const double xj = x[j];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = fma(L[(int64_t)min(b0 + wv + 16 * u, k - 1) * ld + j], xj, acc[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = wv + 16 * u, row = b0 + rr;
                const double a = wave_sum(acc[u]);
                if (lane == 0) t[rr] = row < k ? x[row] - a : 0.0;
            }
        }
        __syncthreads();
        {                                                          // z_b = Linv_b t_b  (64 x 64, lower triangular)
            const double* Lb = Linv + (int64_t)b * 64 * 64;
            const int row = lane, c0 = wv * 4;
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = fma(Lb[row * 64 + c0 + c], t[c0 + c], acc);
            part[wv * 64 + row] = acc;
        }
        __syncthreads();
        if (tid < 64 && b0 + tid < k) {
            double acc = 0.0;
#pragma unroll
            for (int p = 0; p < 16; ++p) acc += part[p * 64 + tid];
            x[b0 + tid] = acc;
        }
        __syncthreads();
    }
    // backward: L' y = z
    for (int b = nb - 1; b >= 0; --b) {
        const int b0 = b << 6, b1 = min(b0 + 64, k);
        {                                                          // t_b = z_b - L[b1:, b]' y[b1:]: thread = (column of the block, row residue mod 16)
            const int c = lane;
            double acc = 0.0;
            if (b0 + c < k)
                for (int r = b1 + wv; r < k; r += 16) acc = fma(L[(int64_t)r * ld + b0 + c], x[r], acc);
            part[wv * 64 + c] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            double acc = 0.0;
#pragma unroll
            for (int p = 0; p < 16; ++p) acc += part[p * 64 + tid];
            t[tid] = b0 + tid < k ? x[b0 + tid] - acc : 0.0;
        }
        __syncthreads();
        {                                                          // y_b = Linv_b' t_b
            const double* Lb = Linv + (int64_t)b * 64 * 64;
            const int col = lane, r0 = wv * 4;
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = fma(Lb[(r0 + r) * 64 + col], t[r0 + r], acc);
            part[wv * 64 + col] = acc;
        }
        __syncthreads();
        if (tid < 64 && b0 + tid < k) {
            double acc = 0.0;
#pragma unroll
            for (int p = 0; p < 16; ++p) acc += part[p * 64 + tid];
            x[b0 + tid] = acc;
        }
        __syncthreads();
    }
}
// out = (L L')^-1 rhs
__global__ __launch_bounds__(1024) void k_small_solve(AsmBt abt, const double* __restrict__ L, int64_t ld, const double* __restrict__ Linv, int k, const double* __restrict__ rhs, double* __restrict__ out) {
    ASM_BARGS(abt, L, ld, Linv, k, rhs, out);
    __shared__ double x[ASM_SMALL_MAX], t[64], part[16 * 64];
    for (int i = threadIdx.x; i < k; i += 1024) x[i] = rhs[i];
    __syncthreads();
    small_chol_solve(L, ld, Linv, k, x, t, part);
    for (int i = threadIdx.x; i < k; i += 1024) out[i] = x[i];
}
// r (LDS) = rhs - N0 x  with N0 symmetric, stored in full: one wavefront per row
__device__ __forceinline__ void small_symv_res(const double* __restrict__ N0, int64_t ld, int k, const double* x, const double* rhs, double* r) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i0 = wv; i0 < k; i0 += 64) {                          // four rows of a wavefront together: their loads in one round trip, each row's sum as before
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int j = lane; j < k; j += 64) {
            const double xj = x[j];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = fma(N0[(int64_t)min(i0 + 16 * u, k - 1) * ld + j], xj, acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 16 * u;
            const double a = wave_sum(acc[u]);
            if (lane == 0 && i < k) r[i] = rhs[i] - a;
        }
    }
    __syncthreads();
}
// The reduced solve of a null-space Newton step in one launch (oracle: solve_ns): du = N^-1 ru, one refinement sweep on the unregularised
// N0, and slot[0] = max(slot[0], max|ru - N0 du| / max(1, max|ru|)).
__global__ __launch_bounds__(1024) void k_ns_reduced_solve(AsmBt abt, const double* __restrict__ L, int64_t ld, const double* __restrict__ Linv, const double* __restrict__ N0, int k, const double* __restrict__ ru, double* __restrict__ du, double* __restrict__ slot) {
    ASM_BARGS(abt, L, ld, Linv, N0, k, ru, du, slot);
    __shared__ double x[ASM_SMALL_MAX], y[ASM_SMALL_MAX], b[ASM_SMALL_MAX], t[64], part[16 * 64], sh[16];
    for (int i = threadIdx.x; i < k; i += 1024) { b[i] = ru[i]; x[i] = ru[i]; }
    __syncthreads();
    small_chol_solve(L, ld, Linv, k, x, t, part);          // x = du0
    small_symv_res(N0, ld, k, x, b, y);                    // y = ru - N0 du0
    small_chol_solve(L, ld, Linv, k, y, t, part);          // y = correction
    for (int i = threadIdx.x; i < k; i += 1024) x[i] += y[i];
    __syncthreads();
    small_symv_res(N0, ld, k, x, b, y);                    // y = ru - N0 du
    double a = 0.0, m = 1.0;
    for (int i = threadIdx.x; i < k; i += 1024) { a = fmax(a, fabs(y[i])); m = fmax(m, fabs(b[i])); du[i] = x[i]; }
    a = blk_reduce_max(a, sh);
    m = blk_reduce_max(m, sh);
    if (threadIdx.x == 0) slot[0] = fmax(slot[0], a / m);
}
// out[j] = sum_c Zt[c, j] u[c]   (k rows of pitch ld, one thread per column; u staged in LDS) - the basis applied in one launch
__global__ __launch_bounds__(256) void k_gemv_t_small(AsmBt abt, const double* __restrict__ Zt, int64_t ld, int k, const double* __restrict__ u, double* __restrict__ out, int64_t ncols) {
    ASM_BARGS(abt, Zt, ld, k, u, out, ncols);
    __shared__ double us[ASM_SMALL_MAX];
    for (int i = threadIdx.x; i < k; i += 256) us[i] = u[i];
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    double acc = 0.0;
    for (int c = 0; c < k; ++c) acc = fma(Zt[(int64_t)c * ld + j], us[c], acc);
    out[j] = acc;
}

// k_gemv_t_small + k_ns_dp: zu[j] = sum_c Zt[c, j] u[c] goes straight into dp = res dpbar + Z du and the bound multipliers' directions
__global__ __launch_bounds__(256) void k_gemv_t_small_dp(AsmBt abt, const double* __restrict__ Zt, int64_t ld, int k, const double* __restrict__ u, IpmPtrs P, IpmDir D, const double* __restrict__ th,
                                                         const double* __restrict__ dpb, double res, int64_t ldn) {
    ASM_BARGS(abt, Zt, ld, k, u, P, D, th, dpb, res, ldn);
    __shared__ double us[ASM_SMALL_MAX];
    for (int i = threadIdx.x; i < k; i += 256) us[i] = u[i];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ldn) return;
    double acc = 0.0;
    for (int c = 0; c < k; ++c) acc = fma(Zt[(int64_t)c * ld + t], us[c], acc);
    double dp = 0.0, dL = 0.0, dU = 0.0;
    if (t < P.n && th[t] != 0.0) {
        dp = res * dpb[t] + acc;
        dL = (P.rcL[t] - P.muL[t] * dp) / P.tL[t];
        dU = (P.rcU[t] + P.muU[t] * dp) / P.tU[t];
    }
    D.dp[t] = dp;
    if (t < P.n) { D.dmuL[t] = dL; D.dmuU[t] = dU; }
}

// ---- banded S0 = A_EF A_EF' built from its structural pattern (equality rows in reverse Cuthill-McKee order)
// clears row i, columns [i - w + 1, i] (w covers the band rounded up to the factorisation's tiles; the factor of the previous LP lives there)
__global__ __launch_bounds__(256) void k_ns_zero_band(AsmBt abt, double* __restrict__ S, int64_t ld, int nE, int w) {
    ASM_BARGS(abt, S, ld, nE, w);
    const int i = blockIdx.y;
    const int c = i - (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < nE && c >= 0 && (int)(blockIdx.x * 256 + threadIdx.x) < w) S[(int64_t)i * ld + c] = 0.0;
}
// one thread per structural entry (pi, pj), pj <= pi: the dot product of rows Eidx[pi] and Eidx[pj] over the free columns (both rows' column
// lists are sorted: a merge)
__global__ __launch_bounds__(256) void k_ns_s0_sparse(AsmBt abt, const int* __restrict__ pairs, int64_t npairs, const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ val, const int* __restrict__ Eidx, const double* __restrict__ Fm, double* __restrict__ S, int64_t ld) {
    ASM_BARGS(abt, pairs, npairs, ptr, col, val, Eidx, Fm, S, ld);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= npairs) return;
    const int pi = pairs[2 * t], pj = pairs[2 * t + 1];
    const int ri = Eidx[pi], rj = Eidx[pj];
    int a = ptr[ri], b = ptr[rj];
    const int ae = ptr[ri + 1], be = ptr[rj + 1];
    double acc = 0.0;
    while (a < ae && b < be) {
        const int ca = col[a], cb = col[b];
        if (ca == cb) { acc += val[a] * val[b] * Fm[ca]; ++a; ++b; }
        else if (ca < cb) ++a;
        else ++b;
    }
    S[(int64_t)pi * ld + pj] = acc;
}

// Gram matrix of a row list in the handle's banded row order, entry by entry: pair (ri, rj) of rows that share a column -> place (cpos[ri], cpos[rj])
// of the list (skipped when either row is not in it); weights theta over the columns, `diag` (indexed by place) added on the diagonal
__global__ __launch_bounds__(256) void k_schur_sparse(AsmBt abt, const int* __restrict__ pairs, int64_t npairs, const int* __restrict__ cpos, const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ theta, const double* __restrict__ diag, double* __restrict__ S, int64_t ld, const int* __restrict__ vpos) {
    ASM_BARGS(abt, pairs, npairs, cpos, ptr, col, val, theta, diag, S, ld, vpos);
    // vpos: the lists (ptr, col) are the CSC side of the pattern (Gram matrix of COLUMNS: the column form's K = Th + A' D^-1 A) and entry a of
    // a list has its value at val[vpos[a]]; null: CSR, values in list order
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= npairs) return;
    const int ri = pairs[2 * t], rj = pairs[2 * t + 1];
    const int ci = cpos[ri], cj = cpos[rj];
    if (ci < 0 || cj < 0) return;
    int a = ptr[ri], b = ptr[rj];
    const int ae = ptr[ri + 1], be = ptr[rj + 1];
    double acc = 0.0;
    while (a < ae && b < be) {
        const int ca = col[a], cb = col[b];
        if (ca == cb) { acc += (vpos ? val[vpos[a]] * val[vpos[b]] : val[a] * val[b]) * theta[ca]; ++a; ++b; }
        else if (ca < cb) ++a;
        else ++b;
    }
    if (ri == rj && diag) acc += diag[ci];
    S[(int64_t)max(ci, cj) * ld + min(ci, cj)] = acc;
}
