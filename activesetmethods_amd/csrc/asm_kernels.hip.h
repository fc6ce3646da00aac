// Device kernels for the SLP sub-LP hot path on MI355X (gfx950 / CDNA4).  FP64 throughout.
//
// Layout in HBM (all row-major, leading dimensions padded to a multiple of 16 doubles = 128 B so that
// every row starts on a cache-line boundary and the MFMA k-loops need no edge code):
//   J    (Mp x ldn)  dense constraint Jacobian incl. the extra range rows        [assembled each call]
//   Ah   (Mp x ldn)  power-of-two scaled copy  Ah = diag(1/rho) J diag(c)        [one pass each call]
//   S    (Mp x Mp)   Schur complement / its Cholesky factor (lower triangle)     [per factorisation]
//
// Kernels (roofline that bounds each one; bytes/flops per launch are tabulated in DESIGN.md):
//   k_assemble        HBM   COO -> dense J, duplicate accumulation in j_str order (bit-exact vs common.jl:12-20)
//   k_scale_rows      HBM   row max + scaled copy
//   k_gemv_n/t        HBM   Ah x , Ah' y
//   k_syrk<T>         MFMA  S = Ah[idx,:] diag(theta) Ah[idx,:]' (+diag)  and  S22 -= P P'   (v_mfma_f64_16x16x4_f64)
//   k_potrf_diag      LDS   64x64 diagonal block Cholesky with static pivot guard
//   k_trsm_panel      LDS   panel  P = S21 L11^-T
//   k_trtri_*, k_wtrsv_*<WB>  HBM   wide (512 / 1024) block inverses and the wide-block forward / backward substitution
//   k_spmv_*          HBM   matrix-vector products on the CSR / CSC copy of sparse patterns
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "asm_bt.hip.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

#define ASM_NB 64          // Cholesky panel width
#define ASM_KC 32          // k-chunk staged through LDS by k_syrk
#define ASM_MAXCHUNKS 1024  // k-chunk list of k_syrk (sparse Schur build); n <= 32768, else the dense sweep is used
#define ASM_PITCH 34       // LDS row pitch in doubles: KC + 2  (pitch = 2 mod 32 -> conflict-free ds_read_b64 fragments)

// ---------------------------------------------------------------------------------------------------
// Assembly.  One thread per distinct (row, col) of the pattern; its duplicates are summed in the
// original j_str order (perm is a stable sort), starting from 0.0 exactly like `J[r,c] += dE[k]`.
// `adj_off` (>=0) is the image of that entry in the extra `<=` row of a range constraint; it is
// refreshed only when one of the terms is non-zero (stored-entry semantics of subproblem.jl:448-457).
__global__ void k_assemble(AsmBt abt, const double* __restrict__ dE, const int64_t* __restrict__ perm, const int64_t* __restrict__ ustart, const int64_t* __restrict__ uoff, const int64_t* __restrict__ adj_off, double* __restrict__ J, int64_t nu) {
    ASM_BARGS(abt, dE, perm, ustart, uoff, adj_off, J, nu);
    for (int64_t u = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; u < nu; u += (int64_t)gridDim.x * blockDim.x) {
        double acc = 0.0;
        bool any = false;
        for (int64_t k = ustart[u]; k < ustart[u + 1]; ++k) {
            double v = dE[perm[k]];
            acc = acc + v;
            any = any || (v != 0.0);
        }
        J[uoff[u]] = acc;
        int64_t a = adj_off[u];
        if (a >= 0 && any) J[a] = acc;
    }
}

// Fast path when the pattern is duplicate-free and already row-major dense (config C2): a strided copy.
__global__ void k_assemble_dense(AsmBt abt, const double* __restrict__ dE, double* __restrict__ J, int64_t m, int64_t n, int64_t ldn) {
    ASM_BARGS(abt, dE, J, m, n, ldn);
    int64_t total = m * n;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = t / n, j = t - i * n;
        J[i * ldn + j] = 0.0 + dE[t];
    }
}

__device__ __forceinline__ double pow2_round_dev(double x) {
    if (!(x > 0.0)) return 1.0;
    int e;
    double f = frexp(x, &e);
    if (f < 0.70710678118654752) e -= 1;
    return ldexp(1.0, e);
}

__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One workgroup per row: rho_i = pow2(max_j |J_ij c_j|),  Ah_ij = J_ij c_j / rho_i  (all factors are powers of two).
__global__ __launch_bounds__(256) void k_scale_rows(AsmBt abt, const double* __restrict__ J, const double* __restrict__ c, double* __restrict__ Ah, double* __restrict__ rho, int64_t n, int64_t ldn) {
    ASM_BARGS(abt, J, c, Ah, rho, n, ldn);
    __shared__ double red[4];
    __shared__ double s_inv;
    int64_t i = blockIdx.x;
    const double* row = J + i * ldn;
    double mx = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 256) mx = fmax(mx, fabs(row[j] * c[j]));
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m4 = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        double r = pow2_round_dev(m4);
        rho[i] = r;
        s_inv = 1.0 / r;
    }
    __syncthreads();
    double inv = s_inv;
    double* out = Ah + i * ldn;
    for (int64_t j = threadIdx.x; j < n; j += 256) out[j] = row[j] * c[j] * inv;
}

// out[i] = sum_j A[i,j] x[j]   (one wavefront per row, 16-byte loads along the row)
__global__ __launch_bounds__(256) void k_gemv_n(AsmBt abt, const double* __restrict__ A, int64_t ld, const double* __restrict__ x, double* __restrict__ out, int64_t M, int64_t ncols) {
    ASM_BARGS(abt, A, ld, x, out, M, ncols);
    int64_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    int lane = threadIdx.x & 63;
    const double2* row = reinterpret_cast<const double2*>(A + i * ld);
    const double2* xv = reinterpret_cast<const double2*>(x);
    double acc = 0.0;
    int64_t n2 = ncols >> 1;                     // ncols is padded to an even count
    for (int64_t j = lane; j < n2; j += 64) {
        double2 a = row[j];
        double2 b = xv[j];
        acc = fma(a.x, b.x, acc);
        acc = fma(a.y, b.y, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[i] = acc;
}

// the same product for FEW LONG rows (the k x n basis of the null-space form: 519 rows of 11 192): one workgroup per row, so that four times
// as many wavefronts keep loads in flight (one wavefront per row: 28 us = 1.6 TB/s for 46 MB)
__global__ __launch_bounds__(256) void k_gemv_n_wide(AsmBt abt, const double* __restrict__ A, int64_t ld, const double* __restrict__ x, double* __restrict__ out, int64_t M, int64_t ncols) {
    ASM_BARGS(abt, A, ld, x, out, M, ncols);
    __shared__ double red[4];
    const int64_t i = blockIdx.x;
    const double2* row = reinterpret_cast<const double2*>(A + i * ld);
    const double2* xv = reinterpret_cast<const double2*>(x);
    double acc = 0.0;
    const int64_t n2 = ncols >> 1;
    for (int64_t j = threadIdx.x; j < n2; j += 256) {
        double2 a = row[j];
        double2 b = xv[j];
        acc = fma(a.x, b.x, acc);
        acc = fma(a.y, b.y, acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[i] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[i] = sum_{j < ncols} A[i,j] x[j] with ncols exact (entries of x beyond ncols are not touched): one wavefront per row.  The transposed
// product A'y of a DENSE matrix through its transposed copy - one launch of 5 us instead of the two-stage column reduction (13 us at 500 x 1000).
__global__ __launch_bounds__(256) void k_gemv_n_exact(AsmBt abt, const double* __restrict__ A, int64_t ld, const double* __restrict__ x, double* __restrict__ out, int64_t rows, int64_t ncols) {
    ASM_BARGS(abt, A, ld, x, out, rows, ncols);
    int64_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const int lane = threadIdx.x & 63;
    const double* row = A + i * ld;
    double acc = 0.0;
    for (int64_t j = lane; j < ncols; j += 64) acc = fma(row[j], x[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[i] = acc;
}

// partial[r][j] = sum_{i in row chunk r} A[i,j] y[i]   ; second stage sums the chunks in order (deterministic).
#define ASM_TMAXCHUNKS 128
__global__ __launch_bounds__(256) void k_gemv_t_stage1(AsmBt abt, const double* __restrict__ A, int64_t ld, const double* __restrict__ y, double* __restrict__ partial, int64_t M, int64_t ncols, int64_t chunk) {
    ASM_BARGS(abt, A, ld, y, partial, M, ncols, chunk);
    int64_t j = blockIdx.x * 256 + threadIdx.x;
    int64_t r = blockIdx.y;
    int64_t i0 = r * chunk, i1 = i0 + chunk;
    if (i1 > M) i1 = M;
    if (j >= ncols) return;
    double acc = 0.0;
    for (int64_t i = i0; i < i1; ++i) acc = fma(A[i * ld + j], y[i], acc);
    partial[r * ncols + j] = acc;
}
__global__ __launch_bounds__(256) void k_gemv_t_stage2(AsmBt abt, const double* __restrict__ partial, double* __restrict__ out, int64_t R, int64_t ncols) {
    ASM_BARGS(abt, partial, out, R, ncols);
    int64_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    double acc = 0.0;
    for (int64_t r = 0; r < R; ++r) acc += partial[r * ncols + j];
    out[j] = acc;
}

// Lower-triangular tile pair (bi >= bj) of workgroup blockIdx.x, XCD-aware.  Workgroups g, g+8, g+16, ... share an XCD (and
// its L2), so the NT = ntr (ntr + 1) / 2 tile pairs are laid along a curve that walks 8 x 8 super-tiles (row-major inside,
// super-rows top to bottom) and the XCD with label g % 8 works through one contiguous eighth of it.  The workgroups an XCD
// holds at a time then cover about one super-tile: 8 + 8 operand row blocks for 64 tile pairs instead of ~4 + 64 with a
// row-major enumeration - the operand re-reads that go beyond L2 drop ~4x (measured: fetch per launch -32 %).  Pure
// placement: any dispatch order is correct.  Grid = 8 * ceil(NT / 8) workgroups (512 * ceil(NT / 512) when dealt); false = no tile
// for this workgroup.
// deal = false: XCD label x takes the contiguous eighth [x per, (x + 1) per) of the curve (equal work per tile: the Cholesky updates);
// deal = true: runs of 64 curve positions are dealt round-robin to the XCD labels (the sparse Schur build, where the chunk lists make
// the work per tile follow the sparsity pattern - contiguous eighths put all the dense tiles on a few XCDs: measured 1.9 -> 3.1 ms).
__device__ __forceinline__ bool tri_tile_xcd(int ntr, int& bi, int& bj, bool deal = false) {
    const int64_t NT = (int64_t)ntr * (ntr + 1) / 2, per = (NT + 7) / 8;
    const int64_t l = blockIdx.x >> 3;
    const int64_t p = deal ? ((l >> 6) * 8 + (blockIdx.x & 7)) * 64 + (l & 63) : (int64_t)(blockIdx.x & 7) * per + l;
    if ((!deal && l >= per) || p >= NT) return false;
    const int nst = (ntr + 7) / 8, rlast = ntr - 8 * (nst - 1);
    int I = (int)((sqrt(16.0 + 128.0 * (double)p) - 4.0) * (1.0 / 64.0));          // 32 I^2 + 4 I tile pairs precede super-row I
    if (I > nst - 1) I = nst - 1;
    while (I < nst - 1 && 32ll * (I + 1) * (I + 1) + 4ll * (I + 1) <= p) ++I;
    while (I > 0 && 32ll * I * I + 4ll * I > p) --I;
    const int rem = (int)(p - (32ll * I * I + 4ll * I));
    const int R = (I == nst - 1) ? rlast : 8;                                       // tile rows of this super-row
    int J, ti, tj;
    if (rem < I * R * 8) {
        J = rem / (R * 8);
        const int wv = rem - J * R * 8;
        ti = wv >> 3;
        tj = wv & 7;
    } else {
        J = I;
        const int wv = rem - I * R * 8;
        ti = (int)((sqrt(8.0 * (double)wv + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= wv) ++ti;
        while (ti * (ti + 1) / 2 > wv) --ti;
        tj = wv - ti * (ti + 1) / 2;
    }
    bi = 8 * I + ti;
    bj = 8 * J + tj;
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Symmetric rank-K kernel on the f64 matrix cores.
//   mode 0:  S[a,b]  = sum_k A[row(a),k] theta[k] A[row(b),k]  (+ diag[a] if a==b)      a >= b
//   mode 1:  S[a,b] -= sum_k A[row(a),k] A[row(b),k]                                    a >= b
// row(a) = idx[a] when idx != nullptr, else row0 + a.  Tile = 32*T x 32*T per 256-thread workgroup;
// the four wavefronts form a 2x2 grid and each owns T x T MFMA tiles of 16x16.
// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md:161): lane l holds A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15]; result register r of lane l is D[row=(l>>4)+4r][col=l&15].
// KC = k-chunk staged through LDS (32: 139 KB of LDS, one workgroup per CU; 16: 74 KB and <= 128 VGPRs, two per CU -
// used for the Cholesky updates so that the latency-bound panel kernels of the look-ahead stream can share the CUs).
template <int T, int NW, int KC, int WPE>
__global__ __launch_bounds__(64 * NW, WPE) void k_syrk(AsmBt abt, const double* __restrict__ A, int64_t ld, const int* __restrict__ idx, int64_t row0, int Ms, int K, const double* __restrict__ theta, const double* __restrict__ diag, double* __restrict__ S, int64_t ldS, int64_t srow0, int mode, int MsB, int ntj, const unsigned char* __restrict__ nzflags, int nzpitch, int64_t ksplit) {
    ASM_BARGS(abt, A, ld, idx, row0, Ms, K, theta, diag, S, ldS, srow0, mode, MsB, ntj, nzflags, nzpitch, ksplit);
    constexpr int TS = 32 * T;
    int slice_tile = -1;                      // >= 0: split-K with slice-major placement - the tile pair of this workgroup by plain triangular enumeration
    if (gridDim.y > 1) {
        // split-K (small matrices with a long k range: the k x k Newton matrix of the null-space form): one slice of the
        // chunks, summed into its own copy of S (`ksplit` doubles apart); the caller adds the copies in a fixed order.
        // Placement: workgroups L, L + 8, ... share an XCD and its L2.  With a multiple of eight slices the workgroups of ONE slice are put
        // on ONE XCD (slice = label + 8 * ...): the XCD streams that slice of the operand once for all its tile pairs instead of every
        // XCD streaming the row blocks of its tile pairs over the whole k range (pure placement: any dispatch order is correct).
        int sl = (int)blockIdx.y;
        if (ntj == 0 && !nzflags && (gridDim.y & 7u) == 0u) {
            const unsigned Lin = blockIdx.x + blockIdx.y * gridDim.x, q = Lin >> 3;
            sl = (int)((Lin & 7u) + 8u * (q / gridDim.x));
            slice_tile = (int)(q % gridDim.x);
        }
        const int per = ((K / KC + (int)gridDim.y - 1) / (int)gridDim.y) * KC;
        const int k0s = sl * per;
        A += k0s;
        if (theta) theta += k0s;
        K = max(0, min(per, K - k0s));
        S += (int64_t)sl * ksplit;
    }
    // two LDS stages: the global loads of chunk c+1 are issued before the MFMAs of chunk c and written to the
    // other stage afterwards, so HBM/L2 latency hides under 16*T*T/4 matrix instructions; one barrier per chunk.
    __shared__ __attribute__((aligned(16))) double As[2][TS * (KC + 2)];
    __shared__ __attribute__((aligned(16))) double Bs[2][TS * (KC + 2)];
    int bi, bj;
    if (ntj > 0) {
        // rectangular enumeration (Cholesky in-panel update): all row tiles x the first ntj column tiles
        bi = blockIdx.x / ntj;
        bj = blockIdx.x - bi * ntj;
        if (bj > bi) return;
    } else if (slice_tile >= 0) {
        const int ntr = (Ms + TS - 1) / TS;
        if (slice_tile >= ntr * (ntr + 1) / 2) return;
        bi = (int)((sqrt(8.0 * (double)slice_tile + 1.0) - 1.0) * 0.5);
        while ((bi + 1) * (bi + 2) / 2 <= slice_tile) ++bi;
        while (bi * (bi + 1) / 2 > slice_tile) --bi;
        bj = slice_tile - bi * (bi + 1) / 2;
    } else {
        if (!tri_tile_xcd((Ms + TS - 1) / TS, bi, bj, nzflags != nullptr)) return;
    }

    // NW wavefronts as a 2 x (NW/2) grid; each owns TI x TJ MFMA tiles (NW = 8: two wavefronts per SIMD, so one
    // can issue matrix instructions while the other waits at the barrier or on LDS)
    constexpr int WCOLS = NW / 2;
    constexpr int TI = TS / 32, TJ = TS / (16 * WCOLS);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w / WCOLS, wc = w % WCOLS;
    // mode 1 (S -= A B'): the accumulators start from the S tile and the A operand is negated when staged, so the
    // read of S overlaps the first operand loads instead of trailing the last MFMA
    v4f64 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
            if (mode != 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int row = bi * TS + wr * 16 * TI + i * 16 + (lane >> 4) + 4 * r;
                    int col = bj * TS + wc * 16 * TJ + j * 16 + (lane & 15);
                    const int rc = min(row, Ms - 1), cc = min(col, rc);                // unconditional load from a clamped address
                    const double v = S[(srow0 + rc) * ldS + (srow0 + cc)];
                    acc[i][j][r] = ((row < Ms) & (col < MsB) & (col <= row)) ? v : 0.0;
                }
            }
        }
    const double asign = mode != 0 ? -1.0 : 1.0;

    // staging assignment: TS rows x KC doubles per operand; each thread moves 2 doubles (16 B) per pass
    constexpr int PER_ROW = KC / 2;                 // threads per row
    constexpr int ROWS_PER_PASS = 64 * NW / PER_ROW;
    constexpr int PASSES = TS / ROWS_PER_PASS;          // = 2T
    const int lr = tid / PER_ROW, lk = (tid % PER_ROW) * 2;
    const double* arow[PASSES];
    const double* brow[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        int r = ps * ROWS_PER_PASS + lr;
        int ga = bi * TS + r, gb = bj * TS + r;
        arow[ps] = nullptr;
        brow[ps] = nullptr;
        if (ga < Ms) arow[ps] = A + (idx ? (int64_t)idx[ga] : row0 + ga) * ld;
        if (gb < MsB) brow[ps] = A + (idx ? (int64_t)idx[gb] : row0 + gb) * ld;
    }
    // gload only issues the loads; the scaling (sign of A, theta on B) happens in lstore, i.e. after the MFMAs of the current
    // chunk - a multiply right behind the load would put the s_waitcnt vmcnt(0) in front of them and expose the load latency
    double2 ra[PASSES], rb[PASSES], rth;
    auto gload = [&](int k0) {
        rth = theta ? *reinterpret_cast<const double2*>(theta + k0 + lk) : make_double2(1.0, 1.0);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            ra[ps] = arow[ps] ? *reinterpret_cast<const double2*>(arow[ps] + k0 + lk) : make_double2(0.0, 0.0);
            rb[ps] = brow[ps] ? *reinterpret_cast<const double2*>(brow[ps] + k0 + lk) : make_double2(0.0, 0.0);
        }
    };
    auto lstore = [&](int st) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            int r = ps * ROWS_PER_PASS + lr;
            *reinterpret_cast<double2*>(&As[st][r * (KC + 2) + lk]) = make_double2(asign * ra[ps].x, asign * ra[ps].y);
            *reinterpret_cast<double2*>(&Bs[st][r * (KC + 2) + lk]) = make_double2(rb[ps].x * rth.x, rb[ps].y * rth.y);
        }
    };
    // k-chunks to visit: all of them, or (Schur build of a sparse Jacobian) only those where both operand tiles
    // hold a non-zero - the skipped products are exact zeros, and the list keeps increasing k order, so the
    // result is bitwise the one of the dense sweep.
    __shared__ int s_list[KC == 32 ? ASM_MAXCHUNKS : 1];        // chunk lists exist at the 32-column granularity of the flags only
    __shared__ int s_cnt;
    int nchunks = K / KC;
    if (nzflags) {
        if (w == 0) {
            const unsigned char* fa = nzflags + (int64_t)bi * nzpitch;
            const unsigned char* fb = nzflags + (int64_t)bj * nzpitch;
            int cnt = 0;
            for (int base = 0; base < nchunks; base += 64) {
                int c = base + lane;
                bool f = c < nchunks && (fa[c] & fb[c]);
                unsigned long long m = __ballot(f);
                if (f) s_list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = c;
                cnt += __popcll(m);
            }
            if (lane == 0) s_cnt = cnt;
        }
        __syncthreads();
        nchunks = s_cnt;
    }
    auto chunk_k0 = [&](int i) { return (nzflags ? s_list[i] : i) * KC; };
    if (nchunks > 0) {
        gload(chunk_k0(0));
        lstore(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int st = c & 1;
        if (c + 1 < nchunks) gload(chunk_k0(c + 1));
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            double af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = As[st][(wr * 16 * TI + i * 16 + (lane & 15)) * (KC + 2) + kk + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = Bs[st][(wc * 16 * TJ + j * 16 + (lane & 15)) * (KC + 2) + kk + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < nchunks) lstore(st ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = bi * TS + wr * 16 * TI + i * 16 + (lane >> 4) + 4 * r;
                int col = bj * TS + wc * 16 * TJ + j * 16 + (lane & 15);
                if (row < Ms && col < MsB && col <= row) {
                    double* dst = S + (srow0 + row) * ldS + (srow0 + col);
                    double v = acc[i][j][r];
                    if (mode == 0 && row == col && diag) v += diag[row];
                    *dst = v;
                }
            }
}

// ---------------------------------------------------------------------------------------------------
// The Cholesky update  S[a,b] -= sum_k P[a,k] P[b,k]  (a >= b; trailing and in-panel updates: nearly all of a factorisation's
// flops), written for the f64 matrix pipe rather than shared with the Schur build.  The loop never waits on anything it has not
// had a chunk's time to get:
//   * tile 128 x 128, FOUR wavefronts as 2 x 2, each owning 64 x 64 = 4 x 4 MFMA tiles (128 accumulator registers): 16 matrix
//     instructions per k-step for 8 fragment loads per two k-steps; two workgroups per CU (256 registers per lane), so one
//     workgroup's prologue / epilogue (the S tile) runs under the other's matrix instructions.
//   * operands go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write, no VALU work): one
//     wave-instruction writes 1 KiB = 16 rows x 8 doubles, lane-linear, so the slot swizzle is applied on the SOURCE address.
//   * LDS image of a chunk of 8 k: row = 8 doubles = 4 slots of 16 B, slot q of row r at q ^ (2 * bit3(r)).  The k index is permuted
//     inside the chunk: lane (row i, quarter kq) of an MFMA operand owns the contiguous doubles k = 2 kq, 2 kq + 1 and uses
//     element s in step s (same permutation on both operands), so the fragment for a chunk's two k-steps is ONE ds_read_b128 and
//     the 16-lane groups the LDS serves per cycle hit 16 distinct slots (MI355X_MICROARCH.md, LDS table).
//   * FOUR LDS stages: in iteration c the loads of chunk c+3 are issued between the matrix instructions (one per four MFMAs),
//     the fragments of chunk c+1 are read while the matrix pipe works on chunk c, and the wait at the end of the iteration is
//     for chunk c+2 only (vmcnt(4): chunk c+3 stays in flight across the barrier), reached with the next fragments in registers.
//   * accumulators start at -S and the result is stored negated: S - P P' with no multiply in the loop, bitwise the same sum.
// Rows past Ms are clamped for the loads (their products land in accumulator entries that are never stored).
#define ASM_UPD_TS 128
#define ASM_UPD_KC 8
#ifdef ASM_UPD_PROF
// diagnostic build only (-DASM_UPD_PROF): cycle sums of wavefront 0 of every workgroup, read by asm_debug_upd_prof
__device__ unsigned long long g_upd_prof[8];
#define UPD_STAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define UPD_STAMP(v) do { } while (0)
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// one LDS-DMA load: 16 B per lane to LDS address lds_dst + 16 * lane (M0 = destination; cdna_hip_programming.md, LDS-DMA recipe).
// Issued from an asm statement, so the compiler does not count it: it puts no vmcnt(0) between such a load and fragment reads of
// the other stages, and the kernel waits for them itself.  No compiler-counted global load may be outstanding when one is issued.
__device__ __forceinline__ void glds16(const double* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_syrk_upd(AsmBt abt, const double* __restrict__ A, int64_t ld, int64_t row0, int Ms, int K, double* __restrict__ S, int64_t ldS, int64_t srow0, int MsB, int ntj) {
    ASM_BARGS(abt, A, ld, row0, Ms, K, S, ldS, srow0, MsB, ntj);
    constexpr int TS = ASM_UPD_TS, KC = ASM_UPD_KC;
    constexpr int STG = 2 * TS * KC;                                           // doubles per stage: A image (8 KB), then B image (8 KB)
    __shared__ __attribute__((aligned(1024))) double lds[4 * STG];            // 64 KB
    int bi, bj;
    if (ntj > 0) {
        bi = blockIdx.x / ntj;
        bj = blockIdx.x - bi * ntj;
        if (bj > bi) return;
    } else {
        if (!tri_tile_xcd((Ms + TS - 1) / TS, bi, bj)) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    // Claim the full 256-register half of the SIMD's file (the kernel needs 244 -> 248 allocated): with 2 x 256 the slot a retiring
    // wavefront frees is one contiguous block of 256, which fits a wavefront of the look-ahead panel kernel (<= 256) as well.
    asm volatile("" ::: "v255");
#ifdef ASM_UPD_PROF
    unsigned long long tb0 = 0, tb1 = 0, tb2 = 0;
    UPD_STAMP(tb0);
#endif

    // ---- LDS-DMA sources: 16 wave-instructions fill a stage (16 rows x 8 doubles each: 8 for the A image, 8 for the B image);
    // wavefront w issues t = 4 j + w, j = 0..3.  lane -> (row t * 16 + lane / 4, physical slot lane % 4); bit3(row) = bit5(lane)
    const int lslot = (lane & 3) ^ (2 * ((lane >> 5) & 1));             // the logical slot this lane fetches
    const double* src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = 4 * j + w;
        const int g = j < 2 ? bi * TS + t * 16 + (lane >> 2) : bj * TS + (t - 8) * 16 + (lane >> 2);
        src[j] = A + (row0 + min(g, Ms - 1)) * ld + 2 * lslot;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)&lds[w * 128]);
    auto dma = [&](int j, int c) { glds16(src[j] + c * KC, lds0 + (unsigned)((c & 3) * STG * 8 + j * 4 * 1024)); };      // instruction t lands at t KiB
    const int nchunks = K / KC;                                         // even (K is a multiple of 64)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
        if (cc < nchunks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma(j, cc);
        }

    // ---- accumulators = -S tile (read while the first chunks are in flight).  Unconditional loads from clamped (always valid)
    // addresses, selected afterwards: a load under a branch would be waited for before the next one is issued.
    v4f64 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bi * TS + wr * 64 + i * 16 + (lane >> 4) + 4 * r;
                const int col = bj * TS + wc * 64 + j * 16 + (lane & 15);
                const int rr = min(row, Ms - 1), cc = min(col, rr);
                const double v = S[(srow0 + rr) * ldS + (srow0 + cc)];
                acc[i][j][r] = ((row < Ms) & (col < MsB) & (col <= row)) ? -v : 0.0;
            }

    // ---- fragment addresses: lane (row i = lane & 15, quarter kq = lane >> 4) reads slot kq ^ (2 * bit3(i))
    const int fslot = (lane >> 4) ^ (2 * ((lane >> 3) & 1));
    const int foA = (wr * 64 + (lane & 15)) * KC + 2 * fslot, foB = TS * KC + (wc * 64 + (lane & 15)) * KC + 2 * fslot;
    double2 fa0[4], fb0[4], fa1[4], fb1[4];
    auto frags = [&](double2* fa, double2* fb, int c) {
        const double* stage = lds + (c & 3) * STG;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const double2*>(stage + foA + i * 16 * KC);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const double2*>(stage + foB + j * 16 * KC);
    };
    // one iteration: matrix instructions of chunk c on (xa, xb); fragments of chunk c+1 into (ya, yb); loads of chunk c+3
    auto body = [&](int c, const double2* xa, const double2* xb, double2* ya, double2* yb) {
        const bool pf = c + 3 < nchunks;
        frags(ya, yb, c + 1);           // unconditional (after the last chunk: unused values from a quiescent stage): a branch here costs the counted waits
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[i].x, xb[j].x, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (pf) dma(i, c + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[i].y, xb[j].y, acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // chunk c+2 has landed (this wave's part; the barrier covers the rest); chunk c+3's four loads may stay in flight
        if (pf) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    UPD_STAMP(tb1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    frags(fa0, fb0, 0);
#ifdef ASM_UPD_PROF
    unsigned long long tk0 = 0, tk1 = 0;
    UPD_STAMP(tk0);
#endif
    for (int c = 0; c < nchunks; c += 2) {
        body(c, fa0, fb0, fa1, fb1);
        body(c + 1, fa1, fb1, fa0, fb0);
    }
#ifdef ASM_UPD_PROF
    UPD_STAMP(tk1);
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bi * TS + wr * 64 + i * 16 + (lane >> 4) + 4 * r;
                const int col = bj * TS + wc * 64 + j * 16 + (lane & 15);
                if (row < Ms && col < MsB && col <= row) S[(srow0 + row) * ldS + (srow0 + col)] = -acc[i][j][r];
            }
#ifdef ASM_UPD_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    UPD_STAMP(tb2);
    if (tid == 0) {
        atomicAdd(&g_upd_prof[0], tk0 - tb0); atomicAdd(&g_upd_prof[1], tb2 - tk1); atomicAdd(&g_upd_prof[2], tb1 - tb0);
        atomicAdd(&g_upd_prof[5], tk1 - tk0); atomicAdd(&g_upd_prof[6], (unsigned long long)nchunks); atomicAdd(&g_upd_prof[7], 1ull);
    }
#endif
}

// diag0[i] = S_ii ; then S_ii += reg.   mode 0: reg_i = rel*S_ii + absv ;  mode 1: reg = rel*max(max_i S_ii, 1e-300)
__global__ __launch_bounds__(1024) void k_diag_prepare(AsmBt abt, double* __restrict__ S, int64_t ldS, int Ms, double* __restrict__ diag0, int mode, double rel, double absv) {
    ASM_BARGS(abt, S, ldS, Ms, diag0, mode, rel, absv);
    __shared__ double red[16];
    __shared__ double s_max;
    double mx = 0.0;
    for (int i = threadIdx.x; i < Ms; i += 1024) {
        double d = S[(int64_t)i * ldS + i];
        diag0[i] = d;
        mx = fmax(mx, d);
    }
    if (mode == 1) {
        mx = wave_max(mx);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            double m = 0.0;
            for (int k = 0; k < 16; ++k) m = fmax(m, red[k]);
            s_max = fmax(m, 1e-300);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < Ms; i += 1024) {
        double d = diag0[i];
        double reg = (mode == 0) ? (rel * d + absv) : (rel * s_max);
        S[(int64_t)i * ldS + i] = d + reg;
    }
}

// ---------------------------------------------------------------------------------------------------
// Cholesky of one NB x NB diagonal block, staged in LDS.  Pivot guard: d <= 1e-14*diag0 -> d := 1e256.
#define ASM_DP (ASM_NB + 1)
// Cholesky of one 64x64 diagonal block + explicit inverse of the factor, on the critical path of every panel step.
// 256 threads, everything in LDS, blocked by 16: (1) one wavefront factors the 16x16 sub-block in registers (row per
// lane, wave shuffles), (2) substitution for the rows below, (3) rank-16 update of the remaining triangle; then the
// inverse by 16x16 blocks: diagonal blocks by substitution (one wavefront each), off-diagonal blocks level by level
// W_ij = -W_ii * sum_k L_ik W_kj.  Static pivot guard: d <= thr*diag0 -> 1e256 (row dropped, see asm_hip.hip).
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// body shared by the stand-alone kernel and the fused panel kernel: D, W are ASM_NB x ASM_DP LDS buffers, T 4 x (16 x 17)
typedef double potrf_T_t[16 * 17];
template <bool SC1, bool OPQ = false, bool PADSKIP = false>
__device__ __forceinline__ void potrf64_body(double* __restrict__ D, double* __restrict__ W, potrf_T_t* __restrict__ T, double* __restrict__ d0,
                                             double* __restrict__ dinv, double* __restrict__ S, int64_t ldS, int k0, int nb,
                                             const double* __restrict__ diag0, double thr, double* __restrict__ Linv) {
    // OPQ: the thread index is made opaque per call.  Inside the panel kernels this body sits in the step loop, and what the compiler derives
    // from a loop-invariant index (address offsets, the identity pattern of the tile fill) is hoisted out of that loop and - at the 256-register
    // cap of k_chol_panel_band - spilled; the reloads sat on the chain of every step (eight dependent scratch round trips, 22 k cycles for the
    // tile fill instead of 4 k).  Kernels with the full register file keep the hoisting: every recomputed instruction costs the chain's lone
    // wavefront ~8 cycles.
    int tid_ = threadIdx.x;
    if (OPQ) asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63, wv = tid >> 6;
#ifdef ASM_POTRF_PROF
    long long stamp_[16]; int ns_ = 0;
#define PSTAMP() do { if (tid == 0) stamp_[ns_++] = clock64(); } while (0)
#else
#define PSTAMP() do {} while (0)
#endif
    PSTAMP();
    {
        // Tile loads here and below: (1) unconditional loads from clamped (always valid) addresses, selected afterwards with '&' rather
        // than '&&' - a load under a branch (or one the compiler can sink into one) is waited for before the next is issued;
        // (2) all sixteen loads of a thread first, the LDS stores after a scheduling fence - one round trip instead of sixteen.
        double tv[ASM_NB * ASM_NB / 256];
#pragma unroll
        for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
            const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
            const int rc = min(rr, nb - 1), cc = min(c, rc);
            tv[it] = S[(int64_t)(k0 + rc) * ldS + k0 + cc];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
            const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
            D[rr * ASM_DP + c] = ((rr < nb) & (c <= rr)) ? tv[it] : (rr == c ? 1.0 : 0.0);
            W[rr * ASM_DP + c] = 0.0;
        }
    }
    if (tid < ASM_NB) d0[tid] = tid < nb ? diag0[k0 + tid] : 1.0;
    __syncthreads();
    PSTAMP();
    // inverse of one finished 16x16 diagonal block of L (columns of the inverse by substitution; lane t < 16 owns column t)
    auto diag_inverse = [&](int c0) {
        const int t = lane & 15;
        double x[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            double sum = 0.0;
#pragma unroll
            for (int q = 0; q < rr; ++q) sum = fma(D[(c0 + rr) * ASM_DP + c0 + q], x[q], sum);
            double rhs = (rr == t) ? 1.0 : 0.0;
            x[rr] = (rr < t) ? 0.0 : (rhs - sum) * dinv[c0 + rr];
        }
        if (lane < 16) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) W[(c0 + rr) * ASM_DP + c0 + t] = x[rr];
        }
    };
    // sub-blocks past the nb live columns are identity padding (k = 519: the ninth diagonal block has 7 columns): no pivot chain, no update,
    // identity inverse - the block costs what its live sub-blocks cost, not a full 64-column chain
    auto identity_inverse = [&](int c0) { if (lane < 16) W[(c0 + lane) * ASM_DP + c0 + lane] = 1.0; };
    for (int sb = 0; sb < 4; ++sb) {
        const int c0 = sb * 16;
        if (PADSKIP && c0 >= nb) {
            if (wv == 0 && lane < 16) dinv[c0 + lane] = 1.0;
            if (wv == 1 && sb > 0) { if (c0 - 16 < nb) diag_inverse(c0 - 16); else identity_inverse(c0 - 16); }
            __syncthreads();
            continue;
        }
        if (wv == 0) {
            // (1) 16 columns of the factor over ALL remaining rows at once: lane r owns row c0 + r.  Lanes 0..15 hold the
            //     diagonal sub-block; the same sixteen broadcast-and-update steps that factor it solve the rows below it
            //     (x L11' = a is exactly the recurrence a[c] -= l * L11[c][j]), so the panel solve costs nothing extra.
            const int row = c0 + lane;
            const bool live = row < ASM_NB;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = live ? D[row * ASM_DP + c0 + c] : 0.0;
            double my_inv = 1.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double d = readlane_f64(a[j], j);
                if (!(d > thr * d0[c0 + j])) d = 1e256;
                // 1/sqrt(d): hardware estimate (v_rsq_f64) + two Newton steps - the shortest dependent chain that is accurate to the
                // last bits (the library rsqrt() adds range handling this pivot never needs: d is a guarded positive number)
                double inv = __builtin_amdgcn_rsq(d);
                inv = inv * fma(-0.5 * d * inv, inv, 1.5);
                inv = inv * fma(-0.5 * d * inv, inv, 1.5);
                double l = a[j] * inv;
                if (lane == j) { l = d * inv; my_inv = inv; }
                a[j] = l;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) a[c] = fma(-l, readlane_f64(l, c), a[c]);
            }
            if (live) {
#pragma unroll
                for (int c = 0; c < 16; ++c) D[row * ASM_DP + c0 + c] = (lane >= 16 || c <= lane) ? a[c] : 0.0;
                if (lane < 16) dinv[c0 + lane] = my_inv;
            }
        } else if (wv == 1 && sb > 0) {
            diag_inverse(c0 - 16);                       // meanwhile: inverse of the previous (finished) diagonal sub-block
        }
        __syncthreads();
        PSTAMP();
        {                                                // (2) rank-16 update of the remaining lower triangle, 16x16 tiles on
            const int rb = c0 + 16;                      //     the matrix cores: A22[ti][tj] -= X[ti] X[tj]'  (tj <= ti)
            const int nt = (ASM_NB - rb) / 16;
            for (int t = wv; t < nt * (nt + 1) / 2; t += 4) {
                int ti = 0;
                while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
                const int tj = t - ti * (ti + 1) / 2;
                v4f64 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = D[(rb + 16 * ti + (lane >> 4) + 4 * r) * ASM_DP + rb + 16 * tj + (lane & 15)];
#pragma unroll
                for (int kk = 0; kk < 16; kk += 4) {
                    double af = -D[(rb + 16 * ti + (lane & 15)) * ASM_DP + c0 + kk + (lane >> 4)];
                    double bf = D[(rb + 16 * tj + (lane & 15)) * ASM_DP + c0 + kk + (lane >> 4)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) D[(rb + 16 * ti + (lane >> 4) + 4 * r) * ASM_DP + rb + 16 * tj + (lane & 15)] = acc[r];
            }
        }
        __syncthreads();
    }
    // ---- the factor goes back to S (wavefronts 0, 2, 3) while wavefront 1 inverts the last diagonal sub-block; then the inverse
    // W = L^-1 by 16x16 blocks (diagonal blocks 0..2 were inverted beside the factorisation of their successors)
    if (wv == 1) {
        if (!PADSKIP || 48 < nb) diag_inverse(48); else identity_inverse(48);
    } else {
        const int t3 = (wv == 0 ? 0 : wv - 1) * 64 + lane;          // 0..191
        for (int e = t3; e < ASM_NB * ASM_NB; e += 192) {
            int rr = e >> 6, c = e & 63;
            if (rr < nb && c <= rr) S[(int64_t)(k0 + rr) * ldS + k0 + c] = D[rr * ASM_DP + c];
        }
    }
    PSTAMP();
    __syncthreads();
    PSTAMP();
    for (int dlev = 1; dlev < 4; ++dlev) {               // off-diagonal blocks (i, j = i - dlev), one wavefront per block,
        const int i = wv + dlev, j = wv;                 // 16x16x16 products on the matrix cores
        if (PADSKIP && dlev * 16 >= nb) break;                      // (uniform) every block of this and the later levels lies in the padding: zero
        if (i < 4) {
            // T = sum_{k=j}^{i-1} L_ik W_kj
            v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; ++k)
#pragma unroll
                for (int kk = 0; kk < 16; kk += 4) {
                    double af = D[(i * 16 + (lane & 15)) * ASM_DP + k * 16 + kk + (lane >> 4)];
                    double bf = W[(k * 16 + kk + (lane >> 4)) * ASM_DP + j * 16 + (lane & 15)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) T[wv][((lane >> 4) + 4 * r) * 17 + (lane & 15)] = acc[r];
        }
        __syncthreads();
        if (i < 4) {
            // W_ij = -W_ii T
            v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; kk += 4) {
                double af = -W[(i * 16 + (lane & 15)) * ASM_DP + i * 16 + kk + (lane >> 4)];
                double bf = T[wv][(kk + (lane >> 4)) * 17 + (lane & 15)];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) W[(i * 16 + (lane >> 4) + 4 * r) * ASM_DP + j * 16 + (lane & 15)] = acc[r];
        }
        __syncthreads();
    }
    PSTAMP();
    double* out = Linv + (int64_t)(k0 / ASM_NB) * ASM_NB * ASM_NB;
#ifdef ASM_POTRF_PROF
    if (tid == 0 && k0 == 64) { stamp_[ns_++] = clock64(); printf("potrf stamps:"); for (int q = 1; q < ns_; ++q) printf(" %lld", stamp_[q] - stamp_[q - 1]); printf("\n"); }
#endif
    _Pragma("unroll") for (int e_it = 0; e_it < ASM_NB * ASM_NB / 256; ++e_it) {      /* constant trip count, fully unrolled: all of a thread's loads in flight */
        const int e = tid + 256 * e_it;
        const double v = W[(e >> 6) * ASM_DP + (e & 63)];
        if (SC1) __hip_atomic_store(out + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through: read by other workgroups of this launch
        else out[e] = v;
    }
}
__global__ __launch_bounds__(256) void k_potrf_diag(AsmBt abt, double* __restrict__ S, int64_t ldS, int k0, int nb, const double* __restrict__ diag0, double thr, double* __restrict__ Linv) {
    ASM_BARGS(abt, S, ldS, k0, nb, diag0, thr, Linv);
    __shared__ double D[ASM_NB * ASM_DP];
    __shared__ double W[ASM_NB * ASM_DP];
    __shared__ potrf_T_t T[4];
    __shared__ double d0[ASM_NB], dinv[ASM_NB];
    potrf64_body<false>(D, W, T, d0, dinv, S, ldS, k0, nb, diag0, thr, Linv);
}

// Panel solve through the explicit inverse:  S[i, k0:k1] <- S[i, k0:k1] * Linv11'  for the 64 rows of this tile, as one
// 64x64x64 product on the matrix cores: wavefront w owns rows 16w..16w+15 (four 16x16 tiles, 16 k-steps of
// v_mfma_f64_16x16x4_f64).  X Linv' = sum_q X[r][q] Linv[c][q], so the right operand is needed transposed - which is the
// row-major Linv itself; both operands sit in LDS with pitch = 2 mod 32 doubles (conflict-free fragment reads).  Linv is
// lower triangular with explicit zeros above the diagonal, so the full product is the triangular one.
#define ASM_XP 66
__global__ __launch_bounds__(256) void k_trsm_panel(AsmBt abt, double* __restrict__ S, int64_t ldS, int k0, int nb, int Ms, const double* __restrict__ Linv) {
    ASM_BARGS(abt, S, ldS, k0, nb, Ms, Linv);
    __shared__ double Xa[ASM_NB * ASM_XP];      // the tile  X[r][q]
    __shared__ double Li[ASM_NB * ASM_XP];      // Linv[c][q]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i0 = k0 + nb + blockIdx.x * ASM_NB;
    const double* Lb = Linv + (int64_t)(k0 / ASM_NB) * ASM_NB * ASM_NB;
    _Pragma("unroll") for (int e_it = 0; e_it < ASM_NB * ASM_NB / 256; ++e_it) {      /* constant trip count, fully unrolled: all of a thread's loads in flight */
        const int e = tid + 256 * e_it;
        int rr = e >> 6, c = e & 63;
        Li[rr * ASM_XP + c] = Lb[e];
        int gi = i0 + rr;
        const double v = S[(int64_t)min(gi, Ms - 1) * ldS + k0 + min(c, nb - 1)];
        Xa[rr * ASM_XP + c] = ((gi < Ms) & (c < nb)) ? v : 0.0;
    }
    __syncthreads();
    v4f64 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < ASM_NB; kk += 4) {
        double af = Xa[(w * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            double bf = Li[(t * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = w * 16 + (lane >> 4) + 4 * r, col = t * 16 + (lane & 15);
            int gi = i0 + row;
            if (gi < Ms && col < nb) S[(int64_t)gi * ldS + k0 + col] = acc[t][r];
        }
}

// In-panel rank-64 update of the Cholesky:  S[r, c] -= P[r, :] . P[c, :]  for rows r >= k1 and columns k1 <= c < c_end
// (c <= r), P = S[:, k0:k0+64] the panel just solved.  One workgroup per 64 x 64 tile: both operand tiles are staged whole
// (pitch = 2 mod 32 doubles), 16 k-steps of v_mfma_f64_16x16x4_f64 per wavefront, the accumulators start from the S tile
// and the left operand is negated - the generic k_syrk pays its k-chunk pipeline and 128 x 128 tiles for K = 64.
__global__ __launch_bounds__(256) void k_panel_update64(AsmBt abt, double* __restrict__ S, int64_t ldS, int k0, int k1, int c_end, int Ms) {
    ASM_BARGS(abt, S, ldS, k0, k1, c_end, Ms);
    __shared__ double Pa[ASM_NB * ASM_XP];
    __shared__ double Pb[ASM_NB * ASM_XP];
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (tj > ti) return;                                   // strictly upper tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r0 = k1 + ti * ASM_NB, c0 = k1 + tj * ASM_NB;
    _Pragma("unroll") for (int e_it = 0; e_it < ASM_NB * ASM_NB / 256; ++e_it) {      /* constant trip count, fully unrolled: all of a thread's loads in flight */
        const int e = tid + 256 * e_it;
        int rr = e >> 6, c = e & 63;
        const double va = S[(int64_t)min(r0 + rr, Ms - 1) * ldS + k0 + c], vb = S[(int64_t)min(c0 + rr, Ms - 1) * ldS + k0 + c];
        Pa[rr * ASM_XP + c] = (r0 + rr < Ms) ? -va : 0.0;
        Pb[rr * ASM_XP + c] = (c0 + rr < Ms) ? vb : 0.0;
    }
    v4f64 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = r0 + w * 16 + (lane >> 4) + 4 * r, col = c0 + t * 16 + (lane & 15);
            const int rc = min(row, Ms - 1);
            const double v = S[(int64_t)rc * ldS + min(col, min(c_end - 1, rc))];
            acc[t][r] = ((row < Ms) & (col < c_end) & (col <= row)) ? v : 0.0;
        }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < ASM_NB; kk += 4) {
        double af = Pa[(w * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            double bf = Pb[(t * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = r0 + w * 16 + (lane >> 4) + 4 * r, col = c0 + t * 16 + (lane & 15);
            if (row < Ms && col < c_end && col <= row) S[(int64_t)row * ldS + col] = acc[t][r];
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused inner panel of the Cholesky: the 64-wide steps of one inner panel [I0, I1) (<= ASM_PNL_NS = 8 steps: diagonal-block
// factorisation, panel solve, rank-64 update of the panel's remaining columns) in ONE launch, as a dataflow over
// 64-row tiles instead of three dependent launches per step.  Row tile rt (rows I0 + 64 rt ...) belongs to workgroup
// rt mod G for the whole launch, so a tile's own history needs no synchronisation; what crosses workgroups is published
// with an agent-scope release and consumed behind a relaxed poll + ONE agent-scope acquire (cdna_hip_programming.md,
// Guideline 16):
//     flags[k]              the factor / inverse of diagonal block k (written by the owner of row tile k)
//     flags[NS + NS k + tj] the solved panel tile X(tj, step k) of a later diagonal row tile tj (the right operand of the
//                           rank-64 update of column tile tj)
// The owner of row tile k+1 factors diagonal block k+1 as soon as that tile has its step-k update (look-ahead): the
// critical path of a step is  panel solve + update of ONE tile + the 64 x 64 factorisation, everything else overlaps.
// Every workgroup must be able to become resident (G <= #CUs); every spin is bounded (tmo[0] != 0 -> the host fails the
// factorisation).  A flag is "set" when it holds the launch's epoch (a per-handle counter, never 0, never repeated: the panel
// launches of a handle are stream-ordered and not replayed from a graph), so the flag words need no reset between launches.
// The payload is stored write-through (sc1: __hip_atomic_store relaxed / agent scope), so publishing needs no release fence
// (which would write back every dirty line of the XCD's L2): every storing wavefront drains its stores, then one lane sets
// the flag (Guideline 16, recipe R1).
__device__ __forceinline__ void pnl_publish(unsigned* flag, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pnl_wait(unsigned* flag, unsigned epoch, unsigned* tmo) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(4);
            ++spins;
            // bounded: a lost producer must not hang the GPU.  Once one workgroup has given up the launch is void (the host reports
            // it): every other wait ends at its next look at the timeout word instead of spinning to its own bound
            if (spins > (1u << 22) || ((spins & 255u) == 0 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
// the same wait for helper workgroups (k_chol_panel_inv, k_chol_panel_band), which are not on the critical path of the factorisation
__device__ __forceinline__ void pnl_wait_helper(unsigned* flag, unsigned epoch, unsigned* tmo) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(24);          // helpers are not on the critical path of the factorisation: they poll rarely
            ++spins;
            if (spins > (1u << 20) || ((spins & 63u) == 0 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
// test hook: workgroups waiting for a flag that no producer of the launch sets (the bounded spin must end and report)
__global__ __launch_bounds__(256) void k_pnl_wait_probe(AsmBt abt, unsigned* flag, unsigned epoch, unsigned* tmo) {
    ASM_BARGS(abt, flag, epoch, tmo);
    pnl_wait(flag, epoch, tmo);
}
#define ASM_PNL_LDS (2 * ASM_NB * ASM_XP + 4 * 16 * 17 + 2 * ASM_NB)
#define ASM_PNL_NRT 40      // BAND variant: most row tiles of one launch (panel + reach of the band)
#define ASM_PNL_NS 10       // most 64-wide steps of one panel launch: eight, or up to ten when the last inner panel absorbs a short remainder (flag words: NS for the diagonal blocks + NS * NS for the panel tiles)
// Register budget: two wavefronts per SIMD = 256 registers per lane, AGPRs included.  The panel kernel runs beside k_syrk_upd
// (256 per wavefront, two per SIMD): a panel wavefront must fit into the slot ONE retiring update wavefront frees.  Left to
// itself the compiler takes 256 VGPRs + 56 AGPRs (occupancy 1 is allowed for a 256-thread kernel), such a wavefront fits
// nowhere while updates are queued and the look-ahead chain starts only after the whole trailing update (measured at
// M = 18637: first panel launch 5.2 ms instead of 1.0 ms).
// BAND: the variant for banded factors whose trailing update runs inside the launch too (k_chol_panel_band below): every solved panel tile is
// published (flag stride ASM_PNL_NRT), not only those of the panel's own diagonal rows.
// THIN: a row tile with at most 16 live rows (k = 519: the ninth has 7) keeps its products to the live 16-row block and deals the four 16-column
// blocks to the four wavefronts - a quarter of the matrix instructions per wavefront.  The owner of the LAST row tile has the most updates per
// step (one per remaining column tile); with nine or ten steps in one launch it finishes when the chain does, so its cost per update counts.
template <bool BAND, bool OPQ, bool PREF, bool THIN = false>
__device__ __forceinline__ void chol_panel_body(double* __restrict__ sm, double* __restrict__ S, int64_t ldS, int I0, int I1, int Ms,
                                                const double* __restrict__ diag0, double thr, double* __restrict__ Linv,
                                                unsigned* __restrict__ flags, unsigned* __restrict__ tmo, unsigned epoch, const int G, const int wg) {
    double* B0 = sm;                                  // the tile being solved, then X
    double* B1 = sm + ASM_NB * ASM_XP;                // block inverse, then the right operand X(tj)
    potrf_T_t* Tt = reinterpret_cast<potrf_T_t*>(sm + 2 * ASM_NB * ASM_XP);
    double* d0 = sm + 2 * ASM_NB * ASM_XP + 4 * 16 * 17;
    double* dinv = d0 + ASM_NB;
    const int nsteps = (I1 - I0 + ASM_NB - 1) / ASM_NB;
    const int nrt = (Ms - I0 + ASM_NB - 1) / ASM_NB;
#ifdef ASM_PANEL_PROF
    const int tid = threadIdx.x;
    long long qs_[12]; int qn_ = 0;
#define QSTAMP(cond) do { if (tid == 0 && (cond)) qs_[qn_++] = clock64(); } while (0)
#else
#define QSTAMP(cond) do {} while (0)
#endif
    if (wg == 0) {
        potrf64_body<true, OPQ, THIN>(B0, B1, Tt, d0, dinv, S, ldS, I0, min(ASM_NB, Ms - I0), diag0, thr, Linv);
        pnl_publish(flags + 0, epoch);
    }
    for (int k = 0; k < nsteps; ++k) {
        const int k0 = I0 + k * ASM_NB;
        const int nb = min(ASM_NB, Ms - k0), k1 = k0 + nb;
        if (k1 >= Ms) break;
        bool waited = false;
        // first row tile > k owned by this workgroup
        int rt = wg;
        while (rt <= k) rt += G;
        for (; rt < nrt; rt += G) {
            // OPQ (the register-capped kernels): the thread index is opaque per tile, as in potrf64_body - nothing derived from it is hoisted out of
            // the step loop and spilled.  The kernels with the full register file keep the hoisting (recomputing it costs the lone
            // wavefronts of the chain instructions: every one is ~8 cycles there)
            int tq_ = threadIdx.x;
            if (OPQ) asm volatile("" : "+v"(tq_));
            const int tid = tq_, lane = tid & 63, w = tid >> 6;
            QSTAMP(k == 1 && rt == 2);
            const int i0 = I0 + rt * ASM_NB;
            const double* Lb = Linv + (int64_t)(k0 / ASM_NB) * ASM_NB * ASM_NB;
            // PREF: the tile's own entries are this workgroup's history (all its earlier updates were made here): their loads - 64 rows a whole
            // matrix row apart, the slow part of the step's fetch in a banded factor - go out BEFORE the wait for the diagonal block
            double tv[ASM_NB * ASM_NB / 256];
            if (PREF) {
#pragma unroll
                for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
                    const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
                    tv[it] = S[(int64_t)min(i0 + rr, Ms - 1) * ldS + k0 + min(c, nb - 1)];
                }
            }
            if (!waited) { pnl_wait(flags + k, epoch, tmo); waited = true; }
            QSTAMP(k == 1 && rt == 2);
            // ---- panel solve of the tile: X = S[tile, k0:k1] Linv'
            __syncthreads();
            {
                double lv[ASM_NB * ASM_NB / 256];
#pragma unroll
                for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
                    const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
                    lv[it] = Lb[e];
                    if (!PREF) tv[it] = S[(int64_t)min(i0 + rr, Ms - 1) * ldS + k0 + min(c, nb - 1)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
                    const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
                    B1[rr * ASM_XP + c] = lv[it];
                    B0[rr * ASM_XP + c] = ((i0 + rr < Ms) & (c < nb)) ? tv[it] : 0.0;
                }
            }
            __syncthreads();
            QSTAMP(k == 1 && rt == 2);
            const bool thin = THIN && Ms - i0 <= 16;
            const int ws = __builtin_amdgcn_readfirstlane(w);      // (scalar: wavefront-uniform block offsets)
            v4f64 acc[4];
            if (thin) {
                // rows 0..15 of the tile x column block ws; rows 16.. of B0 stay the zeros of the fill
                v4f64 a1 = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
                for (int kk = 0; kk < ASM_NB; kk += 4) {
                    const double af = B0[(lane & 15) * ASM_XP + kk + (lane >> 4)];
                    const double bf = B1[(ws * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
                    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, a1, 0, 0, 0);
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (lane >> 4) + 4 * r, col = ws * 16 + (lane & 15), gi = i0 + row;
                    B0[row * ASM_XP + col] = a1[r];
                    if (gi < Ms && col < nb) {
                        double* dst = S + (int64_t)gi * ldS + k0 + col;
                        if (BAND || rt < nsteps) __hip_atomic_store(dst, a1[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else *dst = a1[r];
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
                for (int kk = 0; kk < ASM_NB; kk += 4) {
                    double af = B0[(w * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        double bf = B1[(t * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[t], 0, 0, 0);
                    }
                }
                __syncthreads();                          // every wavefront has read its rows of B0 and all of B1
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int row = w * 16 + (lane >> 4) + 4 * r, col = t * 16 + (lane & 15);
                        int gi = i0 + row;
                        B0[row * ASM_XP + col] = acc[t][r];
                        if (gi < Ms && col < nb) {
                            double* dst = S + (int64_t)gi * ldS + k0 + col;
                            if (BAND || rt < nsteps) __hip_atomic_store(dst, acc[t][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // operand for others
                            else *dst = acc[t][r];
                        }
                    }
            }
            QSTAMP(k == 1 && rt == 2);
            if (BAND || rt < nsteps) pnl_publish(flags + ASM_PNL_NS + (BAND ? ASM_PNL_NRT : ASM_PNL_NS) * k + rt, epoch);      // a later diagonal row tile (BAND: any row tile): its X is an operand for others
            else __syncthreads();
            QSTAMP(k == 1 && rt == 2);
            // ---- rank-64 update of the panel's remaining column tiles tj = k+1 .. min(rt, nsteps-1)
            const int tj_hi = min(rt, nsteps - 1);
            for (int tj = k + 1; tj <= tj_hi; ++tj) {
                const int c0 = I0 + tj * ASM_NB, c_end = min(c0 + ASM_NB, min(I1, Ms));
                const double* Pb = B0;
                if (tj != rt) {
                    pnl_wait(flags + ASM_PNL_NS + (BAND ? ASM_PNL_NRT : ASM_PNL_NS) * k + tj, epoch, tmo);
                    {
                        double tv[ASM_NB * ASM_NB / 256];
#pragma unroll
                        for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
                            const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
                            tv[it] = S[(int64_t)min(c0 + rr, Ms - 1) * ldS + k0 + min(c, nb - 1)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
                            const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
                            B1[rr * ASM_XP + c] = ((c0 + rr < Ms) & (c < nb)) ? tv[it] : 0.0;
                        }
                    }
                    __syncthreads();
                    Pb = B1;
                }
                if (thin) {
                    v4f64 a1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i0 + (lane >> 4) + 4 * r, col = c0 + ws * 16 + (lane & 15);
                        const int rc = min(row, Ms - 1);
                        const double v = S[(int64_t)rc * ldS + min(col, min(c_end - 1, rc))];
                        a1[r] = ((row < Ms) & (col < c_end) & (col <= row)) ? v : 0.0;
                    }
#pragma unroll 4
                    for (int kk = 0; kk < ASM_NB; kk += 4) {
                        const double af = -B0[(lane & 15) * ASM_XP + kk + (lane >> 4)];
                        const double bf = Pb[(ws * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, a1, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i0 + (lane >> 4) + 4 * r, col = c0 + ws * 16 + (lane & 15);
                        if (row < Ms && col < c_end && col <= row) S[(int64_t)row * ldS + col] = a1[r];
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int row = i0 + w * 16 + (lane >> 4) + 4 * r, col = c0 + t * 16 + (lane & 15);
                            const int rc = min(row, Ms - 1);
                            const double v = S[(int64_t)rc * ldS + min(col, min(c_end - 1, rc))];
                            acc[t][r] = ((row < Ms) & (col < c_end) & (col <= row)) ? v : 0.0;
                        }
#pragma unroll 4
                    for (int kk = 0; kk < ASM_NB; kk += 4) {
                        double af = -B0[(w * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            double bf = Pb[(t * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[t], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int row = i0 + w * 16 + (lane >> 4) + 4 * r, col = c0 + t * 16 + (lane & 15);
                            if (row < Ms && col < c_end && col <= row) S[(int64_t)row * ldS + col] = acc[t][r];
                        }
                }
                __syncthreads();                      // B1 is reloaded for the next column tile
            }
            // ---- look-ahead: this tile is the next diagonal block and has all its updates now
            if (rt == k + 1 && rt < nsteps) {
                QSTAMP(k == 1 && rt == 2);
                const int kn = I0 + rt * ASM_NB;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                QSTAMP(k == 1 && rt == 2);
                potrf64_body<true, OPQ, THIN>(B0, B1, Tt, d0, dinv, S, ldS, kn, min(ASM_NB, Ms - kn), diag0, thr, Linv);
                QSTAMP(k == 1 && rt == 2);
                pnl_publish(flags + rt, epoch);
                QSTAMP(k == 1 && rt == 2);
#ifdef ASM_PANEL_PROF
                if (tid == 0 && k == 1 && (I0 == 0 || I0 == 1024)) { printf("panel stamps (wait, load, trsm+store, publish, update, drain, potrf, publish):"); for (int q = 1; q < qn_; ++q) printf(" %lld", qs_[q] - qs_[q - 1]); printf("\n"); }
#endif
            }
        }
    }
}

// k_chol_panel: the version that runs BESIDE the trailing update (look-ahead stream), capped at 256 registers;
// k_chol_panel_solo: the same body without the cap (312 registers, no spills) for chains that have the chip to themselves.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chol_panel(AsmBt abt, double* __restrict__ S, int64_t ldS, int I0, int I1, int Ms, const double* __restrict__ diag0, double thr, double* __restrict__ Linv, unsigned* __restrict__ flags, unsigned* __restrict__ tmo, unsigned epoch) {
    ASM_BARGS(abt, S, ldS, I0, I1, Ms, diag0, thr, Linv, flags, tmo, epoch);
    __shared__ __attribute__((aligned(16))) double sm[ASM_PNL_LDS];
    __builtin_amdgcn_s_setprio(3);      // beside the trailing update: the chain's wavefronts win the issue slots of the SIMDs they share
    chol_panel_body<false, true, true>(sm, S, ldS, I0, I1, Ms, diag0, thr, Linv, flags, tmo, epoch, (int)gridDim.x, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void k_chol_panel_solo(AsmBt abt, double* __restrict__ S, int64_t ldS, int I0, int I1, int Ms, const double* __restrict__ diag0, double thr, double* __restrict__ Linv, unsigned* __restrict__ flags, unsigned* __restrict__ tmo, unsigned epoch) {
    ASM_BARGS(abt, S, ldS, I0, I1, Ms, diag0, thr, Linv, flags, tmo, epoch);
    __shared__ __attribute__((aligned(16))) double sm[ASM_PNL_LDS];
    chol_panel_body<false, false, true, true>(sm, S, ldS, I0, I1, Ms, diag0, thr, Linv, flags, tmo, epoch, (int)gridDim.x, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Banded factors (the S0 = A_EF A_EF' of the null-space form in reverse Cuthill-McKee order: 10 673 rows, band 1 398 at case1354pegase
// size): the whole step of the blocked factorisation for one 512-wide panel in ONE launch.  Everything the panel reaches lies within
// `band` rows of its last column (Ms = that limit here), i.e. ~30 row tiles, and the rank-512 update of the tiles behind the panel - two
// k_syrk launches of 60-70 us on the factorisation's chain per 1024 columns before - is a few hundred 64 x 64 tiles with eight 64^3
// products each.  Helper workgroups (behind the main ones in the grid, one per trailing tile (rt, ct), ct >= the panel's steps) subtract
// X(rt, k) X(ct, k)' as the solved tiles of step k arrive (every panel tile is published in this variant) and store the tile once.  After
// the last step's panel solve two products remain.  The helpers only wait for main workgroups; all workgroups must be resident.
#define ASM_BAND_TPH 1      // trailing tiles per helper workgroup (2 and 3 - all workgroups then fit one per CU - measured the same within 0.3 %, with 159 spilled registers)
__device__ __forceinline__ void chol_band_update_tile(double* __restrict__ sm, double* __restrict__ S, int64_t ldS, int I0, int I1, int Ms,
                                                      unsigned* __restrict__ flags, unsigned* __restrict__ tmo, unsigned epoch, int hx, int nhelp) {
    double* Pa = sm;
    double* Pb = sm + ASM_NB * ASM_XP;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nsteps = (I1 - I0 + ASM_NB - 1) / ASM_NB;
    const int nrt = (Ms - I0 + ASM_NB - 1) / ASM_NB, m = nrt - nsteps, ntiles = m * (m + 1) / 2;
    // trailing tiles in rows of a triangle: index -> (a, b), b <= a;  rt = nsteps + a, ct = nsteps + b.  This workgroup owns the tiles hx, hx + nhelp, ...
    int rtv[ASM_BAND_TPH], ctv[ASM_BAND_TPH];
    v4f64 acc[ASM_BAND_TPH][4];
#pragma unroll
    for (int j = 0; j < ASM_BAND_TPH; ++j) {
        const int ti = hx + j * nhelp;
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= ti) ++a;
        rtv[j] = ti < ntiles ? nsteps + a : -1;
        ctv[j] = nsteps + (ti - a * (a + 1) / 2);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = I0 + rtv[j] * ASM_NB + w * 16 + (lane >> 4) + 4 * r, col = I0 + ctv[j] * ASM_NB + t * 16 + (lane & 15);
                const int rc = max(0, min(row, Ms - 1));
                const double v = S[(int64_t)rc * ldS + max(0, min(col, rc))];
                acc[j][t][r] = ((rtv[j] >= 0) & (row < Ms) & (col <= row)) ? v : 0.0;
            }
    }
    for (int k = 0; k < nsteps; ++k) {
        const int k0 = I0 + k * ASM_NB;
        const int nb = min(ASM_NB, Ms - k0);
#pragma unroll
        for (int j = 0; j < ASM_BAND_TPH; ++j) {
            if (rtv[j] < 0) continue;                 // (uniform over the workgroup)
            const int r0 = I0 + rtv[j] * ASM_NB, c0 = I0 + ctv[j] * ASM_NB;
            pnl_wait_helper(flags + ASM_PNL_NS + ASM_PNL_NRT * k + rtv[j], epoch, tmo);
            if (ctv[j] != rtv[j]) pnl_wait_helper(flags + ASM_PNL_NS + ASM_PNL_NRT * k + ctv[j], epoch, tmo);
            double pv[ASM_NB * ASM_NB / 256], qv[ASM_NB * ASM_NB / 256];
#pragma unroll
            for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
                const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
                pv[it] = S[(int64_t)min(r0 + rr, Ms - 1) * ldS + k0 + min(c, nb - 1)];
                qv[it] = S[(int64_t)min(c0 + rr, Ms - 1) * ldS + k0 + min(c, nb - 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                          // the previous product has consumed the LDS images
#pragma unroll
            for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
                const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
                Pa[rr * ASM_XP + c] = ((r0 + rr < Ms) & (c < nb)) ? -pv[it] : 0.0;
                Pb[rr * ASM_XP + c] = ((c0 + rr < Ms) & (c < nb)) ? qv[it] : 0.0;
            }
            __syncthreads();
#pragma unroll 4
            for (int kk = 0; kk < ASM_NB; kk += 4) {
                double af = Pa[(w * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double bf = Pb[(t * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
                    acc[j][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[j][t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < ASM_BAND_TPH; ++j) {
        if (rtv[j] < 0) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = I0 + rtv[j] * ASM_NB + w * 16 + (lane >> 4) + 4 * r, col = I0 + ctv[j] * ASM_NB + t * 16 + (lane & 15);
                if (row < Ms && col <= row) S[(int64_t)row * ldS + col] = acc[j][t][r];
            }
    }
}
// (two wavefronts per SIMD like k_chol_panel: main + helper workgroups are up to ~360 and must all be resident, two per CU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chol_panel_band(AsmBt abt, double* __restrict__ S, int64_t ldS, int I0, int I1, int Ms, const double* __restrict__ diag0, double thr, double* __restrict__ Linv, unsigned* __restrict__ flags, unsigned* __restrict__ tmo, unsigned epoch, int G) {
    ASM_BARGS(abt, S, ldS, I0, I1, Ms, diag0, thr, Linv, flags, tmo, epoch, G);
    __shared__ __attribute__((aligned(16))) double sm[ASM_PNL_LDS];
    if ((int)blockIdx.x < G) {
        __builtin_amdgcn_s_setprio(3);      // the chain's wavefronts win the issue slots of a SIMD they share with a helper's
        chol_panel_body<true, true, true>(sm, S, ldS, I0, I1, Ms, diag0, thr, Linv, flags, tmo, epoch, G, (int)blockIdx.x);
    } else chol_band_update_tile(sm, S, ldS, I0, I1, Ms, flags, tmo, epoch, (int)blockIdx.x - G, (int)gridDim.x - G);
}

// ---------------------------------------------------------------------------------------------------------------
// The explicit inverse of a factor that is ONE wide block (Ms <= wb), made INSIDE the panel launch by helper workgroups instead of
// 2 log2(wb / 64) + 2 dependent launches afterwards (k_trtri_*): block row r of W = L^-1 follows from the rows above it,
//     W[r][r] = Linv_r,      W[r][j] = -Linv_r * sum_{i=j}^{r-1} L[r][i] W[i][j]      (j < r),
// and everything on the right is final soon after the factorisation reaches block row r: L[r][i] once step i has solved row tile r
// (panel-tile flag), W[i][j] once its own helper is done (flags[FW + ...]), Linv_r with the diagonal flag.  Helper hx owns tile
// (r, j) = (s0 + hx / T, hx % T): it accumulates the sum as its terms arrive, so after the last diagonal block only two 64^3
// products remain.  It writes X(r, j) and the transposed copy XT(j, r) (zeros above the diagonal, as k_trtri_init leaves them);
// rows past the end of the matrix: identity on the diagonal, zero elsewhere (Linv_r carries the identity there, the rows of L
// read as zero).  Helpers come after the main workgroups in the grid and only wait for workgroups with smaller indices.
#define ASM_PNL_FW (ASM_PNL_NS * (ASM_PNL_NS + 1))
#define ASM_PNL_WT 16          // most 64-blocks of one wide block (wb <= 1024)
#define ASM_PNL_FLAGS (ASM_PNL_NS * (ASM_PNL_NRT + 1) > ASM_PNL_FW + ASM_PNL_NS * ASM_PNL_WT ? ASM_PNL_NS * (ASM_PNL_NRT + 1) : ASM_PNL_FW + ASM_PNL_NS * ASM_PNL_WT)
__device__ __forceinline__ void chol_inv_tile(double* __restrict__ sm, const double* __restrict__ S, int64_t ldS, int I0, int Ms, const double* __restrict__ Linv,
                                              unsigned* __restrict__ flags, unsigned* __restrict__ tmo, unsigned epoch, double* __restrict__ X, double* __restrict__ XT,
                                              int wb, int hx) {
    double* Pa = sm;                                  // left operand P[r][k]; afterwards the finished tile
    double* Qt = sm + ASM_NB * ASM_XP;                // right operand transposed Qt[c][k] = Q[k][c]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int T = (Ms + ASM_NB - 1) / ASM_NB, s0 = I0 / ASM_NB;
    const int r = s0 + hx / T, j = hx % T, rl = r - s0;
    v4f64 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    if (j < r) {
        for (int i = j; i <= r; ++i) {
            // terms i < r: P = L[r][i], Q = W[i][j];   the closing product i == r: P = -Linv_r, Q = the sum (read back from the accumulators)
            const double* Qsrc = nullptr;             // 64 x 64 row-major block (Linv) or rows of pitch wb (X)
            int64_t qld = ASM_NB;
            if (i < r) {
                if (i >= s0) pnl_wait_helper(flags + ASM_PNL_NS + ASM_PNL_NS * (i - s0) + rl, epoch, tmo);      // L[r][i]: step i - s0 has solved row tile rl
                if (i == j) {
                    if (j >= s0) pnl_wait_helper(flags + (j - s0), epoch, tmo);
                    Qsrc = Linv + (int64_t)j * ASM_NB * ASM_NB;
                } else {
                    if (i >= s0) pnl_wait_helper(flags + ASM_PNL_FW + ASM_PNL_WT * (i - s0) + j, epoch, tmo);
                    Qsrc = X + (int64_t)i * ASM_NB * wb + (int64_t)j * ASM_NB;
                    qld = wb;
                }
            } else {
                pnl_wait_helper(flags + rl, epoch, tmo);
            }
            double pv[ASM_NB * ASM_NB / 256], qv[ASM_NB * ASM_NB / 256];
#pragma unroll
            for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
                const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
                if (i < r) {
                    const int gi = r * ASM_NB + rr;
                    const double v = S[(int64_t)min(gi, Ms - 1) * ldS + i * ASM_NB + c];
                    pv[it] = gi < Ms ? v : 0.0;
                    qv[it] = Qsrc[(int64_t)rr * qld + c];
                } else {
                    pv[it] = -Linv[(int64_t)r * ASM_NB * ASM_NB + e];
                    qv[it] = 0.0;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                          // the previous product has consumed the LDS images
            if (i == r) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = w * 16 + (lane >> 4) + 4 * q, col = t * 16 + (lane & 15);
                        Qt[col * ASM_XP + row] = acc[t][q];
                        acc[t][q] = 0.0;
                    }
            }
#pragma unroll
            for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
                const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
                Pa[rr * ASM_XP + c] = pv[it];
                if (i < r) Qt[c * ASM_XP + rr] = qv[it];
            }
            __syncthreads();
#pragma unroll 4
            for (int kk = 0; kk < ASM_NB; kk += 4) {
                double af = Pa[(w * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double bf = Qt[(t * 16 + (lane & 15)) * ASM_XP + kk + (lane >> 4)];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) Pa[(w * 16 + (lane >> 4) + 4 * q) * ASM_XP + t * 16 + (lane & 15)] = acc[t][q];
    } else if (j == r) {
        pnl_wait_helper(flags + rl, epoch, tmo);
#pragma unroll
        for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
            const int e = tid + 256 * it;
            Pa[(e >> 6) * ASM_XP + (e & 63)] = Linv[(int64_t)r * ASM_NB * ASM_NB + e];
        }
    } else {
#pragma unroll
        for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
            const int e = tid + 256 * it;
            Pa[(e >> 6) * ASM_XP + (e & 63)] = 0.0;
        }
    }
    __syncthreads();
    // the tile and its transpose, both with coalesced rows; write-through (read by the helpers of later block rows)
#pragma unroll
    for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
        const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
        __hip_atomic_store(X + (int64_t)(r * ASM_NB + rr) * wb + j * ASM_NB + c, Pa[rr * ASM_XP + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        XT[(int64_t)(j * ASM_NB + rr) * wb + r * ASM_NB + c] = Pa[c * ASM_XP + rr];
    }
    if (j < r) pnl_publish(flags + ASM_PNL_FW + ASM_PNL_WT * rl + j, epoch);
}
// k_chol_panel_inv: k_chol_panel_solo plus the helper workgroups of the inverse (grid = G + steps * T)
__global__ __launch_bounds__(256) void k_chol_panel_inv(AsmBt abt, double* __restrict__ S, int64_t ldS, int I0, int I1, int Ms, const double* __restrict__ diag0, double thr, double* __restrict__ Linv, unsigned* __restrict__ flags, unsigned* __restrict__ tmo, unsigned epoch, double* __restrict__ Binv, double* __restrict__ BinvT, int wb, int G) {
    ASM_BARGS(abt, S, ldS, I0, I1, Ms, diag0, thr, Linv, flags, tmo, epoch, Binv, BinvT, wb, G);
    __shared__ __attribute__((aligned(16))) double sm[ASM_PNL_LDS];
    if ((int)blockIdx.x < G) chol_panel_body<false, false, true, true>(sm, S, ldS, I0, I1, Ms, diag0, thr, Linv, flags, tmo, epoch, G, (int)blockIdx.x);
    else chol_inv_tile(sm, S, ldS, I0, Ms, Linv, flags, tmo, epoch, Binv, BinvT, wb, (int)blockIdx.x - G);
}

// out[i] = || A[i, :] ||_2   (one wavefront per row) - KT_residuals / compute_nu! (common.jl:41, slp.jl:58)
__global__ __launch_bounds__(256) void k_row_norms(AsmBt abt, const double* __restrict__ A, int64_t ld, double* __restrict__ out, int64_t M, int64_t ncols) {
    ASM_BARGS(abt, A, ld, out, M, ncols);
    int64_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    int lane = threadIdx.x & 63;
    const double* row = A + i * ld;
    double acc = 0.0;
    for (int64_t j = lane; j < ncols; j += 64) acc = fma(row[j], row[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[i] = sqrt(acc);
}

// the same norms from the CSR copy of a sparse pattern, bit-identical to k_row_norms: lane l adds the entries of the columns = l (mod 64) in
// column order (the zeros the dense sweep adds change nothing), then the same wavefront reduction
__global__ __launch_bounds__(256) void k_sp_row_norms(AsmBt abt, const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ v, double* __restrict__ out, int64_t M) {
    ASM_BARGS(abt, ptr, col, v, out, M);
    int64_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k)
        if ((col[k] & 63) == lane) acc = fma(v[k], v[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[i] = sqrt(acc);
}

// Scaling helpers (oracle/lp_solver.py: scale_lp): rmax[i] = max_j |J_ij| (1 if the row is empty), then
// rel[j] = max_i |J_ij| / rmax[i] with the same deterministic two-stage column reduction as k_gemv_t.
__global__ __launch_bounds__(256) void k_row_absmax(AsmBt abt, const double* __restrict__ A, int64_t ld, double* __restrict__ out, int64_t M, int64_t ncols) {
    ASM_BARGS(abt, A, ld, out, M, ncols);
    int64_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    int lane = threadIdx.x & 63;
    const double* row = A + i * ld;
    double mx = 0.0;
    for (int64_t j = lane; j < ncols; j += 64) mx = fmax(mx, fabs(row[j]));
    mx = wave_max(mx);
    if (lane == 0) out[i] = mx > 0.0 ? mx : 1.0;
}
__global__ __launch_bounds__(256) void k_col_relmax_stage1(AsmBt abt, const double* __restrict__ A, int64_t ld, const double* __restrict__ rmax, double* __restrict__ partial, int64_t M, int64_t ncols, int64_t chunk) {
    ASM_BARGS(abt, A, ld, rmax, partial, M, ncols, chunk);
    int64_t j = blockIdx.x * 256 + threadIdx.x;
    int64_t r = blockIdx.y;
    int64_t i0 = r * chunk, i1 = i0 + chunk;
    if (i1 > M) i1 = M;
    if (j >= ncols) return;
    double mx = 0.0;
    for (int64_t i = i0; i < i1; ++i) mx = fmax(mx, fabs(A[i * ld + j]) / rmax[i]);
    partial[r * ncols + j] = mx;
}
__global__ __launch_bounds__(256) void k_col_relmax_stage2(AsmBt abt, const double* __restrict__ partial, double* __restrict__ out, int64_t R, int64_t ncols) {
    ASM_BARGS(abt, partial, out, R, ncols);
    int64_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    double mx = 0.0;
    for (int64_t r = 0; r < R; ++r) mx = fmax(mx, partial[r * ncols + j]);
    out[j] = mx;
}

// The same three scaling passes on the CSR / CSC pattern of a sparse Jacobian (values gathered in pattern order: v[k] = J[off[k]]): the
// maxima are order-independent, so the results are those of the dense kernels - which spend 0.9 ms per pass on the zeros at 18 637 x 11 192.
__global__ __launch_bounds__(256) void k_sp_row_absmax(AsmBt abt, const int* __restrict__ ptr, const double* __restrict__ v, double* __restrict__ out, int64_t M) {
    ASM_BARGS(abt, ptr, v, out, M);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    double mx = 0.0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) mx = fmax(mx, fabs(v[k]));
    out[i] = mx > 0.0 ? mx : 1.0;
}
__global__ __launch_bounds__(256) void k_sp_col_relmax(AsmBt abt, const int* __restrict__ cptr, const int* __restrict__ crow, const int* __restrict__ cpos, const double* __restrict__ v, const double* __restrict__ rmax, double* __restrict__ out, int64_t n, int64_t ncols) {
    ASM_BARGS(abt, cptr, crow, cpos, v, rmax, out, n, ncols);
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    double mx = 0.0;
    if (j < n)
        for (int q = cptr[j]; q < cptr[j + 1]; ++q) mx = fmax(mx, fabs(v[cpos[q]]) / rmax[crow[q]]);
    out[j] = mx;
}
// rho_i = pow2(max |J_ij c_j|), Ah_ij = J_ij c_j / rho_i at the pattern's entries (the rest of the dense copy is zero and stays zero), and the
// same values in pattern order (vAh)
__global__ __launch_bounds__(256) void k_sp_scale_rows(AsmBt abt, const int* __restrict__ ptr, const int* __restrict__ col, const int64_t* __restrict__ off, const double* __restrict__ vJ, const double* __restrict__ c, double* __restrict__ Ah, double* __restrict__ vAh, double* __restrict__ rho, int64_t M) {
    ASM_BARGS(abt, ptr, col, off, vJ, c, Ah, vAh, rho, M);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    double mx = 0.0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) mx = fmax(mx, fabs(vJ[k] * c[col[k]]));
    const double r = pow2_round_dev(mx);
    rho[i] = r;
    const double inv = 1.0 / r;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) {
        const double a = vJ[k] * c[col[k]] * inv;
        Ah[off[k]] = a;
        vAh[k] = a;
    }
}

// FP64 matrix-core peak probe: back-to-back v_mfma_f64_16x16x4_f64 on 8 independent accumulators per wavefront,
// operands in registers, no memory traffic.  Used only to measure the roofline denominator on the box at hand.
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_f64_peak(AsmBt abt, double* __restrict__ out, int iters) {
    ASM_BARGS(abt, out, iters);
    v4f64 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = 1.0 + 1e-9 * (threadIdx.x + i); b[i] = 1.0 - 1e-9 * (threadIdx.x + 2 * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;      // keep the accumulators alive
}

// nz[t][c] = 1 iff the 32T-row tile t of A has a non-zero in k-chunk c (32 columns).  One workgroup per (tile, group of
// 8 chunks); feeds the chunk skipping of the Schur build for sparse Jacobians (ACOPF: ~3 non-zeros per row).
__global__ __launch_bounds__(256) void k_tile_nzflags(AsmBt abt, const double* __restrict__ A, int64_t ld, int64_t M, int tile_rows, int nchunks, unsigned char* __restrict__ nz, int nzpitch, const int* __restrict__ idx) {
    ASM_BARGS(abt, A, ld, M, tile_rows, nchunks, nz, nzpitch, idx);
    const int t = blockIdx.x, c0 = blockIdx.y * 8;
    __shared__ int any[8];
    if (threadIdx.x < 8) any[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)t * tile_rows;
    // 256 threads = 8 chunks x 32 columns; loop over the rows of the tile
    const int col = c0 * ASM_KC + threadIdx.x;
    const int cc = threadIdx.x >> 5;
    bool f = false;
    if (c0 + cc < nchunks)
        for (int r = 0; r < tile_rows; ++r) {
            int64_t i = r0 + r;
            if (i >= M) break;
            const int64_t ri = idx ? idx[i] : i;      // gathered row set of an active-set build
            f = f || (A[ri * ld + col] != 0.0);
        }
    if (f) any[cc] = 1;
    __syncthreads();
    if (threadIdx.x < 8 && c0 + threadIdx.x < nchunks) nz[(int64_t)t * nzpitch + c0 + threadIdx.x] = (unsigned char)any[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------
// Wide-block triangular solves.  After the factorisation the 512 x 512 diagonal blocks of L are inverted explicitly
// (k_trtri_init / k_trtri_level, from the 64 x 64 block inverses of k_potrf_diag), so a solve needs 2-3 launches per wide
// block and direction instead of one per 64-wide block.
// WB = width of a wide block (template parameter: 512 for small systems, 1024 otherwise - asm_hip.hip: pick_wb)
// Block inversion by divide and conquer (dependent chain of 2*log2 launches).  The inverse of a lower-triangular 2h x 2h block [L11 0; L21 L22] is [X11 0; -X22 L21 X11  X22];
// with X11, X22 (h x h) known, one level is two launches:
//   stage 0:  T   = L21 X11      tile (i,j) = sum_{k>=j} L21[i][k] X11[k][j]      (X11 lower triangular)
//   stage 1:  X21 = -X22 T       tile (i,j) = sum_{k<=i} X22[i][k] T[k][j]
// h = 1, 2, 4, ... 64-blocks; T lives at the X21 position of the scratch buffer (BinvT, transposed copy written later).
// 64x64x64 products on the matrix cores: wavefront w owns rows 16w..16w+15, four 16x16 tiles, 16 k-steps of
// v_mfma_f64_16x16x4_f64; the right operand is staged transposed so that both fragment reads are conflict-free
// (pitch = 2 mod 32 doubles).
#define ASM_TP 66
template <int WB>
__device__ __forceinline__ void trtri_init_tile(const double* __restrict__ Linv, int Ms, double* __restrict__ Binv, int B, int i, int j) {
    const int b0 = B * WB;
    const int nsub = min((WB / ASM_NB), (Ms - b0 + ASM_NB - 1) / ASM_NB);
    double* X = Binv + (int64_t)B * WB * WB;
    const double* src = (i == j && i < nsub) ? Linv + (int64_t)((b0 / ASM_NB) + i) * ASM_NB * ASM_NB : nullptr;
    _Pragma("unroll") for (int e_it = 0; e_it < ASM_NB * ASM_NB / 256; ++e_it) {
        const int e = threadIdx.x + 256 * e_it;
        int rr = e >> 6, c = e & 63;
        double v = src ? src[e] : ((i == j && rr == c) ? 1.0 : 0.0);      // missing sub-blocks: identity
        X[(int64_t)(i * ASM_NB + rr) * WB + j * ASM_NB + c] = v;
    }
}
template <int WB>
__global__ __launch_bounds__(256) void k_trtri_init(AsmBt abt, const double* __restrict__ Linv, int Ms, double* __restrict__ Binv) {
    ASM_BARGS(abt, Linv, Ms, Binv);
    trtri_init_tile<WB>(Linv, Ms, Binv, blockIdx.x, blockIdx.y / (WB / ASM_NB), blockIdx.y % (WB / ASM_NB));
}
template <int WB>
__device__ __forceinline__ void trtri_level_tile(double* __restrict__ Pa, double* __restrict__ Qt, const double* __restrict__ L, int64_t ld, int Ms, double* __restrict__ Binv,
                                                 double* __restrict__ Tbuf, int h, int stage, int B, int pr, int ti, int tj) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b0 = B * WB;
    const int c0 = pr * 2 * h * ASM_NB, r0 = c0 + h * ASM_NB;          // inside the wide block: cols of "1", rows of "2"
    double* X = Binv + (int64_t)B * WB * WB;
    double* T = Tbuf + (int64_t)B * WB * WB;
    v4f64 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int k_lo = stage == 0 ? tj : 0, k_hi = stage == 0 ? h - 1 : ti;
    // rows of this output tile past the end of the matrix: the block inverse is the identity there, the off-diagonal tile is zero
    const bool dead = b0 + r0 + ti * ASM_NB >= Ms;
    // operand tiles of step k: sixteen loads per operand and thread, all in flight (unconditional, clamped rows), kept in registers
    // until the LDS image of the previous step has been consumed - the loads of step k+1 run under the matrix instructions of step k
    double pv[ASM_NB * ASM_NB / 256], qv[ASM_NB * ASM_NB / 256];
    auto load = [&](int k) {
#pragma unroll
        for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
            const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
            if (stage == 0) {          // P = L21[ti][k] (rows of the factor), Q = X11[k][tj]
                const int gi = b0 + r0 + ti * ASM_NB + rr;
                const double v = L[(int64_t)min(gi, Ms - 1) * ld + b0 + c0 + k * ASM_NB + c];
                pv[it] = gi < Ms ? v : 0.0;
                qv[it] = X[(int64_t)(c0 + k * ASM_NB + rr) * WB + c0 + tj * ASM_NB + c];
            } else {                   // P = X22[ti][k], Q = T[k][tj]
                pv[it] = X[(int64_t)(r0 + ti * ASM_NB + rr) * WB + r0 + k * ASM_NB + c];
                qv[it] = T[(int64_t)(r0 + k * ASM_NB + rr) * WB + c0 + tj * ASM_NB + c];
            }
        }
    };
    if (!dead) load(k_lo);
    for (int k = k_lo; k <= k_hi && !dead; ++k) {
#pragma unroll
        for (int it = 0; it < ASM_NB * ASM_NB / 256; ++it) {
            const int e = tid + 256 * it, rr = e >> 6, c = e & 63;
            Pa[rr * ASM_TP + c] = pv[it];
            Qt[c * ASM_TP + rr] = qv[it];
        }
        __syncthreads();
        if (k < k_hi) load(k + 1);
#pragma unroll 4
        for (int kk = 0; kk < ASM_NB; kk += 4) {
            double af = Pa[(w * 16 + (lane & 15)) * ASM_TP + kk + (lane >> 4)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                double bf = Qt[(t * 16 + (lane & 15)) * ASM_TP + kk + (lane >> 4)];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    double* out = stage == 0 ? T : X;
    const double sgn = stage == 0 ? 1.0 : -1.0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = w * 16 + (lane >> 4) + 4 * r, col = t * 16 + (lane & 15);
            out[(int64_t)(r0 + ti * ASM_NB + row) * WB + c0 + tj * ASM_NB + col] = sgn * acc[t][r];
        }
}
template <int WB>
__global__ __launch_bounds__(256) void k_trtri_level(AsmBt abt, const double* __restrict__ L, int64_t ld, int Ms, double* __restrict__ Binv, double* __restrict__ Tbuf, int h, int stage) {
    ASM_BARGS(abt, L, ld, Ms, Binv, Tbuf, h, stage);
    __shared__ double Pa[ASM_NB * ASM_TP];      // left operand  P[r][k]
    __shared__ double Qt[ASM_NB * ASM_TP];      // right operand transposed  Qt[c][k] = Q[k][c]
    trtri_level_tile<WB>(Pa, Qt, L, ld, Ms, Binv, Tbuf, h, stage, blockIdx.x, blockIdx.y, asm_bz(abt) / h, asm_bz(abt) % h);
}

// forward, wide block B:  z_B = X_B w_B   (one wavefront per row: 128 workgroups of 4 rows; all loads of a row in flight)
template <int WB>
__global__ __launch_bounds__(256) void k_wtrsv_fwd_diag(AsmBt abt, const double* __restrict__ Binv, int B, int Ms, const double* __restrict__ w, double* __restrict__ z) {
    ASM_BARGS(abt, Binv, B, Ms, w, z);
    const int b0 = B * WB;
    const double* X = Binv + (int64_t)B * WB * WB;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wv;
    if (b0 + row >= Ms) return;
    double v[WB / 64];
#pragma unroll
    for (int u = 0; u < WB / 64; ++u) {
        int c = u * 64 + lane;
        // unconditional loads (the block and the clamped vector entry are always valid), selected afterwards: loads under a branch
        // are waited for one pair at a time - sixteen serialised round trips instead of "all loads of a row in flight"
        const double xv = X[(int64_t)row * WB + c], wv_ = w[min(b0 + c, Ms - 1)];
        v[u] = (c <= row) ? xv * wv_ : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < WB / 64; ++u) acc += v[u];
    acc = wave_sum(acc);
    if (lane == 0) z[b0 + row] = acc;
}
// forward panel update:  w[i] -= L[i, b0:b1] . z[b0:b1]   for i >= b1.  A wavefront owns ASM_FWD_RPW rows: their loads are issued
// together, the sums are reduced, and the first lanes apply the read-modify-writes in parallel.  (Round 3: 8 rows per wavefront, i.e. 44
// workgroups for the 1 400 rows a wide block of the banded S0 reaches - 13 us on a sixth of the chip; 2 rows: 175 workgroups.)
#ifndef ASM_FWD_RPW
#define ASM_FWD_RPW 2
#endif
template <int WB>
__global__ __launch_bounds__(256) void k_wtrsv_fwd_panel(AsmBt abt, const double* __restrict__ L, int64_t ld, int B, int Ms, const double* __restrict__ z, double* __restrict__ w) {
    ASM_BARGS(abt, L, ld, B, Ms, z, w);
    __shared__ double zs[WB];
    const int b0 = B * WB, b1 = min(b0 + WB, Ms), wdt = b1 - b0;
    for (int c = threadIdx.x; c < WB; c += 256) zs[c] = c < wdt ? z[b0 + c] : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i0 = b1 + blockIdx.x * (4 * ASM_FWD_RPW) + wv * ASM_FWD_RPW;
    double wold = 0.0;
    if (lane < ASM_FWD_RPW && i0 + lane < Ms) wold = w[i0 + lane];
    double acc[ASM_FWD_RPW];
#pragma unroll
    for (int rr = 0; rr < ASM_FWD_RPW; ++rr) {
        int i = i0 + rr;
        double a = 0.0;
        if (i < Ms) {
            const double* row = L + (int64_t)i * ld + b0;
#pragma unroll
            for (int u = 0; u < WB / 64; ++u) {
                int c = u * 64 + lane;
                a = fma(c < wdt ? row[c] : 0.0, zs[c], a);
            }
        }
        acc[rr] = a;
    }
    double mine = 0.0;
#pragma unroll
    for (int rr = 0; rr < ASM_FWD_RPW; ++rr) {
        double t = wave_sum(acc[rr]);
        if (lane == rr) mine = t;
    }
    if (lane < ASM_FWD_RPW && i0 + lane < Ms) w[i0 + lane] = wold - mine;
}
// backward partial sums for wide block B over chunks of 64 rows i >= b1:  part[g][c] = sum_i L[i, b0+c] x[i]
#ifndef ASM_WBROWS
#define ASM_WBROWS 16      // rows per partial sum of the backward panel product (round 3: 64, i.e. 22 workgroups for the reach of a wide block of S0)
#endif
template <int WB>
__global__ __launch_bounds__(256) void k_wtrsv_bwd_panel(AsmBt abt, const double* __restrict__ L, int64_t ld, int B, int Ms, const double* __restrict__ x, double* __restrict__ part) {
    ASM_BARGS(abt, L, ld, B, Ms, x, part);
    __shared__ double red[4][WB];
    const int b0 = B * WB, b1 = min(b0 + WB, Ms), wdt = b1 - b0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc[WB / 64];
#pragma unroll
    for (int u = 0; u < WB / 64; ++u) acc[u] = 0.0;
    const int base = b1 + blockIdx.x * ASM_WBROWS;
    for (int r0 = wv; r0 < ASM_WBROWS; r0 += 16) {            // 4 rows per batch and wavefront: 32 loads in flight
        double xi[4];
        const double* rowp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int i = base + r0 + 4 * q;
            bool ok = i < Ms;
            xi[q] = ok ? x[i] : 0.0;
            rowp[q] = L + (int64_t)(ok ? i : b1) * ld + b0;
        }
        double v[4][WB / 64];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < WB / 64; ++u) {
                int c = u * 64 + lane;
                v[q][u] = c < wdt ? rowp[q][c] : 0.0;
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < WB / 64; ++u) acc[u] = fma(v[q][u], xi[q], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < WB / 64; ++u) red[wv][u * 64 + lane] = acc[u];
    __syncthreads();
    for (int c = threadIdx.x; c < WB; c += 256)
        part[(int64_t)blockIdx.x * WB + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
// t = z_B - sum_g part[g]   (8 workgroups of 64 columns; partials split over the 4 wavefronts in a fixed order)
template <int WB>
__global__ __launch_bounds__(256) void k_wtrsv_bwd_reduce(AsmBt abt, int B, int Ms, const double* __restrict__ z, const double* __restrict__ part, int n_part, double* __restrict__ t) {
    ASM_BARGS(abt, B, Ms, z, part, n_part, t);
    __shared__ double red[4][64];
    const int b0 = B * WB, b1 = min(b0 + WB, Ms), wdt = b1 - b0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double acc = 0.0;
    for (int g0 = wv; g0 < n_part; g0 += 32) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int g = g0 + 4 * q;
            v[q] = g < n_part ? part[(int64_t)g * WB + c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    red[wv][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 64) t[c] = c < wdt ? z[b0 + c] - ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) : 0.0;
}
// x_B = X_B' t  through the transposed block inverse (row c of XT = column c of X): one wavefront per unknown
template <int WB>
__global__ __launch_bounds__(256) void k_wtrsv_bwd_diag(AsmBt abt, const double* __restrict__ BinvT, int B, int Ms, const double* __restrict__ t, double* __restrict__ x, int tlen) {
    ASM_BARGS(abt, BinvT, B, Ms, t, x, tlen);
    const int b0 = B * WB;
    const double* XT = BinvT + (int64_t)B * WB * WB;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wv;
    if (b0 + row >= Ms) return;
    double v[WB / 64];
#pragma unroll
    for (int u = 0; u < WB / 64; ++u) {
        int c = u * 64 + lane;
        const double xv = XT[(int64_t)row * WB + c], tv = t[min(c, tlen - 1)];          // unconditional, as in k_wtrsv_fwd_diag (t has tlen valid entries)
        v[u] = (c >= row && c < tlen) ? xv * tv : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < WB / 64; ++u) acc += v[u];
    acc = wave_sum(acc);
    if (lane == 0) x[b0 + row] = acc;
}
// XT_B = X_B'  (LDS-tiled transpose of every 512 x 512 block inverse)
template <int WB>
__device__ __forceinline__ void transpose_wb_tile(double* __restrict__ tile, const double* __restrict__ Binv, double* __restrict__ BinvT, int B, int ti, int tj) {
    const double* X = Binv + (int64_t)B * WB * WB;
    double* XT = BinvT + (int64_t)B * WB * WB;
    _Pragma("unroll") for (int e_it = 0; e_it < 16; ++e_it) {
        const int e = threadIdx.x + 256 * e_it;
        int r = e >> 6, c = e & 63;
        tile[r * 65 + c] = X[(int64_t)(ti * 64 + r) * WB + tj * 64 + c];
    }
    __syncthreads();
    _Pragma("unroll") for (int e_it = 0; e_it < 16; ++e_it) {
        const int e = threadIdx.x + 256 * e_it;
        int r = e >> 6, c = e & 63;
        XT[(int64_t)(tj * 64 + r) * WB + ti * 64 + c] = tile[c * 65 + r];
    }
}
template <int WB>
__global__ __launch_bounds__(256) void k_transpose_wb(AsmBt abt, const double* __restrict__ Binv, double* __restrict__ BinvT) {
    ASM_BARGS(abt, Binv, BinvT);
    __shared__ double tile[64 * 65];
    transpose_wb_tile<WB>(tile, Binv, BinvT, blockIdx.x, blockIdx.y / (WB / ASM_NB), blockIdx.y % (WB / ASM_NB));
}

// ---------------------------------------------------------------------------------------------------------------
// Sparse matrix-vector products on the fixed Jacobian pattern (ACOPF: 0.03 % fill).  The dense row-major matrix stays
// the operand of the MFMA kernels; these read a gathered copy of its pattern entries (CSR values, CSC via positions).
// One thread per row / column, fixed summation order -> deterministic.
__global__ __launch_bounds__(256) void k_sp_gather(AsmBt abt, const double* __restrict__ A, const int64_t* __restrict__ off, double* __restrict__ vals, int64_t nnz) {
    ASM_BARGS(abt, A, off, vals, nnz);
    int64_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < nnz) vals[k] = A[off[k]];
}
__global__ __launch_bounds__(256) void k_spmv_n(AsmBt abt, const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ out, int64_t M) {
    ASM_BARGS(abt, ptr, col, vals, x, out, M);
    int64_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    double acc = 0.0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) acc += vals[k] * x[col[k]];
    out[i] = acc;
}
__global__ __launch_bounds__(256) void k_spmv_t(AsmBt abt, const int* __restrict__ cptr, const int* __restrict__ row, const int* __restrict__ pos, const double* __restrict__ vals, const double* __restrict__ y, double* __restrict__ out, int64_t n, int64_t ldn) {
    ASM_BARGS(abt, cptr, row, pos, vals, y, out, n, ldn);
    // eight lanes per column (a bus-voltage column of the ACOPF Jacobian has 10-30 entries, each behind two dependent loads: one thread
    // per column walked them one after the other, 10-16 us per product at 2 400-11 000 columns); partial sums combined by lane shuffles
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    double acc = 0.0;
    if (j < n)
        for (int k = cptr[j] + sub; k < cptr[j + 1]; k += 8) acc += vals[pos[k]] * y[row[k]];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (sub == 0 && j < ldn) out[j] = acc;
}

// out[j][i] = A[i][j]   (64 x 64 LDS tiles; A is rows x ld_in, out is cols x ld_out) - transposed copy of the scaled LP
// matrix for the column form of the Newton system
// band >= 0: square lower-triangular input (a Cholesky factor) whose entries beyond `band` sub-diagonals are zero (band = rows: dense): only
// the tiles that hold factor entries are moved - the transposed factor is read in its upper band only (Dev::trsm_rows, backward pass)
__global__ __launch_bounds__(256) void k_transpose_dense(AsmBt abt, const double* __restrict__ A, int64_t ld_in, int64_t rows, int64_t cols, double* __restrict__ out, int64_t ld_out, int64_t band) {
    ASM_BARGS(abt, A, ld_in, rows, cols, out, ld_out, band);
    __shared__ double tile[64 * 65];
    const int64_t i0 = (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;
    if (band >= 0 && (j0 > i0 + 63 || i0 > j0 + 63 + band)) return;
    _Pragma("unroll") for (int e_it = 0; e_it < 16; ++e_it) {
        const int e = threadIdx.x + 256 * e_it;
        int r = e >> 6, c = e & 63;
        tile[r * 65 + c] = (i0 + r < rows && j0 + c < cols) ? A[(i0 + r) * ld_in + j0 + c] : 0.0;
    }
    __syncthreads();
    _Pragma("unroll") for (int e_it = 0; e_it < 16; ++e_it) {
        const int e = threadIdx.x + 256 * e_it;
        int r = e >> 6, c = e & 63;          // out row j0 + r, out col i0 + c
        if (j0 + r < cols && i0 + c < rows) out[(j0 + r) * ld_out + i0 + c] = tile[c * 65 + r];
    }
}
