// libasmhip: host-side solver logic + C ABI (include/asm_hip.h) above the gfx950 kernels in
// asm_kernels.hip.h.  The algorithm (scaling -> warm active-set verify -> Mehrotra IPM in Schur form ->
// partition identification -> active-set Schur/Cholesky polish) is specified in DESIGN.md; reference
// citations for the formulation are given at each step (file:line under the reference tree).
//
// Round-1 split of work: every O(M*n) and O(M^3) operation (assembly, scaling, Ah x, Ah' y, the Schur
// SYRK, the Cholesky factorisation and the triangular solves) runs on the GPU; the O(M+n) vector
// algebra between them runs on the host and exchanges vectors through pinned staging buffers.
#include "asm_kernels.hip.h"
#include "asm_ipm_kernels.hip.h"
#include "asm_as_kernels.hip.h"
#include "asm_ns_kernels.hip.h"
#include "asm_eval_kernels.hip.h"
#include "asm_batch.hip.h"
#include "../../include/asm_hip.h"

#include <sched.h>
#include <time.h>
#include <atomic>
#include <thread>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

const double INF = std::numeric_limits<double>::infinity();
const double TOL_P = 1e-9, TOL_D = 1e-6;
const int IPM_MAXIT = 60;
const double IPM_GAP_DONE = 1e-3, IPM_DINF_FLOOR = 1e-6;      // stage complete with the dual residual on the solves' accuracy floor (oracle/lp_solver.py)
const double JAM_PINF = 1e-6;    // below this a stagnating primal residual is rounding-level, not a jam (oracle/lp_solver.py)
const int IPM_MCC = 2;           // Gondzio centrality correctors per iteration (oracle/lp_solver.py)
const double MCC_DELTA = 0.3, MCC_BMIN = 0.1, MCC_BMAX = 10.0, MCC_GAMMA = 0.1;
// inner / outer panel widths of the three-level Cholesky.  Inner = 512 = eight 64-wide steps per dataflow launch (ASM_PNL_NS) and ONE
// K = 512 update of the rest of the outer panel; measured at M = 1024 / 1725 / 3889 / 11192 / 18637: 256 -> 0.603 / 1.093 / 2.79 / 14.7 /
// 44.95 ms, 512 -> 0.583 / 1.062 / 2.60 / 14.0 / 43.95 ms, 768 -> 0.674 / 1.226 / 2.96 / 15.0 / 45.0 ms, 1024 -> 0.847 / 1.365 / 3.24 /
// 15.7 / 46.4 ms (wider: the rank-64 updates inside the launch grow onto the critical path; narrower: more launches and more of the
// poorly filled in-panel updates)
const int CHOL_NBI = 512, CHOL_NBO = 1024;
// column (Sherman-Morrison-Woodbury) form of the Newton system for restoration LPs (oracle/lp_solver.py: COL_*)
const int COL_MIN_M = 64, COL_MAX_CG = 6;
const double COL_MAX_RATIO = 0.8, COL_FIXED = 1e200;
// reduced row form of the normal-phase Newton system, large sparse problems only (oracle/lp_solver.py: RED_*)
const int RED_MIN_M = 4096, RED_MAX_CG = 10;
const double RED_TAU = 100.0, RED_MIN_FRAC = 0.1;
// null-space form of the normal-phase Newton system (oracle/lp_solver.py: NS_*)
const int NS_MIN_E = 64;
const double NS_MAX_RATIO = 0.3, NS_WARM_THR = 1e-6, NS_ZWARM_THR = 0.25, NS_BIG = 0.5e128, NS_RERR = 1e-6;
const int NS_CMAX = 2;
const double IPM_DEGRADE = 10.0;  // oracle/lp_solver.py: a stage that ends this much worse than the best stage so far is undone (best-iterate safeguard)
const int WARM_BACKOFF_MAX = 6;  // oracle/lp_solver.py: pause after consecutive failed warm attempts doubles up to 2^6 - 1 LPs
const double EQP_RUNAWAY = 10.0, EQP_MAXCHG = 0.03;
const int EQP_MINCHG = 32;  // oracle/lp_solver.py: growth of the primal residual between two rounds of a bulk correction that ends the attempt
const int64_t RCM_MAX_PAIRS = 50000000;      // sum over the columns of (rows in the column)^2 beyond which no row order is computed
const double IPM_MU0_NORMAL = 0.3;      // oracle/lp_solver.py: initial complementarity of a normal-phase LP in units of scale_q
const double IPM_ACCEPT = 1e-8, IPM_ACCEPT_DUAL = 1e-8;      // oracle/lp_solver.py: last-resort acceptance of a converged iterate (primal residual and gap; dual residual)
const int NS_MAX_SPLIT = 8;
const double NS_SEL_THR[4] = {1e-2, 1e-4, 1e-7, 1e-10};
const int PCG_MAXIT = 20;       // conjugate-gradient steps per Newton solve (preconditioner = the Cholesky factor)
const double PCG_KAPPA = 1e-3;  // Newton-system residual tolerance relative to the current primal residual
const double IPM_RHO_P = 1e-8;   // primal proximal regularisation of the Newton system
// canonical pair of a non-unique optimum (oracle/lp_solver.py: FACE_*)
const int FACE_BULK = 12, FACE_STEPS = 400;
const double FACE_TOL_M = 1e-9;

struct HipError : std::runtime_error {
    explicit HipError(const std::string& s) : std::runtime_error(s) {}
};
#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" +   \
                           std::to_string(__LINE__) + ")");                                              \
    } while (0)

inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

inline double pow2_round(double x) {
    if (!(x > 0.0)) return 1.0;
    int e;
    double f = std::frexp(x, &e);
    if (f < 0.70710678118654752) e -= 1;
    return std::ldexp(1.0, e);
}

typedef std::vector<double> vec;
typedef std::vector<int8_t> ivec;

struct ActiveSet {
    ivec rowst, bst, sst;
    bool valid = false;
};

// adaptive decisions carried from one LP of a phase to the next (oracle/lp_solver.py: solve_scaled `hint`)
struct SolveHint {
    int warm_fail = 0, warm_skip = 1;     // the first re-solve of a phase is not attempted
    bool stable = false;                  // the last two LPs of the phase ended on the same active sets
    bool prefer_ref = false;
    std::vector<int> ns_J;                // basis columns of the null-space form retained from the previous LP of the phase
};

struct TimedRegion {
    hipEvent_t a, b;
    int kind;
    hipStream_t stream;     // the stream the timed launches go to (the look-ahead stream has its own regions)
};

// one Cholesky factor with the explicit inverses the substitution kernels use (the null-space form keeps two besides the main one)
// Reverse Cuthill-McKee order of the rows `rows` of the CSR pattern (sp_ptr, sp_col); two rows are adjacent when they share a column.
// Deterministic (oracle: rcm_order): components are started from the unvisited row of least degree (ties: first in `rows`), the
// breadth-first search appends the unvisited neighbours by (degree, position in `rows`), the whole order is reversed.  Returns positions
// into `rows`; *bandwidth = the largest distance between two adjacent rows in the new order; *pairs = the (row, column) positions, in the
// new order, of the structural non-zeros of the lower triangle of the rows' Gram matrix.
static std::vector<int> rcm_order(const std::vector<int>& rows, const std::vector<int>& sp_ptr, const std::vector<int>& sp_col, int64_t ncols, int* bandwidth,
                                  std::vector<int>* pairs = nullptr) {
    const int nR = (int)rows.size();
    std::vector<std::vector<int>> col_rows((size_t)ncols);
    for (int i = 0; i < nR; ++i)
        for (int k = sp_ptr[rows[i]]; k < sp_ptr[rows[i] + 1]; ++k) col_rows[sp_col[k]].push_back(i);
    // a column shared by very many rows makes the coupling graph (and its Gram matrix) dense: no order then (oracle: RCM_MAX_PAIRS)
    {
        int64_t npair = 0;
        for (const auto& v : col_rows) npair += (int64_t)v.size() * (int64_t)v.size();
        if (npair > RCM_MAX_PAIRS) {
            std::vector<int> id(nR);
            for (int i = 0; i < nR; ++i) id[i] = i;
            if (bandwidth) *bandwidth = nR;
            if (pairs) pairs->clear();
            return id;
        }
    }
    std::vector<std::vector<int>> nbr((size_t)nR);
    for (int i = 0; i < nR; ++i) {
        std::vector<int>& v = nbr[i];
        for (int k = sp_ptr[rows[i]]; k < sp_ptr[rows[i] + 1]; ++k) v.insert(v.end(), col_rows[sp_col[k]].begin(), col_rows[sp_col[k]].end());
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        v.erase(std::remove(v.begin(), v.end(), i), v.end());
    }
    std::vector<int> deg(nR);
    for (int i = 0; i < nR; ++i) deg[i] = (int)nbr[i].size();
    auto by_deg = [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; };
    for (int i = 0; i < nR; ++i) std::sort(nbr[i].begin(), nbr[i].end(), by_deg);
    std::vector<int> starts(nR), order;
    for (int i = 0; i < nR; ++i) starts[i] = i;
    std::sort(starts.begin(), starts.end(), by_deg);
    std::vector<char> seen(nR, 0);
    order.reserve(nR);
    for (int s0 : starts) {
        if (seen[s0]) continue;
        seen[s0] = 1;
        size_t head = order.size();
        order.push_back(s0);
        while (head < order.size()) {
            const int v = order[head++];
            for (int u : nbr[v])
                if (!seen[u]) { seen[u] = 1; order.push_back(u); }
        }
    }
    std::reverse(order.begin(), order.end());
    std::vector<int> pos(nR);
    for (int q = 0; q < nR; ++q) pos[order[q]] = q;
    int bw = 0;
    for (int i = 0; i < nR; ++i)
        for (int u : nbr[i]) bw = std::max(bw, std::abs(pos[i] - pos[u]));
    if (bandwidth) *bandwidth = bw;
    if (pairs) {                    // structural non-zeros (new row, new column <= row) of the rows' Gram matrix, diagonal included
        pairs->clear();
        for (int i = 0; i < nR; ++i) {
            pairs->push_back(pos[i]); pairs->push_back(pos[i]);
            for (int u : nbr[i])
                if (pos[u] < pos[i]) { pairs->push_back(pos[i]); pairs->push_back(pos[u]); }
        }
    }
    return order;
}

struct FacBuf {
    double *S = nullptr, *Linv = nullptr, *Binv = nullptr, *BinvT = nullptr;
    int64_t ld = 0;
    int wb = 512;                   // wide-block width of its substitution kernels
    bool small = false;             // k x k systems of the null-space form: solved in one workgroup when the order is <= ASM_SMALL_MAX
    int band = 0;                   // > 0: the matrix is banded (entry (i, j) is zero when |i - j| > band) and so is its factor: every panel
                                    // operation stops `band` rows below the panel's last column
};

}  // namespace

struct asm_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // look-ahead stream of the Cholesky (next panel's block chain runs beside the trailing update)
    std::vector<hipEvent_t> la_events;
    std::string err;
    bool setup_done = false, inputs_ready = false;

    // ---- problem (subproblem.jl:51-215) ----
    int64_t n = 0, m = 0, nnz = 0, nadj = 0, M = 0, Mp = 0, ldn = 0, ns = 0;
    vec c_lb, c_ub, v_lb, v_ub;
    std::vector<int> kind;          // 0 EQ, 2 range, +1 lower only, -1 upper only
    std::vector<int64_t> adj;       // rows with two distinct finite bounds
    std::vector<int> rtype;         // LP row types (M)
    std::vector<int> srow;          // slack -> LP row
    vec scoef;                      // slack coefficient (+1 / -1)
    std::vector<int64_t> sown;      // slack -> original row (for p_slack)
    std::vector<int> nslack;        // slacks per original row (1 or 2)

    // ---- assembly plan ----
    int64_t nu = 0;
    bool dense_fast = false;
    int64_t *d_perm = nullptr, *d_ustart = nullptr, *d_uoff = nullptr, *d_adjoff = nullptr;

    // ---- device buffers ----
    double *d_dE = nullptr, *d_J = nullptr, *d_Ah = nullptr, *d_S = nullptr;
    double *d_c = nullptr, *d_rho = nullptr, *d_theta = nullptr, *d_diag = nullptr, *d_diag0 = nullptr;
    double *d_vecN = nullptr, *d_vecM = nullptr, *d_vecM2 = nullptr, *d_partial = nullptr;
    double *d_Linv = nullptr, *d_Binv = nullptr, *d_BinvT = nullptr, *d_wpart = nullptr, *d_wt = nullptr;
    int wb = 512;                  // wide-block width of the triangular solves
    // column form: transposed copy of Ah (n x ldT), its chunk flags, work vectors
    bool col_capable = false, ahT_valid = false, nzT_valid = false;
    int64_t ldT = 0;
    double nzT_fraction = 1.0;
    double *d_AhT = nullptr, *d_cdinv = nullptr, *d_cth = nullptr, *d_cu = nullptr, *d_ct = nullptr, *d_cv = nullptr, *d_cw = nullptr;
    unsigned char* d_nzT = nullptr;
    // reduced row form: dropped-row index list, its diagonal, gathered work vectors, s_ii
    int* d_idxI = nullptr;
    double *d_rdI = nullptr, *d_rce = nullptr, *d_rze = nullptr, *d_sdiag = nullptr;
    unsigned char* d_nz = nullptr;  // (row tile, k-chunk) non-zero flags of Ah; second half: flags of a gathered row set
    int64_t nz_half = 0;
    // sparse copy of the fixed Jacobian pattern for the matrix-vector products (sparse patterns only)
    bool sp_ok = false, spv_Ah_valid = false, spv_J_valid = false;
    int64_t sp_nnz = 0;
    int *d_sp_ptr = nullptr, *d_sp_col = nullptr, *d_sc_ptr = nullptr, *d_sc_row = nullptr, *d_sc_pos = nullptr;
    int64_t* d_sp_off = nullptr;
    double *d_spv_Ah = nullptr, *d_spv_J = nullptr;
    bool nz_valid = false;
    int nz_T = 0, nz_pitch = 0;
    double nz_fraction = 1.0;       // executed share of the (tile pair, k-chunk) products of the Schur build
    double nz_frac_cache[2] = {-1.0, -1.0};   // ... cached per handle: all rows / the equality rows (the pattern is fixed)
    // ---- null-space form of the interior-point Newton system (asm_ns_kernels.hip.h; oracle: class NullSpace)
    bool ns_cap = false;            // the LP skeleton qualifies (sparse pattern, enough hard equality rows, small null space)
    int ns_nE = 0, ns_nI = 0, ns_nEp = 0, ns_nIp = 0, ns_kcap = 0;
    int64_t ns_ldg = 0;
    int *d_nsEidx = nullptr, *d_nsEpos = nullptr, *d_nsIidx = nullptr, *d_nsIpos = nullptr, *d_nsJ = nullptr, *d_nscnt = nullptr;
    FacBuf ns_f0, ns_fN;            // factors of S0 = A_EF A_EF' (per LP) and of the k x k reduced matrix (per iteration)
    double *d_nsLt = nullptr, *d_nsR = nullptr, *d_nsX = nullptr, *d_nsG = nullptr, *d_nsth = nullptr, *d_nsFm = nullptr, *d_nsv = nullptr, *d_nsYt = nullptr, *d_nsNp = nullptr, *d_nsN0 = nullptr, *d_nsZT = nullptr;
    FacBuf ns_fC;                   // factor of the Gram matrix of the active constraints in reduced coordinates (active-set solves)
    int ns_ccap = 0;                // most constraints it is sized for
    int64_t ns_npairs = 0;          // structural non-zeros of the lower triangle of S0 (banded S0 only)
    int* d_nsS0pairs = nullptr;
    // all M rows in reverse Cuthill-McKee order (sparse patterns whose bandwidth is below M / 2): the Gram matrices of row subsets taken in
    // that order are banded with half-bandwidth <= row_band
    int row_band = 0;
    int *d_rowperm = nullptr, *d_rowpos = nullptr, *d_rowpairs = nullptr, *d_cpos = nullptr;      // position -> row, row -> position, structural pairs (row_i, row_j), pos_i >= pos_j
    int64_t n_rowpairs = 0;
    std::vector<int> row_perm_h, ns_eidx_h;      // host copies: rows by position (all rows; the equality rows of the null-space form)
    int col_band = 0;               // the same for the n columns (column form of the restoration-phase Newton system)
    int *d_colperm = nullptr, *d_colpos = nullptr, *d_colpairs = nullptr;
    int64_t n_colpairs = 0;
    int main_band_cur = 0;          // band of the matrix now in the main factor buffers (set by the banded builds, 0 after every other build)
    double* d_AhTg = nullptr;       // transposed copy of a dense Ah for the products Ah'y (ldn rows of pitch Mp; made once per LP)
    bool ahTg_valid = false;
    double* d_redpart = nullptr;    // partial results / arrival counter of the multi-workgroup interior-point reductions
    unsigned* d_redcnt = nullptr;
    int main_band = 0;              // band of the matrix in the main factor buffers (test hook asm_test_set_band; 0 = dense)
    int ns_Zk = 0;                  // rows of the orthonormal basis of the previous LP still resident in d_nsG (0: none)
    int *d_nsqi = nullptr;          // sel | bpos | rpos | cnt
    double *d_nsq = nullptr;        // Csel | d, u, lam, v, w | pbar, tbar, u0, qh
    std::vector<void*> ns_bufs;     // everything above, for release
    double* d_ipm_snap = nullptr;   // best-iterate safeguard: copy of the iterate at the end of the best interior-point stage so far
    double* d_ipm = nullptr;        // arena of the device-resident interior-point state
    int* d_ipm_i = nullptr;
    double* d_as = nullptr;         // arena of the device-resident active-set machinery (asm_as_kernels.hip.h)
    int* d_as_i = nullptr;
    int* h_ascnt = nullptr;         // pinned read-back of its counters / scalars
    double* h_asscal = nullptr;
    double *d_Zbuf = nullptr, *d_nsu = nullptr, *d_nsdots = nullptr, *h_nsdots = nullptr;   // null-space active-set method (face_primal_anchored)
    // ---- device-side evaluator (asm_eval_*): flattened function store, NLP block parameters, evaluation results in HBM
    bool ev_ready = false;
    int ev_nlp_kind = 0;
    int64_t ev_nlp_rows = 0, ev_nlp_nnz = 0, ev_fn_nnz = 0;
    FnStore ev_F;
    std::vector<void*> ev_bufs;
    int64_t* d_ev_ipar = nullptr;
    double *d_ev_dpar = nullptr, *d_ev_x = nullptr, *d_ev_xt = nullptr, *d_ev_df = nullptr, *d_ev_E = nullptr, *d_ev_Et = nullptr, *d_ev_f = nullptr;
    double *d_ev_vecs = nullptr, *h_ev = nullptr;     // reduction inputs (lambda, multipliers, nu, slacks, p, bounds) / pinned staging
    bool J_valid = false;                               // the dense J in HBM matches the dE in HBM
    int64_t nsp = 0;
    double* h_scal = nullptr;       // pinned scalar read-back; host-mapped: the reduction kernels store the block there themselves (scal_publish)
    double* d_hscal = nullptr;      // its device address
    unsigned* h_seq = nullptr;      // sequence word the host spins on (host-mapped), d_hseq its device address
    unsigned* d_hseq = nullptr;
    unsigned scal_seq = 0;          // last sequence number handed to a publishing kernel
    bool spin_read = true;          // ASM_HIP_SPIN=0: hipMemcpyAsync + hipStreamSynchronize instead (14.5 us per read-back instead of 6.7 us)
    int* d_idx = nullptr;
    double* h_pin = nullptr;        // pinned staging (max(ldn, Mp) doubles) x 2
    double* d_dl = nullptr;         // the answer of an LP packed on the device (k_as_pack) and its pinned host image: one device-to-host copy per LP instead of eight
    double* h_dl = nullptr;
    double* h_up = nullptr;         // pinned staging of the LP vectors an LP uploads (3 ldn + Mp + 3 nsp doubles, zero beyond the vectors' lengths)
    int64_t pin_len = 0;

    // ---- host copies of the evaluation results (slp.jl:8-21) ----
    vec df, E, x_k;
    double f = 0.0;

    // ---- warm start (retained active set per phase; GLPK keeps its basis, slp.jl:38-40) ----
    ActiveSet warm[2];
    SolveHint hint[2];
    int64_t stats_pcg = 0;          // conjugate-gradient steps since creation (verbose diagnostics)
    ActiveSet last;
    asm_solve_stats stats;

    // ---- kernel timing ----
    asm_kernel_stats kstats;
    std::vector<TimedRegion> regions;
    std::vector<hipEvent_t> event_pool;
    bool batch_slot = false;        // slot of an asm_batch: the stream belongs to the batch, no look-ahead stream, no event timing
    bool fused_panel = true;        // Cholesky inner panels as one dataflow launch (k_chol_panel) instead of three launches per 64-wide step
    int panel_wgs = 240;            // its grid bound: every workgroup must be able to become resident
    int num_cus = 256;              // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    unsigned *d_pflags = nullptr, *d_ptmo = nullptr;
    bool test_no_polish = false;    // test hook: the active-set attempts of an LP all fail (asm_test_no_polish)
    unsigned panel_epoch = 0;
    int timing = 1;                 // HIP-event timing: 0 off, 1 the dominant kernel only (every k_syrk launch), 2 every kernel family
    bool verbose = false;
};

namespace {

template <class T>
void dmalloc(T** p, int64_t count) {
    HIPCHK(hipMalloc((void**)p, std::max<int64_t>(count, 1) * sizeof(T)));
}

// after host -> device copies whose source must stay untouched until they have run: in a scenario batch the payload was copied when the
// operation was recorded, nothing to wait for
inline void h2d_done(asm_handle* h) {
    if (!asmb::in_fiber()) HIPCHK(hipStreamSynchronize(h->stream));
}

double* ns_dalloc(asm_handle* h, int64_t count) {
    double* d = nullptr;
    dmalloc(&d, count);
    h->ns_bufs.push_back((void*)d);
    HIPCHK(hipMemsetAsync(d, 0, std::max<int64_t>(count, 1) * sizeof(double), h->stream));
    return d;
}
// buffers of one Cholesky factor of order <= N (pitch = N rounded up to 32) with the block inverses of the substitution kernels
void ns_alloc_factor(asm_handle* h, FacBuf& f, int64_t N, int band_hint = 0) {
    f.ld = round_up(std::max<int64_t>(N, 1), 32);
    f.wb = (f.ld <= 1024 || f.ld > 1536) ? 1024 : 512;      // one wide block (= the whole inverse) when the factor fits into it
    // a narrow band: the explicit inverse of a wide diagonal block is dense whatever the band, its cost grows with the square of the block
    // width - half the width is a quarter of the inverse (case300-sized S0, band 268 of 2 100: 16 % of a scenario batch's kernel time at 1024)
    if (band_hint > 0 && band_hint <= 512 && f.ld > 1024) f.wb = 512;
    f.S = ns_dalloc(h, f.ld * f.ld);
    f.Linv = ns_dalloc(h, (f.ld / ASM_NB + 1) * ASM_NB * ASM_NB);
    f.Binv = ns_dalloc(h, (f.ld / f.wb + 1) * (int64_t)f.wb * f.wb);
    f.BinvT = ns_dalloc(h, (f.ld / f.wb + 1) * (int64_t)f.wb * f.wb);
}

// =====================================================================================================
// device helpers
// =====================================================================================================
struct Dev {
    asm_handle* h;
    hipStream_t cur;                 // stream the factorisation kernels are launched on (h->stream, or the look-ahead stream)
    double* solve_w = nullptr;       // buffer the wide-block substitution runs in (default d_vecM2)
    const double* solve_src = nullptr;   // one-block systems: the right-hand side the forward product reads (no copy into solve_w first)
    // the factor the Cholesky / substitution launches work on: matrix (lower triangle, pitch fld), inverses of its 64-wide diagonal
    // blocks, explicit inverses of its wide diagonal blocks and their transposes.  Default: the handle's main buffers.
    double *fS, *fLinv, *fBinv, *fBinvT;
    int64_t fld;
    int fwb;
    bool fsmall = false;
    int fband = 0;
    explicit Dev(asm_handle* hh) : h(hh), cur(hh->stream) { use_main(); }
    void use_main() { fS = h->d_S; fld = h->Mp; fLinv = h->d_Linv; fBinv = h->d_Binv; fBinvT = h->d_BinvT; fwb = h->wb; fsmall = false; fband = h->main_band > 0 ? h->main_band : h->main_band_cur; }
    void use_factor(const FacBuf& f) { fS = f.S; fld = f.ld; fLinv = f.Linv; fBinv = f.Binv; fBinvT = f.BinvT; fwb = f.wb; fsmall = f.small; fband = f.band; }
    // the matrix about to be built in the main buffers has this band (0 = dense)
    void set_main_band(int b) { h->main_band_cur = b; if (fS == h->d_S) fband = h->main_band > 0 ? h->main_band : b; }
    // first row that the columns [.., c1) of a banded matrix / factor cannot reach (the order Ms when the matrix is dense)
    int rowlim(int Ms, int c1) const { return fband > 0 ? (int)std::min<int64_t>(Ms, round_up((int64_t)c1 + fband, 64)) : Ms; }

    hipEvent_t get_event() {
        if (!h->event_pool.empty()) {
            hipEvent_t e = h->event_pool.back();
            h->event_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        return e;
    }
    int begin(int kind, double flops, double bytes, hipStream_t on = nullptr) {
        h->kstats.flops[kind] += flops;
        h->kstats.bytes[kind] += bytes;
        h->kstats.calls[kind] += 1;
        if (h->timing < 2 && !(h->timing == 1 && (kind == ASM_K_SYRK_KERNEL || kind == ASM_K_PANEL_KERNEL))) return -1;
        TimedRegion r;
        r.a = get_event();
        r.b = get_event();
        r.kind = kind;
        r.stream = on ? on : h->stream;
        HIPCHK(hipEventRecord(r.a, r.stream));
        h->regions.push_back(r);
        return (int)h->regions.size() - 1;
    }
    void end(int id) {
        if (id < 0) return;
        HIPCHK(hipEventRecord(h->regions[id].b, h->regions[id].stream));
    }
    void resolve_timing() {
        if (h->regions.empty()) return;
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipStreamSynchronize(h->stream2));
        for (auto& r : h->regions) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
            h->kstats.ms[r.kind] += ms;
            h->event_pool.push_back(r.a);
            h->event_pool.push_back(r.b);
        }
        h->regions.clear();
    }

    void h2d(double* dst, const double* src, int64_t cnt, int64_t padded) {
        double* st = h->h_pin;
        std::memcpy(st, src, cnt * sizeof(double));
        for (int64_t i = cnt; i < padded; ++i) st[i] = 0.0;
        HIPCHK(hipMemcpyAsync(dst, st, padded * sizeof(double), hipMemcpyHostToDevice, h->stream));
        h2d_done(h);   // staging buffer is reused by the next call
    }
    void d2h(double* dst, const double* src, int64_t cnt) {
        double* st = h->h_pin + h->pin_len;
        HIPCHK(hipMemcpyAsync(st, src, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        std::memcpy(dst, st, cnt * sizeof(double));
    }

    // out[M] = A x   (A = Ah or J, M rows)
    // gathered pattern values of A (d_Ah or d_J) for the sparse products, nullptr when the pattern is dense
    const double* sparse_vals(const double* A) {
        if (!h->sp_ok) return nullptr;
        double* v = nullptr;
        bool* valid = nullptr;
        if (A == h->d_Ah) { v = h->d_spv_Ah; valid = &h->spv_Ah_valid; }
        else if (A == h->d_J) { v = h->d_spv_J; valid = &h->spv_J_valid; }
        else return nullptr;
        if (!*valid) {
            hipLaunchKernelGGL(k_sp_gather, dim3((unsigned)((h->sp_nnz + 255) / 256)), dim3(256), 0, h->stream, A, h->d_sp_off, v, h->sp_nnz);
            *valid = true;
        }
        return v;
    }
    void launch_gemv_n(const double* A, const double* x, double* out) {
        if (h->M == 0) return;                         // LP without rows
        if (const double* v = sparse_vals(A)) {
            int id = begin(ASM_K_GEMV, 2.0 * h->sp_nnz, 20.0 * h->sp_nnz + 12.0 * h->M);
            hipLaunchKernelGGL(k_spmv_n, dim3((unsigned)((h->M + 255) / 256)), dim3(256), 0, h->stream, h->d_sp_ptr, h->d_sp_col, v, x, out, h->M);
            end(id);
            return;
        }
        int id = begin(ASM_K_GEMV, 2.0 * h->M * h->n, 8.0 * h->M * h->ldn);
        hipLaunchKernelGGL(k_gemv_n, dim3((unsigned)((h->M + 3) / 4)), dim3(256), 0, h->stream, A, h->ldn, x, out, h->M, h->ldn);
        end(id);
    }
    void launch_gemv_t(const double* A, const double* y, double* out) {
        if (h->M == 0) {                               // LP without rows: A'y = 0
            HIPCHK(hipMemsetAsync(out, 0, h->ldn * sizeof(double), h->stream));
            return;
        }
        if (const double* v = sparse_vals(A)) {
            int id = begin(ASM_K_GEMV, 2.0 * h->sp_nnz, 24.0 * h->sp_nnz + 12.0 * h->n);
            hipLaunchKernelGGL(k_spmv_t, dim3((unsigned)((h->ldn * 8 + 255) / 256)), dim3(256), 0, h->stream, h->d_sc_ptr, h->d_sc_row, h->d_sc_pos, v,
                               y, out, h->n, h->ldn);
            end(id);
            return;
        }
        if (A == h->d_Ah && (int64_t)h->ldn * h->Mp <= ((int64_t)1 << 27)) {
            // dense LP matrix of moderate size: through a transposed copy (made once per LP), one row-wise launch
            if (!h->d_AhTg) {
                dmalloc(&h->d_AhTg, h->ldn * h->Mp);
                HIPCHK(hipMemsetAsync(h->d_AhTg, 0, h->ldn * h->Mp * sizeof(double), h->stream));
            }
            if (!h->ahTg_valid) {
                hipLaunchKernelGGL(k_transpose_dense, dim3((unsigned)((h->n + 63) / 64), (unsigned)((h->M + 63) / 64)), dim3(256), 0, h->stream, h->d_Ah,
                                   h->ldn, h->M, h->n, h->d_AhTg, h->Mp, (int64_t)-1);
                h->ahTg_valid = true;
            }
            int id = begin(ASM_K_GEMV, 2.0 * h->M * h->n, 8.0 * h->M * h->ldn);
            hipLaunchKernelGGL(k_gemv_n_exact, dim3((unsigned)((h->ldn + 3) / 4)), dim3(256), 0, h->stream, (const double*)h->d_AhTg, h->Mp, y, out, h->ldn, h->M);
            end(id);
            return;
        }
        int64_t R = std::min<int64_t>((h->M + 31) / 32, ASM_TMAXCHUNKS);
        int64_t chunk = (h->M + R - 1) / R;
        R = (h->M + chunk - 1) / chunk;
        int id = begin(ASM_K_GEMV, 2.0 * h->M * h->n, 8.0 * h->M * h->ldn);
        hipLaunchKernelGGL(k_gemv_t_stage1, dim3((unsigned)((h->ldn + 255) / 256), (unsigned)R), dim3(256), 0, h->stream, A, h->ldn, y,
                           h->d_partial, h->M, h->ldn, chunk);
        hipLaunchKernelGGL(k_gemv_t_stage2, dim3((unsigned)((h->ldn + 255) / 256)), dim3(256), 0, h->stream, h->d_partial, out, R, h->ldn);
        end(id);
    }
    // out[M] = A x   (host vectors)
    void gemv_n(const double* A, const double* x, double* out) {
        h2d(h->d_vecN, x, h->n, h->ldn);
        launch_gemv_n(A, h->d_vecN, h->d_vecM);
        d2h(out, h->d_vecM, h->M);
    }
    // out[n] = A' y
    void gemv_t(const double* A, const double* y, double* out) {
        h2d(h->d_vecM, y, h->M, h->Mp);
        launch_gemv_t(A, h->d_vecM, h->d_vecN);
        d2h(out, h->d_vecN, h->n);
    }

    // device-pointer variants (the IPM keeps its vectors in HBM)
    void gemv_n_dev(const double* A, const double* x, double* out) { launch_gemv_n(A, x, out); }
    void gemv_t_dev(const double* A, const double* y, double* out) { launch_gemv_t(A, y, out); }
    void syrk_dev(const int* idx_dev, int Ms, const double* theta_dev, const double* diag_dev) {
        set_main_band(0);
        const bool skip = h->nz_valid && idx_dev == nullptr && Ms == (int)h->M && pick_tile(Ms) == h->nz_T;
        int id = begin(ASM_K_SYRK, (skip ? h->nz_fraction : 1.0) * (double)Ms * (Ms + 1) * h->ldn,
                       8.0 * (Ms * (double)h->ldn + 0.5 * Ms * (double)Ms));
        launch_syrk(pick_tile(Ms), h->d_Ah, h->ldn, idx_dev, 0, Ms, (int)h->ldn, theta_dev, diag_dev, h->d_S, h->Mp, 0, 0, -1,
                    skip ? h->d_nz : nullptr, h->nz_pitch);
        end(id);
    }
    // S[0:Ms,0:Ms] (lower) = Ah[idx,:] diag(theta) Ah[idx,:]' + diag   with idx / theta / diag already on the device and the
    // chunk flags of the gathered row set (reduced row form of the interior-point system)
    // S[0:Ms,0:Ms] (lower) of the same matrix for a row list in the handle's reverse Cuthill-McKee order (asm_handle::row_band): banded,
    // built entry by entry from the structural pairs - cpos maps a row to its place in the list (-1: not in it).  The factorisation and the
    // substitutions that follow stop at the band (main_band_cur).
    void schur_banded_dev(const int* cpos_dev, int Ms, const double* theta_dev, const double* diag_dev) {
        // the band plus what the blocked factorisation reads beyond it (an outer panel of CHOL_NBO columns, tile rounding)
        const int64_t wz = std::min<int64_t>(round_up(h->row_band + 1, 64) + CHOL_NBO + 128, h->Mp);
        hipLaunchKernelGGL(k_ns_zero_band, dim3((unsigned)((wz + 255) / 256), (unsigned)Ms), dim3(256), 0, h->stream, h->d_S, h->Mp, Ms, (int)wz);
        hipLaunchKernelGGL(k_schur_sparse, dim3((unsigned)((h->n_rowpairs + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_rowpairs, h->n_rowpairs, cpos_dev,
                           h->d_sp_ptr, h->d_sp_col, sparse_vals(h->d_Ah), theta_dev, diag_dev, h->d_S, h->Mp, (const int*)nullptr);
        set_main_band(h->row_band);
    }
    // K = diag + Ah' diag(dinv) Ah (n x n, lower) with the COLUMNS in their reverse Cuthill-McKee order (asm_handle::col_band): the column form
    // of the restoration-phase Newton system, banded and built from the structural column pairs; `diag_place` is indexed by position
    void schur_banded_cols_dev(const double* dinv_dev, const double* diag_place) {
        const int n = (int)h->n;
        const int64_t wz = std::min<int64_t>(round_up(h->col_band + 1, 64) + CHOL_NBO + 128, h->Mp);
        hipLaunchKernelGGL(k_ns_zero_band, dim3((unsigned)((wz + 255) / 256), (unsigned)n), dim3(256), 0, h->stream, h->d_S, h->Mp, n, (int)wz);
        hipLaunchKernelGGL(k_schur_sparse, dim3((unsigned)((h->n_colpairs + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_colpairs, h->n_colpairs,
                           (const int*)h->d_colpos, h->d_sc_ptr, h->d_sc_row, sparse_vals(h->d_Ah), dinv_dev, diag_place, h->d_S, h->Mp, (const int*)h->d_sc_pos);
        set_main_band(h->col_band);
    }
    void syrk_gathered_dev(const int* idx_dev, int Ms, const double* theta_dev, const double* diag_dev) {
        set_main_band(0);
        const int nch = (int)(h->ldn / ASM_KC);
        const int T = pick_tile(Ms), TS = 32 * T;
        const int nt = (Ms + TS - 1) / TS;
        const bool skip = h->nz_valid && nt > 0;
        unsigned char* nz2 = h->d_nz + h->nz_half;
        double frac = 1.0;
        if (skip) {
            hipLaunchKernelGGL(k_tile_nzflags, dim3((unsigned)nt, (unsigned)((nch + 7) / 8)), dim3(256), 0, h->stream, h->d_Ah, h->ldn, (int64_t)Ms, TS, nch,
                               nz2, nch, idx_dev);
            frac = executed_fraction(nz2, nt, nch);
        }
        int id = begin(ASM_K_SYRK, frac * (double)Ms * (Ms + 1) * h->ldn, 8.0 * (Ms * (double)h->ldn + 0.5 * Ms * (double)Ms));
        launch_syrk(T, h->d_Ah, h->ldn, idx_dev, 0, Ms, (int)h->ldn, theta_dev, diag_dev, h->d_S, h->Mp, 0, 0, -1, skip ? nz2 : nullptr, nch, frac);
        end(id);
    }
    // S[0:Ms,0:Ms] (lower, pitch ldS) = Ah[idx,:] diag(theta) Ah[idx,:]'  into an arbitrary buffer (null-space form: S0 = A_EF A_EF')
    void syrk_gathered_into(const int* idx_dev, int Ms, const double* theta_dev, double* S, int64_t ldS) {
        const int nch = (int)(h->ldn / ASM_KC);
        const int T = pick_tile(Ms), TS = 32 * T;
        const int nt = (Ms + TS - 1) / TS;
        const bool skip = h->nz_valid && nt > 0;
        unsigned char* nz2 = h->d_nz + h->nz_half;
        double frac = 1.0;
        if (skip) {
            hipLaunchKernelGGL(k_tile_nzflags, dim3((unsigned)nt, (unsigned)((nch + 7) / 8)), dim3(256), 0, h->stream, h->d_Ah, h->ldn, (int64_t)Ms, TS, nch,
                               nz2, nch, idx_dev);
            frac = executed_fraction(nz2, nt, nch, idx_dev == h->d_nsEidx ? 1 : -1);
        }
        int id = begin(ASM_K_SYRK, frac * (double)Ms * (Ms + 1) * h->ldn, 8.0 * (Ms * (double)h->ldn + 0.5 * Ms * (double)Ms));
        launch_syrk(T, h->d_Ah, h->ldn, idx_dev, 0, Ms, (int)h->ldn, theta_dev, nullptr, S, ldS, 0, 0, -1, skip ? nz2 : nullptr, nch, frac);
        end(id);
    }
    // C = (C0) -/+ A B'  on the matrix cores (k_gemm_nt); K a multiple of 32
    void gemm_nt(const double* A, int64_t lda, const double* B, int64_t ldb, const double* C0, int64_t ldc0, double* C, int64_t ldc, int Ma, int Mb, int K, int mode) {
        if (Ma <= 0 || Mb <= 0) return;
        int id = begin(ASM_K_TRSV, 2.0 * Ma * (double)Mb * K, 8.0 * ((double)(Ma + Mb) * K + (double)Ma * Mb));
        // 64 x 64 tiles leave CUs idle when there are few right-hand sides: 32-row tiles then (same sums, same order)
        const int64_t t64 = (int64_t)((Mb + 63) / 64) * ((Ma + 63) / 64);
        static const int ta_env = [] { const char* v = std::getenv("ASM_GEMM_TA"); return v ? std::atoi(v) : 0; }();
        // ... and 96 columns per workgroup when 32 x 64 tiles overshoot one workgroup per CU and 32 x 96 tiles do not
        const int64_t t3264 = (int64_t)((Mb + 63) / 64) * ((Ma + 31) / 32), t3296 = (int64_t)((Mb + 95) / 96) * ((Ma + 31) / 32);
        static const int tb_env = [] { const char* v = std::getenv("ASM_GEMM_TB"); return v ? std::atoi(v) : 0; }();
        if ((ta_env == 32 || (ta_env == 0 && t64 < 2 * (int64_t)h->num_cus)) && Ma > 32 && (tb_env == 96 || (tb_env == 0 && t3264 > h->num_cus && t3296 <= h->num_cus)))
            hipLaunchKernelGGL(k_gemm_nt32w, dim3((unsigned)((Mb + 95) / 96), (unsigned)((Ma + 31) / 32)), dim3(256), 0, h->stream, A, lda, B, ldb, C0, ldc0, C, ldc, Ma, Mb, K, mode);
        else if ((ta_env == 32 || (ta_env == 0 && t64 < 2 * (int64_t)h->num_cus)) && Ma > 32)
            hipLaunchKernelGGL(k_gemm_nt32, dim3((unsigned)((Mb + 63) / 64), (unsigned)((Ma + 31) / 32)), dim3(256), 0, h->stream, A, lda, B, ldb, C0, ldc0, C, ldc, Ma, Mb, K, mode);
        else
            hipLaunchKernelGGL(k_gemm_nt, dim3((unsigned)((Mb + 63) / 64), (unsigned)((Ma + 63) / 64)), dim3(256), 0, h->stream, A, lda, B, ldb, C0, ldc0, C, ldc, Ma, Mb, K, mode);
        end(id);
    }
    // Rows of R (nrhs x ldr, zero beyond column Ms) are right-hand sides of  L x = r  (forward) and then  L' x = z  (backward) with the
    // CURRENT factor (fS, its wide-block inverses) and its transposed copy Lt (same pitch).  Right-looking block substitution over the
    // wide blocks: the block's solution is a product with the explicit inverse of the diagonal block, then ONE update of all the
    // remaining columns  R[:, rest] -= X_blk L[rest, blk]'  (K = the block width, every tile of the remainder in parallel) - both on the
    // matrix cores.  Forward: R -> X.  Backward (if Lt): X -> R.  The solution ends in R (backward) or X (forward only).
    void trsm_rows(double* R, double* X, int64_t ldr, int nrhs, int Ms, const double* Lt) {
        const int WB = fwb;
        const int nB = (Ms + WB - 1) / WB;
        for (int B = 0; B < nB; ++B) {
            const int b0 = B * WB, wv = std::min(WB, Ms - b0), b1 = b0 + wv, Kb = (int)round_up(wv, 32);
            gemm_nt(R + b0, ldr, fBinv + (int64_t)B * WB * WB, WB, nullptr, 0, X + b0, ldr, nrhs, wv, Kb, 0);
            if (b1 < Ms) gemm_nt(X + b0, ldr, fS + (int64_t)b1 * fld + b0, fld, R + b1, ldr, R + b1, ldr, nrhs, rowlim(Ms, b1) - b1, Kb, 1);
        }
        if (!Lt) return;
        for (int B = nB - 1; B >= 0; --B) {
            const int b0 = B * WB, wv = std::min(WB, Ms - b0), Kb = (int)round_up(wv, 32);
            gemm_nt(X + b0, ldr, fBinvT + (int64_t)B * WB * WB, WB, nullptr, 0, R + b0, ldr, nrhs, wv, Kb, 0);
            // rows of L in this block reach back `band` columns at most
            const int c0 = fband > 0 ? std::max(0, (b0 - fband) / 64 * 64) : 0;
            if (b0 > 0) gemm_nt(R + b0, ldr, Lt + (int64_t)c0 * fld + b0, fld, X + c0, ldr, X + c0, ldr, nrhs, b0 - c0, Kb, 1);
        }
    }
    // out[i] = sum_j Ah_ij^2 thinv_j   (sparse patterns only)
    void schur_diag(const double* thinv_dev, double* out_dev) {
        const double* v = sparse_vals(h->d_Ah);
        hipLaunchKernelGGL(k_ipm_sdiag_csr, dim3((unsigned)((h->M + 255) / 256)), dim3(256), 0, h->stream, h->d_sp_ptr, h->d_sp_col, v, thinv_dev, out_dev, h->M);
    }
    // transposed copy of Ah and its chunk flags (column form of the restoration-phase Newton system), once per LP
    void ensure_AhT() {
        if (h->ahT_valid) return;
        hipLaunchKernelGGL(k_transpose_dense, dim3((unsigned)((h->n + 63) / 64), (unsigned)((h->M + 63) / 64)), dim3(256), 0, h->stream, h->d_Ah,
                           h->ldn, h->M, h->n, h->d_AhT, h->ldT, (int64_t)-1);
        h->nzT_valid = false;
        const int nch = (int)(h->ldT / ASM_KC);
        if (!h->dense_fast && nch <= ASM_MAXCHUNKS && h->nnz * 8 <= h->M * h->n) {
            const int T = pick_tile(h->n), TS = 32 * T;
            const int nt = (int)((h->n + TS - 1) / TS);
            hipLaunchKernelGGL(k_tile_nzflags, dim3((unsigned)nt, (unsigned)((nch + 7) / 8)), dim3(256), 0, h->stream, h->d_AhT, h->ldT, h->n, TS, nch,
                               h->d_nzT, nch, (const int*)nullptr);
            h->nzT_fraction = executed_fraction(h->d_nzT, nt, nch);
            h->nzT_valid = true;
        }
        h->ahT_valid = true;
    }
    // S[0:n,0:n] (lower) = AhT diag(dinv) AhT' + diag(th)
    void syrk_col(const double* dinv_dev, const double* th_dev) {
        set_main_band(0);
        ensure_AhT();
        const int n = (int)h->n;
        const int nch = (int)(h->ldT / ASM_KC);
        const double frac = h->nzT_valid ? h->nzT_fraction : 1.0;
        int id = begin(ASM_K_SYRK, frac * (double)n * (n + 1) * h->ldT, 8.0 * (n * (double)h->ldT + 0.5 * n * (double)n));
        launch_syrk(pick_tile(n), h->d_AhT, h->ldT, nullptr, 0, n, (int)h->ldT, dinv_dev, th_dev, h->d_S, h->Mp, 0, 0, -1,
                    h->nzT_valid ? h->d_nzT : nullptr, nch, frac);
        end(id);
    }
    // per (row tile, k-chunk) non-zero flags of Ah for the chunk-skipping Schur build (sparse Jacobians only)
    void tile_flags() {
        h->nz_valid = false;
        if (h->M == 0) return;
        const int nch = (int)(h->ldn / ASM_KC);
        if (h->dense_fast || nch > ASM_MAXCHUNKS || h->nnz * 8 > h->M * h->n) return;     // dense pattern: nothing to skip
        if (h->row_band > 0) return;      // banded row order: the full-row Schur matrix is built from its structural entries, not by k_syrk
        h->nz_T = pick_tile(h->M);
        const int TS = 32 * h->nz_T;
        const int nt = (int)((h->M + TS - 1) / TS);
        h->nz_pitch = nch;
        hipLaunchKernelGGL(k_tile_nzflags, dim3((unsigned)nt, (unsigned)((nch + 7) / 8)), dim3(256), 0, h->stream, h->d_Ah, h->ldn, h->M, TS,
                           nch, h->d_nz, h->nz_pitch, (const int*)nullptr);
        h->nz_valid = true;
        h->nz_fraction = executed_fraction(h->d_nz, nt, nch, 0);
    }
    // fraction of (tile pair, chunk) products actually executed: keeps the flop accounting of the roofline honest
    double executed_fraction(const unsigned char* d_flags, int nt, int nch, int cache_slot = -1) {
        if (h->timing == 0) return h->nz_valid ? h->nz_fraction : 1.0;      // only the flop accounting needs it: no read-back when timing is off
        // the flags follow the fixed pattern of the Jacobian: for the row sets that recur every LP (all rows; the equality rows of the
        // null-space form) the share is computed once per handle - the read-back and the O(nt^2 nch) host loop cost milliseconds per LP
        if (cache_slot >= 0 && h->nz_frac_cache[cache_slot] >= 0.0) return h->nz_frac_cache[cache_slot];
        const double fr_ = executed_fraction_now(d_flags, nt, nch);
        if (cache_slot >= 0) h->nz_frac_cache[cache_slot] = fr_;
        return fr_;
    }
    double executed_fraction_now(const unsigned char* d_flags, int nt, int nch) {
        std::vector<unsigned char> fl((size_t)nt * nch);
        HIPCHK(hipMemcpyAsync(fl.data(), d_flags, fl.size(), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        double act = 0.0, tot = 0.0;
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b <= a; ++b) {
                const unsigned char *fa = &fl[(size_t)a * nch], *fb = &fl[(size_t)b * nch];
                int cnt = 0;
                for (int c = 0; c < nch; ++c) cnt += fa[c] & fb[c];
                act += cnt;
                tot += nch;
            }
        return tot > 0 ? act / tot : 1.0;
    }
    void chol_solve_dev(const double* rhs_dev, double* out_dev, int Ms) {
        // the substitution runs in place in the caller's output buffer (w), z in d_vecM
        if (fsmall && Ms <= ASM_SMALL_USE) {       // small systems: one workgroup, factor + inverses of its 64-wide diagonal blocks
            hipLaunchKernelGGL(k_small_solve, dim3(1), dim3(1024), 0, h->stream, (const double*)fS, fld, (const double*)fLinv, Ms, rhs_dev, out_dev);
            return;
        }
        // a system of one wide block only READS its right-hand side (forward diagonal product); with more blocks the panel updates work in place
        const bool one_block = Ms <= fwb;
        if (out_dev != rhs_dev && !one_block) HIPCHK(hipMemcpyAsync(out_dev, rhs_dev, Ms * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        int id = begin(ASM_K_TRSV, 2.0 * Ms * (double)Ms, 8.0 * Ms * (double)Ms);
        solve_w = out_dev;
        solve_src = one_block ? rhs_dev : nullptr;
        solve_launches(Ms);
        solve_src = nullptr;
        solve_w = h->d_vecM2;
        end(id);
    }

    static int pick_tile(int64_t Ms) { return Ms >= 3072 ? 4 : (Ms >= 768 ? 2 : 1); }

    void launch_syrk(int T, const double* A, int64_t ld, const int* idx, int64_t row0, int Ms, int K, const double* theta,
                     const double* diag, double* S, int64_t ldS, int64_t srow0, int mode, int MsB = -1,
                     const unsigned char* nz = nullptr, int nzpitch = 0, double nzfrac = -1.0, int nsplit = 1, int64_t ksplit = 0) {
        int TS = 32 * T;
        int64_t nt = (Ms + TS - 1) / TS;
        int ntj = 0;
        int64_t blocks = (nt * (nt + 1) / 2 + 7) / 8 * 8;      // triangular: eight equal runs of the tile curve, one per XCD label (tri_tile_xcd)
        if (MsB >= 0) {                       // rectangular: all row tiles x the column tiles covering MsB columns
            ntj = (int)((MsB + TS - 1) / TS);
            blocks = nt * ntj;
        } else {
            MsB = Ms;
        }
        if (blocks <= 0) return;
        if (nz && ntj == 0) blocks = (nt * (nt + 1) / 2 + 511) / 512 * 512;      // sparse build: runs of 64 tile pairs dealt to the XCD labels (tri_tile_xcd)
        // per-launch timing of the MFMA kernel itself (algorithmic flops: Ms*MsB*K over the stored triangle/rectangle)
        double fl = (MsB == Ms && ntj == 0) ? (double)Ms * (Ms + 1) * K : 2.0 * ((double)Ms * MsB - 0.5 * (double)MsB * MsB) * K;
        if (nz) fl *= nzfrac >= 0.0 ? nzfrac : h->nz_fraction;
        // timed on the stream it is launched on (HIP events see only their own stream)
        int kid = begin(ASM_K_SYRK_KERNEL, fl, 8.0 * ((double)(Ms + MsB) * K + (double)Ms * MsB), cur);
        struct EndGuard { Dev* d; int id; ~EndGuard() { d->end(id); } } guard_{this, kid};
        // ASM_SYRK_UPD=0: the Cholesky updates through the generic kernel (ablation: what the dedicated kernel is worth)
        static const bool use_upd = [] { const char* v = std::getenv("ASM_SYRK_UPD"); return !(v && v[0] == '0'); }();
        if (T == 4 && mode == 1 && !nz && !idx && !theta && K % (2 * ASM_UPD_KC) == 0 && use_upd)      // Cholesky updates: their own kernel
            hipLaunchKernelGGL(k_syrk_upd, dim3((unsigned)blocks), dim3(256), 0, cur, A, ld, row0, Ms, K, S, ldS, srow0, MsB, ntj);
        else if (T == 4 && mode == 1 && !nz && K % 16 == 0)      // 16-wide k-chunks, two workgroups per CU
            hipLaunchKernelGGL((k_syrk<4, 8, 16, 4>), dim3((unsigned)blocks, (unsigned)nsplit), dim3(512), 0, cur, A, ld, idx, row0, Ms, K, theta, diag, S,
                               ldS, srow0, mode, MsB, ntj, nz, nzpitch, ksplit);
        else if (T == 4)
            hipLaunchKernelGGL((k_syrk<4, 8, 32, 2>), dim3((unsigned)blocks, (unsigned)nsplit), dim3(512), 0, cur, A, ld, idx, row0, Ms, K, theta, diag, S,
                               ldS, srow0, mode, MsB, ntj, nz, nzpitch, ksplit);
        else if (T == 2)
            hipLaunchKernelGGL((k_syrk<2, 4, 32, 1>), dim3((unsigned)blocks, (unsigned)nsplit), dim3(256), 0, cur, A, ld, idx, row0, Ms, K, theta, diag, S,
                               ldS, srow0, mode, MsB, ntj, nz, nzpitch, ksplit);
        else
            hipLaunchKernelGGL((k_syrk<1, 4, 32, 1>), dim3((unsigned)blocks, (unsigned)nsplit), dim3(256), 0, cur, A, ld, idx, row0, Ms, K, theta, diag, S,
                               ldS, srow0, mode, MsB, ntj, nz, nzpitch, ksplit);
    }

    // S[0:Ms,0:Ms] (lower) = Ah[idx,:] diag(theta) Ah[idx,:]' + diag     idx == nullptr -> identity
    void syrk(const int* idx_host, int Ms, const double* theta, const double* diag) {
        set_main_band(0);
        h2d(h->d_theta, theta, h->n, h->ldn);
        if (diag) h2d(h->d_diag, diag, Ms, Ms);
        if (idx_host) {
            HIPCHK(hipMemcpyAsync(h->d_idx, idx_host, Ms * sizeof(int), hipMemcpyHostToDevice, h->stream));
            h2d_done(h);
        }
        // sparse Jacobians: chunk flags of the gathered row set (second half of the flag buffer), as in the full Schur build
        const int nch = (int)(h->ldn / ASM_KC);
        const int T = pick_tile(Ms), TS = 32 * T;
        const int nt = (Ms + TS - 1) / TS;
        const bool skip = h->nz_valid && idx_host != nullptr && nt > 0;
        unsigned char* nz2 = h->d_nz + h->nz_half;
        if (skip)
            hipLaunchKernelGGL(k_tile_nzflags, dim3((unsigned)nt, (unsigned)((nch + 7) / 8)), dim3(256), 0, h->stream, h->d_Ah, h->ldn, (int64_t)Ms, TS,
                               nch, nz2, nch, (const int*)h->d_idx);
        const double frac = skip ? executed_fraction(nz2, nt, nch) : 1.0;
        int id = begin(ASM_K_SYRK, frac * (double)Ms * (Ms + 1) * h->ldn, 8.0 * (Ms * (double)h->ldn + 0.5 * Ms * (double)Ms));
        launch_syrk(T, h->d_Ah, h->ldn, idx_host ? h->d_idx : nullptr, 0, Ms, (int)h->ldn, h->d_theta,
                    diag ? h->d_diag : nullptr, h->d_S, h->Mp, 0, 0, -1, skip ? nz2 : nullptr, nch, frac);
        end(id);
    }
    void diag_prepare(int Ms, int mode, double rel, double absv) {
        hipLaunchKernelGGL(k_diag_prepare, dim3(1), dim3(1024), 0, h->stream, fS, fld, Ms, h->d_diag0, mode, rel, absv);
    }
    template <int WB>
    void trtri_launches(int Ms) {
        constexpr int WSUB = WB / ASM_NB;
        const unsigned nW = (unsigned)((Ms + WB - 1) / WB);
        hipLaunchKernelGGL((k_trtri_init<WB>), dim3(nW, WSUB * WSUB), dim3(256), 0, h->stream, fLinv, Ms, fBinv);
        for (int hh = 1; hh < WSUB; hh *= 2)
            for (int stage = 0; stage < 2; ++stage)
                hipLaunchKernelGGL((k_trtri_level<WB>), dim3(nW, (unsigned)(WSUB / (2 * hh)), (unsigned)(hh * hh)), dim3(256), 0, h->stream,
                                   fS, fld, Ms, fBinv, fBinvT, hh, stage);
        hipLaunchKernelGGL((k_transpose_wb<WB>), dim3(nW, WSUB * WSUB), dim3(256), 0, h->stream, fBinv, fBinvT);
    }
    // want_inverse = false: the caller only solves against the factor and the factor is a "small" one (one-workgroup solves): the
    // explicit inverses of the wide blocks are not built
    bool panel_inv_now = false;      // the panel launches of the factorisation in progress also make the explicit inverse
    void chol(int Ms, double thr = 1e-14, bool want_inverse = true) {
        if (Ms <= 0) return;
        // a banded factor (band b) costs about Ms (b + 64)^2 flops and touches Ms (b + 64) entries, not Ms^3 / 3 and Ms^2 / 2
        const double bw = fband > 0 ? (double)std::min<int64_t>(Ms, (int64_t)fband + 64) : (double)Ms;
        int id = begin(ASM_K_CHOL, fband > 0 ? (double)Ms * bw * bw : (double)Ms * Ms * Ms / 3.0, 8.0 * 1.5 * Ms * bw);
        const bool skip_inv = !want_inverse && fsmall && Ms <= ASM_SMALL_USE;
        // a factor of ONE wide block gets its explicit inverse inside the panel launches (helper workgroups of k_chol_panel_inv)
        static const bool inv_env = [] { const char* v = std::getenv("ASM_PANEL_INV"); return !(v && v[0] == '0'); }();
        panel_inv_now = !skip_inv && inv_env && h->fused_panel && Ms <= fwb && fband == 0 && fBinv && fBinvT && ASM_PNL_WT * (ASM_PNL_NS + 1) <= h->panel_wgs;
        const bool inv_done = panel_inv_now;
        chol_launches(Ms, thr);
        panel_inv_now = false;
        if (skip_inv || inv_done) {
            end(id);
            h->stats.nfact += 1;
            return;
        }
        // explicit inverses of the wide diagonal blocks by divide and conquer over the 64-wide sub-blocks: diagonal
        // blocks from k_potrf_diag, then log2 levels of two launches each (the scratch T uses the buffer of the
        // transposed copy, which is written afterwards)
        if (fwb == 1024) trtri_launches<1024>(Ms); else trtri_launches<512>(Ms);
        end(id);
        h->stats.nfact += 1;
    }
    // block chain of one outer panel [K0, K1): 64-wide potrf / panel solve steps whose rank-64 updates stay inside a
    // 512-wide inner panel; the rest of the outer panel is updated once per inner panel with K = 512
    void chol_chain(int Ms, double thr, int K0, int K1, bool beside_updates = false) {
        for (int I0 = K0, Inext = K0; I0 < K1; I0 = Inext) {
            // a remainder of at most two 64-wide steps joins the last inner panel (k = 519: one launch of nine steps instead of a panel launch, an
            // in-panel update and a second panel launch for the last seven columns)
            const int I1 = (K1 - I0 <= CHOL_NBI + 2 * ASM_NB) ? K1 : std::min(I0 + CHOL_NBI, K1);
            Inext = I1;
            const int Mi = rowlim(Ms, I1);           // banded factor: the rows below are out of this inner panel's reach
            if (h->fused_panel) {
                // the <= 8 steps of this inner panel in one dataflow launch (k_chol_panel): row tiles are owned by workgroups,
                // diagonal-block factors and the panel tiles other workgroups need travel through release / acquire flags
                const int nrt = (Mi - I0 + ASM_NB - 1) / ASM_NB;
                // grid: one workgroup per row tile while they all fit (77 KB of LDS: two per CU); measured: fewer workgroups with several
                // tiles each lengthen every step (M = 11192: 16.7 ms with one tile per workgroup, 21.1 ms with three)
                const int G = std::max(1, std::min(nrt, h->panel_wgs));
                h->panel_epoch += 1;              // flags are "set" when they hold this launch's epoch: no reset between launches
                if (h->panel_epoch == 0) h->panel_epoch = 1;
                // algorithmic flops of the launch: per 64-wide step the factor + inverse of the diagonal block, the panel solve of the rows below
                // and the rank-64 update of the panel's remaining columns
                double pfl = 0.0;
                for (int k0 = I0; k0 < std::min(I1, Ms); k0 += ASM_NB) {
                    const double r = std::max(0, Mi - (k0 + ASM_NB)), w = std::max(0, std::min(I1, Mi) - (k0 + ASM_NB));
                    pfl += 2.0 / 3.0 * ASM_NB * ASM_NB * ASM_NB + 2.0 * r * ASM_NB * ASM_NB + 2.0 * ASM_NB * (r * w - 0.5 * w * w);
                }
                int pid = begin(ASM_K_PANEL_KERNEL, pfl, 8.0 * 2.0 * (double)(Mi - I0) * (double)(std::min(I1, Mi) - I0), cur);
                struct PEnd { Dev* d; int id; ~PEnd() { d->end(id); } } pend_{this, pid};
                // beside the trailing update the register-capped build must be used (its wavefronts have to fit into freed update slots)
                if (panel_inv_now && !beside_updates) {
                    // one wide block: its explicit inverse is made inside the launch by helper workgroups, one per 64 x 64 tile of the block
                    // rows this inner panel finishes (k_chol_panel_inv) - no k_trtri_* launches afterwards
                    const int nst = (std::min(I1, Ms) - I0 + ASM_NB - 1) / ASM_NB, T = (Ms + ASM_NB - 1) / ASM_NB;
                    asmb::launch_resident(k_chol_panel_inv, dim3((unsigned)(G + nst * T)), dim3(256), 0, cur, fS, fld, I0, std::min(I1, Ms), Mi, (const double*)h->d_diag0, thr,
                                       fLinv, h->d_pflags, h->d_ptmo, h->panel_epoch, fBinv, fBinvT, fwb, G);
                } else if (beside_updates)
                    asmb::launch_resident(k_chol_panel, dim3((unsigned)G), dim3(256), 0, cur, fS, fld, I0, std::min(I1, Ms), Mi, (const double*)h->d_diag0, thr,
                                       fLinv, h->d_pflags, h->d_ptmo, h->panel_epoch);
                else
                    asmb::launch_resident(k_chol_panel_solo, dim3((unsigned)G), dim3(256), 0, cur, fS, fld, I0, std::min(I1, Ms), Mi, (const double*)h->d_diag0, thr,
                                       fLinv, h->d_pflags, h->d_ptmo, h->panel_epoch);
            } else
            for (int k0 = I0; k0 < I1; k0 += ASM_NB) {
                int nb = std::min(ASM_NB, Mi - k0);
                hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), 0, cur, fS, fld, k0, nb, h->d_diag0, thr, fLinv);
                int k1 = k0 + nb;
                if (k1 < Mi) {
                    int rem = Mi - k1;
                    hipLaunchKernelGGL(k_trsm_panel, dim3((unsigned)((rem + 63) / 64)), dim3(256), 0, cur, fS, fld, k0, nb, Mi, fLinv);
                    if (k1 < I1)   // update the remaining columns of this inner panel only (rank 64: dedicated 64 x 64-tile kernel)
                        hipLaunchKernelGGL(k_panel_update64, dim3((unsigned)((rem + 63) / 64), (unsigned)((std::min(I1, Mi) - k1 + 63) / 64)), dim3(256), 0,
                                           cur, fS, fld, k0, k1, std::min(I1, Mi), Mi);
                }
            }
            if (I1 < K1 && I1 < Mi) {
                int rem = Mi - I1;
                launch_syrk(pick_tile(rem), fS + I0, fld, nullptr, I1, rem, I1 - I0, nullptr, nullptr, fS, fld, I1, 1, std::min(K1 - I1, rem));
            }
        }
    }
    // Banded factor, every inner panel with its whole trailing update in ONE launch (k_chol_panel_band): no rank-K launches between the panels.
    // Possible when a panel and the band's reach fit ASM_PNL_NRT row tiles and the workgroups (one per row tile + one per trailing tile) are
    // all resident; a batch slot keeps the launch sequence (its panel launches are merged across scenarios).
    bool band_panels_ok(int Ms) const {
        static const bool env = [] { const char* v = std::getenv("ASM_BAND_PANELS"); return !(v && v[0] == '0'); }();
        if (!env || fband <= 0 || !h->fused_panel || h->batch_slot || Ms <= CHOL_NBI + 2 * ASM_NB) return false;
        const int nrt = (CHOL_NBI + 2 * ASM_NB + (int)round_up(fband, 64) + ASM_NB - 1) / ASM_NB + 1;
        const int m = nrt - CHOL_NBI / ASM_NB;
        return nrt <= ASM_PNL_NRT && nrt + m * (m + 1) / 2 <= h->panel_wgs;
    }
    void chol_launches_band(int Ms, double thr) {
        cur = h->stream;
        for (int I0 = 0, I1 = 0; I0 < Ms; I0 = I1) {
            I1 = (Ms - I0 <= CHOL_NBI + 2 * ASM_NB) ? Ms : I0 + CHOL_NBI;
            const int Mi = rowlim(Ms, I1);
            const int nrt = (Mi - I0 + ASM_NB - 1) / ASM_NB, nst = (I1 - I0 + ASM_NB - 1) / ASM_NB, m = nrt - nst;
            h->panel_epoch += 1;
            if (h->panel_epoch == 0) h->panel_epoch = 1;
            double pfl = 0.0;
            for (int k0 = I0; k0 < I1; k0 += ASM_NB) {
                const double r = std::max(0, Mi - (k0 + ASM_NB));
                pfl += 2.0 / 3.0 * ASM_NB * ASM_NB * ASM_NB + 2.0 * r * ASM_NB * ASM_NB + 2.0 * ASM_NB * (0.5 * r * r);      // factor, panel solve, update of everything in reach
            }
            int pid = begin(ASM_K_PANEL_KERNEL, pfl, 8.0 * 2.0 * (double)(Mi - I0) * (double)(Mi - I0) * 0.5, cur);
            const int nhelp = (m * (m + 1) / 2 + ASM_BAND_TPH - 1) / ASM_BAND_TPH;      // helper workgroups: one per trailing tile
            asmb::launch_resident(k_chol_panel_band, dim3((unsigned)(nrt + nhelp)), dim3(256), 0, cur, fS, fld, I0, I1, Mi, (const double*)h->d_diag0, thr,
                                  fLinv, h->d_pflags, h->d_ptmo, h->panel_epoch, nrt);
            end(pid);
        }
    }
    void chol_launches(int Ms, double thr) {
        if (band_panels_ok(Ms)) { chol_launches_band(Ms, thr); return; }
        // Two-level right-looking blocking with look-ahead.  64-wide steps inside a 1024-wide outer panel touch only the
        // panel's own columns; the trailing matrix is read-modify-written once per outer panel (K = 1024), in two parts:
        // (a) the columns of the NEXT outer panel, (b) the rest.  The next panel's serial block chain then runs on a second
        // stream beside (b), so the latency-bound chain hides under the MFMA-bound update.
        const int NBO = CHOL_NBO;
        const int nP = (Ms + NBO - 1) / NBO;
        const bool la = nP > 2 && !h->batch_slot;      // (a batch slot records its launches for ONE stream)
        while ((int)h->la_events.size() < 2 * nP + 2) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->la_events.push_back(e);
        }
        cur = h->stream;
        chol_chain(Ms, thr, 0, std::min(NBO, Ms));
        for (int p = 0; p < nP; ++p) {
            const int K0 = p * NBO, K1 = std::min(K0 + NBO, Ms);
            if (K1 >= Ms) break;
            const int rem = rowlim(Ms, K1) - K1, wa = std::min(NBO, rem);   // next panel = first `wa` trailing columns (banded: `rem` stops where the panel's reach ends)
            hipEvent_t e_a = h->la_events[2 * p], e_c = h->la_events[2 * p + 1];
            cur = h->stream;
            // (a) rows >= K1, columns of the next outer panel
            launch_syrk(pick_tile(rem), fS + K0, fld, nullptr, K1, rem, K1 - K0, nullptr, nullptr, fS, fld, K1, 1, wa);
            if (la) {
                HIPCHK(hipEventRecord(e_a, h->stream));
                HIPCHK(hipStreamWaitEvent(h->stream2, e_a, 0));
                cur = h->stream2;
                chol_chain(Ms, thr, K1, std::min(K1 + NBO, Ms), true);     // next panel's chain beside (b)
                HIPCHK(hipEventRecord(e_c, h->stream2));
                cur = h->stream;
            }
            // (b) the rest of the trailing matrix
            const int rem2 = rem - wa;
            if (rem2 > 0)
                launch_syrk(pick_tile(rem2), fS + K0, fld, nullptr, K1 + wa, rem2, K1 - K0, nullptr, nullptr, fS, fld, K1 + wa, 1);
            if (la) HIPCHK(hipStreamWaitEvent(h->stream, e_c, 0));
            else chol_chain(Ms, thr, K1, std::min(K1 + NBO, Ms));
        }
        cur = h->stream;
    }
    // out = (L L')^-1 rhs   (compact vectors of length Ms)
    void chol_solve(const double* rhs, double* out, int Ms) {
        h2d(h->d_vecM2, rhs, Ms, Ms);
        int id = begin(ASM_K_TRSV, 2.0 * Ms * (double)Ms, 8.0 * Ms * (double)Ms);
        solve_launches(Ms);
        end(id);
        d2h(out, h->d_vecM2, Ms);
    }
    void solve_launches(int Ms) {
        if (fwb == 1024) solve_launches_wb<1024>(Ms); else solve_launches_wb<512>(Ms);
    }
    template <int WB>
    void solve_launches_wb(int Ms) {
        // forward: w = copy of rhs (d_vecM2, updated in place), z -> d_vecM ; backward: x -> d_vecM2 (w is dead by then)
        double* w = solve_w ? solve_w : h->d_vecM2;
        double* z = h->d_vecM;
        const int nB = (Ms + WB - 1) / WB;
        for (int B = 0; B < nB; ++B) {
            int b1 = std::min((B + 1) * WB, Ms);
            hipLaunchKernelGGL((k_wtrsv_fwd_diag<WB>), dim3(WB / 4), dim3(256), 0, h->stream, fBinv, B, Ms, (const double*)(solve_src ? solve_src : w), z);
            const int Me = rowlim(Ms, b1);
            int rem = Me - b1;
            if (rem > 0)
                hipLaunchKernelGGL((k_wtrsv_fwd_panel<WB>), dim3((unsigned)((rem + 4 * ASM_FWD_RPW - 1) / (4 * ASM_FWD_RPW))), dim3(256), 0, h->stream, fS, fld, B, Me, z, w);
        }
        for (int B = nB - 1; B >= 0; --B) {
            int b1 = std::min((B + 1) * WB, Ms);
            const int Me = rowlim(Ms, b1);
            int rem = Me - b1;
            int np = 0;
            if (rem > 0) {
                np = (rem + ASM_WBROWS - 1) / ASM_WBROWS;
                hipLaunchKernelGGL((k_wtrsv_bwd_panel<WB>), dim3((unsigned)np), dim3(256), 0, h->stream, fS, fld, B, Me, w, h->d_wpart);
            }
            if (np > 0) {
                hipLaunchKernelGGL((k_wtrsv_bwd_reduce<WB>), dim3(WB / ASM_NB), dim3(256), 0, h->stream, B, Ms, z, h->d_wpart, np, h->d_wt);
                hipLaunchKernelGGL((k_wtrsv_bwd_diag<WB>), dim3(WB / 4), dim3(256), 0, h->stream, fBinvT, B, Ms, h->d_wt, w, WB);
            } else {
                // last wide block (the only one of a small system): nothing to subtract, the diagonal product reads z itself
                hipLaunchKernelGGL((k_wtrsv_bwd_diag<WB>), dim3(WB / 4), dim3(256), 0, h->stream, fBinvT, B, Ms, (const double*)(z + (int64_t)B * WB), w, Ms - B * WB);
            }
        }
    }
    void assemble() {
        if (h->J_valid) return;                 // J in HBM already matches dE (one assembly per evaluation, shared by the norms and the LP)
        h->J_valid = true;
        h->spv_J_valid = false;
        int id = begin(ASM_K_ASSEMBLE, 0.0, 8.0 * h->nnz + 8.0 * h->nu + (h->dense_fast ? 0.0 : 24.0 * h->nu + 8.0 * h->nnz));
        if (h->dense_fast) {
            int64_t total = h->m * h->n;
            unsigned g = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
            hipLaunchKernelGGL(k_assemble_dense, dim3(g), dim3(256), 0, h->stream, h->d_dE, h->d_J, h->m, h->n, h->ldn);
        } else if (h->nu > 0) {
            unsigned g = (unsigned)std::min<int64_t>((h->nu + 255) / 256, 4096);
            hipLaunchKernelGGL(k_assemble, dim3(g), dim3(256), 0, h->stream, h->d_dE, h->d_perm, h->d_ustart, h->d_uoff,
                               h->d_adjoff, h->d_J, h->nu);
        }
        end(id);
    }
    // rel[j] = max_i |J_ij| / max_k |J_ik|   (host, n) - matrix-based cap of the column scale
    void col_relmax(double* rel) {
        if (h->M == 0) { std::fill(rel, rel + h->n, 0.0); return; }
        int64_t R = std::min<int64_t>((h->M + 31) / 32, ASM_TMAXCHUNKS);
        int64_t chunk = (h->M + R - 1) / R;
        R = (h->M + chunk - 1) / chunk;
        if (h->sp_ok) {                      // sparse pattern: the same maxima over the stored entries only
            const double* vJ = sparse_vals(h->d_J);
            int id = begin(ASM_K_SCALE, 0.0, 8.0 * 3.0 * h->sp_nnz);
            hipLaunchKernelGGL(k_sp_row_absmax, dim3((unsigned)((h->M + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_sp_ptr, vJ, h->d_rho, h->M);
            hipLaunchKernelGGL(k_sp_col_relmax, dim3((unsigned)((h->ldn + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_sc_ptr, (const int*)h->d_sc_row,
                               (const int*)h->d_sc_pos, vJ, (const double*)h->d_rho, h->d_vecN, h->n, h->ldn);
            end(id);
            d2h(rel, h->d_vecN, h->n);
            return;
        }
        int id = begin(ASM_K_SCALE, 0.0, 8.0 * 2.0 * h->M * h->ldn);
        hipLaunchKernelGGL(k_row_absmax, dim3((unsigned)((h->M + 3) / 4)), dim3(256), 0, h->stream, h->d_J, h->ldn, h->d_rho, h->M, h->ldn);
        hipLaunchKernelGGL(k_col_relmax_stage1, dim3((unsigned)((h->ldn + 255) / 256), (unsigned)R), dim3(256), 0, h->stream, h->d_J,
                           h->ldn, h->d_rho, h->d_partial, h->M, h->ldn, chunk);
        hipLaunchKernelGGL(k_col_relmax_stage2, dim3((unsigned)((h->ldn + 255) / 256)), dim3(256), 0, h->stream, h->d_partial, h->d_vecN, R,
                           h->ldn);
        end(id);
        d2h(rel, h->d_vecN, h->n);
    }
    // Ah = diag(1/rho) J diag(c);  rho (host, M)
    void scale(const double* c, double* rho) {
        h->spv_Ah_valid = false;
        h->ahT_valid = false;
        h->ahTg_valid = false;
        if (h->M == 0) return;
        h2d(h->d_c, c, h->n, h->ldn);
        if (h->sp_ok) {
            const double* vJ = sparse_vals(h->d_J);
            int id = begin(ASM_K_SCALE, 0.0, 8.0 * 4.0 * h->sp_nnz);
            hipLaunchKernelGGL(k_sp_scale_rows, dim3((unsigned)((h->M + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_sp_ptr, (const int*)h->d_sp_col,
                               (const int64_t*)h->d_sp_off, vJ, (const double*)h->d_c, h->d_Ah, h->d_spv_Ah, h->d_rho, h->M);
            end(id);
            h->spv_Ah_valid = true;
            d2h(rho, h->d_rho, h->M);
            return;
        }
        int id = begin(ASM_K_SCALE, 0.0, 8.0 * 3.0 * h->M * h->ldn);
        hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)h->M), dim3(256), 0, h->stream, h->d_J, h->d_c, h->d_Ah, h->d_rho, h->n,
                           h->ldn);
        end(id);
        d2h(rho, h->d_rho, h->M);
    }
};

// =====================================================================================================
// LP in scaled units (oracle/lp_solver.py: class LP / scale_lp)
// =====================================================================================================
struct SLP {
    int64_t n, M, ns;
    vec q, r, lb, ub, w, slo;
    const int* rtype;
    const int* srow;
    const double* scoef;
    double scale_q;
};

struct Solver {
    // out = A x for a k x ncols matrix of few, long rows (the basis Zt): one workgroup per row when that fills the chip better
    void gemv_rows(const double* A, int64_t ld, const double* x, double* out, int64_t rows, int64_t ncols) {
        if (rows <= 2048 && ncols >= 2048)
            hipLaunchKernelGGL(k_gemv_n_wide, dim3((unsigned)rows), dim3(256), 0, h->stream, A, ld, x, out, rows, ncols);
        else
            hipLaunchKernelGGL(k_gemv_n, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->stream, A, ld, x, out, rows, ncols);
    }
    // workgroups of the interior-point reductions (k_ipm_measures / _steps / _muaff): 1024 elements per workgroup and sweep, at most IPM_RED_MAXWG
    unsigned red_grid() const { return (unsigned)std::min<int64_t>(IPM_RED_MAXWG, std::max<int64_t>(1, (std::max(std::max(lp.n, lp.M), lp.ns) + 4095) / 4096)); }
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    asm_handle* h;
    Dev dev;
    SLP lp;
    vec tmpM, tmpN;

    explicit Solver(asm_handle* hh) : h(hh), dev(hh) {}

    // t = Ah p + E s
    void rowact(const vec& p, const vec& s, vec& t) {
        t.resize(lp.M);
        dev.gemv_n(h->d_Ah, p.data(), t.data());
        for (int64_t k = 0; k < lp.ns; ++k) t[lp.srow[k]] += lp.scoef[k] * s[k];
    }
    void atv(const vec& y, vec& out) {
        out.resize(lp.n);
        dev.gemv_t(h->d_Ah, y.data(), out.data());
    }

    // ---------------------------------------------------------------- interior point (oracle: class IPM)
    struct IpmState {
        vec p, s, g, y, tL, tU, muL, muU, ts, mus, pi, rp, rdp, rds;
        std::vector<char> ineq, free_;
        vec sg;
        int64_t ncomp = 1;
        vec pinf_hist;
        bool stalled = false;
        bool col_ok = false, col_off = false;     // column form available / abandoned for this LP
        int col_iters = 0;
        bool red_ok = false, red_off = false;     // reduced row form (normal phase, large sparse problems)
        int red_iters = 0;
        bool ns_ok = false, ns_off = false, ns_ready = false;   // null-space form (normal phase, many hard equality rows)
        int ns_iters = 0, ns_k = 0;
        bool ns_e_ready = false;  // the component e of the iterate outside pbar + null(A_EF) has been split off
        int iters = 0;
        int status = ASM_OTHER;
        double mu = 0, pinf = 0, dinf = 0, gap = 0, ymax = 0, rpmax = 0;
    } ip;

    // ---- device-resident IPM state (asm_ipm_kernels.hip.h) ---------------------------------------------------
    IpmPtrs P;
    IpmDir dirA, dirC;
    double *d_sres = nullptr, *d_corr = nullptr, *d_tN = nullptr, *d_pcg = nullptr;

    void ipm_bind() {
        double* a = h->d_ipm;
        const int64_t ln = h->ldn, lm = h->Mp, ls = h->nsp;
        auto N = [&]() { double* r_ = a; a += ln; return r_; };
        auto Mv = [&]() { double* r_ = a; a += lm; return r_; };
        auto Sv = [&]() { double* r_ = a; a += ls; return r_; };
        double *q = N(), *lb = N(), *ub = N();
        P.q = q; P.lb = lb; P.ub = ub;
        P.p = N(); P.tL = N(); P.tU = N(); P.muL = N(); P.muU = N(); P.aty = N(); P.rdp = N(); P.thp_inv = N();
        P.hp = N(); P.tmpn = N(); P.rcL = N(); P.rcU = N();
        dirA.dp = N(); dirA.dmuL = N(); dirA.dmuU = N(); dirC.dp = N(); dirC.dmuL = N(); dirC.dmuU = N();
        d_tN = N();
        double* r = Mv();
        P.r = r;
        P.g = Mv(); P.y = Mv(); P.pi = Mv(); P.act = Mv(); P.rp = Mv(); P.dS = Mv(); P.t1 = Mv(); P.rhs = Mv(); P.res = Mv(); P.rcg = Mv();
        dirA.dg = Mv(); dirA.dy = Mv(); dirA.dpi = Mv(); dirC.dg = Mv(); dirC.dy = Mv(); dirC.dpi = Mv();
        d_sres = Mv(); d_corr = Mv(); d_pcg = Mv();
        double *w = Sv(), *slo = Sv(), *scoef = Sv();
        P.w = w; P.slo = slo; P.scoef = scoef;
        P.s = Sv(); P.ts = Sv(); P.mus = Sv(); P.rds = Sv(); P.ths_inv = Sv(); P.hs = Sv(); P.rcs = Sv();
        dirA.ds = Sv(); dirA.dmus = Sv(); dirC.ds = Sv(); dirC.dmus = Sv();
        P.scal = a;
        P.hscal = h->d_hscal; P.hseq = h->d_hseq;
        P.rpart = h->d_redpart; P.rcnt = h->d_redcnt;
        P.rtype = h->d_ipm_i; P.rs0 = h->d_ipm_i + lm; P.rs1 = h->d_ipm_i + 2 * lm; P.srow = h->d_ipm_i + 3 * lm;
    }
    void up(const double* dst, const vec& v) {
        if (!v.empty()) HIPCHK(hipMemcpyAsync((void*)dst, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    void down(vec& v, const double* src, int64_t cnt) {
        v.resize(cnt);
        if (cnt) HIPCHK(hipMemcpyAsync(v.data(), src, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    // push the (scaled) LP data of `lp` to the device
    void ipm_upload_lp() {
        ipm_bind();
        P.n = lp.n; P.M = lp.M; P.ns = lp.ns; P.scale_q = lp.scale_q;
        if (!asmb::in_fiber() && h->h_up) {
            // seven vectors as three copies from pinned staging (q | lb | ub and w | slo | scoef are neighbours in the arena): a copy from a
            // pageable std::vector is staged by the runtime and blocks the host for tens of microseconds each, twice per LP
            const int64_t ln = h->ldn, lm = h->Mp, ls = h->nsp;
            double* st = h->h_up;
            std::memcpy(st, lp.q.data(), lp.n * sizeof(double));
            std::memcpy(st + ln, lp.lb.data(), lp.n * sizeof(double));
            std::memcpy(st + 2 * ln, lp.ub.data(), lp.n * sizeof(double));
            std::memcpy(st + 3 * ln, lp.r.data(), lp.M * sizeof(double));
            HIPCHK(hipMemcpyAsync((void*)P.q, st, 3 * ln * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync((void*)P.r, st + 3 * ln, lp.M * sizeof(double), hipMemcpyHostToDevice, h->stream));
            if (lp.ns) {
                double* ss = st + 3 * ln + lm;
                std::memcpy(ss, lp.w.data(), lp.ns * sizeof(double));
                std::memcpy(ss + ls, lp.slo.data(), lp.ns * sizeof(double));
                std::memcpy(ss + 2 * ls, h->scoef.data(), lp.ns * sizeof(double));
                HIPCHK(hipMemcpyAsync((void*)P.w, ss, 3 * ls * sizeof(double), hipMemcpyHostToDevice, h->stream));
            }
            h2d_done(h);
            return;
        }
        up(P.q, lp.q); up(P.lb, lp.lb); up(P.ub, lp.ub); up(P.r, lp.r);
        up(P.w, lp.w); up(P.slo, lp.slo);
        vec sc(h->scoef.begin(), h->scoef.begin() + lp.ns);
        up(P.scoef, sc);
        h2d_done(h);
    }
    unsigned grid_all() const { return (unsigned)((std::max(std::max(lp.n, lp.M), std::max<int64_t>(lp.ns, 1)) + 255) / 256); }
    // Sequence number for the next publishing kernel (0 = the kernel does not publish: copy path)
    unsigned pub_next() {
        if (!h->spin_read) return 0;
        h->scal_seq += 1;
        if (h->scal_seq == 0) h->scal_seq = 1;
        return h->scal_seq;
    }
    // The scalar block of the last reduction kernel on the host.  pub != 0: that kernel stored the block into host-mapped memory
    // and then the sequence word; spin on the word (the stream is in order: everything before that kernel has finished too).
    void read_scal(unsigned pub) {
        if (pub == 0) {
            HIPCHK(hipMemcpyAsync(h->h_scal, P.scal, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            return;
        }
        if (asmb::in_fiber()) {      // scenario batch: the publishing kernel is recorded; the round that launches it ends before this fiber resumes
            asmb::flush_wait();
            if (__atomic_load_n(h->h_seq, __ATOMIC_ACQUIRE) != pub) throw HipError("read_scal: the batch round ended without the publishing kernel's sequence word");
            return;
        }
        // hot spin for the common case (the kernel is next on an otherwise idle stream: 6.7 us), then yield, then sleep: with several
        // solver threads per process (scenario batches) the kernel may be queued behind other streams' work for milliseconds, and a
        // spinning thread would burn the CPU share the launching threads need
        const double t0 = now_ms();
        for (unsigned long spins = 1;; ++spins) {
            if (__atomic_load_n(h->h_seq, __ATOMIC_ACQUIRE) == pub) return;
            if (spins < 20000) __builtin_ia32_pause();
            else if (spins < 20200) sched_yield();
            else { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
            if (((spins < 20200 && (spins & 0xfffff) == 0) || (spins >= 20200 && (spins & 0x3ff) == 0)) && now_ms() - t0 > 30000.0) {
                HIPCHK(hipStreamSynchronize(h->stream));            // a device fault surfaces here
                if (__atomic_load_n(h->h_seq, __ATOMIC_ACQUIRE) == pub) return;
                throw HipError("read_scal: the publishing kernel finished without setting its sequence word");
            }
        }
    }

    void ipm_init() {
        const int64_t n = lp.n, M = lp.M;
        IpmState fresh;
        ip = fresh;
        int64_t nfree = 0, nineq = 0;
        for (int64_t i = 0; i < M; ++i) nineq += lp.rtype[i] != 0;
        for (int64_t j = 0; j < n; ++j) nfree += lp.ub[j] > lp.lb[j];
        ip.ncomp = std::max<int64_t>(2 * nfree + lp.ns + nineq, 1);
        ip.red_ok = M >= RED_MIN_M && h->sp_ok;
        ip.ns_ok = ns_lp && lp.ns == 0;      // basis made by solve_scaled before the warm attempt
        ip.ns_ready = true;
        ip.ns_k = ns_k;
        ip.col_ok = h->col_capable && lp.ns > 0 && M >= COL_MIN_M && (double)n <= COL_MAX_RATIO * (double)M;   // every row owns a slack (setup)
        ipm_upload_lp();
        P.ncomp = ip.ncomp;
        static const bool origin_env = [] { const char* v = std::getenv("ASM_IPM_ORIGIN_START"); return !(v && v[0] == '0'); }();      // (measurement knob)
        hipLaunchKernelGGL(k_ipm_init_p, dim3(grid_all()), dim3(256), 0, h->stream, P, (origin_env && lp.ns == 0) ? 1 : 0);
        dev.gemv_n_dev(h->d_Ah, P.p, P.act);
        static const double mu_env = [] { const char* v = std::getenv("ASM_IPM_MU0"); return v ? std::atof(v) : IPM_MU0_NORMAL; }();      // (measurement knob)
        hipLaunchKernelGGL(k_ipm_init_rest, dim3(grid_all()), dim3(256), 0, h->stream, P, lp.ns == 0 ? mu_env : 1.0);
    }

    void ipm_measures() {
        dev.gemv_n_dev(h->d_Ah, P.p, P.act);
        dev.gemv_t_dev(h->d_Ah, P.y, P.aty);
        const unsigned pub = pub_next();
        if (ns_live()) {
            // null-space form: the equality rows' multipliers are carried as 0, the dual residual that counts is Z'rdp (oracle: IPM.measures)
            hipLaunchKernelGGL(k_ipm_measures, dim3(red_grid()), dim3(1024), 0, h->stream, P, 0u);
            gemv_rows((const double*)h->d_nsG, h->ns_ldg, (const double*)P.rdp, nsv(12), (int64_t)ip.ns_k, h->ldn);
            hipLaunchKernelGGL(k_ns_dinf, dim3(1), dim3(1024), 0, h->stream, P, (const double*)nsv(12), ip.ns_k, pub);
        } else {
            hipLaunchKernelGGL(k_ipm_measures, dim3(red_grid()), dim3(1024), 0, h->stream, P, pub);
        }
        read_scal(pub);
        ip.pinf = h->h_scal[SC_PINF];
        ip.dinf = h->h_scal[SC_DINF];
        ip.mu = h->h_scal[SC_MU];
        ip.gap = ip.mu / lp.scale_q;
        ip.ymax = h->h_scal[SC_YMAX];
        ip.rpmax = h->h_scal[SC_RPMAX];
    }

    // pull the iterate back to the host (identification, certificates, unpolished fallback)
    void ipm_download() {
        down(ip.p, P.p, lp.n); down(ip.tL, P.tL, lp.n); down(ip.tU, P.tU, lp.n); down(ip.muL, P.muL, lp.n); down(ip.muU, P.muU, lp.n);
        down(ip.g, P.g, lp.M); down(ip.pi, P.pi, lp.M); down(ip.y, P.y, lp.M);
        down(ip.s, P.s, lp.ns); down(ip.ts, P.ts, lp.ns); down(ip.mus, P.mus, lp.ns);
        HIPCHK(hipStreamSynchronize(h->stream));
        ip.free_.resize(lp.n);
        for (int64_t j = 0; j < lp.n; ++j) ip.free_[j] = lp.ub[j] > lp.lb[j];
    }

    // rigorous primal-infeasibility certificate test (oracle: farkas_margin)
    double farkas_margin(const vec& y) {
        double ymax = 0.0;
        for (double v : y) ymax = std::max(ymax, std::fabs(v));
        ymax = std::max(ymax, 1e-300);
        vec yn(lp.M), rho;
        for (int64_t i = 0; i < lp.M; ++i) yn[i] = y[i] / ymax;
        atv(yn, rho);
        double cmax = -1.0, sl = 0.0;
        for (int64_t k = 0; k < lp.ns; ++k) {
            double coef = lp.scoef[k] * yn[lp.srow[k]];
            cmax = std::max(cmax, coef);
            sl += coef * lp.slo[k];
        }
        if (lp.ns && cmax > 1e-12) return -INF;
        double lhs = sl, ynr = 0.0;
        for (int64_t j = 0; j < lp.n; ++j) lhs += std::max(rho[j] * lp.lb[j], rho[j] * lp.ub[j]);
        for (int64_t i = 0; i < lp.M; ++i) ynr += yn[i] * lp.r[i];
        return ynr - lhs;
    }

    // ------------------------------------------------------------ null-space form (oracle: class NullSpace / IPM.run use_ns)
    bool use_ns = false, ns_was_cold = false;
    // ASM_NS_SPLIT=T,n : tile size and number of k slices of the reduced Newton matrix's build (tuning knob; default by size)
    int ns_split_T = [] { const char* v = std::getenv("ASM_NS_SPLIT"); return v ? std::atoi(v) : 0; }();
    int ns_split_n = [] { const char* v = std::getenv("ASM_NS_SPLIT"); const char* c = v ? std::strchr(v, ',') : nullptr; return c ? std::atoi(c + 1) : 0; }();
    bool ns_lp = false;       // this LP has a valid null-space basis (set up before the warm attempt: the active-set solves use it too)
    int ns_k = 0;
    SolveHint* cur_hint = nullptr;
    NsIdx nsX() const { NsIdx X; X.Eidx = h->d_nsEidx; X.Epos = h->d_nsEpos; X.Iidx = h->d_nsIidx; X.Ipos = h->d_nsIpos; X.nE = h->ns_nE; X.nI = h->ns_nI; return X; }
    int ns_read_cnt() {
        int v = 0;
        HIPCHK(hipMemcpyAsync(&v, h->d_nscnt, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return v;
    }
    // buffers sized by the null-space dimension: right-hand-side blocks R, X (k x nEp), Gt = [Zt | GI'] (k x ldg), the k x k factor
    void ns_reserve(int k) {
        if (k <= h->ns_kcap) return;
        if (h->ns_kcap > 0) {
            // the null space grew beyond what the first LP of the form reserved (fewer fixed columns than then): the k-sized buffers are
            // released and re-made; the carried basis goes with them
            if (h->verbose) std::fprintf(stderr, "[asm] null-space form: dimension %d exceeds the reserved %d - buffers re-allocated\n", k, h->ns_kcap);
            HIPCHK(hipStreamSynchronize(h->stream));
            auto drop = [&](void* q) {
                if (!q) return;
                auto it = std::find(h->ns_bufs.begin(), h->ns_bufs.end(), q);
                if (it != h->ns_bufs.end()) h->ns_bufs.erase(it);
                (void)hipFree(q);
            };
            drop(h->d_nsR); drop(h->d_nsX); drop(h->d_nsG); drop(h->d_nsN0); drop(h->d_nsNp); drop(h->d_nsZT); drop(h->d_nsJ); drop(h->d_nsqi); drop(h->d_nsq);
            for (FacBuf* f : {&h->ns_fN, &h->ns_fC}) { drop(f->S); drop(f->Linv); drop(f->Binv); drop(f->BinvT); *f = FacBuf(); }
            h->d_nsR = h->d_nsX = h->d_nsG = h->d_nsN0 = h->d_nsNp = h->d_nsZT = h->d_nsq = nullptr;
            h->d_nsJ = h->d_nsqi = nullptr;
            h->ns_kcap = 0; h->ns_ccap = 0; h->ns_Zk = 0;
        }
        const int cap = (int)round_up(k + k / 4 + 64, 64);
        h->d_nsR = ns_dalloc(h, (int64_t)cap * h->ns_nEp);
        h->d_nsX = ns_dalloc(h, (int64_t)cap * h->ns_nEp);
        h->d_nsG = ns_dalloc(h, (int64_t)cap * h->ns_ldg);
        ns_alloc_factor(h, h->ns_fN, cap);
        h->ns_fN.small = true;
        h->d_nsN0 = ns_dalloc(h, h->ns_fN.ld * h->ns_fN.ld);
        h->d_nsNp = ns_dalloc(h, NS_MAX_SPLIT * h->ns_fN.ld * h->ns_fN.ld);       // split-K slices of the reduced Newton matrix
        h->d_nsZT = ns_dalloc(h, h->ldn * h->ns_fN.ld);        // transposed copy of the basis rows (right operand of the orthonormalisation product)
        int* dj = nullptr;
        dmalloc(&dj, cap);
        h->ns_bufs.push_back((void*)dj);
        h->d_nsJ = dj;
        h->ns_kcap = cap;
        // active-set solves in reduced coordinates: up to 2 cap constraints (an over-determined working set has more than k)
        h->ns_ccap = (int)std::min<int64_t>(2 * cap, h->Mp);
        ns_alloc_factor(h, h->ns_fC, h->ns_ccap);
        h->ns_fC.small = true;
        int* qi = nullptr;
        dmalloc(&qi, (int64_t)h->ns_ccap + h->ldn + h->ns_nIp + 16);
        h->ns_bufs.push_back((void*)qi);
        h->d_nsqi = qi;
        h->d_nsq = ns_dalloc(h, (int64_t)h->ns_ccap * h->ns_fN.ld + 5 * (int64_t)h->ns_ccap + h->ldn + h->Mp + 2 * h->ns_fN.ld + 64);
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    // In place  Zt = L^-1 Zt  with the k x k factor just made in h->ns_fN.  When the factor fits into one wide block the explicit inverse of
    // that block IS L^-1: one transpose + one product on the matrix cores (the rows are n long); otherwise forward substitution per column.
    void ns_ortho(int k) {
        if (k <= h->ns_fN.wb) {
            const int kp = (int)round_up(k, 32);
            hipLaunchKernelGGL(k_transpose_dense, dim3((unsigned)((h->ldn + 63) / 64), (unsigned)((k + 63) / 64)), dim3(256), 0, h->stream, (const double*)h->d_nsG, h->ns_ldg,
                               (int64_t)k, h->ldn, h->d_nsZT, h->ns_fN.ld, (int64_t)-1);
            dev.gemm_nt(h->ns_fN.Binv, h->ns_fN.wb, h->d_nsZT, h->ns_fN.ld, nullptr, 0, h->d_nsG, h->ns_ldg, k, (int)h->ldn, kp, 0);
        } else {
            hipLaunchKernelGGL(k_ns_ortho, dim3((unsigned)((h->ldn + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->ns_fN.S, h->ns_fN.ld, k, h->d_nsG, h->ns_ldg, h->ldn);
        }
    }
    // The k rows in d_nsG (an approximate or an outdated basis) projected onto null(A_EF) of THIS LP and orthonormalised with their own
    // Gram matrix (pivot guard `thr`, absolute), then GI' = (A_I Z)'.  Used as the second pass of ns_basis_from and, with the previous
    // LP's basis, as the whole set-up (oracle: NullSpace.__init__, warm_Z).  The factor of S0 must be current in h->ns_f0.
    bool ns_reproject(int k, double thr) {
        const int nE = h->ns_nE;
        const NsIdx X = nsX();
        const double* vals = dev.sparse_vals(h->d_Ah);
        // second pass (oracle: NullSpace.basis_from): project the rows once more and orthonormalise with their own Gram matrix (~ I) -
        // the columns picked in index order can be badly conditioned, and the active-set solves need A_EF Z = 0 to 1e-13
        hipLaunchKernelGGL(k_ns_rows_e, dim3((unsigned)((h->ns_nEp + 255) / 256), (unsigned)k), dim3(256), 0, h->stream, h->d_sp_ptr, h->d_sp_col, vals, X,
                           (const double*)h->d_nsFm, (const double*)h->d_nsG, h->ns_ldg, h->d_nsR, (int64_t)h->ns_nEp);
        dev.use_factor(h->ns_f0);
        dev.trsm_rows(h->d_nsR, h->d_nsX, h->ns_nEp, k, nE, h->d_nsLt);
        hipLaunchKernelGGL(k_ns_pj, dim3((unsigned)((h->ldn + 255) / 256), (unsigned)k), dim3(256), 0, h->stream, h->d_sc_ptr, h->d_sc_row, h->d_sc_pos, vals, X,
                           (const int*)h->d_nsJ, (const double*)h->d_nsFm, (const double*)h->d_nsR, (int64_t)h->ns_nEp, h->d_nsG, h->ns_ldg, lp.n, h->ldn, 1);
        dev.use_factor(h->ns_fN);
        dev.launch_syrk(Dev::pick_tile(k), h->d_nsG, h->ns_ldg, nullptr, 0, k, (int)h->ldn, nullptr, nullptr, h->ns_fN.S, h->ns_fN.ld, 0, 0);
        hipLaunchKernelGGL(k_ns_fill, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, h->stream, h->d_diag0, 1.0, (int64_t)k);
        dev.chol(k, thr);
        hipLaunchKernelGGL(k_ns_count_big, dim3(1), dim3(1024), 0, h->stream, (const double*)h->ns_fN.S, h->ns_fN.ld, k, NS_BIG, h->d_nscnt);
        const int bad2 = ns_read_cnt();
        dev.use_main();
        if (bad2 > 0) return false;
        ns_ortho(k);
        hipLaunchKernelGGL(k_ns_gi, dim3((unsigned)((h->ns_nIp + 255) / 256), (unsigned)k), dim3(256), 0, h->stream, h->d_sp_ptr, h->d_sp_col, vals, X,
                           (const double*)h->d_nsG, h->ns_ldg, h->d_nsG + h->ldn, h->ns_nIp);
        return true;
    }
    // Orthonormal basis from the columns J of the projector P (oracle: NullSpace.basis_from): W = S0^-1 A_EF[:, J] by block
    // substitution with all k right-hand sides at once, P[J, :] = E_J' - W' A_EF, L_J L_J' = P[J, J] (guard: pivot <= NS_WARM_THR
    // -> not a basis), Zt = L_J^-1 P[J, :], GI' = (A_I Z)'.  The factor of S0 must be current in h->ns_f0.
    bool ns_basis_from(const std::vector<int>& J) {
        const int k = (int)J.size(), nE = h->ns_nE;
        const NsIdx X = nsX();
        const double* vals = dev.sparse_vals(h->d_Ah);
        HIPCHK(hipMemcpyAsync(h->d_nsJ, J.data(), k * sizeof(int), hipMemcpyHostToDevice, h->stream));
        h2d_done(h);
        hipLaunchKernelGGL(k_ns_rhs_cols, dim3((unsigned)k), dim3(256), 0, h->stream, h->d_sc_ptr, h->d_sc_row, h->d_sc_pos, vals, X, (const int*)h->d_nsJ,
                           (const double*)h->d_nsFm, h->d_nsR, (int64_t)h->ns_nEp);
        dev.use_factor(h->ns_f0);
        dev.trsm_rows(h->d_nsR, h->d_nsX, h->ns_nEp, k, nE, h->d_nsLt);
        hipLaunchKernelGGL(k_ns_pj, dim3((unsigned)((h->ldn + 255) / 256), (unsigned)k), dim3(256), 0, h->stream, h->d_sc_ptr, h->d_sc_row, h->d_sc_pos, vals, X,
                           (const int*)h->d_nsJ, (const double*)h->d_nsFm, (const double*)h->d_nsR, (int64_t)h->ns_nEp, h->d_nsG, h->ns_ldg, lp.n, h->ldn, 0);
        dev.use_factor(h->ns_fN);
        hipLaunchKernelGGL(k_ns_gather_t, dim3((unsigned)((k + 255) / 256), (unsigned)k), dim3(256), 0, h->stream, (const double*)h->d_nsG, h->ns_ldg, (const int*)h->d_nsJ, k,
                           h->ns_fN.S, h->ns_fN.ld);
        hipLaunchKernelGGL(k_ns_fill, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, h->stream, h->d_diag0, 1.0, (int64_t)k);
        dev.chol(k, NS_WARM_THR);
        hipLaunchKernelGGL(k_ns_count_big, dim3(1), dim3(1024), 0, h->stream, (const double*)h->ns_fN.S, h->ns_fN.ld, k, NS_BIG, h->d_nscnt);
        const int bad = ns_read_cnt();
        dev.use_main();
        if (bad > 0) return false;
        ns_ortho(k);
        return ns_reproject(k, NS_WARM_THR);
    }
    // Per LP (oracle: NullSpace.__init__): factor S0, null-space dimension, basis columns (retained ones, else a guarded Cholesky of
    // P in index order), orthonormal basis.  False: the LP keeps the row form.
    bool ns_setup() {
        const int nE = h->ns_nE;
        const int64_t n = lp.n;
        const NsIdx X = nsX();
        int64_t nF = 0;
        for (int64_t j = 0; j < n; ++j) nF += lp.ub[j] > lp.lb[j];
        {
            if (!asmb::in_fiber()) {                // (pinned staging: the copy does not go through the runtime's pageable path)
                double* fm = h->h_pin;
                for (int64_t j = 0; j < h->ldn; ++j) fm[j] = (j < n && lp.ub[j] > lp.lb[j]) ? 1.0 : 0.0;
                HIPCHK(hipMemcpyAsync(h->d_nsFm, fm, h->ldn * sizeof(double), hipMemcpyHostToDevice, h->stream));
            } else {
                vec fm(h->ldn, 0.0);
                for (int64_t j = 0; j < n; ++j) fm[j] = lp.ub[j] > lp.lb[j] ? 1.0 : 0.0;
                HIPCHK(hipMemcpyAsync(h->d_nsFm, fm.data(), h->ldn * sizeof(double), hipMemcpyHostToDevice, h->stream));
            }
            h2d_done(h);
        }
        double t_v = now_ms();
        auto vlap = [&](const char* what) {
            if (!h->verbose) return;
            HIPCHK(hipStreamSynchronize(h->stream));
            const double t = now_ms();
            std::fprintf(stderr, "[asm] ns set-up %-10s +%.2f ms\n", what, t - t_v);
            t_v = t;
        };
        dev.use_factor(h->ns_f0);
        if (h->ns_f0.band > 0) {
            // banded S0: the band is cleared (the last factor filled it) and the ~20 structural entries per row are written as merged
            // sparse dot products of the two rows - the dense rank-K build spends 3 ms on the zeros at n = 11 192
            const int64_t wz = std::min<int64_t>(round_up(h->ns_f0.band + 1, 64) + CHOL_NBO + 128, h->ns_f0.ld);      // band + what the blocked factorisation reads beyond it
            hipLaunchKernelGGL(k_ns_zero_band, dim3((unsigned)((wz + 255) / 256), (unsigned)nE), dim3(256), 0, h->stream, h->ns_f0.S, h->ns_f0.ld, nE, (int)wz);
            hipLaunchKernelGGL(k_ns_s0_sparse, dim3((unsigned)((h->ns_npairs + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_nsS0pairs, h->ns_npairs, h->d_sp_ptr,
                               h->d_sp_col, dev.sparse_vals(h->d_Ah), (const int*)h->d_nsEidx, (const double*)h->d_nsFm, h->ns_f0.S, h->ns_f0.ld);
        } else {
            dev.syrk_gathered_into(h->d_nsEidx, nE, h->d_nsFm, h->ns_f0.S, h->ns_f0.ld);
        }
        vlap("S0 build");
        dev.diag_prepare(nE, 1, 0.0, 0.0);
        dev.chol(nE, 1e-10);
        vlap("S0 factor");
        hipLaunchKernelGGL(k_ns_count_big, dim3(1), dim3(1024), 0, h->stream, (const double*)h->ns_f0.S, h->ns_f0.ld, nE, NS_BIG, h->d_nscnt);
        const int dropped = ns_read_cnt();
        dev.use_main();
        const int64_t k = nF - (nE - dropped);
        if (k < 1 || (double)k > 1.5 * NS_MAX_RATIO * (double)lp.M + 8.0) return false;
        ns_reserve((int)k);
        hipLaunchKernelGGL(k_transpose_dense, dim3((unsigned)((nE + 63) / 64), (unsigned)((nE + 63) / 64)), dim3(256), 0, h->stream, (const double*)h->ns_f0.S, h->ns_f0.ld,
                           (int64_t)nE, (int64_t)nE, h->d_nsLt, h->ns_f0.ld, (int64_t)(h->ns_f0.band > 0 ? h->ns_f0.band : nE));
        std::vector<int>& J = cur_hint->ns_J;
        bool have = false;
        ns_was_cold = false;
        vlap("Lt");
        if (h->ns_Zk == (int)k) have = ns_reproject((int)k, NS_ZWARM_THR);      // the previous LP's basis, one projection pass
        vlap("reproject");
        h->ns_Zk = 0;
        if (!have && (int64_t)J.size() == k) {
            bool free_all = true;
            for (int j : J) free_all = free_all && j >= 0 && j < n && lp.ub[j] > lp.lb[j];
            if (free_all) have = ns_basis_from(J);
        }
        if (!have) {
            ns_was_cold = true;
            // cold selection: Y = L0^-1 A_EF for ALL columns (forward substitution only), T = I_F - Y'Y in the main matrix buffer,
            // guarded Cholesky of T in index order with the absolute thresholds NS_SEL_THR in turn until exactly k columns are kept
            if (!h->d_nsYt) h->d_nsYt = ns_dalloc(h, (int64_t)h->ldn * h->ns_nEp + (int64_t)h->ldn * h->ns_nEp);
            double* Yr = h->d_nsYt;                                  // right-hand sides, then garbage
            double* Yt = h->d_nsYt + (int64_t)h->ldn * h->ns_nEp;   // L0^-1 a_j as rows
            const double* vals = dev.sparse_vals(h->d_Ah);
            hipLaunchKernelGGL(k_ns_rhs_cols, dim3((unsigned)n), dim3(256), 0, h->stream, h->d_sc_ptr, h->d_sc_row, h->d_sc_pos, vals, X, (const int*)nullptr,
                               (const double*)h->d_nsFm, Yr, (int64_t)h->ns_nEp);
            dev.use_factor(h->ns_f0);
            dev.trsm_rows(Yr, Yt, h->ns_nEp, (int)n, nE, nullptr);
            dev.use_main();
            std::vector<double> dg(n);
            for (int a = 0; a < 4 && !have; ++a) {
                hipLaunchKernelGGL(k_ns_set_diag, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, h->stream, h->d_S, h->Mp, (int)n, (const double*)h->d_nsFm);
                dev.use_main();
                dev.set_main_band(0);
                dev.launch_syrk(Dev::pick_tile(n), Yt, h->ns_nEp, nullptr, 0, (int)n, h->ns_nEp, nullptr, nullptr, h->d_S, h->Mp, 0, 1);
                hipLaunchKernelGGL(k_ns_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_diag0, 1.0, n);
                dev.chol((int)n, NS_SEL_THR[a]);
                hipLaunchKernelGGL(k_ns_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->d_S, h->Mp, (int)n, h->d_vecN);
                HIPCHK(hipMemcpyAsync(dg.data(), h->d_vecN, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
                std::vector<int> Jc;
                for (int64_t j = 0; j < n; ++j)
                    if (dg[j] < NS_BIG) Jc.push_back((int)j);
                if ((int64_t)Jc.size() != k) continue;
                if (ns_basis_from(Jc)) { J = Jc; have = true; }
            }
        }
        if (!have) { J.clear(); return false; }
        ns_k = (int)k;
        h->ns_Zk = (int)k;
        return true;
    }
    double* nsv(int which) const {      // work vectors: 0..4 n-sized, 5..7 M-sized, 8..9 E-sized, 10..13 k-sized (k <= n)
        if (which < 5) return h->d_nsv + (int64_t)which * h->ldn;
        if (which < 8) return h->d_nsv + 5 * h->ldn + (int64_t)(which - 5) * h->Mp;
        if (which < 10) return h->d_nsv + 5 * h->ldn + 3 * h->Mp + (int64_t)(which - 8) * h->ns_nEp;
        return h->d_nsv + 5 * h->ldn + 3 * h->Mp + 2 * h->ns_nEp + (int64_t)(which - 10) * h->ldn;      // 10..13 k-sized, 14..15 n-sized (14 = e)
    }
    // oracle: ns_applicable
    bool ns_applicable() const {
        if (!h->ns_cap || lp.ns != 0) return false;
        int64_t nE = 0, nF = 0;
        for (int64_t i = 0; i < lp.M; ++i) nE += lp.rtype[i] == 0;
        for (int64_t j = 0; j < lp.n; ++j) nF += lp.ub[j] > lp.lb[j];
        return nE >= NS_MIN_E && (double)(nF - nE) <= NS_MAX_RATIO * (double)lp.M;
    }
    NsEq nsq() const {
        NsEq Q;
        int* qi = h->d_nsqi;
        Q.sel = qi; Q.bpos = qi + h->ns_ccap; Q.rpos = Q.bpos + h->ldn; Q.cnt = Q.rpos + h->ns_nIp;
        double* b = h->d_nsq;
        Q.ldc = h->ns_fN.ld;
        Q.Csel = b; b += (int64_t)h->ns_ccap * Q.ldc;
        Q.d = b; b += h->ns_ccap; Q.lam = b; b += h->ns_ccap; Q.v = b; b += h->ns_ccap; Q.w = b; b += h->ns_ccap; Q.u = b; b += h->ns_ccap;
        Q.pbar = b; b += h->ldn; Q.tbar = b; b += h->Mp; Q.u0 = b; b += Q.ldc; Q.qh = b;
        return Q;
    }
    // dense products with the gathered constraint matrix Csel (nact rows of pitch ldc)
    void nsq_gemv_n(const NsEq& Q, int nact, const double* x, double* out) {
        hipLaunchKernelGGL(k_gemv_n, dim3((unsigned)((nact + 3) / 4)), dim3(256), 0, h->stream, (const double*)Q.Csel, Q.ldc, x, out, (int64_t)nact, Q.ldc);
    }
    void nsq_gemv_t(const NsEq& Q, int nact, const double* y, double* out) {
        int64_t R = std::min<int64_t>((nact + 31) / 32, ASM_TMAXCHUNKS);
        int64_t chunk = (nact + R - 1) / R;
        R = (nact + chunk - 1) / chunk;
        hipLaunchKernelGGL(k_gemv_t_stage1, dim3((unsigned)((Q.ldc + 255) / 256), (unsigned)R), dim3(256), 0, h->stream, (const double*)Q.Csel, Q.ldc, y, h->d_partial, (int64_t)nact, Q.ldc, chunk);
        hipLaunchKernelGGL(k_gemv_t_stage2, dim3((unsigned)((Q.ldc + 255) / 256)), dim3(256), 0, h->stream, h->d_partial, out, R, Q.ldc);
    }
    // per LP (oracle: eqp_ns, the part that does not depend on the working set): pbar, A pbar, u0 = Z'(p_ref - pbar), Z'q
    void ns_lp_vectors() {
        const int k = ns_k, nE = h->ns_nE;
        const int64_t M = lp.M, ldn = h->ldn;
        const NsIdx X = nsX();
        const NsEq Q = nsq();
        const unsigned gM = (unsigned)((M + 255) / 256), gN = (unsigned)((ldn + 255) / 256), gE = (unsigned)((nE + 255) / 256);
        double *pfix = nsv(0), *x = nsv(1), *vz = nsv(2), *yM = nsv(5), *aM = nsv(6), *rE = nsv(8), *tE = nsv(9);
        hipLaunchKernelGGL(k_nseq_pfix, dim3(gN), dim3(256), 0, h->stream, A, pfix, ldn);
        dev.gemv_n_dev(h->d_Ah, pfix, aM);
        hipLaunchKernelGGL(k_nseq_be, dim3(gE), dim3(256), 0, h->stream, A, X, (const double*)aM, rE);
        dev.use_factor(h->ns_f0);
        dev.chol_solve_dev(rE, tE, nE);
        dev.use_main();
        hipLaunchKernelGGL(k_ns_rowvec_e, dim3(gM), dim3(256), 0, h->stream, X, (const double*)tE, yM, M);
        dev.gemv_t_dev(h->d_Ah, yM, x);
        hipLaunchKernelGGL(k_nseq_pbar, dim3(gN), dim3(256), 0, h->stream, A, (const double*)pfix, (const double*)x, (const double*)d_zero, Q.pbar, vz, ldn);
        dev.gemv_n_dev(h->d_Ah, Q.pbar, Q.tbar);
        HIPCHK(hipMemsetAsync(Q.u0, 0, 2 * Q.ldc * sizeof(double), h->stream));      // u0 and qh (contiguous)
        gemv_rows((const double*)h->d_nsG, h->ns_ldg, (const double*)vz, Q.u0, (int64_t)k, ldn);
        gemv_rows((const double*)h->d_nsG, h->ns_ldg, (const double*)A.q, Q.qh, (int64_t)k, ldn);
    }
    // Equality-constrained solve on the working set `cur` in reduced coordinates (oracle: eqp_ns).  Leaves p, y, t = Ah p, tN = Ah' y
    // for the tail kernel like as_solve.  False: more active constraints than the buffers hold (the caller uses as_solve).
    bool as_solve_ns(const AsSets& cur) {
        const int k = ns_k, nE = h->ns_nE;
        const int64_t M = lp.M, n = lp.n, ldn = h->ldn;
        const NsIdx X = nsX();
        const NsEq Q = nsq();
        const unsigned gM = (unsigned)((M + 255) / 256), gN = (unsigned)((ldn + 255) / 256), gE = (unsigned)((nE + 255) / 256);
        hipLaunchKernelGGL(k_nseq_setup, dim3(1), dim3(1024), 0, h->stream, A, cur, X, Q, ldn);
        int cnt[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(cnt, Q.cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        const int nact = cnt[1];
        if (nact > NS_CMAX * k || nact > h->ns_ccap) return false;             // oracle: eqp_ns returns None - the polish attempt ends (eqp_loop)
        const unsigned gC = (unsigned)((Q.ldc + 255) / 256), gA = (unsigned)((nact + 255) / 256);
        double *t1 = nsv(12), *t2 = nsv(13);
        HIPCHK(hipMemcpyAsync(Q.u, Q.u0, Q.ldc * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemsetAsync(Q.lam, 0, (size_t)h->ns_ccap * sizeof(double), h->stream));
        if (nact > 0) {
            hipLaunchKernelGGL(k_nseq_gather, dim3(gC, (unsigned)nact), dim3(256), 0, h->stream, A, cur, X, Q, (const double*)h->d_nsG, h->ns_ldg, k, ldn);
            dev.use_factor(h->ns_fC);
            dev.launch_syrk(Dev::pick_tile(nact), Q.Csel, Q.ldc, nullptr, 0, nact, (int)Q.ldc, nullptr, nullptr, h->ns_fC.S, h->ns_fC.ld, 0, 0);
            dev.diag_prepare(nact, 1, 0.0, 0.0);
            dev.chol(nact, 1e-10, false);
            for (int sw = 0; sw < 3; ++sw) {
                nsq_gemv_n(Q, nact, Q.u, Q.v);                                                                           // C u
                hipLaunchKernelGGL(k_nseq_sub, dim3(gA), dim3(256), 0, h->stream, (const double*)Q.d, (const double*)Q.v, Q.v, (int64_t)nact);
                dev.chol_solve_dev(Q.v, Q.w, nact);
                nsq_gemv_t(Q, nact, Q.w, t1);
                hipLaunchKernelGGL(k_ns_add, dim3(gC), dim3(256), 0, h->stream, (const double*)Q.u, (const double*)t1, Q.u, Q.ldc);
                nsq_gemv_t(Q, nact, Q.lam, t1);                                                                          // C' lam
                hipLaunchKernelGGL(k_nseq_sub, dim3(gC), dim3(256), 0, h->stream, (const double*)Q.qh, (const double*)t1, t2, Q.ldc);
                nsq_gemv_n(Q, nact, t2, Q.v);
                dev.chol_solve_dev(Q.v, Q.w, nact);
                hipLaunchKernelGGL(k_ns_add, dim3(gA), dim3(256), 0, h->stream, (const double*)Q.lam, (const double*)Q.w, Q.lam, (int64_t)nact);
            }
            dev.use_main();
        }
        double *zu = nsv(3), *atw = nsv(4), *wN = nsv(2), *yM = nsv(5), *aM = nsv(6), *rE = nsv(8), *tE = nsv(9);
        ns_gemv_t_dense(Q.u, k, zu);
        hipLaunchKernelGGL(k_nseq_p, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, A, cur, Q, (const double*)zu, ldn);
        hipLaunchKernelGGL(k_nseq_yi, dim3(gM), dim3(256), 0, h->stream, A, X, Q, yM);
        dev.gemv_t_dev(h->d_Ah, yM, atw);
        hipLaunchKernelGGL(k_nseq_w, dim3(gN), dim3(256), 0, h->stream, A, Q, (const double*)atw, wN, ldn);
        dev.gemv_n_dev(h->d_Ah, wN, aM);
        hipLaunchKernelGGL(k_ns_gather_e, dim3(gE), dim3(256), 0, h->stream, X, (const double*)aM, 1.0, rE);
        dev.use_factor(h->ns_f0);
        dev.chol_solve_dev(rE, tE, nE);
        dev.use_main();
        hipLaunchKernelGGL(k_nseq_y, dim3(gM), dim3(256), 0, h->stream, A, X, (const double*)yM, (const double*)tE);
        dev.gemv_n_dev(h->d_Ah, A.p, A.t);
        dev.gemv_t_dev(h->d_Ah, A.y, A.tN);
        h->stats.eqp += 1;
        return true;
    }
    bool ns_live() const { return ip.ns_ok && !ip.ns_off; }
    // least-squares multipliers of the equality rows for the current iterate (oracle: IPM.ns_finish_y); P.rdp is the dual residual of the last
    // measures (with the equality multipliers as they stand: 0, or the values recovered at the end of the previous stage)
    void ns_finish_y() {
        const int nE = h->ns_nE;
        const NsIdx X = nsX();
        const unsigned gE = (unsigned)((nE + 255) / 256);
        double *aM = nsv(6), *rE = nsv(8), *tE = nsv(9);
        dev.gemv_n_dev(h->d_Ah, P.rdp, aM);
        hipLaunchKernelGGL(k_ns_gather_e, dim3(gE), dim3(256), 0, h->stream, X, (const double*)aM, 1.0, rE);
        dev.use_factor(h->ns_f0);
        dev.chol_solve_dev(rE, tE, nE);
        dev.use_main();
        // P.rdp already contains -A_E'y_E of the multipliers recovered at the end of an earlier stage: the solve gives the correction
        hipLaunchKernelGGL(k_ns_scatter_e, dim3(gE), dim3(256), 0, h->stream, X, (const double*)tE, P.y, 1);
    }
    // Per iteration (oracle: IPM.run, use_ns branch): reduced matrix N = Zt Th Zt' + GI' D_I^-1 GI (an unregularised copy is kept for the
    // refinement sweep), its factor, dpbar = A_EF' S0^-1 (-rp_E) and K dpbar (shared by predictor and corrector)
    void ns_iter_setup() {
        const int k = ip.ns_k;
        const int64_t M = lp.M, ldn = h->ldn;
        const NsIdx X = nsX();
        const unsigned gM = (unsigned)((M + 255) / 256), gN = (unsigned)((ldn + 255) / 256);
        double *dpb = nsv(0), *kdpb = nsv(1), *atw = nsv(4), *yM = nsv(5), *aM = nsv(6);
        const double* th = h->d_nsth;
        const double* thI = h->d_nsth + ldn;
        // (theta~ was formed with the interior-point theta: k_ipm_theta_ns in ipm_run)
        dev.use_factor(h->ns_fN);
        int id = dev.begin(ASM_K_SYRK, (double)k * (k + 1) * h->ns_ldg, 8.0 * (k * (double)h->ns_ldg + 0.5 * k * (double)k));
        // the k range (free columns + inequality rows, 19 000 at n = 11 192) is long and the matrix small (k = 519: 45 tiles of 64 x 64):
        // split-K fills the chip; the slices are added in a fixed order while the unregularised copy N0 is made
        // (measured at k = 519, k range 19 000: 32 x 32 tiles x 8 slices 1.19 ms per iteration, 64 x 64 x 8 1.21, 32 x 32 x 4 1.21, unsplit 1.40)
        const int T = ns_split_T > 0 ? ns_split_T : Dev::pick_tile(k);
        const int64_t ntile = ((k + 32 * T - 1) / (32 * T));
        int nsplit = ns_split_n > 0 ? ns_split_n : (int)std::min<int64_t>(NS_MAX_SPLIT, std::max<int64_t>(1, 1224 / std::max<int64_t>(1, ntile * (ntile + 1) / 2)));
        nsplit = (int)std::min<int64_t>(nsplit, std::max<int64_t>(1, h->ns_ldg / 512));
        if (nsplit > 1) {
            const int64_t pstride = h->ns_fN.ld * h->ns_fN.ld;
            dev.launch_syrk(T, h->d_nsG, h->ns_ldg, nullptr, 0, k, (int)h->ns_ldg, h->d_nsth, nullptr, h->d_nsNp, h->ns_fN.ld, 0, 0, -1, nullptr, 0, -1.0, nsplit, pstride);
            dev.end(id);
            hipLaunchKernelGGL(k_ns_reduce_lower, dim3((unsigned)((k + 255) / 256), (unsigned)k), dim3(256), 0, h->stream, (const double*)h->d_nsNp, nsplit, pstride, h->ns_fN.ld,
                               h->ns_fN.S, h->d_nsN0, k, k <= ASM_SMALL_USE ? 1 : 0, h->d_diag0, 1e-13, 1e-30);      // (+ k_diag_prepare, mode 0)
        } else {
            dev.launch_syrk(Dev::pick_tile(k), h->d_nsG, h->ns_ldg, nullptr, 0, k, (int)h->ns_ldg, h->d_nsth, nullptr, h->ns_fN.S, h->ns_fN.ld, 0, 0);
            dev.end(id);
            hipLaunchKernelGGL(k_ns_copy_lower, dim3((unsigned)((k + 255) / 256), (unsigned)k), dim3(256), 0, h->stream, (const double*)h->ns_fN.S, h->ns_fN.ld, h->d_nsN0, h->ns_fN.ld, k, k <= ASM_SMALL_USE ? 1 : 0);
            dev.diag_prepare(k, 0, 1e-13, 1e-30);
        }
        dev.chol(k, 1e-14, false);
        // dpbar = -e: the component of the iterate outside pbar + null(A_EF), split off once per LP and shrunk by (1 - a) with every step
        double* e = nsv(14);
        if (!ip.ns_e_ready) {
            ip.ns_e_ready = true;
            double *d0 = nsv(2), *zz = nsv(3), *tk = nsv(12);
            hipLaunchKernelGGL(k_ns_e0, dim3(gN), dim3(256), 0, h->stream, P, (const double*)nsq().pbar, d0, ldn);
            gemv_rows((const double*)h->d_nsG, h->ns_ldg, (const double*)d0, tk, (int64_t)k, ldn);
            ns_gemv_t_dense(tk, k, zz);
            hipLaunchKernelGGL(k_ns_e1, dim3(gN), dim3(256), 0, h->stream, P, (const double*)d0, (const double*)zz, e, ldn);
        }
        // (fused launches: negation + clearing of the residual measure; sparse product + its row- / column-wise kernel)
        const double* vals = dev.sparse_vals(h->d_Ah);
        hipLaunchKernelGGL(k_ns_spmvn_wm_neg, dim3(std::max(gM, gN)), dim3(256), 0, h->stream, (const int*)h->d_sp_ptr, (const int*)h->d_sp_col, vals, (const double*)e, dpb, ldn,
                           P.scal + SC_NSERR, X, thI, yM, M);
        hipLaunchKernelGGL(k_ns_spmvt_kx, dim3((unsigned)((ldn * 8 + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_sc_ptr, (const int*)h->d_sc_row, (const int*)h->d_sc_pos, vals,
                           (const double*)yM, th, (const double*)dpb, kdpb, lp.n, ldn);
        (void)aM; (void)atw;
    }
    // One Newton solve in null-space form (oracle: IPM.run, solve_ns): mode 0 affine, 1 Mehrotra corrector on `base`.  The relative residual of
    // the reduced solve (after its refinement sweep) is accumulated in SC_NSERR.
    void ns_newton(int mode, const IpmDir& base, IpmDir& D) {
        const int k = ip.ns_k;
        const int64_t M = lp.M, n = lp.n, ldn = h->ldn;
        const NsIdx X = nsX();
        const unsigned g = grid_all(), gM = (unsigned)((M + 255) / 256), gN = (unsigned)((ldn + 255) / 256), gK = (unsigned)((k + 255) / 256);
        double *dpb = nsv(0), *kdpb = nsv(1), *ht = nsv(2), *v = nsv(3), *atw = nsv(4), *yM = nsv(5), *aM = nsv(6), *bI = nsv(7);
        double *ru = nsv(10), *du = nsv(11), *rr = nsv(12), *dd = nsv(13);
        const double* th = h->d_nsth;
        const double* thI = h->d_nsth + ldn;
        const double res = 1.0;
        const double* vals = dev.sparse_vals(h->d_Ah);
        hipLaunchKernelGGL(k_ns_rhs1_bi, dim3(g), dim3(256), 0, h->stream, P, base, mode, X, thI, res, bI, yM);
        hipLaunchKernelGGL(k_ns_spmvt_ht, dim3((unsigned)((ldn * 8 + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_sc_ptr, (const int*)h->d_sc_row, (const int*)h->d_sc_pos, vals,
                           (const double*)yM, th, (const double*)P.hp, (const double*)kdpb, res, ht, v, n, ldn);
        gemv_rows((const double*)h->d_nsG, h->ns_ldg, (const double*)v, ru, (int64_t)k, ldn);
        if (k <= ASM_SMALL_USE) {
            // solve, refinement sweep on the unregularised matrix and the residual check in ONE one-workgroup launch
            hipLaunchKernelGGL(k_ns_reduced_solve, dim3(1), dim3(1024), 0, h->stream, (const double*)h->ns_fN.S, h->ns_fN.ld, (const double*)h->ns_fN.Linv,
                               (const double*)h->d_nsN0, k, (const double*)ru, du, P.scal + SC_NSERR);
        } else {
            dev.use_factor(h->ns_fN);
            dev.chol_solve_dev(ru, du, k);
            hipLaunchKernelGGL(k_ns_symv_res, dim3((unsigned)((k + 3) / 4)), dim3(256), 0, h->stream, (const double*)h->d_nsN0, h->ns_fN.ld, k, (const double*)du, (const double*)ru, rr);
            dev.chol_solve_dev(rr, dd, k);
            dev.use_main();
            hipLaunchKernelGGL(k_ns_add, dim3(gK), dim3(256), 0, h->stream, (const double*)du, (const double*)dd, du, (int64_t)k);
            hipLaunchKernelGGL(k_ns_symv_res, dim3((unsigned)((k + 3) / 4)), dim3(256), 0, h->stream, (const double*)h->d_nsN0, h->ns_fN.ld, k, (const double*)du, (const double*)ru, rr);
            hipLaunchKernelGGL(k_ns_relres, dim3(1), dim3(1024), 0, h->stream, (const double*)rr, (const double*)ru, k, P.scal + SC_NSERR);
        }
        if (k <= ASM_SMALL_USE) {
            hipLaunchKernelGGL(k_gemv_t_small_dp, dim3(gN), dim3(256), 0, h->stream, (const double*)h->d_nsG, h->ns_ldg, k, (const double*)du, P, D, th, (const double*)dpb, res, ldn);
        } else {
            ns_gemv_t_dense(du, k, v);
            hipLaunchKernelGGL(k_ns_dp, dim3(gN), dim3(256), 0, h->stream, P, D, th, (const double*)dpb, res, (const double*)v, ldn);
        }
        hipLaunchKernelGGL(k_ns_spmvn_rows, dim3(gM), dim3(256), 0, h->stream, (const int*)h->d_sp_ptr, (const int*)h->d_sp_col, vals, P, D, X, thI, (const double*)bI, yM);
        (void)aM; (void)atw;
    }
    // out[n] = Zt' u   (Zt dense, k rows of pitch ldg)
    void ns_gemv_t_dense(const double* u, int k, double* out) {
        if (k <= ASM_SMALL_USE) {
            hipLaunchKernelGGL(k_gemv_t_small, dim3((unsigned)((h->ldn + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->d_nsG, h->ns_ldg, k, u, out, h->ldn);
            return;
        }
        int64_t R = std::min<int64_t>((k + 31) / 32, ASM_TMAXCHUNKS);
        int64_t chunk = (k + R - 1) / R;
        R = (k + chunk - 1) / chunk;
        hipLaunchKernelGGL(k_gemv_t_stage1, dim3((unsigned)((h->ldn + 255) / 256), (unsigned)R), dim3(256), 0, h->stream, (const double*)h->d_nsG, h->ns_ldg, u,
                           h->d_partial, (int64_t)k, h->ldn, chunk);
        hipLaunchKernelGGL(k_gemv_t_stage2, dim3((unsigned)((h->ldn + 255) / 256)), dim3(256), 0, h->stream, h->d_partial, out, R, h->ldn);
    }

    bool use_red = false;     // reduced row form: factor of the rows redE, diagonal on the rows redI
    bool use_perm = false;    // full row form with the rows in the handle's banded order (row_band > 0): solves gather / scatter through d_rowperm
    std::vector<int> redE, redI;
    int nE = 0, nI = 0;
    vec tmpM2;
    bool use_col = false;     // form of the current factorisation
    int cg_max = 0;           // most CG steps any solve of the current iteration needed
    bool cg_fail = false;     // a solve of the current iteration left its CG loop without reaching the tolerance
    // out = (approximate) S^-1 in : the Cholesky factor of S (row form) or Sherman-Morrison-Woodbury through the factor of K
    void precond(const double* in, double* out) {
        const int M = (int)lp.M;
        const unsigned gm = (unsigned)((lp.M + 255) / 256);
        if (use_red) {
            hipLaunchKernelGGL(k_red_gather, dim3((unsigned)((nE + 255) / 256)), dim3(256), 0, h->stream, h->d_idx, nE, in, h->d_rce);
            dev.chol_solve_dev(h->d_rce, h->d_rze, nE);
            hipLaunchKernelGGL(k_red_scatter, dim3(gm), dim3(256), 0, h->stream, h->d_idx, nE, h->d_rze, h->d_idxI, nI, h->d_rdI, in, out);
            return;
        }
        if (use_perm) {
            hipLaunchKernelGGL(k_red_gather, dim3(gm), dim3(256), 0, h->stream, (const int*)h->d_rowperm, M, in, h->d_rce);
            dev.chol_solve_dev(h->d_rce, h->d_rze, M);
            hipLaunchKernelGGL(k_red_scatter, dim3(gm), dim3(256), 0, h->stream, (const int*)h->d_rowperm, M, (const double*)h->d_rze, (const int*)h->d_rowperm, 0,
                               (const double*)h->d_rze, in, out);
            return;
        }
        if (!use_col) { dev.chol_solve_dev(in, out, M); return; }
        hipLaunchKernelGGL(k_col_scale, dim3(gm), dim3(256), 0, h->stream, h->d_cdinv, in, h->d_cu, lp.M);      // u = D^-1 r
        dev.gemv_t_dev(h->d_Ah, h->d_cu, h->d_ct);                                                                 // Ah' u
        if (h->col_band > 0) {                                                                                     // K^-1 (columns in banded order)
            const unsigned gn = (unsigned)((lp.n + 255) / 256);
            hipLaunchKernelGGL(k_red_gather, dim3(gn), dim3(256), 0, h->stream, (const int*)h->d_colperm, (int)lp.n, (const double*)h->d_ct, h->d_rce);
            dev.chol_solve_dev(h->d_rce, h->d_rze, (int)lp.n);
            hipLaunchKernelGGL(k_red_scatter, dim3(gn), dim3(256), 0, h->stream, (const int*)h->d_colperm, (int)lp.n, (const double*)h->d_rze, (const int*)h->d_colperm, 0,
                               (const double*)h->d_rze, (const double*)h->d_ct, h->d_cv);
        } else {
            dev.chol_solve_dev(h->d_ct, h->d_cv, (int)lp.n);                                                       // K^-1
        }
        dev.gemv_n_dev(h->d_Ah, h->d_cv, h->d_cw);                                                                 // Ah v
        hipLaunchKernelGGL(k_col_finish, dim3(gm), dim3(256), 0, h->stream, h->d_cdinv, h->d_cu, h->d_cw, out, lp.M);
    }
    // one Newton solve with the current factor (oracle: IPM.run.solve); mode 0 affine, 1 Mehrotra corrector built on
    // `base`, 2 Gondzio centrality corrector for `base` at the trial steps (tp, td)
    // spec = 0: as the oracle states it.  spec = 1 / 2 (first / later solve of an iteration, only with the factor of S itself as preconditioner):
    // the residual check is not waited for - its verdict travels in scal[SC_SPEC] with the next block the iteration reads (ipm_run redoes the
    // iteration's solves with spec = 0 when one of them missed its tolerance: one host round trip per solve less in the common case).
    void ipm_solve(int mode, const IpmDir& base, IpmDir& D, double tp = 0.0, double td = 0.0, int spec = 0) {
        const unsigned g = grid_all();
        
        hipLaunchKernelGGL(k_ipm_rhs1, dim3(g), dim3(256), 0, h->stream, P, base, mode, tp, td, MCC_BMIN, MCC_BMAX);
        dev.gemv_n_dev(h->d_Ah, P.tmpn, P.t1);
        hipLaunchKernelGGL(k_ipm_rhs2, dim3(g), dim3(256), 0, h->stream, P, mode == 2 ? 0.0 : 1.0);
        precond(P.rhs, D.dy);
        if (spec) {
            dev.gemv_t_dev(h->d_Ah, D.dy, d_tN);
            hipLaunchKernelGGL(k_vec_mul, dim3((unsigned)((lp.n + 255) / 256)), dim3(256), 0, h->stream, d_tN, P.thp_inv, lp.n);
            dev.gemv_n_dev(h->d_Ah, d_tN, d_sres);
            hipLaunchKernelGGL(k_ipm_res, dim3(1), dim3(1024), 0, h->stream, P, d_sres, D.dy, 0u, spec, 1e-10, PCG_KAPPA * ip.rpmax);
        } else {
            // preconditioned CG on the unregularised Schur system, the Cholesky factor as preconditioner (oracle: IPM.run.solve).
            // The residual of this system is exactly the primal residual the step leaves behind, hence the tolerance.
            auto applyS = [&](const double* v) {          // d_sres = Ah Th^-1 Ah' v
                dev.gemv_t_dev(h->d_Ah, v, d_tN);
                hipLaunchKernelGGL(k_vec_mul, dim3((unsigned)((lp.n + 255) / 256)), dim3(256), 0, h->stream, d_tN, P.thp_inv, lp.n);
                dev.gemv_n_dev(h->d_Ah, d_tN, d_sres);
            };
            applyS(D.dy);
            unsigned pub = pub_next();
            hipLaunchKernelGGL(k_ipm_res, dim3(1), dim3(1024), 0, h->stream, P, d_sres, D.dy, pub, 0, 0.0, 0.0);
            read_scal(pub);
            // the approximate preconditioner (column form) gets the tighter floor (oracle: IPM.run.solve)
            const double tol = std::max(((use_col || use_red) ? 1e-13 : 1e-10) * h->h_scal[SC_RMAX], PCG_KAPPA * ip.rpmax);
            if (h->h_scal[SC_EMAX] > tol) {
                precond(P.res, d_corr);
                hipLaunchKernelGGL(k_pcg_start, dim3(1), dim3(1024), 0, h->stream, P, d_corr, d_pcg);
                bool converged = false;
                for (int it = 0; it < PCG_MAXIT; ++it) {
                    applyS(d_pcg);
                    pub = pub_next();
                    hipLaunchKernelGGL(k_pcg_step1, dim3(1), dim3(1024), 0, h->stream, P, d_sres, d_pcg, D.dy, pub);
                    read_scal(pub);
                    h->stats_pcg += 1;
                    cg_max = std::max(cg_max, it + 1);
                    if (h->h_scal[SC_STOP] != 0.0) break;
                    if (h->h_scal[SC_EMAX] <= tol) { converged = true; break; }
                    precond(P.res, d_corr);
                    hipLaunchKernelGGL(k_pcg_step2, dim3(1), dim3(1024), 0, h->stream, P, d_corr, d_pcg);
                }
                if (!converged) cg_fail = true;
            }
        }
        dev.gemv_t_dev(h->d_Ah, D.dy, d_tN);
        hipLaunchKernelGGL(k_ipm_dir, dim3(g), dim3(256), 0, h->stream, P, D, d_tN);
    }

    // accuracy bound of the reduced solves of a null-space iteration (ASM_NS_RERR: test knob - a tiny bound sends every LP through the fall-back to the row form)
    static double ns_rerr() { static const double e = [] { const char* v = std::getenv("ASM_NS_RERR"); return v ? std::atof(v) : NS_RERR; }(); return e; }
    static int sig_exp() { static const int e = [] { const char* v = std::getenv("ASM_IPM_SIGEXP"); return v ? std::atoi(v) : 3; }(); return e; }      // (measurement knob)
    static double eta0() { static const double e = [] { const char* v = std::getenv("ASM_IPM_ETA0"); return v ? std::atof(v) : 0.995; }(); return e; }
    int btag = 100;      // alignment tags of a scenario batch grow in program order inside one LP (asm_batch.hip.h)
    int ipm_run(double tol, int max_more) {
        const int M = (int)lp.M;
        int done = 0;
        // null-space iterations apply their step on the device (k_ns_update_dev) and are checked with the NEXT measures: one read-back per
        // iteration.  ns_pending: the last iteration was one of those and its accuracy check is still owed
        static const bool ns_defer = [] { const char* v = std::getenv("ASM_NS_DEFER"); return !(v && v[0] == '0'); }();      // (measurement knob)
        bool ns_pending = false;
        while (true) {
            asmb::barrier(btag);                 // scenario batch: iterations of different scenarios run in lockstep (min-PC-first)
            ipm_measures();
            if (ns_pending) {
                ns_pending = false;
                if (h->verbose) std::fprintf(stderr, "[asm]     ap %.3e ad %.3e (null-space step, applied on the device)\n", h->h_scal[SC_AP], h->h_scal[SC_AD]);
                if (h->h_scal[SC_NSERR] > ns_rerr()) {
                    // the reduced system lost its accuracy and the device left the iterate alone: redo the iteration in row form
                    // (oracle: IPM.run) - measured again below as the row form measures it
                    ns_finish_y();
                    ip.ns_off = true;
                    continue;
                }
            }
            if (h->verbose) std::fprintf(stderr, "[asm] ipm %3d pinf %.3e dinf %.3e gap %.3e\n", ip.iters, ip.pinf, ip.dinf, ip.gap);
            if (ip.pinf <= tol && ip.gap <= tol && (ip.dinf <= tol || (ip.gap <= IPM_GAP_DONE * tol && ip.dinf <= IPM_DINF_FLOOR))) {
                if (ns_live()) ns_finish_y();
                return ip.status = ASM_OPTIMAL;
            }
            if (ip.iters >= 3 && ip.ymax > 1e3 * lp.scale_q) {
                if (ns_live()) ns_finish_y();
                down(ip.y, P.y, lp.M);
                HIPCHK(hipStreamSynchronize(h->stream));
                if (farkas_margin(ip.y) > 1e-9) return ip.status = ASM_INFEASIBLE;
            }
            if (done >= max_more) { if (ns_live()) ns_finish_y(); return ip.status = ASM_OTHER; }
            // jammed: complementarity collapsed but the primal residual no longer decreases (oracle: IPM.run)
            ip.pinf_hist.push_back(ip.pinf);
            if (ip.iters >= 10 && ip.pinf > JAM_PINF && ip.gap <= 1e-2 * ip.pinf && ip.pinf > 0.5 * ip.pinf_hist[ip.pinf_hist.size() - 4]) {
                ip.stalled = true;
                if (ns_live()) ns_finish_y();
                return ip.status = ASM_OTHER;
            }
            if (ip.ns_ok && !ip.ns_off)      // null-space form: its theta~ in the same launch (ns_iter_setup)
                hipLaunchKernelGGL(k_ipm_theta_ns, dim3(std::max(grid_all(), (unsigned)((std::max<int64_t>(h->ldn, h->ns_nIp) + 255) / 256))), dim3(256), 0, h->stream, P, IPM_RHO_P, nsX(), h->d_nsth, h->ldn,
                                   h->ns_nIp);
            else
                hipLaunchKernelGGL(k_ipm_theta, dim3(grid_all()), dim3(256), 0, h->stream, P, IPM_RHO_P);
            // column form (oracle: IPM.run): K = Th + Ah' D^-1 Ah (n x n) while its Sherman-Morrison-Woodbury preconditioner
            // keeps the CG short, the row form S = Ah Th^-1 Ah' + D (M x M) otherwise
            use_red = false;
            use_perm = false;
            // null-space form (oracle: IPM.run): set up once per LP, k x k factorisation per iteration
            use_ns = false;
            if (ip.ns_ok && !ip.ns_off) {
                use_ns = !ip.ns_off;
            }
            use_col = !use_ns && ip.col_ok && !ip.col_off;
            if (use_ns) {
                ip.ns_iters += 1;
                ns_iter_setup();
            } else if (use_col) {
                ip.col_iters += 1;
                hipLaunchKernelGGL(k_ipm_col_prep, dim3(grid_all()), dim3(256), 0, h->stream, P, IPM_RHO_P, COL_FIXED, h->d_cdinv, h->d_cth);
                if (h->col_band > 0) {
                    // columns in their banded order: K built from the structural column pairs, factor and substitutions stop at the band
                    hipLaunchKernelGGL(k_red_gather, dim3((unsigned)((lp.n + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_colperm, (int)lp.n, (const double*)h->d_cth, h->d_diag);
                    dev.schur_banded_cols_dev(h->d_cdinv, h->d_diag);
                } else {
                    dev.syrk_col(h->d_cdinv, h->d_cth);
                }
                dev.diag_prepare((int)lp.n, 0, 1e-13, 1e-30);
                dev.chol((int)lp.n);
            } else {
                // reduced row form (oracle: IPM.run): inequality rows whose slack term dominates their Schur diagonal stay out
                // of the factor and get a diagonal preconditioner
                if (ip.red_ok && !ip.red_off) {
                    dev.schur_diag(P.thp_inv, h->d_sdiag);
                    down(tmpM, P.dS, lp.M);
                    down(tmpM2, h->d_sdiag, lp.M);
                    HIPCHK(hipStreamSynchronize(h->stream));
                    redE.clear(); redI.clear();
                    for (int64_t q = 0; q < lp.M; ++q) {                 // kept rows in the order of the factorisations
                        const int64_t i = h->row_band > 0 ? h->row_perm_h[q] : q;
                        if (tmpM[i] > RED_TAU * tmpM2[i]) redI.push_back((int)i);
                        else redE.push_back((int)i);
                    }
                    use_red = (double)redI.size() >= RED_MIN_FRAC * (double)lp.M && !redE.empty();
                }
                if (use_red) {
                    ip.red_iters += 1;
                    nE = (int)redE.size(); nI = (int)redI.size();
                    vec dE_(nE), dI_(nI);
                    for (int a = 0; a < nE; ++a) dE_[a] = tmpM[redE[a]];
                    for (int b = 0; b < nI; ++b) dI_[b] = tmpM2[redI[b]] + tmpM[redI[b]];
                    HIPCHK(hipMemcpyAsync(h->d_idx, redE.data(), nE * sizeof(int), hipMemcpyHostToDevice, h->stream));
                    HIPCHK(hipMemcpyAsync(h->d_idxI, redI.data(), nI * sizeof(int), hipMemcpyHostToDevice, h->stream));
                    HIPCHK(hipMemcpyAsync(h->d_diag, dE_.data(), nE * sizeof(double), hipMemcpyHostToDevice, h->stream));
                    HIPCHK(hipMemcpyAsync(h->d_rdI, dI_.data(), nI * sizeof(double), hipMemcpyHostToDevice, h->stream));
                    if (h->row_band > 0) {
                        std::vector<int> cp(lp.M, -1);
                        for (int a = 0; a < nE; ++a) cp[redE[a]] = a;
                        HIPCHK(hipMemcpyAsync(h->d_cpos, cp.data(), lp.M * sizeof(int), hipMemcpyHostToDevice, h->stream));
                        h2d_done(h);
                        dev.schur_banded_dev(h->d_cpos, nE, P.thp_inv, h->d_diag);
                    } else {
                        h2d_done(h);      // the host vectors go out of scope
                        dev.syrk_gathered_dev(h->d_idx, nE, P.thp_inv, h->d_diag);
                    }
                    dev.diag_prepare(nE, 0, 1e-13, 1e-30);
                    dev.chol(nE);
                } else if (h->row_band > 0) {
                    // full row form, rows in the banded order: S is built entry by entry, factor and substitutions stop at the band
                    use_perm = true;
                    hipLaunchKernelGGL(k_red_gather, dim3((unsigned)((lp.M + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->d_rowperm, M, (const double*)P.dS, h->d_diag);
                    dev.schur_banded_dev(h->d_rowpos, M, P.thp_inv, h->d_diag);
                    dev.diag_prepare(M, 0, 1e-13, 1e-30);
                    dev.chol(M);
                } else {
                    dev.syrk_dev(nullptr, M, P.thp_inv, P.dS);
                    dev.diag_prepare(M, 0, 1e-13, 1e-30);
                    dev.chol(M);
                }
            }
            ip.iters += 1;
            done += 1;
            double ap = 0.0, ad = 0.0;
            // the solves of the iteration; deferred = the residual checks of the solves are not waited for one by one (ipm_solve, spec):
            // false when one of them missed its tolerance - the solves are then redone as the oracle states them (same iterate, same
            // factor: every vector they write is written again)
            auto solves = [&](const bool deferred) -> bool {
                cg_max = 0;
                cg_fail = false;
                if (use_ns) ns_newton(0, dirA, dirA); else ipm_solve(0, dirA, dirA, 0.0, 0.0, deferred ? 1 : 0);
                hipLaunchKernelGGL(k_ipm_steps, dim3(red_grid()), dim3(1024), 0, h->stream, P, dirA, 0u);
                hipLaunchKernelGGL(k_ipm_muaff, dim3(red_grid()), dim3(1024), 0, h->stream, P, dirA, sig_exp());
                if (use_ns) ns_newton(1, dirA, dirC); else ipm_solve(1, dirA, dirC, 0.0, 0.0, deferred ? 2 : 0);
                if (use_ns && ns_defer) {       // step lengths stay on the device (k_ns_update_dev below)
                    hipLaunchKernelGGL(k_ipm_steps, dim3(red_grid()), dim3(1024), 0, h->stream, P, dirC, 0u);
                    return true;
                }
                unsigned pub = pub_next();
                hipLaunchKernelGGL(k_ipm_steps, dim3(red_grid()), dim3(1024), 0, h->stream, P, dirC, pub);
                read_scal(pub);
                if (deferred && h->h_scal[SC_SPEC] != 0.0) return false;
                ap = h->h_scal[SC_AP]; ad = h->h_scal[SC_AD];
                // Gondzio multiple centrality correctors (oracle: IPM.run): dirA is free again and receives the candidate
                for (int kc = 0; kc < (use_ns ? 0 : IPM_MCC); ++kc) {      // (no correctors in null-space form: a Newton solve costs more than the factorisation there)
                    if (std::min(ap, ad) >= 0.9) break;
                    const double tp = std::min(1.0, ap + MCC_DELTA), td = std::min(1.0, ad + MCC_DELTA);
                    ipm_solve(2, dirC, dirA, tp, td, deferred ? 2 : 0);
                    hipLaunchKernelGGL(k_ipm_diradd, dim3(grid_all()), dim3(256), 0, h->stream, P, dirA, dirC);
                    pub = pub_next();
                    hipLaunchKernelGGL(k_ipm_steps, dim3(red_grid()), dim3(1024), 0, h->stream, P, dirA, pub);
                    read_scal(pub);
                    if (deferred && h->h_scal[SC_SPEC] != 0.0) return false;
                    const double ap2 = h->h_scal[SC_AP], ad2 = h->h_scal[SC_AD];
                    if (!(ap2 >= ap && ad2 >= ad && ap2 + ad2 >= ap + ad + MCC_GAMMA * MCC_DELTA)) break;
                    std::swap(dirA, dirC);
                    ap = ap2; ad = ad2;
                }
                return true;
            };
            static const bool spec_env = [] { const char* v = std::getenv("ASM_IPM_DEFER_CHECK"); return !(v && v[0] == '0'); }();
            const bool defer = spec_env && !use_ns && !use_col && !use_red;      // the factor of S itself is the preconditioner
            if (!(defer && solves(true))) solves(false);
            if (use_ns && ns_defer) {
                const double eta = ip.mu >= 1.0 ? eta0() : std::min(std::max(eta0(), 1.0 - ip.mu / lp.scale_q), 0.999999);
                hipLaunchKernelGGL(k_ns_update_dev, dim3(grid_all()), dim3(256), 0, h->stream, P, dirC, eta, nsv(14), h->ldn, ns_rerr());
                ns_pending = true;
                continue;
            }
            if (h->verbose) std::fprintf(stderr, "[asm]     ap %.3e ad %.3e  cg steps so far %lld\n", ap, ad, (long long)h->stats_pcg);
            if (use_ns && h->h_scal[SC_NSERR] > ns_rerr()) {
                ns_finish_y();
                ip.ns_off = true;          // the reduced system lost its accuracy: redo the iteration in row form (oracle: IPM.run)
                continue;
            }
            if (use_col && cg_fail) {      // column-form preconditioner lost its accuracy: redo this iteration in row form (oracle: IPM.run)
                ip.col_off = true;
                continue;
            }
            if (use_red && cg_fail) {      // same safety net for the reduced row form
                ip.red_off = true;
                continue;
            }
            const double eta = ip.mu >= 1.0 ? eta0() : std::min(std::max(eta0(), 1.0 - ip.mu / lp.scale_q), 0.999999);
            if (use_ns)
                hipLaunchKernelGGL(k_ns_update, dim3(grid_all()), dim3(256), 0, h->stream, P, dirC, std::min(1.0, eta * ap), std::min(1.0, eta * ad), nsv(14),
                                   1.0 - std::min(1.0, eta * ap), h->ldn);
            else
                hipLaunchKernelGGL(k_ipm_update, dim3(grid_all()), dim3(256), 0, h->stream, P, dirC, std::min(1.0, eta * ap), std::min(1.0, eta * ad));
            if (use_col && cg_max > COL_MAX_CG) ip.col_off = true;
            if (use_red && cg_max > RED_MAX_CG) ip.red_off = true;
        }
    }

    // ---------------------------------------------------------------- active-set machinery (device resident)
    // The working sets, index lists and every O(M+n) vector of the equality-constrained solves live in HBM
    // (asm_as_kernels.hip.h); the host reads back one block of counters / scalars after the set-up kernel (the size of the
    // gathered Schur system fixes the launch grids) and one after the tail kernel of a solve.
    struct EqpOut {
        vec p, s, y, act, z;   // act = Ah p + E s ; z = q - Ah' y
    };
    AsPtrs A;
    AsSets S_[6];              // 0..2: rotation of the correction loop; 3: partition of the iterate; 4: primal working set; 5: dual
    double *d_pref = nullptr, *d_zero = nullptr;
    double *d_p0 = nullptr, *d_s0 = nullptr, *d_y0 = nullptr, *d_act0 = nullptr, *d_z0 = nullptr;      // projection of the iterate
    double *d_pa = nullptr, *d_sa = nullptr, *d_acta = nullptr;                                          // anchor of the primal method
    double *d_pf = nullptr, *d_sf = nullptr, *d_actf = nullptr, *d_yf = nullptr, *d_zf = nullptr;       // least-norm point (scratch) / multipliers
    int final_sets = 0;
    int as_nH = 0, as_nF = 0;

    void as_bind() {
        const int64_t ln = h->ldn, lm = h->Mp, ls = h->nsp;
        double* a = h->d_as;
        auto N = [&]() { double* r_ = a; a += ln; return r_; };
        auto Mv = [&]() { double* r_ = a; a += lm; return r_; };
        auto Sv = [&]() { double* r_ = a; a += ls; return r_; };
        A.q = P.q; A.lb = P.lb; A.ub = P.ub; A.r = P.r; A.w = P.w; A.slo = P.slo; A.scoef = P.scoef;
        A.rtype = P.rtype; A.srow = P.srow; A.rs0 = P.rs0; A.rs1 = P.rs1;
        A.n = lp.n; A.M = lp.M; A.ns = lp.ns; A.scale_q = lp.scale_q;
        A.Fmask = N(); A.p = N(); A.z = N(); A.pB = N(); A.pF = N(); A.cF = N(); A.rd = N(); A.tN = N(); A.xfull = N(); A.nu = N();
        d_pref = N(); d_zero = N(); d_p0 = N(); d_z0 = N(); d_pa = N(); d_pf = N(); d_zf = N();
        A.Hmask = Mv(); A.sl = Mv(); A.y = Mv(); A.act = Mv(); A.t = Mv(); A.bH = Mv(); A.v = Mv(); A.u = Mv(); A.yH = Mv();
        A.yfull = Mv(); A.uacc = Mv(); A.ax = Mv();
        d_y0 = Mv(); d_act0 = Mv(); d_acta = Mv(); d_actf = Mv(); d_yf = Mv();
        A.s = Sv(); d_s0 = Sv(); d_sa = Sv(); d_sf = Sv();
        A.scal = a;
        int* ia = h->d_as_i;
        auto Ni = [&]() { int* r_ = ia; ia += ln; return r_; };
        auto Mi = [&]() { int* r_ = ia; ia += lm; return r_; };
        auto Si = [&]() { int* r_ = ia; ia += ls; return r_; };
        for (int k = 0; k < 6; ++k) { S_[k].rowst = Mi(); S_[k].bst = Ni(); S_[k].sst = Si(); }
        A.ksoft = Mi(); A.Hidx = Mi(); A.hpos = Mi(); A.Fidx = Ni(); A.fpos = Ni();
        A.rperm = h->row_band > 0 ? h->d_rowperm : nullptr;
        A.cnt = ia;
    }
    // per LP: reference point of the unique-optimum polish (0 clipped into the box), slack offsets of the rows
    void as_begin_lp() {
        as_bind();
        hipLaunchKernelGGL(k_as_sl, dim3((unsigned)((lp.M + 255) / 256 + 1)), dim3(256), 0, h->stream, A);
        hipLaunchKernelGGL(k_as_clip0, dim3((unsigned)((lp.n + 255) / 256)), dim3(256), 0, h->stream, A.lb, A.ub, (const double*)nullptr, d_zero, lp.n);
    }
    void as_read() {
        HIPCHK(hipMemcpyAsync(h->h_ascnt, A.cnt, AC_COUNT * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->h_asscal, A.scal, AS_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    void as_copy_sets(int dst, int src) {
        hipLaunchKernelGGL(k_as_copy_sets, dim3(grid_all()), dim3(256), 0, h->stream, S_[dst], S_[src], lp.M, lp.n, lp.ns);
    }
    void as_upload_sets(const ActiveSet& as, int dst) {
        std::vector<int> buf((size_t)(lp.M + lp.n + lp.ns));
        for (int64_t i = 0; i < lp.M; ++i) buf[i] = as.rowst[i];
        for (int64_t j = 0; j < lp.n; ++j) buf[lp.M + j] = as.bst[j];
        for (int64_t k = 0; k < lp.ns; ++k) buf[lp.M + lp.n + k] = as.sst[k];
        if (lp.M) HIPCHK(hipMemcpyAsync(S_[dst].rowst, buf.data(), lp.M * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(S_[dst].bst, buf.data() + lp.M, lp.n * sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (lp.ns) HIPCHK(hipMemcpyAsync(S_[dst].sst, buf.data() + lp.M + lp.n, lp.ns * sizeof(int), hipMemcpyHostToDevice, h->stream));
        h2d_done(h);
    }
    void dcopy(double* dst, const double* src, int64_t cnt) {
        if (cnt > 0) HIPCHK(hipMemcpyAsync(dst, src, cnt * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    // final answer of the LP (device -> host): p, s, y, z, act and the working set `final_sets`
    void as_download(EqpOut& o, ActiveSet& as) {
        const int64_t n = lp.n, M = lp.M, ns = lp.ns;
        if (!asmb::in_fiber() && h->d_dl) {
            // outside a batch: packed on the device, one copy into pinned memory (eight copies into pageable vectors cost the host tens of
            // microseconds each, with the GPU idle in between)
            hipLaunchKernelGGL(k_as_pack, dim3(grid_all()), dim3(256), 0, h->stream, (const double*)A.p, (const double*)A.z, (const double*)A.y, (const double*)A.act,
                               (const double*)A.s, S_[final_sets], n, M, ns, h->d_dl);
            const size_t bytes = (size_t)(2 * n + 2 * M + ns) * sizeof(double) + (size_t)(M + n + ns) * sizeof(int);
            HIPCHK(hipMemcpyAsync(h->h_dl, h->d_dl, bytes, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            const double* st = h->h_dl;
            o.p.assign(st, st + n); o.z.assign(st + n, st + 2 * n); o.y.assign(st + 2 * n, st + 2 * n + M); o.act.assign(st + 2 * n + M, st + 2 * n + 2 * M);
            o.s.assign(st + 2 * n + 2 * M, st + 2 * n + 2 * M + ns);
            const int* bi = reinterpret_cast<const int*>(st + 2 * n + 2 * M + ns);
            as.rowst.resize(M); as.bst.resize(n); as.sst.resize(ns);
            for (int64_t i = 0; i < M; ++i) as.rowst[i] = (int8_t)bi[i];
            for (int64_t j = 0; j < n; ++j) as.bst[j] = (int8_t)bi[M + j];
            for (int64_t k = 0; k < ns; ++k) as.sst[k] = (int8_t)bi[M + n + k];
            as.valid = true;
            return;
        }
        down(o.p, A.p, n); down(o.z, A.z, n); down(o.y, A.y, M); down(o.act, A.act, M); down(o.s, A.s, ns);
        std::vector<int> buf((size_t)(M + n + ns));
        if (M) HIPCHK(hipMemcpyAsync(buf.data(), S_[final_sets].rowst, M * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(buf.data() + M, S_[final_sets].bst, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (ns) HIPCHK(hipMemcpyAsync(buf.data() + M + n, S_[final_sets].sst, ns * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        as.rowst.resize(M); as.bst.resize(n); as.sst.resize(ns);
        for (int64_t i = 0; i < M; ++i) as.rowst[i] = (int8_t)buf[i];
        for (int64_t j = 0; j < n; ++j) as.bst[j] = (int8_t)buf[M + j];
        for (int64_t k = 0; k < ns; ++k) as.sst[k] = (int8_t)buf[M + n + k];
        as.valid = true;
    }
    void identify_dev(int dst) {
        hipLaunchKernelGGL(k_as_identify, dim3(grid_all()), dim3(256), 0, h->stream, P, S_[dst]);
    }

    // Equality-constrained solve on the working set `cur` (oracle: eqp / _face_primal_solve / face_dual's solve).
    //   mode 0: both projections from (p_ref, y_ref), 4 refinement sweeps              (eqp)
    //   mode 1: primal least-norm point only (p_ref = 0), 3 sweeps, the multipliers of that problem accumulated in uacc
    //   mode 2: basic least-squares multipliers only (y_ref = 0), 4 sweeps
    // Leaves t = Ah p and tN = Ah' y (mode 1: tN = Ah' u_full) for the tail kernel.
    bool part_factor = false;   // d_S holds the factor of the partition's Schur matrix (face_polish re-uses it for the rounds on the same sets)
    void as_solve(const AsSets& cur, const double* p_ref, const double* y_ref, int mode, bool reuse_factor = false) {
        const unsigned gA = grid_all(), gM = (unsigned)((lp.M + 255) / 256 + 1), gN = (unsigned)((lp.n + 255) / 256);
        hipLaunchKernelGGL(k_as_setup, dim3(1), dim3(1024), 0, h->stream, A, cur, p_ref, h->ldn, h->Mp);
        HIPCHK(hipMemcpyAsync(h->h_ascnt, A.cnt, AC_COUNT * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        const int nH = h->h_ascnt[AC_NH], nF = h->h_ascnt[AC_NF];
        const bool any_soft = h->h_ascnt[AC_ANYSOFT] != 0;
        as_nH = nH; as_nF = nF;
        if (nH > 0) {
            dev.gemv_n_dev(h->d_Ah, A.pB, A.t);
            if (any_soft) dev.gemv_t_dev(h->d_Ah, A.y, A.tN);
            hipLaunchKernelGGL(k_as_rhs, dim3(gA), dim3(256), 0, h->stream, A, y_ref);
        }
        if (nH > 0 && nF > 0) {
            if (!(reuse_factor && part_factor)) {
                if (h->row_band > 0) dev.schur_banded_dev(A.hpos, nH, A.Fmask, nullptr);        // Hidx is in the banded row order (k_as_setup)
                else dev.syrk_gathered_dev(A.Hidx, nH, A.Fmask, nullptr);
                dev.diag_prepare(nH, 1, 0.0, 0.0);
                dev.chol(nH, 1e-10);
                part_factor = false;
            }
            const int sweeps = mode == 1 ? 3 : 4;
            const unsigned gH = (unsigned)((nH + 255) / 256);
            for (int it = 0; it < sweeps; ++it) {
                if (mode != 2) {
                    dev.gemv_n_dev(h->d_Ah, A.pF, A.t);                                                  // A_HF pF
                    hipLaunchKernelGGL(k_as_res_p, dim3(gH), dim3(256), 0, h->stream, A);
                    dev.chol_solve_dev(A.v, A.u, nH);
                    hipLaunchKernelGGL(k_as_scatter_h, dim3(gM), dim3(256), 0, h->stream, A, (const double*)A.u, mode == 1 ? 1 : 0);
                    dev.gemv_t_dev(h->d_Ah, A.yfull, A.tN);                                              // A_HF' u
                    hipLaunchKernelGGL(k_as_add_f, dim3(gN), dim3(256), 0, h->stream, A);
                }
                if (mode != 1) {
                    hipLaunchKernelGGL(k_as_scatter_h, dim3(gM), dim3(256), 0, h->stream, A, (const double*)A.yH, 0);
                    dev.gemv_t_dev(h->d_Ah, A.yfull, A.tN);                                              // A_HF' yH
                    hipLaunchKernelGGL(k_as_rd, dim3(gN), dim3(256), 0, h->stream, A);
                    dev.gemv_n_dev(h->d_Ah, A.rd, A.t);                                                  // A_HF rd
                    hipLaunchKernelGGL(k_as_gather_h, dim3(gH), dim3(256), 0, h->stream, A);
                    dev.chol_solve_dev(A.v, A.u, nH);
                    hipLaunchKernelGGL(k_as_add_yh, dim3(gH), dim3(256), 0, h->stream, A);
                }
            }
        }
        if (nH > 0) hipLaunchKernelGGL(k_as_merge, dim3(gA), dim3(256), 0, h->stream, A, mode != 1 ? 1 : 0);
        dev.gemv_n_dev(h->d_Ah, A.p, A.t);
        if (mode == 1) {
            hipLaunchKernelGGL(k_as_scatter_h, dim3(gM), dim3(256), 0, h->stream, A, (const double*)A.uacc, 0);
            dev.gemv_t_dev(h->d_Ah, A.yfull, A.tN);
        } else {
            dev.gemv_t_dev(h->d_Ah, A.y, A.tN);
        }
        h->stats.eqp += 1;
    }

    // oracle: eqp_loop - solve, LP optimality test, bulk correction of the working set, at most `rounds` corrections.
    // Starts from the sets in S_[0]; on return `final_sets` is the buffer holding the last working set.
    bool eqp_loop(const double* p_ref, const double* y_ref, int rounds) {
        if (h->test_no_polish) return false;      // test hook (asm_test_no_polish): every active-set attempt fails -> the last resort decides
        int cur = 0, nx = 1, prev = 2;
        bool have_prev = false;
        double pr_last = -1.0;
        double t_round = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
        for (int k = 0; k <= rounds; ++k) {
            asmb::barrier(btag + 60 + k);
            if (ns_lp && p_ref == d_zero && y_ref == nullptr) {
                if (!as_solve_ns(S_[cur])) return false;
            } else {
                as_solve(S_[cur], p_ref, y_ref, 0);
            }
            hipLaunchKernelGGL(k_as_finish, dim3(1), dim3(1024), 0, h->stream, A, S_[cur], S_[nx], S_[prev], have_prev ? 1 : 0, TOL_P, TOL_D);
            as_read();
            const double pr = h->h_asscal[AS_PR], du = h->h_asscal[AS_DU];
            h->stats.kkt_pr = pr;
            h->stats.kkt_du = du;
            final_sets = cur;
            if (h->verbose) {
                const double t_now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
                std::fprintf(stderr, "[asm] eqp round %d: pr %.3e du %.3e changes %d  (%.2f ms)\n", k, pr, du, h->h_ascnt[AC_NCHG], t_now - t_round);
                t_round = t_now;
            }
            if (pr <= TOL_P && du <= TOL_D) return true;
            if (k == rounds) break;
            // oracle: EQP_RUNAWAY - a correction that made the primal residual 1000 times worse has left the neighbourhood of the partition
            if (pr_last >= 0.0 && pr > EQP_RUNAWAY * std::max(pr_last, TOL_P)) break;
            pr_last = pr;
            if (h->h_ascnt[AC_NCHG] == 0 || (have_prev && h->h_ascnt[AC_NDIFF] == 0)) break;
            // oracle: EQP_MAXCHG - a correction that moves more than 3 % of all constraints at once has run away: the next solve is not made
            if ((double)h->h_ascnt[AC_NCHG] > std::max((double)EQP_MINCHG, EQP_MAXCHG * (double)(lp.n + lp.M + lp.ns))) break;
            const int old_prev = prev;
            prev = cur; cur = nx; nx = old_prev;
            have_prev = true;
            final_sets = cur;
        }
        return false;
    }

    // bordered Cholesky of the small matrix Z'Z (host)
    struct SmallChol {
        std::vector<vec> T, L;
        bool factor_row(size_t k) {                 // row k of L from rows 0..k-1 and T[k]
            vec l(k + 1, 0.0);
            for (size_t a = 0; a < k; ++a) {
                double v = T[k][a];
                for (size_t b = 0; b < a; ++b) v -= l[b] * L[a][b];
                l[a] = v / L[a][a];
            }
            double d2 = T[k][k];
            for (size_t b = 0; b < k; ++b) d2 -= l[b] * l[b];
            if (!(d2 > 1e-14 * T[k][k])) return false;
            l[k] = std::sqrt(d2);
            L.push_back(l);
            return true;
        }
        bool append(const vec& trow) {              // trow: products with the members so far, then the diagonal entry
            const size_t k = T.size();
            for (size_t a = 0; a < k; ++a) T[a].push_back(trow[a]);
            T.push_back(trow);
            if (factor_row(k)) return true;
            T.pop_back();
            for (size_t a = 0; a < k; ++a) T[a].pop_back();
            return false;
        }
        bool remove_swap(size_t j) {                // member j leaves, the last member takes its slot
            const size_t last = T.size() - 1;
            if (j != last) {
                std::swap(T[j], T[last]);
                for (auto& r_ : T) std::swap(r_[j], r_[last]);
            }
            T.pop_back();
            for (auto& r_ : T) r_.pop_back();
            L.clear();
            for (size_t k = 0; k < T.size(); ++k)
                if (!factor_row(k)) return false;
            return true;
        }
        vec solve(const vec& g) const {
            const size_t k = L.size();
            vec x(g.begin(), g.begin() + k);
            for (size_t a = 0; a < k; ++a) {
                for (size_t b = 0; b < a; ++b) x[a] -= L[a][b] * x[b];
                x[a] /= L[a][a];
            }
            for (size_t a = k; a-- > 0;) {
                for (size_t b = a + 1; b < k; ++b) x[a] -= L[b][a] * x[b];
                x[a] /= L[a][a];
            }
            return x;
        }
    };

    bool face_primal_anchored() {
        const int64_t n = lp.n, M = lp.M, ns = lp.ns, ldz = h->ldn;
        const unsigned gM = (unsigned)((M + 255) / 256 + 1), gN = (unsigned)((n + 255) / 256);
        as_copy_sets(4, 3);
        as_solve(S_[4], nullptr, nullptr, 1);            // p0 = least-norm point of the partition; its factor, Hidx, Fmask stay
        const int nH0 = as_nH;
        double *p0 = d_pf, *t0 = d_actf;
        dcopy(p0, A.p, n); dcopy(t0, A.t, M);
        dcopy(d_pa, d_p0, n); dcopy(d_sa, d_s0, ns); dcopy(d_acta, d_act0, M);
        std::vector<std::pair<int, int64_t>> members;
        vec g, sign, u;
        SmallChol sc;
        for (int st = 0; st < FACE_STEPS; ++st) {
            const int k = (int)members.size();
            if (k > 0) {
                HIPCHK(hipMemcpyAsync(h->d_nsu, u.data(), k * sizeof(double), hipMemcpyHostToDevice, h->stream));
                hipLaunchKernelGGL(k_face_ns_combine, dim3(gN), dim3(256), 0, h->stream, (const double*)p0, (const double*)h->d_Zbuf, ldz,
                                   (const double*)h->d_nsu, k, A.p, n);
            } else {
                dcopy(A.p, p0, n);
            }
            dev.gemv_n_dev(h->d_Ah, A.p, A.t);
            hipLaunchKernelGGL(k_face_ns_step, dim3(1), dim3(1024), 0, h->stream, A, S_[4], d_pa, d_sa, d_acta, TOL_P);
            as_read();                                    // (the host copy of u is no longer needed by the device after this)
            const int nviol = h->h_ascnt[AC_NVIOL];
            if (h->verbose && (st < 5 || st % 20 == 0 || nviol == 0))
                std::fprintf(stderr, "[asm] face primal anchored %d: members %d viol %d hres %.2e\n", st, k, nviol, h->h_asscal[AS_HARDRES]);
            if (h->h_asscal[AS_HARDRES] > TOL_P) return false;
            if (nviol == 0) {
                int jw = -1;
                double worst = FACE_TOL_M;
                for (int j = 0; j < k; ++j)
                    if (-sign[j] * u[j] > worst) { worst = -sign[j] * u[j]; jw = j; }
                if (jw < 0) return true;                  // feasible, every multiplier has the right sign: THE least-norm point
                hipLaunchKernelGGL(k_face_ns_unmark, dim3(1), dim3(64), 0, h->stream, A, S_[4], members[jw].first, members[jw].second);
                dcopy(d_pa, A.p, n); dcopy(d_sa, A.s, ns); dcopy(d_acta, A.act, M);
                const int last = k - 1;
                if (jw != last) {
                    dcopy(h->d_Zbuf + (int64_t)jw * ldz, h->d_Zbuf + (int64_t)last * ldz, ldz);
                    members[jw] = members[last]; g[jw] = g[last]; sign[jw] = sign[last];
                }
                members.pop_back(); g.pop_back(); sign.pop_back();
                if (!sc.remove_swap((size_t)jw)) return false;
                u = sc.solve(g);
                continue;
            }
            const int fam = h->h_ascnt[AC_NCHG];
            const int64_t e = h->h_ascnt[AC_NDIFF];
            hipLaunchKernelGGL(k_face_ns_col, dim3(1), dim3(1024), 0, h->stream, A, (const double*)h->d_Ah, h->ldn, fam, e, (const double*)p0, (const double*)t0);
            if (nH0 > 0) {
                dev.gemv_n_dev(h->d_Ah, A.rd, A.t);
                hipLaunchKernelGGL(k_as_gather_h, dim3((unsigned)((nH0 + 255) / 256)), dim3(256), 0, h->stream, A);
                dev.chol_solve_dev(A.v, A.u, nH0);
                hipLaunchKernelGGL(k_as_scatter_h, dim3(gM), dim3(256), 0, h->stream, A, (const double*)A.u, 0);
                dev.gemv_t_dev(h->d_Ah, A.yfull, A.tN);
            } else {
                HIPCHK(hipMemsetAsync(A.tN, 0, h->ldn * sizeof(double), h->stream));
            }
            double* znew = h->d_Zbuf + (int64_t)k * ldz;
            hipLaunchKernelGGL(k_face_ns_z, dim3(gN), dim3(256), 0, h->stream, A, znew);
            hipLaunchKernelGGL(k_gemv_n, dim3((unsigned)((k + 1 + 3) / 4)), dim3(256), 0, h->stream, (const double*)h->d_Zbuf, ldz, (const double*)znew,
                               h->d_nsdots, (int64_t)(k + 1), ldz);
            HIPCHK(hipMemcpyAsync(h->h_nsdots, h->d_nsdots, (k + 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(h->h_asscal, A.scal, AS_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            const double zz = h->h_nsdots[k], cc = h->h_asscal[AS_PR];
            if (!(zz > 1e-10 * cc)) return false;        // dependent on the working set although it blocks
            vec trow(h->h_nsdots, h->h_nsdots + k + 1);
            if (!sc.append(trow)) return false;
            members.emplace_back(fam, e);
            g.push_back(h->h_asscal[AS_EQRES]);
            sign.push_back(fam == 0 ? (double)lp.rtype[e] : (fam == 1 ? -lp.scoef[e] : (fam == 2 ? 1.0 : -1.0)));
            u = sc.solve(g);
            h->stats.eqp += 1;
        }
        return false;
    }

    // oracle: face_polish - canonical pair of a non-unique optimum on the partition in S_[3].
    // Returns 2 ('face': least-norm point + basic multipliers), 1 ('ref': projection of the iterate), 0 (partition not optimal).
    int face_polish() {
        if (h->test_no_polish) return 0;
        const int64_t n = lp.n, M = lp.M, ns = lp.ns;
        hipLaunchKernelGGL(k_as_clip0, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, A.lb, A.ub, (const double*)P.p, d_pref, n);
        as_solve(S_[3], d_pref, P.y, 0);
        hipLaunchKernelGGL(k_as_finish, dim3(1), dim3(1024), 0, h->stream, A, S_[3], S_[1], S_[2], 0, TOL_P, TOL_D);
        as_read();
        h->stats.kkt_pr = h->h_asscal[AS_PR];
        h->stats.kkt_du = h->h_asscal[AS_DU];
        if (h->verbose) std::fprintf(stderr, "[asm] face polish: projection of the iterate on its partition pr %.3e du %.3e\n", h->h_asscal[AS_PR], h->h_asscal[AS_DU]);
        if (!(h->h_asscal[AS_PR] <= TOL_P && h->h_asscal[AS_DU] <= TOL_D)) return 0;
        dcopy(d_p0, A.p, n); dcopy(d_s0, A.s, ns); dcopy(d_y0, A.y, M); dcopy(d_act0, A.act, M); dcopy(d_z0, A.z, n);
        part_factor = true;            // the factor in d_S belongs to the partition: the first dual / primal round below re-use it
        // ---- dual: basic least-squares multipliers on the partition, sign repair (oracle: face_dual; independent of the primal
        // stage, run first so that its round 0 and the primal round 0 share the factorisation of the projection above)
        bool okd = false;
        as_copy_sets(5, 3);
        for (int r = 0; r < FACE_BULK; ++r) {
            as_solve(S_[5], nullptr, nullptr, 2, r == 0);
            hipLaunchKernelGGL(k_face_dual_finish, dim3(1), dim3(1024), 0, h->stream, A, S_[5], FACE_TOL_M);
            as_read();
            if (h->verbose) std::fprintf(stderr, "[asm] face dual %d: nH %d nF %d viol %d\n", r, as_nH, as_nF, h->h_ascnt[AC_NVIOL]);
            if (h->h_ascnt[AC_NVIOL] == 0) { okd = true; break; }
        }
        if (okd) { dcopy(d_yf, A.y, M); dcopy(d_zf, A.z, n); }
        // ---- primal: bulk rounds
        bool okp = false;
        if (okd) {
            as_copy_sets(4, 3);
            for (int r = 0; r < FACE_BULK; ++r) {
                as_solve(S_[4], nullptr, nullptr, 1, r == 0);
                hipLaunchKernelGGL(k_face_primal_finish, dim3(1), dim3(1024), 0, h->stream, A, S_[4], S_[3], TOL_P, FACE_TOL_M, 0);
                as_read();
                if (h->verbose) std::fprintf(stderr, "[asm] face primal bulk %d: nH %d nF %d viol %d rel %d hres %.2e\n", r, as_nH, as_nF, h->h_ascnt[AC_NVIOL], h->h_ascnt[AC_NREL], h->h_asscal[AS_HARDRES]);
                if (h->h_asscal[AS_HARDRES] > TOL_P) break;            // over-determined working set
                if (h->h_ascnt[AC_NVIOL] > 0) continue;
                if (h->h_ascnt[AC_NREL] == 0) { okp = true; break; }
            }
            // ---- primal: anchored method in the null space of the partition (oracle: _face_primal_anchored) - one factorisation,
            // one solve with it per added constraint, the k x k matrix Z'Z of the added constraints on the host
            if (!okp && face_primal_anchored()) {
                // the answer is the least-norm point of the FINAL working set, computed like any other (fresh factorisation)
                as_solve(S_[4], nullptr, nullptr, 1);
                hipLaunchKernelGGL(k_face_primal_finish, dim3(1), dim3(1024), 0, h->stream, A, S_[4], S_[3], TOL_P, FACE_TOL_M, 1);
                as_read();
                okp = h->h_asscal[AS_HARDRES] <= TOL_P && h->h_ascnt[AC_NVIOL] == 0;
            }
        }
        part_factor = false;
        if (okp && okd) {
            dcopy(A.y, d_yf, M); dcopy(A.z, d_zf, n);      // (p, s, act) are the primal stage's last solve; y, z come from the dual stage
            hipLaunchKernelGGL(k_face_kkt, dim3(1), dim3(1024), 0, h->stream, A, S_[5]);
            as_read();
            h->stats.kkt_pr = h->h_asscal[AS_PR];
            h->stats.kkt_du = h->h_asscal[AS_DU];
            if (h->verbose) std::fprintf(stderr, "[asm] face kkt pr %.2e du %.2e\n", h->h_asscal[AS_PR], h->h_asscal[AS_DU]);
            if (h->h_asscal[AS_PR] <= TOL_P && h->h_asscal[AS_DU] <= TOL_D) { final_sets = 4; return 2; }
        }
        dcopy(A.p, d_p0, n); dcopy(A.s, d_s0, ns); dcopy(A.y, d_y0, M); dcopy(A.act, d_act0, M); dcopy(A.z, d_z0, n);
        final_sets = 3;
        return 1;
    }


    // oracle: phase1_infeasible - elastic LP over the same rows/box; its optimal multipliers are a Farkas
    // certificate of the original LP, verified rigorously before INFEASIBLE is reported.
    bool phase1_infeasible() {
        SLP saved = lp;
        IpmState saved_ip = ip;
        lp.ns = h->ns;
        std::fill(lp.q.begin(), lp.q.end(), 0.0);
        lp.w.assign(lp.ns, 1.0);
        lp.slo.assign(lp.ns, 0.0);
        lp.scale_q = 1.0;
        ipm_init();
        ipm_run(1e-8, IPM_MAXIT);
        vec y1;
        down(y1, P.y, lp.M);
        HIPCHK(hipStreamSynchronize(h->stream));
        int its = ip.iters;
        lp = saved;
        ip = saved_ip;
        ipm_upload_lp();              // the original LP again (the interrupted iterate is not resumed after phase 1)
        h->stats.ipm_iters += its;
        return farkas_margin(y1) > 1e-9;
    }

    // oracle: solve_scaled
    double t_warm = 0, t_ipm = 0, t_polish = 0;
    bool snap_e = false;      // the snapshot of the best iterate holds the null-space form's component e
    int solve_scaled(const ActiveSet* warm, SolveHint& hint) {
        int st = solve_scaled_impl(warm, hint);
        if (h->verbose) std::fprintf(stderr, "[asm] phases: warm %.2f ms, ipm %.2f ms (%d its), polish %.2f ms, path %d\n", t_warm, t_ipm, ip.iters, t_polish, h->stats.path);
        return st;
    }
    int solve_scaled_impl(const ActiveSet* warm, SolveHint& hint) {
        const int64_t n = lp.n, M = lp.M, ns = lp.ns;
        h->stats.path = -1;
        h->stats.polished = 1;
        h->stats.restored = 0;
        cur_hint = &hint;
        btag = 30;
        asmb::barrier(btag);
        ipm_upload_lp();
        as_begin_lp();
        // null-space basis of the equality rows (oracle: solve_scaled): made first, the active-set solves of the warm attempt and of the
        // polish go through it as well as the interior-point iterations
        ns_lp = false;
        h->stats.ns_dim = 0;
        h->stats.ns_cold = 0;
        if (ns_applicable()) {
            const size_t had = hint.ns_J.size();
            const double t0 = now_ms();
            ns_lp = ns_setup();
            if (ns_lp) ns_lp_vectors();
            h->stats.ns_dim = ns_lp ? ns_k : 0;
            h->stats.ns_cold = (ns_lp && (had == 0 || ns_was_cold)) ? 1 : 0;
            if (h->verbose) {
                HIPCHK(hipStreamSynchronize(h->stream));
                std::fprintf(stderr, "[asm] null-space set-up: %s, k = %d, %s basis columns, %.2f ms\n", ns_lp ? "ok" : "not usable", ns_k, ns_was_cold ? "fresh" : "retained", now_ms() - t0);
            }
        }
        if (warm && warm->valid && (int64_t)warm->rowst.size() == M && (int64_t)warm->bst.size() == n && (int64_t)warm->sst.size() == ns) {
            // attempt when the last two LPs ended on the same sets or the back-off has run out (oracle: solve_scaled)
            if (!hint.stable && hint.warm_skip > 0) {
                hint.warm_skip -= 1;
            } else {
                double t0 = now_ms();
                btag = 40 - 60;
                as_upload_sets(*warm, 0);
                bool okw = eqp_loop(d_zero, nullptr, 1);
                t_warm += now_ms() - t0;
                if (okw) { hint.warm_fail = 0; hint.warm_skip = 0; h->stats.path = 0; return ASM_OPTIMAL; }
                hint.warm_fail = std::min(hint.warm_fail + 1, WARM_BACKOFF_MAX);
                hint.warm_skip = (1 << hint.warm_fail) - 1;
                hint.stable = false;
            }
        }
        const bool prefer_ref = hint.prefer_ref;
        btag = 90;
        asmb::barrier(btag);
        ipm_init();
        static const double tol0_env = [] { const char* v = std::getenv("ASM_IPM_TOL0"); return v ? std::atof(v) : 3e-10; }();      // (measurement knob: tolerance of the first identification)
        const double tols[3] = {tol0_env, 0.1 * tol0_env, 1e-12};      // oracle: IPM_STAGES
        const int more[3] = {IPM_MAXIT, 6, 6};
        bool have_sets = false;
        double best_m = INF, m_last = INF;
        bool have_snap = false;
        for (int stage = 0; stage < 3; ++stage) {
            double t0 = now_ms();
            btag = 100 + 100 * stage;
            int st = ipm_run(tols[stage], more[stage]);
            asmb::barrier(btag + 50);
            t_ipm += now_ms() - t0;
            h->stats.ipm_iters = ip.iters;
            h->stats.col_iters = ip.col_iters;
            h->stats.ns_iters = ip.ns_iters;
            h->stats.ipm_pinf = ip.pinf; h->stats.ipm_dinf = ip.dinf; h->stats.ipm_gap = ip.gap;
            if (st == ASM_INFEASIBLE) { h->stats.path = 6; return ASM_INFEASIBLE; }
            {
                // best-iterate safeguard, first half (oracle: solve_scaled): the iterate at the end of the best stage so far is kept
                m_last = std::max(ip.pinf, std::max(ip.dinf, ip.gap));
                double* e_ns = (ip.ns_ok && ip.ns_e_ready) ? nsv(14) : nullptr;
                if (m_last < best_m) {
                    hipLaunchKernelGGL(k_ipm_snapshot, dim3(grid_all()), dim3(256), 0, h->stream, P, h->d_ipm_snap, e_ns, h->ldn, h->Mp, h->nsp, 0);
                    best_m = m_last; have_snap = true; snap_e = e_ns != nullptr;
                }
            }
            if (st == ASM_OTHER && stage == 0) {
                // the IPM is only the identifier: a jammed / slow run that is already close is still handed to
                // the active-set solve, whose LP optimality test decides (oracle: solve_scaled)
                if (ip.pinf <= 1e-3 && ip.dinf <= 1e-3 && ip.gap <= 1e-4) {
                    identify_dev(3);
                    have_sets = true;
                    as_copy_sets(0, 3);
                    if (eqp_loop(d_zero, nullptr, 3)) { h->stats.path = 8; return ASM_OPTIMAL; }
                }
                if (lp.ns == 0 && phase1_infeasible()) { h->stats.path = 7; return ASM_INFEASIBLE; }
                break;
            }
            double t1 = now_ms();
            identify_dev(3);
            if (h->verbose) { HIPCHK(hipStreamSynchronize(h->stream)); std::fprintf(stderr, "[asm] stage %d identify %.2f ms\n", stage, now_ms() - t1); }
            have_sets = true;
            bool tried_ln = false;
            if (ns_lp) {
                // the least-norm polish in reduced coordinates costs two solves with the factor of S0: tried first whatever the last LP
                // needed (oracle: solve_scaled)
                as_copy_sets(0, 3);
                const bool okn = eqp_loop(d_zero, nullptr, 2);
                if (okn) { t_polish += now_ms() - t1; hint.prefer_ref = false; h->stats.path = 1 + stage; return ASM_OPTIMAL; }
                tried_ln = true;
            }
            if (prefer_ref) {
                // non-unique optimum expected: the canonical pair as soon as the partition passes the LP optimality test
                // (oracle: solve_scaled)
                const int how = face_polish();
                t_polish += now_ms() - t1;
                if (how == 2) { h->stats.path = 4; return ASM_OPTIMAL; }
                if (how == 1) { h->stats.path = 9; return ASM_OPTIMAL; }
                continue;
            }
            if (tried_ln) { t_polish += now_ms() - t1; continue; }
            as_copy_sets(0, 3);
            bool okp = eqp_loop(d_zero, nullptr, 2);
            t_polish += now_ms() - t1;
            if (okp) { h->stats.path = 1 + stage; return ASM_OPTIMAL; }
        }
        btag = 400;
        asmb::barrier(btag);
        if (have_snap && have_sets && m_last > IPM_DEGRADE * best_m) {
            // best-iterate safeguard, second half (oracle: solve_scaled): no stage ended in a successful polish and the last one ended IPM_DEGRADE
            // times worse than the best - the best iterate comes back, the final attempts run on it and on the partition identified from it
            hipLaunchKernelGGL(k_ipm_snapshot, dim3(grid_all()), dim3(256), 0, h->stream, P, h->d_ipm_snap, snap_e ? nsv(14) : (double*)nullptr, h->ldn, h->Mp, h->nsp, 1);
            ipm_measures();
            if (h->verbose) std::fprintf(stderr, "[asm] last stage ended %.1e against %.1e at best: best iterate restored (pinf %.3e dinf %.3e gap %.3e)\n", m_last, best_m, ip.pinf, ip.dinf, ip.gap);
            h->stats.ipm_pinf = ip.pinf; h->stats.ipm_dinf = ip.dinf; h->stats.ipm_gap = ip.gap;
            h->stats.restored = 1;
            identify_dev(3);
        }
        if (have_sets) {
            double t1 = now_ms();
            // non-unique optimum: canonical (least-norm) pair of the optimal faces the partition describes (oracle: face_polish)
            const int how = prefer_ref ? 0 : face_polish();
            t_polish += now_ms() - t1;
            if (how == 2) { hint.prefer_ref = true; h->stats.path = 4; return ASM_OPTIMAL; }
            if (how == 1) { hint.prefer_ref = true; h->stats.path = 9; return ASM_OPTIMAL; }
            // the partition is not optimal as it stands: bulk corrections from the iterate's projection
            as_copy_sets(0, 3);
            bool okr = eqp_loop(d_pref, P.y, 2);
            if (okr) {
                // the corrected working set passes the LP optimality test, i.e. it describes a face of optimal points: return that
                // face's canonical pair (a function of the discrete set) rather than the projection of the iterate onto it
                // (oracle: solve_scaled).  face_polish leaves the projection in place when its own stages do not succeed.
                hint.prefer_ref = true;
                if (final_sets != 3) as_copy_sets(3, final_sets);
                const int eqp_sets = final_sets;
                const int how2 = face_polish();
                if (how2 == 0) final_sets = eqp_sets;
                h->stats.path = how2 == 2 ? 4 : 9;
                return ASM_OPTIMAL;
            }
            hint.prefer_ref = false;
            if (prefer_ref && !ns_lp) {                     // the least-norm polish has not been tried on this LP yet
                as_copy_sets(0, 3);
                if (eqp_loop(d_zero, nullptr, 2)) { h->stats.path = 3; return ASM_OPTIMAL; }
            }
        }
        // last resort (oracle: solve_scaled, 'ipm-conv'): an iterate converged to IPM_ACCEPT in all three measures is an optimal point of the
        // LP to that accuracy; it is handed out through the active-set arena (clipped into the box, partition of the last identification)
        bool conv = have_sets && ip.pinf <= IPM_ACCEPT && ip.dinf <= IPM_ACCEPT_DUAL && ip.gap <= IPM_ACCEPT;
        if (!conv && have_sets && have_snap && best_m <= IPM_ACCEPT) {
            // ... or the best stage end did (the last iterations drifted out of the acceptance, but by less than IPM_DEGRADE): that iterate then
            hipLaunchKernelGGL(k_ipm_snapshot, dim3(grid_all()), dim3(256), 0, h->stream, P, h->d_ipm_snap, snap_e ? nsv(14) : (double*)nullptr, h->ldn, h->Mp, h->nsp, 1);
            ipm_measures();
            h->stats.ipm_pinf = ip.pinf; h->stats.ipm_dinf = ip.dinf; h->stats.ipm_gap = ip.gap;
            h->stats.restored = 1;
            identify_dev(3);
            conv = true;
        }
        if (conv) {
            hipLaunchKernelGGL(k_as_clip0, dim3((unsigned)((lp.n + 255) / 256)), dim3(256), 0, h->stream, A.lb, A.ub, (const double*)P.p, A.p, lp.n);
            dcopy(A.y, P.y, lp.M);
            if (lp.ns) {
                hipLaunchKernelGGL(k_as_smax, dim3((unsigned)((lp.ns + 255) / 256)), dim3(256), 0, h->stream, (const double*)P.s, A.slo, A.s, lp.ns);
                hipLaunchKernelGGL(k_as_sl_values, dim3((unsigned)((lp.M + 255) / 256 + 1)), dim3(256), 0, h->stream, A);      // the tail kernel must not recompute slacks from a stale working set
            }
            dev.gemv_n_dev(h->d_Ah, A.p, A.t);
            dev.gemv_t_dev(h->d_Ah, A.y, A.tN);
            hipLaunchKernelGGL(k_as_finish, dim3(1), dim3(1024), 0, h->stream, A, S_[3], S_[1], S_[2], 0, TOL_P, TOL_D);
            as_read();
            final_sets = 3;
            h->stats.path = 10;
            return ASM_OPTIMAL;
        }
        // no active-set solve passed the LP optimality test: the interior iterate is not returned as a solution
        // (status OTHER; the SLP caller stops with a warning, slp_line_search.jl:127-133)
        h->stats.path = 5;
        h->stats.polished = 0;
        return ASM_OTHER;
    }

};

// =====================================================================================================
// formulation + extraction (subproblem.jl:229-542)
// =====================================================================================================
int row_kind(double lb, double ub) {
    if (lb == ub) return 0;
    if (lb != -INF && ub != INF && lb < ub) return 2;
    if (lb != -INF) return 1;
    if (ub != INF) return -1;
    return 9;
}

void free_device(asm_handle* h) {
    auto F = [](void* p) { if (p) (void)hipFree(p); };
    F(h->d_perm); F(h->d_ustart); F(h->d_uoff); F(h->d_adjoff);
    F(h->d_dE); F(h->d_J); F(h->d_Ah); F(h->d_S); F(h->d_c); F(h->d_rho); F(h->d_theta); F(h->d_diag); F(h->d_diag0);
    F(h->d_vecN); F(h->d_vecM); F(h->d_vecM2); F(h->d_partial); F(h->d_idx); F(h->d_Linv); F(h->d_Binv); F(h->d_wpart); F(h->d_BinvT); F(h->d_wt);
    h->d_Binv = h->d_wpart = h->d_BinvT = h->d_wt = nullptr; F(h->d_ipm); F(h->d_ipm_i); F(h->d_nz);
    F(h->d_ipm_snap); h->d_ipm_snap = nullptr;
    h->d_nz = nullptr; h->nz_valid = false; h->nz_frac_cache[0] = h->nz_frac_cache[1] = -1.0;
    F(h->d_idxI); F(h->d_rdI); F(h->d_rce); F(h->d_rze); F(h->d_sdiag);
    F(h->d_redpart); F(h->d_redcnt); h->d_redpart = nullptr; h->d_redcnt = nullptr;
    F(h->d_AhTg); h->d_AhTg = nullptr; h->ahTg_valid = false;
    h->d_idxI = nullptr; h->d_rdI = h->d_rce = h->d_rze = h->d_sdiag = nullptr;
    F(h->d_AhT); F(h->d_cdinv); F(h->d_cth); F(h->d_cu); F(h->d_ct); F(h->d_cv); F(h->d_cw); F(h->d_nzT);
    h->d_AhT = h->d_cdinv = h->d_cth = h->d_cu = h->d_ct = h->d_cv = h->d_cw = nullptr; h->d_nzT = nullptr;
    h->col_capable = h->ahT_valid = h->nzT_valid = false;
    F(h->d_sp_ptr); F(h->d_sp_col); F(h->d_sc_ptr); F(h->d_sc_row); F(h->d_sc_pos); F(h->d_sp_off); F(h->d_spv_Ah); F(h->d_spv_J);
    h->d_sp_ptr = h->d_sp_col = h->d_sc_ptr = h->d_sc_row = h->d_sc_pos = nullptr;
    h->d_sp_off = nullptr; h->d_spv_Ah = h->d_spv_J = nullptr;
    h->sp_ok = h->spv_Ah_valid = h->spv_J_valid = false; h->sp_nnz = 0;
    if (h->h_scal) (void)hipHostFree(h->h_scal);
    if (h->h_seq) (void)hipHostFree(h->h_seq);
    h->d_ipm = nullptr; h->d_ipm_i = nullptr; h->h_scal = nullptr; h->h_seq = nullptr; h->d_hscal = nullptr; h->d_hseq = nullptr;
    F(h->d_pflags); F(h->d_ptmo);
    h->d_pflags = h->d_ptmo = nullptr;
    for (void* q : h->ns_bufs) F(q);
    h->ns_bufs.clear();
    h->ns_cap = false; h->ns_kcap = 0; h->ns_ccap = 0; h->ns_Zk = 0; h->ns_fC = FacBuf(); h->d_nsqi = nullptr; h->d_nsq = nullptr;
    h->d_nsEidx = h->d_nsEpos = h->d_nsIidx = h->d_nsIpos = h->d_nsJ = h->d_nscnt = nullptr;
    h->d_nsS0pairs = nullptr; h->ns_npairs = 0;
    h->ns_f0 = FacBuf(); h->ns_fN = FacBuf();
    h->d_nsLt = h->d_nsR = h->d_nsX = h->d_nsG = h->d_nsth = h->d_nsFm = h->d_nsv = h->d_nsYt = h->d_nsN0 = h->d_nsZT = nullptr;
    F(h->d_as); F(h->d_as_i);
    if (h->h_ascnt) (void)hipHostFree(h->h_ascnt);
    if (h->h_asscal) (void)hipHostFree(h->h_asscal);
    h->d_as = nullptr; h->d_as_i = nullptr; h->h_ascnt = nullptr; h->h_asscal = nullptr;
    for (void* q : h->ev_bufs) F(q);
    h->ev_bufs.clear();
    h->ev_ready = false; h->d_ev_ipar = nullptr;
    h->d_ev_dpar = h->d_ev_x = h->d_ev_xt = h->d_ev_df = h->d_ev_E = h->d_ev_Et = h->d_ev_f = h->d_ev_vecs = nullptr;
    if (h->h_ev) (void)hipHostFree(h->h_ev);
    h->h_ev = nullptr;
    F(h->d_Zbuf); F(h->d_nsu); F(h->d_nsdots);
    if (h->h_nsdots) (void)hipHostFree(h->h_nsdots);
    h->d_Zbuf = h->d_nsu = h->d_nsdots = h->h_nsdots = nullptr;
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    if (h->h_up) (void)hipHostFree(h->h_up);
    h->h_up = nullptr;
    if (h->h_dl) (void)hipHostFree(h->h_dl);
    F(h->d_dl);
    h->h_dl = h->d_dl = nullptr;
    h->d_perm = h->d_ustart = h->d_uoff = h->d_adjoff = nullptr;
    h->d_dE = h->d_J = h->d_Ah = h->d_S = h->d_c = h->d_rho = h->d_theta = h->d_diag = h->d_diag0 = nullptr;
    h->d_vecN = h->d_vecM = h->d_vecM2 = h->d_partial = nullptr;
    h->d_Linv = nullptr;
    h->d_idx = nullptr;
    h->h_pin = nullptr;
}

void check_panel_timeout(asm_handle* h) {
    if (!h->fused_panel || !h->d_ptmo) return;
    unsigned tmo = 0;
    HIPCHK(hipMemcpy(&tmo, h->d_ptmo, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (tmo != 0) {
        HIPCHK(hipMemset(h->d_ptmo, 0, sizeof(unsigned)));      // reported once: the handle stays usable
        throw HipError("k_chol_panel: a workgroup timed out waiting for a producer (grid not resident?)");
    }
}

void do_setup(asm_handle* h, int64_t n, int64_t m, int64_t nnz, const int64_t* j_row, const int64_t* j_col, const double* c_lb,
              const double* c_ub, const double* v_lb, const double* v_ub) {
    HIPCHK(hipSetDevice(h->device));
    free_device(h);
    h->setup_done = false;
    h->inputs_ready = false;
    h->J_valid = false;
    h->n = n; h->m = m; h->nnz = nnz;
    h->c_lb.assign(c_lb, c_lb + m); h->c_ub.assign(c_ub, c_ub + m);
    h->v_lb.assign(v_lb, v_lb + n); h->v_ub.assign(v_ub, v_ub + n);
    h->kind.resize(m);
    h->adj.clear();
    for (int64_t i = 0; i < m; ++i) {
        h->kind[i] = row_kind(c_lb[i], c_ub[i]);
        if (h->kind[i] == 9) throw std::invalid_argument("free constraint row (c_lb=-Inf, c_ub=+Inf) is not representable");
        if (h->kind[i] == 2) h->adj.push_back(i);
    }
    h->nadj = (int64_t)h->adj.size();
    h->M = m + h->nadj;
    h->Mp = round_up(std::max<int64_t>(h->M, 1), 16);
    h->ldn = round_up(n, 32);              // multiple of the SYRK k-chunk (ASM_KC)
    h->rtype.assign(h->M, 0);
    for (int64_t i = 0; i < m; ++i) h->rtype[i] = h->kind[i] == 0 ? 0 : (h->kind[i] == -1 ? -1 : 1);
    for (int64_t k = 0; k < h->nadj; ++k) h->rtype[m + k] = -1;
    // slack layout (subproblem.jl:83-112): one per row, two when both bounds are finite
    h->srow.clear(); h->scoef.clear(); h->sown.clear(); h->nslack.assign(m, 1);
    std::vector<int64_t> adjpos(m, -1);
    for (int64_t k = 0; k < h->nadj; ++k) adjpos[h->adj[k]] = k;
    for (int64_t i = 0; i < m; ++i) {
        h->nslack[i] = (c_lb[i] > -INF && c_ub[i] < INF) ? 2 : 1;
        int kd = h->kind[i];
        if (kd == 0) { h->srow.push_back((int)i); h->scoef.push_back(1.0); h->srow.push_back((int)i); h->scoef.push_back(-1.0); h->sown.push_back(i); h->sown.push_back(i); }
        else if (kd == 2) { h->srow.push_back((int)i); h->scoef.push_back(1.0); h->srow.push_back((int)(m + adjpos[i])); h->scoef.push_back(-1.0); h->sown.push_back(i); h->sown.push_back(i); }
        else if (kd == 1) { h->srow.push_back((int)i); h->scoef.push_back(1.0); h->sown.push_back(i); }
        else { h->srow.push_back((int)i); h->scoef.push_back(-1.0); h->sown.push_back(i); }
    }
    h->ns = (int64_t)h->srow.size();

    // assembly plan: stable sort of the COO entries by (row, col)
    for (int64_t k = 0; k < nnz; ++k)
        if (j_row[k] < 1 || j_row[k] > m || j_col[k] < 1 || j_col[k] > n) throw std::invalid_argument("j_row/j_col out of range (1-based)");
    bool dense = (nnz == m * n) && h->nadj == 0 && nnz > 0;
    if (dense)
        for (int64_t k = 0; k < nnz && dense; ++k) dense = (j_row[k] - 1) * n + (j_col[k] - 1) == k;
    h->dense_fast = dense;
    std::vector<int64_t> perm(nnz), ustart, uoff, adjoff;
    std::iota(perm.begin(), perm.end(), 0);
    if (!dense) {
        std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) {
            int64_t ka = (j_row[a] - 1) * n + (j_col[a] - 1), kb = (j_row[b] - 1) * n + (j_col[b] - 1);
            return ka < kb;
        });
        int64_t prev = -1;
        for (int64_t t = 0; t < nnz; ++t) {
            int64_t k = perm[t];
            int64_t r = j_row[k] - 1, c = j_col[k] - 1, key = r * n + c;
            if (key != prev) {
                ustart.push_back(t);
                uoff.push_back(r * h->ldn + c);
                adjoff.push_back(adjpos[r] >= 0 ? (m + adjpos[r]) * h->ldn + c : -1);
                prev = key;
            }
        }
        ustart.push_back(nnz);
    }
    h->nu = dense ? nnz : (int64_t)uoff.size();
    // sparse pattern (CSR + CSC) of the LP matrix rows [0, M): the unique Jacobian entries plus the copies of the range
    // rows; used by the matrix-vector products when the fill is below 1/16
    std::vector<int> sp_ptr, sp_col, sc_ptr, sc_row, sc_pos;
    std::vector<int64_t> sp_off;
    {
        int64_t nadjent = 0;
        for (int64_t v : adjoff) nadjent += v >= 0;
        const int64_t nnzS = (int64_t)uoff.size() + nadjent;
        if (!dense && nnzS > 0 && nnzS * 16 <= h->M * n && nnzS < (int64_t)1 << 30) {
            sp_off.reserve(nnzS);
            for (int64_t v : uoff) sp_off.push_back(v);                 // sorted by (row, col), rows < m
            for (int64_t v : adjoff) if (v >= 0) sp_off.push_back(v);   // rows m.., same order
            sp_ptr.assign(h->M + 1, 0);
            sp_col.resize(nnzS);
            for (int64_t k = 0; k < nnzS; ++k) {
                sp_ptr[sp_off[k] / h->ldn + 1] += 1;
                sp_col[k] = (int)(sp_off[k] % h->ldn);
            }
            for (int64_t i = 0; i < h->M; ++i) sp_ptr[i + 1] += sp_ptr[i];
            sc_ptr.assign(n + 1, 0);
            for (int64_t k = 0; k < nnzS; ++k) sc_ptr[sp_col[k] + 1] += 1;
            for (int64_t j = 0; j < n; ++j) sc_ptr[j + 1] += sc_ptr[j];
            std::vector<int> fill(sc_ptr.begin(), sc_ptr.end() - 1);
            sc_row.resize(nnzS); sc_pos.resize(nnzS);
            for (int64_t k = 0; k < nnzS; ++k) {                          // stable: rows ascending inside a column
                int q = fill[sp_col[k]]++;
                sc_row[q] = (int)(sp_off[k] / h->ldn);
                sc_pos[q] = (int)k;
            }
            h->sp_nnz = nnzS;
        }
    }

    dmalloc(&h->d_dE, nnz);
    dmalloc(&h->d_J, h->Mp * h->ldn);
    dmalloc(&h->d_Ah, h->Mp * h->ldn);
    dmalloc(&h->d_S, h->Mp * h->Mp);
    dmalloc(&h->d_c, h->ldn); dmalloc(&h->d_rho, h->Mp); dmalloc(&h->d_theta, h->ldn);
    dmalloc(&h->d_diag, h->Mp); dmalloc(&h->d_diag0, h->Mp);
    dmalloc(&h->d_vecN, h->ldn); dmalloc(&h->d_vecM, h->Mp); dmalloc(&h->d_vecM2, h->Mp);
    dmalloc(&h->d_partial, (int64_t)ASM_TMAXCHUNKS * h->ldn);
    dmalloc(&h->d_idx, h->Mp);
    dmalloc(&h->d_Linv, (h->Mp / ASM_NB + 1) * ASM_NB * ASM_NB);
    dmalloc(&h->d_pflags, ASM_PNL_FLAGS);
    dmalloc(&h->d_ptmo, 4);
    HIPCHK(hipMemsetAsync(h->d_ptmo, 0, 4 * sizeof(unsigned), h->stream));
    HIPCHK(hipMemsetAsync(h->d_pflags, 0, ASM_PNL_FLAGS * sizeof(unsigned), h->stream));
    h->wb = h->M > 1536 ? 1024 : 512;      // wide-block width of the triangular solves (k_wtrsv_*<WB>)
    if (h->M >= RED_MIN_M) {
        dmalloc(&h->d_idxI, h->Mp); dmalloc(&h->d_rdI, h->Mp); dmalloc(&h->d_rce, h->Mp); dmalloc(&h->d_rze, h->Mp); dmalloc(&h->d_sdiag, h->Mp);
    }
    // column form of the restoration-phase Newton system (every row owns a slack column there): n x n instead of M x M
    h->col_capable = h->M >= COL_MIN_M && (double)n <= COL_MAX_RATIO * (double)h->M && n <= h->Mp;
    if (h->col_capable) {
        h->ldT = round_up(h->M, 32);
        dmalloc(&h->d_AhT, n * h->ldT);
        HIPCHK(hipMemsetAsync(h->d_AhT, 0, n * h->ldT * sizeof(double), h->stream));
        dmalloc(&h->d_cdinv, h->ldT); dmalloc(&h->d_cth, h->ldn); dmalloc(&h->d_cu, h->Mp); dmalloc(&h->d_ct, h->ldn);
        dmalloc(&h->d_cv, h->ldn); dmalloc(&h->d_cw, h->Mp);
        HIPCHK(hipMemsetAsync(h->d_cdinv, 0, h->ldT * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(h->d_cth, 0, h->ldn * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(h->d_cu, 0, h->Mp * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(h->d_ct, 0, h->ldn * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(h->d_cv, 0, h->ldn * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(h->d_cw, 0, h->Mp * sizeof(double), h->stream));
        dmalloc(&h->d_nzT, (n / 32 + 2) * (h->ldT / ASM_KC + 1));
    }
    dmalloc(&h->d_Binv, (h->Mp / h->wb + 1) * (int64_t)h->wb * h->wb);
    dmalloc(&h->d_BinvT, (h->Mp / h->wb + 1) * (int64_t)h->wb * h->wb);
    dmalloc(&h->d_wpart, (h->Mp / ASM_WBROWS + 2) * (int64_t)1024);
    dmalloc(&h->d_wt, 1024);
    h->nz_half = (h->Mp / 32 + 1) * (h->ldn / ASM_KC + 1);
    dmalloc(&h->d_nz, 2 * h->nz_half);
    if (h->sp_nnz > 0) {
        dmalloc(&h->d_sp_ptr, h->M + 1); dmalloc(&h->d_sp_col, h->sp_nnz); dmalloc(&h->d_sp_off, h->sp_nnz);
        dmalloc(&h->d_sc_ptr, n + 1); dmalloc(&h->d_sc_row, h->sp_nnz); dmalloc(&h->d_sc_pos, h->sp_nnz);
        dmalloc(&h->d_spv_Ah, h->sp_nnz); dmalloc(&h->d_spv_J, h->sp_nnz);
        HIPCHK(hipMemcpy(h->d_sp_ptr, sp_ptr.data(), sp_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_sp_col, sp_col.data(), sp_col.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_sp_off, sp_off.data(), sp_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_sc_ptr, sc_ptr.data(), sc_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_sc_row, sc_row.data(), sc_row.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_sc_pos, sc_pos.data(), sc_pos.size() * sizeof(int), hipMemcpyHostToDevice));
        h->sp_ok = true;
    }
    h->nsp = round_up(std::max<int64_t>(h->ns, 1), 16);
    {
        int64_t nd = 24 * h->ldn + 23 * h->Mp + 16 * h->nsp + 64;
        dmalloc(&h->d_ipm, nd);
        HIPCHK(hipMemsetAsync(h->d_ipm, 0, nd * sizeof(double), h->stream));
        dmalloc(&h->d_ipm_snap, 6 * h->ldn + 3 * h->Mp + 3 * h->nsp);
        dmalloc(&h->d_ipm_i, 3 * h->Mp + h->nsp);
        std::vector<int> iv(3 * h->Mp + h->nsp, -1);
        for (int64_t i = 0; i < h->M; ++i) iv[i] = h->rtype[i];
        for (int64_t k = 0; k < h->ns; ++k) {
            int r_ = h->srow[k];
            if (iv[h->Mp + r_] < 0) iv[h->Mp + r_] = (int)k; else iv[2 * h->Mp + r_] = (int)k;
            iv[3 * h->Mp + k] = r_;
        }
        HIPCHK(hipMemcpy(h->d_ipm_i, iv.data(), iv.size() * sizeof(int), hipMemcpyHostToDevice));
        dmalloc(&h->d_redpart, IPM_RED_MAXWG * IPM_RED_SLOTS);
        dmalloc(&h->d_redcnt, 4);
        HIPCHK(hipMemsetAsync(h->d_redcnt, 0, 4 * sizeof(unsigned), h->stream));
        HIPCHK(hipHostMalloc((void**)&h->h_scal, 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostMalloc((void**)&h->h_seq, 64, hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer((void**)&h->d_hscal, h->h_scal, 0));
        HIPCHK(hipHostGetDevicePointer((void**)&h->d_hseq, h->h_seq, 0));
        *h->h_seq = 0;
        h->scal_seq = 0;
        const int64_t nas = 17 * h->ldn + 17 * h->Mp + 4 * h->nsp + 64, nasi = 6 * (h->Mp + h->ldn + h->nsp) + 3 * h->Mp + 2 * h->ldn + 64;
        dmalloc(&h->d_as, nas);
        HIPCHK(hipMemsetAsync(h->d_as, 0, nas * sizeof(double), h->stream));
        dmalloc(&h->d_as_i, nasi);
        HIPCHK(hipMemsetAsync(h->d_as_i, 0, nasi * sizeof(int), h->stream));
        HIPCHK(hipHostMalloc((void**)&h->h_ascnt, 64 * sizeof(int)));
        HIPCHK(hipHostMalloc((void**)&h->h_asscal, 64 * sizeof(double)));
        dmalloc(&h->d_Zbuf, (int64_t)(FACE_STEPS + 1) * h->ldn);
        HIPCHK(hipMemsetAsync(h->d_Zbuf, 0, (int64_t)(FACE_STEPS + 1) * h->ldn * sizeof(double), h->stream));
        dmalloc(&h->d_nsu, FACE_STEPS + 8);
        dmalloc(&h->d_nsdots, FACE_STEPS + 8);
        HIPCHK(hipHostMalloc((void**)&h->h_nsdots, (FACE_STEPS + 8) * sizeof(double)));
    }
    // null-space form of the normal-phase Newton system: static part (index lists, factor of S0, work vectors); the buffers
    // sized by the null-space dimension k are allocated by the first LP that uses the form (Solver::ns_reserve)
    {
        int nE = 0;
        for (int64_t i = 0; i < h->M; ++i) nE += h->rtype[i] == 0;
        h->ns_cap = h->sp_ok && nE >= NS_MIN_E && (double)(n - nE) <= NS_MAX_RATIO * (double)h->M && n <= h->Mp && h->M < (int64_t)1 << 30;
        if (h->ns_cap) {
            h->ns_nE = nE; h->ns_nI = (int)(h->M - nE);
            h->ns_nEp = (int)round_up(nE, 32); h->ns_nIp = (int)round_up(std::max(h->ns_nI, 1), 32);
            h->ns_ldg = h->ldn + h->ns_nIp;
            std::vector<int> eidx, epos(h->M, -1), iidx, ipos(h->M, -1);
            for (int64_t i = 0; i < h->M; ++i) {
                if (h->rtype[i] == 0) { epos[i] = (int)eidx.size(); eidx.push_back((int)i); }
                else { ipos[i] = (int)iidx.size(); iidx.push_back((int)i); }
            }
            auto ialloc = [&](const std::vector<int>& v, int64_t cnt) {
                int* d = nullptr;
                dmalloc(&d, cnt);
                h->ns_bufs.push_back((void*)d);
                HIPCHK(hipMemset(d, 0, std::max<int64_t>(cnt, 1) * sizeof(int)));
                if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
                return d;
            };
            // the equality rows in reverse Cuthill-McKee order of their coupling graph: S0 = A_EF A_EF' and its factor are banded in that order
            int s0_band = 0;
            std::vector<int> s0_pairs;
            {
                const std::vector<int> ord = rcm_order(eidx, sp_ptr, sp_col, h->ldn, &s0_band, &s0_pairs);
                std::vector<int> e2(nE);
                for (int q = 0; q < nE; ++q) e2[q] = eidx[ord[q]];
                eidx.swap(e2);
                for (int q = 0; q < nE; ++q) epos[eidx[q]] = q;
            }
            h->ns_eidx_h = eidx;
            h->d_nsEidx = ialloc(eidx, nE); h->d_nsEpos = ialloc(epos, h->M); h->d_nsIidx = ialloc(iidx, h->ns_nI); h->d_nsIpos = ialloc(ipos, h->M);
            h->d_nscnt = ialloc({}, 16);
            const int f0_band = (std::getenv("ASM_HIP_NO_BAND") || 2 * (int64_t)s0_band >= nE) ? 0 : std::max(s0_band, 1);
            ns_alloc_factor(h, h->ns_f0, h->ns_nEp, f0_band);
            h->ns_f0.band = f0_band;
            if (h->ns_f0.band > 0) {            // S0 is then built entry by entry from its structural pattern (k_ns_s0_sparse)
                h->ns_npairs = (int64_t)s0_pairs.size() / 2;
                h->d_nsS0pairs = ialloc(s0_pairs, (int64_t)s0_pairs.size());
            }
            h->d_nsLt = ns_dalloc(h, (int64_t)h->ns_nEp * h->ns_nEp);
            h->d_nsth = ns_dalloc(h, h->ns_ldg);
            h->d_nsFm = ns_dalloc(h, h->ldn);
            h->d_nsv = ns_dalloc(h, 11 * h->ldn + 3 * h->Mp + 2 * h->ns_nEp);
        }
    }
    // row order of the factorisations (see asm_handle::row_band)
    h->row_band = 0; h->n_rowpairs = 0; h->row_perm_h.clear(); h->main_band_cur = 0;
    h->d_rowperm = h->d_rowpos = h->d_rowpairs = h->d_cpos = nullptr;
    if (h->sp_ok && !std::getenv("ASM_HIP_NO_BAND") && h->M >= 256) {
        std::vector<int> all(h->M), pairs_pos;
        for (int64_t i = 0; i < h->M; ++i) all[i] = (int)i;
        int bw = 0;
        const std::vector<int> ord = rcm_order(all, sp_ptr, sp_col, h->ldn, &bw, &pairs_pos);
        if (2 * (int64_t)bw < h->M) {
            h->row_band = std::max(bw, 1);
            h->row_perm_h = ord;
            std::vector<int> pos(h->M), pairs(pairs_pos.size());
            for (int64_t q = 0; q < h->M; ++q) pos[ord[q]] = (int)q;
            for (size_t t = 0; t < pairs_pos.size(); ++t) pairs[t] = ord[pairs_pos[t]];
            h->n_rowpairs = (int64_t)pairs.size() / 2;
            auto up = [&](const std::vector<int>& v) {
                int* d = nullptr;
                dmalloc(&d, (int64_t)v.size());
                h->ns_bufs.push_back((void*)d);
                HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
                return d;
            };
            h->d_rowperm = up(ord); h->d_rowpos = up(pos); h->d_rowpairs = up(pairs);
            h->d_cpos = up(pos);                      // place of every row in the current row list of the interior-point factor (rewritten per iteration)
            if (!h->d_rce) { dmalloc(&h->d_rce, h->Mp); dmalloc(&h->d_rze, h->Mp); }
        }
    }
    // column order of the column form (K = Th + A' D^-1 A couples two columns when they share a row)
    h->col_band = 0; h->n_colpairs = 0; h->d_colperm = h->d_colpos = h->d_colpairs = nullptr;
    if (h->sp_ok && h->col_capable && !std::getenv("ASM_HIP_NO_BAND") && n >= 256) {
        std::vector<int> all(n), pairs_pos;
        for (int64_t j = 0; j < n; ++j) all[j] = (int)j;
        int bw = 0;
        const std::vector<int> ord = rcm_order(all, sc_ptr, sc_row, h->Mp, &bw, &pairs_pos);
        if (2 * (int64_t)bw < n) {
            h->col_band = std::max(bw, 1);
            std::vector<int> pos(n), pairs(pairs_pos.size());
            for (int64_t q = 0; q < n; ++q) pos[ord[q]] = (int)q;
            for (size_t t = 0; t < pairs_pos.size(); ++t) pairs[t] = ord[pairs_pos[t]];
            h->n_colpairs = (int64_t)pairs.size() / 2;
            auto up = [&](const std::vector<int>& v) {
                int* d = nullptr;
                dmalloc(&d, (int64_t)v.size());
                h->ns_bufs.push_back((void*)d);
                HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
                return d;
            };
            h->d_colperm = up(ord); h->d_colpos = up(pos); h->d_colpairs = up(pairs);
            if (!h->d_rce) { dmalloc(&h->d_rce, h->Mp); dmalloc(&h->d_rze, h->Mp); }
        }
    }
    h->pin_len = std::max(std::max(h->ldn, h->Mp), h->nsp);
    HIPCHK(hipHostMalloc((void**)&h->h_pin, 2 * h->pin_len * sizeof(double)));
    HIPCHK(hipHostMalloc((void**)&h->h_up, (3 * h->ldn + h->Mp + 3 * h->nsp + 16) * sizeof(double)));
    std::memset(h->h_up, 0, (3 * h->ldn + h->Mp + 3 * h->nsp + 16) * sizeof(double));
    {
        const int64_t dl = 2 * h->ldn + 2 * h->Mp + h->nsp + (h->ldn + h->Mp + h->nsp + 1) / 2 + 16;
        dmalloc(&h->d_dl, dl);
        HIPCHK(hipHostMalloc((void**)&h->h_dl, dl * sizeof(double)));
    }
    HIPCHK(hipMemsetAsync(h->d_J, 0, h->Mp * h->ldn * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->d_Ah, 0, h->Mp * h->ldn * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->d_vecN, 0, h->ldn * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->d_vecM, 0, h->Mp * sizeof(double), h->stream));
    if (!dense) {
        dmalloc(&h->d_perm, nnz); dmalloc(&h->d_ustart, h->nu + 1); dmalloc(&h->d_uoff, h->nu); dmalloc(&h->d_adjoff, h->nu);
        HIPCHK(hipMemcpy(h->d_perm, perm.data(), nnz * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_ustart, ustart.data(), (h->nu + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_uoff, uoff.data(), h->nu * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_adjoff, adjoff.data(), h->nu * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->warm[0] = ActiveSet(); h->warm[1] = ActiveSet(); h->last = ActiveSet();
    h->hint[0] = SolveHint(); h->hint[1] = SolveHint();
    h->ns_Zk = 0;
    h->hint[1].prefer_ref = true;     // restoration LPs usually have a non-unique optimum (oracle/subproblem.py)
    std::memset(&h->stats, 0, sizeof(h->stats));
    h->setup_done = true;
}

void do_upload(asm_handle* h, const double* dE, const double* df, double f, const double* E, const double* x_k) {
    if (!h->setup_done) throw std::logic_error("asm_sublp_setup has not been called");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->d_dE, dE, h->nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h2d_done(h);
    h->J_valid = false;
    h->df.assign(df, df + h->n); h->E.assign(E, E + h->m); h->x_k.assign(x_k, x_k + h->n);
    h->f = f;
    h->inputs_ready = true;
}

// the LP in caller units: min q'p + w's  s.t.  J_i p + E s (=,>=,<=) r_i (rows m.. = the extra `<=` rows of range constraints),
// lb <= p <= ub, s >= slo (slack columns only in the restoration phase; layout of create_model!, subproblem.jl:83-112)
struct LpRaw {
    vec q, r, lb, ub, w, slo;
    bool slacks = false;
};
struct LpSol {
    int status = ASM_OTHER;
    vec p, s, y, z;          // p exactly on its bound where bound-active; y per LP row; z = q - J'y (reduced costs)
    ActiveSet as;
};

// assemble J from the dE in HBM, scale, solve, unscale (oracle: solve_lp).  `slot`: which retained active set / hints to use.
void solve_raw(asm_handle* h, const LpRaw& L, int slot, LpSol& out) {
    const int64_t n = h->n, M = h->M;
    Solver sv(h);
    SLP& lp = sv.lp;
    std::memset(&h->stats, 0, sizeof(h->stats));
    lp.n = n; lp.M = M; lp.ns = L.slacks ? h->ns : 0;
    lp.rtype = h->rtype.data(); lp.srow = h->srow.data(); lp.scoef = h->scoef.data();
    h->stats.M = (int)M; h->stats.n = (int)n; h->stats.ns = (int)lp.ns;
    // Jacobian -> dense rows incl. range rows; scaled copy (column scale = min(box, matrix cap), oracle: scale_lp)
    vec c(n), rel(n);
    const double tr0 = Solver::now_ms();
    auto lap = [&](const char* what) {
        if (!h->verbose) return;
        HIPCHK(hipStreamSynchronize(h->stream));
        static thread_local double last = 0.0;
        const double t = Solver::now_ms();
        std::fprintf(stderr, "[asm] solve_raw %-10s +%.2f ms\n", what, t - (last > tr0 ? last : tr0));
        last = t;
    };
    asmb::barrier(20);
    sv.dev.assemble();
    lap("assemble");
    sv.dev.col_relmax(rel.data());
    lap("relmax");
    for (int64_t j = 0; j < n; ++j) {
        double c_mat = rel[j] > 0.0 ? 1.0 / rel[j] : 1.0;
        c[j] = pow2_round(std::min(std::max(L.ub[j], -L.lb[j]), c_mat));
    }
    vec rho(M);
    sv.dev.scale(c.data(), rho.data());
    lap("scale");
    sv.dev.tile_flags();
    lap("flags");
    lp.q.resize(n); lp.lb.resize(n); lp.ub.resize(n); lp.r.resize(M); lp.w.resize(lp.ns); lp.slo.resize(lp.ns);
    double qmax = 0.0;
    for (int64_t j = 0; j < n; ++j) { lp.q[j] = L.q[j] * c[j]; qmax = std::max(qmax, std::fabs(lp.q[j])); }
    for (int64_t k = 0; k < lp.ns; ++k) { lp.w[k] = L.w[k] * rho[h->srow[k]]; qmax = std::max(qmax, std::fabs(lp.w[k])); }
    double kap = pow2_round(qmax);
    for (int64_t j = 0; j < n; ++j) { lp.q[j] /= kap; lp.lb[j] = L.lb[j] / c[j]; lp.ub[j] = L.ub[j] / c[j]; }
    for (int64_t k = 0; k < lp.ns; ++k) { lp.w[k] /= kap; lp.slo[k] = L.slo[k] / rho[h->srow[k]]; }
    for (int64_t i = 0; i < M; ++i) lp.r[i] = L.r[i] / rho[i];
    lp.scale_q = 1.0;
    for (double v : lp.q) lp.scale_q = std::max(lp.scale_q, std::fabs(v));
    for (double v : lp.w) lp.scale_q = std::max(lp.scale_q, std::fabs(v));

    Solver::EqpOut o;
    lap("host-lp");
    out.status = sv.solve_scaled(&h->warm[slot], h->hint[slot]);
    lap("solve");
    asmb::barrier(900);
    if (out.status == ASM_OPTIMAL) {
        sv.as_download(o, out.as);
        const ActiveSet& prev = h->warm[slot];
        h->hint[slot].stable = prev.valid && prev.rowst == out.as.rowst && prev.bst == out.as.bst && prev.sst == out.as.sst;
        h->warm[slot] = out.as;
        h->last = out.as;
        // unscale - bound-active components are exactly on their bound
        out.p.resize(n); out.z.resize(n); out.y.resize(M); out.s.resize(lp.ns);
        for (int64_t j = 0; j < n; ++j) {
            double pj = o.p[j] * c[j];
            pj = std::min(std::max(pj, L.lb[j]), L.ub[j]);
            if (out.as.bst[j] < 0) pj = L.lb[j];
            else if (out.as.bst[j] > 0) pj = L.ub[j];
            out.p[j] = pj;
            out.z[j] = o.z[j] * kap / c[j];                  // z = q - J'y  ==  kap * zhat / c
        }
        for (int64_t i = 0; i < M; ++i) {
            double yi = o.y[i] * kap / rho[i];
            // multipliers with the sign their row type admits (a simplex code returns sign-feasible duals)
            if (h->rtype[i] == 1) yi = std::max(yi, 0.0);
            else if (h->rtype[i] == -1) yi = std::min(yi, 0.0);
            out.y[i] = yi;
        }
        for (int64_t k = 0; k < lp.ns; ++k) out.s[k] = o.s[k] * rho[h->srow[k]];
    } else {
        h->last = ActiveSet();
    }
    lap("extract");
    sv.dev.resolve_timing();
    lap("timing");
    check_panel_timeout(h);
    lap("ptmo");
}

void do_solve(asm_handle* h, double delta, int feasibility, double* p_out, double* lambda, double* mult_x_U, double* mult_x_L,
              double* p_slack, int32_t* status) {
    if (!h->setup_done || !h->inputs_ready) throw std::logic_error("inputs have not been uploaded");
    HIPCHK(hipSetDevice(h->device));
    auto t0 = std::chrono::steady_clock::now();
    const int64_t n = h->n, m = h->m, M = h->M;
    const bool fr = feasibility != 0;
    LpRaw L;
    L.slacks = fr;
    // trust region intersected with the variable bounds (subproblem.jl:427-434)
    L.lb.resize(n); L.ub.resize(n);
    for (int64_t j = 0; j < n; ++j) {
        L.ub[j] = std::min(delta, h->v_ub[j] - h->x_k[j]);
        L.lb[j] = std::max(-delta, h->v_lb[j] - h->x_k[j]);
    }
    // feasibility-restoration shift (subproblem.jl:287-295) and slack lower bounds (:298-381)
    vec b(h->E);
    if (fr) {
        L.slo.reserve(h->ns);
        for (int64_t i = 0; i < m; ++i) {
            double v = 0.0;
            if (h->E[i] > h->c_ub[i]) v = h->c_ub[i] - h->E[i];
            else if (h->E[i] < h->c_lb[i]) v = h->c_lb[i] - h->E[i];
            b[i] -= std::fabs(v);
            if (h->nslack[i] == 2) {
                if (v < 0) { L.slo.push_back(0.0); L.slo.push_back(v); }
                else { L.slo.push_back(-v); L.slo.push_back(0.0); }
            } else {
                L.slo.push_back(-std::fabs(v));
            }
        }
    }
    // right-hand sides (subproblem.jl:461-484)
    L.r.resize(M);
    for (int64_t i = 0; i < m; ++i) L.r[i] = h->kind[i] == -1 ? h->c_ub[i] - b[i] : h->c_lb[i] - b[i];
    for (int64_t k = 0; k < h->nadj; ++k) L.r[m + k] = h->c_ub[h->adj[k]] - b[h->adj[k]];
    // objective (subproblem.jl:250-272 | 384-405)
    L.q.assign(n, 0.0);
    L.w.assign(fr ? h->ns : 0, 1.0);
    if (!fr) L.q = h->df;

    LpSol sol;
    solve_raw(h, L, fr ? 1 : 0, sol);
    *status = sol.status;
    for (int64_t j = 0; j < n; ++j) { p_out[j] = 0.0; mult_x_U[j] = 0.0; mult_x_L[j] = 0.0; }
    for (int64_t i = 0; i < m; ++i) { lambda[i] = 0.0; p_slack[2 * i] = 0.0; p_slack[2 * i + 1] = h->nslack[i] == 2 ? 0.0 : std::nan(""); }
    if (sol.status == ASM_OPTIMAL) {
        for (int64_t j = 0; j < n; ++j) p_out[j] = sol.p[j];                                    // subproblem.jl:502
        if (fr) {
            int64_t k = 0;
            for (int64_t i = 0; i < m; ++i)
                for (int t = 0; t < h->nslack[i]; ++t, ++k) p_slack[2 * i + t] = sol.s[k];       // :503-505
        }
        for (int64_t i = 0; i < m; ++i) lambda[i] = sol.y[i];                                   // :510-512
        for (int64_t k = 0; k < h->nadj; ++k) lambda[h->adj[k]] += sol.y[m + k];                // :513-515
        for (int64_t j = 0; j < n; ++j) {                                                       // :519-529
            bool fixed = L.ub[j] <= L.lb[j];
            double mL = sol.as.bst[j] < 0 ? std::max(sol.z[j], 0.0) : 0.0;
            double mU = sol.as.bst[j] > 0 ? std::min(sol.z[j], 0.0) : 0.0;
            if (fixed) { mU = std::min(sol.z[j], 0.0); mL = std::max(sol.z[j], 0.0); }      // a fixed column reports both halves of its reduced cost
            if (p_out[j] < h->v_ub[j] - h->x_k[j]) mU = 0.0;
            if (p_out[j] > h->v_lb[j] - h->x_k[j]) mL = 0.0;
            mult_x_L[j] = mL;
            mult_x_U[j] = mU;
        }
    }
    h->stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// The LP itself, as an MOI optimizer receives it from the reference (subproblem.jl:250-484): for AsmHip.Optimizer
void do_lp_solve(asm_handle* h, const double* dE, const double* q, const double* r, const double* lb, const double* ub, int use_slacks,
                 const double* w, const double* slo, double* p, double* s, double* y, double* z, int32_t* bound_state, int32_t* status) {
    if (!h->setup_done) throw std::logic_error("asm_lp_solve: asm_sublp_setup first");
    HIPCHK(hipSetDevice(h->device));
    auto t0 = std::chrono::steady_clock::now();
    const int64_t n = h->n, M = h->M;
    for (int64_t j = 0; j < n; ++j)
        if (!(lb[j] > -INF && ub[j] < INF && lb[j] <= ub[j])) throw std::invalid_argument("asm_lp_solve: every structural column needs a finite box (the trust region)");
    HIPCHK(hipMemcpyAsync(h->d_dE, dE, h->nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->J_valid = false;
    LpRaw L;
    L.slacks = use_slacks != 0;
    L.q.assign(q, q + n); L.r.assign(r, r + M); L.lb.assign(lb, lb + n); L.ub.assign(ub, ub + n);
    if (L.slacks) { L.w.assign(w, w + h->ns); L.slo.assign(slo, slo + h->ns); }
    LpSol sol;
    solve_raw(h, L, L.slacks ? 1 : 0, sol);
    *status = sol.status;
    for (int64_t j = 0; j < n; ++j) { p[j] = 0.0; z[j] = 0.0; if (bound_state) bound_state[j] = 0; }
    for (int64_t i = 0; i < M; ++i) y[i] = 0.0;
    if (s) for (int64_t k = 0; k < h->ns; ++k) s[k] = L.slacks ? slo[k] : 0.0;
    if (sol.status == ASM_OPTIMAL) {
        for (int64_t j = 0; j < n; ++j) { p[j] = sol.p[j]; z[j] = sol.z[j]; if (bound_state) bound_state[j] = sol.as.bst[j]; }
        for (int64_t i = 0; i < M; ++i) y[i] = sol.y[i];
        if (s && L.slacks) for (int64_t k = 0; k < h->ns; ++k) s[k] = sol.s[k];
    }
    h->stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

template <class F>
int guarded(asm_handle* h, F&& fn) {
    if (!h) return ASM_ERR_ARG;
    try {
        fn();
        return ASM_OK;
    } catch (const HipError& e) {
        h->err = e.what();
        return ASM_ERR_HIP;
    } catch (const std::invalid_argument& e) {
        h->err = e.what();
        return std::string(e.what()).find("not representable") != std::string::npos ? ASM_ERR_UNSUPPORTED : ASM_ERR_ARG;
    } catch (const std::logic_error& e) {
        h->err = e.what();
        return ASM_ERR_STATE;
    } catch (const std::exception& e) {
        h->err = e.what();
        return ASM_ERR_ARG;
    }
}

}  // namespace

namespace {
template <class T>
T* ev_upload(asm_handle* h, const T* src, int64_t count) {
    T* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, std::max<int64_t>(count, 1) * sizeof(T)));
    h->ev_bufs.push_back((void*)d);
    if (count > 0) HIPCHK(hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
// kernels of one evaluation at the point in `xd`: values into Ed (m), objective into fd, optionally gradient / Jacobian values
// `ntrial` > 1 (values only): the trial points of a batched line search, xd / Ed / fd advancing by ldx / ldE / 1 per point
void ev_launch(asm_handle* h, const double* xd, double* Ed, double* fd, bool full, int ntrial = 1, int64_t ldx = 0, int64_t ldE = 0) {
    const FnStore& F = h->ev_F;
    const unsigned nt = (unsigned)ntrial;
    if (F.n_rows > 0)
        hipLaunchKernelGGL(k_fn_rows, dim3((unsigned)((F.n_rows + 255) / 256), nt), dim3(256), 0, h->stream, F, xd, Ed, h->d_dE, full ? 1 : 0, ldx, ldE);
    hipLaunchKernelGGL(k_fn_objective, dim3(nt), dim3(256), 0, h->stream, F, xd, fd, ldx);
    if (full) hipLaunchKernelGGL(k_fn_gradient, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, F, xd, h->d_ev_df);
    if (h->ev_nlp_kind == 1) {
        const int64_t nl = h->ev_nlp_rows / 4;
        hipLaunchKernelGGL(k_nlp_acopf_ohm, dim3((unsigned)((nl + 255) / 256), nt), dim3(256), 0, h->stream, (const int64_t*)h->d_ev_ipar, (const double*)h->d_ev_dpar,
                           xd, Ed, h->d_dE, F.n_rows, h->ev_fn_nnz, full ? 1 : 0, ldx, ldE);
    } else if (h->ev_nlp_kind == 2) {
        hipLaunchKernelGGL(k_nlp_dense_quadratic, dim3((unsigned)((h->ev_nlp_rows + 3) / 4), nt), dim3(256), 0, h->stream, (const double*)h->d_ev_dpar,
                           h->ev_nlp_rows, h->n, xd, Ed, h->d_dE, F.n_rows, h->ev_fn_nnz, full ? 1 : 0, ldx, ldE);
    }
}
SlpVecs ev_vecs(asm_handle* h, const double* lam, const double* mU, const double* mL, const double* jtl, const double* rown) {
    SlpVecs V;
    double* b = h->d_ev_vecs;      // layout: g_L, g_U (m each) | x_L, x_U (n each), written by asm_eval_setup
    V.g_L = b; V.g_U = b + h->m; V.x_L = b + 2 * h->m; V.x_U = b + 2 * h->m + h->n;
    V.E = h->d_ev_E; V.x = h->d_ev_x; V.df = h->d_ev_df; V.lam = lam; V.mU = mU; V.mL = mL; V.jtl = jtl; V.rown = rown;
    V.n = h->n; V.m = h->m;
    return V;
}
}  // namespace

// =========================================================================================================
// C ABI
// =========================================================================================================
extern "C" {

int asm_create(int device, asm_handle** out) {
    if (!out) return ASM_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return ASM_ERR_HIP;
    asm_handle* h = new (std::nothrow) asm_handle();
    if (!h) return ASM_ERR_ARG;
    h->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess ||
        hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, -1) != hipSuccess) {   // look-ahead chain: high priority
        delete h;
        return ASM_ERR_HIP;
    }
    std::memset(&h->kstats, 0, sizeof(h->kstats));
    std::memset(&h->stats, 0, sizeof(h->stats));
    const char* tm = std::getenv("ASM_HIP_TIMING");
    h->timing = (tm && tm[0] >= '0' && tm[0] <= '2') ? tm[0] - '0' : 1;
    const char* vb = std::getenv("ASM_HIP_VERBOSE");
    h->verbose = vb && vb[0] == '1';
    const char* sp = std::getenv("ASM_HIP_SPIN");
    h->spin_read = !(sp && sp[0] == '0');
    const char* fp = std::getenv("ASM_HIP_FUSED_PANEL");
    h->fused_panel = !(fp && fp[0] == '0');
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 16) { h->num_cus = prop.multiProcessorCount; h->panel_wgs = 2 * prop.multiProcessorCount - 32; }
        if (const char* pw = std::getenv("ASM_PANEL_WGS")) h->panel_wgs = std::max(1, std::atoi(pw));
    }
    *out = h;
    return ASM_OK;
}

int asm_destroy(asm_handle* h) {
    if (!h) return ASM_ERR_ARG;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);   // look-ahead chain may still be running after an exception
    for (auto& r : h->regions) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    free_device(h);
    for (auto e : h->la_events) (void)hipEventDestroy(e);
    if (!h->batch_slot) {
        if (h->stream2) (void)hipStreamDestroy(h->stream2);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return ASM_OK;
}

const char* asm_last_error(const asm_handle* h) { return h ? h->err.c_str() : "null handle"; }

int asm_sublp_setup(asm_handle* h, int64_t n, int64_t m, int64_t nnz, const int64_t* j_row, const int64_t* j_col,
                    const double* c_lb, const double* c_ub, const double* v_lb, const double* v_ub) {
    return guarded(h, [&] {
        if (n <= 0 || m < 0 || nnz < 0 || (nnz > 0 && (!j_row || !j_col)) || (m > 0 && (!c_lb || !c_ub)) || !v_lb || !v_ub)
            throw std::invalid_argument("asm_sublp_setup: bad dimensions or null pointer");
        do_setup(h, n, m, nnz, j_row, j_col, c_lb, c_ub, v_lb, v_ub);
    });
}

int asm_sublp_set_bounds(asm_handle* h, const double* c_lb, const double* c_ub, const double* v_lb, const double* v_ub) {
    return guarded(h, [&] {
        if (!h->setup_done) throw std::logic_error("asm_sublp_set_bounds: asm_sublp_setup first");
        if ((h->m > 0 && (!c_lb || !c_ub)) || !v_lb || !v_ub) throw std::invalid_argument("asm_sublp_set_bounds: null pointer");
        for (int64_t i = 0; i < h->m; ++i)
            if (row_kind(c_lb[i], c_ub[i]) != h->kind[i])
                throw std::invalid_argument("asm_sublp_set_bounds: the kind of a row changes - the LP skeleton is not representable (call asm_sublp_setup)");
        HIPCHK(hipSetDevice(h->device));
        h->c_lb.assign(c_lb, c_lb + h->m); h->c_ub.assign(c_ub, c_ub + h->m);
        h->v_lb.assign(v_lb, v_lb + h->n); h->v_ub.assign(v_ub, v_ub + h->n);
        if (h->ev_ready) {              // the reductions' copy of the bounds
            HIPCHK(hipMemcpy(h->d_ev_vecs, h->c_lb.data(), h->m * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(h->d_ev_vecs + h->m, h->c_ub.data(), h->m * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(h->d_ev_vecs + 2 * h->m, h->v_lb.data(), h->n * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(h->d_ev_vecs + 2 * h->m + h->n, h->v_ub.data(), h->n * sizeof(double), hipMemcpyHostToDevice));
        }
        // the retained working sets and the basis Z belong to the old instance; the basis COLUMNS of the null-space form are kept - the
        // pattern is the same, and a set that no longer spans null(A_EF) is detected and re-selected by the next LP (Solver::ns_setup)
        std::vector<int> keepJ = h->hint[0].ns_J;
        h->warm[0] = ActiveSet(); h->warm[1] = ActiveSet(); h->last = ActiveSet();
        h->hint[0] = SolveHint(); h->hint[1] = SolveHint();
        h->hint[0].ns_J.swap(keepJ);
        h->ns_Zk = 0;
        h->hint[1].prefer_ref = true;
        h->inputs_ready = false;
    });
}

int asm_sublp_upload(asm_handle* h, const double* dE, const double* df, double f, const double* E, const double* x_k) {
    return guarded(h, [&] {
        if ((h->nnz > 0 && !dE) || !df || (h->m > 0 && !E) || !x_k) throw std::invalid_argument("asm_sublp_upload: null pointer");
        do_upload(h, dE, df, f, E, x_k);
    });
}

int asm_sublp_solve_resident(asm_handle* h, double delta, int feasibility, double* p, double* lambda, double* mult_x_U,
                             double* mult_x_L, double* p_slack, int32_t* status) {
    return guarded(h, [&] {
        if (!p || (h->m > 0 && (!lambda || !p_slack)) || !mult_x_U || !mult_x_L || !status)
            throw std::invalid_argument("asm_sublp_solve: null output pointer");
        if (!(delta >= 0.0)) throw std::invalid_argument("asm_sublp_solve: delta must be >= 0");
        do_solve(h, delta, feasibility, p, lambda, mult_x_U, mult_x_L, p_slack, status);
    });
}

int asm_sublp_solve(asm_handle* h, const double* dE, const double* df, double f, const double* E, const double* x_k, double delta,
                    int feasibility, double* p, double* lambda, double* mult_x_U, double* mult_x_L, double* p_slack,
                    int32_t* status) {
    int rc = asm_sublp_upload(h, dE, df, f, E, x_k);
    if (rc != ASM_OK) return rc;
    return asm_sublp_solve_resident(h, delta, feasibility, p, lambda, mult_x_U, mult_x_L, p_slack, status);
}

int asm_lp_solve(asm_handle* h, const double* dE, const double* q, const double* r, const double* lb, const double* ub, int use_slacks,
                 const double* w, const double* slo, double* p, double* s, double* y, double* z, int32_t* bound_state, int32_t* status) {
    return guarded(h, [&] {
        if ((h->nnz > 0 && !dE) || !q || (h->M > 0 && (!r || !y)) || !lb || !ub || !p || !z || !status || (use_slacks && (!w || !slo)))
            throw std::invalid_argument("asm_lp_solve: null pointer");
        do_lp_solve(h, dE, q, r, lb, ub, use_slacks, w, slo, p, s, y, z, bound_state, status);
    });
}

int asm_sublp_active_set(const asm_handle* h, int32_t* row_state, int32_t* bound_state, int32_t* slack_state, int64_t* n_rows,
                         int64_t* n_slack) {
    if (!h) return ASM_ERR_ARG;
    if (!h->last.valid) return ASM_ERR_STATE;
    if (n_rows) *n_rows = (int64_t)h->last.rowst.size();
    if (n_slack) *n_slack = (int64_t)h->last.sst.size();
    if (row_state) for (size_t i = 0; i < h->last.rowst.size(); ++i) row_state[i] = h->last.rowst[i];
    if (bound_state) for (size_t i = 0; i < h->last.bst.size(); ++i) bound_state[i] = h->last.bst[i];
    if (slack_state) for (size_t i = 0; i < h->last.sst.size(); ++i) slack_state[i] = h->last.sst[i];
    return ASM_OK;
}

int asm_sublp_reset_warm(asm_handle* h) {
    if (!h) return ASM_ERR_ARG;
    h->warm[0] = ActiveSet();
    h->warm[1] = ActiveSet();
    h->hint[0] = SolveHint();
    h->hint[1] = SolveHint();
    h->ns_Zk = 0;
    h->hint[1].prefer_ref = true;
    return ASM_OK;
}

int asm_sublp_ns_basis(const asm_handle* h, int32_t* J, int64_t* k) {
    if (!h || !k) return ASM_ERR_ARG;
    const std::vector<int>& v = h->hint[0].ns_J;
    *k = (int64_t)v.size();
    if (J) for (size_t a = 0; a < v.size(); ++a) J[a] = v[a];
    return ASM_OK;
}

int asm_sublp_row_order(const asm_handle* h, int32_t* perm, int64_t* band, int32_t* e_rows, int64_t* n_e, int64_t* e_band) {
    if (!h || !band || !n_e || !e_band) return ASM_ERR_ARG;
    *band = h->row_band;
    if (perm && h->row_band > 0) for (int64_t q = 0; q < h->M; ++q) perm[q] = h->row_perm_h[q];
    *n_e = h->ns_cap ? h->ns_nE : 0;
    *e_band = h->ns_cap ? h->ns_f0.band : 0;
    if (e_rows && h->ns_cap) for (int q = 0; q < h->ns_nE; ++q) e_rows[q] = h->ns_eidx_h[q];
    return ASM_OK;
}

int asm_sublp_last_stats(const asm_handle* h, asm_solve_stats* out) {
    if (!h || !out) return ASM_ERR_ARG;
    *out = h->stats;
    return ASM_OK;
}

int asm_kernel_stats_get(asm_handle* h, asm_kernel_stats* out) {
    if (!out) return ASM_ERR_ARG;
    return guarded(h, [&] {
        Dev d(h);
        d.resolve_timing();
        *out = h->kstats;
    });
}

int asm_kernel_timing(asm_handle* h, int level) {
    return guarded(h, [&] {
        if (level < 0 || level > 2) throw std::invalid_argument("asm_kernel_timing: level 0, 1 or 2");
        Dev d(h);
        d.resolve_timing();
        h->timing = h->batch_slot ? 0 : level;
    });
}

int asm_kernel_stats_reset(asm_handle* h) {
    return guarded(h, [&] {
        Dev d(h);
        d.resolve_timing();
        std::memset(&h->kstats, 0, sizeof(h->kstats));
    });
}

// --------------------------------------------------------------------------------- norms on the resident Jacobian
int asm_jac_row_norms(asm_handle* h, double* out_m) {
    return guarded(h, [&] {
        if (!h->setup_done || !h->inputs_ready || !out_m) throw std::logic_error("asm_jac_row_norms: no assembled Jacobian");
        HIPCHK(hipSetDevice(h->device));
        Dev d(h);
        d.assemble();
        if (h->m == 0) return;
        hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((h->m + 3) / 4)), dim3(256), 0, h->stream, h->d_J, h->ldn, h->d_vecM, h->m, h->ldn);
        d.d2h(out_m, h->d_vecM, h->m);
    });
}

int asm_kt_residuals(asm_handle* h, const double* df, const double* lambda, const double* mult_x_U, const double* mult_x_L, double* out) {
    return guarded(h, [&] {
        if (!h->setup_done || !h->inputs_ready || !df || !mult_x_U || !mult_x_L || !out || (h->m > 0 && !lambda))
            throw std::logic_error("asm_kt_residuals: no assembled Jacobian or null pointer");
        HIPCHK(hipSetDevice(h->device));
        Dev d(h);
        d.assemble();
        const int64_t n = h->n, m = h->m;
        vec lam(h->M, 0.0), jtl(n), rn(std::max<int64_t>(m, 1));
        for (int64_t i = 0; i < m; ++i) lam[i] = lambda[i];
        d.gemv_t(h->d_J, lam.data(), jtl.data());
        if (m > 0) {
            hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, h->stream, h->d_J, h->ldn, h->d_vecM, m, h->ldn);
            d.d2h(rn.data(), h->d_vecM, m);
        }
        // common.jl:38-43
        double res = 0.0, ndf = 0.0;
        for (int64_t j = 0; j < n; ++j) {
            double v = df[j] - jtl[j] - mult_x_U[j] - mult_x_L[j];
            res += v * v;
            ndf += df[j] * df[j];
        }
        double scalar = std::max(1.0, std::sqrt(ndf));
        for (int64_t i = 0; i < m; ++i) scalar = std::max(scalar, std::fabs(lambda[i]) * rn[i]);
        *out = std::sqrt(res) / scalar;
    });
}

// --------------------------------------------------------------------------------- device-side evaluators (rows a2 / f3)
int asm_eval_setup(asm_handle* h, int64_t n_rows, const int64_t* aff_ptr, const int64_t* aff_var, const double* aff_coef, const int64_t* quad_ptr,
                   const int64_t* q_v1, const int64_t* q_v2, const double* q_coef, const double* constant, const int64_t* jac_off,
                   const int64_t* g_ptr, const int64_t* g_kind, const double* g_coef, const int64_t* g_other, double objective_scale, int nlp_kind,
                   int64_t nlp_rows, int64_t nlp_nnz, const int64_t* nlp_ipar, int64_t n_ipar, const double* nlp_dpar, int64_t n_dpar) {
    return guarded(h, [&] {
        if (!h->setup_done) throw std::logic_error("asm_eval_setup: asm_sublp_setup first (it fixes n, m and the j_str order of dE)");
        if (n_rows < 0 || !aff_ptr || !quad_ptr || !constant || !jac_off || !g_ptr || nlp_kind < 0 || nlp_kind > 2)
            throw std::invalid_argument("asm_eval_setup: bad argument");
        const int64_t fn_nnz = jac_off[n_rows];
        if (n_rows + nlp_rows != h->m || fn_nnz + nlp_nnz != h->nnz)
            throw std::invalid_argument("asm_eval_setup: row / Jacobian-entry counts do not match asm_sublp_setup");
        if (nlp_kind == 1 && (nlp_rows % 4 != 0 || nlp_nnz != 5 * nlp_rows || n_ipar != 7 + 2 * (nlp_rows / 4) || n_dpar != 8 * (nlp_rows / 4)))
            throw std::invalid_argument("asm_eval_setup: ACOPF block parameter sizes");
        if (nlp_kind == 2 && (nlp_nnz != nlp_rows * h->n || n_dpar != 2 * nlp_rows * h->n)) throw std::invalid_argument("asm_eval_setup: dense block parameter sizes");
        HIPCHK(hipSetDevice(h->device));
        for (void* q : h->ev_bufs) (void)hipFree(q);
        h->ev_bufs.clear();
        FnStore& F = h->ev_F;
        F.n_rows = n_rows; F.n = h->n; F.objective_scale = objective_scale;
        const int64_t na = aff_ptr[n_rows + 1], nq = quad_ptr[n_rows + 1], ng = g_ptr[h->n];
        F.aff_ptr = ev_upload(h, aff_ptr, n_rows + 2); F.aff_var = ev_upload(h, aff_var, na); F.aff_coef = ev_upload(h, aff_coef, na);
        F.quad_ptr = ev_upload(h, quad_ptr, n_rows + 2); F.q_v1 = ev_upload(h, q_v1, nq); F.q_v2 = ev_upload(h, q_v2, nq); F.q_coef = ev_upload(h, q_coef, nq);
        F.constant = ev_upload(h, constant, n_rows + 1); F.jac_off = ev_upload(h, jac_off, n_rows + 1);
        F.g_ptr = ev_upload(h, g_ptr, h->n + 1); F.g_kind = ev_upload(h, g_kind, ng); F.g_coef = ev_upload(h, g_coef, ng); F.g_other = ev_upload(h, g_other, ng);
        h->ev_nlp_kind = nlp_kind; h->ev_nlp_rows = nlp_rows; h->ev_nlp_nnz = nlp_nnz; h->ev_fn_nnz = fn_nnz;
        h->d_ev_ipar = ev_upload(h, nlp_ipar, n_ipar);
        h->d_ev_dpar = ev_upload(h, nlp_dpar, n_dpar);
        h->d_ev_x = ev_upload<double>(h, nullptr, 0); (void)h->d_ev_x;
        const int64_t n = h->n, m = std::max<int64_t>(h->m, 1);
        auto dalloc = [&](int64_t cnt) { double* d = nullptr; HIPCHK(hipMalloc((void**)&d, std::max<int64_t>(cnt, 1) * sizeof(double))); h->ev_bufs.push_back((void*)d);
                                          HIPCHK(hipMemset(d, 0, std::max<int64_t>(cnt, 1) * sizeof(double))); return d; };
        h->d_ev_x = dalloc(n); h->d_ev_xt = dalloc(8 * round_up(n, 32)); h->d_ev_df = dalloc(n); h->d_ev_E = dalloc(m); h->d_ev_Et = dalloc(8 * round_up(std::max<int64_t>(m, 1), 32));
        h->d_ev_f = dalloc(16);      // xt / Et / f[1..8]: eight trial points of the batched line search
        // bounds for the reductions + staging area: [g_L, g_U, x_L, x_U | lam, mU, mL, nu, ps(2m), p, jtl(ldn), rown(Mp), out(8)]
        h->d_ev_vecs = dalloc(2 * m + 2 * n + 2 * m + 2 * n + 2 * m + n + h->ldn + h->Mp + 16);
        HIPCHK(hipMemcpy(h->d_ev_vecs, h->c_lb.data(), h->m * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_ev_vecs + h->m, h->c_ub.data(), h->m * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_ev_vecs + 2 * h->m, h->v_lb.data(), n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_ev_vecs + 2 * h->m + n, h->v_ub.data(), n * sizeof(double), hipMemcpyHostToDevice));
        if (!h->h_ev) HIPCHK(hipHostMalloc((void**)&h->h_ev, (4 * (n + m) + 64) * sizeof(double)));
        h->ev_ready = true;
    });
}

// eval_functions! (slp.jl:186-191) on the device + what asm_sublp_upload does with the results: dE stays in HBM
int asm_eval_functions(asm_handle* h, const double* x, double* f, double* df, double* E) {
    return guarded(h, [&] {
        if (!h->ev_ready || !x || !f || !df || (h->m > 0 && !E)) throw std::logic_error("asm_eval_functions: asm_eval_setup first / null pointer");
        HIPCHK(hipSetDevice(h->device));
        const int64_t n = h->n, m = h->m;
        std::memcpy(h->h_ev, x, n * sizeof(double));
        HIPCHK(hipMemcpyAsync(h->d_ev_x, h->h_ev, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        ev_launch(h, h->d_ev_x, h->d_ev_E, h->d_ev_f, true);
        double* st = h->h_ev + n;
        HIPCHK(hipMemcpyAsync(st, h->d_ev_df, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (m) HIPCHK(hipMemcpyAsync(st + n, h->d_ev_E, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(st + n + m, h->d_ev_f, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        std::memcpy(df, st, n * sizeof(double));
        if (m) std::memcpy(E, st + n, m * sizeof(double));
        *f = st[n + m];
        h->df.assign(df, df + n); h->E.assign(E, E + m); h->x_k.assign(x, x + n);
        h->f = *f;
        h->inputs_ready = true;
        h->J_valid = false;
    });
}

// eval_f + eval_g at a trial point (compute_alpha, slp_line_search.jl:222-244; step_quality, slp_trust_region.jl:213-251)
int asm_eval_constraints(asm_handle* h, const double* x, double* f, double* E) {
    return guarded(h, [&] {
        if (!h->ev_ready || !x || !f || (h->m > 0 && !E)) throw std::logic_error("asm_eval_constraints: asm_eval_setup first / null pointer");
        HIPCHK(hipSetDevice(h->device));
        const int64_t n = h->n, m = h->m;
        std::memcpy(h->h_ev, x, n * sizeof(double));
        HIPCHK(hipMemcpyAsync(h->d_ev_xt, h->h_ev, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        ev_launch(h, h->d_ev_xt, h->d_ev_Et, h->d_ev_f + 1, false);
        double* st = h->h_ev + n;
        if (m) HIPCHK(hipMemcpyAsync(st, h->d_ev_Et, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(st + m, h->d_ev_f + 1, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (m) std::memcpy(E, st, m * sizeof(double));
        *f = st[m];
    });
}

int asm_eval_jacobian_values(asm_handle* h, double* dE_out) {
    return guarded(h, [&] {
        if (!h->setup_done || !dE_out) throw std::logic_error("asm_eval_jacobian_values: no setup / null pointer");
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipMemcpy(dE_out, h->d_dE, h->nnz * sizeof(double), hipMemcpyDeviceToHost));
    });
}

// --------------------------------------------------------------------------------- per-iteration reductions on the device (row f1)
// out[4] = { norm_violations(Inf), norm_violations(1), KT_residuals, norm_complementarity(Inf) }  (common.jl:35-98) from the
// evaluation results of the last asm_eval_functions and the Jacobian assembled from its dE (assembled once, shared with the LP)
int asm_slp_norms(asm_handle* h, const double* lambda, const double* mult_x_U, const double* mult_x_L, double* out4) {
    return guarded(h, [&] {
        if (!h->ev_ready || !h->inputs_ready || !mult_x_U || !mult_x_L || !out4 || (h->m > 0 && !lambda))
            throw std::logic_error("asm_slp_norms: asm_eval_functions first / null pointer");
        HIPCHK(hipSetDevice(h->device));
        const int64_t n = h->n, m = h->m;
        Dev d(h);
        d.assemble();
        double* v = h->d_ev_vecs + 2 * m + 2 * n;       // lam | mU | mL
        double *lam = v, *mU = v + m, *mL = mU + n, *jtl = mL + n + m + 2 * m + n /* after nu (m), ps (2m), p (n) */, *rown = jtl + h->ldn, *outd = rown + h->Mp;
        double* st = h->h_ev;
        if (m) std::memcpy(st, lambda, m * sizeof(double));
        std::memcpy(st + m, mult_x_U, n * sizeof(double));
        std::memcpy(st + m + n, mult_x_L, n * sizeof(double));
        HIPCHK(hipMemcpyAsync(lam, st, (m + 2 * n) * sizeof(double), hipMemcpyHostToDevice, h->stream));
        // J' lambda over the first m rows: lambda padded with zeros on the extra range rows
        HIPCHK(hipMemsetAsync(h->d_vecM, 0, h->Mp * sizeof(double), h->stream));
        if (m) HIPCHK(hipMemcpyAsync(h->d_vecM, lam, m * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        d.launch_gemv_t(h->d_J, h->d_vecM, jtl);
        if (m) {
            if (const double* vJ = d.sparse_vals(h->d_J))      // sparse pattern: from the CSR copy (13 k entries instead of a 75 MB dense sweep at case300 size)
                hipLaunchKernelGGL(k_sp_row_norms, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, h->stream, (const int*)h->d_sp_ptr, (const int*)h->d_sp_col, vJ, rown, m);
            else
                hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, h->stream, h->d_J, h->ldn, rown, m, h->ldn);
        }
        hipLaunchKernelGGL(k_slp_norms, dim3(1), dim3(1024), 0, h->stream, ev_vecs(h, lam, mU, mL, jtl, rown), outd);
        HIPCHK(hipMemcpyAsync(st, outd, RN_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int k = 0; k < RN_COUNT; ++k) out4[k] = st[k];
        d.resolve_timing();
    });
}

// compute_phi(x, alpha, p) (slp.jl:79-115; mode 0) and compute_derivative (slp.jl:122-147; mode 1) with the trial evaluation on the
// device.  p_slack: 2 entries per row as asm_sublp_solve returns them.
int asm_slp_merit(asm_handle* h, int mode, double alpha, const double* p, const double* nu, const double* p_slack, int feasibility, double prim_infeas,
                  double* out) {
    return guarded(h, [&] {
        if (!h->ev_ready || !h->inputs_ready || !p || !out || (h->m > 0 && (!nu || !p_slack)) || mode < 0 || mode > 1)
            throw std::logic_error("asm_slp_merit: asm_eval_functions first / bad argument");
        HIPCHK(hipSetDevice(h->device));
        const int64_t n = h->n, m = h->m;
        double* v = h->d_ev_vecs + 2 * m + 2 * n + m + 2 * n;    // nu | ps | p
        double *nud = v, *psd = v + m, *pd = psd + 2 * m, *outd = pd + n + h->ldn + h->Mp;
        double* st = h->h_ev;
        if (m) { std::memcpy(st, nu, m * sizeof(double)); std::memcpy(st + m, p_slack, 2 * m * sizeof(double)); }
        std::memcpy(st + 3 * m, p, n * sizeof(double));
        HIPCHK(hipMemcpyAsync(nud, st, (3 * m + n) * sizeof(double), hipMemcpyHostToDevice, h->stream));
        const double* Et = h->d_ev_E;
        const double* ft = h->d_ev_f;
        if (mode == 0 && alpha != 0.0) {
            hipLaunchKernelGGL(k_axpy_out, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->d_ev_x, alpha, (const double*)pd, h->d_ev_xt, n);
            ev_launch(h, h->d_ev_xt, h->d_ev_Et, h->d_ev_f + 1, false);
            Et = h->d_ev_Et;
            ft = h->d_ev_f + 1;
        }
        hipLaunchKernelGGL(k_slp_merit, dim3(1), dim3(1024), 0, h->stream, ev_vecs(h, nullptr, nullptr, nullptr, nullptr, nullptr), Et, (const double*)nud,
                           (const double*)psd, (const double*)pd, alpha, feasibility, prim_infeas, ft, mode, outd, TrialAlphas(), (int64_t)0);
        HIPCHK(hipMemcpyAsync(st, outd, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        *out = st[0];
    });
}

// compute_alpha (slp_line_search.jl:222-244) with the trial evaluations on the device: alpha = 1, tau, tau^2, ... until
//   phi(alpha) <= phi0 + eta alpha D      (accepted: *ok = 1)      or      alpha < min_alpha with the test still failing (*ok = 0).
// The trials are pure function evaluations, so they are launched eight at a time without waiting for the verdict of the earlier ones
// (one read-back per eight); nu, the slacks and p are uploaded once.  Same alpha, same merit values as calling asm_slp_merit per trial.
int asm_slp_line_search(asm_handle* h, const double* p, const double* nu, const double* p_slack, int feasibility, double prim_infeas, double phi0,
                        double D, double eta, double tau, double min_alpha, double* alpha_out, double* phi_out, int* trials_out, int* ok_out) {
    return guarded(h, [&] {
        if (!h->ev_ready || !h->inputs_ready || !p || !alpha_out || !ok_out || (h->m > 0 && (!nu || !p_slack)) || !(tau > 0.0 && tau < 1.0))
            throw std::logic_error("asm_slp_line_search: asm_eval_functions first / bad argument");
        HIPCHK(hipSetDevice(h->device));
        const int64_t n = h->n, m = h->m;
        double* v = h->d_ev_vecs + 2 * m + 2 * n + m + 2 * n;    // nu | ps | p
        double *nud = v, *psd = v + m, *pd = psd + 2 * m, *outd = pd + n + h->ldn + h->Mp;
        double* st = h->h_ev;
        if (m) { std::memcpy(st, nu, m * sizeof(double)); std::memcpy(st + m, p_slack, 2 * m * sizeof(double)); }
        std::memcpy(st + 3 * m, p, n * sizeof(double));
        HIPCHK(hipMemcpyAsync(nud, st, (3 * m + n) * sizeof(double), hipMemcpyHostToDevice, h->stream));
        const SlpVecs V = ev_vecs(h, nullptr, nullptr, nullptr, nullptr, nullptr);
        constexpr int CH = 8;
        double alpha = 1.0, a[CH];
        int trials = 0;
        *ok_out = -1;
        while (*ok_out < 0) {
            asmb::barrier(920);
            // eight trial points per set of launches (trial index in the grid): x + alpha_t p, the function values there, the merit values
            TrialAlphas al;
            for (int t = 0; t < CH; ++t) { a[t] = al.a[t] = alpha; alpha *= tau; }
            const int64_t ldx = round_up(n, 32), ldE = round_up(std::max<int64_t>(m, 1), 32);
            hipLaunchKernelGGL(k_axpy_trials, dim3((unsigned)((n + 255) / 256), CH), dim3(256), 0, h->stream, (const double*)h->d_ev_x, al, (const double*)pd, h->d_ev_xt, n, ldx);
            ev_launch(h, h->d_ev_xt, h->d_ev_Et, h->d_ev_f + 1, false, CH, ldx, ldE);
            hipLaunchKernelGGL(k_slp_merit, dim3(CH), dim3(1024), 0, h->stream, V, (const double*)h->d_ev_Et, (const double*)nud, (const double*)psd, (const double*)pd, 0.0,
                               feasibility, prim_infeas, (const double*)(h->d_ev_f + 1), 0, outd, al, ldE);
            HIPCHK(hipMemcpyAsync(st, outd, CH * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            for (int t = 0; t < CH && *ok_out < 0; ++t) {
                trials += 1;
                if (!(st[t] > phi0 + eta * a[t] * D)) { *ok_out = 1; *alpha_out = a[t]; if (phi_out) *phi_out = st[t]; }
                else if (a[t] < min_alpha) { *ok_out = 0; *alpha_out = a[t]; if (phi_out) *phi_out = st[t]; }
            }
        }
        if (trials_out) *trials_out = trials;
    });
}

// The native SLP driver's step: compute_phi(x, 0, p), compute_derivative and compute_alpha in one upload and - when one of the first eight
// trial steps is accepted, i.e. nearly always - one read-back (asm_slp_merit twice + asm_slp_line_search: three uploads of nu | slacks | p and
// three read-backs).  The trial points do not depend on phi0 and D, only the acceptance test does: same launches per quantity, same values.
static void slp_merit_and_search(asm_handle* h, const double* p, const double* nu, const double* p_slack, int feasibility, double prim_infeas, double eta, double tau,
                                 double min_alpha, double* phi0_out, double* D_out, double* alpha_out, double* phi_out, int* trials_out, int* ok_out) {
    const int64_t n = h->n, m = h->m;
    double* v = h->d_ev_vecs + 2 * m + 2 * n + m + 2 * n;    // nu | ps | p
    double *nud = v, *psd = v + m, *pd = psd + 2 * m, *outd = pd + n + h->ldn + h->Mp;
    double* st = h->h_ev;
    if (m) { std::memcpy(st, nu, m * sizeof(double)); std::memcpy(st + m, p_slack, 2 * m * sizeof(double)); }
    std::memcpy(st + 3 * m, p, n * sizeof(double));
    HIPCHK(hipMemcpyAsync(nud, st, (3 * m + n) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const SlpVecs V = ev_vecs(h, nullptr, nullptr, nullptr, nullptr, nullptr);
    constexpr int CH = 8;
    for (int mode = 0; mode < 2; ++mode)
        hipLaunchKernelGGL(k_slp_merit, dim3(1), dim3(1024), 0, h->stream, V, (const double*)h->d_ev_E, (const double*)nud, (const double*)psd, (const double*)pd, 0.0, feasibility,
                           prim_infeas, (const double*)h->d_ev_f, mode, outd + CH + mode, TrialAlphas(), (int64_t)0);
    double alpha = 1.0, a[CH], phi0 = 0.0, D = 0.0;
    int trials = 0;
    *ok_out = -1;
    for (int round = 0; *ok_out < 0; ++round) {
        asmb::barrier(920);
        TrialAlphas al;
        for (int t = 0; t < CH; ++t) { a[t] = al.a[t] = alpha; alpha *= tau; }
        const int64_t ldx = round_up(n, 32), ldE = round_up(std::max<int64_t>(m, 1), 32);
        hipLaunchKernelGGL(k_axpy_trials, dim3((unsigned)((n + 255) / 256), CH), dim3(256), 0, h->stream, (const double*)h->d_ev_x, al, (const double*)pd, h->d_ev_xt, n, ldx);
        ev_launch(h, h->d_ev_xt, h->d_ev_Et, h->d_ev_f + 1, false, CH, ldx, ldE);
        hipLaunchKernelGGL(k_slp_merit, dim3(CH), dim3(1024), 0, h->stream, V, (const double*)h->d_ev_Et, (const double*)nud, (const double*)psd, (const double*)pd, 0.0,
                           feasibility, prim_infeas, (const double*)(h->d_ev_f + 1), 0, outd, al, ldE);
        HIPCHK(hipMemcpyAsync(st, outd, (CH + (round == 0 ? 2 : 0)) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (round == 0) { phi0 = st[CH]; D = st[CH + 1]; }
        for (int t = 0; t < CH && *ok_out < 0; ++t) {
            trials += 1;
            if (!(st[t] > phi0 + eta * a[t] * D)) { *ok_out = 1; *alpha_out = a[t]; *phi_out = st[t]; }
            else if (a[t] < min_alpha) { *ok_out = 0; *alpha_out = a[t]; *phi_out = st[t]; }
        }
    }
    *phi0_out = phi0; *D_out = D; *trials_out = trials;
}

// --------------------------------------------------------------------------------- kernel test hooks
static void test_alloc(asm_handle* h, int64_t M, int64_t K) {
    // minimal "problem" so that the generic buffers exist: dense pattern M x K
    std::vector<int64_t> jr(1, 1), jc(1, 1);
    vec lo(M, 0.0), hi(M, 0.0), vl(K, -1.0), vu(K, 1.0);
    do_setup(h, K, M, 1, jr.data(), jc.data(), lo.data(), hi.data(), vl.data(), vu.data());
    h->sp_ok = false;     // the hooks write arbitrary matrices into the buffers
}

int asm_test_syrk(asm_handle* h, const double* A, int64_t M, int64_t K, const int32_t* idx, int64_t Ms, const double* theta,
                  const double* diag, double* S_out, int tile) {
    return guarded(h, [&] {
        test_alloc(h, M, K);
        Dev d(h);
        for (int64_t i = 0; i < M; ++i)
            HIPCHK(hipMemcpy(h->d_Ah + i * h->ldn, A + i * K, K * sizeof(double), hipMemcpyHostToDevice));
        d.h2d(h->d_theta, theta, K, h->ldn);
        if (diag) d.h2d(h->d_diag, diag, Ms, Ms);
        if (idx) HIPCHK(hipMemcpy(h->d_idx, idx, Ms * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(h->d_S, 0, h->Mp * h->Mp * sizeof(double)));
        d.launch_syrk(tile > 0 ? tile : Dev::pick_tile(Ms), h->d_Ah, h->ldn, idx ? h->d_idx : nullptr, 0, (int)Ms, (int)h->ldn, h->d_theta,
                      diag ? h->d_diag : nullptr, h->d_S, h->Mp, 0, 0);
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < Ms; ++i)
            HIPCHK(hipMemcpy(S_out + i * Ms, h->d_S + i * h->Mp, Ms * sizeof(double), hipMemcpyDeviceToHost));
    });
}

// S[a,b] -= sum_k P[a,k] P[b,k] for a >= b (and b < MsB when MsB >= 0): the Cholesky update as the factorisation launches it
// (k_syrk_upd for tile 4, the generic kernel otherwise), with an offset origin inside a larger matrix (srow0) like a trailing update
int asm_test_syrk_update(asm_handle* h, const double* Pm, int64_t Ms, int64_t K, int64_t MsB, int64_t srow0, double* S_inout, int tile) {
    return guarded(h, [&] {
        if (K % 64 != 0) throw HipError("asm_test_syrk_update: K must be a multiple of 64 (panel widths)");
        const int64_t N = srow0 + Ms;
        test_alloc(h, N, std::max<int64_t>(K, 16));
        Dev d(h);
        HIPCHK(hipMemset(h->d_Ah, 0, h->Mp * h->ldn * sizeof(double)));
        for (int64_t i = 0; i < Ms; ++i)
            HIPCHK(hipMemcpy(h->d_Ah + (srow0 + i) * h->ldn, Pm + i * K, K * sizeof(double), hipMemcpyHostToDevice));
        for (int64_t i = 0; i < Ms; ++i)
            HIPCHK(hipMemcpy(h->d_S + (srow0 + i) * h->Mp + srow0, S_inout + i * Ms, Ms * sizeof(double), hipMemcpyHostToDevice));
        d.launch_syrk(tile > 0 ? tile : Dev::pick_tile(Ms), h->d_Ah, h->ldn, nullptr, srow0, (int)Ms, (int)K, nullptr, nullptr, h->d_S, h->Mp, srow0, 1, (int)MsB);
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < Ms; ++i)
            HIPCHK(hipMemcpy(S_inout + i * Ms, h->d_S + (srow0 + i) * h->Mp + srow0, Ms * sizeof(double), hipMemcpyDeviceToHost));
        d.resolve_timing();
    });
}

static void test_load_S(asm_handle* h, const double* S, int64_t N) {
    test_alloc(h, N, 16);
    h->main_band_cur = 0;
    for (int64_t i = 0; i < N; ++i)
        HIPCHK(hipMemcpy(h->d_S + i * h->Mp, S + i * N, N * sizeof(double), hipMemcpyHostToDevice));
}

int asm_test_cholesky(asm_handle* h, const double* S, int64_t N, double* L_out) {
    return guarded(h, [&] {
        test_load_S(h, S, N);
        Dev d(h);
        d.diag_prepare((int)N, 0, 0.0, 0.0);
        d.chol((int)N);
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < N; ++i) {
            HIPCHK(hipMemcpy(L_out + i * N, h->d_S + i * h->Mp, N * sizeof(double), hipMemcpyDeviceToHost));
            for (int64_t j = i + 1; j < N; ++j) L_out[i * N + j] = 0.0;
        }
        d.resolve_timing();
        check_panel_timeout(h);
    });
}

// The bounded wait of the panel kernel with a producer that never publishes: the probe's workgroups must give up (the first after the
// full bound, the others at their next look at the timeout word), the host must report ASM_ERR_HIP once, and the handle must stay usable.
// the matrices the following kernel hooks load are banded with this half-bandwidth (0 = dense again): their factorisation and
// substitutions then stop at the band, as for the S0 of the null-space form
int asm_test_set_band(asm_handle* h, int band) {
    return guarded(h, [&] {
        if (band < 0) throw std::invalid_argument("asm_test_set_band: bad argument");
        h->main_band = band;
    });
}

// every active-set attempt of the following LPs fails (on != 0): the LP solve ends on its last resort, the converged interior iterate
int asm_test_no_polish(asm_handle* h, int on) {
    return guarded(h, [&] { h->test_no_polish = on != 0; });
}

int asm_test_panel_timeout(asm_handle* h, int workgroups) {
    return guarded(h, [&] {
        if (workgroups < 1 || workgroups > 64) throw std::invalid_argument("asm_test_panel_timeout: 1..64 workgroups");
        test_alloc(h, 64, 16);
        h->panel_epoch += 1;
        if (h->panel_epoch == 0) h->panel_epoch = 1;
        hipLaunchKernelGGL(k_pnl_wait_probe, dim3((unsigned)workgroups), dim3(256), 0, h->stream, h->d_pflags, h->panel_epoch, h->d_ptmo);
        HIPCHK(hipStreamSynchronize(h->stream));
        check_panel_timeout(h);
    });
}

int asm_test_chol_solve(asm_handle* h, const double* S, int64_t N, const double* b, double* x) {
    return guarded(h, [&] {
        test_load_S(h, S, N);
        Dev d(h);
        d.diag_prepare((int)N, 0, 0.0, 0.0);
        d.chol((int)N);
        d.chol_solve(b, x, (int)N);
        d.resolve_timing();
    });
}

int asm_test_gemm_nt(asm_handle* h, const double* A, const double* B, const double* C0, int64_t Ma, int64_t Mb, int64_t K, int mode, double* C_out) {
    return guarded(h, [&] {
        if (Ma <= 0 || Mb <= 0 || K <= 0 || K % 32 != 0 || !A || !B || !C_out || (mode != 0 && !C0)) throw std::invalid_argument("asm_test_gemm_nt: bad argument");
        HIPCHK(hipSetDevice(h->device));
        double *dA = nullptr, *dB = nullptr, *dC = nullptr;
        dmalloc(&dA, Ma * K); dmalloc(&dB, Mb * K); dmalloc(&dC, Ma * Mb);
        HIPCHK(hipMemcpy(dA, A, Ma * K * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dB, B, Mb * K * sizeof(double), hipMemcpyHostToDevice));
        if (mode != 0) HIPCHK(hipMemcpy(dC, C0, Ma * Mb * sizeof(double), hipMemcpyHostToDevice));
        const char* var = std::getenv("ASM_TEST_GEMM");      // tile variant under test: 32 (32 x 64), 32w (32 x 96), default 64 x 64
        if (var && std::string(var) == "32w")
            hipLaunchKernelGGL(k_gemm_nt32w, dim3((unsigned)((Mb + 95) / 96), (unsigned)((Ma + 31) / 32)), dim3(256), 0, h->stream, (const double*)dA, K, (const double*)dB, K,
                               (const double*)(mode != 0 ? dC : nullptr), Mb, dC, Mb, (int)Ma, (int)Mb, (int)K, mode);
        else if (var && std::string(var) == "32")
            hipLaunchKernelGGL(k_gemm_nt32, dim3((unsigned)((Mb + 63) / 64), (unsigned)((Ma + 31) / 32)), dim3(256), 0, h->stream, (const double*)dA, K, (const double*)dB, K,
                               (const double*)(mode != 0 ? dC : nullptr), Mb, dC, Mb, (int)Ma, (int)Mb, (int)K, mode);
        else
        hipLaunchKernelGGL(k_gemm_nt, dim3((unsigned)((Mb + 63) / 64), (unsigned)((Ma + 63) / 64)), dim3(256), 0, h->stream, (const double*)dA, K, (const double*)dB, K,
                           (const double*)(mode != 0 ? dC : nullptr), Mb, dC, Mb, (int)Ma, (int)Mb, (int)K, mode);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(C_out, dC, Ma * Mb * sizeof(double), hipMemcpyDeviceToHost));
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    });
}

int asm_test_trsm_rows(asm_handle* h, const double* S, int64_t N, const double* R, int64_t nrhs, int backward, double* X_out) {
    return guarded(h, [&] {
        if (N <= 0 || nrhs <= 0 || !S || !R || !X_out) throw std::invalid_argument("asm_test_trsm_rows: bad argument");
        test_load_S(h, S, N);
        Dev d(h);
        d.diag_prepare((int)N, 0, 0.0, 0.0);
        d.chol((int)N);
        const int64_t ldr = round_up(N, 32);
        double *dR = nullptr, *dX = nullptr, *dLt = nullptr;
        dmalloc(&dR, nrhs * ldr); dmalloc(&dX, nrhs * ldr); dmalloc(&dLt, h->Mp * h->Mp);
        HIPCHK(hipMemset(dR, 0, nrhs * ldr * sizeof(double)));
        HIPCHK(hipMemset(dX, 0, nrhs * ldr * sizeof(double)));
        HIPCHK(hipMemset(dLt, 0, h->Mp * h->Mp * sizeof(double)));
        for (int64_t r = 0; r < nrhs; ++r) HIPCHK(hipMemcpy(dR + r * ldr, R + r * N, N * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_transpose_dense, dim3((unsigned)((N + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0, h->stream, (const double*)h->d_S, h->Mp, N, N, dLt, h->Mp, (int64_t)-1);
        d.trsm_rows(dR, dX, ldr, (int)nrhs, (int)N, backward ? dLt : nullptr);
        HIPCHK(hipStreamSynchronize(h->stream));
        const double* out = backward ? dR : dX;
        for (int64_t r = 0; r < nrhs; ++r) HIPCHK(hipMemcpy(X_out + r * N, out + r * ldr, N * sizeof(double), hipMemcpyDeviceToHost));
        (void)hipFree(dR); (void)hipFree(dX); (void)hipFree(dLt);
        d.resolve_timing();
    });
}

int asm_test_gemv(asm_handle* h, const double* A, int64_t M, int64_t K, const double* x, const double* y, double* Ax, double* ATy) {
    return guarded(h, [&] {
        test_alloc(h, M, K);
        Dev d(h);
        for (int64_t i = 0; i < M; ++i)
            HIPCHK(hipMemcpy(h->d_Ah + i * h->ldn, A + i * K, K * sizeof(double), hipMemcpyHostToDevice));
        d.gemv_n(h->d_Ah, x, Ax);
        d.gemv_t(h->d_Ah, y, ATy);
        d.resolve_timing();
    });
}

int asm_test_mfma_peak(asm_handle* h, int iters, int waves_per_simd, double* tflops) {
    return guarded(h, [&] {
        if (!tflops || iters <= 0 || waves_per_simd < 1 || waves_per_simd > 8) throw std::invalid_argument("asm_test_mfma_peak: bad argument");
        HIPCHK(hipSetDevice(h->device));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, h->device));
        int blocks = prop.multiProcessorCount * waves_per_simd;          // 256-thread blocks: 4 wavefronts = one per SIMD
        double* d_out = nullptr;
        HIPCHK(hipMalloc((void**)&d_out, 64));
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_mfma_f64_peak<4>, dim3(blocks), dim3(256), 0, h->stream, d_out, iters / 10 + 1);   // warm-up
        HIPCHK(hipEventRecord(e0, h->stream));
        hipLaunchKernelGGL(k_mfma_f64_peak<4>, dim3(blocks), dim3(256), 0, h->stream, d_out, iters);
        HIPCHK(hipEventRecord(e1, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)blocks * 4.0 * (double)iters * 4.0 * 2.0 * 16 * 16 * 4;
        *tflops = flops / (ms * 1e-3) / 1e12;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(d_out);
    });
}

int asm_test_assemble(asm_handle* h, const double* dE, double* J_out) {
    return guarded(h, [&] {
        if (!h->setup_done) throw std::logic_error("setup first");
        HIPCHK(hipMemcpy(h->d_dE, dE, h->nnz * sizeof(double), hipMemcpyHostToDevice));
        h->J_valid = false;
        Dev d(h);
        d.assemble();
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < h->M; ++i)
            HIPCHK(hipMemcpy(J_out + i * h->n, h->d_J + i * h->ldn, h->n * sizeof(double), hipMemcpyDeviceToHost));
        d.resolve_timing();
    });
}


int asm_sublp_set_ns_basis(asm_handle* h, const int32_t* J, int64_t k) {
    return guarded(h, [&] {
        if (!h->setup_done || k < 0 || (k > 0 && !J)) throw std::invalid_argument("asm_sublp_set_ns_basis: setup first / bad argument");
        for (int64_t a = 0; a < k; ++a)
            if (J[a] < 0 || J[a] >= h->n) throw std::invalid_argument("asm_sublp_set_ns_basis: column out of range");
        h->hint[0].ns_J.assign(J, J + k);
        h->ns_Zk = 0;
    });
}

}  // extern "C"

// =========================================================================================================
// Native SLP driver (rows f2 of SURVEY.md section 8): run!(::SlpLS), slp_line_search.jl:78-215, on the device evaluator - the same
// sequence of library calls as activesetmethods_amd/slp.py (SlpLS.run with Parameters(device_eval=True)), statement by statement, so
// that a whole scenario solve is one C call and B of them can run as fibers of one host thread (asm_batch_slp_run).
// =========================================================================================================
namespace {

struct SlpRunLS {
    asm_handle* h;
    const asm_slp_params& o;
    int64_t n, m;
    vec x, p, lam, mU, mL, ps, df, E, nu;
    double f = 0.0, phi = INF, alpha = 1.0, D = 0.0, prim_infeas = INF, dual_infeas = INF, compl_ = INF;
    bool fr = false;
    int iter = 1, ret = -5, lp_solves = 0, fr_solves = 0, ls_trials = 0;
    asm_slp_result* res;

    SlpRunLS(asm_handle* hh, const asm_slp_params& par, asm_slp_result* r) : h(hh), o(par), n(hh->n), m(hh->m), res(r) {
        x.assign(n, 0.0); p.assign(n, 0.0); lam.assign(m, 0.0); mU.assign(n, 0.0); mL.assign(n, 0.0);
        ps.assign(2 * std::max<int64_t>(m, 1), 0.0); df.assign(n, 0.0); E.assign(std::max<int64_t>(m, 1), 0.0); nu.assign(m, 0.0);
    }
    void chk(int rc, const char* what) {
        if (rc != ASM_OK) throw HipError(std::string(what) + ": " + h->err);
    }
    bool feasible_enough() const { return prim_infeas <= o.tol_infeas; }
    double finite_or_zero(double v) const { return std::isfinite(v) ? v : 0.0; }

    void run(const double* x0) {
        // slp_line_search.jl:96-104 (the clamp tests x_U > -Inf, sic)
        for (int64_t j = 0; j < n; ++j) {
            double v = x0[j];
            if (h->v_lb[j] > -INF) v = std::max(v, h->v_lb[j]);
            if (h->v_ub[j] > -INF) v = std::min(v, h->v_ub[j]);
            x[j] = v;
        }
        iter = 1;
        while (true) {
            if (o.max_lp_solves > 0 && lp_solves >= o.max_lp_solves) break;
            asmb::next_cycle();
            asmb::barrier(10);
            chk(asm_eval_functions(h, x.data(), &f, df.data(), E.data()), "asm_eval_functions");      // :109-110
            alpha = 0.0;
            double nrm[4];
            chk(asm_slp_norms(h, lam.data(), mU.data(), mL.data(), nrm), "asm_slp_norms");             // :113-118, the previous LP's multipliers
            prim_infeas = nrm[0]; dual_infeas = nrm[2]; compl_ = nrm[3];
            int32_t status = 0;
            chk(asm_sublp_solve_resident(h, 1000.0, fr ? 1 : 0, p.data(), lam.data(), mU.data(), mL.data(), ps.data(), &status), "asm_sublp_solve_resident");   // :122-123
            lp_solves += 1;
            if (fr) fr_solves += 1;
            if (res) {
                const int pth = h->stats.path;
                if (pth >= 0 && pth < 12) res->paths[pth] += 1;
                res->ipm_iters += h->stats.ipm_iters;
                res->ns_cold += h->stats.ns_cold;
            }
            if (status != ASM_OPTIMAL && status != ASM_INFEASIBLE) {                                     // :127-133
                if (feasible_enough()) ret = 6;
                break;
            }
            if (status == ASM_INFEASIBLE) {                                                              // :135-147
                if (fr) { ret = feasible_enough() ? 6 : 2; break; }
                fr = true;
                continue;
            }
            for (int64_t i = 0; i < m; ++i) nu[i] = iter == 1 ? std::fabs(lam[i]) : std::max(nu[i], std::fabs(lam[i]));       // :251-261
            asmb::barrier(910);
            int trials = 0, ok = 0;
            double phi_a = 0.0;
            // compute_phi, compute_derivative, compute_alpha (:222-244) - as asm_slp_merit (modes 0, 1) + asm_slp_line_search, with one upload
            chk(guarded(h, [&] { slp_merit_and_search(h, p.data(), nu.data(), ps.data(), fr ? 1 : 0, finite_or_zero(prim_infeas), o.eta, o.tau, o.min_alpha, &phi, &D, &alpha,
                                                      &phi_a, &trials, &ok); }), "slp_merit_and_search");
            ls_trials += trials;
            if (!ok && fr) ret = -3;
            const bool valid = ok != 0;
            if (iter >= o.max_iter) { ret = feasible_enough() ? 6 : -1; break; }                         // :160-166
            double pmax = 0.0;
            for (int64_t j = 0; j < n; ++j) pmax = std::max(pmax, std::fabs(p[j]));
            if ((feasible_enough() && compl_ <= o.tol_residual) || pmax <= o.tol_direction) {            // :168-182
                if (fr) { fr = false; iter += 1; continue; }
                if (dual_infeas <= o.tol_residual) { ret = 0; break; }
            }
            if (!valid) {                                                                                 // :184-199
                if (ret == -3) { ret = feasible_enough() ? 6 : 2; break; }
                fr = true;
                iter += 1;
                continue;
            }
            for (int64_t j = 0; j < n; ++j) x[j] = x[j] + alpha * p[j];                                  // :201-203
            iter += 1;
        }
        // :208-214: objective at the final point
        double f_end = 0.0;
        vec Et(std::max<int64_t>(m, 1));
        asmb::barrier(990);
        chk(asm_eval_constraints(h, x.data(), &f_end, Et.data()), "asm_eval_constraints");
        if (res) {
            res->status = ret; res->iter = iter; res->lp_solves = lp_solves; res->restoration_solves = fr_solves; res->ls_trials = ls_trials;
            res->obj_val = f_end; res->prim_infeas = prim_infeas; res->dual_infeas = dual_infeas; res->compl_ = compl_;
        }
    }
};

void slp_run_ls(asm_handle* h, const asm_slp_params* par, const double* x0, double* x_out, double* lambda, double* mult_x_U, double* mult_x_L, double* g_out,
                asm_slp_result* res) {
    if (!h->ev_ready) throw std::logic_error("asm_slp_run: asm_eval_setup first (the native driver evaluates on the device)");
    if (res) std::memset(res, 0, sizeof(*res));
    SlpRunLS r(h, *par, res);
    r.run(x0);
    if (x_out) std::memcpy(x_out, r.x.data(), h->n * sizeof(double));
    if (lambda && h->m) std::memcpy(lambda, r.lam.data(), h->m * sizeof(double));
    if (mult_x_U) std::memcpy(mult_x_U, r.mU.data(), h->n * sizeof(double));
    if (mult_x_L) std::memcpy(mult_x_L, r.mL.data(), h->n * sizeof(double));
    if (g_out && h->m) std::memcpy(g_out, r.E.data(), h->m * sizeof(double));
}

}  // namespace

// A batch is split into groups: each group = a contiguous range of slots, one stream, one scheduler, one host thread (group 0 runs on the
// calling thread).  While one group's round executes on the device the other groups' host threads merge and launch theirs: two groups
// hide the host side of the rounds (64 case300-sized scenarios on one GPU: 23 -> 32 solves/s).
struct BatchGroup {
    hipStream_t stream = nullptr;
    asmb::Sched sched;
    int lo = 0, hi = 0;             // slots [lo, hi)
    std::exception_ptr err;
};
struct asm_batch {
    int device = 0;
    std::vector<asm_handle*> slots;
    std::vector<BatchGroup*> groups;
    std::string err;
    std::vector<int> J_ref;         // basis columns of the null-space form every scenario starts from (first cold selection of the batch)
    bool setup_done = false;
    asm_batch_stats stats;
};

namespace {
template <class F>
int bguarded(asm_batch* b, F&& fn) {
    if (!b) return ASM_ERR_ARG;
    try {
        fn();
        return ASM_OK;
    } catch (const HipError& e) { b->err = e.what(); return ASM_ERR_HIP; }
    catch (const asmb::BatchError& e) { b->err = e.what(); return ASM_ERR_HIP; }
    catch (const std::invalid_argument& e) { b->err = e.what(); return ASM_ERR_ARG; }
    catch (const std::logic_error& e) { b->err = e.what(); return ASM_ERR_STATE; }
    catch (const std::exception& e) { b->err = e.what(); return ASM_ERR_ARG; }
}
void bcheck(asm_batch* b, int slot, int rc, const char* what) {
    if (rc != ASM_OK) throw HipError(std::string(what) + " (slot " + std::to_string(slot) + "): " + b->slots[slot]->err);
}
// run `work(slot)` for every slot in [0, count): the slots of a group are fibers of the group's thread, launches merged across them
template <class W>
void run_fibers(asm_batch* b, int count, W&& work) {
    const double t0 = asmb::Sched::now_ms();
    auto run_group = [&](BatchGroup* g) {
        try {
            HIPCHK(hipSetDevice(b->device));
            asmb::Sched& S = g->sched;
            for (asmb::Fiber* f : S.fibers) { if (f->stack) munmap(f->stack, f->stack_size); delete f; }
            S.fibers.clear();
            for (int s = g->lo; s < std::min(g->hi, count); ++s) S.add_fiber([&work, s] { work(s); });
            if (!S.fibers.empty()) S.run();
        } catch (...) {
            g->err = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < b->groups.size(); ++k)
        if (b->groups[k]->lo < count) th.emplace_back(run_group, b->groups[k]);
    run_group(b->groups[0]);
    for (auto& t : th) t.join();
    b->stats.wall_ms += asmb::Sched::now_ms() - t0;
    asm_batch_stats& st = b->stats;
    st.rounds = st.ops = st.launches = st.releases = st.blob_bytes = 0;
    st.emit_ms = st.wait_ms = st.host_ms = 0.0;
    st.panel_ms = st.panel_flops = st.panel_bytes = 0.0;
    st.panel_launches = st.panel_ops = 0;
    for (asm_handle* sh : b->slots) { st.panel_flops += sh->kstats.flops[ASM_K_PANEL_KERNEL]; st.panel_bytes += sh->kstats.bytes[ASM_K_PANEL_KERNEL]; }
    for (BatchGroup* g : b->groups) {
        const asmb::Sched& S = g->sched;
        st.panel_ms += S.res_ms; st.panel_launches += (int64_t)S.res_launches; st.panel_ops += (int64_t)S.res_ops;
        st.rounds += (int64_t)S.n_rounds; st.ops += (int64_t)S.n_ops; st.launches += (int64_t)S.n_launches; st.releases += (int64_t)S.n_releases;
        st.blob_bytes += (int64_t)S.blob_bytes; st.emit_ms += S.t_emit_ms; st.wait_ms += S.t_wait_ms; st.host_ms += S.t_host_ms;
    }
    for (BatchGroup* g : b->groups)
        if (g->err) { std::exception_ptr e = g->err; g->err = nullptr; std::rethrow_exception(e); }
}
void batch_free_groups(asm_batch* b) {
    for (BatchGroup* g : b->groups) {
        g->sched.release();
        if (g->stream) (void)hipStreamDestroy(g->stream);
        delete g;
    }
    b->groups.clear();
}
// `n_groups` groups of (almost) equal size over the slots; the slots' launches go to their group's stream
void batch_make_groups(asm_batch* b, int n_groups) {
    HIPCHK(hipSetDevice(b->device));
    for (BatchGroup* g : b->groups) HIPCHK(hipStreamSynchronize(g->stream));
    batch_free_groups(b);
    const int n = (int)b->slots.size();
    n_groups = std::max(1, std::min(n_groups, n));
    for (int k = 0; k < n_groups; ++k) {
        BatchGroup* g = new BatchGroup();
        b->groups.push_back(g);
        HIPCHK(hipStreamCreate(&g->stream));
        g->lo = (int)((int64_t)n * k / n_groups);
        g->hi = (int)((int64_t)n * (k + 1) / n_groups);
        // the all-resident panel kernels of the groups run side by side: together they must fit the chip (workgroups are dealt to the XCDs
        // round-robin and each XCD places its share on its own - a consumer can become resident before its producer, and with the chip
        // full of spinning consumers the producer never would)
        g->sched.init(b->device, g->stream, std::max(16, b->slots[0]->panel_wgs / n_groups));
        if (const char* nb = std::getenv("ASM_BATCH_NO_BARRIERS")) g->sched.use_barriers = !(nb[0] == '1');
        for (int s = g->lo; s < g->hi; ++s) { b->slots[s]->stream = g->stream; b->slots[s]->stream2 = g->stream; }
    }
}
}  // namespace

extern "C" {

int asm_slp_run(asm_handle* h, const asm_slp_params* par, const double* x0, double* x, double* lambda, double* mult_x_U, double* mult_x_L, double* g,
                asm_slp_result* res) {
    return guarded(h, [&] {
        if (!par || !x0) throw std::invalid_argument("asm_slp_run: null pointer");
        slp_run_ls(h, par, x0, x, lambda, mult_x_U, mult_x_L, g, res);
    });
}

int asm_batch_create(int device, int n_slots, asm_batch** out) {
    if (!out || n_slots < 1 || n_slots > 4096) return ASM_ERR_ARG;
    *out = nullptr;
    asm_batch* b = new (std::nothrow) asm_batch();
    if (!b) return ASM_ERR_ARG;
    b->device = device;
    std::memset(&b->stats, 0, sizeof(b->stats));
    int rc = ASM_OK;
    if (hipSetDevice(device) != hipSuccess) { delete b; return ASM_ERR_HIP; }
    for (int s = 0; s < n_slots && rc == ASM_OK; ++s) {
        asm_handle* h = nullptr;
        rc = asm_create(device, &h);
        if (rc != ASM_OK) break;
        // the slot's launches are recorded and merged onto its group's stream: no streams of its own, no look-ahead stream, no event timing
        (void)hipStreamDestroy(h->stream2);
        (void)hipStreamDestroy(h->stream);
        h->stream = nullptr; h->stream2 = nullptr;
        h->batch_slot = true;
        h->timing = 0;
        b->slots.push_back(h);
    }
    if (rc == ASM_OK) {
        try {
            // measured on 64 case300-sized scenarios, one MI355X: 1 group 23.4 solves/s, 2 groups 31.0, 3 groups 35.3, 4 groups 21.5 (the
            // all-resident panel kernels of four streams crowd each other out of the compute units)
            int ng = n_slots >= 48 ? 3 : (n_slots >= 16 ? 2 : 1);
            if (const char* e = std::getenv("ASM_BATCH_GROUPS")) ng = std::max(1, std::atoi(e));
            batch_make_groups(b, ng);
        } catch (const std::exception&) { rc = ASM_ERR_HIP; }
    }
    if (rc != ASM_OK) {
        batch_free_groups(b);
        for (asm_handle* h : b->slots) (void)asm_destroy(h);
        delete b;
        return rc;
    }
    *out = b;
    return ASM_OK;
}

int asm_batch_destroy(asm_batch* b) {
    if (!b) return ASM_ERR_ARG;
    (void)hipSetDevice(b->device);
    for (BatchGroup* g : b->groups) (void)hipStreamSynchronize(g->stream);
    for (asm_handle* h : b->slots) (void)asm_destroy(h);
    batch_free_groups(b);
    delete b;
    return ASM_OK;
}

int asm_batch_set_groups(asm_batch* b, int n_groups) {
    return bguarded(b, [&] {
        if (n_groups < 1) throw std::invalid_argument("asm_batch_set_groups: at least one group");
        batch_make_groups(b, n_groups);
    });
}
int asm_batch_groups(const asm_batch* b) { return b ? (int)b->groups.size() : 0; }

const char* asm_batch_last_error(const asm_batch* b) { return b ? b->err.c_str() : "null batch"; }
int asm_batch_slots(const asm_batch* b) { return b ? (int)b->slots.size() : 0; }
asm_handle* asm_batch_handle(asm_batch* b, int slot) { return (b && slot >= 0 && slot < (int)b->slots.size()) ? b->slots[slot] : nullptr; }

int asm_batch_setup(asm_batch* b, int64_t n, int64_t m, int64_t nnz, const int64_t* j_row, const int64_t* j_col, const double* c_lb, const double* c_ub,
                    const double* v_lb, const double* v_ub) {
    return bguarded(b, [&] {
        for (size_t s = 0; s < b->slots.size(); ++s) bcheck(b, (int)s, asm_sublp_setup(b->slots[s], n, m, nnz, j_row, j_col, c_lb, c_ub, v_lb, v_ub), "asm_sublp_setup");
        b->J_ref.clear();
        b->setup_done = true;
    });
}

int asm_batch_eval_setup(asm_batch* b, int64_t n_rows, const int64_t* aff_ptr, const int64_t* aff_var, const double* aff_coef, const int64_t* quad_ptr,
                         const int64_t* q_v1, const int64_t* q_v2, const double* q_coef, const double* constant, const int64_t* jac_off, const int64_t* g_ptr,
                         const int64_t* g_kind, const double* g_coef, const int64_t* g_other, double objective_scale, int nlp_kind, int64_t nlp_rows,
                         int64_t nlp_nnz, const int64_t* nlp_ipar, int64_t n_ipar, const double* nlp_dpar, int64_t n_dpar) {
    return bguarded(b, [&] {
        if (!b->setup_done) throw std::logic_error("asm_batch_eval_setup: asm_batch_setup first");
        for (size_t s = 0; s < b->slots.size(); ++s)
            bcheck(b, (int)s, asm_eval_setup(b->slots[s], n_rows, aff_ptr, aff_var, aff_coef, quad_ptr, q_v1, q_v2, q_coef, constant, jac_off, g_ptr, g_kind, g_coef,
                                             g_other, objective_scale, nlp_kind, nlp_rows, nlp_nnz, nlp_ipar, n_ipar, nlp_dpar, n_dpar), "asm_eval_setup");
    });
}

int asm_batch_set_ns_basis(asm_batch* b, const int32_t* J, int64_t k) {
    return bguarded(b, [&] {
        if (!b->setup_done || k < 0 || (k > 0 && !J)) throw std::invalid_argument("asm_batch_set_ns_basis: setup first / bad argument");
        b->J_ref.assign(J, J + k);
        for (size_t s = 0; s < b->slots.size(); ++s) bcheck(b, (int)s, asm_sublp_set_ns_basis(b->slots[s], J, k), "asm_sublp_set_ns_basis");
    });
}

// `count` LPs in lockstep, one per slot: asm_sublp_solve with a leading scenario dimension on every array
int asm_batch_sublp_solve(asm_batch* b, int count, const double* c_lb, const double* c_ub, const double* v_lb, const double* v_ub, const double* dE,
                          const double* df, const double* f, const double* E, const double* x_k, const double* delta, const int32_t* feasibility, double* p,
                          double* lambda, double* mult_x_U, double* mult_x_L, double* p_slack, int32_t* status) {
    return bguarded(b, [&] {
        if (!b->setup_done) throw std::logic_error("asm_batch_sublp_solve: asm_batch_setup first");
        if (count < 1 || count > (int)b->slots.size() || !dE || !df || !f || !x_k || !delta || !feasibility || !p || !mult_x_U || !mult_x_L || !status)
            throw std::invalid_argument("asm_batch_sublp_solve: bad count or null pointer");
        const int64_t n = b->slots[0]->n, m = b->slots[0]->m, nnz = b->slots[0]->nnz;
        if (m > 0 && (!E || !lambda || !p_slack)) throw std::invalid_argument("asm_batch_sublp_solve: null pointer");
        HIPCHK(hipSetDevice(b->device));
        run_fibers(b, count, [&](int s) {
            asm_handle* h = b->slots[s];
            if (c_lb && c_ub && v_lb && v_ub) bcheck(b, s, asm_sublp_set_bounds(h, c_lb + s * m, c_ub + s * m, v_lb + s * n, v_ub + s * n), "asm_sublp_set_bounds");
            asmb::next_cycle();
            bcheck(b, s, asm_sublp_solve(h, dE + s * nnz, df + s * n, f[s], E ? E + s * m : nullptr, x_k + s * n, delta[s], feasibility[s], p + s * n,
                                         lambda ? lambda + s * m : nullptr, mult_x_U + s * n, mult_x_L + s * n, p_slack ? p_slack + s * 2 * m : nullptr, status + s),
                   "asm_sublp_solve");
        });
    });
}

// n_scen complete SLP runs (Line Search): the slots' fibers take the scenarios in index order; every array has a leading scenario dimension
int asm_batch_slp_run(asm_batch* b, int64_t n_scen, const double* c_lb, const double* c_ub, const double* v_lb, const double* v_ub, const double* x0,
                      const asm_slp_params* par, double* x, double* lambda, double* mult_x_U, double* mult_x_L, double* g, asm_slp_result* res) {
    return bguarded(b, [&] {
        if (!b->setup_done) throw std::logic_error("asm_batch_slp_run: asm_batch_setup first");
        if (n_scen < 1 || !c_lb || !c_ub || !v_lb || !v_ub || !x0 || !par || !res) throw std::invalid_argument("asm_batch_slp_run: bad argument");
        const int64_t n = b->slots[0]->n, m = b->slots[0]->m;
        HIPCHK(hipSetDevice(b->device));
        if (b->J_ref.empty()) {
            // reference selection of the null-space basis columns: one LP of scenario 0 at its start point on slot 0 (cold selection); EVERY
            // scenario, 0 included, then starts from these columns - the results do not depend on the slot or on what it solved before
            asm_handle* h0 = b->slots[0];
            bcheck(b, 0, asm_sublp_set_bounds(h0, c_lb, c_ub, v_lb, v_ub), "asm_sublp_set_bounds");
            h0->hint[0].ns_J.clear();
            vec xs(n), dfs(n), Es(std::max<int64_t>(m, 1)), pp(n), ll(std::max<int64_t>(m, 1)), uu(n), lo(n), sl(2 * std::max<int64_t>(m, 1));
            for (int64_t j = 0; j < n; ++j) {
                double v = x0[j];
                if (v_lb[j] > -INF) v = std::max(v, v_lb[j]);
                if (v_ub[j] > -INF) v = std::min(v, v_ub[j]);
                xs[j] = v;
            }
            double f0 = 0.0;
            int32_t st0 = 0;
            bcheck(b, 0, asm_eval_functions(h0, xs.data(), &f0, dfs.data(), Es.data()), "asm_eval_functions");
            bcheck(b, 0, asm_sublp_solve_resident(h0, 1000.0, 0, pp.data(), ll.data(), uu.data(), lo.data(), sl.data(), &st0), "asm_sublp_solve_resident");
            b->J_ref = h0->hint[0].ns_J;
        }
        const int count = (int)std::min<int64_t>(n_scen, (int64_t)b->slots.size());
        std::atomic<int64_t> next{0};
        run_fibers(b, count, [&](int s) {
            asm_handle* h = b->slots[s];
            for (;;) {
                const int64_t sc = next.fetch_add(1);
                if (sc >= n_scen) break;
                bcheck(b, s, asm_sublp_set_bounds(h, c_lb + sc * m, c_ub + sc * m, v_lb + sc * n, v_ub + sc * n), "asm_sublp_set_bounds");
                // every scenario starts from the batch's reference basis columns (results do not depend on which slot solved what before)
                if (!b->J_ref.empty()) h->hint[0].ns_J = b->J_ref;
                else h->hint[0].ns_J.clear();
                slp_run_ls(h, par, x0 + sc * n, x ? x + sc * n : nullptr, lambda ? lambda + sc * m : nullptr, mult_x_U ? mult_x_U + sc * n : nullptr,
                           mult_x_L ? mult_x_L + sc * n : nullptr, g ? g + sc * m : nullptr, res + sc);
                res[sc].slot = s;
            }
        });
    });
}

int asm_batch_ns_basis(const asm_batch* b, int32_t* J, int64_t* k) {
    if (!b || !k) return ASM_ERR_ARG;
    *k = (int64_t)b->J_ref.size();
    if (J) for (size_t a = 0; a < b->J_ref.size(); ++a) J[a] = b->J_ref[a];
    return ASM_OK;
}

int asm_batch_get_stats(const asm_batch* b, asm_batch_stats* out) {
    if (!b || !out) return ASM_ERR_ARG;
    *out = b->stats;
    return ASM_OK;
}

}  // extern "C"

#ifdef ASM_UPD_PROF
extern "C" int asm_debug_upd_prof(unsigned long long* out8) {       // diagnostic build only: read and clear the k_syrk_upd phase sums
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_upd_prof), sizeof(z)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_upd_prof), z, sizeof(z)) != hipSuccess;
}
#endif
