/* libasmhip - C ABI of the MI355X-native SLP sub-problem solver.
 *
 * This is the drop-in boundary for the per-iteration sub-LP path of exanauts/ActiveSetMethods:
 * everything `sub_optimize!(slp, Δ)` (src/algorithms/slp.jl:23-47) does below the SLP outer loop -
 * Jacobian assembly (src/algorithms/common.jl:12-20), LP formulation for the normal and the
 * feasibility-restoration phase (src/algorithms/subproblem.jl:51-215, 229-484), the LP solve that the
 * reference hands to an external MOI optimizer (subproblem.jl:490, GLPK in all its tests) and the
 * extraction of step / multipliers (subproblem.jl:494-541).
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every entry returns 0 on success, <0 on misuse or a
 *     HIP error (asm_last_error gives the text).  The LP outcome is reported only through `status`.
 *   - the caller owns every array; the library copies what it needs during the call and keeps no host
 *     pointer afterwards.  Device buffers belong to the handle and are freed by asm_destroy.
 *   - reals are IEEE double, indices int64 and 1-BASED exactly as the reference's `j_str`
 *     (src/MOI_wrapper.jl:726-746); +-Inf bounds are IEEE infinities.
 *   - one handle <-> one HIP stream; distinct handles may be used concurrently from different host
 *     threads / devices (needed for scenario batches); a single handle is not re-entrant.
 *
 * Reference-side binding (Julia `ccall`): see INTEGRATION.md.
 */
#ifndef ASM_HIP_H
#define ASM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct asm_handle asm_handle;

/* MOI.TerminationStatusCode values this path can produce (subproblem.jl:491, 500-539). */
enum { ASM_OPTIMAL = 1, ASM_INFEASIBLE = 2, ASM_DUAL_INFEASIBLE = 3, ASM_OTHER = 4 };

/* error codes */
enum { ASM_OK = 0, ASM_ERR_ARG = -1, ASM_ERR_HIP = -2, ASM_ERR_STATE = -3, ASM_ERR_UNSUPPORTED = -11 };

/* Replaces `MOI.instantiate(slp.options.external_optimizer)` (slp.jl:32). */
int asm_create(int device, asm_handle** out);
int asm_destroy(asm_handle* h);
const char* asm_last_error(const asm_handle* h);

/* Replaces QpModel(...) + create_model! (subproblem.jl:16-215): fixes the pattern and the row/slack
 * layout.  j_row/j_col: nnz entries, 1-based, duplicates allowed, any order (order defines the
 * accumulation order of duplicates).  Rows with c_lb=-Inf and c_ub=+Inf are rejected
 * (ASM_ERR_UNSUPPORTED): create_model! adds no row for them and the reference's indexing breaks. */
int asm_sublp_setup(asm_handle* h, int64_t n, int64_t m, int64_t nnz,
                    const int64_t* j_row, const int64_t* j_col,
                    const double* c_lb, const double* c_ub,
                    const double* v_lb, const double* v_ub);

/* Same pattern, new bounds (a new scenario of a batch: the loads of an ACOPF scenario are constraint bounds): keeps every
 * device buffer, the assembly plan and the evaluator's function store; the retained active sets are dropped.  The kind of
 * each row (==, range, >= only, <= only) must not change - that would be another LP skeleton (create_model!,
 * subproblem.jl:137-214): ASM_ERR_UNSUPPORTED. */
int asm_sublp_set_bounds(asm_handle* h, const double* c_lb, const double* c_ub, const double* v_lb, const double* v_ub);

/* Replaces sub_optimize!(qp, x_k, Δ, feasibility) (subproblem.jl:229-542) with data = LpData(slp)
 * (slp.jl:8-21): dE = Jacobian values in j_str order, df = gradient (c), f = objective value (c0),
 * E = constraint values (b).
 * Outputs (caller-allocated): p[n] (Xsol), lambda[m], mult_x_U[n], mult_x_L[n],
 *   p_slack[2m]: p_slack[2i], p_slack[2i+1] = slack values of row i, second entry NaN when the row has
 *   one slack (Dict{Int,Vector{Float64}} of subproblem.jl:495-505);  status: enum above.
 * INFEASIBLE -> all outputs zero (subproblem.jl:532-536). */
int asm_sublp_solve(asm_handle* h, const double* dE, const double* df, double f, const double* E,
                    const double* x_k, double delta, int feasibility,
                    double* p, double* lambda, double* mult_x_U, double* mult_x_L, double* p_slack,
                    int32_t* status);

/* Split form of the same call for callers that keep the evaluation results resident in HBM:
 * upload once (or write the device buffers directly), then solve from device-resident inputs.
 * asm_sublp_solve == asm_sublp_upload + asm_sublp_solve_resident. */
int asm_sublp_upload(asm_handle* h, const double* dE, const double* df, double f, const double* E,
                     const double* x_k);
int asm_sublp_solve_resident(asm_handle* h, double delta, int feasibility,
                             double* p, double* lambda, double* mult_x_U, double* mult_x_L, double* p_slack,
                             int32_t* status);

/* The LP itself, in the form an MOI `external_optimizer` receives it from the UNMODIFIED reference (subproblem.jl:250-484):
 * after asm_sublp_setup has fixed the skeleton (row kinds -> row types and slack layout of create_model!, subproblem.jl:83-214),
 *   min q'p + w's   s.t.  J_i p + (slack terms of row i) (= | >= | <=) r_i ,  lb <= p <= ub (finite) ,  s >= slo
 * with J from dE (j_str order), r of length m + n_adj (rows m.. = the extra `<=` rows of range constraints, subproblem.jl:200-214).
 * use_slacks = 0: all slack columns fixed at 0 (normal phase, subproblem.jl:410-423), w / slo ignored.
 * Outputs: p[n], s[n_slack] (may be NULL), y[m + n_adj] row duals in the MOI sign convention, z[n] = q - J'y reduced costs,
 * bound_state[n] (-1 at lower, +1 at upper, 0 between; may be NULL), status.  This is what AsmHip.Optimizer calls
 * from MOI.optimize!; asm_sublp_solve is the same solve with the formulation done inside the library. */
int asm_lp_solve(asm_handle* h, const double* dE, const double* q, const double* r, const double* lb, const double* ub,
                 int use_slacks, const double* w, const double* slo,
                 double* p, double* s, double* y, double* z, int32_t* bound_state, int32_t* status);

/* Active set of the last OPTIMAL solve, as the reference could observe it from the LP solution:
 * row_state[m+nadj]: 1 active / 0 inactive (rows m.. are the extra `<=` rows of range constraints,
 * in the order of `adj`, subproblem.jl:200-214); bound_state[n]: -1 at lower, +1 at upper, 0 free;
 * slack_state[nslack]: 1 basic / 0 at its bound (all 0 in the normal phase).
 * Any pointer may be NULL.  n_rows/n_slack receive the array lengths. */
int asm_sublp_active_set(const asm_handle* h, int32_t* row_state, int32_t* bound_state, int32_t* slack_state,
                         int64_t* n_rows, int64_t* n_slack);

/* Drop the retained active sets (the analogue of GLPK's retained basis, slp.jl:38-40). */
int asm_sublp_reset_warm(asm_handle* h);

/* The other retained state of the normal phase: the columns J (0-based) whose projections span null(A_EF) - the null-space form's
 * counterpart of a simplex basis (its complement holds a basis of the equality rows).  k = 0 when the form is not in use.
 * J may be NULL to query k. */
int asm_sublp_ns_basis(const asm_handle* h, int32_t* J, int64_t* k);
/* Observability of the row order of the factorisations (nothing in the reference; oracle/lp_solver.py: rcm_order, row_order): the M rows
 * of the internal LP (m constraint rows, then the extra <= rows of the range constraints) by position in the reverse Cuthill-McKee
 * order of their coupling graph (perm[q] = row at position q; *band = half-bandwidth of their Gram matrix in that order, 0 = natural
 * order in use and perm untouched) and the hard equality rows of the null-space form in their own order (e_rows, *n_e of them, *e_band;
 * *n_e = 0 when the form is not available).  perm / e_rows may be NULL. */
int asm_sublp_row_order(const asm_handle* h, int32_t* perm, int64_t* band, int32_t* e_rows, int64_t* n_e, int64_t* e_band);

/* Statistics of the last solve / accumulated device-kernel timing. */
typedef struct {
    int32_t path;          /* 0 warm, 1 ipm stage0+polish, 2 stage1, 3 stage2, 4 ipm+face (non-unique optimum: least-norm point of the optimal
                            * face + basic multipliers), 5 unpolished (status OTHER), 6 infeasible (IPM duals), 7 infeasible (phase-1 duals),
                            * 8 jammed ipm + polish, 9 ipm+ref (non-unique optimum, projection of the iterate: fallback of 4),
                            * 10 ipm-conv (no active-set solve passed its test, the iterate converged to 1e-10 in all measures is the answer) */
    int32_t polished;
    int32_t ipm_iters;
    int32_t nfact;         /* Cholesky factorisations */
    int32_t eqp;           /* active-set (EQP) solves */
    int32_t M, n, ns;      /* LP rows, columns, slack columns */
    int32_t col_iters;     /* interior-point iterations that factored the n x n column form (restoration LPs) */
    int32_t ns_iters;      /* interior-point iterations that factored the k x k null-space form (normal-phase LPs with many equality rows) */
    int32_t ns_dim;        /* k = dimension of null(A_EF) when that form was set up, else 0 */
    int32_t ns_cold;       /* 1: the basis columns were selected from scratch in this LP (0: the previous LP's columns were re-used) */
    int32_t restored;      /* 1: no interior-point stage ended in a successful polish and the last stage ended 10x worse than the best one - the best
                            * iterate was brought back for the final attempts (best-iterate safeguard) */
    double  ipm_pinf, ipm_dinf, ipm_gap;
    double  kkt_pr, kkt_du;
    double  wall_ms;       /* host wall time of the solve */
} asm_solve_stats;
int asm_sublp_last_stats(const asm_handle* h, asm_solve_stats* out);

/* Device time per kernel family, measured with HIP events on the handle's stream. */
/* ASM_K_SYRK: the Schur build; ASM_K_CHOL / ASM_K_TRSV: whole factorisation / solve (many launches);
 * ASM_K_SYRK_KERNEL: every single launch of the rank-K MFMA kernels k_syrk_upd (Cholesky updates) and k_syrk<T> (Schur builds). */
/* ASM_K_PANEL_KERNEL: every single launch of the dataflow panel kernels k_chol_panel / k_chol_panel_solo (the 64-wide steps of the Cholesky
 * factorisations: the dependent chain the path is bound by since round 3). */
enum { ASM_K_ASSEMBLE = 0, ASM_K_SCALE, ASM_K_GEMV, ASM_K_SYRK, ASM_K_CHOL, ASM_K_TRSV, ASM_K_SYRK_KERNEL, ASM_K_PANEL_KERNEL, ASM_K_COUNT };
typedef struct {
    double  ms[ASM_K_COUNT];      /* accumulated device milliseconds */
    int64_t calls[ASM_K_COUNT];   /* number of timed regions */
    double  flops[ASM_K_COUNT];   /* algorithmic flops issued */
    double  bytes[ASM_K_COUNT];   /* algorithmic bytes moved */
} asm_kernel_stats;
int asm_kernel_stats_get(asm_handle* h, asm_kernel_stats* out);
int asm_kernel_stats_reset(asm_handle* h);
/* 0: no HIP-event timing; 1: every launch of the rank-K and panel kernels (ASM_K_SYRK_KERNEL, ASM_K_PANEL_KERNEL); 2: every family
 * (perturbs latency-bound workloads).  Default from the environment (ASM_HIP_TIMING), else 1. */
int asm_kernel_timing(asm_handle* h, int level);

/* ---- per-iteration reductions that consume J, lambda already in HBM (common.jl:35-44) ----------- */
/* KT_residuals(df, lambda, mult_x_U, mult_x_L, J) with J = the Jacobian assembled by the last
 * upload/solve on this handle. */
int asm_kt_residuals(asm_handle* h, const double* df, const double* lambda,
                     const double* mult_x_U, const double* mult_x_L, double* out);
/* per-row 2-norms of the assembled Jacobian (compute_nu!, slp.jl:54-66). */
int asm_jac_row_norms(asm_handle* h, double* out_m);

/* ---- device-side evaluators: eval_functions! (slp.jl:186-191) without the host ------------------------------------
 * The affine / quadratic evaluator of the MOI wrapper (MOI_wrapper.jl:776-944) on a flattened function store, plus one
 * optional NLP block kernel; Jacobian values are written straight into the handle's dE buffer in the j_str order given to
 * asm_sublp_setup (which must precede this call).  Rows 0..n_rows-1 are the constraint functions in the wrapper's block
 * order (linear <=, >=, ==, quadratic <=, >=, ==; MOI_wrapper.jl:683-689), row n_rows is the objective (MOI_wrapper.jl:809-820).
 *   aff_ptr/quad_ptr [n_rows+2]  term ranges per row;  aff_var, q_v1, q_v2: 0-based variables;  constant [n_rows+1]
 *   jac_off [n_rows+1]           offset of each row's values in dE (affine terms, then per quadratic term 1 or 2 values:
 *                                fill_constraint_jacobian!, MOI_wrapper.jl:889-918)
 *   g_ptr [n+1], g_kind, g_coef, g_other   per-variable contributions to the objective gradient in term order
 *                                (fill_gradient!, MOI_wrapper.jl:827-850; kind 0: += coef, 1: += coef * x[other])
 *   objective_scale              +1 MIN, -1 MAX, 0 FEASIBILITY (MOI_wrapper.jl:1037-1054)
 *   nlp_kind                     0 none; 1 Ohm's-law rows of the polar ACOPF model (4 rows and 20 values per branch;
 *                                ipar = [n_branch, va0, vm0, pf0, pt0, qf0, qt0, f_bus.., t_bus..], dpar = 8 coefficient
 *                                arrays); 2 dense quadratic rows g = A x + 1/2 Q x^2 (dpar = A then Q, row-major)
 * The affine / quadratic part is bit-identical to the host evaluator (no fused multiply-add, the reference's term order). */
int asm_eval_setup(asm_handle* h, int64_t n_rows, const int64_t* aff_ptr, const int64_t* aff_var, const double* aff_coef,
                   const int64_t* quad_ptr, const int64_t* q_v1, const int64_t* q_v2, const double* q_coef,
                   const double* constant, const int64_t* jac_off,
                   const int64_t* g_ptr, const int64_t* g_kind, const double* g_coef, const int64_t* g_other,
                   double objective_scale, int nlp_kind, int64_t nlp_rows, int64_t nlp_nnz,
                   const int64_t* nlp_ipar, int64_t n_ipar, const double* nlp_dpar, int64_t n_dpar);
/* f, df, E at x (returned) and dE (kept in HBM); equivalent to evaluating on the host + asm_sublp_upload. */
int asm_eval_functions(asm_handle* h, const double* x, double* f, double* df, double* E);
/* eval_f + eval_g at a trial point (line search / step quality); does not touch the inputs of the next LP. */
int asm_eval_constraints(asm_handle* h, const double* x, double* f, double* E);
/* copy of the dE buffer (tests). */
int asm_eval_jacobian_values(asm_handle* h, double* dE_out);

/* ---- per-iteration reductions of the SLP callers on the evaluation results in HBM (need asm_eval_functions) ----------
 * out4 = { norm_violations(Inf), norm_violations(1), KT_residuals, norm_complementarity(Inf) }   (common.jl:35-98). */
int asm_slp_norms(asm_handle* h, const double* lambda, const double* mult_x_U, const double* mult_x_L, double* out4);
/* mode 0: compute_phi(x, alpha, p) (slp.jl:79-115; the trial point is evaluated on the device);
 * mode 1: compute_derivative (slp.jl:122-147).  p_slack as asm_sublp_solve returns it. */
int asm_slp_merit(asm_handle* h, int mode, double alpha, const double* p, const double* nu, const double* p_slack,
                  int feasibility, double prim_infeas, double* out);

/* compute_alpha (slp_line_search.jl:222-244): backtracking alpha = 1, tau, tau^2, ... on the merit function, the trial points evaluated on
 * the device eight at a time - one set of launches with the trial index in the grid (same alpha and merit values as one asm_slp_merit call per
 * trial).  *ok = 1: Armijo test passed at *alpha;
 * *ok = 0: alpha fell below min_alpha with the test still failing (*alpha is that last trial, as the reference leaves it). */
int asm_slp_line_search(asm_handle* h, const double* p, const double* nu, const double* p_slack, int feasibility, double prim_infeas, double phi0,
                        double D, double eta, double tau, double min_alpha, double* alpha, double* phi_alpha, int* trials, int* ok);

/* Seed the retained basis columns of the null-space form (0-based; what asm_sublp_ns_basis returns): the next normal-phase LP builds its
 * basis from them instead of selecting columns from scratch.  asm_sublp_set_bounds keeps the columns (same pattern), drops the basis. */
int asm_sublp_set_ns_basis(asm_handle* h, const int32_t* J, int64_t k);

/* ---- native SLP caller: run!(::SlpLS) (slp_line_search.jl:78-215) with every evaluation and reduction on the device ------------------
 * The same sequence of library calls as the host drivers make per outer iteration (asm_eval_functions, asm_slp_norms,
 * asm_sublp_solve_resident, asm_slp_merit x 2, asm_slp_line_search), so that one scenario solve is ONE call.  Parameters: src/parameters.jl:17-28. */
typedef struct {
    int32_t max_iter;          /* parameters.jl: max_iter */
    int32_t max_lp_solves;     /* 0 = no cap (bench: fixed number of steps) */
    double  tol_direction, tol_residual, tol_infeas, eta, tau, min_alpha;
} asm_slp_params;
typedef struct {
    int32_t status;            /* slp.ret: 0 optimal, 2 infeasible, 6 almost feasible, -1 iteration limit, -3 line-search failure, -5 not finished */
    int32_t iter, lp_solves, restoration_solves, ls_trials, slot;
    int32_t paths[12];         /* histogram of asm_solve_stats.path over the LPs of the run */
    int32_t ipm_iters, ns_cold;
    double  obj_val, prim_infeas, dual_infeas, compl_;
} asm_slp_result;
/* x0[n] -> x[n], lambda[m], mult_x_U[n], mult_x_L[n], g[m] (constraint values at the last evaluated iterate); any output may be NULL.
 * Needs asm_sublp_setup + asm_eval_setup on the handle. */
int asm_slp_run(asm_handle* h, const asm_slp_params* par, const double* x0, double* x, double* lambda, double* mult_x_U, double* mult_x_L,
                double* g, asm_slp_result* res);

/* ---- scenario batches: B sub-problems with the same pattern advance through ONE launch sequence on ONE stream --------------------------
 * The reference has no batching (one Optimizer <-> one Model <-> one SLP object, src/MOI_wrapper.jl:1093-1152); these entries are
 * SURVEY.md section 8(b)'s "batch variants with a leading scenario dimension".  An asm_batch owns n_slots handles on one device; a batch
 * call runs one fiber per slot on the calling thread, records every slot's kernel launches and merges equal launches of different slots
 * into one (scenario index in the grid, argument table in HBM).  Results are bit-identical to the per-handle calls.
 * Arrays carry a leading scenario dimension (row-major, scenario s at offset s * length). */
typedef struct asm_batch asm_batch;
int asm_batch_create(int device, int n_slots, asm_batch** out);
int asm_batch_destroy(asm_batch* b);
const char* asm_batch_last_error(const asm_batch* b);
int asm_batch_slots(const asm_batch* b);
/* The slots are split into groups: one stream and one host thread each (group 0 on the calling thread), so that one group's host work
 * (merging, launching) overlaps the other groups' device work.  Default: 2 groups from 16 slots on, 3 from 48 on (ASM_BATCH_GROUPS overrides). */
int asm_batch_set_groups(asm_batch* b, int n_groups);
int asm_batch_groups(const asm_batch* b);
/* the handle of a slot: the per-handle entries (statistics, asm_sublp_active_set, ...) work on it between batch calls */
asm_handle* asm_batch_handle(asm_batch* b, int slot);
/* asm_sublp_setup / asm_eval_setup for every slot (same pattern, same functions) */
int asm_batch_setup(asm_batch* b, int64_t n, int64_t m, int64_t nnz, const int64_t* j_row, const int64_t* j_col,
                    const double* c_lb, const double* c_ub, const double* v_lb, const double* v_ub);
int asm_batch_eval_setup(asm_batch* b, int64_t n_rows, const int64_t* aff_ptr, const int64_t* aff_var, const double* aff_coef,
                         const int64_t* quad_ptr, const int64_t* q_v1, const int64_t* q_v2, const double* q_coef,
                         const double* constant, const int64_t* jac_off,
                         const int64_t* g_ptr, const int64_t* g_kind, const double* g_coef, const int64_t* g_other,
                         double objective_scale, int nlp_kind, int64_t nlp_rows, int64_t nlp_nnz,
                         const int64_t* nlp_ipar, int64_t n_ipar, const double* nlp_dpar, int64_t n_dpar);
/* basis columns every scenario of asm_batch_slp_run starts from (default: selected by one LP of scenario 0 on slot 0) */
int asm_batch_set_ns_basis(asm_batch* b, const int32_t* J, int64_t k);
int asm_batch_ns_basis(const asm_batch* b, int32_t* J, int64_t* k);
/* asm_sublp_set_bounds (when the four bound arrays are given) + asm_sublp_solve for `count` <= n_slots scenarios in lockstep:
 * c_lb, c_ub [count x m]; v_lb, v_ub, df, x_k, p, mult_x_U, mult_x_L [count x n]; dE [count x nnz]; f, delta, feasibility, status [count];
 * E, lambda [count x m]; p_slack [count x 2m]. */
int asm_batch_sublp_solve(asm_batch* b, int count, const double* c_lb, const double* c_ub, const double* v_lb, const double* v_ub,
                          const double* dE, const double* df, const double* f, const double* E, const double* x_k,
                          const double* delta, const int32_t* feasibility,
                          double* p, double* lambda, double* mult_x_U, double* mult_x_L, double* p_slack, int32_t* status);
/* n_scen complete SLP runs (asm_slp_run per scenario; n_scen may exceed n_slots: a slot takes the next scenario in index order when
 * it finishes one).  c_lb, c_ub, g [n_scen x m]; v_lb, v_ub, x0, x, mult_x_U, mult_x_L [n_scen x n]; lambda [n_scen x m]; res [n_scen].
 * Needs asm_batch_setup + asm_batch_eval_setup. */
int asm_batch_slp_run(asm_batch* b, int64_t n_scen, const double* c_lb, const double* c_ub, const double* v_lb, const double* v_ub,
                      const double* x0, const asm_slp_params* par,
                      double* x, double* lambda, double* mult_x_U, double* mult_x_L, double* g, asm_slp_result* res);
/* what the launch merging did since the batch was created */
typedef struct {
    int64_t rounds;        /* scheduler rounds (one blob copy + one completion wait each) */
    int64_t ops;           /* operations the slots recorded (= launches the per-handle path would have made) */
    int64_t launches;      /* launches actually made */
    int64_t releases;      /* barrier groups released */
    int64_t blob_bytes;    /* argument tables + copy payloads sent to the device */
    double  emit_ms, wait_ms, host_ms, wall_ms;   /* merging + launching, waiting for the device, solver host code, total */
    /* the dataflow panel kernels (k_chol_panel*: ASM_K_PANEL_KERNEL of a handle) as the batch runs them: HIP events around every merged launch
     * on its group's stream (ASM_HIP_TIMING != 0), the launches the slots recorded, and their algorithmic flops / bytes summed over the slots */
    double  panel_ms;
    int64_t panel_launches, panel_ops;
    double  panel_flops, panel_bytes;
} asm_batch_stats;
int asm_batch_get_stats(const asm_batch* b, asm_batch_stats* out);

/* Test hook: the matrices loaded by asm_test_cholesky / asm_test_chol_solve / asm_test_trsm_rows are banded with this half-bandwidth
 * (0 = dense): factorisation and substitutions stop at the band, as they do for S0 = A_EF A_EF' of the null-space form (its equality
 * rows are put in reverse Cuthill-McKee order at set-up). */
int asm_test_set_band(asm_handle* h, int band);

/* ---- kernel-level test hooks (used by tests/ to check each kernel against NumPy) ---------------- */
int asm_test_syrk(asm_handle* h, const double* A, int64_t M, int64_t K, const int32_t* idx, int64_t Ms,
                  const double* theta, const double* diag, double* S_out /* Ms*Ms, lower valid */, int tile);
/* S[a,b] -= sum_k P[a,k] P[b,k], a >= b (b < MsB when MsB >= 0), the block at (srow0, srow0) of a larger matrix: the Cholesky update */
int asm_test_syrk_update(asm_handle* h, const double* P /* Ms*K */, int64_t Ms, int64_t K, int64_t MsB, int64_t srow0,
                         double* S_inout /* Ms*Ms */, int tile);
int asm_test_cholesky(asm_handle* h, const double* S /* N*N sym */, int64_t N, double* L_out /* N*N lower */);
int asm_test_chol_solve(asm_handle* h, const double* S, int64_t N, const double* b, double* x);
/* the bounded wait of the dataflow panel kernel with a producer that never publishes: returns ASM_ERR_HIP (reported once), the
 * handle stays usable */
int asm_test_panel_timeout(asm_handle* h, int workgroups);
/* on != 0: every active-set attempt (polish) of the following LPs on this handle fails, so that the solve ends on its last resort - the
 * converged interior iterate, asm_solve_stats.path 10 (oracle: tests patch eqp_loop / face_polish the same way) */
int asm_test_no_polish(asm_handle* h, int on);
/* C = (mode 1: C0) -/+ A B'  (A: Ma x K, B: Mb x K, row-major, K a multiple of 32) - the product kernel of the multi-right-hand-side
 * triangular solves of the null-space form */
int asm_test_gemm_nt(asm_handle* h, const double* A, const double* B, const double* C0, int64_t Ma, int64_t Mb, int64_t K, int mode,
                     double* C_out);
/* rows of R (nrhs x N) solved against the Cholesky factor of S: forward only (L x = r) or forward + backward (S x = r) */
int asm_test_trsm_rows(asm_handle* h, const double* S, int64_t N, const double* R, int64_t nrhs, int backward, double* X_out);
int asm_test_gemv(asm_handle* h, const double* A, int64_t M, int64_t K, const double* x, const double* y,
                  double* Ax, double* ATy);
int asm_test_assemble(asm_handle* h, const double* dE, double* J_out /* (m+nadj)*n */);
/* FP64 MFMA peak probe (back-to-back v_mfma_f64_16x16x4_f64, registers only): measured roofline denominator. */
int asm_test_mfma_peak(asm_handle* h, int iters, int waves_per_simd, double* tflops);

#ifdef __cplusplus
}
#endif
#endif
