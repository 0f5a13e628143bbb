/*
 * bluest_hip.h -- C-ABI of libbluest_hip.so: the MI355X (gfx950) implementation of BLUEST's sample-allocation
 * hot path (Phi(m) assembly, V = e0^T Phi^-1 e0, grad V, simplex projection for the SPG solver).
 *
 * This is the drop-in boundary.  Part 1 mirrors, one entry point per function, the reference's only native
 * module `_cmisc_bluest` (/root/reference/bluest/cmisc.cpp:99-110): the same arguments in the same order and
 * the same in-place accumulation contract, as plain pointers and sizes.  Part 2 is the device-resident "plan"
 * that the Python operators SAP / MOSAP (bluest/sap.py:131-143, bluest/mosap.py:86-100) sit on: group tables
 * and per-group inverse covariances stay in HBM, one call evaluates V and grad V for every output.
 *
 * Conventions
 *   - every function returns BLUEST_OK (0) or a BLUEST_ERR_* code; bluest_last_error() gives the message
 *     of the last failure on the calling thread.  Nothing throws, nothing takes ownership of caller memory.
 *   - "hd" pointers (Part 1) may be HOST or DEVICE pointers (detected with hipPointerGetAttributes); host
 *     buffers are staged through HBM and the call is synchronous.  "dev" pointers (Part 2) must be device
 *     pointers; those calls are asynchronous on `stream` (a hipStream_t passed as void*, NULL = default).
 *   - arithmetic is IEEE float64, indices int64, exactly as the reference (cmisc.cpp uses `long int`).
 *   - there is NO CPU fallback: without a usable GPU every compute entry point fails with BLUEST_ERR_NOGPU.
 */
#ifndef BLUEST_HIP_H
#define BLUEST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLUEST_ABI_VERSION 1

#define BLUEST_OK          0
#define BLUEST_ERR_ARG     1   /* bad argument (null pointer, size out of range) */
#define BLUEST_ERR_HIP     2   /* a HIP runtime call failed */
#define BLUEST_ERR_NOGPU   3   /* no gfx950 device visible */
#define BLUEST_ERR_STATE   4   /* plan not finalized / already finalized */

#define BLUEST_MAX_MODELS  64  /* n  (Phi is n x n) */
#define BLUEST_MAX_GROUP   16  /* k_max */

/* per-(candidate,output) evaluation status written by bluest_plan_solve / bluest_plan_eval */
#define BLUEST_EVAL_OK         0
#define BLUEST_EVAL_INF        1  /* max|m| < 0.05  -> V = +inf, grad = +inf   (bluest/misc.py:464,484)      */
#define BLUEST_EVAL_NO_MODEL0  2  /* model 0 not sampled -> reference asserts   (bluest/misc.py:470)          */
#define BLUEST_EVAL_SINGULAR   3  /* restricted Phi not positive definite       (bluest/misc.py:473-474)      */

int         bluest_abi_version(void);
const char *bluest_last_error(void);
int         bluest_device_count(int *count);
int         bluest_device_name(char *buf, int buflen);

/* ------------------------------------------------------------------------------------------------------
 * Part 1 -- stateless mirrors of _cmisc_bluest (reference: bluest/cmisc.cpp)
 * ---------------------------------------------------------------------------------------------------- */

/* replaces assemble_psi_c (cmisc.cpp:10-23): psi[(N*g_j+g_l), i] += invcov_i[j,l];
 * psi is C-order (N*N, Lk), zero-filled by the caller (bluest/misc.py:601). */
int bluest_assemble_psi(double *psi_hd, int N, int k, int64_t Lk, const int64_t *groupsk_hd,
                        const double *invcovsk_hd);

/* replaces objectiveK_c<double> (cmisc.cpp:25-40, overload :104): PHI[N*g_j+g_l] += m_i*invcov_i[j,l] */
int bluest_objectiveK_f64(double *PHI_hd, int N, int k, int64_t Lk, const double *mk_hd,
                          const int64_t *groupsk_hd, const double *invcovsk_hd);

/* replaces objectiveK_c<long int> (cmisc.cpp:25-40, overload :105): integer sample counts */
int bluest_objectiveK_i64(double *PHI_hd, int N, int k, int64_t Lk, const int64_t *mk_hd,
                          const int64_t *groupsk_hd, const double *invcovsk_hd);

/* replaces gradK_c (cmisc.cpp:58-72): grad_i += v[g]^T invcov_i v[g], v = invPHI[0] (length >= max(g)+1) */
int bluest_gradK(double *grad_hd, int k, int64_t Lk, const int64_t *groupsk_hd, const double *invcovsk_hd,
                 const double *invPHI_0_hd, int n_models);

/* replaces cleanupK_c (cmisc.cpp:42-56), INCLUDING its `=` (not `+=`) at line 51: only the l=k-1 term
 * survives, X[g_j, i] = invcov_i[j,k-1]*v[g_{k-1}].  X is C-order (N, Lk). */
int bluest_cleanupK(double *X_hd, int k, int64_t Lk, const int64_t *groupsk_hd, const double *invcovsk_hd,
                    const double *invPHI_0_hd, int n_models);

/* replaces hessKQ_c (cmisc.cpp:74-97): hess[ik,iq] += a_k(ik)^T invPHI[g^k,g^q] a_q(iq),
 * a_k(ik)_j = sum_l v[g^k_l] invcov^k[l,j]; hess is C-order (Lk, Lq); invPHI is C-order (N, N). */
int bluest_hessKQ(double *hess_hd, int N, int k, int q, int64_t Lk, int64_t Lq, const int64_t *groupsk_hd,
                  const int64_t *groupsq_hd, const double *invcovsk_hd, const double *invcovsq_hd,
                  const double *invPHI_hd);

/* replaces the per-group numpy.linalg.pinv(C[g,g]) loop of SAP.__init__ (bluest/sap.py:69-79):
 * invcov_i = pinv(C[g_i,g_i]) (symmetric eigen-decomposition, cut-off 1e-15*max|lambda| as numpy's default
 * rcond), flattened C-order (Lk*k*k). */
int bluest_group_pinv(const double *C_hd, int N, int k, int64_t Lk, const int64_t *groupsk_hd,
                      double *invcovsk_hd);

/* ------------------------------------------------------------------------------------------------------
 * Part 2 -- device-resident plan (operator level: bluest/sap.py:131-143, bluest/mosap.py:86-100)
 *
 * A plan holds, for each output o, the groups and inverse covariances of SAP_o re-laid out in HBM for the two
 * streaming passes (destination-major symmetric CSR for the Phi pass, group-major 64-wide tiles for the
 * gradient pass), plus the map from SAP_o's local group numbering to the global allocation vector m
 * (`mappings[o]` of bluest/mosap.py:54-67).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct bluest_plan_s *bluest_plan_t;

/* n_models = N (<= BLUEST_MAX_MODELS); L_global = length of the global allocation vector m */
int bluest_plan_create(bluest_plan_t *plan, int n_models, int64_t L_global);
int bluest_plan_destroy(bluest_plan_t plan);

/* Plan lifetime vs hipGraph capture.  Releasing a plan frees device memory, and a free issued while some stream of the
 * process is being captured invalidates that capture.  Callers whose host language may drop a plan at any time (garbage
 * collection, reference counting) bracket their captures: bluest_capture_guard(1) before hipStreamBeginCapture,
 * bluest_capture_guard(0) after hipStreamEndCapture.  While a guard is open bluest_plan_destroy only parks the plan; the
 * closing call releases everything parked.  Guards nest.  bluest_deferred_plans reports how many plans are parked. */
int bluest_capture_guard(int on);
int bluest_deferred_plans(int *count);

/* Add output o (call once per output, in order).  K = max group size of this output; sizes[k-1] = L_k for
 * k = 1..K; groups = concat_k (L_k*k) model indices; invcovs = concat_k (L_k*k*k) (HOST pointers, copied);
 * mapping = L_o global indices (m_o = m[mapping]) or NULL for the identity (requires L_o == L_global). */
int bluest_plan_add_output(bluest_plan_t plan, int K, const int64_t *sizes, const int64_t *groups,
                           const double *invcovs, const int64_t *mapping);

/* Same, but the per-group pseudo-inverses are computed on the GPU from the n x n covariance C (host pointer);
 * if invcovs_out (host) is non-NULL the inverses are also returned in the reference layout. */
int bluest_plan_add_output_cov(bluest_plan_t plan, const double *C, int K, const int64_t *sizes,
                               const int64_t *groups, const int64_t *mapping, double *invcovs_out);

/* Reference-layout pseudo-inverses of output o (concat_k L_k*k*k doubles) back to a host buffer: they stay on the device
 * after bluest_plan_add_output_cov (the plan is built from them there) and are only fetched when the host asks, e.g. for the
 * `invcovs` attribute of SAP (bluest/sap.py:69-79).  bluest_plan_gather_invcovs fetches the k x k blocks of SOME groups of
 * output o (local indices, any order, blocks written one after the other) -- what a restricted plan needs. */
int bluest_plan_get_invcovs(bluest_plan_t plan, int output, double *invcovs_out);
int bluest_plan_gather_invcovs(bluest_plan_t plan, int output, const int64_t *local_idx, int64_t n, double *out);

/* A NEW finalized plan over the sub-list `keep` (n_keep strictly ascending global group indices) of a finalized plan's groups:
 * allocation vectors of the new plan have length n_keep (entry i = group keep[i]); every output keeps the groups whose global
 * index is in `keep`, with the parent's pseudo-inverses gathered on the device (no host round trip).  This is the operator of
 * bluest/sap.py:53-79 built on a subset of the groups -- the working set of the solver.  Fails with BLUEST_ERR_ARG if some
 * output would be left without a group containing model 0 (its variance would be infinite whatever the allocation).
 * bluest_plan_output_layout returns what the host mirror needs of one output: K, the sizes L_k (K entries) and, if `mapping`
 * is non-NULL, the global indices of its groups (sum L_k entries); pass sizes = NULL to query K alone. */
int bluest_plan_restrict(bluest_plan_t parent, const int64_t *keep, int64_t n_keep, int max_candidates, bluest_plan_t *restricted);
int bluest_plan_output_layout(bluest_plan_t plan, int output, int *K, int64_t *sizes, int64_t *mapping);

/* Build the HBM layouts.  max_candidates = largest number of allocation vectors evaluated per call. */
int bluest_plan_finalize(bluest_plan_t plan, int max_candidates);

int bluest_plan_n_outputs(bluest_plan_t plan, int *n_outputs);
/* total length of the concatenated per-output gradient (sum_o L_o) and the offset of output o inside it */
int bluest_plan_grad_layout(bluest_plan_t plan, int64_t *grad_len, int64_t *offsets /* n_outputs */);
/* HBM bytes the Phi pass / gradient pass stream per candidate (actual layout, for roofline accounting) */
int bluest_plan_traffic(bluest_plan_t plan, int64_t *phi_bytes, int64_t *grad_bytes);
/* MATRIX-FREE evaluation (csrc/matfree.hip; the form BASELINE.json's north star names): the inverse of every group's covariance
 * block (bluest/sap.py:69-79) is recomputed in registers where bluest/cmisc.cpp:25-40,58-72 read the stored one, so an evaluation
 * reads the groups' model indices and m only.  Chosen at bluest_plan_finalize for plans that qualify (outputs given by their
 * covariance, group sizes <= 8, <= 48 models, every block safely positive definite) when BLUEST_MATFREE=1, or on its own when the
 * stored streams exceed 64 MB; single-candidate bluest_plan_eval / bluest_plan_phi / bluest_plan_solve_grad then take it.
 * The GRADIENT pass alone is matrix-free on every plan that qualifies (BLUEST_MATFREE=2 forces exactly that, 0 forbids everything): the
 * stored Phi pass is then followed by one kernel that folds its partials, solves and recomputes the group factors for the gradient.
 * *matfree = 1: Phi and gradient matrix-free, 2: the gradient only, 0: stored inverses everywhere; *mf_bytes = bytes a fully
 * matrix-free evaluation reads and writes per candidate. */
int bluest_plan_matfree(bluest_plan_t plan, int *matfree, int64_t *mf_bytes);
/* size in doubles of one candidate's Phi-pass result: n_outputs * (N*N + 2*N + 1), see bluest_plan_phi */
int bluest_plan_phi_len(bluest_plan_t plan, int64_t *len);

/* Phase A (a4/a5/a6 of SURVEY.md 8a): for every candidate c and output o write into phi_dev[c][o] a record of
 * N*N + 2*N + 1 doubles: Phi_o(m_c) WITHOUT the delta*I term (N*N); then for each model a, 1.0 if some group
 * containing a has |m_i| > 1e-6 else 0.0 (N doubles: the `idx` of bluest/misc.py:453-457); then for each model
 * a, 1.0 if some group containing a has m_i != 0 (N doubles: the support of Phi); then 1.0 if max|m| >= 0.05
 * (1 double, bluest/misc.py:464).  All fields are sums/indicators, so partial records from several GPUs (each
 * holding a shard of the groups) combine with one all-reduce(SUM) and are then tested with `> 0`.
 * m_dev: n_cand vectors of length L_global, m_stride doubles apart. */
int bluest_plan_phi(bluest_plan_t plan, const double *m_dev, int n_cand, int64_t m_stride, double *phi_dev,
                    void *stream);

/* Diagnostics: launch ONLY the Phi chunk kernel (partials stay in the plan's workspace); lets bench.py time
 * the streaming kernel alone with HIP events. */
int bluest_plan_phi_chunks(bluest_plan_t plan, const double *m_dev, int n_cand, int64_t m_stride, void *stream);

/* Phase B (a7/a8): from the (all-reduced) records, V = (Phi[idx,idx]^-1)_00 on the sampled models
 * (bluest/misc.py:467-472,490), v = row 0 of pinv(Phi + delta I) (bluest/misc.py:487), status codes above.
 * var_dev: n_cand*n_outputs, v_dev: n_cand*n_outputs*N, status_dev: n_cand*n_outputs int32. */
int bluest_plan_solve(bluest_plan_t plan, const double *phi_dev, int n_cand, double delta, double *var_dev,
                      double *v_dev, int32_t *status_dev, void *stream);

/* Phase B for a RANK-DEFICIENT restricted Phi (status BLUEST_EVAL_SINGULAR from bluest_plan_solve / _eval): what the
 * reference's variance_GH returns there -- V = pinv(Phi[idx,idx])[0,0] (bluest/misc.py:490) and v = row 0 of pinv(Phi)
 * (bluest/misc.py:487) with numpy's cut-off (eigenvalues <= 1e-15 * max|lambda| dropped), by a cyclic Jacobi
 * eigen-decomposition.  Same arguments as bluest_plan_solve; the status is never BLUEST_EVAL_SINGULAR.  Rare path:
 * one wavefront per (candidate, output), ~ms. */
int bluest_plan_solve_pinv(bluest_plan_t plan, const double *phi_dev, int n_cand, double delta, double *var_dev,
                           double *v_dev, int32_t *status_dev, void *stream);

/* Phase C (a9): grad_o,i = -v[g_i]^T invcov_i v[g_i] for every group of every output (this GPU's shard);
 * grad_dev: n_cand rows of grad_len doubles (row stride grad_stride); +inf where status == BLUEST_EVAL_INF. */
int bluest_plan_grad(bluest_plan_t plan, const double *v_dev, const int32_t *status_dev, int n_cand,
                     double *grad_dev, int64_t grad_stride, void *stream);

/* A + B (+ C if grad_dev != NULL) back to back on one stream, single GPU (MOSAP.variances / variance_GH). */
int bluest_plan_eval(bluest_plan_t plan, const double *m_dev, int n_cand, int64_t m_stride, double delta,
                     double *var_dev, double *grad_dev, int64_t grad_stride, int32_t *status_dev, void *stream);

/* Fold the per-output gradients back onto the global allocation vector:
 * out[c][j] = scale[j] * sum_o coef[c][o] * grad_o[c][local_o(j)]   (0 where output o has no group j).
 * coef_dev: n_cand*n_outputs; scale_dev: L_global or NULL (=1). */
int bluest_plan_combine_grad(bluest_plan_t plan, const double *grad_dev, int64_t grad_stride,
                             const double *coef_dev, const double *scale_dev, int n_cand, double *out_dev,
                             int64_t out_stride, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Part 3 -- SPG building block (NEW solver="spg"; algorithm of bluest/spg.py:39-132 with proj = simplex)
 *
 * p = argmin sum_i (p_i - u_i)^2 / s_i  over {p >= 0, sum p = z},  u = x - lambda * s * g,  d = p - x.
 *   floor == 0 : s = 1, the plain Euclidean projection P_simplex(x - lambda*g) (reference-style SPG step);
 *   floor  > 0 : s_i = max(x_i, floor), the variable-metric ("entropic") SPG step used by solver="spg" by default.
 * stats_dev[0] = g.d, stats_dev[1] = max|d|, stats_dev[2] = tau (threshold on the ratios, after shifting by their
 * max), stats_dev[3] = number of positive entries of p.  g_dev may be NULL (then lambda is ignored: p = P(x)).
 * d_dev or p_dev may be NULL.
 * work_dev: NULL, or bluest_simplex_workspace_doubles(L) doubles of scratch, ZERO-FILLED once (hipMemset) before its
 * first use and then left alone between calls: with it, vectors longer than 4096 are projected by ONE launch of up to 64
 * workgroups (one workgroup alone moves only ~25-60 GB/s) that exchange their partial sums through tagged mailboxes in
 * the scratch, and the threshold search is warm-started from the previous call on the same scratch (typically 3 passes
 * instead of 10-15).  The result does not depend on the hint.  Keep one scratch per stream: two projections running
 * concurrently on the same scratch would read each other's mailboxes.
 */
int bluest_simplex_workspace_doubles(int64_t L, int64_t *n_doubles);
int bluest_simplex_project(const double *x_dev, const double *g_dev, double lambda, double z, double floor, int64_t L,
                           double *p_dev, double *d_dev, double *stats_dev, double *work_dev, void *stream);

/* Device-resident SPG iteration (bluest/spg.py:68-106 with the control flow on the GPU).  The solver state is an
 * array of BLUEST_SPG_STATE_DOUBLES doubles in HBM (layout: csrc/spg.hip SPG_*, bluest_amd/spg_device.py);
 * every kernel below reads its scalars (lambda, alpha, f, history ...) from it and is a no-op once state[DONE] or
 * state[FAIL] is set, so the loop is a fixed launch sequence of identical STEPS -- direction, T predicated line-search slots,
 * (gradient,) Barzilai-Borwein update -- that needs no host round trip (plain stream launches, or captured in a hipGraph).
 * A trial rejected in the last slot of a step sets state[PENDING]: the next direction launch then forms the next trial point
 * x + alpha*d instead of a new direction and the update stays gated off, i.e. the line search of spg.py:9-35 simply continues
 * in the next step; a step length below 1e-300 or state[MAXFEV] evaluations set state[FAIL].
 *   bluest_plan_set_gate   : the plan's kernels (Phi chunks, solve, gradient, combine) skip when *enable_dev == 0;
 *                            always_v != 0 makes every solve also produce v (kept in the plan workspace)
 *   bluest_spg_direction   : d = P_s(x - lambda*s*g) - x with lambda = state[LAMBDA]; g.d -> state (the multi-workgroup kernel
 *                            leaves per-workgroup partials that bluest_spg_decide folds); optionally also the first trial
 *                            point (alpha = 1): xnew = x + d, m = scale*xnew, *enable = 1.  While state[PENDING]: no new
 *                            direction, xnew = x + alpha*d, m = scale*xnew instead
 *   bluest_spg_trial       : xnew = x + alpha*d, m = scale*xnew, *enable = 1 -- unless the iteration already accepted
 *   bluest_spg_decide      : F(trial) from the per-output variances (p-norm / max), nonmonotone Armijo test
 *                            (spg.py:17,32), safeguarded quadratic interpolation of alpha (spg.py:18-26)
 *                            on the last slot of an iteration also *enable = accepted (gates gradient + combine)
 *   bluest_spg_update      : s, y, Barzilai-Borwein lambda (spg.py:91-106), x <- xnew, g <- gnew, history
 *                            (work_dev: 1024 doubles of scratch for the per-block partial sums)
 *   bluest_spg_converged   : gpmax = max|P_s(x - s*g) - x| -> state; sets state[DONE] when gpmax <= state[EPS] (spg.py:68,99-101)
 */
#define BLUEST_SPG_STATE_DOUBLES 256
int bluest_plan_set_gate(bluest_plan_t plan, const int32_t *enable_dev, int always_v);
/* One line-search slot in two launches instead of three: bluest_plan_eval (value only, single candidate) with
 * bluest_spg_decide fused into the tail of the solve kernel -- the output workgroup that finishes last evaluates the decision. */
int bluest_plan_eval_decide(bluest_plan_t plan, const double *m_dev, double delta, double *var_dev, int32_t *status_dev,
                            double *state_dev, int last_slot, int32_t *enable_dev, void *stream);
/* The same with the gradient of the trial point (bluest_plan_eval's fused solve + gradient launch, decision in its tail): the
 * gradient of a rejected trial is wasted work (~2 us), the gradient of the accepted one is what the update needs -- no separate
 * gradient launch after the line search.  grad_dev: grad_len doubles (bluest_plan_grad_layout). */
int bluest_plan_eval_grad_decide(bluest_plan_t plan, const double *m_dev, double delta, double *var_dev, double *grad_dev,
                                 int32_t *status_dev, double *state_dev, int last_slot, int32_t *enable_dev, void *stream);
/* Group-sharded plans: solve + gradient of this GPU's shard FROM the all-reduced Phi record (bluest_plan_phi, summed over the
 * ranks) in one launch; state_dev != NULL adds the line-search decision in its tail (as bluest_plan_eval_grad_decide). */
int bluest_plan_solve_grad(bluest_plan_t plan, const double *rec_dev, double delta, double *var_dev, double *grad_dev,
                           int32_t *status_dev, double *state_dev, int last_slot, int32_t *enable_dev, void *stream);
int bluest_plan_v_workspace(bluest_plan_t plan, const double **v_dev, const int32_t **status_dev);
int bluest_spg_direction(const double *x_dev, const double *g_dev, double *state_dev, double z, double floor, int64_t L,
                         double *d_dev, const double *scale_dev, double *xnew_dev, double *m_dev, int32_t *enable_dev,
                         double *work_dev, void *stream);
int bluest_spg_converged(const double *x_dev, const double *g_dev, double *state_dev, double z, double floor, int64_t L,
                         double *work_dev, void *stream);
int bluest_spg_trial(const double *x_dev, const double *d_dev, const double *scale_dev, const double *state_dev,
                     double *xnew_dev, double *m_dev, int32_t *enable_dev, int64_t L, void *stream);
int bluest_spg_decide(double *state_dev, const double *var_dev, const int32_t *status_dev, int n_out, int last_slot,
                      int32_t *enable_dev, void *stream);
int bluest_spg_update(double *x_dev, double *g_dev, const double *xnew_dev, const double *gnew_dev, double *state_dev,
                      double floor, int64_t L, double *work_dev, void *stream);
/* the same with the gradient fold of bluest_plan_combine_grad fused in (coefficients from state[COEF..]): gnew is formed on
 * the fly from the per-output gradients grad_dev (single candidate) */
int bluest_spg_update_fused(bluest_plan_t plan, double *x_dev, double *g_dev, const double *xnew_dev, const double *grad_dev,
                            const double *scale_dev, double *state_dev, double floor, double *work_dev, void *stream);
/* gradient of the accepted trial point + bluest_spg_update_fused in one call: bluest_plan_grad followed by the fused update, or --
 * for small plans (K_tot <= 4096: the solver's working set) -- ONE single-workgroup kernel that does both.  v_dev / status_dev as
 * handed out by bluest_plan_v_workspace. */
int bluest_spg_finish(bluest_plan_t plan, const double *v_dev, const int32_t *status_dev, double *x_dev, double *g_dev,
                      const double *xnew_dev, double *grad_dev, const double *scale_dev, double *state_dev, double floor,
                      double *work_dev, void *stream);

/* n_iterations steps of the loop (bluest_spg_direction, `slots` x [bluest_spg_trial,] bluest_plan_eval_grad_decide,
 * bluest_spg_update_fused; on plans with K_tot <= 4096: bluest_plan_eval_decide and bluest_spg_finish), then
 * bluest_spg_converged if check_last: one host call enqueues the launch sequence of a whole window on `stream`.  Arguments as in the calls it is made of; v_ws_dev from bluest_plan_v_workspace; proj_work_dev as work_dev of
 * bluest_simplex_project (may be NULL for L <= 4096). */
int bluest_spg_window(bluest_plan_t plan, double *x_dev, double *g_dev, double *d_dev, double *xnew_dev, double *m_dev,
                      const double *scale_dev, double *state_dev, double *var_dev, int32_t *status_dev, double *grad_dev,
                      int32_t *enable_dev, double *work_dev, double *proj_work_dev, const double *v_ws_dev, double floor, int slots,
                      int n_iterations, int check_last, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Part 4 -- integer projection batch (bluest/misc.py:228-311 multi, :313-382 single)
 *
 * The reference forms phis = basephi + psi[:, idx] @ ms for up to 2^LL integer candidates and takes
 * pinv(phis)[:,0,0] (misc.py:293-294, 368-369).  Here: base_dev = n_out x (N*N) base information matrices,
 * cols_dev = n_out x LL x (N*N) columns of psi for the LL free groups (zero where an output lacks the group),
 * ms_dev = n_cand x LL candidate values (row-major), V_dev = n_cand x n_out variances (+inf where the candidate's
 * information matrix is singular on its support or does not sample model 0).
 * ---------------------------------------------------------------------------------------------------- */
int bluest_intproj_eval(int N, int n_out, int LL, const double *base_dev, const double *cols_dev, const double *ms_dev,
                        int64_t n_cand, double *V_dev, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Part 5 -- one-shot all-reduce of the Phi records between the GPUs of ONE node
 *
 * NEW (the reference's optimiser runs on one MPI rank, bluest/blue_models.py:508-526).  One process per GPU.  Each rank
 * creates its mailboxes, the ranks swap the 64-byte handles by any means (torch.distributed all_gather in bluest_amd/dist.py),
 * connect, and then every bluest_xchg_allreduce_sum call -- issued by ALL ranks, in the same order -- replaces buf by the sum over
 * ranks in ONE kernel launch per rank: direct peer writes of tagged 8-byte granules into fine-grained memory shared with
 * hipIpc (csrc/xchg.hip).  The sum is formed in rank order, so all ranks hold identical bits.  Asynchronous on `stream`;
 * bluest_xchg_status reports the number of completed calls and whether a wait ever timed out (the result is then NaN).
 * ---------------------------------------------------------------------------------------------------- */
#define BLUEST_XCHG_HANDLE_BYTES 64
typedef struct bluest_xchg_s *bluest_xchg_t;
int bluest_xchg_create(bluest_xchg_t *xchg, int world, int rank, int64_t max_doubles, void *handle_out /* 64 bytes */);
int bluest_xchg_connect(bluest_xchg_t xchg, const void *all_handles /* world * 64 bytes, rank order */);
int bluest_xchg_allreduce_sum(bluest_xchg_t xchg, double *buf_dev, int64_t n_doubles, void *stream);
int bluest_xchg_status(bluest_xchg_t xchg, int64_t *calls, int *timed_out);
int bluest_xchg_destroy(bluest_xchg_t xchg);

/* ------------------------------------------------------------------------------------------------------
 * Part 6 -- second-order finish of solver="spg" (NEW; csrc/newton.hip, restated in numpy in oracle/master_newton.py)
 *
 * The reference hands  min_m max_o V_o(m)/s_o  s.t.  cost.m = B, m >= 0  to third-party NLP solvers together with the Hessian
 * of bluest/misc.py:497-503 / bluest/cmisc.cpp:74-97 (bluest/sap.py:387-456, bluest/mosap.py:578-673).  Here, in the scaled
 * variable x_i = cost_i m_i / B on the unit simplex:
 *   phase 1  bluest_ma_update       multiplicative algorithm on ALL groups (one evaluation + one elementwise kernel per step)
 *   phase 2  bluest_master_newton   Levenberg-Marquardt damped active-set Newton (SQP) on a support of <= 64 groups, one launch of
 *                                   one workgroup per master problem;
 *            bluest_support_point   support vector -> allocation of the full problem (+ uniform background of weight eps);
 *            bluest_price           reduced costs c_i = (B/cost_i) sum_o (mu_o/s_o) v_{o,g_i}^T C_{i,o}^-1 v_{o,g_i} of all groups
 *                                   from the gradient at that point: candidates to enter the support, and max_i c_i, which gives
 *                                   the certified bound  F* >= A^2 / (4 max_i c_i),  A = 2 sum_o (mu_o/s_o) V_o  (weak duality).
 * ---------------------------------------------------------------------------------------------------- */
/* largest support the single-workgroup master can hold for this plan (LDS budget); 0: does not fit */
int bluest_master_max_support(bluest_plan_t plan, int *s_max);
/* support_host: S strictly ascending GLOBAL group indices (host); cc_host: B / cost_j (host, S); s_dev: output scales (n_out);
 * bg_dev: n_out x N x N matrices eps_bg * Phi_o(uniform allocation) (device; may be NULL when eps_bg == 0);
 * x_dev (S): start in, solution out; mu_dev (n_out): multipliers in (any, e.g. 1/n_out) / out;
 * out_dev (16 + n_out doubles): F, lam, kkt residual, spread, iterations, evaluations, linear solves, status (0 converged,
 * 1 stalled, 2 start not evaluable), damping, lam at x, ..., then r_o = V_o/s_o at the solution.  Asynchronous on `stream`
 * after one small synchronous descriptor upload. */
int bluest_master_newton(bluest_plan_t plan, int S, const int64_t *support_host, const double *cc_host, const double *s_dev,
                         const double *bg_dev, double eps_bg, double *x_dev, double *mu_dev, double tol, int maxit,
                         double *out_dev, void *stream);
/* the same master problem with per-model sample caps (max_model_samples, bluest/sap.py:222-240, bluest/mosap.py:326-344):
 * sum over the support groups j containing model cap_model[c] of (1 - eps_bg) cc_j x_j <= cap_b[c]  (ncap <= 64; host arrays;
 * the start x must satisfy them).  Caps at their bound are equality rows of the SQP step, a cap whose multiplier comes out
 * negative leaves, trial points are repaired / pulled back onto the feasible set.  nu_dev (ncap): cap multipliers out. */
int bluest_master_newton_capped(bluest_plan_t plan, int S, const int64_t *support_host, const double *cc_host, const double *s_dev,
                                const double *bg_dev, double eps_bg, double *x_dev, double *mu_dev, double tol, int maxit,
                                double *out_dev, int ncap, const int32_t *cap_model_host, const double *cap_b_host, double *nu_dev,
                                void *stream);
/* x_i <- x_i * cc_i * sum_o wgt_o q_{o,i}/s_o / sum_o wgt_o r_o,  m_i = cc_i x_i;  var/status/grad as bluest_plan_eval left them
 * for the allocation m; wgt_o ~ r_o^(p-1) (p-norm surrogate of the max) */
int bluest_ma_update(bluest_plan_t plan, const double *var_dev, const int32_t *status_dev, const double *grad_dev,
                     const double *s_dev, const double *cc_dev, double p, double *x_dev, double *m_dev, void *stream);
/* single-output plans on all groups (identity mapping): one step of phase 1 in TWO launches -- the evaluation of m_dev, whose fused
 * solve + gradient kernel applies the update above to x_dev / m_dev in its tile wavefronts instead of writing the gradient.  Same
 * iterates as bluest_plan_eval + bluest_ma_update, bit for bit.  BLUEST_ERR_STATE for any other plan. */
int bluest_plan_is_identity(bluest_plan_t plan, int *yes);      /* every output on all groups, local index = global index */
int bluest_plan_eval_ma(bluest_plan_t plan, double *m_dev, double *var_dev, int32_t *status_dev, const double *s_dev,
                        const double *cc_dev, double *x_dev, void *stream);
/* m_i = cc_i ((1 - eps) x_S[i in S] + eps / L); sup_dev ascending */
int bluest_support_point(int64_t L, int S, const int64_t *sup_dev, const double *xs_dev, const double *cc_dev, double eps,
                         double *m_dev, void *stream);
/* after bluest_plan_eval (with gradient) of the priced allocation on the same stream.  c_sup_dev (S): c_i of the support
 * entries; top_val_dev / top_idx_dev (64 * 16): per workgroup the 16 largest (c_i, i); y0_dev (n_out): component 0 of the
 * vectors v_o the quadratic forms were taken with -- the bound's A = 2 sum_o (mu_o/s_o) y0_o */
#define BLUEST_PRICE_CANDIDATES 1024
int bluest_price(bluest_plan_t plan, const double *grad_dev, const double *mu_dev, const double *s_dev, const double *cc_dev,
                 int S, const int64_t *sup_dev, double *c_sup_dev, double *top_val_dev, int64_t *top_idx_dev, double *y0_dev,
                 void *stream);

/* bluest_price with sample caps: capmask_dev (L_global x uint64, bit c: the model of cap c is in the group), nu_dev (cap
 * multipliers of bluest_master_newton_capped), master_out_dev (its result record: F at [0]); the reduced costs become
 * c_i - (B/cost_i) F^2 sum_{c in mask_i} nu_c */
int bluest_price_capped(bluest_plan_t plan, const double *grad_dev, const double *mu_dev, const double *s_dev, const double *cc_dev,
                        int S, const int64_t *sup_dev, double *c_sup_dev, double *top_val_dev, int64_t *top_idx_dev, double *y0_dev,
                        const uint64_t *capmask_dev, const double *nu_dev, const double *master_out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BLUEST_HIP_H */
