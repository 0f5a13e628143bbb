// solve.hpp -- device code shared by the plan kernels and the integer-projection kernel: fold of the chunk partials into
// Phi (LDS) and the register-resident Gauss-Jordan solve of one (candidate, output) by one wavefront.
#pragma once
#include "common.hpp"

#ifndef PHASE          // in-kernel timestamps exist only in the experiment build of plan.hip
#define PHASE(i)
#endif

struct RowDesc {      // one symmetric destination (a <= b) of one output
    int32_t first_chunk;
    int32_t n_chunks;
    int16_t out, a, b, pad;
};


// ---- solve: one workgroup of 256 threads folds the chunk partials, then wavefront 0 factorises in REGISTERS ----
//
// Ordering trick: the restricted system is permuted so that the TARGET model (model 0 if it is sampled, else the
// smallest sampled model, as pinv(PHI[idx])[0,0] of misc.py:490 would pick) comes LAST.  With A = L L^T and
// e = e_last:  L y = e  =>  y = e_last / L_nn, so V = e^T A^-1 e = 1/L_nn^2 needs NO triangular solve, and
// x = A^-1 e needs only the backward one.  Lane i holds row i of A / L in registers (static indices after full
// unrolling); column broadcasts are v_readlane (SGPR operands), no LDS and no barriers inside the factorisation.
template <int NT>   // NT >= N: LDS footprint follows the problem (6.7 KB at NT = 20), not the 64-model maximum
struct SolveLds {
    static constexpr int LDA = NT + 1;
    double phi[NT * NT];        // full symmetric Phi (no delta), row stride N
    double lt[NT * (NT + 1)];   // L, for the transposed read of the backward solve
    double amax[NT];            // per model: max |m_i| over groups containing it
    double vout[NT];            // row 0 of pinv(Phi) for the fused gradient pass
    int model_of_pos[NT];
    int status;
};

__device__ __forceinline__ double readlane_f64(double x, int l)
{   // l must be wave-uniform (here: a compile-time constant after unrolling)
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// fold chunk partials of rows [row_begin, row_begin+n_rows) into lds.phi / lds.amax; all threads of the block.
// FOUR adjacent lanes share a row: lane q sums chunks q, q+4, q+8, ... (up to 8 independent loads in flight, so a row of
// <= 32 chunks costs ONE memory round trip), then the quad combines as (s0+s1)+(s2+s3) -- a fixed order, so the result
// is deterministic and identical in every kernel that folds.  nthreads must be a multiple of 4.
template <int NT>
__device__ __forceinline__ void fold_rows(SolveLds<NT> &lds, int N, const RowDesc *__restrict__ rows, int row_begin,
                                          int n_rows, const double2 *__restrict__ partial, int tid, int nthreads)
{
    const int q = tid & 3;
    for (int r = tid >> 2; r < n_rows; r += nthreads >> 2) {
        const RowDesc rd = rows[row_begin + r];
        const double2 *p = partial + rd.first_chunk;
        const int n = rd.n_chunks;
        double s = 0.0, am = 0.0;
        for (int c0 = q; c0 < n; c0 += 32) {
            double2 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (c0 + 4 * i < n) ? p[c0 + 4 * i] : make_double2(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < 8; i++) { s += v[i].x; am = fmax(am, v[i].y); }
        }
        s += __shfl_xor(s, 1);
        am = fmax(am, __shfl_xor(am, 1));
        s += __shfl_xor(s, 2);
        am = fmax(am, __shfl_xor(am, 2));
        if (q == 0) {
            lds.phi[rd.a * N + rd.b] = s;
            lds.phi[rd.b * N + rd.a] = s;
            if (rd.a == rd.b) lds.amax[rd.a] = am;
        }
    }
}

// single-wavefront LDS ordering: LDS operations of one wave execute in order; this only stops the compiler from
// moving LDS accesses across it and drains lgkmcnt
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int uniform_i(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ double rcp_f64(double x)
{   // v_rcp_f64 seed + two Newton steps (y <- y + y*(1 - x y)), the refinement the compiler's own f64 division uses
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// Gauss-Jordan elimination (no pivoting: the matrix is symmetric positive definite) of an NT x NT matrix whose row
// `lane` sits in a[0..NT), straight-line code: no predicates, no LDS, no barriers.  Step j subtracts multiples of row j
// from ALL other rows (the lanes above the diagonal are there anyway, so eliminating upwards is free and replaces the
// backward substitution of a Cholesky solve).  Only columns c > j are touched.  The pivots are the same Schur-complement
// diagonals a Cholesky factorisation squares-roots, so "not positive definite" is detected identically.
// The right-hand side is e_last and never stored: it stays e_last until the last step, hence on return
//   x_last = 1/p_last,   x_i = -a[NT-1](lane i, before the last step) / (p_i p_last)   (i != last)
// with p_i = pivot i; rinv_mine = 1/p_lane, last_pivot = p_last.
template <int NT>
__device__ __forceinline__ void gj_regs(double (&a)[NT], int lane, double &rinv_mine, double &last_pivot, int &bad)
{
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const double piv = readlane_f64(a[j], j);
        bad |= (!(piv > 0.0) || !isfinite(piv)) ? 1 : 0;
        const double rinv = rcp_f64(piv);
        rinv_mine = (lane == j) ? rinv : rinv_mine;
        if (j == NT - 1) { last_pivot = piv; break; }
        const double f = (lane == j) ? 0.0 : -a[j] * rinv;
        // pivot row first (scalar registers), then the updates: the broadcasts do not depend on each other, so issuing
        // them in a block hides the VALU-writes-SGPR -> VALU-reads-it wait states that a readlane/fma ping-pong pays
        double u[NT];
#pragma unroll
        for (int c = j + 1; c < NT; c++) u[c] = readlane_f64(a[c], j);
#pragma unroll
        for (int c = j + 1; c < NT; c++) a[c] = fma(f, u[c], a[c]);
    }
}

// One (candidate, output): masks -> permuted, identity-padded restricted matrix in registers -> Cholesky -> V (-> v).
// Called by ONE wavefront (lane = 0..63); lds.phi is ready.  Order of the NT positions:
//   [ NT-nr identity pads | sampled models except the target, ascending | target ]
// so the target always sits at the static position NT-1.
template <int NT>
__device__ __forceinline__ void solve_wave(SolveLds<NT> &lds, int N, double delta, bool s1, bool s2, bool big_in, bool want_v,
                                        double *__restrict__ var_out, double *__restrict__ v_out,
                                        int32_t *__restrict__ status_out, int lane)
{
    // mask1: models touched by a group with |m| > 1e-6 (misc.py:453-457) -> V; mask2: support of Phi+delta*I -> v
    const unsigned long long mask1 = __ballot(lane < N && s1);
    const unsigned long long mask2 = (delta != 0.0) ? __ballot(lane < N) : __ballot(lane < N && s2);
    const bool big = uniform_i(big_in ? 1 : 0) != 0;
    int status = BLUEST_EVAL_OK;
    double V = 0.0, vfill = 0.0, xpos = 0.0;
    int xrow = -1;
    unsigned long long xmask = 0ull;
    bool have_x = false;
    if (!big) {
        status = BLUEST_EVAL_INF;
        V = INFINITY;
    } else if (mask1 == 0ull) {
        status = BLUEST_EVAL_NO_MODEL0;
        V = NAN;
    } else {
        if (!(mask1 & 1ull)) status = BLUEST_EVAL_NO_MODEL0;
        const int npass = (mask1 == mask2 || !want_v) ? 1 : 2;
        for (int pass = 0; pass < npass; pass++) {
            const unsigned long long mask = (pass == 0) ? mask1 : mask2;
            if (pass == 1 && !(mask & 1ull)) break;                     // row 0 of pinv(Phi) is zero
            const int nr = __popcll(mask);
            const int npad = NT - nr;
            const int target = __ffsll((long long)mask) - 1;           // smallest sampled model, ordered LAST
            // model at each position = inverse of "position of each model": every lane sends (its model + 1) to its
            // position with one ds_permute (a bijection of the 64 lanes: sampled models -> their positions, everything else ->
            // the pad / unused positions, carrying 0), then the columns' models are lane broadcasts of the result
            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            const bool mine = lane < N && ((mask >> lane) & 1ull);
            const int jfree = lane - below;                            // rank among the lanes that are not sampled models
            const int dest = mine ? ((lane == target) ? NT - 1 : npad + below - 1) : (jfree < npad ? jfree : NT + (jfree - npad));
            const int rowm = __builtin_amdgcn_ds_permute(dest << 2, mine ? lane + 1 : 0) - 1;
            int colm[NT];
#pragma unroll
            for (int c = 0; c < NT; c++) colm[c] = __builtin_amdgcn_readlane(rowm, c);
            double a[NT];
            PHASE(9);
#pragma unroll
            for (int c = 0; c < NT; c++) {
                const bool real = rowm >= 0 && colm[c] >= 0;
                const double x = lds.phi[real ? rowm * N + colm[c] : 0];
                const double diag = (c == lane) ? 1.0 : 0.0;
                a[c] = real ? ((c == lane) ? x + delta : x) : diag;    // pads: identity
            }
            double last_pivot = 1.0, rinv_mine = 0.0;
            int bad = 0;
            PHASE(5);
            gj_regs<NT>(a, lane, rinv_mine, last_pivot, bad);
            PHASE(6);
            if (uniform_i(bad)) {
                if (status == BLUEST_EVAL_OK) status = BLUEST_EVAL_SINGULAR;
                if (pass == 0) V = NAN;
                vfill = NAN;
                have_x = false;
                continue;
            }
            if (pass == 0) V = 1.0 / last_pivot;     // = (A^-1)_{target,target}
            if (want_v && pass == npass - 1) {
                const double rl = readlane_f64(rinv_mine, NT - 1);
                xpos = (lane == NT - 1) ? rl : -a[NT - 1] * rinv_mine * rl;   // x = A^-1 e_last at position `lane`
                xrow = rowm;
                xmask = mask;
                vfill = 0.0;
                have_x = (mask & 1ull) != 0ull;      // row 0 of pinv(Phi) is zero when model 0 is not in the support
            }
        }
    }
    PHASE(7);
    if (want_v) {   // v in model order: the support scattered from position order, zero (NaN if singular) elsewhere
        if (lane < N && !(have_x && ((xmask >> lane) & 1ull))) v_out[lane] = vfill;
        if (have_x && lane < NT && xrow >= 0) v_out[xrow] = xpos;
        wave_lds_sync();
    }
    if (lane == 0) { *var_out = V; *status_out = status; }
}

// threads of the fold + solve workgroups: 1024 (one quad per row for up to 256 rows per pass) while the register-resident
// matrix of the solving wavefront fits the 128-VGPR budget that comes with it, else 256
__host__ __device__ constexpr int fold_threads(int NT) { return NT <= 26 ? 1024 : 256; }

// host-side choice of the register-array size NT >= N
#define NT_DISPATCH(N, LAUNCH)                                                                                \
    do {                                                                                                     \
        if (N <= 8) { LAUNCH(8); }                                                                           \
        else if (N <= 12) { LAUNCH(12); }                                                                    \
        else if (N <= 16) { LAUNCH(16); }                                                                    \
        else if (N <= 20) { LAUNCH(20); }                                                                    \
        else if (N <= 26) { LAUNCH(26); }                                                                    \
        else if (N <= 32) { LAUNCH(32); }                                                                    \
        else if (N <= 48) { LAUNCH(48); }                                                                    \
        else { LAUNCH(64); }                                                                                 \
    } while (0)

static inline int pick_nt(int N) { return N <= 8 ? 8 : N <= 12 ? 12 : N <= 16 ? 16 : N <= 20 ? 20 : N <= 26 ? 26 : N <= 32 ? 32 : N <= 48 ? 48 : 64; }
