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


// ---- solve: the workgroup folds the chunk partials into Phi (LDS), then ONE wavefront eliminates in REGISTERS ----
//
// Ordering: position p of the elimination holds model NT-1-p, so the TARGET model 0 comes LAST and the identity pads
// (models N..NT-1 do not exist) come first -- a STATIC map: no permutation has to be computed, lane p reads its row as
// NT contiguous doubles.  Phi is therefore kept REVERSED in LDS: entry (a, b) lives at [(NT-1-a) * LDP + (NT-1-b)], row
// stride LDP = NT + 2 doubles (an odd number of 16-byte words: the 16-byte row reads of the 64 lanes do not collide on banks).
// With A the restricted matrix and e = e_last: Gauss-Jordan leaves V = e^T A^-1 e = 1/p_last and x = A^-1 e without any
// triangular solve (gj_regs below).  Models that are not sampled keep an identity row / column in place.
template <int NT>   // NT >= N: LDS footprint follows the problem (3.8 KB at NT = 20), not the 64-model maximum
struct SolveLds {
    static constexpr int LDP = NT + 2;
    double phi[NT * LDP];       // reversed symmetric Phi (no delta); rows / columns of the pads are zero
    double amax[NT];            // per model: max |m_i| over groups containing it
    double vout[NT];            // row 0 of pinv(Phi) for the fused gradient pass
    double scratch[NT <= 32 && NT > 16 ? 16 * (NT - 15) : 1];   // back-substitution of the DPP elimination (GjMap::scratch_doubles)
    int status;
    __device__ __forceinline__ double &at(int a, int b) { return phi[(NT - 1 - a) * LDP + (NT - 1 - b)]; }
};

// zero-fill before a fold when pads exist (N < NT); the caller synchronises afterwards.  With N == NT every entry is
// written by the fold itself (every symmetric destination has a row descriptor), so nothing needs clearing.
template <int NT>
__device__ __forceinline__ void clear_pads(SolveLds<NT> &lds, int N, int tid, int nthreads)
{
    if (N < NT)
        for (int t = tid; t < NT * SolveLds<NT>::LDP; t += nthreads) lds.phi[t] = 0.0;
}

__device__ __forceinline__ double readlane_f64(double x, int l)
{   // l must be wave-uniform (here: a compile-time constant after unrolling)
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// Cross-lane reductions without the LDS crossbar.  __shfl_xor compiles to ds_bpermute (two per double, ~100 cycles each, in a
// dependent chain of six for a wave reduction); the master is a latency chain of short reductions, so they are done with DPP
// moves instead: quad permutes, then row_half_mirror / row_mirror inside the 16-lane rows (every lane ends with its row's result),
// then the four rows are combined through v_readlane.  Fixed order, all lanes get the same value.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum(double v)
{   // sum over the lane's 16-lane row, in every lane of the row
    v += dpp_f64<0xB1>(v);      // quad_perm:[1,0,3,2]
    v += dpp_f64<0x4E>(v);      // quad_perm:[2,3,0,1]
    v += dpp_f64<0x141>(v);     // row_half_mirror
    v += dpp_f64<0x140>(v);     // row_mirror
    return v;
}
__device__ __forceinline__ double fast_sum(double v)
{
    v = row_sum(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double fast_max(double v)
{
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
// fold chunk partials of rows [row_begin, row_begin+n_rows) into lds.phi / lds.amax; all threads of the block.
// FOUR adjacent lanes share a row: lane q sums chunks q, q+4, q+8, ... (up to 8 independent loads in flight, so a row of
// <= 32 chunks costs ONE memory round trip), then the quad combines as (s0+s1)+(s2+s3) -- a fixed order, so the result
// is deterministic and identical in every kernel that folds.  nthreads must be a multiple of 4.
// (Staging one output's partials in LDS first -- one coalesced copy, then the fold out of LDS -- was measured and rejected:
//  step 15.7 vs 15.3 us at the headline size; the LDS round trip and the extra barrier cost more than the dependent HBM
//  round trip descriptor -> partials they replace.)
// 0.0 the optimiser cannot see through.  With a literal zero it rewrites `0.0 + (cond ? load : 0.0)` into a sum INSIDE the
// conditional block of the first load, i.e. it waits for that load before it issues the other seven: one extra round trip.
__device__ __forceinline__ double opaque_zero()
{
    double z;
    asm("v_mov_b64 %0, 0" : "=v"(z));
    return z;
}
// REGULAR rows (the usual plans: every model in equally many groups): the partials of an output sit at fixed strides -- first
// the N diagonal destinations with Cd slots each, then the off-diagonal ones (a < b, row by row) with Co slots each; slots a
// row does not use stay zero -- so the fold needs no descriptor: one memory round trip instead of two dependent ones
// (descriptor -> partials), measured 0.47 us of the 1.5 us fold at the headline size.  Cd = 0: the descriptor path.
struct FoldReg { int Cd, Co; const uint16_t *rank_ab; };     // rank_ab[r] = a | b << 8 of the destination with rank r
template <int NT>
__device__ __forceinline__ void fold_rows(SolveLds<NT> &lds, int N, const RowDesc *__restrict__ rows, int row_begin,
                                          int n_rows, const double2 *__restrict__ partial, int tid, int nthreads, FoldReg reg)
{
    const int q = tid & 3;
    if (reg.Cd > 0) {
        const int slots = N * reg.Cd + (n_rows - N) * reg.Co;
        const double2 *po = partial + (int64_t)(row_begin / n_rows) * slots;
        for (int r = tid >> 2; r < n_rows; r += nthreads >> 2) {
            const bool dg = r < N;
            const int n = dg ? reg.Cd : reg.Co;
            const double2 *p = po + (dg ? r * reg.Cd : N * reg.Cd + (r - N) * reg.Co);
            const unsigned ab = reg.rank_ab[r];            // (an independent load, in flight with the partials)
            double s = opaque_zero(), am = opaque_zero();
            for (int c0 = q; c0 < n; c0 += 32) {
                double2 v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = (c0 + 4 * i < n) ? p[c0 + 4 * i] : make_double2(0.0, 0.0);
#pragma unroll
                for (int i = 0; i < 8; i++) { s += v[i].x; am = fmax(am, v[i].y); }
            }
            const int a = (int)(ab & 0xffu), b = (int)(ab >> 8);
            s += __shfl_xor(s, 1);
            am = fmax(am, __shfl_xor(am, 1));
            s += __shfl_xor(s, 2);
            am = fmax(am, __shfl_xor(am, 2));
            if (q == 0) {
                lds.at(a, b) = s;
                lds.at(b, a) = s;
                if (dg) lds.amax[a] = am;
            }
        }
        return;
    }
    for (int r = tid >> 2; r < n_rows; r += nthreads >> 2) {
        const RowDesc rd = rows[row_begin + r];
        const double2 *p = partial + rd.first_chunk;
        const int n = rd.n_chunks;
        double s = opaque_zero(), am = opaque_zero();
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
            lds.at(rd.a, rd.b) = s;
            lds.at(rd.b, rd.a) = s;
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
// diagonals a Cholesky factorisation squares-roots.  "Not positive definite" = some pivot is not larger than
// PIVOT_TOL times the ORIGINAL diagonal entry of its row (a relative test, in the spirit of the rcond of numpy's pinv;
// each lane checks its own pivot when its turn comes, one compare per step).
// The right-hand side is e_last and never stored: it stays e_last until the last step, hence on return
//   x_last = 1/p_last,   x_i = -a[NT-1](lane i, before the last step) / (p_i p_last)   (i != last)
// with p_i = pivot i; rinv_mine = 1/p_lane, last_pivot = p_last.
#define BLUEST_PIVOT_TOL 1.0e-14
template <int NT>
__device__ __forceinline__ void gj_regs(double (&a)[NT], int lane, double diag0, double &rinv_mine, double &last_pivot, int &bad)
{
    const double floor_mine = BLUEST_PIVOT_TOL * diag0;
    bool flag = false;
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const double piv = readlane_f64(a[j], j);
        const bool is = lane == j;
        flag = is ? !(a[j] > floor_mine) : flag;           // NaN and non-positive pivots included
        const double rinv = rcp_f64(piv);
        rinv_mine = is ? rinv : rinv_mine;
        if (j == NT - 1) { last_pivot = piv; break; }
        const double f = is ? 0.0 : -a[j] * rinv;
        // pivot row first (scalar registers), then the updates: the broadcasts do not depend on each other, so issuing
        // them in a block hides the VALU-writes-SGPR -> VALU-reads-it wait states that a readlane/fma ping-pong pays
        double u[NT];
#pragma unroll
        for (int c = j + 1; c < NT; c++) u[c] = readlane_f64(a[c], j);
#pragma unroll
        for (int c = j + 1; c < NT; c++) a[c] = fma(f, u[c], a[c]);
    }
    bad |= (__ballot(flag && lane < NT) != 0ull || !isfinite(last_pivot)) ? 1 : 0;
}

// ---- the same elimination with DPP broadcasts (NT <= 32) ------------------------------------------------------------------
// The Schur complement of a symmetric matrix stays symmetric, so the pivot row's entry a_jc is also entry j of row c: it sits
// in register a[j] of the LANE that holds row c.  `v_fmac_f64_dpp ... row_newbcast:l` takes one operand from lane l of the
// 16-lane DPP row: ONE instruction per column update instead of two v_readlane + FMA.  A DPP row has 16 lanes, so
//   * positions E .. NT-1 (E = max(NT-16, 0)), the MAIN rows, are held by lanes 0 .. 15 (DPP row 0);
//   * positions 0 .. E-1, the EXTRA rows, are pivoted first.  Lanes 16 .. 16+E-1 hold them, but only their leading E x E
//     block A11 is ever used: the rest of an extra row is, by symmetry, column e of the main rows -- registers a[e] of
//     lanes 0..15, which nobody touches after step e.  Step j < E therefore updates the main rows with DPP for the columns
//     c >= E (source a[j] of lane c-E) and with a v_readlane broadcast of the A11 entry for the few columns j < c < E; the
//     extra rows below j take the same v_readlane update (forward elimination only, a pivoted extra row stays frozen);
//   * steps E .. NT-1 are Gauss-Jordan on the main rows with DPP only (row_mask 1: DPP row 0 is written);
//   * the extra unknowns follow by back-substitution: r_e = sum_l a_l[e] x_l over the main rows through a small LDS
//     scratch, then x_e = -(r_e + sum_{e<c<E} u_ec x_c) / p_e down the extra rows with DPP broadcasts inside DPP row 1.
// Hazards (inline assembly is opaque to the compiler's hazard recogniser): a DPP operand must have been written >= 2 wait
// states earlier -- a[j] is written by the first column update of step j-1 and read again only after the reciprocal chain
// of step j; the broadcasts that read a just-written register carry their own s_nop; EXEC does not change inside the
// straight-line code (selects only) and the DPP phases start with s_nop 4.
template <int NT> struct GjMap {
    static constexpr bool dpp = NT <= 32;
    static constexpr int E = (dpp && NT > 16) ? NT - 16 : 0;
    static constexpr int lane_of(int pos) { return !dpp ? pos : (pos < E ? 16 + pos : pos - E); }
    static constexpr int n_lanes = dpp ? (E > 0 ? 16 + E : NT) : NT;      // lanes [0, n_lanes) hold a position
    static constexpr int scratch_doubles = E > 0 ? 16 * (E + 1) : 1;      // LDS scratch of the back-substitution
    // position held by a lane; lanes without one shadow the last position (results unused)
    static __device__ __forceinline__ int pos_of(int lane)
    {
        if (!dpp) return lane < NT ? lane : NT - 1;
        const int p = lane < 16 ? lane + E : lane - 16;
        return (lane < n_lanes && p < NT) ? p : NT - 1;
    }
};

template <int L>
__device__ __forceinline__ void fmac_row0_bcast(double &acc, double src, double mul)
{   // lanes 0..15: acc += (src of lane L) * mul; the other lanes keep acc
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0x1 bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(L));
}
template <int L>
__device__ __forceinline__ void fmac_row1_bcast_nop(double &acc, double src, double mul)
{   // lanes 16..31: acc += (src of lane 16+L) * mul; src may have been written by the previous instruction
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0x2 bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(L));
}
template <int L>
__device__ __forceinline__ double mov_bcast(double src)
{   // every lane: src of lane L of its own 16-lane row
    double out;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(src), "n"(L));
    return out;
}

template <int NT, int J, int C>
struct GjDppCols {     // main rows: column updates C .. NT-1 (C >= E) of step J
    static __device__ __forceinline__ void run(double (&a)[NT], double f)
    {
        if constexpr (C < NT) {
            fmac_row0_bcast<C - GjMap<NT>::E>(a[C], a[J], f);
            GjDppCols<NT, J, C + 1>::run(a, f);
        }
    }
};
template <int NT, int J>
struct GjDppStep {     // steps J .. NT-1 (positions held by lanes 0..15)
    static __device__ __forceinline__ void run(double (&a)[NT], int lane, double &rinv_mine, double &last_pivot)
    {
        constexpr int LJ = J - GjMap<NT>::E;
        const double piv = mov_bcast<LJ>(a[J]);
        const bool is = lane == LJ;
        const double rinv = rcp_f64(piv);
        rinv_mine = is ? rinv : rinv_mine;
        if constexpr (J == NT - 1) { last_pivot = piv; }
        else {
            const double f = is ? 0.0 : -a[J] * rinv;
            GjDppCols<NT, J, J + 1>::run(a, f);
            GjDppStep<NT, J + 1>::run(a, lane, rinv_mine, last_pivot);
        }
    }
};
template <int NT, int J>
struct GjExtraStep {   // steps J .. E-1: pivots of the extra rows (lanes 16 ..)
    static __device__ __forceinline__ void run(double (&a)[NT], int lane, double &rinv_mine)
    {
        constexpr int E = GjMap<NT>::E;
        if constexpr (J < E) {
            const double piv = readlane_f64(a[J], 16 + J);
            const double rinv = rcp_f64(piv);
            rinv_mine = (lane == 16 + J) ? rinv : rinv_mine;
            const double f = (lane >= 16 && lane <= 16 + J) ? 0.0 : -a[J] * rinv;      // pivoted extra rows stay frozen
            double u[E];
#pragma unroll
            for (int c = J + 1; c < E; c++) u[c] = readlane_f64(a[c], 16 + J);
#pragma unroll
            for (int c = J + 1; c < E; c++) a[c] = fma(f, u[c], a[c]);
            GjDppCols<NT, J, E>::run(a, f);
            GjExtraStep<NT, J + 1>::run(a, lane, rinv_mine);
        }
    }
};
template <int NT, int Ei>
struct GjBackSub {     // extra unknowns Ei-1, Ei-2, .., 0 (lanes 16 ..): acc holds r_e plus the terms of the unknowns already known
    static __device__ __forceinline__ void run(const double (&a)[NT], int lane, double rinv_mine, double &acc, double &xe)
    {
        if constexpr (Ei > 0) {
            constexpr int e = Ei - 1;
            xe = (lane == 16 + e) ? -acc * rinv_mine : xe;
            if constexpr (e > 0) fmac_row1_bcast_nop<e>(acc, xe, a[e]);      // rows e' < e: acc += x_e * u_e'e
            GjBackSub<NT, Ei - 1>::run(a, lane, rinv_mine, acc, xe);
        }
    }
};

// Elimination + solution: returns x = (A^-1 e_last) at the position this lane holds (GjMap); last_pivot = 1 / (A^-1)_last,last
// (wave-uniform).  scratch: GjMap<NT>::scratch_doubles doubles of LDS private to this wavefront.  "Not positive definite" as in
// gj_regs, judged from the reciprocals: a pivot p passes when 1/p is positive and (1/p) * floor < 1.
template <int NT>
__device__ __forceinline__ double gj_solve_last(double (&a)[NT], int lane, double diag0, double *scratch, double &last_pivot, int &bad)
{
    using M = GjMap<NT>;
    double rinv_mine = 0.0;
    if constexpr (!M::dpp) {
        gj_regs<NT>(a, lane, diag0, rinv_mine, last_pivot, bad);
        const double rl = readlane_f64(rinv_mine, NT - 1);
        return (lane == NT - 1) ? rl : -a[NT - 1] * rinv_mine * rl;
    } else {
        constexpr int E = M::E;
        asm volatile("s_nop 4" ::: "memory");
        GjExtraStep<NT, 0>::run(a, lane, rinv_mine);
        GjDppStep<NT, E>::run(a, lane, rinv_mine, last_pivot);
        last_pivot = readlane_f64(last_pivot, 0);
        const bool ok = rinv_mine > 0.0 && rinv_mine * (BLUEST_PIVOT_TOL * diag0) < 1.0;     // false for NaN as well
        bad |= (__ballot(!ok && lane < M::n_lanes) != 0ull || !isfinite(last_pivot)) ? 1 : 0;
        const double rl = readlane_f64(rinv_mine, M::lane_of(NT - 1));
        double x = (lane == M::lane_of(NT - 1)) ? rl : -a[NT - 1] * rinv_mine * rl;
        if constexpr (E > 0) {
            // r_e = sum over the main rows l of a_l[e] x_l: lane l leaves (a_l[0..E), x_l) in the scratch, lane 16+e sums column e
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < E; e++) scratch[lane * (E + 1) + e] = a[e];
                scratch[lane * (E + 1) + E] = x;
            }
            wave_lds_sync();
            const int e_mine = (lane >= 16 && lane < 16 + E) ? lane - 16 : 0;
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < 16; l++) acc = fma(scratch[l * (E + 1) + e_mine], scratch[l * (E + 1) + E], acc);
            wave_lds_sync();
            double xe = 0.0;
            asm volatile("s_nop 4" ::: "memory");
            GjBackSub<NT, E>::run(a, lane, rinv_mine, acc, xe);
            x = (lane >= 16) ? xe : x;
        }
        return x;
    }
}

// One (candidate, output): masks -> identity-padded restricted matrix in registers (static order) -> Gauss-Jordan -> V (-> v).
// Called by ONE wavefront (lane = 0..63); lds.phi is ready.  s1 / s2 (lane = model): model touched by a group with
// |m| > 1e-6 (misc.py:453-457, the rows of V's restricted system) / by a group with m != 0 (support of Phi, for v).
template <int NT>
__device__ __forceinline__ void solve_wave(SolveLds<NT> &lds, int N, double delta, bool s1, bool s2, bool big_in, bool want_v,
                                        double *__restrict__ var_out, double *__restrict__ v_out,
                                        int32_t *__restrict__ status_out, int lane)
{
    constexpr int LDP = SolveLds<NT>::LDP;
    const unsigned long long mask1 = __ballot(lane < N && s1);
    const unsigned long long mask2 = (delta != 0.0) ? __ballot(lane < N) : __ballot(lane < N && s2);
    const unsigned long long all = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
    const bool big = uniform_i(big_in ? 1 : 0) != 0;
    int status = BLUEST_EVAL_OK;
    double V = 0.0;
    // v defaults: zero everywhere (NaN after a singular elimination); filled per model below
    double vmine = 0.0;                     // value of v for the model this lane's POSITION holds
    unsigned long long vmask = 0ull;        // models for which vmine is meaningful
    int vswap = 0;
    const int p = GjMap<NT>::pos_of(lane);            // position of the elimination this lane holds (lanes without one shadow the last)
    if (__builtin_expect(!big, 0)) {
        status = BLUEST_EVAL_INF;
        V = INFINITY;
    } else if (__builtin_expect(mask1 == 0ull, 0)) {
        status = BLUEST_EVAL_NO_MODEL0;
        V = NAN;
    } else {
        if (__builtin_expect(!(mask1 & 1ull), 0)) status = BLUEST_EVAL_NO_MODEL0;
        const int npass = (mask1 == mask2 || !want_v) ? 1 : 2;
        for (int pass = 0; pass < npass; pass++) {
            unsigned long long mask = (pass == 0) ? mask1 : mask2;
            if (__builtin_expect(pass == 1 && !(mask & 1ull), 0)) break;                     // row 0 of pinv(Phi) is zero
            // target = smallest sampled model (model 0 unless it is unsampled: pinv(PHI[idx])[0,0] of misc.py:490 then
            // picks the first row of the restricted matrix); it must sit at the last position: swap models 0 <-> t
            const int t = __ffsll((long long)mask) - 1;
            int mp = NT - 1 - p;                                        // ORIGINAL model whose row this position holds
            if (__builtin_expect(t != 0, 0)) {
                mp = (mp == 0) ? t : (mp == t ? 0 : mp);
                mask = (mask | 1ull) & ~(1ull << t);                    // membership by POSITION index: bits 0 and t exchanged
            }
            const bool mine = (NT - 1 - p) < N && ((mask >> (NT - 1 - p)) & 1ull);
            const double *row = lds.phi + (NT - 1 - mp) * LDP;
            double a[NT];
            PHASE(9);
#pragma unroll
            for (int c = 0; c < NT; c += 2) {
                const double2 x = *reinterpret_cast<const double2 *>(row + c);
                a[c] = x.x; a[c + 1] = x.y;
            }
            if (__builtin_expect(t != 0, 0)) {   // column swap: position NT-1 <-> position NT-1-t
#pragma unroll
                for (int c = 0; c < NT - 1; c++)
                    if (c == NT - 1 - t) { const double tmp = a[c]; a[c] = a[NT - 1]; a[NT - 1] = tmp; }
            }
            const unsigned long long smask = mask;                      // bit i: model at position NT-1-i is in the system
            if (__builtin_expect(smask != all, 0)) {
#pragma unroll
                for (int c = 0; c < NT; c++) {
                    const bool colin = (NT - 1 - c) < N && ((smask >> (NT - 1 - c)) & 1ull);
                    a[c] = (mine && colin) ? a[c] : 0.0;
                }
            }
            const double diag = mine ? delta : 1.0;                     // pads / unsampled models: identity row
            double diag0 = 1.0;                                         // my original diagonal entry (lanes >= NT: unused)
#pragma unroll
            for (int c = 0; c < NT; c++) {
                a[c] = (c == p) ? a[c] + diag : a[c];
                diag0 = (c == p) ? a[c] : diag0;
            }
            double last_pivot = 1.0;
            int bad = 0;
            PHASE(5);
            const double x = gj_solve_last<NT>(a, lane, diag0, lds.scratch, last_pivot, bad);      // x = A^-1 e_last at position p
            PHASE(6);
            if (__builtin_expect(uniform_i(bad) != 0, 0)) {
                if (status == BLUEST_EVAL_OK) status = BLUEST_EVAL_SINGULAR;
                if (pass == 0) V = NAN;
                vmine = NAN; vmask = all; vswap = 0;
                continue;
            }
            if (pass == 0) V = 1.0 / last_pivot;     // = (A^-1)_{target,target}
            if (want_v && pass == npass - 1) {
                // row 0 of pinv(Phi) is zero when model 0 is not in the support; otherwise x on the support, 0 elsewhere
                const bool have = (mask2 & 1ull) != 0ull || delta != 0.0;
                vmine = (have && mine) ? x : 0.0;
                vmask = all;
                vswap = t;
            }
        }
    }
    PHASE(7);
    if (want_v) {   // position p holds model NT-1-p (0 <-> vswap exchanged)
        int mp = NT - 1 - p;
        if (vswap != 0) mp = (mp == 0) ? vswap : (mp == vswap ? 0 : mp);
        if (lane < GjMap<NT>::n_lanes && mp < N) v_out[mp] = vmask ? vmine : 0.0;
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
