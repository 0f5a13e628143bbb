// spg_state.hpp -- layout of the device-resident SPG state and the line-search decision, shared by spg.hip (stand-alone decision
// kernel) and plan.hip (decision fused into the tail of the solve kernel: one launch less per line-search slot).
#pragma once
#include "common.hpp"

// ---- device-resident SPG state (doubles in HBM; layout mirrored in bluest_amd/spg_device.py) -----------------------
#define SPG_F        0    // objective at x (normalised)
#define SPG_FNEW     1    // objective at the accepted trial point
#define SPG_LAMBDA   2    // spectral step
#define SPG_ALPHA    3    // line-search step of the NEXT trial
#define SPG_GD       4    // g.d            } written by the direction kernel
#define SPG_DMAX     5    // max|d|         }
#define SPG_TAU      6    //                }
#define SPG_NPOS     7    //                }
#define SPG_ACCEPT   8    // 1 once a trial of this iteration satisfied the nonmonotone Armijo test
#define SPG_FAIL     9    // 1 = the line search failed (step length underflow / evaluation budget spent): every kernel is a no-op
#define SPG_DONE     10   // 1 = every kernel is a no-op
#define SPG_IT       11
#define SPG_COUNT    12   // objective evaluations
#define SPG_NORM     13   // objective normalisation
#define SPG_P        14   // smoothing exponent (inf = plain max)
#define SPG_LMIN     15
#define SPG_LMAX     16
#define SPG_HLEN     17   // history length (<= 16)
#define SPG_SDOTS    18
#define SPG_SDOTY    19
#define SPG_FTRIAL   20   // objective of the last evaluated trial
#define SPG_EPS      21   // stop when max|P(x-g)-x| <= eps
#define SPG_PENDING  22   // 1 = the line search of the current iteration goes on: the NEXT direction launch forms the next trial
                          //     point (alpha from the state) instead of a new direction, the finishing launches stay gated off
#define SPG_MAXFEV   23   // evaluation budget (0 = none): a line search that exhausts it fails
#define SPG_GPSTATS  24   // g.gp, max|gp| (= gpmax), tau, npos of the convergence projection
#define SPG_TICKET   28   // (32-bit counter in this slot) arrival ticket of the update launch's workgroups
#define SPG_GDPARTS  29   // (pointer bits) per-workgroup partials of g.d left by the multi-workgroup direction kernel, stride 4
#define SPG_GDPARTS_N 30  // how many (0: SPG_GD holds the folded value)
#define SPG_THETA    31   // warm start of the single-workgroup direction search: (1 - tau_abs)/lambda of the previous one
#define SPG_HIST     32   // 16 slots
#define SPG_COEF     64   // dF/dV_o of the accepted trial (n_out <= 64)
#define SPG_S        128  // normalisers s_o (1 or eps_o^2)
#define SPG_STATE_DOUBLES 256
#define SPG_MAX_OUT  64


__device__ __forceinline__ void spg_wave_lds_sync()
{   // one wavefront: its LDS operations execute in order; keep the compiler from moving them and drain the counter
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// objective of the trial from the per-output variances, nonmonotone Armijo test, safeguarded quadratic interpolation
// (bluest/spg.py:9-35).  ONE wavefront (lane = 0..63): the state is staged through `ls` (SPG_STATE_DOUBLES doubles of LDS) with
// coalesced loads, lane o handles output o, lane 0 takes the decision.  On the last slot of an iteration it also sets the
// gate of the finishing launches; a trial rejected there leaves the line search PENDING (continued by the next direction launch,
// so a window of the solver is a sequence of identical "steps" with no host round trip) unless the step length underflowed or
// the evaluation budget is spent (FAIL).
// What the decision needs that no workgroup of the SAME launch changes: the state (written by earlier launches) and the
// per-workgroup partials of g.d (written by the direction launch).  A kernel that ends with the decision issues these loads
// early -- spg_prefetch_state before its own work, spg_prefetch_parts once the state has arrived -- so that the last arriver's
// serial tail is ticket -> other outputs' values -> arithmetic, without two more dependent round trips to memory.
struct SpgPrefetch {
    double w[SPG_STATE_DOUBLES / 64];   // state words t*64 + lane
    double gdpart;                      // this lane's partial of g.d (0 beyond the count)
    int n_parts;
};
__device__ __forceinline__ void spg_prefetch_state(const double *st, int lane, SpgPrefetch &pf)
{
#pragma unroll
    for (int t = 0; t < SPG_STATE_DOUBLES / 64; t++) pf.w[t] = st[t * 64 + lane];
    pf.gdpart = 0.0;
    pf.n_parts = 0;
}
__device__ __forceinline__ void spg_prefetch_parts(SpgPrefetch &pf, int lane)
{
    const double ptr_bits = __shfl(pf.w[0], SPG_GDPARTS);
    pf.n_parts = (int)__shfl(pf.w[0], SPG_GDPARTS_N);
    if (pf.n_parts > 0) {
        const double *parts = reinterpret_cast<const double *>((uintptr_t)__double_as_longlong(ptr_bits));
        pf.gdpart = lane < pf.n_parts ? parts[4 * lane + 1] : 0.0;
    }
}

__device__ __forceinline__ void spg_decide_wave(double *__restrict__ st, const double *var, const int32_t *status, int n_out,
                                                int last_slot, int32_t *__restrict__ enable, double *ls, int lane,
                                                const SpgPrefetch &pf, bool own = false, double own_V = 0.0, int32_t own_status = 0)
{   // own: single-output plan, lane 0 passes V and status of its own solve in registers (nothing is read back from memory)
#pragma unroll
    for (int t = 0; t < SPG_STATE_DOUBLES / 64; t++) ls[t * 64 + lane] = pf.w[t];
    spg_wave_lds_sync();
    const bool idle = ls[SPG_DONE] != 0.0 || ls[SPG_FAIL] != 0.0;
    if (idle || ls[SPG_ACCEPT] != 0.0) {
        if (last_slot && lane == 0) *enable = (!idle && ls[SPG_ACCEPT] != 0.0) ? 1 : 0;
        return;
    }
    // objective F = || (V_o/s_o) ||_p / norm, coefficients dF/dV_o
    const bool mine = lane < n_out;
    const double so = mine ? ls[SPG_S + lane] : 1.0;
    // agent-scope loads: in the fused form these values were written a moment ago by OTHER workgroups of the same launch
    double vo = 0.0;
    int32_t so_status = BLUEST_EVAL_OK;
    if (own) {
        vo = mine ? own_V : 0.0;
        so_status = mine ? own_status : BLUEST_EVAL_OK;
    } else if (mine) {
        vo = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(var) + lane, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT));
        so_status = __hip_atomic_load(status + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double r = mine ? vo / so : 0.0;
    const bool bad = mine && (so_status != BLUEST_EVAL_OK || !isfinite(r));
    const bool ok = __ballot(bad) == 0ull;
    const double rmax = wave_max(mine ? r : -INFINITY);
    const double p = ls[SPG_P], norm = ls[SPG_NORM];
    double F = INFINITY, coef = 0.0;
    if (ok) {
        if (isinf(p) || n_out == 1) {
            const unsigned long long is_max = __ballot(mine && r == rmax);
            const int omax = __ffsll((long long)is_max) - 1;
            F = rmax;
            coef = (lane == omax) ? 1.0 / so : 0.0;
        } else {
            const double q = mine ? r / rmax : 0.0;
            const double tq = mine ? pow(q, p - 1.0) : 0.0;       // q^(p-1); q^p = tq*q
            const double tsum = wave_sum(tq * q);
            const double root = pow(tsum, 1.0 / p);
            F = rmax * root;
            coef = tq * (root / tsum) / so;
        }
        F /= norm;
    }
    const int H = (int)ls[SPG_HLEN];
    double fmax = -INFINITY;
    for (int h = 0; h < H; h++) fmax = fmax > ls[SPG_HIST + h] ? fmax : ls[SPG_HIST + h];
    double gd = ls[SPG_GD];
    if (pf.n_parts > 0) gd = wave_sum(pf.gdpart);   // the direction kernel left per-workgroup partials (<= 64): fixed-order fold
    const double alpha = ls[SPG_ALPHA], f = ls[SPG_F];
    const bool accept = F <= fmax + 1.0e-4 * alpha * gd;
    if (accept && mine) st[SPG_COEF + lane] = coef / norm;
    if (lane == 0) {
        st[SPG_COUNT] = ls[SPG_COUNT] + 1.0;
        st[SPG_FTRIAL] = F;
        if (accept) {
            st[SPG_ACCEPT] = 1.0;
            st[SPG_FNEW] = F;
        } else {
            double a = alpha;
            if (a <= 0.1) {
                a *= 0.5;
            } else {
                double at = -0.5 * (a * a) * gd / (F - f - a * gd);
                if (!(at >= 0.1) || at > 0.9 * a) at = 0.5 * a;   // also catches F = inf (at = -0) and NaN
                a = at;
            }
            st[SPG_ALPHA] = a;
            if (last_slot) {
                const double maxfev = ls[SPG_MAXFEV];
                const bool dead = !(a >= 1.0e-300) || (maxfev > 0.0 && ls[SPG_COUNT] + 1.0 >= maxfev);
                st[dead ? SPG_FAIL : SPG_PENDING] = 1.0;
            }
        }
        if (accept) st[SPG_PENDING] = 0.0;
        if (last_slot) *enable = accept ? 1 : 0;
    }
}

// the decision without an early prefetch (stand-alone kernel, predicated-off launches)
__device__ __forceinline__ void spg_decide_wave(double *__restrict__ st, const double *var, const int32_t *status, int n_out,
                                                int last_slot, int32_t *__restrict__ enable, double *ls, int lane)
{
    SpgPrefetch pf;
    spg_prefetch_state(st, lane, pf);
    spg_prefetch_parts(pf, lane);
    spg_decide_wave(st, var, status, n_out, last_slot, enable, ls, lane, pf);
}
