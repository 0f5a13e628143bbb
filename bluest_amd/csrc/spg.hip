// spg.hip -- Part 3 of include/bluest_hip.h: the simplex projection (one workgroup / single launch of many workgroups with
// tagged mailboxes / multi-launch fallback) and the kernels of the device-resident SPG iteration (state in HBM, control flow by
// predication).  Algorithm: bluest/spg.py:3-132 with the projection onto the simplex.
#include "plan.hpp"
#include "spg_state.hpp"

// ------------------------------------------------------------------------------------------------------
// Part 3 -- simplex projection (single workgroup of 1024 threads)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool proj_idle(const double *spg_state) { return spg_state && (spg_state[SPG_DONE] != 0.0 || spg_state[SPG_FAIL] != 0.0); }
// direction launches (spg_mode 1) while a line search is pending: no new direction, the next trial point instead
__device__ __forceinline__ bool proj_pending(const double *spg_state, int spg_mode) { return spg_state && spg_mode == 1 && spg_state[SPG_PENDING] != 0.0; }

// the next trial point of a pending line search, element i: xnew = x + alpha*d, m = scale*xnew (bluest/spg.py:13,28)
__device__ __forceinline__ void spg_trial_point(int64_t i, double alpha, const double *__restrict__ x, const double *d,
                                                const double *__restrict__ scale, double *__restrict__ xnew, double *__restrict__ m)
{
    const double xn = fma(alpha, d[i], x[i]);
    xnew[i] = xn;
    m[i] = scale[i] * xn;
}

// xnew = x + alpha*d, m = scale*xnew for the next line-search slot; sets the plan gate (bluest/spg.py:13,28)
__global__ __launch_bounds__(1024) void k_spg_trial(const double *__restrict__ x, const double *__restrict__ d,
                                                    const double *__restrict__ scale, const double *__restrict__ st,
                                                    double *__restrict__ xnew, double *__restrict__ m,
                                                    int32_t *__restrict__ enable, int64_t L)
{
    const bool run = st[SPG_DONE] == 0.0 && st[SPG_FAIL] == 0.0 && st[SPG_ACCEPT] == 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *enable = run ? 1 : 0;
    if (!run) return;
    const double alpha = st[SPG_ALPHA];
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < L) {
        const double xn = fma(alpha, d[i], x[i]);
        xnew[i] = xn;
        m[i] = scale[i] * xn;
    }
}

// objective of the trial from the per-output variances, nonmonotone Armijo test, safeguarded quadratic interpolation
// (bluest/spg.py:9-35).  One wavefront: the state is staged through LDS with coalesced loads, lane o handles output o,
// lane 0 takes the decision.  On the last slot of an iteration it also sets the gate of the finishing launches.
__global__ __launch_bounds__(64) void k_spg_decide(double *__restrict__ st, const double *__restrict__ var,
                                                   const int32_t *__restrict__ status, int n_out, int last_slot,
                                                   int32_t *__restrict__ enable)
{
    __shared__ double ls[SPG_STATE_DOUBLES];
    spg_decide_wave(st, var, status, n_out, last_slot, enable, ls, threadIdx.x);
}

// Workgroup reductions for the projection / SPG kernels: wavefront butterflies, one LDS hand-off and ONE barrier per call.
// The LDS slots are double-buffered by the caller-held phase bit, so a wavefront that is already in the next reduction
// cannot overwrite values a slower wavefront is still reading (it would first have to pass the barrier in between).
// Every wavefront folds the <= 16 per-wavefront partials itself with the same butterfly: identical results everywhere.
struct ProjLds {
    double d0[2][16];
    double d1[2][16];
    long long c[2][16];
    double d2[2][16];
    double d3[2][16];
    double xf[8];           // results of the cross-workgroup fold of a Newton pass (k_proj_fused)
};

__device__ __forceinline__ double block_max(double x, ProjLds &s, int tid, int &ph)
{
    x = wave_max(x);
    const int lane = tid & 63, nw = blockDim.x >> 6;
    if (lane == 0) s.d0[ph][tid >> 6] = x;
    __syncthreads();
    double r = (lane < nw) ? s.d0[ph][lane] : -INFINITY;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) r = fmax(r, __shfl_xor(r, off, WAVE));
    ph ^= 1;
    return __shfl(r, 0, WAVE);
}
__device__ __forceinline__ void block_sum_cnt(double &x, long long &n, ProjLds &s, int tid, int &ph)
{
    x = wave_sum(x);
    n = wave_sum_ll(n);
    const int lane = tid & 63, nw = blockDim.x >> 6;
    if (lane == 0) { s.d0[ph][tid >> 6] = x; s.c[ph][tid >> 6] = n; }
    __syncthreads();
    double r = (lane < nw) ? s.d0[ph][lane] : 0.0;
    long long c = (lane < nw) ? s.c[ph][lane] : 0;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { r += __shfl_xor(r, off, WAVE); c += __shfl_xor(c, off, WAVE); }
    ph ^= 1;
    x = __shfl(r, 0, WAVE);
    n = __shfl(c, 0, WAVE);
}
__device__ __forceinline__ void block_sum2_cnt(double &x, double &y, long long &n, ProjLds &s, int tid, int &ph)
{
    x = wave_sum(x);
    y = wave_sum(y);
    n = wave_sum_ll(n);
    const int lane = tid & 63, nw = blockDim.x >> 6;
    if (lane == 0) { s.d0[ph][tid >> 6] = x; s.d1[ph][tid >> 6] = y; s.c[ph][tid >> 6] = n; }
    __syncthreads();
    double r = (lane < nw) ? s.d0[ph][lane] : 0.0;
    double q = (lane < nw) ? s.d1[ph][lane] : 0.0;
    long long c = (lane < nw) ? s.c[ph][lane] : 0;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { r += __shfl_xor(r, off, WAVE); q += __shfl_xor(q, off, WAVE); c += __shfl_xor(c, off, WAVE); }
    ph ^= 1;
    x = __shfl(r, 0, WAVE);
    y = __shfl(q, 0, WAVE);
    n = __shfl(c, 0, WAVE);
}

// sums of x and y together with max of lo and min of hi: ONE barrier for the four values of a Newton pass of the projection
__device__ __forceinline__ void block_pass4(double &x, double &y, double &lo, double &hi, ProjLds &s, int tid, int &ph)
{
    x = wave_sum(x);
    y = wave_sum(y);
    lo = wave_max(lo);
    hi = -wave_max(-hi);
    const int lane = tid & 63, nw = blockDim.x >> 6;
    if (lane == 0) { s.d0[ph][tid >> 6] = x; s.d1[ph][tid >> 6] = y; s.d2[ph][tid >> 6] = lo; s.d3[ph][tid >> 6] = hi; }
    __syncthreads();
    double r = (lane < nw) ? s.d0[ph][lane] : 0.0;
    double q = (lane < nw) ? s.d1[ph][lane] : 0.0;
    double l = (lane < nw) ? s.d2[ph][lane] : -INFINITY;
    double h = (lane < nw) ? s.d3[ph][lane] : INFINITY;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        r += __shfl_xor(r, off, WAVE); q += __shfl_xor(q, off, WAVE);
        l = fmax(l, __shfl_xor(l, off, WAVE)); h = fmin(h, __shfl_xor(h, off, WAVE));
    }
    ph ^= 1;
    x = __shfl(r, 0, WAVE);
    y = __shfl(q, 0, WAVE);
    lo = __shfl(l, 0, WAVE);
    hi = __shfl(h, 0, WAVE);
}

// accept the step (bluest/spg.py:85-106), two launches:
// Barzilai-Borwein bookkeeping after an accepted step, one wavefront: fixed-order sum of the per-workgroup partials of
// s^T D^-1 s and s.y, the new spectral step, history, reset of the line-search state (bluest/spg.py:100-120).
__device__ __forceinline__ void spg_update_tail(double *__restrict__ st, const double2 *partial, int nblocks, int lane)
{
    double a = 0.0, b = 0.0;
    for (int t = lane; t < nblocks; t += 64) {
        const unsigned long long *q = reinterpret_cast<const unsigned long long *>(partial + t);
        a += __longlong_as_double((long long)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        b += __longlong_as_double((long long)__hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    const double sdots = wave_sum(a), sdoty = wave_sum(b);
    if (lane == 0) {
        st[SPG_SDOTS] = sdots;
        st[SPG_SDOTY] = sdoty;
        const double lmin = st[SPG_LMIN], lmax = st[SPG_LMAX], fnew = st[SPG_FNEW];
        st[SPG_LAMBDA] = (sdoty <= 0.0) ? lmax : fmin(lmax, fmax(lmin, sdots / sdoty));
        const double it = st[SPG_IT] + 1.0;
        st[SPG_IT] = it;
        st[SPG_F] = fnew;
        const int H = (int)st[SPG_HLEN];
        st[SPG_HIST + ((long long)it % H)] = fnew;
        st[SPG_ALPHA] = 1.0;
        st[SPG_ACCEPT] = 0.0;
    }
}

// The workgroups of an update launch publish their partial sums, take a ticket, and the LAST one to arrive runs the tail above:
// no second launch (same publication protocol as the decision in the tail of k_solve_from_chunks, csrc/plan.hip).
__device__ __forceinline__ void spg_update_publish(double *__restrict__ st, double2 *__restrict__ partial, double sdots, double sdoty,
                                                   int tid, int *s_last)
{
    if (tid == 0) {
        partial[blockIdx.x] = make_double2(sdots, sdoty);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int *ticket = reinterpret_cast<unsigned int *>(st + SPG_TICKET);
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == gridDim.x - 1u) ? 1 : 0;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        *s_last = last;
    }
    __syncthreads();
    if (!*s_last || tid >= 64) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    spg_update_tail(st, partial, (int)gridDim.x, tid);
}

//  update launch (multi-block): s = xnew - x, y = gnew - g, per-block partial sums of s^T D^-1 s (D = diag(max(x,floor))) and
//  s.y, x <- xnew, g <- gnew; the last workgroup to finish runs spg_update_tail.
#define SPG_UPD_BLOCKS_MAX 512
__global__ __launch_bounds__(1024) void k_spg_update_a(double *__restrict__ x, double *__restrict__ g,
                                                       const double *__restrict__ xnew, const double *__restrict__ gnew,
                                                       double *__restrict__ st, double floor, int64_t L,
                                                       double2 *__restrict__ partial)
{
    __shared__ ProjLds sm;
    __shared__ int s_last;
    int ph = 0;
    const int tid = threadIdx.x;
    if (st[SPG_DONE] != 0.0 || st[SPG_FAIL] != 0.0 || st[SPG_ACCEPT] == 0.0) return;
    double sdots = 0.0, sdoty = 0.0;
    long long dummy = 0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + tid; i < L; i += (int64_t)gridDim.x * 1024) {
        const double xi = x[i], gi = g[i], xn = xnew[i], gn = gnew[i];
        const double sv = xn - xi, yv = gn - gi;
        sdots += (floor > 0.0) ? sv * sv / fmax(xi, floor) : sv * sv;
        sdoty = fma(sv, yv, sdoty);
        x[i] = xn;
        g[i] = gn;
    }
    block_sum2_cnt(sdots, sdoty, dummy, sm, tid, ph);
    spg_update_publish(st, partial, sdots, sdoty, tid, &s_last);
}

// A with the gradient fold fused in: gnew_j = scale_j * sum_o coef_o * grad_o[local_o(j)] is formed on the fly (no gnew
// vector, one launch less per iteration).
__global__ __launch_bounds__(1024) void k_spg_update_a_fused(double *__restrict__ x, double *__restrict__ g,
                                                             const double *__restrict__ xnew, const double *__restrict__ grad,
                                                             const int64_t *__restrict__ goff, const int32_t *__restrict__ invmap,
                                                             int n_out, const double *__restrict__ scale,
                                                             double *__restrict__ st, double floor, int64_t L,
                                                             double2 *__restrict__ partial, int identity)
{
    __shared__ ProjLds sm;
    __shared__ int s_last;
    int ph = 0;
    const int tid = threadIdx.x;
    if (st[SPG_DONE] != 0.0 || st[SPG_FAIL] != 0.0 || st[SPG_ACCEPT] == 0.0) return;
    double sdots = 0.0, sdoty = 0.0;
    long long dummy = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + tid; i < L; i += (int64_t)gridDim.x * blockDim.x) {
        double gn = 0.0;
        if (identity) {   // every output over all groups in global order: coalesced streams, no index loads
            for (int o = 0; o < n_out; o++) gn = fma(st[SPG_COEF + o], grad[(int64_t)o * L + i], gn);
        } else {
            for (int o = 0; o < n_out; o++) {
                const int32_t li = invmap[(int64_t)o * L + i];
                if (li >= 0) gn = fma(st[SPG_COEF + o], grad[goff[o] + li], gn);
            }
        }
        gn *= scale[i];
        const double xi = x[i], gi = g[i], xn = xnew[i];
        const double sv = xn - xi, yv = gn - gi;
        sdots += (floor > 0.0) ? sv * sv / fmax(xi, floor) : sv * sv;
        sdoty = fma(sv, yv, sdoty);
        x[i] = xn;
        g[i] = gn;
    }
    block_sum2_cnt(sdots, sdoty, dummy, sm, tid, ph);
    spg_update_publish(st, partial, sdots, sdoty, tid, &s_last);
}

// Small plans (K_tot <= 4096, a few hundred gradient tiles: the working set of the solver): the accepted step's gradient tiles,
// the fused gradient fold + update and the Barzilai-Borwein bookkeeping in ONE single-workgroup kernel instead of three launches
// (k_grad_tiles, k_spg_update_a_fused) -- at this size every launch is pure latency.
template <int KU>
__global__ __launch_bounds__(1024) void k_spg_finish_small(const TileDesc *__restrict__ tiles, int64_t n_tiles,
                                                           const double *__restrict__ tvals,
                                                           const double *__restrict__ v, const int32_t *__restrict__ status,
                                                           int N, int n_out, double *__restrict__ grad,
                                                           double *__restrict__ x, double *__restrict__ g,
                                                           const double *__restrict__ xnew, const int64_t *__restrict__ goff,
                                                           const int32_t *__restrict__ invmap, const double *__restrict__ scale,
                                                           double *__restrict__ st, double floor, int64_t L)
{
    __shared__ ProjLds sm;
    int ph = 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (st[SPG_DONE] != 0.0 || st[SPG_FAIL] != 0.0 || st[SPG_ACCEPT] == 0.0) return;
    // (1) gradient tiles of the accepted trial point: one wavefront per tile, 16 at a time
    for (int64_t t = wave; t < n_tiles; t += (blockDim.x >> 6)) {
        const TileDesc td = tiles[t];
        if ((td.n_valid & 0xffff) == 0) continue;
#define GT(KK) case KK: if (KK <= KU) { grad_tile<(KK <= KU ? KK : 1)>(td, tvals, v, status, N, n_out, 1, grad, 0, lane); break; }
        switch (td.k) {
            GT(1) GT(2) GT(3) GT(4) GT(5) GT(6) GT(7) GT(8) GT(9) GT(10) GT(11) GT(12)
            default: grad_tile_generic(td, tvals, v, status, N, n_out, 1, grad, 0, lane);
        }
#undef GT
    }
    __threadfence_block();
    __syncthreads();
    // (2) gnew_j = scale_j * sum_o coef_o grad_o[local_o(j)], s = xnew - x, y = gnew - g, x <- xnew, g <- gnew
    double sdots = 0.0, sdoty = 0.0;
    long long dummy = 0;
    for (int64_t i = tid; i < L; i += blockDim.x) {
        double gn = 0.0;
        for (int o = 0; o < n_out; o++) {
            const int32_t li = invmap[(int64_t)o * L + i];
            if (li >= 0) gn = fma(st[SPG_COEF + o], grad[goff[o] + li], gn);
        }
        gn *= scale[i];
        const double xi = x[i], gi = g[i], xn = xnew[i];
        const double sv = xn - xi, yv = gn - gi;
        sdots += (floor > 0.0) ? sv * sv / fmax(xi, floor) : sv * sv;
        sdoty = fma(sv, yv, sdoty);
        x[i] = xn;
        g[i] = gn;
    }
    block_sum2_cnt(sdots, sdoty, dummy, sm, tid, ph);
    // (3) Barzilai-Borwein step, history, reset of the line-search state (as spg_update_tail)
    if (tid == 0) {
        const double lmin = st[SPG_LMIN], lmax = st[SPG_LMAX], fnew = st[SPG_FNEW];
        st[SPG_SDOTS] = sdots;
        st[SPG_SDOTY] = sdoty;
        st[SPG_LAMBDA] = (sdoty <= 0.0) ? lmax : fmin(lmax, fmax(lmin, sdots / sdoty));
        const double it = st[SPG_IT] + 1.0;
        st[SPG_IT] = it;
        st[SPG_F] = fnew;
        const int H = (int)st[SPG_HLEN];
        st[SPG_HIST + ((long long)it % H)] = fnew;
        st[SPG_ALPHA] = 1.0;
        st[SPG_ACCEPT] = 0.0;
    }
}

// p = argmin sum_i (p_i - u_i)^2 / s_i  s.t. p >= 0, sum p = z, with u = x - lambda*s*g:
//   p_i = s_i * max(r_i - tau, 0),  r_i = x_i/s_i - lambda*g_i,  sum_i s_i max(r_i - tau, 0) = z.
// floor == 0: s = 1 (plain Euclidean projection, the reference-style SPG step);
// floor  > 0: s_i = max(x_i, floor) (variable "entropic" metric: the scaled SPG step).
// ITEMS > 0: ratios r and weights s are cached in registers (L <= 512*ITEMS);
// ITEMS == 0: everything is recomputed from x,g in every pass.
// spg_mode 1 (direction of the device-resident SPG): lambda from the state, and the FIRST trial point of the line search
// (alpha = 1: xnew = x + d, m = scale*xnew, gate open) is written by the same kernel.  spg_mode 2: convergence projection.
#define SIMPLEX_BLOCK 512   // 8 wavefronts: up to 256 VGPRs per lane, so (r, s) for 48 items stay in registers
template <int ITEMS>
__global__ __launch_bounds__(SIMPLEX_BLOCK) void k_simplex(const double *__restrict__ x, const double *__restrict__ g,
                                                  double lambda, double z, double floor, int64_t L,
                                                  double *__restrict__ p, double *__restrict__ d,
                                                  double *__restrict__ stats, double *__restrict__ spg_state, int spg_mode,
                                                  const double *__restrict__ scale, double *__restrict__ xnew,
                                                  double *__restrict__ mtrial, int32_t *__restrict__ enable)
{
    __shared__ ProjLds s;
    int ph = 0;
    const int tid = threadIdx.x;
    if (spg_state) {   // device-resident SPG: a finished / failed run is a no-op
        if (spg_state[SPG_DONE] != 0.0 || spg_state[SPG_FAIL] != 0.0) {
            if (enable && tid == 0) *enable = 0;
            return;
        }
        if (proj_pending(spg_state, spg_mode)) {   // line search pending: the next trial point instead of a new direction
            if (!xnew) return;                     // (without the fused trial outputs the caller launches bluest_spg_trial)
            const double alpha = spg_state[SPG_ALPHA];
            for (int64_t i = tid; i < L; i += blockDim.x) spg_trial_point(i, alpha, x, d, scale, xnew, mtrial);
            if (tid == 0) *enable = 1;
            return;
        }
        if (spg_mode == 1) lambda = spg_state[SPG_LAMBDA];   // direction: the step length lives in HBM
    }
    constexpr int R = ITEMS > 0 ? ITEMS : 1;
    const int B = (int)blockDim.x;   // <= SIMPLEX_BLOCK; short vectors run with fewer wavefronts: every workgroup reduction is cheaper
    double r[R], sc[R];
    auto scale_of = [&](double xi) -> double { return floor > 0.0 ? fmax(xi, floor) : 1.0; };
    auto ratio_of = [&](int64_t i, double xi) -> double {
        const double q = (floor > 0.0) ? ((xi >= floor) ? 1.0 : xi / floor) : xi;
        return g ? fma(-lambda, g[i], q) : q;
    };

    double rmax = -INFINITY;
    if (ITEMS > 0) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int64_t i = (int64_t)k * B + tid;
            const double xi = (i < L) ? x[i] : 0.0;
            sc[k] = (i < L) ? scale_of(xi) : 0.0;
            r[k] = (i < L) ? ratio_of(i, xi) : -INFINITY;
            rmax = fmax(rmax, r[k]);
        }
    } else {
        for (int64_t i = tid; i < L; i += B) rmax = fmax(rmax, ratio_of(i, x[i]));
    }
    rmax = block_max(rmax, s, tid, ph);
    if (ITEMS > 0) {
#pragma unroll
        for (int k = 0; k < R; k++) r[k] -= rmax;   // all ratios <= 0; the threshold lies in [-z/min s, 0)
    }
    // cold start: everything active (Michelot).  Direction of the device-resident SPG: warm start from the previous direction's
    // multiplier theta = (1 - tau_abs)/lambda, which moves slowly along the iteration; the search ends at the unique fixed point
    // whatever the start, so the result is the same bit for bit (measured: 8.9 -> 8.6 us per launch on the working set)
    const double cold = (floor > 0.0) ? -z / floor : -z;
    const bool use_theta = spg_state && spg_mode == 1 && g && lambda > 0.0;
    const double hint = use_theta ? 1.0 - lambda * spg_state[SPG_THETA] - rmax : cold;
    bool warm = use_theta && isfinite(hint) && hint > cold && hint < 0.0;
    double tau = warm ? hint : cold;
    long long prev = -1;
    for (int iter = 0; iter < 300; iter++) {
        double s1 = 0.0, s0 = 0.0;
        long long cnt = 0;
        if (ITEMS > 0) {
#pragma unroll
            for (int k = 0; k < R; k++) {
                const bool act = r[k] > tau;
                s1 = act ? fma(sc[k], r[k], s1) : s1;
                s0 = act ? s0 + sc[k] : s0;
                cnt += act ? 1 : 0;
            }
        } else {
            for (int64_t i = tid; i < L; i += B) {
                const double xi = x[i];
                const double ri = ratio_of(i, xi) - rmax;
                if (ri > tau) { const double si = scale_of(xi); s1 = fma(si, ri, s1); s0 += si; cnt++; }
            }
        }
        block_sum2_cnt(s1, s0, cnt, s, tid, ph);
        if (cnt == 0 && warm) { warm = false; tau = cold; prev = -1; continue; }   // hint right of every r_i: cold start
        if (cnt == prev || cnt == 0) break;
        prev = cnt;
        tau = (s1 - z) / s0;
    }
    double gd = 0.0, dmax = 0.0;
    long long npos = 0;
    auto emit = [&](int64_t i, double ri, double si) {
        const double pi = si * fmax(ri - tau, 0.0);
        const double di = pi - x[i];
        if (p) p[i] = pi;
        if (d) d[i] = di;
        if (xnew) { xnew[i] = pi; mtrial[i] = scale[i] * pi; }   // x + 1.0*d = p
        if (g) gd = fma(g[i], di, gd);
        dmax = fmax(dmax, fabs(di));
        npos += (pi > 0.0);
    };
    if (ITEMS > 0) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int64_t i = (int64_t)k * B + tid;
            if (i < L) emit(i, r[k], sc[k]);
        }
    } else {
        for (int64_t i = tid; i < L; i += B) { const double xi = x[i]; emit(i, ratio_of(i, xi) - rmax, scale_of(xi)); }
    }
    block_sum_cnt(gd, npos, s, tid, ph);
    dmax = block_max(dmax, s, tid, ph);
    if (tid == 0 && spg_state && spg_mode == 1) {
        spg_state[SPG_GDPARTS_N] = 0.0;   // g.d below is the folded value
        if (use_theta) spg_state[SPG_THETA] = (1.0 - (tau + rmax)) / lambda;
    }
    if (tid == 0 && stats) {
        stats[0] = gd;
        stats[1] = dmax;
        stats[2] = tau;
        stats[3] = (double)npos;
    }
    if (tid == 0 && enable) *enable = 1;
    // convergence projection of the device-resident SPG: gpmax = max|P(x - s*g) - x| <= eps ends the run (spg.py:68)
    if (tid == 0 && spg_state && spg_mode == 2 && dmax <= spg_state[SPG_EPS]) spg_state[SPG_DONE] = 1.0;
}

// ---- the same projection for long vectors: one CU cannot stream x, g, p, d fast enough (a single workgroup moves
// ~25-60 GB/s), so the streaming parts run on many CUs and only the threshold search is a single workgroup:
//   A (multi-block) r_i, s_i -> workspace, per-block max r
//   B (one workgroup) Michelot/Newton search for tau on (r, s) held in registers
//   C (multi-block) p, d (and the fused first trial point), per-block partials of g.d, max|d|, #positive
//   D (one wavefront) fold the partials -> stats, gate, convergence flag
// Mailboxes of the single-launch projection (k_proj_fused) inside the workspace.  The workspace should be ZERO-FILLED
// once before its first use (tag 0 is never sent, so an all-zero mailbox reads as "nothing there yet").
struct FusedProj {
    static constexpr int MAXB = 256;      // workgroups (<= compute units: all of them are resident at once)
    static constexpr int MAXP = 60;       // Newton passes per search
    static constexpr int EPOCH_STEP = 64; // tags used per launch (passes + final statistics)
    // a mailbox = 8 x 64-bit words = four doubles, each split into two (tag << 32 | 32 payload bits) words: ONE 64-byte store
    // (a 12-word message for the pass data was tried: every pass cost ~10 us instead of ~5, two lines to become visible)
    // doubles: [0, 2*MAXB*8) double-buffered pass mailboxes   [.., +MAXB*8) final-statistics mailboxes
    static constexpr int PART = 0, FIN = 2 * MAXB * 8, DOUBLES = FIN + MAXB * 8;
};
struct ProjWs {            // layout of the caller-provided workspace (doubles)
    // [0, 2L): interleaved (r_i, s_i) pairs
    static __host__ __device__ int64_t part_off(int64_t L) { return 2 * L; }            // 4 doubles per block
    // (room for at least 64 blocks of partials: the single-launch path may run 64 workgroups of 256 threads on a short vector)
    static __host__ __device__ int64_t tau_off(int64_t L, int nb) { return 2 * L + 4LL * (nb > 64 ? nb : 64); }   // tau, rmax
    static __host__ __device__ int64_t sync_off(int64_t L, int nb) { return tau_off(L, nb) + 16; }   // single-launch path
    static __host__ __device__ int64_t total(int64_t L, int nb) { return sync_off(L, nb) + FusedProj::DOUBLES; }
};


__global__ __launch_bounds__(1024) void k_proj_a(const double *__restrict__ x, const double *__restrict__ g, double lambda,
                                                 double floor, int64_t L, double *__restrict__ ws, int nb,
                                                 const double *__restrict__ spg_state, int spg_mode)
{
    __shared__ ProjLds sm;
    int ph = 0;
    if (proj_idle(spg_state) || proj_pending(spg_state, spg_mode)) return;
    if (spg_state && spg_mode == 1) lambda = spg_state[SPG_LAMBDA];
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    double ri = -INFINITY;
    if (i < L) {
        const double xi = x[i];
        const double si = floor > 0.0 ? fmax(xi, floor) : 1.0;
        const double q = (floor > 0.0) ? ((xi >= floor) ? 1.0 : xi / floor) : xi;
        ri = g ? fma(-lambda, g[i], q) : q;
        reinterpret_cast<double2 *>(ws)[i] = make_double2(ri, si);   // interleaved (r, s): one 16-byte access per item
    }
    const double bm = block_max(ri, sm, tid, ph);
    if (tid == 0) ws[ProjWs::part_off(L) + 4LL * blockIdx.x] = bm;
    if (tid == 0 && blockIdx.x == 0) ws[ProjWs::tau_off(L, nb) + 8] = g ? lambda : 0.0;   // for the warm start of the search
}

// Warm start: the search is Newton's method on the convex piecewise-linear function sum_i s_i max(r_i - tau, 0) - z, which
// converges from ANY starting point with a non-empty active set (one step lands left of the root, then it climbs
// monotonically; active sets are nested, so "same count twice" still means "same set").  With r_i = q_i - lambda g_i and
// q_i = 1 on the support, entry i is active iff g_i < theta := (1 - tau_abs)/lambda -- the multiplier of the simplex
// constraint, which settles as SPG converges while lambda (the spectral step) jumps around.  So every search starts from
// the previous theta of its kind (ws[tau_off + 4 + mode]) and typically needs 2-3 passes instead of 10-15; an unusable hint
// (not finite, empty active set) falls back to the cold start with everything active.  Without a gradient the hint is the
// previous absolute threshold itself.
//   ws[tau_off + 0] tau  [+1] rmax  [+2] previous active count  [+3] converged flag  [+4..6] hints  [+7] cold-start fallback
//   [+8] lambda of this projection (0: no gradient)  [+9] passes so far (diagnostics)  [+10] searches so far
__device__ __forceinline__ double proj_start(const double *t, int mode, double rmax, double cold, bool &warm)
{
    const double lambda = t[8];
    const double hint_abs = (lambda > 0.0) ? 1.0 - lambda * t[4 + mode] : t[4 + mode];
    const double hint = hint_abs - rmax;
    warm = isfinite(hint) && hint > cold && hint < 0.0;
    return warm ? hint : cold;
}
__device__ __forceinline__ void proj_remember(double *t, int mode, double tau_abs, int passes)
{
    const double lambda = t[8];
    t[4 + mode] = (lambda > 0.0) ? (1.0 - tau_abs) / lambda : tau_abs;
    t[9] += (double)passes;
    t[10] += 1.0;
}

template <int ITEMS>   // ITEMS*1024 >= L, or ITEMS == 0: stream (r, s) from the workspace in every pass
__global__ __launch_bounds__(1024) void k_proj_b(double z, double floor, int64_t L, double *__restrict__ ws, int nb,
                                                 const double *__restrict__ spg_state, int mode)
{
    __shared__ ProjLds sm;
    int ph = 0;
    if (proj_idle(spg_state) || proj_pending(spg_state, mode)) return;
    const int tid = threadIdx.x;
    double rmax = -INFINITY;
    for (int b = tid; b < nb; b += 1024) rmax = fmax(rmax, ws[ProjWs::part_off(L) + 4LL * b]);
    rmax = block_max(rmax, sm, tid, ph);
    constexpr int R = ITEMS > 0 ? ITEMS : 1;
    double r[R], sc[R];
    const double2 *rs = reinterpret_cast<const double2 *>(ws);
    if (ITEMS > 0) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int64_t i = (int64_t)k * 1024 + tid;
            const double2 q = (i < L) ? rs[i] : make_double2(-INFINITY, 0.0);
            r[k] = q.x - rmax;
            sc[k] = q.y;
        }
    }
    double *t = ws + ProjWs::tau_off(L, nb);
    const double cold = (floor > 0.0) ? -z / floor : -z;
    bool warm;
    double tau = proj_start(t, mode, rmax, cold, warm);
    long long prev = -1;
    int passes = 0;
    for (int iter = 0; iter < 300; iter++) {
        double s1 = 0.0, s0 = 0.0;
        long long cnt = 0;
        passes++;
        if (ITEMS > 0) {
#pragma unroll
            for (int k = 0; k < R; k++) {
                const bool act = r[k] > tau;
                s1 = act ? fma(sc[k], r[k], s1) : s1;
                s0 = act ? s0 + sc[k] : s0;
                cnt += act ? 1 : 0;
            }
        } else {
            for (int64_t i = tid; i < L; i += 1024) {
                const double2 q = rs[i];
                const double ri = q.x - rmax;
                if (ri > tau) { s1 = fma(q.y, ri, s1); s0 += q.y; cnt++; }
            }
        }
        block_sum2_cnt(s1, s0, cnt, sm, tid, ph);
        if (cnt == 0 && warm) { warm = false; tau = cold; prev = -1; continue; }   // hint right of every r_i: cold start
        if (cnt == prev || cnt == 0) break;
        prev = cnt;
        tau = (s1 - z) / s0;
    }
    if (tid == 0) { t[0] = tau; t[1] = rmax; proj_remember(t, mode, tau + rmax, passes); }
}

// Threshold search for vectors too long for one workgroup's registers (L > 24576): every Michelot/Newton pass is a
// multi-block launch (P: per-block partial sums over the current active set) plus a one-wavefront launch (Q: new tau,
// convergence flag).  A fixed number of passes is enqueued; passes after convergence exit on the flag, and k_proj_b<0>
// (single workgroup, streaming) finishes the search in the rare case the flag is still clear.
//   ws[tau_off + 0] tau   [+1] rmax   [+2] previous active count   [+3] converged flag
__global__ __launch_bounds__(64) void k_proj_q0(double z, double floor, int64_t L, double *__restrict__ ws, int nb,
                                                const double *__restrict__ spg_state, int mode)
{
    if (proj_idle(spg_state) || proj_pending(spg_state, mode)) return;
    const int lane = threadIdx.x;
    double rmax = -INFINITY;
    for (int b = lane; b < nb; b += 64) rmax = fmax(rmax, ws[ProjWs::part_off(L) + 4LL * b]);
    rmax = wave_max(rmax);
    if (lane == 0) {
        double *t = ws + ProjWs::tau_off(L, nb);
        const double cold = (floor > 0.0) ? -z / floor : -z;
        bool warm;
        t[0] = proj_start(t, mode, rmax, cold, warm);
        t[1] = rmax;
        t[2] = -1.0;
        t[3] = 0.0;
        t[7] = warm ? cold : 0.0;   // non-zero: the cold start to fall back to if the hint's active set is empty
    }
}

__global__ __launch_bounds__(1024) void k_proj_p(int64_t L, double *__restrict__ ws, int nb, const double *__restrict__ spg_state,
                                                 int mode)
{
    __shared__ ProjLds sm;
    int ph = 0;
    if (proj_idle(spg_state) || proj_pending(spg_state, mode)) return;
    const double *t = ws + ProjWs::tau_off(L, nb);
    if (t[3] != 0.0) return;
    const double tau = t[0], rmax = t[1];
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    double s1 = 0.0, s0 = 0.0;
    long long cnt = 0;
    if (i < L) {
        const double2 q = reinterpret_cast<const double2 *>(ws)[i];
        const double ri = q.x - rmax;
        if (ri > tau) { s1 = q.y * ri; s0 = q.y; cnt = 1; }
    }
    block_sum2_cnt(s1, s0, cnt, sm, tid, ph);
    if (tid == 0) {
        double *pp = ws + ProjWs::part_off(L) + 4LL * blockIdx.x;
        pp[1] = s1; pp[2] = s0; pp[3] = (double)cnt;
    }
}

__global__ __launch_bounds__(64) void k_proj_q(double z, int64_t L, double *__restrict__ ws, int nb, const double *__restrict__ spg_state,
                                               int mode)
{
    if (proj_idle(spg_state) || proj_pending(spg_state, mode)) return;
    double *t = ws + ProjWs::tau_off(L, nb);
    if (t[3] != 0.0) return;
    const int lane = threadIdx.x;
    double s1 = 0.0, s0 = 0.0, cnt = 0.0;
    for (int b = lane; b < nb; b += 64) {
        const double *pp = ws + ProjWs::part_off(L) + 4LL * b;
        s1 += pp[1]; s0 += pp[2]; cnt += pp[3];
    }
    s1 = wave_sum(s1); s0 = wave_sum(s0); cnt = wave_sum(cnt);
    if (lane == 0) {
        if (cnt == 0.0 && t[7] != 0.0) { t[0] = t[7]; t[7] = 0.0; t[2] = -1.0; return; }   // unusable hint: cold start
        t[9] += 1.0;
        if (cnt == t[2] || cnt == 0.0) { t[3] = 1.0; proj_remember(t, mode, t[0] + t[1], 0); return; }
        t[2] = cnt;
        t[0] = (s1 - z) / s0;
    }
}

// finishing search (single workgroup, streaming) if the enqueued passes did not reach the fixed point
__global__ __launch_bounds__(1024) void k_proj_b_finish(double z, int64_t L, double *__restrict__ ws, int nb,
                                                        const double *__restrict__ spg_state, int mode)
{
    __shared__ ProjLds sm;
    int ph = 0;
    if (proj_idle(spg_state) || proj_pending(spg_state, mode)) return;
    double *t = ws + ProjWs::tau_off(L, nb);
    if (t[3] != 0.0) return;
    const int tid = threadIdx.x;
    const double rmax = t[1];
    double tau = t[0], cold = t[7];
    long long prev = (long long)t[2];
    const double2 *rs = reinterpret_cast<const double2 *>(ws);
    for (int iter = 0; iter < 1000; iter++) {
        double s1 = 0.0, s0 = 0.0;
        long long cnt = 0;
        for (int64_t i = tid; i < L; i += 1024) {
            const double2 q = rs[i];
            const double ri = q.x - rmax;
            if (ri > tau) { s1 = fma(q.y, ri, s1); s0 += q.y; cnt++; }
        }
        block_sum2_cnt(s1, s0, cnt, sm, tid, ph);
        if (cnt == 0 && cold != 0.0) { tau = cold; cold = 0.0; prev = -1; continue; }   // unusable hint: cold start
        if (cnt == prev || cnt == 0) break;
        prev = cnt;
        tau = (s1 - z) / s0;
    }
    if (tid == 0) { t[0] = tau; t[3] = 1.0; proj_remember(t, mode, tau + rmax, 100); }   // 100: the fallback ran (diagnostics)
}

__global__ __launch_bounds__(1024) void k_proj_c(const double *__restrict__ x, const double *__restrict__ g, int64_t L,
                                                 double *__restrict__ ws, int nb, double *__restrict__ p, double *__restrict__ d,
                                                 const double *__restrict__ scale, double *__restrict__ xnew,
                                                 double *__restrict__ mtrial, const double *__restrict__ spg_state, int mode)
{
    __shared__ ProjLds sm;
    int ph = 0;
    if (proj_idle(spg_state)) return;
    if (proj_pending(spg_state, mode)) {   // line search pending: the next trial point instead of a new direction
        const int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x;
        if (xnew && j < L) spg_trial_point(j, spg_state[SPG_ALPHA], x, d, scale, xnew, mtrial);
        return;
    }
    const int tid = threadIdx.x;
    const double tau = ws[ProjWs::tau_off(L, nb)], rmax = ws[ProjWs::tau_off(L, nb) + 1];
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    double gd = 0.0, dm = 0.0;
    long long npos = 0;
    if (i < L) {
        const double2 q = reinterpret_cast<const double2 *>(ws)[i];
        const double pi = q.y * fmax(q.x - rmax - tau, 0.0);
        const double di = pi - x[i];
        if (p) p[i] = pi;
        if (d) d[i] = di;
        if (xnew) { xnew[i] = pi; mtrial[i] = scale[i] * pi; }
        if (g) gd = g[i] * di;
        dm = fabs(di);
        npos = pi > 0.0;
    }
    block_sum_cnt(gd, npos, sm, tid, ph);
    dm = block_max(dm, sm, tid, ph);
    if (tid == 0) {
        double *pp = ws + ProjWs::part_off(L) + 4LL * blockIdx.x;
        pp[1] = gd; pp[2] = dm; pp[3] = (double)npos;
    }
}

__global__ __launch_bounds__(64) void k_proj_d(int64_t L, const double *__restrict__ ws, int nb, double *__restrict__ stats,
                                               double *__restrict__ spg_state, int spg_mode, int32_t *__restrict__ enable)
{
    const int lane = threadIdx.x;
    if (proj_idle(spg_state)) { if (enable && lane == 0) *enable = 0; return; }
    if (proj_pending(spg_state, spg_mode)) { if (enable && lane == 0) *enable = 1; return; }
    double gd = 0.0, dm = 0.0, np = 0.0;
    for (int b = lane; b < nb; b += 64) {
        const double *pp = ws + ProjWs::part_off(L) + 4LL * b;
        gd += pp[1]; dm = fmax(dm, pp[2]); np += pp[3];
    }
    gd = wave_sum(gd); dm = wave_max(dm); np = wave_sum(np);
    if (lane == 0) {
        if (stats) { stats[0] = gd; stats[1] = dm; stats[2] = ws[ProjWs::tau_off(L, nb)]; stats[3] = np; }
        if (spg_state && spg_mode == 1) spg_state[SPG_GDPARTS_N] = 0.0;   // g.d above is the folded value
        if (enable) *enable = 1;
        if (spg_state && spg_mode == 2 && dm <= spg_state[SPG_EPS]) spg_state[SPG_DONE] = 1.0;
    }
}

// ---- single-launch projection: A + threshold search + C + D in ONE kernel -----------------------------------------
// nb <= (number of compute units) workgroups of 1024 threads, so every workgroup is resident and waiting on each other
// cannot starve anybody.  Workgroups talk through MAILBOXES in HBM, with no read-modify-write atomics and no fences:
// a message of four doubles is stored as eight 64-bit words, each word = (tag << 32 | 32 payload bits).  Aligned 64-bit
// stores and loads are single-copy atomic, so a reader that sees the expected tag in ALL eight words has the complete
// message of exactly this launch and pass, whatever order the words became visible in; stale or half-written mailboxes
// simply fail the tag check and are polled again (device-coherent loads, s_sleep back-off).  Tags advance by EPOCH_STEP
// per launch (kept in the workspace), so nothing ever has to be reset.  Every wait is bounded: after PROJ_SPIN_LIMIT polls
// a workgroup gives up, the projection returns NaN and a sticky error flag is raised, so every wavefront terminates.
// All workgroups fold the messages in the same fixed order, hence compute bit-identical thresholds and take identical
// branches.  (r, s) of ITEMS entries per thread stay in registers over the whole search.
#define PROJ_SPIN_LIMIT 400000u
__device__ __forceinline__ void mailbox_send(double *box, unsigned int tag, double v0, double v1, double v2, double v3)
{   // called by lanes 0..7 of one wavefront: one coalesced 64-byte store
    const int l = threadIdx.x & 7;
    const double v = (l < 2) ? v0 : (l < 4) ? v1 : (l < 6) ? v2 : v3;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned long long half = (l & 1) ? (bits >> 32) : (bits & 0xffffffffull);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(box) + l, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// every thread with `active` polls its own mailbox; returns (uniformly for the workgroup) whether all of them arrived
__device__ __forceinline__ bool mailbox_recv(const double *box, unsigned int tag, bool active, double (&v)[4])
{
    int ok = 1;
    v[0] = v[1] = v[2] = v[3] = 0.0;
    if (active) {
        const unsigned long long *w = reinterpret_cast<const unsigned long long *>(box);
        unsigned int spins = 0;
        for (;;) {
            unsigned long long q[8];
#pragma unroll
            for (int i = 0; i < 8; i++) q[i] = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool all = true;
#pragma unroll
            for (int i = 0; i < 8; i++) all = all && (unsigned int)(q[i] >> 32) == tag;
            if (all) {
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = __longlong_as_double((long long)((q[2 * i] & 0xffffffffull) | (q[2 * i + 1] << 32)));
                break;
            }
            if (++spins > PROJ_SPIN_LIMIT) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return __syncthreads_and(ok) != 0;
}
// the same without the barrier: per-lane result (1 = arrived or inactive, 0 = timed out)
__device__ __forceinline__ int mailbox_poll(const double *box, unsigned int tag, bool active, double (&v)[4])
{
    int ok = 1;
    v[0] = v[1] = v[2] = v[3] = 0.0;
    if (active) {
        const unsigned long long *w = reinterpret_cast<const unsigned long long *>(box);
        unsigned int spins = 0;
        for (;;) {
            unsigned long long q[8];
#pragma unroll
            for (int i = 0; i < 8; i++) q[i] = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool all = true;
#pragma unroll
            for (int i = 0; i < 8; i++) all = all && (unsigned int)(q[i] >> 32) == tag;
            if (all) {
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = __longlong_as_double((long long)((q[2 * i] & 0xffffffffull) | (q[2 * i + 1] << 32)));
                break;
            }
            if (++spins > PROJ_SPIN_LIMIT) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return ok;
}

template <int ITEMS>
__global__ __launch_bounds__(1024) void k_proj_fused(const double *__restrict__ x, const double *__restrict__ g, double lambda,
                                                     double z, double floor, int64_t L, double *__restrict__ ws, int nb_ws,
                                                     double *__restrict__ p, double *__restrict__ d,
                                                     double *__restrict__ stats, double *__restrict__ spg_state, int spg_mode,
                                                     const double *__restrict__ scale, double *__restrict__ xnew,
                                                     double *__restrict__ mtrial, int32_t *__restrict__ enable, int maxp, int bracket)
{
    __shared__ ProjLds sm;
    int ph = 0;
    const int tid = threadIdx.x, nb = gridDim.x, b = blockIdx.x;
    const int64_t BT = blockDim.x;   // 1024, or 256 for vectors that 64 workgroups of 256 hold: cheaper workgroup reductions
    double *t = ws + ProjWs::tau_off(L, nb_ws);
    double *sy = ws + ProjWs::sync_off(L, nb_ws);
    if (proj_idle(spg_state) || t[12] != 0.0) {   // t[12]: sticky "a wait timed out" flag
        if (enable && b == 0 && tid == 0) *enable = 0;
        return;
    }
    if (proj_pending(spg_state, spg_mode)) {   // line search pending: the next trial point instead of a new direction
        if (!xnew) return;
        const double alpha = spg_state[SPG_ALPHA];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const int64_t i = ((int64_t)k * nb + b) * BT + tid;
            if (i < L) spg_trial_point(i, alpha, x, d, scale, xnew, mtrial);
        }
        if (b == 0 && tid == 0) *enable = 1;
        return;
    }
    const unsigned int epoch = (unsigned int)t[11];   // tags of this launch: epoch + 1 ... epoch + EPOCH_STEP
    if (spg_state && spg_mode == 1) lambda = spg_state[SPG_LAMBDA];
    const int mode = (spg_mode >= 0 && spg_mode <= 2) ? spg_mode : 0;
    const double hint_raw = t[4 + mode];

    // ---- A: ratios and weights into registers ----
    double r[ITEMS], sc[ITEMS];
    double rmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const int64_t i = ((int64_t)k * nb + b) * BT + tid;
        r[k] = -INFINITY; sc[k] = 0.0;
        if (i < L) {
            const double xi = x[i];
            sc[k] = floor > 0.0 ? fmax(xi, floor) : 1.0;
            const double q = (floor > 0.0) ? ((xi >= floor) ? 1.0 : xi / floor) : xi;
            r[k] = g ? fma(-lambda, g[i], q) : q;
            rmax = fmax(rmax, r[k]);
        }
    }

    // ---- B: warm-started Newton search for tau (see proj_start).  The search runs in coordinates shifted by the grid-wide
    // max r (differences r_i - rmax are exact where it matters, tau stays small), but no barrier is spent on that maximum:
    // pass 0 picks its active set with the hint in absolute coordinates (tau0 = -inf without a hint: everything active,
    // Michelot's first step), each workgroup sums relative to ITS OWN max m_b and publishes (s1_b, s0_b, count_b, m_b);
    // after the barrier everybody re-bases the partials to rmax = max_b m_b:  S1 = sum_b s1_b + s0_b (m_b - rmax).
    const bool use_theta = g != nullptr && lambda > 0.0;
    const double hint = use_theta ? 1.0 - lambda * hint_raw : hint_raw;
    const double mb = block_max(rmax, sm, tid, ph);          // this workgroup's max r
    bool ok = true, first = true;
    double tau = isfinite(hint) ? hint : -INFINITY;          // pass 0: absolute; later passes: relative to rmax
    int passes = 0;
    rmax = mb;
    // Every message is four doubles in ONE 64-byte mailbox: (s1_b, s0_b, m_b, bracket_b) in pass 0, (s1_b, s0_b, lo_b, hi_b) later;
    // the bracket = the largest ratio NOT in the active set and the smallest one in it (pass 0: as two floats rounded outwards).  After the exchange every
    // workgroup knows the new threshold AND whether it separates the same entries as the one the sums were taken at -- then it IS
    // the fixed point, and the confirming pass (one more device-wide exchange, ~5 us) is skipped; otherwise the search goes on
    // and ends when a pass reproduces the threshold bit for bit.  (No counts in the message: the weights are positive, so an
    // empty active set is s0 == 0.)
    for (int iter = 0; iter < maxp; iter++) {
        double s1 = 0.0, s0 = 0.0;
        double lo = -INFINITY, hi = INFINITY;                // largest inactive / smallest active ratio (coordinates of tau)
        passes++;
        const double ref = first ? mb : 0.0;                 // later passes: r[] is already shifted
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const bool act = r[k] > tau;
            s1 = act ? fma(sc[k], r[k] - ref, s1) : s1;
            s0 = act ? s0 + sc[k] : s0;
            hi = act ? fmin(hi, r[k]) : hi;
            lo = act ? lo : fmax(lo, r[k]);
        }
        block_pass4(s1, s0, lo, hi, sm, tid, ph);
        const unsigned int tag = epoch + 1u + (unsigned int)iter;
        // pass 0 needs the slot of the third double for m_b, so its bracket travels as two floats; later passes send it exactly
        const double bracket_bits = __hiloint2double(__float_as_int(__double2float_rd(hi)), __float_as_int(__double2float_ru(lo)));
        if (tid < 8)
            mailbox_send(sy + FusedProj::PART + ((iter & 1) * FusedProj::MAXB + b) * 8, tag, s1, s0, first ? mb : lo, first ? bracket_bits : hi);
        // the <= 64 messages are received and folded by wavefront 0 alone (wavefront reductions, no LDS, no barrier; the same
        // operation order as a workgroup-wide fold whose other wavefronts contribute identities), then ONE barrier hands the
        // five results to everybody
        const bool was_first = first;
        if (tid < WAVE) {
            double msg[4];
            const int got = mailbox_poll(sy + FusedProj::PART + ((iter & 1) * FusedProj::MAXB + (tid < nb ? tid : 0)) * 8, tag, tid < nb, msg);
            const bool all_ok = __ballot(got == 0) == 0ull;
            double q1 = msg[0], q0 = msg[1], ql, qh, rm = 0.0;
            if (first) {
                ql = (tid < nb) ? (double)__int_as_float(__double2loint(msg[3])) : -INFINITY;
                qh = (tid < nb) ? (double)__int_as_float(__double2hiint(msg[3])) : INFINITY;
                const double m = (tid < nb) ? msg[2] : -INFINITY;
                rm = wave_max(m);
                if (q0 > 0.0) q1 = fma(q0, m - rm, q1);          // re-base this workgroup's partial to the grid-wide max
                ql -= rm; qh -= rm;                              // pass 0 compared in absolute coordinates
            } else {
                ql = (tid < nb) ? msg[2] : -INFINITY;
                qh = (tid < nb) ? msg[3] : INFINITY;
            }
            q1 = wave_sum(q1); q0 = wave_sum(q0); ql = wave_max(ql); qh = -wave_max(-qh);
            if (tid == 0) { sm.xf[0] = all_ok ? 1.0 : 0.0; sm.xf[1] = q1; sm.xf[2] = q0; sm.xf[3] = ql; sm.xf[4] = qh; sm.xf[5] = rm; }
        }
        __syncthreads();
        ok = sm.xf[0] != 0.0;
        if (!ok) break;
        s1 = sm.xf[1]; s0 = sm.xf[2]; lo = sm.xf[3]; hi = sm.xf[4];
        if (first) rmax = sm.xf[5];
        if (first) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) r[k] -= rmax;
            first = false;
            if (s0 == 0.0) {                                 // hint right of every r_i (or no entries): all-active restart
                if (tau == -INFINITY) break;
                tau = -INFINITY;                             // r[] is shifted now, so the restart is an ordinary later pass
                continue;
            }
        } else if (s0 == 0.0) {
            break;
        }
        const double tau_new = (s1 - z) / s0;
        if (!was_first && tau_new == tau) break;             // this pass reproduced the threshold: fixed point
        tau = tau_new;
        if (bracket && lo <= tau && tau < hi) break;         // same active set on both sides of the update: fixed point
    }
    if (!ok) tau = NAN;

    // ---- C: p, d, the fused first trial point; partials of g.d, max|d|, #positive ----
    double gd = 0.0, dm = 0.0;
    long long npos = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const int64_t i = ((int64_t)k * nb + b) * BT + tid;
        if (i < L) {
            const double pi = sc[k] * fmax(r[k] - tau, 0.0);
            const double di = pi - x[i];
            if (p) p[i] = pi;
            if (d) d[i] = di;
            if (xnew) { xnew[i] = pi; mtrial[i] = scale[i] * pi; }
            if (g) gd = fma(g[i], di, gd);
            dm = fmax(dm, fabs(di));
            npos += pi > 0.0;
        }
    }
    if (spg_state && spg_mode == 1) {
        // direction of the device-resident SPG: the only statistic anybody consumes is g.d (the Armijo test, two launches later),
        // so no workgroup waits for the others here: each publishes its partial of g.d and spg_decide_wave folds them in fixed
        // order -- one mailbox exchange (~4 us) less per iteration.  Every workgroup holds the same tau, so workgroup 0 does
        // the bookkeeping alone; all others read epoch/hint/flag before the first exchange, i.e. before it can get here.
        long long unused = 0;
        block_sum_cnt(gd, unused, sm, tid, ph);
        if (tid == 0) {
            ws[ProjWs::part_off(L) + 4LL * b + 1] = ok ? gd : NAN;
            if (!ok) t[12] = 1.0;                                      // sticky: a wait timed out
            if (b == 0) {
                spg_state[SPG_GDPARTS] = __longlong_as_double((long long)(uintptr_t)(ws + ProjWs::part_off(L)));
                spg_state[SPG_GDPARTS_N] = (double)nb;
                if (enable) *enable = 1;
                t[0] = tau; t[1] = rmax;
                if (ok) t[4 + mode] = use_theta ? (1.0 - (tau + rmax)) / lambda : tau + rmax;
                t[9] += (double)passes;
                t[10] += 1.0;
                t[11] = (double)(epoch + (unsigned int)FusedProj::EPOCH_STEP);
            }
        }
        return;
    }
    block_sum_cnt(gd, npos, sm, tid, ph);
    dm = block_max(dm, sm, tid, ph);
    const unsigned int ftag = epoch + (unsigned int)FusedProj::EPOCH_STEP - 1u;
    if (tid < 8) mailbox_send(sy + FusedProj::FIN + b * 8, ftag, gd, dm, (double)npos, 0.0);
    if (b != 0) return;

    // ---- D: workgroup 0 collects everybody's statistics, remembers the hint and advances the epoch ----
    double msg[4];
    const bool got = mailbox_recv(sy + FusedProj::FIN + (tid < nb ? tid : 0) * 8, ftag, tid < nb, msg);
    gd = msg[0]; dm = msg[1]; npos = (long long)msg[2];
    block_sum_cnt(gd, npos, sm, tid, ph);
    dm = block_max(dm, sm, tid, ph);
    if (tid == 0) {
        if (!ok || !got) { t[12] = 1.0; tau = NAN; gd = NAN; dm = NAN; }   // sticky: a wait timed out, results are NaN from here on
        if (stats) { stats[0] = gd; stats[1] = dm; stats[2] = tau; stats[3] = (double)npos; }
        if (enable) *enable = 1;
        if (spg_state && spg_mode == 2 && dm <= spg_state[SPG_EPS]) spg_state[SPG_DONE] = 1.0;
        t[0] = tau; t[1] = rmax;
        if (ok && got) t[4 + mode] = use_theta ? (1.0 - (tau + rmax)) / lambda : tau + rmax;
        t[9] += (double)passes;
        t[10] += 1.0;
        t[11] = (double)(epoch + (unsigned int)FusedProj::EPOCH_STEP);   // exact in a double; wraps modulo 2^32 with the cast above
    }
}

// ------------------------------------------------------------------------------------------------------
// Part 3 host side
// ------------------------------------------------------------------------------------------------------
#define PROJ_PASSES 5   // enqueued multi-block Newton passes (warm-started searches need 2-3; the finishing kernel covers the rest)
static int simplex_impl(const double *x_dev, const double *g_dev, double lambda, double z, double floor, int64_t L, double *p_dev,
                        double *d_dev, double *stats_dev, double *spg_state, int spg_mode, void *stream,
                        const double *trial_scale = nullptr, double *trial_xnew = nullptr, double *trial_m = nullptr,
                        int32_t *trial_enable = nullptr, double *ws = nullptr)
{
    int rc = require_gpu(); if (rc) return rc;
    if (!x_dev || L <= 0) return fail(BLUEST_ERR_ARG, "bad x / L");
    if (!(z > 0.0)) return fail(BLUEST_ERR_ARG, "z must be positive");
    if (!(floor >= 0.0)) return fail(BLUEST_ERR_ARG, "floor must be >= 0");
    hipStream_t st = (hipStream_t)stream;
    if (ws && L > 4096) {   // long vector: streaming parts on many CUs
        const int nb = (int)((L + 1023) / 1024);
        {   // single launch with grid barriers while (r, s) of the whole vector fit the registers of <= #CU workgroups
            static int ncu_of[64] = {0};          // compute units per device (index = HIP device id)
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64) return fail(BLUEST_ERR_ARG, "device id %d out of range", dev);
            if (!ncu_of[dev]) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, dev));
                ncu_of[dev] = prop.multiProcessorCount;
            }
            const int ncu = ncu_of[dev];
            static int maxp = getenv("BLUEST_PROJ_MAXP") ? atoi(getenv("BLUEST_PROJ_MAXP")) : (int)FusedProj::MAXP;   // timing experiments
            static int bracket = getenv("BLUEST_PROJ_NO_BRACKET") ? 0 : 1;   // timing experiments
            static int maxb = getenv("BLUEST_PROJ_MAXB") ? atoi(getenv("BLUEST_PROJ_MAXB")) : 64;   // 64 measured best at L = 245505 (61 -> 42 us)
            // workgroup size: 256 threads when 64 of them hold the vector with <= 4 entries per thread (K_tot <= 65536), else 1024
            static int bt_env = getenv("BLUEST_PROJ_THREADS") ? atoi(getenv("BLUEST_PROJ_THREADS")) : 0;   // timing experiments
            const int bt = bt_env ? bt_env : (L <= 64LL * 256 * 4 ? 256 : 1024);
            const int nb_bt = (int)((L + bt - 1) / bt);
            const int nbf = std::max(1, std::min(std::min(nb_bt, ncu), std::min(std::min(maxb, 64), (int)FusedProj::MAXB)));   // <= 64: one wavefront folds the messages
            const int64_t items = (L + (int64_t)bt * nbf - 1) / ((int64_t)bt * nbf);
            if (items <= 16 && !getenv("BLUEST_PROJ_MULTI_LAUNCH")) {
#define PF(IT) hipLaunchKernelGGL((k_proj_fused<IT>), dim3(nbf), dim3(bt), 0, st, x_dev, g_dev, lambda, z, floor, L, ws, nb, p_dev, d_dev, \
                                  stats_dev, spg_state, spg_mode, trial_scale, trial_xnew, trial_m, trial_enable, maxp, bracket)
                if (items <= 1) PF(1); else if (items <= 2) PF(2); else if (items <= 4) PF(4); else if (items <= 8) PF(8); else PF(16);
#undef PF
                HIP_TRY(hipGetLastError());
                return BLUEST_OK;
            }
        }
        hipLaunchKernelGGL(k_proj_a, dim3(nb), dim3(1024), 0, st, x_dev, g_dev, lambda, floor, L, ws, nb, spg_state, spg_mode);
        const int mode = (spg_mode >= 0 && spg_mode <= 2) ? spg_mode : 0;   // one warm-start slot per kind of projection
#define PB(IT) hipLaunchKernelGGL((k_proj_b<IT>), dim3(1), dim3(1024), 0, st, z, floor, L, ws, nb, spg_state, mode)
        if (L <= 1024 * 8) PB(8);
        else if (L <= 1024 * 24) PB(24);
        else {   // long vector: multi-block Newton passes, then the (normally idle) finishing search
            hipLaunchKernelGGL(k_proj_q0, dim3(1), dim3(64), 0, st, z, floor, L, ws, nb, spg_state, mode);
            for (int pass = 0; pass < PROJ_PASSES; pass++) {
                hipLaunchKernelGGL(k_proj_p, dim3(nb), dim3(1024), 0, st, L, ws, nb, spg_state, mode);
                hipLaunchKernelGGL(k_proj_q, dim3(1), dim3(64), 0, st, z, L, ws, nb, spg_state, mode);
            }
            hipLaunchKernelGGL(k_proj_b_finish, dim3(1), dim3(1024), 0, st, z, L, ws, nb, spg_state, mode);
        }
#undef PB
        hipLaunchKernelGGL(k_proj_c, dim3(nb), dim3(1024), 0, st, x_dev, g_dev, L, ws, nb, p_dev, d_dev, trial_scale, trial_xnew, trial_m,
                           spg_state, mode);
        hipLaunchKernelGGL(k_proj_d, dim3(1), dim3(64), 0, st, L, ws, nb, stats_dev, spg_state, spg_mode, trial_enable);
        HIP_TRY(hipGetLastError());
        return BLUEST_OK;
    }
    // short vectors (the solver's working set is a few hundred entries): as few wavefronts as hold the vector with 4 entries per
    // thread -- the kernel is a chain of workgroup reductions, and those cost ~1.8 us each across 16 wavefronts
    int sblock = SIMPLEX_BLOCK;
    while (sblock > 64 && (int64_t)(sblock / 2) * 4 >= L) sblock /= 2;
#define SP(IT) hipLaunchKernelGGL((k_simplex<IT>), dim3(1), dim3(sblock), 0, st, x_dev, g_dev, lambda, z, floor, L, p_dev, d_dev, stats_dev, spg_state, spg_mode, \
                                  trial_scale, trial_xnew, trial_m, trial_enable)
    if (L <= SIMPLEX_BLOCK * 4) SP(4);
    else if (L <= SIMPLEX_BLOCK * 12) SP(12);
    else if (L <= SIMPLEX_BLOCK * 24) SP(24);
    else if (L <= SIMPLEX_BLOCK * 48) SP(48);
    else SP(0);
#undef SP
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_simplex_workspace_doubles(int64_t L, int64_t *n)
{
    if (!n || L <= 0) return fail(BLUEST_ERR_ARG, "bad argument");
    *n = ProjWs::total(L, (int)((L + 1023) / 1024));
    return BLUEST_OK;
}

extern "C" int bluest_simplex_project(const double *x_dev, const double *g_dev, double lambda, double z, double floor, int64_t L,
                                      double *p_dev, double *d_dev, double *stats_dev, double *work_dev, void *stream)
{
    return simplex_impl(x_dev, g_dev, lambda, z, floor, L, p_dev, d_dev, stats_dev, nullptr, 0, stream, nullptr, nullptr, nullptr, nullptr,
                        work_dev);
}

// ---- device-resident SPG (state in HBM, control flow by predication; see include/bluest_hip.h Part 3) -----------
extern "C" int bluest_plan_set_gate(bluest_plan_t plan, const int32_t *enable_dev, int always_v)
{
    if (!plan) return fail(BLUEST_ERR_ARG, "plan is NULL");
    plan->gate = enable_dev;
    plan->always_v = always_v != 0;
    return BLUEST_OK;
}

extern "C" int bluest_plan_v_workspace(bluest_plan_t plan, const double **v_dev, const int32_t **status_dev)
{
    if (!plan || !plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    if (v_dev) *v_dev = plan->d_v;
    if (status_dev) *status_dev = plan->d_status;
    return BLUEST_OK;
}

extern "C" int bluest_spg_direction(const double *x_dev, const double *g_dev, double *state_dev, double z, double floor, int64_t L,
                                    double *d_dev, const double *scale_dev, double *xnew_dev, double *m_dev, int32_t *enable_dev,
                                    double *work_dev, void *stream)
{
    if (!state_dev || !g_dev || !d_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (xnew_dev && (!scale_dev || !m_dev || !enable_dev)) return fail(BLUEST_ERR_ARG, "first-trial outputs need scale, m and enable");
    return simplex_impl(x_dev, g_dev, 0.0, z, floor, L, nullptr, d_dev, state_dev + SPG_GD, state_dev, 1, stream, scale_dev, xnew_dev,
                        xnew_dev ? m_dev : nullptr, xnew_dev ? enable_dev : nullptr, work_dev);
}

extern "C" int bluest_spg_converged(const double *x_dev, const double *g_dev, double *state_dev, double z, double floor, int64_t L,
                                    double *work_dev, void *stream)
{
    if (!state_dev || !g_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    return simplex_impl(x_dev, g_dev, 1.0, z, floor, L, nullptr, nullptr, state_dev + SPG_GPSTATS, state_dev, 2, stream, nullptr, nullptr,
                        nullptr, nullptr, work_dev);
}

extern "C" int bluest_spg_trial(const double *x_dev, const double *d_dev, const double *scale_dev, const double *state_dev,
                                double *xnew_dev, double *m_dev, int32_t *enable_dev, int64_t L, void *stream)
{
    int rc = require_gpu(); if (rc) return rc;
    if (!x_dev || !d_dev || !scale_dev || !state_dev || !xnew_dev || !m_dev || !enable_dev || L <= 0) return fail(BLUEST_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_spg_trial, dim3((unsigned)((L + 1023) / 1024)), dim3(1024), 0, (hipStream_t)stream, x_dev, d_dev, scale_dev,
                       state_dev, xnew_dev, m_dev, enable_dev, L);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_spg_decide(double *state_dev, const double *var_dev, const int32_t *status_dev, int n_out, int last_slot,
                                 int32_t *enable_dev, void *stream)
{
    int rc = require_gpu(); if (rc) return rc;
    if (!state_dev || !var_dev || !status_dev || !enable_dev || n_out <= 0 || n_out > SPG_MAX_OUT) return fail(BLUEST_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_spg_decide, dim3(1), dim3(64), 0, (hipStream_t)stream, state_dev, var_dev, status_dev, n_out, last_slot, enable_dev);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_spg_update_fused(bluest_plan_t plan, double *x_dev, double *g_dev, const double *xnew_dev, const double *grad_dev,
                                       const double *scale_dev, double *state_dev, double floor, double *work_dev, void *stream)
{
    int rc = plan_ready(plan, 1); if (rc) return rc;
    if (!x_dev || !g_dev || !xnew_dev || !grad_dev || !scale_dev || !state_dev || !work_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    const int64_t L = plan->L;
    // 256-thread workgroups: at K_tot = 21699 that is 85 compute units instead of 22 for a kernel made of dependent gathers
    const int nblocks = (int)std::min<int64_t>((L + 255) / 256, SPG_UPD_BLOCKS_MAX);
    hipLaunchKernelGGL(k_spg_update_a_fused, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, x_dev, g_dev, xnew_dev, grad_dev, plan->d_goff,
                       plan->d_invmap, (int)plan->outs.size(), scale_dev, state_dev, floor, L, (double2 *)work_dev, plan->identity ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_spg_finish(bluest_plan_t plan, const double *v_dev, const int32_t *status_dev, double *x_dev, double *g_dev,
                                 const double *xnew_dev, double *grad_dev, const double *scale_dev, double *state_dev, double floor,
                                 double *work_dev, void *stream)
{
    int rc = plan_ready(plan, 1); if (rc) return rc;
    if (!v_dev || !status_dev || !x_dev || !g_dev || !xnew_dev || !grad_dev || !scale_dev || !state_dev || !work_dev)
        return fail(BLUEST_ERR_ARG, "null pointer");
    if (plan->L <= 4096 && plan->n_tiles <= 1024) {
        int kmax = 0;
        for (const auto &od : plan->outs) kmax = std::max(kmax, od.K);
        // 1024 threads: the tile rounds dominate (measured 13.0 / 14.1 / 18.0 us with 1024 / 512 / 256 threads)
#define LFS(KU) hipLaunchKernelGGL((k_spg_finish_small<KU>), dim3(1), dim3(1024), 0, (hipStream_t)stream, plan->d_tiles, plan->n_tiles, \
                                   plan->d_tvals, v_dev, status_dev, plan->N, (int)plan->outs.size(), grad_dev, x_dev, g_dev,    \
                                   xnew_dev, plan->d_goff, plan->d_invmap, scale_dev, state_dev, floor, plan->L)
        if (kmax <= 5) LFS(5);
        else if (kmax <= 8) LFS(8);
        else LFS(12);
#undef LFS
        HIP_TRY(hipGetLastError());
        return BLUEST_OK;
    }
    rc = bluest_plan_grad(plan, v_dev, status_dev, 1, grad_dev, plan->grad_len, stream);
    if (rc) return rc;
    return bluest_spg_update_fused(plan, x_dev, g_dev, xnew_dev, grad_dev, scale_dev, state_dev, floor, work_dev, stream);
}

extern "C" int bluest_spg_update(double *x_dev, double *g_dev, const double *xnew_dev, const double *gnew_dev, double *state_dev,
                                 double floor, int64_t L, double *work_dev, void *stream)
{
    int rc = require_gpu(); if (rc) return rc;
    if (!x_dev || !g_dev || !xnew_dev || !gnew_dev || !state_dev || !work_dev || L <= 0) return fail(BLUEST_ERR_ARG, "bad argument");
    const int nblocks = (int)std::min<int64_t>((L + 1023) / 1024, SPG_UPD_BLOCKS_MAX);
    hipLaunchKernelGGL(k_spg_update_a, dim3(nblocks), dim3(1024), 0, (hipStream_t)stream, x_dev, g_dev, xnew_dev, gnew_dev, state_dev, floor, L,
                       (double2 *)work_dev);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}


// A whole window of SPG iterations enqueued by ONE host call: direction (+ fused first trial point), T line-search slots
// (trial point, Phi pass, solve + decision), gradient + update; the last iteration optionally followed by the convergence
// projection.  Plain stream launches -- measured against replaying a captured hipGraph of the same sequence: the graph saves
// ~1 us of dispatch per kernel, but capturing 18 graphs costs 13-55 ms per solve and DESTROYING them 60-80 ms, which the
// runtime processes asynchronously inside whatever synchronises next (the "teardown stall" that kept landing in the next
// set-up or solve).  20 iterations x 6-7 launches x ~2.5 us of host time stay well below the ~0.7-1.4 ms the GPU needs for them.
extern "C" int bluest_spg_window(bluest_plan_t plan, double *x_dev, double *g_dev, double *d_dev, double *xnew_dev, double *m_dev,
                                 const double *scale_dev, double *state_dev, double *var_dev, int32_t *status_dev, double *grad_dev,
                                 int32_t *enable_dev, double *work_dev, double *proj_work_dev, const double *v_ws_dev,
                                 double floor, int slots, int n_iterations, int check_last, void *stream)
{
    int rc = plan_ready(plan, 1); if (rc) return rc;
    if (!x_dev || !g_dev || !d_dev || !xnew_dev || !m_dev || !scale_dev || !state_dev || !var_dev || !status_dev || !grad_dev ||
        !enable_dev || !work_dev || !v_ws_dev)
        return fail(BLUEST_ERR_ARG, "null pointer");
    if (slots < 1 || slots > 8 || n_iterations < 0) return fail(BLUEST_ERR_ARG, "slots=%d, n_iterations=%d out of range", slots, n_iterations);
    const int64_t L = plan->L;
    // small plans: the gradient tiles of the accepted point run inside the single-workgroup finishing kernel; large plans: every
    // evaluation is the fused solve + gradient launch (decision in its tail), so the update follows the line search directly
    const bool small = plan->L <= 4096 && plan->n_tiles <= 1024;
    for (int it = 0; it < n_iterations; it++) {
        if ((rc = bluest_spg_direction(x_dev, g_dev, state_dev, 1.0, floor, L, d_dev, scale_dev, xnew_dev, m_dev, enable_dev, proj_work_dev, stream))) return rc;
        for (int t = 0; t < slots; t++) {
            const int last = t == slots - 1 ? 1 : 0;
            if (t > 0 && (rc = bluest_spg_trial(x_dev, d_dev, scale_dev, state_dev, xnew_dev, m_dev, enable_dev, L, stream))) return rc;
            if (small) rc = bluest_plan_eval_decide(plan, m_dev, 0.0, var_dev, status_dev, state_dev, last, enable_dev, stream);
            else rc = bluest_plan_eval_grad_decide(plan, m_dev, 0.0, var_dev, grad_dev, status_dev, state_dev, last, enable_dev, stream);
            if (rc) return rc;
        }
        if (small) rc = bluest_spg_finish(plan, v_ws_dev, status_dev, x_dev, g_dev, xnew_dev, grad_dev, scale_dev, state_dev, floor, work_dev, stream);
        else rc = bluest_spg_update_fused(plan, x_dev, g_dev, xnew_dev, grad_dev, scale_dev, state_dev, floor, work_dev, stream);
        if (rc) return rc;
    }
    if (check_last && (rc = bluest_spg_converged(x_dev, g_dev, state_dev, 1.0, floor, L, proj_work_dev, stream))) return rc;
    return BLUEST_OK;
}
