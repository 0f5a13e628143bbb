// newton.hip -- Part 6 of include/bluest_hip.h: the second-order finish of solver="spg".
//
//   k_ma_update      one step of the multiplicative algorithm on the full problem (phase 1: locates the support)
//   k_master_newton  active-set Newton (SQP, Levenberg-Marquardt damped) on a support of <= 64 groups: ONE workgroup runs the
//                    whole solve -- evaluations, Hessian, KKT system, step control -- out of LDS
//   k_price          reduced costs of ALL groups at the priced point + the dual bound + the most violating candidates
//   k_support_point  allocation vector of the full problem from a support vector (+ the uniform background)
//
// The algorithm is restated in numpy by the test infrastructure; tests compare the two.
// The reference hands this problem to third-party NLP solvers with the Hessian of bluest/misc.py:497-503 /
// bluest/cmisc.cpp:74-97 (bluest/sap.py:387-456); there is no reference code to follow.
#include "plan.hpp"     // brings solve.hpp: readlane_f64, wave_lds_sync

#ifdef MASTER_TIMING      // experiment build only: per-phase wall-clock (100 MHz counter) of thread 0, accumulated in L.scal[240..248] (as
                          // 64-bit integers; slot 8 = the last stamp) and returned in out[8..15] -- the record's damping / multiplier
                          // slots are overwritten, nothing on the host reads them.  Phases: 0 load, 1 Phi assembly, 2 elimination of the
                          // evaluations, 3 active set + derivatives, 4 free set / step formation / bookkeeping, 5 Hessian, 6 elimination of
                          // the Newton system, 7 K + the small KKT system
#define TSTAMP(k) do { __syncthreads(); if (threadIdx.x == 0) { long long *tq_ = reinterpret_cast<long long *>(L.scal + 240); const long long t_ = wall_clock64(); tq_[k] += t_ - tq_[12]; tq_[12] = t_; } } while (0)
#else
#define TSTAMP(k)
#endif
#define MASTER_THREADS 512
#define MASTER_SMAX 64
#ifndef MASTER_PACT
#define MASTER_PACT 8      // candidate outputs (within act_tol of the maximum) of one step of the master.  6 until round 4: with 7 or 8
                           // outputs tied at the optimum (headline shape, other covariance seeds) the step cannot equalise them and the
                           // master ends at a KKT measure of 3e-5, certified gap 1.4e-5 -- profiles/r04_seed_sweep.txt.  The scal[] blocks
                           // SQ / SQT / SMU (8 slots each) and the 16-lane KKT system (8 + MASTER_MCAP + 2 rows) are sized for 8
#endif
#define MASTER_MCAP 4      // sample caps (max_model_samples rows) that can be in the step as equality rows at once
#define MASTER_NE (MASTER_PACT + MASTER_MCAP + 1)      // columns of E: active outputs, active caps, the simplex row
#define MASTER_OUT 16      // doubles in front of r[] in the result record
#ifndef MASTER_DAMP_DOWN
#define MASTER_DAMP_DOWN 0.3   // damping after a step whose decrease was more than half of the predicted one.  x0.1 (until round 4) made
                               // the masters oscillate -- accepted, x0.1, the next step too long and rejected, x10, accepted ... : 45 of 116
                               // factorisations of a headline solve ended in a rejected step; profiles/r04_damp_ab.txt (tools/damp_ab.sh)
#endif
#ifndef MASTER_DAMP_HOLD
#define MASTER_DAMP_HOLD 0     // 1: a step that was accepted only after a rejection keeps its damping for the next iteration (A/B switch: with
                               // x4 after a rejection it is 0-3 % faster on the synthetic shapes and takes the ill-conditioned Navier-Stokes problem's
                               // certified gap to 6e-5 under perturbed parameters -- profiles/r04_damp_ab.txt; not shipped)
#endif
#ifndef MASTER_DAMP_UP
#define MASTER_DAMP_UP 10.0    // ... after a rejected step
#endif
#define MASTER_STATIC_LDS 2048   // room kept for the kernel's static LDS (the small KKT system: 1.5 KB) next to the dynamic part

struct MasterArgs {
    int N, n_out, S, KM;               // models, outputs, support size (<= 64), largest group size in the support
    double eps_bg, tol, act_tol, floor_x;
    double fb;                         // > 0: fraction-to-the-boundary rule (no entry reaches zero: the polish without background)
    int maxit;
    const double *const *invcov;       // [n_out] reference-layout pseudo-inverses of every output (plan memory)
    const int64_t *boff;               // [n_out * S] offset (doubles) of block (o, j) inside invcov[o]; -1: output o lacks group j
    const double *cc;                  // [S] B / w_j
    const uint8_t *idx;                // [S * KM] model indices of the support groups (padded)
    const int32_t *kk;                 // [S] group sizes
    const double *s;                   // [n_out] output scales
    const double *bg;                  // [n_out * N * N] eps_bg * Phi_o(uniform allocation), or NULL (eps_bg == 0)
    double *x;                         // [S] in: start (>= 0), out: solution (sum 1)
    double *mu;                        // [n_out] in: multipliers of the previous master (or < 0: none), out: multipliers
    double *out;                       // [MASTER_OUT + n_out]: F, lam, kkt, spread, it, evals, solves, status, damp, ..., r[n_out]
    int ncap;                          // per-model sample caps (bluest/sap.py:222-240): sum_{j: model in group j} (1-eps) cc_j x_j <= cap_b
    const int32_t *cap_model;          // [ncap] the capped model
    const double *cap_b;               // [ncap] right-hand side in the scaled variable (the background's share already taken off)
    double *nu;                        // [ncap] out: cap multipliers (>= 0)
};

// register size of the per-output elimination: the matrix of an output is kept REVERSED (solve.hpp: position p = model NT-1-p,
// row stride NT + 2) so that the DPP elimination of the evaluation reads its row as NT contiguous doubles; 0 = more than 32
// models: natural layout, elimination out of LDS
// nt: the register tile of the instantiation (12 .. 64: Phi reversed with stride nt + 2), or 0 (natural layout, eliminations in LDS)
__host__ __device__ constexpr int master_phi_doubles(int N, int nt) { return nt ? nt * (nt + 2) : N * (N + 1); }

struct MasterLds {                     // carved out of dynamic LDS by master_carve()
    double *PHI, *TACT, *BLK, *M, *AAC, *GQ;
    double *x, *xt, *xp, *d, *cc, *Dm, *glv, *mvec, *hd, *r, *rt, *rp, *mu, *mup, *muh, *scal, *capb, *capslack, *nu;      // hd: undamped diagonal of the iteration's Hessian (free-set order)
    double *fcol;                      // [2][72]: multiplier column + pivot of the factorisation's current step (double-buffered)
    int *kk, *fi, *act, *istate, *capmodel, *actc;
    signed char *pos;
    unsigned char *idx;
    unsigned long long *memb;          // [N] bit j: support group j contains the model
    unsigned short *dl_off, *dl_ab, *dl_ent;   // destination lists of the Phi assembly: offsets [ND + 1], (a | b << 8) [ND], entries j | e << 7
    int LDN, LDM, KE, ND, PHS;         // PHS: doubles per output in PHI
};

__host__ __device__ inline size_t master_lds_bytes(int N, int n_out, int S, int KM, int nt)
{
    const size_t LDN = N + 1, LDM = (S + MASTER_NE + 1) | 1, KE = (size_t)KM * (KM + 1) / 2, ND = (size_t)N * (N + 1) / 2;
    const size_t PA = n_out < MASTER_PACT ? n_out : MASTER_PACT;      // active outputs there can be: T and the a_{o,j} are kept for those only
    size_t d = (size_t)n_out * master_phi_doubles(N, nt) + PA * N * LDN + (size_t)S * n_out * KE + (size_t)S * LDM +
               PA * S * KM + (size_t)S * MASTER_PACT + 9 * (size_t)S + 6 * (size_t)n_out + 256 + 3 * 64 + (size_t)N + 2 * 72;
    size_t bytes = d * sizeof(double) + (3 * (size_t)S + 2 * MASTER_PACT + 48 + 64 + 2 * MASTER_MCAP) * sizeof(int) + (size_t)S * N + (size_t)S * KM + 64;
    bytes = (bytes + 7) & ~(size_t)7;
    bytes += (2 * ND + 2 + (size_t)S * KE + 8) * sizeof(unsigned short);
    return (bytes + 15) & ~(size_t)15;
}

__device__ inline void master_carve(MasterLds &L, unsigned char *base, int N, int n_out, int S, int KM, int nt)
{
    L.LDN = N + 1; L.LDM = (S + MASTER_NE + 1) | 1; L.KE = KM * (KM + 1) / 2; L.ND = N * (N + 1) / 2; L.PHS = master_phi_doubles(N, nt);
    double *p = reinterpret_cast<double *>(base);
    L.PHI = p;  p += (size_t)n_out * L.PHS;
    const size_t PA = n_out < MASTER_PACT ? n_out : MASTER_PACT;
    L.TACT = p; p += PA * N * L.LDN;
    L.BLK = p;  p += (size_t)S * n_out * L.KE;
    L.M = p;    p += (size_t)S * L.LDM;
    L.AAC = p;  p += PA * S * KM;
    L.GQ = p;   p += (size_t)S * MASTER_PACT;
    L.x = p; p += S; L.xt = p; p += S; L.xp = p; p += S; L.d = p; p += S; L.cc = p; p += S; L.Dm = p; p += S; L.glv = p; p += S; L.mvec = p; p += S; L.hd = p; p += S;
    L.r = p; p += n_out; L.rt = p; p += n_out; L.rp = p; p += n_out; L.mu = p; p += n_out; L.mup = p; p += n_out; L.muh = p; p += n_out;
    L.scal = p; p += 256;
    L.capb = p; p += 64; L.capslack = p; p += 64; L.nu = p; p += 64;
    L.fcol = p; p += 2 * 72;
    L.memb = reinterpret_cast<unsigned long long *>(p); p += N;
    int *q = reinterpret_cast<int *>(p);
    L.kk = q; q += S; L.fi = q; q += S; L.act = q; q += 2 * MASTER_PACT; L.istate = q; q += 48;      // act: current list, then the iteration's list (act0)
    L.capmodel = q; q += 64; L.actc = q; q += 2 * MASTER_MCAP;                                       // actc: current caps of the step, then the iteration's (actc0)
    q += S;     // spare
    L.pos = reinterpret_cast<signed char *>(q);
    L.idx = reinterpret_cast<unsigned char *>(L.pos + (size_t)S * N);
    size_t used = (size_t)((L.idx + (size_t)S * KM) - base);
    used = (used + 7) & ~(size_t)7;
    L.dl_off = reinterpret_cast<unsigned short *>(base + used);
    L.dl_ab = L.dl_off + L.ND + 1;
    L.dl_ent = L.dl_ab + L.ND + 1;
}

// scal[] slots
enum { SC_F = 0, SC_LAMEST, SC_DAMP, SC_TAU, SC_LAM, SC_PRED, SC_KKT, SC_SPREAD, SC_QMAX, SC_FT, SC_LAMX, SC_MBEST, SC_FBEST, SC_KKTBEST, SC_SPREADBEST };
// istate[] slots
enum { IS_NACT = 0, IS_NF, IS_OK, IS_DONE, IS_ACCEPT, IS_IT, IS_EVALS, IS_SOLVES, IS_STATUS, IS_NACT0, IS_NOISE, IS_NALIVE, IS_NFAIL, IS_ALIVE /* .. +PACT */,
       IS_NCACT0 = IS_ALIVE + MASTER_PACT, IS_NCALIVE, IS_CALIVE /* .. +MCAP */, IS_CLOCK = IS_CALIVE + MASTER_MCAP /* .. +MCAP */,
       IS_OLOCK = IS_CLOCK + MASTER_MCAP /* .. +PACT */, IS_COUNT = IS_OLOCK + MASTER_PACT };
// scal[] blocks: SQ.. q of the act0 outputs, SQT.. q at the trial point, SMU.. new multipliers (act0 order), SNU.. new cap
// multipliers (actc0 order), SCK.. K = E^T M^-1 E (MASTER_NE x MASTER_NE)
#define SQ 16
#define SQT 24
#define SMU 32
#define SNU 40
#define SCK 64

// coefficient of support entry j in cap c: (1 - eps) cc_j if the capped model is in group j
__device__ __forceinline__ double cap_a(const MasterArgs &A, const MasterLds &L, int c, int j)
{
    return L.pos[j * A.N + L.capmodel[c]] >= 0 ? (1.0 - A.eps_bg) * L.cc[j] : 0.0;
}

// packed symmetric index of (l, l2), l <= l2, in a k x k block
__device__ __forceinline__ int sym_e(int l, int l2, int k) { return l * k - l * (l - 1) / 2 + (l2 - l); }

// address of entry (a, b) of one output's information matrix in L.PHI (reversed with stride NT + 2 for the register
// eliminations, natural with stride N + 1 beyond 32 models)
template <int NT>
__device__ __forceinline__ int phi_idx(int a, int b, int LDN)
{
    if constexpr (NT > 0) return (NT - 1 - a) * (NT + 2) + (NT - 1 - b);
    else return a * LDN + b;
}

// in-place Gauss-Jordan inverse of the SPD matrix P (N x N, row stride LDN, LDS, natural order) by one wavefront out of LDS,
// lane = row: more than 32 models only (the register versions below serve the usual sizes).
__device__ __forceinline__ bool inverse_wave_lds(double *P, int N, int LDN, int lane)
{
    const bool mine = lane < N;
    bool bad = false;
    bool untouched = false;
    if (mine && !(P[lane * LDN + lane] > 0.0)) { untouched = true; P[lane * LDN + lane] = 1.0; }
    wave_lds_sync();
    for (int p = 0; p < N; p++) {
        const double piv = P[p * LDN + p];
        if (!(piv > 0.0) || !isfinite(piv)) { bad = true; break; }            // wave-uniform (same address)
        const double rinv = 1.0 / piv;
        if (mine && lane != p) {
            const double f = P[lane * LDN + p] * rinv;
            for (int c0 = 0; c0 < N; c0 += 8) {                               // loads first, then the updates: 16 LDS reads in flight
                double u[8], w[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { const int c = c0 + q < N ? c0 + q : N - 1; u[q] = P[p * LDN + c]; w[q] = P[lane * LDN + c]; }
#pragma unroll
                for (int q = 0; q < 8; q++) if (c0 + q < N && c0 + q != p) P[lane * LDN + c0 + q] = fma(-f, u[q], w[q]);
            }
            P[lane * LDN + p] = -f;
        }
        wave_lds_sync();
        if (lane == p) {
            for (int c = 0; c < N; c++)
                if (c != p) P[p * LDN + c] *= rinv;
            P[p * LDN + p] = rinv;
        }
        wave_lds_sync();
    }
    if (untouched) P[lane * LDN + lane] = 0.0;
    wave_lds_sync();
    return bad;
}

// T = P^-1 of one output by one wavefront, row `lane` (= model) in NT registers: P is read from the reversed layout of L.PHI and
// left alone, T is written in natural order (row stride LDN) -- the matrix the Hessian and the gradients of an iteration are made
// from.  Gauss-Jordan without pivoting.  Per step the pivot lane first scales its own row by 1/piv (one multiply per entry under
// its EXEC mask), then every row takes a_c <- fma(g, u_c, a_c) with u_c the scaled pivot row's entry (v_readlane) and g = -a_p
// (0 for the pivot lane, whose row is final), and a_p <- 1/piv or g / piv: four instructions per entry instead of a multiply, an fma
// and two 64-bit selects.  (Folding the pivot row into the same fma as a + (1/piv - 1) a costs the low bits of 1/piv whenever the
// pivot is large -- relative error eps * piv in T: the master then stalls at a KKT residual of 1e-8.)
// A model no group of the support touches (no background) has a zero row: identity in, zero out.
// (The pivot loop is a template recursion: left as a loop, `#pragma unroll` gives up on it at NT = 32 -- 32 steps of 3 NT
// instructions --, the row is then indexed dynamically and lives in SCRATCH memory: 153 us per inverse instead of 5.)
template <int NT, int P>
struct InvRegsStep {
    static __device__ __forceinline__ void run(double (&a)[NT], int lane, bool &bad)
    {
        const double piv = readlane_f64(a[P], P);
        bad = bad || !(piv > 0.0) || !isfinite(piv);
        const double rinv = rcp_f64(piv);
        const bool is = lane == P;
        if (is) {
#pragma unroll
            for (int c = 0; c < NT; c++) if (c != P) a[c] *= rinv;
        }
        const double g = is ? 0.0 : -a[P];
        constexpr int CH = NT <= 20 ? NT : 16;          // broadcasts in flight at once (the row itself takes 2 NT registers)
#pragma unroll
        for (int c0 = 0; c0 < NT; c0 += CH) {
            double u[CH];
#pragma unroll
            for (int q = 0; q < CH; q++) if (c0 + q < NT && c0 + q != P) u[q] = readlane_f64(a[c0 + q], P);
#pragma unroll
            for (int q = 0; q < CH; q++) if (c0 + q < NT && c0 + q != P) a[c0 + q] = fma(g, u[q], a[c0 + q]);
        }
        a[P] = is ? rinv : g * rinv;
        InvRegsStep<NT, P + 1>::run(a, lane, bad);
    }
};
template <int NT>
struct InvRegsStep<NT, NT> { static __device__ __forceinline__ void run(double (&)[NT], int, bool &) {} };

template <int NT>
__device__ __forceinline__ bool inverse_regs(const double *P, double *T, int N, int LDN, int lane)
{
    static_assert(NT > 0, "register inverse");
    constexpr int LDP = NT + 2;
    const bool mine = lane < N;
    double a[NT];
    const double *row = P + (NT - 1 - (mine ? lane : 0)) * LDP;
#pragma unroll
    for (int c = 0; c < NT; c++) a[c] = (mine && c < N) ? row[NT - 1 - c] : ((c == lane) ? 1.0 : 0.0);
    bool untouched = false;
#pragma unroll
    for (int c = 0; c < NT; c++) if (c == lane && mine && !(a[c] > 0.0)) { untouched = true; a[c] = 1.0; }
    bool bad = false;
    InvRegsStep<NT, 0>::run(a, lane, bad);
#pragma unroll
    for (int c = 0; c < NT; c++) if (c == lane && untouched) a[c] = 0.0;
    if (mine)
#pragma unroll
        for (int c = 0; c < NT; c++) if (c < N) T[lane * LDN + c] = a[c];
    return bad;
}

// arg-extremum over lanes 0..15 (one DPP row) of (value, lane) pairs; MAXI: largest value, else smallest; ties go to the lower
// lane.  Every lane of the row ends with the winner.
template <bool MAXI, int CTRL>
__device__ __forceinline__ void row16_pick_step(double &best, int &who)
{
    const double b2 = dpp_f64<CTRL>(best);
    const int w2 = __builtin_amdgcn_mov_dpp(who, CTRL, 0xf, 0xf, true);
    const bool take = MAXI ? (b2 > best || (b2 == best && w2 < who)) : (b2 < best || (b2 == best && w2 < who));
    best = take ? b2 : best;
    who = take ? w2 : who;
}
template <bool MAXI>
__device__ __forceinline__ void row16_pick(double &best, int &who)
{
    row16_pick_step<MAXI, 0xB1>(best, who);
    row16_pick_step<MAXI, 0x4E>(best, who);
    row16_pick_step<MAXI, 0x141>(best, who);
    row16_pick_step<MAXI, 0x140>(best, who);
}

// x of the lane whose index within its quad differs in bit 0 (CTRL 0xB1 = quad_perm:[1,0,3,2]) or bit 1 (0x4E = quad_perm:[2,3,0,1]):
// a DPP move per 32-bit half, no LDS crossbar (ds_bpermute) involved
template <int CTRL>
__device__ __forceinline__ double quad_xor(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// forward elimination beyond 32 models (no DPP: a DPP row has 16 lanes), rows in registers, only what the LAST pivot needs: the
// elimination of solve.hpp's gj_regs as a template recursion with the broadcasts in chunks of 16 (a[NT] and NT broadcasts at once
// do not fit the 256 registers a wavefront of this 512-thread kernel has)
template <int NT, int J>
struct Inv00Step {
    static __device__ __forceinline__ void run(double (&a)[NT], int lane, double floor_mine, int &flag, double &last_pivot)
    {
        const double piv = readlane_f64(a[J], J);
        const bool is = lane == J;
        flag = is ? (!(a[J] > floor_mine) ? 1 : 0) : flag;        // NaN and non-positive pivots included
        if constexpr (J == NT - 1) { last_pivot = piv; }
        else {
            const double f = is ? 0.0 : -a[J] * rcp_f64(piv);
#pragma unroll
            for (int c0 = J + 1; c0 < NT; c0 += 16) {
                double u[16];
#pragma unroll
                for (int q = 0; q < 16; q++) if (c0 + q < NT) u[q] = readlane_f64(a[c0 + q], J);
#pragma unroll
                for (int q = 0; q < 16; q++) if (c0 + q < NT) a[c0 + q] = fma(f, u[q], a[c0 + q]);
            }
            Inv00Step<NT, J + 1>::run(a, lane, floor_mine, flag, last_pivot);
        }
    }
};

// (P^-1)_00 of one output by one wavefront: the DPP elimination of the plan's solve (solve.hpp: gj_solve_last without the
// solution vector), rows in registers, P (reversed layout) left alone.  Returns +inf when P is not positive definite.
template <int NT>
__device__ __forceinline__ double inv00_regs(const double *P, int N, int lane)
{
    static_assert(NT > 0, "register elimination");
    using G = GjMap<NT>;
    constexpr int LDP = NT + 2;
    const int p = G::pos_of(lane);
    const int model = NT - 1 - p;
    const double *row = P + p * LDP;
    double a[NT];
#pragma unroll
    for (int c = 0; c < NT; c += 2) {
        const double2 x = *reinterpret_cast<const double2 *>(row + c);
        a[c] = x.x; a[c + 1] = x.y;
    }
    // pads (models that do not exist) and models no group touches: identity row (their rows and columns are zero)
    double diag0 = 1.0;
    bool live = false;
#pragma unroll
    for (int c = 0; c < NT; c++) {
        if (c == p) {
            live = model < N && a[c] > 0.0;
            a[c] = live ? a[c] : 1.0;
            diag0 = a[c];
        }
    }
    if (!((__ballot(live) >> G::lane_of(NT - 1)) & 1ull)) return INFINITY;      // model 0 itself is not sampled (wave-uniform)
    double rinv_mine = 0.0, last_pivot = 1.0;
    if constexpr (!G::dpp) {
        int bad0 = 0;
        Inv00Step<NT, 0>::run(a, lane, BLUEST_PIVOT_TOL * diag0, bad0, last_pivot);
        bad0 = (__ballot(bad0 != 0 && lane < NT) != 0ull || !isfinite(last_pivot)) ? 1 : 0;
        return (!bad0 && last_pivot > 0.0) ? 1.0 / last_pivot : INFINITY;
    } else {
        asm volatile("s_nop 4" ::: "memory");
        GjExtraStep<NT, 0>::run(a, lane, rinv_mine);
        GjDppStep<NT, G::E>::run(a, lane, rinv_mine, last_pivot);
        last_pivot = readlane_f64(last_pivot, 0);
        const bool ok = rinv_mine > 0.0 && rinv_mine * (BLUEST_PIVOT_TOL * diag0) < 1.0;     // false for NaN as well
        const bool bad = __ballot(!ok && lane < G::n_lanes) != 0ull || !isfinite(last_pivot) || !(last_pivot > 0.0);
        return bad ? INFINITY : 1.0 / last_pivot;
    }
}

// r[o] = V_o / s_o of the allocation with support vector xv (LDS) -- all threads.  Leaves Phi_o in L.PHI (NT > 0: the matrix
// itself, for inverse_regs; beyond 32 models T_o = Phi_o^-1 on the touched models, zero elsewhere).  Fixed summation orders: the
// result is bit-reproducible (every rank of a sharded solve runs this redundantly and must get the same bits).
// Assembly: every symmetric destination (a <= b) has the list of its contributions (support group, packed entry) made at load
// time; a QUAD of lanes shares a destination -- lane q takes entries q, q + 4, .. for all outputs, a transposing butterfly over the
// quad combines them -- so the destinations of model 0 (a member of nearly every group: lists as long as the support) cost 16
// entries per lane instead of one dependent chain of 64 look-ups, and 128 destinations are served per pass.
template <int NT>
__device__ void master_eval(const MasterArgs &A, MasterLds &L, const double *xv, double *rout, int tid)
{
    const int N = A.N, n_out = A.n_out, S = A.S, LDN = L.LDN, KE = L.KE, ND = L.ND, PHS = L.PHS;
    const int wave = tid >> 6, lane = tid & 63, nw = MASTER_THREADS / 64;
    // the background's share first: all its loads are in flight together (one memory round trip per evaluation; loaded where
    // the sums are stored, every pass of the loop below waited for its own)
    if (A.bg)
        for (int t0 = tid; t0 < n_out * N * N; t0 += 8 * MASTER_THREADS) {
            double v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { const int t = t0 + i * MASTER_THREADS; v[i] = t < n_out * N * N ? A.bg[t] : 0.0; }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int t = t0 + i * MASTER_THREADS;
                if (t < n_out * N * N) {
                    const int o = t / (N * N), rem = t - o * N * N;
                    L.PHI[(size_t)o * PHS + phi_idx<NT>(rem / N, rem % N, LDN)] = v[i];
                }
            }
        }
    for (int j = tid; j < S; j += MASTER_THREADS) L.mvec[j] = (1.0 - A.eps_bg) * L.cc[j] * xv[j];
    __syncthreads();
#ifdef MASTER_TIMING_FINE
    TSTAMP(8);
#endif
    const int sub = tid & 3;
    for (int d0 = 0; d0 < ND; d0 += MASTER_THREADS / 4) {
        const int d = d0 + (tid >> 2);
        const bool valid = d < ND;
        const int beg = valid ? L.dl_off[d] : 0, end = valid ? L.dl_off[d + 1] : 0;
        const unsigned ab = valid ? L.dl_ab[d] : 0u;
        const int ia = phi_idx<NT>((int)(ab & 0xffu), (int)(ab >> 8), LDN), ib = phi_idx<NT>((int)(ab >> 8), (int)(ab & 0xffu), LDN);
        for (int o0 = 0; o0 < n_out; o0 += 8) {
            const int om = o0 + 2 * sub;                      // this lane stores outputs om and om + 1
            double *P0 = L.PHI + (size_t)om * PHS, *P1 = P0 + PHS;
            const bool st0 = valid && om < n_out, st1 = valid && om + 1 < n_out;
            const double base0 = (A.bg && st0) ? P0[ia] : 0.0, base1 = (A.bg && st1) ? P1[ia] : 0.0;      // the background's share (in flight)
            double acc[8];
#pragma unroll
            for (int q = 0; q < 8; q++) acc[q] = 0.0;
            for (int i = beg + sub; i < end; i += 4) {
                const unsigned ent = L.dl_ent[i];
                const int j = (int)(ent & 127u), e = (int)(ent >> 7);
                const double mj = L.mvec[j];
                const double *B = L.BLK + (size_t)j * n_out * KE + e;
#pragma unroll
                for (int q = 0; q < 8; q++) if (o0 + q < n_out) acc[q] = fma(mj, B[(size_t)(o0 + q) * KE], acc[q]);
            }
            // the quad's partial sums of up to eight outputs -> lane `sub` ends with the totals of outputs o0 + 2 sub, + 1: a transposing
            // butterfly (4 + 2 exchanges; every lane folds the half it keeps and hands over the other), DPP quad permutes, fixed order
            const bool h2 = (sub & 2) != 0, h1 = (sub & 1) != 0;
            double k4[4], k2[2];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double give = h2 ? acc[i] : acc[4 + i], keep = h2 ? acc[4 + i] : acc[i];
                k4[i] = keep + quad_xor<0x4E>(give);
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const double give = h1 ? k4[i] : k4[2 + i], keep = h1 ? k4[2 + i] : k4[i];
                k2[i] = keep + quad_xor<0xB1>(give);
            }
            if (st0) { const double v = k2[0] + base0; P0[ia] = v; P0[ib] = v; }
            if (st1) { const double v = k2[1] + base1; P1[ia] = v; P1[ib] = v; }
        }
    }
    __syncthreads();
    TSTAMP(1);
    for (int o = wave; o < n_out; o += nw) {
        double *P = L.PHI + (size_t)o * PHS;
        double v00;
        if constexpr (NT > 0) {
            v00 = inv00_regs<NT>(P, N, lane);
        } else {
            const bool bad = inverse_wave_lds(P, N, LDN, lane);
            v00 = bad ? INFINITY : P[0];
        }
        if (lane == 0) rout[o] = (v00 > 0.0 && isfinite(v00)) ? v00 / A.s[o] : INFINITY;
    }
    __syncthreads();
    TSTAMP(2);
}

// Hessian of the Lagrangian in reciprocal form + damping into M (free x free, compressed), E columns appended -- all threads
// (reuse: a further attempt of the same iteration -- the Hessian is the same, only the damping on the diagonal changed; the
// elimination works in registers and leaves the S x S block of L.M as it was, so the diagonal is restored from L.hd and damped anew)
__device__ void master_build_system(const MasterArgs &A, MasterLds &L, int tid, bool reuse)
{
    const int N = A.N, S = A.S, KM = A.KM, LDN = L.LDN, LDM = L.LDM;
    const int nf = L.istate[IS_NF], nact = L.istate[IS_NACT], nact0 = L.istate[IS_NACT0];
    const double dl = L.scal[SC_DAMP] * fabs(L.scal[SC_LAMEST]);
    if (reuse)
        for (int fa = tid; fa < nf; fa += MASTER_THREADS) {
            double h = L.hd[fa];
            h += dl * L.Dm[L.fi[fa]];
            L.M[fa * LDM + fa] = h;
        }
    for (int t = tid; t < (reuse ? 0 : nf * nf); t += MASTER_THREADS) {
        const int fa = t / nf, fb = t % nf;
        if (fa > fb) continue;
        const int i = L.fi[fa], j = L.fi[fb];
        const int ki = L.kk[i], kj = L.kk[j];
        double h = 0.0;
        for (int a = 0; a < nact0; a++) {              // curvature of every output of the iteration's active set
            const int o = L.act[a + MASTER_PACT];      // act0 list lives behind the current list
            const double ro = L.r[o];
            const double *T = L.TACT + (size_t)a * N * LDN;
            const double *ai = L.AAC + ((size_t)a * S + i) * KM, *aj = L.AAC + ((size_t)a * S + j) * KM;
            double acc = 0.0;
            for (int l = 0; l < ki; l++) {
                const double *Trow = T + (size_t)L.idx[i * KM + l] * LDN;
                double tt = 0.0;
                for (int l2 = 0; l2 < kj; l2++) tt = fma(Trow[L.idx[j * KM + l2]], aj[l2], tt);
                acc = fma(ai[l], tt, acc);
            }
            const double cci = (1.0 - A.eps_bg) * L.cc[i], ccj = (1.0 - A.eps_bg) * L.cc[j];
            const double gi = L.GQ[i * MASTER_PACT + a], gj = L.GQ[j * MASTER_PACT + a];
            h += L.muh[o] * ((2.0 / A.s[o]) * cci * ccj * acc / (ro * ro) - 2.0 * gi * gj * ro);
        }
        if (fa == fb) { L.hd[fa] = h; h += dl * L.Dm[i]; }
        L.M[fa * LDM + fb] = h;
        L.M[fb * LDM + fa] = h;
    }
    // E = [GQ columns of the active outputs (act0 order), rows of the iteration's caps (actc0 order), 1]
    const int ncact = L.istate[IS_NCACT0], ne = nact + ncact + 1;
    (void)nact0;
    for (int t = tid; t < nf * ne; t += MASTER_THREADS) {
        const int fa = t / ne, e = t % ne;
        double v = 1.0;
        if (e < nact) v = L.GQ[L.fi[fa] * MASTER_PACT + e];
        else if (e < nact + ncact) v = cap_a(A, L, L.actc[MASTER_MCAP + e - nact], L.fi[fa]);
        L.M[fa * LDM + nf + e] = v;
    }
    __syncthreads();
}

// ALL wavefronts: Gauss-Jordan on [M | E] -> Y = M^-1 E (left in the E columns of M), K = E^T Y (L.scal[SCK..]), L.istate[IS_OK].
// The matrix lives in REGISTERS, column-cyclic over the eight wavefronts (wavefront w holds columns w, w + 8, .. of M and of
// E; lane = row), so a pivot step costs every wavefront one fma per live column with the pivot row's entry broadcast by
// v_readlane -- out of LDS the same elimination by one wavefront moved 7 MB through the LDS pipe (24 us at 64 free variables).
// Per step the owner of the pivot column publishes the multipliers a_ip / a_pp (and the pivot) through a double-buffered LDS
// column and one workgroup barrier; the owner of the NEXT pivot column updates that column first and publishes it before it
// touches its other columns, so its division overlaps the other wavefronts' updates.  Same operations in the same order per
// entry as the single-wavefront version: bit-identical results.
__device__ __forceinline__ void factor_publish(MasterLds &L, double col, int p, int nf, int lane)
{
    const double piv = readlane_f64(col, p);
    const double rinv = rcp_f64(piv);                     // (on the critical path of every pivot step: the IEEE division is ~4x longer)
    double *fc = L.fcol + (p & 1) * 72;
    fc[lane] = (lane == p || lane >= nf) ? 0.0 : col * rinv;
    if (lane == 0) fc[64] = piv;
}

// pivot steps P, P + 1, .. of the elimination as straight-line code (static register indices); returns false when a pivot is not
// positive (the same decision in every wavefront: all read the same published value)
template <int P>
struct FactorStep {
    static __device__ __forceinline__ bool run(MasterLds &L, double (&am)[8], double (&ae)[2], double &mypiv, int nf, int ne, int wave, int lane)
    {
        if constexpr (P >= 64) return true;
        else {
            if (P >= nf) return true;
            __syncthreads();
            const double *fc = L.fcol + (P & 1) * 72;
            const double f = fc[lane], piv = fc[64];
            if (!(piv > 0.0) || !isfinite(piv)) return false;
            mypiv = (lane == P) ? piv : mypiv;
            constexpr int Q1 = (P + 1) >> 3;                                    // slot of the next pivot column
            const bool ahead = (P + 1 < 64) && (P + 1 < nf) && wave == ((P + 1) & 7);
            if constexpr (P + 1 < 64) {
                if (ahead) {
                    const double u = readlane_f64(am[Q1], P);
                    am[Q1] = fma(-f, u, am[Q1]);
                    factor_publish(L, am[Q1], P + 1, nf, lane);
                }
            }
#pragma unroll
            for (int q = P / 8; q < 8; q++) {                                   // slots below hold only columns behind the pivot
                const int c = wave + 8 * q;
                if (c > P && c < nf && !(ahead && c == P + 1)) {                // wave-uniform
                    const double u = readlane_f64(am[q], P);
                    am[q] = fma(-f, u, am[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (wave + 8 * q < ne) {
                    const double u = readlane_f64(ae[q], P);
                    ae[q] = fma(-f, u, ae[q]);
                }
            }
            return FactorStep<P + 1>::run(L, am, ae, mypiv, nf, ne, wave, lane);
        }
    }
};

__device__ void master_factor_all(const MasterArgs &A, MasterLds &L, int tid)
{
    const int LDM = L.LDM, wave = tid >> 6, lane = tid & 63;
    const int nf = L.istate[IS_NF], nact = L.istate[IS_NACT], ncact = L.istate[IS_NCACT0];
    const int ne = nact + ncact + 1;
    const bool row = lane < nf;
    double am[8], ae[2];
#pragma unroll
    for (int q = 0; q < 8; q++) { const int c = wave + 8 * q; am[q] = (row && c < nf) ? L.M[lane * LDM + c] : 0.0; }
#pragma unroll
    for (int q = 0; q < 2; q++) { const int e = wave + 8 * q; ae[q] = (row && e < ne) ? L.M[lane * LDM + nf + e] : 0.0; }
    double mypiv = 1.0;
    if (wave == 0) factor_publish(L, am[0], 0, nf, lane);
    const bool ok = FactorStep<0>::run(L, am, ae, mypiv, nf, ne, wave, lane);
    TSTAMP(6);
    __syncthreads();
    if (!ok) { if (tid == 0) L.istate[IS_OK] = 0; __syncthreads(); return; }
    const double dinv = row ? rcp_f64(mypiv) : 0.0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int e = wave + 8 * q;
        if (e < ne && row) L.M[lane * LDM + nf + e] = ae[q] * dinv;
    }
    __syncthreads();
    for (int t = wave; t < ne * ne; t += MASTER_THREADS / 64) {            // K = E^T Y, one entry per wavefront and pass
        const int e = t / ne, e2 = t % ne;
        double eo = 0.0, y = 0.0;
        if (row) {
            eo = e < nact ? L.GQ[L.fi[lane] * MASTER_PACT + e]
               : (e < nact + ncact ? cap_a(A, L, L.actc[MASTER_MCAP + e - nact], L.fi[lane]) : 1.0);
            y = L.M[lane * LDM + nf + e2];
        }
        const double k = fast_sum(eo * y);
        if (lane == 0) L.scal[SCK + e * MASTER_NE + e2] = k;
    }
    if (tid == 0) {
        L.istate[IS_OK] = 1;
        for (int e = 0; e < MASTER_PACT; e++) L.istate[IS_ALIVE + e] = e < nact;
        for (int e = 0; e < MASTER_MCAP; e++) { L.istate[IS_CALIVE + e] = e < ncact; L.istate[IS_CLOCK + e] = 0; }
        for (int e = 0; e < MASTER_PACT; e++) L.istate[IS_OLOCK + e] = 0;
    }
    __syncthreads();
}

// wave 0: the small KKT system over the alive outputs (p), alive caps (pc), the simplex row and tau:
//     K z + [1_p; 0; 0] tau = [rhs_q; rhs_c; 0],   sum z[0..p) = rhs_sum,
// then the step out_j = -Y z on the free rows.  soc = false: the SQP step (rhs_q = q at x, rhs_c = -(b_c - a_c.x), rhs_sum = 1):
// outputs / caps whose multiplier comes out negative leave the alive sets; a cap AT its bound that was dropped comes back (and
// stays) if the step would violate it; multipliers / lam / tau are published.  soc = true: the second-order correction (rhs_q = q
// at the trial point, rhs_c = 0, rhs_sum = 0) with the alive sets as the step left them.  The (<= 12 x 12) system is eliminated
// with one row per lane in registers; results are broadcast through LDS.
__device__ void master_small_solve(const MasterArgs &A, MasterLds &L, int lane, bool soc, double *outvec)
{
    constexpr int NU = MASTER_PACT + MASTER_MCAP + 2;
    __shared__ double sol[16];
    __shared__ double zf[MASTER_PACT], nuf[MASTER_MCAP], zlt[2];      // multipliers (act0 / actc0 order), lam, tau
    const int S = A.S, LDM = L.LDM;
    const int nf = L.istate[IS_NF], nact = L.istate[IS_NACT], ncact = L.istate[IS_NCACT0];
    // alive / locked sets as wave-uniform bit masks (bit e: candidate output e of act0, cap e of actc0)
    unsigned alive = (unsigned)__ballot(lane < MASTER_PACT && L.istate[IS_ALIVE + (lane < MASTER_PACT ? lane : 0)] != 0);
    unsigned olock = (unsigned)__ballot(lane < MASTER_PACT && L.istate[IS_OLOCK + (lane < MASTER_PACT ? lane : 0)] != 0);
    unsigned calive = (unsigned)__ballot(lane < MASTER_MCAP && L.istate[IS_CALIVE + (lane < MASTER_MCAP ? lane : 0)] != 0);
    unsigned clock = (unsigned)__ballot(lane < MASTER_MCAP && L.istate[IS_CLOCK + (lane < MASTER_MCAP ? lane : 0)] != 0);
    alive &= (1u << nact) - 1u;
    calive &= (1u << ncact) - 1u;
    for (int round = 0; round < 6 * (MASTER_PACT + MASTER_MCAP) + 6; round++) {
        // unknown a (= row a = lane a): the a-th alive output, then the alive caps, the simplex multiplier, tau
        const int p = __popc(alive), pc = __popc(calive), nk = p + pc + 1, n = nk + 1;
        int mymap = -1;                                                       // column of K behind unknown `lane`
        {
            int cnt = 0;
#pragma unroll
            for (int e = 0; e < MASTER_PACT; e++) if ((alive >> e) & 1u) { if (cnt == lane) mymap = e; cnt++; }
#pragma unroll
            for (int e = 0; e < MASTER_MCAP; e++) if ((calive >> e) & 1u) { if (cnt == lane) mymap = nact + e; cnt++; }
            if (lane == p + pc) mymap = nact + ncact;
        }
        // the (p + pc + 2)-square system, ONE ROW PER LANE in registers: Gauss-Jordan with partial pivoting -- the pivot row is the
        // unpivoted lane with the largest entry of the column (wave argmax, ties to the lower lane) and is broadcast with
        // v_readlane; rows are never swapped, a lane remembers the column it pivoted.  (One lane eliminating out of LDS paid a
        // memory round trip per entry: ~9 us per Newton system.)
        const bool rowin = lane < n;
        double r[NU], rhs = 0.0;
#pragma unroll
        for (int bq = 0; bq < NU; bq++) {
            const int mb = __builtin_amdgcn_readlane(mymap, bq);              // wave-uniform
            double v = 0.0;
            if (rowin && bq < n) {
                if (lane < nk && bq < nk) v = L.scal[SCK + mymap * MASTER_NE + mb];
                else if ((lane < p && bq == nk) || (lane == nk && bq < p)) v = 1.0;
            }
            r[bq] = v;
        }
        if (lane < p) rhs = L.scal[(soc ? SQT : SQ) + mymap];
        else if (lane < p + pc) rhs = soc ? 0.0 : -L.capslack[L.actc[MASTER_MCAP + mymap - nact]];
        else if (lane == nk) rhs = soc ? 0.0 : 1.0;
        bool pivoted = !rowin;                                               // lanes without a row never pivot
        int mycol = -1;
        double mydiag = 1.0;
        bool sing = false;
#pragma unroll
        for (int c = 0; c < NU; c++) {
            if (c < n && !sing) {                                            // wave-uniform
                double best = pivoted ? -1.0 : fabs(r[c]);
                int who = lane;
                row16_pick<true>(best, who);                                 // rows live in lanes 0..11: one DPP row
                best = readlane_f64(best, 0);
                if (!(best > 0.0)) { sing = true; }
                else {
                    const int pr = __builtin_amdgcn_readfirstlane(who);
                    const double pvc = readlane_f64(r[c], pr);
                    const double f = (lane == pr) ? 0.0 : r[c] / pvc;
#pragma unroll
                    for (int bq = 0; bq < NU; bq++)
                        if (bq >= c && bq < n) r[bq] = fma(-f, readlane_f64(r[bq], pr), r[bq]);
                    rhs = fma(-f, readlane_f64(rhs, pr), rhs);
                    if (lane == pr) { pivoted = true; mycol = c; mydiag = pvc; }
                }
            }
        }
        if (sing) { if (lane == 0) L.istate[IS_OK] = 0; wave_lds_sync(); return; }
        if (mycol >= 0) sol[mycol] = rhs / mydiag;
        if (lane < MASTER_PACT) zf[lane] = 0.0;
        if (lane < MASTER_MCAP) nuf[lane] = 0.0;
        wave_lds_sync();
        const double z = rowin ? sol[lane] : 0.0;                            // the value of unknown `lane`
        // a multiplier that comes out negative leaves the alive set (the most negative one, outputs first; locked ones stay)
        bool redo = false;
        if (!soc) {
            const bool isout = lane < p, iscap = lane >= p && lane < p + pc;
            bool elig = isout && p > 1 && z < -1.0e-12 && !((olock >> (mymap < 0 ? 0 : mymap)) & 1u);
            if (__ballot(elig) == 0ull) elig = iscap && z < -1.0e-12 && !((clock >> (iscap ? mymap - nact : 0)) & 1u);
            const unsigned long long em = __ballot(elig);
            if (em != 0ull) {
                double best = elig ? z : INFINITY;
                int who = lane;
                row16_pick<false>(best, who);
                who = __builtin_amdgcn_readfirstlane(who);
                const int col = __builtin_amdgcn_readlane(mymap, who);
                if (who < p) alive &= ~(1u << col); else calive &= ~(1u << (col - nact));
                redo = true;
            }
        }
        if (redo) continue;
        if (lane < p) zf[mymap] = z;
        else if (lane < p + pc) nuf[mymap - nact] = z;
        else if (lane == nk - 1) zlt[0] = z;
        else if (lane == nk) zlt[1] = z;
        wave_lds_sync();
        double di = 0.0;
        if (lane < nf) {
            for (int e = 0; e < nact; e++) di = fma(L.M[lane * LDM + nf + e], zf[e], di);
            for (int e = 0; e < ncact; e++) di = fma(L.M[lane * LDM + nf + nact + e], nuf[e], di);
            di = fma(L.M[lane * LDM + nf + nact + ncact], zlt[0], di);
        }
        for (int j = lane; j < S; j += 64) outvec[j] = 0.0;
        wave_lds_sync();
        if (lane < nf) outvec[L.fi[lane]] = -di;
        wave_lds_sync();
        // a cap at its bound that was dropped must not be violated by the step it was dropped from: it comes back, locked
        bool back = false;
        if (!soc)
            for (int e = 0; e < ncact; e++) {
                if ((calive >> e) & 1u) continue;
                const int c = L.actc[MASTER_MCAP + e];
                const double bc = L.capb[c];
                if (!(L.capslack[c] <= 1.0e-10 * fmax(fabs(bc), 1.0))) continue;
                double part = 0.0;
                for (int j = lane; j < S; j += 64) part = fma(cap_a(A, L, c, j), outvec[j], part);
                part = fast_sum(part);
                if (part > 1.0e-12 * fmax(fabs(bc), 1.0)) { calive |= 1u << e; clock |= 1u << e; back = true; }
            }
        // ... and a candidate output that was dropped must not rise above the level tau of the others to first order
        if (!soc)
            for (int e = 0; e < nact; e++) {
                if ((alive >> e) & 1u) continue;
                double part = 0.0;
                for (int j = lane; j < S; j += 64) part = fma(L.GQ[j * MASTER_PACT + e], outvec[j], part);
                part = fast_sum(part);
                const double tau = zlt[1];
                if (L.scal[SQ + e] + part > tau + 1.0e-10 * fabs(tau)) { alive |= 1u << e; olock |= 1u << e; back = true; }
            }
        if (back) { wave_lds_sync(); continue; }
        break;
    }
    if (!soc) {
        const double zp = (lane < nact && zf[lane < MASTER_PACT ? lane : 0] > 0.0) ? zf[lane] : 0.0;
        const double tot = fast_sum(zp);
        if (lane < nact) L.scal[SMU + lane] = tot > 0.0 ? zp / tot : 0.0;                         // new multipliers, act order
        if (lane < MASTER_MCAP) { const double nv = nuf[lane]; L.scal[SNU + lane] = (lane < ncact && tot > 0.0 && nv > 0.0) ? nv / tot : 0.0; }
        if (lane < MASTER_PACT) { L.istate[IS_ALIVE + lane] = (alive >> lane) & 1u; L.istate[IS_OLOCK + lane] = (olock >> lane) & 1u; }
        if (lane < MASTER_MCAP) { L.istate[IS_CALIVE + lane] = (calive >> lane) & 1u; L.istate[IS_CLOCK + lane] = (clock >> lane) & 1u; }
        if (lane == 0) { L.scal[SC_LAM] = zlt[0]; L.scal[SC_TAU] = zlt[1]; L.istate[IS_NALIVE] = __popc(alive); }
    }
    if (lane == 0) L.istate[IS_OK] = 1;
    wave_lds_sync();
}

// wave 0: make the trial point L.xt respect every cap (L.x does).  Clipping negative entries and renormalising moves mass
// between capped and uncapped groups, so a cap the step kept at its bound can end slightly violated.  First REPAIR -- scale the
// entries of the most violated cap's groups down to its bound and hand the freed mass to the other entries in proportion (a few
// rounds: caps overlap) --, then, if something is still violated, the furthest feasible point of the segment x -> xt.
__device__ void master_cap_feasible(const MasterArgs &A, MasterLds &L, int lane)
{
    const int S = A.S, ncap = A.ncap;
    for (int round = 0; round < 8; round++) {
        double ratio = 0.0;
        if (lane < ncap) {
            double ax = 0.0;
            for (int j = 0; j < S; j++) ax = fma(cap_a(A, L, lane, j), L.xt[j], ax);
            ratio = ax / (L.capb[lane] > 0.0 ? L.capb[lane] : 1.0);
        }
        // argmax over the caps (ties: smaller index), wave-uniform
        double best = ratio; int who = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double b2 = __shfl_xor(best, off, 64);
            const int w2 = __shfl_xor(who, off, 64);
            if (b2 > best || (b2 == best && w2 < who)) { best = b2; who = w2; }
        }
        if (!(best > 1.0 + 1.0e-13)) break;
        const bool msk = lane < S && cap_a(A, L, who, lane) > 0.0;
        const double xv = lane < S ? L.xt[lane] : 0.0;
        const double capped = fast_sum(msk ? xv : 0.0), rest = fast_sum(msk ? 0.0 : xv);
        if (!(rest > 0.0)) break;
        const double freed = (1.0 - 1.0 / best) * capped;
        if (lane < S) L.xt[lane] = msk ? xv / best : xv * (1.0 + freed / rest);
        wave_lds_sync();
    }
    double theta = INFINITY;
    if (lane < ncap) {
        double axt = 0.0;
        for (int j = 0; j < S; j++) axt = fma(cap_a(A, L, lane, j), L.xt[j], axt);
        const double ax = L.capb[lane] - L.capslack[lane];
        if (axt > L.capb[lane] * (1.0 + 1.0e-12) + 1.0e-300) theta = L.capslack[lane] / (axt - ax);
    }
    theta = -fast_max(-theta);
    if (theta < INFINITY) {
        theta = fmax(theta, 0.0);
        if (lane < S) L.xt[lane] = fma(theta, L.xt[lane] - L.x[lane], L.x[lane]);
    }
    wave_lds_sync();
}

template <int NT>
__global__ __launch_bounds__(MASTER_THREADS) void k_master_newton(MasterArgs A)
{
    extern __shared__ __align__(16) unsigned char master_sm[];
    MasterLds L;
    const int N = A.N, n_out = A.n_out, S = A.S, KM = A.KM;
    master_carve(L, master_sm, N, n_out, S, KM, NT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LDN = L.LDN, KE = L.KE;
    // ---- load the support ---------------------------------------------------------------------------
    for (int j = tid; j < S; j += MASTER_THREADS) { L.kk[j] = A.kk[j]; L.cc[j] = A.cc[j]; const double v = A.x[j]; L.x[j] = v > 0.0 ? v : 0.0; }
    for (int t = tid; t < S * KM; t += MASTER_THREADS) L.idx[t] = A.idx[t];
    for (int t = tid; t < S * N; t += MASTER_THREADS) L.pos[t] = -1;
    for (int o = tid; o < n_out; o += MASTER_THREADS) L.mu[o] = A.mu[o];
    if (tid < 48) L.istate[tid] = 0;
    for (int t = tid; t < 256; t += MASTER_THREADS) L.scal[t] = 0.0;
    for (int t = tid; t < n_out * L.PHS; t += MASTER_THREADS) L.PHI[t] = 0.0;      // pads of the reversed layout stay zero
    if (tid < 64) { L.capmodel[tid] = tid < A.ncap ? A.cap_model[tid] : 0; L.capb[tid] = tid < A.ncap ? A.cap_b[tid] : 0.0; L.nu[tid] = 0.0; L.capslack[tid] = 0.0; }
    __syncthreads();
#ifdef MASTER_TIMING
    if (tid == 0) reinterpret_cast<long long *>(L.scal + 240)[12] = wall_clock64();
#endif
    for (int t = tid; t < S * KM; t += MASTER_THREADS) {
        const int j = t / KM, l = t % KM;
        if (l < L.kk[j]) L.pos[j * N + L.idx[t]] = (signed char)l;
    }
    for (int t = tid; t < S * n_out * KE; t += MASTER_THREADS) {
        const int e = t % KE, o = (t / KE) % n_out, j = t / (KE * n_out);
        const int k = L.kk[j];
        double v = 0.0;
        const int64_t off = A.boff[(size_t)o * S + j];
        if (off >= 0 && e < k * (k + 1) / 2) {
            int l = 0, rem = e;
            while (rem >= k - l) { rem -= k - l; l++; }
            const int l2 = l + rem;
            const double *b = A.invcov[o] + off;
            v = 0.5 * (b[l * k + l2] + b[l2 * k + l]);
        }
        L.BLK[t] = v;
    }
    // membership masks from the positions: lane = support group, one ballot per model
    __syncthreads();
    for (int a = wave; a < N; a += MASTER_THREADS / 64) {
        const unsigned long long mk = __ballot(lane < S && L.pos[lane * N + a] >= 0);
        if (lane == 0) L.memb[a] = mk;
    }
    // destination lists of the Phi assembly (master_eval): per symmetric destination (a <= b) the support groups containing both
    // models with the packed entry (pos_a, pos_b) of their blocks, groups ascending
    __syncthreads();
    for (int t = tid; t < N * N; t += MASTER_THREADS) {
        const int a = t / N, b = t % N;
        if (a > b) continue;
        const int d = a * N - a * (a - 1) / 2 + (b - a);
        L.dl_off[d + 1] = (unsigned short)__popcll(L.memb[a] & L.memb[b]);
        L.dl_ab[d] = (unsigned short)(a | (b << 8));
    }
    __syncthreads();
    if (wave == 0) {   // offsets: inclusive scan of the counts, 64 destinations per pass
        int run = 0;
        for (int base = 0; base < L.ND; base += 64) {
            const int d = base + lane;
            int v = d < L.ND ? L.dl_off[d + 1] : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int o2 = __shfl_up(v, off, WAVE); if (lane >= off) v += o2; }
            if (d < L.ND) L.dl_off[d + 1] = (unsigned short)(run + v);
            run += __shfl(v, 63, WAVE);
        }
        if (lane == 0) L.dl_off[0] = 0;
    }
    __syncthreads();
    for (int t = tid; t < N * N; t += MASTER_THREADS) {
        const int a = t / N, b = t % N;
        if (a > b) continue;
        const int d = a * N - a * (a - 1) / 2 + (b - a);
        unsigned long long both = L.memb[a] & L.memb[b];
        int o = L.dl_off[d];
        while (both) {
            const int j = __ffsll((long long)both) - 1;
            both &= both - 1ull;
            const int pa = L.pos[j * N + a], pb = L.pos[j * N + b];
            const int lo = pa < pb ? pa : pb, hi = pa < pb ? pb : pa;
            L.dl_ent[o++] = (unsigned short)(j | (sym_e(lo, hi, L.kk[j]) << 7));
        }
    }
    if (wave == 0) {   // normalise the start
        double sx = 0.0;
        for (int j = lane; j < S; j += 64) sx += L.x[j];
        sx = fast_sum(sx);
        for (int j = lane; j < S; j += 64) L.x[j] = sx > 0.0 ? L.x[j] / sx : 1.0 / S;
    }
    if (tid == 0) {
        const double damp0 = 1.0e-2;      // (carrying the previous master's final damping over, x1 .. x1e4, changed nothing: profiles/r04_damping_warm_start.txt)
        L.scal[SC_DAMP] = damp0; L.istate[IS_STATUS] = 0; L.scal[SC_MBEST] = INFINITY;
    }
    __syncthreads();
    TSTAMP(0);                                          // 0: load
    master_eval<NT>(A, L, L.x, L.r, tid);
    if (tid == 0) {
        L.istate[IS_EVALS] = 1;
        double F = 0.0;
        for (int o = 0; o < n_out; o++) F = fmax(F, L.r[o]);
        if (!isfinite(F)) { L.istate[IS_DONE] = 1; L.istate[IS_STATUS] = 2; }       // start not evaluable
        L.scal[SC_F] = F;
        L.scal[SC_KKT] = INFINITY; L.scal[SC_SPREAD] = INFINITY;
    }
    __syncthreads();

    for (int it = 0; it < A.maxit && !L.istate[IS_DONE]; it++) {
        if (A.ncap > 0) {   // slack of every cap at x (lane = cap), then the caps of this iteration's step: at their bound or carrying
            if (wave == 0) {   // a multiplier, at most MASTER_MCAP, smallest slack first
                if (lane < A.ncap) {
                    double ax = 0.0;
                    for (int j = 0; j < S; j++) ax = fma(cap_a(A, L, lane, j), L.x[j], ax);
                    L.capslack[lane] = L.capb[lane] - ax;
                }
                wave_lds_sync();
                if (lane == 0) {
                    int n0 = 0;
                    for (int pick = 0; pick < MASTER_MCAP; pick++) {
                        int best = -1;
                        for (int c = 0; c < A.ncap; c++) {
                            bool taken = false;
                            for (int q = 0; q < n0; q++) if (L.actc[MASTER_MCAP + q] == c) taken = true;
                            if (taken) continue;
                            if (!(L.capslack[c] <= 1.0e-10 * fmax(fabs(L.capb[c]), 1.0) || L.nu[c] > 0.0)) continue;
                            if (best < 0 || L.capslack[c] < L.capslack[best]) best = c;
                        }
                        if (best < 0) break;
                        L.actc[MASTER_MCAP + n0++] = best;
                    }
                    for (int a = 0; a < n0; a++) for (int b = a + 1; b < n0; b++)      // ascending cap index
                        if (L.actc[MASTER_MCAP + b] < L.actc[MASTER_MCAP + a]) { const int t = L.actc[MASTER_MCAP + a]; L.actc[MASTER_MCAP + a] = L.actc[MASTER_MCAP + b]; L.actc[MASTER_MCAP + b] = t; }
                    L.istate[IS_NCACT0] = n0;
                }
            }
            __syncthreads();
        }
        // ---- active outputs and their curvature weights (wavefront 0, lane = output) --------------------
        if (wave == 0) {
            const double F = L.scal[SC_F];
            const bool in = lane < n_out;
            const double ro = in ? L.r[lane] : 0.0, muo = in ? L.mu[lane] : 0.0;
            // candidates: within act_tol of the maximum, or carrying a multiplier; the MASTER_PACT largest r (ties: smaller index)
            const bool cand = in && ((ro >= F * (1.0 - A.act_tol)) || (muo > 1.0e-12));
            const unsigned long long cmask = __ballot(cand);
            int rank = 0;
            for (int o2 = 0; o2 < n_out; o2++) {
                const double r2 = readlane_f64(ro, o2);
                if (((cmask >> o2) & 1ull) && (r2 > ro || (r2 == ro && o2 < lane))) rank++;
            }
            const bool sel = cand && rank < MASTER_PACT;
            const unsigned long long smask = __ballot(sel);
            const int nact = __popcll(smask);
            const int slot = __popcll(smask & ((1ull << lane) - 1ull));          // ascending output order (matches the restatement)
            if (sel) { L.act[slot] = lane; L.act[slot + MASTER_PACT] = lane; }   // act0 behind the list: the iteration's set (GQ / TACT / AAC order)
            const double m_o = sel ? (muo > 0.0 ? muo : 0.0) : 0.0;
            double tot = fast_sum(m_o);
            double wgt = sel ? (tot > 0.0 ? m_o / tot : 1.0 / nact) : 0.0;
            wgt = sel ? fmax(wgt, 1.0e-3 / nact) : 0.0;
            tot = fast_sum(wgt);
            if (in) L.muh[lane] = sel ? wgt / tot : 0.0;
            if (lane == 0) { L.istate[IS_NACT] = nact; L.istate[IS_NACT0] = nact; }
        }
        __syncthreads();
        // ---- derivatives at x: T (kept for the active outputs), a_{o,j}, gradients in reciprocal form ---------------
        // T_o = Phi_o^-1 of the current x is in L.PHI: left there by the initial evaluation or by the accepted trial evaluation
        const int nact0 = L.istate[IS_NACT0];
        if constexpr (NT > 0) {     // L.PHI holds Phi_o(x): one wavefront per active output inverts it in registers
            for (int a = wave; a < nact0; a += MASTER_THREADS / 64)
                (void)inverse_regs<NT>(L.PHI + (size_t)L.act[a + MASTER_PACT] * L.PHS, L.TACT + (size_t)a * N * LDN, N, LDN, lane);
        } else {                    // L.PHI holds T_o already (inverted in place by the evaluation)
            for (int t = tid; t < nact0 * N * LDN; t += MASTER_THREADS) {
                const int a = t / (N * LDN);
                L.TACT[t] = L.PHI[(size_t)L.act[a + MASTER_PACT] * L.PHS + (t - a * N * LDN)];
            }
        }
        __syncthreads();
#ifdef MASTER_TIMING_FINE
        TSTAMP(9);
#endif
        for (int t = tid; t < nact0 * S * KM; t += MASTER_THREADS) {
            const int l = t % KM, j = (t / KM) % S, a = t / (KM * S);
            const int o = L.act[a + MASTER_PACT], k = L.kk[j];
            double acc = 0.0;
            if (l < k) {
                const double *T = L.TACT + (size_t)a * N * LDN;
                const double *B = L.BLK + ((size_t)j * n_out + o) * KE;
                for (int l2 = 0; l2 < k; l2++) {
                    const int lo = l < l2 ? l : l2, hi = l < l2 ? l2 : l;
                    acc = fma(B[sym_e(lo, hi, k)], T[(size_t)L.idx[j * KM + l2] * LDN + 0], acc);
                }
            }
            L.AAC[t] = acc;
        }
        __syncthreads();
#ifdef MASTER_TIMING_FINE
        TSTAMP(10);
#endif
        for (int t = tid; t < S * nact0; t += MASTER_THREADS) {
            const int a = t % nact0, j = t / nact0;
            const int o = L.act[a + MASTER_PACT], k = L.kk[j];
            const double *T = L.TACT + (size_t)a * N * LDN;
            double acc = 0.0;
            for (int l = 0; l < k; l++) acc = fma(L.AAC[((size_t)a * S + j) * KM + l], T[(size_t)L.idx[j * KM + l] * LDN + 0], acc);
            const double ro = L.r[o];
            L.GQ[j * MASTER_PACT + a] = -(1.0 - A.eps_bg) * L.cc[j] * acc / A.s[o] / (ro * ro);
        }
        if (tid == 0) {
            for (int a = 0; a < nact0; a++) L.scal[SQ + a] = -1.0 / L.r[L.act[a + MASTER_PACT]];      // q of the act0 outputs
            double qm = -INFINITY;
            for (int o = 0; o < n_out; o++) qm = fmax(qm, -1.0 / L.r[o]);
            L.scal[SC_QMAX] = qm;
        }
        __syncthreads();
        TSTAMP(3);                                      // 3: active set + derivatives
        if (wave == 0) {   // reduced costs with the curvature weights -> free set
            double part = 0.0;
            for (int j = lane; j < S; j += 64) {
                double g = 0.0;
                for (int a = 0; a < nact0; a++) g = fma(L.muh[L.act[a + MASTER_PACT]], L.GQ[j * MASTER_PACT + a], g);
                for (int c = 0; c < A.ncap; c++) if (L.nu[c] > 0.0) g = fma(L.nu[c], cap_a(A, L, c, j), g);
                L.glv[j] = g;
                part = fma(g, L.x[j], part);
            }
            const double lam_est = -fast_sum(part);
            int nf = 0;
            for (int base = 0; base < S; base += 64) {
                const int j = base + lane;
                const bool fr = j < S && (L.x[j] > 0.0 || L.glv[j] + lam_est < 0.0);
                const unsigned long long bal = __ballot(fr);
                if (fr) L.fi[nf + __popcll(bal & ((1ull << lane) - 1ull))] = j;
                nf += __popcll(bal);
                if (j < S) L.Dm[j] = 1.0 / fmax(L.x[j], A.floor_x);
            }
            if (lane == 0) { L.scal[SC_LAMEST] = lam_est; L.istate[IS_NF] = nf; L.istate[IS_ACCEPT] = 0; }
        }
        __syncthreads();
        // ---- damped attempts -------------------------------------------------------------------------------------
        bool converged = false;
        int noise_stop = 0;                      // 1: stopped by the KKT monitor
        double pred0 = -INFINITY;                // decrease the model promised at the iteration's first attempt
        for (int attempt = 0; attempt < 40; attempt++) {
            if (tid == 0) L.istate[IS_NACT] = L.istate[IS_NACT0];
            if (tid < MASTER_PACT) L.act[tid] = L.act[tid + MASTER_PACT];
            __syncthreads();
            TSTAMP(4);                                  // 4: free set, KKT bookkeeping, step formation
            master_build_system(A, L, tid, attempt > 0);
            TSTAMP(5);                                  // 5: Hessian / system assembly
            master_factor_all(A, L, tid);
#ifdef MASTER_TIMING_FINE
            TSTAMP(11);
#endif
            if (wave == 0) {
                if (L.istate[IS_OK]) master_small_solve(A, L, lane, false, L.d);
                if (lane == 0) L.istate[IS_SOLVES] += 1;
            }
            __syncthreads();
            TSTAMP(7);                                  // 7: K + small system
            if (!L.istate[IS_OK]) {              // M not positive definite (or singular small system): more damping
                __syncthreads();
#ifdef MASTER_COUNTS
                if (tid == 0) L.scal[200] += 1.0;
#endif
                if (tid == 0) L.scal[SC_DAMP] *= 10.0;
                __syncthreads();
                if (L.scal[SC_DAMP] > 1.0e12) break;
                continue;
            }
            // new multipliers (full vector) and, at the first attempt, the KKT residual at x
            if (wave == 0) {
                const int nact = L.istate[IS_NACT0];
                if (attempt == 0) {
                    double part = 0.0;
                    for (int j = lane; j < S; j += 64) {
                        double g = 0.0;
                        for (int a = 0; a < nact; a++) g = fma(L.scal[SMU + a], L.GQ[j * MASTER_PACT + a], g);
                        for (int e = 0; e < L.istate[IS_NCACT0]; e++)
                            if (L.scal[SNU + e] > 0.0) g = fma(L.scal[SNU + e], cap_a(A, L, L.actc[MASTER_MCAP + e], j), g);
                        L.xt[j] = g;                    // scratch
                        part = fma(g, L.x[j], part);
                    }
                    const double lam_x = -fast_sum(part);
                    double worst = 0.0;
                    for (int j = lane; j < S; j += 64) {
                        const double rc = L.xt[j] + lam_x;
                        worst = fmax(worst, L.x[j] > 1.0e-10 ? fabs(rc) : fmax(-rc, 0.0));      // entries below 1e-10 count as at the bound
                    }
                    worst = fast_max(worst) / fmax(fabs(lam_x), 1.0e-300);
                    if (lane == 0) {
                        double sp = 0.0;
                        const double F = L.scal[SC_F];
                        for (int a = 0; a < nact; a++) if (L.scal[SMU + a] > 0.0) sp = fmax(sp, (F - L.r[L.act[a + MASTER_PACT]]) / F);
                        L.scal[SC_KKT] = worst; L.scal[SC_SPREAD] = sp; L.scal[SC_LAMX] = lam_x;
                    }
                }
            }
            __syncthreads();
            if (attempt == 0 && L.scal[SC_KKT] <= A.tol && L.scal[SC_SPREAD] <= A.tol) { converged = true; break; }
            if (attempt == 0) {
                // the best point seen (smallest KKT measure), with the multipliers estimated at it.  Steps taken on trust (IS_NOISE, see
                // the acceptance test) are judged here: three in a row that do not improve on the best point end the master there
                const double m_now = fmax(L.scal[SC_KKT], L.scal[SC_SPREAD]);
                const bool better = m_now < L.scal[SC_MBEST];
                const bool giveup = !better && L.istate[IS_NOISE] && L.istate[IS_NFAIL] >= 2;
                __syncthreads();
                if (better) {
                    for (int j = tid; j < S; j += MASTER_THREADS) L.xp[j] = L.x[j];
                    for (int o = tid; o < n_out; o += MASTER_THREADS) { L.rp[o] = L.r[o]; L.mup[o] = 0.0; }
                    __syncthreads();
                    if (tid == 0) {
                        for (int a = 0; a < L.istate[IS_NACT0]; a++) L.mup[L.act[a + MASTER_PACT]] = L.scal[SMU + a];
                        L.scal[SC_MBEST] = m_now; L.scal[SC_FBEST] = L.scal[SC_F]; L.scal[SC_KKTBEST] = L.scal[SC_KKT]; L.scal[SC_SPREADBEST] = L.scal[SC_SPREAD];
                        L.istate[IS_NFAIL] = 0;
                    }
                } else if (L.istate[IS_NOISE]) {
                    if (tid == 0) L.istate[IS_NFAIL] += 1;
                }
                __syncthreads();
                if (giveup) { noise_stop = 1; break; }
            }
            const double F = L.scal[SC_F];
            const double pred = (L.scal[SC_TAU] - L.scal[SC_QMAX]) * F * F;
            if (attempt == 0) pred0 = pred;
            if (pred < -0.5 * F) {               // the model promises more than half of a positive objective: shorter step
                __syncthreads();
#ifdef MASTER_COUNTS
                if (tid == 0) L.scal[201] += 1.0;
#endif
                if (tid == 0) L.scal[SC_DAMP] *= 10.0;
                __syncthreads();
                if (L.scal[SC_DAMP] > 1.0e12) break;
                continue;
            }
            // projected step, renormalised.  fb > 0 (no background): every entry keeps at least 1 - fb of its value, so that no
            // model drops out of the information matrix inside the master (V has a kink there)
            if (wave == 0) {
                double sx = 0.0;
                for (int j = lane; j < S; j += 64) {
                    const double v = fmax(L.x[j] + L.d[j], A.fb > 0.0 ? (1.0 - A.fb) * L.x[j] : 0.0);
                    L.xt[j] = v; sx += v;
                }
                sx = fast_sum(sx);
                for (int j = lane; j < S; j += 64) L.xt[j] = L.xt[j] / sx;
                if (A.ncap > 0) { wave_lds_sync(); master_cap_feasible(A, L, lane); }
            }
            __syncthreads();
            TSTAMP(4);
            master_eval<NT>(A, L, L.xt, L.rt, tid);
            // acceptance; near a tie of several outputs second-order errors split the tie and the exact max rejects a good SQP
            // step (the Maratos effect): one second-order correction -- the minimum-norm (in M) step c that re-equalises the
            // alive outputs at the trial point to first order, same K, another right-hand side -- gets a second evaluation
            for (int pass = 0; pass < 2; pass++) {
                if (tid == 0) {
                    L.istate[IS_EVALS] += 1;
                    double Ft = 0.0;
                    for (int o = 0; o < n_out; o++) Ft = fmax(Ft, L.rt[o]);
                    const double actual = Ft - F;
                    L.istate[IS_OK] = 0;                           // reused: "try the correction"
                    // below a promised decrease of 1e-11 F the objective no longer tells a good step from a bad one (its own rounding
                    // is of that size): such a step is taken when the objective does not RISE by more than that, and the next
                    // iteration judges it by the KKT residual instead (IS_NOISE, see there) -- Newton's method on the KKT system
                    // converges where the objective has stopped resolving, which is what takes the residual from 1e-6 to 1e-10
                    const bool noise = pred > -1.0e-11 * F && fmax(L.scal[SC_KKT], L.scal[SC_SPREAD]) <= 1.0e-4 && L.scal[SC_DAMP] <= 1.0e-2;      // (a tiny step of a heavily damped system is not noise)
                    if (isfinite(Ft) && (actual <= 1.0e-4 * fmin(pred, 0.0) + 1.0e-15 * F || (noise && actual <= 1.0e-11 * F))) {
                        const double ratio = (pred < 0.0 && !noise) ? actual / pred : 1.0;      // a step taken on trust counts as a good one
                        if (ratio > 0.5 && (attempt == 0 || !MASTER_DAMP_HOLD)) L.scal[SC_DAMP] = fmax(L.scal[SC_DAMP] * MASTER_DAMP_DOWN, 1.0e-14);
                        else if (ratio < 0.1) L.scal[SC_DAMP] *= 10.0;
                        L.istate[IS_ACCEPT] = 1;
                        L.scal[SC_FT] = Ft;
                        L.istate[IS_NOISE] = noise ? 1 : 0;
                    } else if (pass == 0 && isfinite(Ft) && L.istate[IS_NALIVE] > 1) {
                        for (int a = 0; a < L.istate[IS_NACT0]; a++) L.scal[SQT + a] = -1.0 / L.rt[L.act[a + MASTER_PACT]];
                        L.istate[IS_OK] = 1;
#ifdef MASTER_COUNTS
                        L.scal[202] += 1.0;
#endif
                    } else {
                        L.scal[SC_DAMP] *= MASTER_DAMP_UP;
#ifdef MASTER_COUNTS
                        L.scal[203] += 1.0; if (!isfinite(Ft)) L.scal[204] += 1.0;
                        if (isfinite(Ft) && pred < 0.0) { const double rr = actual / pred; if (rr > -1.0) L.scal[205] += 1.0; }
#endif
                    }
                }
                __syncthreads();
                if (pass == 1 || !L.istate[IS_OK]) break;
                __syncthreads();
                if (wave == 0) {
                    master_small_solve(A, L, lane, true, L.glv);          // c into glv (free after the free-set step)
                    if (L.istate[IS_OK]) {
                        double sx = 0.0;
                        for (int j = lane; j < S; j += 64) {
                            const double v = fmax(L.x[j] + L.d[j] + L.glv[j], A.fb > 0.0 ? (1.0 - A.fb) * L.x[j] : 0.0);
                            L.xt[j] = v; sx += v;
                        }
                        sx = fast_sum(sx);
                        for (int j = lane; j < S; j += 64) L.xt[j] = L.xt[j] / sx;
                        if (A.ncap > 0) { wave_lds_sync(); master_cap_feasible(A, L, lane); }
                    }
                }
                __syncthreads();
                if (!L.istate[IS_OK]) {                             // singular correction system: plain rejection
                    __syncthreads();
                    if (tid == 0) L.scal[SC_DAMP] *= 10.0;
                    break;
                }
                master_eval<NT>(A, L, L.xt, L.rt, tid);
            }
            __syncthreads();
            if (L.istate[IS_ACCEPT] || L.scal[SC_DAMP] > 1.0e12) break;
        }
        __syncthreads();
        // ---- take the multipliers; take the step if one was accepted ------------------------------------------------
        if (converged || L.istate[IS_ACCEPT]) {
            if (tid == 0) {
                for (int o = 0; o < n_out; o++) L.mu[o] = 0.0;
                for (int a = 0; a < L.istate[IS_NACT0]; a++) L.mu[L.act[a + MASTER_PACT]] = L.scal[SMU + a];
                for (int c = 0; c < A.ncap; c++) L.nu[c] = 0.0;
                for (int e = 0; e < L.istate[IS_NCACT0]; e++) L.nu[L.actc[MASTER_MCAP + e]] = L.scal[SNU + e];
            }
        }
        if (noise_stop) {      // the iteration has reached what this arithmetic resolves: back to the best point, with its multipliers
            for (int j = tid; j < S; j += MASTER_THREADS) L.x[j] = L.xp[j];
            for (int o = tid; o < n_out; o += MASTER_THREADS) { L.r[o] = L.rp[o]; L.mu[o] = L.mup[o]; }
            if (tid == 0) {
                L.scal[SC_F] = L.scal[SC_FBEST]; L.scal[SC_KKT] = L.scal[SC_KKTBEST]; L.scal[SC_SPREAD] = L.scal[SC_SPREADBEST];
                L.istate[IS_DONE] = 1;
            }
            __syncthreads();
            break;
        }
        if (converged) { if (tid == 0) L.istate[IS_DONE] = 1; __syncthreads(); break; }
        if (!L.istate[IS_ACCEPT]) {
            // no step was accepted.  If even the first attempt's model promised less than the objective can resolve (1e-12 F) at a KKT
            // residual of 1e-6 or better, the rejections are rounding noise of the evaluations: this IS the optimum to working precision
            // (status 0); otherwise the master stalled (status 1)
            const bool resolved = L.scal[SC_KKT] <= 1.0e-6 && L.scal[SC_SPREAD] <= 1.0e-6 && pred0 > -1.0e-12 * L.scal[SC_F];
            __syncthreads();
            if (tid == 0) { L.istate[IS_DONE] = 1; L.istate[IS_STATUS] = resolved ? 0 : 1; }
            __syncthreads();
            break;
        }
        for (int j = tid; j < S; j += MASTER_THREADS) L.x[j] = L.xt[j];
        for (int o = tid; o < n_out; o += MASTER_THREADS) L.r[o] = L.rt[o];
        if (tid == 0) { L.scal[SC_F] = L.scal[SC_FT]; L.istate[IS_IT] = it + 1; }
        __syncthreads();
    }
    __syncthreads();
    // ---- results ------------------------------------------------------------------------------------------------------
    for (int j = tid; j < S; j += MASTER_THREADS) A.x[j] = L.x[j];
    for (int o = tid; o < n_out; o += MASTER_THREADS) { A.mu[o] = L.mu[o]; A.out[MASTER_OUT + o] = L.r[o]; }
    for (int c = tid; c < A.ncap; c += MASTER_THREADS) A.nu[c] = L.nu[c];
    if (tid == 0) {
        A.out[0] = L.scal[SC_F]; A.out[1] = L.scal[SC_LAM]; A.out[2] = L.scal[SC_KKT]; A.out[3] = L.scal[SC_SPREAD];
        A.out[4] = L.istate[IS_IT]; A.out[5] = L.istate[IS_EVALS]; A.out[6] = L.istate[IS_SOLVES]; A.out[7] = L.istate[IS_STATUS];
        A.out[8] = L.scal[SC_DAMP]; A.out[9] = L.scal[SC_LAMX];
#ifdef MASTER_COUNTS
        printf("COUNTS S %d it %d solves %d evals %d | notPD %g modelTooBig %g soc %g rejected %g (inf %g, mild %g) status %d\n", S, L.istate[IS_IT], L.istate[IS_SOLVES], L.istate[IS_EVALS],
               L.scal[200], L.scal[201], L.scal[202], L.scal[203], L.scal[204], L.scal[205], L.istate[IS_STATUS]);
#endif
#ifdef MASTER_TIMING
        for (int k = 0; k < 8; k++) A.out[8 + k] = (double)reinterpret_cast<long long *>(L.scal + 240)[k] * 0.01;      // microseconds
#ifdef MASTER_TIMING_FINE
        printf("FINE %lld %lld %lld %lld\n", reinterpret_cast<long long *>(L.scal + 240)[8], reinterpret_cast<long long *>(L.scal + 240)[9],
               reinterpret_cast<long long *>(L.scal + 240)[10], reinterpret_cast<long long *>(L.scal + 240)[11]);
#endif
#endif
    }
}


// ------------------------------------------------------------------------------------------------------
// phase 1: multiplicative algorithm on the full problem.  For the p-norm surrogate  f = || (r_o)_o ||_p  the update
//     x_i <- x_i * ( sum_o wgt_o c_i q_{o,i} / s_o ) / ( sum_o wgt_o r_o ),     wgt_o ~ r_o^(p-1),  q_{o,i} = -dV_o/dm_i >= 0
// keeps sum x = 1 exactly in exact arithmetic (V_o is homogeneous of degree -1: sum_i x_i c_i q_{o,i} = V_o) and is monotone
// for a single output (the classical algorithm for c-optimal design weights).  No projection, no line search, no reduction:
// one evaluation (Phi pass + solve + gradient tiles) and this elementwise kernel per iteration.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ma_update(int64_t L, int n_out, const double *__restrict__ var, const int32_t *__restrict__ status,
                                                   const double *__restrict__ grad, const int64_t *__restrict__ goff,
                                                   const int32_t *__restrict__ invmap, const double *__restrict__ s,
                                                   const double *__restrict__ cc, double p, double *__restrict__ x, double *__restrict__ m)
{
    __shared__ double wgt[64], sden;                       // the n_out weights are the same for every group: once per workgroup
    __shared__ int64_t sgo[64];                            // ... and so are the gradient offsets (a dependent load per output otherwise)
    __shared__ int sok;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ double sr[64], sterm[64];
    __shared__ int sst[64];
    if (threadIdx.x < n_out) {                             // the per-output scalars in parallel (one dependent round trip, not 3 n_out)
        const int o = threadIdx.x;
        const double so = s[o];
        sr[o] = var[o] / so; sst[o] = status[o]; sgo[o] = goff[o]; wgt[o] = so;
    }
    __syncthreads();
    if (threadIdx.x < n_out) {                             // ... and the weights too: one pow() per thread, not n_out of them on one
        const int o = threadIdx.x;
        double rmax = 0.0;
        bool ok = true;
        for (int oo = 0; oo < n_out; oo++) { ok = ok && sst[oo] == BLUEST_EVAL_OK; rmax = fmax(rmax, sr[oo]); }
        ok = ok && rmax > 0.0 && isfinite(rmax);
        const double ro = sr[o];
        const double w = (n_out == 1 || !ok) ? 1.0 : pow(ro / rmax, p - 1.0);
        sterm[o] = w;                                      // (the denominator is summed below, in output order)
        wgt[o] = w / wgt[o];
        if (o == 0) sok = ok ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                // the denominator in output order (same sum as before)
        double den = 0.0;
        for (int o = 0; o < n_out; o++) den = fma(sterm[o], sr[o], den);
        sden = den;
    }
    __syncthreads();
    if (i >= L || !sok) return;                            // not evaluable: the iterate is left alone (uniform across the grid)
    double num = 0.0;
    if (!invmap) {
        // identity mapping: the n_out gradient entries are independent loads -- issued eight at a time, then summed in output order
        for (int o0 = 0; o0 < n_out; o0 += 8) {
            double g[8];
#pragma unroll
            for (int q = 0; q < 8; q++) g[q] = (o0 + q < n_out) ? grad[sgo[o0 + q] + i] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; q++) if (o0 + q < n_out) num = fma(wgt[o0 + q], -g[q], num);
        }
    } else {
        for (int o = 0; o < n_out; o++) {
            const int32_t li = invmap[(int64_t)o * L + i];
            if (li >= 0) num = fma(wgt[o], -grad[sgo[o] + li], num);
        }
    }
    // (over-relaxed steps x * ratio^delta, delta 1.5 .. 3, alternating with plain ones: fewer iterations of this phase, up to 2.7x more
    // Newton iterations afterwards at the headline size -- profiles/r04_ma_power_negative_result.txt)
    const double xn = x[i] * cc[i] * num / sden;
    x[i] = xn;
    m[i] = cc[i] * xn;
}

// allocation of the full problem from a support vector: m_i = c_i ((1 - eps) x_S[i in S] + eps / L)
__global__ __launch_bounds__(256) void k_support_point(int64_t L, int S, const int64_t *__restrict__ sup, const double *__restrict__ xs,
                                                       const double *__restrict__ cc, double eps, double *__restrict__ m)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    int lo = 0, hi = S;                                   // sup is ascending: binary search
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sup[mid] < i) lo = mid + 1; else hi = mid; }
    const double xi = (lo < S && sup[lo] == i) ? xs[lo] : 0.0;
    m[i] = cc[i] * ((1.0 - eps) * xi + eps / (double)L);
}

// pricing: c_i = cc_i sum_o (mu_o / s_o) q_{o,i} for every group.  Every thread keeps the largest entry of its strided scan and
// the workgroup reports the PRICE_TOP largest of those per-thread maxima (value, index): the first is the workgroup's true
// maximum (all the bound needs), the others are candidates -- two of the true top 16 that fall on the same thread's stride
// yield one candidate, the other enters in a later round.
#define PRICE_TOP 16
#define PRICE_BLOCKS 64
__global__ __launch_bounds__(256) void k_price(int64_t L, int n_out, const double *__restrict__ grad, const int64_t *__restrict__ goff,
                                               const int32_t *__restrict__ invmap, const double *__restrict__ mu,
                                               const double *__restrict__ s, const double *__restrict__ cc, int S,
                                               const int64_t *__restrict__ sup, double *__restrict__ c_sup,
                                               double *__restrict__ top_val, int64_t *__restrict__ top_idx,
                                               const double *__restrict__ v_ws, int N, double *__restrict__ y0,
                                               const unsigned long long *__restrict__ capmask, const double *__restrict__ nu,
                                               const double *__restrict__ master_out)
{
    // sample caps: c_i - cc_i F^2 sum_{caps c whose model is in group i} nu_c  (the master's multipliers are those of its
    // reciprocal form; F^2 nu_c are the multipliers of the caps in the problem as posed).  capmask: bit c of entry i.
    const double F2 = capmask ? master_out[0] * master_out[0] : 0.0;
    __shared__ double wv[4];
    __shared__ int64_t wi[4];
    __shared__ int wt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double best = -INFINITY;                               // (with caps a corrected reduced cost can be negative)
    int64_t besti = -1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < L; i += (int64_t)gridDim.x * 256) {
        double c = 0.0;
        for (int o = 0; o < n_out; o++) {
            const int32_t li = invmap ? invmap[(int64_t)o * L + i] : (int32_t)i;
            if (li >= 0 && mu[o] > 0.0) c = fma(mu[o] / s[o], -grad[goff[o] + li], c);
        }
        c *= cc[i];
        if (capmask) {
            unsigned long long mk = capmask[i];
            double corr = 0.0;
            while (mk) { const int b = __ffsll((long long)mk) - 1; mk &= mk - 1ull; corr += nu[b]; }
            c = fma(-cc[i] * F2, corr, c);
        }
        if (c > best) { best = c; besti = i; }             // ascending scan: ties keep the smaller index
    }
    // y_{o,0}: component 0 of the vector the quadratic forms were taken with (row 0 of the inverse of the WHOLE information
    // matrix).  The bound needs THIS number, not V: V restricts Phi to the models sampled with |m| > 1e-6 (bluest/misc.py:453-457),
    // which the tiny background does not reach -- using V made the bound invalid on ill-conditioned data (6 % on the NS problem)
    if (blockIdx.x == 0 && tid < n_out) y0[tid] = v_ws[(int64_t)tid * N];
    if (blockIdx.x == 0)                                   // reduced costs of the support entries themselves
        for (int j = tid; j < S; j += 256) {
            const int64_t i = sup[j];
            double c = 0.0;
            for (int o = 0; o < n_out; o++) {
                const int32_t li = invmap ? invmap[(int64_t)o * L + i] : (int32_t)i;
                if (li >= 0 && mu[o] > 0.0) c = fma(mu[o] / s[o], -grad[goff[o] + li], c);
            }
            c *= cc[i];
            if (capmask) {
                unsigned long long mk = capmask[i];
                double corr = 0.0;
                while (mk) { const int b = __ffsll((long long)mk) - 1; mk &= mk - 1ull; corr += nu[b]; }
                c = fma(-cc[i] * F2, corr, c);
            }
            c_sup[j] = c;
        }
    for (int r = 0; r < PRICE_TOP; r++) {
        // block argmax (ties: smaller index), fixed order
        double v = best; int64_t ix = besti; int who = tid;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double v2 = __shfl_xor(v, off, 64);
            const long long i2 = __shfl_xor((long long)ix, off, 64);
            const int w2 = __shfl_xor(who, off, 64);
            if (v2 > v || (v2 == v && i2 >= 0 && (ix < 0 || i2 < ix))) { v = v2; ix = i2; who = w2; }
        }
        if (lane == 0) { wv[wave] = v; wi[wave] = ix; wt[wave] = who; }
        __syncthreads();
        double bv = wv[0]; int64_t bi = wi[0]; int bt = wt[0];
        for (int w = 1; w < 4; w++)
            if (wv[w] > bv || (wv[w] == bv && wi[w] >= 0 && (bi < 0 || wi[w] < bi))) { bv = wv[w]; bi = wi[w]; bt = wt[w]; }
        if (tid == 0) { top_val[blockIdx.x * PRICE_TOP + r] = bv; top_idx[blockIdx.x * PRICE_TOP + r] = bi; }
        if (tid == bt) { best = -INFINITY; besti = -1; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
static int master_lds_limit()
{
    static int limit_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 64 << 10;
    if (!limit_of[dev]) {
        hipDeviceProp_t prop;
        limit_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? (int)std::min<size_t>(prop.sharedMemPerBlock, 160u << 10) : (64 << 10);
    }
    return limit_of[dev];
}

// Which instantiation runs a problem, and the largest support it holds.  Up to 32 models the register tiles 12 / 20 / 26 / 32.
// Beyond: 40 / 48 / 64 in registers (one evaluation's elimination 3 us instead of 150 in LDS) when the larger Phi stride (nt + 2
// instead of N + 1) still leaves the support the natural layout would allow, or N + 16 entries; else the LDS eliminations (nt = 0).
static int master_choose_nt(int N, int n_out, int KM, int *s_max)
{
    const size_t limit = (size_t)master_lds_limit() - MASTER_STATIC_LDS;
    auto fit = [&](int nt) {
        int S = MASTER_SMAX;
        while (S > 0 && master_lds_bytes(N, n_out, S, KM, nt) > limit) S -= 2;
        return S;
    };
    int nt = N <= 12 ? 12 : N <= 20 ? 20 : N <= 26 ? 26 : N <= 32 ? 32 : 0;
    int S = fit(nt);
    if (!nt) {
        const int cand = N <= 40 ? 40 : N <= 48 ? 48 : 64;
        const int s_reg = fit(cand);
        if (s_reg >= std::min(S, std::min(MASTER_SMAX, N + 16))) { nt = cand; S = s_reg; }
    }
    if (s_max) *s_max = S;
    return nt;
}

extern "C" int bluest_master_max_support(bluest_plan_t plan, int *s_max)
{
    if (!plan || !s_max) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    int rc = require_gpu(); if (rc) return rc;
    int KM = 0;
    for (const auto &od : plan->outs) KM = std::max(KM, od.K);
    const int n_out = (int)plan->outs.size();
    (void)master_choose_nt(plan->N, n_out, KM, s_max);     // 0: this problem does not fit the single-workgroup master
    return BLUEST_OK;
}

static int master_launch(bluest_plan_t plan, int S, const int64_t *support_host, const double *cc_host, const double *s_dev,
                         const double *bg_dev, double eps_bg, double *x_dev, double *mu_dev, double tol, int maxit,
                         double *out_dev, int ncap, const int32_t *cap_model_host, const double *cap_b_host, double *nu_dev, void *stream);

extern "C" int bluest_master_newton(bluest_plan_t plan, int S, const int64_t *support_host, const double *cc_host, const double *s_dev,
                                    const double *bg_dev, double eps_bg, double *x_dev, double *mu_dev, double tol, int maxit,
                                    double *out_dev, void *stream)
{
    return master_launch(plan, S, support_host, cc_host, s_dev, bg_dev, eps_bg, x_dev, mu_dev, tol, maxit, out_dev, 0, nullptr, nullptr, nullptr, stream);
}

extern "C" int bluest_master_newton_capped(bluest_plan_t plan, int S, const int64_t *support_host, const double *cc_host, const double *s_dev,
                                           const double *bg_dev, double eps_bg, double *x_dev, double *mu_dev, double tol, int maxit,
                                           double *out_dev, int ncap, const int32_t *cap_model_host, const double *cap_b_host,
                                           double *nu_dev, void *stream)
{
    if (ncap < 0 || ncap > 64) return fail(BLUEST_ERR_ARG, "ncap=%d out of range (0..64)", ncap);
    if (ncap > 0 && (!cap_model_host || !cap_b_host || !nu_dev)) return fail(BLUEST_ERR_ARG, "null pointer");
    return master_launch(plan, S, support_host, cc_host, s_dev, bg_dev, eps_bg, x_dev, mu_dev, tol, maxit, out_dev, ncap, cap_model_host,
                         cap_b_host, nu_dev, stream);
}

static int master_launch(bluest_plan_t plan, int S, const int64_t *support_host, const double *cc_host, const double *s_dev,
                         const double *bg_dev, double eps_bg, double *x_dev, double *mu_dev, double tol, int maxit,
                         double *out_dev, int ncap, const int32_t *cap_model_host, const double *cap_b_host, double *nu_dev, void *stream)
{
    if (!plan || !support_host || !cc_host || !s_dev || !x_dev || !mu_dev || !out_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    int rc = require_gpu(); if (rc) return rc;
    if (S <= 0 || S > MASTER_SMAX) return fail(BLUEST_ERR_ARG, "support size %d out of range (1..%d)", S, MASTER_SMAX);
    if (plan->outs.size() > 64) return fail(BLUEST_ERR_ARG, "the master problem takes at most 64 outputs");
    if (eps_bg < 0.0 || eps_bg >= 1.0 || (eps_bg > 0.0 && !bg_dev)) return fail(BLUEST_ERR_ARG, "background weight / matrices inconsistent");
    const int n_out = (int)plan->outs.size(), N = plan->N;
    // host-side maps, built once per plan: global group index -> (local index per output); offsets of the size classes
    if (plan->inv_host.empty()) {
        plan->inv_host.resize(n_out);
        for (int o = 0; o < n_out; o++) {
            const OutputDesc &od = plan->outs[o];
            plan->inv_host[o].assign((size_t)plan->L, -1);
            for (int64_t t = 0; t < od.L_o; t++) plan->inv_host[o][(size_t)od.mapping[t]] = (int32_t)t;
        }
    }
    int KM = 1;
    std::vector<int32_t> kk((size_t)S);
    std::vector<int64_t> boff((size_t)n_out * S, -1);
    std::vector<uint8_t> idx;
    std::vector<std::vector<int64_t>> members((size_t)S);
    for (int j = 0; j < S; j++) {
        const int64_t gi = support_host[j];
        if (gi < 0 || gi >= plan->L || (j > 0 && gi <= support_host[j - 1])) return fail(BLUEST_ERR_ARG, "support must be strictly ascending indices in [0, L_global)");
        bool found = false;
        for (int o = 0; o < n_out; o++) {
            const OutputDesc &od = plan->outs[o];
            const int32_t li = plan->inv_host[o][(size_t)gi];
            if (li < 0) continue;
            int64_t first = 0, goff = 0, ioff = 0;         // size class of local index li
            int k = 1;
            for (; k <= od.K; k++) {
                if (li < first + od.sizes[k - 1]) break;
                first += od.sizes[k - 1]; goff += od.sizes[k - 1] * k; ioff += od.sizes[k - 1] * k * k;
            }
            boff[(size_t)o * S + j] = ioff + (li - first) * (int64_t)k * k;
            if (!found) {
                found = true;
                kk[j] = k;
                members[j].assign(od.groups.begin() + goff + (li - first) * k, od.groups.begin() + goff + (li - first + 1) * k);
            }
        }
        if (!found) return fail(BLUEST_ERR_ARG, "group %lld belongs to no output", (long long)gi);
        KM = std::max(KM, kk[j]);
    }
    const int nt = master_choose_nt(N, n_out, KM, nullptr);
    const size_t lds = master_lds_bytes(N, n_out, S, KM, nt);
    if (lds > (size_t)master_lds_limit() - MASTER_STATIC_LDS) return fail(BLUEST_ERR_ARG, "master problem needs %zu bytes of LDS (limit %d)", lds, master_lds_limit() - MASTER_STATIC_LDS);
    idx.assign((size_t)S * KM, 0);
    for (int j = 0; j < S; j++) for (int l = 0; l < kk[j]; l++) idx[(size_t)j * KM + l] = (uint8_t)members[j][l];
    // one descriptor blob: [invcov pointers][boff][cc][kk][idx]
    const size_t b_ptr = (size_t)n_out * sizeof(void *), b_off = (size_t)n_out * S * sizeof(int64_t), b_cc = (size_t)S * sizeof(double),
                 b_kk = (((size_t)S * sizeof(int32_t)) + 7) & ~(size_t)7, b_idx = ((size_t)S * KM + 7) & ~(size_t)7;
    for (int c = 0; c < ncap; c++)
        if (cap_model_host[c] < 0 || cap_model_host[c] >= N) return fail(BLUEST_ERR_ARG, "capped model %d out of range", cap_model_host[c]);
    const size_t b_cm = (((size_t)ncap * sizeof(int32_t)) + 7) & ~(size_t)7, b_cb = (size_t)ncap * sizeof(double);
    const size_t total = b_ptr + b_off + b_cc + b_kk + b_idx + b_cm + b_cb;
    std::vector<unsigned char> blob(total);
    unsigned char *h = blob.data();
    for (int o = 0; o < n_out; o++) { const double *p = plan->outs[o].d_invcov; memcpy(h + o * sizeof(void *), &p, sizeof(void *)); }
    memcpy(h + b_ptr, boff.data(), b_off);
    memcpy(h + b_ptr + b_off, cc_host, b_cc);
    memcpy(h + b_ptr + b_off + b_cc, kk.data(), (size_t)S * sizeof(int32_t));
    memcpy(h + b_ptr + b_off + b_cc + b_kk, idx.data(), (size_t)S * KM);
    if (ncap > 0) {
        memcpy(h + b_ptr + b_off + b_cc + b_kk + b_idx, cap_model_host, (size_t)ncap * sizeof(int32_t));
        memcpy(h + b_ptr + b_off + b_cc + b_kk + b_idx + b_cm, cap_b_host, b_cb);
    }
    DeviceScopeN scope(plan->device);
    if (plan->master_bytes < total) {
        // from the library's block cache (plan.hip): a hipMalloc here cost every NEW problem's first master launch 0.1-0.3 ms
        if (plan->d_master) { (void)hipStreamSynchronize((hipStream_t)stream); (void)pool_free(plan->d_master); plan->d_master = nullptr; }
        const size_t want = std::max<size_t>(total * 2, 64u << 10);
        HIP_TRY(pool_alloc(&plan->d_master, want));
        plan->master_bytes = want;
    }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(plan->d_master, h, total, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));                    // the staging vector dies with this call
    unsigned char *d = (unsigned char *)plan->d_master;
    MasterArgs A;
    A.N = N; A.n_out = n_out; A.S = S; A.KM = KM;
    A.eps_bg = eps_bg; A.tol = tol; A.act_tol = 1.0e-3; A.floor_x = 1.0e-6; A.maxit = maxit;
    A.fb = eps_bg > 0.0 ? 0.0 : 0.9;       // without the background V has kinks where a model drops out: stay inside the face
    A.invcov = (const double *const *)d;
    A.boff = (const int64_t *)(d + b_ptr);
    A.cc = (const double *)(d + b_ptr + b_off);
    A.kk = (const int32_t *)(d + b_ptr + b_off + b_cc);
    A.idx = (const uint8_t *)(d + b_ptr + b_off + b_cc + b_kk);
    A.s = s_dev; A.bg = eps_bg > 0.0 ? bg_dev : nullptr;
    A.x = x_dev; A.mu = mu_dev; A.out = out_dev;
    A.ncap = ncap;
    A.cap_model = (const int32_t *)(d + b_ptr + b_off + b_cc + b_kk + b_idx);
    A.cap_b = (const double *)(d + b_ptr + b_off + b_cc + b_kk + b_idx + b_cm);
    A.nu = nu_dev;
    // register-resident inverses for the usual sizes, the LDS version beyond 32 models.  Dynamic LDS beyond the default needs the
    // attribute (the kernel also has ~1 KB of static LDS: ask for what is needed, not for the whole limit)
#define LAUNCH_MASTER(NT)                                                                                                     \
    do {                                                                                                                      \
        static size_t granted[64] = {0};      /* per device: the attribute belongs to the device's copy of the kernel */      \
        const int dv = (plan->device >= 0 && plan->device < 64) ? plan->device : 0;                                           \
        if (lds > granted[dv] || dv != plan->device) {                                                                        \
            HIP_TRY(hipFuncSetAttribute((const void *)k_master_newton<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            granted[dv] = lds;                                                                                                \
        }                                                                                                                     \
        hipLaunchKernelGGL((k_master_newton<NT>), dim3(1), dim3(MASTER_THREADS), lds, st, A);                                 \
    } while (0)
    switch (nt) {
        case 12: LAUNCH_MASTER(12); break;
        case 20: LAUNCH_MASTER(20); break;
        case 26: LAUNCH_MASTER(26); break;
        case 32: LAUNCH_MASTER(32); break;
        case 40: LAUNCH_MASTER(40); break;
        case 48: LAUNCH_MASTER(48); break;
        case 64: LAUNCH_MASTER(64); break;
        default: LAUNCH_MASTER(0);
    }
#undef LAUNCH_MASTER
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_ma_update(bluest_plan_t plan, const double *var_dev, const int32_t *status_dev, const double *grad_dev,
                                const double *s_dev, const double *cc_dev, double p, double *x_dev, double *m_dev, void *stream)
{
    if (!plan || !var_dev || !status_dev || !grad_dev || !s_dev || !cc_dev || !x_dev || !m_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    const int n_out = (int)plan->outs.size();
    if (n_out > 64) return fail(BLUEST_ERR_ARG, "bluest_ma_update: more than 64 outputs");
    hipLaunchKernelGGL(k_ma_update, dim3((unsigned)((plan->L + 255) / 256)), dim3(256), 0,      // (64 / 128 threads per workgroup: no difference)
                       (hipStream_t)stream, plan->L, n_out, var_dev,
                       status_dev, grad_dev, plan->d_goff, plan->identity ? nullptr : plan->d_invmap, s_dev, cc_dev, p, x_dev, m_dev);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_support_point(int64_t L, int S, const int64_t *sup_dev, const double *xs_dev, const double *cc_dev, double eps,
                                    double *m_dev, void *stream)
{
    if (!sup_dev || !xs_dev || !cc_dev || !m_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (L <= 0 || S <= 0) return fail(BLUEST_ERR_ARG, "sizes out of range");
    int rc = require_gpu(); if (rc) return rc;
    hipLaunchKernelGGL(k_support_point, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L, S, sup_dev, xs_dev, cc_dev, eps, m_dev);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_price_capped(bluest_plan_t plan, const double *grad_dev, const double *mu_dev, const double *s_dev, const double *cc_dev,
                                   int S, const int64_t *sup_dev, double *c_sup_dev, double *top_val_dev, int64_t *top_idx_dev, double *y0_dev,
                                   const uint64_t *capmask_dev, const double *nu_dev, const double *master_out_dev, void *stream);

extern "C" int bluest_price(bluest_plan_t plan, const double *grad_dev, const double *mu_dev, const double *s_dev, const double *cc_dev,
                            int S, const int64_t *sup_dev, double *c_sup_dev, double *top_val_dev, int64_t *top_idx_dev, double *y0_dev,
                            void *stream)
{
    return bluest_price_capped(plan, grad_dev, mu_dev, s_dev, cc_dev, S, sup_dev, c_sup_dev, top_val_dev, top_idx_dev, y0_dev, nullptr, nullptr,
                               nullptr, stream);
}

extern "C" int bluest_price_capped(bluest_plan_t plan, const double *grad_dev, const double *mu_dev, const double *s_dev, const double *cc_dev,
                                   int S, const int64_t *sup_dev, double *c_sup_dev, double *top_val_dev, int64_t *top_idx_dev, double *y0_dev,
                                   const uint64_t *capmask_dev, const double *nu_dev, const double *master_out_dev, void *stream)
{
    if (capmask_dev && (!nu_dev || !master_out_dev)) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!plan || !grad_dev || !mu_dev || !s_dev || !cc_dev || !sup_dev || !c_sup_dev || !top_val_dev || !top_idx_dev || !y0_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    const int n_out = (int)plan->outs.size();
    hipLaunchKernelGGL(k_price, dim3(PRICE_BLOCKS), dim3(256), 0, (hipStream_t)stream, plan->L, n_out, grad_dev, plan->d_goff,
                       plan->identity ? nullptr : plan->d_invmap, mu_dev, s_dev, cc_dev, S, sup_dev, c_sup_dev, top_val_dev, top_idx_dev, plan->d_v, plan->N, y0_dev,
                       (const unsigned long long *)capmask_dev, nu_dev, master_out_dev);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}
