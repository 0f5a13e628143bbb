// matfree.hip -- the MATRIX-FREE evaluation: Phi(m), V and grad V without the stored per-group inverses.
//
// BASELINE.json's north star asks for this form of the path: the inverse of every group's covariance block C[g, g]
// (bluest/sap.py:69-79 builds and stores them; bluest/cmisc.cpp:25-40,58-72 stream them) is RECOMPUTED in registers each
// time it is needed -- lane = group, the N x N covariance staged in LDS, an in-register Cholesky factorisation of the k x k
// block -- so an evaluation reads the k model indices of a group (bytes), its allocation m_g and nothing else: 3.4 MB instead
// of 153 MB at K_tot = 245 505 (the matrix-free floor).
//
//   k_phi_matfree     Phi pass.  One wavefront per tile of 64 groups: C block -> L L^T -> (L L^T)^-1 -> m_g times its
//                     k(k+1)/2 packed entries, scatter-ADDED into an accumulator of Phi_o that is PRIVATE TO THE WAVEFRONT in
//                     LDS (ds_add_f64; the model-wise max |m_g| rides along with ds_max_f64).  No atomic leaves the
//                     wavefront: the lanes of one instruction are resolved by the LDS in lane order, the instructions of a
//                     wavefront in program order, and the wavefronts' accumulators are summed in wavefront order, the
//                     workgroups' partials by k_mf_reduce in workgroup order -- the sums have ONE order per plan and are
//                     bit-reproducible (tests/test_gpu_matfree.py compares two runs bit for bit).
//   k_mf_reduce       partials of the workgroups -> the Phi record (N*N sums + the sampled-model flags, the layout of
//                     k_fold_to_record: what the multi-GPU exchange all-reduces and k_solve_grad / k_solve_from_record read).
//   k_solve_grad_mf   record -> Phi in LDS -> one wavefront eliminates (solve.hpp) WHILE the tile wavefronts factorise their
//                     groups' blocks; after the barrier they forward-substitute v[g]: grad_g = -|L^-1 v[g]|^2.
//
// Which plans: every output was given by its covariance (bluest_plan_add_output_cov), group sizes <= 8, at most 64 models,
// and every block is safely positive definite (k_mf_check: smallest Cholesky pivot > 1e-10 of its diagonal entry) -- the
// reference's inverse is an SVD pseudo-inverse with cut-off 1e-15 (bluest/sap.py:74), which the Cholesky inverse equals to
// rounding on such blocks only; the stored path stays for all others (the `singular` and paper fixtures among them).
// When: BLUEST_MATFREE=1 forces it on eligible plans, =0 forbids it, unset: plans whose stored streams exceed
// MF_AUTO_BYTES (the evaluation is then bound by bytes, profiles/r04_matfree_ab.txt).
#include "plan.hpp"

#ifndef MF_ABLATE          // experiment builds only (timing; the ablated builds compute wrong numbers): 1 no LDS adds, 2 no run sums,
#define MF_ABLATE 0        // 3 no inverse (the factor's entries are scattered), 4 tiles load their inputs and do nothing else
#endif
#define MF_KMAX 8
#define MF_TILE_WAVES 15                       // tile wavefronts of k_solve_grad_mf (+ the solving one)
static const int64_t MF_AUTO_BYTES = 64ll << 20;

struct MfTile {            // 64 consecutive groups of one size of one output (8 bytes: one scalar load)
    int32_t first;         // local index (within the output) of the first group
    int16_t n, k;          // groups in the tile (<= 64), their size
};

struct MfState {           // hangs off bluest_plan_s::mf
    void *blob = nullptr;  // one device allocation
    MfTile *d_tiles = nullptr;
    int32_t *d_tile_begin = nullptr;           // [n_out + 1]
    double *d_C = nullptr;                     // [n_out][N * N]
    const uint64_t **d_groups = nullptr;       // [n_out] device pointers to the packed model indices (8 bytes per group)
    uint64_t *d_packed = nullptr;              // the packed lists (one per distinct group list)
    int32_t *d_map = nullptr;                  // local -> global group index, concatenated like the gradient (NULL: identity)
    double *d_partial = nullptr;               // [n_out][wgs][nsym]
    double *d_amax = nullptr;                  // [n_out][wgs][N + 1]: model-wise max |m|, then their maximum
    double *d_rec = nullptr;                   // [n_out][N*N + 2N + 1]  (evaluations that never leave this GPU)
    int32_t *d_flag = nullptr;                 // k_mf_check: number of blocks that are not safely positive definite
    std::vector<int32_t> tile_begin;           // host copy
    int wgs = 0, nw = 8, nsym = 0;
    int64_t n_tiles = 0;
    size_t lds_phi = 0, lds_grad = 0;
};

__host__ __device__ constexpr int mf_ke(int K) { return K * (K + 1) / 2; }
__device__ __forceinline__ int mf_sym(int a, int b, int N) { const int lo = a < b ? a : b, hi = a < b ? b : a; return lo * N - lo * (lo - 1) / 2 + (hi - lo); }

__device__ __forceinline__ double mf_rsqrt(double d)
{   // 1 / sqrt(d): v_rsq_f64 seed (~2^-26) + two Newton steps y <- y (1.5 - 0.5 d y^2)
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// x of the lane CTRL places before this one in its 16-lane row (0x111/2/4/8 = row_shr:1/2/4/8); 0 where the row ends
template <int CTRL>
__device__ __forceinline__ double dpp_shr(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Cholesky factor of the packed lower triangle a (entry (i, j), j <= i, at i(i+1)/2 + j), in place; r[j] = 1 / L_jj.
// Returns the smallest pivot relative to its diagonal entry (<= 0: not positive definite).
template <int K>
__device__ __forceinline__ double mf_chol(double (&a)[mf_ke(K)], double (&r)[K])
{
    double worst = 1.0;
#pragma unroll
    for (int j = 0; j < K; j++) {
        double d = a[j * (j + 1) / 2 + j];
        const double d0 = d;
#pragma unroll
        for (int p = 0; p < j; p++) d = fma(-a[j * (j + 1) / 2 + p], a[j * (j + 1) / 2 + p], d);
        worst = fmin(worst, d / d0);
        const double rj = mf_rsqrt(d);
        r[j] = rj;
        a[j * (j + 1) / 2 + j] = d * rj;
#pragma unroll
        for (int i = j + 1; i < K; i++) {
            double s = a[i * (i + 1) / 2 + j];
#pragma unroll
            for (int p = 0; p < j; p++) s = fma(-a[i * (i + 1) / 2 + p], a[j * (j + 1) / 2 + p], s);
            a[i * (i + 1) / 2 + j] = s * rj;
        }
    }
    return worst;
}

// the model indices of lane's group (one byte each in the packed word) and its covariance block from LDS; lanes without a group get
// the identity
template <int K>
__device__ __forceinline__ void mf_load_block(uint64_t pk, bool valid, const double *__restrict__ Cs, int N,
                                              int (&idx)[K], double (&a)[mf_ke(K)])
{
#pragma unroll
    for (int j = 0; j < K; j++) idx[j] = valid ? (int)((pk >> (8 * j)) & 0xffull) : 0;
#pragma unroll
    for (int i = 0; i < K; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) a[i * (i + 1) / 2 + j] = valid ? Cs[idx[i] * N + idx[j]] : (i == j ? 1.0 : 0.0);
}

// Phi contribution of one tile (K static): into this wavefront's accumulators
template <int K>
__device__ __forceinline__ void mf_phi_tile(int n_valid, uint64_t pk, double mg, const double *__restrict__ Cs, int N,
                                            double *__restrict__ acc, double *__restrict__ amx, int lane)
{
    const bool valid = lane < n_valid;
    int idx[K];
    double a[mf_ke(K)], r[K];
    mf_load_block<K>(pk, valid, Cs, N, idx, a);
    if (MF_ABLATE == 4) { if (a[0] == 1.2345 && mg == 5.4321) acc[idx[0]] = 1.0; return; }
    if (MF_ABLATE != 3) (void)mf_chol<K>(a, r);
    else { _Pragma("unroll") for (int i = 0; i < K; i++) r[i] = 1.0; }
    // M = L^-1 (lower) in place of L: M_jj = r_j, M_ij = -r_i sum_{p=j}^{i-1} L_ip M_pj  (column by column, rows top down)
    double M[mf_ke(K)];
#pragma unroll
    for (int j = 0; j < K; j++) {
        M[j * (j + 1) / 2 + j] = r[j];
#pragma unroll
        for (int i = j + 1; i < K; i++) {
            double s = 0.0;
#pragma unroll
            for (int p = j; p < i; p++) s = fma(a[i * (i + 1) / 2 + p], M[p * (p + 1) / 2 + j], s);
            M[i * (i + 1) / 2 + j] = -r[i] * s;
        }
    }
    // (L L^T)^-1 = M^T M: entry (i, j), j <= i, = sum_{q >= i} M_qi M_qj; times m_g into Phi[idx_i, idx_j].
    // The groups of a tile are consecutive in lexicographic order, so a destination (idx_i, idx_j) is shared by RUNS of lanes (all 64
    // of them when neither position is one of the last two): left to the LDS such an add serialises 64 ways (measured: 37 us for
    // the Phi pass at K_tot = 245 505).  So the products are first summed along the runs inside every 16-lane DPP row -- a
    // segmented scan, the segment heads being the lanes where the prefix (idx_0 .. idx_i) changes -- and only the LAST lane of a
    // run adds the run's total: at most a few lanes per instruction meet at one address.  The order of the additions is fixed by
    // the plan (lane order inside a run, then the LDS's lane order, then program order).
    // All run sums first, as straight-line code (21 independent scans at k = 6 for the scheduler to interleave: behind a branch per
    // destination they ran one after the other, each a chain of four dependent DPP steps), then the adds.
    const double am = fabs(mg);
    unsigned head = (lane & 15) == 0 || !valid ? 1u : 0u;      // cumulative over the positions: prefix (idx_0 .. idx_i) differs from lane - 1's
    double tot[mf_ke(K)], mxs[K];
    bool tails[K];
#pragma unroll
    for (int i = 0; i < K; i++) {
        const int prev = __builtin_amdgcn_mov_dpp(idx[i], 0x111, 0xf, 0xf, true);          // row_shr:1
        head |= (prev != idx[i]) ? 1u : 0u;
        // multipliers of the four scan steps for runs of this prefix, and the run's last lane
        unsigned f = head;
        double nf[4];
        nf[0] = f ? 0.0 : 1.0;
        f |= (unsigned)__builtin_amdgcn_mov_dpp((int)f, 0x111, 0xf, 0xf, true);
        nf[1] = f ? 0.0 : 1.0;
        f |= (unsigned)__builtin_amdgcn_mov_dpp((int)f, 0x112, 0xf, 0xf, true);           // row_shr:2
        nf[2] = f ? 0.0 : 1.0;
        f |= (unsigned)__builtin_amdgcn_mov_dpp((int)f, 0x114, 0xf, 0xf, true);           // row_shr:4
        nf[3] = f ? 0.0 : 1.0;
        // (the DPP move first, for ALL lanes: behind an || it would run with lane 15 masked off and lane 14 would read a zero)
        const int next_head = __builtin_amdgcn_mov_dpp((int)head, 0x101, 0xf, 0xf, true);      // row_shl:1: the next lane starts a run
        tails[i] = ((lane & 15) == 15) | (next_head != 0);
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double sij = 0.0;
#pragma unroll
            for (int q = i; q < K; q++) sij = fma(M[q * (q + 1) / 2 + i], M[q * (q + 1) / 2 + j], sij);
            double v = mg * sij;
            if (MF_ABLATE != 2) {
                v = fma(dpp_shr<0x111>(v), nf[0], v);
                v = fma(dpp_shr<0x112>(v), nf[1], v);
                v = fma(dpp_shr<0x114>(v), nf[2], v);
                v = fma(dpp_shr<0x118>(v), nf[3], v);
            }
            tot[i * (i + 1) / 2 + j] = v;
        }
        // max |m_g| over the groups containing model idx_i: same runs (|m| >= 0, lanes outside the row read 0)
        double mx = am;
        mx = fmax(mx, dpp_shr<0x111>(mx) * nf[0]);
        mx = fmax(mx, dpp_shr<0x112>(mx) * nf[1]);
        mx = fmax(mx, dpp_shr<0x114>(mx) * nf[2]);
        mx = fmax(mx, dpp_shr<0x118>(mx) * nf[3]);
        mxs[i] = mx;
    }
    if (MF_ABLATE == 1) { double z = 0.0; _Pragma("unroll") for (int e = 0; e < mf_ke(K); e++) z += tot[e]; if (z == 1.2345) acc[0] = z; return; }
#pragma unroll
    for (int i = 0; i < K; i++) {
        if (tails[i]) {
#pragma unroll
            for (int j = 0; j <= i; j++) {
                const double v = tot[i * (i + 1) / 2 + j];
                if (v != 0.0) atomicAdd(&acc[mf_sym(idx[i], idx[j], N)], v);      // ds_add_f64, wavefront-private accumulator
            }
            if (mxs[i] > 0.0) __hip_atomic_fetch_max(&amx[idx[i]], mxs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // ds_max_f64
        }
    }
}

struct MfArgs {
    int N, nsym, n_out, wgs;
    const MfTile *tiles;
    const int32_t *tile_begin;
    const double *C;
    const uint64_t *const *groups;
    const int32_t *map;
    const int64_t *goff;
};

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_phi_matfree(const MfArgs A, const double *__restrict__ m, double *__restrict__ partial,
                                                         double *__restrict__ amax)
{
    extern __shared__ double mf_sm[];          // [N*N covariance][NW * nsym sums][NW * N maxima]
    const int N = A.N, nsym = A.nsym, o = blockIdx.y, b = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double *Cs = mf_sm, *acc = Cs + N * N, *amx = acc + NW * nsym;
    for (int t = tid; t < N * N; t += 64 * NW) Cs[t] = A.C[(int64_t)o * N * N + t];
    for (int t = tid; t < NW * nsym; t += 64 * NW) acc[t] = 0.0;
    for (int t = tid; t < NW * N; t += 64 * NW) amx[t] = 0.0;
    const int tb = A.tile_begin[o], te = A.tile_begin[o + 1];
    const uint64_t *groups = A.groups[o];
    const int64_t map_off = A.goff[o];
    double *my_acc = acc + wave * nsym, *my_amx = amx + wave * N;
    // tiles of this output, strided over the workgroups and then over the wavefronts (the sizes mix evenly).  The inputs of the NEXT
    // tile (descriptor: a scalar load; packed indices and m: one load each per lane) are in flight while this one is worked on
    auto fetch = [&](int t, MfTile &td, uint64_t &pk, double &mg) {
        td = A.tiles[__builtin_amdgcn_readfirstlane(t)];
        const bool v = lane < td.n;
        const int64_t li = (int64_t)td.first + lane;
        pk = v ? groups[li] : 0ull;
        mg = v ? m[A.map ? (int64_t)A.map[map_off + li] : li] : 0.0;
    };
    const int stride = NW * A.wgs;
    int t = tb + b + wave * A.wgs;
    MfTile td, tdn;
    uint64_t pk = 0ull, pkn = 0ull;
    double mg = 0.0, mgn = 0.0;
    td.first = 0; td.n = 0; td.k = 0; tdn = td;
    if (t < te) fetch(t, td, pk, mg);
    __syncthreads();                               // (the covariance and the zeroed accumulators)
    while (t < te) {
        const int tn = t + stride;
        if (tn < te) fetch(tn, tdn, pkn, mgn);
#define MFP(KK) case KK: mf_phi_tile<KK>(td.n, pk, mg, Cs, N, my_acc, my_amx, lane); break;
        switch (__builtin_amdgcn_readfirstlane((int)td.k)) { MFP(1) MFP(2) MFP(3) MFP(4) MFP(5) MFP(6) MFP(7) MFP(8) default: break; }
#undef MFP
        td = tdn; pk = pkn; mg = mgn; t = tn;
    }
    __syncthreads();
    double *pout = partial + ((int64_t)o * A.wgs + b) * nsym;
    for (int d = tid; d < nsym; d += 64 * NW) {
        double s = acc[d];
#pragma unroll
        for (int w = 1; w < NW; w++) s += acc[w * nsym + d];      // wavefront order: fixed
        pout[d] = s;
    }
    // model-wise maxima of this workgroup, and their maximum in slot N (the "some group has |m| >= 0.05" test of bluest/misc.py:464)
    double am = 0.0;
    if (tid < N) {
        am = amx[tid];
#pragma unroll
        for (int w = 1; w < NW; w++) am = fmax(am, amx[w * N + tid]);
        amax[((int64_t)o * A.wgs + b) * (N + 1) + tid] = am;
    }
    if (tid < 64) {      // N <= 48: one wavefront holds every model
        am = fast_max(am);
        if (tid == 0) amax[((int64_t)o * A.wgs + b) * (N + 1) + N] = am;
    }
}

// partials of the workgroups -> record of every output.  ONE WAVEFRONT PER RESULT, four per workgroup (block 256): the first nsym
// results are the destinations, the next N + 1 the model-wise maxima and their maximum (-> the sampled-model flags of
// bluest/misc.py:453-457, :464).  Lane l folds workgroups l, l + 64, .. with four independent loads in flight (sixteen dependent
// round trips, as a 16-lane version had them, cost 10 us at 256 workgroups); the lanes combine with the fixed DPP reduction.
__global__ __launch_bounds__(256) void k_mf_reduce(int N, int nsym, int wgs, const double *__restrict__ partial, const double *__restrict__ amax,
                                                   double *__restrict__ rec)
{
    const int o = blockIdx.y, tid = threadIdx.x, ln = tid & 63, d = blockIdx.x * 4 + (tid >> 6);
    const int reclen = N * N + 2 * N + 1;
    double *r = rec + (int64_t)o * reclen;
    if (d < nsym) {
        const double *p = partial + (int64_t)o * wgs * nsym + d;
        double s = 0.0;
        for (int w0 = ln; w0 < wgs; w0 += 256) {
            double v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = (w0 + 64 * i < wgs) ? p[(int64_t)(w0 + 64 * i) * nsym] : 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) s += v[i];
        }
        s = fast_sum(s);
        if (ln == 0) {
            // destination d = (a, b), a <= b, in the row-major order of the upper triangle
            int a = 0, rem = d;
            while (rem >= N - a) { rem -= N - a; a++; }
            const int b = a + rem;
            r[a * N + b] = s;
            r[b * N + a] = s;
        }
    } else if (d <= nsym + N) {
        const int a = d - nsym;                  // model a, or a == N: the maximum over all models
        double am = 0.0;
        for (int w0 = ln; w0 < wgs; w0 += 256) {
            double v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = (w0 + 64 * i < wgs) ? amax[((int64_t)o * wgs + w0 + 64 * i) * (N + 1) + a] : 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) am = fmax(am, v[i]);
        }
        am = fast_max(am);
        if (ln == 0) {
            if (a < N) { r[N * N + a] = (am > 1.0e-6) ? 1.0 : 0.0; r[N * N + N + a] = (am > 0.0) ? 1.0 : 0.0; }
            else r[N * N + 2 * N] = (am >= 0.05) ? 1.0 : 0.0;
        }
    }
}

// gradient of one tile: factor before the barrier, forward substitution after it
template <int K>
__device__ __forceinline__ double mf_quad(const double (&a)[mf_ke(K)], const double (&r)[K], const int (&idx)[K], const double *__restrict__ v)
{   // |L^-1 v_g|^2 = v_g^T (L L^T)^-1 v_g
    double y[K], q = 0.0;
#pragma unroll
    for (int i = 0; i < K; i++) {
        double s = v[idx[i]];
#pragma unroll
        for (int p = 0; p < i; p++) s = fma(-a[i * (i + 1) / 2 + p], y[p], s);
        y[i] = s * r[i];
        q = fma(y[i], y[i], q);
    }
    return q;
}

template <int NT, int KU>
__global__ __launch_bounds__(64 * (MF_TILE_WAVES + 1)) void k_solve_grad_mf(const MfArgs A, const double *__restrict__ rec, double delta,
                                                                           const RowDesc *__restrict__ rows, FoldReg reg, const double2 *__restrict__ partial,
                                                                           const int32_t *__restrict__ wg_begin, int bpo,
                                                                           double *__restrict__ var, double *__restrict__ v_ws,
                                                                           int32_t *__restrict__ status, double *__restrict__ grad, const MaTail ma)
{
    constexpr int NTHREADS = 64 * (MF_TILE_WAVES + 1);
    __shared__ SolveLds<NT> lds;
    extern __shared__ double mf_cs[];          // [N*N] covariance of this workgroup's output
    const int N = A.N, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // which output: the workgroups of output o are [wg_begin[o], wg_begin[o + 1]) -- arithmetic when every output has the same number
    // (bpo > 0, the usual case: a search through the table is a chain of dependent loads in front of everything else)
    int o = 0, b = 0;
    if (bpo > 0) { o = blockIdx.x / bpo; b = blockIdx.x - o * bpo; }
    else {
        while (o + 1 < A.n_out && (int)blockIdx.x >= wg_begin[o + 1]) o++;
        b = blockIdx.x - wg_begin[o];
    }
    const bool first = b == 0;
    // Phi_o from the record (matrix-free Phi pass, all-reduced record of a sharded plan), or -- rec == NULL -- folded from the chunk
    // partials of the STORED Phi pass (plans whose Phi layout fits the L2s keep that pass: it is a pure stream there, and without the
    // gradient pass's tile stream next to it, it stays L2-resident from step to step)
    const double *rec_o = rec ? rec + (int64_t)o * (N * N + 2 * N + 1) : nullptr;
    if (N < NT) { clear_pads(lds, N, tid, NTHREADS); __syncthreads(); }
    for (int t = tid; t < N * N; t += NTHREADS) mf_cs[t] = A.C[(int64_t)o * N * N + t];
    if (rec_o) { for (int t = tid; t < N * N; t += NTHREADS) lds.at(t / N, t % N) = rec_o[t]; }
    else fold_rows<NT>(lds, N, rows, o * A.nsym, A.nsym, partial, tid, NTHREADS, reg);
    const int t_mine = A.tile_begin[o] + b * MF_TILE_WAVES + wave - 1;
    MfTile td;
    td.first = 0; td.n = 0; td.k = 0;
    if (wave > 0 && t_mine < A.tile_begin[o + 1]) td = A.tiles[t_mine];
    __syncthreads();
    // the tile wavefronts keep their factor in registers across the barrier (KU = largest group size of the plan: 21 + 6 doubles at 6)
    int idx[KU];
    double a[mf_ke(KU)], r[KU];
    const bool valid = lane < td.n;
    const uint64_t gl = valid ? A.groups[o][(int64_t)td.first + lane] : 0ull;      // the group's model indices, a byte each
    if (wave == 0) {
        bool s1, s2, big;
        if (rec_o) {
            s1 = lane < N && rec_o[N * N + lane] > 0.0;
            s2 = lane < N && rec_o[N * N + N + lane] > 0.0;
            big = rec_o[N * N + 2 * N] > 0.0;
        } else {
            const double am = (lane < N) ? lds.amax[lane] : 0.0;
            s1 = am > 1.0e-6;
            s2 = am > 0.0;
            big = __ballot(am >= 0.05) != 0ull;          // max |m| >= 0.05 (bluest/misc.py:464)
        }
        double V = 0.0;
        int32_t st = 0;
        solve_wave<NT>(lds, N, delta, s1, s2, big, true, &V, lds.vout, &st, lane);
        if (lane == 0) { lds.status = st; if (ma.x) lds.scratch[0] = V; }
        if (first) {
            if (lane == 0) { var[o] = V; status[o] = st; }
            if (lane < N) v_ws[(int64_t)o * N + lane] = lds.vout[lane];
        }
    } else {
#define MFC(KK) case KK: if (KK <= KU) {                                                                      \
            int ik[KK]; double ak[mf_ke(KK)], rk[KK];                                                          \
            mf_load_block<KK>(gl, valid, mf_cs, N, ik, ak);                                                    \
            (void)mf_chol<KK>(ak, rk);                                                                         \
            _Pragma("unroll") for (int i = 0; i < KK; i++) if (i < KU) { idx[i] = ik[i]; r[i] = rk[i]; }       \
            _Pragma("unroll") for (int i = 0; i < mf_ke(KK); i++) if (i < mf_ke(KU)) a[i] = ak[i];             \
            } break;
        switch (td.k) { MFC(1) MFC(2) MFC(3) MFC(4) MFC(5) MFC(6) MFC(7) MFC(8) default: break; }
#undef MFC
    }
    __syncthreads();
    if (wave == 0 || td.k == 0) return;
    const bool inf = lds.status == BLUEST_EVAL_INF;
    double q = 0.0;
#define MFQ(KK) case KK: if (KK <= KU) {                                                                      \
        int ik[KK]; double ak[mf_ke(KK)], rk[KK];                                                              \
        _Pragma("unroll") for (int i = 0; i < KK; i++) { ik[i] = i < KU ? idx[i < KU ? i : 0] : 0; rk[i] = i < KU ? r[i < KU ? i : 0] : 0.0; } \
        _Pragma("unroll") for (int i = 0; i < mf_ke(KK); i++) ak[i] = i < mf_ke(KU) ? a[i < mf_ke(KU) ? i : 0] : 0.0;  \
        q = mf_quad<KK>(ak, rk, ik, lds.vout);                                                                 \
        } break;
    switch (td.k) { MFQ(1) MFQ(2) MFQ(3) MFQ(4) MFQ(5) MFQ(6) MFQ(7) MFQ(8) default: break; }
#undef MFQ
    if (valid && !ma.x) grad[A.goff[o] + td.first + lane] = inf ? INFINITY : -q;
    else if (valid && lds.status == BLUEST_EVAL_OK) {      // single output, phase 1 of the second-order finish: k_ma_update's arithmetic
        const int64_t i = (int64_t)td.first + lane;
        const double so = ma.s[0], cci = ma.cc[i];
        const double xn = ma.x[i] * cci * ((1.0 / so) * q) / (lds.scratch[0] / so);
        ma.x[i] = xn;
        ma.m[i] = cci * xn;
    }
}

// gradient pass alone from a given v (bluest_plan_grad on a matrix-free plan): one wavefront per tile, four per workgroup; the same
// factor and the same forward substitution as the fused kernel: identical bits
__global__ __launch_bounds__(256) void k_grad_mf(const MfArgs A, const double *__restrict__ v, const int32_t *__restrict__ status, double *__restrict__ grad)
{
    extern __shared__ double mf_gs[];          // [N*N covariance][N v]
    const int N = A.N, o = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double *vs = mf_gs + N * N;
    for (int t = tid; t < N * N; t += 256) mf_gs[t] = A.C[(int64_t)o * N * N + t];
    if (tid < N) vs[tid] = v[(int64_t)o * N + tid];
    __syncthreads();
    const bool inf = status[o] == BLUEST_EVAL_INF;
    for (int t = A.tile_begin[o] + blockIdx.x * 4 + wave; t < A.tile_begin[o + 1]; t += gridDim.x * 4) {
        const MfTile td = A.tiles[t];
        const bool valid = lane < td.n;
        const uint64_t gl = valid ? A.groups[o][(int64_t)td.first + lane] : 0ull;
        double q = 0.0;
#define MFG(KK) case KK: { int ik[KK]; double ak[mf_ke(KK)], rk[KK]; mf_load_block<KK>(gl, valid, mf_gs, N, ik, ak); (void)mf_chol<KK>(ak, rk); q = mf_quad<KK>(ak, rk, ik, vs); break; }
        switch (td.k) { MFG(1) MFG(2) MFG(3) MFG(4) MFG(5) MFG(6) MFG(7) MFG(8) default: break; }
#undef MFG
        if (valid) grad[A.goff[o] + td.first + lane] = inf ? INFINITY : -q;
    }
}

// the k model indices of every group of one size (bytes, the plan's list) -> one 8-byte word per group: one coalesced load per lane
__global__ __launch_bounds__(256) void k_mf_pack(const uint8_t *__restrict__ gk, int k, int64_t Lk, uint64_t *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= Lk) return;
    uint64_t w = 0ull;
    for (int j = 0; j < k; j++) w |= (uint64_t)gk[g * k + j] << (8 * j);
    out[g] = w;
}

// eligibility: every block safely positive definite (see the header)
__global__ __launch_bounds__(256) void k_mf_check(const MfArgs A, int32_t *__restrict__ flag)
{
    extern __shared__ double mf_cc[];
    const int N = A.N, o = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int t = tid; t < N * N; t += 256) mf_cc[t] = A.C[(int64_t)o * N * N + t];
    __syncthreads();
    int bad = 0;
    for (int t = A.tile_begin[o] + blockIdx.x * 4 + wave; t < A.tile_begin[o + 1]; t += gridDim.x * 4) {
        const MfTile td = A.tiles[t];
        const bool valid = lane < td.n;
        const uint64_t gl = valid ? A.groups[o][(int64_t)td.first + lane] : 0ull;
        double worst = 1.0;
#define MFK(KK) case KK: { int ik[KK]; double ak[mf_ke(KK)], rk[KK]; mf_load_block<KK>(gl, valid, mf_cc, N, ik, ak); worst = mf_chol<KK>(ak, rk); break; }
        switch (td.k) { MFK(1) MFK(2) MFK(3) MFK(4) MFK(5) MFK(6) MFK(7) MFK(8) default: worst = 0.0; }
#undef MFK
        if (valid && !(worst > 1.0e-10)) bad++;
    }
    if (bad) atomicAdd(flag, bad);
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
void mf_release(bluest_plan_s *p)
{
    MfState *S = reinterpret_cast<MfState *>(p->mf);
    if (!S) return;
    if (S->blob) (void)pool_free(S->blob);
    if (p->mf_wg_begin_dev) { (void)pool_free(p->mf_wg_begin_dev); p->mf_wg_begin_dev = nullptr; }
    delete S;
    p->mf = nullptr;
    p->matfree = false;
    p->mf_gradient = false;
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// called at the end of bluest_plan_finalize.  Leaves plan->matfree = false (and no state) when the plan does not qualify.
int mf_finalize(bluest_plan_t plan)
{
    plan->matfree = false;
    plan->mf_gradient = false;
    // BLUEST_MATFREE: 0 stored inverses everywhere, 1 matrix-free Phi AND gradient, 2 matrix-free gradient behind the stored Phi pass
    // (an A/B mode: 3 % at the headline size, 6 % at K_tot = 245505 against 12 % for the full form, profiles/r04_matfree_ab.txt);
    // unset: matrix-free where the stored streams exceed MF_AUTO_BYTES -- there the evaluation is bound by bytes --, stored below
    const char *env = getenv("BLUEST_MATFREE");
    const int mode = env ? atoi(env) : -1;
    if (mode == 0) return BLUEST_OK;
    if (mode < 0 && plan->phi_bytes + plan->grad_bytes < MF_AUTO_BYTES) return BLUEST_OK;
    const bool full = mode != 2;
    const int n_out = (int)plan->outs.size(), N = plan->N;
    if (N > 48 || n_out < 1) return BLUEST_OK;      // (LDS of k_solve_grad_mf: the elimination's matrix + the covariance)
    for (const auto &od : plan->outs) if (od.h_C.size() != (size_t)N * N || od.K > MF_KMAX || !od.d_groups) return BLUEST_OK;
    MfState *S = new MfState();
    S->nsym = N * (N + 1) / 2;
    S->nw = N <= 32 ? 8 : 4;
    // tiles
    std::vector<MfTile> tiles;
    S->tile_begin.assign(n_out + 1, 0);
    for (int o = 0; o < n_out; o++) {
        const OutputDesc &od = plan->outs[o];
        S->tile_begin[o] = (int32_t)tiles.size();
        int64_t first = 0;
        for (int k = 1; k <= od.K; k++) {
            const int64_t Lk = od.sizes[k - 1];
            for (int64_t t = 0; t < Lk; t += 64) {
                MfTile td;
                td.first = (int32_t)(first + t); td.n = (int16_t)std::min<int64_t>(64, Lk - t); td.k = (int16_t)k;
                tiles.push_back(td);
            }
            first += Lk;
        }
    }
    S->tile_begin[n_out] = (int32_t)tiles.size();
    S->n_tiles = (int64_t)tiles.size();
    // workgroups of the Phi pass per output: all compute units busy, a tile or more per wavefront
    static int cus_of[64] = {0};           // (hipGetDeviceProperties costs tens of milliseconds: once per device)
    const int dv = (plan->device >= 0 && plan->device < 64) ? plan->device : 0;
    if (!cus_of[dv]) {
        int c = 0;
        cus_of[dv] = (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, plan->device) == hipSuccess && c > 0) ? c : 256;
    }
    const int cus = cus_of[dv];
    int64_t most = 0;
    for (int o = 0; o < n_out; o++) most = std::max<int64_t>(most, S->tile_begin[o + 1] - S->tile_begin[o]);
    S->wgs = (int)std::max<int64_t>(1, std::min<int64_t>((cus + n_out - 1) / n_out, (most + S->nw - 1) / S->nw));
    const char *wenv = getenv("BLUEST_MATFREE_WGS");
    if (wenv && atoi(wenv) >= 1) S->wgs = atoi(wenv);
    // mappings (identity plans need none)
    std::vector<int32_t> map;
    if (!plan->identity) {
        map.resize((size_t)plan->grad_len);
        for (int o = 0; o < n_out; o++) {
            const OutputDesc &od = plan->outs[o];
            for (int64_t t = 0; t < od.L_o; t++) map[(size_t)(plan->grad_off[o] + t)] = (int32_t)od.mapping[(size_t)t];
        }
    }
    const int reclen = N * N + 2 * N + 1;
    // packed model indices: one list per DISTINCT group list (outputs that share output 0's list share its packing)
    std::vector<int64_t> pk_off((size_t)n_out, 0);
    int64_t pk_words = 0;
    for (int o = 0; o < n_out; o++) {
        if (o > 0 && plan->outs[o].d_groups == plan->outs[0].d_groups) { pk_off[(size_t)o] = 0; continue; }
        pk_off[(size_t)o] = pk_words;
        pk_words += plan->outs[o].L_o;
    }
    const size_t b_pk = al((size_t)pk_words * 8);
    const size_t b_tiles = al(tiles.size() * sizeof(MfTile)), b_tb = al((n_out + 1) * sizeof(int32_t)), b_C = al((size_t)n_out * N * N * 8),
                 b_gp = al(n_out * sizeof(void *)), b_map = al(map.size() * sizeof(int32_t)), b_part = al((size_t)n_out * S->wgs * S->nsym * sizeof(double)),
                 b_amax = al((size_t)n_out * S->wgs * (N + 1) * 8), b_rec = al((size_t)n_out * reclen * 8), b_flag = al(sizeof(int32_t));
    const size_t total = b_tiles + b_tb + b_C + b_gp + b_map + b_part + b_amax + b_rec + b_flag + b_pk;
    DeviceScopeN scope(plan->device);
    hipError_t e = pool_alloc(&S->blob, total);
    if (e != hipSuccess) { delete S; HIP_TRY(e); }
    unsigned char *d = reinterpret_cast<unsigned char *>(S->blob);
    S->d_tiles = reinterpret_cast<MfTile *>(d); d += b_tiles;
    S->d_tile_begin = reinterpret_cast<int32_t *>(d); d += b_tb;
    S->d_C = reinterpret_cast<double *>(d); d += b_C;
    S->d_groups = reinterpret_cast<const uint64_t **>(d); d += b_gp;
    S->d_map = map.empty() ? nullptr : reinterpret_cast<int32_t *>(d); d += b_map;
    S->d_partial = reinterpret_cast<double *>(d); d += b_part;
    S->d_amax = reinterpret_cast<double *>(d); d += b_amax;
    S->d_rec = reinterpret_cast<double *>(d); d += b_rec;
    S->d_flag = reinterpret_cast<int32_t *>(d); d += b_flag;
    S->d_packed = reinterpret_cast<uint64_t *>(d);
    plan->mf = S;
    std::vector<const uint64_t *> gp((size_t)n_out);
    for (int o = 0; o < n_out; o++) {
        gp[(size_t)o] = S->d_packed + pk_off[(size_t)o];
        if (o > 0 && plan->outs[o].d_groups == plan->outs[0].d_groups) continue;
        const OutputDesc &od = plan->outs[o];
        int64_t first = 0, gb = 0;
        for (int k = 1; k <= od.K; k++) {
            const int64_t Lk = od.sizes[k - 1];
            if (Lk > 0) hipLaunchKernelGGL(k_mf_pack, dim3((unsigned)((Lk + 255) / 256)), dim3(256), 0, 0, od.d_groups + gb, k, Lk, S->d_packed + pk_off[(size_t)o] + first);
            first += Lk; gb += Lk * k;
        }
    }
    hipError_t err = hipMemcpy(S->d_tiles, tiles.data(), tiles.size() * sizeof(MfTile), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(S->d_tile_begin, S->tile_begin.data(), (n_out + 1) * sizeof(int32_t), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy((void *)S->d_groups, gp.data(), n_out * sizeof(void *), hipMemcpyHostToDevice);
    if (err == hipSuccess && !map.empty()) err = hipMemcpy(S->d_map, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    for (int o = 0; o < n_out && err == hipSuccess; o++)
        err = hipMemcpy(S->d_C + (size_t)o * N * N, plan->outs[o].h_C.data(), (size_t)N * N * 8, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemset(S->d_flag, 0, sizeof(int32_t));
    if (err != hipSuccess) { mf_release(plan); HIP_TRY(err); }
    S->lds_phi = ((size_t)N * N + (size_t)S->nw * S->nsym + (size_t)S->nw * N) * 8;
    S->lds_grad = (size_t)N * N * 8;
    // every block safely positive definite?
    MfArgs A;
    A.N = N; A.nsym = S->nsym; A.n_out = n_out; A.wgs = S->wgs; A.tiles = S->d_tiles; A.tile_begin = S->d_tile_begin; A.C = S->d_C;
    A.groups = S->d_groups; A.map = S->d_map; A.goff = plan->d_goff;
    const int cgrid = (int)std::max<int64_t>(1, std::min<int64_t>(256, (most + 3) / 4));
    hipLaunchKernelGGL(k_mf_check, dim3(cgrid, n_out), dim3(256), (size_t)N * N * 8, 0, A, S->d_flag);
    int32_t flag = 1;
    err = hipMemcpy(&flag, S->d_flag, sizeof(int32_t), hipMemcpyDeviceToHost);
    if (err != hipSuccess) { mf_release(plan); HIP_TRY(err); }
    if (flag != 0) { mf_release(plan); return BLUEST_OK; }      // some block is (nearly) singular: the stored pseudo-inverses are the path
    // dynamic LDS beyond the default needs the attribute (per device: the attribute belongs to the device's copy of the kernel)
    if (S->lds_phi > (48u << 10)) {
        if (S->nw == 8) (void)hipFuncSetAttribute((const void *)k_phi_matfree<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S->lds_phi);
        else (void)hipFuncSetAttribute((const void *)k_phi_matfree<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S->lds_phi);
    }
    plan->mf_gradient = true;
    plan->matfree = full;
    return BLUEST_OK;
}

static MfArgs mf_args(bluest_plan_t plan)
{
    MfState *S = reinterpret_cast<MfState *>(plan->mf);
    MfArgs A;
    A.N = plan->N; A.nsym = S->nsym; A.n_out = (int)plan->outs.size(); A.wgs = S->wgs; A.tiles = S->d_tiles; A.tile_begin = S->d_tile_begin;
    A.C = S->d_C; A.groups = S->d_groups; A.map = S->d_map; A.goff = plan->d_goff;
    return A;
}

// Phi record of ONE allocation vector into rec_dev (NULL: the plan's own record buffer, returned through *rec_used)
int mf_phi_record(bluest_plan_t plan, const double *m_dev, double *rec_dev, const double **rec_used, hipStream_t st)
{
    MfState *S = reinterpret_cast<MfState *>(plan->mf);
    if (!S) return fail(BLUEST_ERR_STATE, "matrix-free state missing");
    const MfArgs A = mf_args(plan);
    double *rec = rec_dev ? rec_dev : S->d_rec;
    const dim3 grid((unsigned)S->wgs, (unsigned)A.n_out);
    if (S->nw == 8) hipLaunchKernelGGL(k_phi_matfree<8>, grid, dim3(512), S->lds_phi, st, A, m_dev, S->d_partial, S->d_amax);
    else hipLaunchKernelGGL(k_phi_matfree<4>, grid, dim3(256), S->lds_phi, st, A, m_dev, S->d_partial, S->d_amax);
    hipLaunchKernelGGL(k_mf_reduce, dim3((unsigned)((S->nsym + A.N + 1 + 3) / 4), (unsigned)A.n_out), dim3(256), 0, st, A.N, S->nsym, S->wgs,
                       S->d_partial, S->d_amax, rec);
    if (rec_used) *rec_used = rec;
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

// solve + gradient of this plan's groups from a record (rec_dev == NULL: from the chunk partials the stored Phi pass just left):
// (var, status, v workspace, grad)
int mf_solve_grad(bluest_plan_t plan, const double *rec_dev, double delta, double *var_dev, int32_t *status_dev, double *grad_dev, hipStream_t st, MaTail ma)
{
    MfState *S = reinterpret_cast<MfState *>(plan->mf);
    if (!S) return fail(BLUEST_ERR_STATE, "matrix-free state missing");
    const MfArgs A = mf_args(plan);
    // workgroups per output: MF_TILE_WAVES tiles each; the table of first workgroups rides in the (host-built) tile_begin pattern
    if (!plan->mf_wg_begin_dev) {
        std::vector<int32_t> wb((size_t)A.n_out + 1, 0);
        for (int o = 0; o < A.n_out; o++) {
            const int nt = S->tile_begin[o + 1] - S->tile_begin[o];
            wb[(size_t)o + 1] = wb[(size_t)o] + std::max(1, (nt + MF_TILE_WAVES - 1) / MF_TILE_WAVES);
        }
        DeviceScopeN scope(plan->device);
        HIP_TRY(pool_alloc((void **)&plan->mf_wg_begin_dev, al((A.n_out + 1) * sizeof(int32_t))));
        HIP_TRY(hipMemcpy(plan->mf_wg_begin_dev, wb.data(), (A.n_out + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
        plan->mf_wgs_grad = wb[(size_t)A.n_out];
        plan->mf_bpo = wb[1];
        for (int o = 0; o < A.n_out; o++) if (wb[(size_t)o + 1] - wb[(size_t)o] != wb[1]) plan->mf_bpo = 0;
    }
    const dim3 grid((unsigned)plan->mf_wgs_grad);
    int kmax = 0;
    for (const auto &od : plan->outs) kmax = std::max(kmax, od.K);
#define LMF2(NT, KU) hipLaunchKernelGGL((k_solve_grad_mf<NT, KU>), grid, dim3(64 * (MF_TILE_WAVES + 1)), S->lds_grad, st, A, rec_dev, delta, \
                                        plan->d_rows, plan->fold_reg, plan->d_partial, (const int32_t *)plan->mf_wg_begin_dev, plan->mf_bpo, var_dev, plan->d_v, status_dev, grad_dev, ma)
#define LMF(NT) do { if (kmax <= 5) LMF2(NT, 5); else if (kmax <= 6) LMF2(NT, 6); else LMF2(NT, 8); } while (0)
    NT_DISPATCH(plan->N, LMF);
#undef LMF
#undef LMF2
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_plan_matfree(bluest_plan_t plan, int *matfree, int64_t *mf_bytes)
{
    if (!plan || !matfree) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    *matfree = plan->matfree ? 1 : (plan->mf_gradient ? 2 : 0);
    if (mf_bytes) {
        // per evaluation: Phi pass and gradient pass each read the groups' model indices (bytes) and m / write the gradient, plus the
        // workgroups' partials (written, then read once) and the covariances
        int64_t b = 0;
        const MfState *S = reinterpret_cast<const MfState *>(plan->mf);
        for (const auto &od : plan->outs) {
            b += 2 * od.L_o * 8 + 2 * od.L_o * 8;      // packed indices (8 bytes per group) in both passes, m in, gradient out
        }
        if (S) b += 2 * (int64_t)plan->outs.size() * S->wgs * (S->nsym * 8 + (plan->N + 1) * 8) + 2 * (int64_t)plan->outs.size() * plan->N * plan->N * 8;
        *mf_bytes = b;
    }
    return BLUEST_OK;
}

// gradient of this plan's groups from given v / status (one candidate)
int mf_grad(bluest_plan_t plan, const double *v_dev, const int32_t *status_dev, double *grad_dev, hipStream_t st)
{
    MfState *S = reinterpret_cast<MfState *>(plan->mf);
    if (!S) return fail(BLUEST_ERR_STATE, "matrix-free state missing");
    const MfArgs A = mf_args(plan);
    int64_t most = 0;
    for (int o = 0; o < A.n_out; o++) most = std::max<int64_t>(most, S->tile_begin[o + 1] - S->tile_begin[o]);
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((most + 3) / 4, 1024 / std::max(1, A.n_out)));
    hipLaunchKernelGGL(k_grad_mf, dim3(gx, (unsigned)A.n_out), dim3(256), (size_t)(A.N * A.N + A.N) * 8, st, A, v_dev, status_dev, grad_dev);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}
