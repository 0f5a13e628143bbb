// plan.hip -- Part 2 of include/bluest_hip.h: the evaluation plan of one (multi-output) sample-allocation problem on MI355X
// (gfx950 / CDNA4).
//
// What is computed (reference files bluest/*.py, bluest/cmisc.cpp; cited per function in include/bluest_hip.h):
//   Phi_o(m) = sum_i m_i R_i^T C_i^-1 R_i      (misc.py:459-461, cmisc.cpp:25-40)
//   V_o      = (Phi_o[idx,idx]^-1)_00          (misc.py:463-477, :490)
//   grad_o,i = -v[g_i]^T C_i^-1 v[g_i]         (misc.py:493, cmisc.cpp:58-72), v = row 0 of pinv(Phi_o)
// for every output o of a multi-output problem and a batch of candidate allocations m, all float64.
//
// Design: HBM/L2-streaming integer + f64 work with ~0.2 flop/byte, so there is no MFMA; the levers are
// coalescing, bytes per entry, dependent round trips and launch count.
//   * Phi pass : destination-major symmetric CSR of psi, cut into wave-sized chunks; one wavefront streams one chunk with
//                16-byte loads and gathers m from L2; fixed-order butterfly sum => bit-reproducible (no float atomics).
//   * solve    : per (candidate, output): quads of lanes fold the chunk partials per row -> Phi in LDS; one wavefront holds
//                the restricted, permuted matrix in registers and eliminates by Gauss-Jordan (solve.hpp).
//   * grad pass: group-major tiles of 64 groups, lane = group; per group the model indices and the packed-symmetric inverse as
//                8-byte slots stored in pairs, so every wave-instruction reads 16 bytes per lane, 1 KiB contiguous (TileDesc,
//                plan.hpp); fused with the solve for the single-candidate evaluation (k_solve_grad: one solver wavefront, the
//                tile wavefronts streaming -- non-temporal loads -- while it factorises; tiles per workgroup chosen per plan
//                so that the workgroups cover every compute unit).
//   * What bounds a step after round 3: the two copies of the inverses (Phi layout + tiles) evict each other from the L2s; see
//                DESIGN.md section 4 and profiles/r03_phi_tiles_negative_result.txt (k_phi_tiles below is the single-copy pass).
// No CPU fallback exists in this library.

#include "common.hpp"

#ifndef BLUEST_ABLATE      // experiment builds only (-DBLUEST_ABLATE=n, tools/ablate.sh): parts of the evaluation kernels switched off
#define BLUEST_ABLATE 0    // for timing: 1 no fold, 2 no elimination, 3 no tile stream, 5 empty k_solve_grad, 6 empty Phi pass, 7 no gradient
                           // store, 8 Phi pass stores its partials elsewhere, 9 Phi pass only stores, 10 / 11 k_phi_tiles: products only / no tile loads, 12 Phi pass without the gather of m
#endif
#ifdef BLUEST_PHASE_TIMING   // experiment builds only (tools/phase_timing.py): 100 MHz timestamps of one workgroup's phases
__device__ long long g_phase[3][12];
#define PHASE(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1)) \
        g_phase[blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : 1)][i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
extern "C" int bluest_debug_phase_times(long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(long long) * 36) == hipSuccess ? 0 : 1; }
// the same for the first TILE wavefront (thread 64) of the three sampled workgroups
__device__ long long g_phase_tile[3][8];
#define PHASE_TILE(i) do { if (threadIdx.x == 64 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1)) \
        g_phase_tile[blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : 1)][i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
extern "C" int bluest_debug_phase_times_tile(long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_tile), sizeof(long long) * 24) == hipSuccess ? 0 : 1; }
// kernel spans in a chain of evaluations: [step % 16][kernel][begin, end] (min / max over a sample of workgroups)
__device__ unsigned long long g_span[16][2][2];
__device__ int g_step;
#define SPAN_SAMPLED() (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x < 8 || blockIdx.x + 8 >= gridDim.x || (blockIdx.x & 31) == 0))
#define SPAN_BEGIN(kid) const int span_step_ = g_step & 15; \
    do { if (SPAN_SAMPLED()) atomicMin(&g_span[span_step_][kid][0], (unsigned long long)wall_clock64()); } while (0)
#define SPAN_END(kid, bump) do { if (SPAN_SAMPLED()) atomicMax(&g_span[span_step_][kid][1], (unsigned long long)wall_clock64()); \
        if (bump && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_step = g_step + 1; } while (0)
extern "C" int bluest_debug_span_reset(void)
{
    unsigned long long h[16][2][2];
    for (int i = 0; i < 16; i++) for (int k = 0; k < 2; k++) { h[i][k][0] = ~0ull; h[i][k][1] = 0ull; }
    int z = 0;
    return (hipMemcpyToSymbol(HIP_SYMBOL(g_span), h, sizeof(h)) == hipSuccess && hipMemcpyToSymbol(HIP_SYMBOL(g_step), &z, sizeof(z)) == hipSuccess) ? 0 : 1;
}
extern "C" int bluest_debug_span_read(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1; }
#else
#define PHASE(i)
#define PHASE_TILE(i)
#define SPAN_BEGIN(kid)
#define SPAN_END(kid, bump)
#endif
static int g_debug_timing = getenv("BLUEST_DEBUG_TIMING") ? 1 : 0;   // stderr phase times of the set-up entry points
struct PhaseTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    const char *what;
    explicit PhaseTimer(const char *w) : what(w) {}
    void lap(const char *phase)
    {
        if (!g_debug_timing) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[bluest timing] %s: %s %.3f ms\n", what, phase, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
static int g_debug_solve = getenv("BLUEST_DEBUG_SOLVE") ? atoi(getenv("BLUEST_DEBUG_SOLVE")) : 0;  // timing experiments only


#include "plan.hpp"
#include "spg_state.hpp"

// ------------------------------------------------------------------------------------------------------
// Part 2 -- plan kernels
// ------------------------------------------------------------------------------------------------------
// columns (global group indices) are stored as uint16 when the allocation vector has at most 65 536 entries (half the bytes of the
// column stream, which the shared kernel re-reads per pair of outputs), else as int32
template <typename COLT> __device__ __forceinline__ int4 load_cols4(const COLT *p);
template <> __device__ __forceinline__ int4 load_cols4<int32_t>(const int32_t *p) { return *reinterpret_cast<const int4 *>(p); }
template <> __device__ __forceinline__ int4 load_cols4<uint16_t>(const uint16_t *p)
{
    const uint2 w = *reinterpret_cast<const uint2 *>(p);
    return make_int4((int)(w.x & 0xffffu), (int)(w.x >> 16), (int)(w.y & 0xffffu), (int)(w.y >> 16));
}
// Phi pass: one wavefront per chunk of CH = 256*iters entries; lane l owns entries [4l, 4l+4) of each 256-block.
template <int WPB, typename COLT>
__global__ __launch_bounds__(64 * WPB) void k_phi_chunks(const double *__restrict__ vals, const COLT *__restrict__ cols,
                                                    int iters, int64_t n_chunks, const double *__restrict__ m,
                                                    int64_t m_stride, int n_cand, double2 *__restrict__ partial,
                                                    const int32_t *__restrict__ pslot, int64_t pstride,
                                                    const int32_t *__restrict__ gate)
{
    if (gate && *gate == 0) return;   // device-side predication (SPG line-search slots)
    const int64_t chunk = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (chunk >= n_chunks) return;
    const int64_t slot = pslot ? (int64_t)pslot[chunk] : chunk;      // where the fold expects this chunk's partial
    const int64_t base = chunk * (int64_t)iters * 256 + lane * 4;
    for (int c = 0; c < n_cand; c++) {
        const double *mc = m + (int64_t)c * m_stride;
        double s = 0.0, amax = 0.0;
        for (int it = 0; it < iters; it++) {
            const double2 v01 = *reinterpret_cast<const double2 *>(vals + base + it * 256);
            const double2 v23 = *reinterpret_cast<const double2 *>(vals + base + it * 256 + 2);
            const int4 cc = load_cols4<COLT>(cols + base + it * 256);
            const double m0 = mc[cc.x], m1 = mc[cc.y], m2 = mc[cc.z], m3 = mc[cc.w];
            s = fma(v01.x, m0, s);
            s = fma(v01.y, m1, s);
            s = fma(v23.x, m2, s);
            s = fma(v23.y, m3, s);
            amax = fmax(fmax(amax, fmax(fabs(m0), fabs(m1))), fmax(fabs(m2), fabs(m3)));
        }
        s = wave_sum(s);
        amax = wave_max(amax);
        if (lane == 0) partial[(int64_t)c * pstride + slot] = make_double2(s, amax);
    }
}

// Phi pass, shared structure: when every output has the same groups and mapping (the usual multi-output case) the
// column indices and the gathered m are common; one wavefront streams the chunk of OB outputs and reads them once.
// vals / partial keep the output-major chunk numbering of the general layout (chunk id = o*ncpo + c).
template <int OB, int WPB, typename COLT>
__global__ __launch_bounds__(64 * WPB) void k_phi_chunks_shared(const double *__restrict__ vals, const COLT *__restrict__ cols,
                                                           int iters, int64_t ncpo, int n_out, const double *__restrict__ m,
                                                           int64_t m_stride, int n_cand, int64_t pstride, const int32_t *__restrict__ pslot,
                                                           int slots_per_output, double2 *__restrict__ partial, const int32_t *__restrict__ gate)
{
    if (gate && *gate == 0) return;   // device-side predication (SPG line-search slots)
    const int64_t chunk = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int o0 = blockIdx.y * OB;
    if (chunk >= ncpo) return;
    if (BLUEST_ABLATE == 6) return;
    // where the fold expects this chunk's partials: output-major chunk numbering, or the regular rows' slots (one structure for all outputs)
    const int64_t slot = pslot ? (int64_t)pslot[chunk] : chunk, ostride = pslot ? (int64_t)slots_per_output : ncpo;
    SPAN_BEGIN(0);
    const int64_t CH = (int64_t)iters * 256;
    const int64_t base = chunk * CH + lane * 4;
    for (int c = 0; c < n_cand; c++) {
        const double *mc = m + (int64_t)c * m_stride;
        double s[OB];
#pragma unroll
        for (int oo = 0; oo < OB; oo++) s[oo] = 0.0;
        double amax = 0.0;
        for (int it = 0; it < (BLUEST_ABLATE == 9 ? 0 : iters); it++) {
            const int4 cc = load_cols4<COLT>(cols + base + it * 256);
            double2 v01[OB], v23[OB];
#pragma unroll
            for (int oo = 0; oo < OB; oo++) {
                const int o = (o0 + oo < n_out) ? o0 + oo : n_out - 1;
                const double *vp = vals + (int64_t)o * ncpo * CH + base + it * 256;
                v01[oo] = *reinterpret_cast<const double2 *>(vp);
                v23[oo] = *reinterpret_cast<const double2 *>(vp + 2);
            }
#if BLUEST_ABLATE == 12      // (experiment: no gather of m)
            const double m0 = 1.0 + cc.x, m1 = 1.0 + cc.y, m2 = 1.0 + cc.z, m3 = 1.0 + cc.w;
#else
            const double m0 = mc[cc.x], m1 = mc[cc.y], m2 = mc[cc.z], m3 = mc[cc.w];
#endif
            amax = fmax(fmax(amax, fmax(fabs(m0), fabs(m1))), fmax(fabs(m2), fabs(m3)));
#pragma unroll
            for (int oo = 0; oo < OB; oo++) {
                s[oo] = fma(v01[oo].x, m0, s[oo]);
                s[oo] = fma(v01[oo].y, m1, s[oo]);
                s[oo] = fma(v23[oo].x, m2, s[oo]);
                s[oo] = fma(v23[oo].y, m3, s[oo]);
            }
        }
        amax = wave_max(amax);
#pragma unroll
        for (int oo = 0; oo < OB; oo++) {
            const double t = wave_sum(s[oo]);
            // (experiment builds: ABLATE 8 stores into the slots of candidate 1 -- needs max_candidates >= 2 --, so the fused kernel
            //  reads partials that were not just rewritten; ABLATE 9 skips the streams above and only stores)
            if (lane == 0 && o0 + oo < n_out) partial[(int64_t)(c + (BLUEST_ABLATE == 8 ? 1 : 0)) * pstride + (int64_t)(o0 + oo) * ostride + slot] = make_double2(t, amax);
        }
    }
    SPAN_END(0, false);
}

// Phi pass FROM THE TILES (plans whose outputs share one group list, at most 32 workgroups per output): the workgroup that owns
// tiles [b*tpb, (b+1)*tpb) of output o in the fused solve + gradient kernel also forms their contribution to Phi_o, so the plan
// holds ONE copy of the inverses (the destination-major copy of k_phi_chunks is not built) and both passes of a step read the
// same bytes through the same compute unit's L2: 21 MB instead of 42 at the headline size, and that does fit the 32 MB of L2.
// No float atomics: lane = group multiplies its packed entries by m_g and leaves the products in LDS ([tile slot][entry][lane]).
// After a barrier the products are summed per symmetric destination in an order fixed with the plan: the destination's
// contributions (positions into the staging area) are cut into SEGMENTS of 16, one thread sums one segment (its 16 positions
// sit in registers since the start of the kernel: 32 bytes per thread, padded with the position of a zero), a second barrier, then
// one thread per destination adds the destination's segment sums in order and writes ONE partial per (destination, workgroup)
// where fold_rows expects the "chunks" of the destination's row.  max |m| over the groups containing a model rides along on
// the diagonal destinations.
struct PhiTilesArgs {
    const TileDesc *tiles; const RowDesc *rows; const double *tvals; const int64_t *goff; const int32_t *gmap;
    const uint32_t *wg_seg_base;   // [bpo + 1] first segment of every workgroup
    const uint16_t *seg_list;      // [segments][16] staging positions
    const uint16_t *seg_dest;      // [segments] destination (bit 15: diagonal)
    const uint16_t *wg_dseg;       // [bpo][nsym + 1] first segment (relative to the workgroup) of every destination
    int bpo, tpb, nsym, stage_stride, seg_cap;   // seg_cap: LDS slots for segment sums (>= segments of any workgroup)
    uint32_t stride_inv;           // ceil(2^32 / stage_stride): tile slot of a position = mulhi(pos, stride_inv)
};
template <int KU, int NW>
__global__ __launch_bounds__(64 * NW) void k_phi_tiles(const PhiTilesArgs A, const double *__restrict__ m, int64_t m_stride, int n_cand,
                                                       int64_t n_chunks, double2 *__restrict__ partial, const int32_t *__restrict__ gate)
{
    extern __shared__ double phi_stage[];      // [tpb * stride products][64 zeros][(tpb + 1) * 64 |m_g|, last row zero][seg_cap sums][seg_cap maxima]
    if (gate && *gate == 0) return;   // device-side predication (SPG line-search slots)
    if (BLUEST_ABLATE == 6) return;
    constexpr int PU = tile_pairs(KU);
    constexpr int NTHREADS = 64 * NW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tpb = A.tpb, nsym = A.nsym, S = A.stage_stride;
    const int o = blockIdx.x / A.bpo, b = blockIdx.x % A.bpo;
    double *stage_m = phi_stage + (int64_t)tpb * S + 64;
    double *seg_sum = stage_m + (tpb + 1) * 64, *seg_max = seg_sum + A.seg_cap;
    const bool has = wave < tpb;
    TileDesc td;
    td.k = 0; td.n_valid = 0; td.val_off = td.grad_off = 0; td.out = (int16_t)o;
    if (has) td = A.tiles[(int64_t)blockIdx.x * tpb + wave];
    // this thread's segment of the first pass: destination and 16 positions (two 16-byte loads)
    const uint32_t seg0 = A.wg_seg_base[b], nseg = A.wg_seg_base[b + 1] - seg0;
    uint4 L0 = make_uint4(0, 0, 0, 0), L1 = L0;
    int sd0 = 0;
    if ((uint32_t)tid < nseg) {
        const uint4 *lp = reinterpret_cast<const uint4 *>(A.seg_list + (int64_t)(seg0 + tid) * 16);
        L0 = lp[0]; L1 = lp[1];
        sd0 = A.seg_dest[seg0 + tid];
    }
    const int k = td.k;
    double2 pr[PU];
    if (BLUEST_ABLATE == 11) { for (int i = 0; i < PU; i++) pr[i] = make_double2(1.0, 1.0); }      // (experiment: no tile loads)
    else if (has && k == KU) {      // the usual tile: straight-line loads
        const double *tl = A.tvals + td.val_off + 2 * lane;
#pragma unroll
        for (int i = 0; i < PU; i++) pr[i] = *reinterpret_cast<const double2 *>(tl + i * 128);
    } else if (has && k < KU) tile_load(pr, A.tvals + td.val_off + 2 * lane, tile_pairs(k));
    const bool valid = has && lane < (td.n_valid & 0xffff);
    int64_t gidx = 0;
    if (valid) {
        const int64_t li = td.grad_off - A.goff[o] + lane;
        gidx = A.gmap ? (int64_t)A.gmap[li] : li;
    }
    if (tid < 64) { phi_stage[(int64_t)tpb * S + tid] = 0.0; stage_m[tpb * 64 + tid] = 0.0; }
    double *my_stage = phi_stage + (int64_t)wave * S + lane;
    // sum of one segment: positions p[0..16) in order
    auto seg_reduce = [&](const uint4 &l0, const uint4 &l1, int sd, int slot) {
        const unsigned w[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        double s = 0.0;
#pragma unroll
        for (int h = 0; h < 2; h++) {      // eight reads in flight at a time (registers)
            double v[8];
#pragma unroll
            for (int i = 0; i < 4; i++) { v[2 * i] = phi_stage[w[4 * h + i] & 0xffffu]; v[2 * i + 1] = phi_stage[w[4 * h + i] >> 16]; }
#pragma unroll
            for (int i = 0; i < 8; i++) s += v[i];
        }
        seg_sum[slot] = s;
        if (sd & 0x8000) {     // diagonal destination: max |m| of the contributing groups
            double am = 0.0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned p0 = w[i] & 0xffffu, p1 = w[i] >> 16;
                am = fmax(am, stage_m[__umulhi(p0, A.stride_inv) * 64 + (p0 & 63)]);
                am = fmax(am, stage_m[__umulhi(p1, A.stride_inv) * 64 + (p1 & 63)]);
            }
            seg_max[slot] = am;
        }
    };
    for (int c = 0; c < n_cand; c++) {
        const double mg = valid ? m[(int64_t)c * m_stride + gidx] : 0.0;
        if (has) {
            stage_m[wave * 64 + lane] = fabs(mg);
#define PT(KK) case KK: if (KK <= KU) {                                                                              \
                _Pragma("unroll") for (int e = 0; e < tile_ne(KK); e++)                                              \
                    my_stage[e * 64] = mg * tile_slot(pr, tile_ni(KK <= KU ? KK : 1) + e);                           \
                break; }
            switch (k) {
                PT(1) PT(2) PT(3) PT(4) PT(5) PT(6) PT(7) PT(8) PT(9) PT(10) PT(11) PT(12)
                default: {      // larger groups: slots straight from global memory
                    const double *tile = A.tvals + td.val_off;
                    const int ni = tile_ni(k), ne = tile_ne(k);
                    for (int e = 0; e < ne; e++) my_stage[e * 64] = mg * tile[tile_slot_off(ni + e, lane)];
                }
            }
#undef PT
        }
        __syncthreads();
        if (BLUEST_ABLATE == 10) continue;      // (experiment: products only)
        if ((uint32_t)tid < nseg) seg_reduce(L0, L1, sd0, tid);
        for (uint32_t t = tid + NTHREADS; t < nseg; t += NTHREADS) {      // (more segments than threads: rare)
            const uint4 *lp = reinterpret_cast<const uint4 *>(A.seg_list + (int64_t)(seg0 + t) * 16);
            seg_reduce(lp[0], lp[1], A.seg_dest[seg0 + t], (int)t);
        }
        __syncthreads();
        for (int d = tid; d < nsym; d += NTHREADS) {
            const int sb = A.wg_dseg[(int64_t)b * (nsym + 1) + d], se = A.wg_dseg[(int64_t)b * (nsym + 1) + d + 1];
            const RowDesc rd = A.rows[(int64_t)o * nsym + d];
            double s = 0.0, am = 0.0;
            for (int t = sb; t < se; t++) s += seg_sum[t];
            if (rd.a == rd.b) for (int t = sb; t < se; t++) am = fmax(am, seg_max[t]);
            partial[(int64_t)c * n_chunks + rd.first_chunk + b] = make_double2(s, am);
        }
        if (c + 1 < n_cand) __syncthreads();
    }
}

// fused: fold chunk partials + solve.  grid = (n_out, n_cand), block = fold_threads (fold) -> wavefront 0 (solve).
// want_v: bit0 = also produce v (gradient wanted); bit1 / bit2 = timing diagnostics (fold only / solve twice).
template <int NT>
__global__ __launch_bounds__(fold_threads(NT)) void k_solve_from_chunks(int N, int n_out, const RowDesc *__restrict__ rows, int nsym, FoldReg reg,
                                                           const double2 *__restrict__ partial, int64_t n_chunks,
                                                           double delta, int want_v, double *__restrict__ var,
                                                           double *__restrict__ v, int32_t *__restrict__ status,
                                                           const int32_t *__restrict__ gate, double *__restrict__ spg_state, int last_slot, int32_t *__restrict__ spg_enable,
                                                           unsigned int *__restrict__ ticket)
{
    __shared__ SolveLds<NT> lds;
    __shared__ double spg_ls[SPG_STATE_DOUBLES];
    const int o = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    if (gate && *gate == 0) {   // device-side predication (SPG line-search slots)
        // the line-search decision still has to close the slot (it sets the gate of the finishing launches on the last one)
        if (spg_state && o == 0 && c == 0 && tid < WAVE) spg_decide_wave(spg_state, var, status, n_out, last_slot, spg_enable, spg_ls, tid);
        return;
    }
    SpgPrefetch pf;
    if (spg_state && tid < WAVE) spg_prefetch_state(spg_state, tid, pf);   // in flight during the fold (see spg_state.hpp)
    if (N < NT) { clear_pads(lds, N, tid, fold_threads(NT)); __syncthreads(); }   // uniform; every real entry is written by the fold
    fold_rows(lds, N, rows, o * nsym, nsym, partial + (int64_t)c * n_chunks, tid, fold_threads(NT), reg);
    __syncthreads();
    if (tid >= WAVE) return;   // single wavefront from here on
    const int lane = tid;
    if (spg_state) spg_prefetch_parts(pf, lane);                           // in flight during the elimination
    const int64_t e = (int64_t)c * n_out + o;
    if (want_v & 2) {   // diagnostics: fold only (timing experiments)
        if (lane == 0) { var[e] = lds.at(0, 0); status[e] = 0; }
        return;
    }
    const double am = (lane < N) ? lds.amax[lane] : 0.0;
    const bool big = __ballot(am >= 0.05) != 0ull;   // max |m| >= 0.05 (misc.py:464)
    const int reps = (want_v & 4) ? 2 : 1;   // diagnostics: run the solve twice
    double V_mine = 0.0;
    int32_t st_mine = 0;
    for (int rep = 0; rep < reps; rep++)
        solve_wave<NT>(lds, N, delta, am > 1.0e-6, am > 0.0, big, (want_v & 1) != 0, &V_mine, v + e * N, &st_mine, lane);
    if (lane == 0) { var[e] = V_mine; status[e] = st_mine; }
    if (spg_state) {
        // line-search decision fused into the tail (single candidate): every output's workgroup publishes V and status, takes a
        // ticket, and the LAST one to arrive evaluates the objective of the trial point and decides -- one launch and one kernel
        // boundary less per slot.  Publication: stores drained, agent-scope release, then the ticket (relaxed agent atomic);
        // the last arriver acquires before it reads the other outputs' values (MI355X_MICROARCH.md, inter-workgroup visibility).
        int last = 0;
        if (n_out == 1) {   // single output: nobody to wait for, V and status stay in lane 0's registers
            spg_decide_wave(spg_state, var, status, n_out, last_slot, spg_enable, spg_ls, lane, pf, true, V_mine, st_mine);
            return;
        }
        if (lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (t == (unsigned int)(n_out - 1)) ? 1 : 0;
            if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        }
        last = __builtin_amdgcn_readfirstlane(last);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            spg_decide_wave(spg_state, var, status, n_out, last_slot, spg_enable, spg_ls, lane, pf);
        }
    }
}

// multi-GPU path, phase A tail: fold chunk partials into an all-reduce-able record.
template <int NT>
__global__ __launch_bounds__(fold_threads(NT)) void k_fold_to_record(int N, int n_out, const RowDesc *__restrict__ rows, int nsym, FoldReg reg,
                                                        const double2 *__restrict__ partial, int64_t n_chunks,
                                                        double *__restrict__ rec)
{
    __shared__ SolveLds<NT> lds;
    const int o = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    fold_rows(lds, N, rows, o * nsym, nsym, partial + (int64_t)c * n_chunks, tid, fold_threads(NT), reg);
    __syncthreads();
    const int reclen = N * N + 2 * N + 1;
    double *r = rec + ((int64_t)c * n_out + o) * reclen;
    for (int t = tid; t < N * N; t += fold_threads(NT)) r[t] = lds.at(t / N, t % N);
    if (tid >= WAVE) return;
    const double am = (tid < N) ? lds.amax[tid] : 0.0;
    if (tid < N) { r[N * N + tid] = (am > 1.0e-6) ? 1.0 : 0.0; r[N * N + N + tid] = (am > 0.0) ? 1.0 : 0.0; }
    const double big = wave_max(am);
    if (tid == 0) r[N * N + 2 * N] = (big >= 0.05) ? 1.0 : 0.0;
}

// multi-GPU path, phase B: solve from an (all-reduced) record.  block = 64.
template <int NT>
__global__ __launch_bounds__(64) void k_solve_from_record(int N, int n_out, const double *__restrict__ rec,
                                                          double delta, int want_v, double *__restrict__ var,
                                                          double *__restrict__ v, int32_t *__restrict__ status)
{
    __shared__ SolveLds<NT> lds;
    const int o = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    const int reclen = N * N + 2 * N + 1;
    const double *r = rec + ((int64_t)c * n_out + o) * reclen;
    clear_pads(lds, N, lane, WAVE);
    __syncthreads();
    for (int t = lane; t < N * N; t += WAVE) lds.at(t / N, t % N) = r[t];
    __syncthreads();
    const bool s1 = lane < N && r[N * N + lane] > 0.0;
    const bool s2 = lane < N && r[N * N + N + lane] > 0.0;
    const bool big = r[N * N + 2 * N] > 0.0;
    const int64_t e = (int64_t)c * n_out + o;
    solve_wave<NT>(lds, N, delta, s1, s2, big, (want_v & 1) != 0, var + e, v + e * N, status + e, lane);
}

// Rare path (bluest_plan_solve_pinv): what the reference's variance_GH returns when the restricted Phi is rank-deficient --
// numpy pinv, i.e. eigenvalues not larger than 1e-15 * max|lambda| dropped (misc.py:487,490).  One wavefront per (candidate,
// output), lane r = row r; cyclic Jacobi exactly as sym_pinv_jacobi (mirrors.hip) does it per group, with the matrix and the
// eigenvectors in LDS because the size is a run-time value here.  Models outside the index set keep zero rows / columns: their
// eigenvalues are zero and drop out, so the pseudo-inverse of the padded matrix is the padded pseudo-inverse.
__device__ __forceinline__ void jacobi_pinv_lds(double *A, double *Q, int N, int LD, int lane)
{
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
        for (int p = 0; p < N - 1; p++)
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p * LD + q], app = A[p * LD + p], aqq = A[q * LD + q];      // same address in every lane
                if (!(fabs(apq) > 1.0e-18 * sqrt(fabs(app * aqq)))) continue;                    // uniform (also skips zeros / NaN)
                rotated = 1;
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                if (lane < N) {
                    const double arp = A[lane * LD + p], arq = A[lane * LD + q];
                    A[lane * LD + p] = c * arp - sn * arq;
                    A[lane * LD + q] = sn * arp + c * arq;
                    const double vrp = Q[lane * LD + p], vrq = Q[lane * LD + q];
                    Q[lane * LD + p] = c * vrp - sn * vrq;
                    Q[lane * LD + q] = sn * vrp + c * vrq;
                }
                wave_lds_sync();
                if (lane < N) {
                    const double apr = A[p * LD + lane], aqr = A[q * LD + lane];
                    A[p * LD + lane] = c * apr - sn * aqr;
                    A[q * LD + lane] = sn * apr + c * aqr;
                }
                wave_lds_sync();
            }
        if (!rotated) break;
    }
}

__global__ __launch_bounds__(64) void k_pinv_from_record(int N, int n_out, const double *__restrict__ rec, double delta,
                                                         double *__restrict__ var, double *__restrict__ v,
                                                         int32_t *__restrict__ status)
{
    extern __shared__ double pinv_sm[];
    const int LD = N + 1;
    double *A = pinv_sm, *Q = pinv_sm + N * LD;
    const int o = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    const int reclen = N * N + 2 * N + 1;
    const double *r = rec + ((int64_t)c * n_out + o) * reclen;
    const int64_t e = (int64_t)c * n_out + o;
    const bool s1 = lane < N && r[N * N + lane] > 0.0;
    const bool s2 = lane < N && r[N * N + N + lane] > 0.0;
    const bool big = r[N * N + 2 * N] > 0.0;
    const unsigned long long mask1 = __ballot(s1);
    const unsigned long long mask2 = (delta != 0.0) ? __ballot(lane < N) : __ballot(s2);
    int st = BLUEST_EVAL_OK;
    double V = 0.0, vmine = 0.0;
    if (!big) { st = BLUEST_EVAL_INF; V = INFINITY; }
    else if (mask1 == 0ull) { st = BLUEST_EVAL_NO_MODEL0; V = NAN; }
    else {
        if (!(mask1 & 1ull)) st = BLUEST_EVAL_NO_MODEL0;
        const int npass = (mask1 == mask2) ? 1 : 2;
        for (int pass = 0; pass < npass; pass++) {
            const unsigned long long mask = pass == 0 ? mask1 : mask2;
            if (lane < N)
                for (int j = 0; j < N; j++) {
                    const bool in = ((mask >> lane) & 1ull) && ((mask >> j) & 1ull);
                    A[lane * LD + j] = in ? 0.5 * (r[lane * N + j] + r[j * N + lane]) + (lane == j ? delta : 0.0) : 0.0;
                    Q[lane * LD + j] = lane == j ? 1.0 : 0.0;
                }
            wave_lds_sync();
            jacobi_pinv_lds(A, Q, N, LD, lane);
            const double w = lane < N ? A[lane * LD + lane] : 0.0;
            const double cut = 1.0e-15 * wave_max(fabs(w));
            if (lane < N) A[lane * LD + lane] = fabs(w) > cut ? 1.0 / w : 0.0;        // the diagonal now holds 1/lambda or 0
            wave_lds_sync();
            const int t = __ffsll((long long)mask1) - 1;       // first row of the restricted matrix (model 0 unless unsampled)
            if (pass == 0) {
                double acc = 0.0;
                for (int ee = 0; ee < N; ee++) acc += Q[t * LD + ee] * A[ee * LD + ee] * Q[t * LD + ee];
                V = acc;
            }
            if (pass == npass - 1) {                           // row 0 of pinv(Phi): zero when model 0 is outside the support
                double acc = 0.0;
                if (lane < N && (mask2 & 1ull))
                    for (int ee = 0; ee < N; ee++) acc += Q[0 * LD + ee] * A[ee * LD + ee] * Q[lane * LD + ee];
                vmine = acc;
            }
            wave_lds_sync();
        }
    }
    if (lane < N) v[e * N + lane] = vmine;
    if (lane == 0) { var[e] = V; status[e] = st; }
}

// KU = largest group size with a fully unrolled register path in this instantiation (the host picks the smallest
// KU covering the plan, so the common small-k plans keep a small register footprint)
template <int KU>
__global__ __launch_bounds__(256) void k_grad_tiles(const TileDesc *__restrict__ tiles, int64_t n_tiles,
                                                    const double *__restrict__ tvals, const double *__restrict__ v,
                                                    const int32_t *__restrict__ status, int N, int n_out, int n_cand,
                                                    double *__restrict__ grad, int64_t grad_stride,
                                                    const int32_t *__restrict__ gate)
{
    if (gate && *gate == 0) return;   // device-side predication (SPG line-search slots)
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const TileDesc td = tiles[t];
#define GT(KK) case KK: if (KK <= KU) { grad_tile<(KK <= KU ? KK : 1)>(td, tvals, v, status, N, n_out, n_cand, grad, grad_stride, lane); break; }
    switch (td.k) {
        GT(1) GT(2) GT(3) GT(4) GT(5) GT(6) GT(7) GT(8) GT(9) GT(10) GT(11) GT(12)
        default: grad_tile_generic(td, tvals, v, status, N, n_out, n_cand, grad, grad_stride, lane);
    }
#undef GT
}

// Fused solve + gradient pass (single candidate): every workgroup owns 4 tiles of ONE output, folds that output's chunk
// partials and factorises Phi itself (redundantly with the other workgroups of the output -- ~2.5 us of one wavefront,
// no inter-workgroup hand-off, so nothing to synchronise), then its 4 wavefronts evaluate their tiles with v read from
// LDS.  Saves one dependent launch per evaluation.  The tile list is padded so that no workgroup straddles two outputs.
// tiles per workgroup of the fused kernel: wavefront 0 solves, wavefronts 1..TPB own one tile each.  15 (1024 threads,
// 128 VGPRs) while the register-resident matrix (2 NT VGPRs) and tile (KU (KU+1) + KU VGPRs) fit, else 7 (512 threads, 256 VGPRs)
__host__ __device__ constexpr int fused_tpb(int NT, int KU) { return (NT <= 26 && KU <= 8) ? 15 : 7; }
static int pick_ku(int kmax) { return kmax <= 5 ? 5 : kmax <= 6 ? 6 : kmax <= 8 ? 8 : 12; }
template <int NT, int KU>
__global__ __launch_bounds__(64 * (fused_tpb(NT, KU) + 1)) void k_solve_grad(int N, int n_out, const RowDesc *__restrict__ rows, int nsym, FoldReg reg,
                                                    const double2 *__restrict__ partial, const double *__restrict__ rec, double delta,
                                                    const TileDesc *__restrict__ tiles, int64_t n_tiles, int bpo, int tpb, int tile_nt,
                                                    const double *__restrict__ tvals,
                                                    double *__restrict__ var, double *__restrict__ v_ws,
                                                    int32_t *__restrict__ status, double *__restrict__ grad,
                                                    const int32_t *__restrict__ gate, double *__restrict__ spg_state, int last_slot,
                                                    int32_t *__restrict__ spg_enable, unsigned int *__restrict__ ticket, const MaTail ma)
{
    constexpr int FUSED_TPB = fused_tpb(NT, KU);
    constexpr int NTHREADS = 64 * (FUSED_TPB + 1);
    constexpr int PU = tile_pairs(KU);
    __shared__ SolveLds<NT> lds;
    __shared__ double spg_ls[SPG_STATE_DOUBLES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (gate && *gate == 0) {
        // predicated off; the line-search decision still has to close the slot (see k_solve_from_chunks)
        if (spg_state && blockIdx.x == 0 && wave == 0) spg_decide_wave(spg_state, var, status, n_out, last_slot, spg_enable, spg_ls, lane);
        return;
    }
    if (BLUEST_ABLATE == 5) return;
    SPAN_BEGIN(1);
    PHASE(0);
    PHASE_TILE(0);
    const int64_t t0 = (int64_t)blockIdx.x * tpb;      // tpb <= FUSED_TPB tiles per workgroup (wavefronts beyond it only fold)
    // which output, and am I its first workgroup: arithmetic when every output has the same number of workgroups (bpo > 0,
    // the usual case), else from the first tile's descriptor (one more dependent load in front of the fold)
    int o, first;
    if (bpo > 0) { o = blockIdx.x / bpo; first = (blockIdx.x % bpo) == 0; }
    else { const TileDesc td0 = tiles[t0]; o = td0.out; first = (td0.n_valid >> 30) & 1; }
    SpgPrefetch pf;
    if (spg_state && first && wave == 0) spg_prefetch_state(spg_state, lane, pf);   // in flight during the fold (spg_state.hpp)
    if (N < NT) { clear_pads(lds, N, tid, NTHREADS); __syncthreads(); }   // uniform; every real entry is written by the fold
    PHASE(1);
    // Phi either folded from this GPU's chunk partials, or (group-sharded plans) taken from the all-reduced record of output o:
    // N*N sums, then the per-model flags "touched with |m| > 1e-6" / "touched at all" and the flag "max|m| >= 0.05" as counts
    const double *rec_o = rec ? rec + (int64_t)o * (N * N + 2 * N + 1) : nullptr;
    if (rec_o) {
        for (int t = tid; t < N * N; t += NTHREADS) lds.at(t / N, t % N) = rec_o[t];
    } else if (BLUEST_ABLATE == 1) {
        for (int t = tid; t < N * N; t += NTHREADS) lds.at(t / N, t % N) = (t / N == t % N) ? 1.0 : 0.0;
        if (tid < N) lds.amax[tid] = 1.0;
    } else {
        fold_rows<NT>(lds, N, rows, o * nsym, nsym, partial, tid, NTHREADS, reg);
    }
    // the list is padded to a multiple of FUSED_TPB tiles per output, so every tile of this workgroup belongs to output o
    // (loaded by the tile wavefronts only: the solving wavefront must not wait for a descriptor it does not use)
    TileDesc td;
    td.k = 1; td.n_valid = 0; td.val_off = td.grad_off = 0; td.out = (int16_t)o;
    if (wave > 0 && wave <= tpb) td = tiles[t0 + wave - 1];
    __syncthreads();
    PHASE(2);
    const int k = td.k;
    double2 pr[PU];
    double V_pub = 0.0;      // wavefront 0: V and status of this workgroup's solve (for the single-output decision)
    int32_t st_pub = 0;
    if (wave == 0) {
        if (spg_state && first) spg_prefetch_parts(pf, lane);                       // in flight during the elimination
        bool s1, s2, big;
        if (rec_o) {
            s1 = lane < N && rec_o[N * N + lane] > 0.0;
            s2 = lane < N && rec_o[N * N + N + lane] > 0.0;
            big = rec_o[N * N + 2 * N] > 0.0;
        } else {
            const double am = (lane < N) ? lds.amax[lane] : 0.0;
            s1 = am > 1.0e-6;
            s2 = am > 0.0;
            big = __ballot(am >= 0.05) != 0ull;          // max |m| >= 0.05 (misc.py:464)
        }
        double V = 0.0;
        int32_t st = 0;
        PHASE(8);
        if (BLUEST_ABLATE == 2) { if (lane < N) lds.vout[lane] = 1.0; }
        else solve_wave<NT>(lds, N, delta, s1, s2, big, true, &V, lds.vout, &st, lane);
        if (lane == 0) { lds.status = st; if (ma.x) lds.scratch[0] = V; }      // (the elimination is done with its scratch)
        V_pub = V; st_pub = st;
        if (first) {   // first workgroup of this output publishes V, status, v
            if (lane == 0) { var[o] = V; status[o] = st; }
            if (lane < N) v_ws[(int64_t)o * N + lane] = lds.vout[lane];
        }
        PHASE(3);
    } else if (k <= KU && BLUEST_ABLATE != 3) {
        // stream the tile into registers while wavefront 0 factorises (after the fold, so these loads do not queue in front of it)
        PHASE_TILE(1);
        if (tile_nt) tile_load<PU, true>(pr, tvals + td.val_off + 2 * lane, tile_pairs(k));
        else tile_load(pr, tvals + td.val_off + 2 * lane, tile_pairs(k));
        PHASE_TILE(2);
#ifdef BLUEST_PHASE_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PHASE_TILE(3);
#endif
    }
    __syncthreads();
    PHASE_TILE(4);
    if (wave == 0) {
        if (spg_state && first) {
            // SPG line search: the first workgroup of every output has published V and status above; they take a ticket and the
            // last one to arrive decides (same protocol as the tail of k_solve_from_chunks), while the tile wavefronts of all
            // workgroups compute the gradient of this trial point -- it is the one the update needs if the trial is accepted
            int last = 0;
            if (n_out == 1) {   // single output: nobody to wait for, V and status stay in lane 0's registers
                spg_decide_wave(spg_state, var, status, n_out, last_slot, spg_enable, spg_ls, lane, pf, true, V_pub, st_pub);
                SPAN_END(1, true);
                return;
            }
            if (lane == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = (t == (unsigned int)(n_out - 1)) ? 1 : 0;
                if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            last = __builtin_amdgcn_readfirstlane(last);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                spg_decide_wave(spg_state, var, status, n_out, last_slot, spg_enable, spg_ls, lane, pf);
            }
        }
        SPAN_END(1, true);
        return;
    }
    if (BLUEST_ABLATE == 3) return;
    const bool valid = lane < (td.n_valid & 0xffff) && (BLUEST_ABLATE != 7 || pr[0].x == 1.2345);
    const bool inf = lds.status == BLUEST_EVAL_INF;
    double *gout = grad + td.grad_off + lane;
    // (single output, phase 1 of the second-order finish: the update itself instead of the gradient -- k_ma_update's arithmetic)
#ifdef BLUEST_NO_MA_TAIL      // A/B build: the kernel without the tail
    const bool MA_ON = false;
#else
    const bool MA_ON = ma.x != nullptr;
#endif
#define GT(KK) case KK: if (KK <= KU) { const double q = tile_form<(KK <= KU ? KK : 1)>(pr, lds.vout);                                          \
        if (valid && !MA_ON) *gout = inf ? INFINITY : -q;                                                                                        \
        else if (valid && lds.status == BLUEST_EVAL_OK) { const int64_t i = td.grad_off + lane; const double so = ma.s[0], cci = ma.cc[i];       \
            const double xn = ma.x[i] * cci * ((1.0 / so) * q) / (lds.scratch[0] / so); ma.x[i] = xn; ma.m[i] = cci * xn; } break; }
    switch (k) {
        GT(1) GT(2) GT(3) GT(4) GT(5) GT(6) GT(7) GT(8) GT(9) GT(10) GT(11) GT(12)
        default: {
            // grad_tile reads v as v[(c*n_out + td.out)*N + model] and status[c*n_out + td.out]: point both at LDS
            const double *vl = lds.vout - (int64_t)td.out * N;
            const int32_t *sl = &lds.status - td.out;
            grad_tile_generic(td, tvals, vl, sl, N, n_out, 1, grad, 0, lane);
        }
    }
#undef GT
    PHASE(4);
    PHASE_TILE(5);
#ifdef BLUEST_PHASE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE_TILE(6);
#endif
    SPAN_END(1, false);
}

// out[c][j] = scale[j] * sum_o coef[c][o] * grad_o[c][invmap_o[j]]
__global__ void k_combine_grad(const double *__restrict__ grad, int64_t grad_stride, const int64_t *__restrict__ goff,
                               const int32_t *__restrict__ invmap, int64_t L, int n_out,
                               const double *__restrict__ coef, const double *__restrict__ scale, int n_cand,
                               double *__restrict__ out, int64_t out_stride, const int32_t *__restrict__ gate)
{
    if (gate && *gate == 0) return;   // device-side predication (SPG line-search slots)
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= L || c >= n_cand) return;
    double s = 0.0;
    for (int o = 0; o < n_out; o++) {
        const int32_t li = invmap[(int64_t)o * L + j];
        if (li >= 0) s = fma(coef[(int64_t)c * n_out + o], grad[(int64_t)c * grad_stride + goff[o] + li], s);
    }
    out[(int64_t)c * out_stride + j] = scale ? scale[j] * s : s;
}

// ------------------------------------------------------------------------------------------------------
// Part 2 host side: the plan
// ------------------------------------------------------------------------------------------------------
// Device memory of plans comes from a small cache owned by the library: a released block is kept (up to CACHE_CAP bytes) and
// handed to the next request of a similar size.  Measured alternatives: hipMalloc / hipFree of the 45 MB arena cost 10-80 ms
// depending on what else the process holds; HIP's stream-ordered pool (hipMallocAsync) was fast to allocate from but every
// second set-up stalled ~75 ms inside the next null-stream synchronisation while the runtime trimmed and re-mapped the pool.
// Blocks are only recycled after a device synchronisation (plan_release), so no kernel of the previous owner is still running.
#include <map>
#include <unordered_map>
// Blocks belong to the DEVICE they were allocated on: the cache is keyed by (device, size) and a block is only ever handed to a
// request made with the same current device (Plan(device=...) is a public argument; a block of cuda:0 inside a plan of cuda:1
// would be a memory fault, not an error code).
struct BlockInfo { size_t size; int device; };
static std::mutex g_cache_mutex;
static std::multimap<std::pair<int, size_t>, void *> g_cache;   // free blocks by (device, size)
static std::unordered_map<void *, BlockInfo> g_block_info;      // every block handed out or cached
static size_t g_cache_bytes = 0;
static const size_t CACHE_CAP = 3ull << 30;

hipError_t pool_alloc(void **p, size_t bytes)
{
    size_t want = (std::max<size_t>(bytes, 1) + 0x3ffff) & ~(size_t)0x3ffff;            // 256 KiB granules
    if (want <= (8u << 20)) {                  // small requests (restricted plans of the solver): power-of-two size classes, so
        size_t cls = 512u << 10;               // that the next plan of a slightly different size reuses the block
        while (cls < want) cls <<= 1;
        want = cls;
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        auto it = g_cache.lower_bound(std::make_pair(dev, want));
        if (it != g_cache.end() && it->first.first == dev && it->first.second <= 2 * want + (1u << 20)) {
            *p = it->second;
            g_cache_bytes -= it->first.second;
            g_cache.erase(it);
            return hipSuccess;
        }
    }
    e = hipMalloc(p, want);
    if (e != hipSuccess) {       // out of memory: give this device's cached blocks back and try once more
        (void)hipGetLastError();
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lock(g_cache_mutex);
            for (auto it = g_cache.begin(); it != g_cache.end();) {
                if (it->first.first != dev) { ++it; continue; }
                drop.push_back(it->second);
                g_cache_bytes -= it->first.second;
                g_block_info.erase(it->second);
                it = g_cache.erase(it);
            }
        }
        for (void *q : drop) (void)hipFree(q);
        e = hipMalloc(p, want);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        g_block_info[*p] = BlockInfo{want, dev};
    }
    return e;
}
// recycle = false: the block is handed back to the runtime instead of the cache (error paths: nothing that a faulted or
// half-finished sequence touched is given to the next plan)
hipError_t pool_free(void *p, bool recycle)
{
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        auto it = g_block_info.find(p);
        if (it == g_block_info.end()) return hipFree(p);
        const BlockInfo bi = it->second;
        if (recycle && g_cache_bytes + bi.size <= CACHE_CAP) {
            g_cache.emplace(std::make_pair(bi.device, bi.size), p);
            g_cache_bytes += bi.size;
            return hipSuccess;
        }
        g_block_info.erase(it);
    }
    return hipFree(p);
}

typedef DeviceScopeN DeviceScope;      // plan.hpp

static void plan_free_device(bluest_plan_s *p)
{
    mf_release(p);
    if (p->d_arena) (void)pool_free(p->d_arena);
    if (p->d_scratch) (void)pool_free(p->d_scratch);
    if (p->d_master) (void)pool_free(p->d_master);
    p->d_arena = p->d_scratch = p->d_master = nullptr;
    p->master_bytes = 0;
    for (auto &od : p->outs) {
        if (od.d_invcov) (void)pool_free(od.d_invcov);
        if (od.d_groups && od.owns_groups) (void)pool_free(od.d_groups);
        od.d_invcov = nullptr; od.d_groups = nullptr;
    }
}

extern "C" int bluest_plan_create(bluest_plan_t *plan, int n_models, int64_t L_global)
{
    if (!plan) return fail(BLUEST_ERR_ARG, "plan is NULL");
    if (n_models <= 0 || n_models > BLUEST_MAX_MODELS)
        return fail(BLUEST_ERR_ARG, "n_models=%d out of range (1..%d)", n_models, BLUEST_MAX_MODELS);
    if (L_global <= 0 || L_global > 0x7fffffffLL) return fail(BLUEST_ERR_ARG, "L_global=%lld out of range", (long long)L_global);
    int rc = require_gpu(); if (rc) return rc;
    bluest_plan_s *p = new bluest_plan_s();
    p->N = n_models;
    p->L = L_global;
    if (hipGetDevice(&p->device) != hipSuccess) { delete p; return fail(BLUEST_ERR_HIP, "hipGetDevice failed"); }
    *plan = p;
    return BLUEST_OK;
}

// Plan lifetime vs stream capture: freeing device memory (hipFreeAsync on the null stream) while ANOTHER stream of the process
// is being captured into a hipGraph invalidates that capture (hipErrorStreamCaptureInvalidated).  A host language with a
// garbage collector or reference counting can drop a plan at any point, so the library does not rely on the caller's timing:
// between bluest_capture_guard(1) and bluest_capture_guard(0) destroyed plans are parked and released when the guard closes.
static std::mutex g_defer_mutex;
static int g_capture_depth = 0;
static std::vector<bluest_plan_s *> g_deferred;

static void plan_release(bluest_plan_s *p)
{
    DeviceScope scope(p->device);       // the caller may be on another device (a destructor runs wherever the last reference dies)
    (void)hipDeviceSynchronize();       // the blocks go back to the cache: nothing of this plan may still be running
    plan_free_device(p);
    delete p;
}

extern "C" int bluest_capture_guard(int on)
{
    std::vector<bluest_plan_s *> drain;
    {
        std::lock_guard<std::mutex> lock(g_defer_mutex);
        if (on) { g_capture_depth++; return BLUEST_OK; }
        if (g_capture_depth > 0) g_capture_depth--;
        if (g_capture_depth == 0) drain.swap(g_deferred);
    }
    for (bluest_plan_s *p : drain) plan_release(p);
    return BLUEST_OK;
}

extern "C" int bluest_deferred_plans(int *count)
{
    if (!count) return fail(BLUEST_ERR_ARG, "count is NULL");
    std::lock_guard<std::mutex> lock(g_defer_mutex);
    *count = (int)g_deferred.size();
    return BLUEST_OK;
}

extern "C" int bluest_plan_destroy(bluest_plan_t plan)
{
    if (!plan) return BLUEST_OK;
    {
        std::lock_guard<std::mutex> lock(g_defer_mutex);
        if (g_capture_depth > 0) { g_deferred.push_back(plan); return BLUEST_OK; }
    }
    plan_release(plan);
    return BLUEST_OK;
}

static int plan_add_common(bluest_plan_t plan, int K, const int64_t *sizes, const int64_t *groups, const int64_t *mapping,
                           OutputDesc &od)
{
    if (!plan) return fail(BLUEST_ERR_ARG, "plan is NULL");
    if (plan->finalized) return fail(BLUEST_ERR_STATE, "plan already finalized");
    if (K <= 0 || K > BLUEST_MAX_GROUP) return fail(BLUEST_ERR_ARG, "K=%d out of range (1..%d)", K, BLUEST_MAX_GROUP);
    if (!sizes || !groups) return fail(BLUEST_ERR_ARG, "null pointer");
    if ((int)plan->outs.size() >= 32767) return fail(BLUEST_ERR_ARG, "too many outputs");
    od.K = K;
    od.sizes.assign(sizes, sizes + K);
    int64_t L_o = 0, ng = 0;
    for (int k = 1; k <= K; k++) {
        if (sizes[k - 1] < 0) return fail(BLUEST_ERR_ARG, "negative size");
        L_o += sizes[k - 1];
        ng += sizes[k - 1] * k;
    }
    if (L_o <= 0) return fail(BLUEST_ERR_ARG, "output has no groups");
    od.L_o = L_o;
    od.groups.assign(groups, groups + ng);
    for (int64_t t = 0; t < ng; t++)
        if (groups[t] < 0 || groups[t] >= plan->N) return fail(BLUEST_ERR_ARG, "model index %lld out of range", (long long)groups[t]);
    if (mapping) {
        od.mapping.assign(mapping, mapping + L_o);
        for (int64_t t = 0; t < L_o; t++)
            if (mapping[t] < 0 || mapping[t] >= plan->L) return fail(BLUEST_ERR_ARG, "mapping index %lld out of range", (long long)mapping[t]);
    } else {
        if (L_o != plan->L) return fail(BLUEST_ERR_ARG, "identity mapping needs L_o == L_global (%lld vs %lld)", (long long)L_o, (long long)plan->L);
        od.mapping.resize(L_o);
        for (int64_t t = 0; t < L_o; t++) od.mapping[t] = t;
    }
    return BLUEST_OK;
}

// device copies of one output's inputs: the group list (shared with output 0 when identical) and room for its inverses
static int output_to_device(bluest_plan_t plan, OutputDesc &od)
{
    int64_t ni = 0, ng = 0;
    for (int k = 1; k <= od.K; k++) { ni += od.sizes[k - 1] * k * k; ng += od.sizes[k - 1] * k; }
    od.n_inv = ni;
    HIP_TRY(pool_alloc((void **)&od.d_invcov, (size_t)std::max<int64_t>(ni, 1) * sizeof(double)));
    if (!plan->outs.empty() && plan->outs[0].groups == od.groups) {
        od.d_groups = plan->outs[0].d_groups;
        od.owns_groups = false;
    } else {
        HIP_TRY(pool_alloc((void **)&od.d_groups, (size_t)std::max<int64_t>(ng, 1)));
        od.owns_groups = true;
        std::vector<uint8_t> narrow((size_t)ng);
        for (int64_t t = 0; t < ng; t++) narrow[(size_t)t] = (uint8_t)od.groups[(size_t)t];   // validated: 0 <= index < N <= 64
        HIP_TRY(hipMemcpy(od.d_groups, narrow.data(), (size_t)ng, hipMemcpyHostToDevice));
    }
    return BLUEST_OK;
}

static void output_release(OutputDesc &od)
{
    if (od.d_invcov) (void)pool_free(od.d_invcov);
    if (od.d_groups && od.owns_groups) (void)pool_free(od.d_groups);
    od.d_invcov = nullptr; od.d_groups = nullptr;
}

extern "C" int bluest_plan_add_output(bluest_plan_t plan, int K, const int64_t *sizes, const int64_t *groups,
                                      const double *invcovs, const int64_t *mapping)
{
    OutputDesc od;
    PhaseTimer timer("plan_add_output");
    int rc = plan_add_common(plan, K, sizes, groups, mapping, od);
    if (rc) return rc;
    if (!invcovs) return fail(BLUEST_ERR_ARG, "invcovs is NULL");
    timer.lap("copy groups / mapping");
    if ((rc = output_to_device(plan, od))) { output_release(od); return rc; }
    timer.lap("device buffers + groups upload");
    hipError_t e = hipMemcpy(od.d_invcov, invcovs, (size_t)od.n_inv * sizeof(double), hipMemcpyHostToDevice);
    timer.lap("H2D inverses");
    if (e != hipSuccess) { output_release(od); HIP_TRY(e); }
    plan->outs.push_back(std::move(od));
    return BLUEST_OK;
}

extern "C" int bluest_plan_add_output_cov(bluest_plan_t plan, const double *C, int K, const int64_t *sizes,
                                          const int64_t *groups, const int64_t *mapping, double *invcovs_out)
{
    OutputDesc od;
    PhaseTimer timer("plan_add_output_cov");
    int rc = plan_add_common(plan, K, sizes, groups, mapping, od);
    if (rc) return rc;
    timer.lap("copy groups / mapping");
    if (!C) return fail(BLUEST_ERR_ARG, "C is NULL");
    const int N = plan->N;
    if ((rc = output_to_device(plan, od))) { output_release(od); return rc; }
    timer.lap("device buffers + groups upload");
    // the covariance goes through a small scratch that is reused across outputs (freed by finalize)
    const size_t need = (size_t)N * N * sizeof(double);
    if (need > plan->scratch_bytes) {
        if (plan->d_scratch) (void)pool_free(plan->d_scratch);
        plan->d_scratch = nullptr; plan->scratch_bytes = 0;
        hipError_t ea = pool_alloc(&plan->d_scratch, need);
        if (ea != hipSuccess) { output_release(od); HIP_TRY(ea); }
        plan->scratch_bytes = need;
    }
    double *dC = reinterpret_cast<double *>(plan->d_scratch);
    hipError_t e = hipMemcpy(dC, C, need, hipMemcpyHostToDevice);     // synchronous: the scratch may be rewritten by the next call
    timer.lap("H2D covariance");
    if (e == hipSuccess) {
        int64_t go = 0, io = 0;
        for (int k = 1; k <= K && rc == BLUEST_OK; k++) {
            const int64_t Lk = sizes[k - 1];
            if (Lk > 0) rc = launch_group_pinv_u8(dC, N, k, Lk, od.d_groups + go, od.d_invcov + io, 0);
            go += Lk * k; io += Lk * k * k;
        }
        // the next output's covariance upload overwrites dC: wait for the kernels (they take ~0.1 ms; the inverses stay on the device)
        od.h_C.assign(C, C + (size_t)N * N);      // the covariance itself is kept (5 KB): the matrix-free evaluation recomputes the inverses from it
        if (rc == BLUEST_OK) e = hipStreamSynchronize(0);
        if (rc == BLUEST_OK && e == hipSuccess && invcovs_out)
            e = hipMemcpy(invcovs_out, od.d_invcov, (size_t)od.n_inv * sizeof(double), hipMemcpyDeviceToHost);
    }
    timer.lap("pseudo-inverse kernels");
    if (rc) { output_release(od); return rc; }
    if (e != hipSuccess) { output_release(od); HIP_TRY(e); }
    plan->outs.push_back(std::move(od));
    return BLUEST_OK;
}

// reference-layout inverses of output o back to the host (lazily: the plan itself never needs them there)
extern "C" int bluest_plan_get_invcovs(bluest_plan_t plan, int output, double *invcovs_out)
{
    if (!plan || !invcovs_out) return fail(BLUEST_ERR_ARG, "null pointer");
    if (output < 0 || output >= (int)plan->outs.size()) return fail(BLUEST_ERR_ARG, "output %d out of range", output);
    const OutputDesc &od = plan->outs[output];
    HIP_TRY(hipMemcpy(invcovs_out, od.d_invcov, (size_t)od.n_inv * sizeof(double), hipMemcpyDeviceToHost));
    return BLUEST_OK;
}

// inverses of SOME groups of output o (local indices, any order): gathered on the device, one small copy back.
// out receives the k x k blocks one after the other (k = size of each requested group).
__global__ void k_gather_blocks(const double *__restrict__ ic, const int64_t *__restrict__ src_off, const int64_t *__restrict__ dst_off,
                                const int32_t *__restrict__ kk, int64_t n, double *__restrict__ out)
{
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    const int k2 = kk[i] * kk[i];
    for (int t = threadIdx.x; t < k2; t += blockDim.x) out[dst_off[i] + t] = ic[src_off[i] + t];
}

extern "C" int bluest_plan_gather_invcovs(bluest_plan_t plan, int output, const int64_t *local_idx, int64_t n, double *out)
{
    if (!plan || !local_idx || !out) return fail(BLUEST_ERR_ARG, "null pointer");
    if (output < 0 || output >= (int)plan->outs.size()) return fail(BLUEST_ERR_ARG, "output %d out of range", output);
    if (n <= 0) return BLUEST_OK;
    const OutputDesc &od = plan->outs[output];
    std::vector<int64_t> cum(od.K + 1, 0), ioff(od.K + 1, 0);
    for (int k = 1; k <= od.K; k++) { cum[k] = cum[k - 1] + od.sizes[k - 1]; ioff[k] = ioff[k - 1] + od.sizes[k - 1] * k * k; }
    std::vector<int64_t> src(n), dst(n);
    std::vector<int32_t> kk(n);
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t li = local_idx[i];
        if (li < 0 || li >= od.L_o) return fail(BLUEST_ERR_ARG, "group index %lld out of range", (long long)li);
        const int k = (int)(std::upper_bound(cum.begin(), cum.end(), li) - cum.begin());
        src[i] = ioff[k - 1] + (li - cum[k - 1]) * k * k;
        dst[i] = total;
        kk[i] = k;
        total += (int64_t)k * k;
    }
    void *d = nullptr;
    const size_t b_off = (size_t)n * sizeof(int64_t), bytes = 2 * b_off + ((size_t)n * sizeof(int32_t) + 7) / 8 * 8 + (size_t)total * sizeof(double);
    HIP_TRY(pool_alloc(&d, bytes));
    int64_t *d_src = (int64_t *)d, *d_dst = d_src + n;
    int32_t *d_k = (int32_t *)(d_dst + n);
    double *d_out = (double *)((char *)d + 2 * b_off + ((size_t)n * sizeof(int32_t) + 7) / 8 * 8);
    hipError_t e = hipMemcpy(d_src, src.data(), b_off, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_dst, dst.data(), b_off, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_k, kk.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gather_blocks, dim3((unsigned)n), dim3(64), 0, 0, od.d_invcov, d_src, d_dst, d_k, n, d_out);
        e = hipMemcpy(out, d_out, (size_t)total * sizeof(double), hipMemcpyDeviceToHost);
    }
    (void)pool_free(d);
    HIP_TRY(e);
    return BLUEST_OK;
}

// ---- device-side construction of the two streaming layouts ----------------------------------------------
// The host only decides WHERE things go (a counting sort of the group list by destination row gives every packed-symmetric
// entry its slot in the CSR; tiles are arithmetic); the VALUES never leave the device: they are scattered from the
// reference-layout inverses (pinv output) by the two kernels below, for one (output, group size) at a time.
__global__ __launch_bounds__(256) void k_fill_csr(const double *__restrict__ ic, int k, int64_t Lk, const int32_t *__restrict__ perm,
                                                  double *__restrict__ vals)
{
    const int ne = k * (k + 1) / 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Lk * ne) return;
    const int64_t gi = t / ne;
    int e = (int)(t - gi * ne), j = 0;
    while (e >= k - j) { e -= k - j; j++; }          // e-th entry of the packed upper triangle: row j, column l = j + e
    const int l = j + e;
    const double *b = ic + gi * k * k;
    vals[perm[t]] = (j == l) ? b[j * k + j] : 0.5 * (b[j * k + l] + b[l * k + j]);
}

__global__ __launch_bounds__(256) void k_fill_tiles(const double *__restrict__ ic, const uint8_t *__restrict__ groups, int k, int64_t Lk,
                                                    double *__restrict__ tvals)
{   // one thread per (group, entry or index byte); the tile layout is described at TileDesc (plan.hpp)
    const int ne = tile_ne(k), ni = tile_ni(k), S = tile_slots(k);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Lk * (ne + k)) return;
    const int64_t gi = t / (ne + k);
    const int r = (int)(t - gi * (ne + k));
    const int64_t tile = gi >> 6;
    const int lane = (int)(gi & 63);
    double *tb = tvals + tile * S * 64;
    if (r < ne) {
        int e = r, j = 0;
        while (e >= k - j) { e -= k - j; j++; }
        const int l = j + e;
        const double *b = ic + gi * k * k;
        tb[tile_slot_off(ni + r, lane)] = (j == l) ? b[j * k + j] : 0.5 * (b[j * k + l] + b[l * k + j]);
    } else {
        const int j = r - ne;
        reinterpret_cast<uint8_t *>(tb + tile_slot_off(j >> 3, lane))[j & 7] = groups[gi * k + j];
    }
}

// the plan's device arrays live in one arena: reserve() collects sizes, then every array is copied to its slot
struct Arena {
    size_t bytes = 0;
    char *base = nullptr;
    size_t reserve(size_t n) { const size_t off = bytes; bytes += (std::max<size_t>(n, 1) + 255) / 256 * 256; return off; }
};
// a host array that is NOT value-initialised (std::vector zero-fills on one thread and touches every page before the worker
// threads do: 4-5 ms for the 64 MB of slots and columns at K_tot = 245 505); every element is written by the layout passes
template <typename T>
struct RawArray {
    std::unique_ptr<T[]> p;
    size_t n;
    explicit RawArray(size_t n_ = 0) : p(new T[std::max<size_t>(n_, 1)]), n(n_) {}
    void reset(size_t n_) { p.reset(new T[std::max<size_t>(n_, 1)]); n = n_; }
    T *data() { return p.get(); }
    const T *data() const { return p.get(); }
    size_t size() const { return n; }
};
template <typename T>
static int upload(Arena &arena, size_t off, T **dst, const RawArray<T> &src)
{
    *dst = reinterpret_cast<T *>(arena.base + off);
    if (src.size()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return BLUEST_OK;
}
template <typename T>
static int upload(Arena &arena, size_t off, T **dst, const std::vector<T> &src)
{
    *dst = reinterpret_cast<T *>(arena.base + off);
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return BLUEST_OK;
}

// host-side layout construction on up to 16 worker threads
static int host_threads()
{
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min<unsigned>(hw ? hw : 1u, 16u));
}
template <class F>
static void parallel_items(int n_items, F fn)
{
    const int nt = std::min(n_items, host_threads());
    if (nt <= 1) { for (int i = 0; i < n_items; i++) fn(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> workers;
    for (int t = 0; t < nt; t++)
        workers.emplace_back([&]() { for (;;) { const int i = next++; if (i >= n_items) break; fn(i); } });
    for (auto &w : workers) w.join();
}

extern "C" int bluest_plan_finalize(bluest_plan_t plan, int max_candidates)
{
    if (!plan) return fail(BLUEST_ERR_ARG, "plan is NULL");
    if (plan->finalized) return fail(BLUEST_ERR_STATE, "plan already finalized");
    if (plan->outs.empty()) return fail(BLUEST_ERR_STATE, "plan has no outputs");
    if (max_candidates <= 0 || max_candidates > 65535) return fail(BLUEST_ERR_ARG, "max_candidates=%d out of range", max_candidates);
    const int N = plan->N, n_out = (int)plan->outs.size();
    const int nsym = N * (N + 1) / 2;
    auto tri = [N](int a, int b) { return a * N - a * (a - 1) / 2 + (b - a); };
    PhaseTimer timer("plan_finalize");

    plan->nsym = nsym;
    plan->shared = true;
    for (int o = 1; o < n_out; o++) {
        const OutputDesc &x = plan->outs[0], &y = plan->outs[o];
        if (x.K != y.K || x.sizes != y.sizes || x.groups != y.groups || x.mapping != y.mapping) { plan->shared = false; break; }
    }
    // distinct STRUCTURES: with identical group lists and mappings every output has the same rows, chunks and slots, so the
    // counting sort below runs once and the other outputs reuse its result shifted by their chunk base
    const int n_struct = plan->shared ? 1 : n_out;

    // ---- gradient pass: group-major tiles (descriptors only; values are scattered on the device) ----------------
    {
        int kmax_all = 0;
        for (const auto &od : plan->outs) kmax_all = std::max(kmax_all, od.K);
        // tiles per workgroup of the fused kernel.  What bounds its tile stream is the rate at which ONE compute unit pulls bytes
        // that miss its L2 (~30 GB/s), so the tiles are spread over as many compute units as the device has (one workgroup of
        // 1024 threads fills a compute unit's registers): 184 workgroups of 15 tiles left 72 of 256 units idle at the headline size
        const int tpb_max = fused_tpb(pick_nt(N), pick_ku(kmax_all));
        int64_t tiles_max = 1;
        for (const auto &od : plan->outs) {
            int64_t t = 0;
            for (int k = 1; k <= od.K; k++) t += (od.sizes[k - 1] + 63) / 64;
            tiles_max = std::max(tiles_max, t);
        }
        static std::atomic<int> cu_of[64];            // compute units per device (hipGetDeviceProperties costs about a millisecond)
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        int ncu = (dev >= 0 && dev < 64) ? cu_of[dev].load() : 0;
        if (ncu <= 0) {
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            ncu = std::max(1, prop.multiProcessorCount);
            if (dev >= 0 && dev < 64) cu_of[dev].store(ncu);
        }
        const int64_t wg_per_output = std::max<int64_t>(1, ncu / n_out);
        const int64_t tpb_cu = std::max<int64_t>(1, (tiles_max + wg_per_output - 1) / wg_per_output);
        // Phi pass from the tiles (k_phi_tiles): one group list for all outputs, at most 32 workgroups per output (the fold reads
        // one partial per destination and workgroup), the products of a workgroup's tiles staged in LDS
        const int64_t tpb_32 = (tiles_max + 31) / 32;
        plan->stage_stride = tile_ne(kmax_all) * 64;
        // LDS of k_phi_tiles (phi_tiles_lds): products + |m| row per tile slot, sums and maxima of <= stride/16 + nsym/tpb segments each
        const int64_t tpb_lds = ((150 << 10) - 16 * (int64_t)nsym - 2048) / ((int64_t)plan->stage_stride * 9 + 512);
        const char *tiles_env = getenv("BLUEST_PHI_TILES");        // opt-in, read per plan (profiles/r03_phi_tiles_negative_result.txt)
        const bool want_tiles = tiles_env && atoi(tiles_env) != 0;
        plan->phi_tiles = want_tiles && (plan->shared || n_out == 1) && tpb_32 <= std::min<int64_t>(tpb_max, tpb_lds) && kmax_all <= 16;
        if (plan->phi_tiles) plan->fused_tpb = (int)std::min<int64_t>(std::min<int64_t>(tpb_max, tpb_lds), std::max(tpb_cu, tpb_32));
        else plan->fused_tpb = (int)std::min<int64_t>(tpb_max, tpb_cu);
        if (const char *tpb_env = getenv("BLUEST_TPB")) {                       // A/B switch (not for the single-copy pass)
            if (!plan->phi_tiles && atoi(tpb_env) >= 1) plan->fused_tpb = (int)std::min<int64_t>(tpb_max, atoi(tpb_env));
        }
    }
    std::vector<TileDesc> tiles;
    std::vector<std::vector<int64_t>> bucket_val(n_out);    // first tile of size bucket k: offset
    size_t n_tvals = 0;
    plan->grad_off.assign(n_out, 0);
    int64_t grad_len = 0;
    for (int o = 0; o < n_out; o++) {
        const OutputDesc &od = plan->outs[o];
        plan->grad_off[o] = grad_len;
        const size_t first_tile_of_output = tiles.size();
        bucket_val[o].assign(od.K + 1, 0);
        int64_t li = 0;
        for (int k = 1; k <= od.K; k++) {
            const int64_t Lk = od.sizes[k - 1];
            bucket_val[o][k] = (int64_t)n_tvals;
            for (int64_t t0 = 0; t0 < Lk; t0 += 64) {
                TileDesc td;
                td.val_off = (int64_t)n_tvals;
                td.grad_off = grad_len + li + t0;
                td.n_valid = (int32_t)std::min<int64_t>(64, Lk - t0);
                td.k = (int16_t)k; td.out = (int16_t)o;
                n_tvals += (size_t)tile_slots(k) * 64;
                tiles.push_back(td);
            }
            li += Lk;
        }
        // first tile of the output is flagged; the list is padded to a multiple of FUSED_TPB tiles per output with empty
        // tiles so that a workgroup of the fused solve+gradient kernel never straddles two outputs
        if (tiles.size() > first_tile_of_output) tiles[first_tile_of_output].n_valid |= (1 << 30);
        while ((tiles.size() - first_tile_of_output) % plan->fused_tpb || tiles.size() == first_tile_of_output) {
            TileDesc td;
            td.val_off = 0; td.grad_off = 0; td.n_valid = (tiles.size() == first_tile_of_output) ? (1 << 30) : 0;
            td.k = 1; td.out = (int16_t)o;
            tiles.push_back(td);
        }
        const int bpo = (int)((tiles.size() - first_tile_of_output) / plan->fused_tpb);
        if (o == 0) plan->fused_bpo = bpo;
        else if (plan->fused_bpo != bpo) plan->fused_bpo = 0;
        grad_len += od.L_o;
    }
    n_tvals = std::max<size_t>(n_tvals, 128);       // the empty padding tiles read (and ignore) one slot pair at offset 0
    plan->grad_len = grad_len;

    plan->identity = plan->shared && plan->outs[0].L_o == plan->L;
    for (int64_t li = 0; plan->identity && li < plan->L; li++) plan->identity = plan->outs[0].mapping[li] == li;
    // ---- Phi pass, layout 1 (plans that cannot use k_phi_tiles): destination-major symmetric CSR, positions only ---------
    int iters = 1;
    int64_t CH = 256, n_chunks = 0;
    std::vector<RowDesc> rows((size_t)n_out * nsym);
    std::vector<int32_t> out_row_begin(n_out + 1);
    std::vector<int64_t> out_chunk_begin(n_out + 1, 0);
    std::vector<int64_t> struct_entries(n_struct + 1, 0), struct_slots(n_struct + 1, 0);
    RawArray<int32_t> perm, cols;
    std::vector<uint32_t> wg_seg_base;
    std::vector<uint16_t> seg_list, seg_dest, wg_dseg;
    std::vector<int32_t> gmap, pslot;
    plan->fold_reg = FoldReg{0, 0, nullptr};
    std::vector<uint16_t> rank_ab;
    plan->slots_per_output = 0;
    if (!plan->phi_tiles) {
        // parallel counting sort: slice s of S owns a contiguous range of the output's groups; it counts its entries per row,
        // the per-slice counts are prefix-summed into start offsets, then every slice writes the SLOT of its own entries -- rows
        // keep their entries in group order whatever S is, so the layout (and every summation order on the GPU) is independent
        // of threading
        std::vector<std::vector<int64_t>> counts(n_struct, std::vector<int64_t>(nsym, 0));
        int64_t max_row = 0;
        // slices per structure: as many as there are threads for them, but no slice below ~250 k (j, l) pairs -- spawning and joining
        // 16 threads costs more than counting the headline problem's 326 k pairs on one (the layout does not depend on S)
        int64_t pairs0 = 0;
        for (int k = 1; k <= plan->outs[0].K; k++) pairs0 += plan->outs[0].sizes[k - 1] * (int64_t)(k * (k + 1) / 2);
        const int S = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads() / n_struct, pairs0 / 250000));
        std::vector<std::vector<int64_t>> slice_cnt((size_t)n_struct * S, std::vector<int64_t>(nsym, 0));
        auto for_groups_of_slice = [&](int o, int slice, auto &&body) {
            const OutputDesc &od = plan->outs[o];
            const int64_t lo = od.L_o * slice / S, hi = od.L_o * (slice + 1) / S;
            int64_t go = 0, eo = 0, l0 = 0;
            for (int k = 1; k <= od.K; k++) {
                const int64_t Lk = od.sizes[k - 1];
                const int ne = k * (k + 1) / 2;
                for (int64_t i = std::max<int64_t>(lo - l0, 0); i < std::min<int64_t>(hi - l0, Lk); i++)
                    body(k, od.groups.data() + go + i * k, eo + i * ne, l0 + i);
                go += Lk * k; eo += Lk * ne; l0 += Lk;
            }
        };
        parallel_items(n_struct * S, [&](int item) {
            std::vector<int64_t> &cnt = slice_cnt[item];
            for_groups_of_slice(item / S, item % S, [&](int k, const int64_t *g, int64_t, int64_t) {
                for (int j = 0; j < k; j++)
                    for (int l = j; l < k; l++) cnt[tri((int)std::min(g[j], g[l]), (int)std::max(g[j], g[l]))]++;
            });
        });
        for (int o = 0; o < n_struct; o++)
            for (int r = 0; r < nsym; r++) {
                int64_t run = 0;
                for (int sl = 0; sl < S; sl++) { const int64_t c = slice_cnt[(size_t)o * S + sl][r]; slice_cnt[(size_t)o * S + sl][r] = run; run += c; }
                counts[o][r] = run;      // slice_cnt now holds each slice's start offset inside the row
                max_row = std::max(max_row, run);
            }
        timer.lap("count");
        while ((max_row + 256LL * iters - 1) / (256LL * iters) > 64 && iters < 1024) iters *= 2;
        CH = 256LL * iters;
        plan->iters = iters;

        for (int o = 0; o < n_out; o++) {
            out_row_begin[o] = o * nsym;
            out_chunk_begin[o] = n_chunks;
            const std::vector<int64_t> &cnt = counts[plan->shared ? 0 : o];
            for (int a = 0; a < N; a++)
                for (int b = a; b < N; b++) {
                    RowDesc &rd = rows[(size_t)o * nsym + tri(a, b)];
                    const int64_t nc = (cnt[tri(a, b)] + CH - 1) / CH;
                    rd.first_chunk = (int32_t)n_chunks;
                    rd.n_chunks = (int32_t)nc;
                    rd.out = (int16_t)o; rd.a = (int16_t)a; rd.b = (int16_t)b; rd.pad = 0;
                    n_chunks += nc;
                }
        }
        out_row_begin[n_out] = n_out * nsym;
        out_chunk_begin[n_out] = n_chunks;
        if (n_chunks <= 0 || n_chunks * CH > 0x7fffffff0LL) return fail(BLUEST_ERR_ARG, "problem too large for one plan (%lld chunks)", (long long)n_chunks);
        // slot of every packed-symmetric entry (reference order: group-major, upper triangle row by row) and the column
        // (= global index of the group) stored at that slot; per structure, relative to the structure's first chunk
        for (int o = 0; o < n_struct; o++) {
            int64_t ne_all = 0;
            for (int k = 1; k <= plan->outs[o].K; k++) ne_all += plan->outs[o].sizes[k - 1] * (k * (k + 1) / 2);
            struct_entries[o + 1] = struct_entries[o] + ne_all;
            struct_slots[o + 1] = struct_slots[o] + (out_chunk_begin[o + 1] - out_chunk_begin[o]) * CH;
        }
        if (struct_slots[n_struct] > 0x7fffffffLL) return fail(BLUEST_ERR_ARG, "problem too large for one plan (%lld slots per structure set)", (long long)struct_slots[n_struct]);
        perm.reset((size_t)struct_entries[n_struct]);
        cols.reset((size_t)struct_slots[n_struct]);
        parallel_items(n_struct * S, [&](int item) {
            const int o = item / S;
            std::vector<int64_t> &next = slice_cnt[item];     // start offsets, advanced as the slice writes
            const OutputDesc &od = plan->outs[o];
            const int64_t chunk0 = out_chunk_begin[o];
            int32_t *pm = perm.data() + struct_entries[o];
            int32_t *cl = cols.data() + struct_slots[o];
            for_groups_of_slice(o, item % S, [&](int k, const int64_t *g, int64_t e0, int64_t li) {
                int e = 0;
                for (int j = 0; j < k; j++)
                    for (int l = j; l < k; l++, e++) {
                        const int rr = tri((int)std::min(g[j], g[l]), (int)std::max(g[j], g[l]));
                        const int64_t pos = ((int64_t)rows[(size_t)o * nsym + rr].first_chunk - chunk0) * CH + next[rr]++;
                        pm[e0 + e] = (int32_t)pos;
                        cl[pos] = (int32_t)od.mapping[li];
                    }
            });
        });
        // padding: value 0 (the device buffer is cleared), column = the row's first column (keeps max|m| per row exact, adds nothing)
        parallel_items(n_struct, [&](int o) {
            int32_t *cl = cols.data() + struct_slots[o];
            for (int rr = 0; rr < nsym; rr++) {
                const RowDesc &rd = rows[(size_t)o * nsym + rr];
                const int64_t beg = ((int64_t)rd.first_chunk - out_chunk_begin[o]) * CH, end = beg + (int64_t)rd.n_chunks * CH;
                for (int64_t pos = beg + counts[o][rr]; pos < end; pos++) cl[pos] = cl[beg];
            }
        });
        // regular rows (solve.hpp FoldReg): fixed partial strides per destination class when that wastes at most a quarter
        {
            int Cd = 0, Co = 0;
            for (const RowDesc &rd : rows) { if (rd.a == rd.b) Cd = std::max<int>(Cd, rd.n_chunks); else Co = std::max<int>(Co, rd.n_chunks); }
            Co = std::max(Co, 1);
            const int64_t slots = (int64_t)N * Cd + (int64_t)(nsym - N) * Co;
            const bool no_reg = getenv("BLUEST_NO_REGULAR_FOLD") != nullptr;              // A/B switch, read per plan
            if (!no_reg && Cd >= 1 && Cd <= 32 && Co <= 32 && slots * n_out * 4 <= n_chunks * 5 && slots < 0x7fffffff / std::max(1, n_out)) {
                plan->fold_reg = FoldReg{Cd, Co, nullptr};
                for (int a = 0; a < N; a++) rank_ab.push_back((uint16_t)(a | (a << 8)));
                for (int a = 0; a < N; a++) for (int b = a + 1; b < N; b++) rank_ab.push_back((uint16_t)(a | (b << 8)));
                plan->slots_per_output = (int)slots;
                pslot.assign((size_t)(plan->shared ? n_chunks / n_out : n_chunks), 0);
                std::vector<int> off_before(N + 1, 0);          // pairs a' < b' with a' < a
                for (int a = 0; a < N; a++) off_before[a + 1] = off_before[a] + (N - 1 - a);
                for (int o = 0; o < n_struct; o++)
                    for (int a = 0; a < N; a++)
                        for (int b = a; b < N; b++) {
                            const RowDesc &rd = rows[(size_t)o * nsym + tri(a, b)];
                            const int64_t first = (a == b) ? (int64_t)a * Cd : (int64_t)N * Cd + (int64_t)(off_before[a] + (b - a - 1)) * Co;
                            for (int j = 0; j < rd.n_chunks; j++)
                                pslot[(size_t)(rd.first_chunk - (plan->shared ? out_chunk_begin[o] : 0)) + j] = (int32_t)(first + (plan->shared ? 0 : (int64_t)o * slots) + j);
                        }
            }
        }
        timer.lap("CSR slots + columns");
    } else {
        // ---- Phi pass, layout 2: per workgroup of the tile assignment and per destination, the staging positions of its products
        const int bpo = plan->fused_bpo, tpb = plan->fused_tpb, S = plan->stage_stride;
        iters = 0; CH = 0;
        n_chunks = (int64_t)n_out * nsym * bpo;
        for (int o = 0; o < n_out; o++) {
            out_row_begin[o] = o * nsym;
            out_chunk_begin[o] = (int64_t)o * nsym * bpo;
            for (int a = 0; a < N; a++)
                for (int b = a; b < N; b++) {
                    RowDesc &rd = rows[(size_t)o * nsym + tri(a, b)];
                    rd.first_chunk = (int32_t)(((int64_t)o * nsym + tri(a, b)) * bpo);
                    rd.n_chunks = (int32_t)bpo;
                    rd.out = (int16_t)o; rd.a = (int16_t)a; rd.b = (int16_t)b; rd.pad = 0;
                }
        }
        out_row_begin[n_out] = n_out * nsym;
        out_chunk_begin[n_out] = n_chunks;
        const OutputDesc &od = plan->outs[0];
        std::vector<int64_t> go_k(od.K + 2, 0), l0_k(od.K + 2, 0);       // first group (member offset, local index) of every size
        for (int k = 1; k <= od.K; k++) { go_k[k + 1] = go_k[k] + od.sizes[k - 1] * k; l0_k[k + 1] = l0_k[k] + od.sizes[k - 1]; }
        // per workgroup: contributions sorted by destination (counting sort, increasing staging position inside a destination),
        // then cut into segments of 16 padded with the position of the zero word behind the products
        const uint16_t zero_pos = (uint16_t)(tpb * S);
        std::vector<std::vector<uint16_t>> lists(bpo), dests(bpo);
        wg_dseg.assign((size_t)bpo * (nsym + 1), 0);
        parallel_items(bpo, [&](int b) {
            std::vector<uint32_t> cnt(nsym + 1, 0);
            auto for_entries = [&](auto &&body) {
                for (int j = 0; j < tpb; j++) {
                    const TileDesc &td = tiles[(size_t)b * tpb + j];          // output 0's tiles come first in the list
                    const int k = td.k, nv = td.n_valid & 0xffff;
                    const int64_t li0 = td.grad_off - plan->grad_off[0];
                    for (int e = 0, jj = 0; jj < k; jj++)
                        for (int ll = jj; ll < k; ll++, e++)
                            for (int lane = 0; lane < nv; lane++) {
                                const int64_t *g = od.groups.data() + go_k[k] + (li0 + lane - l0_k[k]) * k;
                                body(tri((int)std::min(g[jj], g[ll]), (int)std::max(g[jj], g[ll])), (uint16_t)(j * S + e * 64 + lane));
                            }
                }
            };
            for_entries([&](int d, uint16_t) { cnt[d + 1]++; });
            for (int d = 0; d < nsym; d++) cnt[d + 1] += cnt[d];
            std::vector<uint16_t> sorted(cnt[nsym]);
            std::vector<uint32_t> next(cnt.begin(), cnt.end() - 1);
            for_entries([&](int d, uint16_t pos) { sorted[next[d]++] = pos; });
            std::vector<bool> diag(nsym, false);
            for (int a = 0; a < N; a++) diag[tri(a, a)] = true;
            for (int d = 0; d < nsym; d++) {
                wg_dseg[(size_t)b * (nsym + 1) + d] = (uint16_t)dests[b].size();
                for (uint32_t i = cnt[d]; i < cnt[d + 1]; i += 16) {
                    for (uint32_t t = i; t < i + 16; t++) lists[b].push_back(t < cnt[d + 1] ? sorted[t] : zero_pos);
                    dests[b].push_back((uint16_t)(d | (diag[d] ? 0x8000 : 0)));
                }
            }
            wg_dseg[(size_t)b * (nsym + 1) + nsym] = (uint16_t)dests[b].size();
        });
        wg_seg_base.assign(bpo + 1, 0);
        plan->seg_cap = 1;
        for (int b = 0; b < bpo; b++) {
            wg_seg_base[b + 1] = wg_seg_base[b] + (uint32_t)dests[b].size();
            plan->seg_cap = std::max<int>(plan->seg_cap, (int)dests[b].size());
            seg_list.insert(seg_list.end(), lists[b].begin(), lists[b].end());
            seg_dest.insert(seg_dest.end(), dests[b].begin(), dests[b].end());
        }
        if (plan->seg_cap > 0xffff) return fail(BLUEST_ERR_ARG, "too many contribution segments per workgroup (%d)", plan->seg_cap);
        if (!plan->identity) { gmap.resize(od.L_o); for (int64_t li = 0; li < od.L_o; li++) gmap[li] = (int32_t)od.mapping[li]; }
        timer.lap("contribution lists");
    }
    // ---- inverse maps for combine_grad ------------------------------------------------------------
    std::vector<int32_t> invmap((size_t)n_out * plan->L, -1);
    parallel_items(n_out, [&](int o) {
        for (int64_t li = 0; li < plan->outs[o].L_o; li++) invmap[(size_t)o * plan->L + plan->outs[o].mapping[li]] = (int32_t)li;
    });

    plan->n_chunks = n_chunks;
    plan->cols16 = plan->L <= 65536 && getenv("BLUEST_COLS32") == nullptr;       // (A/B switch: BLUEST_COLS32=1 keeps int32 columns)
    plan->partial_stride = plan->fold_reg.Cd > 0 ? (int64_t)plan->slots_per_output * n_out : n_chunks;
    plan->n_rows = (int64_t)rows.size();
    plan->n_tiles = (int64_t)tiles.size();
    plan->max_cand = max_candidates;
    plan->phi_bytes = plan->phi_tiles ? (int64_t)n_tvals * 8 + (int64_t)seg_list.size() * 2 + (int64_t)seg_dest.size() * 2 + plan->outs[0].L_o * 8 + n_chunks * 16
                                      : n_chunks * CH * 8 + (plan->shared ? n_chunks / n_out : n_chunks) * CH * (plan->cols16 ? 2 : 4) + n_chunks * 16;
    plan->n_segments = (int64_t)seg_dest.size();
    plan->grad_bytes = (int64_t)n_tvals * 8 + grad_len * 8;
    {
        const char *nt_env = getenv("BLUEST_TILE_NT");              // A/B switch, read per plan: 0 = plain loads
        plan->tile_nt = nt_env ? atoi(nt_env) != 0 : true;
    }

    timer.lap("tile descriptors + inverse maps");
    int rc;
    if (plan->d_scratch) { (void)pool_free(plan->d_scratch); plan->d_scratch = nullptr; plan->scratch_bytes = 0; }
    Arena arena;
    const size_t o_vals = arena.reserve((size_t)n_chunks * CH * sizeof(double)), o_cols = arena.reserve((size_t)n_chunks * CH * sizeof(int32_t));
    const size_t o_rows = arena.reserve(rows.size() * sizeof(RowDesc)), o_orb = arena.reserve(out_row_begin.size() * sizeof(int32_t));
    const size_t o_tiles = arena.reserve(tiles.size() * sizeof(TileDesc)), o_tvals = arena.reserve(n_tvals * sizeof(double));
    const size_t o_invmap = arena.reserve(invmap.size() * sizeof(int32_t));
    const size_t o_goff = arena.reserve(plan->grad_off.size() * sizeof(int64_t));
    const size_t o_perm = arena.reserve(perm.size() * sizeof(int32_t));
    const size_t o_ocb = arena.reserve(out_chunk_begin.size() * sizeof(int64_t));
    const size_t o_segbase = arena.reserve(wg_seg_base.size() * sizeof(uint32_t)), o_seglist = arena.reserve(seg_list.size() * sizeof(uint16_t));
    const size_t o_segdest = arena.reserve(seg_dest.size() * sizeof(uint16_t)), o_dseg = arena.reserve(wg_dseg.size() * sizeof(uint16_t));
    const size_t o_gmap = arena.reserve(gmap.size() * sizeof(int32_t));
    const size_t o_partial = arena.reserve((size_t)max_candidates * plan->partial_stride * sizeof(double2));
    const size_t o_pslot = arena.reserve(pslot.size() * sizeof(int32_t)), o_rankab = arena.reserve(rank_ab.size() * sizeof(uint16_t));
    const size_t o_v = arena.reserve((size_t)max_candidates * n_out * N * sizeof(double));
    const size_t o_status = arena.reserve((size_t)max_candidates * n_out * sizeof(int32_t));
    const size_t o_ticket = arena.reserve(256);
    HIP_TRY(pool_alloc(&plan->d_arena, arena.bytes));
    arena.base = (char *)plan->d_arena;
    plan->d_vals = reinterpret_cast<double *>(arena.base + o_vals);
    plan->d_cols = reinterpret_cast<int32_t *>(arena.base + o_cols);
    plan->d_tvals = reinterpret_cast<double *>(arena.base + o_tvals);
    int32_t *d_perm = nullptr;
    // clear what the scatter kernels do not write (padding slots, padding lanes); then the small tables
    if (!plan->phi_tiles) HIP_TRY(hipMemsetAsync(plan->d_vals, 0, (size_t)n_chunks * CH * sizeof(double), 0));
    HIP_TRY(hipMemsetAsync(plan->d_tvals, 0, n_tvals * sizeof(double), 0));
    // columns: the structure's list sits at the structure's own chunk range (shared plans: output 0's range is the one the
    // Phi kernel reads; the other ranges stay unused)
    if (plan->cols16) {
        RawArray<uint16_t> c16(cols.size());
        for (size_t i = 0; i < cols.size(); i++) c16.data()[i] = (uint16_t)cols.data()[i];
        uint16_t *d16 = reinterpret_cast<uint16_t *>(plan->d_cols);
        for (int o = 0; o < n_struct && !plan->phi_tiles; o++)
            HIP_TRY(hipMemcpy(d16 + out_chunk_begin[o] * CH, c16.data() + struct_slots[o],
                              (size_t)(struct_slots[o + 1] - struct_slots[o]) * sizeof(uint16_t), hipMemcpyHostToDevice));
    } else
    for (int o = 0; o < n_struct && !plan->phi_tiles; o++)
        HIP_TRY(hipMemcpy(plan->d_cols + out_chunk_begin[o] * CH, cols.data() + struct_slots[o],
                          (size_t)(struct_slots[o + 1] - struct_slots[o]) * sizeof(int32_t), hipMemcpyHostToDevice));
    if ((rc = upload(arena, o_rows, &plan->d_rows, rows))) return rc;
    if ((rc = upload(arena, o_orb, &plan->d_out_row_begin, out_row_begin))) return rc;
    if ((rc = upload(arena, o_tiles, &plan->d_tiles, tiles))) return rc;
    if ((rc = upload(arena, o_invmap, &plan->d_invmap, invmap))) return rc;
    if ((rc = upload(arena, o_goff, &plan->d_goff, plan->grad_off))) return rc;
    if ((rc = upload(arena, o_perm, &d_perm, perm))) return rc;
    if ((rc = upload(arena, o_ocb, &plan->d_out_chunk_begin, out_chunk_begin))) return rc;
    if ((rc = upload(arena, o_segbase, &plan->d_wg_seg_base, wg_seg_base))) return rc;
    if ((rc = upload(arena, o_seglist, &plan->d_seg_list, seg_list))) return rc;
    if ((rc = upload(arena, o_segdest, &plan->d_seg_dest, seg_dest))) return rc;
    if ((rc = upload(arena, o_dseg, &plan->d_wg_dseg, wg_dseg))) return rc;
    if ((rc = upload(arena, o_gmap, &plan->d_gmap, gmap))) return rc;
    if ((rc = upload(arena, o_pslot, &plan->d_pslot, pslot))) return rc;
    if (pslot.empty()) plan->d_pslot = nullptr;
    {
        uint16_t *d_rank_ab = nullptr;
        if ((rc = upload(arena, o_rankab, &d_rank_ab, rank_ab))) return rc;
        plan->fold_reg.rank_ab = rank_ab.empty() ? nullptr : d_rank_ab;
    }
    if (gmap.empty()) plan->d_gmap = nullptr;
    plan->max_chunks_per_output = 0;
    for (int o = 0; o < n_out; o++) plan->max_chunks_per_output = std::max<int>(plan->max_chunks_per_output, (int)(out_chunk_begin[o + 1] - out_chunk_begin[o]));
    timer.lap("device arena + small uploads");
    // scatter the values on the device: one launch per (output, group size) and layout
    for (int o = 0; o < n_out; o++) {
        const OutputDesc &od = plan->outs[o];
        const int st = plan->shared ? 0 : o;
        int64_t io = 0, go = 0, eo = 0;
        for (int k = 1; k <= od.K; k++) {
            const int64_t Lk = od.sizes[k - 1];
            const int ne = k * (k + 1) / 2;
            if (Lk > 0) {
                if (!plan->phi_tiles)
                hipLaunchKernelGGL(k_fill_csr, dim3((unsigned)((Lk * ne + 255) / 256)), dim3(256), 0, 0, od.d_invcov + io, k, Lk,
                                   d_perm + struct_entries[st] + eo, plan->d_vals + out_chunk_begin[o] * CH);
                hipLaunchKernelGGL(k_fill_tiles, dim3((unsigned)((Lk * (ne + k) + 255) / 256)), dim3(256), 0, 0, od.d_invcov + io,
                                   od.d_groups + go, k, Lk, plan->d_tvals + bucket_val[o][k]);
            }
            io += Lk * k * k; go += Lk * k; eo += Lk * ne;
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(0));
    plan->d_partial = reinterpret_cast<double2 *>(arena.base + o_partial);
    // regular rows read slots no chunk writes: they stay zero for the plan's life
    if (plan->fold_reg.Cd > 0) {
        HIP_TRY(hipMemsetAsync(plan->d_partial, 0, (size_t)max_candidates * plan->partial_stride * sizeof(double2), 0));
        HIP_TRY(hipStreamSynchronize(0));
    }
    plan->d_v = reinterpret_cast<double *>(arena.base + o_v);
    plan->d_status = reinterpret_cast<int32_t *>(arena.base + o_status);
    plan->d_ticket = reinterpret_cast<unsigned int *>(arena.base + o_ticket);
    HIP_TRY(hipMemset(plan->d_ticket, 0, 256));
    plan->finalized = true;
    if ((rc = mf_finalize(plan))) return rc;      // matrix-free evaluation for plans that qualify (matfree.hip)

    timer.lap("device scatter of the values");
    return BLUEST_OK;
}

int plan_ready(bluest_plan_t plan, int n_cand)
{
    if (!plan) return fail(BLUEST_ERR_ARG, "plan is NULL");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    if (n_cand <= 0 || n_cand > plan->max_cand) return fail(BLUEST_ERR_ARG, "n_cand=%d outside 1..%d", n_cand, plan->max_cand);
    return BLUEST_OK;
}

extern "C" int bluest_plan_output_layout(bluest_plan_t plan, int output, int *K, int64_t *sizes, int64_t *mapping)
{
    if (!plan || !K) return fail(BLUEST_ERR_ARG, "null pointer");
    if (output < 0 || output >= (int)plan->outs.size()) return fail(BLUEST_ERR_ARG, "output %d out of range", output);
    const OutputDesc &od = plan->outs[output];
    *K = od.K;
    if (sizes) std::copy(od.sizes.begin(), od.sizes.end(), sizes);
    if (mapping) std::copy(od.mapping.begin(), od.mapping.end(), mapping);
    return BLUEST_OK;
}

// The working set of the solver: a plan over a sub-list of the parent's groups, built natively -- group lists and mappings are
// filtered on the host (they live there), the k x k pseudo-inverse blocks are gathered device to device.
extern "C" int bluest_plan_restrict(bluest_plan_t parent, const int64_t *keep, int64_t n_keep, int max_candidates, bluest_plan_t *restricted)
{
    if (!parent || !keep || !restricted) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!parent->finalized) return fail(BLUEST_ERR_STATE, "parent plan not finalized");
    if (n_keep <= 0 || n_keep > parent->L) return fail(BLUEST_ERR_ARG, "n_keep=%lld out of range", (long long)n_keep);
    PhaseTimer timer("plan_restrict");
    DeviceScope scope(parent->device);                   // the sub-plan lives where its parent's inverses are
    std::vector<int32_t> pos((size_t)parent->L, -1);
    for (int64_t i = 0; i < n_keep; i++) {
        if (keep[i] < 0 || keep[i] >= parent->L || (i > 0 && keep[i] <= keep[i - 1]))
            return fail(BLUEST_ERR_ARG, "keep must be strictly ascending indices in [0, L_global)");
        pos[(size_t)keep[i]] = (int32_t)i;
    }
    bluest_plan_t sub = nullptr;
    int rc = bluest_plan_create(&sub, parent->N, n_keep);
    if (rc) return rc;
    const int n_out = (int)parent->outs.size();
    // selection per output (host), and ONE upload of all gather descriptors
    std::vector<int64_t> src, dst;
    std::vector<int32_t> kk;
    std::vector<int64_t> first_of_output(n_out + 1, 0);
    std::vector<OutputDesc> nods((size_t)n_out);
    for (int o = 0; o < n_out; o++) {
        const OutputDesc &od = parent->outs[o];
        OutputDesc &nd = nods[o];
        nd.K = od.K;
        nd.sizes.assign(od.K, 0);
        bool has0 = false;
        int64_t t0 = 0, goff = 0, ioff = 0, dtot = 0;
        for (int k = 1; k <= od.K; k++) {
            const int64_t Lk = od.sizes[k - 1];
            for (int64_t t = 0; t < Lk; t++) {
                const int32_t q = pos[(size_t)od.mapping[t0 + t]];
                if (q < 0) continue;
                const int64_t *gp = od.groups.data() + goff + t * k;
                for (int j = 0; j < k; j++) has0 = has0 || gp[j] == 0;
                nd.groups.insert(nd.groups.end(), gp, gp + k);
                nd.mapping.push_back(q);
                nd.sizes[k - 1]++;
                src.push_back(ioff + t * k * k);
                dst.push_back(dtot);
                kk.push_back(k);
                dtot += (int64_t)k * k;
            }
            t0 += Lk; goff += Lk * k; ioff += Lk * k * k;
        }
        nd.L_o = (int64_t)nd.mapping.size();
        first_of_output[o + 1] = (int64_t)src.size();
        if (nd.L_o == 0 || !has0) {
            bluest_plan_destroy(sub);
            return fail(BLUEST_ERR_ARG, "restricted plan: output %d would not sample model 0", o);
        }
    }
    timer.lap("select groups");
    const int64_t n_all = (int64_t)src.size();
    void *d = nullptr;
    const size_t b64 = (size_t)n_all * sizeof(int64_t), b32 = ((size_t)n_all * sizeof(int32_t) + 7) / 8 * 8;
    hipError_t e = pool_alloc(&d, 2 * b64 + b32);
    if (e != hipSuccess) { bluest_plan_destroy(sub); HIP_TRY(e); }
    int64_t *d_src = (int64_t *)d, *d_dst = d_src + n_all;
    int32_t *d_k = (int32_t *)(d_dst + n_all);
    e = hipMemcpy(d_src, src.data(), b64, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_dst, dst.data(), b64, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_k, kk.data(), (size_t)n_all * sizeof(int32_t), hipMemcpyHostToDevice);
    for (int o = 0; o < n_out && e == hipSuccess; o++) {
        OutputDesc &nd = nods[o];
        if (output_to_device(sub, nd) != BLUEST_OK) { output_release(nd); e = hipErrorOutOfMemory; break; }
        const int64_t f = first_of_output[o], n = first_of_output[o + 1] - f;
        hipLaunchKernelGGL(k_gather_blocks, dim3((unsigned)n), dim3(64), 0, 0, parent->outs[o].d_invcov, d_src + f, d_dst + f, d_k + f, n,
                           nd.d_invcov);
        sub->outs.push_back(std::move(nd));
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();     // the descriptors are released below
    (void)pool_free(d, e == hipSuccess);                 // after a HIP error the block goes back to the runtime, not to the cache
    if (e != hipSuccess) { bluest_plan_destroy(sub); HIP_TRY(e); }
    timer.lap("gather inverses (device)");
    rc = bluest_plan_finalize(sub, max_candidates);
    if (rc) { bluest_plan_destroy(sub); return rc; }
    *restricted = sub;
    return BLUEST_OK;
}

extern "C" int bluest_plan_n_outputs(bluest_plan_t plan, int *n)
{
    if (!plan || !n) return fail(BLUEST_ERR_ARG, "null pointer");
    *n = (int)plan->outs.size();
    return BLUEST_OK;
}

extern "C" int bluest_plan_grad_layout(bluest_plan_t plan, int64_t *grad_len, int64_t *offsets)
{
    if (!plan) return fail(BLUEST_ERR_ARG, "plan is NULL");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    if (grad_len) *grad_len = plan->grad_len;
    if (offsets) for (size_t o = 0; o < plan->outs.size(); o++) offsets[o] = plan->grad_off[o];
    return BLUEST_OK;
}

extern "C" int bluest_plan_traffic(bluest_plan_t plan, int64_t *phi_bytes, int64_t *grad_bytes)
{
    if (!plan) return fail(BLUEST_ERR_ARG, "plan is NULL");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    if (phi_bytes) *phi_bytes = plan->phi_bytes;
    if (grad_bytes) *grad_bytes = plan->grad_bytes;
    return BLUEST_OK;
}

extern "C" int bluest_plan_phi_len(bluest_plan_t plan, int64_t *len)
{
    if (!plan || !len) return fail(BLUEST_ERR_ARG, "null pointer");
    *len = (int64_t)plan->outs.size() * (plan->N * plan->N + 2 * plan->N + 1);
    return BLUEST_OK;
}

static void launch_grad(bluest_plan_t plan, const double *v_dev, const int32_t *status_dev, int n_cand, double *grad_dev,
                        int64_t grad_stride, hipStream_t st)
{
    const int n_out = (int)plan->outs.size();
    int kmax = 0;
    for (const auto &od : plan->outs) kmax = std::max(kmax, od.K);
#define LG(KU) hipLaunchKernelGGL((k_grad_tiles<KU>), dim3((unsigned)((plan->n_tiles + 3) / 4)), dim3(256), 0, st, plan->d_tiles, plan->n_tiles, \
                                  plan->d_tvals, v_dev, status_dev, plan->N, n_out, n_cand, grad_dev, grad_stride, plan->gate)
    if (kmax <= 5) LG(5);
    else if (kmax <= 8) LG(8);
    else LG(12);
#undef LG
}

// dynamic LDS of k_phi_tiles: products, 64 zeros, |m| rows (one more, zero), segment sums and maxima
static size_t phi_tiles_lds(int tpb, int stage_stride, int seg_cap)
{
    return ((size_t)tpb * stage_stride + 64 + (size_t)(tpb + 1) * 64 + 2 * (size_t)seg_cap) * sizeof(double);
}

static void launch_chunks(bluest_plan_t p, const double *m, int n_cand, int64_t m_stride, hipStream_t st)
{
    const int n_out = (int)p->outs.size();
    if (p->phi_tiles) {
        int kmax = 0;
        for (const auto &od : p->outs) kmax = std::max(kmax, od.K);
        const size_t lds = phi_tiles_lds(p->fused_tpb, p->stage_stride, p->seg_cap);
        const dim3 grid((unsigned)(p->n_tiles / p->fused_tpb));
        PhiTilesArgs A;
        A.tiles = p->d_tiles; A.rows = p->d_rows; A.tvals = p->d_tvals; A.goff = p->d_goff; A.gmap = p->d_gmap;
        A.wg_seg_base = p->d_wg_seg_base; A.seg_list = p->d_seg_list; A.seg_dest = p->d_seg_dest; A.wg_dseg = p->d_wg_dseg;
        A.bpo = p->fused_bpo; A.tpb = p->fused_tpb; A.nsym = p->nsym; A.stage_stride = p->stage_stride; A.seg_cap = p->seg_cap;
        A.stride_inv = (uint32_t)((0x100000000ull + (uint64_t)p->stage_stride - 1) / (uint64_t)p->stage_stride);
        // same block size as the fused kernel of this plan (16 wavefronts while the tile fits 128 registers, else 8)
#define LPT(KU, NW) do {                                                                                                          \
            static size_t lds_set = 0;     /* dynamic LDS beyond 64 KB has to be declared once per kernel */                     \
            if (lds > lds_set) { (void)hipFuncSetAttribute((const void *)k_phi_tiles<KU, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set = lds; } \
            hipLaunchKernelGGL((k_phi_tiles<KU, NW>), grid, dim3(64 * NW), lds, st, A, m, m_stride, n_cand, p->partial_stride, p->d_partial, p->gate); \
        } while (0)
        const bool wide = fused_tpb(pick_nt(p->N), pick_ku(kmax)) == 15;
        // groups of up to 5 models keep their tile in registers; larger ones read the slots from (L2-resident) global memory in
        // the product phase -- with the tile AND the segment positions in registers the 1024-thread form would spill
        if (kmax <= 5) { if (wide) LPT(5, 16); else LPT(5, 8); }
        else { if (wide) LPT(1, 16); else LPT(1, 8); }
#undef LPT
        return;
    }
    // wavefronts per workgroup of the chunk kernels: single-wavefront workgroups drain earliest at the kernel's end (same-box A/B
    // at the headline size, two runs each: step 12.86 / 12.26 / 12.05 us with 4 / 2 / 1 wavefronts, 13.8 with 16)
    static const int wpb = getenv("BLUEST_PHI_WPB") ? atoi(getenv("BLUEST_PHI_WPB")) : 1;
    if (p->shared && n_out >= 2) {
        const int64_t ncpo = p->n_chunks / n_out;
        // outputs per wavefront: sharing the column stream saves bytes, but the pass is latency-bound, so keep at least
        // ~4096 wavefronts in flight (measured at n=20, n_out=8: OB=8 6.9 us, OB=4 5.5 us, OB=2 5.1 us, OB=1 6.1 us)
        int ob = 8;
        while (ob > 2 && (ncpo * ((n_out + ob - 1) / ob) < 4096 || ob > n_out)) ob /= 2;
        static const int ob_env = getenv("BLUEST_PHI_OB") ? atoi(getenv("BLUEST_PHI_OB")) : 0;        // A/B switch
        if (ob_env == 2 || ob_env == 4 || ob_env == 8) ob = ob_env;
#define LCS2(OB, WPB, COLT) hipLaunchKernelGGL((k_phi_chunks_shared<OB, WPB, COLT>), dim3((unsigned)((ncpo + WPB - 1) / WPB), (n_out + OB - 1) / OB), dim3(64 * WPB), 0, st, \
                                        p->d_vals, reinterpret_cast<const COLT *>(p->d_cols), p->iters, ncpo, n_out, m, m_stride, n_cand, p->partial_stride, p->d_pslot, p->slots_per_output, p->d_partial, p->gate)
#define LCS(OB, WPB) do { if (p->cols16) LCS2(OB, WPB, uint16_t); else LCS2(OB, WPB, int32_t); } while (0)
        if (wpb == 1) { if (ob == 8) LCS(8, 1); else if (ob == 4) LCS(4, 1); else LCS(2, 1); }
        else { if (ob == 8) LCS(8, 4); else if (ob == 4) LCS(4, 4); else LCS(2, 4); }
#undef LCS
#undef LCS2
        return;
    }
#define LPC(WPB, COLT) hipLaunchKernelGGL((k_phi_chunks<WPB, COLT>), dim3((unsigned)((p->n_chunks + WPB - 1) / WPB)), dim3(64 * WPB), 0, st, p->d_vals, \
                                       reinterpret_cast<const COLT *>(p->d_cols), p->iters, p->n_chunks, m, m_stride, n_cand, p->d_partial, p->d_pslot, p->partial_stride, p->gate)
    if (wpb == 1) { if (p->cols16) LPC(1, uint16_t); else LPC(1, int32_t); }
    else { if (p->cols16) LPC(4, uint16_t); else LPC(4, int32_t); }
#undef LPC
}

extern "C" int bluest_plan_phi_chunks(bluest_plan_t plan, const double *m_dev, int n_cand, int64_t m_stride, void *stream)
{
    int rc = plan_ready(plan, n_cand); if (rc) return rc;
    if (!m_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (n_cand > 1 && m_stride < plan->L) return fail(BLUEST_ERR_ARG, "m_stride < L_global");
    launch_chunks(plan, m_dev, n_cand, m_stride, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_plan_phi(bluest_plan_t plan, const double *m_dev, int n_cand, int64_t m_stride, double *phi_dev, void *stream)
{
    int rc = plan_ready(plan, n_cand); if (rc) return rc;
    if (!m_dev || !phi_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (n_cand > 1 && m_stride < plan->L) return fail(BLUEST_ERR_ARG, "m_stride < L_global");
    hipStream_t st = (hipStream_t)stream;
    const int n_out = (int)plan->outs.size();
    if (plan->matfree && n_cand == 1 && !plan->gate) return mf_phi_record(plan, m_dev, phi_dev, nullptr, st);
    launch_chunks(plan, m_dev, n_cand, m_stride, st);
#define LFR(NT) hipLaunchKernelGGL((k_fold_to_record<NT>), dim3(n_out, n_cand), dim3(fold_threads(NT)), 0, st, plan->N, n_out, plan->d_rows, \
                                   plan->nsym, plan->fold_reg, plan->d_partial, plan->partial_stride, phi_dev)
    NT_DISPATCH(plan->N, LFR);
#undef LFR
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_plan_solve(bluest_plan_t plan, const double *phi_dev, int n_cand, double delta, double *var_dev,
                                 double *v_dev, int32_t *status_dev, void *stream)
{
    int rc = plan_ready(plan, n_cand); if (rc) return rc;
    if (!phi_dev || !var_dev || !v_dev || !status_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    const int n_out = (int)plan->outs.size();
#define LSR(NT) hipLaunchKernelGGL((k_solve_from_record<NT>), dim3(n_out, n_cand), dim3(64), 0, (hipStream_t)stream, plan->N, n_out, \
                                   phi_dev, delta, 1, var_dev, v_dev, status_dev)
    NT_DISPATCH(plan->N, LSR);
#undef LSR
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_plan_solve_pinv(bluest_plan_t plan, const double *phi_dev, int n_cand, double delta, double *var_dev,
                                      double *v_dev, int32_t *status_dev, void *stream)
{
    int rc = plan_ready(plan, n_cand); if (rc) return rc;
    if (!phi_dev || !var_dev || !v_dev || !status_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    const int n_out = (int)plan->outs.size(), N = plan->N;
    hipLaunchKernelGGL(k_pinv_from_record, dim3(n_out, n_cand), dim3(64), (size_t)2 * N * (N + 1) * sizeof(double),
                       (hipStream_t)stream, N, n_out, phi_dev, delta, var_dev, v_dev, status_dev);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_plan_grad(bluest_plan_t plan, const double *v_dev, const int32_t *status_dev, int n_cand,
                                double *grad_dev, int64_t grad_stride, void *stream)
{
    int rc = plan_ready(plan, n_cand); if (rc) return rc;
    if (!v_dev || !status_dev || !grad_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (n_cand > 1 && grad_stride < plan->grad_len) return fail(BLUEST_ERR_ARG, "grad_stride < grad_len");
    const int n_out = (int)plan->outs.size();
    if (plan->mf_gradient && n_cand == 1 && !plan->gate) return mf_grad(plan, v_dev, status_dev, grad_dev, (hipStream_t)stream);
    launch_grad(plan, v_dev, status_dev, n_cand, grad_dev, grad_stride, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

static int plan_eval(bluest_plan_t plan, const double *m_dev, int n_cand, int64_t m_stride, double delta,
                     double *var_dev, double *grad_dev, int64_t grad_stride, int32_t *status_dev, void *stream,
                     double *dec_state, int dec_last, int32_t *dec_enable, MaTail ma = MaTail{nullptr, nullptr, nullptr, nullptr});

extern "C" int bluest_plan_eval(bluest_plan_t plan, const double *m_dev, int n_cand, int64_t m_stride, double delta,
                                double *var_dev, double *grad_dev, int64_t grad_stride, int32_t *status_dev, void *stream)
{
    return plan_eval(plan, m_dev, n_cand, m_stride, delta, var_dev, grad_dev, grad_stride, status_dev, stream, nullptr, 0, nullptr);
}

// group-sharded plans: the second half of an evaluation FROM the all-reduced Phi record in one launch -- redundant solve in
// every workgroup, gradient tiles of this GPU's shard, optionally the SPG line-search decision in the tail
extern "C" int bluest_plan_solve_grad(bluest_plan_t plan, const double *rec_dev, double delta, double *var_dev, double *grad_dev,
                                      int32_t *status_dev, double *state_dev, int last_slot, int32_t *enable_dev, void *stream)
{
    int rc = plan_ready(plan, 1); if (rc) return rc;
    if (!rec_dev || !var_dev || !grad_dev || !status_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (state_dev && !enable_dev) return fail(BLUEST_ERR_ARG, "the decision needs enable_dev");
    const int n_out = (int)plan->outs.size();
    if (state_dev && n_out > SPG_MAX_OUT) return fail(BLUEST_ERR_ARG, "more than %d outputs", SPG_MAX_OUT);
    hipStream_t st = (hipStream_t)stream;
    if (plan->mf_gradient && !state_dev && !plan->gate) return mf_solve_grad(plan, rec_dev, delta, var_dev, status_dev, grad_dev, st);
    int kmax = 0;
    for (const auto &od : plan->outs) kmax = std::max(kmax, od.K);
    const dim3 grid((unsigned)(plan->n_tiles / plan->fused_tpb));
#define LSR2(NT, KU) hipLaunchKernelGGL((k_solve_grad<NT, KU>), grid, dim3(64 * (fused_tpb(NT, KU) + 1)), 0, st, plan->N, n_out, plan->d_rows, plan->nsym, plan->fold_reg, plan->d_partial, \
                                        rec_dev, delta, plan->d_tiles, plan->n_tiles, plan->fused_bpo, plan->fused_tpb, plan->tile_nt ? 1 : 0, plan->d_tvals, var_dev, plan->d_v, status_dev,  \
                                        grad_dev, plan->gate, state_dev, last_slot, enable_dev, plan->d_ticket, MaTail{nullptr, nullptr, nullptr, nullptr})
#define LSR(NT) do { if (kmax <= 5) LSR2(NT, 5); else if (kmax <= 6) LSR2(NT, 6); else if (kmax <= 8) LSR2(NT, 8); else LSR2(NT, 12); } while (0)
    NT_DISPATCH(plan->N, LSR);
#undef LSR
#undef LSR2
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_plan_eval_decide(bluest_plan_t plan, const double *m_dev, double delta, double *var_dev, int32_t *status_dev,
                                       double *state_dev, int last_slot, int32_t *enable_dev, void *stream)
{
    if (!state_dev || !enable_dev || !status_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (plan && (int)plan->outs.size() > SPG_MAX_OUT) return fail(BLUEST_ERR_ARG, "more than %d outputs", SPG_MAX_OUT);
    return plan_eval(plan, m_dev, 1, 0, delta, var_dev, nullptr, 0, status_dev, stream, state_dev, last_slot, enable_dev);
}

extern "C" int bluest_plan_eval_grad_decide(bluest_plan_t plan, const double *m_dev, double delta, double *var_dev, double *grad_dev,
                                            int32_t *status_dev, double *state_dev, int last_slot, int32_t *enable_dev, void *stream)
{
    if (!state_dev || !enable_dev || !status_dev || !grad_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (plan && (int)plan->outs.size() > SPG_MAX_OUT) return fail(BLUEST_ERR_ARG, "more than %d outputs", SPG_MAX_OUT);
    return plan_eval(plan, m_dev, 1, 0, delta, var_dev, grad_dev, plan ? plan->grad_len : 0, status_dev, stream, state_dev, last_slot, enable_dev);
}

static int plan_eval(bluest_plan_t plan, const double *m_dev, int n_cand, int64_t m_stride, double delta,
                     double *var_dev, double *grad_dev, int64_t grad_stride, int32_t *status_dev, void *stream,
                     double *dec_state, int dec_last, int32_t *dec_enable, MaTail ma)
{
    int rc = plan_ready(plan, n_cand); if (rc) return rc;
    if (!m_dev || !var_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (n_cand > 1 && m_stride < plan->L) return fail(BLUEST_ERR_ARG, "m_stride < L_global");
    if (grad_dev && n_cand > 1 && grad_stride < plan->grad_len) return fail(BLUEST_ERR_ARG, "grad_stride < grad_len");
    hipStream_t st = (hipStream_t)stream;
    const int n_out = (int)plan->outs.size();
    int32_t *status = status_dev ? status_dev : plan->d_status;
    if (plan->matfree && n_cand == 1 && !dec_state && !plan->gate && !g_debug_solve) {
        // matrix-free: Phi pass -> record -> (solve + gradient | solve) -- no stored inverse is read
        const double *rec = nullptr;
        if ((rc = mf_phi_record(plan, m_dev, nullptr, &rec, st))) return rc;
        if (grad_dev) return mf_solve_grad(plan, rec, delta, var_dev, status, grad_dev, st, ma);
#define LSRM(NT) hipLaunchKernelGGL((k_solve_from_record<NT>), dim3(n_out, 1), dim3(64), 0, st, plan->N, n_out, rec, delta, plan->always_v ? 1 : 0, \
                                    var_dev, plan->d_v, status)
        NT_DISPATCH(plan->N, LSRM);
#undef LSRM
        HIP_TRY(hipGetLastError());
        return BLUEST_OK;
    }
    launch_chunks(plan, m_dev, n_cand, m_stride, st);
    if (plan->mf_gradient && grad_dev && n_cand == 1 && !dec_state && !plan->gate && !g_debug_solve)
        return mf_solve_grad(plan, nullptr, delta, var_dev, status, grad_dev, st, ma);  // stored Phi pass + fold, solve and matrix-free gradient
    int kmax = 0;
    for (const auto &od : plan->outs) kmax = std::max(kmax, od.K);
    // fused solve + gradient pass (2 launches per evaluation); groups larger than 12 take the generic tile code inside it
    if (grad_dev && n_cand == 1 && !g_debug_solve) {
        const dim3 grid((unsigned)(plan->n_tiles / plan->fused_tpb));
#define LSG2(NT, KU) hipLaunchKernelGGL((k_solve_grad<NT, KU>), grid, dim3(64 * (fused_tpb(NT, KU) + 1)), 0, st, plan->N, n_out, plan->d_rows, plan->nsym, plan->fold_reg, plan->d_partial, nullptr, \
                                        delta, plan->d_tiles, plan->n_tiles, plan->fused_bpo, plan->fused_tpb, plan->tile_nt ? 1 : 0, plan->d_tvals, var_dev, plan->d_v, status, grad_dev, plan->gate, \
                                        dec_state, dec_last, dec_enable, plan->d_ticket, ma)
#define LSG(NT) do { if (kmax <= 5) LSG2(NT, 5); else if (kmax <= 6) LSG2(NT, 6); else if (kmax <= 8) LSG2(NT, 8); else LSG2(NT, 12); } while (0)
        NT_DISPATCH(plan->N, LSG);
#undef LSG
#undef LSG2
        HIP_TRY(hipGetLastError());
        return BLUEST_OK;
    }
    const int want = ((grad_dev || plan->always_v) ? 1 : 0) | g_debug_solve;
#define LSC(NT) hipLaunchKernelGGL((k_solve_from_chunks<NT>), dim3(n_out, n_cand), dim3(fold_threads(NT)), 0, st, plan->N, n_out, plan->d_rows, \
                                   plan->nsym, plan->fold_reg, plan->d_partial, plan->partial_stride, delta, want, var_dev, plan->d_v, status, plan->gate, \
                                   dec_state, dec_last, dec_enable, plan->d_ticket)
    NT_DISPATCH(plan->N, LSC);
#undef LSC
    if (grad_dev)
        launch_grad(plan, plan->d_v, status, n_cand, grad_dev, grad_stride, st);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_plan_is_identity(bluest_plan_t plan, int *yes)
{
    if (!plan || !yes) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    *yes = plan->identity ? 1 : 0;
    return BLUEST_OK;
}

// phase 1 of the second-order finish on a single-output plan: one evaluation of m_dev whose fused solve + gradient kernel applies the
// multiplicative update to x_dev / m_dev itself (no gradient array, no third launch); var / status as bluest_plan_eval leaves them
extern "C" int bluest_plan_eval_ma(bluest_plan_t plan, double *m_dev, double *var_dev, int32_t *status_dev, const double *s_dev,
                                   const double *cc_dev, double *x_dev, void *stream)
{
    if (!plan || !m_dev || !var_dev || !status_dev || !s_dev || !cc_dev || !x_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    int kmax = 0;
    for (const auto &od : plan->outs) kmax = std::max(kmax, od.K);
    if (plan->outs.size() != 1 || !plan->identity || plan->gate || g_debug_solve || kmax > (plan->matfree ? 8 : 12))
        return fail(BLUEST_ERR_STATE, "bluest_plan_eval_ma: one output on all groups under the identity mapping only");
    // (the gradient pointer only selects the fused kernel: with the tail on, nothing is written through it)
    return plan_eval(plan, m_dev, 1, 0, 0.0, var_dev, plan->d_v, plan->grad_len, status_dev, stream, nullptr, 0, nullptr, MaTail{x_dev, cc_dev, m_dev, s_dev});
}

extern "C" int bluest_plan_combine_grad(bluest_plan_t plan, const double *grad_dev, int64_t grad_stride, const double *coef_dev,
                                        const double *scale_dev, int n_cand, double *out_dev, int64_t out_stride, void *stream)
{
    int rc = plan_ready(plan, n_cand); if (rc) return rc;
    if (!grad_dev || !coef_dev || !out_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    const int n_out = (int)plan->outs.size();
    hipLaunchKernelGGL(k_combine_grad, dim3((unsigned)((plan->L + 255) / 256), n_cand), dim3(256), 0, (hipStream_t)stream, grad_dev,
                       grad_stride, plan->d_goff, plan->d_invmap, plan->L, n_out, coef_dev, scale_dev, n_cand, out_dev, out_stride, plan->gate);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

