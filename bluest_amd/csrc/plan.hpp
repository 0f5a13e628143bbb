// plan.hpp -- the evaluation plan (Part 2 of include/bluest_hip.h) as the translation units see it.
#pragma once
#include "solve.hpp"

// Tile = 64 groups of equal size k of one output, lane = group.  Per group S(k) = ne + ni (+1 to make it even) 8-byte SLOTS:
// slots [0, ni) hold the k model indices as bytes (ni = ceil(k/8)), slots [ni, ni+ne) the ne = k(k+1)/2 packed-symmetric
// entries of the group's inverse.  Slots are stored in PAIRS, pair-major: pair p of lane l sits at doubles [p*128 + 2l, +2),
// so one wave-instruction loads 16 bytes per lane, 1 KiB contiguous, and the indices come with the same loads (k = 5: 8 loads
// per tile instead of the 15 + 5 of the 8-byte-per-lane form of rounds 1-2; step 14.4 -> 14.1 us at the headline size).
struct TileDesc {
    int64_t val_off;  // doubles: first slot pair of the tile
    int64_t grad_off; // position of the tile's first group inside the concatenated gradient
    int32_t n_valid;  // groups in this tile (<= 64); bit 30 set on the first tile of an output
    int16_t k, out;
};
__host__ __device__ constexpr int tile_ne(int k) { return k * (k + 1) / 2; }
__host__ __device__ constexpr int tile_ni(int k) { return (k + 7) / 8; }
__host__ __device__ constexpr int tile_slots(int k) { return (tile_ne(k) + tile_ni(k) + 1) & ~1; }
__host__ __device__ constexpr int tile_pairs(int k) { return tile_slots(k) / 2; }
// offset (doubles, relative to the tile) of slot sl of lane l
__host__ __device__ constexpr int64_t tile_slot_off(int sl, int lane) { return (int64_t)(sl >> 1) * 128 + lane * 2 + (sl & 1); }


struct OutputDesc {
    int K = 0;
    std::vector<int64_t> sizes;    // K entries
    std::vector<int64_t> groups;   // concat (L_k * k), host copy (kept: restricted plans, descriptors)
    std::vector<int64_t> mapping;  // L_o global indices
    int64_t L_o = 0;
    int64_t n_inv = 0;             // doubles of the reference-layout inverses: sum L_k k^2
    double *d_invcov = nullptr;    // DEVICE, reference layout concat (L_k * k * k): pinv output or uploaded; lives as long as the plan
    uint8_t *d_groups = nullptr;   // DEVICE copy of `groups`, one byte per model index (n <= 64): 8x less to push over PCIe
                                   // than the int64 list (11 MB at K_tot = 245 505); shared with output 0 when the lists are identical
    bool owns_groups = false;
    std::vector<double> h_C;       // the covariance (N x N) when the output was given by it (host copy: 5 KB; the matrix-free evaluation uploads it)
};

struct bluest_plan_s {
    int device = -1;      // HIP device the plan's memory lives on (current device at bluest_plan_create)
    int N = 0;
    int64_t L = 0;
    std::vector<OutputDesc> outs;
    bool finalized = false;
    int max_cand = 0;
    int iters = 1;  // chunk = 256*iters entries
    bool shared = false;  // all outputs have identical groups + mapping
    bool identity = false; // ... and that mapping is the identity (local index = global index for every output)
    int fused_bpo = 0;    // workgroups of k_solve_grad per output when that is the same for every output, else 0
    int fused_tpb = 15;   // tiles per workgroup of k_solve_grad for this plan (tile list is padded to it per output)
    bool tile_nt = true;  // k_solve_grad streams its tiles with non-temporal loads (same-box A/B over six shapes: 12.8 vs 14.3 us
                          // per step at the headline size, the others within +-2 %; BLUEST_TILE_NT=0 switches it off)
    const int32_t *gate = nullptr;  // optional device word: 0 = skip the plan's kernels (bluest_plan_set_gate)
    bool always_v = false;          // compute v in every solve (device-side SPG keeps the accepted trial's v)
    int nsym = 0;
    int64_t n_chunks = 0, n_rows = 0, n_tiles = 0, grad_len = 0;
    std::vector<int64_t> grad_off;
    int64_t phi_bytes = 0, grad_bytes = 0;
    // device
    double *d_vals = nullptr;
    int32_t *d_cols = nullptr;          // columns of the Phi layout; holds uint16 entries when cols16 (allocation vectors of <= 65 536 entries)
    bool cols16 = false;
    RowDesc *d_rows = nullptr;
    int32_t *d_out_row_begin = nullptr;
    int64_t *d_out_chunk_begin = nullptr;   // n_out + 1: first chunk of every output (its partials are contiguous)
    int max_chunks_per_output = 0;
    // Phi pass from the tiles (k_phi_tiles): the plan keeps ONE copy of the inverses.  Per workgroup of the fused kernel's tile
    // assignment and per symmetric destination, the list of that workgroup's contributions as positions into its LDS staging area
    bool phi_tiles = false;
    int stage_stride = 0;               // doubles of staging per tile slot of a workgroup: ne(k_max) * 64
    int seg_cap = 0;                    // most segments any workgroup has
    uint32_t *d_wg_seg_base = nullptr;  // [bpo + 1]
    uint16_t *d_seg_list = nullptr;     // [segments][16] staging positions (tile slot * stage_stride + entry * 64 + lane)
    uint16_t *d_seg_dest = nullptr;     // [segments] destination, bit 15 = diagonal
    uint16_t *d_wg_dseg = nullptr;      // [bpo][nsym + 1]
    int32_t *d_gmap = nullptr;          // local group index -> global index (NULL: identity)
    int64_t n_segments = 0;
    TileDesc *d_tiles = nullptr;
    double *d_tvals = nullptr;   // tiles: slot pairs (see TileDesc)
    int32_t *d_invmap = nullptr;
    int64_t *d_goff = nullptr;
    FoldReg fold_reg = {0, 0, nullptr};          // regular rows: fixed partial strides per destination class (solve.hpp), else {0, 0}
    int32_t *d_pslot = nullptr;         // chunk -> partial slot (regular rows; per structure)
    int64_t partial_stride = 0;         // partial slots per candidate
    int slots_per_output = 0;
    double2 *d_partial = nullptr;
    double *d_v = nullptr;       // workspace for eval
    void *d_arena = nullptr;     // ONE device allocation behind all the d_* arrays (hipMalloc is 1-3 ms a piece on a busy process)
    void *d_scratch = nullptr;   // set-up scratch of bluest_plan_add_output_cov (C, groups, inverses), reused across outputs
    size_t scratch_bytes = 0;
    int32_t *d_status = nullptr; // workspace for eval when caller passes NULL
    unsigned int *d_ticket = nullptr;   // arrival counter of the fused solve + line-search decision (bluest_plan_eval_decide)
    // second-order finish (newton.hip): descriptor blob of the master problem (plain hipMalloc, grown on demand) and the
    // host-side global -> local group maps it is built from
    // matrix-free evaluation (matfree.hip): chosen at finalize for plans that qualify
    bool matfree = false;        // Phi pass and gradient pass matrix-free
    bool mf_gradient = false;    // gradient pass matrix-free (also true when matfree): the stored Phi pass + k_solve_grad_mf folding its partials
    void *mf = nullptr;
    int32_t *mf_wg_begin_dev = nullptr;
    int mf_wgs_grad = 0, mf_bpo = 0;
    void *d_master = nullptr;
    size_t master_bytes = 0;
    std::vector<std::vector<int32_t>> inv_host;
};

// current device for the lifetime of the object (a plan's memory and kernels live on plan->device)
struct DeviceScopeN {
    int prev = -1;
    bool switched = false;
    explicit DeviceScopeN(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev && dev >= 0) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScopeN() { if (switched) (void)hipSetDevice(prev); }
};


// ---- tile device code shared by plan.hip (gradient pass, fused kernel) and spg.hip (small-plan finish kernel) ----
// the slot pairs of one lane's group, PU pairs of registers (PU >= tile_pairs(K)); loaded with 16-byte loads
typedef double tile_d2v __attribute__((ext_vector_type(2)));
template <int PU, bool NT = false>
__device__ __forceinline__ void tile_load(double2 (&pr)[PU], const double *__restrict__ tile_lane, int n_pairs)
{   // tile_lane = tvals + td.val_off + 2 * lane; NT: non-temporal loads (the fused kernel's stream, see bluest_plan_s::tile_nt)
#pragma unroll
    for (int i = 0; i < PU; i++)
        if (i < n_pairs) {
            if (NT) { const tile_d2v v = __builtin_nontemporal_load(reinterpret_cast<const tile_d2v *>(tile_lane + i * 128)); pr[i] = make_double2(v.x, v.y); }
            else pr[i] = *reinterpret_cast<const double2 *>(tile_lane + i * 128);
        }
}
template <int PU>
__device__ __forceinline__ double tile_slot(const double2 (&pr)[PU], int sl) { return (sl & 1) ? pr[sl >> 1].y : pr[sl >> 1].x; }   // sl static after unrolling

// the quadratic form of one group from its slot pairs: q = v_g^T S v_g (packed symmetric S, K static);
// q = sum_j v_j (s_jj v_j + 2 sum_{l>j} s_jl v_l)
template <int K, int PU>
__device__ __forceinline__ double tile_form(const double2 (&pr)[PU], const double *__restrict__ vc)
{
    constexpr int NI = tile_ni(K);
    double vj[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(tile_slot(pr, j >> 3));
        vj[j] = vc[(int)((bits >> (8 * (j & 7))) & 0xffull)];
    }
    double q = 0.0;
    int e = NI;
#pragma unroll
    for (int j = 0; j < K; j++) {
        double t = 0.0;
#pragma unroll
        for (int l = j + 1; l < K; l++) t = fma(tile_slot(pr, e + (l - j)), vj[l], t);
        t = fma(tile_slot(pr, e), vj[j], 2.0 * t);
        q = fma(vj[j], t, q);
        e += K - j;
    }
    return q;
}

// gradient pass: one wavefront per tile, lane = group
template <int K>
__device__ __forceinline__ void grad_tile(const TileDesc &td, const double *__restrict__ tvals, const double *__restrict__ v,
                                          const int32_t *__restrict__ status, int N, int n_out, int n_cand,
                                          double *__restrict__ grad, int64_t grad_stride, int lane)
{
    double2 pr[tile_pairs(K)];
    tile_load(pr, tvals + td.val_off + 2 * lane, tile_pairs(K));
    for (int c = 0; c < n_cand; c++) {
        const int64_t eo = (int64_t)c * n_out + td.out;
        const double q = tile_form<K>(pr, v + eo * N);
        if (lane < (td.n_valid & 0xffff))
            grad[(int64_t)c * grad_stride + td.grad_off + lane] = (status[eo] == BLUEST_EVAL_INF) ? INFINITY : -q;
    }
}

// generic k (13..16): slots re-read per candidate, no big register arrays
__device__ __forceinline__ void grad_tile_generic(const TileDesc &td, const double *__restrict__ tvals, const double *__restrict__ v,
                                                  const int32_t *__restrict__ status, int N, int n_out, int n_cand,
                                                  double *__restrict__ grad, int64_t grad_stride, int lane)
{
    const int K = td.k;
    const double *tile = tvals + td.val_off;
    const int ni = tile_ni(K);
    auto model = [&](int j) { return (int)reinterpret_cast<const uint8_t *>(tile + tile_slot_off(j >> 3, lane))[j & 7]; };
    for (int c = 0; c < n_cand; c++) {
        const int64_t eo = (int64_t)c * n_out + td.out;
        const double *vc = v + eo * N;
        double q = 0.0;
        int e = ni;
        for (int j = 0; j < K; j++) {
            const double vjj = vc[model(j)];
            double t = 0.0;
            for (int l = j + 1; l < K; l++) t = fma(tile[tile_slot_off(e + (l - j), lane)], vc[model(l)], t);
            t = fma(tile[tile_slot_off(e, lane)], vjj, 2.0 * t);
            q = fma(vjj, t, q);
            e += K - j;
        }
        if (lane < (td.n_valid & 0xffff))
            grad[(int64_t)c * grad_stride + td.grad_off + lane] = (status[eo] == BLUEST_EVAL_INF) ? INFINITY : -q;
    }
}


// single-output plans: the multiplicative update of phase 1 (newton.hip, k_ma_update) done by the fused solve + gradient kernel's
// tile wavefronts instead of writing the gradient: x_i <- x_i cc_i (q_i / s_0) / (V / s_0), m_i = cc_i x_i (x == NULL: off)
struct MaTail { double *x; const double *cc; double *m; const double *s; };

int plan_ready(bluest_plan_t plan, int n_cand);
// the library's device block cache (plan.hip)
hipError_t pool_alloc(void **p, size_t bytes);
hipError_t pool_free(void *p, bool recycle = true);
// matrix-free evaluation (matfree.hip)
int mf_finalize(bluest_plan_t plan);
void mf_release(bluest_plan_s *p);
int mf_phi_record(bluest_plan_t plan, const double *m_dev, double *rec_dev, const double **rec_used, hipStream_t st);
int mf_solve_grad(bluest_plan_t plan, const double *rec_dev, double delta, double *var_dev, int32_t *status_dev, double *grad_dev, hipStream_t st,
                  MaTail ma = MaTail{nullptr, nullptr, nullptr, nullptr});
int mf_grad(bluest_plan_t plan, const double *v_dev, const int32_t *status_dev, double *grad_dev, hipStream_t st);
