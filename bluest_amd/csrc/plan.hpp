// plan.hpp -- the evaluation plan (Part 2 of include/bluest_hip.h) as the translation units see it.
#pragma once
#include "solve.hpp"

struct TileDesc {     // 64 groups of equal size k of one output, for the gradient pass
    int64_t val_off;  // doubles: packed-symmetric entries, [k(k+1)/2][64]
    int64_t idx_off;  // bytes:   model indices, [k][64]
    int64_t grad_off; // position of the tile's first group inside the concatenated gradient
    int32_t n_valid;  // groups in this tile (<= 64); bit 30 set on the first tile of an output
    int16_t k, out;
};


struct OutputDesc {
    int K = 0;
    std::vector<int64_t> sizes;    // K entries
    std::vector<int64_t> groups;   // concat (L_k * k)
    std::vector<double> invcovs;   // concat (L_k * k * k)
    std::vector<int64_t> mapping;  // L_o global indices
    int64_t L_o = 0;
};

struct bluest_plan_s {
    int N = 0;
    int64_t L = 0;
    std::vector<OutputDesc> outs;
    bool finalized = false;
    int max_cand = 0;
    int iters = 1;  // chunk = 256*iters entries
    bool shared = false;  // all outputs have identical groups + mapping
    int fused_bpo = 0;    // workgroups of k_solve_grad per output when that is the same for every output, else 0
    int fused_tpb = 15;   // tiles per workgroup of k_solve_grad for this plan (tile list is padded to it per output)
    const int32_t *gate = nullptr;  // optional device word: 0 = skip the plan's kernels (bluest_plan_set_gate)
    bool always_v = false;          // compute v in every solve (device-side SPG keeps the accepted trial's v)
    int nsym = 0;
    int64_t n_chunks = 0, n_rows = 0, n_tiles = 0, grad_len = 0;
    std::vector<int64_t> grad_off;
    int64_t phi_bytes = 0, grad_bytes = 0;
    // device
    double *d_vals = nullptr;
    int32_t *d_cols = nullptr;
    RowDesc *d_rows = nullptr;
    int32_t *d_out_row_begin = nullptr;
    TileDesc *d_tiles = nullptr;
    double *d_tvals = nullptr;
    uint8_t *d_tidx = nullptr;
    int32_t *d_invmap = nullptr;
    int64_t *d_goff = nullptr;
    double2 *d_partial = nullptr;
    double *d_v = nullptr;       // workspace for eval
    void *d_arena = nullptr;     // ONE device allocation behind all the d_* arrays (hipMalloc is 1-3 ms a piece on a busy process)
    void *d_scratch = nullptr;   // set-up scratch of bluest_plan_add_output_cov (C, groups, inverses), reused across outputs
    size_t scratch_bytes = 0;
    int32_t *d_status = nullptr; // workspace for eval when caller passes NULL
};


int plan_ready(bluest_plan_t plan, int n_cand);
