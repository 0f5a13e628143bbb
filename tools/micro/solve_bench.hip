// Experiment (not product): time the register Gauss-Jordan solve of one wavefront in isolation, several variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I bluest_amd/csrc tools/micro/solve_bench.hip -o /tmp/solve_bench
#include "common.hpp"
#include "solve.hpp"

int fail(int code, const char *, ...) { return code; }
int require_gpu() { return 0; }

// ---- variant 1: reciprocal of the NEXT pivot started as soon as its column is updated (software pipelining) ----
template <int NT>
__device__ __forceinline__ void gj_pipe(double (&a)[NT], int lane, double &rinv_mine, double &last_pivot, int &bad)
{
    double piv = readlane_f64(a[0], 0);
    double rinv = rcp_f64(piv);
#pragma unroll
    for (int j = 0; j < NT; j++) {
        bad |= (!(piv > 0.0) || !isfinite(piv)) ? 1 : 0;
        rinv_mine = (lane == j) ? rinv : rinv_mine;
        if (j == NT - 1) { last_pivot = piv; break; }
        const double f = (lane == j) ? 0.0 : -a[j] * rinv;
        // first the column of the next pivot, then its broadcast + reciprocal chain, then the rest of the row
        const double u1 = readlane_f64(a[j + 1], j);
        a[j + 1] = fma(f, u1, a[j + 1]);
        const double pivn = readlane_f64(a[j + 1], j + 1);
        const double rinvn = rcp_f64(pivn);
        double u[NT];
#pragma unroll
        for (int c = j + 2; c < NT; c++) u[c] = readlane_f64(a[c], j);
#pragma unroll
        for (int c = j + 2; c < NT; c++) a[c] = fma(f, u[c], a[c]);
        piv = pivn; rinv = rinvn;
    }
}

// ---- variant 2: DPP row_newbcast FMA (NT <= 16: all rows inside one 16-lane DPP row) ----
template <int J>
__device__ __forceinline__ double fmac_bcast(double acc, double src, double mul)
{   // acc += (src of lane J of my 16-lane row) * mul
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
    return acc;
}
template <int J>
__device__ __forceinline__ double hb_mov_bcast(double src)
{
    double out;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(src), "n"(J));
    return out;
}
template <int NT, int J>
struct GjDpp {
    static __device__ __forceinline__ void run(double (&a)[NT], int lane, double piv, double rinv, double &rinv_mine, double &last_pivot, int &bad)
    {
        bad |= (!(piv > 0.0) || !isfinite(piv)) ? 1 : 0;
        rinv_mine = ((lane & 15) == J) ? rinv : rinv_mine;
        if constexpr (J == NT - 1) { last_pivot = piv; }
        else {
            const double f = ((lane & 15) == J) ? 0.0 : -a[J] * rinv;
            a[J + 1] = fmac_bcast<J>(a[J + 1], a[J + 1], f);
            const double pivn = hb_mov_bcast<J + 1>(a[J + 1]);
            const double rinvn = rcp_f64(pivn);
#pragma unroll
            for (int c = J + 2; c < NT; c++) a[c] = fmac_bcast<J>(a[c], a[c], f);
            GjDpp<NT, J + 1>::run(a, lane, pivn, rinvn, rinv_mine, last_pivot, bad);
        }
    }
};
template <int NT>
__device__ __forceinline__ void gj_dpp(double (&a)[NT], int lane, double &rinv_mine, double &last_pivot, int &bad)
{
    const double piv = hb_mov_bcast<0>(a[0]);
    GjDpp<NT, 0>::run(a, lane, piv, rcp_f64(piv), rinv_mine, last_pivot, bad);
}


// ---- variant 3: 2x2 block pivots (one reciprocal chain per TWO eliminated columns); NT even.  Returns x = A^-1 e_last per lane.
template <int NT>
__device__ __forceinline__ double gj_block2(double (&a)[NT], int lane, double &V, int &bad)
{
    static_assert(NT % 2 == 0, "NT even");
    double d_own = 0.0, d_oth = 0.0;
#pragma unroll
    for (int b = 0; b < NT - 2; b += 2) {
        const double al = readlane_f64(a[b], b), be = readlane_f64(a[b + 1], b), ga = readlane_f64(a[b + 1], b + 1);
        const double det = fma(al, ga, -be * be);
        bad |= (!(al > 0.0) || !(det > 0.0) || !isfinite(det)) ? 1 : 0;
        const double rdet = rcp_f64(det);
        const double gr = ga * rdet, br = be * rdet, ar = al * rdet;
        const bool inb = (lane >> 1) == (b >> 1);
        d_oth = inb ? -br : d_oth;
        d_own = (lane == b) ? gr : ((lane == b + 1) ? ar : d_own);
        // multipliers of rows b, b+1 for my row: [g0 g1] = [a_b a_b1] P^-1
        const double g0 = inb ? 0.0 : fma(a[b], gr, -a[b + 1] * br);
        const double g1 = inb ? 0.0 : fma(a[b + 1], ar, -a[b] * br);
        double u0[NT], u1[NT];
#pragma unroll
        for (int c = b + 2; c < NT; c++) { u0[c] = readlane_f64(a[c], b); u1[c] = readlane_f64(a[c], b + 1); }
#pragma unroll
        for (int c = b + 2; c < NT; c++) a[c] = fma(-g1, u1[c], fma(-g0, u0[c], a[c]));
    }
    // last block (positions NT-2, NT-1): right-hand side e_last
    const double al = readlane_f64(a[NT - 2], NT - 2), be = readlane_f64(a[NT - 1], NT - 2), ga = readlane_f64(a[NT - 1], NT - 1);
    const double det = fma(al, ga, -be * be);
    bad |= (!(al > 0.0) || !(det > 0.0) || !isfinite(det)) ? 1 : 0;
    const double rdet = rcp_f64(det);
    V = al * rdet;
    // rhs of my row after eliminating the last block: -( a_{i,L0} * (P^-1)_{01} + a_{i,L1} * (P^-1)_{11} )
    const double rhs = -fma(a[NT - 1], al * rdet, -a[NT - 2] * (be * rdet));
    const double rhs_p = __shfl_xor(rhs, 1, 64);
    const double x = fma(d_own, rhs, d_oth * rhs_p);
    return (lane == NT - 1) ? V : ((lane == NT - 2) ? -be * rdet : x);
}

// ---- variant 4: lean per-pivot bookkeeping (sign bits OR-ed on the scalar unit, one Halley step after v_rcp_f64) ----
__device__ __forceinline__ double rcp_halley(double x)
{   // v_rcp_f64 (relative error <= 2^-23) + one third-order step: y = y0 + y0 (e + e^2), e = 1 - x y0  -> error e^3
    const double y0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y0, 1.0);
    return fma(y0, fma(e, e, e), y0);
}
template <int NT>
__device__ __forceinline__ void gj_lean(double (&a)[NT], int lane, double &rinv_mine, double &last_pivot, int &bad)
{
    int signs = 0;
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(a[j]), j), hi = __builtin_amdgcn_readlane(__double2hiint(a[j]), j);
        signs |= hi;
        const double piv = __hiloint2double(hi, lo);
        const double rinv = rcp_halley(piv);
        const bool is = lane == j;
        rinv_mine = is ? rinv : rinv_mine;
        if (j == NT - 1) { last_pivot = piv; break; }
        const double f = is ? 0.0 : -a[j] * rinv;
        double u[NT];
#pragma unroll
        for (int c = j + 1; c < NT; c++) u[c] = readlane_f64(a[c], j);
#pragma unroll
        for (int c = j + 1; c < NT; c++) a[c] = fma(f, u[c], a[c]);
    }
    bad |= (signs < 0 || !(last_pivot > 0.0) || !isfinite(last_pivot)) ? 1 : 0;
}
// ---- variant 5: lean + DPP row_newbcast column updates (rows inside one 16-lane DPP row: NT <= 16) ----
template <int J>
__device__ __forceinline__ double fmac_bcast0(double acc, double src, double mul)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
    return acc;
}
template <int NT, int J>
struct GjDppLean {
    static __device__ __forceinline__ void run(double (&a)[NT], int lane16, int &signs, double &rinv_mine, double &last_pivot)
    {
        const double piv = hb_mov_bcast<J>(a[J]);           // (s_nop 1 inside: a[J] may have been written two instructions ago)
        signs |= __builtin_amdgcn_readfirstlane(__double2hiint(piv));
        const double rinv = rcp_halley(piv);
        const bool is = lane16 == J;
        rinv_mine = is ? rinv : rinv_mine;
        if constexpr (J == NT - 1) { last_pivot = piv; }
        else {
            const double f = is ? 0.0 : -a[J] * rinv;
#pragma unroll
            for (int c = J + 1; c < NT; c++) a[c] = fmac_bcast0<J>(a[c], a[c], f);
            GjDppLean<NT, J + 1>::run(a, lane16, signs, rinv_mine, last_pivot);
        }
    }
};
template <int NT>
__device__ __forceinline__ void gj_dpp_lean(double (&a)[NT], int lane, double &rinv_mine, double &last_pivot, int &bad)
{
    int signs = 0;
    GjDppLean<NT, 0>::run(a, lane & 15, signs, rinv_mine, last_pivot);
    bad |= (signs < 0 || !(last_pivot > 0.0) || !isfinite(last_pivot)) ? 1 : 0;
}

template <int NT, int VAR>
__global__ __launch_bounds__(64) void k_bench(const double *__restrict__ A, int reps, double *__restrict__ out, long long *__restrict__ cycles)
{
    const int lane = threadIdx.x;
    __shared__ double sA[NT * NT];
    __shared__ double sScratch[16 * 17];
    for (int t = lane; t < NT * NT; t += 64) sA[t] = A[t];
    __syncthreads();
    double acc = 0.0;
    long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    for (int rep = 0; rep < reps + 1; rep++) {
        if (rep < 4 && lane == 0) cycles[2 + rep] = __builtin_amdgcn_s_memtime();
        if (rep == 1) { t0 = __builtin_amdgcn_s_memtime(); r0 = wall_clock64(); }
        double a[NT];
        const int row = (VAR == 2 || VAR == 5) ? (lane & 15) : (VAR == 6 ? GjMap<NT>::pos_of(lane) : lane);
#pragma unroll
        for (int c = 0; c < NT; c++) a[c] = (row < NT) ? sA[row * NT + c] + acc * 1e-300 : ((c == row) ? 1.0 : 0.0);
        double rinv_mine = 0.0, last = 1.0;
        int bad = 0;
        if (VAR == 0) gj_regs<NT>(a, lane, (row < NT) ? sA[row * NT + row] : 1.0, rinv_mine, last, bad);
        else if (VAR == 1) gj_pipe<NT>(a, lane, rinv_mine, last, bad);
        else if (VAR == 2) { if constexpr (NT <= 16) gj_dpp<NT>(a, lane, rinv_mine, last, bad); }
        else if (VAR == 4) gj_lean<NT>(a, lane, rinv_mine, last, bad);
        else if (VAR == 5) { if constexpr (NT <= 16) gj_dpp_lean<NT>(a, lane, rinv_mine, last, bad); }
        double x;
        if (VAR == 6) x = gj_solve_last<NT>(a, lane, (row < NT) ? sA[row * NT + row] : 1.0, sScratch, last, bad);
        else if (VAR == 3) { double V; x = gj_block2<NT>(a, lane, V, bad); }
        else {
            const double rl = (VAR == 2 || VAR == 5) ? __shfl(rinv_mine, NT - 1, 64) : readlane_f64(rinv_mine, NT - 1);
            x = (row == NT - 1) ? rl : -a[NT - 1] * rinv_mine * rl;
        }
        acc += x + bad;
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = wall_clock64();
    if (VAR == 6) {      // back to "lane = position" for the check on the host
        __shared__ double xs[64];
        xs[lane] = 0.0;
        __syncthreads();
        if (lane < GjMap<NT>::n_lanes) xs[GjMap<NT>::pos_of(lane)] = acc;
        __syncthreads();
        acc = xs[lane];
    }
    out[lane] = acc;
    if (lane == 0) { cycles[0] = t1 - t0; cycles[1] = r1 - r0; }
}

template <int NT, int VAR>
static void run(const char *name, int reps)
{
    std::vector<double> G(NT * NT), A(NT * NT, 0.0);
    srand(1);
    for (auto &g : G) g = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < NT; i++) for (int j = 0; j < NT; j++) { double s = (i == j) ? 1.0 : 0.0; for (int k = 0; k < NT; k++) s += G[i * NT + k] * G[j * NT + k]; A[i * NT + j] = s; }
    double *dA, *dout; long long *dc;
    hipMalloc(&dA, sizeof(double) * NT * NT); hipMalloc(&dout, sizeof(double) * 64); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), sizeof(double) * NT * NT, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_bench<NT, VAR>), dim3(1), dim3(64), 0, 0, dA, reps, dout, dc);
    hipDeviceSynchronize();
    long long c[8]; double out[64];
    hipMemcpy(c, dc, 64, hipMemcpyDeviceToHost); hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    // reference x = A^-1 e_last (host, Gauss)
    std::vector<double> M(A), b(NT, 0.0); b[NT - 1] = 1.0;
    for (int j = 0; j < NT; j++) for (int i = 0; i < NT; i++) if (i != j) { double f = M[i * NT + j] / M[j * NT + j]; for (int cc = 0; cc < NT; cc++) M[i * NT + cc] -= f * M[j * NT + cc]; b[i] -= f * b[j]; }
    double err = 0.0;
    for (int i = 0; i < NT; i++) err = std::max(err, std::fabs(out[i] / (reps + 1) - b[i] / M[i * NT + i]) / std::fabs(b[NT - 1] / M[NT * NT - 1]));
    printf("%-28s NT=%2d: %8.1f cycles (s_memtime) %7.1f ns per solve, max rel err %.1e | first executions in a launch: %lld %lld %lld cycles\n", name, NT, (double)c[0] / reps, (double)c[1] * 10.0 / reps, err, c[3] - c[2], c[4] - c[3], c[5] - c[4]);
    hipFree(dA); hipFree(dout); hipFree(dc);
}


// ---- the product's whole solve_wave (masks, loads, elimination, v) from LDS, as the kernels call it ----
template <int NT>
__global__ __launch_bounds__(64) void k_wave(const double *__restrict__ A, int N, int reps, double *__restrict__ out, long long *__restrict__ cycles)
{
    __shared__ SolveLds<NT> lds;
    const int lane = threadIdx.x;
    clear_pads(lds, N, lane, 64);
    __syncthreads();
    for (int t = lane; t < N * N; t += 64) lds.at(t / N, t % N) = A[t];
    if (lane < N) lds.amax[lane] = 1.0 + lane;
    __syncthreads();
    double acc = 0.0;
    long long t0 = 0;
    for (int rep = 0; rep < reps + 1; rep++) {
        if (rep == 1) t0 = __builtin_amdgcn_s_memtime();
        const double am = (lane < N) ? lds.amax[lane] : 0.0;
        const bool big = __ballot(am >= 0.05) != 0ull;
        double V = 0.0;
        int32_t st = 0;
        solve_wave<NT>(lds, N, acc * 1e-300, am > 1.0e-6, am > 0.0, big, true, &V, lds.vout, &st, lane);
        acc += V + lds.vout[lane % N] + st;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = acc;
    if (lane == 0) cycles[0] = t1 - t0;
}
template <int NT>
static void run_wave(int N, int reps)
{
    std::vector<double> G(N * N), A(N * N, 0.0);
    srand(1);
    for (auto &g : G) g = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { double s = (i == j) ? 1.0 : 0.0; for (int k = 0; k < N; k++) s += G[i * N + k] * G[j * N + k]; A[i * N + j] = s; }
    double *dA, *dout; long long *dc;
    hipMalloc(&dA, sizeof(double) * N * N); hipMalloc(&dout, sizeof(double) * 64); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_wave<NT>), dim3(1), dim3(64), 0, 0, dA, N, reps, dout, dc);
    hipDeviceSynchronize();
    long long c[8];
    hipMemcpy(c, dc, 64, hipMemcpyDeviceToHost);
    printf("solve_wave<%d> (N = %d), whole call: %8.1f cycles = %7.1f ns at 2.4 GHz\n", NT, N, (double)c[0] / reps, (double)c[0] / reps / 2.4);
    hipFree(dA); hipFree(dout); hipFree(dc);
}

int main()
{
    run_wave<20>(20, 2000);
    run_wave<26>(25, 2000);
    run_wave<12>(12, 2000);
    const int reps = 2000;
    run<8, 6>("product: dpp, extras first", reps);
    run<12, 6>("product: dpp, extras first", reps);
    run<16, 6>("product: dpp, extras first", reps);
    run<20, 6>("product: dpp, extras first", reps);
    run<26, 6>("product: dpp, extras first", reps);
    run<32, 6>("product: dpp, extras first", reps);
    run<32, 0>("readlane (round 2)", reps);
    run<20, 0>("readlane (round 2)", reps);
    run<20, 1>("readlane, pipelined rcp", reps);
    run<20, 4>("readlane, lean bookkeeping", reps);
    run<26, 4>("readlane, lean bookkeeping", reps);
    run<16, 4>("readlane, lean bookkeeping", reps);
    run<12, 4>("readlane, lean bookkeeping", reps);
    run<16, 5>("dpp, lean bookkeeping", reps);
    run<12, 5>("dpp, lean bookkeeping", reps);
    run<8, 5>("dpp, lean bookkeeping", reps);
    run<8, 0>("readlane (product)", reps);
    run<16, 0>("readlane (product)", reps);
    run<16, 1>("readlane, pipelined rcp", reps);
    run<16, 2>("dpp row_newbcast", reps);
    run<12, 0>("readlane (product)", reps);
    run<12, 1>("readlane, pipelined rcp", reps);
    run<12, 2>("dpp row_newbcast", reps);
    run<26, 0>("readlane (product)", reps);
    run<26, 1>("readlane, pipelined rcp", reps);
    return 0;
}
