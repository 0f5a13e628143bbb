// bluest_hip.hip -- MI355X (gfx950 / CDNA4) kernels + C-ABI for BLUEST's sample-allocation hot path.
//
// What is computed (reference files bluest/*.py, bluest/cmisc.cpp; cited per function in include/bluest_hip.h):
//   Phi_o(m) = sum_i m_i R_i^T C_i^-1 R_i      (misc.py:459-461, cmisc.cpp:25-40)
//   V_o      = (Phi_o[idx,idx]^-1)_00          (misc.py:463-477, :490)
//   grad_o,i = -v[g_i]^T C_i^-1 v[g_i]         (misc.py:493, cmisc.cpp:58-72), v = row 0 of pinv(Phi_o)
// for every output o of a multi-output problem and a batch of candidate allocations m, all float64.
//
// Design: everything here is HBM/L2-streaming integer+f64 work with ~0.2 flop/byte, so there
// is no MFMA; the levers are coalescing, bytes per entry and launch count.
//   * Phi pass : destination-major symmetric CSR of psi, cut into wave-sized chunks; one wavefront streams one
//                chunk with 16-byte loads and gathers m from L2; fixed-order butterfly sum => bit-reproducible
//                (no float atomics).  Chunk partials are folded per row in a fixed order.
//   * solve    : one wavefront per (candidate, output): row fold -> LDS, sampled-model masks by ballot,
//                in-LDS Cholesky restricted to the sampled models, two triangular solves.
//   * grad pass: group-major tiles of 64 groups, lane = group, packed-symmetric inverse stored entry-major so
//                every wave-instruction reads 512 contiguous bytes.
//   * simplex projection: one 1024-thread workgroup, values cached in registers, Michelot/Newton iteration on
//                the threshold with wavefront shuffles + LDS reductions.
// No CPU fallback exists in this file.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>

#include "bluest_hip.h"

// ------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
#ifdef BLUEST_PHASE_TIMING   // experiment builds only (tools/phase_timing.py): 100 MHz timestamps of one workgroup's phases
__device__ long long g_phase[3][12];
#define PHASE(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1)) \
        g_phase[blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : 1)][i] = wall_clock64(); } while (0)
extern "C" int bluest_debug_phase_times(long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(long long) * 36) == hipSuccess ? 0 : 1; }
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

static int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            return fail(BLUEST_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                        __LINE__);                                                                           \
    } while (0)

static int require_gpu()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(BLUEST_ERR_NOGPU, "no HIP device visible (hipGetDeviceCount: %s); libbluest_hip has no CPU path",
                    e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    }
    return BLUEST_OK;
}

extern "C" int bluest_abi_version(void) { return BLUEST_ABI_VERSION; }
extern "C" const char *bluest_last_error(void) { return g_last_error.c_str(); }

extern "C" int bluest_device_count(int *count)
{
    if (!count) return fail(BLUEST_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return BLUEST_OK;
}

extern "C" int bluest_device_name(char *buf, int buflen)
{
    if (!buf || buflen <= 0) return fail(BLUEST_ERR_ARG, "bad buffer");
    int rc = require_gpu();
    if (rc) return rc;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return BLUEST_OK;
}

// ------------------------------------------------------------------------------------------------------
// host<->device staging for the "hd" pointers of Part 1
// ------------------------------------------------------------------------------------------------------
static bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

template <typename T>
struct Staged {  // device view of a host-or-device array; copies back on finish() if writable
    T *dev = nullptr;
    T *host = nullptr;
    size_t count = 0;
    bool owned = false;
    int init(const T *p, size_t n, bool copy_in)
    {
        count = n;
        if (n == 0) { dev = nullptr; return BLUEST_OK; }
        if (is_device_ptr(p)) { dev = const_cast<T *>(p); return BLUEST_OK; }
        host = const_cast<T *>(p);
        owned = true;
        HIP_TRY(hipMalloc((void **)&dev, n * sizeof(T)));
        if (copy_in) HIP_TRY(hipMemcpy(dev, p, n * sizeof(T), hipMemcpyHostToDevice));
        return BLUEST_OK;
    }
    int finish(bool copy_out)
    {
        if (owned && copy_out && count) HIP_TRY(hipMemcpy(host, dev, count * sizeof(T), hipMemcpyDeviceToHost));
        return BLUEST_OK;
    }
    ~Staged() { if (owned && dev) (void)hipFree(dev); }
};

// ------------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------------
#define WAVE 64

__device__ __forceinline__ double wave_sum(double x)
{   // fixed xor-butterfly: every lane ends with the same, order-independent-of-timing sum
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, WAVE);
    return x;
}
__device__ __forceinline__ double wave_max(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off, WAVE));
    return x;
}
__device__ __forceinline__ long long wave_sum_ll(long long x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, WAVE);
    return x;
}

// ------------------------------------------------------------------------------------------------------
// Part 1 kernels -- reference layout (int64 groups (Lk,k), f64 invcov (Lk,k,k) row-major)
// ------------------------------------------------------------------------------------------------------

// cmisc.cpp:10-23.  thread = group i; column i of psi is written with stride Lk => coalesced across lanes.
__global__ void k_assemble_psi(double *__restrict__ psi, int N, int k, int64_t Lk, const int64_t *__restrict__ g,
                               const double *__restrict__ ic)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Lk) return;
    const int64_t *gi = g + i * k;
    const double *ici = ic + i * (int64_t)k * k;
    for (int j = 0; j < k; j++) {
        const int64_t gj = gi[j];
        for (int l = 0; l < k; l++) psi[Lk * (N * gj + gi[l]) + i] += ici[k * j + l];
    }
}

// cmisc.cpp:25-40.  Each workgroup accumulates its slice of groups into an LDS copy of Phi with LDS f64 atomics
// (ds_add_f64), then writes it as one slab; k_fold_slabs adds the slabs to PHI in a fixed order.
template <typename MT>
__global__ void k_objectiveK(double *__restrict__ slabs, int N, int k, int64_t Lk, const MT *__restrict__ mk,
                             const int64_t *__restrict__ g, const double *__restrict__ ic)
{
    extern __shared__ __attribute__((aligned(16))) double sphi[];
    const int NN = N * N;
    for (int t = threadIdx.x; t < NN; t += blockDim.x) sphi[t] = 0.0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Lk; i += (int64_t)gridDim.x * blockDim.x) {
        const double mi = (double)mk[i];
        const int64_t *gi = g + i * k;
        const double *ici = ic + i * (int64_t)k * k;
        for (int j = 0; j < k; j++) {
            const int gj = (int)gi[j];
            for (int l = 0; l < k; l++) atomicAdd(&sphi[N * gj + (int)gi[l]], mi * ici[k * j + l]);
        }
    }
    __syncthreads();
    double *out = slabs + (int64_t)blockIdx.x * NN;
    for (int t = threadIdx.x; t < NN; t += blockDim.x) out[t] = sphi[t];
}

__global__ void k_fold_slabs(double *__restrict__ PHI, const double *__restrict__ slabs, int NN, int nslabs)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NN) return;
    double s = 0.0;
    for (int b = 0; b < nslabs; b++) s += slabs[(int64_t)b * NN + t];
    PHI[t] += s;
}

// cmisc.cpp:58-72.  thread = group.
__global__ void k_gradK(double *__restrict__ grad, int k, int64_t Lk, const int64_t *__restrict__ g,
                        const double *__restrict__ ic, const double *__restrict__ v, int n_models)
{
    __shared__ double sv[BLUEST_MAX_MODELS * 4];
    for (int t = threadIdx.x; t < n_models; t += blockDim.x) sv[t] = v[t];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Lk) return;
    const int64_t *gi = g + i * k;
    const double *ici = ic + i * (int64_t)k * k;
    double acc = 0.0;
    for (int j = 0; j < k; j++) {
        const double vj = sv[gi[j]];
        for (int l = 0; l < k; l++) acc += vj * ici[k * j + l] * sv[gi[l]];
    }
    grad[i] += acc;
}

// cmisc.cpp:42-56 with the `=` of line 51: the last l written wins, i.e. l = k-1 (a later j with the same
// model would overwrite too, exactly as the sequential reference does).
__global__ void k_cleanupK(double *__restrict__ X, int k, int64_t Lk, const int64_t *__restrict__ g,
                           const double *__restrict__ ic, const double *__restrict__ v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Lk) return;
    const int64_t *gi = g + i * k;
    const double *ici = ic + i * (int64_t)k * k;
    for (int j = 0; j < k; j++)
        for (int l = 0; l < k; l++) X[Lk * gi[j] + i] = ici[k * j + l] * v[gi[l]];
}

// cmisc.cpp:74-97 factored: a_k(ik)_j = sum_l v[gk_l] Ck[l,j]  (k doubles per group), then
// hess[ik,iq] += sum_{j,j'} a_k(ik)_j invPHI[gk_j, gq_j'] a_q(iq)_j'.
__global__ void k_hess_avec(double *__restrict__ a, int k, int64_t Lk, const int64_t *__restrict__ g,
                            const double *__restrict__ ic, const double *__restrict__ v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Lk) return;
    const int64_t *gi = g + i * k;
    const double *ici = ic + i * (int64_t)k * k;
    for (int j = 0; j < k; j++) {
        double s = 0.0;
        for (int l = 0; l < k; l++) s += v[gi[l]] * ici[k * l + j];
        a[i * k + j] = s;
    }
}
__global__ void k_hessKQ(double *__restrict__ hess, int N, int k, int q, int64_t Lk, int64_t Lq,
                         const int64_t *__restrict__ gk, const int64_t *__restrict__ gq,
                         const double *__restrict__ ak, const double *__restrict__ aq,
                         const double *__restrict__ invPHI)
{
    const int64_t iq = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ik = blockIdx.y;
    if (iq >= Lq || ik >= Lk) return;
    double s = 0.0;
    for (int j = 0; j < k; j++) {
        const double akj = ak[ik * k + j];
        const double *row = invPHI + (int64_t)N * gk[ik * k + j];
        for (int jq = 0; jq < q; jq++) s += akj * row[gq[iq * q + jq]] * aq[iq * q + jq];
    }
    hess[ik * Lq + iq] += s;
}

// sap.py:69-79: pinv(C[g,g]) per group by cyclic Jacobi (symmetric eigen-decomposition), thread = group.
// Matches numpy.linalg.pinv for symmetric input: drop |lambda| <= 1e-15*max|lambda|.
template <int K>
__device__ __forceinline__ void sym_pinv_jacobi(double (&A)[K * K], double (&V)[K * K], double *__restrict__ out)
{
#pragma unroll
    for (int i = 0; i < K; i++)
#pragma unroll
        for (int j = 0; j < K; j++) V[i * K + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            diag += A[i * K + i] * A[i * K + i];
#pragma unroll
            for (int j = i + 1; j < K; j++) off += A[i * K + j] * A[i * K + j];
        }
        if (off <= 1e-60 * (diag + off) || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < K - 1; p++)
#pragma unroll
            for (int q = p + 1; q < K; q++) {
                const double apq = A[p * K + q];
                if (apq != 0.0) {
                    const double theta = (A[q * K + q] - A[p * K + p]) / (2.0 * apq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        const double arp = A[r * K + p], arq = A[r * K + q];
                        A[r * K + p] = c * arp - s * arq;
                        A[r * K + q] = s * arp + c * arq;
                    }
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        const double apr = A[p * K + r], aqr = A[q * K + r];
                        A[p * K + r] = c * apr - s * aqr;
                        A[q * K + r] = s * apr + c * aqr;
                    }
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        const double vrp = V[r * K + p], vrq = V[r * K + q];
                        V[r * K + p] = c * vrp - s * vrq;
                        V[r * K + q] = s * vrp + c * vrq;
                    }
                }
            }
    }
    double wmax = 0.0;
#pragma unroll
    for (int i = 0; i < K; i++) wmax = fmax(wmax, fabs(A[i * K + i]));
    const double cut = 1e-15 * wmax;
#pragma unroll
    for (int i = 0; i < K; i++)
#pragma unroll
        for (int j = 0; j < K; j++) {
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < K; e++) {
                const double w = A[e * K + e];
                const double inv = (fabs(w) > cut) ? 1.0 / w : 0.0;
                s += V[i * K + e] * inv * V[j * K + e];
            }
            out[i * K + j] = s;
        }
}

template <int K>
__global__ __launch_bounds__(64) void k_group_pinv(const double *__restrict__ C, int N, int64_t Lk, const int64_t *__restrict__ g,
                             double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Lk) return;
    double A[K * K], V[K * K];
    const int64_t *gi = g + i * K;
#pragma unroll
    for (int j = 0; j < K; j++)
#pragma unroll
        for (int l = 0; l < K; l++) {
            const int64_t a = gi[j], b = gi[l];
            A[j * K + l] = 0.5 * (C[a * N + b] + C[b * N + a]);
        }
    sym_pinv_jacobi<K>(A, V, out + i * (int64_t)K * K);
}

// ------------------------------------------------------------------------------------------------------
// Part 2 -- plan kernels
// ------------------------------------------------------------------------------------------------------

struct RowDesc {      // one symmetric destination (a <= b) of one output
    int32_t first_chunk;
    int32_t n_chunks;
    int16_t out, a, b, pad;
};

struct TileDesc {     // 64 groups of equal size k of one output, for the gradient pass
    int64_t val_off;  // doubles: packed-symmetric entries, [k(k+1)/2][64]
    int64_t idx_off;  // bytes:   model indices, [k][64]
    int64_t grad_off; // position of the tile's first group inside the concatenated gradient
    int32_t n_valid;  // groups in this tile (<= 64); bit 30 set on the first tile of an output
    int16_t k, out;
};

// Phi pass: one wavefront per chunk of CH = 256*iters entries; lane l owns entries [4l, 4l+4) of each 256-block.
__global__ __launch_bounds__(256) void k_phi_chunks(const double *__restrict__ vals, const int32_t *__restrict__ cols,
                                                    int iters, int64_t n_chunks, const double *__restrict__ m,
                                                    int64_t m_stride, int n_cand, double2 *__restrict__ partial,
                                                    const int32_t *__restrict__ gate)
{
    if (gate && *gate == 0) return;   // device-side predication (SPG line-search slots)
    const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (chunk >= n_chunks) return;
    const int64_t base = chunk * (int64_t)iters * 256 + lane * 4;
    for (int c = 0; c < n_cand; c++) {
        const double *mc = m + (int64_t)c * m_stride;
        double s = 0.0, amax = 0.0;
        for (int it = 0; it < iters; it++) {
            const double2 v01 = *reinterpret_cast<const double2 *>(vals + base + it * 256);
            const double2 v23 = *reinterpret_cast<const double2 *>(vals + base + it * 256 + 2);
            const int4 cc = *reinterpret_cast<const int4 *>(cols + base + it * 256);
            const double m0 = mc[cc.x], m1 = mc[cc.y], m2 = mc[cc.z], m3 = mc[cc.w];
            s = fma(v01.x, m0, s);
            s = fma(v01.y, m1, s);
            s = fma(v23.x, m2, s);
            s = fma(v23.y, m3, s);
            amax = fmax(fmax(amax, fmax(fabs(m0), fabs(m1))), fmax(fabs(m2), fabs(m3)));
        }
        s = wave_sum(s);
        amax = wave_max(amax);
        if (lane == 0) partial[(int64_t)c * n_chunks + chunk] = make_double2(s, amax);
    }
}

// Phi pass, shared structure: when every output has the same groups and mapping (the usual multi-output case) the
// column indices and the gathered m are common; one wavefront streams the chunk of OB outputs and reads them once.
// vals / partial keep the output-major chunk numbering of the general layout (chunk id = o*ncpo + c).
template <int OB>
__global__ __launch_bounds__(256) void k_phi_chunks_shared(const double *__restrict__ vals, const int32_t *__restrict__ cols,
                                                           int iters, int64_t ncpo, int n_out, const double *__restrict__ m,
                                                           int64_t m_stride, int n_cand, int64_t n_chunks,
                                                           double2 *__restrict__ partial, const int32_t *__restrict__ gate)
{
    if (gate && *gate == 0) return;   // device-side predication (SPG line-search slots)
    const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int o0 = blockIdx.y * OB;
    if (chunk >= ncpo) return;
    SPAN_BEGIN(0);
    const int64_t CH = (int64_t)iters * 256;
    const int64_t base = chunk * CH + lane * 4;
    for (int c = 0; c < n_cand; c++) {
        const double *mc = m + (int64_t)c * m_stride;
        double s[OB];
#pragma unroll
        for (int oo = 0; oo < OB; oo++) s[oo] = 0.0;
        double amax = 0.0;
        for (int it = 0; it < iters; it++) {
            const int4 cc = *reinterpret_cast<const int4 *>(cols + base + it * 256);
            double2 v01[OB], v23[OB];
#pragma unroll
            for (int oo = 0; oo < OB; oo++) {
                const int o = (o0 + oo < n_out) ? o0 + oo : n_out - 1;
                const double *vp = vals + (int64_t)o * ncpo * CH + base + it * 256;
                v01[oo] = *reinterpret_cast<const double2 *>(vp);
                v23[oo] = *reinterpret_cast<const double2 *>(vp + 2);
            }
            const double m0 = mc[cc.x], m1 = mc[cc.y], m2 = mc[cc.z], m3 = mc[cc.w];
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
            if (lane == 0 && o0 + oo < n_out) partial[(int64_t)c * n_chunks + (int64_t)(o0 + oo) * ncpo + chunk] = make_double2(t, amax);
        }
    }
    SPAN_END(0, false);
}

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

__device__ __forceinline__ double rsqrt_f64(double x)
{   // v_rsq_f64 seed + two Newton steps (y <- y + y*(1 - x y^2)/2): full double accuracy for normal x > 0
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
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

// fused: fold chunk partials + solve.  grid = (n_out, n_cand), block = fold_threads (fold) -> wavefront 0 (solve).
// want_v: bit0 = also produce v (gradient wanted); bit1 / bit2 = timing diagnostics (fold only / solve twice).
template <int NT>
__global__ __launch_bounds__(fold_threads(NT)) void k_solve_from_chunks(int N, int n_out, const RowDesc *__restrict__ rows, int nsym,
                                                           const double2 *__restrict__ partial, int64_t n_chunks,
                                                           double delta, int want_v, double *__restrict__ var,
                                                           double *__restrict__ v, int32_t *__restrict__ status,
                                                           const int32_t *__restrict__ gate)
{
    __shared__ SolveLds<NT> lds;
    if (gate && *gate == 0) return;   // device-side predication (SPG line-search slots)
    const int o = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    if (tid < N) lds.amax[tid] = 0.0;
    for (int t = tid; t < N * N; t += fold_threads(NT)) lds.phi[t] = 0.0;
    __syncthreads();
    fold_rows(lds, N, rows, o * nsym, nsym, partial + (int64_t)c * n_chunks, tid, fold_threads(NT));
    __syncthreads();
    if (tid >= WAVE) return;   // single wavefront from here on
    const int lane = tid;
    const int64_t e = (int64_t)c * n_out + o;
    if (want_v & 2) {   // diagnostics: fold only (timing experiments)
        if (lane == 0) { var[e] = lds.phi[0]; status[e] = 0; }
        return;
    }
    const double am = (lane < N) ? lds.amax[lane] : 0.0;
    const bool big = __ballot(am >= 0.05) != 0ull;   // max |m| >= 0.05 (misc.py:464)
    const int reps = (want_v & 4) ? 2 : 1;   // diagnostics: run the solve twice
    for (int rep = 0; rep < reps; rep++)
        solve_wave<NT>(lds, N, delta, am > 1.0e-6, am > 0.0, big, (want_v & 1) != 0, var + e, v + e * N, status + e, lane);
}

// multi-GPU path, phase A tail: fold chunk partials into an all-reduce-able record.
template <int NT>
__global__ __launch_bounds__(fold_threads(NT)) void k_fold_to_record(int N, int n_out, const RowDesc *__restrict__ rows, int nsym,
                                                        const double2 *__restrict__ partial, int64_t n_chunks,
                                                        double *__restrict__ rec)
{
    __shared__ SolveLds<NT> lds;
    const int o = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    if (tid < N) lds.amax[tid] = 0.0;
    for (int t = tid; t < N * N; t += fold_threads(NT)) lds.phi[t] = 0.0;
    __syncthreads();
    fold_rows(lds, N, rows, o * nsym, nsym, partial + (int64_t)c * n_chunks, tid, fold_threads(NT));
    __syncthreads();
    const int reclen = N * N + 2 * N + 1;
    double *r = rec + ((int64_t)c * n_out + o) * reclen;
    for (int t = tid; t < N * N; t += fold_threads(NT)) r[t] = lds.phi[t];
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
    for (int t = lane; t < N * N; t += WAVE) lds.phi[t] = r[t];
    __syncthreads();
    const bool s1 = lane < N && r[N * N + lane] > 0.0;
    const bool s2 = lane < N && r[N * N + N + lane] > 0.0;
    const bool big = r[N * N + 2 * N] > 0.0;
    const int64_t e = (int64_t)c * n_out + o;
    solve_wave<NT>(lds, N, delta, s1, s2, big, (want_v & 1) != 0, var + e, v + e * N, status + e, lane);
}

// integer projection (misc.py:293-294, 368-369): for a batch of integer candidates that differ from a base allocation
// in LL entries, Phi_cand = Phi_base + sum_j ms[cand][j] * psi[:, idx_j], V = pinv(Phi_cand)[0,0].
// One wavefront per (candidate, output); psi columns (LL x N*N per output) are L2-resident.
template <int NT>
__global__ __launch_bounds__(64) void k_intproj(int N, int n_out, int LL, const double *__restrict__ base,
                                                const double *__restrict__ cols, const double *__restrict__ ms,
                                                int64_t n_cand, double *__restrict__ V)
{
    __shared__ SolveLds<NT> lds;
    __shared__ double sms[32];
    const int64_t cand = blockIdx.x;
    const int o = blockIdx.y, lane = threadIdx.x;
    if (lane < LL) sms[lane] = ms[cand * LL + lane];
    __syncthreads();
    const int NN = N * N;
    const double *b = base + (int64_t)o * NN;
    const double *cl = cols + (int64_t)o * LL * NN;
    for (int t = lane; t < NN; t += WAVE) {
        double x = b[t];
        for (int j = 0; j < LL; j++) x = fma(sms[j], cl[(int64_t)j * NN + t], x);
        lds.phi[t] = x;
    }
    __syncthreads();
    const bool sup = lane < N && lds.phi[lane * N + lane] > 0.0;
    double var = 0.0, vdummy = 0.0;
    int32_t status = 0;
    solve_wave<NT>(lds, N, 0.0, sup, sup, true, false, &var, &vdummy, &status, lane);
    if (lane == 0) V[cand * n_out + o] = (status == BLUEST_EVAL_OK) ? var : INFINITY;
}

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

// gradient pass: one wavefront per tile, lane = group.  q = sum_j v_j (s_jj v_j + 2 sum_{l>j} s_jl v_l).
template <int K>
__device__ __forceinline__ void grad_tile(const TileDesc &td, const double *__restrict__ tvals,
                                          const uint8_t *__restrict__ tidx, const double *__restrict__ v,
                                          const int32_t *__restrict__ status, int N, int n_out, int n_cand,
                                          double *__restrict__ grad, int64_t grad_stride, int lane)
{
    const double *vals = tvals + td.val_off + lane;
    const uint8_t *idx = tidx + td.idx_off + lane;
    int gi[K];
#pragma unroll
    for (int j = 0; j < K; j++) gi[j] = idx[j * 64];
    double s[K * (K + 1) / 2];
#pragma unroll
    for (int e = 0; e < K * (K + 1) / 2; e++) s[e] = vals[e * 64];
    for (int c = 0; c < n_cand; c++) {
        const int64_t eo = (int64_t)c * n_out + td.out;
        const double *vc = v + eo * N;
        double vj[K];
#pragma unroll
        for (int j = 0; j < K; j++) vj[j] = vc[gi[j]];
        double q = 0.0;
        int e = 0;
#pragma unroll
        for (int j = 0; j < K; j++) {
            double t = 0.0;
#pragma unroll
            for (int l = j + 1; l < K; l++) t = fma(s[e + (l - j)], vj[l], t);
            t = fma(s[e], vj[j], 2.0 * t);
            q = fma(vj[j], t, q);
            e += K - j;
        }
        if (lane < (td.n_valid & 0xffff))
            grad[(int64_t)c * grad_stride + td.grad_off + lane] = (status[eo] == BLUEST_EVAL_INF) ? INFINITY : -q;
    }
}

// the quadratic form of one group from a tile already in registers: q = v_g^T S v_g (packed symmetric S, K static)
template <int K, int NE, int KU>
__device__ __forceinline__ double tile_form(const double (&s)[NE], const int (&gi)[KU], const double *__restrict__ vc)
{
    double vj[K];
#pragma unroll
    for (int j = 0; j < K; j++) vj[j] = vc[gi[j]];
    double q = 0.0;
    int e = 0;
#pragma unroll
    for (int j = 0; j < K; j++) {
        double t = 0.0;
#pragma unroll
        for (int l = j + 1; l < K; l++) t = fma(s[e + (l - j)], vj[l], t);
        t = fma(s[e], vj[j], 2.0 * t);
        q = fma(vj[j], t, q);
        e += K - j;
    }
    return q;
}

// generic k (13..16): entries re-read per candidate, no big register arrays
__device__ __forceinline__ void grad_tile_generic(const TileDesc &td, const double *__restrict__ tvals,
                                                  const uint8_t *__restrict__ tidx, const double *__restrict__ v,
                                                  const int32_t *__restrict__ status, int N, int n_out, int n_cand,
                                                  double *__restrict__ grad, int64_t grad_stride, int lane)
{
    const int K = td.k;
    const double *vals = tvals + td.val_off + lane;
    const uint8_t *idx = tidx + td.idx_off + lane;
    for (int c = 0; c < n_cand; c++) {
        const int64_t eo = (int64_t)c * n_out + td.out;
        const double *vc = v + eo * N;
        double q = 0.0;
        int e = 0;
        for (int j = 0; j < K; j++) {
            const double vjj = vc[idx[j * 64]];
            double t = 0.0;
            for (int l = j + 1; l < K; l++) t = fma(vals[(e + (l - j)) * 64], vc[idx[l * 64]], t);
            t = fma(vals[e * 64], vjj, 2.0 * t);
            q = fma(vjj, t, q);
            e += K - j;
        }
        if (lane < (td.n_valid & 0xffff))
            grad[(int64_t)c * grad_stride + td.grad_off + lane] = (status[eo] == BLUEST_EVAL_INF) ? INFINITY : -q;
    }
}

// KU = largest group size with a fully unrolled register path in this instantiation (the host picks the smallest
// KU covering the plan, so the common small-k plans keep a small register footprint)
template <int KU>
__global__ __launch_bounds__(256) void k_grad_tiles(const TileDesc *__restrict__ tiles, int64_t n_tiles,
                                                    const double *__restrict__ tvals,
                                                    const uint8_t *__restrict__ tidx, const double *__restrict__ v,
                                                    const int32_t *__restrict__ status, int N, int n_out, int n_cand,
                                                    double *__restrict__ grad, int64_t grad_stride,
                                                    const int32_t *__restrict__ gate)
{
    if (gate && *gate == 0) return;   // device-side predication (SPG line-search slots)
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const TileDesc td = tiles[t];
#define GT(KK) case KK: if (KK <= KU) { grad_tile<(KK <= KU ? KK : 1)>(td, tvals, tidx, v, status, N, n_out, n_cand, grad, grad_stride, lane); break; }
    switch (td.k) {
        GT(1) GT(2) GT(3) GT(4) GT(5) GT(6) GT(7) GT(8) GT(9) GT(10) GT(11) GT(12)
        default: grad_tile_generic(td, tvals, tidx, v, status, N, n_out, n_cand, grad, grad_stride, lane);
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
static int pick_nt(int N) { return N <= 8 ? 8 : N <= 12 ? 12 : N <= 16 ? 16 : N <= 20 ? 20 : N <= 26 ? 26 : N <= 32 ? 32 : N <= 48 ? 48 : 64; }
static int pick_ku(int kmax) { return kmax <= 5 ? 5 : kmax <= 6 ? 6 : kmax <= 8 ? 8 : 12; }
template <int NT, int KU>
__global__ __launch_bounds__(64 * (fused_tpb(NT, KU) + 1)) void k_solve_grad(int N, int n_out, const RowDesc *__restrict__ rows, int nsym,
                                                    const double2 *__restrict__ partial, double delta,
                                                    const TileDesc *__restrict__ tiles, int64_t n_tiles, int bpo,
                                                    const double *__restrict__ tvals, const uint8_t *__restrict__ tidx,
                                                    double *__restrict__ var, double *__restrict__ v_ws,
                                                    int32_t *__restrict__ status, double *__restrict__ grad,
                                                    const int32_t *__restrict__ gate)
{
    constexpr int FUSED_TPB = fused_tpb(NT, KU);
    constexpr int NTHREADS = 64 * (FUSED_TPB + 1);
    constexpr int NE = KU * (KU + 1) / 2;
    __shared__ SolveLds<NT> lds;
    if (gate && *gate == 0) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    SPAN_BEGIN(1);
    PHASE(0);
    const int64_t t0 = (int64_t)blockIdx.x * FUSED_TPB;
    // which output, and am I its first workgroup: arithmetic when every output has the same number of workgroups (bpo > 0,
    // the usual case), else from the first tile's descriptor (one more dependent load in front of the fold)
    int o, first;
    if (bpo > 0) { o = blockIdx.x / bpo; first = (blockIdx.x % bpo) == 0; }
    else { const TileDesc td0 = tiles[t0]; o = td0.out; first = (td0.n_valid >> 30) & 1; }
    if (tid < N) lds.amax[tid] = 0.0;
    for (int t = tid; t < N * N; t += NTHREADS) lds.phi[t] = 0.0;
    __syncthreads();
    PHASE(1);
    fold_rows<NT>(lds, N, rows, o * nsym, nsym, partial, tid, NTHREADS);
    __syncthreads();
    PHASE(2);
    // the list is padded to a multiple of FUSED_TPB tiles per output, so every tile of this workgroup belongs to output o
    const TileDesc td = tiles[t0 + (wave > 0 ? wave - 1 : 0)];
    const int k = td.k;
    double s[NE];
    int gi[KU];
    if (wave == 0) {
        const double am = (lane < N) ? lds.amax[lane] : 0.0;
        const bool big = __ballot(am >= 0.05) != 0ull;   // max |m| >= 0.05 (misc.py:464)
        double V = 0.0;
        int32_t st = 0;
        PHASE(8);
        solve_wave<NT>(lds, N, delta, am > 1.0e-6, am > 0.0, big, true, &V, lds.vout, &st, lane);
        if (lane == 0) lds.status = st;
        if (first) {   // first workgroup of this output publishes V, status, v
            if (lane == 0) { var[o] = V; status[o] = st; }
            if (lane < N) v_ws[(int64_t)o * N + lane] = lds.vout[lane];
        }
        PHASE(3);
    } else if (k <= KU) {
        // stream the tile into registers while wavefront 0 factorises (after the fold, so these loads do not queue in front of it)
        const double *vals = tvals + td.val_off + lane;
        const uint8_t *idx = tidx + td.idx_off + lane;
        const int ne = k * (k + 1) / 2;
#pragma unroll
        for (int j = 0; j < KU; j++) if (j < k) gi[j] = idx[j * 64];
#pragma unroll
        for (int e = 0; e < NE; e++) if (e < ne) s[e] = vals[e * 64];
    }
    __syncthreads();
    if (wave == 0) { SPAN_END(1, true); return; }
    const bool valid = lane < (td.n_valid & 0xffff);
    const bool inf = lds.status == BLUEST_EVAL_INF;
    double *gout = grad + td.grad_off + lane;
#define GT(KK) case KK: if (KK <= KU) { const double q = tile_form<(KK <= KU ? KK : 1)>(s, gi, lds.vout); if (valid) *gout = inf ? INFINITY : -q; break; }
    switch (k) {
        GT(1) GT(2) GT(3) GT(4) GT(5) GT(6) GT(7) GT(8) GT(9) GT(10) GT(11) GT(12)
        default: {
            // grad_tile reads v as v[(c*n_out + td.out)*N + model] and status[c*n_out + td.out]: point both at LDS
            const double *vl = lds.vout - (int64_t)td.out * N;
            const int32_t *sl = &lds.status - td.out;
            grad_tile_generic(td, tvals, tidx, vl, sl, N, n_out, 1, grad, 0, lane);
        }
    }
#undef GT
    PHASE(4);
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
// Part 3 -- simplex projection (single workgroup of 1024 threads)
// ------------------------------------------------------------------------------------------------------
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
#define SPG_FAIL     9    // 1 if all slots of an iteration were rejected (host continues the line search)
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
#define SPG_GPSTATS  24   // g.gp, max|gp| (= gpmax), tau, npos of the convergence projection
#define SPG_HIST     32   // 16 slots
#define SPG_COEF     64   // dF/dV_o of the accepted trial (n_out <= 64)
#define SPG_S        128  // normalisers s_o (1 or eps_o^2)
#define SPG_STATE_DOUBLES 256
#define SPG_MAX_OUT  64

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
    const int lane = threadIdx.x;
#pragma unroll
    for (int t = 0; t < SPG_STATE_DOUBLES / 64; t++) ls[t * 64 + lane] = st[t * 64 + lane];
    __syncthreads();
    const bool idle = ls[SPG_DONE] != 0.0 || ls[SPG_FAIL] != 0.0;
    if (idle || ls[SPG_ACCEPT] != 0.0) {
        if (last_slot && lane == 0) *enable = (!idle && ls[SPG_ACCEPT] != 0.0) ? 1 : 0;
        return;
    }
    // objective F = || (V_o/s_o) ||_p / norm, coefficients dF/dV_o
    const bool mine = lane < n_out;
    const double so = mine ? ls[SPG_S + lane] : 1.0;
    const double r = mine ? var[lane] / so : 0.0;
    const bool bad = mine && (status[lane] != BLUEST_EVAL_OK || !isfinite(r));
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
    const double alpha = ls[SPG_ALPHA], gd = ls[SPG_GD], f = ls[SPG_F];
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
            if (last_slot) st[SPG_FAIL] = 1.0;
        }
        if (last_slot) *enable = accept ? 1 : 0;
    }
}

// Workgroup reductions for the projection / SPG kernels: wavefront butterflies, one LDS hand-off and ONE barrier per call.
// The LDS slots are double-buffered by the caller-held phase bit, so a wavefront that is already in the next reduction
// cannot overwrite values a slower wavefront is still reading (it would first have to pass the barrier in between).
// Every wavefront folds the <= 16 per-wavefront partials itself with the same butterfly: identical results everywhere.
struct ProjLds {
    double d0[2][16];
    double d1[2][16];
    long long c[2][16];
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

// accept the step (bluest/spg.py:85-106), two launches:
//  A (multi-block): s = xnew - x, y = gnew - g, per-block partial sums of s^T D^-1 s (D = diag(max(x,floor))) and s.y,
//                   x <- xnew, g <- gnew;
//  B (one wavefront): fixed-order sum of the partials, Barzilai-Borwein lambda, history, reset of the line-search state.
#define SPG_UPD_BLOCKS_MAX 512
__global__ __launch_bounds__(1024) void k_spg_update_a(double *__restrict__ x, double *__restrict__ g,
                                                       const double *__restrict__ xnew, const double *__restrict__ gnew,
                                                       const double *__restrict__ st, double floor, int64_t L,
                                                       double2 *__restrict__ partial)
{
    __shared__ ProjLds sm;
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
    if (tid == 0) partial[blockIdx.x] = make_double2(sdots, sdoty);
}

// A with the gradient fold fused in: gnew_j = scale_j * sum_o coef_o * grad_o[local_o(j)] is formed on the fly (no gnew
// vector, one launch less per iteration).
__global__ __launch_bounds__(1024) void k_spg_update_a_fused(double *__restrict__ x, double *__restrict__ g,
                                                             const double *__restrict__ xnew, const double *__restrict__ grad,
                                                             const int64_t *__restrict__ goff, const int32_t *__restrict__ invmap,
                                                             int n_out, const double *__restrict__ scale,
                                                             const double *__restrict__ st, double floor, int64_t L,
                                                             double2 *__restrict__ partial)
{
    __shared__ ProjLds sm;
    int ph = 0;
    const int tid = threadIdx.x;
    if (st[SPG_DONE] != 0.0 || st[SPG_FAIL] != 0.0 || st[SPG_ACCEPT] == 0.0) return;
    double sdots = 0.0, sdoty = 0.0;
    long long dummy = 0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + tid; i < L; i += (int64_t)gridDim.x * 1024) {
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
    if (tid == 0) partial[blockIdx.x] = make_double2(sdots, sdoty);
}

__global__ __launch_bounds__(64) void k_spg_update_b(double *__restrict__ st, const double2 *__restrict__ partial, int nblocks)
{
    __shared__ double ls[64];
    const int lane = threadIdx.x;
    ls[lane] = st[lane];          // scalars live in st[0..63]
    __syncthreads();
    if (ls[SPG_DONE] != 0.0 || ls[SPG_FAIL] != 0.0 || ls[SPG_ACCEPT] == 0.0) return;
    double a = 0.0, b = 0.0;
    for (int t = lane; t < nblocks; t += 64) { const double2 q = partial[t]; a += q.x; b += q.y; }
    const double sdots = wave_sum(a), sdoty = wave_sum(b);
    if (lane == 0) {
        st[SPG_SDOTS] = sdots;
        st[SPG_SDOTY] = sdoty;
        const double lmin = ls[SPG_LMIN], lmax = ls[SPG_LMAX];
        st[SPG_LAMBDA] = (sdoty <= 0.0) ? lmax : fmin(lmax, fmax(lmin, sdots / sdoty));
        const double it = ls[SPG_IT] + 1.0;
        st[SPG_IT] = it;
        st[SPG_F] = ls[SPG_FNEW];
        const int H = (int)ls[SPG_HLEN];
        st[SPG_HIST + ((long long)it % H)] = ls[SPG_FNEW];
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
        if (spg_mode == 1) lambda = spg_state[SPG_LAMBDA];   // direction: the step length lives in HBM
    }
    constexpr int R = ITEMS > 0 ? ITEMS : 1;
    constexpr int B = SIMPLEX_BLOCK;
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
    double tau = (floor > 0.0) ? -z / floor : -z;
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
    // a mailbox = 8 x 64-bit words = four doubles, each split into two (tag << 32 | 32 payload bits) words
    // doubles: [0, 2*MAXB*8) double-buffered pass mailboxes   [.., +MAXB*8) final-statistics mailboxes
    static constexpr int PART = 0, FIN = 2 * MAXB * 8, DOUBLES = FIN + MAXB * 8;
};
struct ProjWs {            // layout of the caller-provided workspace (doubles)
    // [0, 2L): interleaved (r_i, s_i) pairs
    static __host__ __device__ int64_t part_off(int64_t L) { return 2 * L; }            // 4 doubles per block
    static __host__ __device__ int64_t tau_off(int64_t L, int nb) { return 2 * L + 4LL * nb; }   // tau, rmax
    static __host__ __device__ int64_t sync_off(int64_t L, int nb) { return 2 * L + 4LL * nb + 16; }   // single-launch path
    static __host__ __device__ int64_t total(int64_t L, int nb) { return sync_off(L, nb) + FusedProj::DOUBLES; }
};

__device__ __forceinline__ bool proj_idle(const double *spg_state) { return spg_state && (spg_state[SPG_DONE] != 0.0 || spg_state[SPG_FAIL] != 0.0); }

__global__ __launch_bounds__(1024) void k_proj_a(const double *__restrict__ x, const double *__restrict__ g, double lambda,
                                                 double floor, int64_t L, double *__restrict__ ws, int nb,
                                                 const double *__restrict__ spg_state, int spg_mode)
{
    __shared__ ProjLds sm;
    int ph = 0;
    if (proj_idle(spg_state)) return;
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
    if (proj_idle(spg_state)) return;
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
    if (proj_idle(spg_state)) return;
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

__global__ __launch_bounds__(1024) void k_proj_p(int64_t L, double *__restrict__ ws, int nb, const double *__restrict__ spg_state)
{
    __shared__ ProjLds sm;
    int ph = 0;
    if (proj_idle(spg_state)) return;
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
    if (proj_idle(spg_state)) return;
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
    if (proj_idle(spg_state)) return;
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
                                                 double *__restrict__ mtrial, const double *__restrict__ spg_state)
{
    __shared__ ProjLds sm;
    int ph = 0;
    if (proj_idle(spg_state)) return;
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
    double gd = 0.0, dm = 0.0, np = 0.0;
    for (int b = lane; b < nb; b += 64) {
        const double *pp = ws + ProjWs::part_off(L) + 4LL * b;
        gd += pp[1]; dm = fmax(dm, pp[2]); np += pp[3];
    }
    gd = wave_sum(gd); dm = wave_max(dm); np = wave_sum(np);
    if (lane == 0) {
        if (stats) { stats[0] = gd; stats[1] = dm; stats[2] = ws[ProjWs::tau_off(L, nb)]; stats[3] = np; }
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

template <int ITEMS>
__global__ __launch_bounds__(1024) void k_proj_fused(const double *__restrict__ x, const double *__restrict__ g, double lambda,
                                                     double z, double floor, int64_t L, double *__restrict__ ws, int nb_ws,
                                                     double *__restrict__ p, double *__restrict__ d,
                                                     double *__restrict__ stats, double *__restrict__ spg_state, int spg_mode,
                                                     const double *__restrict__ scale, double *__restrict__ xnew,
                                                     double *__restrict__ mtrial, int32_t *__restrict__ enable, int maxp)
{
    __shared__ ProjLds sm;
    int ph = 0;
    const int tid = threadIdx.x, nb = gridDim.x, b = blockIdx.x;
    double *t = ws + ProjWs::tau_off(L, nb_ws);
    double *sy = ws + ProjWs::sync_off(L, nb_ws);
    if (proj_idle(spg_state) || t[12] != 0.0) {   // t[12]: sticky "a wait timed out" flag
        if (enable && b == 0 && tid == 0) *enable = 0;
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
        const int64_t i = ((int64_t)k * nb + b) * 1024 + tid;
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
    long long prev = -1;
    int passes = 0;
    rmax = mb;
    for (int iter = 0; iter < maxp; iter++) {
        double s1 = 0.0, s0 = 0.0;
        long long cnt = 0;
        passes++;
        const double ref = first ? mb : 0.0;                 // later passes: r[] is already shifted
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const bool act = r[k] > tau;
            s1 = act ? fma(sc[k], r[k] - ref, s1) : s1;
            s0 = act ? s0 + sc[k] : s0;
            cnt += act ? 1 : 0;
        }
        block_sum2_cnt(s1, s0, cnt, sm, tid, ph);
        const unsigned int tag = epoch + 1u + (unsigned int)iter;
        if (tid < 8) mailbox_send(sy + FusedProj::PART + ((iter & 1) * FusedProj::MAXB + b) * 8, tag, s1, s0, (double)cnt, mb);
        double msg[4];
        ok = mailbox_recv(sy + FusedProj::PART + ((iter & 1) * FusedProj::MAXB + (tid < nb ? tid : 0)) * 8, tag, tid < nb, msg);
        if (!ok) break;
        s1 = msg[0]; s0 = msg[1]; cnt = (long long)msg[2];
        if (first) {
            const double m = (tid < nb) ? msg[3] : -INFINITY;
            rmax = block_max(m, sm, tid, ph);
            if (cnt > 0) s1 = fma(s0, m - rmax, s1);         // re-base this workgroup's partial to the grid-wide max
        }
        block_sum2_cnt(s1, s0, cnt, sm, tid, ph);
        if (first) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) r[k] -= rmax;
            if (cnt == 0) {                                  // hint right of every r_i (or no entries): all-active restart
                if (tau == -INFINITY) break;
                tau = -INFINITY;
                // r[] is shifted now, so the restart is an ordinary later pass
                first = false;
                continue;
            }
            first = false;
        } else if (cnt == prev || cnt == 0) {
            break;
        }
        prev = cnt;
        tau = (s1 - z) / s0;
    }
    if (!ok) tau = NAN;

    // ---- C: p, d, the fused first trial point; partials of g.d, max|d|, #positive ----
    double gd = 0.0, dm = 0.0;
    long long npos = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const int64_t i = ((int64_t)k * nb + b) * 1024 + tid;
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
// Part 1 host entry points
// ------------------------------------------------------------------------------------------------------
static int check_cmisc_args(int N, int k, int64_t Lk)
{
    if (N <= 0 || N > 4096) return fail(BLUEST_ERR_ARG, "N=%d out of range", N);
    if (k <= 0 || k > 64) return fail(BLUEST_ERR_ARG, "k=%d out of range", k);
    if (Lk < 0) return fail(BLUEST_ERR_ARG, "Lk=%lld negative", (long long)Lk);
    return BLUEST_OK;
}

extern "C" int bluest_assemble_psi(double *psi, int N, int k, int64_t Lk, const int64_t *g, const double *ic)
{
    int rc = require_gpu(); if (rc) return rc;
    rc = check_cmisc_args(N, k, Lk); if (rc) return rc;
    if (Lk == 0) return BLUEST_OK;
    if (!psi || !g || !ic) return fail(BLUEST_ERR_ARG, "null pointer");
    Staged<double> s_psi, s_ic; Staged<int64_t> s_g;
    if ((rc = s_psi.init(psi, (size_t)N * N * Lk, true))) return rc;
    if ((rc = s_g.init(g, (size_t)Lk * k, true))) return rc;
    if ((rc = s_ic.init(ic, (size_t)Lk * k * k, true))) return rc;
    hipLaunchKernelGGL(k_assemble_psi, dim3((unsigned)((Lk + 255) / 256)), dim3(256), 0, 0, s_psi.dev, N, k, Lk, s_g.dev, s_ic.dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return s_psi.finish(true);
}

template <typename MT>
static int objectiveK_impl(double *PHI, int N, int k, int64_t Lk, const MT *mk, const int64_t *g, const double *ic)
{
    int rc = require_gpu(); if (rc) return rc;
    rc = check_cmisc_args(N, k, Lk); if (rc) return rc;
    if (N > 128) return fail(BLUEST_ERR_ARG, "N=%d > 128 unsupported by the LDS-privatised Phi kernel", N);
    if (Lk == 0) return BLUEST_OK;
    if (!PHI || !mk || !g || !ic) return fail(BLUEST_ERR_ARG, "null pointer");
    Staged<double> s_phi, s_ic; Staged<int64_t> s_g; Staged<MT> s_m;
    if ((rc = s_phi.init(PHI, (size_t)N * N, true))) return rc;
    if ((rc = s_m.init(mk, (size_t)Lk, true))) return rc;
    if ((rc = s_g.init(g, (size_t)Lk * k, true))) return rc;
    if ((rc = s_ic.init(ic, (size_t)Lk * k * k, true))) return rc;
    const int NN = N * N;
    int nblocks = (int)std::min<int64_t>((Lk + 255) / 256, 1024);
    double *slabs = nullptr;
    HIP_TRY(hipMalloc((void **)&slabs, (size_t)nblocks * NN * sizeof(double)));
    hipLaunchKernelGGL((k_objectiveK<MT>), dim3(nblocks), dim3(256), NN * sizeof(double), 0, slabs, N, k, Lk, s_m.dev, s_g.dev, s_ic.dev);
    hipLaunchKernelGGL(k_fold_slabs, dim3((NN + 255) / 256), dim3(256), 0, 0, s_phi.dev, slabs, NN, nblocks);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(slabs);
    HIP_TRY(e);
    return s_phi.finish(true);
}

extern "C" int bluest_objectiveK_f64(double *PHI, int N, int k, int64_t Lk, const double *mk, const int64_t *g, const double *ic)
{ return objectiveK_impl<double>(PHI, N, k, Lk, mk, g, ic); }
extern "C" int bluest_objectiveK_i64(double *PHI, int N, int k, int64_t Lk, const int64_t *mk, const int64_t *g, const double *ic)
{ return objectiveK_impl<int64_t>(PHI, N, k, Lk, mk, g, ic); }

extern "C" int bluest_gradK(double *grad, int k, int64_t Lk, const int64_t *g, const double *ic, const double *v, int n_models)
{
    int rc = require_gpu(); if (rc) return rc;
    rc = check_cmisc_args(n_models, k, Lk); if (rc) return rc;
    if (n_models > BLUEST_MAX_MODELS * 4) return fail(BLUEST_ERR_ARG, "n_models=%d > %d", n_models, BLUEST_MAX_MODELS * 4);
    if (Lk == 0) return BLUEST_OK;
    if (!grad || !g || !ic || !v) return fail(BLUEST_ERR_ARG, "null pointer");
    Staged<double> s_grad, s_ic, s_v; Staged<int64_t> s_g;
    if ((rc = s_grad.init(grad, (size_t)Lk, true))) return rc;
    if ((rc = s_g.init(g, (size_t)Lk * k, true))) return rc;
    if ((rc = s_ic.init(ic, (size_t)Lk * k * k, true))) return rc;
    if ((rc = s_v.init(v, (size_t)n_models, true))) return rc;
    hipLaunchKernelGGL(k_gradK, dim3((unsigned)((Lk + 255) / 256)), dim3(256), 0, 0, s_grad.dev, k, Lk, s_g.dev, s_ic.dev, s_v.dev, n_models);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return s_grad.finish(true);
}

extern "C" int bluest_cleanupK(double *X, int k, int64_t Lk, const int64_t *g, const double *ic, const double *v, int n_models)
{
    int rc = require_gpu(); if (rc) return rc;
    rc = check_cmisc_args(n_models, k, Lk); if (rc) return rc;
    if (Lk == 0) return BLUEST_OK;
    if (!X || !g || !ic || !v) return fail(BLUEST_ERR_ARG, "null pointer");
    Staged<double> s_X, s_ic, s_v; Staged<int64_t> s_g;
    if ((rc = s_X.init(X, (size_t)n_models * Lk, true))) return rc;
    if ((rc = s_g.init(g, (size_t)Lk * k, true))) return rc;
    if ((rc = s_ic.init(ic, (size_t)Lk * k * k, true))) return rc;
    if ((rc = s_v.init(v, (size_t)n_models, true))) return rc;
    hipLaunchKernelGGL(k_cleanupK, dim3((unsigned)((Lk + 255) / 256)), dim3(256), 0, 0, s_X.dev, k, Lk, s_g.dev, s_ic.dev, s_v.dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return s_X.finish(true);
}

extern "C" int bluest_hessKQ(double *hess, int N, int k, int q, int64_t Lk, int64_t Lq, const int64_t *gk, const int64_t *gq,
                             const double *ick, const double *icq, const double *invPHI)
{
    int rc = require_gpu(); if (rc) return rc;
    rc = check_cmisc_args(N, k, Lk); if (rc) return rc;
    rc = check_cmisc_args(N, q, Lq); if (rc) return rc;
    if (Lk == 0 || Lq == 0) return BLUEST_OK;
    if (Lk > 65535) return fail(BLUEST_ERR_ARG, "Lk=%lld > 65535: the (Lk,Lq) Hessian block is not meant for this size", (long long)Lk);
    if (!hess || !gk || !gq || !ick || !icq || !invPHI) return fail(BLUEST_ERR_ARG, "null pointer");
    Staged<double> s_h, s_ick, s_icq, s_P; Staged<int64_t> s_gk, s_gq;
    if ((rc = s_h.init(hess, (size_t)Lk * Lq, true))) return rc;
    if ((rc = s_gk.init(gk, (size_t)Lk * k, true))) return rc;
    if ((rc = s_gq.init(gq, (size_t)Lq * q, true))) return rc;
    if ((rc = s_ick.init(ick, (size_t)Lk * k * k, true))) return rc;
    if ((rc = s_icq.init(icq, (size_t)Lq * q * q, true))) return rc;
    if ((rc = s_P.init(invPHI, (size_t)N * N, true))) return rc;
    double *ak = nullptr, *aq = nullptr;
    HIP_TRY(hipMalloc((void **)&ak, (size_t)Lk * k * sizeof(double)));
    hipError_t e = hipMalloc((void **)&aq, (size_t)Lq * q * sizeof(double));
    if (e != hipSuccess) { (void)hipFree(ak); HIP_TRY(e); }
    hipLaunchKernelGGL(k_hess_avec, dim3((unsigned)((Lk + 255) / 256)), dim3(256), 0, 0, ak, k, Lk, s_gk.dev, s_ick.dev, s_P.dev);
    hipLaunchKernelGGL(k_hess_avec, dim3((unsigned)((Lq + 255) / 256)), dim3(256), 0, 0, aq, q, Lq, s_gq.dev, s_icq.dev, s_P.dev);
    hipLaunchKernelGGL(k_hessKQ, dim3((unsigned)((Lq + 127) / 128), (unsigned)Lk), dim3(128), 0, 0, s_h.dev, N, k, q, Lk, Lq,
                       s_gk.dev, s_gq.dev, ak, aq, s_P.dev);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(ak); (void)hipFree(aq);
    HIP_TRY(e);
    return s_h.finish(true);
}

static int launch_group_pinv(const double *dC, int N, int k, int64_t Lk, const int64_t *dg, double *dout, hipStream_t st)
{
    const dim3 grid((unsigned)((Lk + 63) / 64)), block(64);
#define GP(KK) case KK: hipLaunchKernelGGL((k_group_pinv<KK>), grid, block, 0, st, dC, N, Lk, dg, dout); break;
    switch (k) {
        GP(1) GP(2) GP(3) GP(4) GP(5) GP(6) GP(7) GP(8) GP(9) GP(10) GP(11) GP(12) GP(13) GP(14) GP(15) GP(16)
        default: return fail(BLUEST_ERR_ARG, "group size k=%d > %d", k, BLUEST_MAX_GROUP);
    }
#undef GP
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_group_pinv(const double *C, int N, int k, int64_t Lk, const int64_t *g, double *out)
{
    int rc = require_gpu(); if (rc) return rc;
    rc = check_cmisc_args(N, k, Lk); if (rc) return rc;
    if (Lk == 0) return BLUEST_OK;
    if (!C || !g || !out) return fail(BLUEST_ERR_ARG, "null pointer");
    Staged<double> s_C, s_out; Staged<int64_t> s_g;
    if ((rc = s_C.init(C, (size_t)N * N, true))) return rc;
    if ((rc = s_g.init(g, (size_t)Lk * k, true))) return rc;
    if ((rc = s_out.init(out, (size_t)Lk * k * k, false))) return rc;
    if ((rc = launch_group_pinv(s_C.dev, N, k, Lk, s_g.dev, s_out.dev, 0))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return s_out.finish(true);
}

// ------------------------------------------------------------------------------------------------------
// Part 2 host side: the plan
// ------------------------------------------------------------------------------------------------------
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
    int32_t *d_status = nullptr; // workspace for eval when caller passes NULL
};

static void plan_free_device(bluest_plan_s *p)
{
    void *ptrs[] = {p->d_vals, p->d_cols, p->d_rows, p->d_out_row_begin, p->d_tiles, p->d_tvals, p->d_tidx,
                    p->d_invmap, p->d_goff, p->d_partial, p->d_v, p->d_status};
    for (void *q : ptrs) if (q) (void)hipFree(q);
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
    *plan = p;
    return BLUEST_OK;
}

extern "C" int bluest_plan_destroy(bluest_plan_t plan)
{
    if (!plan) return BLUEST_OK;
    plan_free_device(plan);
    delete plan;
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

extern "C" int bluest_plan_add_output(bluest_plan_t plan, int K, const int64_t *sizes, const int64_t *groups,
                                      const double *invcovs, const int64_t *mapping)
{
    OutputDesc od;
    int rc = plan_add_common(plan, K, sizes, groups, mapping, od);
    if (rc) return rc;
    if (!invcovs) return fail(BLUEST_ERR_ARG, "invcovs is NULL");
    int64_t ni = 0;
    for (int k = 1; k <= K; k++) ni += sizes[k - 1] * k * k;
    od.invcovs.assign(invcovs, invcovs + ni);
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
    int64_t ni = 0, ng = 0;
    for (int k = 1; k <= K; k++) { ni += sizes[k - 1] * k * k; ng += sizes[k - 1] * k; }
    od.invcovs.resize(ni);
    double *dC = nullptr, *dic = nullptr;
    int64_t *dg = nullptr;
    HIP_TRY(hipMalloc((void **)&dC, (size_t)N * N * sizeof(double)));
    hipError_t e = hipMalloc((void **)&dic, (size_t)ni * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&dg, (size_t)ng * sizeof(int64_t));
    if (e == hipSuccess) e = hipMemcpy(dC, C, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dg, groups, (size_t)ng * sizeof(int64_t), hipMemcpyHostToDevice);
    rc = BLUEST_OK;
    if (e == hipSuccess) {
        int64_t go = 0, io = 0;
        for (int k = 1; k <= K && rc == BLUEST_OK; k++) {
            const int64_t Lk = sizes[k - 1];
            if (Lk > 0) rc = launch_group_pinv(dC, N, k, Lk, dg + go, dic + io, 0);
            go += Lk * k; io += Lk * k * k;
        }
        if (rc == BLUEST_OK) e = hipMemcpy(od.invcovs.data(), dic, (size_t)ni * sizeof(double), hipMemcpyDeviceToHost);
    }
    (void)hipFree(dC); (void)hipFree(dic); (void)hipFree(dg);
    timer.lap("device pinv round trip (malloc, H2D, kernels, D2H, free)");
    if (rc) return rc;
    HIP_TRY(e);
    if (invcovs_out) memcpy(invcovs_out, od.invcovs.data(), (size_t)ni * sizeof(double));
    plan->outs.push_back(std::move(od));
    return BLUEST_OK;
}

template <typename T>
static int upload(T **dst, const std::vector<T> &src)
{
    *dst = nullptr;
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void **)dst, bytes));
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

    // ---- Phi pass: destination-major symmetric CSR ------------------------------------------------
    // count entries per (output,row); pick the chunk size so that no row needs more than 64 chunks
    std::vector<std::vector<int64_t>> counts(n_out, std::vector<int64_t>(nsym, 0));
    int64_t max_row = 0;
    // parallel counting sort: slice s of S owns a contiguous range of the output's groups; it counts its entries per row,
    // the per-slice counts are prefix-summed into start offsets, then every slice writes its own entries -- rows keep their
    // entries in group order whatever S is, so the layout (and every summation order on the GPU) is independent of threading
    const int S = std::max(1, host_threads() / n_out);      // slices per output
    std::vector<std::vector<int64_t>> slice_cnt((size_t)n_out * S, std::vector<int64_t>(nsym, 0));
    auto for_groups_of_slice = [&](int o, int slice, auto &&body) {
        const OutputDesc &od = plan->outs[o];
        const int64_t lo = od.L_o * slice / S, hi = od.L_o * (slice + 1) / S;
        int64_t go = 0, io = 0, l0 = 0;
        for (int k = 1; k <= od.K; k++) {
            const int64_t Lk = od.sizes[k - 1];
            for (int64_t i = std::max<int64_t>(lo - l0, 0); i < std::min<int64_t>(hi - l0, Lk); i++)
                body(k, od.groups.data() + go + i * k, od.invcovs.data() + io + i * k * k, l0 + i);
            go += Lk * k; io += Lk * k * k; l0 += Lk;
        }
    };
    parallel_items(n_out * S, [&](int item) {
        std::vector<int64_t> &cnt = slice_cnt[item];
        for_groups_of_slice(item / S, item % S, [&](int k, const int64_t *g, const double *, int64_t) {
            for (int j = 0; j < k; j++)
                for (int l = j; l < k; l++) cnt[tri((int)std::min(g[j], g[l]), (int)std::max(g[j], g[l]))]++;
        });
    });
    for (int o = 0; o < n_out; o++)
        for (int r = 0; r < nsym; r++) {
            int64_t run = 0;
            for (int sl = 0; sl < S; sl++) { const int64_t c = slice_cnt[(size_t)o * S + sl][r]; slice_cnt[(size_t)o * S + sl][r] = run; run += c; }
            counts[o][r] = run;      // slice_cnt now holds each slice's start offset inside the row
        }
    for (int o = 0; o < n_out; o++)
        for (int r = 0; r < nsym; r++) max_row = std::max(max_row, counts[o][r]);
    timer.lap("count");
    int iters = 1;
    while ((max_row + 256LL * iters - 1) / (256LL * iters) > 64 && iters < 1024) iters *= 2;
    const int64_t CH = 256LL * iters;
    plan->iters = iters;
    plan->nsym = nsym;
    plan->shared = true;
    for (int o = 1; o < n_out; o++) {
        const OutputDesc &x = plan->outs[0], &y = plan->outs[o];
        if (x.K != y.K || x.sizes != y.sizes || x.groups != y.groups || x.mapping != y.mapping) { plan->shared = false; break; }
    }

    std::vector<RowDesc> rows((size_t)n_out * nsym);
    std::vector<int32_t> out_row_begin(n_out + 1);
    int64_t n_chunks = 0;
    for (int o = 0; o < n_out; o++) {
        out_row_begin[o] = o * nsym;
        for (int a = 0; a < N; a++)
            for (int b = a; b < N; b++) {
                RowDesc &rd = rows[(size_t)o * nsym + tri(a, b)];
                const int64_t nc = (counts[o][tri(a, b)] + CH - 1) / CH;
                rd.first_chunk = (int32_t)n_chunks;
                rd.n_chunks = (int32_t)nc;
                rd.out = (int16_t)o; rd.a = (int16_t)a; rd.b = (int16_t)b; rd.pad = 0;
                n_chunks += nc;
            }
    }
    out_row_begin[n_out] = n_out * nsym;
    if (n_chunks <= 0 || n_chunks * CH > 0x7fffffff0LL) return fail(BLUEST_ERR_ARG, "problem too large for one plan (%lld chunks)", (long long)n_chunks);
    std::vector<double> vals((size_t)(n_chunks * CH), 0.0);
    std::vector<int32_t> cols((size_t)(n_chunks * CH), 0);
    {
        parallel_items(n_out * S, [&](int item) {
            const int o = item / S;
            std::vector<int64_t> &next = slice_cnt[item];     // start offsets, advanced as the slice writes
            const OutputDesc &od = plan->outs[o];
            for_groups_of_slice(o, item % S, [&](int k, const int64_t *g, const double *ic, int64_t li) {
                for (int j = 0; j < k; j++)
                    for (int l = j; l < k; l++) {
                        const int rr = tri((int)std::min(g[j], g[l]), (int)std::max(g[j], g[l]));
                        const int64_t pos = (int64_t)rows[(size_t)o * nsym + rr].first_chunk * CH + next[rr]++;
                        vals[pos] = (j == l) ? ic[j * k + j] : 0.5 * (ic[j * k + l] + ic[l * k + j]);
                        cols[pos] = (int32_t)od.mapping[li];
                    }
            });
        });
        // padding: value 0, column = the row's first column (keeps max|m| per row exact, adds nothing)
        parallel_items(n_out, [&](int o) {
            for (int rr = 0; rr < nsym; rr++) {
                const size_t r = (size_t)o * nsym + rr;
                const int64_t beg = (int64_t)rows[r].first_chunk * CH, end = beg + (int64_t)rows[r].n_chunks * CH;
                for (int64_t pos = beg + counts[o][rr]; pos < end; pos++) cols[pos] = cols[beg];
            }
        });
    }
    timer.lap("CSR alloc + fill");
    // ---- gradient pass: group-major tiles ------------------------------------------------------------
    {
        int kmax_all = 0;
        for (const auto &od : plan->outs) kmax_all = std::max(kmax_all, od.K);
        plan->fused_tpb = fused_tpb(pick_nt(N), pick_ku(kmax_all));
    }
    // phase 1 (serial, cheap): descriptors and offsets; phase 2 (one thread per output): fill values and indices
    std::vector<TileDesc> tiles;
    std::vector<int64_t> tile_t0;            // index of the tile's first group inside its size bucket
    std::vector<size_t> tile_begin(n_out + 1, 0);
    size_t n_tvals = 0, n_tidx = 0;
    plan->grad_off.assign(n_out, 0);
    int64_t grad_len = 0;
    for (int o = 0; o < n_out; o++) {
        const OutputDesc &od = plan->outs[o];
        plan->grad_off[o] = grad_len;
        const size_t first_tile_of_output = tiles.size();
        tile_begin[o] = first_tile_of_output;
        int64_t li = 0;
        for (int k = 1; k <= od.K; k++) {
            const int64_t Lk = od.sizes[k - 1];
            const int ne = k * (k + 1) / 2;
            for (int64_t t0 = 0; t0 < Lk; t0 += 64) {
                TileDesc td;
                td.val_off = (int64_t)n_tvals;
                n_tidx = (n_tidx + 15) / 16 * 16;
                td.idx_off = (int64_t)n_tidx;
                td.grad_off = grad_len + li + t0;
                td.n_valid = (int32_t)std::min<int64_t>(64, Lk - t0);
                td.k = (int16_t)k; td.out = (int16_t)o;
                n_tvals += (size_t)ne * 64;
                n_tidx += (size_t)k * 64;
                tiles.push_back(td);
                tile_t0.push_back(t0);
            }
            li += Lk;
        }
        // first tile of the output is flagged; the list is padded to a multiple of FUSED_TPB tiles per output with empty
        // tiles so that a workgroup of the fused solve+gradient kernel never straddles two outputs
        if (tiles.size() > first_tile_of_output) tiles[first_tile_of_output].n_valid |= (1 << 30);
        tile_begin[o + 1] = tiles.size();       // real tiles of this output end here (padding follows)
        while ((tiles.size() - first_tile_of_output) % plan->fused_tpb || tiles.size() == first_tile_of_output) {
            TileDesc td;
            td.val_off = 0; td.idx_off = 0; td.grad_off = 0; td.n_valid = (tiles.size() == first_tile_of_output) ? (1 << 30) : 0;
            td.k = 1; td.out = (int16_t)o;
            tiles.push_back(td);
            tile_t0.push_back(0);
        }
        const int bpo = (int)((tiles.size() - first_tile_of_output) / plan->fused_tpb);
        if (o == 0) plan->fused_bpo = bpo;
        else if (plan->fused_bpo != bpo) plan->fused_bpo = 0;
        grad_len += od.L_o;
    }
    std::vector<double> tvals(n_tvals, 0.0);
    std::vector<uint8_t> tidx(n_tidx, 0);
    {
        std::vector<std::vector<int64_t>> gofs(n_out), iofs(n_out);   // start of size bucket k in groups / invcovs
        for (int o = 0; o < n_out; o++) {
            const OutputDesc &od = plan->outs[o];
            gofs[o].assign(od.K + 2, 0); iofs[o].assign(od.K + 2, 0);
            for (int k = 1; k <= od.K; k++) { gofs[o][k + 1] = gofs[o][k] + od.sizes[k - 1] * k; iofs[o][k + 1] = iofs[o][k] + od.sizes[k - 1] * k * k; }
        }
        const int n_slices = host_threads() * 4;
        const size_t nt_all = tiles.size();
        parallel_items(n_slices, [&](int slice) {
            for (size_t t = (size_t)slice * nt_all / n_slices; t < (size_t)(slice + 1) * nt_all / n_slices; t++) {
                const TileDesc &td = tiles[t];
                const int nv = td.n_valid & 0xffff;
                if (nv == 0) continue;
                const OutputDesc &od = plan->outs[td.out];
                const int k = td.k;
                for (int lane = 0; lane < nv; lane++) {
                    const int64_t i = tile_t0[t] + lane;
                    const int64_t *g = od.groups.data() + gofs[td.out][k] + i * k;
                    const double *ic = od.invcovs.data() + iofs[td.out][k] + i * k * k;
                    int e = 0;
                    for (int j = 0; j < k; j++) {
                        tidx[td.idx_off + j * 64 + lane] = (uint8_t)g[j];
                        for (int l = j; l < k; l++, e++)
                            tvals[td.val_off + e * 64 + lane] = (j == l) ? ic[j * k + j] : 0.5 * (ic[j * k + l] + ic[l * k + j]);
                    }
                }
            }
        });
    }
    plan->grad_len = grad_len;

    // ---- inverse maps for combine_grad ------------------------------------------------------------
    std::vector<int32_t> invmap((size_t)n_out * plan->L, -1);
    for (int o = 0; o < n_out; o++)
        for (int64_t li = 0; li < plan->outs[o].L_o; li++) invmap[(size_t)o * plan->L + plan->outs[o].mapping[li]] = (int32_t)li;

    plan->n_chunks = n_chunks;
    plan->n_rows = (int64_t)rows.size();
    plan->n_tiles = (int64_t)tiles.size();
    plan->max_cand = max_candidates;
    plan->phi_bytes = n_chunks * CH * 8 + (plan->shared ? n_chunks / n_out : n_chunks) * CH * 4 + n_chunks * 16;
    plan->grad_bytes = (int64_t)tvals.size() * 8 + (int64_t)tidx.size() + grad_len * 8;

    timer.lap("tiles + inverse maps");
    int rc;
    if ((rc = upload(&plan->d_vals, vals))) return rc;
    if ((rc = upload(&plan->d_cols, cols))) return rc;
    if ((rc = upload(&plan->d_rows, rows))) return rc;
    if ((rc = upload(&plan->d_out_row_begin, out_row_begin))) return rc;
    if ((rc = upload(&plan->d_tiles, tiles))) return rc;
    if ((rc = upload(&plan->d_tvals, tvals))) return rc;
    if ((rc = upload(&plan->d_tidx, tidx))) return rc;
    if ((rc = upload(&plan->d_invmap, invmap))) return rc;
    if ((rc = upload(&plan->d_goff, plan->grad_off))) return rc;
    HIP_TRY(hipMalloc((void **)&plan->d_partial, (size_t)max_candidates * n_chunks * sizeof(double2)));
    HIP_TRY(hipMalloc((void **)&plan->d_v, (size_t)max_candidates * n_out * N * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&plan->d_status, (size_t)max_candidates * n_out * sizeof(int32_t)));
    plan->finalized = true;
    timer.lap("uploads + device allocations");
    // host copies of the reference-layout inputs are no longer needed
    for (auto &od : plan->outs) { std::vector<double>().swap(od.invcovs); std::vector<int64_t>().swap(od.groups); }
    return BLUEST_OK;
}

static int plan_ready(bluest_plan_t plan, int n_cand)
{
    if (!plan) return fail(BLUEST_ERR_ARG, "plan is NULL");
    if (!plan->finalized) return fail(BLUEST_ERR_STATE, "plan not finalized");
    if (n_cand <= 0 || n_cand > plan->max_cand) return fail(BLUEST_ERR_ARG, "n_cand=%d outside 1..%d", n_cand, plan->max_cand);
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
                                  plan->d_tvals, plan->d_tidx, v_dev, status_dev, plan->N, n_out, n_cand, grad_dev, grad_stride, plan->gate)
    if (kmax <= 5) LG(5);
    else if (kmax <= 8) LG(8);
    else LG(12);
#undef LG
}

static void launch_chunks(bluest_plan_t p, const double *m, int n_cand, int64_t m_stride, hipStream_t st)
{
    const int n_out = (int)p->outs.size();
    if (p->shared && n_out >= 2) {
        const int64_t ncpo = p->n_chunks / n_out;
        const unsigned gx = (unsigned)((ncpo + 3) / 4);
#define LCS(OB) hipLaunchKernelGGL((k_phi_chunks_shared<OB>), dim3(gx, (n_out + OB - 1) / OB), dim3(256), 0, st, p->d_vals, p->d_cols, \
                                   p->iters, ncpo, n_out, m, m_stride, n_cand, p->n_chunks, p->d_partial, p->gate)
        // outputs per wavefront: sharing the column stream saves bytes, but the pass is latency-bound, so keep at least
        // ~4096 wavefronts in flight (measured at n=20, n_out=8: OB=8 6.9 us, OB=4 5.5 us, OB=2 5.1 us, OB=1 6.1 us)
        int ob = 8;
        while (ob > 2 && (ncpo * ((n_out + ob - 1) / ob) < 4096 || ob > n_out)) ob /= 2;
        if (ob == 8) LCS(8);
        else if (ob == 4) LCS(4);
        else LCS(2);
#undef LCS
        return;
    }
    hipLaunchKernelGGL(k_phi_chunks, dim3((unsigned)((p->n_chunks + 3) / 4)), dim3(256), 0, st, p->d_vals, p->d_cols,
                       p->iters, p->n_chunks, m, m_stride, n_cand, p->d_partial, p->gate);
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
    launch_chunks(plan, m_dev, n_cand, m_stride, st);
#define LFR(NT) hipLaunchKernelGGL((k_fold_to_record<NT>), dim3(n_out, n_cand), dim3(fold_threads(NT)), 0, st, plan->N, n_out, plan->d_rows, \
                                   plan->nsym, plan->d_partial, plan->n_chunks, phi_dev)
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

extern "C" int bluest_plan_grad(bluest_plan_t plan, const double *v_dev, const int32_t *status_dev, int n_cand,
                                double *grad_dev, int64_t grad_stride, void *stream)
{
    int rc = plan_ready(plan, n_cand); if (rc) return rc;
    if (!v_dev || !status_dev || !grad_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (n_cand > 1 && grad_stride < plan->grad_len) return fail(BLUEST_ERR_ARG, "grad_stride < grad_len");
    const int n_out = (int)plan->outs.size();
    launch_grad(plan, v_dev, status_dev, n_cand, grad_dev, grad_stride, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_plan_eval(bluest_plan_t plan, const double *m_dev, int n_cand, int64_t m_stride, double delta,
                                double *var_dev, double *grad_dev, int64_t grad_stride, int32_t *status_dev, void *stream)
{
    int rc = plan_ready(plan, n_cand); if (rc) return rc;
    if (!m_dev || !var_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (n_cand > 1 && m_stride < plan->L) return fail(BLUEST_ERR_ARG, "m_stride < L_global");
    if (grad_dev && n_cand > 1 && grad_stride < plan->grad_len) return fail(BLUEST_ERR_ARG, "grad_stride < grad_len");
    hipStream_t st = (hipStream_t)stream;
    const int n_out = (int)plan->outs.size();
    int32_t *status = status_dev ? status_dev : plan->d_status;
    launch_chunks(plan, m_dev, n_cand, m_stride, st);
    int kmax = 0;
    for (const auto &od : plan->outs) kmax = std::max(kmax, od.K);
    // fused solve + gradient pass (2 launches per evaluation); groups larger than 12 take the generic tile code inside it
    if (grad_dev && n_cand == 1 && !g_debug_solve) {
        const dim3 grid((unsigned)(plan->n_tiles / plan->fused_tpb));
#define LSG2(NT, KU) hipLaunchKernelGGL((k_solve_grad<NT, KU>), grid, dim3(64 * (fused_tpb(NT, KU) + 1)), 0, st, plan->N, n_out, plan->d_rows, plan->nsym, plan->d_partial, \
                                        delta, plan->d_tiles, plan->n_tiles, plan->fused_bpo, plan->d_tvals, plan->d_tidx, var_dev, plan->d_v, status, grad_dev, plan->gate)
#define LSG(NT) do { if (kmax <= 5) LSG2(NT, 5); else if (kmax <= 6) LSG2(NT, 6); else if (kmax <= 8) LSG2(NT, 8); else LSG2(NT, 12); } while (0)
        NT_DISPATCH(plan->N, LSG);
#undef LSG
#undef LSG2
        HIP_TRY(hipGetLastError());
        return BLUEST_OK;
    }
    const int want = ((grad_dev || plan->always_v) ? 1 : 0) | g_debug_solve;
#define LSC(NT) hipLaunchKernelGGL((k_solve_from_chunks<NT>), dim3(n_out, n_cand), dim3(fold_threads(NT)), 0, st, plan->N, n_out, plan->d_rows, \
                                   plan->nsym, plan->d_partial, plan->n_chunks, delta, want, var_dev, plan->d_v, status, plan->gate)
    NT_DISPATCH(plan->N, LSC);
#undef LSC
    if (grad_dev)
        launch_grad(plan, plan->d_v, status, n_cand, grad_dev, grad_stride, st);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
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
            static int ncu = 0;
            if (!ncu) {
                int dev = 0;
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDevice(&dev));
                HIP_TRY(hipGetDeviceProperties(&prop, dev));
                ncu = prop.multiProcessorCount;
            }
            static int maxp = getenv("BLUEST_PROJ_MAXP") ? atoi(getenv("BLUEST_PROJ_MAXP")) : (int)FusedProj::MAXP;   // timing experiments
            static int maxb = getenv("BLUEST_PROJ_MAXB") ? atoi(getenv("BLUEST_PROJ_MAXB")) : 64;   // 64 measured best at L = 245505 (61 -> 42 us)
            const int nbf = std::max(1, std::min(std::min(nb, ncu), std::min(maxb, (int)FusedProj::MAXB)));
            const int64_t items = (L + 1024LL * nbf - 1) / (1024LL * nbf);
            if (items <= 16 && !getenv("BLUEST_PROJ_MULTI_LAUNCH")) {
#define PF(IT) hipLaunchKernelGGL((k_proj_fused<IT>), dim3(nbf), dim3(1024), 0, st, x_dev, g_dev, lambda, z, floor, L, ws, nb, p_dev, d_dev, \
                                  stats_dev, spg_state, spg_mode, trial_scale, trial_xnew, trial_m, trial_enable, maxp)
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
                hipLaunchKernelGGL(k_proj_p, dim3(nb), dim3(1024), 0, st, L, ws, nb, spg_state);
                hipLaunchKernelGGL(k_proj_q, dim3(1), dim3(64), 0, st, z, L, ws, nb, spg_state, mode);
            }
            hipLaunchKernelGGL(k_proj_b_finish, dim3(1), dim3(1024), 0, st, z, L, ws, nb, spg_state, mode);
        }
#undef PB
        hipLaunchKernelGGL(k_proj_c, dim3(nb), dim3(1024), 0, st, x_dev, g_dev, L, ws, nb, p_dev, d_dev, trial_scale, trial_xnew, trial_m,
                           spg_state);
        hipLaunchKernelGGL(k_proj_d, dim3(1), dim3(64), 0, st, L, ws, nb, stats_dev, spg_state, spg_mode, trial_enable);
        HIP_TRY(hipGetLastError());
        return BLUEST_OK;
    }
#define SP(IT) hipLaunchKernelGGL((k_simplex<IT>), dim3(1), dim3(SIMPLEX_BLOCK), 0, st, x_dev, g_dev, lambda, z, floor, L, p_dev, d_dev, stats_dev, spg_state, spg_mode, \
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
    const int nblocks = (int)std::min<int64_t>((L + 1023) / 1024, SPG_UPD_BLOCKS_MAX);
    hipLaunchKernelGGL(k_spg_update_a_fused, dim3(nblocks), dim3(1024), 0, (hipStream_t)stream, x_dev, g_dev, xnew_dev, grad_dev, plan->d_goff,
                       plan->d_invmap, (int)plan->outs.size(), scale_dev, state_dev, floor, L, (double2 *)work_dev);
    hipLaunchKernelGGL(k_spg_update_b, dim3(1), dim3(64), 0, (hipStream_t)stream, state_dev, (const double2 *)work_dev, nblocks);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_spg_update(double *x_dev, double *g_dev, const double *xnew_dev, const double *gnew_dev, double *state_dev,
                                 double floor, int64_t L, double *work_dev, void *stream)
{
    int rc = require_gpu(); if (rc) return rc;
    if (!x_dev || !g_dev || !xnew_dev || !gnew_dev || !state_dev || !work_dev || L <= 0) return fail(BLUEST_ERR_ARG, "bad argument");
    const int nblocks = (int)std::min<int64_t>((L + 1023) / 1024, SPG_UPD_BLOCKS_MAX);
    hipLaunchKernelGGL(k_spg_update_a, dim3(nblocks), dim3(1024), 0, (hipStream_t)stream, x_dev, g_dev, xnew_dev, gnew_dev, state_dev, floor, L,
                       (double2 *)work_dev);
    hipLaunchKernelGGL(k_spg_update_b, dim3(1), dim3(64), 0, (hipStream_t)stream, state_dev, (const double2 *)work_dev, nblocks);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

// ------------------------------------------------------------------------------------------------------
// integer projection batch
// ------------------------------------------------------------------------------------------------------
extern "C" int bluest_intproj_eval(int N, int n_out, int LL, const double *base_dev, const double *cols_dev, const double *ms_dev,
                                   int64_t n_cand, double *V_dev, void *stream)
{
    int rc = require_gpu(); if (rc) return rc;
    if (N <= 0 || N > BLUEST_MAX_MODELS) return fail(BLUEST_ERR_ARG, "N=%d out of range", N);
    if (n_out <= 0 || n_out > 65535) return fail(BLUEST_ERR_ARG, "n_out=%d out of range", n_out);
    if (LL <= 0 || LL > 32) return fail(BLUEST_ERR_ARG, "LL=%d out of range (1..32)", LL);
    if (n_cand <= 0 || n_cand > 0x7fffffffLL) return fail(BLUEST_ERR_ARG, "n_cand out of range");
    if (!base_dev || !cols_dev || !ms_dev || !V_dev) return fail(BLUEST_ERR_ARG, "null pointer");
#define LIP(NT) hipLaunchKernelGGL((k_intproj<NT>), dim3((unsigned)n_cand, n_out), dim3(64), 0, (hipStream_t)stream, N, n_out, LL, \
                                   base_dev, cols_dev, ms_dev, n_cand, V_dev)
    NT_DISPATCH(N, LIP);
#undef LIP
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}
