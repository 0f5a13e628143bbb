// mirrors.hip -- Part 1 of include/bluest_hip.h: same-name, same-argument-order mirrors of the reference's native module
// _cmisc_bluest (bluest/cmisc.cpp) on the reference data layout, plus the per-group pseudo-inverse of sap.py:69-79.
// Pointers may be host or device; host buffers are staged through HBM (compatibility path, PCIe-inclusive).
#include "common.hpp"

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

template <int K, typename G>   // G: element type of the group list (int64_t: the reference layout; uint8_t: the plan's device copy)
__global__ __launch_bounds__(64) void k_group_pinv(const double *__restrict__ C, int N, int64_t Lk, const G *__restrict__ g,
                             double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Lk) return;
    double A[K * K], V[K * K];
    const G *gi = g + i * K;
#pragma unroll
    for (int j = 0; j < K; j++)
#pragma unroll
        for (int l = 0; l < K; l++) {
            const int64_t a = (int64_t)gi[j], b = (int64_t)gi[l];
            A[j * K + l] = 0.5 * (C[a * N + b] + C[b * N + a]);
        }
    sym_pinv_jacobi<K>(A, V, out + i * (int64_t)K * K);
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

template <typename G>
static int launch_group_pinv_t(const double *dC, int N, int k, int64_t Lk, const G *dg, double *dout, hipStream_t st)
{
    const dim3 grid((unsigned)((Lk + 63) / 64)), block(64);
#define GP(KK) case KK: hipLaunchKernelGGL((k_group_pinv<KK, G>), grid, block, 0, st, dC, N, Lk, dg, dout); break;
    switch (k) {
        GP(1) GP(2) GP(3) GP(4) GP(5) GP(6) GP(7) GP(8) GP(9) GP(10) GP(11) GP(12) GP(13) GP(14) GP(15) GP(16)
        default: return fail(BLUEST_ERR_ARG, "group size k=%d > %d", k, BLUEST_MAX_GROUP);
    }
#undef GP
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}
int launch_group_pinv(const double *dC, int N, int k, int64_t Lk, const int64_t *dg, double *dout, hipStream_t st)
{
    return launch_group_pinv_t<int64_t>(dC, N, k, Lk, dg, dout, st);
}
int launch_group_pinv_u8(const double *dC, int N, int k, int64_t Lk, const uint8_t *dg, double *dout, hipStream_t st)
{
    return launch_group_pinv_t<uint8_t>(dC, N, k, Lk, dg, dout, st);
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

