// intproj.hip -- Part 4 of include/bluest_hip.h: candidate batch of the integer projection (bluest/misc.py:293-294, 368-369).
#include "solve.hpp"

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
    clear_pads(lds, N, lane, WAVE);
    __syncthreads();
    for (int t = lane; t < NN; t += WAVE) {
        double x = b[t];
        for (int j = 0; j < LL; j++) x = fma(sms[j], cl[(int64_t)j * NN + t], x);
        lds.at(t / N, t % N) = x;
    }
    __syncthreads();
    const bool sup = lane < N && lds.at(lane, lane) > 0.0;
    double var = 0.0, vdummy = 0.0;
    int32_t status = 0;
    solve_wave<NT>(lds, N, 0.0, sup, sup, true, false, &var, &vdummy, &status, lane);
    if (lane == 0) V[cand * n_out + o] = (status == BLUEST_EVAL_OK) ? var : INFINITY;
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

