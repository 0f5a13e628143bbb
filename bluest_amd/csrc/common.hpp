// common.hpp -- shared by every translation unit of libbluest_hip.so: error plumbing, HIP_TRY, wavefront reductions.
#pragma once
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
#include <mutex>

#include "bluest_hip.h"

// ---- error plumbing (definitions in runtime.hip) -----------------------------------------------------
int fail(int code, const char *fmt, ...);
int require_gpu();

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            return fail(BLUEST_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                        __LINE__);                                                                           \
    } while (0)

// per-group pseudo-inverse launcher (mirrors.hip), also used by the plan's set-up
int launch_group_pinv(const double *dC, int N, int k, int64_t Lk, const int64_t *dg, double *dout, hipStream_t st);
int launch_group_pinv_u8(const double *dC, int N, int k, int64_t Lk, const uint8_t *dg, double *dout, hipStream_t st);

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

