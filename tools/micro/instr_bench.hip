// Experiment (not product): issue cost of the instructions the register solve is made of, ONE wave on a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int WHICH>
__global__ __launch_bounds__(64) void k(long long *out, double *sink, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3, a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7, f = 1e-9 * seed;
    long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < 4; it++) {
        if (WHICH == 0) {   // independent readlanes to different SGPRs
            asm volatile(REP64("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 5\n v_readlane_b32 s22, %0, 7\n v_readlane_b32 s23, %0, 9\n") :: "v"(__double2loint(a0)) : "s20", "s21", "s22", "s23");
        } else if (WHICH == 1) {   // independent f64 fma (8 accumulators)
            asm volatile(REP64("v_fmac_f64 %0, %8, %1\n v_fmac_f64 %1, %8, %2\n v_fmac_f64 %2, %8, %3\n v_fmac_f64 %3, %8, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(f));
        } else if (WHICH == 2) {   // dependent f64 fma chain
            asm volatile(REP64("v_fmac_f64 %0, %1, %0\n v_fmac_f64 %0, %1, %0\n v_fmac_f64 %0, %1, %0\n v_fmac_f64 %0, %1, %0\n") : "+v"(a0) : "v"(f));
        } else if (WHICH == 3) {   // the solve's column update: 2 readlanes -> fmac with the SGPR pair
            asm volatile(REP64("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_fmac_f64 %2, s[20:21], %4\n v_readlane_b32 s22, %0, 5\n v_readlane_b32 s23, %1, 5\n v_fmac_f64 %3, s[22:23], %4\n")
                         :: "v"(__double2loint(a0)), "v"(__double2hiint(a0)), "v"(a1), "v"(a2), "v"(f) : "s20", "s21", "s22", "s23");
        } else if (WHICH == 4) {   // block of 8 readlanes then 4 fmacs (as the compiler emits)
            asm volatile(REP64("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %0, 5\n v_readlane_b32 s23, %1, 5\n v_readlane_b32 s24, %0, 6\n v_readlane_b32 s25, %1, 6\n v_readlane_b32 s26, %0, 7\n v_readlane_b32 s27, %1, 7\n"
                               "v_fmac_f64 %2, s[20:21], %6\n v_fmac_f64 %3, s[22:23], %6\n v_fmac_f64 %4, s[24:25], %6\n v_fmac_f64 %5, s[26:27], %6\n")
                         :: "v"(__double2loint(a0)), "v"(__double2hiint(a0)), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(f) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (WHICH == 5) {   // dpp broadcast fmac, independent accumulators, with the required s_nop
            asm volatile(REP64("s_nop 1\n v_fmac_f64_dpp %0, %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f64_dpp %1, %1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                               "s_nop 1\n v_fmac_f64_dpp %2, %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f64_dpp %3, %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
        } else if (WHICH == 6) {   // dpp broadcast fmac without nops between independent ones
            asm volatile(REP64("v_fmac_f64_dpp %0, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f64_dpp %2, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(f));
        } else if (WHICH == 7) {   // dependent rcp chain
            asm volatile(REP64("v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n") : "+v"(a0));
        } else if (WHICH == 8) {   // independent f32 fma for reference
            float b0 = (float)a0, b1 = (float)a1, b2 = (float)a2, b3 = (float)a3, g = (float)f;
            asm volatile(REP64("v_fmac_f32 %0, %4, %1\n v_fmac_f32 %1, %4, %2\n v_fmac_f32 %2, %4, %3\n v_fmac_f32 %3, %4, %0\n") : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(g));
            a0 += b0 + b1 + b2 + b3;
        } else if (WHICH == 9) {   // v_mov_b64 dpp broadcast
            asm volatile(REP64("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b64_dpp %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (WHICH == 10) {  // f64 mul independent
            asm volatile(REP64("v_mul_f64 %0, %4, %5\n v_mul_f64 %1, %5, %6\n v_mul_f64 %2, %6, %7\n v_mul_f64 %3, %7, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
template <int W> void run(const char *name, int per64)
{
    long long *d; double *s; hipMalloc(&d, 8); hipMalloc(&s, 512);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<W>, dim3(1), dim3(64), 0, 0, d, s, 1.0);
    long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
    printf("%-60s %6.2f cycles per instruction (%d instructions)\n", name, (double)c / (4.0 * 64 * per64), 4 * 64 * per64);
}
int main()
{
    run<0>("v_readlane_b32, independent", 4);
    run<1>("v_fmac_f64, independent", 4);
    run<2>("v_fmac_f64, dependent chain", 4);
    run<3>("2 readlane + fmac(sgpr pair), interleaved", 6);
    run<4>("8 readlane then 4 fmac(sgpr pair)", 12);
    run<5>("s_nop 1 + v_fmac_f64_dpp row_newbcast (src = dst)", 8);
    run<6>("v_fmac_f64_dpp row_newbcast, independent, no nop", 4);
    run<7>("v_rcp_f64, dependent chain", 4);
    run<8>("v_fmac_f32, independent", 4);
    run<9>("v_mov_b64_dpp row_newbcast", 4);
    run<10>("v_mul_f64, independent", 4);
    return 0;
}
