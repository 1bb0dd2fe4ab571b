// micro-benchmark: VALU issue rate per SIMD by instruction class and waves per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 11u, a5 = a0 + 13u, a6 = a0 ^ 77u, a7 = a0 ^ 99u;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
    const double dm = 1.0000001, da = 1e-9;
    const float fm = 1.0001f, fa = 1e-5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) { // v_add_u32 (8 independent chains)
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
            } else if (KIND == 1) { // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm), "v"(da));
            } else if (KIND == 2) { // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fm), "v"(fa));
            } else if (KIND == 3) { // v_add_f64
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(da));
            } else if (KIND == 4) { // v_cmp_lt_f64 + v_cndmask (pairs)
                asm volatile("v_cmp_lt_f64 vcc, %0, %4\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f64 vcc, %1, %4\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_lt_f64 vcc, %0, %4\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f64 vcc, %1, %4\n v_cndmask_b32 %3, %3, %2, vcc"
                    : "+v"(d0), "+v"(d1), "+v"(a2), "+v"(a3) : "v"(dm) : "vcc");
            } else if (KIND == 5) { // v_and_b32 / v_lshlrev / v_bcnt mix (bit ops)
                asm volatile("v_and_b32 %0, %0, %8\n v_lshlrev_b32 %1, 1, %1\n v_bcnt_u32_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_lshlrev_b32 %5, 1, %5\n v_bcnt_u32_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
            } else if (KIND == 6) { // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm));
            } else if (KIND == 7) { // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm), "v"(da));
            } else if (KIND == 8) { // v_cvt_f32_f64 / v_cvt_f64_f32 alternating
                asm volatile("v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1\n v_cvt_f32_f64 %6, %2\n v_cvt_f32_f64 %7, %3\n v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
            } else if (KIND == 9) { // s_ (SALU) adds: scalar issue rate
                asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "s20", "s21", "s22", "s23");
            } else if (KIND == 10) { // mixed: v_add_u32 + s_add (can they co-issue from one wave? no; across waves yes)
                asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %4\n s_add_u32 s21, s21, 1\n v_add_u32 %2, %2, %4\n s_add_u32 s22, s22, 1\n v_add_u32 %3, %3, %4\n s_add_u32 s23, s23, 1"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "s20", "s21", "s22", "s23");
            }
        }
    }
    unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) ^ (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if (r == 0x12345678u) out[0] = r;
}

template <int KIND> float run(unsigned *d, int wgs_per_cu, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    unsigned *d; CHECK(hipMalloc(&d, 64));
    const char *names[] = {"v_add_u32", "v_fma_f64", "v_fma_f32", "v_add_f64", "v_cmp_f64+cndmask", "bit ops mix", "v_mul_f64", "v_pk_fma_f32", "v_cvt f32<->f64", "s_add_u32", "v_add_u32+s_add interleaved"};
    const int iters = 20000;
    for (int kind = 0; kind < 11; ++kind) {
        printf("%-28s", names[kind]);
        for (int w : {1, 2, 4, 6, 8}) {           // 256-thread WGs per CU = waves per SIMD
            float ms = 0;
            switch (kind) {
            case 0: ms = run<0>(d, w, iters); break; case 1: ms = run<1>(d, w, iters); break; case 2: ms = run<2>(d, w, iters); break;
            case 3: ms = run<3>(d, w, iters); break; case 4: ms = run<4>(d, w, iters); break; case 5: ms = run<5>(d, w, iters); break;
            case 6: ms = run<6>(d, w, iters); break; case 7: ms = run<7>(d, w, iters); break; case 8: ms = run<8>(d, w, iters); break;
            case 9: ms = run<9>(d, w, iters); break; case 10: ms = run<10>(d, w, iters); break;
            }
            // instructions per wave = iters * 64; per SIMD = w * that; cycles at 2.4 GHz (nominal)
            const double instr = (double)iters * 64 * w;
            const double cyc = ms * 1e-3 * 2.4e9;
            printf("  w=%d: %.2f cyc/instr", w, cyc / instr);
        }
        printf("\n");
    }
    return 0;
}
