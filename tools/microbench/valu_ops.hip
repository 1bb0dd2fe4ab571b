// micro-benchmark: per-opcode VALU throughput on one SIMD (6 waves/SIMD, 8 independent chains per wave), gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X) \
  X(0, "v_add_u32 %0, %0, %8") X(1, "v_and_b32 %0, %0, %8") X(2, "v_lshlrev_b32 %0, 1, %0") X(3, "v_lshrrev_b32 %0, 1, %0") \
  X(4, "v_bcnt_u32_b32 %0, %8, %0") X(5, "v_ffbl_b32 %0, %0") X(6, "v_bfe_u32 %0, %0, 3, 7") X(7, "v_cndmask_b32 %0, %0, %8, vcc") \
  X(8, "v_cmp_lt_u32 vcc, %0, %8") X(9, "v_cmp_lt_f32 vcc, %0, %8") X(10, "v_mov_b32 %0, %8") X(11, "v_lshl_add_u32 %0, %0, 2, %8") \
  X(12, "v_lshl_or_b32 %0, %0, 5, %8") X(13, "v_add3_u32 %0, %0, %8, %8") X(14, "v_mul_lo_u32 %0, %0, %8") X(15, "v_mad_u32_u24 %0, %0, %8, %8") \
  X(16, "v_cvt_f32_i32 %0, %0") X(17, "v_cvt_i32_f32 %0, %0") X(18, "v_floor_f32 %0, %0") X(19, "v_sqrt_f32 %0, %0") \
  X(20, "v_max_f32 %0, %0, %8") X(21, "v_min_i32 %0, %0, %8") X(22, "v_med3_f32 %0, %0, %8, %8") X(23, "v_sub_f32 %0, %0, %8") \
  X(24, "v_mul_f32 %0, %0, %8") X(25, "v_fma_f32 %0, %0, %8, %8") X(26, "v_xor_b32 %0, %0, %8") X(27, "v_or_b32 %0, %0, %8") \
  X(28, "v_bitop3_b32 %0, %0, %8, %8 bitop3:0x40") X(29, "v_sub_u32 %0, %0, %8") X(30, "v_not_b32 %0, %0") X(31, "v_and_or_b32 %0, %0, %8, %8") \
  X(32, "v_cndmask_b32_e64 %0, %0, %8, s[20:21]") X(33, "v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]") X(34, "v_ashrrev_i32 %0, 3, %0") X(35, "v_bfi_b32 %0, %8, %0, %8") \
  X(36, "v_cvt_f32_u32 %0, %0") X(37, "v_rcp_f32 %0, %0") X(38, "v_ceil_f32 %0, %0") X(39, "v_max_i32 %0, %0, %8") \
  X(40, "v_cmp_eq_u32 vcc, 0, %0") X(41, "v_cmp_ne_u32_e64 s[24:25], 0, %0") X(42, "v_readlane_b32 s26, %0, 3") X(43, "v_perm_b32 %0, %0, %8, %8")

template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 11u, a5 = a0 + 13u, a6 = a0 ^ 77u, a7 = a0 ^ 99u;
    asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define X(ID, STR) if (KIND == ID) { \
            asm volatile(STR : "+v"(a0) : "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26"); \
            asm volatile(STR : "+v"(a1) : "v"(a0), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26"); \
            asm volatile(STR : "+v"(a2) : "v"(a1), "v"(a0), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26"); \
            asm volatile(STR : "+v"(a3) : "v"(a1), "v"(a2), "v"(a0), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26"); \
            asm volatile(STR : "+v"(a4) : "v"(a1), "v"(a2), "v"(a3), "v"(a0), "v"(a5), "v"(a6), "v"(a7), "v"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26"); \
            asm volatile(STR : "+v"(a5) : "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a0), "v"(a6), "v"(a7), "v"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26"); \
            asm volatile(STR : "+v"(a6) : "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a0), "v"(a7), "v"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26"); \
            asm volatile(STR : "+v"(a7) : "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a0), "v"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26"); }
            OPS(X)
#undef X
        }
    }
    unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345678u) out[0] = r;
}

template <int KIND> float run(unsigned *d, int iters)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 6), dim3(256), 0, 0, d, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 6), dim3(256), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    unsigned *d; if (hipMalloc(&d, 64) != hipSuccess) return 1;
    const int iters = 4000;
    // calibrate the clock on v_add_u32 = 2 cycles nominal? just print relative to kind 0
    float base = 0;
#define X(ID, STR) { float ms = run<ID>(d, iters); if (ID == 0) base = ms; printf("%-48s %6.2f x v_add_u32   (%.2f cyc at 2.4 GHz)\n", STR, ms / base, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * 6)); }
    OPS(X)
#undef X
    return 0;
}
