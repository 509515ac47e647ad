// ubench_valu.hip -- issue-rate microbenchmark for the VALU instructions the PairHMM kernel is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip ; run on the GPU box.
// Prints wave-instructions per cycle per SIMD at 1, 2, 4, 8 waves/SIMD (clock measured with s_memtime/s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(S) S S S S S S S S
template <int MODE>
__global__ void k(float* out, int iters, unsigned long long* clk)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 0.999f, c = 0.001f;
    unsigned m0 = threadIdx.x * 2654435761u, m1 = m0 ^ 0x9e3779b9u;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = 0.999, dc = 0.001;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (MODE == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
        if (MODE == 2) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_fmac_f32 %1, %8, %9\n v_mul_f32 %2, %2, %8\n v_fmac_f32 %3, %8, %9\n v_mul_f32 %4, %4, %8\n v_fmac_f32 %5, %8, %9\n v_mul_f32 %6, %6, %8\n v_fmac_f32 %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (MODE == 3) { REP8(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (MODE == 4) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %0\n v_cndmask_b32 %2, %8, %9, vcc\n v_add_co_u32 %1, vcc, %1, %1\n v_cndmask_b32 %3, %8, %9, vcc\n v_add_co_u32 %0, vcc, %0, %0\n v_cndmask_b32 %4, %8, %9, vcc\n v_add_co_u32 %1, vcc, %1, %1\n v_cndmask_b32 %5, %8, %9, vcc" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) }
        if (MODE == 5) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));) }
        if (MODE == 6) { REP8(asm volatile("s_nop 1\n v_mul_f32_dpp %0, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %2, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mul_f32_dpp %4, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %6, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mul_f32_dpp %1, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %3, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mul_f32_dpp %5, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %7, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (MODE == 8) { REP8(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2" : "+v"(m0), "+v"(m1) : "v"(m0 | 1u), "v"(a2), "v"(a3));) }
        if (MODE == 9) { REP8(asm volatile("v_max_i32 %0, %0, %2\n v_max_i32 %1, %1, %2\n v_max_i32 %0, %0, %3\n v_max_i32 %1, %1, %3\n v_max_i32 %0, %0, %2\n v_max_i32 %1, %1, %2\n v_max_i32 %0, %0, %3\n v_max_i32 %1, %1, %3" : "+v"(m0), "+v"(m1) : "v"(a2), "v"(a3));) }
        if (MODE == 10) { REP8(asm volatile("v_sub_u32 %2, %0, %1\n v_alignbit_b32 %3, %3, %2, 31\n v_sub_u32 %4, %1, %0\n v_alignbit_b32 %3, %3, %4, 31\n v_sub_u32 %2, %0, %1\n v_alignbit_b32 %3, %3, %2, 31\n v_sub_u32 %4, %1, %0\n v_alignbit_b32 %3, %3, %4, 31" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 11) { REP8(asm volatile("v_cmp_eq_u32_e64 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %2, %4, %5, s[20:21]\n v_cmp_gt_i32_e64 s[22:23], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %3, %4, %5, s[22:23]\n v_cmp_eq_u32_e64 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %2, %4, %5, s[20:21]\n v_cmp_gt_i32_e64 s[22:23], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %3, %4, %5, s[22:23]" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23");) }
        if (MODE == 12) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %2, %2, %1, 31\n v_alignbit_b32 %3, %3, %1, 31\n v_alignbit_b32 %4, %4, %1, 31\n v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %2, %2, %1, 31\n v_alignbit_b32 %3, %3, %1, 31\n v_alignbit_b32 %4, %4, %1, 31" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        // compare -> select -> carry-append chains as in the Smith-Waterman cell: (13) hazard covered by s_nop 1, (14) by two independent adds,
        // (15) by one add + s_nop 0; 4 VALU per group in 13, 6 in 14, 5 in 15 (per_iter is 64 VALU: 16 / 10.67 / 12.8 groups -- use 48 VALU for 14/15: see run())
        if (MODE == 13) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %0, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_cmp_lt_u32 vcc, %1, %0\n s_nop 1\n v_cndmask_b32 %4, %0, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4) : : "vcc");) }
        if (MODE == 14) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %5, %5, %0\n v_add_u32 %6, %6, %1\n v_cndmask_b32 %2, %0, %1, vcc\n v_cmp_lt_u32 vcc, %1, %0\n v_add_u32 %5, %5, %0\n v_add_u32 %6, %6, %1\n v_cndmask_b32 %4, %0, %1, vcc" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : : "vcc");) }
        if (MODE == 15) { REP8(asm volatile("v_add_u32 %5, %5, %0\n v_add_u32 %6, %6, %1\n v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %0, %1, vcc\n v_add_u32 %5, %5, %0\n v_add_u32 %6, %6, %1\n v_cmp_lt_u32 vcc, %1, %0\n s_nop 1\n v_cndmask_b32 %4, %0, %1, vcc" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : : "vcc");) }
        if (MODE == 16) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %0, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %4, %0, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4) : : "vcc");) }
        // the same chains with the compare done by v_sub_co_u32 (borrow = a < b) instead of VOPC
        if (MODE == 17) { REP8(asm volatile("v_sub_co_u32 %5, vcc, %0, %1\n v_cndmask_b32 %2, %0, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_sub_co_u32 %5, vcc, %1, %0\n v_cndmask_b32 %4, %0, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : : "vcc");) }
        if (MODE == 18) { REP8(asm volatile("v_sub_co_u32 %5, vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %0, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_sub_co_u32 %5, vcc, %1, %0\n s_nop 1\n v_cndmask_b32 %4, %0, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : : "vcc");) }
        if (MODE == 19) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0" : "+v"(m0), "+v"(m1) : : "vcc");) }
        if (MODE == 20) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %0, %1, vcc\n v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %3, %0, %1, vcc\n v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %4, %0, %1, vcc\n v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %2, %0, %1, vcc" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4) : : "vcc");) }
        if (MODE == 21) { REP8(asm volatile("v_sub_co_u32 %5, vcc, %0, %1\n v_cndmask_b32 %2, %0, %1, vcc\n v_sub_co_u32 %5, vcc, %1, %0\n v_cndmask_b32 %3, %0, %1, vcc\n v_sub_co_u32 %5, vcc, %0, %1\n v_cndmask_b32 %4, %0, %1, vcc\n v_sub_co_u32 %5, vcc, %1, %0\n v_cndmask_b32 %2, %0, %1, vcc" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : : "vcc");) }
        if (MODE == 22) { REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3) : : "vcc");) }
        if (MODE == 23) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %2\n v_lshl_add_u32 %1, %1, 1, %2\n v_lshl_add_u32 %3, %3, 1, %2\n v_lshl_add_u32 %4, %4, 1, %2\n v_lshl_add_u32 %0, %0, 1, %2\n v_lshl_add_u32 %1, %1, 1, %2\n v_lshl_add_u32 %3, %3, 1, %2\n v_lshl_add_u32 %4, %4, 1, %2" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 24) { REP8(asm volatile("v_sub_co_u32 %5, vcc, %0, %1\n v_cndmask_b32 %2, %0, %1, vcc\n v_lshrrev_b32 %5, 31, %5\n v_sub_co_u32 %6, vcc, %1, %0\n v_cndmask_b32 %4, %0, %1, vcc\n v_lshrrev_b32 %6, 31, %6\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : : "vcc");) }
        if (MODE == 25) { REP8(asm volatile("v_and_or_b32 %0, %0, %2, %1\n v_and_or_b32 %1, %1, %2, %0\n v_and_or_b32 %3, %3, %2, %1\n v_and_or_b32 %4, %4, %2, %0\n v_and_or_b32 %0, %0, %2, %1\n v_and_or_b32 %1, %1, %2, %0\n v_and_or_b32 %3, %3, %2, %1\n v_and_or_b32 %4, %4, %2, %0" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 26) { REP8(asm volatile("v_min_u32 %0, %0, %2\n v_min_u32 %1, %1, %2\n v_max_u32 %3, %3, %2\n v_max_u32 %4, %4, %2\n v_min_u32 %0, %0, %2\n v_min_u32 %1, %1, %2\n v_max_u32 %3, %3, %2\n v_max_u32 %4, %4, %2" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 27) { REP8(asm volatile("v_xor_b32 %0, %0, %2\n v_and_b32 %1, %1, %2\n v_or_b32 %3, %3, %2\n v_lshlrev_b32 %4, 1, %4\n v_xor_b32 %0, %0, %2\n v_and_b32 %1, %1, %2\n v_or_b32 %3, %3, %2\n v_lshrrev_b32 %4, 1, %4" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        // packed 16-bit integer ops (two Smith-Waterman cells per instruction if they issue at full rate)
        if (MODE == 28) { REP8(asm volatile("v_pk_add_u16 %0, %0, %2\n v_pk_add_u16 %1, %1, %2\n v_pk_sub_i16 %3, %3, %2\n v_pk_sub_i16 %4, %4, %2\n v_pk_add_u16 %0, %0, %2\n v_pk_add_u16 %1, %1, %2\n v_pk_sub_i16 %3, %3, %2\n v_pk_sub_i16 %4, %4, %2" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 29) { REP8(asm volatile("v_pk_max_i16 %0, %0, %2\n v_pk_max_i16 %1, %1, %2\n v_pk_min_i16 %3, %3, %2\n v_pk_min_u16 %4, %4, %2\n v_pk_max_i16 %0, %0, %2\n v_pk_max_i16 %1, %1, %2\n v_pk_min_i16 %3, %3, %2\n v_pk_min_u16 %4, %4, %2" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 30) { REP8(asm volatile("v_pk_lshrrev_b16 %0, 15, %0\n v_pk_lshlrev_b16 %1, 1, %1\n v_pk_ashrrev_i16 %3, 15, %3\n v_pk_mad_u16 %4, %4, %2, %1\n v_pk_lshrrev_b16 %0, 15, %0\n v_pk_lshlrev_b16 %1, 1, %1\n v_pk_ashrrev_i16 %3, 15, %3\n v_pk_mad_u16 %4, %4, %2, %1" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 31) { REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));) }
        if (MODE == 32) { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));) }
        if (MODE == 33) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));) }
        if (MODE == 34) { REP8(asm volatile("v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %0, %3, %2, vcc\n v_cndmask_b32 %1, %3, %2, vcc\n v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %0, %3, %2, vcc\n v_cndmask_b32 %1, %3, %2, vcc" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c) : );) }
        if (MODE == 35) { REP8(asm volatile("v_add_u32_sdwa %0, %0, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %1, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %3, %3, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %4, %4, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_add_u32_sdwa %0, %0, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %1, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %3, %3, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %4, %4, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 36) { REP8(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2" : "+v"(m0), "+v"(m1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 7) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3) + (float)(m0 + m1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE> void run(const char* name, float* out, unsigned long long* clk)
{
    const int iters = 2000, per_iter = MODE == 13 ? 48 : 64;  // VALU instructions per loop iteration (s_nop not counted)
    printf("%-34s", name); fflush(stdout);
    for (int wps : {1, 2, 4, 8}) {
        int threads = 256, blocks = 256 * wps;  // 256 CUs x (wps waves/SIMD x 4 SIMD / 4 waves per block)
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10, clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        double ghz = (double)h[0] / ((double)h[1] * 10.0);  // memrealtime ticks at 100 MHz
        double winstr = (double)blocks * 4 * iters * per_iter;
        double cyc_per_simd = ms * 1e-3 * ghz * 1e9;       // cycles elapsed (each SIMD runs wps waves)
        double ipc = winstr / (256.0 * 4) / cyc_per_simd;  // wave-instr per cycle per SIMD
        printf("  w%d: %5.3f ipc (%4.2f cyc/instr, %.2f GHz)", wps, ipc, 1.0 / ipc, ghz);
    }
    printf("\n");
    fflush(stdout);
}


#define SALU2 "s_xor_b64 s[20:21], s[20:21], s[22:23]\n s_nor_b64 s[24:25], s[20:21], s[22:23]\n"
template <int NS>
__global__ void ksalu(float* out, int iters, unsigned long long* clk)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 0.999f, c = 0.001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#define BODY(S) asm volatile("v_fma_f32 %0, %0, %8, %9\n" S "v_fma_f32 %1, %1, %8, %9\n" S "v_fma_f32 %2, %2, %8, %9\n" S "v_fma_f32 %3, %3, %8, %9\n" S "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "scc");
        if (NS == 0) { REP8(BODY("")) }
        if (NS == 2) { REP8(BODY("s_xor_b64 s[20:21], s[20:21], s[22:23]\n")) }   // 4 SALU per 8 VALU
        if (NS == 4) { REP8(BODY(SALU2)) }                                          // 8 SALU per 8 VALU
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NS> void run_salu(const char* name, float* out, unsigned long long* clk)
{
    const int iters = 2000, per_iter = 64;
    printf("%-34s", name); fflush(stdout);
    for (int wps : {1, 2, 4, 8}) {
        int threads = 256, blocks = 256 * wps;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(ksalu<NS>, dim3(blocks), dim3(threads), 0, 0, out, 10, clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(ksalu<NS>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        double ghz = (double)h[0] / ((double)h[1] * 10.0);
        double winstr = (double)blocks * 4 * iters * per_iter;
        double cyc = ms * 1e-3 * ghz * 1e9;
        double ipc = winstr / (256.0 * 4) / cyc;
        printf("  w%d: %4.2f cyc/VALU (%.2f GHz)", wps, 1.0 / ipc, ghz);
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float))); CHECK(hipMalloc(&clk, 16));
    run<36>("v_add_u32 (VOP2)", out, clk);
    run<35>("v_add_u32_sdwa sext WORD/BYTE", out, clk);
    if (getenv("UBENCH_NEW_ONLY")) return 0;
    run<0>("v_fma_f32 (8 indep)", out, clk);
    run<7>("v_fma_f32 (1 dependent chain)", out, clk);
    run<1>("v_pk_fma_f32 (4 indep)", out, clk);
    run<2>("v_mul_f32 / v_fmac_f32", out, clk);
    run<3>("v_mov_b32_dpp row_shr:1", out, clk);
    run<6>("v_mul/fmac_f32_dpp row_shr:1", out, clk);
    run<4>("v_add_co_u32 + v_cndmask_b32", out, clk);
    run<5>("v_fma_f64 (4 indep)", out, clk);
    run<31>("v_mul_f64 (4 indep)", out, clk);
    run<32>("v_add_f64 (4 indep)", out, clk);
    run<33>("v_fma_f64 (1 dependent chain)", out, clk);
    run<8>("v_add_u32 (4 indep)", out, clk);
    run<9>("v_max_i32 (2 chains)", out, clk);
    run<10>("v_sub_u32 + v_alignbit_b32", out, clk);
    run<12>("v_alignbit_b32 (4 indep)", out, clk);
    run<13>("cmp, s_nop 1, cndmask, addc", out, clk);
    run<14>("cmp, add, add, cndmask", out, clk);
    run<15>("add, add, cmp, s_nop 1, cndmask", out, clk);
    run<16>("cmp, cndmask, addc (NO nop) + adds", out, clk);
    run<17>("sub_co, cndmask, addc (no nop) + adds", out, clk);
    run<18>("sub_co, s_nop 1, cndmask, addc + adds", out, clk);
    run<19>("v_cmp_lt_u32 to vcc (8 indep)", out, clk);
    run<20>("v_cmp, cndmask back to back", out, clk);
    run<21>("v_sub_co, cndmask back to back", out, clk);
    run<22>("v_addc_co_u32 (carry in and out)", out, clk);
    run<23>("v_lshl_add_u32", out, clk);
    run<24>("sub_co, cndmask, lshrrev + adds", out, clk);
    run<25>("v_and_or_b32", out, clk);
    run<26>("v_min_u32 / v_max_u32", out, clk);
    run<27>("v_xor/and/or/lshl/lshr (VOP2)", out, clk);
    run<28>("v_pk_add_u16 / v_pk_sub_i16", out, clk);
    run<29>("v_pk_max_i16 / v_pk_min_i16 / min_u16", out, clk);
    run<30>("v_pk_lshr/lshl/ashr_b16 + v_pk_mad_u16", out, clk);
    run<11>("v_cmp_e64 sgpr + s_nop 1 + v_cndmask_e64 (counted as 2)", out, clk);
    run_salu<0>("8 v_fma + 0 SALU", out, clk);
    run_salu<2>("8 v_fma + 4 SALU (interleaved)", out, clk);
    run_salu<4>("8 v_fma + 8 SALU (interleaved)", out, clk);
    return 0;
}
