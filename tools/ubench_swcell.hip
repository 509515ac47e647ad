// ubench_swcell.hip -- what the Smith-Waterman cell of sw_kernels.hip costs per form of recording its four comparison bits (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_swcell tools/ubench_swcell.hip ; run on the GPU box.
//   form 0: the shipped cell (v_addc_co_u32 appends each bit to a VGPR word)
//   form 1: the borrow masks copied to SGPR pairs (s_mov_b64), never stored      -- the SALU cost alone
//   form 2: form 1 + two s_store_dwordx4 per cell into a per-wave stream          -- bit planes written by the scalar unit
//   form 3: bits dropped                                                          -- lower bound
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int K = 10;

template <int FORM>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint8_t *planes, size_t wave_stride, int iters, unsigned long long *clk)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t Hp[K], F[K], b2[K];
    for (int i = 0; i < K; ++i) { Hp[i] = (1u << 30) + tid * 3 + i; F[i] = (1u << 30) - 100 + i; b2[i] = (tid >> i) & 3; }
    uint32_t wm = 200, wmm = (uint32_t)-150, wo = (uint32_t)-260, we = (uint32_t)-11;
    asm volatile("" : "+v"(wm), "+v"(wmm), "+v"(wo), "+v"(we));
    uint32_t hl = 1u << 30, e = (1u << 30) - 50, hd = 1u << 30, base1 = tid & 3, acc = 0;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    uint8_t *wp = planes + (size_t)wave * wave_stride;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        uint8_t *sp = wp + (size_t)it * (K * 32);
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            uint32_t e11, f11 = F[kk], h11, t_exth, t_openh, t_sc, t_m11, t_extv, t_openv, t_h1, t_d;
            if (FORM == 0) {
                asm volatile(
                    "v_cmp_eq_u32 vcc, %[b1], %[b2]\n\t"
                    "v_add_u32 %[exth], %[e], %[wext]\n\t"
                    "v_add_u32 %[openh], %[hl], %[wopen]\n\t"
                    "v_cndmask_b32 %[sc], %[wmm], %[wm], vcc\n\t"
                    "v_sub_co_u32 %[d], vcc, %[exth], %[openh]\n\t"
                    "v_add_u32 %[m11], %[hd], %[sc]\n\t"
                    "v_add_u32 %[extv], %[f11], %[wext]\n\t"
                    "v_cndmask_b32 %[e11], %[exth], %[openh], vcc\n\t"
                    "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                    "v_add_u32 %[openv], %[hp], %[wopen]\n\t"
                    "v_sub_co_u32 %[d], vcc, %[extv], %[openv]\n\t"
                    "s_nop 1\n\t"
                    "v_cndmask_b32 %[f11], %[extv], %[openv], vcc\n\t"
                    "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                    "v_sub_co_u32 %[d], vcc, %[m11], %[e11]\n\t"
                    "s_nop 1\n\t"
                    "v_cndmask_b32 %[h1], %[m11], %[e11], vcc\n\t"
                    "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                    "v_sub_co_u32 %[d], vcc, %[h1], %[f11]\n\t"
                    "s_nop 1\n\t"
                    "v_cndmask_b32 %[h11], %[h1], %[f11], vcc\n\t"
                    "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc"
                    : [acc] "+v"(acc), [e11] "=&v"(e11), [f11] "+v"(f11), [h11] "=&v"(h11), [exth] "=&v"(t_exth), [openh] "=&v"(t_openh),
                      [sc] "=&v"(t_sc), [m11] "=&v"(t_m11), [extv] "=&v"(t_extv), [openv] "=&v"(t_openv), [h1] "=&v"(t_h1), [d] "=&v"(t_d)
                    : [b1] "v"(base1), [b2] "v"(b2[kk]), [e] "v"(e), [hl] "v"(hl), [hd] "v"(hd), [hp] "v"(Hp[kk]), [wext] "v"(we),
                      [wopen] "v"(wo), [wm] "v"(wm), [wmm] "v"(wmm)
                    : "vcc");
            }
            else if (FORM == 6) {
                asm volatile(
                    "v_cmp_eq_u32 vcc, %[b1], %[b2]\n\t"
                    "v_add_u32 %[exth], %[e], %[wext]\n\t"
                    "v_add_u32 %[openh], %[hl], %[wopen]\n\t"
                    "v_cndmask_b32 %[sc], %[wmm], %[wm], vcc\n\t"
                    "v_sub_co_u32 %[d], vcc, %[exth], %[openh]\n\t"
                    "v_add_u32 %[m11], %[hd], %[sc]\n\t"
                    "v_add_u32 %[extv], %[f11], %[wext]\n\t"
                    "v_cndmask_b32 %[e11], %[exth], %[openh], vcc\n\t"
                    "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                    "v_add_u32 %[openv], %[hp], %[wopen]\n\t"
                    "v_sub_co_u32 %[d], vcc, %[extv], %[openv]\n\t"
                    "v_cndmask_b32 %[f11], %[extv], %[openv], vcc\n\t"
                    "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                    "v_sub_co_u32 %[d], vcc, %[m11], %[e11]\n\t"
                    "v_cndmask_b32 %[h1], %[m11], %[e11], vcc\n\t"
                    "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                    "v_sub_co_u32 %[d], vcc, %[h1], %[f11]\n\t"
                    "v_cndmask_b32 %[h11], %[h1], %[f11], vcc\n\t"
                    "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc"
                    : [acc] "+v"(acc), [e11] "=&v"(e11), [f11] "+v"(f11), [h11] "=&v"(h11), [exth] "=&v"(t_exth), [openh] "=&v"(t_openh),
                      [sc] "=&v"(t_sc), [m11] "=&v"(t_m11), [extv] "=&v"(t_extv), [openv] "=&v"(t_openv), [h1] "=&v"(t_h1), [d] "=&v"(t_d)
                    : [b1] "v"(base1), [b2] "v"(b2[kk]), [e] "v"(e), [hl] "v"(hl), [hd] "v"(hd), [hp] "v"(Hp[kk]), [wext] "v"(we),
                      [wopen] "v"(wo), [wm] "v"(wm), [wmm] "v"(wmm)
                    : "vcc");
            }
            else {
                // the same cell; each borrow mask goes to an SGPR pair right after the select that consumes it
#define CELL(M1, M2, M3, M4, TAIL)                                                                                                         \
    asm volatile("v_cmp_eq_u32 vcc, %[b1], %[b2]\n\t"                                                                                      \
                 "v_add_u32 %[exth], %[e], %[wext]\n\t"                                                                                    \
                 "v_add_u32 %[openh], %[hl], %[wopen]\n\t"                                                                                 \
                 "v_cndmask_b32 %[sc], %[wmm], %[wm], vcc\n\t"                                                                             \
                 "v_sub_co_u32 %[d], vcc, %[exth], %[openh]\n\t"                                                                           \
                 "v_add_u32 %[m11], %[hd], %[sc]\n\t"                                                                                      \
                 "v_add_u32 %[extv], %[f11], %[wext]\n\t"                                                                                  \
                 "v_cndmask_b32 %[e11], %[exth], %[openh], vcc\n\t" M1                                                                     \
                 "v_add_u32 %[openv], %[hp], %[wopen]\n\t"                                                                                 \
                 "v_sub_co_u32 %[d], vcc, %[extv], %[openv]\n\t"                                                                           \
                 "s_nop 1\n\t"                                                                                                             \
                 "v_cndmask_b32 %[f11], %[extv], %[openv], vcc\n\t" M2                                                                     \
                 "v_sub_co_u32 %[d], vcc, %[m11], %[e11]\n\t"                                                                              \
                 "s_nop 1\n\t"                                                                                                             \
                 "v_cndmask_b32 %[h1], %[m11], %[e11], vcc\n\t" M3                                                                         \
                 "v_sub_co_u32 %[d], vcc, %[h1], %[f11]\n\t"                                                                               \
                 "s_nop 1\n\t"                                                                                                             \
                 "v_cndmask_b32 %[h11], %[h1], %[f11], vcc\n\t" M4 TAIL                                                                    \
                 : [e11] "=&v"(e11), [f11] "+v"(f11), [h11] "=&v"(h11), [exth] "=&v"(t_exth), [openh] "=&v"(t_openh), [sc] "=&v"(t_sc),    \
                   [m11] "=&v"(t_m11), [extv] "=&v"(t_extv), [openv] "=&v"(t_openv), [h1] "=&v"(t_h1), [d] "=&v"(t_d)                      \
                 : [b1] "v"(base1), [b2] "v"(b2[kk]), [e] "v"(e), [hl] "v"(hl), [hd] "v"(hd), [hp] "v"(Hp[kk]), [wext] "v"(we),            \
                   [wopen] "v"(wo), [wm] "v"(wm), [wmm] "v"(wmm), [sp] "s"(sp), [off] "n"(0)                                               \
                 : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "memory")
                if (FORM == 1) {
                    CELL("s_mov_b64 s[20:21], vcc\n\t", "s_mov_b64 s[22:23], vcc\n\t", "s_mov_b64 s[24:25], vcc\n\t", "s_mov_b64 s[26:27], vcc\n\t", "");
                }
                else if (FORM == 2) {
                    uint8_t *cp = sp + kk * 32;
                    asm volatile("" : "+s"(cp));
                    sp = cp - kk * 32;
                    asm volatile("" ::: "memory");
                    // two 16-byte scalar stores per cell (the previous cell's stores must have left the SGPRs: s_waitcnt lgkmcnt(0) first)
                    CELL("s_waitcnt lgkmcnt(0)\n\ts_mov_b64 s[20:21], vcc\n\t", "s_mov_b64 s[22:23], vcc\n\t", "s_mov_b64 s[24:25], vcc\n\t",
                         "s_mov_b64 s[26:27], vcc\n\t", "s_store_dwordx4 s[20:23], %[sp], 0x0\n\ts_store_dwordx4 s[24:27], %[sp], 0x10\n\t");
                    sp += 32;
                }
                else if (FORM == 4) {  // form 2 without the wait (timing only: is it the wait or the stores themselves)
                    uint8_t *cp = sp + kk * 32;
                    asm volatile("" : "+s"(cp));
                    sp = cp - kk * 32;
                    asm volatile("" ::: "memory");
                    CELL("s_mov_b64 s[20:21], vcc\n\t", "s_mov_b64 s[22:23], vcc\n\t", "s_mov_b64 s[24:25], vcc\n\t",
                         "s_mov_b64 s[26:27], vcc\n\t", "s_store_dwordx4 s[20:23], %[sp], 0x0\n\ts_store_dwordx4 s[24:27], %[sp], 0x10\n\t");
                    sp += 32;
                }
                else if (FORM == 5) {  // one store per cell (half the bits): does the cost scale with the store count
                    uint8_t *cp = sp + kk * 32;
                    asm volatile("" : "+s"(cp));
                    sp = cp - kk * 32;
                    asm volatile("" ::: "memory");
                    CELL("s_mov_b64 s[20:21], vcc\n\t", "s_mov_b64 s[22:23], vcc\n\t", "s_mov_b64 s[24:25], vcc\n\t",
                         "s_mov_b64 s[26:27], vcc\n\t", "s_store_dwordx4 s[20:23], %[sp], 0x0\n\t");
                    sp += 32;
                }
                else {
                    CELL("", "", "", "", "");
                }
            }
            hd = Hp[kk];
            Hp[kk] = h11;
            F[kk] = f11;
            hl = h11;
            e = e11;
        }
        if (FORM == 2 || FORM == 4 || FORM == 5) sp -= K * 32;
        base1 = (base1 + 1) & 3;
    }
    if (FORM == 2 || FORM == 4 || FORM == 5) asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = acc ^ hl ^ e;
    for (int i = 0; i < K; ++i) s ^= Hp[i] ^ F[i];
    out[tid] = s;
    if (tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int FORM> void run(const char *name, uint32_t *out, uint8_t *planes, size_t plane_bytes, unsigned long long *clk)
{
    printf("%-44s", name); fflush(stdout);
    for (int wps : {1, 2, 4}) {
        const int threads = 256, blocks = 256 * wps, iters = 400;
        const size_t waves = (size_t)blocks * 4, stride = (size_t)iters * K * 32;
        if (waves * stride > plane_bytes) { printf("  (buffer too small)"); continue; }
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(threads), 0, 0, out, planes, stride, 10, clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(threads), 0, 0, out, planes, stride, iters, clk); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);
        const double cells = (double)waves * iters * K;          // wave-cells
        const double cyc = ms * 1e-3 * ghz * 1e9;
        printf("  w%d: %5.1f cyc/cell/SIMD (%.2f GHz, %.0f GCUPS-equivalent)", wps, cyc / (cells / (256.0 * 4)), ghz, cells * 64 / (ms * 1e-3) / 1e9);
    }
    printf("\n"); fflush(stdout);
}

int main()
{
    uint32_t *out; uint8_t *planes; unsigned long long *clk;
    const size_t plane_bytes = (size_t)4096 * 400 * K * 32 + 4096;
    CHECK(hipMalloc(&out, 256 * 4 * 256 * sizeof(uint32_t))); CHECK(hipMalloc(&planes, plane_bytes)); CHECK(hipMalloc(&clk, 16));
    run<0>("shipped cell (4 v_addc_co_u32)", out, planes, plane_bytes, clk);
    run<6>("shipped cell without its three s_nop (timing)", out, planes, plane_bytes, clk);
    run<3>("bits dropped (lower bound)", out, planes, plane_bytes, clk);
    run<1>("borrow masks to SGPR pairs (4 s_mov_b64)", out, planes, plane_bytes, clk);
    run<2>("... + 2 s_store_dwordx4 per cell", out, planes, plane_bytes, clk);
    run<4>("... the same without s_waitcnt (timing only)", out, planes, plane_bytes, clk);
    run<5>("... one s_store_dwordx4 per cell, no wait", out, planes, plane_bytes, clk);
    // check that scalar stores landed: the last cell of wave 0's last iteration holds four masks (not all zero in general)
    uint64_t tail[4]; CHECK(hipMemcpy(tail, planes + (size_t)(400 - 1) * K * 32 + (K - 1) * 32, 32, hipMemcpyDeviceToHost));
    printf("last cell's masks of wave 0: %016llx %016llx %016llx %016llx\n", (unsigned long long)tail[0], (unsigned long long)tail[1],
           (unsigned long long)tail[2], (unsigned long long)tail[3]);
    return 0;
}
