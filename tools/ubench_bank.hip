// ubench_bank.hip -- does the VGPR bank (register number mod 4) of VALU source operands change the issue rate on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define REP8(S) S S S S S S S S
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131"
template <int MODE>
__global__ void k(float* out, int iters, unsigned long long* clk)
{
    asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 1.0\n v_mov_b32 v102, 1.0\n v_mov_b32 v103, 1.0\n v_mov_b32 v104, 1.0\n v_mov_b32 v105, 1.0\n v_mov_b32 v106, 1.0\n v_mov_b32 v107, 1.0\n"
                 "v_mov_b32 v108, 1.0\n v_mov_b32 v109, 1.0\n v_mov_b32 v110, 1.0\n v_mov_b32 v111, 1.0\n v_mov_b32 v112, 0.5\n v_mov_b32 v113, 0.5\n v_mov_b32 v114, 0.5\n v_mov_b32 v115, 0.5\n"
                 "v_mov_b32 v116, 0.5\n v_mov_b32 v117, 0.5\n v_mov_b32 v118, 0.5\n v_mov_b32 v119, 0.5\n v_mov_b32 v120, 0.5\n v_mov_b32 v121, 0.5\n v_mov_b32 v122, 0.5\n v_mov_b32 v123, 0.5\n" ::: CLOB);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        // 8 instructions per statement, 8 statements
        if (MODE == 0) { REP8(asm volatile("v_fma_f32 v100, v100, v113, v118\n v_fma_f32 v101, v101, v114, v119\n v_fma_f32 v102, v102, v115, v116\n v_fma_f32 v103, v103, v112, v117\n v_fma_f32 v104, v104, v113, v118\n v_fma_f32 v105, v105, v114, v119\n v_fma_f32 v106, v106, v115, v116\n v_fma_f32 v107, v107, v112, v117" ::: CLOB);) }  // 3 sources in 3 different banks
        if (MODE == 1) { REP8(asm volatile("v_fma_f32 v100, v100, v112, v116\n v_fma_f32 v101, v101, v113, v117\n v_fma_f32 v102, v102, v114, v118\n v_fma_f32 v103, v103, v115, v119\n v_fma_f32 v104, v104, v112, v116\n v_fma_f32 v105, v105, v113, v117\n v_fma_f32 v106, v106, v114, v118\n v_fma_f32 v107, v107, v115, v119" ::: CLOB);) }  // all 3 sources in the SAME bank
        if (MODE == 2) { REP8(asm volatile("v_fmac_f32 v100, v113, v118\n v_fmac_f32 v101, v114, v119\n v_fmac_f32 v102, v115, v116\n v_fmac_f32 v103, v112, v117\n v_fmac_f32 v104, v113, v118\n v_fmac_f32 v105, v114, v119\n v_fmac_f32 v106, v115, v116\n v_fmac_f32 v107, v112, v117" ::: CLOB);) }  // fmac, different banks
        if (MODE == 3) { REP8(asm volatile("v_fmac_f32 v100, v112, v116\n v_fmac_f32 v101, v113, v117\n v_fmac_f32 v102, v114, v118\n v_fmac_f32 v103, v115, v119\n v_fmac_f32 v104, v112, v116\n v_fmac_f32 v105, v113, v117\n v_fmac_f32 v106, v114, v118\n v_fmac_f32 v107, v115, v119" ::: CLOB);) }  // fmac, same bank
        if (MODE == 4) { REP8(asm volatile("v_mul_f32 v100, v113, v118\n v_mul_f32 v101, v114, v119\n v_mul_f32 v102, v115, v116\n v_mul_f32 v103, v112, v117\n v_mul_f32 v104, v113, v118\n v_mul_f32 v105, v114, v119\n v_mul_f32 v106, v115, v116\n v_mul_f32 v107, v112, v117" ::: CLOB);) }  // mul 2 src, different banks
        if (MODE == 5) { REP8(asm volatile("v_mul_f32 v100, v112, v116\n v_mul_f32 v101, v113, v117\n v_mul_f32 v102, v114, v118\n v_mul_f32 v103, v115, v119\n v_mul_f32 v104, v112, v116\n v_mul_f32 v105, v113, v117\n v_mul_f32 v106, v114, v118\n v_mul_f32 v107, v115, v119" ::: CLOB);) }  // mul 2 src, same bank
        if (MODE == 6) { REP8(asm volatile("v_fmac_f32 v100, v107, v118\n v_fmac_f32 v101, v100, v119\n v_fmac_f32 v102, v101, v116\n v_fmac_f32 v103, v102, v117\n v_fmac_f32 v104, v103, v118\n v_fmac_f32 v105, v104, v119\n v_fmac_f32 v106, v105, v116\n v_fmac_f32 v107, v106, v117" ::: CLOB);) }  // each fmac reads the previous result (dependent chain)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s;
    asm volatile("v_add_f32 %0, v100, v101\n v_add_f32 %0, %0, v102\n v_add_f32 %0, %0, v103\n v_add_f32 %0, %0, v104\n v_add_f32 %0, %0, v105\n v_add_f32 %0, %0, v106\n v_add_f32 %0, %0, v107" : "=v"(s) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int MODE> void run(const char* name, float* out, unsigned long long* clk)
{
    const int iters = 2000, per_iter = 64;
    printf("%-44s", name); fflush(stdout);
    for (int wps : {1, 2, 4}) {
        int threads = 256, blocks = 256 * wps;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10, clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        double ghz = (double)h[0] / ((double)h[1] * 10.0);
        double winstr = (double)blocks * 4 * iters * per_iter;
        double cyc = ms * 1e-3 * ghz * 1e9;
        printf("  w%d: %4.2f cyc/instr (%.2f GHz)", wps, cyc / (winstr / (256.0 * 4)), ghz);
    }
    printf("\n"); fflush(stdout);
}
int main()
{
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float))); CHECK(hipMalloc(&clk, 16));
    run<0>("v_fma_f32 3 src, 3 different banks", out, clk);
    run<1>("v_fma_f32 3 src, same bank", out, clk);
    run<2>("v_fmac_f32 (acc + 2 src), different banks", out, clk);
    run<3>("v_fmac_f32 (acc + 2 src), same bank", out, clk);
    run<4>("v_mul_f32 2 src, different banks", out, clk);
    run<5>("v_mul_f32 2 src, same bank", out, clk);
    run<6>("v_fmac_f32 dependent chain", out, clk);
    return 0;
}
