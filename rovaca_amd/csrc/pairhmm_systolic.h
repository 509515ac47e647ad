// pairhmm_systolic.h -- device code shared by the gfx950 PairHMM kernels: the cross-lane helpers, the per-row constants, and the
// sweep of ONE workgroup of the single-pass systolic kernels as a function template of the rows per lane K.  Included by
// pairhmm_kernels.hip (one launch per K: the whole launch is one length class) and pairhmm_kernels_ragged.hip (one launch for several
// length classes: every workgroup picks the K of its class).  See pairhmm_kernels.hip for what is computed and how it maps to CDNA4.
#ifndef ROVACA_AMD_PAIRHMM_SYSTOLIC_H_
#define ROVACA_AMD_PAIRHMM_SYSTOLIC_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "pairhmm_device.h"

namespace phmm
{

// ----------------------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_row_shr1(float old, float src)
{
    // lane i of each 16-lane row receives lane i-1; lane 0 of the row keeps `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), 0x111, 0xF, 0xF, false));
}

__device__ __forceinline__ double dpp_row_shr1(double old, double src)
{
    union { double d; int i[2]; } o, s, r;
    o.d = old;
    s.d = src;
    r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], 0x111, 0xF, 0xF, false);
    r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], 0x111, 0xF, 0xF, false);
    return r.d;
}

__device__ __forceinline__ float dpp_wave_shr1(float old, float src)
{
    // lane i of the wave receives lane i-1 (crossing the 16-lane rows); lane 0 keeps `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), 0x138, 0xF, 0xF, false));
}

__device__ __forceinline__ double dpp_wave_shr1(double old, double src)
{
    union { double d; int i[2]; } o, s, r;
    o.d = old;
    s.d = src;
    r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], 0x138, 0xF, 0xF, false);
    r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], 0x138, 0xF, 0xF, false);
    return r.d;
}

// 32 lanes per pair (two pairs per wave): lanes 1..31 and 33..63 receive lane i-1, lanes 0 and 32 keep `old`.  Two DPP moves:
// wave_shr:1 on rows 0, 1 and 3 (lane 16 takes lane 15, lane 48 lane 47; row 2 is left alone), then row_shr:1 on row 2, whose
// first lane -- lane 32, the first of the second pair -- has no source inside the row and keeps what it had.
__device__ __forceinline__ float dpp_half_shr1(float old, float src)
{
    const int r = __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), 0x138, 0xB, 0xF, false);
    return __int_as_float(__builtin_amdgcn_update_dpp(r, __float_as_int(src), 0x111, 0x4, 0xF, false));
}

__device__ __forceinline__ double dpp_half_shr1(double old, double src)
{
    union { double d; int i[2]; } o, s, r;
    o.d = old;
    s.d = src;
    r.i[0] = __builtin_amdgcn_update_dpp(__builtin_amdgcn_update_dpp(o.i[0], s.i[0], 0x138, 0xB, 0xF, false), s.i[0], 0x111, 0x4, 0xF, false);
    r.i[1] = __builtin_amdgcn_update_dpp(__builtin_amdgcn_update_dpp(o.i[1], s.i[1], 0x138, 0xB, 0xF, false), s.i[1], 0x111, 0x4, 0xF, false);
    return r.d;
}

template <int G, typename T>
__device__ __forceinline__ T dpp_shr1(T old, T src)
{
    if constexpr (G == 64)
        return dpp_wave_shr1(old, src);
    else if constexpr (G == 32)
        return dpp_half_shr1(old, src);
    else
        return dpp_row_shr1(old, src);
}

// read symbol -> plane class: A C G T N other  (one-hot semantics of pairhmm_internal.cpp:27-41)
__device__ __forceinline__ uint32_t read_class(uint8_t b)
{
    return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : b == 'N' ? 4u : 5u;
}

template <typename T> struct Num;
template <> struct Num<float>
{
    __device__ static float init_const() { return __builtin_ldexpf(1.0f, 120); }  // core/context.h:148
    __device__ static float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    __device__ static float div3(float x) { return __fdiv_rn(x, 3.0f); }
};
template <> struct Num<double>
{
    __device__ static double init_const() { return __builtin_ldexp(1.0, 1020); }  // core/context.h:106
    __device__ static double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
    __device__ static double div3(double x) { return __ddiv_rn(x, 3.0); }
};

// ----------------------------------------------------------------------------------------------------
// the systolic kernel
// ----------------------------------------------------------------------------------------------------
template <typename T, int K, bool UG>
struct Rows
{
    // per-row constants.  UG is the Rovaca case: the gap-continuation penalty is one constant (read_record.cpp:205-210
    // fixes gcp = 10) and the deletion gap-open array is the insertion one (pairhmm_internal.cpp:425), so p_gapm and
    // p_xx are shared by all rows and p_my is p_mx.  p_yy stays per row because a padding row needs p_yy = 1.
    T p_mm[K], p_mx[K], p_yy[K], lo[K], hi[K];
    T p_my_r[UG ? 1 : K];
    T p_gapm_r[UG ? 1 : K], p_xx_r[UG ? 1 : K];
    __device__ __forceinline__ T p_my(int k) const { return UG ? p_mx[k] : p_my_r[UG ? 0 : k]; }
    __device__ __forceinline__ T p_gapm(int k) const { return p_gapm_r[UG ? 0 : k]; }
    __device__ __forceinline__ T p_xx(int k) const { return p_xx_r[UG ? 0 : k]; }
};

// Occupancy target (waves per SIMD) handed to the register allocator, from measurements per K (tools/bench_by_k.py): the UG
// kernel at K = 10 (150-bp reads) needs 131 VGPRs unconstrained, 3 more than the 128 that admit a fourth wave; K = 11 still
// wins at 4 waves with a few spills outside the steady loop; K = 12..14 run best at 3 waves.
#define PHMM_MIN_WAVES(T, K, UG) (sizeof(T) == 8 ? 2 : ((UG) && (K) <= 11) ? 4 : ((UG) && (K) <= 14) ? 3 : 1)

// G = lanes per pair.  16: four pairs per wave (DPP row_shr), the throughput shape.  32: two pairs per wave -- half the rows per lane,
// so a read of 193-256 bases needs 7-8 rows (about 100 VGPRs) instead of 13-16 (155-200): the shape that lets EVERY length class of a
// chunk-sized launch run under one 128-register budget (pairhmm_kernels_ragged.hip), at 1.16x the instructions.  64: one pair per wave (DPP wave_shr) in
// one-wave workgroups, the latency shape for launches with so few pairs that four-per-wave would leave most SIMDs empty (a
// single region): a quarter of the rows per lane makes the sweep of a pair about twice as short.
template <int G> constexpr int systolic_block_threads() { return G == 64 ? 64 : kBlockThreads; }

// which workgroup of the processing order this hardware workgroup takes (KernelArgs::xcd_run_log2); wave-uniform scalar work
__device__ __forceinline__ uint32_t xcd_local_workgroup(uint32_t b, uint32_t run_log2)
{
    if (run_log2 == 0) return b;
    const uint32_t xcd = b & 7u, i = b >> 3;
    return ((((i >> run_log2) << 3) + xcd) << run_log2) | (i & ((1u << run_log2) - 1u));
}

// The sweep of one workgroup, shared by the one-K kernels (pairhmm_kernels.hip: the whole launch is one rows-per-lane class) and the
// ragged kernels (pairhmm_kernels_ragged.hip: a launch holds several classes, every workgroup picks its K).  `wg` is the workgroup's
// place in the processing order of its class, [pos_first, pos_first + count) the class's range of a.order, and the two rescue pointers
// the queue of that class.
template <typename T, int K, bool UG, int G, int BT = systolic_block_threads<G>()>
__device__ __forceinline__ void systolic_sweep(const KernelArgs &a, uint32_t wg, uint32_t pos_first, uint32_t count, uint32_t *rescue_list,
                                               uint32_t *rescue_count)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & (uint32_t)(G - 1);                      // lane inside the pair's G-lane array
    const uint32_t grp_in_block = threadIdx.x / (uint32_t)G;                    // 0..15 (G = 16), 0 (G = 64)
    const uint32_t gidx = wg * (BT / G) + grp_in_block;  // BT: threads of the workgroup this sweep runs in
    const bool valid = gidx < count;
    const uint32_t pair = a.order[valid ? pos_first + gidx : 0];
#define PHMM_SWEEP_RESCUE_LIST rescue_list
#define PHMM_SWEEP_RESCUE_COUNT rescue_count
#include "pairhmm_systolic_sweep.inc"
#undef PHMM_SWEEP_RESCUE_LIST
#undef PHMM_SWEEP_RESCUE_COUNT
}

// ----------------------------------------------------------------------------------------------------
// the systolic kernel for batches with ONE gap-open quality (PCR-free data: --pcr-indel-model NONE leaves every base at 45,
// read_record.cpp:191-196; BASELINE's synthetic batches use 40 throughout) on top of the Rovaca case above.  Then p_mm and p_mx are
// the same for every row, and M[i][j] * p_my_i (needed by Y[i][j+1]) is the very number M[i][j] * p_mx_{i+1} (needed by X[i+1][j]):
// pass 2 computes it, the next step's pass 1 uses it again.  K - 1 multiplications fewer per step (101 VALU instructions instead of
// 110 at K = 10), 2K - (K - 1) - 2 vector registers fewer, the same bits -- every product is the same operation on the same operands,
// rounded once.  A copy of the kernel above rather than another template switch in it: that kernel's register allocation is tuned
// and sensitive to any change of its text ("Register numbers matter").
// ----------------------------------------------------------------------------------------------------
// occupancy targets of this kernel, measured per K (tools/bench_by_k.py): it has 2K - 1 fewer live registers than the general one, so
// K = 12 still fits four waves (+1 %) and K = 15 three (5 670 -> 6 050 GCUPS); K = 16 at three waves spills inside the loop (4 980)
#define PHMM_UGO_MIN_WAVES(T, K) (sizeof(T) == 8 ? 2 : (K) <= 12 ? 4 : (K) <= 15 ? 3 : 1)
template <typename T, int K, int G, int BT = systolic_block_threads<G>()>
__device__ __forceinline__ void systolic_ugo_sweep(const KernelArgs &a, uint32_t wg, uint32_t pos_first, uint32_t count, uint32_t *rescue_list,
                                                   uint32_t *rescue_count)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & (uint32_t)(G - 1);                      // lane inside the pair's G-lane array
    const uint32_t grp_in_block = threadIdx.x / (uint32_t)G;                    // 0..15 (G = 16), 0 (G = 64)
    const uint32_t gidx = wg * (BT / G) + grp_in_block;  // BT: threads of the workgroup this sweep runs in
    const bool valid = gidx < count;
    const uint32_t pair = a.order[valid ? pos_first + gidx : 0];
#define PHMM_SWEEP_RESCUE_LIST rescue_list
#define PHMM_SWEEP_RESCUE_COUNT rescue_count
#include "pairhmm_systolic_ugo_sweep.inc"
#undef PHMM_SWEEP_RESCUE_LIST
#undef PHMM_SWEEP_RESCUE_COUNT
}

}  // namespace phmm
#endif
