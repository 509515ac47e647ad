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
    const uint32_t r = a.pair_read[pair], h = a.pair_hap[pair];
    const uint32_t R = a.read_len[r], H = a.hap_len[h];
    const uint32_t roff = a.read_off[r];
    const uint32_t W = planes_words(H);
    const T *__restrict__ ph2pr = reinterpret_cast<const T *>(a.ph2pr);
    const T *__restrict__ mmtab = reinterpret_cast<const T *>(a.mm_table);

    // ---- stage this pair's haplotype planes into LDS (kPlanes * W words, 16 lanes copy) ----
    uint32_t *pl = lds + grp_in_block * a.lds_words_per_group;
    {
        const uint32_t *src = a.planes + a.plane_off[h];
        for (uint32_t i = lane; i < kPlanes * W; i += G) pl[i] = src[i];
    }

    // ---- per-row constants (pairhmm_internal.cpp:466-480 == core/avx512_s.cpp:90-101) ----
    Rows<T, K, UG> P;
    T M[K], X[K], Y[K];
    uint32_t cls_lo = 0, cls_hi = 0;  // 3-bit plane class per row, packed (K <= 16 -> 48 bits)
    const T init_y = Num<T>::init_const() / static_cast<T>(H);  // rovaca/avx_512_float.cc:32
    const int pad = G * K - static_cast<int>(R);                  // rows [0, pad) are boundary replicas
    if (UG) {
        uint32_t c0 = a.read_gcp[roff] & 127u;
        P.p_gapm_r[0] = static_cast<T>(1.0) - ph2pr[c0];
        P.p_xx_r[0] = ph2pr[c0];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int row = static_cast<int>(lane) * K + k - pad;
        uint32_t cls = 5;
        if (row >= 0) {
            const uint32_t at = roff + static_cast<uint32_t>(row);
            const uint32_t _i = a.read_ins[at] & 127u, _d = a.read_del[at] & 127u, _c = a.read_gcp[at] & 127u,
                           _q = a.read_qual[at] & 127u;
            const uint32_t mx = _i > _d ? _i : _d, mn = _i > _d ? _d : _i;
            P.p_mm[k] = mmtab[((mx * (mx + 1)) >> 1) + mn];  // core/context.h:162-173
            P.p_mx[k] = ph2pr[_i];
            if (!UG) P.p_my_r[UG ? 0 : k] = ph2pr[_d];
            P.p_yy[k] = ph2pr[_c];
            if (!UG) {
                P.p_gapm_r[UG ? 0 : k] = static_cast<T>(1.0) - ph2pr[_c];
                P.p_xx_r[UG ? 0 : k] = ph2pr[_c];
            }
            const T dist = ph2pr[_q];
            P.hi[k] = static_cast<T>(1.0) - dist;
            P.lo[k] = Num<T>::div3(dist);
            cls = read_class(a.read_bases[at]);
            M[k] = 0;
            X[k] = 0;
            Y[k] = 0;
        }
        else {
            // boundary replica: M = X = 0 and Y = init at every column (prior 0 kills M, p_yy = 1 holds Y)
            P.p_mm[k] = 0;
            P.p_mx[k] = 0;
            if (!UG) P.p_my_r[UG ? 0 : k] = 0;
            P.p_yy[k] = 1;
            if (!UG) {
                P.p_gapm_r[UG ? 0 : k] = 0;
                P.p_xx_r[UG ? 0 : k] = 0;
            }
            P.hi[k] = 0;
            P.lo[k] = 0;
            M[k] = 0;
            X[k] = 0;
            Y[k] = init_y;
        }
        if (k < 10)
            cls_lo |= cls << (3 * k);
        else
            cls_hi |= cls << (3 * (k - 10));
    }
    __syncthreads();  // planes visible (each group only reads its own, but keep the block-wide fence simple)

    // ---- the sweep ----
    const uint32_t Tg = valid ? H + (G - 1) : 0;  // last step of this group
    uint32_t Tmax = Tg;                           // wave-uniform loop bound
    if (G <= 16) Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 16));
    if (G <= 32) Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 32));
    Tmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmax);

    uint32_t Tmin = Tg;  // until here every lane of the wave is busy once the pipeline is full
    if (G <= 16) Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 16));
    if (G <= 32) Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 32));
    Tmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmin);

    uint32_t mask[K];
    // The row above slot 0 travels between lanes in two register sets that swap roles every step: one holds that
    // row at the current column (written by the DPP shift of this step), the other at the previous column (written
    // one step earlier).  Lane 0 never receives anything, so its copies keep the boundary row (0, 0, init) for ever;
    // the DPP's `old` operand is the destination itself, which costs no instruction.  Before a lane's first step
    // both sets hold column 0 of the row above: Y[0][0] = init when that row is the boundary row or a replica of it
    // (lane * K <= pad), and 0 when it is a read row (M[i][0] = X[i][0] = Y[i][0] = 0).
    const T y_col0 = (static_cast<int>(lane) * K <= pad) ? init_y : static_cast<T>(0);
    T aM = 0, aX = 0, aY = y_col0, bM = 0, bX = 0, bY = y_col0;
    T sumM = 0, sumX = 0;
    const uint32_t wmax = W - 2;

    auto reload = [&](uint32_t t) {
        // every row's 32-column window of its match plane, starting at this lane's next column
        const uint32_t jn = t > lane ? t - lane : 1u;  // 1-based
        uint32_t w = (jn - 1) >> 5;
        const uint32_t s = (jn - 1) & 31u;
        w = w < wmax ? w : wmax;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t cls = k < 10 ? (cls_lo >> (3 * k)) & 7u : (cls_hi >> (3 * (k - 10))) & 7u;
            const uint32_t *p = pl + cls * W + w;
            const uint32_t w0 = p[0], w1 = p[1];
            mask[k] = (w0 << s) | ((w1 >> 1) >> (31u - s));  // bits [s, s+32) of w0:w1, MSB first (s = 0..31)
        }
    };

    // one anti-diagonal.  (dM, dX, dY): row above slot 0 at the previous column; (uM0, uX0, uY0): receives it at this column.
    auto step = [&](T &dM, T &dX, T &dY, T &uM0, T &uX0, T &uY0) {
        uM0 = dpp_shr1<G>(uM0, M[K - 1]);
        uX0 = dpp_shr1<G>(uX0, X[K - 1]);
        uY0 = dpp_shr1<G>(uY0, Y[K - 1]);
        // Pass 1, bottom row first: Y' and M' read only the previous anti-diagonal, so walking upwards lets every
        // value be overwritten in place (row k's old M/X/Y are last read by row k+1, which is already done).
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
            uint32_t nm;
            const bool hit = __builtin_uadd_overflow(mask[k], mask[k], &nm);
            mask[k] = nm;
            const T prior = hit ? P.hi[k] : P.lo[k];
            Y[k] = Num<T>::fma_(Y[k], P.p_yy[k], M[k] * P.p_my(k));
            const T gM = k ? M[k ? k - 1 : 0] : dM, gX = k ? X[k ? k - 1 : 0] : dX, gY = k ? Y[k ? k - 1 : 0] : dY;
            T tt = gM * P.p_mm[k];
            tt = Num<T>::fma_(gX, P.p_gapm(k), tt);
            tt = Num<T>::fma_(gY, P.p_gapm(k), tt);
            M[k] = tt * prior;
            // Keep the rows in program order: measured on MI355X, hipcc's own interleavings of this block run 5-25 %
            // slower (tools/try_variants.sh), whatever the scheduling strategy.
            __builtin_amdgcn_sched_barrier(0);
        }
        // Pass 2, top row first: the insertion state chains down the column through the NEW M and X.
        T uM = uM0, uX = uX0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            X[k] = Num<T>::fma_(uX, P.p_xx(k), uM * P.p_mx[k]);
            uM = M[k];
            uX = X[k];
        }
        __builtin_amdgcn_sched_barrier(0);
        sumM = sumM + M[K - 1];  // rovaca/avx_512_float.cc:166-171 (only lane 15 is read back)
        sumX = sumX + X[K - 1];
    };
    auto step_t = [&](uint32_t t) {
        if (t & 1u)
            step(aM, aX, aY, bM, bX, bY);
        else
            step(bM, bX, bY, aM, aX, aY);
    };

    uint32_t t = 1;
    // fill: lane l starts at step l + 1
    const uint32_t Tfill = Tmax < (uint32_t)(G - 1) ? Tmax : (uint32_t)(G - 1);
    for (; t <= Tfill; ++t) {
        if ((t & 31u) == 1u) reload(t);
        if ((t > lane) && (t <= Tg)) step_t(t);
    }
    // steady state: every lane of all four pairs has work, no per-lane predicate
    if (t <= Tmin && !(t & 1u)) {
        step(bM, bX, bY, aM, aX, aY);
        ++t;
    }
    for (; t + 1 <= Tmin; t += 2) {  // t is odd here: the two register sets swap roles without a single move
        if ((t & 31u) == 1u) reload(t);
        step(aM, aX, aY, bM, bX, bY);
        step(bM, bX, bY, aM, aX, aY);
    }
    if (t <= Tmin) {
        if ((t & 31u) == 1u) reload(t);
        step(aM, aX, aY, bM, bX, bY);
        ++t;
    }
    // drain: pairs with a shorter haplotype than the longest of the wave drop out group by group
    for (; t <= Tmax; ++t) {
        if ((t & 31u) == 1u) reload(t);
        if (t <= Tg) step_t(t);
    }
    if (valid && lane == G - 1) {
        const T res = sumM + sumX;
        reinterpret_cast<T *>(a.out_raw)[pair] = res;
        if (sizeof(T) == 4 && rescue_list && res < static_cast<T>(1e-28f))  // pairhmm_internal.cpp:578: queue for fp64
            rescue_list[atomicAdd(rescue_count, 1u)] = pair;
    }
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
    const uint32_t r = a.pair_read[pair], h = a.pair_hap[pair];
    const uint32_t R = a.read_len[r], H = a.hap_len[h];
    const uint32_t roff = a.read_off[r];
    const uint32_t W = planes_words(H);
    const T *__restrict__ ph2pr = reinterpret_cast<const T *>(a.ph2pr);
    const T *__restrict__ mmtab = reinterpret_cast<const T *>(a.mm_table);

    // ---- stage this pair's haplotype planes into LDS (kPlanes * W words, 16 lanes copy) ----
    uint32_t *pl = lds + grp_in_block * a.lds_words_per_group;
    {
        const uint32_t *src = a.planes + a.plane_off[h];
        for (uint32_t i = lane; i < kPlanes * W; i += G) pl[i] = src[i];
    }

    // ---- constants.  One gap-open and one gap-continuation quality for the whole batch (the host checked): the transition
    // probabilities are four wave-uniform scalars; per row only what depends on the base quality and on padding remains ----
    T p_yy[K], lo[K], hi[K];
    T M[K], X[K], Y[K];
    T PM[K > 1 ? K - 1 : 1];  // M[k] * p_mx of the anti-diagonal before: computed once in pass 2 (for X of the row below), used again
                              // in the next step's pass 1 (for Y of the same row) -- the two products are the same number here
    uint32_t cls_lo = 0, cls_hi = 0;  // 3-bit plane class per row, packed (K <= 16 -> 48 bits)
    const T init_y = Num<T>::init_const() / static_cast<T>(H);  // rovaca/avx_512_float.cc:32
    const int pad = G * K - static_cast<int>(R);                  // rows [0, pad) are boundary replicas
    // kept per lane on purpose: with the four shared factors in SGPRs the sweep ran 6 % slower (a VALU instruction with an SGPR
    // source issues more slowly on this chip than one whose sources are all VGPRs; measured on config 2, 8.25 ms against 7.76 ms)
    const uint32_t g0 = a.read_ins[roff] & 127u, c0 = a.read_gcp[roff] & 127u;
    const T p_mm = mmtab[((g0 * (g0 + 1)) >> 1) + g0];  // core/context.h:162-173 with ins == del
    const T p_mx = ph2pr[g0];                            // == p_my
    const T p_gapm = static_cast<T>(1.0) - ph2pr[c0];
    const T p_xx = ph2pr[c0];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int row = static_cast<int>(lane) * K + k - pad;
        uint32_t cls = 5;
        if (row >= 0) {
            const uint32_t at = roff + static_cast<uint32_t>(row);
            const uint32_t _q = a.read_qual[at] & 127u;
            p_yy[k] = p_xx;
            const T dist = ph2pr[_q];
            hi[k] = static_cast<T>(1.0) - dist;
            lo[k] = Num<T>::div3(dist);
            cls = read_class(a.read_bases[at]);
            M[k] = 0;
            X[k] = 0;
            Y[k] = 0;
        }
        else {
            // boundary replica: prior 0 kills M whatever p_mm is, X stays 0 because everything above it is 0, p_yy = 1 holds Y
            p_yy[k] = 1;
            hi[k] = 0;
            lo[k] = 0;
            M[k] = 0;
            X[k] = 0;
            Y[k] = init_y;
        }
        if (k < 10)
            cls_lo |= cls << (3 * k);
        else
            cls_hi |= cls << (3 * (k - 10));
    }
#pragma unroll
    for (int k = 0; k + 1 < K; ++k) PM[k] = 0;
    __syncthreads();  // planes visible (each group only reads its own, but keep the block-wide fence simple)

    // ---- the sweep ----
    const uint32_t Tg = valid ? H + (G - 1) : 0;  // last step of this group
    uint32_t Tmax = Tg;                           // wave-uniform loop bound
    if (G <= 16) Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 16));
    if (G <= 32) Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 32));
    Tmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmax);

    uint32_t Tmin = Tg;  // until here every lane of the wave is busy once the pipeline is full
    if (G <= 16) Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 16));
    if (G <= 32) Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 32));
    Tmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmin);

    uint32_t mask[K];
    // The row above slot 0 travels between lanes in two register sets that swap roles every step: one holds that
    // row at the current column (written by the DPP shift of this step), the other at the previous column (written
    // one step earlier).  Lane 0 never receives anything, so its copies keep the boundary row (0, 0, init) for ever;
    // the DPP's `old` operand is the destination itself, which costs no instruction.  Before a lane's first step
    // both sets hold column 0 of the row above: Y[0][0] = init when that row is the boundary row or a replica of it
    // (lane * K <= pad), and 0 when it is a read row (M[i][0] = X[i][0] = Y[i][0] = 0).
    const T y_col0 = (static_cast<int>(lane) * K <= pad) ? init_y : static_cast<T>(0);
    T aM = 0, aX = 0, aY = y_col0, bM = 0, bX = 0, bY = y_col0;
    T sumM = 0, sumX = 0;
    const uint32_t wmax = W - 2;

    auto reload = [&](uint32_t t) {
        // every row's 32-column window of its match plane, starting at this lane's next column
        const uint32_t jn = t > lane ? t - lane : 1u;  // 1-based
        uint32_t w = (jn - 1) >> 5;
        const uint32_t s = (jn - 1) & 31u;
        w = w < wmax ? w : wmax;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t cls = k < 10 ? (cls_lo >> (3 * k)) & 7u : (cls_hi >> (3 * (k - 10))) & 7u;
            const uint32_t *p = pl + cls * W + w;
            const uint32_t w0 = p[0], w1 = p[1];
            mask[k] = (w0 << s) | ((w1 >> 1) >> (31u - s));  // bits [s, s+32) of w0:w1, MSB first (s = 0..31)
        }
    };

    // one anti-diagonal.  (dM, dX, dY): row above slot 0 at the previous column; (uM0, uX0, uY0): receives it at this column.
    auto step = [&](T &dM, T &dX, T &dY, T &uM0, T &uX0, T &uY0) {
        uM0 = dpp_shr1<G>(uM0, M[K - 1]);
        uX0 = dpp_shr1<G>(uX0, X[K - 1]);
        uY0 = dpp_shr1<G>(uY0, Y[K - 1]);
        // Pass 1, bottom row first (in place, as in the general kernel); the M * p_my term of Y comes from the product pass 2 made
        // of the same M one step ago -- except for the last row, whose M goes to the next lane
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
            uint32_t nm;
            const bool hit = __builtin_uadd_overflow(mask[k], mask[k], &nm);
            mask[k] = nm;
            const T prior = hit ? hi[k] : lo[k];
            Y[k] = Num<T>::fma_(Y[k], p_yy[k], k == K - 1 ? M[K - 1] * p_mx : PM[k < K - 1 ? k : 0]);
            const T gM = k ? M[k ? k - 1 : 0] : dM, gX = k ? X[k ? k - 1 : 0] : dX, gY = k ? Y[k ? k - 1 : 0] : dY;
            T tt = gM * p_mm;
            tt = Num<T>::fma_(gX, p_gapm, tt);
            tt = Num<T>::fma_(gY, p_gapm, tt);
            M[k] = tt * prior;
            __builtin_amdgcn_sched_barrier(0);
        }
        // Pass 2, top row first: the insertion state chains down the column through the NEW M and X
        T uM = uM0, uX = uX0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const T prod = uM * p_mx;  // k >= 1: the product of the new M[k - 1], kept for Y[k - 1] of the next step
            X[k] = Num<T>::fma_(uX, p_xx, prod);
            if (k >= 1) PM[k - 1] = prod;
            uM = M[k];
            uX = X[k];
        }
        __builtin_amdgcn_sched_barrier(0);
        sumM = sumM + M[K - 1];  // rovaca/avx_512_float.cc:166-171 (only lane 15 is read back)
        sumX = sumX + X[K - 1];
    };
    auto step_t = [&](uint32_t t) {
        if (t & 1u)
            step(aM, aX, aY, bM, bX, bY);
        else
            step(bM, bX, bY, aM, aX, aY);
    };

#ifndef PHMM_EXPERIMENT_REPEAT
#define PHMM_EXPERIMENT_REPEAT 1  // tuning experiment (tools/build_variant.sh): sweep every loaded pair this many times -- what a
#endif                            // read-stationary chain of haplotypes would save on per-pair set-up, measured without building it
#pragma unroll 1
    for (int rep = 0; rep < PHMM_EXPERIMENT_REPEAT; ++rep) {
    if (PHMM_EXPERIMENT_REPEAT > 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool padrow = static_cast<int>(lane) * K + k < pad;
            M[k] = 0;
            X[k] = 0;
            Y[k] = padrow ? init_y : static_cast<T>(0);
        }
#pragma unroll
        for (int k = 0; k + 1 < K; ++k) PM[k] = 0;
        aM = 0, aX = 0, aY = y_col0, bM = 0, bX = 0, bY = y_col0;
        sumM = 0, sumX = 0;
    }
    uint32_t t = 1;
    // fill: lane l starts at step l + 1
    const uint32_t Tfill = Tmax < (uint32_t)(G - 1) ? Tmax : (uint32_t)(G - 1);
    for (; t <= Tfill; ++t) {
        if ((t & 31u) == 1u) reload(t);
        if ((t > lane) && (t <= Tg)) step_t(t);
    }
    // steady state: every lane of all four pairs has work, no per-lane predicate
    if (t <= Tmin && !(t & 1u)) {
        step(bM, bX, bY, aM, aX, aY);
        ++t;
    }
    for (; t + 1 <= Tmin; t += 2) {  // t is odd here: the two register sets swap roles without a single move
        if ((t & 31u) == 1u) reload(t);
        step(aM, aX, aY, bM, bX, bY);
        step(bM, bX, bY, aM, aX, aY);
    }
    if (t <= Tmin) {
        if ((t & 31u) == 1u) reload(t);
        step(aM, aX, aY, bM, bX, bY);
        ++t;
    }
    // drain: pairs with a shorter haplotype than the longest of the wave drop out group by group
    for (; t <= Tmax; ++t) {
        if ((t & 31u) == 1u) reload(t);
        if (t <= Tg) step_t(t);
    }
    }
    if (valid && lane == G - 1) {
        const T res = sumM + sumX;
        reinterpret_cast<T *>(a.out_raw)[pair] = res;
        if (sizeof(T) == 4 && rescue_list && res < static_cast<T>(1e-28f))  // pairhmm_internal.cpp:578: queue for fp64
            rescue_list[atomicAdd(rescue_count, 1u)] = pair;
    }
}

}  // namespace phmm
#endif
