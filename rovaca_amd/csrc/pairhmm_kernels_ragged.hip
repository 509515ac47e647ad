// pairhmm_kernels_ragged.hip -- "ragged" launches of the single-pass systolic sweep: ONE launch for several rows-per-lane classes.
//
// Why.  The single-pass kernels are templates of K = rows per lane (the read's rows live in VGPRs, so K decides the register budget and
// with it the waves per SIMD), and a batch of mixed read lengths was one launch per K: thirteen for BASELINE config 3 (reads of 50-250
// bases), each a thirteenth of the pairs.  A launch pays about one and a half workgroup lifetimes of ramp and tail whatever its size
// ("How long a launch is"), so the chunks of a one-shot call (13 launches of ~6 000 pairs each, none of them a
// full round of the chip's 16 384 pair slots) ran 20-25 % below the uniform-length rate, and a stage of seven regions with trimmed
// reads either paid seven launches or padded every read to the longest one's K.  The reference packs near-equal lengths side by side
// in one pass too (pairhmm_internal.cpp:268-391: groups of 16 reads within +-5 rows of each other).
//
// How.  The sweep of one workgroup is a function template of K (pairhmm_systolic.h).  A ragged kernel holds the sweeps of a RANGE of K
// values that share one occupancy target (the same __launch_bounds__ the one-K kernels use), and every workgroup looks up the class it
// belongs to in a small table passed by value (RaggedArgs: per class its first workgroup, its range of the processing order, its K and
// its rescue queue) and branches -- a scalar, workgroup-uniform branch -- into that K's sweep.  All 16 pairs of a workgroup are of one
// class; classes are listed longest rows first, so the launch starts its longest workgroups first and the tail is made of short ones.
// Bits cannot change: it is the same sweep on the same pair, only the launch it rides in differs
// (tests: every mixed-length parity test runs both ways, PHMM_NO_RAGGED).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pairhmm_device.h"
#include "pairhmm_systolic.h"

namespace phmm
{

// MODE 1: the Rovaca case (one gap-continuation quality, deletion gap-open == insertion gap-open); 2: one gap-open quality as well
template <typename T, int MODE, int G, int K, int KHI>
__device__ __forceinline__ void ragged_dispatch(uint32_t rows, const KernelArgs &a, uint32_t wg, const RaggedSeg &sg)
{
    if (rows == (uint32_t)K) {
        if constexpr (MODE == 2)
            systolic_ugo_sweep<T, K, G>(a, wg, sg.pos_first, sg.count, a.rescue_list + sg.list_base, a.rescue_count + sg.queue);
        else
            systolic_sweep<T, K, true, G>(a, wg, sg.pos_first, sg.count, a.rescue_list + sg.list_base, a.rescue_count + sg.queue);
        return;
    }
    if constexpr (K < KHI) ragged_dispatch<T, MODE, G, K + 1, KHI>(rows, a, wg, sg);
}

// occupancy target of a range: that of its largest K (the ranges are cut where the target changes, ragged_ranges below)
template <typename T, int MODE, int G, int KHI>
constexpr int ragged_min_waves()
{
    return G == 64 ? 4 : MODE == 2 ? PHMM_UGO_MIN_WAVES(T, KHI) : PHMM_MIN_WAVES(T, KHI, true);
}

template <typename T, int MODE, int G, int KLO, int KHI>
__global__ void __launch_bounds__(systolic_block_threads<G>(), (ragged_min_waves<T, MODE, G, KHI>()))
    pairhmm_systolic_ragged_kernel(KernelArgs a, RaggedArgs m)
{
    const uint32_t b = xcd_local_workgroup(blockIdx.x, a.xcd_run_log2);
    if (b >= m.n_wg) return;  // (the XCD-local order rounds the grid up)
    // the class of this workgroup: the last one that begins at or before it (scalar loop over at most kRaggedMaxSegs entries)
    uint32_t s = 0;
    for (uint32_t i = 1; i < m.n_seg; ++i)
        if (b >= m.seg[i].wg_first) s = i;
    const RaggedSeg sg = m.seg[s];
    ragged_dispatch<T, MODE, G, KLO, KHI>(sg.rows, a, b - sg.wg_first, sg);
}

// ----------------------------------------------------------------------------------------------------
// One register budget for every class: the ragged kernel of chunk- and stage-sized launches.
//
// The K ranges above run side by side as two or three kernels with budgets of 128, 168 and 200+ VGPRs.  In a launch of a million pairs
// that costs nothing: the big-K kernels are launched first, finish in the first third of the pass, and the small-K kernel has the chip
// to itself for the rest.  In the chunks of a one-shot call (or a stage) the mix never ends -- two chunks are always in flight -- and a
// SIMD that holds one 168-register wave has room for two more waves, not three: the small-K waves, three quarters of the work, spend
// the pass at three waves per SIMD instead of four (config 3's one-shot call, round 4: every chunk 16 % slower than the resident
// rate, 10.1 ms of device time for 7.9 ms of sweeps; config 2, one class and one kernel per chunk, loses nothing).  Here the classes
// whose 16-lane sweep needs much more than 128 registers (reads of 209-256 bases; from 177 with per-base gap-opens) take the 32-lane shape
// instead -- two pairs per wave, 6-8 rows per lane, about 100 registers -- so that ONE kernel at four waves per SIMD holds every class.
// Those classes issue about 1.2x the instructions (eight rows carry the per-step overhead that sixteen did); everything else is the
// sweep it always was.  Same bits either way.
// ----------------------------------------------------------------------------------------------------
template <typename T, int MODE, int K, int KHI>
__device__ __forceinline__ void ragged_dispatch32(uint32_t rows, const KernelArgs &a, uint32_t wg, const RaggedSeg &sg)
{
    if (rows == (uint32_t)K) {
        if constexpr (MODE == 2)
            systolic_ugo_sweep<T, K, 32, kBlockThreads>(a, wg, sg.pos_first, sg.count, a.rescue_list + sg.list_base, a.rescue_count + sg.queue);
        else
            systolic_sweep<T, K, true, 32, kBlockThreads>(a, wg, sg.pos_first, sg.count, a.rescue_list + sg.list_base, a.rescue_count + sg.queue);
        return;
    }
    if constexpr (K < KHI) ragged_dispatch32<T, MODE, K + 1, KHI>(rows, a, wg, sg);
}

// the largest rows per lane the 16-lane sweep runs at four waves per SIMD, and the rows per lane of the 32-lane classes above it
// (-DPHMM_UNIFORM_K16_MAX_UGO=n / _UG=n move the border: tuning hooks.)  Beyond 12 / 11 rows the 16-lane sweeps no longer fit 128
// registers and the compiler spills: 14 dwords per lane at 13 rows with one gap-open quality, 68 at 14, 91 at 16; 68 at 12 rows with
// per-base gap-opens.  Fourteen dwords of scratch traffic per step cost less than the 32-lane shape's 1.2x instructions -- config 3's
// one-shot call 9.04-9.26 -> 8.81-8.84 ms with the border at 13, 9.5 at 14, 13.0 at 15, 19.2 at 16 (round 4, A/B on one box) -- so the
// one-gap-open border is 13; the per-base one stays at 11.
#ifndef PHMM_UNIFORM_K16_MAX_UGO
#define PHMM_UNIFORM_K16_MAX_UGO 13
#endif
#ifndef PHMM_UNIFORM_K16_MAX_UG
#define PHMM_UNIFORM_K16_MAX_UG 11
#endif
template <int MODE> constexpr int uniform_k16_max() { return MODE == 2 ? PHMM_UNIFORM_K16_MAX_UGO : PHMM_UNIFORM_K16_MAX_UG; }
template <int MODE> constexpr int uniform_k32_min() { return (16 * uniform_k16_max<MODE>() + 1 + 31) / 32; }

template <typename T, int MODE>
__global__ void __launch_bounds__(kBlockThreads, 4) pairhmm_systolic_ragged_uniform_kernel(KernelArgs a, RaggedArgs m)
{
    const uint32_t b = xcd_local_workgroup(blockIdx.x, a.xcd_run_log2);
    if (b >= m.n_wg) return;
    uint32_t s = 0;
    for (uint32_t i = 1; i < m.n_seg; ++i)
        if (b >= m.seg[i].wg_first) s = i;
    const RaggedSeg sg = m.seg[s];
    if constexpr (uniform_k32_min<MODE>() <= kMaxRowsPerLaneMid) {
        if (sg.lanes == 32u) {
            ragged_dispatch32<T, MODE, uniform_k32_min<MODE>(), kMaxRowsPerLaneMid>(sg.rows, a, b - sg.wg_first, sg);
            return;
        }
    }
        ragged_dispatch<T, MODE, 16, 1, uniform_k16_max<MODE>()>(sg.rows, a, b - sg.wg_first, sg);
}

int ragged_uniform_lanes_of(int rows_per_lane_16, bool uniform_gop)
{
    return rows_per_lane_16 <= (uniform_gop ? uniform_k16_max<2>() : uniform_k16_max<1>()) ? 16 : 32;
}

hipError_t launch_systolic_ragged_uniform_f32(bool uniform_gop, const KernelArgs &a, const RaggedArgs &m, hipStream_t s)
{
    if (m.n_seg == 0 || m.n_wg == 0) return hipSuccess;
    if (m.n_seg > (uint32_t)kRaggedMaxSegs) return hipErrorInvalidValue;
    for (uint32_t i = 0; i < m.n_seg; ++i) {
        const RaggedSeg &g = m.seg[i];
        const int k16 = uniform_gop ? uniform_k16_max<2>() : uniform_k16_max<1>(), k32 = uniform_gop ? uniform_k32_min<2>() : uniform_k32_min<1>();
        const bool ok = g.lanes == 16u ? (g.rows >= 1 && g.rows <= (uint32_t)k16) : g.lanes == 32u ? (g.rows >= (uint32_t)k32 && g.rows <= (uint32_t)kMaxRowsPerLaneMid) : false;
        if (!ok) return hipErrorInvalidValue;
    }
    uint32_t blocks = m.n_wg;
    if (a.xcd_run_log2) {
        const uint32_t round = 8u << a.xcd_run_log2;
        blocks = (blocks + round - 1) / round * round;
    }
    const size_t shmem = static_cast<size_t>(kBlockThreads / 16) * a.lds_words_per_group * sizeof(uint32_t);
    if (uniform_gop)
        hipLaunchKernelGGL((pairhmm_systolic_ragged_uniform_kernel<float, 2>), dim3(blocks), dim3(kBlockThreads), shmem, s, a, m);
    else
        hipLaunchKernelGGL((pairhmm_systolic_ragged_uniform_kernel<float, 1>), dim3(blocks), dim3(kBlockThreads), shmem, s, a, m);
    return hipGetLastError();
}

template <typename T, int MODE, int G, int KLO, int KHI>
static hipError_t launch_ragged_one(const KernelArgs &a, const RaggedArgs &m, hipStream_t s)
{
    constexpr uint32_t threads = systolic_block_threads<G>();
    uint32_t blocks = m.n_wg;
    if (a.xcd_run_log2) {
        const uint32_t round = 8u << a.xcd_run_log2;
        blocks = (blocks + round - 1) / round * round;
    }
    const size_t shmem = static_cast<size_t>(threads / G) * a.lds_words_per_group * sizeof(uint32_t);
    hipLaunchKernelGGL((pairhmm_systolic_ragged_kernel<T, MODE, G, KLO, KHI>), dim3(blocks), dim3(threads), shmem, s, a, m);
    return hipGetLastError();
}

// The K ranges that share a kernel: cut where the occupancy target of the one-K kernels changes (pairhmm_systolic.h: PHMM_MIN_WAVES,
// PHMM_UGO_MIN_WAVES), so that no class runs at fewer waves per SIMD than it would alone.
int ragged_range_of(int rows_per_lane, int lanes_per_pair, bool uniform_gop)
{
    if (lanes_per_pair == 64) return rows_per_lane >= 1 && rows_per_lane <= kMaxRowsPerLaneWide ? 0 : -1;
    if (rows_per_lane < 1 || rows_per_lane > kMaxRowsPerLane) return -1;
    if (uniform_gop) return rows_per_lane <= 12 ? 0 : rows_per_lane <= 15 ? 1 : 2;
    return rows_per_lane <= 11 ? 0 : rows_per_lane <= 14 ? 1 : 2;
}

hipError_t launch_systolic_ragged_f32(int lanes_per_pair, bool uniform_gop, const KernelArgs &a, const RaggedArgs &m, hipStream_t s)
{
    if (m.n_seg == 0 || m.n_wg == 0) return hipSuccess;
    if (m.n_seg > (uint32_t)kRaggedMaxSegs) return hipErrorInvalidValue;
    const int range = ragged_range_of((int)m.seg[0].rows, lanes_per_pair, uniform_gop);
    for (uint32_t i = 0; i < m.n_seg; ++i)
        if (ragged_range_of((int)m.seg[i].rows, lanes_per_pair, uniform_gop) != range || range < 0) return hipErrorInvalidValue;
    if (lanes_per_pair == 64)
        return uniform_gop ? launch_ragged_one<float, 2, 64, 1, kMaxRowsPerLaneWide>(a, m, s)
                           : launch_ragged_one<float, 1, 64, 1, kMaxRowsPerLaneWide>(a, m, s);
    if (uniform_gop) {
        if (range == 0) return launch_ragged_one<float, 2, 16, 1, 12>(a, m, s);
        if (range == 1) return launch_ragged_one<float, 2, 16, 13, 15>(a, m, s);
        return launch_ragged_one<float, 2, 16, 16, 16>(a, m, s);
    }
    if (range == 0) return launch_ragged_one<float, 1, 16, 1, 11>(a, m, s);
    if (range == 1) return launch_ragged_one<float, 1, 16, 12, 14>(a, m, s);
    return launch_ragged_one<float, 1, 16, 15, 16>(a, m, s);
}

}  // namespace phmm
