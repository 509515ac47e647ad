// pairhmm_kernels_tab.hip -- the one-K launches of the table sweep (pairhmm_systolic_tab.h): the single-pass systolic sweep with the
// prior looked up in an LDS table per read instead of selected by a match bit, gfx950.  One launch per rows-per-lane class, 16 lanes per
// pair, 256-thread workgroups (the four waves of a workgroup land on the four SIMDs of a CU; one-wave workgroups measured 8 % slower).
//
// Compiled three times, -DPHMM_TAB_SCHED=0 / 1 / 2, each with another instruction scheduler of the compiler (csrc/pairhmm_tab_sched.mk);
// a build holds the kernels of the classes that tab_plan (pairhmm_systolic_tab.h) assigns to it, build 0 also everything that is not
// a template.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "pairhmm_device.h"
#include "pairhmm_systolic_tab.h"
#ifndef PHMM_TAB_SCHED
#define PHMM_TAB_SCHED 0
#endif
#if PHMM_TAB_SCHED != 0
#undef PHMM_WG_TIMELINE  // (the time line's ring lives in build 0: tools/build_tab_variant.sh wgtl -DPHMM_WG_TIMELINE -DPHMM_TAB_ONE_BUILD)
#endif
#include "pairhmm_wg_timeline.h"  // (nothing unless built with -DPHMM_WG_TIMELINE: tools/wg_timeline.py)

namespace phmm
{

template <int K, bool UGO>
__global__ void __launch_bounds__(256, kTabWaves) pairhmm_systolic_tab_kernel(KernelArgs a)
{
    PHMM_WG_TIMELINE_BEGIN();
    systolic_tab_sweep<K, UGO, tab_plan(K, UGO).one_loop>(a, xcd_local_workgroup(blockIdx.x, a.xcd_run_log2), 0u, a.n_pairs, a.rescue_list, a.rescue_count, a.tab_slots);
    PHMM_WG_TIMELINE_END(a);
}

template <int K, bool UGO>
static hipError_t launch_tab_one(const KernelArgs &a, hipStream_t s)
{
    if constexpr (tab_plan(K, UGO).sched != PHMM_TAB_SCHED) {
        return hipErrorInvalidValue;  // another build's class (launch_systolic_tab_f32 routes)
    }
    else {
    const uint32_t blocks = (a.n_pairs + 15u) / 16u;
    uint32_t grid = blocks;
    if (a.xcd_run_log2) {  // whole rounds of 8 runs (the extra workgroups find no pair)
        const uint32_t round = 8u << a.xcd_run_log2;
        grid = (blocks + round - 1) / round * round;
    }
    // (the bit-plane sweep a workgroup falls back to for haplotypes with N needs its planes' words)
    const size_t shmem = std::max<size_t>(tab_lds_bytes(K, a.tab_slots, a.tab_max_hap), (size_t)16 * a.lds_words_per_group * sizeof(uint32_t));
    hipLaunchKernelGGL((pairhmm_systolic_tab_kernel<K, UGO>), dim3(grid), dim3(256), shmem, s, a);
    return hipGetLastError();
    }
}

template <bool UGO, int K>
static hipError_t launch_tab_k(int k, const KernelArgs &a, hipStream_t s)
{
    if (k == K) return launch_tab_one<K, UGO>(a, s);
    if constexpr (K < kTabMaxRows)
        return launch_tab_k<UGO, K + 1>(k, a, s);
    else
        return hipErrorInvalidValue;
}

// this build's classes
#if PHMM_TAB_SCHED == 0
hipError_t launch_systolic_tab_build0(int rows_per_lane, bool uniform_gop, const KernelArgs &a, hipStream_t s)
#elif PHMM_TAB_SCHED == 1
hipError_t launch_systolic_tab_build1(int rows_per_lane, bool uniform_gop, const KernelArgs &a, hipStream_t s)
#else
hipError_t launch_systolic_tab_build2(int rows_per_lane, bool uniform_gop, const KernelArgs &a, hipStream_t s)
#endif
{
    return uniform_gop ? launch_tab_k<true, 1>(rows_per_lane, a, s) : launch_tab_k<false, 1>(rows_per_lane, a, s);
}

#if PHMM_TAB_SCHED == 0
hipError_t launch_systolic_tab_build1(int rows_per_lane, bool uniform_gop, const KernelArgs &a, hipStream_t s);
hipError_t launch_systolic_tab_build2(int rows_per_lane, bool uniform_gop, const KernelArgs &a, hipStream_t s);

uint32_t tab_lds_budget(int waves)
{
    // the LDS of a CU (160 KiB) shared out among as many 256-thread workgroups as the register budget admits waves per SIMD
    return (160u * 1024u) / (uint32_t)(waves < 1 ? 1 : waves) - 256u;
}

uint32_t tab_slots_for(int rows_per_lane, uint32_t max_hap, uint32_t lds_budget)
{
    if (rows_per_lane < 1 || rows_per_lane > kTabMaxRows) return 0;
    if (tab_lds_bytes(rows_per_lane, 0, max_hap) >= lds_budget) return 0;
    uint32_t slots = 16;
    while (slots && tab_lds_bytes(rows_per_lane, slots, max_hap) > lds_budget) --slots;
    return slots;
}

hipError_t launch_systolic_tab_f32(int rows_per_lane, bool uniform_gop, const KernelArgs &a, hipStream_t s)
{
    if (a.n_pairs == 0) return hipSuccess;
    if (!a.hap_bases || !a.hap_off || a.tab_slots == 0 || a.tab_slots > 16) return hipErrorInvalidValue;
    if (rows_per_lane < 1 || rows_per_lane > kTabMaxRows) return hipErrorInvalidValue;
    switch (tab_plan(rows_per_lane, uniform_gop).sched) {
        case kTabSchedMaxIlp: return launch_systolic_tab_build1(rows_per_lane, uniform_gop, a, s);
        case kTabSchedIterIlp: return launch_systolic_tab_build2(rows_per_lane, uniform_gop, a, s);
        default: return launch_systolic_tab_build0(rows_per_lane, uniform_gop, a, s);
    }
}

// the ragged forms live in pairhmm_kernels_ragged_tab.hip, compiled once per arithmetic form
hipError_t launch_systolic_ragged_tab_ugo_f32(const KernelArgs &a, const RaggedArgs &m, hipStream_t s);
hipError_t launch_systolic_ragged_tab_ug_f32(const KernelArgs &a, const RaggedArgs &m, hipStream_t s);
hipError_t launch_systolic_ragged_tab_f32(bool uniform_gop, const KernelArgs &a, const RaggedArgs &m, hipStream_t s)
{
    return uniform_gop ? launch_systolic_ragged_tab_ugo_f32(a, m, s) : launch_systolic_ragged_tab_ug_f32(a, m, s);
}
#endif
}  // namespace phmm
