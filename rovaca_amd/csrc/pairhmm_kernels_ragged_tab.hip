// pairhmm_kernels_ragged_tab.hip -- ragged launches of the table sweep (pairhmm_systolic_tab.h): ONE launch for several rows-per-lane
// classes, every workgroup picks the K of its class, exactly as pairhmm_kernels_ragged.hip does for the bit-plane sweeps (which says
// why).  Compiled twice, once per arithmetic form (-DPHMM_RAGGED_TAB_MODE=1: per-base gap-opens, the Rovaca case under a PCR indel
// model; =2: one gap-open quality), so that the two halves build side by side.
//
// One kernel per form: every class of the table sweep fits one register budget (128, four waves per SIMD), so -- unlike the bit-plane
// sweeps, which need K ranges for big launches and a uniform-budget kernel with 32-lane classes for chunk- and stage-sized ones -- the
// same kernel serves launches of every size.  The LDS of a launch is the largest any of its classes needs; every class gets as many
// table slots as fit it (RaggedSeg::tab_slots, set by the host).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pairhmm_device.h"
#include "pairhmm_systolic_tab.h"

#ifndef PHMM_RAGGED_TAB_MODE
#error "compile with -DPHMM_RAGGED_TAB_MODE=1 or 2"
#endif

namespace phmm
{

namespace
{
constexpr int kMode = PHMM_RAGGED_TAB_MODE;
constexpr bool kUgo = kMode == 2;
}

// (MODE is a template parameter of everything with external linkage here: the two builds of this file go into one library)
template <int MODE, int K, int KHI>
__device__ __forceinline__ void ragged_tab_dispatch(uint32_t rows, const KernelArgs &a, uint32_t wg, const RaggedSeg &sg)
{
    if (rows == (uint32_t)K) {
        systolic_tab_sweep<K, kUgo>(a, wg, sg.pos_first, sg.count, a.rescue_list + sg.list_base, a.rescue_count + sg.queue, sg.tab_slots);
        return;
    }
    if constexpr (K < KHI) ragged_tab_dispatch<MODE, K + 1, KHI>(rows, a, wg, sg);
}

template <int MODE>
__global__ void __launch_bounds__(256, kTabWaves) pairhmm_systolic_ragged_tab_kernel(KernelArgs a, RaggedArgs m)
{
    const uint32_t b = xcd_local_workgroup(blockIdx.x, a.xcd_run_log2);
    if (b >= m.n_wg) return;  // (the XCD-local order rounds the grid up)
    uint32_t s = 0;
    for (uint32_t i = 1; i < m.n_seg; ++i)
        if (b >= m.seg[i].wg_first) s = i;
    const RaggedSeg sg = m.seg[s];
    ragged_tab_dispatch<MODE, 1, kTabMaxRows>(sg.rows, a, b - sg.wg_first, sg);
}

// LDS of a launch: the most any of its classes needs -- its tables and haplotype entries, and the planes of the bit-plane sweep a
// workgroup with an N in a haplotype falls back to
static size_t ragged_tab_lds(const KernelArgs &a, const RaggedArgs &m)
{
    size_t bytes = (size_t)16 * a.lds_words_per_group * sizeof(uint32_t);
    for (uint32_t i = 0; i < m.n_seg; ++i)
        bytes = std::max<size_t>(bytes, tab_lds_bytes((int)m.seg[i].rows, m.seg[i].tab_slots, a.tab_max_hap));
    return bytes;
}

static uint32_t ragged_grid(const KernelArgs &a, const RaggedArgs &m)
{
    uint32_t blocks = m.n_wg;
    if (a.xcd_run_log2) {
        const uint32_t round = 8u << a.xcd_run_log2;
        blocks = (blocks + round - 1) / round * round;
    }
    return blocks;
}

#if PHMM_RAGGED_TAB_MODE == 2
hipError_t launch_systolic_ragged_tab_ugo_f32(const KernelArgs &a, const RaggedArgs &m, hipStream_t s)
#else
hipError_t launch_systolic_ragged_tab_ug_f32(const KernelArgs &a, const RaggedArgs &m, hipStream_t s)
#endif
{
    if (m.n_seg == 0 || m.n_wg == 0) return hipSuccess;
    if (m.n_seg > (uint32_t)kRaggedMaxSegs || !a.hap_bases || !a.hap_off) return hipErrorInvalidValue;
    for (uint32_t i = 0; i < m.n_seg; ++i) {
        const RaggedSeg &g = m.seg[i];
        if (g.rows < 1 || g.rows > (uint32_t)kTabMaxRows || g.tab_slots < 1 || g.tab_slots > 16) return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL((pairhmm_systolic_ragged_tab_kernel<kMode>), dim3(ragged_grid(a, m)), dim3(256), ragged_tab_lds(a, m), s, a, m);
    return hipGetLastError();
}

}  // namespace phmm
