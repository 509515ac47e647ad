// pairhmm_wg_timeline.h -- tuning builds only (tools/build_tab_variant.sh NAME -DPHMM_WG_TIMELINE -DPHMM_TAB_ONE_BUILD, read by tools/wg_timeline.py): when,
// at what shader clock and for which launch every workgroup of the table sweep ran.  A ring of 2^18 records over launches, four words
// each: start and end on the 100 MHz s_memrealtime clock, shader clock cycles in between (s_memtime), and the launch's grid, bits of the
// address of its pair list and the XCC_ID.  In a regular build the two macros are empty and nothing else is declared.
#ifndef ROVACA_AMD_PAIRHMM_WG_TIMELINE_H_
#define ROVACA_AMD_PAIRHMM_WG_TIMELINE_H_
#ifdef PHMM_WG_TIMELINE
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace phmm
{
constexpr uint32_t kWgTimelineMax = 1u << 18;
__device__ unsigned long long g_wg_timeline[4 * kWgTimelineMax];
__device__ unsigned int g_wg_timeline_cursor;
}  // namespace phmm

#define PHMM_WG_TIMELINE_BEGIN() \
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime(), wg_c0 = __builtin_amdgcn_s_memtime()
#define PHMM_WG_TIMELINE_END(a)                                                                                                          \
    do {                                                                                                                                 \
        __syncthreads();                                                                                                                 \
        if (threadIdx.x == 0) {                                                                                                          \
            unsigned long long *w = phmm::g_wg_timeline + 4ull * (atomicAdd(&phmm::g_wg_timeline_cursor, 1u) & (phmm::kWgTimelineMax - 1)); \
            w[0] = wg_t0;                                                                                                                \
            w[1] = __builtin_amdgcn_s_memrealtime();                                                                                     \
            w[2] = __builtin_amdgcn_s_memtime() - wg_c0;                                                                                 \
            w[3] = ((unsigned long long)gridDim.x << 40) | ((unsigned long long)((reinterpret_cast<uintptr_t>((a).order) >> 6) & 0xFFFFFFFFu) << 8) | \
                   (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFF);                                                                  \
        }                                                                                                                                \
    } while (0)

// (include in ONE translation unit)  the whole ring (4 x 2^18 words) and the number of records written so far (the newest is at (cursor - 1) mod 2^18)
extern "C" __attribute__((visibility("default"), used)) int phmm_debug_wg_timeline(unsigned long long *host, unsigned int *cursor)
{
    hipError_t e = hipMemcpyFromSymbol(cursor, HIP_SYMBOL(phmm::g_wg_timeline_cursor), sizeof(unsigned int));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(phmm::g_wg_timeline), (size_t)phmm::kWgTimelineMax * 4 * sizeof(unsigned long long));
}
#else
#define PHMM_WG_TIMELINE_BEGIN() ((void)0)
#define PHMM_WG_TIMELINE_END(a) ((void)0)
#endif
#endif
