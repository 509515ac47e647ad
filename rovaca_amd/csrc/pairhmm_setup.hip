// pairhmm_setup.hip -- gfx950 device code that prepares a batch for the sweep kernels (pairhmm_kernels.hip): haplotype bit planes,
// device-side read preparation (H3 quality modification + H4 PCR indel model), and the one-launch set-up of a region batch that
// arrives through pinned staging (include/pairhmm_hip.h, phmm_stage).  Integer/byte work, exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pairhmm_device.h"

namespace phmm
{

// ----------------------------------------------------------------------------------------------------
// haplotype bit planes.  One thread per (haplotype, 32-column word).  Column j (1-based) is bit
// 31 - ((j-1) & 31) of word (j-1) >> 5: MSB first, so that `v_add_co_u32 m, vcc, m, m` pops the next column.
//   plane 0..3: read base A/C/G/T matches this column (hap base equal, or hap base N)
//   plane 4   : read base N matches this column (hap base is any of ACGTN)
//   plane 5   : read symbol outside ACGTN: never matches (all zero)
// Words past the haplotype end are zero.
// ----------------------------------------------------------------------------------------------------
// A wave builds TWO global word slots at a time (one column per lane, one ballot per plane); the two may belong to different
// haplotypes.  (Until round 3 one thread built one word from 32 dependent byte loads: 24-29 us per launch whatever its size.)
__device__ __forceinline__ void hap_planes_words2(const uint8_t *__restrict__ hap_bases, const uint32_t *__restrict__ hap_off,
                                                  const uint32_t *__restrict__ hap_len, const uint32_t *__restrict__ plane_off,
                                                  const uint32_t *__restrict__ word_owner, uint32_t n_words,
                                                  uint32_t *__restrict__ planes, uint32_t g0)
{
    if (g0 >= n_words) return;  // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = g0 + (lane >> 5);  // this half-wave's word slot
    const bool live = g < n_words;
    // word_owner[g] = haplotype that owns global word slot g; the slot index inside that haplotype follows
    const uint32_t h = word_owner[live ? g : g0];
    const uint32_t H = hap_len[h], W = planes_words(H), po = plane_off[h];
    const uint32_t w = g - po / kPlanes;  // plane_off is in words and counts kPlanes * W per haplotype
    const uint32_t j0 = w * 32u + (lane & 31u);
    const uint8_t c = live && j0 < H ? hap_bases[hap_off[h] + j0] : (uint8_t)0;  // 0 matches nothing
    const bool isN = c == 'N';
    const bool m[kPlanes - 1] = {c == 'A' || isN, c == 'C' || isN, c == 'G' || isN, c == 'T' || isN,
                                 c == 'A' || c == 'C' || c == 'G' || c == 'T' || isN};
    uint32_t *dst = planes + po;
#pragma unroll
    for (int p = 0; p < kPlanes - 1; ++p) {
        const uint64_t b = __ballot(m[p]);  // lane i -> bit i; a plane word wants column 32 w + k at bit 31 - k
        const uint32_t mine = __brev((lane >> 5) ? (uint32_t)(b >> 32) : (uint32_t)b);
        if (live && (lane & 31u) == 0) dst[p * W + w] = mine;
    }
    if (live && (lane & 31u) == 0) dst[(kPlanes - 1) * W + w] = 0;  // plane 5: a read symbol outside ACGTN never matches
}

// ----------------------------------------------------------------------------------------------------
// device-side read preparation: H3 quality modification and H4 PCR indel model, one workgroup
// per read, one thread per base.  Integer work, exact.  Follows pairhmm_internal.cpp:434-439 and :65-159 / :195-229
// including the quirks the host adapter documents (rovaca_amd/host/pairhmm_host_logic.cpp).
// ----------------------------------------------------------------------------------------------------
__device__ int32_t count_tiled_units_dev(const uint8_t *unit, int32_t ulen, const uint8_t *text, int32_t tlen, bool from_front)
{
    if (tlen == 0) return 0;
    int32_t copies = 0;
    const int32_t last = tlen - ulen;
    if (from_front) {
        for (int32_t at = 0; at <= last; at += ulen, ++copies) {
            bool same = true;
            for (int32_t i = 0; i < ulen; ++i) same = same && (text[at + i] == unit[i]);
            if (!same) break;
        }
    }
    else {
        for (int32_t at = last; at >= 0; at -= ulen, ++copies) {
            bool same = true;
            for (int32_t i = 0; i < ulen; ++i) same = same && (text[at + i] == unit[i]);
            if (!same) break;
        }
    }
    return copies;
}

__device__ int32_t tandem_repeat_length_dev(const uint8_t *bases, int32_t len, int32_t offset)
{
    const int32_t prefix = offset + 1;
    int32_t backward = 0;
    for (int32_t unit = 1; unit <= 8 && unit <= prefix; ++unit) {
        backward = count_tiled_units_dev(bases + prefix - unit, unit, bases, prefix, false);
        if (backward > 1) break;
    }
    int32_t total = backward;
    if (offset < len - 1) {
        const int32_t suffix = len - prefix;
        int32_t forward = 0, fw_unit = 1;
        for (int32_t unit = 1; unit <= 8 && unit <= suffix; ++unit) {
            forward = count_tiled_units_dev(bases + prefix, unit, bases + prefix, suffix, true);
            if (forward > 1) {
                fw_unit = unit;
                break;
            }
        }
        backward = count_tiled_units_dev(bases + prefix, fw_unit, bases, prefix, false);
        total = backward + forward;
    }
    return total < 20 ? total : 20;
}

__constant__ uint8_t kPcrCap[4][21] = {
    {255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255},
    {40, 40, 39, 38, 37, 36, 34, 32, 28, 23, 17, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10},   // HOSTILE      adapter.h:13
    {40, 40, 40, 39, 39, 39, 38, 38, 37, 37, 36, 35, 34, 33, 32, 30, 28, 26, 23, 20, 17},   // AGGRESSIVE   adapter.h:14
    {40, 40, 40, 40, 39, 39, 39, 39, 39, 38, 38, 38, 37, 37, 37, 36, 36, 35, 34, 33, 33}};  // CONSERVATIVE adapter.h:15

// ----------------------------------------------------------------------------------------------------
// H4 for a whole read at once.  The search above walks tiles byte by byte: a chain of a few hundred dependent loads and branches per
// base, 16 us per launch on a chip that clocks down between region-sized submissions.  The same counts follow from eight
// bit masks.  Let E_u[j] = (bases[j] == bases[j - u]), j >= u.  A tile equals the unit iff it equals the tile next to it (all tiles in
// between having matched), so with p = offset + 1:
//   copies of the unit bases[p .. p + u) counted forward from p   = 1 + (ones of E_u starting at p + u) / u
//   copies of that unit counted backward over bases[0 .. p)       =     (ones of E_u ending at p + u - 1) / u
// (pairhmm_internal.cpp:195-229 with the unit taken from the text itself; runs cannot leave the read, so the `last` bounds hold by
// themselves).  For every base the reference processes (all but the last) the count of its backward search is overwritten by the
// re-count with the forward unit (pairhmm_internal.cpp:108, see tandem_repeat_length in pairhmm_host_logic.cpp), so only the two
// lines above are needed.  Runs are followed for 160 bits: 20 copies of the longest unit, beyond which the total is capped at 20.
// The masks live in LDS with two zero words in front and four behind, so that windows never leave them.
// ----------------------------------------------------------------------------------------------------
constexpr int kStrUnits = 8;                                     // adapter.h:7
constexpr uint32_t kMaskPadFront = 2, kMaskPadBack = 4;
__device__ __forceinline__ uint32_t mask_words(uint32_t n) { return kMaskPadFront + ((n + 31u) >> 5) + kMaskPadBack; }

// workgroup-wide: masks[u - 1][...] for u = 1..8 from the read's bases in LDS.  Bit j of E_u is bit (j & 31) of word j >> 5.
__device__ __forceinline__ void build_repeat_masks(const uint8_t *sb, uint32_t n, uint32_t *masks)
{
    const uint32_t W = mask_words(n), chunks = (n + 63u) >> 6;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (uint32_t i = threadIdx.x; i < (uint32_t)kStrUnits * W; i += blockDim.x) masks[i] = 0;
    __syncthreads();
    for (uint32_t t = wave; t < (uint32_t)kStrUnits * chunks; t += n_waves) {
        const uint32_t u = t / chunks + 1u, j = (t % chunks) * 64u + lane;
        const bool e = j < n && j >= u && sb[j] == sb[j - u];
        const uint64_t b = __ballot(e);
        uint32_t *m = masks + (u - 1u) * W + kMaskPadFront + (t % chunks) * 2u;
        if (lane == 0) m[0] = (uint32_t)b;
        if (lane == 1 && (t % chunks) * 2u + 1u < ((n + 31u) >> 5)) m[1] = (uint32_t)(b >> 32);
    }
    __syncthreads();
}
// ones of the mask starting at bit j going up (j may lie behind the read: zero), followed for at most 160 bits
__device__ __forceinline__ uint32_t ones_from(const uint32_t *m, uint32_t j)
{
    uint32_t run = 0;
    for (int k = 0; k < 5; ++k, j += 32u) {
        const uint32_t w = kMaskPadFront + (j >> 5), sh = j & 31u;
        const uint32_t x = sh ? (m[w] >> sh) | (m[w + 1] << (32u - sh)) : m[w];
        const uint32_t t = x == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~x);
        run += t;
        if (t < 32u) break;
    }
    return run;
}
// ones of the mask ending at bit j going down
__device__ __forceinline__ uint32_t ones_until(const uint32_t *m, uint32_t j)
{
    uint32_t run = 0;
    int32_t jj = (int32_t)j;
    for (int k = 0; k < 5 && jj >= 0; ++k, jj -= 32) {
        // the 32 bits [jj - 31, jj]: bit 31 of x is bit jj
        const int32_t lo = jj - 31;
        const uint32_t w = (uint32_t)((int32_t)kMaskPadFront + (lo >> 5)), sh = (uint32_t)lo & 31u;  // (arithmetic shift: lo may be negative, the pad words are zero)
        const uint32_t x = sh ? (m[w] >> sh) | (m[w + 1] << (32u - sh)) : m[w];
        const uint32_t t = x == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_clz(~x);
        run += t;
        if (t < 32u) break;
    }
    return run;
}
// tandem_repeat_length for a base that is not the read's last one (offset < n - 1)
__device__ __forceinline__ int32_t tandem_repeat_length_masks(const uint32_t *masks, uint32_t n, uint32_t offset)
{
    const uint32_t W = mask_words(n), p = offset + 1u, suffix = n - p;
    uint32_t forward = 1, fu = 1;  // (no unit repeats: the count left over is that of the last unit tried, which is 1, and the re-count uses unit 1)
    for (uint32_t u = 1; u <= (uint32_t)kStrUnits && u <= suffix; ++u) {
        const uint32_t cf = 1u + ones_from(masks + (u - 1u) * W, p + u) / u;
        if (cf > 1u) {
            forward = cf;
            fu = u;
            break;
        }
    }
    const uint32_t backward = ones_until(masks + (fu - 1u) * W, p + fu - 1u) / fu;
    const uint32_t total = backward + forward;
    return (int32_t)(total < 20u ? total : 20u);
}

// H3 + H4 + the constant gap-continuation penalty for one read, by one workgroup.  raw_q may alias out_qual (in-place preparation).
__device__ __forceinline__ void prepare_read(const uint8_t *__restrict__ bases_g, const uint8_t *raw_q, uint8_t *out_qual,
                                             uint8_t *__restrict__ out_gop, uint8_t *__restrict__ out_gcp, uint32_t n, uint8_t mq,
                                             int32_t min_q, int32_t pcr_model, uint8_t gap_open, uint8_t gap_cont, uint8_t *sb, uint32_t *masks)
{
    const bool fast = pcr_model != 0 && n <= (uint32_t)kStageLdsBytes;  // (longer reads search their repeats byte by byte in global memory)
    if (fast) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) sb[i] = bases_g[i];
        __syncthreads();
        build_repeat_masks(sb, n, masks);
    }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        uint8_t q = raw_q[i];
        q = q < mq ? q : mq;                        // pairhmm_internal.cpp:435-436
        q = (int32_t)q < min_q ? (uint8_t)6 : q;    // :437-439, adapter.h:5
        out_qual[i] = q;
        out_gcp[i] = gap_cont;
        uint8_t g = gap_open;
        if (pcr_model != 0 && i + 1 < n) {  // the last base keeps its gap-open
            const int32_t rep = fast ? tandem_repeat_length_masks(masks, n, i) : tandem_repeat_length_dev(bases_g, (int32_t)n, (int32_t)i);
            const uint8_t cap = kPcrCap[pcr_model][rep];
            g = g < cap ? g : cap;
        }
        out_gop[i] = g;
    }
}

__device__ __forceinline__ void prep_one_read(const uint8_t *__restrict__ read_bases, uint8_t *read_qual,
                                              uint8_t *__restrict__ read_gop, uint8_t *__restrict__ read_gcp,
                                              const uint32_t *__restrict__ read_off, const uint32_t *__restrict__ read_len,
                                              const uint8_t *__restrict__ read_mapq, int32_t min_q, int32_t pcr_model,
                                              uint8_t gap_open, uint8_t gap_cont, uint32_t r, uint8_t *sb, uint32_t *masks)
{
    const uint32_t off = read_off[r], n = read_len[r];
    prepare_read(read_bases + off, read_qual + off, read_qual + off, read_gop + off, read_gcp + off, n, read_mapq[r], min_q, pcr_model, gap_open,
                 gap_cont, sb, masks);
}

// One launch sets a batch up: workgroups [0, n_prep_reads) prepare one read each (H3 + H4, only for batches that carry a
// phmm_read_prep), the rest build the haplotype bit planes, one wave per two 32-column words; the rescue queue is emptied on
// the way.  A region-sized call is a chain of dependent launches and copies, each worth 5-10 us: fewer links, lower latency.
__global__ void __launch_bounds__(kSetupThreads) setup_kernel(SetupArgs s)
{
    __shared__ uint8_t sb[kStageLdsBytes];
    __shared__ uint32_t masks[kStrUnits * (kMaskPadFront + kStageLdsBytes / 32 + kMaskPadBack)];
    if (blockIdx.x == 0 && threadIdx.x < kRescueQueues && s.rescue_count) s.rescue_count[threadIdx.x] = 0;
    if (blockIdx.x < s.n_prep_reads) {
        prep_one_read(s.read_bases, s.read_qual, s.read_gop, s.read_gcp, s.read_off, s.read_len, s.read_mapq, s.min_q, s.pcr_model,
                      s.gap_open, s.gap_cont, blockIdx.x, sb, masks);
        return;
    }
    hap_planes_words2(s.hap_bases, s.hap_off, s.hap_len, s.plane_off, s.word_owner, s.n_words, s.planes,
                      ((blockIdx.x - s.n_prep_reads) * (kSetupThreads / 64) + (threadIdx.x >> 6)) * 2u);
}

hipError_t launch_setup(const SetupArgs &a, hipStream_t s)
{
    const uint32_t words_per_block = 2u * (kSetupThreads / 64);  // two word slots per wave
    const uint32_t blocks = a.n_prep_reads + (a.n_words + words_per_block - 1) / words_per_block;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(setup_kernel, dim3(blocks), dim3(kSetupThreads), 0, s, a);
    return hipGetLastError();
}


// ----------------------------------------------------------------------------------------------------
// Region batches through pinned staging: ONE launch turns what the workers wrote into the stage (copied to HBM as one block:
// the byte arena, then the descriptors the submitting thread put behind it) into everything the sweep kernels index.  Workgroup
// r < n_reads takes read r: applies H3 to its raw qualities and H4 to its gap-opens (pairhmm_internal.cpp:434-439, :65-159) and
// spells out the read's row of its region's cross product -- pair ids, (read, hap) indices and the processing order the host
// binned per read.  Workgroup n_reads + h takes haplotype h: its six match planes.  The host's share is O(reads + haplotypes)
// descriptors; the tables' bytes are written once by the workers and copied once by the DMA engine.
// ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kStageSetupThreads) stage_setup_kernel(StageSetupArgs s)
{
    __shared__ uint8_t sb[kStageLdsBytes];
    __shared__ uint32_t masks[kStrUnits * (kMaskPadFront + kStageLdsBytes / 32 + kMaskPadBack)];
    if (blockIdx.x == 0 && threadIdx.x < kRescueQueues) s.rescue_count[threadIdx.x] = 0;
    if (blockIdx.x < s.n_reads) {
        const uint32_t r = blockIdx.x;
        const StageReadDesc d = s.read_desc[r];
        const uint32_t off = d.off, n = d.len;
        if (threadIdx.x == 0) {
            s.read_off[r] = off;
            s.read_len[r] = n;
        }
        // the read's row of the cross product
        for (uint32_t j = threadIdx.x; j < d.n_haps; j += blockDim.x) {
            const uint32_t p = d.pair_first + j;
            s.pair_read[p] = r;
            s.pair_hap[p] = d.hap_first + j;
            s.order[d.slot_first + j] = p;
        }
        prepare_read(s.arena + off, s.arena + d.qual_off, s.read_qual + off, s.read_gop + off, s.read_gcp + off, n, (uint8_t)d.mapq, s.min_q,
                     s.pcr_model, s.gap_open, s.gap_cont, sb, masks);
        return;
    }
    const uint32_t h = blockIdx.x - s.n_reads;
    if (h >= s.n_haps) return;
    const StageHapDesc d = s.hap_desc[h];
    const uint32_t H = d.len, W = planes_words(H);
    if (threadIdx.x == 0) {
        s.hap_len[h] = H;
        s.plane_off[h] = d.plane_off;
        s.hap_off[h] = d.off;  // the table kernels read the haplotype's bases (in the arena) themselves
    }
    // One column per lane, one ballot per plane: a wave turns 64 columns into two words of each plane with no loop at all.  (The
    // first form of this kernel gave every word to one thread that walked its 32 columns -- 32 dependent steps on a chip that
    // clocks down between region-sized launches: 22 us per launch, more than the sweep of a small region.)
    const uint8_t *hb = s.arena + d.off;
    uint32_t *dst = s.planes + d.plane_off;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (uint32_t w0 = 2u * wave; w0 < W; w0 += 2u * n_waves) {  // words w0 and w0 + 1: columns 32 w0 .. 32 w0 + 63 (0-based)
        const uint32_t j0 = w0 * 32u + lane;
        const uint8_t c = j0 < H ? hb[j0] : (uint8_t)0;  // 0 matches nothing: columns past the end are zero in every plane
        const bool isN = c == 'N';
        const bool m[kPlanes - 1] = {c == 'A' || isN, c == 'C' || isN, c == 'G' || isN, c == 'T' || isN,
                                     c == 'A' || c == 'C' || c == 'G' || c == 'T' || isN};
#pragma unroll
        for (int p = 0; p < kPlanes - 1; ++p) {
            const uint64_t b = __ballot(m[p]);  // lane i -> bit i; a plane word wants column 32 w + k at bit 31 - k
            const uint32_t lo = __brev((uint32_t)b), hi = __brev((uint32_t)(b >> 32));
            if (lane == 0) dst[p * W + w0] = lo;
            if (lane == 1 && w0 + 1 < W) dst[p * W + w0 + 1] = hi;
        }
        if (lane == 0) dst[(kPlanes - 1) * W + w0] = 0;  // plane 5: a read symbol outside ACGTN never matches
        if (lane == 1 && w0 + 1 < W) dst[(kPlanes - 1) * W + w0 + 1] = 0;
    }
}

hipError_t launch_stage_setup(const StageSetupArgs &a, hipStream_t s)
{
    const uint32_t blocks = a.n_reads + a.n_haps;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_setup_kernel, dim3(blocks), dim3(kStageSetupThreads), 0, s, a);
    return hipGetLastError();
}

}  // namespace phmm
