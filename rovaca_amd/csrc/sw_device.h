// sw_device.h -- device-side declarations of the Smith-Waterman aligner: shared by sw_kernels.hip and
// sw_capi.cpp.  The algorithm is the reference's sw_avx.c (common/base/lib/sw_avx/sw_avx.c:41-362), bit for bit.
#ifndef ROVACA_AMD_SW_DEVICE_H_
#define ROVACA_AMD_SW_DEVICE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace swhip
{

constexpr int kBlockThreads = 256;
constexpr int kMaxSeqLen = 2048;                    // sw_avx.c:17: sequences of this length or more are not aligned
constexpr int32_t kLowInit = INT32_MIN / 2;         // sw_avx.c:19
// The kernel keeps scores biased by 2^30 so that unsigned compares order them.  LOW_INIT itself would be 0 there and
// LOW_INIT + gap_extend would wrap, so the kernel's "minus infinity" is 2^20 above it: still a billion below any real score
// (|score| < 2^20 for sequences under 2048 bases and penalties under 512), which is all the recurrence needs of it.
constexpr uint32_t kBias = 1u << 30;
constexpr uint32_t kLowBiased = 1u << 20;
constexpr int kMaxColsPerLaneNarrow = 16;           // 16 lanes per alignment: seq2 up to 256 bases
constexpr int kMaxColsPerLaneWide = 32;             // 64 lanes per alignment: seq2 up to 2048 bases

// backtrack flags per cell, 4 bits (smithwaterman_common.h:41-46): low two bits 0 match / 1 insert / 2 delete,
// bit 2 "the insertion was an extension", bit 3 "the deletion was an extension"
enum : uint32_t { kBtInsert = 1, kBtDelete = 2, kBtInsertExt = 4, kBtDeleteExt = 8 };
// BAM operators of the emitted elements (hc_marco.h:22-26), element = length << 4 | op (sw_avx.c:234-241)
enum : uint32_t { kOpM = 0, kOpI = 1, kOpD = 2, kOpS = 4 };

struct Alignment  // mirrors swhip_alignment of include/smithwaterman_hip.h
{
    int32_t score;
    int16_t max_i, max_j;
    int16_t alignment_offset;
    uint16_t flags;
    uint32_t cigar_count;
    uint32_t cigar_off;
};

struct SwArgs
{
    const uint8_t *ref_bases;   // seq1 of the reference's handle ("ref"; rows)
    const uint32_t *ref_off, *ref_len;
    const uint8_t *seq_bases;   // seq2 ("alt"; columns, spread over the lanes)
    const uint32_t *seq_off, *seq_len;
    const uint32_t *pair_ref, *pair_seq;
    const uint32_t *order;      // pairs of this launch, in processing order
    uint32_t n_pairs;
    uint32_t lds_stride;        // bytes of LDS per alignment: the longest seq1 of the launch, rounded up
    uint32_t bt_in_lds;         // 1: backtrack flags live in LDS (lds_bt_bytes behind seq1), one alignment per workgroup
    uint32_t lds_bt_bytes;
    uint32_t score_table;       // 1: the launch looks the scores up in a per-alignment table in LDS (sw_kernels.hip)
    int32_t w_match, w_mismatch, w_open, w_extend;
    uint32_t *bt;               // backtrack flags; pair p starts at bt_off[p] (see bt_words)
    const uint64_t *bt_off;
    int32_t *scratch;           // per pair: last-column H (len1), last-row H (len2), element buffer
    const uint64_t *scratch_off;
    Alignment *out;             // [pair]
    uint32_t *cigars;           // compact element array, allocated with atomicAdd on *cigar_cursor
    uint32_t *cigar_cursor;
};

// backtrack flags: 4 bits per cell; a lane's K cells of a step are K / 8 full dwords plus a tail of 1, 2 or 4 bytes
inline __host__ __device__ constexpr uint32_t bt_tail_bytes(int cols_per_lane)
{
    return cols_per_lane % 8 == 0 ? 0u : cols_per_lane % 8 <= 2 ? 1u : cols_per_lane % 8 <= 4 ? 2u : 4u;
}
// dwords of backtrack storage of one alignment: [steps][lanes][K / 8] dwords, then [steps][lanes] tails, rounded up to 128 B
inline __host__ __device__ uint64_t bt_words(uint32_t len1, int lanes, int cols_per_lane)
{
    const uint64_t slots = (uint64_t)(len1 + lanes - 1) * lanes;
    const uint64_t words = slots * (uint64_t)(cols_per_lane / 8) + (slots * bt_tail_bytes(cols_per_lane) + 3) / 4;
    return (words + 31) & ~uint64_t(31);
}
// int32 slots of scratch a pair needs with the given geometry
inline __host__ __device__ uint64_t scratch_slots(uint32_t len1, uint32_t len2, int cols_per_lane)
{
    (void)cols_per_lane;
    return (uint64_t)len1 + len2 + (uint64_t)len1 + len2 + 8u;  // last column, last row, element buffer
}

// LDS of the score table of one alignment (sw_kernels.hip; up to 16 columns per lane): four bases x its lanes x 16 or 32 bytes
inline __host__ __device__ uint32_t score_table_bytes(int cols_per_lane, int lanes)
{
    return 4u * (uint32_t)lanes * 16u * (cols_per_lane > 8 ? 2u : 1u);
}

hipError_t launch_sw(int cols_per_lane, int lanes_per_pair, const SwArgs &a, hipStream_t s);
// the cols_per_lane values that are instantiated: round a requirement up to one of them (0 if too large)
int pick_cols_per_lane(uint32_t len2, int lanes_per_pair);

}  // namespace swhip
#endif
