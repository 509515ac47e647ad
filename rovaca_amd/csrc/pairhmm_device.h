// pairhmm_device.h -- shared between the device code (pairhmm_kernels.hip, pairhmm_setup.hip) and the C-ABI host code.
#ifndef ROVACA_AMD_PAIRHMM_DEVICE_H_
#define ROVACA_AMD_PAIRHMM_DEVICE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace phmm
{

constexpr int kLanesPerPair = 16;     // one DPP row per (read, haplotype) pair; four pairs per wave64
#ifndef PHMM_BLOCK_THREADS
#define PHMM_BLOCK_THREADS 256  // tuning aid (tools/build_variant.sh): 128 and 64 measured within 0.5 % on config 2, 1 % worse on config 3
#endif
constexpr int kBlockThreads = PHMM_BLOCK_THREADS;    // 4 waves = 16 pairs per workgroup
constexpr int kMaxRowsPerLane = 16;   // K: reads up to 16 * 16 = 256 rows take the single-pass kernel
constexpr int kMaxRowsPerLaneWide = 8;  // the one-pair-per-wave shape goes further: 64 * 8 = 512 rows in one pass
constexpr int kMaxRowsPerLaneMid = 8;   // two pairs per wave (32 lanes each): reads up to 32 * 8 = 256 rows, as the 16-lane shape
constexpr int kPlanes = 6;            // haplotype match planes: read symbol A, C, G, T, N, other
#ifndef PHMM_STRIPED_ROWS_F32
#define PHMM_STRIPED_ROWS_F32 8
#endif
constexpr int kStripedRowsF32 = PHMM_STRIPED_ROWS_F32;   // rows per lane of the striped fp32 kernel (stripes of 256 read rows)
// rows per lane of the striped fp64 kernel: instantiated for 4, 5 and 6 (stripes of 64, 80, 96 read rows with 16 lanes per
// pair); the launch picks the one that wastes the fewest padding rows on the batch's read lengths
constexpr int kStripedRowsF64Min = 4, kStripedRowsF64Max = 6;
// rows per lane of the band kernel (teams of several waves, fp32 and fp64 alike): measured on config 5, fp32 pass 4.7 ms at 8
// rows, 3.6 at 6, 2.7 at 4; fp64 re-run 2 520 / 2 740 / 2 760 GCUPS overall at 6 / 5 / 4
constexpr int kBandRows = 4;
// Per type since round 4 (-DPHMM_BAND_ROWS_F32=n / -DPHMM_BAND_ROWS_F64=n are tuning hooks).  The fp32 pass of a long pair is one wave's
// walk along the whole haplotype -- in fp32 nothing survives more than ~75 rows off the diagonal, so only the first stripe is ever full
// width, and the pass lasts as long as that stripe's 12 000 dependent steps take -- and a step of two rows is shorter than one of
// four: config 5's fp32 pass 2.42 ms at 4 rows per lane, 2.11 at 3, 2.04 at 2, 2.14 at 1 (profiles/r4_c5_band_rows.txt).  fp64, whose
// full-width strip is ~1 000 rows deep and issue-bound, is fastest at 4 (pass 12.75 ms; 14.1 at 3, 15.1 at 2).
#ifndef PHMM_BAND_ROWS_F32
#define PHMM_BAND_ROWS_F32 2
#endif
#ifndef PHMM_BAND_ROWS_F64
#define PHMM_BAND_ROWS_F64 kBandRows
#endif
constexpr int kBandRowsF32 = PHMM_BAND_ROWS_F32, kBandRowsF64 = PHMM_BAND_ROWS_F64;
constexpr int kStripedGroupsPerBlock = 4;  // a team (1, 4 or 8 waves) of the striped kernel sweeps four pairs

// 32-bit words per haplotype match plane (column j, 1-based, is bit 31-((j-1)&31) of word (j-1)>>5); one spare
// word so that a window read [w, w+1] at the last sweep step (column H + 15) stays inside the plane
__host__ __device__ inline uint32_t planes_words(uint32_t hap_len) { return ((hap_len + 14u) >> 5) + 2u; }

constexpr int kMaxRescueQueuesPerLaunch = 16;
struct KernelArgs
{
    // read table (raw phred bytes, ::TestCase semantics, core/common.h:13-24)
    const uint8_t *read_bases, *read_qual, *read_ins, *read_del, *read_gcp;
    const uint32_t *read_off, *read_len;
    // haplotype table: lengths and the precomputed match planes
    const uint32_t *hap_len;
    const uint32_t *planes;
    const uint32_t *plane_off;
    // pair list and the length-binned processing order of this launch
    const uint32_t *pair_read, *pair_hap;
    const uint32_t *order;
    uint32_t n_pairs;
    uint32_t lds_words_per_group;
    uint32_t no_early_exit;  // test hook: 1 = sweep every superstep even where everything is zero
    // lookup tables (core/context.h) in the arithmetic type of the launch
    const void *ph2pr;
    const void *mm_table;
    void *out_raw;  // [n_pairs of the whole batch], indexed by pair id
    // fp32 launches: pairs whose sum fell below 1e-28f (rovaca/common.h:9) are appended here for the fp64 re-run
    uint32_t *rescue_list;
    uint32_t *rescue_count;
    // fp64 re-run: the number of pairs in `order` is read from device memory (no host round trip between the passes)
    const uint32_t *n_pairs_dev;
    // single-pass kernels, big launches: log2 of the run of consecutive workgroups (in processing order) that one XCD takes, 0 = plain
    // order.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its own L2), and the pairs of a
    // region -- which share their reads and haplotypes -- sit next to each other in the processing order: with runs of 64 workgroups
    // (1 024 pairs) per XCD a region's bytes are fetched into one L2 instead of eight.
    uint32_t xcd_run_log2;
    // fp64 re-run over SEVERAL rescue queues in one launch (striped kernels; n_rescue_queues = 0: the one queue n_pairs_dev counts).
    // `order` is then the batch's whole rescue list, `rescue_count` the base of its counters; queue i of the launch is counter
    // rescue_queue[i] and begins at rescue_base[i] of the list.  The teams stride over the queues' pairs kGroups at a time, every
    // group of kGroups from ONE queue (pairs of one length class re-run in step).  A batch of mixed lengths has a queue per class;
    // one launch each was thirteen launches per pass, nearly all of them empty, 5-50 us apiece at every chunk boundary of a one-shot call.
    uint32_t n_rescue_queues;
    uint32_t rescue_queue[kMaxRescueQueuesPerLaunch];
    uint32_t rescue_base[kMaxRescueQueuesPerLaunch];
    // tuning aid (PHMM_BAND_TIMELINE): the band kernel's waves log every stripe they sweep -- kTimelineWords 64-bit words per (team,
    // stripe), the first timeline_teams teams only; null otherwise.  Times are s_memrealtime ticks (100 MHz).
    unsigned long long *timeline;
    uint32_t timeline_teams;
    // striped kernels, round 4: the rescue queue of a striped size range is a list IN PROCESSING ORDER with holes.  The fp32 launch
    // writes an underflowed pair's id at the pair's own position of the launch (rescue_list[position], kNoPair elsewhere: the host fills
    // the list before the pass) instead of appending it, and the fp64 launch strides over all n_pairs positions (n_pairs_dev only says
    // whether there is anything at all) and skips the holes: the re-run then takes the pairs longest first, as the fp32 pass did,
    // instead of in the order the fp32 teams happened to finish -- shortest first, the worst order for the makespan (config 5: fp64
    // pass 10.4 -> 9.4 ms).  (At the end of the struct: the single-pass kernels' argument offsets are what they were.)
    uint32_t sparse_order;
    // fp64 FIRST (round 4, striped ranges whose pairs have been underflowing fp32: pairhmm_capi.cpp, "fp64 first"): an fp64 launch over
    // ALL pairs of the range with rescue_list set queues -- at the pair's position, sparse_order -- every pair whose sum is at least
    // need32_above: the pairs whose fp32 sum might reach 1e-28, for which the fp32 kernel then runs and decides as the reference does
    double need32_above;
    // table kernels (pairhmm_kernels_tab.hip): the haplotypes' bytes, and the longest haplotype of the launch (sizes the LDS of a pair)
    const uint8_t *hap_bases;
    const uint32_t *hap_off;
    uint32_t tab_max_hap;
    uint32_t tab_slots;  // tables per workgroup (LDS)
    uint32_t band_tab;      // the fp64 band kernel may take its table form (hap_bases / hap_off / tab_max_hap set)
};
constexpr uint32_t kNoPair = 0xFFFFFFFFu;
constexpr int kTimelineStripes = 64, kTimelineWords = 8;

// lanes_per_pair 16: rows_per_lane = ceil(R / 16) in 1..kMaxRowsPerLane; 64: ceil(R / 64) in 1..kMaxRowsPerLaneWide
// uniform_gop (only with uniform_gcp): every base of the batch has the same gap-open quality too -> the variant that shares p_mm / p_mx
hipError_t launch_systolic_f32(int rows_per_lane, int lanes_per_pair, bool uniform_gcp, const KernelArgs &a, hipStream_t s,
                               bool uniform_gop = false);
// Table kernels (pairhmm_kernels_tab.hip, pairhmm_systolic_tab.h): the same sweep with the prior looked up in an LDS table per read
// instead of selected by a match bit.  The Rovaca case only (one gap-continuation quality, deletion == insertion gap-opens), 16 lanes
// per pair; uniform_gop as in launch_systolic_f32.  KernelArgs::hap_bases / hap_off / tab_max_hap / tab_slots must be set;
// tab_slots_for says how many tables fit a workgroup at the kernel's occupancy (0: the class cannot take the table kernel).
constexpr int kTabMaxRows = 16;
// (registers: 7 K + 25 with one gap-open quality, 8 K + 25 with per-base gap-opens)
// Every class of the table sweep runs at four waves per SIMD: with the priors single-buffered the steady loop of 16 rows per lane
// fits 128 registers in both arithmetic forms (what spills is the prologue and the bit-plane fallback, outside the loop)
#ifndef PHMM_TAB_WAVES
#define PHMM_TAB_WAVES 4  // (tools/build_tab_variant.sh: tuning and debugging)
#endif
constexpr int kTabWaves = PHMM_TAB_WAVES;
// LDS a workgroup of a table kernel may use at `waves` per SIMD, and how many tables then fit next to the 16 haplotypes (0: none)
uint32_t tab_lds_budget(int waves);
uint32_t tab_slots_for(int rows_per_lane, uint32_t max_hap, uint32_t lds_budget);
constexpr uint32_t kTabMinSlots = 4;  // fewer tables than this per workgroup: the class takes the bit-plane kernel
hipError_t launch_systolic_tab_f32(int rows_per_lane, bool uniform_gop, const KernelArgs &a, hipStream_t s);
// Ragged launches (pairhmm_kernels_ragged.hip): ONE launch over several rows-per-lane classes of a batch with mixed read lengths.
// Every workgroup belongs to one class; the classes are listed in launch order (longest rows first).  KernelArgs::order / rescue_list /
// rescue_count are the batch's whole arrays, n_pairs is not read.
constexpr int kRaggedMaxSegs = 16;
struct RaggedSeg
{
    uint32_t wg_first;   // first workgroup of the class in the launch (in processing order, before the XCD-local remap)
    uint32_t pos_first;  // the class's range of `order`: [pos_first, pos_first + count)
    uint32_t count;
    uint32_t rows;       // rows per lane of the class's sweep
    uint32_t queue;      // its rescue queue (index into rescue_count) ...
    uint32_t list_base;  // ... and where that queue's list begins in rescue_list
    uint32_t lanes;      // lanes per pair of the class's sweep (the uniform-budget kernel: 16 or 32; the K-range kernels: not read)
    uint32_t tab_slots;  // table kernels: prior tables a workgroup of this class holds in LDS
};
struct RaggedArgs
{
    uint32_t n_seg, n_wg;  // classes, workgroups in all
    RaggedSeg seg[kRaggedMaxSegs];
};
// Ragged launches of the table sweep (pairhmm_kernels_ragged_tab.hip), as launch_systolic_ragged_f32 below: several rows-per-lane
// classes in one launch.  ONE kernel per arithmetic form holds every class (one register budget, four waves per SIMD), so launches of
// any size take it.  RaggedSeg::tab_slots per class (RaggedSeg::lanes is not read); KernelArgs::tab_max_hap for the launch.
hipError_t launch_systolic_ragged_tab_f32(bool uniform_gop, const KernelArgs &a, const RaggedArgs &m, hipStream_t s);
// which ragged kernel a class belongs to (classes of one range share a kernel: the same occupancy target); -1: none
int ragged_range_of(int rows_per_lane, int lanes_per_pair, bool uniform_gop);
// uniform gap-continuation penalty and deletion == insertion gap-opens are required (the Rovaca case); uniform_gop as in launch_systolic_f32
hipError_t launch_systolic_ragged_f32(int lanes_per_pair, bool uniform_gop, const KernelArgs &a, const RaggedArgs &m, hipStream_t s);
// The uniform-budget form: ONE kernel at four waves per SIMD for every class of the launch; a class whose 16-lane sweep needs more than
// 128 registers is given as 32 lanes per pair (RaggedSeg::lanes / rows = ceil(R / 32)).  ragged_uniform_lanes_of: which, by the class's
// 16-lane rows per lane.
int ragged_uniform_lanes_of(int rows_per_lane_16, bool uniform_gop);
hipError_t launch_systolic_ragged_uniform_f32(bool uniform_gop, const KernelArgs &a, const RaggedArgs &m, hipStream_t s);
// striped ("team") kernels: a workgroup of waves_per_team waves (1, 4 or 8) sweeps four pairs; lds_words_per_group is
// kPlanes * planes_words(longest haplotype of the launch); carry holds 3 * carry_stride values per (team, wave, group).
// lanes_per_pair: 16 (four pairs per wave) or 64 (one pair per wave)
hipError_t launch_striped_f32(bool uniform_gcp, int waves_per_team, int lanes_per_pair, const KernelArgs &a, float *carry,
                              uint32_t carry_stride, uint32_t teams, hipStream_t s);
hipError_t launch_striped_f64(int rows_per_lane, bool uniform_gcp, int waves_per_team, int lanes_per_pair, const KernelArgs &a, double *carry,
                              uint32_t carry_stride, uint32_t teams, hipStream_t s);
inline size_t team_lds_bytes(uint32_t plane_words_per_group, int waves_per_team, size_t value_bytes, int lanes_per_pair)
{
    const size_t groups = 64u / (size_t)lanes_per_pair;
    // planes of the team's pairs, one carry ring per (wave, group), and the waves' progress records (2 rounds x waves x 8 words)
    return groups * (size_t)plane_words_per_group * sizeof(uint32_t) + (size_t)waves_per_team * groups * 3u * 64u * value_bytes +
           2u * (size_t)waves_per_team * 8u * sizeof(uint32_t);
}
// the band kernel's table form (fp64, 4 rows per lane, 64 lanes per pair): haplotype entries (16 bits per column, 64 in front, 72
// behind, whole 64-entry rounds) and one 12 KB table per wave in front of the rings
constexpr uint32_t kBandTabBytes = 2u * 6u * 1024u;
inline __host__ __device__ uint32_t band_tab_hap_bytes(uint32_t max_hap) { return 2u * ((64u + max_hap + 72u + 63u) & ~63u); }
inline __host__ __device__ uint32_t band_tab_front_bytes(uint32_t max_hap, int waves_per_team)
{
    return band_tab_hap_bytes(max_hap) + (uint32_t)waves_per_team * kBandTabBytes;
}
inline size_t team_carry_values(uint32_t teams, int waves_per_team, uint32_t carry_stride, int lanes_per_pair)
{
    return (size_t)teams * waves_per_team * (64u / (size_t)lanes_per_pair) * 3u * carry_stride;
}
// batch set-up (one launch): optional device-side read preparation, haplotype bit planes, rescue queue reset
constexpr int kSetupThreads = 128;
// rescue queues per batch: 0 = all single-pass bins of a small batch, 1..4 = the striped size ranges, 5..20 = one per
// single-pass rows-per-lane class of a large batch (pairs of one class re-run together: equal stripes, sorted haplotypes)
constexpr int kRescueQueues = 24;
struct SetupArgs
{
    // haplotype planes
    const uint8_t *hap_bases;
    const uint32_t *hap_off, *hap_len, *plane_off, *word_owner;
    uint32_t n_words;
    uint32_t *planes;
    // read preparation (n_prep_reads = 0: the read table arrived prepared)
    const uint8_t *read_bases;
    uint8_t *read_qual, *read_gop, *read_gcp;
    const uint32_t *read_off, *read_len;
    const uint8_t *read_mapq;
    uint32_t n_prep_reads;
    int32_t min_q, pcr_model;
    uint8_t gap_open, gap_cont;
    uint32_t *rescue_count;  // kRescueQueues counters, zeroed if non-null
};
hipError_t launch_setup(const SetupArgs &a, hipStream_t s);
// set-up of a region batch that arrived through pinned staging (phmm_stage): one launch, see pairhmm_setup.hip
constexpr int kStageSetupThreads = 256;
constexpr int kStageLdsBytes = 4096;  // sequence bytes a set-up workgroup parks in LDS at a time
struct StageReadDesc  // written by the submitting thread behind the used part of the arena, one per read (32 bytes)
{
    uint32_t off, len;        // the read's bases in the arena
    uint32_t qual_off;        // its raw base qualities in the arena
    uint32_t slot_first;      // position in `order` of its first pair
    uint32_t pair_first;      // id of its first pair (its region's first pair + row * n_haps)
    uint32_t hap_first, n_haps;
    uint32_t mapq;
};
struct StageHapDesc  // 16 bytes
{
    uint32_t off, len, plane_off, pad;
};
struct StageSetupArgs
{
    // the submission as it was copied from the stage: the byte arena and, behind its used part, the descriptors (HBM)
    const uint8_t *arena;
    const StageReadDesc *read_desc;
    const StageHapDesc *hap_desc;
    // what the sweep kernels index (HBM).  The read table keeps the arena's offsets: read_bases IS the arena, the prepared
    // qualities / gap-opens / gap-continuations are parallel arrays indexed by the same offsets
    uint8_t *read_qual, *read_gop, *read_gcp;
    uint32_t *read_off, *read_len, *hap_len, *plane_off, *planes, *pair_read, *pair_hap, *order;
    uint32_t n_reads, n_haps;
    int32_t min_q, pcr_model;
    uint8_t gap_open, gap_cont;
    uint32_t *rescue_count;  // kRescueQueues counters, zeroed
    uint32_t *hap_off;       // each haplotype's offset in the arena (table kernels)
};
hipError_t launch_stage_setup(const StageSetupArgs &a, hipStream_t s);
// allele marginalisation (pairhmm_marginal.hip): device pointers; old [n_old][stride], out [n_new][evidence_count]
hipError_t launch_marginalize(const double *old, uint32_t n_old, uint32_t stride, uint32_t evidence_count, const int32_t *old2new,
                              uint32_t n_new, double *out, hipStream_t s);

}  // namespace phmm
#endif
