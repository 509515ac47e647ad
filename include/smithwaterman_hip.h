/*
 * smithwaterman_hip.h -- C ABI of the MI355X Smith-Waterman aligner (librovaca_sw_hip.so).  SURVEY.md 8f "next-4": the
 * SIMD kernel next to the PairHMM path, used right after it to realign every read to its best haplotype
 * (haplotypecaller_engine.cpp:264 -> assembly_based_caller_utils.cpp:146-169 -> alignment_utils.cpp:572-579) and by the
 * assembler for haplotype-vs-reference CIGARs (apply/assemble/hc_assemble_cigar_cacl.c:417-428).
 *
 * The reference's seam is the handle lib_sw_avx with its function pointer avx_function
 * (common/base/include/smithwaterman_common.h:61-98), bound in sw_avx_init (common/base/lib/sw_avx/sw_avx.c:401-450) to
 * sw_avx512_run / sw_avx2_run (:488-500) and called once per alignment.  One alignment is far too little work for a GPU
 * launch, so the entry point below takes a BATCH of alignments (all reads of a region, or of many regions); INTEGRATION.md
 * shows the loop in realign_reads_to_their_best_haplotype restructured around it.  Per alignment the inputs and outputs are
 * exactly the handle's: seq1/len1 ("ref", rows), seq2/len2 ("alt", columns) in; score, max_i, max_j, alignment_offset,
 * cigar[], cigar_count out -- bit for bit what sw_avx512_run leaves in the handle.
 */
#ifndef ROVACA_AMD_SMITHWATERMAN_HIP_H_
#define ROVACA_AMD_SMITHWATERMAN_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWHIP_ABI_VERSION 1

enum { SWHIP_OK = 0, SWHIP_E_INVALID = 1, SWHIP_E_NO_DEVICE = 2, SWHIP_E_HIP = 3, SWHIP_E_NOMEM = 4, SWHIP_E_CAPACITY = 5 };

#define SWHIP_MAX_SEQ_LEN 2048 /* sw_avx.c:17 */

typedef struct swhip_context swhip_context;

/* smithwaterman_common.h:53-59; the reference's two sets are sw_avx.c:13-14 */
typedef struct swhip_params {
    int32_t match, mismatch, gap_open, gap_extend;
} swhip_params;
#define SWHIP_PARAMS_ASSEMBLE {200, -150, -260, -11}
#define SWHIP_PARAMS_GENOTYPE {10, -15, -30, -5}

/* N alignments over two sequence tables: pair p aligns seq table entry pair_seq[p] (lib_sw_avx.seq2/len2) against ref
 * table entry pair_ref[p] (lib_sw_avx.seq1/len1). */
typedef struct swhip_batch {
    uint32_t n_refs, n_seqs;
    uint64_t n_pairs;
    const uint8_t *ref_bases;
    const uint32_t *ref_off, *ref_len;
    const uint8_t *seq_bases;
    const uint32_t *seq_off, *seq_len;
    const uint32_t *pair_ref, *pair_seq;
} swhip_batch;

/* what sw_avx512_run leaves in the handle (smithwaterman_common.h:63-73) */
typedef struct swhip_alignment {
    int32_t score;
    int16_t max_i, max_j;
    int16_t alignment_offset;
    uint16_t too_long;    /* 1: max(len1, len2) >= 2048; alignment_offset is 1 and there is no CIGAR, as sw_avx.c:490-495 */
    uint32_t cigar_count;
    uint32_t cigar_off;   /* first element of this alignment in the `cigars` array of swhip_align */
} swhip_alignment;

typedef struct swhip_stats {
    uint64_t cells;       /* sum of len1 * len2 over the aligned pairs */
    double kernel_ms;     /* device time of the launches, HIP events */
    double total_ms;      /* host wall time of the call */
    uint32_t n_launches;
} swhip_stats;

int swhip_abi_version(void);
const char *swhip_kernel_build_id(void); /* hash of the kernel sources and compile flags this library was linked from */
int swhip_device_count(void); /* visible HIP devices, 0 without a GPU */
int swhip_create(int device_id, void *stream, swhip_context **out); /* no CPU fallback: fails without a GPU */
void swhip_destroy(swhip_context *ctx);
const char *swhip_last_error(void);

/* cigars: caller buffer of cigar_capacity BAM elements (length << 4 | op; M=0 I=1 D=2 S=4, sw_avx.c:234-254);
 * *cigars_used receives the number written.  SWHIP_E_CAPACITY (and *cigars_used = the number needed) when it is too small;
 * sum(len1 + len2 + 2) always suffices. */
int swhip_align(swhip_context *ctx, const swhip_params *params, const swhip_batch *batch, swhip_alignment *out,
                uint32_t *cigars, uint64_t cigar_capacity, uint64_t *cigars_used, swhip_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
