// sw_seam.h -- the part of Rovaca's Smith-Waterman interface the GPU candidate touches, restated so that the adapter builds
// and is tested outside Rovaca's tree: the handle lib_sw_avx with its function pointer
// (common/base/include/smithwaterman_common.h:53-103).  Inside the tree (-DROVACA_AMD_IN_ROVACA_TREE) the real header is
// used instead; field order and types below follow it exactly, because callers fill the handle directly
// (alignment_utils.cpp:572-579, hc_assemble_cigar_cacl.c:417-428).
#ifndef ROVACA_AMD_SW_SEAM_H_
#define ROVACA_AMD_SW_SEAM_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scoring constants of one handle (two sets exist: assembly and genotyping) */
typedef struct sw_parameters_t {
    int32_t match, mismatch, gap_open, gap_extend;
} sw_parameters;

/* The handle.  Only the members marked [in] / [out] are touched by the GPU candidate; the rest is the AVX
 * implementation's private state and is kept solely so that offsets agree with the reference's definition. */
typedef struct lib_sw_avx_st {
    uint8_t *seq1; /* [in]  rows: haplotype / padded reference */
    uint8_t *seq2; /* [in]  columns: read / haplotype path */
    int16_t len1;  /* [in]  */
    int16_t len2;  /* [in]  */
    int32_t score; /* [out] best end-cell score */
    int16_t max_i; /* [out] row of the best end cell */
    int16_t max_j; /* [out] column of the best end cell */
    int16_t *btrack;        /* AVX: backtrack matrix */
    uint32_t *cigar;        /* [out] BAM elements, first to last */
    uint32_t cigar_count;   /* [out] */
    uint32_t cigar_max_len; /* [in]  capacity of cigar[] */
    int16_t alignment_offset; /* [out] rows of seq1 before the alignment starts; 1 when nothing was aligned */

    int32_t max_seq_len; /* [in]  length limit (2048) */
    int32_t seq_len_changed;

    struct { /* AVX: aligned work buffers */
        int32_t *e;
        int16_t *back_track;
        int16_t *cigar_buf;
    } mm_aloc;

    sw_parameters para; /* [in] */

    uint32_t data_len;
    uint32_t max_data_len;

    struct { /* AVX: widened copies of the sequences */
        uint32_t *seq1_rev;
        uint32_t *seq2;
    } cacl_cache;
    void (*avx_function)(struct lib_sw_avx_st *p); /* the seam: bound once, called per alignment */

    uint8_t data[]; /* storage the pointers above point into */
} lib_sw_avx, *p_lib_sw_avx;

#ifdef __cplusplus
}
#endif
#endif
