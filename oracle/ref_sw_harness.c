/*
 * ref_sw_harness.c -- thin driver around the REFERENCE's own Smith-Waterman (sw_avx.c / sw_avx2.c).  TEST INFRASTRUCTURE ONLY.
 *
 * Built only where /root/reference exists: oracle/Makefile compiles common/base/lib/sw_avx/sw_avx.c and sw_avx2.c from where
 * they lie (headers from the reference tree: common/base/include and third_lib/htslib/include, macros only) and links them
 * with this file into oracle/_ref/libsw_ref.so.  No reference source is copied.  This file only does what the callers of
 * the seam do (alignment_utils.cpp:572-579, hc_assemble_cigar_cacl.c:417-428): fill seq1/seq2/len1/len2 of the
 * lib_sw_avx handle and call the function pointer.
 */
#include <stdint.h>
#include <string.h>

#include "smithwaterman_common.h"

static p_lib_sw_avx g_handle[2]; /* [is_assemble] */

/* which: 0 = the function sw_avx_init selected (AVX-512 where supported), 1 = sw_avx512_run, 2 = sw_avx2_run */
int ref_sw_align(int is_assemble, int which, const uint8_t *seq1, int32_t len1, const uint8_t *seq2, int32_t len2,
                 int32_t *score, int32_t *max_i, int32_t *max_j, int32_t *alignment_offset, uint32_t *cigar,
                 uint32_t cigar_cap, uint32_t *cigar_count)
{
    is_assemble = is_assemble ? 1 : 0;
    if (!g_handle[is_assemble]) g_handle[is_assemble] = sw_avx_init((uint8_t)is_assemble);
    p_lib_sw_avx p = g_handle[is_assemble];
    if (!p) return -2;
    if (len1 < 0 || len2 < 0 || len1 > 32767 || len2 > 32767) return -1;
    if (len1 < p->max_seq_len) memcpy(p->seq1, seq1, (size_t)len1);
    if (len2 < p->max_seq_len) memcpy(p->seq2, seq2, (size_t)len2);
    p->len1 = (int16_t)len1;
    p->len2 = (int16_t)len2;
    p->cigar_count = 0;
    p->alignment_offset = 0;
    if (which == 1)
        sw_avx512_run(p);
    else if (which == 2)
        sw_avx2_run(p);
    else
        p->avx_function(p);
    *score = p->score;
    *max_i = p->max_i;
    *max_j = p->max_j;
    *alignment_offset = p->alignment_offset;
    *cigar_count = p->cigar_count;
    for (uint32_t k = 0; k < p->cigar_count && k < cigar_cap; ++k) cigar[k] = p->cigar[k];
    return 0;
}

int ref_sw_params(int is_assemble, int32_t out[4])
{
    is_assemble = is_assemble ? 1 : 0;
    if (!g_handle[is_assemble]) g_handle[is_assemble] = sw_avx_init((uint8_t)is_assemble);
    if (!g_handle[is_assemble]) return -2;
    out[0] = g_handle[is_assemble]->para.match;
    out[1] = g_handle[is_assemble]->para.mismatch;
    out[2] = g_handle[is_assemble]->para.gap_open;
    out[3] = g_handle[is_assemble]->para.gap_extend;
    return 0;
}

/* time `reps` passes over n alignments on the calling thread; returns seconds.  Used by tools/bench_sw.py for the CPU baseline. */
#include <time.h>
double ref_sw_batch(int is_assemble, const uint8_t *ref_bases, const uint32_t *ref_off, const uint32_t *ref_len,
                    const uint8_t *seq_bases, const uint32_t *seq_off, const uint32_t *seq_len, const uint32_t *pair_ref,
                    const uint32_t *pair_seq, uint64_t n, int reps, int64_t *checksum)
{
    is_assemble = is_assemble ? 1 : 0;
    if (!g_handle[is_assemble]) g_handle[is_assemble] = sw_avx_init((uint8_t)is_assemble);
    p_lib_sw_avx p = g_handle[is_assemble];
    struct timespec t0, t1;
    int64_t sum = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int rep = 0; rep < reps; ++rep)
        for (uint64_t k = 0; k < n; ++k) {
            const uint32_t r = pair_ref[k], s = pair_seq[k];
            memcpy(p->seq1, ref_bases + ref_off[r], ref_len[r]);
            memcpy(p->seq2, seq_bases + seq_off[s], seq_len[s]);
            p->len1 = (int16_t)ref_len[r];
            p->len2 = (int16_t)seq_len[s];
            p->avx_function(p);
            sum += p->score + p->alignment_offset + (int64_t)p->cigar_count;
        }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (checksum) *checksum = sum;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* layout of the REAL handle (smithwaterman_common.h), for tests/test_capi_cpu.py to compare with the adapter's mirror */
#include <stddef.h>
int ref_sw_layout(int32_t *out, int cap)
{
    const int32_t v[] = {(int32_t)sizeof(lib_sw_avx),
                         (int32_t)offsetof(lib_sw_avx, seq1), (int32_t)offsetof(lib_sw_avx, seq2), (int32_t)offsetof(lib_sw_avx, len1),
                         (int32_t)offsetof(lib_sw_avx, len2), (int32_t)offsetof(lib_sw_avx, score), (int32_t)offsetof(lib_sw_avx, max_i),
                         (int32_t)offsetof(lib_sw_avx, max_j), (int32_t)offsetof(lib_sw_avx, cigar), (int32_t)offsetof(lib_sw_avx, cigar_count),
                         (int32_t)offsetof(lib_sw_avx, cigar_max_len), (int32_t)offsetof(lib_sw_avx, alignment_offset),
                         (int32_t)offsetof(lib_sw_avx, max_seq_len), (int32_t)offsetof(lib_sw_avx, para),
                         (int32_t)offsetof(lib_sw_avx, avx_function), (int32_t)offsetof(lib_sw_avx, data)};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int k = 0; k < n && k < cap; ++k) out[k] = v[k];
    return n;
}
