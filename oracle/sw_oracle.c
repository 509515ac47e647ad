/*
 * sw_oracle.c -- CPU restatement of the reference's Smith-Waterman with backtrack.  TEST INFRASTRUCTURE ONLY:
 * only tests/, __graft_entry__.smoke() and tools/ may call it; the product never links or loads it.
 *
 * Follows /root/reference/src/haplotypecaller/common/base/lib/sw_avx/sw_avx.c (an int32 affine-gap aligner with the
 * "soft clip" overhang strategy, derived from Intel GKL):
 *   scoring + backtrack flags   sw_avx.c:41-77 (MAIN_CODE) and :79-225 (smith_waterman_back_track)
 *   best end cell               sw_avx.c:200-217
 *   traceback and CIGAR         sw_avx.c:256-362 (sw_avx_get_cigar), :234-254 (element encoding)
 *   length guard                sw_avx.c:488-500 (sw_avx512_run)
 * The reference walks anti-diagonals with 16 int32 lanes; the arithmetic per cell does not depend on that order, so the
 * restatement fills plain (len1+1) x (len2+1) matrices row by row.  Pinned against the reference's own object code by
 * tests/test_sw_oracle.py (oracle/_ref/libsw_ref.so) and the golden vectors generated from it.
 *
 * Naming as in the reference: seq1 = "ref" (rows i = 1..len1), seq2 = "alt" (columns j = 1..len2).
 */
#include "sw_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define SW_MATRIX_MIN_CUTOFF (-100000000) /* sw_avx.c:18 */
#define SW_LOW_INIT_VALUE    (INT32_MIN / 2) /* sw_avx.c:19 */

/* backtrack flags, smithwaterman_common.h:41-51 */
enum { BT_MATCH = 0, BT_INSERT = 1, BT_DELETE = 2, BT_INSERT_EXT = 4, BT_DELETE_EXT = 8, BT_SOFT_CLIP = 9 };
/* BAM operator codes the elements are emitted with, hc_marco.h:22-26; length << 4 | op (sw_avx.c:234-241) */
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_S = 4 };

static uint32_t make_element(int state, int length)
{
    uint32_t op;
    switch (state) { /* sw_avx.c:243-254 */
        case BT_MATCH: op = OP_M; break;
        case BT_INSERT: op = OP_I; break;
        case BT_DELETE: op = OP_D; break;
        case BT_SOFT_CLIP: op = OP_S; break;
        default: return 0;
    }
    return ((uint32_t)length << 4) | op;
}

int sw_oracle_align(const uint8_t *seq1, int32_t len1, const uint8_t *seq2, int32_t len2, const int32_t params[4],
                    int32_t max_seq_len, sw_oracle_result *res, uint32_t *cigar, uint32_t cigar_cap)
{
    const int32_t w_match = params[0], w_mismatch = params[1], w_open = params[2], w_extend = params[3];
    memset(res, 0, sizeof(*res));
    if (len1 < 1 || len2 < 1) return -1;
    /* sw_avx.c:490-495: sequences of max_seq_len or more are not aligned; the caller sees alignment_offset = 1 */
    if ((len1 > len2 ? len1 : len2) >= max_seq_len) {
        res->alignment_offset = 1;
        res->too_long = 1;
        return 0;
    }
    const int32_t nrow = len1, ncol = len2;
    const size_t W = (size_t)ncol + 1;
    int32_t *H = (int32_t *)calloc((size_t)(nrow + 1) * W, sizeof(int32_t)); /* H[0][*] = H[*][0] = 0 (sw_avx.c:196-197) */
    int32_t *E = (int32_t *)malloc((size_t)(nrow + 1) * sizeof(int32_t));     /* per row, the cell to the left */
    int32_t *F = (int32_t *)malloc(W * sizeof(int32_t));                       /* per column, the cell above */
    uint8_t *bt = (uint8_t *)calloc((size_t)(nrow + 1) * W, 1);
    if (!H || !E || !F || !bt) {
        free(H);
        free(E);
        free(F);
        free(bt);
        return -2;
    }
    for (int32_t j = 0; j <= ncol; ++j) F[j] = SW_LOW_INIT_VALUE; /* sw_avx.c:117-119, :198 */
    for (int32_t i = 1; i <= nrow; ++i) {
        int32_t e = SW_LOW_INIT_VALUE; /* E[i][0], sw_avx.c:120-122, :199 */
        for (int32_t j = 1; j <= ncol; ++j) {
            /* MAIN_CODE, sw_avx.c:41-77 */
            const int32_t ext_h = e + w_extend, open_h = H[i * W + (j - 1)] + w_open;
            const int32_t e11 = open_h > ext_h ? open_h : ext_h;
            int ext = open_h > ext_h ? 0 : BT_INSERT_EXT;
            const int32_t ext_v = F[j] + w_extend, open_v = H[(i - 1) * W + j] + w_open;
            const int32_t f11 = ext_v > open_v ? ext_v : open_v;
            if (!(open_v > ext_v)) ext |= BT_DELETE_EXT;
            const int32_t m11 = H[(i - 1) * W + (j - 1)] + (seq1[i - 1] == seq2[j - 1] ? w_match : w_mismatch);
            int32_t h11 = m11 > SW_MATRIX_MIN_CUTOFF ? m11 : SW_MATRIX_MIN_CUTOFF;
            int b = e11 > h11 ? BT_INSERT : BT_MATCH;
            if (e11 > h11) h11 = e11;
            if (f11 > h11) {
                b = BT_DELETE;
                h11 = f11;
            }
            H[i * W + j] = h11;
            bt[i * W + j] = (uint8_t)(b | ext);
            e = e11;
            F[j] = f11;
        }
        E[i] = e;
    }

    /* best end cell: last row and last column, visited anti-diagonal by anti-diagonal (sw_avx.c:200-217) */
    int32_t max_score = INT32_MIN, max_i = 0, max_j = 0;
    for (int32_t ad = 1; ad <= nrow + ncol; ++ad) {
        if (ad >= nrow + 1) { /* ilo == nrow + 1: cell (nrow, ad - nrow) */
            const int32_t i = nrow, j = ad - nrow;
            const int32_t score = H[i * W + j];
            if (max_score < score || (max_score == score && abs(i - j) < abs(max_i - max_j))) {
                max_score = score;
                max_i = i;
                max_j = j;
            }
        }
        if (ad >= ncol + 1) { /* jhi == ncol + 1: cell (ad - ncol, ncol) */
            const int32_t i = ad - ncol, j = ncol;
            const int32_t score = H[i * W + j];
            if (max_score < score || (max_score == score && (max_j == ncol || abs(i - j) <= abs(max_i - max_j)))) {
                max_score = score;
                max_i = i;
                max_j = j;
            }
        }
    }
    res->score = max_score;
    res->max_i = (int16_t)max_i;
    res->max_j = (int16_t)max_j;

    /* traceback, sw_avx.c:256-362.  (state, length) pairs are collected end to start, merged, then reversed. */
    int16_t *el = (int16_t *)malloc((size_t)(2 * (nrow + ncol) + 8) * 2 * sizeof(int16_t));
    if (!el) {
        free(H);
        free(E);
        free(F);
        free(bt);
        return -2;
    }
    int32_t n_el = 0;
    int32_t i = max_i, j = max_j;
    if (j < ncol) {
        el[0] = BT_SOFT_CLIP;
        el[1] = (int16_t)(ncol - j);
        n_el = 1;
    }
    int state = 0;
    while (i > 0 && j > 0) {
        const int btr = bt[i * W + j];
        if (state == BT_INSERT_EXT) {
            --j;
            el[n_el * 2 - 1]++;
            state = btr & BT_INSERT_EXT;
        }
        else if (state == BT_DELETE_EXT) {
            --i;
            el[n_el * 2 - 1]++;
            state = btr & BT_DELETE_EXT;
        }
        else {
            switch (btr & 3) {
                case BT_MATCH:
                    --i;
                    --j;
                    el[n_el * 2] = BT_MATCH;
                    el[n_el * 2 + 1] = 1;
                    state = 0;
                    ++n_el;
                    break;
                case BT_INSERT:
                    --j;
                    el[n_el * 2] = BT_INSERT;
                    el[n_el * 2 + 1] = 1;
                    state = btr & BT_INSERT_EXT;
                    ++n_el;
                    break;
                case BT_DELETE:
                    --i;
                    el[n_el * 2] = BT_DELETE;
                    el[n_el * 2 + 1] = 1;
                    state = btr & BT_DELETE_EXT;
                    ++n_el;
                    break;
                default: break; /* 3 is never written */
            }
        }
    }
    if (j > 0) {
        el[n_el * 2] = BT_SOFT_CLIP;
        el[n_el * 2 + 1] = (int16_t)j;
        ++n_el;
    }
    res->alignment_offset = (int16_t)i;
    /* merge neighbours with the same state (sw_avx.c:333-347) */
    int32_t last = 0;
    for (int32_t k = 1; k < n_el; ++k) {
        if (el[k * 2] == el[last * 2]) {
            el[last * 2 + 1] = (int16_t)(el[last * 2 + 1] + el[k * 2 + 1]);
        }
        else {
            ++last;
            el[last * 2] = el[k * 2];
            el[last * 2 + 1] = el[k * 2 + 1];
        }
    }
    /* reverse into BAM elements (sw_avx.c:349-358); the reference's buffer holds max_seq_len elements */
    uint32_t count = 0;
    for (int32_t k = last; k >= 0; --k) {
        if (el[2 * k + 1] <= 0 || (int32_t)count + 1 >= max_seq_len) continue;
        if (count < cigar_cap) cigar[count] = make_element(el[2 * k], el[2 * k + 1]);
        ++count;
    }
    res->cigar_count = count;
    free(el);
    free(H);
    free(E);
    free(F);
    free(bt);
    return count > cigar_cap ? -3 : 0;
}
