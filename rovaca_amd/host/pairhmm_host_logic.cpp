// pairhmm_host_logic.cpp -- see pairhmm_host_logic.h.  Written from the behaviour of the reference functions cited at each
// definition; quirks that change results are kept and called out.
#include "pairhmm_host_logic.h"

#include <algorithm>
#include <limits>
#include <cmath>
#include <cstring>

namespace rovaca_amd
{

void modify_read_qualities(uint8_t *qual, uint32_t n, uint8_t mapq, int32_t min_q)
{
    for (uint32_t k = 0; k < n; ++k) {
        const uint8_t capped = std::min(qual[k], mapq);
        qual[k] = (int32_t)capped < min_q ? kMinQuality : capped;
    }
}

namespace
{

// How many back-to-back copies of unit[0..ulen) tile text[0..tlen), counted from the front or from the back
// (pairhmm_internal.cpp:195-229, find_number_of_repetitions).
int32_t count_tiled_units(const uint8_t *unit, int32_t ulen, const uint8_t *text, int32_t tlen, bool from_front)
{
    if (tlen == 0) return 0;
    int32_t copies = 0;
    const int32_t last = tlen - ulen;
    if (from_front) {
        for (int32_t at = 0; at <= last; at += ulen, ++copies)
            if (memcmp(text + at, unit, (size_t)ulen) != 0) break;
    }
    else {
        for (int32_t at = last; at >= 0; at -= ulen, ++copies)
            if (memcmp(text + at, unit, (size_t)ulen) != 0) break;
    }
    return copies;
}

const int kPcrHostile[kMaxRepeatLength + 1] = {40, 40, 39, 38, 37, 36, 34, 32, 28, 23, 17, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10};
const int kPcrAggressive[kMaxRepeatLength + 1] = {40, 40, 40, 39, 39, 39, 38, 38, 37, 37, 36, 35, 34, 33, 32, 30, 28, 26, 23, 20, 17};
const int kPcrConservative[kMaxRepeatLength + 1] = {40, 40, 40, 40, 39, 39, 39, 39, 39, 38, 38, 38, 37, 37, 37, 36, 36, 35, 34, 33, 33};

}  // namespace

// pairhmm_internal.cpp:65-119 (find_tandem_repeat_units).  The reference looks for the shortest unit (1..8 bases) that
// repeats at least twice ending at `offset` (backward) and starting at offset+1 (forward), then -- because the backward
// unit's recorded start (offset - str - 1) can never equal the forward unit's start (offset + 1) -- ALWAYS re-counts the
// backward run with the forward unit (:108).  When no unit repeats, the count of the last unit length tried is what is
// left in the variable and is what gets used.  Both are reproduced.
int32_t tandem_repeat_length(const uint8_t *bases, uint32_t n, uint32_t offset_u)
{
    const int32_t offset = (int32_t)offset_u, len = (int32_t)n;
    const int32_t prefix = offset + 1;  // bases[0..offset]
    int32_t backward = 0;
    for (int32_t unit = 1; unit <= kMaxStrUnitLength && unit <= prefix; ++unit) {
        backward = count_tiled_units(bases + prefix - unit, unit, bases, prefix, false);
        if (backward > 1) break;
    }
    int32_t total = backward;
    if (offset < len - 1) {
        const int32_t suffix = len - prefix;  // bases[offset+1..len)
        int32_t forward = 0, fw_unit = 1;
        for (int32_t unit = 1; unit <= kMaxStrUnitLength && unit <= suffix; ++unit) {
            forward = count_tiled_units(bases + prefix, unit, bases + prefix, suffix, true);
            if (forward > 1) {
                fw_unit = unit;
                break;
            }
        }
        backward = count_tiled_units(bases + prefix, fw_unit, bases, prefix, false);
        total = backward + forward;
    }
    return std::min(total, kMaxRepeatLength);
}

void apply_pcr_error_model(const uint8_t *bases, uint8_t *gap_open, uint32_t n, int model)
{
    const int *cap = model == 1 ? kPcrHostile : model == 2 ? kPcrAggressive : model == 3 ? kPcrConservative : nullptr;
    if (!cap) return;
    for (uint32_t at = 0; at + 1 < n; ++at) {  // the last base keeps its gap-open (loop is i = 1..n-1 over i-1)
        const int limit = cap[tandem_repeat_length(bases, n, at)];
        if ((int)gap_open[at] >= limit) gap_open[at] = (uint8_t)limit;
    }
}

void normalize_likelihoods(double *lik, size_t n_reads, size_t n_haps)
{
    for (size_t r = 0; r < n_reads; ++r) {
        double *row = lik + r * n_haps;
        const double floor_v = *std::max_element(row, row + n_haps) + kMaximumBestAltLikelihoodDifference;
        for (size_t h = 0; h < n_haps; ++h)
            if (row[h] < floor_v) row[h] = floor_v;  // NaN < x is false: NaN stays, as in the reference
    }
}

std::vector<size_t> poorly_modelled_reads(const double *lik, const int32_t *read_len, size_t n_reads, size_t n_haps)
{
    std::vector<size_t> drop;
    for (size_t r = 0; r < n_reads; ++r) {
        const double *row = lik + r * n_haps;
        const double best = *std::max_element(row, row + n_haps);
        const double worst_ok =
            std::min(kMaximumExpectedErrorPerRead, std::ceil(read_len[r] * kExpectedErrorRatePerBase)) * kLog10QualityPerBase;
        if (best < worst_ok) drop.push_back(r);
    }
    return drop;
}

void marginal_likelihoods_direct(const double *old, size_t n_old, size_t stride, size_t evidence_count, const int32_t *old2new,
                                 size_t n_new, double *out)
{
    std::fill(out, out + n_new * evidence_count, -std::numeric_limits<double>::infinity());  // allele_likelihoods.hpp:712
    for (size_t a = 0; a < n_old; ++a) {  // old alleles in ascending order, as the reference walks them (:715)
        const int32_t target = old2new[a];
        if (target == kMissingAlleleIndex) continue;
        const double *src = old + a * stride;
        double *dst = out + (size_t)target * evidence_count;
        for (size_t r = 0; r < evidence_count; ++r)
            if (src[r] > dst[r]) dst[r] = src[r];  // strict: a NaN is never taken, the first of two equal values stays (:724-727)
    }
}

}  // namespace rovaca_amd
