// pairhmm_host_logic.h -- the integer / table work the reference does on the host around its PairHMM kernels
// (SURVEY.md rows H3, H4, H8), as free functions over plain buffers.  Product code (never the oracle).
#ifndef ROVACA_AMD_PAIRHMM_HOST_LOGIC_H_
#define ROVACA_AMD_PAIRHMM_HOST_LOGIC_H_
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace rovaca_amd
{

// adapter.h:5-7, :20-23
constexpr uint8_t kMinQuality = 6;
constexpr int kMaxRepeatLength = 20;
constexpr int kMaxStrUnitLength = 8;
constexpr uint8_t kDefaultGapOpen = '-';      // 45, read_record.cpp:191-196
constexpr uint8_t kDefaultGapCont = '+' - 33;  // 10, read_record.cpp:205-210
constexpr double kLog10QualityPerBase = -4.0;
constexpr double kExpectedErrorRatePerBase = 0.02;
constexpr double kMaximumExpectedErrorPerRead = 2.0;
constexpr double kMaximumBestAltLikelihoodDifference = -4.5;

// H3  pairhmm_internal.cpp:434-439: q = min(q, MQ); q < threshold -> 6
void modify_read_qualities(uint8_t *qual, uint32_t n, uint8_t mapping_quality, int32_t min_quality_threshold);

// H4  pairhmm_internal.cpp:65-119 / :195-229 / :121-159.  model: 0 NONE, 1 HOSTILE, 2 AGGRESSIVE, 3 CONSERVATIVE
int32_t tandem_repeat_length(const uint8_t *bases, uint32_t n, uint32_t offset);
void apply_pcr_error_model(const uint8_t *bases, uint8_t *gap_open, uint32_t n, int model);

// H8  pairhmm_internal.cpp:53-63 and :161-182 on a row-major [n_reads][n_haps] matrix
void normalize_likelihoods(double *lik, size_t n_reads, size_t n_haps);
// returns indices of reads to DROP, ascending
std::vector<size_t> poorly_modelled_reads(const double *lik, const int32_t *read_len, size_t n_reads, size_t n_haps);


// next-3  genotype/allele_likelihoods.hpp:702-732 (marginal_likelihoods_direct), one sample:
// out[a'][r] = max over the old alleles a with old2new[a] == a' of old[a][r], -inf where no old allele maps to a'; entries
// of old2new equal to kMissingAlleleIndex (s_missing_index, :26) are skipped.  old is [n_old][stride] (the reference's rows
// are as wide as the ORIGINAL read count; only the first evidence_count values are evidence), out is [n_new][evidence_count].
constexpr int32_t kMissingAlleleIndex = -1;
void marginal_likelihoods_direct(const double *old, size_t n_old, size_t stride, size_t evidence_count, const int32_t *old2new,
                                 size_t n_new, double *out);

// Which devices the adapters put a batcher on when the caller asks for all of them (init_pairhmm_ptr_gpu(-1), the Smith-Waterman
// seam's default): every visible device once.  Test hook: `env_name` = "0,0,0" lists device ids explicitly, the same one several
// times if need be -- N batchers on ONE card, so that the N-device path (round-robin over batchers, one stage pool and dispatcher set
// each) runs on a one-GPU box.  Honoured only with PHMM_ENABLE_TEST_HOOKS=1 AND PHMM_MULTI_ALLOW_SHARED_DEVICE=1 (the same pair that
// lets phmm_create_multi take a device twice); ids outside the visible devices are dropped.
inline std::vector<int> batcher_devices(const char *env_name, int n_visible)
{
    std::vector<int> out;
    const char *master = getenv("PHMM_ENABLE_TEST_HOOKS"), *shared = getenv("PHMM_MULTI_ALLOW_SHARED_DEVICE"), *list = getenv(env_name);
    if (list && shared && master && master[0] && !(master[0] == '0' && master[1] == 0)) {
        for (const char *q = list; *q;) {
            char *end = nullptr;
            const long d = strtol(q, &end, 10);
            if (end == q) break;
            if (d >= 0 && d < n_visible) out.push_back((int)d);
            q = *end ? end + 1 : end;
        }
    }
    if (out.empty())
        for (int d = 0; d < n_visible; ++d) out.push_back(d);
    return out;
}

}  // namespace rovaca_amd
#endif
