// host_capi.cpp -- plain-C entry points over the C++ host adapter, so that it can be driven without Rovaca's types
// (the Python tests, other FFI users).  One call = one region = one rovaca::call_pairhmm invocation.
#include <cmath>
#include <cstring>
#include <memory>
#include <memory_resource>
#include <vector>

#include "pairhmm_gpu_engine.h"
#include "pairhmm_host_logic.h"

extern "C" {

// host logic, exposed for CPU-side parity tests (no device involved)
void rovaca_amd_modify_read_qualities(uint8_t *qual, uint32_t n, uint8_t mapq, int32_t min_q)
{
    rovaca_amd::modify_read_qualities(qual, n, mapq, min_q);
}
int32_t rovaca_amd_tandem_repeat_length(const uint8_t *bases, uint32_t n, uint32_t offset)
{
    return rovaca_amd::tandem_repeat_length(bases, n, offset);
}
void rovaca_amd_apply_pcr_error_model(const uint8_t *bases, uint8_t *gap_open, uint32_t n, int model)
{
    rovaca_amd::apply_pcr_error_model(bases, gap_open, n, model);
}
void rovaca_amd_normalize_likelihoods(double *lik, uint32_t n_reads, uint32_t n_haps)
{
    rovaca_amd::normalize_likelihoods(lik, n_reads, n_haps);
}
void rovaca_amd_marginal_likelihoods_direct(const double *old, uint32_t n_old, uint32_t stride, uint32_t evidence_count,
                                            const int32_t *old2new, uint32_t n_new, double *out)
{
    rovaca_amd::marginal_likelihoods_direct(old, n_old, stride, evidence_count, old2new, n_new, out);
}
// the same through the adapter-level function, which speaks the reference's types (pmr vectors from an arena)
void rovaca_amd_marginalize_through_adapter(const double *old, uint32_t n_old, uint32_t stride, uint32_t evidence_count,
                                            const int32_t *old2new, uint32_t n_new, double *out)
{
    std::pmr::monotonic_buffer_resource arena;
    rovaca::DoubleVector2D values(&arena);
    for (uint32_t a = 0; a < n_old; ++a)  // (the pmr outer vector hands its resource to the row it constructs)
        values.emplace_back(old + (size_t)a * stride, old + (size_t)(a + 1) * stride);
    rovaca::Int32Vector map(old2new, old2new + n_old, &arena);
    const rovaca::DoubleVector2D res = rovaca::marginal_likelihoods_direct(values, evidence_count, n_new, map, &arena);
    for (uint32_t a = 0; a < n_new; ++a)
        for (uint32_t r = 0; r < evidence_count; ++r) out[(size_t)a * evidence_count + r] = res[a][r];
}
uint32_t rovaca_amd_poorly_modelled_reads(const double *lik, const int32_t *read_len, uint32_t n_reads, uint32_t n_haps,
                                          uint8_t *drop)
{
    std::vector<size_t> d = rovaca_amd::poorly_modelled_reads(lik, read_len, n_reads, n_haps);
    memset(drop, 0, n_reads);
    for (size_t r : d) drop[r] = 1;
    return (uint32_t)d.size();
}

// 1 when an MI355X engine was installed into rovaca::call_pairhmm, 0 when no device is visible
int rovaca_amd_init_pairhmm_ptr_gpu(int device_id) { return rovaca::init_pairhmm_ptr_gpu(device_id) ? 1 : 0; }

void rovaca_amd_batching_stats(uint64_t *regions, uint64_t *submissions) { rovaca::gpu_pairhmm_batching_stats(regions, submissions); }
int rovaca_amd_batcher_stats(int batcher, int *device, uint64_t *regions, uint64_t *submissions)
{
    return rovaca::gpu_pairhmm_batcher_stats(batcher, device, regions, submissions);
}

// One region through rovaca::call_pairhmm (must have been installed).  Inputs are raw BAM-style reads (bases, quals,
// mapping quality) and haplotype strings; out_hap_read is [n_haps][n_reads] (tail NaN), survivors[k] the original index
// of the k-th read left in the read vector.  Returns the number of surviving reads, or -1 if the engine is not installed.
int32_t rovaca_amd_call_region(const uint8_t *read_bases, const uint8_t *read_qual, const uint32_t *read_off,
                               const uint32_t *read_len, const uint8_t *read_mq, uint32_t n_reads, const uint8_t *hap_bases,
                               const uint32_t *hap_off, const uint32_t *hap_len, uint32_t n_haps, int32_t min_quality_threshold,
                               int pcr_model, double *out_hap_read, uint32_t *survivors)
{
    using namespace rovaca;
    if (!call_pairhmm) return -1;
    std::pmr::monotonic_buffer_resource arena;  // stands in for the per-region RovacaMemoryPool
    std::vector<ReadRecord> reads;
    std::vector<Haplotype> haps;
    reads.reserve(n_reads);
    haps.reserve(n_haps);
    for (uint32_t r = 0; r < n_reads; ++r)
        reads.emplace_back(read_bases + read_off[r], read_qual + read_off[r], read_len[r], read_mq[r]);
    for (uint32_t h = 0; h < n_haps; ++h) haps.emplace_back(hap_bases + hap_off[h], hap_len[h]);
    ReadVector rs{&arena};
    HaplotypeVector hs{&arena};
    for (auto &r : reads) rs.push_back(&r);
    for (auto &h : haps) hs.push_back(&h);
    DoubleVector2D lik = call_pairhmm(hs, rs, min_quality_threshold, static_cast<PcrIndelModel>(pcr_model), &arena);
    for (size_t k = 0; k < (size_t)n_haps * n_reads; ++k) out_hap_read[k] = NAN;
    for (size_t h = 0; h < lik.size(); ++h)
        for (size_t k = 0; k < lik[h].size(); ++k) out_hap_read[h * n_reads + k] = lik[h][k];
    for (size_t k = 0; k < rs.size(); ++k) survivors[k] = (uint32_t)(rs[k] - reads.data());
    return (int32_t)rs.size();
}

// Several regions through the begin / finish form of the seam, `depth` of them in flight at a time (depth 1 = begin and finish
// back to back = the synchronous call).  Reads and haplotypes of all regions are concatenated; region g owns reads
// [region_read[g], region_read[g+1]) and haplotypes [region_hap[g], region_hap[g+1]).  out holds the matrices one after the
// other ([n_haps][n_reads] each, tail NaN), survivors the surviving original read indices of each region at its read offset
// (relative to the region), n_survivors[g] their number.  Returns 0, or -1 if the engine is not installed.
int32_t rovaca_amd_call_regions_in_flight(const uint8_t *read_bases, const uint8_t *read_qual, const uint32_t *read_off,
                                          const uint32_t *read_len, const uint8_t *read_mq, const uint8_t *hap_bases,
                                          const uint32_t *hap_off, const uint32_t *hap_len, const uint32_t *region_read,
                                          const uint32_t *region_hap, uint32_t n_regions, int32_t min_quality_threshold, int pcr_model,
                                          uint32_t depth, double *out, uint32_t *survivors, uint32_t *n_survivors)
{
    using namespace rovaca;
    if (!gpu_pairhmm_active()) return -1;
    if (depth == 0) depth = 1;
    struct InFlight
    {
        std::pmr::monotonic_buffer_resource arena;
        std::vector<ReadRecord> reads;
        std::vector<Haplotype> haps;
        ReadVector rs{&arena};
        HaplotypeVector hs{&arena};
        PairhmmTicket ticket;
        uint32_t region = 0;
    };
    std::vector<size_t> out_at(n_regions + 1, 0);
    for (uint32_t g = 0; g < n_regions; ++g)
        out_at[g + 1] = out_at[g] + (size_t)(region_read[g + 1] - region_read[g]) * (region_hap[g + 1] - region_hap[g]);
    auto collect = [&](InFlight &f) {
        const uint32_t g = f.region, r0 = region_read[g], nr = region_read[g + 1] - r0, nh = region_hap[g + 1] - region_hap[g];
        DoubleVector2D lik = call_gpu_pairhmm_finish(std::move(f.ticket), f.rs, &f.arena);
        double *dst = out + out_at[g];
        for (size_t k = 0; k < (size_t)nh * nr; ++k) dst[k] = NAN;
        for (size_t h = 0; h < lik.size(); ++h)
            for (size_t k = 0; k < lik[h].size(); ++k) dst[h * nr + k] = lik[h][k];
        for (size_t k = 0; k < f.rs.size(); ++k) survivors[r0 + k] = (uint32_t)(f.rs[k] - f.reads.data());
        n_survivors[g] = (uint32_t)f.rs.size();
    };
    std::vector<std::unique_ptr<InFlight>> ring(depth);
    for (uint32_t g = 0; g < n_regions + depth - 1; ++g) {
        std::unique_ptr<InFlight> &slot = ring[g % depth];
        if (slot) {
            collect(*slot);
            slot.reset();
        }
        if (g < n_regions) {
            slot.reset(new InFlight());
            InFlight &f = *slot;
            f.region = g;
            f.reads.reserve(region_read[g + 1] - region_read[g]);
            f.haps.reserve(region_hap[g + 1] - region_hap[g]);
            for (uint32_t r = region_read[g]; r < region_read[g + 1]; ++r)
                f.reads.emplace_back(read_bases + read_off[r], read_qual + read_off[r], read_len[r], read_mq[r]);
            for (uint32_t h = region_hap[g]; h < region_hap[g + 1]; ++h) f.haps.emplace_back(hap_bases + hap_off[h], hap_len[h]);
            for (auto &r : f.reads) f.rs.push_back(&r);
            for (auto &h : f.haps) f.hs.push_back(&h);
            f.ticket = call_gpu_pairhmm_begin(f.hs, f.rs, min_quality_threshold, static_cast<PcrIndelModel>(pcr_model));
        }
    }
    for (auto &slot : ring)
        if (slot) collect(*slot);
    return 0;
}

}  // extern "C"
