// pairhmm_gpu_engine.h -- the MI355X candidate for Rovaca's PairHMM seam.
//
// The reference selects its likelihood engine once, in rovaca::init_pairhmm_ptr (pairhmm/pairhmm_engine.cpp:21-53), by
// assigning the process-global function pointer rovaca::call_pairhmm (pairhmm/pairhmm_engine.h:14-15).  This header
// declares a third candidate with the identical C++ signature, plus the init hook that installs it.  See
// INTEGRATION.md for the patch to pairhmm_engine.cpp.
#ifndef ROVACA_AMD_PAIRHMM_GPU_ENGINE_H_
#define ROVACA_AMD_PAIRHMM_GPU_ENGINE_H_

#include <memory>

#ifdef ROVACA_AMD_IN_ROVACA_TREE
#include "../common/enum.h"
#include "../genotype/forward.h"
#else
#include "rovaca_seam.h"
#endif

namespace rovaca
{

// Same contract as call_old_pairhmm / call_new_pairhmm002 (pairhmm/pairhmm_internal.cpp:505-594, :596-753):
//   * empty hs or rs -> {} (:508-510);
//   * returns [hap][read] log10 likelihoods allocated from `pool`, row width = the ORIGINAL read count, valid prefix =
//     surviving reads, tail NaN (:516, :589-593);
//   * `rs` is mutated: poorly modelled reads are erased in place (:161-182) and the caller uses the shrunken list;
//   * reads need not be sorted by length (the AVX-512 driver requires it, :602; this engine bins by length itself);
//   * callable concurrently from many worker threads (haplotypecaller_engine.cpp:119): each thread gets its own device
//     context (stream + staging) on first use.
DoubleVector2D call_gpu_pairhmm(const HaplotypeVector &hs, ReadVector &rs, int32_t min_quality_threshold,
                                PcrIndelModel pcr_option, pMemoryPool pool);

// The same evaluation in two halves ("... one GPU submission; return futures").  begin flattens the region,
// hands it to the cross-region batcher and returns at once; finish waits for it, runs H8 (normalise, filter, transpose) and
// returns what call_gpu_pairhmm returns -- begin + finish back to back IS call_gpu_pairhmm.  In between the worker is free:
// with the next region's begin issued before this region's finish, a worker keeps two regions in flight and the device sees
// batches twice as large (tools/adapter_bench.cpp measures both forms; INTEGRATION.md shows the caller-side loop).  The
// ticket owns copies of everything the device reads; `rs` handed to finish must be the list begin saw (it is shrunk there).
// A ticket that is destroyed without finish waits for its region and discards it.
class PairhmmTicket
{
public:
    PairhmmTicket();
    PairhmmTicket(PairhmmTicket &&) noexcept;
    PairhmmTicket &operator=(PairhmmTicket &&) noexcept;
    ~PairhmmTicket();
    explicit operator bool() const { return st_ != nullptr; }  // false: an empty region (finish returns {})
    struct State;

private:
    std::unique_ptr<State> st_;
    friend PairhmmTicket call_gpu_pairhmm_begin(const HaplotypeVector &, const ReadVector &, int32_t, PcrIndelModel);
    friend DoubleVector2D call_gpu_pairhmm_finish(PairhmmTicket &&, ReadVector &, pMemoryPool);
};
PairhmmTicket call_gpu_pairhmm_begin(const HaplotypeVector &hs, const ReadVector &rs, int32_t min_quality_threshold, PcrIndelModel pcr_option);
DoubleVector2D call_gpu_pairhmm_finish(PairhmmTicket &&ticket, ReadVector &rs, pMemoryPool pool);

// Mirrors the seam's global: what haplotypecaller_engine.cpp:250 calls.
extern DoubleVector2D (*call_pairhmm)(const HaplotypeVector &hs, ReadVector &rs, int32_t min_quality_threshold,
                                      PcrIndelModel pcr_option, pMemoryPool pool);

// Installs call_gpu_pairhmm into call_pairhmm.  Returns false (and leaves the pointer alone) when no MI355X is visible,
// so that the reference's init_pairhmm_ptr can fall through to its AVX selection; it never substitutes a CPU path of its
// own.  `device_id` < 0 means round-robin over all visible devices per calling thread.
bool init_pairhmm_ptr_gpu(int device_id = -1);

// Regions evaluated and device submissions made by the cross-region batcher: workers that call
// call_pairhmm concurrently are coalesced into one device batch per round trip.  ROVACA_AMD_PAIRHMM_BATCHING=0 disables it.
void gpu_pairhmm_batching_stats(uint64_t *regions, uint64_t *submissions);
// the same per batcher (one per device in use); returns the number of batchers whatever `batcher` is
int gpu_pairhmm_batcher_stats(int batcher, int *device, uint64_t *regions, uint64_t *submissions);

// Where the adapter's host time went so far: seconds per phase summed over all worker and dispatcher threads (names[i] describes
// seconds[i]); returns the number of phases.  tools/adapter_bench.cpp prints it; profiles/ keeps a run.
int gpu_pairhmm_profile(double *seconds, const char **names, int cap);

// the step right after the seam.  One sample of AlleleLikelihoods::marginal_likelihoods_direct
// (genotype/allele_likelihoods.hpp:702-732): the read x haplotype matrix collapsed onto a smaller allele set by a maximum
// over the old alleles that map to each new one.  Same values, bit for bit (a maximum of doubles); the rows of the result
// come from `pool` and are `evidence_count` long.  Measured (tools/measure_marginalize.py): a region's matrix is a few
// kilobytes and the host loop takes about a microsecond, a device round trip tens -- it runs on the host, next to H8.
DoubleVector2D marginal_likelihoods_direct(const DoubleVector2D &old_sample_values, size_t evidence_count, size_t new_allele_count,
                                           const Int32Vector &old2new_idx_map, pMemoryPool pool);

// true once init_pairhmm_ptr_gpu succeeded (the caller skips its AVX-512 read pre-sort, haplotypecaller_engine.cpp:245-248)
bool gpu_pairhmm_active();

}  // namespace rovaca
#endif
