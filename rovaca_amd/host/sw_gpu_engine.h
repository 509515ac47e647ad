// sw_gpu_engine.h -- the MI355X candidate for Rovaca's Smith-Waterman seam.
//
// The reference binds lib_sw_avx.avx_function once, in sw_avx_init (common/base/lib/sw_avx/sw_avx.c:401-450), to
// sw_avx512_run or sw_avx2_run (:488-500), and its two call sites call it once per alignment
// (genotype/utils/alignment_utils.cpp:579; apply/assemble/hc_assemble_cigar_cacl.c:427).  This header offers
//   * sw_hip_run      -- a third candidate with the same signature, one alignment per call (correct, but a device round
//                        trip per alignment: meant for completeness and tests);
//   * sw_hip_align_batch -- the form that pays: every read of a region (or of many regions) against its best haplotype in
//                        one device call.  INTEGRATION.md shows realign_reads_to_their_best_haplotype restructured around it.
// Both return exactly what sw_avx512_run leaves in the handle.
#ifndef ROVACA_AMD_SW_GPU_ENGINE_H_
#define ROVACA_AMD_SW_GPU_ENGINE_H_

#ifdef ROVACA_AMD_IN_ROVACA_TREE
#include "smithwaterman_common.h"
#else
#include "sw_seam.h"
#endif

#ifdef __cplusplus
#include <cstddef>
#include <memory>
#include <vector>

extern "C" {
#endif

// same contract as sw_avx512_run (sw_avx.c:488-500): reads seq1/len1, seq2/len2, para; writes score, max_i, max_j,
// alignment_offset, cigar[0..cigar_count), cigar_count; max(len1, len2) >= max_seq_len -> alignment_offset = 1, nothing else.
// Unrecoverable device failures log to stderr and exit(EXIT_FAILURE), the reference's convention.
void sw_hip_run(p_lib_sw_avx p);

// Every handle as sw_hip_run would leave it, in one device call per parameter set -- the C entry for a caller with several
// alignments at hand, e.g. the assembler, which aligns every haplotype path of a region to the padded reference
// (apply/assemble/hc_assemble_cigar_cacl.c:417-428; INTEGRATION.md has the patch).  Each handle carries its own sequences,
// parameters and CIGAR buffer (handles from sw_avx_init).  Returns 0; on a device failure logs to stderr and returns 1.
int sw_hip_align_many(p_lib_sw_avx *handles, size_t n);

// Points p->avx_function at sw_hip_run when an MI355X is visible and returns 1; returns 0 and leaves the handle alone
// otherwise (the AVX selection of sw_avx_init stays in force; there is no CPU path in this library).
int sw_hip_install(p_lib_sw_avx p);

// Device of the contexts this adapter creates from now on: >= 0 pins it (pass what init_pairhmm_ptr_gpu was given), negative
// (the default) spreads creating threads round-robin over the visible devices, like the PairHMM adapter.  Call before the
// first alignment.
void sw_hip_set_device(int device_id);

// Where the batched realignment seam's host time went so far (always on; seconds summed over threads, the last two entries are counts
// of requests and of device calls): fills up to `cap` entries of `values` / `names` (either may be null), returns the number of entries.
int sw_hip_profile(double *values, const char **names, int cap);

#ifdef __cplusplus
}

namespace rovaca
{

struct SwJob
{
    const uint8_t *ref;  // lib_sw_avx.seq1
    int32_t ref_len;
    const uint8_t *seq;  // lib_sw_avx.seq2
    int32_t seq_len;
};

struct SwAlignment
{
    int32_t score = 0;
    int16_t max_i = 0, max_j = 0, alignment_offset = 0;
    bool too_long = false;          // max(len1, len2) >= 2048: alignment_offset is 1, no CIGAR (sw_avx.c:490-495)
    std::vector<uint32_t> cigar;    // BAM elements, as lib_sw_avx.cigar[0..cigar_count)
};

// All jobs in one device call.  Concurrent callers (the reference's worker pool) are coalesced across regions by a couple
// of dispatcher threads, as for PairHMM; ROVACA_AMD_SW_BATCHING=0 makes every caller use its own device context instead.
// Returns false on a device error (text on stderr).
bool sw_hip_align_batch(const SwJob *jobs, size_t n, const sw_parameters &para, std::vector<SwAlignment> &out);
// The same in two halves: begin queues the alignments with the dispatchers and returns at once, finish waits and hands the results
// over -- begin + finish back to back is sw_hip_align_batch.  A worker can thus have a region's realignment on the device while it
// prepares the next region (tools/adapter_bench.cpp).  The job records are copied; the sequence bytes they point at must stay valid
// until finish.  A ticket destroyed without finish waits for its alignments and drops them.
class SwTicket
{
public:
    SwTicket();
    SwTicket(SwTicket &&) noexcept;
    SwTicket &operator=(SwTicket &&) noexcept;
    ~SwTicket();
    explicit operator bool() const { return st_ != nullptr; }
    struct State;

private:
    std::unique_ptr<State> st_;
    friend SwTicket sw_hip_align_batch_begin(const SwJob *, size_t, const sw_parameters &);
    friend bool sw_hip_align_batch_finish(SwTicket &&, std::vector<SwAlignment> &);
};
SwTicket sw_hip_align_batch_begin(const SwJob *jobs, size_t n, const sw_parameters &para);
bool sw_hip_align_batch_finish(SwTicket &&ticket, std::vector<SwAlignment> &out);
// requests served and device calls made by the dispatchers
void sw_hip_batching_stats(uint64_t *requests, uint64_t *submissions);
int sw_hip_batcher_count();  // queues in use (one per device; 0 before the first batched call)

}  // namespace rovaca
#endif
#endif
