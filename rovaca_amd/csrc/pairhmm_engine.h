// pairhmm_engine.h -- internals shared by the translation units behind include/pairhmm_hip.h (pairhmm_capi.cpp: contexts, resident
// batches, launches, multi-device handle; pairhmm_stage.cpp: region-descriptor batches through pinned staging; pairhmm_binning.cpp:
// the length binning, host only).  Not part of the ABI.
#ifndef ROVACA_AMD_PAIRHMM_ENGINE_H_
#define ROVACA_AMD_PAIRHMM_ENGINE_H_
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pairhmm_hip.h"
#include "pairhmm_binning.h"
#include "pairhmm_device.h"

namespace phmm_detail
{
extern thread_local char g_err[512];           // text of the calling thread's last failure (phmm_last_error)
int fail(int code, const char *fmt, ...);      // formats g_err, returns code
}  // namespace phmm_detail
using phmm_detail::fail;

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess) return fail(PHMM_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));        \
    } while (0)

template <typename T>
struct DevBuf
{
    T *p = nullptr;
    size_t n = 0;    // elements in use
    size_t cap = 0;  // elements allocated (grow-only, so a context's scratch batch stops allocating after a few calls)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    void reset()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = cap = 0;
    }
    hipError_t alloc(size_t count)
    {
        if (count <= cap) {
            n = count;
            return hipSuccess;
        }
        if (p) (void)hipFree(p);
        p = nullptr;
        n = cap = 0;
        const size_t want = count + count / 4 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
        if (e != hipSuccess) return e;
        cap = want;
        n = count;
        return hipSuccess;
    }
    hipError_t upload(const T *src, size_t count, hipStream_t s)
    {
        hipError_t e = alloc(count);
        if (e != hipSuccess || count == 0) return e;
        return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s);
    }
};

// a typed window into the resident batch's single input blob (not owned)
template <typename T>
struct DevView
{
    T *p = nullptr;
    size_t n = 0;
};

// pinned host staging, grow-only
struct HostBuf
{
    void *p = nullptr;
    size_t cap = 0;
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf()
    {
        if (p) (void)hipHostFree(p);
    }
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
};


struct phmm_context
{
    // BASELINE config 1: a context made by phmm_create_cpu computes on the host (pairhmm_cpu.cpp) and owns no HIP object
    bool cpu = false;
    unsigned cpu_threads = 1;
    bool cpu_simd = true;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevBuf<float> ph2pr_f32, mm_f32;
    DevBuf<double> ph2pr_f64, mm_f64;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the per-K launches of one pass are independent: they are spread over a few side streams so that the tail of one
    // launch overlaps the head of the next (mixed-length batches have up to 17 launches per pass)
    static constexpr int kSideStreams = 3;  // created
    int n_side = 2;                         // used (PHMM_SIDE_STREAMS overrides: tuning; 2 and 3 measure the same, 0 loses 40 % on config 5)
    hipStream_t side[kSideStreams] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[kSideStreams] = {nullptr, nullptr, nullptr};
    struct phmm_resident *scratch = nullptr;  // reused by phmm_compute: device buffers survive between calls
    // fp64 first: the share of long pairs (striped ranges) that fp64 decided in the last pass any batch of this context finished --
    // what a new batch's first pass goes by
    float f64_share = 0.f;
    bool f64_share_known = false;
    // big one-shot batches are cut into chunks that go round three resident slots: chunk k is binned, packed and uploaded (on
    // side[2]) while chunk k - 1 is swept and chunk k - 2 is converted on the host by a collector thread (pairhmm_capi.cpp:
    // compute_chunked); the sums come back on a stream of their own so that the next sweep does not queue behind the copy
    static constexpr int kChunkSlots = 16;  // (six until round 4: with a slot per chunk no chunk ever waits for an earlier one to be collected)
    struct phmm_resident *chunk_slot[kChunkSlots] = {};
    hipEvent_t chunk_up[kChunkSlots] = {}, chunk_swept[kChunkSlots] = {}, chunk_done[kChunkSlots] = {};
    hipStream_t down = nullptr;  // result copies of the chunked call
    // the chunked call's second sweep lane (odd chunks): a stream set of its own, swapped in for the duration of a chunk's enqueue, so
    // that one chunk's tail and the next one's head share the chip (compute_chunked)
    struct SweepLane
    {
        hipStream_t stream = nullptr, side[kSideStreams] = {nullptr, nullptr, nullptr};
        hipEvent_t fork = nullptr, join[kSideStreams] = {nullptr, nullptr, nullptr}, done = nullptr;
    } lane2, rescue_lane;  // rescue_lane: where the chunk pipeline's collector re-runs a chunk's underflowed pairs
    phmm_detail::HostPool *pack_pool = nullptr, *fin_pool = nullptr;  // helper threads of the two sides of that pipeline
    unsigned pack_pool_asked = 0;  // helpers pack_pool was made for (it may have fewer: thread creation can fail)
    phmm_detail::HostPool *prep_pool = nullptr;  // its prepare workers: chunks k + 1, k + 2, ... are packed and binned while chunk k is enqueued
    phmm_detail::BinScratch prep_scratch[8];     // one binning scratch per prepare worker
    phmm_detail::BinScratch bin_scratch;       // host scratch of the length binning, kept between calls
    std::vector<hipEvent_t> timing;            // event pool of phmm_sweep_timed
    struct phmm_stage *region_stage = nullptr; // phmm_compute_regions: the stage host buffers are copied into (pairhmm_stage.cpp)
    DevBuf<uint8_t> marg_dev;                  // phmm_marginalize: [old matrix][index map][result]
    HostBuf marg_stage;
};

// Geometry of a striped launch.  lanes: 16 puts four pairs in a wave, 64 gives a whole wave to one pair (chosen when the
// launch has so few pairs that four-per-wave would leave most of the chip idle and the reads are long enough to fill 64-lane
// stripes).  waves: how many waves pipeline the stripes of the same pairs -- enough stripes to keep them busy, a haplotype
// long enough for the pipeline lag (4 or 6 supersteps of 32 columns per wave), and only when the batch is small.
struct TeamShape
{
    int lanes, waves;
};
static TeamShape pick_team_shape(uint32_t n_pairs, uint32_t max_read, uint32_t max_hap, uint32_t rows_per_lane)
{
    TeamShape sh{16, 1};
    if ((n_pairs + 3) / 4 < 1024 && max_read > 64u * rows_per_lane) sh.lanes = 64;
    // long reads take a wave each whatever the batch size: the zero-superstep shortcut works per wave, and four pairs in a
    // wave are rarely all zero at the same columns (measured on 6 000 pairs of 2 000 x 2 400: fp32 4.7 -> 3.5 ms, fp64 15.7
    // -> 12.9 ms; on 1 000 x 1 200 the fp64 re-run gains, the fp32 sweep does not)
    if (max_read >= (rows_per_lane >= 8 ? 1536u : 768u)) sh.lanes = 64;
    if (const char *f = phmm_hook("PHMM_FORCE_TEAM_LANES")) {  // tests
        const int l = atoi(f);
        if (l == 16 || l == 64) sh.lanes = l;
    }
    const uint32_t per_wave = 64u / (uint32_t)sh.lanes;
    const uint32_t batches = (n_pairs + per_wave - 1) / per_wave;
    // teams of several waves run the band kernel, whose stripes are kBandRows rows per lane tall whatever the one-wave shape is
    const uint32_t stripes = (max_read + sh.lanes * phmm::kBandRows - 1) / (sh.lanes * phmm::kBandRows);
    const uint32_t supersteps = (max_hap + (uint32_t)sh.lanes - 1 + 31) / 32;
    const uint32_t lag = sh.lanes == 64 ? 6u : 4u;  // team_lag<G>() of the kernel
    // (one pair per wave: since all-zero supersteps are skipped only one or two waves of a team have work at any time, and
    // eight waves per pair measured slower than four on config 5)
    if (sh.lanes == 16 && batches < 2048 && stripes >= 8 && supersteps >= lag * 8 + 12)
        sh.waves = 8;
    else if (batches < 4096 && stripes >= 4 && supersteps >= lag * 4 + 12)
        sh.waves = 4;
    if (const char *f = phmm_hook("PHMM_FORCE_TEAM_WAVES")) {  // tests: exercise every team width on small batches
        const int w = atoi(f);
        if ((w == 1 || w == 4 || w == 8) && supersteps >= lag * w + 12u) sh.waves = w;
    }
    return sh;
}

// Striped pairs are binned into up to four size ranges so that each can get the team shape that fits it.  When neighbouring ranges
// get the SAME shape anyway (config 5: every range runs four waves of 64 lanes), separate launches only interleave their workgroups on
// the chip in no particular order, and the longest pairs -- the pass lasts as long as they do -- start late; one launch hands its
// workgroups out in the processing order, longest first.  Measured on config 5: 13.45 -> 12.7 ms per pass.  The merged range keeps the
// first range's rescue queue, so the fp64 re-run is one launch as well.  PHMM_NO_STRIPED_MERGE: the A/B hook.
static void merge_striped_bins_of_one_shape(std::vector<Bin> &bins)
{
    if (bins.size() < 2 || phmm_hook("PHMM_NO_STRIPED_MERGE")) return;
    auto shape_of = [](const Bin &b) { return pick_team_shape(b.count, b.max_read_len, b.max_hap_len, phmm::kStripedRowsF32); };
    std::vector<Bin> out;
    out.reserve(bins.size());
    for (const Bin &b : bins) {
        if (b.rows_per_lane == 0 && !out.empty() && out.back().rows_per_lane == 0 && out.back().first + out.back().count == b.first) {
            Bin u = out.back();
            u.count += b.count;
            u.lds_words_per_group = std::max(u.lds_words_per_group, b.lds_words_per_group);
            u.max_hap_len = std::max(u.max_hap_len, b.max_hap_len);
            u.max_read_len = std::max(u.max_read_len, b.max_read_len);
            const TeamShape sa = shape_of(out.back()), sb = shape_of(b), su = shape_of(u);
            if (sa.lanes == sb.lanes && sa.waves == sb.waves && su.lanes == sa.lanes && su.waves == sa.waves) {
                out.back() = u;  // (first and queue stay the first range's)
                continue;
            }
        }
        out.push_back(b);
    }
    bins.swap(out);
}

// a batch kept by a CPU context: deep copies of the tables (prepared: H3/H4 already applied when the batch asked for it)
struct CpuBatch
{
    std::vector<uint8_t> bases, qual, ins, del, gcp, haps;
    std::vector<uint32_t> roff, rlen, hoff, hlen, pr, ph;
    std::vector<float> raw32;
    std::vector<double> raw64;
    std::vector<uint8_t> rescued;
    phmm_batch view{};
};

// what upload_prepare (host) leaves for upload_enqueue (HIP): the layout of the resident's staging buffer
struct UploadPlan
{
    size_t total = 0;
    uint64_t read_bytes = 0, hap_bytes = 0, slots = 0;
    size_t o_bases = 0, o_qual = 0, o_mapq = 0, o_ins = 0, o_del = 0, o_gcp = 0, o_roff = 0, o_rlen = 0, o_hap = 0, o_hoff = 0, o_hlen = 0,
           o_pr = 0, o_ph = 0, o_order = 0, o_poff = 0, o_owner = 0;
    bool has_prep = false, slim = false;
    uint8_t gcp0 = 0;
    phmm_read_prep prep{};
};

struct phmm_resident
{
    UploadPlan plan;
    CpuBatch *cpu = nullptr;
    ~phmm_resident()
    {
        delete cpu;
        if (hist_ready) (void)hipEventDestroy(hist_ready);
    }
    uint64_t n_pairs = 0, cells = 0;
    uint32_t n_reads = 0, n_haps = 0, max_hap_len = 0, max_read_len = 0;
    bool uniform_gcp = false, uniform_gop = false;
    bool ragged = false;             // single-pass classes share launches (pairhmm_kernels_ragged.hip); decided at upload
    uint32_t last_fp32_launches = 0; // launches of the last fp32 pass
    // inputs travel as ONE pinned staging buffer and ONE host-to-device copy per upload (a region-sized submission is
    // latency-bound: fifteen small copies cost more than the sweep); the views below point into `blob`
    HostBuf stage, out_stage;
    DevBuf<uint8_t> blob;
    DevView<uint8_t> read_bases, read_qual, hap_bases, read_mapq;
    DevView<uint32_t> read_off, read_len, hap_off, hap_len, pair_read, pair_hap, order, plane_off, word_owner;
    DevView<uint8_t> read_ins, read_del, read_gcp;  // prepared batches: inside `blob`; otherwise the two buffers below
    DevBuf<uint8_t> prep_gop, prep_gcp;             // written by the set-up launch (device-side read preparation)
    DevBuf<uint32_t> planes;
    DevBuf<float> raw32;
    DevBuf<double> raw64;
    DevBuf<uint32_t> rescue_list;
    DevView<uint32_t> rescue_count;  // inside raw32, behind the sums
    // a batch that lives in a phmm_stage: the kernels write their sums into the stage's pinned host arrays instead of raw32 / raw64
    float *out32_host = nullptr;
    double *out64_host = nullptr;
    bool queue_is_empty = false;     // the set-up launch has just zeroed rescue_count
    bool rescue_deferred = false;    // phmm_compute: the fp64 re-run of this pass is phmm_download's to launch, if any pair underflowed
    // fp64 first (pairhmm_sweep.cpp): striped ranges whose fp32 pass is skipped this pass; what the last finished pass observed per
    // striped queue (pairs, of which fp64 decided), read back asynchronously into `hist_host` behind `hist_ready`
    bool queue_f64_first[phmm::kRescueQueues] = {};
    bool fp32_first_only = false;  // a chunk slot of the one-shot pipeline: its two passes are launched by different threads, fp32 first always
    struct StripedLaunchPlan
    {
        TeamShape sh{16, 1};
        uint32_t teams = 0;
        size_t carry_at = 0;
    };
    std::vector<StripedLaunchPlan> fp32_plans;  // per bin (striped bins only), made by the fp32 pass
    HostBuf hist_host;
    hipEvent_t hist_ready = nullptr;
    bool hist_pending = false, hist_f64_first[phmm::kRescueQueues] = {};
    uint32_t hist_cap[phmm::kRescueQueues] = {};
    float f64_share[phmm::kRescueQueues] = {};  // share of a striped queue's pairs that fp64 decided, as last observed ...
    bool f64_share_known[phmm::kRescueQueues] = {};  // ... if at all
    bool queue_sparse[phmm::kRescueQueues] = {};  // set by the fp32 pass: the queue is a list in processing order with holes (KernelArgs::sparse_order)
    DevBuf<float> carry32;
    DevBuf<double> carry64;
    std::vector<Bin> bins;
    uint32_t last_rescued = 0;
    double last_kernel_ms = 0.0;
};


// PHMM_PHASE_TIMING=1: host phases of phmm_compute on stderr (tuning aid)
struct PhaseClock
{
    bool on = getenv("PHMM_PHASE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double us[8] = {0};
    void lap(int i)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        us[i] += std::chrono::duration<double, std::micro>(n - t).count();
        t = n;
    }
};
extern thread_local PhaseClock *g_phase;  // pairhmm_capi.cpp
#define PHASE(i) do { if (g_phase) g_phase->lap(i); } while (0)

// How a batch reaches the device.  The defaults are the plain upload; the chunked one-shot call (phmm_compute on a big batch)
// uploads on a side stream without waiting for the context's stream, and hands in windows of the tables with the pair
// indices still counted from the start of the caller's tables.
struct UploadOpts
{
    hipStream_t stream = nullptr;  // the copy; null: the context's stream
    hipStream_t setup_stream = nullptr;  // the fill and the set-up launch; null: the stream of the copy
    bool caller_synchronised = false;  // the caller knows that nothing on the device still reads this resident's buffers
    uint32_t read_bias = 0, hap_bias = 0;  // subtracted from pair_read / pair_hap
};

// A window of the caller's tables that holds everything a contiguous range of the pair list refers to (pairhmm_chunked.cpp).
struct ChunkWindow
{
    uint64_t p0 = 0, p1 = 0;               // pairs
    uint32_t r0 = 0, r1 = 0, h0 = 0, h1 = 0;  // reads [r0, r1), haplotypes [h0, h1)
    uint64_t rb0 = 0, rb1 = 0, hb0 = 0, hb1 = 0;  // byte ranges of the two tables
};
constexpr uint64_t kChunkedMinPairs = 40000;  // below this one upload + one sweep is as fast (2^19 until the pipeline had two sweep lanes)
constexpr int kChunkedUnavailable = -1000;  // compute_chunked could not start (no thread to be had): not an error, take the plain path
static inline uint64_t chunked_min_pairs()
{
    if (const char *e = phmm_hook("PHMM_CHUNKED_MIN_PAIRS")) return (uint64_t)std::max(1, atoi(e));  // tuning
    return kChunkedMinPairs;
}

namespace phmm_detail
{
// the C boundary: nothing is thrown across it (every extern "C" entry point runs its body through this)
template <typename F>
int guarded(const char *what, F &&fn)
{
    try {
        return fn();
    }
    catch (const std::bad_alloc &) {
        return fail(PHMM_E_NOMEM, "%s: out of host memory", what);
    }
    catch (const std::exception &e) {
        return fail(PHMM_E_INVALID, "%s: %s", what, e.what());
    }
    catch (...) {
        return fail(PHMM_E_INVALID, "%s: unexpected exception", what);
    }
}
}  // namespace phmm_detail
using phmm_detail::guarded;

// The entry points are implemented in this namespace with C++ linkage; the extern "C" wrappers at the end of pairhmm_capi.cpp
// catch whatever a body may throw, so that nothing crosses the C boundary.
namespace impl
{
// pairhmm_sweep.cpp: one pass over a resident batch, enqueued on the context's stream set; pairhmm_capi.cpp: the CPU backend's two steps
int ensure_side_streams(phmm_context *c);  // the context's side streams and join events, made on first use
int sweep_fp32(phmm_context *c, phmm_resident *rb);
int sweep_rescue(phmm_context *c, phmm_resident *rb, const phmm_context::SweepLane *on = nullptr);
void harvest_history(phmm_context *c, phmm_resident *rb);  // fp64 first: fold what the batch's last finished pass observed into the history
int cpu_upload(phmm_context *c, const phmm_batch *b, phmm_resident *rb);
int cpu_sweep(phmm_context *c, phmm_resident *rb);
int phmm_compute(phmm_context *c, const phmm_batch *b, float *out_raw_f32, double *out_log10, phmm_stats *stats);
int upload_into(phmm_context *c, const phmm_batch *b, phmm_resident *rb, const UploadOpts &opt);  // bin, pack, copy, set-up launch: the two below
int upload_prepare(phmm_context *c, const phmm_batch *b, phmm_resident *rb, const UploadOpts &opt, phmm_detail::BinScratch &bin_scratch);  // host only
int upload_enqueue(phmm_context *c, phmm_resident *rb, const UploadOpts &opt);  // HIP calls only
int phmm_create(int device_id, void *stream, phmm_context **out);
int phmm_create_cpu(int n_threads, int allow_simd, phmm_context **out);
void phmm_destroy(phmm_context *c);
int phmm_synchronize(phmm_context *c);
int phmm_upload(phmm_context *c, const phmm_batch *b, phmm_resident **out);
int phmm_sweep(phmm_context *c, phmm_resident *rb);
int phmm_sweep_timed(phmm_context *c, phmm_resident *rb, int iters, double *ms_per_pass, double *fp32_ms_per_pass);
int phmm_download(phmm_context *c, phmm_resident *rb, float *out_raw_f32, double *out_log10, phmm_stats *stats);
void phmm_release(phmm_context *c, phmm_resident *rb);
// pairhmm_chunked.cpp: the one-shot call on a big batch as a pipeline of chunks
bool plan_chunks(const phmm_batch *b, std::vector<ChunkWindow> &plan);
int compute_chunked(phmm_context *c, const phmm_batch *b, const std::vector<ChunkWindow> &plan, float *out_raw_f32, double *out_log10,
                    phmm_stats *stats);
// pairhmm_stage.cpp
int phmm_stage_create(phmm_context *c, const phmm_stage_caps *caps, phmm_stage **out);
void phmm_stage_destroy(phmm_context *c, phmm_stage *st);
int phmm_stage_get_view(const phmm_stage *st, phmm_stage_view *view, phmm_stage_caps *caps);
int phmm_stage_submit(phmm_context *c, phmm_stage *st, uint32_t n_regions, uint32_t n_reads, uint32_t n_haps, const phmm_stage_params *prm);
int phmm_stage_wait(phmm_context *c, phmm_stage *st, phmm_stats *stats);
uint64_t phmm_stage_pair_first(const phmm_stage *st, uint32_t region);
int phmm_stage_log10(const phmm_stage *st, uint64_t first_pair, uint64_t n, double *out, float *out_raw_f32, uint64_t *n_rescued);
int phmm_compute_regions(phmm_context *c, const phmm_batch *t, const phmm_region *regions, uint32_t n_regions, float *out_raw_f32,
                         double *out_log10, phmm_stats *stats);
}  // namespace impl
#endif
