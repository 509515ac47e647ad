// pairhmm_binning.h -- the length binning of a batch: which kernel shape every pair takes and in which order the pairs are
// processed.  Host only (no HIP call), shared by the pair-list upload (pairhmm_capi.cpp) and the region-descriptor path
// (pairhmm_stage.cpp); phmm_plan_bins / phmm_plan_region_bins in the C ABI expose it to the CPU-side tests.
#ifndef ROVACA_AMD_PAIRHMM_BINNING_H_
#define ROVACA_AMD_PAIRHMM_BINNING_H_
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/pairhmm_hip.h"
#include "pairhmm_device.h"

struct Bin
{
    int rows_per_lane;      // 1..16: single-pass kernel <K>; 0: striped kernel (long read or long haplotype)
    uint32_t first, count;  // range of `order`
    uint32_t lds_words_per_group;
    uint32_t max_hap_len, max_read_len;
    int queue;              // rescue queue the launch feeds (phmm::kRescueQueues)
};

// Striped pairs are launched in up to four size ranges (48+ stripes of 128 rows, 24..47, 8..23, 1..7), each with the team
// shape that fits it: a batch of a thousand long reads of very different sizes is otherwise bound by its longest pair
// sweeping alone in a team sized for the average one.  Length classes: 1..16 single-pass rows per lane, then
// 17 + 63 - min(stripes, 63).
constexpr int kQueues = phmm::kRescueQueues;
static inline int bin_id_of_class(uint32_t k) { return k < 17 ? (int)k : k < 33 ? 17 : k < 57 ? 18 : k < 73 ? 19 : 20; }

// single-pass kernels keep 16 pairs' haplotype planes in LDS per workgroup; longer haplotypes take the striped kernel
constexpr uint32_t kSinglePassMaxHap = 4096;
constexpr uint32_t kMaxLen = 1u << 20;
constexpr uint64_t kQueuePerClassMinPairs = 50000;  // batches above this re-run each single-pass class from its own queue
constexpr uint32_t kXcdLocalMinPairs = 1u << 18;  // single-pass launches from this many pairs on order their workgroups XCD-locally
constexpr uint32_t kWideMaxPairs = 1536;       // single-pass launches up to this many pairs run one pair per wave
constexpr uint64_t kRaggedUniformBelowPairs = 1u << 18;  // ragged launches below this take the uniform-budget kernel (one launch for all classes)
constexpr uint32_t kRescueMaxTeams = 4096;    // workgroups of the fp64 re-run (they stride over the rescue queue)
constexpr uint32_t kStripedMaxTeams = 4096;   // workgroups of the striped kernel per launch, divided by waves per team


// Test and tuning hooks are environment variables named PHMM_*, looked up where they apply so that a test can flip them between
// calls -- and ALL of them are dead unless PHMM_ENABLE_TEST_HOOKS=1 is set as well: they change launch geometry, force slower kernel
// shapes or lift safety checks (PHMM_MULTI_ALLOW_SHARED_DEVICE), and a production process that happens to inherit one must not run a
// shape three times slower for it.  An entry point on the latency-critical path (phmm_stage_submit, phmm_compute) opens a HookScope,
// which reads the master switch ONCE, and every phmm_hook() inside it is then a thread-local flag test instead of a getenv (a dozen
// of them cost a region-sized submission 3-5 us).  Outside a scope phmm_hook looks the switch up itself.
// (Not hooks, read directly: PHMM_PHASE_TIMING -- diagnostics on stderr -- and PHMM_STAGE_BLOCKING_SYNC -- an operating choice.)
namespace phmm_detail
{
extern thread_local int t_hooks;  // -1: no scope open (look up); 0: hooks are off; 1: PHMM_ENABLE_TEST_HOOKS is set
bool test_hooks_enabled();
struct HookScope
{
    int saved;
    HookScope() : saved(t_hooks) { if (t_hooks < 0) t_hooks = test_hooks_enabled() ? 1 : 0; }
    ~HookScope() { t_hooks = saved; }
    HookScope(const HookScope &) = delete;
    HookScope &operator=(const HookScope &) = delete;
};
}  // namespace phmm_detail
const char *phmm_hook(const char *name);

// Length class of a pair: rows per lane 1..16 for the single-pass kernels (one launch each); pairs with a read above 256 bases
// or a haplotype above the single-pass LDS budget go to the striped kernel, most stripes first
constexpr uint32_t kClasses = 17 + 64;  // 1..16, then 17 + 63 - min(stripes, 63)
static inline uint32_t class_of(uint32_t R, uint32_t H)
{
    const uint32_t K = (R + phmm::kLanesPerPair - 1) / phmm::kLanesPerPair;
    if (K > (uint32_t)phmm::kMaxRowsPerLane || H > kSinglePassMaxHap) {
        const uint32_t S = phmm::kLanesPerPair * phmm::kStripedRowsF32;
        return 17u + (63u - std::min<uint32_t>((R + S - 1) / S, 63u));  // most stripes first: the longest pairs start first
    }
    return K;
}
// Rescue queue of a bin.  A large batch gives every single-pass class its own queue: the fp64 re-run of a queue is one launch
// whose waves hold four pairs each, and pairs of one class (read lengths within 16, haplotypes arriving longest first) keep
// those four in step -- mixed lengths in one queue cost the re-run a quarter of its rate.  A small batch keeps one queue: a
// launch per class would add more latency than the divergence costs.
static inline int queue_of_bin(int bin_id, bool queue_per_class) { return bin_id >= 17 ? bin_id - 16 : queue_per_class ? 4 + bin_id : 0; }
bool queue_per_class_for(uint64_t n_pairs);  // n_pairs > kQueuePerClassMinPairs, or PHMM_FORCE_QUEUE_PER_CLASS (tests)
// Ragged launches (pairhmm_kernels_ragged.hip): the single-pass classes of a batch in the Rovaca case (one gap-continuation quality,
// deletion == insertion gap-opens: every batch the adapter makes) share launches, so its bins are left one per class -- no ride-along,
// no padding rows.  PHMM_NO_RAGGED: the A/B hook (one launch per bin, small bins ride along, as until round 3).
bool ragged_launches_for(bool uniform_gcp);

// Host loops over the pairs of a BIG batch (binning, packing, the final log10) are split over a few threads; a region-sized call
// never gets here.  Two ways to get the threads: a HostPool the calling thread has made current (the chunked one-shot call keeps
// two in its context: spawning threads for every loop of every chunk cost it ~1 ms per call), or plain std::threads otherwise.
constexpr uint64_t kParallelMinPairs = 1u << 17;

namespace phmm_detail
{
// A few helper threads that take numbered parts of one job at a time.  run() hands out parts [0, parts) to the helpers and to
// the caller and returns when all are done.  Helpers poll for ~100 us after a job before they sleep: the loops of one call come
// back to back.
class HostPool
{
public:
    explicit HostPool(unsigned helpers)
    {
        try {
            for (unsigned i = 0; i < helpers; ++i) threads_.emplace_back([this] { loop(); });
        }
        catch (...) {  // (a process out of threads: the caller does the parts itself)
        }
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            stop_atomic_.store(true, std::memory_order_relaxed);
            ++generation_;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    HostPool(const HostPool &) = delete;
    HostPool &operator=(const HostPool &) = delete;
    unsigned helpers() const { return (unsigned)threads_.size(); }
    // While hot the helpers never go to sleep between jobs: the chunk pipeline of a one-shot call hands them a dozen short jobs per
    // chunk with serial stretches in between, and a helper that has to be woken through the condition variable (tens of microseconds on
    // a busy box) arrives when the job is half done.  The owner switches it on for the duration of a call (8-10 ms), never for longer.
    void set_hot(bool on) { hot_.store(on, std::memory_order_release); }
    template <typename F>
    void run(unsigned parts, F &&fn)  // fn(part index)
    {
        if (parts == 0) return;
        if (parts == 1 || threads_.empty()) {
            for (unsigned p = 0; p < parts; ++p) fn(p);
            return;
        }
        std::lock_guard<std::mutex> one_job(run_mu_);  // one job at a time (callers that overlap use pools of their own)
        std::function<void(unsigned)> f = std::ref(fn);
        {
            std::unique_lock<std::mutex> lk(mu_);
            while (in_job_ != 0) {  // a helper that woke up late for the previous job is still looking at its counters
                lk.unlock();
                std::this_thread::yield();
                lk.lock();
            }
            fn_ = &f;
            parts_ = parts;
            next_.store(0, std::memory_order_relaxed);
            done_.store(0, std::memory_order_relaxed);
            ++generation_;
            gen_atomic_.store(generation_, std::memory_order_release);
        }
        cv_.notify_all();
        work(f, parts);
        // the last part may still be running on a helper
        for (unsigned spins = 0; done_.load(std::memory_order_acquire) != parts; ++spins)
            if (spins > 2000) std::this_thread::yield();
        std::lock_guard<std::mutex> lk(mu_);  // no helper is inside this job any more: f may go
        fn_ = nullptr;
    }

    // The same job handed to the helpers ALONE, without the caller taking part or waiting: begin() returns at once, the helpers take
    // the parts in ascending order, end() waits for the last one.  (The chunk pipeline's prepare workers: the caller enqueues chunk k on
    // the device while the helpers pack and bin chunks k + 1, k + 2, ...)  Needs at least one helper.
    void begin(unsigned parts, std::function<void(unsigned)> fn)
    {
        run_mu_.lock();  // one job at a time; released by end()
        async_fn_ = std::move(fn);
        {
            std::unique_lock<std::mutex> lk(mu_);
            while (in_job_ != 0) {
                lk.unlock();
                std::this_thread::yield();
                lk.lock();
            }
            fn_ = &async_fn_;
            parts_ = parts;
            next_.store(0, std::memory_order_relaxed);
            done_.store(0, std::memory_order_relaxed);
            ++generation_;
            gen_atomic_.store(generation_, std::memory_order_release);
        }
        cv_.notify_all();
    }
    void end()
    {
        for (unsigned spins = 0; done_.load(std::memory_order_acquire) != parts_; ++spins)
            if (spins > 2000) std::this_thread::yield();
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = nullptr;
        }
        run_mu_.unlock();
    }

private:
    void work(std::function<void(unsigned)> &f, unsigned parts)
    {
        for (;;) {
            const unsigned p = next_.fetch_add(1, std::memory_order_acq_rel);
            if (p >= parts) return;
            f(p);
            done_.fetch_add(1, std::memory_order_acq_rel);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            // poll briefly for the next job, then sleep
            const auto t0 = std::chrono::steady_clock::now();
            bool fresh = false;
            for (unsigned spins = 0;; ++spins) {
                if (gen_atomic_.load(std::memory_order_acquire) != seen) {
                    fresh = true;
                    break;
                }
                if ((spins & 63u) == 63u && !hot_.load(std::memory_order_acquire) &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100))
                    break;
                if (stop_atomic_.load(std::memory_order_relaxed)) break;
                __builtin_ia32_pause();
            }
            std::function<void(unsigned)> *f = nullptr;
            unsigned parts = 0;
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (!fresh) cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                f = fn_;
                parts = parts_;
                if (f) in_job_ += 1;
            }
            if (f) {
                work(*f, parts);
                std::lock_guard<std::mutex> lk(mu_);
                in_job_ -= 1;
            }
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_;
    std::vector<std::thread> threads_;
    std::function<void(unsigned)> *fn_ = nullptr;
    std::function<void(unsigned)> async_fn_;
    unsigned parts_ = 0, in_job_ = 0;
    std::atomic<unsigned> next_{0}, done_{0};
    uint64_t generation_ = 0;
    std::atomic<uint64_t> gen_atomic_{0};
    std::atomic<bool> hot_{false}, stop_atomic_{false};
    bool stop_ = false;
};
extern thread_local HostPool *t_pool;  // the pool parallel_ranges uses on this thread, or null: spawn
extern thread_local bool t_serial;     // this thread is itself one of several workers: its loops stay on it (no helpers, no spawning)
struct PoolScope
{
    HostPool *saved;
    explicit PoolScope(HostPool *p) : saved(t_pool) { t_pool = p; }
    ~PoolScope() { t_pool = saved; }
};
}  // namespace phmm_detail

template <typename F>
static void parallel_ranges(uint64_t n, unsigned n_threads, F &&fn)  // fn(lo, hi, thread index)
{
    if (phmm_detail::t_serial) {
        for (unsigned t = 0; t < n_threads; ++t) fn(n * t / n_threads, n * (t + 1) / n_threads, t);
        return;
    }
    if (phmm_detail::HostPool *pool = phmm_detail::t_pool) {
        pool->run(n_threads, [&](unsigned t) { fn(n * t / n_threads, n * (t + 1) / n_threads, t); });
        return;
    }
    std::vector<std::thread> th;
    th.reserve(n_threads);
    unsigned started = 1;  // ranges [1, started) run on helper threads; the rest on the caller (a process with 128 workers may
                           // be out of threads: std::thread then throws, and the work is done here instead)
    try {
        for (; started < n_threads; ++started) {
            const unsigned t = started;
            th.emplace_back([&fn, n, n_threads, t] { fn(n * t / n_threads, n * (t + 1) / n_threads, t); });
        }
    }
    catch (...) {
    }
    fn(0, n / n_threads, 0u);
    for (unsigned t = started; t < n_threads; ++t) fn(n * t / n_threads, n * (t + 1) / n_threads, t);
    for (auto &x : th) x.join();
}
static unsigned host_threads()
{
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(8u, hw ? hw : 1u));
}


namespace phmm_detail
{
struct BinScratch  // host scratch of the pair binning, kept between calls by its owner
{
    std::vector<uint32_t> hist, slot, read_hmax;
};
// Pair-list form.  Order = (class, longest haplotype OF THE PAIR'S READ descending, pair index), as a counting sort over (class,
// length) -- a comparison sort was the largest host cost of a region-sized call -- or a comparison sort when the lengths are sparse.
// The length key is the read's, not the pair's (round 5): the pairs of one read stay next to each other whatever the lengths of its
// haplotypes, because the table kernels share a read's prior table among the pairs of a workgroup (pairhmm_systolic_tab.h); the
// haplotypes of one read are those of one region, of nearly one length, so the four pairs of a wave still finish together.
// cross: the pair list is the full cross product in (read, hap) order and pair_read / pair_hap are not read.  read_bias /
// hap_bias are subtracted from the indices of a pair list that counts from the start of the caller's tables while b's tables
// are windows of them.  Writes pr[p], ph[p] (the pair list as the kernels index it) and order[0 .. n_pairs), appends the
// launches to `bins`.  PHMM_E_INVALID (with the message set) when a pair indexes outside the tables.
// ride_along: small single-pass bins are handed to the next bin up (merge_small_bins, pairhmm_binning.cpp) -- the way a batch of mixed
// lengths keeps its launch count down when its classes cannot share a ragged launch (ragged_launches_for below).
int bin_pairs(const phmm_batch *b, bool cross, uint64_t n_pairs, uint32_t read_bias, uint32_t hap_bias, uint32_t *pr, uint32_t *ph,
              uint32_t *order, BinScratch &scratch, std::vector<Bin> &bins, uint64_t *cells, bool ride_along = true);

// Region form: every region is the cross product of a run of reads with a run of haplotypes, so the placement is decided per
// READ (O(reads) host work; the pair list itself is spelled out on the device): order = (class, region, read, hap), and
// slot_first[r] is the position in `order` of read r's first pair, pair_first[g] the id of region g's first pair (pair ids
// count region by region, [read][hap] inside one).  Regions must list their reads in ascending, non-overlapping runs;
// haplotypes may be shared.  A region with a haplotype above the single-pass budget sends all its pairs to the striped kernel.
int bin_regions(const phmm_region *regions, uint32_t n_regions, const uint32_t *read_len, uint32_t n_reads, const uint32_t *hap_len,
                uint32_t n_haps, uint32_t *slot_first, uint32_t *pair_first, std::vector<Bin> &bins, uint64_t *cells, uint64_t *n_pairs,
                bool ride_along = true);
}  // namespace phmm_detail
#endif
