// pairhmm_gpu_engine.cpp -- rovaca::call_gpu_pairhmm: flattens one region into a POD batch, runs it through the C ABI of
// include/pairhmm_hip.h, and applies the reference's post-processing.  The only translation unit that knows Rovaca types.
#include "pairhmm_gpu_engine.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>

#include <climits>
#include <cpuid.h>
#include <linux/futex.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <vector>

#include "../../include/pairhmm_hip.h"
#include "pairhmm_host_logic.h"

namespace rovaca
{

#ifndef ROVACA_AMD_IN_ROVACA_TREE
// stand-alone build: this library owns the seam's global.  Inside Rovaca it is defined at pairhmm_engine.cpp:16-17.
DoubleVector2D (*call_pairhmm)(const HaplotypeVector &hs, ReadVector &rs, int32_t min_quality_threshold,
                               PcrIndelModel pcr_option, pMemoryPool pool) = nullptr;
#endif

namespace
{

std::atomic<bool> g_active{false};
std::atomic<int> g_device{-1};
std::atomic<int> g_next_device{0};

// one device context per calling thread, created on first use and destroyed with the thread
struct ThreadContext
{
    phmm_context *ctx = nullptr;
    ~ThreadContext()
    {
        if (ctx) phmm_destroy(ctx);
    }
    phmm_context *get()
    {
        if (ctx) return ctx;
        int dev = g_device.load();
        if (dev < 0) {
            const int n = phmm_device_count();
            dev = n > 0 ? g_next_device.fetch_add(1) % n : 0;
        }
        if (phmm_create(dev, nullptr, &ctx) != PHMM_OK) {
            // the reference's convention for an unrecoverable engine failure: log, exit (pairhmm_engine.cpp:32-35)
            fprintf(stderr, "[rovaca_amd] pairhmm: cannot create a device context: %s\n", phmm_last_error());
            exit(EXIT_FAILURE);
        }
        return ctx;
    }
};
thread_local ThreadContext t_ctx;

// ---------------------------------------------------------------------------------------------------------------------
// Cross-region batching.  The caller evaluates one region per worker thread, synchronously
// (haplotypecaller_engine.cpp:119-285); a region is 1e7-1e8 cells, i.e. a few microseconds of MI355X time against a fixed
// per-submission cost of tens.  Workers therefore share STAGES: blocks of pinned host memory owned by the engine and laid out
// as the device reads them (include/pairhmm_hip.h, phmm_stage).  A worker reserves ranges of the stage that is currently
// filling (a mutex-guarded bump of six cursors), decodes its reads and copies its haplotypes STRAIGHT into the pinned tables
// -- the only copy the bytes ever get on the host -- and commits.  Dispatcher threads (a few per device) take whatever has
// gathered in the open stage as soon as they are free, make ONE asynchronous submission of it (region descriptors: the device
// spells out the cross products, prepares the reads and builds the haplotype planes in one set-up launch; the kernels write the
// sums back into the stage), wait for it and wake the stage's workers, each of which converts and post-processes its own
// region from the pinned results.  The dispatcher touches O(reads + haplotypes) descriptors, never the bytes and never the
// pairs.  No timer: a batch is whatever arrived while the dispatchers were busy, so a single caller pays one hand-off and
// 30-128 callers are coalesced automatically.
// ---------------------------------------------------------------------------------------------------------------------
// Where the adapter's host time goes (always on: a handful of clock reads per region and per submission).  Nanoseconds, summed over
// threads; read with gpu_pairhmm_profile (tools/adapter_bench.cpp prints them, profiles/ keeps a run).
enum ProfileSlot {
    kProfReserve, kProfFill, kProfWaitResults, kProfConvert, kProfPostprocess,              // workers
    kProfDispIdle, kProfDispCommitSpin, kProfDispSubmit, kProfDispDeviceWait, kProfDispWake,  // dispatchers
    kProfSlots
};
std::atomic<uint64_t> g_prof[kProfSlots];
struct ProfClock
{
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(ProfileSlot slot)
    {
        const auto n = std::chrono::steady_clock::now();
        g_prof[slot].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count(), std::memory_order_relaxed);
        t = n;
    }
};

static long futex_call(std::atomic<uint32_t> *addr, int op, uint32_t val)
{
    return syscall(SYS_futex, reinterpret_cast<uint32_t *>(addr), op, val, nullptr, nullptr, 0);
}

// capacity of one stage: sixty regions of the usual size (48 reads x 9 haplotypes of 150 / 300 bases are 17 KB and 432 pairs), or one
// very deep one; what does not fit an EMPTY stage takes the direct path (the worker's own context).  2.3 MB of pinned memory and
// 3.5 MB of HBM each.  All stages are made at start-up (ROVACA_AMD_PAIRHMM_STAGES, default 32 per device: workers that keep several
// regions in flight hold finished stages until they collect them); the pool grows up to kStagesMax when they run out, which costs
// the worker that hits it a few milliseconds of allocation.
constexpr phmm_stage_caps kStageCaps{64, 4096, 1024, 1u << 20, 64u * 1024u};
constexpr int kStagesMax = 96;
static int stages_at_start()
{
    const char *e = getenv("ROVACA_AMD_PAIRHMM_STAGES");
    return std::max(2, std::min(kStagesMax, e ? atoi(e) : 32));
}

// The fill cursors of a stage as ONE 128-bit word, reserved with a compare-and-swap: regions | reads | haplotypes | generation |
// closed in the low half, bytes | pairs in the high half.  Until round 3 they were five integers behind the batcher's mutex, and
// with 64-128 workers on a 16-core share a worker preempted inside that critical section held everybody up: `reserve` cost 11.7 /
// 33 us per region, more than filling the stage, the log10 and H8 together (profiles/r3_adapter_host_profile.txt).  Now a
// reservation is lock-free; the mutex is taken to roll a full stage over, by a dispatcher, and by the last reader that recycles one.
struct alignas(16) Cursor
{
    uint64_t lo = kClosed, hi = 0;  // (a stage that is not the open one is closed: a stale reservation cannot land in it)
    static constexpr uint64_t kClosed = 1;
    bool closed() const { return (lo & kClosed) != 0; }
    uint32_t generation() const { return (uint32_t)(lo >> 1) & 0x7FFFu; }
    uint32_t regions() const { return (uint32_t)(lo >> 48); }
    uint32_t reads() const { return (uint32_t)(lo >> 32) & 0xFFFFu; }
    uint32_t haps() const { return (uint32_t)(lo >> 16) & 0xFFFFu; }
    uint32_t bytes() const { return (uint32_t)(hi >> 32); }
    uint32_t pairs() const { return (uint32_t)hi; }
    static Cursor make(uint32_t regions, uint32_t reads, uint32_t haps, uint32_t bytes, uint32_t pairs, uint32_t gen, bool closed)
    {
        Cursor c;
        c.lo = ((uint64_t)regions << 48) | ((uint64_t)reads << 32) | ((uint64_t)haps << 16) | ((uint64_t)(gen & 0x7FFFu) << 1) | (closed ? kClosed : 0);
        c.hi = ((uint64_t)bytes << 32) | pairs;
        return c;
    }
};
// lock cmpxchg16b (every x86-64 this code can meet has it; the attribute spares the including build a -mcx16)
__attribute__((target("cx16"))) static inline bool cursor_cas(Cursor *p, Cursor &expected, const Cursor &desired)
{
    const unsigned __int128 e = ((unsigned __int128)expected.hi << 64) | expected.lo, d = ((unsigned __int128)desired.hi << 64) | desired.lo;
    const unsigned __int128 seen = __sync_val_compare_and_swap(reinterpret_cast<unsigned __int128 *>(p), e, d);
    if (seen == e) return true;
    expected.lo = (uint64_t)seen;
    expected.hi = (uint64_t)(seen >> 64);
    return false;
}
static inline Cursor cursor_peek(const Cursor *p)  // two relaxed loads: may be torn, which the compare-and-swap that follows catches
{
    Cursor c;
    c.lo = __atomic_load_n(&p->lo, __ATOMIC_ACQUIRE);
    c.hi = __atomic_load_n(&p->hi, __ATOMIC_ACQUIRE);
    return c;
}

struct StageSlot
{
    phmm_stage *st = nullptr;
    phmm_stage_view v{};
    Cursor cur;  // the fill cursors while this is the open stage
    // ... and as frozen by whoever closed it (a dispatcher, or the worker that found it full)
    uint32_t n_regions = 0, n_reads = 0, n_haps = 0;
    // written under the batcher's mutex before the stage is published as the open one (the release store of its cursor word); a worker
    // reads them after it has seen that word open -- possibly through a stale pointer to a stage that has been recycled and opened
    // again meanwhile, which is then simply the open stage
    std::atomic<int32_t> min_q{0}, pcr_model{0};
    std::atomic<uint32_t> committed{0};  // regions whose bytes are in place
    std::atomic<uint32_t> state{0};      // futex word: 0 filling / on the device, 1 results in the stage, 2 failed
    std::atomic<uint32_t> waiters{0};    // workers asleep on `state`
    std::atomic<uint32_t> readers{0};    // regions not yet collected; the last one recycles the stage
    int status = PHMM_OK;
    char error[256] = "";  // phmm_last_error() is thread-local to the dispatcher: the text travels with the stage
};

struct Reservation
{
    StageSlot *slot = nullptr;
    uint32_t region = 0, read_first = 0, hap_first = 0, byte_first = 0;
};

// How long a thread polls before it sleeps (ROVACA_AMD_PAIRHMM_SPIN_US, default 0 = never).  Polling saves the futex wake and
// the core's idle-state exit on each hand-off, but measured on the GPU box (CPU time capped per job) 30 workers fell from
// 74 000 to 59 000 regions/s with a 300 us budget and 128 workers to a third: the pollers eat the cycles the dispatchers need.
// Worth trying only where worker threads have cores to themselves.
static int spin_budget_us()
{
    static const int us = [] {
        const char *e = getenv("ROVACA_AMD_PAIRHMM_SPIN_US");
        return e ? std::max(0, atoi(e)) : 0;
    }();
    return us;
}

template <typename Pred>
static bool spin_until(Pred ready)
{
    const int budget = spin_budget_us();
    if (budget <= 0) return ready();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < 64; ++i) {
            if (ready()) return true;
            __builtin_ia32_pause();
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(budget)) return ready();
    }
}

class RegionBatcher
{
public:
    RegionBatcher(int device, int n_dispatchers) : device_(device)
    {
        if (phmm_create(device_, nullptr, &owner_) != PHMM_OK) die("cannot create a device context");
        for (int i = 0, n = stages_at_start(); i < n; ++i) free_.push_back(new_slot());
        for (int i = 0; i < n_dispatchers; ++i) threads_.emplace_back([this] { run(); });
    }
    // (never destroyed: the dispatcher threads would otherwise call into HIP during static teardown)

    // A place for one region in the stage that is filling.  false: the region does not fit even an empty stage, or every stage
    // is taken and the pool is at its limit -- the caller evaluates it on its own context instead (never blocks on other workers:
    // a worker that keeps several regions in flight may hold the very stages it would be waiting for).
    bool reserve(uint32_t n_reads, uint32_t n_haps, uint32_t bytes, int32_t min_q, int32_t pcr_model, Reservation &rv)
    {
        const uint64_t n_pairs = (uint64_t)n_reads * n_haps;
        if (n_reads > kStageCaps.max_reads || n_haps > kStageCaps.max_haps || bytes > kStageCaps.max_bytes || n_pairs > kStageCaps.max_pairs)
            return false;
        for (;;) {
            StageSlot *s = open_.load(std::memory_order_acquire);
            if (s) {
                Cursor c = cursor_peek(&s->cur);
                // one device batch per distinct (threshold, PCR model); in practice they are process-wide constants
                while (!c.closed() && s->min_q.load(std::memory_order_relaxed) == min_q && s->pcr_model.load(std::memory_order_relaxed) == pcr_model && c.regions() < kStageCaps.max_regions &&
                       c.reads() + n_reads <= kStageCaps.max_reads && c.haps() + n_haps <= kStageCaps.max_haps &&
                       (uint64_t)c.bytes() + bytes <= kStageCaps.max_bytes && c.pairs() + n_pairs <= kStageCaps.max_pairs) {
                    const Cursor n = Cursor::make(c.regions() + 1, c.reads() + n_reads, c.haps() + n_haps, c.bytes() + bytes,
                                                  c.pairs() + (uint32_t)n_pairs, c.generation(), false);
                    if (!cursor_cas(&s->cur, c, n)) continue;  // somebody else moved the cursors (c has what is there now): look again
                    rv.slot = s;
                    rv.region = c.regions();
                    rv.read_first = c.reads();
                    rv.hap_first = c.haps();
                    rv.byte_first = c.bytes();
                    s->readers.fetch_add(1, std::memory_order_relaxed);  // (before this region's commit, which every reader's release follows)
                    s->v.regions[rv.region] = phmm_region{rv.read_first, n_reads, rv.hap_first, n_haps};
                    if (idle_.load(std::memory_order_acquire) != 0) {
                        // a dispatcher is asleep (or about to be): it wakes up while this worker is still writing its bytes.  Passing
                        // through the mutex orders this reservation before the dispatcher's predicate check or after its wait.
                        { std::lock_guard<std::mutex> lk(mu_); }
                        cv_.notify_one();
                    }
                    return true;
                }
            }
            // the open stage is full, closed, of another (threshold, model), or there is none: roll over, under the mutex
            bool wake = false;
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (open_.load(std::memory_order_relaxed) != s) continue;  // somebody rolled over meanwhile
                if (s) {
                    if (freeze(*s) != 0) {
                        closed_.push_back(s);  // full: a dispatcher takes it as it is
                        wake = true;
                    }
                    else {
                        free_.push_back(s);  // (nothing in it: a stage opened for another (threshold, model))
                    }
                    open_.store(nullptr, std::memory_order_relaxed);
                }
                if (free_.empty()) {
                    if (n_slots_ >= kStagesMax) {
                        lk.unlock();
                        if (wake) cv_.notify_one();
                        return false;
                    }
                    ++n_slots_;  // counted before the lock is dropped: the pool never overshoots
                    lk.unlock();
                    if (wake) cv_.notify_one();
                    StageSlot *fresh = new_slot(false);
                    lk.lock();
                    free_.push_back(fresh);
                    wake = false;
                    if (open_.load(std::memory_order_relaxed) != nullptr) continue;  // somebody opened one while this one was being made
                }
                StageSlot *n = free_.back();
                free_.pop_back();
                n->min_q.store(min_q, std::memory_order_relaxed);
                n->pcr_model.store(pcr_model, std::memory_order_relaxed);
                n->n_regions = n->n_reads = n->n_haps = 0;
                const Cursor fresh_cur = Cursor::make(0, 0, 0, 0, 0, cursor_peek(&n->cur).generation() + 1, false);
                __atomic_store_n(&n->cur.hi, fresh_cur.hi, __ATOMIC_RELAXED);
                __atomic_store_n(&n->cur.lo, fresh_cur.lo, __ATOMIC_RELEASE);  // (no reservation can be in flight on a closed stage)
                open_.store(n, std::memory_order_release);
            }
            if (wake) cv_.notify_one();
        }
    }
    static void commit(const Reservation &rv) { rv.slot->committed.fetch_add(1, std::memory_order_release); }
    // blocks until the stage's results are in; returns the submission's status
    static int wait(StageSlot &s)
    {
        if (!spin_until([&] { return s.state.load(std::memory_order_acquire) != 0; })) {
            s.waiters.fetch_add(1);
            while (s.state.load() == 0) futex_call(&s.state, FUTEX_WAIT_PRIVATE, 0);
            s.waiters.fetch_sub(1);
        }
        return s.status;
    }
    // a region's results have been read (or are not wanted): the last one hands the stage back
    void release(StageSlot &s)
    {
        if (s.readers.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
        s.n_regions = s.n_reads = s.n_haps = 0;  // (the cursor word stays closed until the stage is opened again)
        s.committed.store(0, std::memory_order_relaxed);
        s.state.store(0, std::memory_order_release);
        std::lock_guard<std::mutex> lk(mu_);
        free_.push_back(&s);
    }
    std::atomic<uint64_t> regions{0}, submissions{0};
    int device() const { return device_; }

private:
    [[noreturn]] static void die(const char *what)
    {
        // the reference's convention for an unrecoverable engine failure: log, exit (pairhmm_engine.cpp:32-35)
        fprintf(stderr, "[rovaca_amd] pairhmm: %s: %s\n", what, phmm_last_error());
        exit(EXIT_FAILURE);
    }
    // closes a stage for reservations and freezes its cursors into the plain fields; returns the regions in it
    static uint32_t freeze(StageSlot &s)
    {
        Cursor c = cursor_peek(&s.cur);
        for (;;) {
            if (c.closed()) break;
            Cursor n = c;
            n.lo |= Cursor::kClosed;
            if (cursor_cas(&s.cur, c, n)) break;
        }
        s.n_regions = c.regions();
        s.n_reads = c.reads();
        s.n_haps = c.haps();
        return s.n_regions;
    }
    bool open_has_regions() const
    {
        const StageSlot *o = open_.load(std::memory_order_acquire);
        return o && cursor_peek(&o->cur).regions() != 0;
    }
    StageSlot *new_slot(bool count = true)
    {
        StageSlot *s = new StageSlot();
        if (phmm_stage_create(owner_, &kStageCaps, &s->st) != PHMM_OK || phmm_stage_get_view(s->st, &s->v, nullptr) != PHMM_OK)
            die("cannot create a pinned stage");
        if (count) ++n_slots_;
        return s;
    }
    void run()
    {
        // named like the reference's own threads (haplotypecaller_engine.cpp:120) so that perf / top -H attribute the time
        pthread_setname_np(pthread_self(), "phmm-gpu-disp");
        phmm_context *ctx = nullptr;
        if (phmm_create(device_, nullptr, &ctx) != PHMM_OK) die("cannot create a device context");
        ProfClock clk;
        for (;;) {
            StageSlot *s = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                idle_.fetch_add(1, std::memory_order_acq_rel);
                cv_.wait(lk, [&] { return !closed_.empty() || open_has_regions(); });
                idle_.fetch_sub(1, std::memory_order_acq_rel);
                if (!closed_.empty()) {
                    s = closed_.front();
                    closed_.pop_front();
                }
                else {
                    s = open_.load(std::memory_order_relaxed);
                    freeze(*s);                                     // no reservation lands in it from here on
                    open_.store(nullptr, std::memory_order_release);  // the next worker opens a fresh stage
                }
                if (idle_.load(std::memory_order_relaxed) != 0 && (!closed_.empty() || open_has_regions())) cv_.notify_one();
            }
            clk.lap(kProfDispIdle);
            // the regions counted in are being written right now: a few microseconds each
            for (int spins = 0; s->committed.load(std::memory_order_acquire) != s->n_regions; ++spins) {
                if (spins < 256)
                    __builtin_ia32_pause();
                else
                    std::this_thread::yield();  // (a writer that lost its core: let it run)
            }
            clk.lap(kProfDispCommitSpin);
            const phmm_stage_params prm{s->min_q.load(std::memory_order_relaxed), s->pcr_model.load(std::memory_order_relaxed), rovaca_amd::kDefaultGapOpen,
                                        rovaca_amd::kDefaultGapCont};
            int rc = phmm_stage_submit(ctx, s->st, s->n_regions, s->n_reads, s->n_haps, &prm);
            clk.lap(kProfDispSubmit);
            if (rc == PHMM_OK) rc = phmm_stage_wait(ctx, s->st, nullptr);
            clk.lap(kProfDispDeviceWait);
            s->status = rc;
            if (rc != PHMM_OK) snprintf(s->error, sizeof(s->error), "%s", phmm_last_error());
            regions += s->n_regions;
            submissions += 1;
            s->state.store(rc == PHMM_OK ? 1u : 2u);  // the stage may be recycled by its last reader from here on
            if (s->waiters.load() != 0) futex_call(&s->state, FUTEX_WAKE_PRIVATE, INT32_MAX);
            clk.lap(kProfDispWake);
        }
    }

    int device_;
    phmm_context *owner_ = nullptr;  // the context the stages are created on (their device buffers belong to its device)
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<StageSlot *> open_{nullptr};  // the stage that is filling (its cursors: StageSlot::cur)
    std::deque<StageSlot *> closed_;          // full stages waiting for a dispatcher (guarded by mu_)
    std::vector<StageSlot *> free_;           // (guarded by mu_)
    int n_slots_ = 0;                         // (guarded by mu_)
    std::atomic<int> idle_{0};                // dispatchers asleep or about to be
    std::vector<std::thread> threads_;
};

// one per device in use; intentionally never destroyed
std::vector<RegionBatcher *> g_batchers;
std::atomic<bool> g_batching{false};
std::atomic<uint32_t> g_next_batcher{0};

}  // namespace

bool init_pairhmm_ptr_gpu(int device_id)
{
    // The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default, and kernels of streams that
    // share one run one after the other.  The two adapters keep five or more streams busy per device (three PairHMM dispatchers, two
    // Smith-Waterman ones), so a 140-us realignment kernel would hold up a PairHMM stage behind it: measured with both seams in the loop,
    // 30 workers 47 000 -> 54 000 regions/s with 16 queues (PairHMM alone 143 000 -> 151 000).  Only effective before the process's
    // first HIP call -- this function is the reference's init_pairhmm_ptr, which runs first thing on the main thread -- and never
    // overrides a value the user has set.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    {
        // the stage reservation is a cmpxchg16b (cursor_cas): CPUID.1:ECX bit 13; no second code path is built for a CPU without it
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (!__get_cpuid(1, &a, &b, &c, &d) || !(c & (1u << 13))) {
            fprintf(stderr, "rovaca_amd: this CPU has no cmpxchg16b; the GPU PairHMM adapter is not available\n");
            return false;
        }
    }
    const int n = phmm_device_count();
    if (n <= 0 || device_id >= n) return false;
    g_device.store(device_id);
    const char *env = getenv("ROVACA_AMD_PAIRHMM_BATCHING");
    const bool batching = !(env && env[0] == '0');
    if (batching && g_batchers.empty()) {
        // a few dispatchers per device: while one submission is on the device the next stage is filling and a third is being
        // converted by its workers
        const char *nd = getenv("ROVACA_AMD_PAIRHMM_DISPATCHERS");
        const int per_device = std::max(1, std::min(8, nd ? atoi(nd) : 3));
        if (device_id >= 0) {
            g_batchers.push_back(new RegionBatcher(device_id, per_device));
        }
        else {
            for (int d : rovaca_amd::batcher_devices("ROVACA_AMD_PAIRHMM_DEVICES", n)) g_batchers.push_back(new RegionBatcher(d, per_device));
        }
    }
    g_batching.store(batching);
    call_pairhmm = call_gpu_pairhmm;
    g_active.store(true);
    return true;
}

int gpu_pairhmm_batcher_stats(int batcher, int *device, uint64_t *regions, uint64_t *submissions)
{
    if (batcher >= 0 && (size_t)batcher < g_batchers.size()) {
        if (device) *device = g_batchers[batcher]->device();
        if (regions) *regions = g_batchers[batcher]->regions.load();
        if (submissions) *submissions = g_batchers[batcher]->submissions.load();
    }
    return (int)g_batchers.size();
}

void gpu_pairhmm_batching_stats(uint64_t *regions, uint64_t *submissions)
{
    uint64_t r = 0, s = 0;
    for (RegionBatcher *b : g_batchers) {
        r += b->regions.load();
        s += b->submissions.load();
    }
    if (regions) *regions = r;
    if (submissions) *submissions = s;
}

bool gpu_pairhmm_active() { return g_active.load(); }

int gpu_pairhmm_profile(double *seconds, const char **names, int cap)
{
    static const char *const kNames[kProfSlots] = {"worker: reserve a place in the open stage", "worker: decode reads + copy haplotypes into the stage",
                                                   "worker: blocked until the stage's results are in", "worker: log10 of the region's sums",
                                                   "worker: normalise, filter, transpose (H8)", "dispatcher: idle (no region waiting)",
                                                   "dispatcher: waiting for regions to be committed", "dispatcher: phmm_stage_submit (bin + 3 launches)",
                                                   "dispatcher: phmm_stage_wait (device)", "dispatcher: waking the stage's workers"};
    for (int i = 0; i < kProfSlots && i < cap; ++i) {
        if (seconds) seconds[i] = (double)g_prof[i].load() * 1e-9;
        if (names) names[i] = kNames[i];
    }
    return kProfSlots;
}

// A region between call_gpu_pairhmm_begin and call_gpu_pairhmm_finish.  Batched: its place in a stage (the bytes live in the
// stage's pinned tables, the results will appear there too).  Direct (batching off, or a region too large for a stage): evaluated
// inside begin on the calling thread's own context; the flattened tables and the [read][hap] result are kept here.  Nothing in it
// refers to the caller's vectors, so the caller may go on to its next region while this one is on the device.
struct PairhmmTicket::State
{
    size_t reads_num = 0, haplotypes_num = 0;
    RegionBatcher *batcher = nullptr;  // non-null while the region holds a place in a stage
    Reservation rv;
    // direct path only
    std::vector<int32_t> seq_len;
    std::vector<double> likelihood;  // [read][hap]
    int rc = PHMM_OK;
    char why[256] = "";
    void abandon()  // the region is not wanted any more: wait until the device is done with the stage, give the place back
    {
        if (!batcher) return;
        (void)RegionBatcher::wait(*rv.slot);
        batcher->release(*rv.slot);
        batcher = nullptr;
    }
};
PairhmmTicket::PairhmmTicket() = default;
PairhmmTicket::PairhmmTicket(PairhmmTicket &&) noexcept = default;
PairhmmTicket &PairhmmTicket::operator=(PairhmmTicket &&o) noexcept
{
    // assigning over a ticket that is still in flight (a look-ahead loop that skipped finish for the previous region): its stage
    // cannot be recycled under the device, so wait first -- as the destructor does
    if (this != &o) {
        if (st_) st_->abandon();
        st_ = std::move(o.st_);
    }
    return *this;
}
PairhmmTicket::~PairhmmTicket()
{
    if (st_) st_->abandon();  // a ticket dropped without finish
}

PairhmmTicket call_gpu_pairhmm_begin(const HaplotypeVector &hs, const ReadVector &rs, int32_t min_quality_threshold, PcrIndelModel pcr_option)
{
    PairhmmTicket t;
    if (hs.empty() || rs.empty()) return t;  // finish returns {} (pairhmm_internal.cpp:508-510)
    t.st_.reset(new PairhmmTicket::State());
    PairhmmTicket::State &st = *t.st_;
    const size_t reads_num = st.reads_num = rs.size();
    const size_t haplotypes_num = st.haplotypes_num = hs.size();
    const int model = static_cast<int>(pcr_option);

    // sizes.  Every read is decoded in place by ReadRecord::decode_base, which writes a Bases record -- a 4-byte count, then the
    // bases (genotype_struct.h:11-15) -- so each read gets a 4-byte-aligned slot: count, bases, raw qualities.
    size_t bytes = 4;
    int32_t max_len = 0;
    for (size_t i = 0; i < reads_num; ++i) {
        const int32_t n = rs[i]->seq_length();
        bytes += ((2 * (size_t)n + 3) & ~size_t(3)) + 4;
        max_len = std::max(max_len, n);
    }
    for (size_t j = 0; j < haplotypes_num; ++j) bytes += hs[j]->get_display_string()->num;

    Reservation rv;
    RegionBatcher *batcher = nullptr;
    ProfClock clk;
    if (g_batching.load() && !g_batchers.empty() && bytes < UINT32_MAX && reads_num < UINT32_MAX && haplotypes_num < UINT32_MAX) {
        batcher = g_batchers[g_next_batcher.fetch_add(1, std::memory_order_relaxed) % g_batchers.size()];
        if (!batcher->reserve((uint32_t)reads_num, (uint32_t)haplotypes_num, (uint32_t)bytes, min_quality_threshold, model, rv)) batcher = nullptr;
    }
    if (batcher) {
        clk.lap(kProfReserve);
        // ---- flatten straight into the stage's pinned arena: the one copy these bytes get on the host.  H3 (quality
        // modification) and H4 (PCR indel model) run on the device; the host only gathers bytes ----
        const phmm_stage_view &v = rv.slot->v;
        uint32_t at = (rv.byte_first + 3u) & ~3u;
        for (size_t i = 0; i < reads_num; ++i) {
            pReadRecord read = rs[i];
            const uint32_t n = (uint32_t)read->seq_length(), off = at + 4;
            read->decode_base(reinterpret_cast<pBases>(v.bytes + at));  // count at `at`, bases at `off`
            memcpy(v.bytes + off + n, read->qual(), n);
            const uint32_t r = rv.read_first + (uint32_t)i;
            v.read_off[r] = off;
            v.read_qual_off[r] = off + n;
            v.read_len[r] = n;
            v.read_mapq[r] = read->mapping_quality();
            at = (off + 2 * n + 3u) & ~3u;
        }
        for (size_t j = 0; j < haplotypes_num; ++j) {
            const pBases hb = hs[j]->get_display_string();
            memcpy(v.bytes + at, hb->data, hb->num);
            v.hap_off[rv.hap_first + j] = at;
            v.hap_len[rv.hap_first + j] = hb->num;
            at += hb->num;
        }
        RegionBatcher::commit(rv);
        clk.lap(kProfFill);
        st.batcher = batcher;
        st.rv = rv;
        return t;
    }

    // ---- direct path: flatten into the ticket, evaluate on this thread's own context ----
    std::vector<uint32_t> read_off(reads_num), read_len(reads_num), hap_off(haplotypes_num), hap_len(haplotypes_num);
    st.seq_len.resize(reads_num);
    size_t total = 0;
    for (size_t i = 0; i < reads_num; ++i) {
        st.seq_len[i] = rs[i]->seq_length();
        read_off[i] = (uint32_t)total;
        read_len[i] = (uint32_t)st.seq_len[i];
        total += (size_t)st.seq_len[i];
    }
    std::vector<uint8_t> bases(total), quals(total), mapq(reads_num);
    std::vector<uint8_t> tmp(sizeof(Bases) + (size_t)max_len + 1);
    pBases read_bases = reinterpret_cast<pBases>(tmp.data());
    for (size_t i = 0; i < reads_num; ++i) {
        pReadRecord read = rs[i];
        read->decode_base(read_bases);
        memcpy(bases.data() + read_off[i], read_bases->data, read_len[i]);
        memcpy(quals.data() + read_off[i], read->qual(), read_len[i]);
        mapq[i] = read->mapping_quality();
    }
    size_t hap_total = 0;
    for (size_t j = 0; j < haplotypes_num; ++j) {
        hap_off[j] = (uint32_t)hap_total;
        hap_len[j] = hs[j]->get_display_string()->num;
        hap_total += hap_len[j];
    }
    std::vector<uint8_t> hap_bases(hap_total);
    for (size_t j = 0; j < haplotypes_num; ++j) memcpy(hap_bases.data() + hap_off[j], hs[j]->get_display_string()->data, hap_len[j]);
    st.likelihood.resize(reads_num * haplotypes_num);
    phmm_read_prep prep{mapq.data(), min_quality_threshold, model, rovaca_amd::kDefaultGapOpen, rovaca_amd::kDefaultGapCont};
    phmm_batch b{};
    b.n_reads = (uint32_t)reads_num;
    b.n_haps = (uint32_t)haplotypes_num;
    b.n_pairs = (uint64_t)reads_num * haplotypes_num;
    b.read_bases = bases.data();
    b.read_qual = quals.data();
    b.read_off = read_off.data();
    b.read_len = read_len.data();
    b.hap_bases = hap_bases.data();
    b.hap_off = hap_off.data();
    b.hap_len = hap_len.data();
    b.prep = &prep;
    st.rc = phmm_compute(t_ctx.get(), &b, nullptr, st.likelihood.data(), nullptr);  // all reads x all haplotypes, (read, hap) order
    if (st.rc != PHMM_OK) snprintf(st.why, sizeof(st.why), "%s", phmm_last_error());
    return t;
}

DoubleVector2D call_gpu_pairhmm_finish(PairhmmTicket &&ticket, ReadVector &rs, pMemoryPool pool)
{
    PairhmmTicket t(std::move(ticket));
    if (!t.st_) return {};
    PairhmmTicket::State &st = *t.st_;
    const size_t reads_num = st.reads_num, haplotypes_num = st.haplotypes_num;
    if (rs.size() != reads_num) {
        fprintf(stderr, "[rovaca_amd] pairhmm: call_gpu_pairhmm_finish was handed a read list of %zu reads for a region begun with %zu\n",
                rs.size(), reads_num);
        exit(EXIT_FAILURE);
    }
    // the result comes from the caller's arena (pairhmm_internal.cpp:516)
    DoubleVector2D final_likelihood{haplotypes_num, DoubleVector{reads_num, NAN, pool}, pool};
    static thread_local std::vector<double> scratch;
    static thread_local std::vector<int32_t> scratch_len;
    double *likelihood_array = nullptr;
    const int32_t *seq_len = nullptr;
    if (st.batcher) {
        // the region's sums are in the stage: convert them here, on the worker (pairhmm_internal.cpp:577-585), then let go of the stage
        StageSlot &slot = *st.rv.slot;
        ProfClock clk;
        st.rc = RegionBatcher::wait(slot);
        clk.lap(kProfWaitResults);
        if (st.rc != PHMM_OK) snprintf(st.why, sizeof(st.why), "%s", slot.error);
        if (st.rc == PHMM_OK) {
            scratch.resize(reads_num * haplotypes_num);
            scratch_len.resize(reads_num);
            st.rc = phmm_stage_log10(slot.st, phmm_stage_pair_first(slot.st, st.rv.region), reads_num * haplotypes_num, scratch.data(), nullptr, nullptr);
            if (st.rc != PHMM_OK) snprintf(st.why, sizeof(st.why), "%s", phmm_last_error());
            for (size_t i = 0; i < reads_num; ++i) scratch_len[i] = (int32_t)slot.v.read_len[st.rv.read_first + i];
        }
        st.batcher->release(slot);
        clk.lap(kProfConvert);
        st.batcher = nullptr;  // nothing of the ticket is in flight any more
        likelihood_array = scratch.data();
        seq_len = scratch_len.data();
    }
    else {
        likelihood_array = st.likelihood.data();
        seq_len = st.seq_len.data();
    }
    if (st.rc != PHMM_OK) {
        fprintf(stderr, "[rovaca_amd] pairhmm: device evaluation failed (status %d): %s\n", st.rc, st.why);
        exit(EXIT_FAILURE);
    }

    // ---- post-processing (pairhmm_internal.cpp:589-591) ----
    struct PostClock : ProfClock
    {
        ~PostClock() { lap(kProfPostprocess); }
    } post_clock;
    rovaca_amd::normalize_likelihoods(likelihood_array, reads_num, haplotypes_num);
    const std::vector<size_t> drop = rovaca_amd::poorly_modelled_reads(likelihood_array, seq_len, reads_num, haplotypes_num);
    if (drop.empty()) {
        for (size_t i = 0; i < reads_num; ++i)
            for (size_t j = 0; j < haplotypes_num; ++j) final_likelihood[j][i] = likelihood_array[i * haplotypes_num + j];
        return final_likelihood;
    }
    std::vector<char> dropped(reads_num, 0);
    for (size_t r : drop) dropped[r] = 1;
    for (auto it = drop.rbegin(); it != drop.rend(); ++it) rs.erase(rs.begin() + (std::ptrdiff_t)*it);  // back to front, :178-181
    size_t k = 0;
    for (size_t i = 0; i < reads_num; ++i) {
        if (dropped[i]) continue;
        for (size_t j = 0; j < haplotypes_num; ++j) final_likelihood[j][k] = likelihood_array[i * haplotypes_num + j];
        ++k;
    }
    return final_likelihood;
}

DoubleVector2D call_gpu_pairhmm(const HaplotypeVector &hs, ReadVector &rs, int32_t min_quality_threshold,
                                PcrIndelModel pcr_option, pMemoryPool pool)
{
    // the synchronous seam (pairhmm_engine.h:14-15) is begin + finish back to back
    return call_gpu_pairhmm_finish(call_gpu_pairhmm_begin(hs, rs, min_quality_threshold, pcr_option), rs, pool);
}

DoubleVector2D marginal_likelihoods_direct(const DoubleVector2D &old_sample_values, size_t evidence_count, size_t new_allele_count,
                                           const Int32Vector &old2new_idx_map, pMemoryPool pool)
{
    // allele_likelihoods.hpp:712: every new row starts at -inf
    DoubleVector2D result(new_allele_count, DoubleVector(evidence_count, -std::numeric_limits<double>::infinity(), pool), pool);
    const size_t old_allele_count = old2new_idx_map.size();
    for (size_t ai = 0; ai < old_allele_count; ++ai) {
        const int32_t target = old2new_idx_map[ai];
        if (target == rovaca_amd::kMissingAlleleIndex) continue;  // :717-719
        // one old row onto one new row: the flat routine with a single allele keeps the reference's comparison (:724-727)
        const DoubleVector &old_lk = old_sample_values.at(ai);
        DoubleVector &new_lk = result.at((size_t)target);
        for (size_t ri = 0; ri < evidence_count; ++ri)
            if (old_lk[ri] > new_lk[ri]) new_lk[ri] = old_lk[ri];
    }
    return result;
}

}  // namespace rovaca
