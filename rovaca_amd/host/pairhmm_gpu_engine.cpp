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

#include <pthread.h>
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
// (haplotypecaller_engine.cpp:119-285); a region is 1e7-1e8 cells, i.e. tens of microseconds of MI355X time against a
// fixed per-submission cost.  Workers therefore hand their flattened region to one dispatcher thread per device and
// block on it; the dispatcher merges everything that queued up while the previous submission was on the device into ONE
// phmm_batch (tables concatenated, pair indices rebased), runs it, and hands each worker its slice.  No timer: the batch
// is whatever arrived during the last device round trip, so a single caller pays one thread hop and 30-128 callers are
// coalesced automatically.
// ---------------------------------------------------------------------------------------------------------------------
struct RegionJob
{
    // inputs (owned by the submitting worker, alive until done)
    const uint8_t *bases, *quals, *mapq, *hap_bases;  // quals are RAW: H3/H4 run on the device (phmm_read_prep)
    const uint32_t *read_off, *read_len, *hap_off, *hap_len;
    uint32_t n_reads, n_haps;
    size_t read_bytes, hap_bytes;
    int32_t min_q, pcr_model;
    double *out;  // [read][hap]
    // completion: `status` (and `error`, when it is a failure) are written before `done` is released
    std::atomic<bool> done{false};
    int status = PHMM_OK;
    char error[256] = "";  // phmm_last_error() is thread-local to the dispatcher: the text travels with the job
};

// How long a thread polls before it sleeps on a condition variable (ROVACA_AMD_PAIRHMM_SPIN_US, default 0 = never).
// Polling saves the futex wake and the core's idle-state exit on each hand-off, but measured on the GPU box (CPU time capped
// per job) 30 workers fell from 74 000 to 59 000 regions/s with a 300 us budget and 128 workers to a third: the pollers eat
// the cycles the dispatchers need.  Worth trying only where worker threads have cores to themselves.
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
    explicit RegionBatcher(int device) : device_(device), worker_([this] { run(); }) {}
    ~RegionBatcher()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        worker_.join();
    }
    // the job (and everything it points at) must stay alive until wait() has returned
    void submit(RegionJob &job)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            queue_.push_back(&job);
            ++pending_;
        }
        if (dispatcher_sleeps_.load(std::memory_order_acquire)) cv_.notify_one();
    }
    int wait(RegionJob &job)
    {
        if (!spin_until([&] { return job.done.load(std::memory_order_acquire); })) {
            std::unique_lock<std::mutex> lk(mu_);
            ++sleepers_;
            done_cv_.wait(lk, [&] { return job.done.load(std::memory_order_acquire); });
            --sleepers_;
        }
        return job.status;
    }
    int submit_and_wait(RegionJob &job)
    {
        submit(job);
        return wait(job);
    }
    std::atomic<uint64_t> regions{0}, submissions{0};

private:
    void run()
    {
        // named like the reference's own threads (haplotypecaller_engine.cpp:120) so that perf / top -H attribute the time
        pthread_setname_np(pthread_self(), "phmm-gpu-disp");
        phmm_context *ctx = nullptr;
        if (phmm_create(device_, nullptr, &ctx) != PHMM_OK) {
            fprintf(stderr, "[rovaca_amd] pairhmm: cannot create a device context: %s\n", phmm_last_error());
            exit(EXIT_FAILURE);
        }
        std::vector<RegionJob *> taken, jobs;
        std::vector<uint8_t> bases, quals, mapq, haps;
        std::vector<uint32_t> roff, rlen, hoff, hlen, pr, ph;
        std::vector<double> out;
        for (;;) {
            if (!spin_until([&] { return pending_.load(std::memory_order_acquire) != 0; })) {
                std::unique_lock<std::mutex> lk(mu_);
                dispatcher_sleeps_.store(true, std::memory_order_release);
                cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
                dispatcher_sleeps_.store(false, std::memory_order_release);
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (stop_ && queue_.empty()) break;
                taken.assign(queue_.begin(), queue_.end());
                queue_.clear();
                pending_.store(0, std::memory_order_release);
            }
            if (taken.empty()) continue;
            while (!taken.empty()) {
                // one device batch per distinct (threshold, PCR model); in practice they are process-wide constants
                jobs.clear();
                const int32_t mq = taken.front()->min_q, pm = taken.front()->pcr_model;
                for (auto it = taken.begin(); it != taken.end();) {
                    if ((*it)->min_q == mq && (*it)->pcr_model == pm) {
                        jobs.push_back(*it);
                        it = taken.erase(it);
                    }
                    else {
                        ++it;
                    }
                }
                // merge: concatenate the tables, rebase the indices, spell the cross products out as one pair list
                bases.clear();
                quals.clear();
                mapq.clear();
                haps.clear();
                roff.clear();
                rlen.clear();
                hoff.clear();
                hlen.clear();
                pr.clear();
                ph.clear();
                for (RegionJob *j : jobs) {
                    const uint32_t r0 = (uint32_t)rlen.size(), h0 = (uint32_t)hlen.size();
                    const uint32_t rb = (uint32_t)bases.size(), hb = (uint32_t)haps.size();
                    bases.insert(bases.end(), j->bases, j->bases + j->read_bytes);
                    quals.insert(quals.end(), j->quals, j->quals + j->read_bytes);
                    mapq.insert(mapq.end(), j->mapq, j->mapq + j->n_reads);
                    haps.insert(haps.end(), j->hap_bases, j->hap_bases + j->hap_bytes);
                    for (uint32_t r = 0; r < j->n_reads; ++r) {
                        roff.push_back(rb + j->read_off[r]);
                        rlen.push_back(j->read_len[r]);
                    }
                    for (uint32_t h = 0; h < j->n_haps; ++h) {
                        hoff.push_back(hb + j->hap_off[h]);
                        hlen.push_back(j->hap_len[h]);
                    }
                    for (uint32_t r = 0; r < j->n_reads; ++r)
                        for (uint32_t h = 0; h < j->n_haps; ++h) {
                            pr.push_back(r0 + r);
                            ph.push_back(h0 + h);
                        }
                }
                phmm_read_prep prep{mapq.data(), mq, pm, rovaca_amd::kDefaultGapOpen, rovaca_amd::kDefaultGapCont};
                phmm_batch b{};
                b.n_reads = (uint32_t)rlen.size();
                b.n_haps = (uint32_t)hlen.size();
                b.n_pairs = pr.size();
                b.read_bases = bases.data();
                b.read_qual = quals.data();
                b.read_off = roff.data();
                b.read_len = rlen.data();
                b.hap_bases = haps.data();
                b.hap_off = hoff.data();
                b.hap_len = hlen.data();
                b.pair_read = pr.data();
                b.pair_hap = ph.data();
                b.prep = &prep;
                out.resize(pr.size());
                const int rc = phmm_compute(ctx, &b, nullptr, out.data(), nullptr);
                const char *why = rc == PHMM_OK ? "" : phmm_last_error();
                regions += jobs.size();
                submissions += 1;
                size_t at = 0;
                for (RegionJob *j : jobs) {
                    const size_t n = (size_t)j->n_reads * j->n_haps;
                    if (rc == PHMM_OK) memcpy(j->out, out.data() + at, n * sizeof(double));
                    at += n;
                    j->status = rc;
                    if (rc != PHMM_OK) snprintf(j->error, sizeof(j->error), "%s", why);
                    j->done.store(true, std::memory_order_release);  // the job may be gone from here on
                }
                bool wake;
                {
                    std::lock_guard<std::mutex> lk(mu_);  // a worker that is about to sleep either sees `done` or is counted
                    wake = sleepers_ != 0;
                }
                if (wake) done_cv_.notify_all();
            }
        }
        phmm_destroy(ctx);
    }

    int device_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<RegionJob *> queue_;
    std::atomic<uint32_t> pending_{0};
    std::atomic<bool> dispatcher_sleeps_{false};
    int sleepers_ = 0;  // workers blocked on done_cv_ (guarded by mu_)
    bool stop_ = false;
    std::thread worker_;
};

// one per device in use; intentionally never destroyed (their threads would otherwise call into HIP during static teardown)
std::vector<RegionBatcher *> g_batchers;
std::atomic<bool> g_batching{false};
std::atomic<uint32_t> g_next_batcher{0};

}  // namespace

bool init_pairhmm_ptr_gpu(int device_id)
{
    const int n = phmm_device_count();
    if (n <= 0 || device_id >= n) return false;
    g_device.store(device_id);
    const char *env = getenv("ROVACA_AMD_PAIRHMM_BATCHING");
    const bool batching = !(env && env[0] == '0');
    if (batching && g_batchers.empty()) {
        // a few dispatchers per device: while one batch is on the device the next one is being merged and uploaded
        const char *nd = getenv("ROVACA_AMD_PAIRHMM_DISPATCHERS");
        const int per_device = std::max(1, std::min(8, nd ? atoi(nd) : 3));
        for (int k = 0; k < per_device; ++k) {
            if (device_id >= 0) {
                g_batchers.push_back(new RegionBatcher(device_id));
            }
            else {
                for (int d = 0; d < n; ++d) g_batchers.push_back(new RegionBatcher(d));
            }
        }
    }
    g_batching.store(batching);
    call_pairhmm = call_gpu_pairhmm;
    g_active.store(true);
    return true;
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

// A region between call_gpu_pairhmm_begin and call_gpu_pairhmm_finish: the flattened tables the dispatcher reads, the
// [read][hap] result it writes, and the job record.  Heap-allocated once per region; nothing in it refers to the caller's
// vectors, so the caller may go on to its next region while this one is on the device.
struct PairhmmTicket::State
{
    std::vector<uint32_t> read_off, read_len, hap_off, hap_len;
    std::vector<int32_t> seq_len;
    std::vector<uint8_t> bases, quals, mapq, hap_bases;
    std::vector<double> likelihood;  // [read][hap]
    size_t reads_num = 0, haplotypes_num = 0;
    RegionJob job{};
    RegionBatcher *batcher = nullptr;  // null: evaluated inside begin (batching off), `rc` / `why` hold the outcome
    int rc = PHMM_OK;
    char why[256] = "";
};
PairhmmTicket::PairhmmTicket() = default;
PairhmmTicket::PairhmmTicket(PairhmmTicket &&) noexcept = default;
PairhmmTicket &PairhmmTicket::operator=(PairhmmTicket &&o) noexcept
{
    // assigning over a ticket that is still in flight (a look-ahead loop that skipped finish for the previous region): the
    // dispatcher reads its tables and writes its result buffer until the job is done, so wait first -- as the destructor does
    if (this != &o) {
        if (st_ && st_->batcher) st_->batcher->wait(st_->job);
        st_ = std::move(o.st_);
    }
    return *this;
}
PairhmmTicket::~PairhmmTicket()
{
    // a ticket dropped without finish: the dispatcher may still be writing into it
    if (st_ && st_->batcher) st_->batcher->wait(st_->job);
}

PairhmmTicket call_gpu_pairhmm_begin(const HaplotypeVector &hs, const ReadVector &rs, int32_t min_quality_threshold, PcrIndelModel pcr_option)
{
    PairhmmTicket t;
    if (hs.empty() || rs.empty()) return t;  // finish returns {} (pairhmm_internal.cpp:508-510)
    t.st_.reset(new PairhmmTicket::State());
    PairhmmTicket::State &st = *t.st_;
    const size_t reads_num = st.reads_num = rs.size();
    const size_t haplotypes_num = st.haplotypes_num = hs.size();

    // ---- flatten: read table ----
    st.read_off.resize(reads_num);
    st.read_len.resize(reads_num);
    st.seq_len.resize(reads_num);
    size_t total = 0;
    int32_t max_len = 0;
    for (size_t i = 0; i < reads_num; ++i) {
        st.seq_len[i] = rs[i]->seq_length();
        st.read_off[i] = (uint32_t)total;
        st.read_len[i] = (uint32_t)st.seq_len[i];
        total += (size_t)st.seq_len[i];
        max_len = std::max(max_len, st.seq_len[i]);
    }
    // H3 (quality modification) and H4 (PCR indel model) run on the device (phmm_read_prep); the host only gathers bytes
    st.bases.resize(total);
    st.quals.resize(total);
    st.mapq.resize(reads_num);
    std::vector<uint8_t> tmp(sizeof(Bases) + (size_t)max_len + 1);
    pBases read_bases = reinterpret_cast<pBases>(tmp.data());
    for (size_t i = 0; i < reads_num; ++i) {
        pReadRecord read = rs[i];
        const uint32_t n = st.read_len[i], o = st.read_off[i];
        read->decode_base(read_bases);
        memcpy(st.bases.data() + o, read_bases->data, n);
        memcpy(st.quals.data() + o, read->qual(), n);
        st.mapq[i] = read->mapping_quality();
    }
    // ---- flatten: haplotype table ----
    st.hap_off.resize(haplotypes_num);
    st.hap_len.resize(haplotypes_num);
    size_t hap_total = 0;
    for (size_t j = 0; j < haplotypes_num; ++j) {
        st.hap_off[j] = (uint32_t)hap_total;
        st.hap_len[j] = hs[j]->get_display_string()->num;
        hap_total += st.hap_len[j];
    }
    st.hap_bases.resize(hap_total);
    for (size_t j = 0; j < haplotypes_num; ++j)
        memcpy(st.hap_bases.data() + st.hap_off[j], hs[j]->get_display_string()->data, st.hap_len[j]);

    // ---- the device: all reads x all haplotypes, (read, hap) order ----
    st.likelihood.resize(reads_num * haplotypes_num);
    const int model = static_cast<int>(pcr_option);
    if (g_batching.load() && !g_batchers.empty()) {
        RegionJob &job = st.job;
        job.bases = st.bases.data();
        job.quals = st.quals.data();
        job.mapq = st.mapq.data();
        job.hap_bases = st.hap_bases.data();
        job.read_off = st.read_off.data();
        job.read_len = st.read_len.data();
        job.hap_off = st.hap_off.data();
        job.hap_len = st.hap_len.data();
        job.n_reads = (uint32_t)reads_num;
        job.n_haps = (uint32_t)haplotypes_num;
        job.read_bytes = total;
        job.hap_bytes = hap_total;
        job.min_q = min_quality_threshold;
        job.pcr_model = model;
        job.out = st.likelihood.data();
        st.batcher = g_batchers[g_next_batcher.fetch_add(1) % g_batchers.size()];
        st.batcher->submit(job);
    }
    else {
        phmm_read_prep prep{st.mapq.data(), min_quality_threshold, model, rovaca_amd::kDefaultGapOpen, rovaca_amd::kDefaultGapCont};
        phmm_batch b{};
        b.n_reads = (uint32_t)reads_num;
        b.n_haps = (uint32_t)haplotypes_num;
        b.n_pairs = (uint64_t)reads_num * haplotypes_num;
        b.read_bases = st.bases.data();
        b.read_qual = st.quals.data();
        b.read_off = st.read_off.data();
        b.read_len = st.read_len.data();
        b.hap_bases = st.hap_bases.data();
        b.hap_off = st.hap_off.data();
        b.hap_len = st.hap_len.data();
        b.prep = &prep;
        st.rc = phmm_compute(t_ctx.get(), &b, nullptr, st.likelihood.data(), nullptr);
        if (st.rc != PHMM_OK) snprintf(st.why, sizeof(st.why), "%s", phmm_last_error());
    }
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
    if (st.batcher) {
        st.rc = st.batcher->wait(st.job);
        if (st.rc != PHMM_OK) snprintf(st.why, sizeof(st.why), "%s", st.job.error);
        st.batcher = nullptr;  // nothing of the ticket is in flight any more
    }
    if (st.rc != PHMM_OK) {
        fprintf(stderr, "[rovaca_amd] pairhmm: device evaluation failed (status %d): %s\n", st.rc, st.why);
        exit(EXIT_FAILURE);
    }

    // ---- post-processing (pairhmm_internal.cpp:589-591) ----
    double *likelihood_array = st.likelihood.data();
    rovaca_amd::normalize_likelihoods(likelihood_array, reads_num, haplotypes_num);
    const std::vector<size_t> drop = rovaca_amd::poorly_modelled_reads(likelihood_array, st.seq_len.data(), reads_num, haplotypes_num);
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
