// sw_gpu_engine.cpp -- see sw_gpu_engine.h.  Host glue only: the alignment itself is include/smithwaterman_hip.h.
#include "sw_gpu_engine.h"
#include "pairhmm_host_logic.h"

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include <climits>
#include <linux/futex.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <vector>

#include "../../include/smithwaterman_hip.h"

#include <algorithm>
#include <chrono>

namespace
{

// device of the contexts created from here on: a fixed one, or (negative, the default) round-robin over the visible
// devices per creating thread -- the rule of the PairHMM adapter (init_pairhmm_ptr_gpu), so that both engines of a run land
// on the same GPUs
std::atomic<int> g_sw_device{-1};
std::atomic<uint32_t> g_sw_next_device{0};

thread_local int t_sw_device = -1;  // a dispatcher thread's own device (set before its context exists)

int pick_device()
{
    if (t_sw_device >= 0) return t_sw_device;
    const int fixed = g_sw_device.load();
    if (fixed >= 0) return fixed;
    const int n = swhip_device_count();
    return n > 0 ? (int)(g_sw_next_device.fetch_add(1) % (uint32_t)n) : 0;
}

struct ThreadContext
{
    swhip_context *ctx = nullptr;
    ~ThreadContext()
    {
        if (ctx) swhip_destroy(ctx);
    }
    swhip_context *get()
    {
        if (!ctx && swhip_create(pick_device(), nullptr, &ctx) != SWHIP_OK) {
            fprintf(stderr, "[rovaca_amd] smith-waterman: cannot create a device context: %s\n", swhip_last_error());
            exit(EXIT_FAILURE);  // as the reference does on an unusable engine (pairhmm_engine.cpp:32-35)
        }
        return ctx;
    }
};
thread_local ThreadContext t_ctx;

}  // namespace

namespace
{

using rovaca::SwAlignment;
using rovaca::SwJob;

// Where the realignment seam's host time goes (always on, as gpu_pairhmm_profile): nanoseconds summed over threads, plus two counters
enum SwProfileSlot {
    kSwWorkerBlocked,   // worker: from handing the request over until it is woken
    kSwWorkerUnpack,    // worker: building the per-alignment vectors the seam returns
    kSwQueueWait,       // requests: from submit until a dispatcher takes them (summed over requests)
    kSwDispGather,      // dispatcher: concatenating the requests' sequences into one batch
    kSwDispDeviceCall,  // dispatcher: swhip_align (classify, one copy, launches, the wait)
    kSwDeviceKernels,   // ... of which device time of the launches (HIP events)
    kSwDispScatter,     // dispatcher: copying each request's slice out and waking it
    kSwDispIdle,        // dispatcher: nothing queued
    kSwRequests, kSwDeviceCalls,  // counters
    kSwProfSlots
};
std::atomic<uint64_t> g_sw_prof[kSwProfSlots];
static uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// reusable host buffers of one device call (per calling thread: a dispatcher, or a worker on the unbatched path)
struct CallBuffers
{
    std::vector<uint8_t> refs, seqs;
    std::vector<uint32_t> roff, rlen, soff, slen, idx;
    std::vector<swhip_alignment> res;
    std::vector<uint32_t> cigars;
};

// one device call for `n` jobs on the calling thread's context; results in buf.res / buf.cigars (cigar_off indexes buf.cigars)
bool align_raw(const SwJob *jobs, size_t n, const sw_parameters &para, CallBuffers &buf)
{
    const uint64_t t0 = now_ns();
    buf.refs.clear();
    buf.seqs.clear();
    buf.roff.resize(n);
    buf.rlen.resize(n);
    buf.soff.resize(n);
    buf.slen.resize(n);
    buf.idx.resize(n);
    size_t cap = 0;
    for (size_t k = 0; k < n; ++k) {
        buf.roff[k] = (uint32_t)buf.refs.size();
        buf.rlen[k] = (uint32_t)jobs[k].ref_len;
        buf.refs.insert(buf.refs.end(), jobs[k].ref, jobs[k].ref + jobs[k].ref_len);
        buf.soff[k] = (uint32_t)buf.seqs.size();
        buf.slen[k] = (uint32_t)jobs[k].seq_len;
        buf.seqs.insert(buf.seqs.end(), jobs[k].seq, jobs[k].seq + jobs[k].seq_len);
        buf.idx[k] = (uint32_t)k;
        cap += (size_t)jobs[k].ref_len + (size_t)jobs[k].seq_len + 2;
    }
    swhip_batch b{};
    b.n_refs = b.n_seqs = (uint32_t)n;
    b.n_pairs = n;
    b.ref_bases = buf.refs.data();
    b.ref_off = buf.roff.data();
    b.ref_len = buf.rlen.data();
    b.seq_bases = buf.seqs.data();
    b.seq_off = buf.soff.data();
    b.seq_len = buf.slen.data();
    b.pair_ref = buf.idx.data();
    b.pair_seq = buf.idx.data();
    const swhip_params prm{para.match, para.mismatch, para.gap_open, para.gap_extend};
    buf.res.resize(n);
    if (buf.cigars.size() < cap) buf.cigars.resize(cap);
    uint64_t used = 0;
    swhip_stats stats{};
    g_sw_prof[kSwDispGather].fetch_add(now_ns() - t0, std::memory_order_relaxed);
    const int rc = swhip_align(t_ctx.get(), &prm, &b, buf.res.data(), buf.cigars.data(), cap, &used, &stats);
    g_sw_prof[kSwDeviceKernels].fetch_add((uint64_t)(stats.kernel_ms * 1e6), std::memory_order_relaxed);
    if (rc != SWHIP_OK) {
        fprintf(stderr, "[rovaca_amd] smith-waterman: %s\n", swhip_last_error());
        return false;
    }
    return true;
}

// what the seam hands back, from the flat results of a call (run by the thread that asked: the per-alignment vectors are its allocations)
void unpack(const swhip_alignment *res, const uint32_t *cigars, size_t n, SwAlignment *out)
{
    for (size_t k = 0; k < n; ++k) {
        out[k].score = res[k].score;
        out[k].max_i = res[k].max_i;
        out[k].max_j = res[k].max_j;
        out[k].alignment_offset = res[k].alignment_offset;
        out[k].too_long = res[k].too_long != 0;
        out[k].cigar.assign(cigars + res[k].cigar_off, cigars + res[k].cigar_off + res[k].cigar_count);
    }
}

bool align_now(const SwJob *jobs, size_t n, const sw_parameters &para, SwAlignment *out)
{
    if (n == 0) return true;
    static thread_local CallBuffers buf;
    if (!align_raw(jobs, n, para, buf)) return false;
    unpack(buf.res.data(), buf.cigars.data(), n, out);
    return true;
}

static long futex_call(std::atomic<uint32_t> *addr, int op, uint32_t val)
{
    return syscall(SYS_futex, reinterpret_cast<uint32_t *>(addr), op, val, nullptr, nullptr, 0);
}

// Cross-region batching, as for PairHMM (pairhmm_gpu_engine.cpp): the reference's workers realign one region at a time
// (haplotypecaller_engine.cpp:264), 48 alignments or so -- a dozen waves.  Dispatcher threads merge every request that queued
// up while the previous device call was in flight into one call.  The dispatcher's share per request is a copy of its slice of
// the flat results (records and CIGAR elements) into buffers the request owns; the per-alignment vectors the seam returns are
// built by the worker that asked, after it has been woken -- through a futex word of its own, so that a finished call wakes
// exactly the workers it served.  ROVACA_AMD_SW_BATCHING=0 bypasses it (each worker calls the device itself);
// ROVACA_AMD_SW_DISPATCHERS sets their number.
struct Request
{
    const SwJob *jobs = nullptr;
    size_t n = 0;
    sw_parameters para{};
    std::vector<swhip_alignment> res;  // [n], cigar_off rebased to `cigars`
    std::vector<uint32_t> cigars;
    bool ok = true;
    std::atomic<uint32_t> done{0};     // futex word
    uint64_t t_submit = 0;             // profile: when it was queued
};

class SwBatcher
{
public:
    // one queue per device, `n_threads` dispatchers behind it: whichever is free takes everything that is waiting, so a request never
    // sits behind a busy dispatcher while another one idles (round 3; before, every dispatcher had a queue of its own)
    SwBatcher(int device, int n_threads)
    {
        for (int i = 0; i < n_threads; ++i) threads_.emplace_back([this, device] { run(device); });
    }
    // the request (and what it points at) must stay alive until wait() has returned
    void submit(Request &rq)
    {
        bool wake;
        rq.t_submit = now_ns();
        {
            std::lock_guard<std::mutex> lk(mu_);
            queue_.push_back(&rq);
            wake = sleeping_ > 0;
        }
        if (wake) cv_.notify_one();
    }
    static bool wait(Request &rq)
    {
        const uint64_t t0 = now_ns();
        while (rq.done.load(std::memory_order_acquire) == 0) futex_call(&rq.done, FUTEX_WAIT_PRIVATE, 0);
        g_sw_prof[kSwWorkerBlocked].fetch_add(now_ns() - t0, std::memory_order_relaxed);
        return rq.ok;
    }
    bool submit_and_wait(Request &rq)
    {
        submit(rq);
        return wait(rq);
    }
    std::atomic<uint64_t> requests{0}, submissions{0};

private:
    static bool same(const sw_parameters &a, const sw_parameters &b)
    {
        return a.match == b.match && a.mismatch == b.mismatch && a.gap_open == b.gap_open && a.gap_extend == b.gap_extend;
    }
    void run(int device)
    {
        t_sw_device = device;
        // named like the reference's own threads (haplotypecaller_engine.cpp:120) so that perf / top -H attribute the time
        pthread_setname_np(pthread_self(), "sw-gpu-disp");
        std::vector<Request *> taken, group;
        std::vector<SwJob> jobs;
        CallBuffers buf;
        for (;;) {
            {
                const uint64_t t_idle = now_ns();
                std::unique_lock<std::mutex> lk(mu_);
                ++sleeping_;
                cv_.wait(lk, [&] { return !queue_.empty(); });
                --sleeping_;
                taken.assign(queue_.begin(), queue_.end());
                queue_.clear();
                lk.unlock();
                const uint64_t t_taken = now_ns();
                g_sw_prof[kSwDispIdle].fetch_add(t_taken - t_idle, std::memory_order_relaxed);
                uint64_t waited = 0;
                for (Request *r : taken) waited += t_taken - r->t_submit;
                g_sw_prof[kSwQueueWait].fetch_add(waited, std::memory_order_relaxed);
                g_sw_prof[kSwRequests].fetch_add(taken.size(), std::memory_order_relaxed);
            }
            while (!taken.empty()) {
                group.clear();
                const sw_parameters para = taken.front()->para;
                for (auto it = taken.begin(); it != taken.end();) {
                    if (same((*it)->para, para)) {
                        group.push_back(*it);
                        it = taken.erase(it);
                    }
                    else {
                        ++it;
                    }
                }
                const uint64_t t_g = now_ns();
                jobs.clear();
                for (Request *r : group) jobs.insert(jobs.end(), r->jobs, r->jobs + r->n);
                const bool ok = align_raw(jobs.data(), jobs.size(), para, buf);
                const uint64_t t_called = now_ns();
                g_sw_prof[kSwDispDeviceCall].fetch_add(t_called - t_g, std::memory_order_relaxed);  // (align_raw reports its gather share itself)
                g_sw_prof[kSwDeviceCalls].fetch_add(1, std::memory_order_relaxed);
                requests += group.size();
                submissions += 1;
                size_t at = 0;
                for (Request *r : group) {
                    if (ok) {
                        // the request's slice: records as they are, CIGAR elements packed into the request's own buffer
                        uint32_t cur = 0;
                        for (size_t k = 0; k < r->n; ++k) {
                            swhip_alignment a = buf.res[at + k];
                            if (r->cigars.size() < (size_t)cur + a.cigar_count) r->cigars.resize((size_t)cur + a.cigar_count + 64);
                            memcpy(r->cigars.data() + cur, buf.cigars.data() + a.cigar_off, (size_t)a.cigar_count * sizeof(uint32_t));
                            a.cigar_off = cur;
                            cur += a.cigar_count;
                            r->res[k] = a;
                        }
                    }
                    at += r->n;
                    r->ok = ok;
                    // From the store on the request may be gone (its owner can return without ever sleeping), so nothing of it is
                    // read again: the wake is unconditional and uses the word's address only as the kernel's key -- at worst a
                    // spurious wake of whoever reuses that address, which every waiter tolerates.
                    std::atomic<uint32_t> *word = &r->done;
                    r->done.store(1);
                    futex_call(word, FUTEX_WAKE_PRIVATE, INT32_MAX);
                }
                g_sw_prof[kSwDispScatter].fetch_add(now_ns() - t_called, std::memory_order_relaxed);
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Request *> queue_;
    int sleeping_ = 0;  // dispatchers waiting for work; guarded by mu_
    std::vector<std::thread> threads_;
};

std::vector<SwBatcher *> g_sw_batchers;  // never destroyed: their threads would call into HIP during static teardown
std::once_flag g_sw_once;
std::atomic<uint32_t> g_sw_next{0};

void start_batchers()
{
    const char *on = getenv("ROVACA_AMD_SW_BATCHING");
    if (on && atoi(on) == 0) return;
    const char *nd = getenv("ROVACA_AMD_SW_DISPATCHERS");
    // A batch of up to ~2 000 alignments takes as long as one of 48 (lone waves), so merging is free on the device; what a second
    // dispatcher buys is a second call in flight while the first is being unpacked.  Measured with PairHMM in front
    // (tools/sw_adapter_sweep.sh, regions of 48 x 9, 16 host cores), 1 / 2 / 3 dispatchers: 30 synchronous workers 43.8 / 45.4 /
    // 42.2 k regions/s, 30 workers x 4 in flight 99.6 / 122 / 112, 128 synchronous 104 / 117 / 104; 8 synchronous 17.3 / 16.3 /
    // 13.5 and 128 x 4 (the cores are the limit there) 161 / 113 / 109.
    int n = nd ? atoi(nd) : 2;
    n = n < 1 ? 1 : n > 16 ? 16 : n;
    // with round-robin devices every device gets its queue and dispatchers (a dispatcher's context is created on its own thread)
    const int fixed = g_sw_device.load();
    if (fixed >= 0) {
        g_sw_batchers.push_back(new SwBatcher(fixed, n));
        return;
    }
    // (test hook ROVACA_AMD_SW_DEVICES=0,0: several queues on one card -- pairhmm_host_logic.h, batcher_devices)
    for (int d : rovaca_amd::batcher_devices("ROVACA_AMD_SW_DEVICES", std::max(1, swhip_device_count()))) g_sw_batchers.push_back(new SwBatcher(d, n));
}

// a request as the worker prepares it: result slots sized, CIGAR buffer with room for the usual handful of elements per alignment
void prepare(Request &rq, const SwJob *jobs, size_t n, const sw_parameters &para)
{
    rq.jobs = jobs;
    rq.n = n;
    rq.para = para;
    rq.res.resize(n);
    rq.cigars.resize(8 * n + 64);
}

}  // namespace

namespace rovaca
{

bool sw_hip_align_batch(const SwJob *jobs, size_t n, const sw_parameters &para, std::vector<SwAlignment> &out)
{
    out.assign(n, SwAlignment{});
    if (n == 0) return true;
    std::call_once(g_sw_once, start_batchers);
    if (g_sw_batchers.empty()) return align_now(jobs, n, para, out.data());
    Request rq;
    prepare(rq, jobs, n, para);
    if (!g_sw_batchers[g_sw_next.fetch_add(1) % g_sw_batchers.size()]->submit_and_wait(rq)) return false;
    const uint64_t t_u = now_ns();
    unpack(rq.res.data(), rq.cigars.data(), n, out.data());
    g_sw_prof[kSwWorkerUnpack].fetch_add(now_ns() - t_u, std::memory_order_relaxed);
    return true;
}

// The same call in two halves, as for PairHMM (pairhmm_gpu_engine.h): begin queues the alignments and returns, finish collects.
struct SwTicket::State
{
    std::vector<SwJob> jobs;  // the job records are copied; the sequence bytes they point at stay the caller's
    std::vector<SwAlignment> out;  // unbatched path only
    Request rq;
    SwBatcher *batcher = nullptr;  // null: evaluated inside begin (batching off)
    bool ok = true;
};
SwTicket::SwTicket() = default;
SwTicket::SwTicket(SwTicket &&) noexcept = default;
SwTicket &SwTicket::operator=(SwTicket &&o) noexcept
{
    // assigning over a ticket whose request is still queued or on the device: its State (job records, result slots) must
    // outlive the dispatcher's writes, exactly as in the destructor
    if (this != &o) {
        if (st_ && st_->batcher) (void)SwBatcher::wait(st_->rq);
        st_ = std::move(o.st_);
    }
    return *this;
}
SwTicket::~SwTicket()
{
    if (st_ && st_->batcher) (void)SwBatcher::wait(st_->rq);  // dropped without finish: the dispatcher may still be writing into it
}

SwTicket sw_hip_align_batch_begin(const SwJob *jobs, size_t n, const sw_parameters &para)
{
    SwTicket t;
    if (n == 0) return t;
    t.st_.reset(new SwTicket::State());
    SwTicket::State &st = *t.st_;
    st.jobs.assign(jobs, jobs + n);
    std::call_once(g_sw_once, start_batchers);
    if (g_sw_batchers.empty()) {
        st.out.assign(n, SwAlignment{});
        st.ok = align_now(st.jobs.data(), n, para, st.out.data());
        return t;
    }
    prepare(st.rq, st.jobs.data(), n, para);
    st.batcher = g_sw_batchers[g_sw_next.fetch_add(1) % g_sw_batchers.size()];
    st.batcher->submit(st.rq);
    return t;
}

bool sw_hip_align_batch_finish(SwTicket &&ticket, std::vector<SwAlignment> &out)
{
    SwTicket t(std::move(ticket));
    out.clear();
    if (!t.st_) return true;
    SwTicket::State &st = *t.st_;
    if (st.batcher) {
        st.ok = SwBatcher::wait(st.rq);
        st.batcher = nullptr;
        if (st.ok) {
            const uint64_t t_u = now_ns();
            out.assign(st.rq.n, SwAlignment{});
            unpack(st.rq.res.data(), st.rq.cigars.data(), st.rq.n, out.data());
            g_sw_prof[kSwWorkerUnpack].fetch_add(now_ns() - t_u, std::memory_order_relaxed);
        }
        return st.ok;
    }
    out = std::move(st.out);
    return st.ok;
}

int sw_hip_batcher_count() { return (int)g_sw_batchers.size(); }

void sw_hip_batching_stats(uint64_t *requests, uint64_t *submissions)
{
    uint64_t r = 0, s = 0;
    for (SwBatcher *b : g_sw_batchers) {
        r += b->requests;
        s += b->submissions;
    }
    if (requests) *requests = r;
    if (submissions) *submissions = s;
}

}  // namespace rovaca

extern "C" {

// what sw_avx512_run leaves in the handle (sw_avx.c:488-500), from one alignment of a batch
static void store_alignment(p_lib_sw_avx p, const rovaca::SwAlignment &a)
{
    p->score = a.score;
    p->max_i = a.max_i;
    p->max_j = a.max_j;
    p->alignment_offset = a.alignment_offset;
    // sw_avx.c:349-358 keeps an element only while cur_size + 1 < cigar_max_len: at most cigar_max_len - 1 of them
    uint32_t count = (uint32_t)a.cigar.size();
    const uint32_t keep = p->cigar_max_len > 0 ? p->cigar_max_len - 1 : 0;
    if (count > keep) count = keep;
    memcpy(p->cigar, a.cigar.data(), count * sizeof(uint32_t));
    p->cigar_count = count;
}
static bool too_long_for(p_lib_sw_avx p)
{
    const int len = p->len1 > p->len2 ? p->len1 : p->len2;
    return len >= p->max_seq_len || len >= SWHIP_MAX_SEQ_LEN;  // sw_avx.c:490-495
}

void sw_hip_run(p_lib_sw_avx p)
{
    if (too_long_for(p)) {
        p->alignment_offset = 1;
        return;
    }
    const rovaca::SwJob job{p->seq1, p->len1, p->seq2, p->len2};
    std::vector<rovaca::SwAlignment> out;
    if (!rovaca::sw_hip_align_batch(&job, 1, p->para, out)) exit(EXIT_FAILURE);
    store_alignment(p, out[0]);
}

// Every handle as sw_hip_run would leave it, in ONE device call per parameter set: the C entry for a caller that has several
// alignments at hand -- the assembler aligns every haplotype path of a region to the padded reference one call at a time
// (apply/assemble/hc_assemble_cigar_cacl.c:417-428; INTEGRATION.md has the patch that collects a region's paths first).  Each
// handle carries its own sequences, parameters and CIGAR buffer.  Returns 0; a device failure logs to stderr and returns 1 with
// no handle written.
int sw_hip_align_many(p_lib_sw_avx *handles, size_t n)
{
    if (n == 0) return 0;
    if (!handles) return 1;
    std::vector<size_t> todo;
    todo.reserve(n);
    for (size_t k = 0; k < n; ++k) {
        if (!handles[k]) return 1;
        if (!too_long_for(handles[k])) todo.push_back(k);
    }
    std::vector<char> taken(todo.size(), 0);
    std::vector<std::vector<rovaca::SwAlignment>> results;
    std::vector<std::vector<size_t>> groups;
    for (size_t i = 0; i < todo.size(); ++i) {
        if (taken[i]) continue;
        const sw_parameters para = handles[todo[i]]->para;  // one device call per distinct parameter set (two in the reference: sw_avx.c:13-14)
        std::vector<size_t> group;
        std::vector<rovaca::SwJob> jobs;
        for (size_t j = i; j < todo.size(); ++j) {
            p_lib_sw_avx q = handles[todo[j]];
            if (taken[j] || q->para.match != para.match || q->para.mismatch != para.mismatch || q->para.gap_open != para.gap_open ||
                q->para.gap_extend != para.gap_extend)
                continue;
            taken[j] = 1;
            group.push_back(todo[j]);
            jobs.push_back(rovaca::SwJob{q->seq1, q->len1, q->seq2, q->len2});
        }
        std::vector<rovaca::SwAlignment> out;
        if (!rovaca::sw_hip_align_batch(jobs.data(), jobs.size(), para, out)) return 1;
        results.push_back(std::move(out));
        groups.push_back(std::move(group));
    }
    for (size_t k = 0; k < n; ++k)
        if (too_long_for(handles[k])) handles[k]->alignment_offset = 1;
    for (size_t g = 0; g < groups.size(); ++g)
        for (size_t i = 0; i < groups[g].size(); ++i) store_alignment(handles[groups[g][i]], results[g][i]);
    return 0;
}

void sw_hip_set_device(int device_id) { g_sw_device.store(device_id); }

int sw_hip_profile(double *values, const char **names, int cap)
{
    static const char *const kNames[kSwProfSlots] = {"sw worker: blocked until its alignments are back", "sw worker: building the per-alignment results",
                                                     "sw requests: queued until a dispatcher took them", "sw dispatcher: gathering the requests' sequences",
                                                     "sw dispatcher: gather + swhip_align (copy, launches, wait)", "sw device: time of the launches (HIP events)",
                                                     "sw dispatcher: handing slices back, waking", "sw dispatcher: idle (nothing queued)",
                                                     "sw requests (count)", "sw device calls (count)"};
    for (int i = 0; i < kSwProfSlots && i < cap; ++i) {
        if (values) values[i] = i >= kSwRequests ? (double)g_sw_prof[i].load() : (double)g_sw_prof[i].load() * 1e-9;
        if (names) names[i] = kNames[i];
    }
    return kSwProfSlots;
}

int sw_hip_install(p_lib_sw_avx p)
{
    if (!p) return 0;
    swhip_context *probe = nullptr;
    const int dev = g_sw_device.load() >= 0 ? g_sw_device.load() : 0;
    if (swhip_create(dev, nullptr, &probe) != SWHIP_OK) return 0;  // no device: the AVX selection stays
    swhip_destroy(probe);
    p->avx_function = sw_hip_run;
    return 1;
}

// layout of the handle as this library sees it (tests compare it with the reference's own header where that is available)
int rovaca_amd_sw_layout(int32_t *out, int cap)
{
    const int32_t v[] = {(int32_t)sizeof(lib_sw_avx),
                         (int32_t)offsetof(lib_sw_avx, seq1), (int32_t)offsetof(lib_sw_avx, seq2), (int32_t)offsetof(lib_sw_avx, len1),
                         (int32_t)offsetof(lib_sw_avx, len2), (int32_t)offsetof(lib_sw_avx, score), (int32_t)offsetof(lib_sw_avx, max_i),
                         (int32_t)offsetof(lib_sw_avx, max_j), (int32_t)offsetof(lib_sw_avx, cigar), (int32_t)offsetof(lib_sw_avx, cigar_count),
                         (int32_t)offsetof(lib_sw_avx, cigar_max_len), (int32_t)offsetof(lib_sw_avx, alignment_offset),
                         (int32_t)offsetof(lib_sw_avx, max_seq_len), (int32_t)offsetof(lib_sw_avx, para),
                         (int32_t)offsetof(lib_sw_avx, avx_function), (int32_t)offsetof(lib_sw_avx, data)};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int k = 0; k < n && k < cap; ++k) out[k] = v[k];
    return n;
}

// Test shim (plain C types for ctypes): builds a handle the way sw_avx_init lays it out (sw_avx.c:401-437, buffers only as
// far as sw_hip_run needs them), installs the GPU candidate, runs one alignment through the function pointer.
int rovaca_amd_sw_run_through_handle(const uint8_t *seq1, int32_t len1, const uint8_t *seq2, int32_t len2, int is_assemble,
                                     int32_t *score, int32_t *max_i, int32_t *max_j, int32_t *alignment_offset, uint32_t *cigar,
                                     uint32_t cigar_cap, uint32_t *cigar_count)
{
    const uint32_t max_seq = SWHIP_MAX_SEQ_LEN;
    const size_t bytes = sizeof(lib_sw_avx) + 2 * (size_t)max_seq + max_seq * sizeof(uint32_t);
    p_lib_sw_avx p = (p_lib_sw_avx)calloc(1, bytes);
    if (!p) return -2;
    p->max_seq_len = (int32_t)max_seq;
    p->seq1 = p->data;
    p->seq2 = p->data + max_seq;
    p->cigar = (uint32_t *)(p->data + 2 * (size_t)max_seq);
    p->cigar_max_len = max_seq;
    p->para = is_assemble ? sw_parameters{200, -150, -260, -11} : sw_parameters{10, -15, -30, -5};  // sw_avx.c:13-14
    if (!sw_hip_install(p)) {
        free(p);
        return -1;
    }
    if ((uint32_t)len1 < max_seq) memcpy(p->seq1, seq1, (size_t)len1);
    if ((uint32_t)len2 < max_seq) memcpy(p->seq2, seq2, (size_t)len2);
    p->len1 = (int16_t)len1;
    p->len2 = (int16_t)len2;
    p->avx_function(p);
    *score = p->score;
    *max_i = p->max_i;
    *max_j = p->max_j;
    *alignment_offset = p->alignment_offset;
    *cigar_count = p->cigar_count;
    for (uint32_t k = 0; k < p->cigar_count && k < cigar_cap; ++k) cigar[k] = p->cigar[k];
    free(p);
    return 0;
}

// Test shim for sw_hip_align_many: n handles laid out as sw_avx_init does, alternating parameter sets when mixed != 0, filled like
// hc_assemle_cigar_cacl_run_sw fills its one handle; results[k] = {score, max_i, max_j, alignment_offset, cigar_count}, the CIGAR of
// alignment k at cigars[cigar_at[k] ..).  Returns what sw_hip_align_many returned, -1 when no device, -2 out of memory.
int rovaca_amd_sw_many_through_handles(const uint8_t *ref_bytes, const uint32_t *ref_off, const uint32_t *ref_len, const uint8_t *seq_bytes,
                                       const uint32_t *seq_off, const uint32_t *seq_len, uint32_t n, int is_assemble, int mixed, int32_t *result,
                                       uint32_t *cigars, const uint32_t *cigar_at)
{
    const uint32_t max_seq = SWHIP_MAX_SEQ_LEN;
    const size_t bytes = sizeof(lib_sw_avx) + 2 * (size_t)max_seq + max_seq * sizeof(uint32_t);
    std::vector<p_lib_sw_avx> h(n, nullptr);
    int rc = 0;
    for (uint32_t k = 0; k < n && rc == 0; ++k) {
        p_lib_sw_avx p = (p_lib_sw_avx)calloc(1, bytes);
        if (!p) {
            rc = -2;
            break;
        }
        h[k] = p;
        p->max_seq_len = (int32_t)max_seq;
        p->seq1 = p->data;
        p->seq2 = p->data + max_seq;
        p->cigar = (uint32_t *)(p->data + 2 * (size_t)max_seq);
        p->cigar_max_len = max_seq;
        const bool asm_k = mixed ? ((k & 1u) != 0) : (is_assemble != 0);
        p->para = asm_k ? sw_parameters{200, -150, -260, -11} : sw_parameters{10, -15, -30, -5};  // sw_avx.c:13-14
        if (k == 0 && !sw_hip_install(p)) rc = -1;
        if (ref_len[k] < max_seq) memcpy(p->seq1, ref_bytes + ref_off[k], ref_len[k]);
        if (seq_len[k] < max_seq) memcpy(p->seq2, seq_bytes + seq_off[k], seq_len[k]);
        p->len1 = (int16_t)ref_len[k];
        p->len2 = (int16_t)seq_len[k];
    }
    if (rc == 0) rc = sw_hip_align_many(h.data(), n);
    for (uint32_t k = 0; k < n; ++k) {
        if (!h[k]) continue;
        if (rc == 0) {
            int32_t *r = result + 5 * (size_t)k;
            r[0] = h[k]->score;
            r[1] = h[k]->max_i;
            r[2] = h[k]->max_j;
            r[3] = h[k]->alignment_offset;
            r[4] = (int32_t)h[k]->cigar_count;
            for (uint32_t c = 0; c < h[k]->cigar_count; ++c) cigars[cigar_at[k] + c] = h[k]->cigar[c];
        }
        free(h[k]);
    }
    return rc;
}

// Test shim for the two-halves form: alignments k in [batch_start[g], batch_start[g+1]) form request g; `depth` requests are in
// flight at a time.  Alignment k is ref_bytes[ref_off[k] .. +ref_len[k]) against seq_bytes[seq_off[k] .. +seq_len[k]).  Results: result[k] =
// {score, max_i, max_j, alignment_offset, too_long, cigar_count}; the CIGAR of alignment k is written at cigars[cigar_at[k] ..).
// Returns 0, or -1 on a device error.
int rovaca_amd_sw_batches_in_flight(const uint8_t *ref_bytes, const uint32_t *ref_off, const uint32_t *ref_len, const uint8_t *seq_bytes,
                                    const uint32_t *seq_off, const uint32_t *seq_len, const uint32_t *batch_start, uint32_t n_batches,
                                    int is_assemble, uint32_t depth, int32_t *result, uint32_t *cigars, const uint32_t *cigar_at)
{
    using namespace rovaca;
    const sw_parameters para = is_assemble ? sw_parameters{200, -150, -260, -11} : sw_parameters{10, -15, -30, -5};
    if (depth == 0) depth = 1;
    struct InFlight
    {
        uint32_t batch = 0;
        SwTicket ticket;
    };
    int rc = 0;
    auto collect = [&](InFlight &f) {
        std::vector<SwAlignment> aln;
        if (!sw_hip_align_batch_finish(std::move(f.ticket), aln)) rc = -1;
        const uint32_t k0 = batch_start[f.batch];
        for (size_t a = 0; a < aln.size(); ++a) {
            int32_t *r = result + 6 * (size_t)(k0 + a);
            r[0] = aln[a].score;
            r[1] = aln[a].max_i;
            r[2] = aln[a].max_j;
            r[3] = aln[a].alignment_offset;
            r[4] = aln[a].too_long ? 1 : 0;
            r[5] = (int32_t)aln[a].cigar.size();
            for (size_t c = 0; c < aln[a].cigar.size(); ++c) cigars[cigar_at[k0 + a] + c] = aln[a].cigar[c];
        }
    };
    std::deque<InFlight> q;
    for (uint32_t g = 0; g < n_batches; ++g) {
        std::vector<SwJob> jobs;
        for (uint32_t k = batch_start[g]; k < batch_start[g + 1]; ++k)
            jobs.push_back(SwJob{ref_bytes + ref_off[k], (int32_t)ref_len[k], seq_bytes + seq_off[k], (int32_t)seq_len[k]});
        InFlight f;
        f.batch = g;
        f.ticket = sw_hip_align_batch_begin(jobs.data(), jobs.size(), para);  // the job records are copied: `jobs` may go
        q.push_back(std::move(f));
        if (q.size() >= depth) {
            collect(q.front());
            q.pop_front();
        }
    }
    while (!q.empty()) {
        collect(q.front());
        q.pop_front();
    }
    return rc;
}

}  // extern "C"
