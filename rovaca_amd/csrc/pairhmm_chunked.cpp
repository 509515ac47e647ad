// pairhmm_chunked.cpp -- the one-shot call (phmm_compute: host buffers in, host buffers out) on a BIG batch: the pair list is cut into
// chunks that go through three resident slots as a two-thread pipeline, so that binning, packing and upload of one chunk, the sweep of
// the one before and the host conversion of the one before that overlap.  SURVEY.md 8(d)'s PCIe-inclusive metric is this path.
#include <atomic>
#include <condition_variable>
#include <mutex>

#include "pairhmm_engine.h"

using phmm_detail::g_err;

namespace impl
{

// ---- the chunked one-shot call ----

constexpr uint64_t kChunkTargetPairs = 20000;  // pairs of the first chunk, roughly
constexpr unsigned kMaxChunks = 12;

// Cuts the pair list into contiguous chunks and finds each chunk's table windows.  Returns false when chunking would not
// pay: a small batch, or a pair list that jumps around the tables so that the windows overlap heavily (each chunk would
// upload most of the tables again).
bool plan_chunks(const phmm_batch *b, std::vector<ChunkWindow> &plan)
{
    const bool cross = b->pair_read == nullptr;
    const uint64_t n = cross ? (uint64_t)b->n_reads * b->n_haps : b->n_pairs;
    if (n < chunked_min_pairs() || b->n_reads == 0 || b->n_haps == 0) return false;
    if (const char *e = phmm_hook("PHMM_NO_CHUNKED_COMPUTE")) {  // tests
        if (e[0] != '0') return false;
    }
    // Chunk sizes rise and fall: the first chunk's packing and the last chunk's conversion are the only host work the device cannot
    // hide, so those are small; in between, packing is ~2.6x faster than sweeping, so a chunk may be up to 2.6x the one before it without
    // the device running dry.  With two sweep lanes (compute_chunked) a chunk boundary costs next to nothing on the device, so the chunks
    // are short: about 20 000 pairs for the first, between four and twelve chunks.  Measured on config 2's pairs, host buffers to log10,
    // against one upload + one sweep: 30 000 pairs 0.57 -> 0.52 ms (1:2:2:1), 100 000 1.75 -> 1.12 (1:2:3:2:1), 300 000 4.71 -> 2.73,
    // 500 000 7.13 -> 4.32, 1 000 000 8.65 (seven chunks, one lane) -> 8.25 (twelve: 1:2:3:4:5:6:6:6:6:5:4:3; ten 8.37, sixteen 8.41).
    // (Round 4, a slot per chunk: fourteen for the biggest batches, below.)
    unsigned C = (unsigned)std::min<uint64_t>(kMaxChunks, std::max<uint64_t>(4, n / kChunkTargetPairs));
    std::vector<unsigned> weight(C);
    {
        const unsigned P = (C + 1) / 2;  // plateau
        for (unsigned k = 0; k < C; ++k) weight[k] = std::min(std::min(k + 1, P), (C - k) + P / 2 - 1);
    }
    if (C == kMaxChunks && (!cross || b->n_reads >= 4096)) {  // (a cross product is cut by reads: with few of them the smallest chunk would be empty)
        // the biggest batches begin with smaller steps still (5 000 pairs, then 2 : 4 : 8): the first sweep starts 0.15 ms into the call
        // instead of 0.35, and with a slot per chunk the extra chunks cost nothing (round 4, configs 2 and 3: 0.1 ms per call, 1 %)
        static const unsigned graded14[] = {1, 2, 4, 8, 12, 16, 20, 24, 24, 24, 24, 20, 16, 12};
        C = 14;
        weight.assign(graded14, graded14 + 14);
    }
    if (phmm_hook("PHMM_CHUNK_ONE_LANE") && C >= 7) {  // the plan of the one-lane form (A/B)
        static const unsigned graded7[] = {2, 6, 14, 26, 26, 18, 8};
        C = 7;
        weight.assign(graded7, graded7 + 7);
    }
    if (const char *e = phmm_hook("PHMM_CHUNK_WEIGHTS")) {  // tuning aid: relative chunk sizes, e.g. "1,2,3,2,1"
        std::vector<unsigned> w;
        for (const char *q = e; *q;) {
            char *end = nullptr;
            const unsigned long v = strtoul(q, &end, 10);
            if (end == q) break;
            if (v) w.push_back((unsigned)std::min<unsigned long>(v, 1000));
            q = *end ? end + 1 : end;
        }
        if (w.size() >= 2 && w.size() <= 16) {
            weight = w;
            C = (unsigned)w.size();
        }
    }
    std::vector<uint64_t> cut(C + 1, 0);
    {
        uint64_t wsum = 0, run = 0;
        for (unsigned k = 0; k < C; ++k) wsum += weight[k];
        for (unsigned k = 0; k < C; ++k) {
            run += weight[k];
            cut[k + 1] = cross ? (uint64_t)b->n_reads * run / wsum : n * run / wsum;
        }
    }
    plan.assign(C, ChunkWindow());
    std::atomic<bool> bad{false};
    parallel_ranges(C, std::min(C, host_threads()), [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t k = lo; k < hi; ++k) {
            ChunkWindow &w = plan[k];
            if (cross) {
                w.r0 = (uint32_t)cut[k];
                w.r1 = (uint32_t)cut[k + 1];
                w.p0 = (uint64_t)w.r0 * b->n_haps;
                w.p1 = (uint64_t)w.r1 * b->n_haps;
                w.h0 = 0;
                w.h1 = b->n_haps;
            }
            else {
                w.p0 = cut[k];
                w.p1 = cut[k + 1];
                uint32_t rmin = UINT32_MAX, rmax = 0, hmin = UINT32_MAX, hmax = 0;
                for (uint64_t p = w.p0; p < w.p1; ++p) {
                    const uint32_t r = b->pair_read[p], h = b->pair_hap[p];
                    rmin = std::min(rmin, r);
                    rmax = std::max(rmax, r);
                    hmin = std::min(hmin, h);
                    hmax = std::max(hmax, h);
                }
                if (w.p1 == w.p0 || rmax >= b->n_reads || hmax >= b->n_haps) {
                    bad.store(true);  // empty chunk or an index out of range: let the plain path report it
                    return;
                }
                w.r0 = rmin;
                w.r1 = rmax + 1;
                w.h0 = hmin;
                w.h1 = hmax + 1;
            }
            uint64_t lo_b = UINT64_MAX, hi_b = 0;
            for (uint32_t r = w.r0; r < w.r1; ++r) {
                lo_b = std::min<uint64_t>(lo_b, b->read_off[r]);
                hi_b = std::max<uint64_t>(hi_b, (uint64_t)b->read_off[r] + b->read_len[r]);
            }
            w.rb0 = w.r1 > w.r0 ? lo_b : 0;
            w.rb1 = w.r1 > w.r0 ? hi_b : 0;
            lo_b = UINT64_MAX;
            hi_b = 0;
            for (uint32_t h = w.h0; h < w.h1; ++h) {
                lo_b = std::min<uint64_t>(lo_b, b->hap_off[h]);
                hi_b = std::max<uint64_t>(hi_b, (uint64_t)b->hap_off[h] + b->hap_len[h]);
            }
            w.hb0 = w.h1 > w.h0 ? lo_b : 0;
            w.hb1 = w.h1 > w.h0 ? hi_b : 0;
        }
    });
    if (bad.load()) return false;
    uint64_t win_bytes = 0, all_lo = UINT64_MAX, all_hi = 0, hall_lo = UINT64_MAX, hall_hi = 0;
    for (const ChunkWindow &w : plan) {
        if (w.p1 == w.p0) return false;
        win_bytes += 5 * (w.rb1 - w.rb0) + (w.hb1 - w.hb0);
        all_lo = std::min(all_lo, w.rb0);
        all_hi = std::max(all_hi, w.rb1);
        hall_lo = std::min(hall_lo, w.hb0);
        hall_hi = std::max(hall_hi, w.hb1);
    }
    const uint64_t once = 5 * (all_hi - all_lo) + (hall_hi - hall_lo);
    return win_bytes <= once + once / 2 + (1u << 20);
}

// The chunked one-shot call as a two-thread pipeline.  The calling thread bins, packs and enqueues chunk after chunk without ever
// waiting for the device (upload + set-up on the upload stream, the sweep behind an event on the context's stream, the result
// copy behind another event on the download stream, so that the next sweep starts the moment this one ends); a collector thread
// takes the chunks in order as their results arrive, fetches the fp64 sums where anything was rescued, and does the host log10.
// Three resident slots: chunk k is packed while k - 1 is swept and k - 2 is being converted.  The device queue stays non-empty
// from the first chunk's sweep to the last one's -- round 2 packed chunk k + 1 only after chunk k - 1 had been collected, and the
// sweeps of one call took 10.0 ms for 8.3 ms of work because the clock ramped down in the gaps.
int compute_chunked(phmm_context *c, const phmm_batch *b, const std::vector<ChunkWindow> &plan, float *out_raw_f32,
                           double *out_log10, phmm_stats *stats)
{
    HIP_TRY(hipSetDevice(c->device));
    {
        const int rc_side = ensure_side_streams(c);  // the upload stream is one of them
        if (rc_side != PHMM_OK) return rc_side;
    }
    constexpr int S = phmm_context::kChunkSlots;
    for (int i = 0; i < S; ++i) {
        if (!c->chunk_slot[i]) {
            c->chunk_slot[i] = new (std::nothrow) phmm_resident();
            if (!c->chunk_slot[i]) return fail(PHMM_E_NOMEM, "phmm_compute: out of host memory");
            c->chunk_slot[i]->fp32_first_only = true;  // (the collector decides pair by pair on the fp32 sums it sees: pairhmm_sweep.cpp, "fp64 first")
        }
        if (!c->chunk_up[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_up[i], hipEventDisableTiming));
        if (!c->chunk_swept[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_swept[i], hipEventDisableTiming));
        if (!c->chunk_done[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_done[i], hipEventDisableTiming));
    }
    if (!c->down) HIP_TRY(hipStreamCreateWithFlags(&c->down, hipStreamNonBlocking));
    // Two sweep lanes: a sweep's last workgroups leave most of the chip idle for a wave's duration ("How long a launch is"),
    // and the next chunk's sweep, which depends on nothing but its own upload, can fill it -- if it sits in another hardware queue.
    // Even chunks sweep on the context's streams, odd chunks on a second set.  Measured on config 2 with 16 hardware queues: device span
    // of a call 8.17 -> 7.9 ms (resident pass: 7.65), and shorter chunks stop costing (twelve instead of seven: 8.65 -> 8.35 ms per call).
    // PHMM_CHUNK_ONE_LANE=1 keeps everything on the context's streams.
    const bool two_lanes = phmm_hook("PHMM_CHUNK_ONE_LANE") == nullptr;
    if (two_lanes && !c->lane2.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->lane2.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->lane2.fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->lane2.done, hipEventDisableTiming));
        for (int i = 0; i < c->n_side; ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&c->lane2.side[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->lane2.join[i], hipEventDisableTiming));
        }
    }
    // The fp64 re-run of a chunk is NOT part of its device chain: the collector thread sees the chunk's fp32 sums on the host, and only
    // when one of them underflowed does it launch that chunk's re-run, on a lane of its own (below).  Enqueued behind every sweep the
    // re-run launches -- four per chunk of mixed lengths, nearly always empty -- each had to push thousands of workgroups with a 256-register
    // budget through a chip full of the other lane's sweep, 50-700 us during which the lane could not start its next chunk: consecutive
    // chunks stopped overlapping and every one paid its own ramp and tail (rocprofv3 time line of config 3, round 4: 10.1 ms of device
    // time for 7.9 ms of sweeps).
    if (!c->rescue_lane.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->rescue_lane.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->rescue_lane.fork, hipEventDisableTiming));
        for (int i = 0; i < c->n_side; ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&c->rescue_lane.side[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->rescue_lane.join[i], hipEventDisableTiming));
        }
    }
    hipStream_t up = c->side[phmm_context::kSideStreams - 1];  // not used by the sweeps (n_side defaults to 2 of 3)
    if (c->n_side >= phmm_context::kSideStreams) up = c->stream;
    hipStream_t down = c->down;
    HIP_TRY(hipStreamSynchronize(c->stream));  // nothing of an earlier call still reads the slots
    if (up != c->stream) HIP_TRY(hipStreamSynchronize(up));
    HIP_TRY(hipStreamSynchronize(down));
    if (c->lane2.stream) HIP_TRY(hipStreamSynchronize(c->lane2.stream));
    const bool cross = b->pair_read == nullptr;
    const size_t n_chunks = plan.size();
    const bool trace = getenv("PHMM_PHASE_TIMING") != nullptr;  // tuning aid: host time line of the pipeline on stderr
    const bool direct = phmm_hook("PHMM_CHUNK_RESULTS_BY_COPY") == nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(); };

    struct InFlight
    {
        const ChunkWindow *w = nullptr;
        size_t count_at = 0;
    } fl[S];
    // hand-off between the two threads: chunks [0, enqueued) are on the device, [0, collected) are finished
    std::mutex mu;
    std::condition_variable cv;
    size_t enqueued = 0, collected = 0;
    bool producer_failed = false;
    int collector_rc = PHMM_OK;
    std::string collector_msg;
    phmm_stats total{};

    auto collect = [&](size_t k) -> int {
        const int slot = (int)(k % S);
        phmm_resident *rb = c->chunk_slot[slot];
        const ChunkWindow &w = *fl[slot].w;
        const uint64_t n = rb->n_pairs;
        HIP_TRY(hipEventSynchronize(c->chunk_done[slot]));
        const double t_ready = trace ? now_us() : 0.0;
        const size_t off_cnt = fl[slot].count_at * sizeof(float), off64 = off_cnt + 256;
        const uint8_t *st = static_cast<const uint8_t *>(rb->out_stage.p);
        const float *rp = reinterpret_cast<const float *>(st);
        const double *r64 = reinterpret_cast<const double *>(st + off64);
        if (out_raw_f32) memcpy(out_raw_f32 + w.p0, rp, n * sizeof(float));
        // pass 1: the accepted pairs' log10 (pairhmm_internal.cpp:577-585), and where the underflowed ones are
        std::atomic<uint64_t> n_resc{0};
        std::mutex resc_mu;
        std::vector<uint32_t> rescued;
        double *dst = out_log10 ? out_log10 + w.p0 : nullptr;
        {
            auto finalize = [&](uint64_t lo, uint64_t hi, unsigned) {
                uint32_t mine[64];
                uint32_t m = 0;
                uint64_t resc = 0;
                auto flush = [&] {
                    std::lock_guard<std::mutex> lk(resc_mu);
                    rescued.insert(rescued.end(), mine, mine + m);
                    m = 0;
                };
                for (uint64_t p = lo; p < hi; ++p) {
                    const float x = rp[p];
                    if (x < PHMM_MIN_ACCEPTED_F32) {
                        ++resc;
                        mine[m++] = (uint32_t)p;
                        if (m == 64) flush();
                    }
                    else if (dst) {
                        dst[p] = (double)(log10f(x) - PHMM_LOG10_INITIAL_CONSTANT_F32);
                    }
                }
                if (m) flush();
                n_resc.fetch_add(resc, std::memory_order_relaxed);
            };
            if (n >= (1u << 14))
                parallel_ranges(n, (c->fin_pool ? c->fin_pool->helpers() : 0u) + 1u, finalize);
            else
                finalize(0, n, 0);
        }
        // pass 2, rare: the fp64 re-run of this chunk's queues (filled by its fp32 kernels, still in the slot's device memory), on the
        // rescue lane, then the rescued pairs' log10
        if (!rescued.empty() && dst) {
            int rc = impl::sweep_rescue(c, rb, &c->rescue_lane);
            if (rc != PHMM_OK) return rc;
            if (!direct)
                HIP_TRY(hipMemcpyAsync(const_cast<uint8_t *>(st) + off64, rb->raw64.p, n * sizeof(double), hipMemcpyDeviceToHost, c->rescue_lane.stream));
            HIP_TRY(hipStreamSynchronize(c->rescue_lane.stream));
            for (uint32_t p : rescued) dst[p] = log10(r64[p]) - PHMM_LOG10_INITIAL_CONSTANT_F64;
        }
        rb->last_rescued = (uint32_t)n_resc.load();
        total.cells += rb->cells;
        total.n_rescued += rb->last_rescued;
        total.n_launches += rb->last_fp32_launches ? rb->last_fp32_launches : (uint32_t)rb->bins.size();
        if (!rb->bins.empty()) total.max_rows_per_lane = std::max<uint32_t>(total.max_rows_per_lane, (uint32_t)rb->bins.back().rows_per_lane);
        if (trace) fprintf(stderr, "phmm chunk %zu: results on the host at %.0f us, converted at %.0f us\n", k, t_ready, now_us());
        return PHMM_OK;
    };

    // (a process at its thread limit -- 128 workers each with a context of their own -- cannot start the collector: nothing has been
    // enqueued yet, the caller takes the one-upload path instead)
    std::thread collector;
    try {
        collector = std::thread([&] {
        (void)hipSetDevice(c->device);
        phmm_detail::HookScope hooks;
        phmm_detail::PoolScope pool(c->fin_pool);
        for (size_t k = 0; k < n_chunks; ++k) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return enqueued > k || producer_failed; });
                if (enqueued <= k) return;  // the producer gave up before this chunk
            }
            int rc = PHMM_OK;
            try {
                rc = collect(k);
            }
            catch (...) {
                rc = fail(PHMM_E_NOMEM, "phmm_compute: the collector thread ran out of memory");
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc != PHMM_OK && collector_rc == PHMM_OK) {
                    collector_rc = rc;
                    collector_msg = g_err;  // (thread-local: carry the text over)
                }
                collected = k + 1;
            }
            cv.notify_all();
        }
        });
    }
    catch (const std::system_error &) {
        return kChunkedUnavailable;
    }

    struct Joiner  // whichever way this function returns, the collector is told and joined first
    {
        std::thread &t;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &failed;
        bool ok = false;
        ~Joiner()
        {
            if (!ok) {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    failed = true;
                }
                cv.notify_all();
            }
            t.join();
        }
    } joiner{collector, mu, cv, producer_failed};

    // ---- the producer side: prepare workers + the calling thread as enqueuer ----
    // Packing and binning a chunk is host work only (upload_prepare) and takes about as long as the device needs to sweep it: 0.65-0.9 ms
    // for 118 000 pairs of mixed lengths against 0.93 ms (round 4's time line of config 3: the producer finished its last chunk at
    // 10-11 ms of a pass that takes the device 7.9 ms).  So the chunks are prepared by a few worker threads, several at a time and each
    // on one thread (no nested helpers), into the slots' pinned staging buffers; the calling thread only enqueues them, in order -- the
    // copy, the set-up launch, the sweeps: 0.1 ms per chunk -- and is never the bottleneck.  PHMM_CHUNK_PREP_THREADS=0: the calling
    // thread prepares every chunk itself (the form until round 3; also what a process out of threads gets).
    unsigned n_prep = 3;
    if (const char *e = phmm_hook("PHMM_CHUNK_PREP_THREADS")) n_prep = (unsigned)std::max(0, std::min(8, atoi(e)));
    if (n_prep && (!c->prep_pool || c->prep_pool->helpers() == 0)) {
        delete c->prep_pool;
        c->prep_pool = new phmm_detail::HostPool(n_prep);
        if (c->prep_pool->helpers() == 0) n_prep = 0;
    }
    struct ChunkInput  // the chunk as a batch of its own; the vectors live as long as the slot holds the chunk
    {
        std::vector<uint32_t> roff, hoff;
        phmm_read_prep prep_k{};
        phmm_batch s{};
        UploadOpts opt;
        PhaseClock clock;
        double t_a = 0.0, t_b = 0.0;
    };
    std::vector<ChunkInput> input(S);
    std::vector<int> prepared(n_chunks, -1);  // -1: not yet; else the status of upload_prepare (guarded by mu)
    std::vector<std::string> prepared_msg(n_chunks);
    const bool tracing = trace;
    auto prepare = [&](size_t k, phmm_detail::BinScratch &scratch) -> int {
        const ChunkWindow &w = plan[k];
        const int slot = (int)(k % S);
        ChunkInput &in = input[slot];
        phmm_resident *rb = c->chunk_slot[slot];
        in.t_a = tracing ? now_us() : 0.0;
        // the chunk as a batch of its own: windows of the tables, offsets rebased to the windows, pair indices biased
        phmm_batch &s = in.s;
        s = *b;
        s.n_reads = w.r1 - w.r0;
        s.n_haps = w.h1 - w.h0;
        s.n_pairs = w.p1 - w.p0;
        in.roff.resize(s.n_reads);
        for (uint32_t r = 0; r < s.n_reads; ++r) in.roff[r] = (uint32_t)(b->read_off[w.r0 + r] - w.rb0);
        in.hoff.resize(s.n_haps);
        for (uint32_t h = 0; h < s.n_haps; ++h) in.hoff[h] = (uint32_t)(b->hap_off[w.h0 + h] - w.hb0);
        s.read_bases = b->read_bases + w.rb0;
        s.read_qual = b->read_qual + w.rb0;
        s.read_ins = b->read_ins ? b->read_ins + w.rb0 : nullptr;
        s.read_del = b->read_del ? b->read_del + w.rb0 : nullptr;
        s.read_gcp = b->read_gcp ? b->read_gcp + w.rb0 : nullptr;
        s.read_off = in.roff.data();
        s.read_len = b->read_len + w.r0;
        s.hap_bases = b->hap_bases + w.hb0;
        s.hap_off = in.hoff.data();
        s.hap_len = b->hap_len + w.h0;
        s.pair_read = cross ? nullptr : b->pair_read + w.p0;
        s.pair_hap = cross ? nullptr : b->pair_hap + w.p0;
        if (b->prep) {
            in.prep_k = *b->prep;
            in.prep_k.read_mapq = b->prep->read_mapq ? b->prep->read_mapq + w.r0 : nullptr;
            s.prep = &in.prep_k;
        }
        in.opt = UploadOpts();
        in.opt.stream = up;
        in.opt.caller_synchronised = true;  // the slot held chunk k - S: collected, so its sweep is long over
        in.opt.read_bias = cross ? 0 : w.r0;
        in.opt.hap_bias = cross ? 0 : w.h0;
        in.clock = PhaseClock();
        struct PhaseScope
        {
            explicit PhaseScope(PhaseClock *p) { g_phase = p; }
            ~PhaseScope() { g_phase = nullptr; }
        } phase_scope(tracing ? &in.clock : nullptr);
        const int rc = upload_prepare(c, &s, rb, in.opt, scratch);
        in.t_b = tracing ? now_us() : 0.0;
        return rc;
    };
    std::atomic<bool> scratch_busy[8];
    for (auto &f : scratch_busy) f.store(false);
    auto slot_is_free = [&](size_t k) { return collected + S > k || producer_failed || collector_rc != PHMM_OK; };
    if (n_prep) {
        c->prep_pool->set_hot(true);
        c->prep_pool->begin((unsigned)n_chunks, [&, n_prep](unsigned k) {
            (void)n_prep;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return slot_is_free(k); });
                if (producer_failed || collector_rc != PHMM_OK) {
                    prepared[k] = PHMM_E_INVALID;
                    prepared_msg[k] = "phmm_compute: the chunk pipeline stopped early";
                    lk.unlock();
                    cv.notify_all();
                    return;
                }
            }
            phmm_detail::HookScope hooks;
            phmm_detail::t_serial = true;
            int rc = PHMM_OK;
            try {
                // a binning scratch of the context for the duration of this chunk (at most 8 workers: there is always a free one)
                unsigned mine = 0;
                for (;; mine = (mine + 1) % 8u) {
                    bool expected = false;
                    if (scratch_busy[mine].compare_exchange_strong(expected, true)) break;
                }
                struct Release
                {
                    std::atomic<bool> &f;
                    ~Release() { f.store(false); }
                } release{scratch_busy[mine]};
                rc = prepare(k, c->prep_scratch[mine]);
            }
            catch (...) {
                rc = fail(PHMM_E_NOMEM, "phmm_compute: a prepare worker ran out of memory");
            }
            phmm_detail::t_serial = false;
            {
                std::lock_guard<std::mutex> lk(mu);
                prepared[k] = rc;
                if (rc != PHMM_OK) prepared_msg[k] = g_err;  // (thread-local: carry the text over)
            }
            cv.notify_all();
        });
    }
    struct PrepJoiner  // whichever way this function returns, the prepare workers are through with the job first
    {
        phmm_detail::HostPool *pool;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &failed;
        bool ok = false;
        ~PrepJoiner()
        {
            if (!pool) return;
            if (!ok) {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    failed = true;
                }
                cv.notify_all();
            }
            pool->end();
            pool->set_hot(false);
        }
    } prep_joiner{n_prep ? c->prep_pool : nullptr, mu, cv, producer_failed};

    phmm_detail::PoolScope pool(c->pack_pool);
    if (c->fin_pool) c->fin_pool->set_hot(true);
    struct CoolDown
    {
        phmm_detail::HostPool *p;
        ~CoolDown() { if (p) p->set_hot(false); }
    } cool_down{c->fin_pool};
    for (size_t k = 0; k < n_chunks; ++k) {
        const ChunkWindow &w = plan[k];
        const int slot = (int)(k % S);
        phmm_resident *rb = c->chunk_slot[slot];
        ChunkInput &in = input[slot];
        if (n_prep) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return prepared[k] >= 0; });
            if (prepared[k] != PHMM_OK) return fail(prepared[k], "%s", prepared_msg[k].c_str());
            if (collector_rc != PHMM_OK) break;
        }
        else {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return slot_is_free(k); });
                if (collector_rc != PHMM_OK) break;
            }
            const int rc = prepare(k, c->bin_scratch);
            if (rc != PHMM_OK) return rc;
        }
        const double t_e = trace ? now_us() : 0.0;
        UploadOpts opt = in.opt;
        // The copy travels on the upload stream (DMA engines: it runs beside the previous chunk's sweep); the set-up launch goes on
        // the context's stream, in front of this chunk's sweep.  On the upload stream it would not start before the previous sweep
        // has handed out its last workgroup -- small kernels of another queue are starved by a grid that fills the chip -- and the
        // sweep behind it would wait with it: 90-110 us of a drained chip at every chunk boundary (rocprofv3 time line, round 3).
        opt.setup_stream = c->stream;
        // Built for 16 hardware queues (GPU_MAX_HW_QUEUES, see phmm_create; the runtime's default of 4 made the upload stream share a queue
        // with the sweeps, which is what the two findings above were): copy and set-up launch both on the upload stream, the sweep behind
        // an event, and the packing thread never waits for a copy.  8.62-8.68 ms per call on config 2 against 8.74-8.75 with the
        // synchronous copy + set-up on the sweep stream (PHMM_CHUNK_SYNC_UPLOAD=1 keeps that form; with 4 queues the two are equal).
        const bool async_upload = phmm_hook("PHMM_CHUNK_SYNC_UPLOAD") == nullptr && up != c->stream;
        if (async_upload) opt.setup_stream = nullptr;
        if (k == 0) HIP_TRY(hipEventRecord(c->ev0, c->stream));
        PhaseClock chunk_clock;
        struct PhaseScope
        {
            explicit PhaseScope(PhaseClock *p) { g_phase = p; }
            ~PhaseScope() { g_phase = nullptr; }
        } phase_scope(trace ? &chunk_clock : nullptr);
        // odd chunks: the second lane's stream set stands in for the context's while this chunk is enqueued (the upload stream is side[2]
        // of the context either way: `up` was read before)
        struct LaneSwap
        {
            phmm_context *c;
            bool on;
            void swap_sets()
            {
                std::swap(c->stream, c->lane2.stream);
                std::swap(c->fork, c->lane2.fork);
                for (int i = 0; i < c->n_side && i < phmm_context::kSideStreams - 1; ++i) {
                    std::swap(c->side[i], c->lane2.side[i]);
                    std::swap(c->join[i], c->lane2.join[i]);
                }
            }
            LaneSwap(phmm_context *c_, bool on_) : c(c_), on(on_) { if (on) swap_sets(); }
            ~LaneSwap() { if (on) swap_sets(); }
        } lane_swap(c, two_lanes && async_upload && (k & 1));
        int rc = upload_enqueue(c, rb, opt);
        if (rc != PHMM_OK) return rc;
        PHASE(3);
        // where the sums go: straight into the slot's pinned result block (the kernels write host memory over PCIe: no copy, no
        // cross-stream event between two sweeps), or -- PHMM_CHUNK_RESULTS_BY_COPY=1 -- into HBM and from there by a copy
        const uint64_t n = rb->n_pairs;
        const size_t count_at = (n + 63) & ~size_t(63);
        HIP_TRY(rb->out_stage.ensure(count_at * sizeof(float) + 256 + n * sizeof(double)));
        rb->out32_host = direct ? static_cast<float *>(rb->out_stage.p) : nullptr;
        rb->out64_host = direct ? reinterpret_cast<double *>(static_cast<uint8_t *>(rb->out_stage.p) + count_at * sizeof(float) + 256) : nullptr;
        if (async_upload) {
            HIP_TRY(hipEventRecord(c->chunk_up[slot], up));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->chunk_up[slot], 0));
        }
        rc = impl::sweep_fp32(c, rb);  // (the fp64 re-run: the collector's business, see collect)
        if (rc != PHMM_OK) return rc;
        PHASE(4);
        if (direct) {
            HIP_TRY(hipEventRecord(c->chunk_done[slot], c->stream));
        }
        else {
            // the sums travel on the download stream: the next chunk's sweep does not queue behind the copy
            HIP_TRY(hipEventRecord(c->chunk_swept[slot], c->stream));
            HIP_TRY(hipStreamWaitEvent(down, c->chunk_swept[slot], 0));
            HIP_TRY(hipMemcpyAsync(rb->out_stage.p, rb->raw32.p, count_at * sizeof(float) + kQueues * sizeof(uint32_t), hipMemcpyDeviceToHost, down));
            HIP_TRY(hipEventRecord(c->chunk_done[slot], down));
        }
        fl[slot].w = &w;
        fl[slot].count_at = count_at;
        {
            std::lock_guard<std::mutex> lk(mu);
            enqueued = k + 1;
        }
        cv.notify_all();
        if (trace) {
            PHASE(5);
            fprintf(stderr, "phmm chunk %zu: %llu pairs, prepared %.0f..%.0f us (validate %.0f pack %.0f bin %.0f), enqueued %.0f..%.0f us (copy+setup %.0f sweep %.0f results %.0f)\n",
                    k, (unsigned long long)(w.p1 - w.p0), in.t_a, in.t_b, in.clock.us[0], in.clock.us[1], in.clock.us[2], t_e, now_us(),
                    chunk_clock.us[3], chunk_clock.us[4], chunk_clock.us[5]);
        }
    }
    if (two_lanes && c->lane2.stream) {  // the device span of the call ends when both lanes are through
        HIP_TRY(hipEventRecord(c->lane2.done, c->lane2.stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->lane2.done, 0));
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return collected >= enqueued || collector_rc != PHMM_OK; });
    }
    joiner.ok = enqueued == n_chunks;
    prep_joiner.ok = joiner.ok;
    if (collector_rc != PHMM_OK) return fail(collector_rc, "%s", collector_msg.c_str());
    if (enqueued != n_chunks) return fail(PHMM_E_INVALID, "phmm_compute: the chunk pipeline stopped early");
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(c->ev1));  // recorded behind the join of the two lanes: the last chunk's results can be in before it
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        total.kernel_ms = ms;
        *stats = total;
    }
    return PHMM_OK;
}

}  // namespace impl
