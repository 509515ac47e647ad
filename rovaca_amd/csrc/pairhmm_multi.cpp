// pairhmm_multi.cpp -- several devices behind one handle (include/pairhmm_hip.h: phmm_create_multi and friends).
#include <atomic>
#include <cmath>

#include "pairhmm_engine.h"

using phmm_detail::g_err;

// ----------------------------------------------------------------------------------------------------
// Several devices behind one handle (SURVEY.md 8b "create(device_ids[], n)", 8e): pairs are independent, so a batch is cut
// into one slice per device -- contiguous ranges of the length-binned order (rows per lane, haplotype length, pair index),
// equal in DP cells, not in pairs -- and every device works its slice on its own host thread with its own context (stream
// set, staging, tables).  There is no exchange step and no collective; results land at the pairs' own positions of the
// caller's arrays, which are disjoint between devices.
// ----------------------------------------------------------------------------------------------------
struct phmm_multi
{
    std::vector<phmm_context *> ctx;
};

struct phmm_multi_resident
{
    uint64_t n_pairs = 0;
    std::vector<phmm_resident *> part;            // one per device (never null; a slice may be empty)
    std::vector<std::vector<uint32_t>> index;     // pair ids of each slice, in slice order
    std::vector<std::vector<uint32_t>> pr, ph;    // the slices' pair lists (kept: phmm_upload_into copies them, but cheap)
    std::vector<uint64_t> cells;
};

namespace impl
{

// run fn(device index) on one thread per device; the first failure's status and text are handed to the calling thread
// `degraded` (nullable) is set before any un-started body runs on the caller when a helper thread could not be created: bodies
// that meet at a gate must then stop waiting for each other (they run one after another)
template <typename F>
static int on_every_device(phmm_multi *m, F &&fn, std::atomic<bool> *degraded = nullptr)
{
    const size_t n = m->ctx.size();
    std::vector<int> rc(n, PHMM_OK);
    std::vector<std::string> msg(n);
    auto body = [&](size_t g) {
        try {
            rc[g] = fn(g);
        }
        catch (const std::bad_alloc &) {
            rc[g] = fail(PHMM_E_NOMEM, "out of host memory on the thread of device %d", m->ctx[g]->device);
        }
        catch (...) {
            rc[g] = fail(PHMM_E_INVALID, "unexpected exception on the thread of device %d", m->ctx[g]->device);
        }
        if (rc[g] != PHMM_OK) msg[g] = g_err;  // g_err is thread-local: copy it out of the worker
    };
    std::vector<std::thread> th;
    th.reserve(n);
    size_t started = 1;
    try {
        for (; started < n; ++started) th.emplace_back(body, started);
    }
    catch (...) {
    }
    if (started < n && degraded) degraded->store(true, std::memory_order_release);
    body(0);
    for (size_t g = started; g < n; ++g) body(g);
    for (auto &t : th) t.join();
    for (size_t g = 0; g < n; ++g)
        if (rc[g] != PHMM_OK) return fail(rc[g], "device %d: %s", m->ctx[g]->device, msg[g].c_str());
    return PHMM_OK;
}

void phmm_destroy_multi(phmm_multi *m)
{
    if (!m) return;
    for (phmm_context *c : m->ctx) impl::phmm_destroy(c);
    delete m;
}

int phmm_create_multi(const int *device_ids, int n_devices, phmm_multi **out)
{
    if (!out || !device_ids || n_devices < 1) return fail(PHMM_E_INVALID, "phmm_create_multi: bad arguments");
    *out = nullptr;
    // PHMM_MULTI_ALLOW_SHARED_DEVICE=1 (tests, and bench.py --oversubscribe): a device may be listed several times -- every
    // entry still gets its own context, host thread and slice, so the whole N-slice path runs on a box with fewer than N GPUs
    const bool shared_ok = phmm_hook("PHMM_MULTI_ALLOW_SHARED_DEVICE") != nullptr;  // (dead without PHMM_ENABLE_TEST_HOOKS, like every hook)
    for (int i = 0; i < n_devices && !shared_ok; ++i)
        for (int j = 0; j < i; ++j)
            if (device_ids[i] == device_ids[j])
                return fail(PHMM_E_INVALID, "phmm_create_multi: device %d is listed twice (one context per device)", device_ids[i]);
    phmm_multi *m = new phmm_multi();
    for (int i = 0; i < n_devices; ++i) {
        phmm_context *c = nullptr;
        const int rc = impl::phmm_create(device_ids[i], nullptr, &c);
        if (rc != PHMM_OK) {
            impl::phmm_destroy_multi(m);
            return rc;
        }
        m->ctx.push_back(c);
    }
    *out = m;
    return PHMM_OK;
}

// the same handle over CPU contexts (phmm_create_cpu): the cut, the per-slice threads, the scatter of the results and the timed
// gate are the code the device slices run, so the N > 1 path is exercised on a machine without GPUs
int phmm_create_multi_cpu(int n_slices, int n_threads_each, int allow_simd, phmm_multi **out)
{
    if (!out || n_slices < 1 || n_threads_each < 0) return fail(PHMM_E_INVALID, "phmm_create_multi_cpu: bad arguments");
    *out = nullptr;
    phmm_multi *m = new phmm_multi();
    for (int i = 0; i < n_slices; ++i) {
        phmm_context *c = nullptr;
        const int rc = impl::phmm_create_cpu(n_threads_each, allow_simd, &c);
        if (rc != PHMM_OK) {
            impl::phmm_destroy_multi(m);
            return rc;
        }
        c->device = -1 - i;  // (only a label: phmm_multi_device_id reports -1, -2, ... for CPU slices)
        m->ctx.push_back(c);
    }
    *out = m;
    return PHMM_OK;
}

void phmm_multi_release(phmm_multi *m, phmm_multi_resident *rb)
{
    if (!rb) return;
    for (size_t g = 0; g < rb->part.size(); ++g) impl::phmm_release(m && g < m->ctx.size() ? m->ctx[g] : nullptr, rb->part[g]);
    delete rb;
}

// the cut: pairs sorted by (rows per lane = ceil(R / 16), haplotype length, index), then n contiguous slices of near-equal
// cells (each cut within one pair of the ideal)
static int partition_equal_cells(const phmm_batch *b, size_t n_dev, phmm_multi_resident *rb)
{
    if ((b->pair_read == nullptr) != (b->pair_hap == nullptr))
        return fail(PHMM_E_INVALID, "phmm_multi_upload: pair_read and pair_hap must both be given or both be NULL");
    const bool cross = b->pair_read == nullptr;
    const uint64_t n = cross ? (uint64_t)b->n_reads * b->n_haps : b->n_pairs;
    if (n > 0xFFFFFFF0ull) return fail(PHMM_E_INVALID, "phmm_multi_upload: more than 2^32 pairs in one batch");
    if (n && (!b->read_len || !b->hap_len || !b->read_off || !b->hap_off || !b->read_bases || !b->read_qual || !b->hap_bases))
        return fail(PHMM_E_INVALID, "phmm_multi_upload: null table pointer");
    if (n && !b->prep && (!b->read_ins || !b->read_del || !b->read_gcp))
        return fail(PHMM_E_INVALID, "phmm_multi_upload: null gap-penalty array (and no phmm_read_prep given)");
    if (b->prep && (!b->prep->read_mapq || b->prep->pcr_model < 0 || b->prep->pcr_model > 3)) return fail(PHMM_E_INVALID, "phmm_multi_upload: bad phmm_read_prep");
    rb->n_pairs = n;
    std::vector<uint64_t> keys(n);
    std::atomic<uint64_t> bad{UINT64_MAX};
    auto pair_of = [&](uint64_t p, uint32_t &r, uint32_t &h) {
        if (cross) {
            r = (uint32_t)(p / b->n_haps);
            h = (uint32_t)(p % b->n_haps);
        }
        else {
            r = b->pair_read[p];
            h = b->pair_hap[p];
        }
    };
    const unsigned T = n >= kParallelMinPairs ? host_threads() : 1u;
    parallel_ranges(n, T, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t p = lo; p < hi; ++p) {
            uint32_t r, h;
            pair_of(p, r, h);
            if (r >= b->n_reads || h >= b->n_haps) {
                uint64_t expect = UINT64_MAX;
                bad.compare_exchange_strong(expect, p);
                return;
            }
            const uint64_t k = (b->read_len[r] + phmm::kLanesPerPair - 1) / phmm::kLanesPerPair;
            keys[p] = (k << 52) | ((uint64_t)(b->hap_len[h] & 0xFFFFFu) << 32) | p;  // lengths were checked < 2^20 by upload
        }
    });
    if (bad.load() != UINT64_MAX)
        return fail(PHMM_E_INVALID, "phmm_multi_upload: pair %llu indexes outside the tables", (unsigned long long)bad.load());
    std::sort(keys.begin(), keys.end());
    std::vector<uint64_t> cum(n);
    uint64_t run = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t r, h;
        pair_of(keys[i] & 0xFFFFFFFFull, r, h);
        run += (uint64_t)b->read_len[r] * b->hap_len[h];
        cum[i] = run;
    }
    rb->index.assign(n_dev, {});
    rb->pr.assign(n_dev, {});
    rb->ph.assign(n_dev, {});
    rb->cells.assign(n_dev, 0);
    uint64_t prev = 0;
    for (size_t g = 0; g < n_dev; ++g) {
        uint64_t cut = n;
        if (g + 1 < n_dev) {
            const long double target = (long double)run * (long double)(g + 1) / (long double)n_dev;
            cut = (uint64_t)(std::lower_bound(cum.begin(), cum.end(), target, [](uint64_t c, long double t) { return (long double)c < t; }) - cum.begin());
            cut = std::max(cut, prev);
        }
        rb->index[g].resize(cut - prev);
        rb->pr[g].resize(cut - prev);
        rb->ph[g].resize(cut - prev);
        for (uint64_t i = prev; i < cut; ++i) {
            const uint32_t p = (uint32_t)(keys[i] & 0xFFFFFFFFull);
            uint32_t r, h;
            pair_of(p, r, h);
            rb->index[g][i - prev] = p;
            rb->pr[g][i - prev] = r;
            rb->ph[g][i - prev] = h;
        }
        rb->cells[g] = (cut ? cum[cut - 1] : 0) - (prev ? cum[prev - 1] : 0);
        prev = cut;
    }
    return PHMM_OK;
}

// A slice as a batch of its own: only the reads and haplotypes its pairs refer to, packed into tables of its own (a slice is a
// run of the length-binned ORDER, so its reads lie all over the caller's tables: windows would not do), indices remapped.  Every
// device therefore packs, uploads and prepares its own share of the tables instead of the whole batch (round 2 handed every
// device the full tables: N times the upload and N times the haplotype planes).
struct SliceTables
{
    std::vector<uint8_t> bases, qual, ins, del, gcp, mapq, haps;
    std::vector<uint32_t> roff, rlen, hoff, hlen, pr, ph;
    phmm_read_prep prep{};
    phmm_batch view{};
};
static void slice_of(const phmm_batch *b, const phmm_multi_resident *rb, size_t g, SliceTables &t)
{
    const std::vector<uint32_t> &gpr = rb->pr[g], &gph = rb->ph[g];
    const size_t n = gpr.size();
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    std::vector<uint32_t> rmap(b->n_reads, kNone), hmap(b->n_haps, kNone);
    t.pr.resize(n ? n : 1);
    t.ph.resize(n ? n : 1);
    const bool has_gap_arrays = b->prep == nullptr;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t r = gpr[i], h = gph[i];
        if (rmap[r] == kNone) {
            rmap[r] = (uint32_t)t.rlen.size();
            const uint32_t o = b->read_off[r], len = b->read_len[r];
            t.roff.push_back((uint32_t)t.bases.size());
            t.rlen.push_back(len);
            t.bases.insert(t.bases.end(), b->read_bases + o, b->read_bases + o + len);
            t.qual.insert(t.qual.end(), b->read_qual + o, b->read_qual + o + len);
            if (has_gap_arrays) {
                t.ins.insert(t.ins.end(), b->read_ins + o, b->read_ins + o + len);
                t.del.insert(t.del.end(), b->read_del + o, b->read_del + o + len);
                t.gcp.insert(t.gcp.end(), b->read_gcp + o, b->read_gcp + o + len);
            }
            else {
                t.mapq.push_back(b->prep->read_mapq[r]);
            }
        }
        if (hmap[h] == kNone) {
            hmap[h] = (uint32_t)t.hlen.size();
            const uint32_t o = b->hap_off[h], len = b->hap_len[h];
            t.hoff.push_back((uint32_t)t.haps.size());
            t.hlen.push_back(len);
            t.haps.insert(t.haps.end(), b->hap_bases + o, b->hap_bases + o + len);
        }
        t.pr[i] = rmap[r];
        t.ph[i] = hmap[h];
    }
    phmm_batch &s = t.view;
    s = phmm_batch{};
    s.n_reads = (uint32_t)t.rlen.size();
    s.n_haps = (uint32_t)t.hlen.size();
    s.n_pairs = n;
    s.read_bases = t.bases.data();
    s.read_qual = t.qual.data();
    s.read_ins = has_gap_arrays ? t.ins.data() : nullptr;
    s.read_del = has_gap_arrays ? t.del.data() : nullptr;
    s.read_gcp = has_gap_arrays ? t.gcp.data() : nullptr;
    s.read_off = t.roff.data();
    s.read_len = t.rlen.data();
    s.hap_bases = t.haps.data();
    s.hap_off = t.hoff.data();
    s.hap_len = t.hlen.data();
    s.pair_read = t.pr.data();  // (never null: an empty slice must not read as "full cross product")
    s.pair_hap = t.ph.data();
    if (b->prep) {
        t.prep = *b->prep;
        t.prep.read_mapq = t.mapq.data();
        s.prep = &t.prep;
    }
}

int phmm_multi_upload(phmm_multi *m, const phmm_batch *b, phmm_multi_resident **out)
{
    if (!m || !b || !out) return fail(PHMM_E_INVALID, "phmm_multi_upload: null argument");
    *out = nullptr;
    phmm_multi_resident *rb = new phmm_multi_resident();
    int rc = partition_equal_cells(b, m->ctx.size(), rb);
    if (rc == PHMM_OK) {
        rb->part.assign(m->ctx.size(), nullptr);
        rc = on_every_device(m, [&](size_t g) {
            SliceTables t;  // (the upload copies what it needs: the slice's tables may go when it returns)
            slice_of(b, rb, g, t);
            return impl::phmm_upload(m->ctx[g], &t.view, &rb->part[g]);
        });
    }
    if (rc != PHMM_OK) {
        impl::phmm_multi_release(m, rb);
        return rc;
    }
    *out = rb;
    return PHMM_OK;
}

int phmm_multi_sweep(phmm_multi *m, phmm_multi_resident *rb)
{
    if (!m || !rb) return fail(PHMM_E_INVALID, "phmm_multi_sweep: null argument");
    // enqueue only (asynchronous): cheap enough to do from the calling thread, device after device
    for (size_t g = 0; g < m->ctx.size(); ++g) {
        const int rc = impl::phmm_sweep(m->ctx[g], rb->part[g]);
        if (rc != PHMM_OK) return rc;
    }
    return PHMM_OK;
}

int phmm_multi_synchronize(phmm_multi *m)
{
    if (!m) return fail(PHMM_E_INVALID, "phmm_multi_synchronize: null argument");
    for (phmm_context *c : m->ctx) {
        const int rc = impl::phmm_synchronize(c);
        if (rc != PHMM_OK) return rc;
    }
    return PHMM_OK;
}

int phmm_multi_sweep_timed(phmm_multi *m, phmm_multi_resident *rb, int iters, double *wall_ms_per_pass, double *device_ms_per_pass,
                           double *fp32_ms_per_pass)
{
    if (!m || !rb || iters < 1 || !wall_ms_per_pass) return fail(PHMM_E_INVALID, "phmm_multi_sweep_timed: bad arguments");
    const size_t n = m->ctx.size();
    // every device thread waits at the gate, sweeps `iters` passes bracketed by HIP events on its own stream and reports its
    // wall time from the common start; the job's time is the slowest device's
    // (the gate is used only while every device has a thread of its own: when a helper thread could not be started the
    // remaining bodies run one after another on the caller, `serial` is set first and nobody waits for a body that has not begun;
    // the common start is then the call's own)
    std::atomic<size_t> arrived{0};
    std::atomic<bool> go{false}, serial{false};
    const auto t_entry = std::chrono::steady_clock::now();
    std::atomic<int64_t> start_ns{0};  // offset of the common start from t_entry
    std::vector<double> wall(n, 0.0), dev(n, 0.0), f32(n, 0.0);
    const int rc = on_every_device(m, [&](size_t g) {
        int r = impl::phmm_synchronize(m->ctx[g]);
        if (arrived.fetch_add(1) + 1 == n && !serial.load(std::memory_order_acquire)) {
            start_ns.store(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_entry).count());
            go.store(true, std::memory_order_release);
        }
        while (!go.load(std::memory_order_acquire) && !serial.load(std::memory_order_acquire)) std::this_thread::yield();
        if (r != PHMM_OK) return r;
        r = impl::phmm_sweep_timed(m->ctx[g], rb->part[g], iters, &dev[g], &f32[g]);
        if (r == PHMM_OK) r = impl::phmm_synchronize(m->ctx[g]);
        const auto t_start = t_entry + std::chrono::nanoseconds(start_ns.load());
        wall[g] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        return r;
    }, &serial);
    if (rc != PHMM_OK) return rc;
    double w = 0.0;
    for (size_t g = 0; g < n; ++g) w = std::max(w, wall[g]);
    *wall_ms_per_pass = w / iters;
    for (size_t g = 0; g < n; ++g) {
        if (device_ms_per_pass) device_ms_per_pass[g] = dev[g];
        if (fp32_ms_per_pass) fp32_ms_per_pass[g] = f32[g];
    }
    return PHMM_OK;
}

int phmm_multi_download(phmm_multi *m, phmm_multi_resident *rb, float *out_raw_f32, double *out_log10, phmm_stats *stats)
{
    if (!m || !rb) return fail(PHMM_E_INVALID, "phmm_multi_download: null argument");
    const size_t n = m->ctx.size();
    std::vector<phmm_stats> st(n);
    const int rc = on_every_device(m, [&](size_t g) -> int {
        const size_t k = rb->index[g].size();
        std::vector<float> raw(out_raw_f32 ? k : 0);
        std::vector<double> l10(out_log10 ? k : 0);
        const int r = impl::phmm_download(m->ctx[g], rb->part[g], out_raw_f32 ? raw.data() : nullptr, out_log10 ? l10.data() : nullptr, &st[g]);
        if (r != PHMM_OK) return r;
        const uint32_t *idx = rb->index[g].data();
        if (out_raw_f32)
            for (size_t i = 0; i < k; ++i) out_raw_f32[idx[i]] = raw[i];
        if (out_log10)
            for (size_t i = 0; i < k; ++i) out_log10[idx[i]] = l10[i];
        return PHMM_OK;
    });
    if (rc != PHMM_OK) return rc;
    if (stats) {
        *stats = phmm_stats{};
        for (size_t g = 0; g < n; ++g) {
            stats->cells += st[g].cells;
            stats->n_rescued += st[g].n_rescued;
            stats->kernel_ms = std::max(stats->kernel_ms, st[g].kernel_ms);
            stats->n_launches += st[g].n_launches;
            stats->max_rows_per_lane = std::max(stats->max_rows_per_lane, st[g].max_rows_per_lane);
        }
    }
    return PHMM_OK;
}

int phmm_multi_compute(phmm_multi *m, const phmm_batch *b, float *out_raw_f32, double *out_log10, phmm_stats *stats)
{
    if (!m || !b) return fail(PHMM_E_INVALID, "phmm_multi_compute: null argument");
    const auto t0 = std::chrono::steady_clock::now();
    phmm_multi_resident part;  // only the cut lives here; every device computes through its context's own scratch batch
    int rc = partition_equal_cells(b, m->ctx.size(), &part);
    if (rc != PHMM_OK) return rc;
    const size_t n = m->ctx.size();
    std::vector<phmm_stats> st(n);
    rc = on_every_device(m, [&](size_t g) -> int {
        SliceTables t;
        slice_of(b, &part, g, t);
        const size_t k = part.index[g].size();
        std::vector<float> raw(out_raw_f32 ? k : 0);
        std::vector<double> l10(out_log10 ? k : 0);
        const int r = impl::phmm_compute(m->ctx[g], &t.view, out_raw_f32 ? raw.data() : nullptr, out_log10 ? l10.data() : nullptr, &st[g]);
        if (r != PHMM_OK) return r;
        const uint32_t *idx = part.index[g].data();
        if (out_raw_f32)
            for (size_t i = 0; i < k; ++i) out_raw_f32[idx[i]] = raw[i];
        if (out_log10)
            for (size_t i = 0; i < k; ++i) out_log10[idx[i]] = l10[i];
        return PHMM_OK;
    });
    if (rc != PHMM_OK) return rc;
    if (stats) {
        *stats = phmm_stats{};
        for (size_t g = 0; g < n; ++g) {
            stats->cells += st[g].cells;
            stats->n_rescued += st[g].n_rescued;
            stats->kernel_ms = std::max(stats->kernel_ms, st[g].kernel_ms);
            stats->n_launches += st[g].n_launches;
            stats->max_rows_per_lane = std::max(stats->max_rows_per_lane, st[g].max_rows_per_lane);
        }
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return PHMM_OK;
}
}  // namespace impl

extern "C" {

int phmm_create_multi(const int *device_ids, int n, phmm_multi **out)
{
    return guarded("phmm_create_multi", [&] { return impl::phmm_create_multi(device_ids, n, out); });
}
int phmm_create_multi_cpu(int n_slices, int n_threads_each, int allow_simd, phmm_multi **out)
{
    return guarded("phmm_create_multi_cpu", [&] { return impl::phmm_create_multi_cpu(n_slices, n_threads_each, allow_simd, out); });
}
void phmm_destroy_multi(phmm_multi *m)
{
    (void)guarded("phmm_destroy_multi", [&] { impl::phmm_destroy_multi(m); return PHMM_OK; });
}
int phmm_multi_device_count(const phmm_multi *m) { return m ? (int)m->ctx.size() : 0; }
int phmm_multi_device_id(const phmm_multi *m, int i) { return (m && i >= 0 && (size_t)i < m->ctx.size()) ? m->ctx[i]->device : -1; }
int phmm_multi_compute(phmm_multi *m, const phmm_batch *b, float *raw, double *l10, phmm_stats *st)
{
    return guarded("phmm_multi_compute", [&] { return impl::phmm_multi_compute(m, b, raw, l10, st); });
}
int phmm_multi_upload(phmm_multi *m, const phmm_batch *b, phmm_multi_resident **out)
{
    return guarded("phmm_multi_upload", [&] { return impl::phmm_multi_upload(m, b, out); });
}
int phmm_multi_sweep(phmm_multi *m, phmm_multi_resident *rb) { return guarded("phmm_multi_sweep", [&] { return impl::phmm_multi_sweep(m, rb); }); }
int phmm_multi_synchronize(phmm_multi *m) { return guarded("phmm_multi_synchronize", [&] { return impl::phmm_multi_synchronize(m); }); }
int phmm_multi_sweep_timed(phmm_multi *m, phmm_multi_resident *rb, int iters, double *wall_ms, double *dev_ms, double *f32_ms)
{
    return guarded("phmm_multi_sweep_timed", [&] { return impl::phmm_multi_sweep_timed(m, rb, iters, wall_ms, dev_ms, f32_ms); });
}
int phmm_multi_download(phmm_multi *m, phmm_multi_resident *rb, float *raw, double *l10, phmm_stats *st)
{
    return guarded("phmm_multi_download", [&] { return impl::phmm_multi_download(m, rb, raw, l10, st); });
}
int phmm_plan_slices(const phmm_batch *b, int n_slices, uint32_t *slice_of_pair, uint64_t *cells)
{
    // the cut alone, no device involved: lets the N > 1 path be checked on a machine without GPUs
    return guarded("phmm_plan_slices", [&]() -> int {
        if (!b || n_slices < 1) return fail(PHMM_E_INVALID, "phmm_plan_slices: bad arguments");
        phmm_multi_resident plan;
        const int rc = impl::partition_equal_cells(b, (size_t)n_slices, &plan);
        if (rc != PHMM_OK) return rc;
        for (int g = 0; g < n_slices; ++g) {
            if (cells) cells[g] = plan.cells[g];
            if (slice_of_pair)
                for (uint32_t p : plan.index[g]) slice_of_pair[p] = (uint32_t)g;
        }
        return PHMM_OK;
    });
}

int phmm_multi_shard_cells(const phmm_multi_resident *rb, uint64_t *cells, uint64_t *pairs)
{
    if (!rb) return fail(PHMM_E_INVALID, "phmm_multi_shard_cells: null argument");
    for (size_t g = 0; g < rb->cells.size(); ++g) {
        if (cells) cells[g] = rb->cells[g];
        if (pairs) pairs[g] = rb->index[g].size();
    }
    return PHMM_OK;
}
void phmm_multi_release(phmm_multi *m, phmm_multi_resident *rb)
{
    (void)guarded("phmm_multi_release", [&] { impl::phmm_multi_release(m, rb); return PHMM_OK; });
}

}  // extern "C"
