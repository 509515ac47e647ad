// sw_capi.cpp -- the C ABI of include/smithwaterman_hip.h: host side of the MI355X Smith-Waterman aligner.
// There is no CPU fallback here: without a GPU swhip_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/smithwaterman_hip.h"
#include "sw_device.h"

// Test and tuning hooks (SWHIP_*) are dead unless PHMM_ENABLE_TEST_HOOKS=1 is set -- the same master switch as the PairHMM library's
// (pairhmm_binning.h): a production process that inherits SWHIP_FORCE_LANES must not run a slower kernel shape for it.
// (SWHIP_BLOCKING_SYNC is an operating choice, read directly.)
static const char *sw_hook(const char *name)
{
    const char *m = getenv("PHMM_ENABLE_TEST_HOOKS");
    if (!m || !m[0] || (m[0] == '0' && m[1] == 0)) return nullptr;
    return getenv(name);
}

static_assert(sizeof(swhip_alignment) == sizeof(swhip::Alignment), "device and ABI result structs must agree");

namespace
{

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(SWHIP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// grow-only device buffer
struct DevBytes
{
    void *p = nullptr;
    size_t cap = 0;
    ~DevBytes()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
};

struct HostBytes
{
    void *p = nullptr;
    size_t cap = 0;
    ~HostBytes()
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
        if (e == hipSuccess) cap = want;
        return e;
    }
};

}  // namespace

struct swhip_context
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_done = nullptr;  // SWHIP_BLOCKING_SYNC=1: the call's one wait sleeps in the driver instead of polling
    DevBytes blob, bt, scratch, cigars;
    HostBytes stage, back, back_cigars;
};

static int swhip_align_impl(swhip_context *c, const swhip_params *params, const swhip_batch *b, swhip_alignment *out, uint32_t *cigars,
                uint64_t cigar_capacity, uint64_t *cigars_used, swhip_stats *stats)
{
    if (!c || !params || !b || (b->n_pairs && (!out || !cigars_used)))
        return fail(SWHIP_E_INVALID, "swhip_align: null argument");
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t n = b->n_pairs;
    if (cigars_used) *cigars_used = 0;
    if (stats) *stats = swhip_stats{};
    if (n == 0) return SWHIP_OK;
    if (n > 0xFFFFFFF0ull) return fail(SWHIP_E_INVALID, "swhip_align: more than 2^32 alignments in one batch");
    if (!b->ref_bases || !b->ref_off || !b->ref_len || !b->seq_bases || !b->seq_off || !b->seq_len || !b->pair_ref || !b->pair_seq)
        return fail(SWHIP_E_INVALID, "swhip_align: null table pointer");
    // the kernel keeps scores in 31 bits around a bias (sw_device.h): penalties of the reference are below 300
    for (int32_t w : {params->match, params->mismatch, params->gap_open, params->gap_extend})
        // the kernel drops sw_avx.c's MATRIX_MIN_CUTOFF clamp and keeps scores biased; that is bit-exact for |penalty| <= 512 on
        // sequences under 2048 bases (sw_device.h), which covers both parameter sets of the reference (sw_avx.c:13-14)
        if (w > 512 || w < -512) return fail(SWHIP_E_INVALID, "swhip_align: scoring parameter %d outside [-512, 512]", w);
    HIP_TRY(hipSetDevice(c->device));

    // ---- validate, find extents ----
    uint64_t ref_bytes = 0, seq_bytes = 0;
    for (uint32_t r = 0; r < b->n_refs; ++r) {
        if (b->ref_len[r] == 0) return fail(SWHIP_E_INVALID, "swhip_align: reference sequence %u is empty", r);
        ref_bytes = std::max<uint64_t>(ref_bytes, (uint64_t)b->ref_off[r] + b->ref_len[r]);
    }
    for (uint32_t s = 0; s < b->n_seqs; ++s) {
        if (b->seq_len[s] == 0) return fail(SWHIP_E_INVALID, "swhip_align: sequence %u is empty", s);
        seq_bytes = std::max<uint64_t>(seq_bytes, (uint64_t)b->seq_off[s] + b->seq_len[s]);
    }

    // ---- classify: (lanes per alignment, columns per lane); longest reference first inside a launch ----
    constexpr uint64_t kWideMaxPairs = 2048;
    constexpr uint32_t kLdsFlagsMaxPairs = 768;  // 256 CUs x up to 3 workgroups of ~50 KB
    int forced_lanes = 0;
    if (const char *e = sw_hook("SWHIP_FORCE_LANES")) forced_lanes = atoi(e);  // tests
    std::vector<uint64_t> keys;  // class << 48 | (0xFFFF - len1) << 32 | pair
    keys.reserve(n);
    uint64_t cells = 0, worst_elems = 0;
    for (uint64_t p = 0; p < n; ++p) {
        const uint32_t r = b->pair_ref[p], s = b->pair_seq[p];
        if (r >= b->n_refs || s >= b->n_seqs)
            return fail(SWHIP_E_INVALID, "swhip_align: alignment %llu indexes outside the tables", (unsigned long long)p);
        const uint32_t l1 = b->ref_len[r], l2 = b->seq_len[s];
        if (std::max(l1, l2) >= (uint32_t)swhip::kMaxSeqLen) {  // sw_avx.c:490-495
            out[p] = swhip_alignment{};
            out[p].alignment_offset = 1;
            out[p].too_long = 1;
            continue;
        }
        // 16 lanes per alignment (four per wave) is the throughput shape; a batch of a region or a few would put a dozen
        // slow waves on a chip of 1024 SIMDs, so small batches take a whole wave per alignment (half the sweep time)
        int lanes = (l2 <= 16u * swhip::kMaxColsPerLaneNarrow && n > kWideMaxPairs) ? 16 : 64;
        if (forced_lanes == 16 && l2 <= 16u * swhip::kMaxColsPerLaneNarrow) lanes = 16;
        if (forced_lanes == 64) lanes = 64;
        const int k = swhip::pick_cols_per_lane(l2, lanes);
        const uint64_t cls = (uint64_t)(lanes == 16 ? 0 : 64) + (uint64_t)k;
        keys.push_back((cls << 48) | ((uint64_t)(0xFFFFu - l1) << 32) | p);
        cells += (uint64_t)l1 * l2;
        worst_elems += (uint64_t)l1 + l2 + 2;
    }
    // Small batches (everything one alignment per wave): a launch per columns-per-lane value costs a lone wave's sweep each, one after
    // the other on the stream, and a region whose reads were trimmed to different lengths has two or three of them.  A class rides along
    // with the next one up -- its alignments simply leave the upper lanes idle -- as long as that at most doubles its columns per lane.
    if (n <= kWideMaxPairs && !sw_hook("SWHIP_NO_CLASS_MERGE")) {
        bool present[128] = {false};
        for (uint64_t key : keys) present[(key >> 48) & 127] = true;
        uint8_t to[128];
        for (int i = 0; i < 128; ++i) to[i] = (uint8_t)i;
        int group_min = 0, members[128], n_members = 0;
        for (int cls = 65; cls < 128; ++cls) {  // 64 + k: the one-alignment-per-wave classes, ascending
            if (!present[cls]) continue;
            const int k = cls - 64;
            if (n_members && k <= 2 * group_min) {
                for (int i = 0; i < n_members; ++i) to[members[i]] = (uint8_t)cls;  // the group now runs with this class's columns per lane
            }
            else {
                n_members = 0;
                group_min = k;
            }
            members[n_members++] = cls;
        }
        for (uint64_t &key : keys) key = (key & 0x0000FFFFFFFFFFFFull) | ((uint64_t)to[(key >> 48) & 127] << 48);
    }
    std::sort(keys.begin(), keys.end());
    const uint64_t m = keys.size();
    if (stats) stats->cells = cells;
    if (m == 0) return SWHIP_OK;

    // ---- per-pair device offsets ----
    std::vector<uint32_t> order(m);
    std::vector<uint64_t> bt_off(n, 0), scr_off(n, 0);
    uint64_t bt_words = 0, bt_peak = 0, scr_slots = 0;
    uint64_t kBtWordsPerLaunch = 1ull << 30;
    if (const char *e = sw_hook("SWHIP_BT_WORDS_PER_LAUNCH")) kBtWordsPerLaunch = std::max<uint64_t>(1, strtoull(e, nullptr, 10));  // tests
    struct Launch
    {
        int lanes, k;
        uint32_t first, count, max_len1;
    };
    std::vector<Launch> launches;
    for (uint64_t i = 0; i < m; ++i) {
        const uint32_t p = (uint32_t)(keys[i] & 0xFFFFFFFFull);
        const uint64_t cls = keys[i] >> 48;
        const int lanes = cls >= 64 ? 64 : 16, k = (int)(cls >= 64 ? cls - 64 : cls);
        order[i] = p;
        const uint32_t l1 = b->ref_len[b->pair_ref[p]], l2 = b->seq_len[b->pair_seq[p]];
        const uint64_t need = swhip::bt_words(l1, lanes, k);
        // launches run one after another on the context's stream, so their backtrack storage can be the same memory: a
        // launch ends where the next alignment would push it past kBtWordsPerLaunch (4 GiB), and the next starts at 0
        if (launches.empty() || launches.back().lanes != lanes || launches.back().k != k || bt_words + need > kBtWordsPerLaunch) {
            launches.push_back(Launch{lanes, k, (uint32_t)i, 0, 0});
            bt_peak = std::max(bt_peak, bt_words);
            bt_words = 0;
        }
        launches.back().count++;
        launches.back().max_len1 = std::max(launches.back().max_len1, l1);
        bt_off[p] = bt_words;
        bt_words += need;
        scr_off[p] = scr_slots;
        scr_slots += swhip::scratch_slots(l1, l2, k);
    }

    // ---- upload: one pinned stage, one copy ----
    hipStream_t st = c->stream;
    HIP_TRY(hipStreamSynchronize(st));
    size_t total = 0;
    auto reserve = [&](size_t bytes) {
        const size_t at = total;
        total += (bytes + 255) & ~size_t(255);
        return at;
    };
    const size_t o_ref = reserve(ref_bytes), o_roff = reserve(b->n_refs * 4ull), o_rlen = reserve(b->n_refs * 4ull);
    const size_t o_seq = reserve(seq_bytes), o_soff = reserve(b->n_seqs * 4ull), o_slen = reserve(b->n_seqs * 4ull);
    const size_t o_pr = reserve(n * 4), o_ps = reserve(n * 4), o_order = reserve(m * 4);
    const size_t o_bt = reserve(n * 8), o_scr = reserve(n * 8), o_cur = reserve(4);
    HIP_TRY(c->stage.ensure(total));
    HIP_TRY(c->blob.ensure(total));
    uint8_t *sp = static_cast<uint8_t *>(c->stage.p);
    memcpy(sp + o_ref, b->ref_bases, ref_bytes);
    memcpy(sp + o_roff, b->ref_off, b->n_refs * 4ull);
    memcpy(sp + o_rlen, b->ref_len, b->n_refs * 4ull);
    memcpy(sp + o_seq, b->seq_bases, seq_bytes);
    memcpy(sp + o_soff, b->seq_off, b->n_seqs * 4ull);
    memcpy(sp + o_slen, b->seq_len, b->n_seqs * 4ull);
    memcpy(sp + o_pr, b->pair_ref, n * 4);
    memcpy(sp + o_ps, b->pair_seq, n * 4);
    memcpy(sp + o_order, order.data(), m * 4);
    memcpy(sp + o_bt, bt_off.data(), n * 8);
    memcpy(sp + o_scr, scr_off.data(), n * 8);
    memset(sp + o_cur, 0, 4);
    HIP_TRY(hipMemcpyAsync(c->blob.p, c->stage.p, total, hipMemcpyHostToDevice, st));
    HIP_TRY(c->bt.ensure(std::max(bt_peak, bt_words) * 4));
    HIP_TRY(c->scratch.ensure(scr_slots * 4));
    // Results never take a copy of their own: the kernel's last lane writes each alignment's record straight into pinned host
    // memory, and for batches whose worst-case element count is small (region-sized calls: the latency case) the CIGAR elements
    // too.  Large batches keep the element array in HBM and fetch the part that was used.
    const size_t res_bytes = n * sizeof(swhip::Alignment);
    HIP_TRY(c->back.ensure(res_bytes));
    constexpr uint64_t kDirectCigarElems = 4ull << 20;  // 16 MB of pinned memory at most
    const bool cigars_direct = worst_elems <= kDirectCigarElems && !sw_hook("SWHIP_CIGARS_BY_COPY");  // (the hook: tests)
    if (cigars_direct)
        HIP_TRY(c->back_cigars.ensure(worst_elems * 4));
    else
        HIP_TRY(c->cigars.ensure(worst_elems * 4));

    uint8_t *base = static_cast<uint8_t *>(c->blob.p);
    swhip::SwArgs a{};
    a.ref_bases = base + o_ref;
    a.ref_off = reinterpret_cast<const uint32_t *>(base + o_roff);
    a.ref_len = reinterpret_cast<const uint32_t *>(base + o_rlen);
    a.seq_bases = base + o_seq;
    a.seq_off = reinterpret_cast<const uint32_t *>(base + o_soff);
    a.seq_len = reinterpret_cast<const uint32_t *>(base + o_slen);
    a.pair_ref = reinterpret_cast<const uint32_t *>(base + o_pr);
    a.pair_seq = reinterpret_cast<const uint32_t *>(base + o_ps);
    a.w_match = params->match;
    a.w_mismatch = params->mismatch;
    a.w_open = params->gap_open;
    a.w_extend = params->gap_extend;
    a.bt = static_cast<uint32_t *>(c->bt.p);
    a.bt_off = reinterpret_cast<const uint64_t *>(base + o_bt);
    a.scratch = static_cast<int32_t *>(c->scratch.p);
    a.scratch_off = reinterpret_cast<const uint64_t *>(base + o_scr);
    a.out = static_cast<swhip::Alignment *>(c->back.p);
    a.cigars = static_cast<uint32_t *>(cigars_direct ? c->back_cigars.p : c->cigars.p);
    a.cigar_cursor = reinterpret_cast<uint32_t *>(base + o_cur);

    if (stats) HIP_TRY(hipEventRecord(c->ev0, st));
    for (const Launch &l : launches) {
        a.order = reinterpret_cast<const uint32_t *>(base + o_order) + l.first;
        a.n_pairs = l.count;
        a.lds_stride = (l.max_len1 + 15u) & ~15u;
        // SWHIP_LDS_FLAGS=1: small one-alignment-per-wave launches keep their flags in LDS (the traceback then waits ~100 cycles per
        // fetch, not ~700).  Off by default since the lanes of an alignment walk the traceback together (a fetch per 64 diagonal
        // moves): a lone region-sized call gains 3 % from it (136 against 140 us), but ~55 KB of LDS per alignment lets a CU hold two,
        // and with several calls in flight -- the adapter's case -- they queue for it (30 workers x 4 regions in flight: 118 000
        // regions/s with, 126 000 without).
        const uint64_t bt_bytes = swhip::bt_words(l.max_len1, l.lanes, l.k) * 4;
        const bool in_lds = l.lanes == 64 && l.count <= kLdsFlagsMaxPairs && a.lds_stride + bt_bytes <= 150u * 1024u && sw_hook("SWHIP_LDS_FLAGS");
        a.bt_in_lds = in_lds ? 1u : 0u;
        a.lds_bt_bytes = in_lds ? (uint32_t)bt_bytes : 0u;
        // the score table (sw_kernels.hip; up to 16 columns per lane): packed 16-bit scores
        // -- if the tables leave room for four workgroups per CU (a launch with seq1 much beyond 500 bases is faster
        // with the fourth workgroup than with the table: 150 x 1000, 3.81 against 3.85 ms)
        // (the scores always fit: parameters are within [-512, 512], above)
        // (one alignment per wave: from 8 columns per lane on -- the assembler's shapes, 2-3 % -- a region's 150-base reads at 3
        // columns per lane lose 3 % to the table's set-up)
        a.score_table = l.k <= 16 && (l.lanes == 16 || l.k >= 8) && !in_lds &&
                                (256u / (uint32_t)l.lanes) * (a.lds_stride + swhip::score_table_bytes(l.k, l.lanes)) <= 160u * 1024u / 4u &&
                                !sw_hook("SWHIP_NO_SCORE_TABLE")
                            ? 1u
                            : 0u;
        HIP_TRY(swhip::launch_sw(l.k, l.lanes, a, st));
    }
    if (stats) HIP_TRY(hipEventRecord(c->ev1, st));
    if (c->ev_done) {  // the one wait of the call
        HIP_TRY(hipEventRecord(c->ev_done, st));
        HIP_TRY(hipEventSynchronize(c->ev_done));
    }
    else {
        HIP_TRY(hipStreamSynchronize(st));
    }

    // every alignment took its elements from the cursor in one piece, so the highest end is the number used
    const swhip_alignment *dev_res = static_cast<const swhip_alignment *>(c->back.p);
    uint64_t used = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const uint32_t p = order[i];
        out[p] = dev_res[p];
        used = std::max<uint64_t>(used, (uint64_t)dev_res[p].cigar_off + dev_res[p].cigar_count);
    }
    *cigars_used = used;
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        stats->kernel_ms = ms;
        stats->n_launches = (uint32_t)launches.size();
    }
    if (used > cigar_capacity)
        return fail(SWHIP_E_CAPACITY, "swhip_align: %llu CIGAR elements, room for %llu", (unsigned long long)used, (unsigned long long)cigar_capacity);
    if (used) {
        if (!cigars) return fail(SWHIP_E_INVALID, "swhip_align: null cigars buffer");
        if (cigars_direct) {
            memcpy(cigars, c->back_cigars.p, (size_t)used * 4);
        }
        else {
            // pinned staging sized for what was produced (the worst case is sum(len1 + len2 + 2) elements)
            HIP_TRY(c->back_cigars.ensure((size_t)used * 4));
            HIP_TRY(hipMemcpyAsync(c->back_cigars.p, c->cigars.p, (size_t)used * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            memcpy(cigars, c->back_cigars.p, (size_t)used * 4);
        }
    }
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SWHIP_OK;
}

extern "C" {

int swhip_abi_version(void) { return SWHIP_ABI_VERSION; }

int swhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
const char *swhip_last_error(void) { return g_err; }

int swhip_create(int device_id, void *stream, swhip_context **out)
{
    if (!out || device_id < 0) return fail(SWHIP_E_INVALID, "swhip_create: bad arguments");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SWHIP_E_NO_DEVICE, "swhip_create: no HIP device visible (this aligner has no CPU fallback)");
    if (device_id >= n) return fail(SWHIP_E_NO_DEVICE, "swhip_create: device %d of %d does not exist", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    swhip_context *c = new (std::nothrow) swhip_context();
    if (!c) return fail(SWHIP_E_NOMEM, "swhip_create: out of host memory");
    c->device = device_id;
    hipError_t e = hipSuccess;
    if (stream) {
        c->stream = static_cast<hipStream_t>(stream);
    }
    else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        c->own_stream = e == hipSuccess;
    }
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess && getenv("SWHIP_BLOCKING_SYNC")) e = hipEventCreateWithFlags(&c->ev_done, hipEventBlockingSync | hipEventDisableTiming);
    if (e != hipSuccess) {
        swhip_destroy(c);
        return fail(SWHIP_E_HIP, "swhip_create: %s", hipGetErrorString(e));
    }
    *out = c;
    return SWHIP_OK;
}

void swhip_destroy(swhip_context *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int swhip_align(swhip_context *c, const swhip_params *params, const swhip_batch *b, swhip_alignment *out, uint32_t *cigars,
                uint64_t cigar_capacity, uint64_t *cigars_used, swhip_stats *stats)
{
    // nothing is thrown across the C boundary: a vector or a thread that cannot be had becomes SWHIP_E_NOMEM
    try {
        return swhip_align_impl(c, params, b, out, cigars, cigar_capacity, cigars_used, stats);
    }
    catch (const std::bad_alloc &) {
        return fail(SWHIP_E_NOMEM, "swhip_align: out of host memory");
    }
    catch (...) {
        return fail(SWHIP_E_NOMEM, "swhip_align: unexpected exception");
    }
}

}  // extern "C"
