// pairhmm_capi.cpp -- the extern "C" boundary of include/pairhmm_hip.h: contexts, resident batches, length binning,
// launches, rescue, host finalisation.  Host C++ over the HIP runtime; no CPU compute path (there is no fallback:
// without a device phmm_create fails with PHMM_E_NO_DEVICE).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pairhmm_hip.h"
#include "../host/pairhmm_host_logic.h"
#include "pairhmm_cpu.h"
#include "pairhmm_device.h"
#include "pairhmm_tables.h"

#include "pairhmm_engine.h"

namespace phmm_detail
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
}  // namespace phmm_detail
using phmm_detail::g_err;
thread_local PhaseClock *g_phase = nullptr;  // PHMM_PHASE_TIMING: the clock of the call in progress on this thread

// The entry points are implemented in this namespace with C++ linkage; the extern "C" wrappers at the end of the file
// catch whatever a body may throw (std::bad_alloc from a vector, std::system_error from a thread), so that nothing
// crosses the C boundary (include/pairhmm_hip.h).
namespace impl
{
void phmm_destroy(phmm_context *c);

int phmm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int phmm_get_tables(float *ph2pr_f32, float *mm_f32, double *ph2pr_f64, double *mm_f64)
{
    const phmm::HostTables &t = phmm::host_tables();
    if (ph2pr_f32) memcpy(ph2pr_f32, t.ph2pr_f32.data(), 128 * sizeof(float));
    if (mm_f32) memcpy(mm_f32, t.mm_f32.data(), phmm::kMmTableSize * sizeof(float));
    if (ph2pr_f64) memcpy(ph2pr_f64, t.ph2pr_f64.data(), 128 * sizeof(double));
    if (mm_f64) memcpy(mm_f64, t.mm_f64.data(), phmm::kMmTableSize * sizeof(double));
    return PHMM_OK;
}

int phmm_create(int device_id, void *stream, phmm_context **out)
{
    if (!out || device_id < 0) return fail(PHMM_E_INVALID, "phmm_create: bad arguments");
    *out = nullptr;
    // A context drives up to ten streams (two sweep lanes of three, upload, download) and a process usually holds several contexts; the
    // HIP runtime multiplexes them onto GPU_MAX_HW_QUEUES hardware queues, 4 by default, and streams that share one take turns.  The
    // one-shot pipeline and the adapters are built for 16: the variable has to be in the environment before the process's first HIP
    // call, which is not a library's to arrange (setenv races with other threads' getenv) -- the adapter's init_pairhmm_ptr_gpu does it
    // on the main thread before the workers exist, the Python bindings at import, bench.py at start; README.md has the line for others.
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(PHMM_E_NO_DEVICE, "phmm_create: no HIP device visible (this engine has no CPU fallback)");
    if (device_id >= n) return fail(PHMM_E_NO_DEVICE, "phmm_create: device %d of %d does not exist", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    phmm_context *c = new (std::nothrow) phmm_context();
    if (!c) return fail(PHMM_E_NOMEM, "phmm_create: out of host memory");
    c->device = device_id;
    if (const char *e = phmm_hook("PHMM_SIDE_STREAMS")) c->n_side = std::max(0, std::min((int)phmm_context::kSideStreams, atoi(e)));
    if (stream) {
        c->stream = static_cast<hipStream_t>(stream);
    }
    else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return fail(PHMM_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        c->own_stream = true;
    }
    const phmm::HostTables &t = phmm::host_tables();
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = c->ph2pr_f32.upload(t.ph2pr_f32.data(), 128, c->stream);
    if (e == hipSuccess) e = c->mm_f32.upload(t.mm_f32.data(), phmm::kMmTableSize, c->stream);
    if (e == hipSuccess) e = c->ph2pr_f64.upload(t.ph2pr_f64.data(), 128, c->stream);
    if (e == hipSuccess) e = c->mm_f64.upload(t.mm_f64.data(), phmm::kMmTableSize, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
    // (the side streams are made when a pass first fans out: ensure_side_streams)
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        impl::phmm_destroy(c);
        return fail(PHMM_E_HIP, "phmm_create: table upload failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return PHMM_OK;
}

int phmm_create_cpu(int n_threads, int allow_simd, phmm_context **out)
{
    if (!out || n_threads < 0) return fail(PHMM_E_INVALID, "phmm_create_cpu: bad arguments");
    *out = nullptr;
    phmm_context *c = new phmm_context();
    c->cpu = true;
    const unsigned hw = std::thread::hardware_concurrency();
    c->cpu_threads = n_threads > 0 ? (unsigned)n_threads : std::max(1u, hw);
    c->cpu_simd = allow_simd != 0;
    (void)phmm::host_tables();  // built once, under std::call_once
    *out = c;
    return PHMM_OK;
}

void phmm_destroy(phmm_context *c)
{
    if (!c) return;
    impl::phmm_stage_destroy(c, c->region_stage);
    c->region_stage = nullptr;
    if (c->cpu) {
        delete c->scratch;
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->fork) (void)hipEventDestroy(c->fork);
    for (hipEvent_t e : c->timing) (void)hipEventDestroy(e);
    for (int i = 0; i < phmm_context::kSideStreams; ++i) {
        if (c->side[i]) (void)hipStreamSynchronize(c->side[i]);
        if (c->join[i]) (void)hipEventDestroy(c->join[i]);
        if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
    }
    delete c->scratch;
    for (int i = 0; i < phmm_context::kChunkSlots; ++i) {
        delete c->chunk_slot[i];
        if (c->chunk_up[i]) (void)hipEventDestroy(c->chunk_up[i]);
        if (c->chunk_swept[i]) (void)hipEventDestroy(c->chunk_swept[i]);
        if (c->chunk_done[i]) (void)hipEventDestroy(c->chunk_done[i]);
    }
    if (c->down) {
        (void)hipStreamSynchronize(c->down);
        (void)hipStreamDestroy(c->down);
    }
    if (c->lane2.stream) {
        (void)hipStreamSynchronize(c->lane2.stream);
        (void)hipStreamDestroy(c->lane2.stream);
    }
    if (c->lane2.fork) (void)hipEventDestroy(c->lane2.fork);
    if (c->lane2.done) (void)hipEventDestroy(c->lane2.done);
    for (int i = 0; i < phmm_context::kSideStreams; ++i) {
        if (c->lane2.side[i]) {
            (void)hipStreamSynchronize(c->lane2.side[i]);
            (void)hipStreamDestroy(c->lane2.side[i]);
        }
        if (c->lane2.join[i]) (void)hipEventDestroy(c->lane2.join[i]);
    }
    if (c->rescue_lane.stream) {
        (void)hipStreamSynchronize(c->rescue_lane.stream);
        (void)hipStreamDestroy(c->rescue_lane.stream);
    }
    if (c->rescue_lane.fork) (void)hipEventDestroy(c->rescue_lane.fork);
    if (c->rescue_lane.done) (void)hipEventDestroy(c->rescue_lane.done);
    for (int i = 0; i < phmm_context::kSideStreams; ++i) {
        if (c->rescue_lane.side[i]) {
            (void)hipStreamSynchronize(c->rescue_lane.side[i]);
            (void)hipStreamDestroy(c->rescue_lane.side[i]);
        }
        if (c->rescue_lane.join[i]) (void)hipEventDestroy(c->rescue_lane.join[i]);
    }
    delete c->pack_pool;
    delete c->fin_pool;
    delete c->prep_pool;
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// The side streams of a context (and their join events) exist from the first pass that needs them.  Most contexts never do: a
// dispatcher of the drop-in adapter submits stages whose reads fall into one launch, and the runtime multiplexes a process's streams onto
// GPU_MAX_HW_QUEUES hardware queues -- sixteen streams made at start-up by the adapter's four contexts, on top of the caller's own, meant
// that two dispatchers' main streams could end up sharing a queue (whether they did depended on which thread created its streams first:
// the synchronous seam measured 2 200 or 2 950 GCUPS from one run of bench.py to the next, round 4).
// Not thread-safe: the caller serialises the calls that launch work on one context (include/pairhmm_hip.h).  All or nothing: the
// streams and events are made into locals and published together, so a failure half way leaves the context without side streams
// (and the next call tries again) instead of with a first stream that says "made" and null ones behind it.
int ensure_side_streams(phmm_context *c)
{
    if (c->cpu || c->side[0]) return PHMM_OK;
    hipStream_t side[phmm_context::kSideStreams] = {};
    hipEvent_t join[phmm_context::kSideStreams] = {};
    hipError_t e = hipSuccess;
    for (int i = 0; i < phmm_context::kSideStreams && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&join[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        for (int i = 0; i < phmm_context::kSideStreams; ++i) {
            if (join[i]) (void)hipEventDestroy(join[i]);
            if (side[i]) (void)hipStreamDestroy(side[i]);
        }
        return fail(PHMM_E_HIP, "side streams: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < phmm_context::kSideStreams; ++i) {
        c->join[i] = join[i];
        c->side[i] = side[i];
    }
    return PHMM_OK;
}

int phmm_synchronize(phmm_context *c)
{
    if (!c) return fail(PHMM_E_INVALID, "null context");
    if (c->cpu) return PHMM_OK;  // the CPU backend computes inside the calls
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PHMM_OK;
}

void phmm_release(phmm_context *c, phmm_resident *rb)
{
    if (!rb) return;
    if (c && !c->cpu) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    delete rb;
}

// (PhaseClock / PHASE, UploadOpts: pairhmm_engine.h)
// A batch reaches the device in two halves.  upload_prepare is host work only -- validate, pack every input into the resident's ONE
// pinned staging buffer, bin the pairs -- and touches nothing of the context but the device number, so the chunk pipeline of a big
// one-shot call runs it for several chunks at once on worker threads (pairhmm_chunked.cpp).  upload_enqueue is the HIP side: one copy,
// the views into the blob, the device buffers, the set-up launch; the pipeline keeps it on one thread, in chunk order.
int upload_prepare(phmm_context *c, const phmm_batch *b, phmm_resident *rb, const UploadOpts &opt, phmm_detail::BinScratch &bin_scratch)
{
    if ((b->pair_read == nullptr) != (b->pair_hap == nullptr))
        return fail(PHMM_E_INVALID, "phmm_upload: pair_read and pair_hap must both be given or both be NULL");
    const bool cross = b->pair_read == nullptr;
    const uint64_t n_pairs = cross ? (uint64_t)b->n_reads * b->n_haps : b->n_pairs;
    if (n_pairs > 0xFFFFFFF0ull) return fail(PHMM_E_INVALID, "phmm_upload: more than 2^32 pairs in one batch");
    const phmm_read_prep *prep = b->prep;
    if (n_pairs && (!b->read_bases || !b->read_qual || !b->read_off || !b->read_len || !b->hap_bases || !b->hap_off ||
                    !b->hap_len))
        return fail(PHMM_E_INVALID, "phmm_upload: null table pointer");
    if (n_pairs && !prep && (!b->read_ins || !b->read_del || !b->read_gcp))
        return fail(PHMM_E_INVALID, "phmm_upload: null gap-penalty array (and no phmm_read_prep given)");
    if (prep && (!prep->read_mapq || prep->pcr_model < 0 || prep->pcr_model > 3))
        return fail(PHMM_E_INVALID, "phmm_upload: bad phmm_read_prep");
    HIP_TRY(hipSetDevice(c->device));

    // ---- validate the tables, find extents ----
    uint64_t read_bytes = 0, hap_bytes = 0;
    uint32_t max_hap = 0, max_read = 0;
    for (uint32_t r = 0; r < b->n_reads; ++r) {
        if (b->read_len[r] == 0) return fail(PHMM_E_INVALID, "phmm_upload: read %u has length 0", r);
        read_bytes = std::max<uint64_t>(read_bytes, (uint64_t)b->read_off[r] + b->read_len[r]);
        max_read = std::max(max_read, b->read_len[r]);
    }
    for (uint32_t h = 0; h < b->n_haps; ++h) {
        if (b->hap_len[h] == 0) return fail(PHMM_E_INVALID, "phmm_upload: haplotype %u has length 0", h);
        hap_bytes = std::max<uint64_t>(hap_bytes, (uint64_t)b->hap_off[h] + b->hap_len[h]);
        max_hap = std::max(max_hap, b->hap_len[h]);
    }
    if (max_read > kMaxLen || max_hap > kMaxLen)
        return fail(PHMM_E_INVALID, "phmm_upload: sequence longer than %u bases", kMaxLen);
    if (phmm::team_lds_bytes((uint32_t)phmm::kPlanes * phmm::planes_words(max_hap), 8, 8, 16) > 160u * 1024u)
        return fail(PHMM_E_INVALID, "phmm_upload: haplotype of %u bases exceeds the LDS plane budget", max_hap);

    PHASE(0);
    rb->bins.clear();
    rb->last_rescued = 0;
    rb->last_kernel_ms = 0.0;
    rb->n_pairs = n_pairs;
    rb->n_reads = b->n_reads;
    rb->n_haps = b->n_haps;
    rb->max_hap_len = max_hap;
    rb->max_read_len = max_read;

    // the Rovaca case -- one gap-continuation penalty for every base and deletion gap-open == insertion gap-open
    // (read_record.cpp:205-210, pairhmm_internal.cpp:425) -- selects the kernel variant that shares those constants
    bool ug = read_bytes > 0;
    uint8_t gcp0 = 0;
    bool ugo = false;  // one gap-open quality for every base as well (PCR-free data, the synthetic configs)
    if (prep) ugo = prep->pcr_model == 0;  // no PCR model: every base keeps prep->gap_open (read_record.cpp:191-196)
    if (!prep && read_bytes) {
        gcp0 = b->read_gcp[b->read_off[0]] & 127;
        const uint8_t gop0 = b->read_ins[b->read_off[0]] & 127;
        uint8_t gop_diff = 0;
        const bool same_array = b->read_del == b->read_ins;
        for (uint32_t r = 0; r < b->n_reads && ug; ++r) {
            const uint8_t *g = b->read_gcp + b->read_off[r];
            const uint8_t *gi = b->read_ins + b->read_off[r], *gd = b->read_del + b->read_off[r];
            const uint32_t n = b->read_len[r];
            uint8_t diff = 0;  // OR of every deviation: no early exit inside a read, so the loop vectorises
            if (same_array)
                for (uint32_t i = 0; i < n; ++i) diff |= (uint8_t)(g[i] ^ gcp0);
            else
                for (uint32_t i = 0; i < n; ++i) diff |= (uint8_t)((g[i] ^ gcp0) | (gi[i] ^ gd[i]));
            for (uint32_t i = 0; i < n; ++i) gop_diff |= (uint8_t)(gi[i] ^ gop0);
            ug = (diff & 127) == 0;
        }
        ugo = ug && (gop_diff & 127) == 0;
    }
    rb->uniform_gcp = ug;
    rb->uniform_gop = ug && ugo && !phmm_hook("PHMM_NO_UNIFORM_GOP");  // (the variable: tests compare the two kernels)
    // In the Rovaca case two of the five read arrays carry no information: the deletion gap-opens are the insertion ones and
    // the gap-continuation array is one value.  They are not packed or copied: on the device `del` aliases `ins` and `gcp` is
    // a fill (a big one-shot batch spends more host time packing tables than binning pairs).
    const bool slim = ug && !prep;

    // ---- stage layout: every input travels in ONE pinned buffer and ONE copy; the index arrays are built in place ----
    uint64_t slots = 0;
    for (uint32_t h = 0; h < b->n_haps; ++h) slots += phmm::planes_words(b->hap_len[h]);
    if (slots * phmm::kPlanes > 0xFFFFFFF0ull) return fail(PHMM_E_INVALID, "phmm_upload: haplotype table too large");
    if (!opt.caller_synchronised)
        HIP_TRY(hipStreamSynchronize(c->stream));  // the previous pass of this resident batch no longer reads the stage or the blob
    size_t total = 0;
    auto reserve = [&](size_t bytes) {
        const size_t at = total;
        total += (bytes + 255) & ~size_t(255);
        return at;
    };
    const size_t o_bases = reserve(read_bytes), o_qual = reserve(read_bytes);
    const size_t o_mapq = reserve(prep ? b->n_reads : 0);
    const size_t o_ins = reserve(prep ? 0 : read_bytes), o_del = reserve(prep || slim ? 0 : read_bytes),
                 o_gcp = reserve(prep || slim ? 0 : read_bytes);
    const size_t o_roff = reserve(b->n_reads * sizeof(uint32_t)), o_rlen = reserve(b->n_reads * sizeof(uint32_t));
    const size_t o_hap = reserve(hap_bytes);
    const size_t o_hoff = reserve(b->n_haps * sizeof(uint32_t)), o_hlen = reserve(b->n_haps * sizeof(uint32_t));
    const size_t o_pr = reserve(n_pairs * sizeof(uint32_t)), o_ph = reserve(n_pairs * sizeof(uint32_t));
    const size_t o_order = reserve(n_pairs * sizeof(uint32_t));
    const size_t o_poff = reserve(b->n_haps * sizeof(uint32_t)), o_owner = reserve(slots * sizeof(uint32_t));
    HIP_TRY(rb->stage.ensure(total));
    uint8_t *stage = static_cast<uint8_t *>(rb->stage.p);
    auto copy = [](void *dst, const void *src, size_t bytes) {  // big tables of a big batch: a few threads share the copy
        // (with the chunk pipeline's helper pool current, handing out parts costs microseconds: a chunk's tables of a few hundred KB each
        // are split too -- they were 0.25 ms of serial memcpy per chunk of 118 000 pairs)
        if (bytes < (phmm_detail::t_pool ? (size_t(1) << 17) : (size_t(1) << 20))) {
            memcpy(dst, src, bytes);
            return;
        }
        parallel_ranges(bytes, std::min(4u, host_threads()), [&](uint64_t lo, uint64_t hi, unsigned) {
            memcpy(static_cast<uint8_t *>(dst) + lo, static_cast<const uint8_t *>(src) + lo, hi - lo);
        });
    };
    if (read_bytes) {
        copy(stage + o_bases, b->read_bases, read_bytes);
        copy(stage + o_qual, b->read_qual, read_bytes);
    }
    if (prep && b->n_reads) copy(stage + o_mapq, prep->read_mapq, b->n_reads);
    if (!prep && read_bytes) {
        copy(stage + o_ins, b->read_ins, read_bytes);
        if (!slim) {
            copy(stage + o_del, b->read_del, read_bytes);
            copy(stage + o_gcp, b->read_gcp, read_bytes);
        }
    }
    if (b->n_reads) {
        copy(stage + o_roff, b->read_off, b->n_reads * sizeof(uint32_t));
        copy(stage + o_rlen, b->read_len, b->n_reads * sizeof(uint32_t));
    }
    if (hap_bytes) copy(stage + o_hap, b->hap_bases, hap_bytes);
    if (b->n_haps) {
        copy(stage + o_hoff, b->hap_off, b->n_haps * sizeof(uint32_t));
        copy(stage + o_hlen, b->hap_len, b->n_haps * sizeof(uint32_t));
    }
    uint32_t *pr = reinterpret_cast<uint32_t *>(stage + o_pr), *ph = reinterpret_cast<uint32_t *>(stage + o_ph);
    uint32_t *order = reinterpret_cast<uint32_t *>(stage + o_order);
    {
        // haplotype plane layout
        uint32_t *plane_off = reinterpret_cast<uint32_t *>(stage + o_poff), *owner = reinterpret_cast<uint32_t *>(stage + o_owner);
        uint64_t s = 0;
        for (uint32_t h = 0; h < b->n_haps; ++h) {
            plane_off[h] = (uint32_t)(s * phmm::kPlanes);
            for (uint32_t w = 0; w < phmm::planes_words(b->hap_len[h]); ++w) owner[s++] = h;
        }
    }
    PHASE(1);

    // ---- pair list (materialise the cross product) and length binning: pairhmm_binning.cpp ----
    uint64_t cells = 0;
    {
        // (what the batch this slot held before observed on its long pairs goes into the context's history in upload_enqueue -- the
        // half that runs on the one enqueuing thread: this half runs on several prepare workers at once and must not write the context)
        rb->ragged = ragged_launches_for(ug);  // the classes share launches: no need for small ones to ride along at more rows per lane
        const int rc = phmm_detail::bin_pairs(b, cross, n_pairs, opt.read_bias, opt.hap_bias, pr, ph, order, bin_scratch, rb->bins, &cells,
                                              !rb->ragged);
        if (rc != PHMM_OK) return rc;
        merge_striped_bins_of_one_shape(rb->bins);
    }
    rb->cells = cells;
    PHASE(2);

    // what upload_enqueue needs of the layout
    UploadPlan &pl = rb->plan;
    pl.total = total;
    pl.read_bytes = read_bytes;
    pl.hap_bytes = hap_bytes;
    pl.slots = slots;
    pl.o_bases = o_bases, pl.o_qual = o_qual, pl.o_mapq = o_mapq, pl.o_ins = o_ins, pl.o_del = o_del, pl.o_gcp = o_gcp;
    pl.o_roff = o_roff, pl.o_rlen = o_rlen, pl.o_hap = o_hap, pl.o_hoff = o_hoff, pl.o_hlen = o_hlen;
    pl.o_pr = o_pr, pl.o_ph = o_ph, pl.o_order = o_order, pl.o_poff = o_poff, pl.o_owner = o_owner;
    pl.has_prep = prep != nullptr;
    pl.slim = slim;
    pl.gcp0 = gcp0;
    if (prep) pl.prep = *prep;  // (read_mapq was packed above: only the scalars are read again)
    return PHMM_OK;
}

int upload_enqueue(phmm_context *c, phmm_resident *rb, const UploadOpts &opt)
{
    HIP_TRY(hipSetDevice(c->device));
    const UploadPlan &pl = rb->plan;
    const size_t total = pl.total;
    const uint64_t read_bytes = pl.read_bytes, hap_bytes = pl.hap_bytes, slots = pl.slots, n_pairs = rb->n_pairs;
    const size_t o_bases = pl.o_bases, o_qual = pl.o_qual, o_mapq = pl.o_mapq, o_ins = pl.o_ins, o_del = pl.o_del, o_gcp = pl.o_gcp,
                 o_roff = pl.o_roff, o_rlen = pl.o_rlen, o_hap = pl.o_hap, o_hoff = pl.o_hoff, o_hlen = pl.o_hlen, o_pr = pl.o_pr,
                 o_ph = pl.o_ph, o_order = pl.o_order, o_poff = pl.o_poff, o_owner = pl.o_owner;
    const bool slim = pl.slim;
    const uint8_t gcp0 = pl.gcp0;
    const phmm_read_prep *prep = pl.has_prep ? &pl.prep : nullptr;
    const uint32_t n_reads = rb->n_reads, n_haps = rb->n_haps;
    hipStream_t st = opt.stream ? opt.stream : c->stream;
    // the batch this slot held before: what its last pass saw goes into the context's history (fp64 first, pairhmm_sweep.cpp); the
    // new batch starts from the context's history
    harvest_history(c, rb);
    for (int qi = 0; qi < kQueues; ++qi) rb->f64_share_known[qi] = rb->queue_f64_first[qi] = false;
    rb->hist_pending = false;
    HIP_TRY(rb->blob.alloc(total));
    // ---- upload ----
    if (opt.setup_stream && opt.setup_stream != st) {
        // The chunked call's synchronous form (PHMM_CHUNK_SYNC_UPLOAD; the default until the process asked for 16 hardware queues): the
        // set-up launch goes on the sweep stream and must see the copy.  A device-side wait there
        // (hipStreamWaitEvent) cost ~50 us of idle queue at every chunk boundary, and an event behind an asynchronous copy on the
        // upload stream does not fire before the sweep that is running has drained (its marker sits in a compute queue the busy
        // chip does not get round to: measured, hipEventSynchronize returned 1.7 ms late).  A synchronous copy waits on the DMA
        // engine's own signal: the calling thread is a chunk ahead of the device and has nothing else to do meanwhile.
        if (total) HIP_TRY(hipMemcpy(rb->blob.p, rb->stage.p, total, hipMemcpyHostToDevice));
        st = opt.setup_stream;
    }
    else if (total) {
        HIP_TRY(hipMemcpyAsync(rb->blob.p, rb->stage.p, total, hipMemcpyHostToDevice, st));
    }
    uint8_t *base = rb->blob.p;
    rb->read_bases = {base + o_bases, (size_t)read_bytes};
    rb->read_qual = {base + o_qual, (size_t)read_bytes};
    rb->read_mapq = {base + o_mapq, prep ? (size_t)n_reads : 0};
    rb->read_off = {reinterpret_cast<uint32_t *>(base + o_roff), n_reads};
    rb->read_len = {reinterpret_cast<uint32_t *>(base + o_rlen), n_reads};
    rb->hap_bases = {base + o_hap, (size_t)hap_bytes};
    rb->hap_off = {reinterpret_cast<uint32_t *>(base + o_hoff), n_haps};
    rb->hap_len = {reinterpret_cast<uint32_t *>(base + o_hlen), n_haps};
    rb->pair_read = {reinterpret_cast<uint32_t *>(base + o_pr), (size_t)n_pairs};
    rb->pair_hap = {reinterpret_cast<uint32_t *>(base + o_ph), (size_t)n_pairs};
    rb->order = {reinterpret_cast<uint32_t *>(base + o_order), (size_t)n_pairs};
    rb->plane_off = {reinterpret_cast<uint32_t *>(base + o_poff), n_haps};
    rb->word_owner = {reinterpret_cast<uint32_t *>(base + o_owner), (size_t)slots};
    if (prep) {
        HIP_TRY(rb->prep_gop.alloc(read_bytes));
        HIP_TRY(rb->prep_gcp.alloc(read_bytes));
        rb->read_ins = {rb->prep_gop.p, (size_t)read_bytes};
        rb->read_del = rb->read_ins;  // pairhmm_internal.cpp:425
        rb->read_gcp = {rb->prep_gcp.p, (size_t)read_bytes};
    }
    else {
        rb->read_ins = {base + o_ins, (size_t)read_bytes};
        if (slim) {
            HIP_TRY(rb->prep_gcp.alloc(read_bytes));
            HIP_TRY(hipMemsetAsync(rb->prep_gcp.p, gcp0, read_bytes, st));
            rb->read_del = rb->read_ins;  // pairhmm_internal.cpp:425 does the same on the host
            rb->read_gcp = {rb->prep_gcp.p, (size_t)read_bytes};
        }
        else {
            rb->read_del = {base + o_del, (size_t)read_bytes};
            rb->read_gcp = {base + o_gcp, (size_t)read_bytes};
        }
    }
    HIP_TRY(rb->planes.alloc(slots * phmm::kPlanes));
    // the rescue count lives behind the fp32 sums so that both come back in one copy
    const size_t count_at = (n_pairs + 63) & ~size_t(63);
    HIP_TRY(rb->raw32.alloc(count_at + 64));
    rb->rescue_count = {reinterpret_cast<uint32_t *>(rb->raw32.p + count_at), (size_t)kQueues};
    HIP_TRY(rb->raw64.alloc(n_pairs));
    HIP_TRY(rb->rescue_list.alloc(n_pairs));
    {
        // H3 + H4 on the device (in place on the uploaded raw qualities) when asked, haplotype planes, empty rescue queue
        phmm::SetupArgs sa{};
        sa.hap_bases = rb->hap_bases.p;
        sa.hap_off = rb->hap_off.p;
        sa.hap_len = rb->hap_len.p;
        sa.plane_off = rb->plane_off.p;
        sa.word_owner = rb->word_owner.p;
        sa.n_words = n_pairs ? (uint32_t)slots : 0;
        sa.planes = rb->planes.p;
        sa.read_bases = rb->read_bases.p;
        sa.read_qual = rb->read_qual.p;
        sa.read_gop = rb->read_ins.p;
        sa.read_gcp = rb->read_gcp.p;
        sa.read_off = rb->read_off.p;
        sa.read_len = rb->read_len.p;
        sa.read_mapq = rb->read_mapq.p;
        sa.n_prep_reads = prep ? n_reads : 0;
        sa.min_q = prep ? prep->min_quality_threshold : 0;
        sa.pcr_model = prep ? prep->pcr_model : 0;
        sa.gap_open = prep ? prep->gap_open : 0;
        sa.gap_cont = prep ? prep->gap_cont : 0;
        sa.rescue_count = rb->rescue_count.p;
        HIP_TRY(phmm::launch_setup(sa, st));
        rb->queue_is_empty = true;
    }
    PHASE(3);
    return PHMM_OK;
}

int upload_into(phmm_context *c, const phmm_batch *b, phmm_resident *rb, const UploadOpts &opt)
{
    const int rc = upload_prepare(c, b, rb, opt, c->bin_scratch);
    return rc != PHMM_OK ? rc : upload_enqueue(c, rb, opt);
}

// ---- CPU backend (phmm_create_cpu): the same entry points, computed on the host by pairhmm_cpu.cpp ----
static int cpu_check_batch(const phmm_batch *b, uint64_t *n_pairs)
{
    if ((b->pair_read == nullptr) != (b->pair_hap == nullptr))
        return fail(PHMM_E_INVALID, "phmm_upload: pair_read and pair_hap must both be given or both be NULL");
    const uint64_t n = b->pair_read ? b->n_pairs : (uint64_t)b->n_reads * b->n_haps;
    if (n > 0xFFFFFFF0ull) return fail(PHMM_E_INVALID, "phmm_upload: more than 2^32 pairs in one batch");
    if (n && (!b->read_bases || !b->read_qual || !b->read_off || !b->read_len || !b->hap_bases || !b->hap_off || !b->hap_len))
        return fail(PHMM_E_INVALID, "phmm_upload: null table pointer");
    if (n && !b->prep && (!b->read_ins || !b->read_del || !b->read_gcp))
        return fail(PHMM_E_INVALID, "phmm_upload: null gap-penalty array (and no phmm_read_prep given)");
    if (b->prep && (!b->prep->read_mapq || b->prep->pcr_model < 0 || b->prep->pcr_model > 3))
        return fail(PHMM_E_INVALID, "phmm_upload: bad phmm_read_prep");
    for (uint32_t r = 0; r < b->n_reads; ++r)
        if (b->read_len[r] == 0) return fail(PHMM_E_INVALID, "phmm_upload: read %u has length 0", r);
    for (uint32_t h = 0; h < b->n_haps; ++h)
        if (b->hap_len[h] == 0) return fail(PHMM_E_INVALID, "phmm_upload: haplotype %u has length 0", h);
    *n_pairs = n;
    return PHMM_OK;
}

// deep copy with the read table PREPARED: a batch that asks for read preparation gets H3 + H4 here (the same routines the
// drop-in adapter's host logic uses: pairhmm_internal.cpp:434-439, :65-159)
static int cpu_keep(const phmm_batch *b, CpuBatch *k)
{
    uint64_t read_bytes = 0, hap_bytes = 0;
    for (uint32_t r = 0; r < b->n_reads; ++r) read_bytes = std::max<uint64_t>(read_bytes, (uint64_t)b->read_off[r] + b->read_len[r]);
    for (uint32_t h = 0; h < b->n_haps; ++h) hap_bytes = std::max<uint64_t>(hap_bytes, (uint64_t)b->hap_off[h] + b->hap_len[h]);
    k->bases.assign(b->read_bases, b->read_bases + read_bytes);
    k->qual.assign(b->read_qual, b->read_qual + read_bytes);
    k->roff.assign(b->read_off, b->read_off + b->n_reads);
    k->rlen.assign(b->read_len, b->read_len + b->n_reads);
    k->haps.assign(b->hap_bases, b->hap_bases + hap_bytes);
    k->hoff.assign(b->hap_off, b->hap_off + b->n_haps);
    k->hlen.assign(b->hap_len, b->hap_len + b->n_haps);
    if (b->prep) {
        k->ins.assign(read_bytes, b->prep->gap_open);
        k->gcp.assign(read_bytes, b->prep->gap_cont);
        for (uint32_t r = 0; r < b->n_reads; ++r) {
            const uint32_t o = k->roff[r], n = k->rlen[r];
            rovaca_amd::modify_read_qualities(k->qual.data() + o, n, b->prep->read_mapq[r], b->prep->min_quality_threshold);
            rovaca_amd::apply_pcr_error_model(k->bases.data() + o, k->ins.data() + o, n, b->prep->pcr_model);
        }
        k->del = k->ins;  // pairhmm_internal.cpp:425
    }
    else {
        k->ins.assign(b->read_ins, b->read_ins + read_bytes);
        k->del.assign(b->read_del, b->read_del + read_bytes);
        k->gcp.assign(b->read_gcp, b->read_gcp + read_bytes);
    }
    if (b->pair_read) {
        k->pr.assign(b->pair_read, b->pair_read + b->n_pairs);
        k->ph.assign(b->pair_hap, b->pair_hap + b->n_pairs);
    }
    phmm_batch &v = k->view;
    v = phmm_batch{};
    v.n_reads = b->n_reads;
    v.n_haps = b->n_haps;
    v.n_pairs = b->pair_read ? b->n_pairs : 0;
    v.read_bases = k->bases.data();
    v.read_qual = k->qual.data();
    v.read_ins = k->ins.data();
    v.read_del = k->del.data();
    v.read_gcp = k->gcp.data();
    v.read_off = k->roff.data();
    v.read_len = k->rlen.data();
    v.hap_bases = k->haps.data();
    v.hap_off = k->hoff.data();
    v.hap_len = k->hlen.data();
    v.pair_read = b->pair_read ? k->pr.data() : nullptr;
    v.pair_hap = b->pair_read ? k->ph.data() : nullptr;
    return PHMM_OK;
}

int cpu_upload(phmm_context *, const phmm_batch *b, phmm_resident *rb)
{
    uint64_t n = 0;
    int rc = cpu_check_batch(b, &n);
    if (rc != PHMM_OK) return rc;
    delete rb->cpu;
    rb->cpu = new CpuBatch();
    rc = cpu_keep(b, rb->cpu);
    if (rc != PHMM_OK) return rc;
    rb->n_pairs = n;
    rb->n_reads = b->n_reads;
    rb->n_haps = b->n_haps;
    rb->cpu->raw32.assign(n, 0.f);
    rb->cpu->raw64.assign(n, 0.0);
    rb->cpu->rescued.assign(n, 0);
    rb->last_rescued = 0;
    return PHMM_OK;
}

int cpu_sweep(phmm_context *c, phmm_resident *rb)
{
    if (!rb->cpu) return fail(PHMM_E_INVALID, "phmm_sweep: this batch was not uploaded through a CPU context");
    CpuBatch &k = *rb->cpu;
    uint64_t nr = 0, cells = 0;
    if (phmm::cpu::compute(k.view, c->cpu_threads, c->cpu_simd, k.raw32.data(), k.raw64.data(), k.rescued.data(), &nr, &cells) != 0)
        return fail(PHMM_E_INVALID, "phmm_sweep: a pair indexes outside the tables");
    rb->last_rescued = (uint32_t)nr;
    rb->cells = cells;
    return PHMM_OK;
}

static int cpu_download(phmm_context *, phmm_resident *rb, float *out_raw_f32, double *out_log10, phmm_stats *stats)
{
    if (!rb->cpu) return fail(PHMM_E_INVALID, "phmm_download: this batch was not uploaded through a CPU context");
    const CpuBatch &k = *rb->cpu;
    const uint64_t n = rb->n_pairs;
    if (out_raw_f32 && n) memcpy(out_raw_f32, k.raw32.data(), n * sizeof(float));
    if (out_log10)
        for (uint64_t p = 0; p < n; ++p) {  // pairhmm_internal.cpp:577-585
            const float x = k.raw32[p];
            out_log10[p] = x < PHMM_MIN_ACCEPTED_F32 ? log10(k.raw64[p]) - PHMM_LOG10_INITIAL_CONSTANT_F64
                                                     : (double)(log10f(x) - PHMM_LOG10_INITIAL_CONSTANT_F32);
        }
    if (stats) {
        *stats = phmm_stats{};
        stats->cells = rb->cells;
        stats->n_rescued = rb->last_rescued;
        stats->kernel_ms = rb->last_kernel_ms;
        stats->n_launches = 0;
    }
    return PHMM_OK;
}

int phmm_upload(phmm_context *c, const phmm_batch *b, phmm_resident **out)
{
    if (!c || !b || !out) return fail(PHMM_E_INVALID, "phmm_upload: null argument");
    *out = nullptr;
    phmm_resident *rb = new (std::nothrow) phmm_resident();
    if (!rb) return fail(PHMM_E_NOMEM, "phmm_upload: out of host memory");
    const int rc = c->cpu ? cpu_upload(c, b, rb) : upload_into(c, b, rb, UploadOpts());
    if (rc != PHMM_OK) {
        delete rb;
        return rc;
    }
    *out = rb;
    return PHMM_OK;
}

int phmm_sweep(phmm_context *c, phmm_resident *rb)
{
    if (!c || !rb) return fail(PHMM_E_INVALID, "phmm_sweep: null argument");
    if (c->cpu) return cpu_sweep(c, rb);
    HIP_TRY(hipSetDevice(c->device));
    int rc = sweep_fp32(c, rb);
    if (rc != PHMM_OK) return rc;
    return sweep_rescue(c, rb);
}

int phmm_sweep_timed(phmm_context *c, phmm_resident *rb, int iters, double *ms_per_pass, double *fp32_ms_per_pass)
{
    if (!c || !rb || iters < 1 || !ms_per_pass) return fail(PHMM_E_INVALID, "phmm_sweep_timed: bad arguments");
    if (c->cpu) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i) {
            const int rc = cpu_sweep(c, rb);
            if (rc != PHMM_OK) return rc;
        }
        *ms_per_pass = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
        if (fp32_ms_per_pass) *fp32_ms_per_pass = *ms_per_pass;
        rb->last_kernel_ms = *ms_per_pass;
        return PHMM_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    // three events per pass (start, after the fp32 launches, end), recorded without waiting; the host waits once per group
    // of passes, so that the passes run back to back as they do without the timing
    constexpr int kGroup = 32;
    const int pool = 3 * std::min(iters, kGroup);
    while ((int)c->timing.size() < pool) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        c->timing.push_back(e);
    }
    double total = 0.0, fp32 = 0.0;
    for (int base = 0; base < iters; base += kGroup) {
        const int m = std::min(kGroup, iters - base);
        for (int i = 0; i < m; ++i) {
            HIP_TRY(hipEventRecord(c->timing[3 * i], c->stream));
            int rc = sweep_fp32(c, rb);
            if (rc != PHMM_OK) return rc;
            HIP_TRY(hipEventRecord(c->timing[3 * i + 1], c->stream));
            rc = sweep_rescue(c, rb);
            if (rc != PHMM_OK) return rc;
            HIP_TRY(hipEventRecord(c->timing[3 * i + 2], c->stream));
        }
        HIP_TRY(hipEventSynchronize(c->timing[3 * (m - 1) + 2]));
        for (int i = 0; i < m; ++i) {
            float ms_a = 0.f, ms_b = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms_a, c->timing[3 * i], c->timing[3 * i + 1]));
            HIP_TRY(hipEventElapsedTime(&ms_b, c->timing[3 * i + 1], c->timing[3 * i + 2]));
            fp32 += ms_a;
            total += (double)ms_a + ms_b;
        }
    }
    *ms_per_pass = total / iters;
    if (fp32_ms_per_pass) *fp32_ms_per_pass = fp32 / iters;
    rb->last_kernel_ms = total / iters;
    return PHMM_OK;
}

int phmm_download(phmm_context *c, phmm_resident *rb, float *out_raw_f32, double *out_log10, phmm_stats *stats)
{
    if (!c || !rb) return fail(PHMM_E_INVALID, "phmm_download: null argument");
    if (c->cpu) return cpu_download(c, rb, out_raw_f32, out_log10, stats);
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t n = rb->n_pairs;
    // (a range that ran fp64 first has no count of underflowed pairs on the device: they are counted here, on the fp32 sums)
    bool any_f64_first = false;
    for (int qi = 1; qi <= 4; ++qi) any_f64_first = any_f64_first || rb->queue_f64_first[qi];
    const bool want = ((out_raw_f32 || out_log10) && n) || (any_f64_first && n);
    // results come back through pinned staging: [n floats, padding, rescue count][n doubles when anything was rescued]
    const size_t count_at = (n + 63) & ~size_t(63);
    const size_t off_cnt = count_at * sizeof(float);
    const size_t off64 = off_cnt + 256;
    if (n) {
        HIP_TRY(rb->out_stage.ensure(off64 + n * sizeof(double)));
        if (want)
            HIP_TRY(hipMemcpyAsync(rb->out_stage.p, rb->raw32.p, off_cnt + kQueues * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        else
            HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(rb->out_stage.p) + off_cnt, rb->rescue_count.p, kQueues * sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, c->stream));
    }
    PHASE(4);
    HIP_TRY(hipStreamSynchronize(c->stream));  // (polling hipStreamQuery instead gains nothing: measured)
    PHASE(5);
    if (n) {
        const uint32_t *counts = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(rb->out_stage.p) + off_cnt);
        rb->last_rescued = 0;
        for (int qi = 0; qi < kQueues; ++qi)
            if (!rb->queue_f64_first[qi]) rb->last_rescued += counts[qi];
        if (any_f64_first) {
            const float *rp = static_cast<const float *>(rb->out_stage.p);
            const uint32_t *order_h = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(rb->stage.p) + rb->plan.o_order);
            for (const Bin &bin : rb->bins)
                if (rb->queue_f64_first[bin.queue])
                    for (uint32_t i = 0; i < bin.count; ++i) rb->last_rescued += rp[order_h[bin.first + i]] < PHMM_MIN_ACCEPTED_F32 ? 1u : 0u;
        }
    }
    const bool want64 = out_log10 && rb->last_rescued && n;
    if (want64 && rb->rescue_deferred) {  // phmm_compute left the fp64 re-run to us: some pairs did underflow
        rb->rescue_deferred = false;
        const int rc = sweep_rescue(c, rb);
        if (rc != PHMM_OK) return rc;
        HIP_TRY(hipEventRecord(c->ev1, c->stream));  // (the call's device span ends here)
    }
    if (want64) {
        HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(rb->out_stage.p) + off64, rb->raw64.p, n * sizeof(double),
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (want) {
        const float *rp = static_cast<const float *>(rb->out_stage.p);
        const double *r64 = reinterpret_cast<const double *>(static_cast<const uint8_t *>(rb->out_stage.p) + off64);
        if (out_raw_f32) memcpy(out_raw_f32, rp, n * sizeof(float));
        if (out_log10) {
            // pairhmm_internal.cpp:577-585 -- libm on the host, as the reference does, so the last ulp agrees
            auto finalize = [&](uint64_t lo, uint64_t hi, unsigned) {
                for (uint64_t p = lo; p < hi; ++p) {
                    const float x = rp[p];
                    if (x < PHMM_MIN_ACCEPTED_F32)
                        out_log10[p] = log10(want64 ? r64[p] : 0.0) - PHMM_LOG10_INITIAL_CONSTANT_F64;
                    else
                        out_log10[p] = (double)(log10f(x) - PHMM_LOG10_INITIAL_CONSTANT_F32);
                }
            };
            if (n >= kParallelMinPairs)
                parallel_ranges(n, host_threads(), finalize);
            else
                finalize(0, n, 0);
        }
    }
    PHASE(6);
    if (stats) {
        stats->cells = rb->cells;
        stats->n_rescued = rb->last_rescued;
        stats->kernel_ms = rb->last_kernel_ms;
        stats->n_launches = rb->last_fp32_launches ? rb->last_fp32_launches : (uint32_t)rb->bins.size();
        stats->max_rows_per_lane = rb->bins.empty() ? 0 : (uint32_t)rb->bins.back().rows_per_lane;
    }
    return PHMM_OK;
}

int phmm_download_read_table(phmm_context *c, phmm_resident *rb, uint8_t *qual, uint8_t *ins, uint8_t *del, uint8_t *gcp)
{
    if (!c || !rb) return fail(PHMM_E_INVALID, "phmm_download_read_table: null argument");
    if (c->cpu) {
        if (!rb->cpu) return fail(PHMM_E_INVALID, "phmm_download_read_table: this batch was not uploaded through a CPU context");
        const CpuBatch &k = *rb->cpu;
        if (qual) memcpy(qual, k.qual.data(), k.qual.size());
        if (ins) memcpy(ins, k.ins.data(), k.ins.size());
        if (del) memcpy(del, k.del.data(), k.del.size());
        if (gcp) memcpy(gcp, k.gcp.data(), k.gcp.size());
        return PHMM_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = rb->read_qual.n;
    const uint8_t *dp = rb->read_del.p;
    if (qual && n) HIP_TRY(hipMemcpyAsync(qual, rb->read_qual.p, n, hipMemcpyDeviceToHost, c->stream));
    if (ins && n) HIP_TRY(hipMemcpyAsync(ins, rb->read_ins.p, n, hipMemcpyDeviceToHost, c->stream));
    if (del && n) HIP_TRY(hipMemcpyAsync(del, dp, n, hipMemcpyDeviceToHost, c->stream));
    if (gcp && n) HIP_TRY(hipMemcpyAsync(gcp, rb->read_gcp.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PHMM_OK;
}

int phmm_compute(phmm_context *c, const phmm_batch *b, float *out_raw_f32, double *out_log10, phmm_stats *stats)
{
    if (!c || !b) return fail(PHMM_E_INVALID, "phmm_compute: null argument");
    phmm_detail::HookScope hooks;
    const auto t0 = std::chrono::steady_clock::now();
    if (c->cpu) {
        if (!c->scratch) c->scratch = new phmm_resident();
        int rc = cpu_upload(c, b, c->scratch);
        if (rc == PHMM_OK) rc = cpu_sweep(c, c->scratch);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        c->scratch->last_kernel_ms = ms;
        if (rc == PHMM_OK) rc = cpu_download(c, c->scratch, out_raw_f32, out_log10, stats);
        if (rc == PHMM_OK && stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    }
    {
        std::vector<ChunkWindow> plan;
        const uint64_t n_all = b->pair_read ? b->n_pairs : (uint64_t)b->n_reads * b->n_haps;
        // helper threads of the chunk pipeline (also used by the plan itself), made on a context's first chunked call: all the cores for
        // a batch of half a million pairs or more, three helpers for a mid-size one (a context per worker thread must not bring a
        // dozen threads each along for the odd large region)
        const bool chunkable = n_all >= chunked_min_pairs();
        if (chunkable) {
            const unsigned all = std::max(1u, host_threads()) - 1u, want = n_all >= (1u << 19) ? all : std::min(3u, all);
            // a large batch after mid-size ones: the pool grows once.  The REQUEST is remembered, not the threads it yielded: in a process
            // near its thread limit a pool comes up short for good, and comparing helpers() rebuilt it -- joining and respawning
            // threads -- on every call
            if (c->pack_pool && c->pack_pool_asked < want) {
                delete c->pack_pool;
                c->pack_pool = nullptr;
            }
            if (!c->pack_pool) {
                c->pack_pool = new phmm_detail::HostPool(want);
                c->pack_pool_asked = want;
            }
        }
        if (chunkable && !c->fin_pool) c->fin_pool = new phmm_detail::HostPool(3);
        phmm_detail::PoolScope pool(chunkable ? c->pack_pool : nullptr);
        const bool tables_ok = b->read_bases && b->read_qual && b->read_off && b->read_len && b->hap_bases && b->hap_off && b->hap_len &&
                               ((b->pair_read == nullptr) == (b->pair_hap == nullptr)) && (b->prep ? b->prep->read_mapq != nullptr : true);
        if (tables_ok && impl::plan_chunks(b, plan)) {
            const int rc = impl::compute_chunked(c, b, plan, out_raw_f32, out_log10, stats);
            if (rc != kChunkedUnavailable) {
                if (rc == PHMM_OK && stats)
                    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                return rc;
            }
        }
    }
    if (!c->scratch) {
        c->scratch = new (std::nothrow) phmm_resident();
        if (!c->scratch) return fail(PHMM_E_NOMEM, "phmm_compute: out of host memory");
    }
    phmm_resident *rb = c->scratch;
    PhaseClock clock;
    struct PhaseScope  // the thread-local pointer never outlives the clock it points to, whichever way the call returns
    {
        explicit PhaseScope(PhaseClock *p) { g_phase = p; }
        ~PhaseScope() { g_phase = nullptr; }
    } phase_scope(clock.on ? &clock : nullptr);
    int rc = upload_into(c, b, rb, UploadOpts());
    if (rc != PHMM_OK) return rc;
    // one pass; the events are read only after the download's synchronisation (no extra round trip for the timing)
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    rc = sweep_fp32(c, rb);
    if (rc != PHMM_OK) return rc;
    // The fp64 re-run: the download below sees how many pairs the fp32 kernels queued and launches it only if there are any -- the
    // nearly always empty launch is off the call's critical path, as in the chunk pipeline and the stages.  (A range of long pairs
    // that runs fp64 FIRST has both its kernels in sweep_rescue: no deferring then.)
    bool rescue_now = out_log10 == nullptr;  // (nobody to fetch the fp64 sums for: keep the device state of a whole pass anyway)
    for (int qi = 1; qi <= 4; ++qi) rescue_now = rescue_now || rb->queue_f64_first[qi];
    if (rescue_now) {
        rc = sweep_rescue(c, rb);
        if (rc != PHMM_OK) return rc;
    }
    rb->rescue_deferred = !rescue_now;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    rc = impl::phmm_download(c, rb, out_raw_f32, out_log10, stats);
    rb->rescue_deferred = false;
    if (rc == PHMM_OK && stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        rb->last_kernel_ms = ms;
        stats->kernel_ms = ms;
    }
    if (rc == PHMM_OK && stats)
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (g_phase) {
        PHASE(7);
        fprintf(stderr, "phmm phases us: validate %.1f pack %.1f bin %.1f enqueue-upload %.1f enqueue-sweep %.1f wait %.1f finalize %.1f stats %.1f\n",
                clock.us[0], clock.us[1], clock.us[2], clock.us[3], clock.us[4], clock.us[5], clock.us[6], clock.us[7]);
    }
    return rc;
}

}  // namespace impl

namespace impl
{
// ---- allele marginalisation ----
int phmm_marginalize(phmm_context *c, const double *old, uint32_t n_old, uint32_t stride, uint32_t evidence_count,
                     const int32_t *old2new, uint32_t n_new, double *out, double *device_ms)
{
    if (!c || !old2new || (n_old && !old) || (n_new && evidence_count && !out) || evidence_count > stride)
        return fail(PHMM_E_INVALID, "phmm_marginalize: bad arguments");
    for (uint32_t a = 0; a < n_old; ++a)
        if (old2new[a] != -1 && (old2new[a] < 0 || (uint32_t)old2new[a] >= n_new))
            return fail(PHMM_E_INVALID, "phmm_marginalize: old allele %u maps to %d, outside [0, %u)", a, old2new[a], n_new);
    if (device_ms) *device_ms = 0.0;
    if (n_new == 0 || evidence_count == 0) return PHMM_OK;
    if (c->cpu) {
        rovaca_amd::marginal_likelihoods_direct(old, n_old, stride, evidence_count, old2new, n_new, out);
        return PHMM_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t in_b = (size_t)n_old * stride * sizeof(double), map_b = ((size_t)n_old * sizeof(int32_t) + 7) & ~size_t(7);
    const size_t out_b = (size_t)n_new * evidence_count * sizeof(double);
    HIP_TRY(c->marg_dev.alloc(in_b + map_b + out_b + 8));
    HIP_TRY(c->marg_stage.ensure(in_b + map_b + out_b + 8));
    uint8_t *hs = static_cast<uint8_t *>(c->marg_stage.p);
    if (in_b) memcpy(hs, old, in_b);
    if (n_old) memcpy(hs + in_b, old2new, (size_t)n_old * sizeof(int32_t));
    if (in_b + map_b) HIP_TRY(hipMemcpyAsync(c->marg_dev.p, hs, in_b + map_b, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(phmm::launch_marginalize(reinterpret_cast<const double *>(c->marg_dev.p), n_old, stride, evidence_count,
                                     reinterpret_cast<const int32_t *>(c->marg_dev.p + in_b), n_new,
                                     reinterpret_cast<double *>(c->marg_dev.p + in_b + map_b), c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipMemcpyAsync(hs + in_b + map_b, c->marg_dev.p + in_b + map_b, out_b, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out, hs + in_b + map_b, out_b);
    if (device_ms) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        *device_ms = ms;
    }
    return PHMM_OK;
}

}  // namespace impl

extern "C" {

int phmm_abi_version(void) { return PHMM_ABI_VERSION; }
const char *phmm_last_error(void) { return g_err; }
int phmm_device_count(void) { return impl::phmm_device_count(); }
int phmm_get_tables(float *a, float *b, double *c, double *d)
{
    return guarded("phmm_get_tables", [&] { return impl::phmm_get_tables(a, b, c, d); });
}
int phmm_create(int device_id, void *stream, phmm_context **out)
{
    return guarded("phmm_create", [&] { return impl::phmm_create(device_id, stream, out); });
}
int phmm_create_cpu(int n_threads, int allow_simd, phmm_context **out)
{
    return guarded("phmm_create_cpu", [&] { return impl::phmm_create_cpu(n_threads, allow_simd, out); });
}
const char *phmm_backend_name(const phmm_context *c)
{
    if (!c) return "";
    return c->cpu ? phmm::cpu::kernel_name(c->cpu_simd) : "hip gfx950";
}
void phmm_destroy(phmm_context *c)
{
    (void)guarded("phmm_destroy", [&] { impl::phmm_destroy(c); return PHMM_OK; });
}
int phmm_synchronize(phmm_context *c) { return guarded("phmm_synchronize", [&] { return impl::phmm_synchronize(c); }); }
void phmm_release(phmm_context *c, phmm_resident *rb)
{
    (void)guarded("phmm_release", [&] { impl::phmm_release(c, rb); return PHMM_OK; });
}
int phmm_upload(phmm_context *c, const phmm_batch *b, phmm_resident **out)
{
    return guarded("phmm_upload", [&] { return impl::phmm_upload(c, b, out); });
}
int phmm_sweep(phmm_context *c, phmm_resident *rb) { return guarded("phmm_sweep", [&] { return impl::phmm_sweep(c, rb); }); }
int phmm_sweep_timed(phmm_context *c, phmm_resident *rb, int iters, double *ms, double *ms32)
{
    return guarded("phmm_sweep_timed", [&] { return impl::phmm_sweep_timed(c, rb, iters, ms, ms32); });
}
int phmm_download(phmm_context *c, phmm_resident *rb, float *raw, double *l10, phmm_stats *st)
{
    return guarded("phmm_download", [&] { return impl::phmm_download(c, rb, raw, l10, st); });
}
int phmm_download_read_table(phmm_context *c, phmm_resident *rb, uint8_t *q, uint8_t *i, uint8_t *d, uint8_t *g)
{
    return guarded("phmm_download_read_table", [&] { return impl::phmm_download_read_table(c, rb, q, i, d, g); });
}
int phmm_compute(phmm_context *c, const phmm_batch *b, float *raw, double *l10, phmm_stats *st)
{
    return guarded("phmm_compute", [&] { return impl::phmm_compute(c, b, raw, l10, st); });
}

int phmm_compute_regions(phmm_context *c, const phmm_batch *t, const phmm_region *regions, uint32_t n_regions, float *raw, double *l10, phmm_stats *st)
{
    return guarded("phmm_compute_regions", [&] { return impl::phmm_compute_regions(c, t, regions, n_regions, raw, l10, st); });
}
int phmm_stage_create(phmm_context *c, const phmm_stage_caps *caps, phmm_stage **out)
{
    return guarded("phmm_stage_create", [&] { return impl::phmm_stage_create(c, caps, out); });
}
void phmm_stage_destroy(phmm_context *c, phmm_stage *st)
{
    (void)guarded("phmm_stage_destroy", [&] { impl::phmm_stage_destroy(c, st); return PHMM_OK; });
}
int phmm_stage_get_view(const phmm_stage *st, phmm_stage_view *view, phmm_stage_caps *caps)
{
    return guarded("phmm_stage_get_view", [&] { return impl::phmm_stage_get_view(st, view, caps); });
}
int phmm_stage_submit(phmm_context *c, phmm_stage *st, uint32_t n_regions, uint32_t n_reads, uint32_t n_haps, const phmm_stage_params *prm)
{
    return guarded("phmm_stage_submit", [&] { return impl::phmm_stage_submit(c, st, n_regions, n_reads, n_haps, prm); });
}
int phmm_stage_wait(phmm_context *c, phmm_stage *st, phmm_stats *stats) { return guarded("phmm_stage_wait", [&] { return impl::phmm_stage_wait(c, st, stats); }); }
uint64_t phmm_stage_pair_first(const phmm_stage *st, uint32_t region) { return impl::phmm_stage_pair_first(st, region); }
int phmm_stage_log10(const phmm_stage *st, uint64_t first_pair, uint64_t n, double *out, float *raw, uint64_t *n_rescued)
{
    return guarded("phmm_stage_log10", [&] { return impl::phmm_stage_log10(st, first_pair, n, out, raw, n_rescued); });
}

// the binning alone (host only): lets the launch plan be checked on a machine without GPUs
static int report_bins(const std::vector<Bin> &bins, int32_t *bin_rows, uint32_t *bin_first, uint32_t *bin_count, int max_bins)
{
    for (size_t i = 0; i < bins.size() && (int)i < max_bins; ++i) {
        if (bin_rows) bin_rows[i] = bins[i].rows_per_lane;
        if (bin_first) bin_first[i] = bins[i].first;
        if (bin_count) bin_count[i] = bins[i].count;
    }
    return (int)bins.size();
}
int phmm_plan_bins(const phmm_batch *b, uint32_t *order, int32_t *bin_rows, uint32_t *bin_first, uint32_t *bin_count, int max_bins)
{
    int n_bins = 0;
    const int rc = guarded("phmm_plan_bins", [&]() -> int {
        if (!b || (b->pair_read == nullptr) != (b->pair_hap == nullptr)) return fail(PHMM_E_INVALID, "phmm_plan_bins: bad arguments");
        const bool cross = b->pair_read == nullptr;
        const uint64_t n = cross ? (uint64_t)b->n_reads * b->n_haps : b->n_pairs;
        if (n > 0xFFFFFFF0ull || (n && (!b->read_len || !b->hap_len))) return fail(PHMM_E_INVALID, "phmm_plan_bins: bad batch");
        std::vector<uint32_t> pr(n), ph(n), ord(order ? 0 : n);
        phmm_detail::BinScratch scratch;
        std::vector<Bin> bins;
        uint64_t cells = 0;
        const int r = phmm_detail::bin_pairs(b, cross, n, 0, 0, pr.data(), ph.data(), order ? order : ord.data(), scratch, bins, &cells);
        if (r != PHMM_OK) return r;
        n_bins = report_bins(bins, bin_rows, bin_first, bin_count, max_bins);
        return PHMM_OK;
    });
    return rc == PHMM_OK ? n_bins : -rc;
}
int phmm_plan_region_bins(const phmm_batch *t, const phmm_region *regions, uint32_t n_regions, uint32_t *slot_first, uint32_t *pair_first,
                          int32_t *bin_rows, uint32_t *bin_first, uint32_t *bin_count, int max_bins)
{
    int n_bins = 0;
    const int rc = guarded("phmm_plan_region_bins", [&]() -> int {
        if (!t || (n_regions && !regions) || !slot_first || !pair_first) return fail(PHMM_E_INVALID, "phmm_plan_region_bins: null argument");
        std::vector<Bin> bins;
        uint64_t cells = 0, n_pairs = 0;
        const int r = phmm_detail::bin_regions(regions, n_regions, t->read_len, t->n_reads, t->hap_len, t->n_haps, slot_first, pair_first, bins, &cells, &n_pairs);
        if (r != PHMM_OK) return r;
        n_bins = report_bins(bins, bin_rows, bin_first, bin_count, max_bins);
        return PHMM_OK;
    });
    return rc == PHMM_OK ? n_bins : -rc;
}

int phmm_marginalize(phmm_context *c, const double *old, uint32_t n_old, uint32_t stride, uint32_t evidence_count, const int32_t *old2new,
                     uint32_t n_new, double *out, double *device_ms)
{
    return guarded("phmm_marginalize", [&] { return impl::phmm_marginalize(c, old, n_old, stride, evidence_count, old2new, n_new, out, device_ms); });
}

}  // extern "C"
