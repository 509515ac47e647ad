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
int ensure_side_streams(phmm_context *c)
{
    if (c->cpu || c->side[0]) return PHMM_OK;
    for (int i = 0; i < phmm_context::kSideStreams; ++i) {
        HIP_TRY(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming));
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
static void harvest_history(phmm_context *c, phmm_resident *rb);

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
        harvest_history(c, rb);  // (the batch this slot held before: what its last pass saw goes into the context's history)
        for (int qi = 0; qi < kQueues; ++qi) rb->f64_share_known[qi] = rb->queue_f64_first[qi] = false;  // a new batch: the context's history decides its first pass
        rb->hist_pending = false;
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

// fp64 first (sweep_fp32): what the batch's last finished pass observed on its striped ranges, if the counters are back
static void harvest_history(phmm_context *c, phmm_resident *rb)
{
    if (!rb->hist_pending || hipEventQuery(rb->hist_ready) != hipSuccess) return;
    const uint32_t *cnt = static_cast<const uint32_t *>(rb->hist_host.p);
    uint64_t all = 0, all_by_f64 = 0;
    for (int qi = 1; qi <= 4; ++qi) {
        if (!rb->hist_cap[qi]) continue;
        // fp32 first: the counter is the number of underflowed pairs; fp64 first: the number sent on to the fp32 kernel
        const uint32_t by_f64 = rb->hist_f64_first[qi] ? rb->hist_cap[qi] - std::min(cnt[qi], rb->hist_cap[qi]) : std::min(cnt[qi], rb->hist_cap[qi]);
        rb->f64_share[qi] = (float)by_f64 / (float)rb->hist_cap[qi];
        rb->f64_share_known[qi] = true;
        all += rb->hist_cap[qi];
        all_by_f64 += by_f64;
    }
    if (all) {
        c->f64_share = (float)all_by_f64 / (float)all;
        c->f64_share_known = true;
    }
    rb->hist_pending = false;
}

int sweep_fp32(phmm_context *c, phmm_resident *rb)
{
    if (rb->n_pairs && !rb->queue_is_empty) HIP_TRY(hipMemsetAsync(rb->rescue_count.p, 0, kQueues * sizeof(uint32_t), c->stream));
    rb->queue_is_empty = false;
    // striped launches run side by side on different streams: each gets its own slice of the carry buffer, sized up front
    using StripedPlan = phmm_resident::StripedLaunchPlan;
    std::vector<StripedPlan> &plans = rb->fp32_plans;  // (kept: an fp64-first range launches its fp32 kernel from sweep_rescue)
    plans.assign(rb->bins.size(), StripedPlan());
    size_t carry_total = 0;
    for (size_t i = 0; i < rb->bins.size(); ++i) {
        const Bin &bin = rb->bins[i];
        if (bin.rows_per_lane > 0) continue;
        StripedPlan &pl = plans[i];
        pl.sh = pick_team_shape(bin.count, bin.max_read_len, bin.max_hap_len, phmm::kStripedRowsF32);
        const uint32_t per_wave = 64u / (uint32_t)pl.sh.lanes;
        pl.teams = std::min<uint32_t>((bin.count + per_wave - 1) / per_wave, kStripedMaxTeams / pl.sh.waves);
        pl.carry_at = carry_total;
        carry_total += (phmm::team_carry_values(pl.teams, pl.sh.waves, bin.max_hap_len + 1, pl.sh.lanes) + 63) & ~size_t(63);
    }
    HIP_TRY(rb->carry32.alloc(carry_total));
    // The rescue queue of a striped range is a list in processing order with holes (KernelArgs::sparse_order) when the team / band
    // kernels feed it (the one-pass kernels that take reads of 257-512 bases append): filled with "no pair" before the pass.
    // PHMM_RESCUE_APPEND=1 (test hook): the appended list of rounds 1-3, whose order is the one the fp32 teams finish in.
    const bool in_place = phmm_hook("PHMM_RESCUE_APPEND") == nullptr;
    auto striped_kernel_bin = [&](const Bin &bin) {
        return bin.rows_per_lane == 0 && !(bin.max_read_len <= 64u * phmm::kMaxRowsPerLaneWide && bin.max_hap_len <= kSinglePassMaxHap &&
                                           !phmm_hook("PHMM_NO_WIDE_SINGLE_PASS"));
    };
    for (int qi = 0; qi < kQueues; ++qi) rb->queue_sparse[qi] = false;
    for (const Bin &bin : rb->bins)
        if (bin.queue >= 1 && bin.queue <= 4 && in_place && striped_kernel_bin(bin)) rb->queue_sparse[bin.queue] = true;
    for (const Bin &bin : rb->bins)
        if (bin.queue >= 1 && bin.queue <= 4 && !striped_kernel_bin(bin)) rb->queue_sparse[bin.queue] = false;  // (a queue is fed one way)
    // ---- fp64 first ----
    // A striped range whose pairs all underflow fp32 -- reads of thousands of bases with ordinary error rates: config 5, every pair --
    // pays for an fp32 pass that decides nothing (config 5: 2.0 of 11.1 ms).  When the last pass over such a range has seen at least
    // kF64FirstShare of its pairs decided by fp64, this pass runs fp64 FIRST, over every pair of the range, and the fp32 kernel only for
    // the pairs whose fp64 sum says fp32 might NOT underflow.  The decision per pair is still the reference's, "fp32 sum < 1e-28f":
    //   * every quantity of the recurrence is non-negative, so the fp32 evaluation can exceed the exact value of the same expression
    //     only by rounding: (1 + 2^-24) per operation along the longest dependence chain, at most 4 (R + H) operations -- 1.06 for
    //     R + H = 2^18, the longest pair that goes this way; flush-to-zero only lowers it.  The fp32 tables differ from the fp64 ones
    //     by at most 1e-6 per factor (pairhmm_tables.cpp: the same Jacobian-log index in both types, one float rounding of the sum and
    //     one of the entry; powf against pow), 3 factors per cell of a path: (1 + 1e-6)^(3 * 2^18) = 2.2.  The fp64 sum carries 1e-12.
    //   * so fp32_sum <= 2^-900 * fp64_sum * 2.4 (the scales are 2^120 and 2^1020), and a pair with
    //     2^-900 * fp64_sum < 1e-28 * 2^-20 underflows fp32 with five decades to spare: its answer is the fp64 one, and its fp32 sum is
    //     REPORTED as 0 (the reference discards it: pairhmm_internal.cpp:577-585).  Every other pair gets its fp32 sum from the fp32
    //     kernel and is decided by it, exactly as in the fp32-first order.
    // Same answers either way (tests: both orders forced, pairs on both sides of the threshold and in the margin); what changes is
    // which kernel runs needlessly when the guess is wrong -- fp64-first on a range whose pairs do not underflow costs the fp64 pass,
    // 4-5x an fp32 one, once: the next pass sees the share and goes back.  PHMM_F64_FIRST=0 / 1 (test hook): never / always.
    // Only where the sums land in device memory (resident batches and the plain one-shot call), since the skipped pairs' fp32 sums
    // are zeroed by a memset.
    constexpr float kF64FirstShare = 0.75f;
    {
        // what the last finished pass observed (the counters come back behind the pass, asynchronously)
        harvest_history(c, rb);
        int forced = -1;
        if (const char *f = phmm_hook("PHMM_F64_FIRST")) forced = atoi(f) ? 1 : 0;
        for (int qi = 0; qi < kQueues; ++qi) rb->queue_f64_first[qi] = false;
        for (const Bin &bin : rb->bins) {
            const int qi = bin.queue;
            if (!(qi >= 1 && qi <= 4) || !rb->queue_sparse[qi] || rb->out32_host) continue;
            if ((uint64_t)bin.max_read_len + bin.max_hap_len > (1u << 18)) continue;  // (the bound above is argued with room to spare up to here)
            const bool known = rb->f64_share_known[qi] || c->f64_share_known;
            const float share = rb->f64_share_known[qi] ? rb->f64_share[qi] : c->f64_share;
            rb->queue_f64_first[qi] = forced >= 0 ? forced == 1 : (known && share >= kF64FirstShare);
        }
    }
    bool any_f64_first = false;
    for (const Bin &bin : rb->bins) {
        if (!rb->queue_sparse[bin.queue]) continue;
        HIP_TRY(hipMemsetAsync(rb->rescue_list.p + bin.first, 0xFF, (size_t)bin.count * sizeof(uint32_t), c->stream));
        any_f64_first = any_f64_first || rb->queue_f64_first[bin.queue];
    }
    if (any_f64_first) HIP_TRY(hipMemsetAsync(rb->raw32.p, 0, (size_t)rb->n_pairs * sizeof(float), c->stream));  // (before any launch of the pass)
    // The launches of the pass.  A striped bin is one launch.  Single-pass bins are one launch each too -- unless the batch takes ragged
    // launches (the Rovaca case; pairhmm_kernels_ragged.hip): then a run of single-pass bins is cut into the K ranges that share a
    // kernel and every range with more than one bin is ONE launch whose workgroups pick their class's K, longest rows first.
    struct FpLaunch
    {
        size_t first, n;  // bins [first, first + n)
        int lanes;        // single-pass launches: lanes per pair (16 / 64); 0: not decided here (striped, or the per-bin rule); -1: uniform-budget ragged kernel (16 or 32 per class)
    };
    uint32_t wide_max = kWideMaxPairs;
    if (const char *f = phmm_hook("PHMM_WIDE_MAX_PAIRS")) wide_max = (uint32_t)std::max(0, atoi(f));  // tuning
    int forced_lanes = 0;
    if (const char *f = phmm_hook("PHMM_FORCE_LANES")) {  // tests
        const int l = atoi(f);
        if (l == 16 || l == 64) forced_lanes = l;
    }
    uint32_t xcd_min = kXcdLocalMinPairs;
    if (const char *f = phmm_hook("PHMM_XCD_LOCAL_MIN_PAIRS")) xcd_min = (uint32_t)std::max(1, atoi(f));  // tests: small launches too
    const bool xcd_local = !phmm_hook("PHMM_NO_XCD_LOCAL_ORDER");
    auto wide_rows = [](const Bin &b) { return (int)((b.max_read_len + 63) / 64); };  // rows per lane at one pair per wave
    std::vector<FpLaunch> launches;
    for (size_t i = 0; i < rb->bins.size();) {
        if (rb->queue_f64_first[rb->bins[i].queue]) {  // (sweep_rescue launches both of its kernels)
            ++i;
            continue;
        }
        if (!(rb->ragged && rb->bins[i].rows_per_lane > 0)) {
            launches.push_back(FpLaunch{i, 1, 0});
            ++i;
            continue;
        }
        size_t j = i;
        uint64_t total = 0;
        for (; j < rb->bins.size() && rb->bins[j].rows_per_lane > 0; ++j) total += rb->bins[j].count;
        // few pairs in the whole run: one pair per wave (see the one-launch-per-bin rule below)
        const int lanes = forced_lanes ? forced_lanes : total <= wide_max ? 64 : 16;
        // A run that is not huge -- a chunk of a one-shot call, a stage -- is ONE launch of the uniform-budget kernel (lanes = -1 here):
        // every class at four waves per SIMD, the classes whose 16-lane sweep needs more than 128 registers as 32 lanes per pair
        // (pairhmm_kernels_ragged.hip says why: kernels of different budgets side by side cost every chunk 16 %).  From 2^18 pairs on
        // the K-range kernels below are better: their mix is over after the first third of the pass.
        uint64_t uniform_below = kRaggedUniformBelowPairs;
        if (const char *f = phmm_hook("PHMM_RAGGED_UNIFORM_BELOW")) uniform_below = (uint64_t)std::max(0ll, atoll(f));  // tests, tuning
        if (lanes == 16 && total < uniform_below && j - i > 1 && j - i <= (size_t)phmm::kRaggedMaxSegs) {
            launches.push_back(FpLaunch{i, j - i, -1});
            i = j;
            continue;
        }
        for (size_t k = i; k < j;) {
            const int range = phmm::ragged_range_of(lanes == 64 ? wide_rows(rb->bins[k]) : rb->bins[k].rows_per_lane, lanes, rb->uniform_gop);
            size_t e = k + 1;
            while (e < j && e - k < (size_t)phmm::kRaggedMaxSegs &&
                   phmm::ragged_range_of(lanes == 64 ? wide_rows(rb->bins[e]) : rb->bins[e].rows_per_lane, lanes, rb->uniform_gop) == range)
                ++e;
            launches.push_back(FpLaunch{k, e - k, lanes});
            k = e;
        }
        i = j;
    }
    // Longest workgroups first: the striped launches, then the single-pass ones by descending rows per lane.  The launches of a pass run
    // side by side on different streams, and a kernel with a bigger register budget cannot move into the hole a finished wave of a
    // smaller one leaves (168 or 200 VGPRs do not fit into a freed 128): launched last, the big-K kernels of a chunk were starved until the
    // small-K kernel had drained and then ran alone, a round and a half of the chip at two or three waves per SIMD (rocprofv3 time line
    // of config 3's one-shot call, round 4: 1.0-1.6 ms for a quarter of a chunk's cells).  Launched first they are resident before the
    // small-K grid arrives, and that one -- which fits into any hole -- fills the chip behind them and makes up the tail.
    std::reverse(launches.begin(), launches.end());
    rb->last_fp32_launches = (uint32_t)launches.size();
    const bool fan_out = launches.size() > 1;
    if (fan_out) {
        const int rc_side = ensure_side_streams(c);
        if (rc_side != PHMM_OK) return rc_side;
        HIP_TRY(hipEventRecord(c->fork, c->stream));
        for (int i = 0; i < c->n_side; ++i) HIP_TRY(hipStreamWaitEvent(c->side[i], c->fork, 0));
    }
    int nth = 0;
    for (const FpLaunch &L : launches) {
        const size_t i = L.first;
        const Bin &bin = rb->bins[i];
        const int lane = fan_out ? nth++ % (c->n_side + 1) : 0;
        hipStream_t stream = lane == 0 ? c->stream : c->side[lane - 1];
        phmm::KernelArgs a{};
        a.read_bases = rb->read_bases.p;
        a.read_qual = rb->read_qual.p;
        a.read_ins = rb->read_ins.p;
        a.read_del = rb->read_del.p;
        a.read_gcp = rb->read_gcp.p;
        a.read_off = rb->read_off.p;
        a.read_len = rb->read_len.p;
        a.hap_len = rb->hap_len.p;
        a.planes = rb->planes.p;
        a.plane_off = rb->plane_off.p;
        a.pair_read = rb->pair_read.p;
        a.pair_hap = rb->pair_hap.p;
        a.order = rb->order.p + bin.first;
        a.n_pairs = bin.count;
        a.lds_words_per_group = bin.lds_words_per_group;
        a.no_early_exit = phmm_hook("PHMM_NO_EARLY_EXIT") ? 1u : 0u;  // tests
        a.ph2pr = c->ph2pr_f32.p;
        a.mm_table = c->mm_f32.p;
        a.out_raw = rb->out32_host ? static_cast<void *>(rb->out32_host) : static_cast<void *>(rb->raw32.p);
        // underflowed pairs go to the queue of the bin's size range (queue 0 starts at 0, a striped range at its own first)
        a.rescue_list = rb->rescue_list.p + (bin.queue == 0 ? 0 : bin.first);
        a.rescue_count = rb->rescue_count.p + bin.queue;
        if (bin.rows_per_lane > 0 && L.n > 1) {
            // a ragged launch over the classes of bins [first, first + n): longest rows first
            phmm::RaggedArgs m{};
            uint32_t wg = 0, lds_words = 0;
            uint64_t total = 0;
            const bool uniform = L.lanes < 0;
            for (size_t q = L.n; q-- > 0;) {
                const Bin &sb = rb->bins[L.first + q];
                phmm::RaggedSeg &sg = m.seg[m.n_seg++];
                const int seg_lanes = uniform ? phmm::ragged_uniform_lanes_of(sb.rows_per_lane, rb->uniform_gop) : L.lanes;
                const uint32_t per_wg = seg_lanes == 64 ? 1u : (uint32_t)phmm::kBlockThreads / (uint32_t)seg_lanes;
                sg.wg_first = wg;
                sg.pos_first = sb.first;
                sg.count = sb.count;
                sg.rows = (uint32_t)(seg_lanes == 64 ? wide_rows(sb) : seg_lanes == 32 ? (int)((sb.max_read_len + 31) / 32) : sb.rows_per_lane);
                sg.lanes = (uint32_t)seg_lanes;
                sg.queue = (uint32_t)sb.queue;
                sg.list_base = sb.queue == 0 ? 0u : sb.first;
                wg += (sb.count + per_wg - 1) / per_wg;
                lds_words = std::max(lds_words, sb.lds_words_per_group);
                total += sb.count;
            }
            m.n_wg = wg;
            a.order = rb->order.p;
            a.n_pairs = 0;
            a.lds_words_per_group = lds_words;
            a.rescue_list = rb->rescue_list.p;
            a.rescue_count = rb->rescue_count.p;
            if (L.lanes != 64 && total >= xcd_min && xcd_local) a.xcd_run_log2 = 6;
            if (uniform)
                HIP_TRY(phmm::launch_systolic_ragged_uniform_f32(rb->uniform_gop, a, m, stream));
            else
                HIP_TRY(phmm::launch_systolic_ragged_f32(L.lanes, rb->uniform_gop, a, m, stream));
        }
        else if (bin.rows_per_lane > 0) {
            // Few pairs: one pair per wave.  Four per wave would occupy count / 4 of the chip's 1024 SIMDs with one slow wave
            // each; 64 lanes per pair makes the sweep of a pair about half as long (a quarter of the rows per lane, 48 more
            // steps) at 1.8x the instructions, which is free while SIMDs idle.  Measured: a 432-pair region 85 -> 46 us, 1728
            // pairs 86 -> 68 us, no gain from 4000 pairs on, and a loss when several submissions share the chip, hence the
            // low threshold.
            const int lanes = L.lanes ? L.lanes : forced_lanes ? forced_lanes : bin.count <= wide_max ? 64 : 16;
            const int k = lanes == 64 ? wide_rows(bin) : bin.rows_per_lane;
            // big four-pairs-per-wave launches: runs of 64 workgroups (1 024 pairs, a region or so) per XCD, see KernelArgs
            if (lanes == 16 && bin.count >= xcd_min && xcd_local) a.xcd_run_log2 = 6;
            HIP_TRY(phmm::launch_systolic_f32(k, lanes, rb->uniform_gcp, a, stream, rb->uniform_gop));
            a.xcd_run_log2 = 0;
        }
        else if (bin.max_read_len <= 64u * phmm::kMaxRowsPerLaneWide && bin.max_hap_len <= kSinglePassMaxHap &&
                 !phmm_hook("PHMM_NO_WIDE_SINGLE_PASS")) {
            // reads of 257..512 bases: still one pass, a whole wave per pair (no stripes, no carry): measured 20 % faster than
            // the team kernel on 400 x 600
            HIP_TRY(phmm::launch_systolic_f32((int)((bin.max_read_len + 63) / 64), 64, rb->uniform_gcp, a, stream, rb->uniform_gop));
        }
        else {
            const StripedPlan &pl = plans[i];
            a.sparse_order = rb->queue_sparse[bin.queue] ? 1u : 0u;
            HIP_TRY(phmm::launch_striped_f32(rb->uniform_gcp, pl.sh.waves, pl.sh.lanes, a, rb->carry32.p + pl.carry_at,
                                             bin.max_hap_len + 1, pl.teams, stream));
        }
    }
    if (fan_out) {
        for (int i = 0; i < c->n_side; ++i) {
            HIP_TRY(hipEventRecord(c->join[i], c->side[i]));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->join[i], 0));
        }
    }
    return PHMM_OK;
}

int sweep_rescue(phmm_context *c, phmm_resident *rb, const phmm_context::SweepLane *on)
{
    // the stream set the launches go to: the context's own, or the one handed in (the chunk pipeline's collector thread re-runs a
    // chunk's underflowed pairs on a lane of its own while the calling thread keeps enqueuing sweeps on the context's)
    hipStream_t s_main = on ? on->stream : c->stream;
    hipStream_t const *s_side = on ? on->side : c->side;  // (the context's own: made on the first fan-out, below)
    hipEvent_t s_fork = on ? on->fork : c->fork;
    hipEvent_t const *s_join = on ? on->join : c->join;

    // The fp64 re-run of the pairs the fp32 kernels queued.  Nothing comes back to the host between the two passes: the
    // team kernel reads the queue length from device memory and its workgroups stride over the queue (all of them leave
    // at once when it is empty), so a pass is one asynchronous chain on the context's stream.  One launch per queue (the
    // single-pass bins share queue 0; each striped size range has its own), shaped for the whole population of the queue
    // because its length is not known on the host.
    if (rb->n_pairs == 0) return PHMM_OK;
    struct QueuePlan
    {
        uint32_t base = 0, cap = 0, max_read = 0, max_hap = 0;
        int rows = phmm::kStripedRowsF64Min;
        TeamShape sh{16, 1};
    } q[kQueues];
    for (const Bin &bin : rb->bins) {
        QueuePlan &p = q[bin.queue];
        if (p.cap == 0) p.base = bin.queue == 0 ? 0 : bin.first;
        p.cap += bin.count;
        p.max_read = std::max(p.max_read, bin.max_read_len);
        p.max_hap = std::max(p.max_hap, bin.max_hap_len);
    }
    size_t carry_total = 0;
    for (int qi = 0; qi < kQueues; ++qi) {
        QueuePlan &p = q[qi];
        if (p.cap == 0) continue;
        // Rows per lane: more rows amortise the per-step work of a stripe better (measured rate per computed cell 1.00 :
        // 1.12 : 1.21 for 4 : 5 : 6), but the last stripe of a read is padded to a whole stripe -- 150-base reads are
        // 3 x 64, 2 x 80 or 2 x 96 rows.  Take the value with the least estimated time over the queue's length bins.
        static const double rate[3] = {1.00, 1.12, 1.21};
        double best = 0.0;
        int forced = 0;
        if (const char *f = phmm_hook("PHMM_FORCE_F64_ROWS")) forced = atoi(f);  // tests
        for (int k = phmm::kStripedRowsF64Min; k <= phmm::kStripedRowsF64Max; ++k) {
            const TeamShape cand = pick_team_shape(p.cap, p.max_read, p.max_hap, (uint32_t)k);
            const uint32_t stripe = (uint32_t)(cand.lanes * k);
            double cost = 0.0;
            for (const Bin &bin : rb->bins)
                if (bin.queue == qi) cost += (double)bin.count * (double)(((bin.max_read_len + stripe - 1) / stripe) * stripe);
            cost /= rate[k - phmm::kStripedRowsF64Min];
            if (k == phmm::kStripedRowsF64Min || cost < best || k == forced) {
                best = k == forced ? -1.0 : cost;
                p.rows = k;
                p.sh = cand;
            }
        }
        // (teams of several waves run the band kernel at kBandRows rows per lane whatever p.rows says: launch_striped_f64)
    }
    // Queues whose re-run takes the same kernel (rows per lane, team shape) share ONE launch: its teams stride over the queues one after
    // the other, four pairs of one queue at a time (KernelArgs::n_rescue_queues).  A batch of mixed lengths has a queue per length class
    // -- thirteen for reads of 50-250 bases -- and nearly all of them are empty nearly always; thirteen launches cost a pass 0.3 ms of a
    // mostly idle chip (rocprofv3 time line of config 3's one-shot call, round 4), at every chunk boundary.  PHMM_NO_RESCUE_MERGE: A/B hook.
    struct RescueLaunch
    {
        int rows;
        TeamShape sh;
        std::vector<int> queues;
        uint32_t cap = 0, max_hap = 0, teams = 0;
        size_t carry_at = 0;
    };
    std::vector<RescueLaunch> launches;
    const bool merge = !phmm_hook("PHMM_NO_RESCUE_MERGE");
    for (int qi = 0; qi < kQueues; ++qi) {
        const QueuePlan &p = q[qi];
        if (p.cap == 0) continue;
        RescueLaunch *into = nullptr;
        if (merge && !rb->queue_sparse[qi])  // (a list with holes is walked position by position: a launch of its own)
            for (RescueLaunch &L : launches)
                if (!rb->queue_sparse[L.queues[0]] && L.rows == p.rows && L.sh.lanes == p.sh.lanes && L.sh.waves == p.sh.waves &&
                    L.queues.size() < (size_t)phmm::kMaxRescueQueuesPerLaunch)
                    into = &L;
        if (!into) {
            launches.push_back(RescueLaunch{p.rows, p.sh, {}, 0, 0, 0, 0});
            into = &launches.back();
        }
        into->queues.push_back(qi);
        into->cap += p.cap;
        into->max_hap = std::max(into->max_hap, p.max_hap);
    }
    carry_total = 0;
    for (RescueLaunch &L : launches) {
        const uint32_t per_wave = 64u / (uint32_t)L.sh.lanes;
        L.teams = std::min<uint32_t>((L.cap + per_wave - 1) / per_wave, kRescueMaxTeams / L.sh.waves);
        L.carry_at = carry_total;
        carry_total += (phmm::team_carry_values(L.teams, L.sh.waves, L.max_hap + 1, L.sh.lanes) + 63) & ~size_t(63);
    }
    HIP_TRY(rb->carry64.alloc(carry_total));
    const bool fan_out = launches.size() > 1;
    if (fan_out && !on) {
        const int rc_side = ensure_side_streams(c);
        if (rc_side != PHMM_OK) return rc_side;
        s_side = c->side;
        s_join = c->join;
    }
    if (fan_out) {
        HIP_TRY(hipEventRecord(s_fork, s_main));
        for (int i = 0; i < c->n_side; ++i) HIP_TRY(hipStreamWaitEvent(s_side[i], s_fork, 0));
    }
    int nth = 0;
    for (const RescueLaunch &L : launches) {
        const int lane = fan_out ? nth++ % (c->n_side + 1) : 0;
        hipStream_t st = lane == 0 ? s_main : s_side[lane - 1];
        phmm::KernelArgs a{};
        a.read_bases = rb->read_bases.p;
        a.read_qual = rb->read_qual.p;
        a.read_ins = rb->read_ins.p;
        a.read_del = rb->read_del.p;
        a.read_gcp = rb->read_gcp.p;
        a.read_off = rb->read_off.p;
        a.read_len = rb->read_len.p;
        a.hap_len = rb->hap_len.p;
        a.planes = rb->planes.p;
        a.plane_off = rb->plane_off.p;
        a.pair_read = rb->pair_read.p;
        a.pair_hap = rb->pair_hap.p;
        a.n_pairs = 0;
        const bool f64_first = L.queues.size() == 1 && rb->queue_f64_first[L.queues[0]];
        if (f64_first) {
            // fp64 FIRST (sweep_fp32): every pair of the range, in processing order; the pairs whose sum says fp32 might not underflow
            // are queued for the fp32 kernel below, at their own positions
            const int qi = L.queues[0];
            a.order = rb->order.p + q[qi].base;
            a.n_pairs = q[qi].cap;
            a.sparse_order = 1;
            a.rescue_list = rb->rescue_list.p + q[qi].base;
            a.rescue_count = rb->rescue_count.p + qi;
            a.need32_above = std::ldexp(1e-28, 900 - 20);
        }
        else if (L.queues.size() == 1) {
            const int qi = L.queues[0];
            a.order = rb->rescue_list.p + q[qi].base;
            a.n_pairs_dev = rb->rescue_count.p + qi;
            if (rb->queue_sparse[qi]) {  // a striped range's queue: every position of the range, holes skipped (KernelArgs::sparse_order)
                a.n_pairs = q[qi].cap;
                a.sparse_order = 1;
            }
        }
        else {
            a.order = rb->rescue_list.p;
            a.rescue_count = rb->rescue_count.p;  // (read, not written, by the fp64 launches)
            a.n_rescue_queues = (uint32_t)L.queues.size();
            for (size_t i = 0; i < L.queues.size(); ++i) {
                a.rescue_queue[i] = (uint32_t)L.queues[i];
                a.rescue_base[i] = q[L.queues[i]].base;
            }
        }
        a.lds_words_per_group = (uint32_t)phmm::kPlanes * phmm::planes_words(L.max_hap);
        a.no_early_exit = phmm_hook("PHMM_NO_EARLY_EXIT") ? 1u : 0u;  // tests
        a.ph2pr = c->ph2pr_f64.p;
        a.mm_table = c->mm_f64.p;
        a.out_raw = rb->out64_host ? static_cast<void *>(rb->out64_host) : static_cast<void *>(rb->raw64.p);
        // tuning aid: PHMM_BAND_TIMELINE=<file> -- the first 64 teams of the band launch log every stripe (begin, first non-skipped
        // superstep, end, waits; 100 MHz ticks), dumped as text after a synchronisation.  The A/B behind DESIGN.md's config 5 section.
        const char *tl_path = L.sh.waves > 1 ? phmm_hook("PHMM_BAND_TIMELINE") : nullptr;
        DevBuf<unsigned long long> tl;
        const uint32_t tl_teams = std::min<uint32_t>(64u, L.teams);
        const size_t tl_n = (size_t)tl_teams * phmm::kTimelineStripes * phmm::kTimelineWords;
        if (tl_path) {
            HIP_TRY(tl.alloc(tl_n));
            HIP_TRY(hipMemsetAsync(tl.p, 0, tl_n * sizeof(unsigned long long), st));
            a.timeline = tl.p;
            a.timeline_teams = tl_teams;
        }
        HIP_TRY(phmm::launch_striped_f64(L.rows, rb->uniform_gcp, L.sh.waves, L.sh.lanes, a, rb->carry64.p + L.carry_at, L.max_hap + 1,
                                         L.teams, st));
        if (f64_first) {
            // the fp32 kernel over the queued pairs (usually none: its teams read the count and leave), behind the fp64 launch
            const int qi = L.queues[0];
            for (size_t i = 0; i < rb->bins.size(); ++i) {
                const Bin &bin = rb->bins[i];
                if (bin.queue != qi || i >= rb->fp32_plans.size()) continue;
                const phmm_resident::StripedLaunchPlan &pl = rb->fp32_plans[i];
                phmm::KernelArgs b = a;
                b.order = rb->rescue_list.p + bin.first;
                b.n_pairs = bin.count;
                b.n_pairs_dev = rb->rescue_count.p + qi;
                b.sparse_order = 1;
                b.rescue_list = nullptr;
                b.rescue_count = nullptr;
                b.need32_above = 0.0;
                b.timeline = nullptr;
                b.timeline_teams = 0;
                b.lds_words_per_group = bin.lds_words_per_group;
                b.ph2pr = c->ph2pr_f32.p;
                b.mm_table = c->mm_f32.p;
                b.out_raw = rb->raw32.p;
                HIP_TRY(phmm::launch_striped_f32(rb->uniform_gcp, pl.sh.waves, pl.sh.lanes, b, rb->carry32.p + pl.carry_at, bin.max_hap_len + 1,
                                                 pl.teams, st));
            }
        }
        if (tl_path) {
            std::vector<unsigned long long> h(tl_n);
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMemcpy(h.data(), tl.p, tl_n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE *f = fopen(tl_path, "w")) {
                fprintf(f, "# team stripe begin_us first_nonskipped_done_us end_us c_first out_first out_last waited_us supersteps_run (times since the launch's first record)\n");
                unsigned long long t0 = ~0ull;
                for (size_t i = 0; i < tl_n; i += phmm::kTimelineWords)
                    if (h[i + 2]) t0 = std::min(t0, h[i]);
                for (uint32_t tm = 0; tm < tl_teams; ++tm)
                    for (int v = 0; v < phmm::kTimelineStripes; ++v) {
                        const unsigned long long *r = h.data() + ((size_t)tm * phmm::kTimelineStripes + v) * phmm::kTimelineWords;
                        if (!r[2]) continue;
                        fprintf(f, "%u %d %.2f %.2f %.2f %llu %lld %llu %.2f %llu\n", tm, v, (r[0] - t0) / 100.0, r[1] ? (r[1] - t0) / 100.0 : -1.0,
                                (r[2] - t0) / 100.0, r[3], (long long)(int)r[4], r[5], r[6] / 100.0, r[7]);
                    }
                fclose(f);
            }
        }
    }
    if (fan_out) {
        for (int i = 0; i < c->n_side; ++i) {
            HIP_TRY(hipEventRecord(s_join[i], s_side[i]));
            HIP_TRY(hipStreamWaitEvent(s_main, s_join[i], 0));
        }
    }
    // what this pass observed on its striped ranges, for the next one's choice of order (sweep_fp32, "fp64 first"): the counters come
    // back behind the pass, nobody waits for them
    bool observe = false;
    for (int qi = 1; qi <= 4; ++qi) observe = observe || (rb->queue_sparse[qi] && q[qi].cap && !rb->out32_host);
    if (observe) {
        HIP_TRY(rb->hist_host.ensure(kQueues * sizeof(uint32_t)));
        if (!rb->hist_ready) HIP_TRY(hipEventCreateWithFlags(&rb->hist_ready, hipEventDisableTiming));
        HIP_TRY(hipMemcpyAsync(rb->hist_host.p, rb->rescue_count.p, kQueues * sizeof(uint32_t), hipMemcpyDeviceToHost, s_main));
        HIP_TRY(hipEventRecord(rb->hist_ready, s_main));
        for (int qi = 0; qi < kQueues; ++qi) {
            rb->hist_cap[qi] = qi >= 1 && qi <= 4 && rb->queue_sparse[qi] ? q[qi].cap : 0u;
            rb->hist_f64_first[qi] = rb->queue_f64_first[qi];
        }
        rb->hist_pending = true;
    }
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
    rc = impl::phmm_sweep(c, rb);
    if (rc != PHMM_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    rc = impl::phmm_download(c, rb, out_raw_f32, out_log10, stats);
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
