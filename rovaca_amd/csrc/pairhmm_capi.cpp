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
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(PHMM_E_NO_DEVICE, "phmm_create: no HIP device visible (this engine has no CPU fallback)");
    if (device_id >= n) return fail(PHMM_E_NO_DEVICE, "phmm_create: device %d of %d does not exist", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    phmm_context *c = new (std::nothrow) phmm_context();
    if (!c) return fail(PHMM_E_NOMEM, "phmm_create: out of host memory");
    c->device = device_id;
    if (const char *e = getenv("PHMM_SIDE_STREAMS")) c->n_side = std::max(0, std::min((int)phmm_context::kSideStreams, atoi(e)));
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
    for (int i = 0; i < phmm_context::kSideStreams && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming);
    }
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
    delete c->pack_pool;
    delete c->fin_pool;
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
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
static thread_local PhaseClock *g_phase = nullptr;
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

static int upload_into(phmm_context *c, const phmm_batch *b, phmm_resident *rb, const UploadOpts &opt = UploadOpts())
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
    hipStream_t st = opt.stream ? opt.stream : c->stream;
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
    HIP_TRY(rb->blob.alloc(total));
    uint8_t *stage = static_cast<uint8_t *>(rb->stage.p);
    auto copy = [](void *dst, const void *src, size_t bytes) {  // big tables of a big batch: a few threads share the copy
        if (bytes < (size_t(1) << 20)) {
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
        const int rc = phmm_detail::bin_pairs(b, cross, n_pairs, opt.read_bias, opt.hap_bias, pr, ph, order, c->bin_scratch, rb->bins, &cells);
        if (rc != PHMM_OK) return rc;
    }
    rb->cells = cells;
    PHASE(2);

    // ---- upload ----
    if (opt.setup_stream && opt.setup_stream != st) {
        // The chunked call: the set-up launch goes on the sweep stream and must see the copy.  A device-side wait there
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
    rb->read_mapq = {base + o_mapq, prep ? (size_t)b->n_reads : 0};
    rb->read_off = {reinterpret_cast<uint32_t *>(base + o_roff), b->n_reads};
    rb->read_len = {reinterpret_cast<uint32_t *>(base + o_rlen), b->n_reads};
    rb->hap_bases = {base + o_hap, (size_t)hap_bytes};
    rb->hap_off = {reinterpret_cast<uint32_t *>(base + o_hoff), b->n_haps};
    rb->hap_len = {reinterpret_cast<uint32_t *>(base + o_hlen), b->n_haps};
    rb->pair_read = {reinterpret_cast<uint32_t *>(base + o_pr), (size_t)n_pairs};
    rb->pair_hap = {reinterpret_cast<uint32_t *>(base + o_ph), (size_t)n_pairs};
    rb->order = {reinterpret_cast<uint32_t *>(base + o_order), (size_t)n_pairs};
    rb->plane_off = {reinterpret_cast<uint32_t *>(base + o_poff), b->n_haps};
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
        sa.n_prep_reads = prep ? b->n_reads : 0;
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
    const int rc = c->cpu ? cpu_upload(c, b, rb) : upload_into(c, b, rb);
    if (rc != PHMM_OK) {
        delete rb;
        return rc;
    }
    *out = rb;
    return PHMM_OK;
}

int sweep_fp32(phmm_context *c, phmm_resident *rb)
{
    if (rb->n_pairs && !rb->queue_is_empty) HIP_TRY(hipMemsetAsync(rb->rescue_count.p, 0, kQueues * sizeof(uint32_t), c->stream));
    rb->queue_is_empty = false;
    const bool fan_out = rb->bins.size() > 1;
    if (fan_out) {
        HIP_TRY(hipEventRecord(c->fork, c->stream));
        for (int i = 0; i < c->n_side; ++i) HIP_TRY(hipStreamWaitEvent(c->side[i], c->fork, 0));
    }
    // striped launches run side by side on different streams: each gets its own slice of the carry buffer, sized up front
    struct StripedPlan
    {
        TeamShape sh;
        uint32_t teams;
        size_t carry_at;
    };
    std::vector<StripedPlan> plans(rb->bins.size());
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
    int nth = 0;
    for (size_t i = 0; i < rb->bins.size(); ++i) {
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
        if (bin.rows_per_lane > 0) {
            // Few pairs: one pair per wave.  Four per wave would occupy count / 4 of the chip's 1024 SIMDs with one slow wave
            // each; 64 lanes per pair makes the sweep of a pair about half as long (a quarter of the rows per lane, 48 more
            // steps) at 1.8x the instructions, which is free while SIMDs idle.  Measured: a 432-pair region 85 -> 46 us, 1728
            // pairs 86 -> 68 us, no gain from 4000 pairs on, and a loss when several submissions share the chip, hence the
            // low threshold.
            uint32_t wide_max = kWideMaxPairs;
            if (const char *f = phmm_hook("PHMM_WIDE_MAX_PAIRS")) wide_max = (uint32_t)std::max(0, atoi(f));  // tuning
            int lanes = bin.count <= wide_max ? 64 : 16;
            if (const char *f = phmm_hook("PHMM_FORCE_LANES")) {  // tests
                const int l = atoi(f);
                if (l == 16 || l == 64) lanes = l;
            }
            const int k = lanes == 64 ? (int)((bin.max_read_len + 63) / 64) : bin.rows_per_lane;
            // big four-pairs-per-wave launches: runs of 64 workgroups (1 024 pairs, a region or so) per XCD, see KernelArgs
            uint32_t xcd_min = kXcdLocalMinPairs;
            if (const char *f = phmm_hook("PHMM_XCD_LOCAL_MIN_PAIRS")) xcd_min = (uint32_t)std::max(1, atoi(f));  // tests: small launches too
            if (lanes == 16 && bin.count >= xcd_min && !phmm_hook("PHMM_NO_XCD_LOCAL_ORDER")) a.xcd_run_log2 = 6;
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

int sweep_rescue(phmm_context *c, phmm_resident *rb)
{
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
        uint32_t teams = 0;
        size_t carry_at = 0;
    } q[kQueues];
    for (const Bin &bin : rb->bins) {
        QueuePlan &p = q[bin.queue];
        if (p.cap == 0) p.base = bin.queue == 0 ? 0 : bin.first;
        p.cap += bin.count;
        p.max_read = std::max(p.max_read, bin.max_read_len);
        p.max_hap = std::max(p.max_hap, bin.max_hap_len);
    }
    size_t carry_total = 0;
    int n_active = 0;
    for (int qi = 0; qi < kQueues; ++qi) {
        QueuePlan &p = q[qi];
        if (p.cap == 0) continue;
        ++n_active;
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
        const uint32_t per_wave = 64u / (uint32_t)p.sh.lanes;
        p.teams = std::min<uint32_t>((p.cap + per_wave - 1) / per_wave, kRescueMaxTeams / p.sh.waves);
        p.carry_at = carry_total;
        carry_total += (phmm::team_carry_values(p.teams, p.sh.waves, p.max_hap + 1, p.sh.lanes) + 63) & ~size_t(63);
    }
    HIP_TRY(rb->carry64.alloc(carry_total));
    const bool fan_out = n_active > 1;
    if (fan_out) {
        HIP_TRY(hipEventRecord(c->fork, c->stream));
        for (int i = 0; i < c->n_side; ++i) HIP_TRY(hipStreamWaitEvent(c->side[i], c->fork, 0));
    }
    int nth = 0;
    for (int qi = 0; qi < kQueues; ++qi) {
        const QueuePlan &p = q[qi];
        if (p.cap == 0) continue;
        const int lane = fan_out ? nth++ % (c->n_side + 1) : 0;
        hipStream_t st = lane == 0 ? c->stream : c->side[lane - 1];
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
        a.order = rb->rescue_list.p + p.base;
        a.n_pairs = 0;
        a.n_pairs_dev = rb->rescue_count.p + qi;
        a.lds_words_per_group = (uint32_t)phmm::kPlanes * phmm::planes_words(p.max_hap);
        a.no_early_exit = phmm_hook("PHMM_NO_EARLY_EXIT") ? 1u : 0u;  // tests
        a.ph2pr = c->ph2pr_f64.p;
        a.mm_table = c->mm_f64.p;
        a.out_raw = rb->out64_host ? static_cast<void *>(rb->out64_host) : static_cast<void *>(rb->raw64.p);
        HIP_TRY(phmm::launch_striped_f64(p.rows, rb->uniform_gcp, p.sh.waves, p.sh.lanes, a, rb->carry64.p + p.carry_at,
                                         p.max_hap + 1, p.teams, st));
    }
    if (fan_out) {
        for (int i = 0; i < c->n_side; ++i) {
            HIP_TRY(hipEventRecord(c->join[i], c->side[i]));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->join[i], 0));
        }
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
    const bool want = (out_raw_f32 || out_log10) && n;
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
        for (int qi = 0; qi < kQueues; ++qi) rb->last_rescued += counts[qi];
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
        stats->n_launches = (uint32_t)rb->bins.size();
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

// ---- the chunked one-shot call ----
// A window of the caller's tables that holds everything a contiguous range of the pair list refers to.
struct ChunkWindow
{
    uint64_t p0 = 0, p1 = 0;               // pairs
    uint32_t r0 = 0, r1 = 0, h0 = 0, h1 = 0;  // reads [r0, r1), haplotypes [h0, h1)
    uint64_t rb0 = 0, rb1 = 0, hb0 = 0, hb1 = 0;  // byte ranges of the two tables
};

constexpr uint64_t kChunkedMinPairs = 1u << 19;  // below this one upload + one sweep is as fast
constexpr uint64_t kChunkTargetPairs = 1u << 17;
constexpr unsigned kMaxChunks = 8;

// Cuts the pair list into contiguous chunks and finds each chunk's table windows.  Returns false when chunking would not
// pay: a small batch, or a pair list that jumps around the tables so that the windows overlap heavily (each chunk would
// upload most of the tables again).
static bool plan_chunks(const phmm_batch *b, std::vector<ChunkWindow> &plan)
{
    const bool cross = b->pair_read == nullptr;
    const uint64_t n = cross ? (uint64_t)b->n_reads * b->n_haps : b->n_pairs;
    if (n < kChunkedMinPairs || b->n_reads == 0 || b->n_haps == 0) return false;
    if (const char *e = phmm_hook("PHMM_NO_CHUNKED_COMPUTE")) {  // tests
        if (e[0] != '0') return false;
    }
    // Chunk sizes rise and fall: the first chunk's packing and the last chunk's conversion are the only host work the device
    // cannot hide, so those two are small -- 2 % and 8 % of the batch from seven chunks on.
    unsigned C = (unsigned)std::min<uint64_t>(kMaxChunks, std::max<uint64_t>(2, n / kChunkTargetPairs + 2));
    std::vector<unsigned> weight(C);
    for (unsigned k = 0; k < C; ++k) weight[k] = std::min(k + 1, std::min(C - k, 3u));
    if (C >= 7) {
        // percent of the batch.  Packing is ~2.6x faster than sweeping, so a chunk may be up to 2.6x the one before it without the
        // device running dry; the first is as small as a launch can usefully be, the last small enough for a short conversion tail
        static const unsigned graded[] = {2, 6, 14, 26, 26, 18, 8};
        C = 7;
        weight.assign(graded, graded + 7);
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
static int compute_chunked(phmm_context *c, const phmm_batch *b, const std::vector<ChunkWindow> &plan, float *out_raw_f32,
                           double *out_log10, phmm_stats *stats)
{
    HIP_TRY(hipSetDevice(c->device));
    constexpr int S = phmm_context::kChunkSlots;
    for (int i = 0; i < S; ++i) {
        if (!c->chunk_slot[i]) {
            c->chunk_slot[i] = new (std::nothrow) phmm_resident();
            if (!c->chunk_slot[i]) return fail(PHMM_E_NOMEM, "phmm_compute: out of host memory");
        }
        if (!c->chunk_up[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_up[i], hipEventDisableTiming));
        if (!c->chunk_swept[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_swept[i], hipEventDisableTiming));
        if (!c->chunk_done[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_done[i], hipEventDisableTiming));
    }
    if (!c->down) HIP_TRY(hipStreamCreateWithFlags(&c->down, hipStreamNonBlocking));
    hipStream_t up = c->side[phmm_context::kSideStreams - 1];  // not used by the sweeps (n_side defaults to 2 of 3)
    if (c->n_side >= phmm_context::kSideStreams) up = c->stream;
    hipStream_t down = c->down;
    HIP_TRY(hipStreamSynchronize(c->stream));  // nothing of an earlier call still reads the slots
    if (up != c->stream) HIP_TRY(hipStreamSynchronize(up));
    HIP_TRY(hipStreamSynchronize(down));
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
        bool want64 = true;  // results written by the kernels straight into the pinned stage: the fp64 sums are there already
        if (!direct) {
            const uint32_t *counts = reinterpret_cast<const uint32_t *>(st + off_cnt);
            rb->last_rescued = 0;
            for (int qi = 0; qi < kQueues; ++qi) rb->last_rescued += counts[qi];
            want64 = out_log10 && rb->last_rescued;
            if (want64) {
                // the rescue kernels of this chunk are done (the event above follows them); the copy runs beside the next sweep
                HIP_TRY(hipMemcpyAsync(const_cast<uint8_t *>(st) + off64, rb->raw64.p, n * sizeof(double), hipMemcpyDeviceToHost, down));
                HIP_TRY(hipStreamSynchronize(down));
            }
        }
        if (out_raw_f32) memcpy(out_raw_f32 + w.p0, rp, n * sizeof(float));
        std::atomic<uint64_t> n_resc{0};
        {
            double *dst = out_log10 ? out_log10 + w.p0 : nullptr;
            auto finalize = [&](uint64_t lo, uint64_t hi, unsigned) {  // pairhmm_internal.cpp:577-585
                uint64_t resc = 0;
                for (uint64_t p = lo; p < hi; ++p) {
                    const float x = rp[p];
                    if (x < PHMM_MIN_ACCEPTED_F32) {
                        ++resc;
                        if (dst) dst[p] = log10(want64 ? r64[p] : 0.0) - PHMM_LOG10_INITIAL_CONSTANT_F64;
                    }
                    else if (dst) {
                        dst[p] = (double)(log10f(x) - PHMM_LOG10_INITIAL_CONSTANT_F32);
                    }
                }
                n_resc.fetch_add(resc, std::memory_order_relaxed);
            };
            if (n >= (1u << 14))
                parallel_ranges(n, c->fin_pool->helpers() + 1u, finalize);
            else
                finalize(0, n, 0);
        }
        rb->last_rescued = (uint32_t)n_resc.load();
        total.cells += rb->cells;
        total.n_rescued += rb->last_rescued;
        total.n_launches += (uint32_t)rb->bins.size();
        if (!rb->bins.empty()) total.max_rows_per_lane = std::max<uint32_t>(total.max_rows_per_lane, (uint32_t)rb->bins.back().rows_per_lane);
        if (trace) fprintf(stderr, "phmm chunk %zu: results on the host at %.0f us, converted at %.0f us\n", k, t_ready, now_us());
        return PHMM_OK;
    };

    std::thread collector([&] {
        (void)hipSetDevice(c->device);
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

    phmm_detail::PoolScope pool(c->pack_pool);
    std::vector<uint32_t> roff, hoff;
    for (size_t k = 0; k < n_chunks; ++k) {
        const ChunkWindow &w = plan[k];
        const int slot = (int)(k % S);
        if (k >= (size_t)S) {  // the slot held chunk k - S: its results must have left it
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return collected + S > k; });
            if (collector_rc != PHMM_OK) break;
        }
        const double t_a = trace ? now_us() : 0.0;
        phmm_resident *rb = c->chunk_slot[slot];
        // the chunk as a batch of its own: windows of the tables, offsets rebased to the windows, pair indices biased
        phmm_batch s = *b;
        s.n_reads = w.r1 - w.r0;
        s.n_haps = w.h1 - w.h0;
        s.n_pairs = w.p1 - w.p0;
        roff.resize(s.n_reads);
        for (uint32_t r = 0; r < s.n_reads; ++r) roff[r] = (uint32_t)(b->read_off[w.r0 + r] - w.rb0);
        hoff.resize(s.n_haps);
        for (uint32_t h = 0; h < s.n_haps; ++h) hoff[h] = (uint32_t)(b->hap_off[w.h0 + h] - w.hb0);
        s.read_bases = b->read_bases + w.rb0;
        s.read_qual = b->read_qual + w.rb0;
        s.read_ins = b->read_ins ? b->read_ins + w.rb0 : nullptr;
        s.read_del = b->read_del ? b->read_del + w.rb0 : nullptr;
        s.read_gcp = b->read_gcp ? b->read_gcp + w.rb0 : nullptr;
        s.read_off = roff.data();
        s.read_len = b->read_len + w.r0;
        s.hap_bases = b->hap_bases + w.hb0;
        s.hap_off = hoff.data();
        s.hap_len = b->hap_len + w.h0;
        s.pair_read = cross ? nullptr : b->pair_read + w.p0;
        s.pair_hap = cross ? nullptr : b->pair_hap + w.p0;
        phmm_read_prep prep_k;
        if (b->prep) {
            prep_k = *b->prep;
            prep_k.read_mapq = b->prep->read_mapq ? b->prep->read_mapq + w.r0 : nullptr;
            s.prep = &prep_k;
        }
        UploadOpts opt;
        opt.stream = up;
        opt.caller_synchronised = true;  // slot `slot` held chunk k - S: collected (see above), so its sweep is long over
        opt.read_bias = cross ? 0 : w.r0;
        opt.hap_bias = cross ? 0 : w.h0;
        // The copy travels on the upload stream (DMA engines: it runs beside the previous chunk's sweep); the set-up launch goes on
        // the context's stream, in front of this chunk's sweep.  On the upload stream it would not start before the previous sweep
        // has handed out its last workgroup -- small kernels of another queue are starved by a grid that fills the chip -- and the
        // sweep behind it would wait with it: 90-110 us of a drained chip at every chunk boundary (rocprofv3 time line, round 3).
        opt.setup_stream = c->stream;
        if (k == 0) HIP_TRY(hipEventRecord(c->ev0, c->stream));
        PhaseClock chunk_clock;
        struct PhaseScope
        {
            explicit PhaseScope(PhaseClock *p) { g_phase = p; }
            ~PhaseScope() { g_phase = nullptr; }
        } phase_scope(trace ? &chunk_clock : nullptr);
        int rc = upload_into(c, &s, rb, opt);
        if (rc != PHMM_OK) return rc;
        // where the sums go: straight into the slot's pinned result block (the kernels write host memory over PCIe: no copy, no
        // cross-stream event between two sweeps), or -- PHMM_CHUNK_RESULTS_BY_COPY=1 -- into HBM and from there by a copy
        const uint64_t n = rb->n_pairs;
        const size_t count_at = (n + 63) & ~size_t(63);
        HIP_TRY(rb->out_stage.ensure(count_at * sizeof(float) + 256 + n * sizeof(double)));
        rb->out32_host = direct ? static_cast<float *>(rb->out_stage.p) : nullptr;
        rb->out64_host = direct ? reinterpret_cast<double *>(static_cast<uint8_t *>(rb->out_stage.p) + count_at * sizeof(float) + 256) : nullptr;
        rc = impl::phmm_sweep(c, rb);
        if (rc != PHMM_OK) return rc;
        PHASE(4);
        if (k + 1 == n_chunks) HIP_TRY(hipEventRecord(c->ev1, c->stream));
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
            fprintf(stderr, "phmm chunk %zu: %llu pairs, packed and enqueued %.0f..%.0f us (validate %.0f pack %.0f bin %.0f copy+setup %.0f sweep-enqueue %.0f results-enqueue %.0f)\n",
                    k, (unsigned long long)(w.p1 - w.p0), t_a, now_us(), chunk_clock.us[0], chunk_clock.us[1], chunk_clock.us[2], chunk_clock.us[3],
                    chunk_clock.us[4], chunk_clock.us[5]);
        }
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return collected >= enqueued || collector_rc != PHMM_OK; });
    }
    joiner.ok = enqueued == n_chunks;
    if (collector_rc != PHMM_OK) return fail(collector_rc, "%s", collector_msg.c_str());
    if (enqueued != n_chunks) return fail(PHMM_E_INVALID, "phmm_compute: the chunk pipeline stopped early");
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        total.kernel_ms = ms;
        *stats = total;
    }
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
        if (n_all >= kChunkedMinPairs) {  // the helper threads of the chunk pipeline (also used by the plan itself)
            if (!c->pack_pool) c->pack_pool = new phmm_detail::HostPool(std::max(1u, host_threads()) - 1u);
            if (!c->fin_pool) c->fin_pool = new phmm_detail::HostPool(3);
        }
        phmm_detail::PoolScope pool(n_all >= kChunkedMinPairs ? c->pack_pool : nullptr);
        const bool tables_ok = b->read_bases && b->read_qual && b->read_off && b->read_len && b->hap_bases && b->hap_off && b->hap_len &&
                               ((b->pair_read == nullptr) == (b->pair_hap == nullptr)) && (b->prep ? b->prep->read_mapq != nullptr : true);
        if (tables_ok && plan_chunks(b, plan)) {
            const int rc = compute_chunked(c, b, plan, out_raw_f32, out_log10, stats);
            if (rc == PHMM_OK && stats)
                stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return rc;
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
    int rc = upload_into(c, b, rb);
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
    const bool shared_ok = getenv("PHMM_MULTI_ALLOW_SHARED_DEVICE") != nullptr;
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

// ----------------------------------------------------------------------------------------------------
// the C boundary: nothing is thrown across it
// ----------------------------------------------------------------------------------------------------
namespace
{
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
}  // namespace

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
