// pairhmm_stage.cpp -- region-descriptor batches (include/pairhmm_hip.h, ABI 5): phmm_stage_* and phmm_compute_regions.
//
// A stage is pinned host memory laid out as the device wants it plus the HBM buffers of one submission.  All sequence bytes
// (read bases, raw read qualities, haplotype bases) live in ONE arena that the caller's threads fill at ranges they hand out among
// themselves; phmm_stage_submit does O(reads + haplotypes) host work (validate, bin per read, one 32-byte descriptor per read
// and 16 bytes per haplotype, written right behind the used part of the arena), enqueues ONE host-to-device copy of that block
// and ONE set-up launch (pairhmm_setup.hip: stage_setup_kernel), then the same sweep and fp64 re-run launches a resident batch
// gets (pairhmm_capi.cpp: sweep_fp32 / sweep_rescue).  The sums are written by the kernels straight into the stage's pinned
// result arrays, so nothing is copied back and a submission is waited for once.  (Round 3 first let the set-up launch read the
// stage over PCIe instead of copying it: 64-byte reads at the link's request rate, 36 us for 8 regions and 171 us for 100 --
// the DMA engine moves the same bytes in a fraction of that.)  No CPU compute path for a device context.
#include <chrono>
#include <cmath>
#include <new>

#include <emmintrin.h>

#include "pairhmm_engine.h"

struct phmm_stage
{
    phmm_stage_caps caps{};
    bool cpu = false;
    HostBuf pinned;               // device contexts: everything the view and the results point into
    std::vector<uint8_t> plain;   // CPU contexts: the same block in ordinary memory
    phmm_stage_view view{};
    size_t arena_cap = 0;                      // max_bytes + room for the descriptors behind the used part
    float *raw32 = nullptr;                    // [max_pairs]   written by the sweep kernels (host memory, over PCIe)
    double *raw64 = nullptr;                   // [max_pairs]   written by the fp64 re-run, only where raw32 < 1e-28f
    phmm_resident rb;                          // the HBM side: rb.blob holds the copy of the arena, the rest are views / buffers
    std::vector<uint32_t> slot_first, pair_first;
    uint32_t n_regions = 0;
    uint64_t n_pairs = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool in_flight = false;
    const phmm_context *submitted_on = nullptr;  // the context of the submission in flight: phmm_stage_wait re-runs fp64 on it
    bool rescue_deferred = false;              // the submission's fp64 re-run is phmm_stage_wait's to launch, if a sum underflowed
    bool valid = false;                        // the last submission succeeded: the results and pair_first describe it
    phmm_stage_params params{};
};

namespace impl
{

static size_t plane_words_cap(const phmm_stage_caps &k)
{
    // planes_words(H) = ((H + 14) >> 5) + 2 <= H / 32 + 3 per haplotype and plane
    return (size_t)phmm::kPlanes * ((size_t)k.max_bytes / 32 + 3 * (size_t)k.max_haps);
}

void phmm_stage_destroy(phmm_context *c, phmm_stage *st)
{
    if (!st) return;
    if (c && !c->cpu) {
        (void)hipSetDevice(c->device);
        // a submission in flight was enqueued on the context it was SUBMITTED on, which need not be `c` (the adapter creates its stages
        // on one context and submits them on its dispatchers'): its event says when the kernels have stopped writing the stage
        if (st->in_flight && st->ev1) (void)hipEventSynchronize(st->ev1);
        (void)hipStreamSynchronize(c->stream);
    }
    else if (st->in_flight && st->ev1) {
        (void)hipEventSynchronize(st->ev1);
    }
    if (st->ev0) (void)hipEventDestroy(st->ev0);
    if (st->ev1) (void)hipEventDestroy(st->ev1);
    delete st;
}

static size_t desc_bytes(uint32_t n_reads, uint32_t n_haps)
{
    return (size_t)n_reads * sizeof(phmm::StageReadDesc) + (size_t)n_haps * sizeof(phmm::StageHapDesc);
}

int phmm_stage_create(phmm_context *c, const phmm_stage_caps *caps, phmm_stage **out)
{
    if (!c || !caps || !out) return fail(PHMM_E_INVALID, "phmm_stage_create: null argument");
    *out = nullptr;
    const phmm_stage_caps k = *caps;
    if (!k.max_regions || !k.max_reads || !k.max_haps || !k.max_bytes || !k.max_pairs || k.max_pairs > 0x7FFFFFF0u || k.max_bytes > 0x7FFFFFF0u)
        return fail(PHMM_E_INVALID, "phmm_stage_create: bad capacities");
    phmm_stage *st = new (std::nothrow) phmm_stage();
    if (!st) return fail(PHMM_E_NOMEM, "phmm_stage_create: out of host memory");
    st->caps = k;
    st->cpu = c->cpu;
    st->arena_cap = (((size_t)k.max_bytes + 255) & ~size_t(255)) + desc_bytes(k.max_reads, k.max_haps) + 256;
    // ---- the host block ----
    size_t total = 0;
    auto reserve = [&](size_t bytes) {
        const size_t at = total;
        total += (bytes + 255) & ~size_t(255);
        return at;
    };
    const size_t o_arena = reserve(st->arena_cap);
    const size_t o_roff = reserve((size_t)k.max_reads * 4), o_qoff = reserve((size_t)k.max_reads * 4), o_rlen = reserve((size_t)k.max_reads * 4);
    const size_t o_mapq = reserve(k.max_reads);
    const size_t o_hoff = reserve((size_t)k.max_haps * 4), o_hlen = reserve((size_t)k.max_haps * 4);
    const size_t o_reg = reserve((size_t)k.max_regions * sizeof(phmm_region));
    const size_t o_raw32 = reserve((size_t)k.max_pairs * sizeof(float)), o_raw64 = reserve((size_t)k.max_pairs * sizeof(double));
    uint8_t *base = nullptr;
    if (st->cpu) {
        st->plain.assign(total, 0);
        base = st->plain.data();
    }
    else {
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = st->pinned.ensure(total);
        if (e != hipSuccess) {
            delete st;
            return fail(PHMM_E_HIP, "phmm_stage_create: %zu bytes of pinned host memory: %s", total, hipGetErrorString(e));
        }
        base = static_cast<uint8_t *>(st->pinned.p);
        memset(base, 0, total);  // touch every page now, not under the first submission
    }
    st->view.bytes = base + o_arena;
    st->view.read_off = reinterpret_cast<uint32_t *>(base + o_roff);
    st->view.read_qual_off = reinterpret_cast<uint32_t *>(base + o_qoff);
    st->view.read_len = reinterpret_cast<uint32_t *>(base + o_rlen);
    st->view.read_mapq = base + o_mapq;
    st->view.hap_off = reinterpret_cast<uint32_t *>(base + o_hoff);
    st->view.hap_len = reinterpret_cast<uint32_t *>(base + o_hlen);
    st->view.regions = reinterpret_cast<phmm_region *>(base + o_reg);
    st->raw32 = reinterpret_cast<float *>(base + o_raw32);
    st->raw64 = reinterpret_cast<double *>(base + o_raw64);
    st->slot_first.resize(k.max_reads);
    st->pair_first.resize((size_t)k.max_regions + 1);
    if (st->cpu) {
        *out = st;
        return PHMM_OK;
    }
    // ---- HBM: the copy of the arena (rb.blob), the prepared read arrays parallel to it, the index arrays, the planes ----
    phmm_resident &rb = st->rb;
    size_t dtotal = 0;
    auto dreserve = [&](size_t bytes) {
        const size_t at = dtotal;
        dtotal += (bytes + 255) & ~size_t(255);
        return at;
    };
    const size_t d_arena = dreserve(st->arena_cap);
    const size_t d_qual = dreserve(k.max_bytes), d_gop = dreserve(k.max_bytes), d_gcp = dreserve(k.max_bytes);
    const size_t d_roff = dreserve((size_t)k.max_reads * 4), d_rlen = dreserve((size_t)k.max_reads * 4);
    const size_t d_hlen = dreserve((size_t)k.max_haps * 4), d_poff = dreserve((size_t)k.max_haps * 4), d_hoff = dreserve((size_t)k.max_haps * 4);
    const size_t d_pr = dreserve((size_t)k.max_pairs * 4), d_ph = dreserve((size_t)k.max_pairs * 4), d_order = dreserve((size_t)k.max_pairs * 4);
    hipError_t e = rb.blob.alloc(dtotal);
    if (e == hipSuccess) e = rb.planes.alloc(plane_words_cap(k));
    if (e == hipSuccess) e = rb.raw32.alloc(64);  // only the rescue counters live here: the sums go to the stage
    if (e == hipSuccess) e = rb.rescue_list.alloc(k.max_pairs);
    // PHMM_STAGE_BLOCKING_SYNC=1: phmm_stage_wait sleeps in the driver instead of polling (frees a core per waiting thread,
    // costs wake-up latency)
    const unsigned ev_flags = getenv("PHMM_STAGE_BLOCKING_SYNC") ? hipEventBlockingSync : hipEventDefault;
    if (e == hipSuccess) e = hipEventCreate(&st->ev0);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&st->ev1, ev_flags);
    if (e != hipSuccess) {
        impl::phmm_stage_destroy(c, st);
        return fail(PHMM_E_HIP, "phmm_stage_create: device buffers: %s", hipGetErrorString(e));
    }
    uint8_t *d = rb.blob.p;
    rb.read_bases = {d + d_arena, k.max_bytes};  // the read table keeps the arena's offsets
    rb.read_qual = {d + d_qual, k.max_bytes};
    rb.read_ins = {d + d_gop, k.max_bytes};
    rb.read_del = rb.read_ins;  // pairhmm_internal.cpp:425
    rb.read_gcp = {d + d_gcp, k.max_bytes};
    rb.read_off = {reinterpret_cast<uint32_t *>(d + d_roff), k.max_reads};
    rb.read_len = {reinterpret_cast<uint32_t *>(d + d_rlen), k.max_reads};
    rb.hap_len = {reinterpret_cast<uint32_t *>(d + d_hlen), k.max_haps};
    rb.plane_off = {reinterpret_cast<uint32_t *>(d + d_poff), k.max_haps};
    rb.hap_off = {reinterpret_cast<uint32_t *>(d + d_hoff), k.max_haps};  // written by the set-up launch: offsets in the arena
    rb.hap_bases = {d + d_arena, k.max_bytes};                             // (the haplotypes' bases lie in the arena)
    rb.pair_read = {reinterpret_cast<uint32_t *>(d + d_pr), k.max_pairs};
    rb.pair_hap = {reinterpret_cast<uint32_t *>(d + d_ph), k.max_pairs};
    rb.order = {reinterpret_cast<uint32_t *>(d + d_order), k.max_pairs};
    rb.rescue_count = {reinterpret_cast<uint32_t *>(rb.raw32.p), (size_t)kQueues};
    rb.out32_host = st->raw32;
    rb.out64_host = st->raw64;
    rb.uniform_gcp = true;  // read_record.cpp:205-210: one gap-continuation quality
    *out = st;
    return PHMM_OK;
}

int phmm_stage_get_view(const phmm_stage *st, phmm_stage_view *view, phmm_stage_caps *caps)
{
    if (!st || !view) return fail(PHMM_E_INVALID, "phmm_stage_get_view: null argument");
    *view = st->view;
    if (caps) *caps = st->caps;
    return PHMM_OK;
}

// the checks every backend makes; fills pair_first / slot_first and the launches
static int stage_plan(phmm_stage *st, uint32_t n_regions, uint32_t n_reads, uint32_t n_haps, std::vector<Bin> &bins, uint64_t *cells,
                      uint32_t *max_read, uint32_t *max_hap, uint32_t *used_bytes, bool ride_along)
{
    const phmm_stage_caps &k = st->caps;
    const phmm_stage_view &v = st->view;
    if (n_regions > k.max_regions || n_reads > k.max_reads || n_haps > k.max_haps)
        return fail(PHMM_E_INVALID, "phmm_stage_submit: %u regions / %u reads / %u haplotypes exceed the stage's capacity", n_regions, n_reads, n_haps);
    uint32_t mr = 0, mh = 0;
    uint64_t used = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint64_t n = v.read_len[r];
        if (n == 0) return fail(PHMM_E_INVALID, "phmm_stage_submit: read %u has length 0", r);
        if (v.read_off[r] + n > k.max_bytes || v.read_qual_off[r] + n > k.max_bytes)
            return fail(PHMM_E_INVALID, "phmm_stage_submit: read %u lies outside the stage's arena", r);
        used = std::max(used, std::max<uint64_t>(v.read_off[r], v.read_qual_off[r]) + n);
        mr = std::max(mr, v.read_len[r]);
    }
    for (uint32_t h = 0; h < n_haps; ++h) {
        if (v.hap_len[h] == 0) return fail(PHMM_E_INVALID, "phmm_stage_submit: haplotype %u has length 0", h);
        if ((uint64_t)v.hap_off[h] + v.hap_len[h] > k.max_bytes)
            return fail(PHMM_E_INVALID, "phmm_stage_submit: haplotype %u lies outside the stage's arena", h);
        used = std::max(used, (uint64_t)v.hap_off[h] + v.hap_len[h]);
        mh = std::max(mh, v.hap_len[h]);
    }
    *used_bytes = (uint32_t)used;
    if (mr > kMaxLen || mh > kMaxLen) return fail(PHMM_E_INVALID, "phmm_stage_submit: sequence longer than %u bases", kMaxLen);
    if (phmm::team_lds_bytes((uint32_t)phmm::kPlanes * phmm::planes_words(mh), 8, 8, 16) > 160u * 1024u)
        return fail(PHMM_E_INVALID, "phmm_stage_submit: haplotype of %u bases exceeds the LDS plane budget", mh);
    bins.clear();
    uint64_t n_pairs = 0;
    const int rc = phmm_detail::bin_regions(v.regions, n_regions, v.read_len, n_reads, v.hap_len, n_haps, st->slot_first.data(),
                                            st->pair_first.data(), bins, cells, &n_pairs, ride_along);
    if (rc != PHMM_OK) return rc;
    merge_striped_bins_of_one_shape(bins);
    if (n_pairs > k.max_pairs) return fail(PHMM_E_INVALID, "phmm_stage_submit: %llu pairs exceed the stage's capacity of %u", (unsigned long long)n_pairs, k.max_pairs);
    st->pair_first[n_regions] = (uint32_t)n_pairs;
    st->n_regions = n_regions;
    st->n_pairs = n_pairs;
    *max_read = mr;
    *max_hap = mh;
    return PHMM_OK;
}

// CPU backend: the stage's regions as a pair-list batch through pairhmm_cpu.cpp (H3 / H4 by the host routines)
static int cpu_stage_submit(phmm_context *c, phmm_stage *st, uint32_t n_regions, uint32_t n_reads, uint32_t n_haps, const phmm_stage_params *prm)
{
    std::vector<Bin> bins;
    uint64_t cells = 0;
    uint32_t mr = 0, mh = 0, used = 0;
    int rc = stage_plan(st, n_regions, n_reads, n_haps, bins, &cells, &mr, &mh, &used, true);
    if (rc != PHMM_OK) return rc;
    const phmm_stage_view &v = st->view;
    std::vector<uint32_t> pr(st->n_pairs), ph(st->n_pairs);
    for (uint32_t g = 0; g < n_regions; ++g) {
        const phmm_region &rg = v.regions[g];
        uint64_t p = st->pair_first[g];
        for (uint32_t r = 0; r < rg.n_reads; ++r)
            for (uint32_t h = 0; h < rg.n_haps; ++h, ++p) {
                pr[p] = rg.read_first + r;
                ph[p] = rg.hap_first + h;
            }
    }
    // a phmm_batch indexes bases and qualities with the same offsets: line the qualities up beside the bases
    std::vector<uint8_t> quals(used ? used : 1);
    for (uint32_t r = 0; r < n_reads; ++r) memcpy(quals.data() + v.read_off[r], v.bytes + v.read_qual_off[r], v.read_len[r]);
    phmm_read_prep prep{v.read_mapq, prm->min_quality_threshold, prm->pcr_model, prm->gap_open, prm->gap_cont};
    phmm_batch b{};
    b.n_reads = n_reads;
    b.n_haps = n_haps;
    b.n_pairs = st->n_pairs;
    b.read_bases = v.bytes;
    b.read_qual = quals.data();
    b.read_off = v.read_off;
    b.read_len = v.read_len;
    b.hap_bases = v.bytes;
    b.hap_off = v.hap_off;
    b.hap_len = v.hap_len;
    b.pair_read = pr.data();
    b.pair_hap = ph.data();
    b.prep = &prep;
    static const uint32_t none = 0;
    if (st->n_pairs == 0) b.pair_read = b.pair_hap = &none;
    rc = impl::cpu_upload(c, &b, &st->rb);
    if (rc == PHMM_OK) rc = impl::cpu_sweep(c, &st->rb);
    if (rc != PHMM_OK) return rc;
    const CpuBatch &k = *st->rb.cpu;
    for (uint64_t p = 0; p < st->n_pairs; ++p) {
        st->raw32[p] = k.raw32[p];
        st->raw64[p] = k.raw64[p];
    }
    st->valid = true;
    return PHMM_OK;
}

int phmm_stage_submit(phmm_context *c, phmm_stage *st, uint32_t n_regions, uint32_t n_reads, uint32_t n_haps, const phmm_stage_params *prm)
{
    if (!c || !st || !prm) return fail(PHMM_E_INVALID, "phmm_stage_submit: null argument");
    phmm_detail::HookScope hooks;
    if (prm->pcr_model < 0 || prm->pcr_model > 3) return fail(PHMM_E_INVALID, "phmm_stage_submit: bad PCR indel model %d", prm->pcr_model);
    if (c->cpu != st->cpu) return fail(PHMM_E_INVALID, "phmm_stage_submit: the stage belongs to the other kind of context");
    if (st->in_flight) return fail(PHMM_E_INVALID, "phmm_stage_submit: the previous submission has not been waited for");
    st->params = *prm;
    // whatever the last submission left behind describes THAT batch: a submission that fails from here on leaves the stage empty, so
    // that phmm_stage_log10 / _pair_first cannot hand out the previous batch's numbers as this one's
    st->valid = false;
    st->n_pairs = 0;
    st->n_regions = 0;
    if (c->cpu) return cpu_stage_submit(c, st, n_regions, n_reads, n_haps, prm);
    HIP_TRY(hipSetDevice(c->device));
    phmm_resident &rb = st->rb;
    uint32_t mr = 0, mh = 0, used = 0;
    rb.ragged = ragged_launches_for(true);  // a stage's reads are prepared on the device: always the Rovaca case
    int rc = stage_plan(st, n_regions, n_reads, n_haps, rb.bins, &rb.cells, &mr, &mh, &used, !rb.ragged);
    if (rc != PHMM_OK) return rc;
    const phmm_stage_view &v = st->view;
    // ---- descriptors, right behind the used part of the arena: 32 bytes per read, 16 per haplotype ----
    const size_t o_rdesc = ((size_t)used + 255) & ~size_t(255), o_hdesc = o_rdesc + (size_t)n_reads * sizeof(phmm::StageReadDesc);
    const size_t copy_bytes = o_hdesc + (size_t)n_haps * sizeof(phmm::StageHapDesc);
    phmm::StageReadDesc *read_desc = reinterpret_cast<phmm::StageReadDesc *>(v.bytes + o_rdesc);
    phmm::StageHapDesc *hap_desc = reinterpret_cast<phmm::StageHapDesc *>(v.bytes + o_hdesc);
    for (uint32_t r = 0; r < n_reads; ++r)
        read_desc[r] = phmm::StageReadDesc{v.read_off[r], v.read_len[r], v.read_qual_off[r], 0, 0, 0, 0, v.read_mapq[r]};
    for (uint32_t g = 0; g < n_regions; ++g) {
        const phmm_region &rg = v.regions[g];
        for (uint32_t i = 0; i < rg.n_reads; ++i) {
            phmm::StageReadDesc &d = read_desc[rg.read_first + i];
            d.slot_first = st->slot_first[rg.read_first + i];
            d.pair_first = st->pair_first[g] + i * rg.n_haps;
            d.hap_first = rg.hap_first;
            d.n_haps = rg.n_haps;
        }
    }
    uint64_t words = 0;
    for (uint32_t h = 0; h < n_haps; ++h) {
        hap_desc[h] = phmm::StageHapDesc{v.hap_off[h], v.hap_len[h], (uint32_t)(words * phmm::kPlanes), 0};
        words += phmm::planes_words(v.hap_len[h]);
    }
    if (words * phmm::kPlanes > rb.planes.cap) return fail(PHMM_E_INVALID, "phmm_stage_submit: haplotype planes exceed the stage's capacity");
    rb.n_pairs = st->n_pairs;
    rb.n_reads = n_reads;
    rb.n_haps = n_haps;
    rb.max_read_len = mr;
    rb.max_hap_len = mh;
    rb.uniform_gop = prm->pcr_model == 0 && !phmm_hook("PHMM_NO_UNIFORM_GOP");  // no PCR model: every base keeps gap_open
    rb.last_rescued = 0;
    if (st->n_pairs == 0) {
        st->valid = true;
        return PHMM_OK;
    }
    // ---- one copy, one set-up launch, then the passes of a resident batch ----
    uint8_t *d_arena = rb.blob.p;  // (the arena is the first thing in the blob)
    phmm::StageSetupArgs sa{};
    sa.arena = d_arena;
    sa.read_desc = reinterpret_cast<const phmm::StageReadDesc *>(d_arena + o_rdesc);
    sa.hap_desc = reinterpret_cast<const phmm::StageHapDesc *>(d_arena + o_hdesc);
    sa.read_qual = rb.read_qual.p;
    sa.read_gop = rb.read_ins.p;
    sa.read_gcp = rb.read_gcp.p;
    sa.read_off = rb.read_off.p;
    sa.read_len = rb.read_len.p;
    sa.hap_len = rb.hap_len.p;
    sa.plane_off = rb.plane_off.p;
    sa.hap_off = rb.hap_off.p;
    sa.planes = rb.planes.p;
    sa.pair_read = rb.pair_read.p;
    sa.pair_hap = rb.pair_hap.p;
    sa.order = rb.order.p;
    sa.n_reads = n_reads;
    sa.n_haps = n_haps;
    sa.min_q = prm->min_quality_threshold;
    sa.pcr_model = prm->pcr_model;
    sa.gap_open = prm->gap_open;
    sa.gap_cont = prm->gap_cont;
    sa.rescue_count = rb.rescue_count.p;
    // from the copy on, every failure goes through one exit that drains the stream: whatever was enqueued reads the arena and writes
    // the stage's results, and the caller is free to refill or destroy the stage as soon as this returns
    auto enqueue = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(d_arena, v.bytes, copy_bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(st->ev0, c->stream));
        HIP_TRY(phmm::launch_stage_setup(sa, c->stream));
        rb.queue_is_empty = true;
        int r = impl::sweep_fp32(c, &rb);
        // The fp64 re-run is phmm_stage_wait's business: it sees the fp32 sums in the stage's pinned memory and launches the re-run only
        // when one of them underflowed.  Enqueued here, behind every sweep, the re-run launch -- nearly always empty -- sat on every
        // submission's critical path (round 4; the one-shot pipeline's collector does the same).  PHMM_STAGE_EAGER_RESCUE=1: A/B hook.
        st->rescue_deferred = r == PHMM_OK && phmm_hook("PHMM_STAGE_EAGER_RESCUE") == nullptr;
        if (r == PHMM_OK && !st->rescue_deferred) r = impl::sweep_rescue(c, &rb);
        if (r != PHMM_OK) return r;
        HIP_TRY(hipEventRecord(st->ev1, c->stream));
        return PHMM_OK;
    };
    rc = enqueue();
    if (rc != PHMM_OK) {
        (void)hipStreamSynchronize(c->stream);
        for (int i = 0; i < c->n_side; ++i)
            if (c->side[i]) (void)hipStreamSynchronize(c->side[i]);
        st->n_pairs = 0;
        st->n_regions = 0;
        return rc;
    }
    st->in_flight = true;
    st->submitted_on = c;
    st->valid = true;
    return PHMM_OK;
}

int phmm_stage_wait(phmm_context *c, phmm_stage *st, phmm_stats *stats)
{
    if (!c || !st) return fail(PHMM_E_INVALID, "phmm_stage_wait: null argument");
    double ms = 0.0;
    if (!c->cpu && st->in_flight) {
        if (st->submitted_on != c)  // (the fp64 re-run below would take the other context's tables and streams)
            return fail(PHMM_E_INVALID, "phmm_stage_wait: the stage was submitted on another context");
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipEventSynchronize(st->ev1));
        if (st->rescue_deferred) {
            st->rescue_deferred = false;
            // (four sums per compare: the scan is on every submission's way to its workers -- 43 000 sums took 17 us one by one)
            bool any = false;
            uint64_t p = 0;
            {
                const __m128 thr = _mm_set1_ps(PHMM_MIN_ACCEPTED_F32);
                __m128 acc0 = _mm_setzero_ps(), acc1 = _mm_setzero_ps();
                for (; p + 8 <= st->n_pairs; p += 8) {  // (raw32 is 64-byte aligned: the stage's arena)
                    acc0 = _mm_or_ps(acc0, _mm_cmplt_ps(_mm_loadu_ps(st->raw32 + p), thr));
                    acc1 = _mm_or_ps(acc1, _mm_cmplt_ps(_mm_loadu_ps(st->raw32 + p + 4), thr));
                }
                any = _mm_movemask_ps(_mm_or_ps(acc0, acc1)) != 0;
            }
            for (; p < st->n_pairs && !any; ++p) any = st->raw32[p] < PHMM_MIN_ACCEPTED_F32;
            if (any) {  // rare: the queues the fp32 kernels filled are still in the stage's device memory
                int rc = impl::sweep_rescue(c, &st->rb);
                if (rc == PHMM_OK && hipEventRecord(st->ev1, c->stream) != hipSuccess) rc = fail(PHMM_E_HIP, "phmm_stage_wait: hipEventRecord failed");
                if (rc == PHMM_OK && hipEventSynchronize(st->ev1) != hipSuccess) rc = fail(PHMM_E_HIP, "phmm_stage_wait: the fp64 re-run failed");
                if (rc != PHMM_OK) {
                    (void)hipStreamSynchronize(c->stream);
                    st->in_flight = false;
                    st->valid = false;
                    return rc;
                }
            }
        }
        st->in_flight = false;
        if (stats) {
            float f = 0.f;
            HIP_TRY(hipEventElapsedTime(&f, st->ev0, st->ev1));
            ms = f;
        }
    }
    if (stats) {
        *stats = phmm_stats{};
        stats->cells = st->rb.cells;
        stats->kernel_ms = ms;
        stats->n_launches = st->rb.last_fp32_launches ? st->rb.last_fp32_launches : (uint32_t)st->rb.bins.size();
        stats->max_rows_per_lane = st->rb.bins.empty() ? 0 : (uint32_t)st->rb.bins.back().rows_per_lane;
    }
    return PHMM_OK;
}

uint64_t phmm_stage_pair_first(const phmm_stage *st, uint32_t region)
{
    if (!st || !st->valid || region > st->n_regions) return 0;
    return st->pair_first[region];
}

int phmm_stage_log10(const phmm_stage *st, uint64_t first_pair, uint64_t n, double *out, float *out_raw_f32, uint64_t *n_rescued)
{
    if (!st) return fail(PHMM_E_INVALID, "phmm_stage_log10: null argument");
    if (!st->valid) return fail(PHMM_E_INVALID, "phmm_stage_log10: the stage's last submission did not succeed (or there was none)");
    if (first_pair + n > st->n_pairs) return fail(PHMM_E_INVALID, "phmm_stage_log10: pairs [%llu, +%llu) of %llu", (unsigned long long)first_pair,
                                                   (unsigned long long)n, (unsigned long long)st->n_pairs);
    const float *r32 = st->raw32 + first_pair;
    const double *r64 = st->raw64 + first_pair;
    uint64_t resc = 0;
    if (out_raw_f32) memcpy(out_raw_f32, r32, n * sizeof(float));
    if (out)
        for (uint64_t p = 0; p < n; ++p) {  // pairhmm_internal.cpp:577-585 -- glibc on the host, as the reference does
            const float x = r32[p];
            if (x < PHMM_MIN_ACCEPTED_F32) {
                out[p] = log10(r64[p]) - PHMM_LOG10_INITIAL_CONSTANT_F64;
                ++resc;
            }
            else {
                out[p] = (double)(log10f(x) - PHMM_LOG10_INITIAL_CONSTANT_F32);
            }
        }
    else
        for (uint64_t p = 0; p < n; ++p) resc += r32[p] < PHMM_MIN_ACCEPTED_F32;
    if (n_rescued) *n_rescued = resc;
    return PHMM_OK;
}

// one-shot form over host buffers: the tables are copied into a stage the context keeps (grown as needed)
int phmm_compute_regions(phmm_context *c, const phmm_batch *t, const phmm_region *regions, uint32_t n_regions, float *out_raw_f32,
                         double *out_log10, phmm_stats *stats)
{
    if (!c || !t || (n_regions && !regions)) return fail(PHMM_E_INVALID, "phmm_compute_regions: null argument");
    const auto t0 = std::chrono::steady_clock::now();
    if (n_regions && (!t->read_bases || !t->read_qual || !t->read_off || !t->read_len || !t->hap_bases || !t->hap_off || !t->hap_len))
        return fail(PHMM_E_INVALID, "phmm_compute_regions: null table pointer");
    if (!t->prep) {
        // prepared read tables (five arrays): spell the pair list out and take the pair-list path
        uint64_t n = 0;
        for (uint32_t g = 0; g < n_regions; ++g) {
            if ((uint64_t)regions[g].read_first + regions[g].n_reads > t->n_reads || (uint64_t)regions[g].hap_first + regions[g].n_haps > t->n_haps)
                return fail(PHMM_E_INVALID, "phmm_compute_regions: region %u leaves the tables", g);
            n += (uint64_t)regions[g].n_reads * regions[g].n_haps;
        }
        if (n > 0xFFFFFFF0ull) return fail(PHMM_E_INVALID, "phmm_compute_regions: more than 2^32 pairs in one batch");
        std::vector<uint32_t> pr(n ? n : 1), ph(n ? n : 1);
        uint64_t p = 0;
        for (uint32_t g = 0; g < n_regions; ++g)
            for (uint32_t r = 0; r < regions[g].n_reads; ++r)
                for (uint32_t h = 0; h < regions[g].n_haps; ++h, ++p) {
                    pr[p] = regions[g].read_first + r;
                    ph[p] = regions[g].hap_first + h;
                }
        phmm_batch b = *t;
        b.n_pairs = n;
        b.pair_read = pr.data();
        b.pair_hap = ph.data();
        return impl::phmm_compute(c, &b, out_raw_f32, out_log10, stats);
    }
    if (!t->prep->read_mapq || t->prep->pcr_model < 0 || t->prep->pcr_model > 3) return fail(PHMM_E_INVALID, "phmm_compute_regions: bad phmm_read_prep");
    uint64_t read_bytes = 0, hap_bytes = 0, n_pairs = 0;
    for (uint32_t r = 0; r < t->n_reads; ++r) read_bytes = std::max<uint64_t>(read_bytes, (uint64_t)t->read_off[r] + t->read_len[r]);
    for (uint32_t h = 0; h < t->n_haps; ++h) hap_bytes = std::max<uint64_t>(hap_bytes, (uint64_t)t->hap_off[h] + t->hap_len[h]);
    for (uint32_t g = 0; g < n_regions; ++g) n_pairs += (uint64_t)regions[g].n_reads * regions[g].n_haps;
    const uint64_t bytes = 2 * read_bytes + hap_bytes;  // arena: [bases][qualities][haplotypes]
    if (bytes > 0x7FFFFFF0ull || n_pairs > 0x7FFFFFF0ull) return fail(PHMM_E_INVALID, "phmm_compute_regions: batch too large for one stage");
    phmm_stage *st = c->region_stage;
    if (!st || st->caps.max_regions < n_regions || st->caps.max_reads < t->n_reads || st->caps.max_haps < t->n_haps ||
        st->caps.max_bytes < bytes || st->caps.max_pairs < n_pairs) {
        impl::phmm_stage_destroy(c, st);
        c->region_stage = nullptr;
        auto grow = [](uint64_t x) { return (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, std::max<uint64_t>(x + x / 4, 64)); };
        const phmm_stage_caps caps{grow(n_regions), grow(t->n_reads), grow(t->n_haps), grow(bytes), grow(n_pairs)};
        const int rc = impl::phmm_stage_create(c, &caps, &c->region_stage);
        if (rc != PHMM_OK) return rc;
        st = c->region_stage;
    }
    const phmm_stage_view &v = st->view;
    if (read_bytes) {
        memcpy(v.bytes, t->read_bases, read_bytes);
        memcpy(v.bytes + read_bytes, t->read_qual, read_bytes);
    }
    for (uint32_t r = 0; r < t->n_reads; ++r) {
        v.read_off[r] = t->read_off[r];
        v.read_qual_off[r] = (uint32_t)read_bytes + t->read_off[r];
        v.read_len[r] = t->read_len[r];
    }
    if (t->n_reads) memcpy(v.read_mapq, t->prep->read_mapq, t->n_reads);
    if (hap_bytes) memcpy(v.bytes + 2 * read_bytes, t->hap_bases, hap_bytes);
    for (uint32_t h = 0; h < t->n_haps; ++h) {
        v.hap_off[h] = (uint32_t)(2 * read_bytes) + t->hap_off[h];
        v.hap_len[h] = t->hap_len[h];
    }
    if (n_regions) memcpy(v.regions, regions, (size_t)n_regions * sizeof(phmm_region));
    const phmm_stage_params prm{t->prep->min_quality_threshold, t->prep->pcr_model, t->prep->gap_open, t->prep->gap_cont};
    int rc = impl::phmm_stage_submit(c, st, n_regions, t->n_reads, t->n_haps, &prm);
    if (rc == PHMM_OK) rc = impl::phmm_stage_wait(c, st, stats);
    uint64_t resc = 0;
    if (rc == PHMM_OK) rc = impl::phmm_stage_log10(st, 0, st->n_pairs, out_log10, out_raw_f32, &resc);
    if (rc == PHMM_OK && stats) {
        stats->n_rescued = resc;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}

}  // namespace impl
