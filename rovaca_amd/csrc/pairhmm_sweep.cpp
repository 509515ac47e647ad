// pairhmm_sweep.cpp -- one pass over a resident batch: the fp32 launches (single-pass, ragged, striped / band kernels), the fp64 re-run of
// what underflowed, and the choice of order for long pairs ("fp64 first").  Host C++ over the HIP runtime; called by phmm_sweep, the
// one-shot paths, the chunk pipeline (pairhmm_chunked.cpp) and the stages (pairhmm_stage.cpp).  Part of pairhmm_capi.cpp until round 4.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/pairhmm_hip.h"
#include "pairhmm_device.h"

#include "pairhmm_engine.h"

using phmm_detail::g_err;

namespace impl
{
// fp64 first (sweep_fp32): what the batch's last finished pass observed on its striped ranges, if the counters are back
void harvest_history(phmm_context *c, phmm_resident *rb)
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
            if (!(qi >= 1 && qi <= 4) || !rb->queue_sparse[qi] || rb->out32_host || rb->fp32_first_only) continue;
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
        bool tab;         // ragged launches: the table sweep (pairhmm_systolic_tab.h) instead of the bit-plane one
    };
    // The table sweep (round 5): the Rovaca case (one gap-continuation quality, deletion == insertion gap-opens), 16 lanes per pair,
    // the haplotypes' bases on the device, and LDS for at least kTabMinSlots prior tables per workgroup next to its 16 haplotypes
    // (haplotypes of a thousand bases and more leave no room: bit-plane kernels).  PHMM_NO_TAB: A/B and test hook.
    const bool tab_on = rb->uniform_gcp && rb->hap_bases.p && rb->hap_off.p && !phmm_hook("PHMM_NO_TAB");
    auto tab_slots_of = [&](int k16, uint32_t max_hap) -> uint32_t {
        if (!tab_on || k16 < 1 || k16 > phmm::kTabMaxRows) return 0u;
        const uint32_t n = phmm::tab_slots_for(k16, max_hap, phmm::tab_lds_budget(phmm::kTabWaves));
        if (const char *f = phmm_hook("PHMM_TAB_SLOTS"))  // tests: few tables per workgroup, so that workgroups sweep in rounds
            return n ? std::min<uint32_t>(n, (uint32_t)std::max(1, atoi(f))) : 0u;
        return n >= phmm::kTabMinSlots ? n : 0u;
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
            launches.push_back(FpLaunch{i, 1, 0, false});
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
        // the run as table sweeps: every class that would run 16 lanes per pair must have room for its tables (the LDS of a launch is
        // sized by the run's longest haplotype)
        uint32_t run_max_hap = 0;
        for (size_t q = i; q < j; ++q) run_max_hap = std::max(run_max_hap, rb->bins[q].max_hap_len);
        // as table sweeps all classes share one kernel at one register budget, whatever the size of the run: one launch (in pieces of
        // kRaggedMaxSegs classes)
        bool tab = tab_on && lanes == 16;
        for (size_t q = i; q < j && tab; ++q)
            if (!tab_slots_of(rb->bins[q].rows_per_lane, run_max_hap)) tab = false;
        if (tab) {
            for (size_t k = i; k < j; k += (size_t)phmm::kRaggedMaxSegs)
                launches.push_back(FpLaunch{k, std::min(j - k, (size_t)phmm::kRaggedMaxSegs), 16, true});
            i = j;
            continue;
        }
        if (lanes == 16 && total < uniform_below && j - i > 1 && j - i <= (size_t)phmm::kRaggedMaxSegs) {
            launches.push_back(FpLaunch{i, j - i, -1, false});
            i = j;
            continue;
        }
        for (size_t k = i; k < j;) {
            const int range = phmm::ragged_range_of(lanes == 64 ? wide_rows(rb->bins[k]) : rb->bins[k].rows_per_lane, lanes, rb->uniform_gop);
            size_t e = k + 1;
            while (e < j && e - k < (size_t)phmm::kRaggedMaxSegs &&
                   phmm::ragged_range_of(lanes == 64 ? wide_rows(rb->bins[e]) : rb->bins[e].rows_per_lane, lanes, rb->uniform_gop) == range)
                ++e;
            launches.push_back(FpLaunch{k, e - k, lanes, false});
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
            uint32_t launch_max_hap = 0;
            for (size_t q = 0; q < L.n; ++q) launch_max_hap = std::max(launch_max_hap, rb->bins[L.first + q].max_hap_len);
            for (size_t q = L.n; q-- > 0;) {
                const Bin &sb = rb->bins[L.first + q];
                phmm::RaggedSeg &sg = m.seg[m.n_seg++];
                const int seg_lanes = uniform ? phmm::ragged_uniform_lanes_of(sb.rows_per_lane, rb->uniform_gop) : L.lanes;
                sg.tab_slots = L.tab && seg_lanes == 16 ? tab_slots_of(sb.rows_per_lane, launch_max_hap) : 0u;
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
            if (L.tab) {
                a.hap_bases = rb->hap_bases.p;
                a.hap_off = rb->hap_off.p;
                a.tab_max_hap = launch_max_hap;
            }
            if (uniform)
                HIP_TRY(phmm::launch_systolic_ragged_uniform_f32(rb->uniform_gop, a, m, stream));
            else
                HIP_TRY(L.tab ? phmm::launch_systolic_ragged_tab_f32(rb->uniform_gop, a, m, stream)
                              : phmm::launch_systolic_ragged_f32(L.lanes, rb->uniform_gop, a, m, stream));
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
            const uint32_t tab_slots = lanes == 16 ? tab_slots_of(k, bin.max_hap_len) : 0u;
            if (tab_slots) {
                a.hap_bases = rb->hap_bases.p;
                a.hap_off = rb->hap_off.p;
                a.tab_max_hap = bin.max_hap_len;
                a.tab_slots = tab_slots;
                HIP_TRY(phmm::launch_systolic_tab_f32(k, rb->uniform_gop, a, stream));
            }
            else
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
        // the fp64 band kernel's table form (round 5): needs the haplotypes' bases; PHMM_NO_BAND_TAB: A/B and test hook
        a.hap_bases = rb->hap_bases.p;
        a.hap_off = rb->hap_off.p;
        a.tab_max_hap = L.max_hap;
        a.band_tab = rb->hap_bases.p && rb->hap_off.p && !phmm_hook("PHMM_NO_BAND_TAB") ? 1u : 0u;
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
                b.band_tab = 0;
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
    // (not from a chunk slot of the one-shot pipeline: its passes always run fp32 first and its collector thread is not the one that
    // reads the history)
    for (int qi = 1; qi <= 4; ++qi) observe = observe || (rb->queue_sparse[qi] && q[qi].cap && !rb->out32_host && !rb->fp32_first_only);
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

}  // namespace impl
