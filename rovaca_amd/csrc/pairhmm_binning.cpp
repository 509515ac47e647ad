// pairhmm_binning.cpp -- see pairhmm_binning.h.  Host code only: no HIP runtime call in this file.
#include "pairhmm_binning.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace phmm_detail
{
int fail(int code, const char *fmt, ...);  // pairhmm_capi.cpp: sets the calling thread's phmm_last_error text
}
using phmm_detail::fail;

namespace phmm_detail
{
thread_local int t_hooks = -1;
thread_local HostPool *t_pool = nullptr;
thread_local bool t_serial = false;
bool test_hooks_enabled()
{
    const char *m = getenv("PHMM_ENABLE_TEST_HOOKS");
    return m && m[0] && !(m[0] == '0' && m[1] == 0);
}
}  // namespace phmm_detail
const char *phmm_hook(const char *name)
{
    if (phmm_detail::t_hooks == 0) return nullptr;
    if (phmm_detail::t_hooks < 0 && !phmm_detail::test_hooks_enabled()) return nullptr;
    return getenv(name);
}

bool ragged_launches_for(bool uniform_gcp) { return uniform_gcp && phmm_hook("PHMM_NO_RAGGED") == nullptr; }
bool queue_per_class_for(uint64_t n_pairs) { return n_pairs > kQueuePerClassMinPairs || phmm_hook("PHMM_FORCE_QUEUE_PER_CLASS") != nullptr; }

namespace phmm_detail
{

// Small launches ride along.  A launch of a few hundred pairs costs a lone wave's whole sweep (50-90 us) whatever its size, and a
// batch of mixed read lengths has one launch per rows-per-lane value: a stage of seven regions whose reads were trimmed to anything
// between 50 and 151 bases is seven launches, six of them nearly empty (measured through the adapter: regions of mixed shape ran at a
// third of the rate of uniform ones).  So a single-pass bin with fewer than kMergeBelowPairs pairs (two rounds of
// the chip's wave slots) is handed to the next bin up: its
// pairs run with that bin's rows per lane -- the kernels pad a shorter read with boundary rows in front, as they do inside a class --
// at the price of idle rows that a launch this small could not have filled the chip with anyway.  A bin that is not small is never
// given more rows per lane, and batches big enough for per-class rescue queues are left alone.
// Threshold, measured (MI355X): through the adapter on regions of mixed shape, 30 synchronous workers 42 -> 70 / 87 / 87 / 90 k regions/s for
// 1 024 / 2 048 / 4 096 / 8 192 (no gain beyond), 30 x 4 in flight 94 -> 99 / 113 / 118 / 126; resident sweeps of 30 000 pairs with reads of
// 50-250 bases 2 250 -> 2 630 (2 048) -> 3 010 GCUPS (8 192), of 50 000 pairs 3 230 -> 3 270 -> 3 590.
constexpr uint32_t kMergeBelowPairs = 8192;
static void merge_small_bins(std::vector<Bin> &bins, bool queue_per_class)
{
    if (queue_per_class || bins.size() < 2 || phmm_hook("PHMM_NO_BIN_MERGE")) return;
    uint32_t below = kMergeBelowPairs;
    if (const char *f = phmm_hook("PHMM_MERGE_BELOW_PAIRS")) below = (uint32_t)std::max(0, atoi(f));  // tuning
    std::vector<Bin> out;
    out.reserve(bins.size());
    bool open = false;  // out.back() is a single-pass bin that the next single-pass bin may absorb
    for (const Bin &b : bins) {
        if (b.rows_per_lane > 0 && open && out.back().count < below && out.back().first + out.back().count == b.first &&
            out.back().queue == b.queue) {
            Bin &a = out.back();  // a's pairs come first in `order`; they take b's (larger) rows per lane
            a.rows_per_lane = std::max(a.rows_per_lane, b.rows_per_lane);
            a.count += b.count;
            a.lds_words_per_group = std::max(a.lds_words_per_group, b.lds_words_per_group);
            a.max_hap_len = std::max(a.max_hap_len, b.max_hap_len);
            a.max_read_len = std::max(a.max_read_len, b.max_read_len);
            continue;
        }
        out.push_back(b);
        open = b.rows_per_lane > 0;
    }
    bins.swap(out);
}

int bin_pairs(const phmm_batch *b, bool cross, uint64_t n_pairs, uint32_t read_bias, uint32_t hap_bias, uint32_t *pr, uint32_t *ph,
              uint32_t *order, BinScratch &scratch, std::vector<Bin> &bins, uint64_t *cells_out, bool ride_along)
{
    const bool queue_per_class = queue_per_class_for(n_pairs);
    uint32_t max_hap = 0;
    for (uint32_t h = 0; h < b->n_haps; ++h) max_hap = std::max(max_hap, b->hap_len[h]);
    uint32_t min_hap = max_hap;
    for (uint32_t h = 0; h < b->n_haps; ++h) min_hap = std::min(min_hap, b->hap_len[h]);
    const uint32_t hrange = b->n_haps ? max_hap - min_hap + 1 : 1;
    uint32_t rank[kClasses], n_used = 0, class_rmax[kClasses];
    {
        bool used[kClasses] = {false};
        const bool long_hap = max_hap > kSinglePassMaxHap;
        for (uint32_t r = 0; r < b->n_reads; ++r) {
            used[class_of(b->read_len[r], 1)] = true;
            if (long_hap) used[class_of(b->read_len[r], max_hap)] = true;
        }
        for (uint32_t k = 0; k < kClasses; ++k) {
            rank[k] = n_used;
            class_rmax[k] = 0;
            if (used[k]) ++n_used;
        }
    }
    uint64_t cells = 0;
    const uint64_t table = (uint64_t)n_used * hrange;
    // the length key of a pair: the longest haplotype its read is paired with (pairhmm_binning.h).  A cross product pairs every read with
    // every haplotype; a pair list takes one pass (bad indices are reported by the counting pass below).
    std::vector<uint32_t> &read_hmax = scratch.read_hmax;
    if (!cross) {
        read_hmax.assign(b->n_reads, 0);
        const unsigned T0 = (n_pairs >= (phmm_detail::t_pool ? (1u << 15) : kParallelMinPairs) && !phmm_detail::t_serial) ? host_threads() : 1u;
        auto max_range = [&](uint64_t lo, uint64_t hi, unsigned) {
            for (uint64_t p = lo; p < hi; ++p) {
                const uint32_t r = b->pair_read[p] - read_bias, h = b->pair_hap[p] - hap_bias;
                if (r >= b->n_reads || h >= b->n_haps) continue;
                const uint32_t H = b->hap_len[h];
                uint32_t cur = __atomic_load_n(&read_hmax[r], __ATOMIC_RELAXED);
                while (cur < H && !__atomic_compare_exchange_n(&read_hmax[r], &cur, H, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
            }
        };
        if (T0 > 1)
            parallel_ranges(n_pairs, T0, max_range);
        else
            max_range(0, n_pairs, 0);
    }
    auto key_len = [&](uint32_t r) { return cross ? max_hap : read_hmax[r]; };
    const bool bin_trace = n_pairs >= 4096 && getenv("PHMM_PHASE_TIMING") != nullptr;  // tuning aid: where the binning's time goes, on stderr
    auto t_prev = std::chrono::steady_clock::now();
    double lap_us[6] = {0};
    auto lap = [&](int i) {
        if (!bin_trace) return;
        const auto n = std::chrono::steady_clock::now();
        lap_us[i] += std::chrono::duration<double, std::micro>(n - t_prev).count();
        t_prev = n;
    };
    lap(0);
    if (table <= std::max<uint64_t>(4 * n_pairs, 1u << 16)) {
        std::vector<uint32_t> &hist = scratch.hist, &slot = scratch.slot;
        hist.assign(table + 1, 0);
        slot.resize(n_pairs);
        // (with a helper pool current -- the chunked call -- handing out parts costs microseconds, so chunks of tens of thousands of pairs
        // are split too: a chunk of 118 000 mixed-length pairs took 0.5-1.0 ms to bin on one thread, longer than its sweep)
        const uint64_t par_min = phmm_detail::t_pool ? (1u << 15) : kParallelMinPairs;
        const unsigned T = (n_pairs >= par_min && table <= (1u << 16) && !phmm_detail::t_serial) ? host_threads() : 1u;
        std::vector<std::vector<uint32_t>> th_hist(T > 1 ? T : 0);
        std::vector<uint64_t> th_cells(T, 0);
        std::vector<uint32_t> th_rmax((size_t)T * kClasses, 0);
        std::atomic<uint64_t> bad_pair{UINT64_MAX};
        auto count_range = [&](uint64_t lo, uint64_t hi, unsigned t) {
            uint32_t *h_t = T > 1 ? th_hist[t].data() : hist.data() + 1;  // one thread counts straight into hist[at + 1]
            uint32_t *rmax_t = th_rmax.data() + (size_t)t * kClasses;
            uint64_t cells_t = 0;
            for (uint64_t p = lo; p < hi; ++p) {
                uint32_t r, h;
                if (cross) {
                    r = (uint32_t)(p / b->n_haps);
                    h = (uint32_t)(p % b->n_haps);
                }
                else {
                    r = b->pair_read[p] - read_bias;  // (an index below the bias wraps and fails the range check)
                    h = b->pair_hap[p] - hap_bias;
                    if (r >= b->n_reads || h >= b->n_haps) {
                        uint64_t expect = UINT64_MAX;
                        bad_pair.compare_exchange_strong(expect, p);
                        return;
                    }
                }
                const uint32_t R = b->read_len[r], H = b->hap_len[h];
                const uint32_t k = class_of(R, H);
                rmax_t[k] = std::max(rmax_t[k], R);
                const uint32_t at = rank[k] * hrange + (max_hap - key_len(r));
                pr[p] = r;
                ph[p] = h;
                slot[p] = at;
                ++h_t[at];
                cells_t += (uint64_t)R * H;
            }
            th_cells[t] = cells_t;
        };
        lap(1);
        if (T > 1) {
            for (auto &v : th_hist) v.assign(table, 0);
            parallel_ranges(n_pairs, T, count_range);
            lap(2);
            for (unsigned t = 0; t < T; ++t)
                for (uint64_t i = 0; i < table; ++i) hist[i + 1] += th_hist[t][i];
        }
        else {
            count_range(0, n_pairs, 0);
            lap(2);
        }
        if (bad_pair.load() != UINT64_MAX)
            return fail(PHMM_E_INVALID, "phmm_upload: pair %llu indexes outside the tables", (unsigned long long)bad_pair.load());
        for (unsigned t = 0; t < T; ++t) {
            cells += th_cells[t];
            for (uint32_t k = 0; k < kClasses; ++k) class_rmax[k] = std::max(class_rmax[k], th_rmax[(size_t)t * kClasses + k]);
        }
        for (uint64_t i = 0; i < table; ++i) hist[i + 1] += hist[i];  // hist[i] = first position of slot i
        // bins: classes 1..16 one each, the striped classes in four size ranges
        for (uint32_t k = 1; k < kClasses;) {
            uint32_t k_end = k + 1;
            while (k_end < kClasses && bin_id_of_class(k_end) == bin_id_of_class(k)) ++k_end;
            const uint32_t lo = rank[k], hi = k_end < kClasses ? rank[k_end] : n_used;
            const uint32_t first = hist[(uint64_t)lo * hrange], end = hist[(uint64_t)hi * hrange];
            if (end > first) {
                uint32_t hmax = 0, rmax = 0;
                for (uint32_t kk = k; kk < k_end; ++kk) rmax = std::max(rmax, class_rmax[kk]);
                for (uint32_t rk = lo; rk < hi; ++rk)  // the longest haplotype of the bin: first occupied slot of a class
                    for (uint32_t d = 0; d < hrange; ++d)
                        if (hist[(uint64_t)rk * hrange + d + 1] > hist[(uint64_t)rk * hrange + d]) {
                            hmax = std::max(hmax, max_hap - d);
                            break;
                        }
                const uint32_t plane_words = (uint32_t)phmm::kPlanes * phmm::planes_words(hmax);
                const int id = bin_id_of_class(k);
                bins.push_back(Bin{id < 17 ? id : 0, first, end - first, plane_words, hmax, rmax, queue_of_bin(id, queue_per_class)});
            }
            k = k_end;
        }
        lap(3);
        if (T > 1) {
            // stable scatter: inside a slot thread 0's pairs come first, then thread 1's, ... -- the order one thread gives
            for (uint64_t i = 0; i < table; ++i) {
                uint32_t run = hist[i];
                for (unsigned t = 0; t < T; ++t) {
                    const uint32_t cnt = th_hist[t][i];
                    th_hist[t][i] = run;
                    run += cnt;
                }
            }
            lap(4);
            parallel_ranges(n_pairs, T, [&](uint64_t lo, uint64_t hi, unsigned t) {
                uint32_t *start = th_hist[t].data();
                for (uint64_t p = lo; p < hi; ++p) order[start[slot[p]]++] = (uint32_t)p;
            });
        }
        else {
            for (uint64_t p = 0; p < n_pairs; ++p) order[hist[slot[p]]++] = (uint32_t)p;
        }
        lap(5);
        if (bin_trace)
            fprintf(stderr, "phmm bin: %llu pairs, table %llu, %u threads: classes %.0f alloc %.0f count %.0f merge+bins %.0f offsets %.0f scatter %.0f us\n",
                    (unsigned long long)n_pairs, (unsigned long long)table, T, lap_us[0], lap_us[1], lap_us[2], lap_us[3], lap_us[4], lap_us[5]);
    }
    else {
        // sparse length distribution: comparison sort on (class, length, index)
        std::vector<uint64_t> keys(n_pairs);
        for (uint64_t p = 0; p < n_pairs; ++p) {
            uint32_t r, h;
            if (cross) {
                r = (uint32_t)(p / b->n_haps);
                h = (uint32_t)(p % b->n_haps);
            }
            else {
                r = b->pair_read[p] - read_bias;
                h = b->pair_hap[p] - hap_bias;
                if (r >= b->n_reads || h >= b->n_haps)
                    return fail(PHMM_E_INVALID, "phmm_upload: pair %llu indexes outside the tables", (unsigned long long)p);
            }
            pr[p] = r;
            ph[p] = h;
            const uint32_t R = b->read_len[r], H = b->hap_len[h];
            cells += (uint64_t)R * H;
            keys[p] = ((uint64_t)class_of(R, H) << 56) | ((uint64_t)(0xFFFFFu - key_len(r)) << 32) | p;  // H < 2^20
        }
        std::sort(keys.begin(), keys.end());
        for (uint64_t i = 0; i < n_pairs; ++i) order[i] = (uint32_t)(keys[i] & 0xFFFFFFFFull);
        for (uint64_t i = 0; i < n_pairs;) {
            const int id = bin_id_of_class((uint32_t)(keys[i] >> 56));
            uint64_t j = i;
            uint32_t hmax = 0, rmax = 0;
            while (j < n_pairs && bin_id_of_class((uint32_t)(keys[j] >> 56)) == id) {
                hmax = std::max(hmax, b->hap_len[ph[order[j]]]);
                rmax = std::max(rmax, b->read_len[pr[order[j]]]);
                ++j;
            }
            const uint32_t plane_words = (uint32_t)phmm::kPlanes * phmm::planes_words(hmax);
            bins.push_back(Bin{id < 17 ? id : 0, (uint32_t)i, (uint32_t)(j - i), plane_words, hmax, rmax, queue_of_bin(id, queue_per_class)});
            i = j;
        }
    }
    if (ride_along) merge_small_bins(bins, queue_per_class);
    *cells_out = cells;
    return PHMM_OK;
}

int bin_regions(const phmm_region *regions, uint32_t n_regions, const uint32_t *read_len, uint32_t n_reads, const uint32_t *hap_len,
                uint32_t n_haps, uint32_t *slot_first, uint32_t *pair_first, std::vector<Bin> &bins, uint64_t *cells_out, uint64_t *n_pairs_out,
                bool ride_along)
{
    // pass 1: validate, count the pairs of every class, note each class's longest read and haplotype
    uint64_t count[kClasses] = {0};
    uint32_t class_rmax[kClasses] = {0}, class_hmax[kClasses] = {0};
    uint64_t n_pairs = 0, cells = 0;
    uint32_t next_read = 0;
    for (uint32_t g = 0; g < n_regions; ++g) {
        const phmm_region &rg = regions[g];
        if (rg.n_reads == 0 || rg.n_haps == 0) return fail(PHMM_E_INVALID, "region %u is empty", g);
        if (rg.read_first < next_read || (uint64_t)rg.read_first + rg.n_reads > n_reads)
            return fail(PHMM_E_INVALID, "region %u: reads [%u, +%u) overlap the region before or leave the table of %u", g, rg.read_first,
                        rg.n_reads, n_reads);
        if ((uint64_t)rg.hap_first + rg.n_haps > n_haps)
            return fail(PHMM_E_INVALID, "region %u: haplotypes [%u, +%u) leave the table of %u", g, rg.hap_first, rg.n_haps, n_haps);
        next_read = rg.read_first + rg.n_reads;
        uint32_t hmax = 0;
        uint64_t hsum = 0;
        for (uint32_t h = rg.hap_first; h < rg.hap_first + rg.n_haps; ++h) {
            hmax = std::max(hmax, hap_len[h]);
            hsum += hap_len[h];
        }
        pair_first[g] = (uint32_t)n_pairs;
        n_pairs += (uint64_t)rg.n_reads * rg.n_haps;
        if (n_pairs > 0xFFFFFFF0ull) return fail(PHMM_E_INVALID, "more than 2^32 pairs in one batch");
        for (uint32_t r = rg.read_first; r < rg.read_first + rg.n_reads; ++r) {
            const uint32_t k = class_of(read_len[r], hmax);  // (one class per read: a long haplotype sends the whole region to the striped kernel)
            count[k] += rg.n_haps;
            class_rmax[k] = std::max(class_rmax[k], read_len[r]);
            class_hmax[k] = std::max(class_hmax[k], hmax);
            cells += (uint64_t)read_len[r] * hsum;
        }
    }
    // launches: classes 1..16 one each, the striped classes in four size ranges -- the rule of bin_pairs
    const bool queue_per_class = queue_per_class_for(n_pairs);
    uint32_t cursor[kClasses];
    uint64_t run = 0;
    for (uint32_t k = 0; k < kClasses; ++k) {
        cursor[k] = (uint32_t)run;
        run += count[k];
    }
    for (uint32_t k = 1; k < kClasses;) {
        uint32_t k_end = k + 1;
        while (k_end < kClasses && bin_id_of_class(k_end) == bin_id_of_class(k)) ++k_end;
        uint64_t n = 0;
        uint32_t hmax = 0, rmax = 0;
        for (uint32_t kk = k; kk < k_end; ++kk) {
            n += count[kk];
            hmax = std::max(hmax, class_hmax[kk]);
            rmax = std::max(rmax, class_rmax[kk]);
        }
        if (n) {
            const int id = bin_id_of_class(k);
            bins.push_back(Bin{id < 17 ? id : 0, cursor[k], (uint32_t)n, (uint32_t)phmm::kPlanes * phmm::planes_words(hmax), hmax, rmax,
                               queue_of_bin(id, queue_per_class)});
        }
        k = k_end;
    }
    // pass 2: every read's run of `order`
    for (uint32_t g = 0; g < n_regions; ++g) {
        const phmm_region &rg = regions[g];
        uint32_t hmax = 0;
        for (uint32_t h = rg.hap_first; h < rg.hap_first + rg.n_haps; ++h) hmax = std::max(hmax, hap_len[h]);
        for (uint32_t r = rg.read_first; r < rg.read_first + rg.n_reads; ++r) {
            const uint32_t k = class_of(read_len[r], hmax);
            slot_first[r] = cursor[k];
            cursor[k] += rg.n_haps;
        }
    }
    if (ride_along) merge_small_bins(bins, queue_per_class);
    *cells_out = cells;
    *n_pairs_out = n_pairs;
    return PHMM_OK;
}

}  // namespace phmm_detail
