// adapter_bench.cpp -- end-to-end throughput of the drop-in adapter: N worker threads call rovaca::call_pairhmm on
// synthetic regions (as haplotypecaller_engine.cpp:119-285 does from its asio pool), through the cross-region batcher.
// Two forms: a library call (librovaca_adapter_bench.so: rovaca_amd_adapter_bench, which bench.py loads -- its `adapter` leg -- with
// regions cut from the same synthetic generator as the resident batch) and a command line over regions made here.
// Build: make -C rovaca_amd/host bench   Run on the GPU box: rovaca_amd/lib/adapter_bench [threads] [regions/thread] [reads] [haps] [sw] [regions in flight per worker]
// sw = 1: after PairHMM every surviving read is realigned to its best haplotype through rovaca::sw_hip_align_batch
// (haplotypecaller_engine.cpp:264), one call per region.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <ctime>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <memory_resource>
#include <random>
#include <thread>
#include <vector>

#include "../rovaca_amd/host/pairhmm_gpu_engine.h"
#include "../rovaca_amd/host/sw_gpu_engine.h"

using namespace rovaca;

struct Region
{
    std::vector<ReadRecord> reads;
    std::vector<Haplotype> haps;
    uint64_t cells = 0;
};

// ---- the measurement ----
struct adapter_bench_args
{
    // regions as flat tables: region g owns reads [region_read[g], region_read[g+1]) and haplotypes [region_hap[g], region_hap[g+1])
    const uint8_t *read_bases, *read_qual;  // raw base qualities
    const uint32_t *read_off, *read_len;
    const uint8_t *read_mapq;
    const uint8_t *hap_bases;
    const uint32_t *hap_off, *hap_len;
    const uint32_t *region_read, *region_hap;
    uint32_t n_regions;
    int32_t threads, regions_per_thread, depth, with_sw;  // depth: regions a worker keeps in flight (1 = the synchronous seam)
    int32_t min_q, pcr_model;
    // optional: the matrices of the first `check_regions` regions as the synchronous seam returns them, [hap][read] with the
    // original row width (tail NaN), one after the other in check_out; check_kept[g] = surviving reads
    uint32_t check_regions;
    double *check_out;
    uint32_t *check_kept;
};
struct adapter_bench_result
{
    double seconds, regions_per_s, gcups, sw_gcups;
    uint64_t regions, cells, kept_reads, submissions, batched_regions;
    double cpu_seconds;       // CPU time of the whole process over the timed region (workers + dispatchers + HIP's own threads)
    double phase_seconds[16]; // gpu_pairhmm_profile over the timed region
    int32_t n_phases;
};

static double process_cpu_seconds()
{
    timespec ts;
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// bench.py's headline must not die with this leg: the adapter follows the reference's convention on an unusable device -- log and
// exit() (pairhmm_engine.cpp:32-35), from a worker thread that no caller's try/except sees.  bench.py hands its record as it stands
// over before the leg and takes it back after; if the process exits in between, the record is written to stdout on the way out.
static std::atomic<char *> g_held_line{nullptr};
static void emit_held_line()
{
    if (char *l = g_held_line.exchange(nullptr)) {
        fputs(l, stdout);
        fputc('\n', stdout);
        fflush(stdout);
        free(l);
    }
}
extern "C" void rovaca_amd_bench_hold_line(const char *line)
{
    static const int registered = atexit(emit_held_line);
    (void)registered;
    free(g_held_line.exchange(line ? strdup(line) : nullptr));
}

extern "C" int rovaca_amd_adapter_bench(const adapter_bench_args *a, adapter_bench_result *res)
{
    if (!a || !res || a->n_regions == 0 || a->threads < 1 || a->regions_per_thread < 1) return 1;
    if (!gpu_pairhmm_active() && !init_pairhmm_ptr_gpu(0)) return 2;  // no MI355X visible: nothing is measured (no CPU path)
    const bool with_sw = a->with_sw != 0;
    const int depth = std::max(1, a->depth), threads = a->threads, per_thread = a->regions_per_thread;
    const PcrIndelModel pcr = static_cast<PcrIndelModel>(a->pcr_model);
    std::vector<Region> regions(a->n_regions);
    for (uint32_t g = 0; g < a->n_regions; ++g) {
        Region &rg = regions[g];
        uint64_t hsum = 0;
        for (uint32_t h = a->region_hap[g]; h < a->region_hap[g + 1]; ++h) {
            rg.haps.emplace_back(a->hap_bases + a->hap_off[h], a->hap_len[h]);
            hsum += a->hap_len[h];
        }
        for (uint32_t r = a->region_read[g]; r < a->region_read[g + 1]; ++r) {
            rg.reads.emplace_back(a->read_bases + a->read_off[r], a->read_qual + a->read_off[r], a->read_len[r], a->read_mapq[r]);
            rg.cells += (uint64_t)a->read_len[r] * hsum;
        }
    }
    std::atomic<uint64_t> cells{0}, kept{0}, sw_cells{0}, sw_elems{0};
    // what follows the seam in the caller (haplotypecaller_engine.cpp:250-285), as far as this benchmark goes
    auto after_pairhmm = [&](Region &rg, HaplotypeVector &hs, ReadVector &rs, DoubleVector2D &lik) {
            cells += rg.cells;
            kept += rs.size();
            if (with_sw && !rs.empty()) {
                std::vector<std::vector<uint8_t>> seqs(rs.size());
                std::vector<SwJob> jobs(rs.size());
                for (size_t r = 0; r < rs.size(); ++r) {
                    size_t best = 0;
                    for (size_t h = 1; h < hs.size(); ++h)
                        if (lik[h][r] > lik[best][r]) best = h;
                    seqs[r].resize(sizeof(Bases) + (size_t)rs[r]->seq_length() + 1);
                    pBases b = reinterpret_cast<pBases>(seqs[r].data());
                    rs[r]->decode_base(b);
                    const pBases hap = hs[best]->get_display_string();
                    jobs[r] = SwJob{hap->data, (int32_t)hap->num, b->data, (int32_t)b->num};
                    sw_cells += (uint64_t)hap->num * b->num;
                }
                std::vector<SwAlignment> aln;
                const sw_parameters genotype{10, -15, -30, -5};  // sw_avx.c:14
                if (!sw_hip_align_batch(jobs.data(), jobs.size(), genotype, aln)) exit(3);
                sw_elems += aln.empty() ? 0 : aln[0].cigar.size();
            }
    };
    struct InFlight
    {
        std::pmr::monotonic_buffer_resource arena;
        ReadVector rs{&arena};
        HaplotypeVector hs{&arena};
        Region *rg = nullptr;
        PairhmmTicket ticket;
        // realignment stage (with_sw): decoded reads the jobs point at, and the ticket of the device call
        std::vector<std::vector<uint8_t>> seqs;
        SwTicket sw_ticket;
    };
    auto work = [&](int t) {
        if (depth <= 1) {  // the seams as the reference calls them: one region at a time, synchronously
            for (int k = 0; k < per_thread; ++k) {
                Region &rg = regions[(t + k) % regions.size()];
                std::pmr::monotonic_buffer_resource arena;
                ReadVector rs{&arena};
                HaplotypeVector hs{&arena};
                for (auto &r : rg.reads) rs.push_back(&r);
                for (auto &h : rg.haps) hs.push_back(&h);
                DoubleVector2D lik = call_pairhmm(hs, rs, a->min_q, pcr, &arena);
                after_pairhmm(rg, hs, rs, lik);
            }
            return;
        }
        // begin/finish forms: up to `depth` regions wait for their PairHMM and, with sw, up to `depth` for their realignment
        std::deque<std::unique_ptr<InFlight>> ph_q, sw_q;
        const sw_parameters genotype{10, -15, -30, -5};  // sw_avx.c:14
        auto finish_sw = [&](InFlight &f) {
            std::vector<SwAlignment> aln;
            if (!sw_hip_align_batch_finish(std::move(f.sw_ticket), aln)) exit(3);
            sw_elems += aln.empty() ? 0 : aln[0].cigar.size();
        };
        auto finish_pairhmm = [&](std::unique_ptr<InFlight> slot) {
            InFlight &f = *slot;
            bool realign = false;
            {
                DoubleVector2D lik = call_gpu_pairhmm_finish(std::move(f.ticket), f.rs, &f.arena);
                cells += f.rg->cells;
                kept += f.rs.size();
                if (with_sw && !f.rs.empty()) {
                    f.seqs.resize(f.rs.size());
                    std::vector<SwJob> jobs(f.rs.size());
                    for (size_t r = 0; r < f.rs.size(); ++r) {
                        size_t best = 0;
                        for (size_t h = 1; h < f.hs.size(); ++h)
                            if (lik[h][r] > lik[best][r]) best = h;
                        f.seqs[r].resize(sizeof(Bases) + (size_t)f.rs[r]->seq_length() + 1);
                        pBases b = reinterpret_cast<pBases>(f.seqs[r].data());
                        f.rs[r]->decode_base(b);
                        const pBases hap = f.hs[best]->get_display_string();
                        jobs[r] = SwJob{hap->data, (int32_t)hap->num, b->data, (int32_t)b->num};
                        sw_cells += (uint64_t)hap->num * b->num;
                    }
                    f.sw_ticket = sw_hip_align_batch_begin(jobs.data(), jobs.size(), genotype);
                    realign = true;
                }
            }  // the matrix lives in the slot's arena: gone before the arena is
            if (realign) {
                sw_q.push_back(std::move(slot));
                if ((int)sw_q.size() >= depth) {
                    finish_sw(*sw_q.front());
                    sw_q.pop_front();
                }
            }
        };
        for (int k = 0; k < per_thread; ++k) {
            std::unique_ptr<InFlight> slot(new InFlight());
            slot->rg = &regions[(t + k) % regions.size()];
            for (auto &r : slot->rg->reads) slot->rs.push_back(&r);
            for (auto &h : slot->rg->haps) slot->hs.push_back(&h);
            slot->ticket = call_gpu_pairhmm_begin(slot->hs, slot->rs, a->min_q, pcr);
            ph_q.push_back(std::move(slot));
            if ((int)ph_q.size() >= depth) {
                std::unique_ptr<InFlight> done = std::move(ph_q.front());
                ph_q.pop_front();
                finish_pairhmm(std::move(done));
            }
        }
        while (!ph_q.empty()) {
            std::unique_ptr<InFlight> done = std::move(ph_q.front());
            ph_q.pop_front();
            finish_pairhmm(std::move(done));
        }
        while (!sw_q.empty()) {
            finish_sw(*sw_q.front());
            sw_q.pop_front();
        }
    };
    work(0);  // warm up: contexts, stages, kernels
    // the check: the first regions through the synchronous seam, matrices handed back
    if (a->check_regions && a->check_out && a->check_kept) {
        size_t at = 0;
        for (uint32_t g = 0; g < a->check_regions && g < a->n_regions; ++g) {
            Region &rg = regions[g];
            std::pmr::monotonic_buffer_resource arena;
            ReadVector rs{&arena};
            HaplotypeVector hs{&arena};
            for (auto &r : rg.reads) rs.push_back(&r);
            for (auto &h : rg.haps) hs.push_back(&h);
            const size_t nr = rs.size(), nh = hs.size();
            DoubleVector2D lik = call_pairhmm(hs, rs, a->min_q, pcr, &arena);
            for (size_t k = 0; k < nh * nr; ++k) a->check_out[at + k] = NAN;
            for (size_t h = 0; h < lik.size(); ++h)
                for (size_t k = 0; k < lik[h].size(); ++k) a->check_out[at + h * nr + k] = lik[h][k];
            a->check_kept[g] = (uint32_t)rs.size();
            at += nh * nr;
        }
    }
    cells = 0;
    kept = 0;
    sw_cells = 0;
    uint64_t nreg0 = 0, nsub0 = 0;
    gpu_pairhmm_batching_stats(&nreg0, &nsub0);
    double ph0[16] = {0}, ph1[16] = {0};
    const int n_ph = gpu_pairhmm_profile(ph0, nullptr, 16);
    const double cpu0 = process_cpu_seconds();
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double cpu1 = process_cpu_seconds();
    gpu_pairhmm_profile(ph1, nullptr, 16);
    uint64_t nreg = 0, nsub = 0;
    gpu_pairhmm_batching_stats(&nreg, &nsub);
    *res = adapter_bench_result{};
    res->seconds = sec;
    res->regions = (uint64_t)threads * per_thread;
    res->regions_per_s = (double)res->regions / sec;
    res->cells = cells.load();
    res->gcups = (double)cells.load() / sec / 1e9;
    res->sw_gcups = (double)sw_cells.load() / sec / 1e9;
    res->kept_reads = kept.load();
    res->submissions = nsub - nsub0;
    res->batched_regions = nreg - nreg0;
    res->cpu_seconds = cpu1 - cpu0;
    res->n_phases = n_ph;
    for (int i = 0; i < n_ph && i < 16; ++i) res->phase_seconds[i] = ph1[i] - ph0[i];
    return 0;
}

#ifdef ADAPTER_BENCH_MAIN
int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 30, per_thread = argc > 2 ? atoi(argv[2]) : 64;
    const int n_reads = argc > 3 ? atoi(argv[3]) : 48, n_haps = argc > 4 ? atoi(argv[4]) : 9;
    const bool with_sw = argc > 5 && atoi(argv[5]) != 0;
    const int depth = argc > 6 ? std::max(1, atoi(argv[6])) : 1;  // regions a worker keeps in flight (begin/finish form of the seam)
    // 16 synthetic regions of 150-base reads against 300-base haplotypes, flattened into the tables the library call takes
    std::vector<uint8_t> rb, rq, mq, hb;
    std::vector<uint32_t> roff, rlen, hoff, hlen, region_read{0}, region_hap{0};
    const char acgt[] = "ACGT";
    // ADAPTER_BENCH_MIXED=1: 64 templates of mixed shape instead (8-135 reads of 151 bases, a fifth of them trimmed to 50-150, against
    // 2-21 haplotypes of 160-500 bases) -- what the stages and the length bins make of regions that are not all alike
    const bool mixed = getenv("ADAPTER_BENCH_MIXED") != nullptr;
    const int n_templates = mixed ? 64 : 16;
    for (int i = 0; i < n_templates; ++i) {
        std::mt19937 rng(1000 + i);
        const int H = mixed ? 160 + (int)(rng() % 341) : 300;
        const int n_reads_i = mixed ? 8 + (int)(rng() % 128) : n_reads, n_haps_i = mixed ? 2 + (int)(rng() % 20) : n_haps;
        std::vector<uint8_t> base(H);
        for (auto &c : base) c = acgt[rng() & 3];
        for (int h = 0; h < n_haps_i; ++h) {
            std::vector<uint8_t> x = base;
            if (h) for (int k = 0; k < 3; ++k) x[rng() % H] = acgt[rng() & 3];
            hoff.push_back((uint32_t)hb.size());
            hlen.push_back(H);
            hb.insert(hb.end(), x.begin(), x.end());
        }
        for (int r = 0; r < n_reads_i; ++r) {
            const int R = !mixed ? 150 : rng() % 5 ? 151 : 50 + (int)(rng() % 101);
            const int st = rng() % (H - R + 1);
            roff.push_back((uint32_t)rb.size());
            rlen.push_back(R);
            for (int k = 0; k < R; ++k) {
                rb.push_back(rng() % 100 == 0 ? acgt[rng() & 3] : base[st + k]);
                const int u = rng() % 100;
                rq.push_back(u < 1 ? 2 : u < 5 ? 11 : u < 20 ? 25 : 37);
            }
            mq.push_back(60);
        }
        region_read.push_back((uint32_t)rlen.size());
        region_hap.push_back((uint32_t)hlen.size());
    }
    adapter_bench_args a{};
    a.read_bases = rb.data();
    a.read_qual = rq.data();
    a.read_off = roff.data();
    a.read_len = rlen.data();
    a.read_mapq = mq.data();
    a.hap_bases = hb.data();
    a.hap_off = hoff.data();
    a.hap_len = hlen.data();
    a.region_read = region_read.data();
    a.region_hap = region_hap.data();
    a.n_regions = n_templates;
    a.threads = threads;
    a.regions_per_thread = per_thread;
    a.depth = depth;
    a.with_sw = with_sw ? 1 : 0;
    a.min_q = 18;
    a.pcr_model = 3;  // CONSERVATIVE, the reference's default (rovaca_tool_args.h:199)
    adapter_bench_result r{};
    const int rc = rovaca_amd_adapter_bench(&a, &r);
    if (rc != 0) {
        fprintf(stderr, rc == 2 ? "no MI355X visible\n" : "bad arguments\n");
        return rc;
    }
    printf("{\"threads\": %d, \"regions\": %llu, \"reads_per_region\": %d, \"haps_per_region\": %d, \"seconds\": %.4f, "
           "\"regions_per_s\": %.0f, \"gcups_end_to_end\": %.1f, \"device_submissions\": %llu, \"regions_per_submission\": %.1f, "
           "\"smith_waterman\": %s, \"sw_gcups_end_to_end\": %.1f, \"regions_in_flight_per_worker\": %d, \"process_cpu_seconds\": %.3f, "
           "\"cpu_us_per_region\": %.1f}\n",
           threads, (unsigned long long)r.regions, n_reads, n_haps, r.seconds, r.regions_per_s, r.gcups, (unsigned long long)r.submissions,
           r.submissions ? (double)r.batched_regions / (double)r.submissions : 0.0, with_sw ? "true" : "false", r.sw_gcups, depth, r.cpu_seconds,
           r.cpu_seconds / (double)r.regions * 1e6);
    if (getenv("ADAPTER_BENCH_PHASES")) {
        const char *names[16] = {nullptr};
        double dummy[16];
        gpu_pairhmm_profile(dummy, names, 16);
        for (int i = 0; i < r.n_phases; ++i)
            fprintf(stderr, "  %-58s %9.3f s  %8.2f us/region\n", names[i], r.phase_seconds[i], r.phase_seconds[i] / (double)r.regions * 1e6);
        if (with_sw) {  // (since process start: the timed run and nothing else in this program)
            const char *sw_names[16] = {nullptr};
            double sw_values[16];
            const int n_sw = sw_hip_profile(sw_values, sw_names, 16);
            for (int i = 0; i < n_sw; ++i) {
                if (i >= n_sw - 2)
                    fprintf(stderr, "  %-58s %9.0f    %8.2f per region\n", sw_names[i], sw_values[i], sw_values[i] / (double)r.regions);
                else
                    fprintf(stderr, "  %-58s %9.3f s  %8.2f us/region\n", sw_names[i], sw_values[i], sw_values[i] / (double)r.regions * 1e6);
            }
        }
    }
    return 0;
}
#endif
