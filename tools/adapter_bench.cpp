// adapter_bench.cpp -- end-to-end throughput of the drop-in adapter: N worker threads call rovaca::call_pairhmm on
// synthetic regions (as haplotypecaller_engine.cpp:119-285 does from its asio pool), through the cross-region batcher.
// Build: make -C rovaca_amd/host bench   Run on the GPU box: rovaca_amd/lib/adapter_bench [threads] [regions/thread] [reads] [haps] [sw] [regions in flight per worker]
// sw = 1: after PairHMM every surviving read is realigned to its best haplotype through rovaca::sw_hip_align_batch
// (haplotypecaller_engine.cpp:264), one call per region.
#include <algorithm>
#include <atomic>
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

static Region make_region(uint32_t seed, int n_reads, int n_haps, int R, int H)
{
    std::mt19937 rng(seed);
    const char acgt[] = "ACGT";
    std::vector<uint8_t> base(H);
    for (auto &c : base) c = acgt[rng() & 3];
    Region rg;
    for (int h = 0; h < n_haps; ++h) {
        std::vector<uint8_t> hb = base;
        if (h) for (int k = 0; k < 3; ++k) hb[rng() % H] = acgt[rng() & 3];
        rg.haps.emplace_back(hb.data(), (uint32_t)H);
    }
    for (int r = 0; r < n_reads; ++r) {
        const int st = rng() % (H - R + 1);
        std::vector<uint8_t> seq(base.begin() + st, base.begin() + st + R), q(R);
        for (int i = 0; i < R; ++i) {
            if (rng() % 100 == 0) seq[i] = acgt[rng() & 3];
            const int u = rng() % 100;
            q[i] = u < 1 ? 2 : u < 5 ? 11 : u < 20 ? 25 : 37;
        }
        rg.reads.emplace_back(seq.data(), q.data(), (uint32_t)R, (uint8_t)60);
        rg.cells += (uint64_t)R * H * n_haps;
    }
    return rg;
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 30, per_thread = argc > 2 ? atoi(argv[2]) : 64;
    const int n_reads = argc > 3 ? atoi(argv[3]) : 48, n_haps = argc > 4 ? atoi(argv[4]) : 9;
    const bool with_sw = argc > 5 && atoi(argv[5]) != 0;
    const int depth = argc > 6 ? std::max(1, atoi(argv[6])) : 1;  // regions a worker keeps in flight (begin/finish form of the seam)
    if (!init_pairhmm_ptr_gpu(0)) {
        fprintf(stderr, "no MI355X visible\n");
        return 2;
    }
    std::vector<Region> regions;
    for (int i = 0; i < 16; ++i) regions.push_back(make_region(1000 + i, n_reads, n_haps, 150, 300));
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
                DoubleVector2D lik = call_pairhmm(hs, rs, 18, PcrIndelModel::CONSERVATIVE, &arena);
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
            slot->ticket = call_gpu_pairhmm_begin(slot->hs, slot->rs, 18, PcrIndelModel::CONSERVATIVE);
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
    work(0);  // warm up: contexts, tables, kernels
    cells = 0;
    sw_cells = 0;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t nreg = 0, nsub = 0;
    gpu_pairhmm_batching_stats(&nreg, &nsub);
    printf("{\"threads\": %d, \"regions\": %d, \"reads_per_region\": %d, \"haps_per_region\": %d, \"seconds\": %.4f, "
           "\"regions_per_s\": %.0f, \"gcups_end_to_end\": %.1f, \"device_submissions\": %llu, \"regions_per_submission\": %.1f, "
           "\"smith_waterman\": %s, \"sw_gcups_end_to_end\": %.1f, \"regions_in_flight_per_worker\": %d}\n",
           threads, threads * per_thread, n_reads, n_haps, sec, threads * per_thread / sec, cells.load() / sec / 1e9,
           (unsigned long long)nsub, nsub ? (double)nreg / nsub : 0.0, with_sw ? "true" : "false", sw_cells.load() / sec / 1e9, depth);
    return 0;
}
