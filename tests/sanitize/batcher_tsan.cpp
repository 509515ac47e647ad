// batcher_tsan.cpp -- ThreadSanitizer run of the two cross-region batchers (pairhmm_gpu_engine.cpp, sw_gpu_engine.cpp) WITHOUT a
// GPU: the C ABIs they call are replaced, in this test binary only, by deterministic stand-ins (a value that depends on the
// read and haplotype bytes alone), so that every worker can check that it got its own region's numbers however the dispatchers
// merged the submissions, while TSan watches the hand-offs.  Test infrastructure: the product libraries are not involved and
// still have no CPU path.  Built and run by tests/test_sanitizers.py.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <memory_resource>
#include <random>
#include <thread>
#include <vector>

#include "../../include/pairhmm_hip.h"
#include "../../include/smithwaterman_hip.h"
#include "../../rovaca_amd/host/pairhmm_gpu_engine.h"
#include "../../rovaca_amd/host/sw_gpu_engine.h"

// ---------------- stand-in device libraries ----------------
struct phmm_context
{
    int dummy;
};
struct swhip_context
{
    int dummy;
};
static uint32_t hash_bytes(const uint8_t *p, uint32_t n)
{
    uint32_t h = 2166136261u;
    for (uint32_t i = 0; i < n; ++i) h = (h ^ p[i]) * 16777619u;
    return h;
}
static double fake_likelihood(const uint8_t *read, uint32_t R, const uint8_t *hap, uint32_t H)
{
    return -1.0 - (double)(hash_bytes(read, R) % 997u) / 100.0 - (double)(hash_bytes(hap, H) % 101u) / 50.0;
}
extern "C" {
int phmm_abi_version(void) { return PHMM_ABI_VERSION; }
int phmm_device_count(void) { return 1; }
const char *phmm_last_error(void) { return ""; }
int phmm_create(int, void *, phmm_context **out)
{
    *out = new phmm_context{0};
    return PHMM_OK;
}
void phmm_destroy(phmm_context *c) { delete c; }
int phmm_compute(phmm_context *, const phmm_batch *b, float *, double *out_log10, phmm_stats *)
{
    // the direct path (a region that fits no stage): the full cross product in (read, hap) order
    std::this_thread::sleep_for(std::chrono::microseconds(150));
    for (uint32_t r = 0; r < b->n_reads; ++r)
        for (uint32_t h = 0; h < b->n_haps; ++h)
            out_log10[(size_t)r * b->n_haps + h] =
                fake_likelihood(b->read_bases + b->read_off[r], b->read_len[r], b->hap_bases + b->hap_off[h], b->hap_len[h]);
    return PHMM_OK;
}
// stand-in stage: ordinary memory laid out like the real one, "results" computed at submit
struct phmm_stage
{
    phmm_stage_caps caps;
    std::vector<uint8_t> bytes, mq;
    std::vector<uint32_t> roff, qoff, rlen, hoff, hlen;
    std::vector<phmm_region> reg;
    std::vector<double> out;
    std::vector<uint64_t> first;
};
int phmm_stage_create(phmm_context *, const phmm_stage_caps *k, phmm_stage **out)
{
    phmm_stage *s = new phmm_stage();
    s->caps = *k;
    s->bytes.resize(k->max_bytes);
    s->mq.resize(k->max_reads);
    s->roff.resize(k->max_reads);
    s->qoff.resize(k->max_reads);
    s->rlen.resize(k->max_reads);
    s->hoff.resize(k->max_haps);
    s->hlen.resize(k->max_haps);
    s->reg.resize(k->max_regions);
    s->out.resize(k->max_pairs);
    s->first.resize((size_t)k->max_regions + 1);
    *out = s;
    return PHMM_OK;
}
void phmm_stage_destroy(phmm_context *, phmm_stage *s) { delete s; }
int phmm_stage_get_view(const phmm_stage *cs, phmm_stage_view *v, phmm_stage_caps *k)
{
    phmm_stage *s = const_cast<phmm_stage *>(cs);
    *v = phmm_stage_view{s->bytes.data(), s->roff.data(), s->qoff.data(), s->rlen.data(), s->mq.data(), s->hoff.data(), s->hlen.data(), s->reg.data()};
    if (k) *k = s->caps;
    return PHMM_OK;
}
int phmm_stage_submit(phmm_context *, phmm_stage *s, uint32_t n_regions, uint32_t n_reads, uint32_t n_haps, const phmm_stage_params *)
{
    uint64_t p = 0;
    for (uint32_t g = 0; g < n_regions; ++g) {
        const phmm_region &rg = s->reg[g];
        if (rg.read_first + rg.n_reads > n_reads || rg.hap_first + rg.n_haps > n_haps) return PHMM_E_INVALID;
        s->first[g] = p;
        for (uint32_t r = rg.read_first; r < rg.read_first + rg.n_reads; ++r) {
            if (s->qoff[r] != s->roff[r] + s->rlen[r]) return PHMM_E_INVALID;  // the adapter puts a read's qualities behind its bases
            for (uint32_t h = rg.hap_first; h < rg.hap_first + rg.n_haps; ++h)
                s->out[p++] = fake_likelihood(s->bytes.data() + s->roff[r], s->rlen[r], s->bytes.data() + s->hoff[h], s->hlen[h]);
        }
    }
    s->first[n_regions] = p;
    return PHMM_OK;
}
int phmm_stage_wait(phmm_context *, phmm_stage *, phmm_stats *)
{
    std::this_thread::sleep_for(std::chrono::microseconds(150));  // a device round trip: lets regions pile up in the open stage
    return PHMM_OK;
}
uint64_t phmm_stage_pair_first(const phmm_stage *s, uint32_t region) { return s->first[region]; }
int phmm_stage_log10(const phmm_stage *s, uint64_t first_pair, uint64_t n, double *out, float *, uint64_t *)
{
    for (uint64_t p = 0; p < n; ++p) out[p] = s->out[first_pair + p];
    return PHMM_OK;
}
int swhip_abi_version(void) { return SWHIP_ABI_VERSION; }
int swhip_device_count(void) { return 1; }
const char *swhip_last_error(void) { return ""; }
int swhip_create(int, void *, swhip_context **out)
{
    *out = new swhip_context{0};
    return SWHIP_OK;
}
void swhip_destroy(swhip_context *c) { delete c; }
int swhip_align(swhip_context *, const swhip_params *, const swhip_batch *b, swhip_alignment *out, uint32_t *cigars, uint64_t cap,
                uint64_t *used, swhip_stats *)
{
    std::this_thread::sleep_for(std::chrono::microseconds(150));
    uint64_t at = 0;
    for (uint64_t p = 0; p < b->n_pairs; ++p) {
        const uint32_t r = b->pair_ref[p], s = b->pair_seq[p];
        out[p] = swhip_alignment{};
        out[p].score = (int32_t)(hash_bytes(b->seq_bases + b->seq_off[s], b->seq_len[s]) % 100000u);
        out[p].alignment_offset = (int16_t)(hash_bytes(b->ref_bases + b->ref_off[r], b->ref_len[r]) % 300u);
        out[p].cigar_off = (uint32_t)at;
        out[p].cigar_count = 1;
        if (at < cap) cigars[at] = (b->seq_len[s] << 4) | 0u;
        ++at;
    }
    *used = at;
    return at > cap ? SWHIP_E_CAPACITY : SWHIP_OK;
}
}

using namespace rovaca;

int main()
{
    if (!init_pairhmm_ptr_gpu(0)) return 2;
    // BATCHER_THREADS: worker threads (default 12; the test also runs 48 on whatever cores there are: a worker preempted in the middle
    // of a reservation must hold nobody up and corrupt nothing)
    const char *env_threads = getenv("BATCHER_THREADS");
    const int threads = env_threads ? std::max(1, atoi(env_threads)) : 12, per_thread = 40;
    std::atomic<int> bad{0};
    struct RegionData  // one synthetic region and everything that must outlive its evaluation
    {
        int n_reads = 0, n_haps = 0;
        std::vector<ReadRecord> reads;
        std::vector<Haplotype> haps;
        std::vector<std::vector<uint8_t>> rb, hb;
        std::pmr::monotonic_buffer_resource arena;
        ReadVector rs{&arena};
        HaplotypeVector hs{&arena};
        PairhmmTicket ticket;
        SwTicket sw_ticket;  // destroyed before the sequences it points at
        explicit RegionData(std::mt19937 &rng)
        {
            const char acgt[] = "ACGT";
            n_reads = 1 + rng() % 12;
            n_haps = 1 + rng() % 5;
            rb.resize(n_reads);
            hb.resize(n_haps);
            reads.reserve(n_reads);
            haps.reserve(n_haps);
            for (auto &h : hb) {
                h.resize(60 + rng() % 100);
                for (auto &c : h) c = acgt[rng() & 3];
                haps.emplace_back(h.data(), (uint32_t)h.size());
            }
            for (auto &r : rb) {
                r.resize(20 + rng() % 60);
                for (auto &c : r) c = acgt[rng() & 3];
                std::vector<uint8_t> q(r.size(), 30);
                reads.emplace_back(r.data(), q.data(), (uint32_t)r.size(), (uint8_t)60);
            }
            for (auto &r : reads) rs.push_back(&r);
            for (auto &h : haps) hs.push_back(&h);
        }
        // every surviving read's row must be this region's own numbers (normalisation only caps from below)
        int mismatches(const DoubleVector2D &lik) const
        {
            int n = 0;
            for (size_t r = 0; r < rs.size(); ++r) {
                const size_t orig = (size_t)(rs[r] - reads.data());
                double best = -1e300;
                for (int h = 0; h < n_haps; ++h) best = std::max(best, fake_likelihood(rb[orig].data(), (uint32_t)rb[orig].size(), hb[h].data(), (uint32_t)hb[h].size()));
                for (int h = 0; h < n_haps; ++h) {
                    const double want = std::max(fake_likelihood(rb[orig].data(), (uint32_t)rb[orig].size(), hb[h].data(), (uint32_t)hb[h].size()), best - 4.5);
                    if (lik[h][r] != want) ++n;
                }
            }
            return n;
        }
    };
    auto work = [&](int t) {
        std::mt19937 rng(100 + t);
        std::unique_ptr<RegionData> in_flight;  // odd workers: the begin / finish form, the previous region still on the "device"
        auto check_sw = [&](const RegionData &rg, const std::vector<SwAlignment> &aln) {
            if ((int)aln.size() != rg.n_reads) {
                ++bad;
                return;
            }
            for (int r = 0; r < rg.n_reads; ++r) {
                if (aln[r].score != (int32_t)(hash_bytes(rg.rb[r].data(), (uint32_t)rg.rb[r].size()) % 100000u)) ++bad;
                if (aln[r].cigar.size() != 1 || aln[r].cigar[0] != (((uint32_t)rg.rb[r].size() << 4) | 0u)) ++bad;
            }
        };
        auto collect = [&](RegionData &rg) {
            {
                DoubleVector2D lik = call_gpu_pairhmm_finish(std::move(rg.ticket), rg.rs, &rg.arena);
                bad += rg.mismatches(lik);
            }
            std::vector<SwAlignment> aln;
            if (!sw_hip_align_batch_finish(std::move(rg.sw_ticket), aln)) ++bad;
            check_sw(rg, aln);
        };
        for (int k = 0; k < per_thread; ++k) {
            std::unique_ptr<RegionData> cur(new RegionData(rng));
            RegionData &rg = *cur;
            const int n_reads = rg.n_reads, n_haps = rg.n_haps;
            auto &rb = rg.rb;
            auto &hb = rg.hb;
            // Smith-Waterman batcher: one request per region
            std::vector<SwJob> jobs;
            for (int r = 0; r < n_reads; ++r) jobs.push_back(SwJob{hb[r % n_haps].data(), (int32_t)hb[r % n_haps].size(), rb[r].data(), (int32_t)rb[r].size()});
            const sw_parameters para{10, -15, -30, -5};
            if (t & 1) {
                // both seams in two halves: this region's evaluation and realignment are queued, the previous region is collected
                rg.ticket = call_gpu_pairhmm_begin(rg.hs, rg.rs, 18, PcrIndelModel::NONE);
                rg.sw_ticket = sw_hip_align_batch_begin(jobs.data(), jobs.size(), para);
                if (in_flight) collect(*in_flight);
                in_flight = std::move(cur);
            }
            else {
                // (every 7th call with another threshold: the open stage rolls over on a parameter mismatch -- the stand-in ignores it)
                DoubleVector2D lik = call_pairhmm(rg.hs, rg.rs, (t + k) % 7 == 0 ? 10 : 18, PcrIndelModel::NONE, &rg.arena);
                bad += rg.mismatches(lik);
                std::vector<SwAlignment> aln;
                if (!sw_hip_align_batch(jobs.data(), jobs.size(), para, aln)) ++bad;
                check_sw(rg, aln);
            }
        }
        if (in_flight) collect(*in_flight);
        // a ticket dropped without finish must wait for its region and let go of it quietly
        if (t & 1) {
            RegionData rg(rng);
            rg.ticket = call_gpu_pairhmm_begin(rg.hs, rg.rs, 18, PcrIndelModel::NONE);
            const SwJob one{rg.hb[0].data(), (int32_t)rg.hb[0].size(), rg.rb[0].data(), (int32_t)rg.rb[0].size()};
            rg.sw_ticket = sw_hip_align_batch_begin(&one, 1, sw_parameters{10, -15, -30, -5});
        }
        // a ticket ASSIGNED OVER while its region is still in flight (a look-ahead loop that skipped finish): the old State
        // must not be freed under the dispatcher (move-assignment waits first, like the destructor)
        if (t & 1) {
            RegionData a(rng), b(rng);
            PairhmmTicket tk = call_gpu_pairhmm_begin(a.hs, a.rs, 18, PcrIndelModel::NONE);
            tk = call_gpu_pairhmm_begin(b.hs, b.rs, 18, PcrIndelModel::NONE);  // a's evaluation is still queued / on the device
            const SwJob ja{a.hb[0].data(), (int32_t)a.hb[0].size(), a.rb[0].data(), (int32_t)a.rb[0].size()};
            const SwJob jb{b.hb[0].data(), (int32_t)b.hb[0].size(), b.rb[0].data(), (int32_t)b.rb[0].size()};
            SwTicket st = sw_hip_align_batch_begin(&ja, 1, sw_parameters{10, -15, -30, -5});
            st = sw_hip_align_batch_begin(&jb, 1, sw_parameters{10, -15, -30, -5});
            DoubleVector2D lik = call_gpu_pairhmm_finish(std::move(tk), b.rs, &b.arena);
            bad += b.mismatches(lik);
            std::vector<SwAlignment> aln;
            if (!sw_hip_align_batch_finish(std::move(st), aln) || aln.size() != 1) ++bad;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    uint64_t regions = 0, subs = 0, rq = 0, sw_subs = 0;
    gpu_pairhmm_batching_stats(&regions, &subs);
    sw_hip_batching_stats(&rq, &sw_subs);
    printf("batchers: %llu regions in %llu submissions, %llu alignment requests in %llu submissions, %d mismatches\n",
           (unsigned long long)regions, (unsigned long long)subs, (unsigned long long)rq, (unsigned long long)sw_subs, bad.load());
    if (bad.load() || regions != (uint64_t)threads * per_thread + 3 * (threads / 2) || subs >= regions || sw_subs >= rq || rq != (uint64_t)threads * per_thread + 3 * (threads / 2)) return 1;
    fflush(stdout);
    _Exit(0);  // the dispatcher threads are never joined (by design); leave without static teardown
}
