// pairhmm_gpu_engine.cpp -- rovaca::call_gpu_pairhmm: flattens one region into a POD batch, runs it through the C ABI of
// include/pairhmm_hip.h, and applies the reference's post-processing.  The only translation unit that knows Rovaca types.
#include "pairhmm_gpu_engine.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/pairhmm_hip.h"
#include "pairhmm_host_logic.h"

namespace rovaca
{

#ifndef ROVACA_AMD_IN_ROVACA_TREE
// stand-alone build: this library owns the seam's global.  Inside Rovaca it is defined at pairhmm_engine.cpp:16-17.
DoubleVector2D (*call_pairhmm)(const HaplotypeVector &hs, ReadVector &rs, int32_t min_quality_threshold,
                               PcrIndelModel pcr_option, pMemoryPool pool) = nullptr;
#endif

namespace
{

std::atomic<bool> g_active{false};
std::atomic<int> g_device{-1};
std::atomic<int> g_next_device{0};

// one device context per calling thread, created on first use and destroyed with the thread
struct ThreadContext
{
    phmm_context *ctx = nullptr;
    ~ThreadContext()
    {
        if (ctx) phmm_destroy(ctx);
    }
    phmm_context *get()
    {
        if (ctx) return ctx;
        int dev = g_device.load();
        if (dev < 0) {
            const int n = phmm_device_count();
            dev = n > 0 ? g_next_device.fetch_add(1) % n : 0;
        }
        if (phmm_create(dev, nullptr, &ctx) != PHMM_OK) {
            // the reference's convention for an unrecoverable engine failure: log, exit (pairhmm_engine.cpp:32-35)
            fprintf(stderr, "[rovaca_amd] pairhmm: cannot create a device context: %s\n", phmm_last_error());
            exit(EXIT_FAILURE);
        }
        return ctx;
    }
};
thread_local ThreadContext t_ctx;

}  // namespace

bool init_pairhmm_ptr_gpu(int device_id)
{
    const int n = phmm_device_count();
    if (n <= 0 || device_id >= n) return false;
    g_device.store(device_id);
    call_pairhmm = call_gpu_pairhmm;
    g_active.store(true);
    return true;
}

bool gpu_pairhmm_active() { return g_active.load(); }

DoubleVector2D call_gpu_pairhmm(const HaplotypeVector &hs, ReadVector &rs, int32_t min_quality_threshold,
                                PcrIndelModel pcr_option, pMemoryPool pool)
{
    if (hs.empty() || rs.empty()) {
        return {};
    }
    const size_t reads_num = rs.size();
    const size_t haplotypes_num = hs.size();
    // the result comes from the caller's arena, before anything else (pairhmm_internal.cpp:516)
    DoubleVector2D final_likelihood{haplotypes_num, DoubleVector{reads_num, NAN, pool}, pool};

    // ---- flatten: read table ----
    std::vector<uint32_t> read_off(reads_num), read_len(reads_num);
    std::vector<int32_t> seq_len(reads_num);
    size_t total = 0;
    int32_t max_len = 0;
    for (size_t i = 0; i < reads_num; ++i) {
        seq_len[i] = rs[i]->seq_length();
        read_off[i] = (uint32_t)total;
        read_len[i] = (uint32_t)seq_len[i];
        total += (size_t)seq_len[i];
        max_len = std::max(max_len, seq_len[i]);
    }
    std::vector<uint8_t> bases(total), quals(total), gops(total), gcps(total, rovaca_amd::kDefaultGapCont);
    std::vector<uint8_t> tmp(sizeof(Bases) + (size_t)max_len + 1);
    pBases read_bases = reinterpret_cast<pBases>(tmp.data());
    const int model = static_cast<int>(pcr_option);
    for (size_t i = 0; i < reads_num; ++i) {
        pReadRecord read = rs[i];
        const uint32_t n = read_len[i], o = read_off[i];
        read->decode_base(read_bases);
        memcpy(bases.data() + o, read_bases->data, n);
        memcpy(quals.data() + o, read->qual(), n);
        rovaca_amd::modify_read_qualities(quals.data() + o, n, read->mapping_quality(), min_quality_threshold);  // :545-550
        memset(gops.data() + o, rovaca_amd::kDefaultGapOpen, n);                                                 // :552-554
        if (pcr_option != PcrIndelModel::NONE) rovaca_amd::apply_pcr_error_model(bases.data() + o, gops.data() + o, n, model);
    }
    // ---- flatten: haplotype table ----
    std::vector<uint32_t> hap_off(haplotypes_num), hap_len(haplotypes_num);
    size_t hap_total = 0;
    for (size_t j = 0; j < haplotypes_num; ++j) {
        hap_off[j] = (uint32_t)hap_total;
        hap_len[j] = hs[j]->get_display_string()->num;
        hap_total += hap_len[j];
    }
    std::vector<uint8_t> hap_bases(hap_total);
    for (size_t j = 0; j < haplotypes_num; ++j) memcpy(hap_bases.data() + hap_off[j], hs[j]->get_display_string()->data, hap_len[j]);

    // ---- the device: all reads x all haplotypes, (read, hap) order ----
    phmm_batch b{};
    b.n_reads = (uint32_t)reads_num;
    b.n_haps = (uint32_t)haplotypes_num;
    b.n_pairs = (uint64_t)reads_num * haplotypes_num;
    b.read_bases = bases.data();
    b.read_qual = quals.data();
    b.read_ins = gops.data();
    b.read_del = gops.data();  // del gap-open aliases the ins buffer (pairhmm_internal.cpp:559)
    b.read_gcp = gcps.data();
    b.read_off = read_off.data();
    b.read_len = read_len.data();
    b.hap_bases = hap_bases.data();
    b.hap_off = hap_off.data();
    b.hap_len = hap_len.data();
    std::vector<double> likelihood_array(reads_num * haplotypes_num);  // [read][hap]
    if (phmm_compute(t_ctx.get(), &b, nullptr, likelihood_array.data(), nullptr) != PHMM_OK) {
        fprintf(stderr, "[rovaca_amd] pairhmm: device evaluation failed: %s\n", phmm_last_error());
        exit(EXIT_FAILURE);
    }

    // ---- post-processing (pairhmm_internal.cpp:589-591) ----
    rovaca_amd::normalize_likelihoods(likelihood_array.data(), reads_num, haplotypes_num);
    const std::vector<size_t> drop = rovaca_amd::poorly_modelled_reads(likelihood_array.data(), seq_len.data(), reads_num, haplotypes_num);
    std::vector<char> dropped(reads_num, 0);
    for (size_t r : drop) dropped[r] = 1;
    for (auto it = drop.rbegin(); it != drop.rend(); ++it) rs.erase(rs.begin() + (std::ptrdiff_t)*it);  // back to front, :178-181
    size_t k = 0;
    for (size_t i = 0; i < reads_num; ++i) {
        if (dropped[i]) continue;
        for (size_t j = 0; j < haplotypes_num; ++j) final_likelihood[j][k] = likelihood_array[i * haplotypes_num + j];
        ++k;
    }
    return final_likelihood;
}

}  // namespace rovaca
