// host_asan.cpp -- AddressSanitizer + UBSan run of the CPU-side code (the reference offers the same switch for its targets,
// cmake/asan.cmake): the product's host logic (H3, H4, H8) and the two oracles, on random inputs including the edge shapes
// (single base, repeats up to the caps, all-NaN rows).  Built and run by tests/test_sanitizers.py; exits 0 when clean.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../oracle/phmm_oracle.h"
#include "../../oracle/sw_oracle.h"
#include "../../rovaca_amd/csrc/pairhmm_cpu.h"
#include "../../rovaca_amd/host/pairhmm_host_logic.h"

int main()
{
    std::mt19937 rng(7);
    const char acgt[] = "ACGTN";
    uint64_t checksum = 0;
    for (int it = 0; it < 400; ++it) {
        const uint32_t R = 1 + rng() % 300, H = 1 + rng() % 400;
        std::vector<uint8_t> read(R), q(R), gop(R, rovaca_amd::kDefaultGapOpen), gcp(R, rovaca_amd::kDefaultGapCont), hap(H);
        const int unit = 1 + rng() % 9;
        for (uint32_t i = 0; i < R; ++i) read[i] = (it % 3 == 0) ? acgt[(i / unit) % 4] : acgt[rng() % 5];  // long tandem repeats too
        for (auto &c : hap) c = acgt[rng() % 5];
        for (auto &x : q) x = (uint8_t)(rng() % 64);
        // H3 + H4 (product) against the oracle's restatement, byte for byte
        std::vector<uint8_t> q2 = q, gop2 = gop;
        rovaca_amd::modify_read_qualities(q.data(), R, (uint8_t)(rng() % 70), 18);
        phmm_oracle_modify_quals(q2.data(), R, 60, 18);
        const int model = rng() % 4;
        rovaca_amd::apply_pcr_error_model(read.data(), gop.data(), R, model);
        phmm_oracle_apply_pcr_error_model(read.data(), gop2.data(), R, model);
        for (uint32_t i = 0; i < R; ++i)
            if (gop[i] != gop2[i]) {
                fprintf(stderr, "PCR model mismatch at %u (iteration %d)\n", i, it);
                return 2;
            }
        for (uint32_t i = 0; i < R; i += 7) checksum += (uint64_t)rovaca_amd::tandem_repeat_length(read.data(), R, i);
        // both oracles walk their matrices without stepping outside
        const float f = phmm_oracle_pair_f32(read.data(), q.data(), gop.data(), gop.data(), gcp.data(), R, hap.data(), H);
        const double d = phmm_oracle_pair_f64(read.data(), q.data(), gop.data(), gop.data(), gcp.data(), R, hap.data(), H);
        checksum += (uint64_t)(f > 0) + (uint64_t)(d > 0);
        const int32_t params[4] = {10, -15, -30, -5};
        sw_oracle_result res;
        std::vector<uint32_t> cigar(R + H + 8);
        if (sw_oracle_align(hap.data(), (int32_t)H, read.data(), (int32_t)R, params, 2048, &res, cigar.data(), (uint32_t)cigar.size()) != 0)
            return 3;
        checksum += res.cigar_count;
        // H8 on a matrix with capped, NaN and -inf entries
        const size_t nr = 1 + rng() % 20, nh = 1 + rng() % 9;
        std::vector<double> lik(nr * nh);
        std::vector<int32_t> rl(nr);
        for (auto &v : lik) v = (rng() % 17 == 0) ? -INFINITY : -(double)(rng() % 4000) / 10.0;
        for (auto &v : rl) v = 1 + (int32_t)(rng() % 250);
        rovaca_amd::normalize_likelihoods(lik.data(), nr, nh);
        checksum += rovaca_amd::poorly_modelled_reads(lik.data(), rl.data(), nr, nh).size();
    }
    // the CPU backend of the C ABI (rovaca_amd/csrc/pairhmm_cpu.cpp): scalar and AVX2 inter-pair kernels on small batches -- groups of
    // equal shape fill vectors, the rest goes scalar -- bit for bit against the oracle's per-pair routine; and the marginalisation
    for (int it = 0; it < 60; ++it) {
        const uint32_t n_reads = 1 + rng() % 12, n_haps = 1 + rng() % 5;
        const bool same_len = it % 2 == 0;
        const uint32_t R0 = 1 + rng() % 120, H0 = 1 + rng() % 200;
        std::vector<uint8_t> rb, rq, ri, rd, rc, hb;
        std::vector<uint32_t> roff, rlen, hoff, hlen;
        for (uint32_t r = 0; r < n_reads; ++r) {
            const uint32_t R = same_len ? R0 : 1 + rng() % 120;
            roff.push_back((uint32_t)rb.size());
            rlen.push_back(R);
            for (uint32_t i = 0; i < R; ++i) {
                rb.push_back((uint8_t)acgt[rng() % 5]);
                rq.push_back((uint8_t)(rng() % 200));  // the kernels mask with 127
                ri.push_back((uint8_t)(10 + rng() % 40));
                rd.push_back((uint8_t)(10 + rng() % 40));
                rc.push_back((uint8_t)(5 + rng() % 20));
            }
        }
        for (uint32_t h = 0; h < n_haps; ++h) {
            const uint32_t H = same_len ? H0 : 1 + rng() % 200;
            hoff.push_back((uint32_t)hb.size());
            hlen.push_back(H);
            for (uint32_t j = 0; j < H; ++j) hb.push_back((uint8_t)acgt[rng() % 5]);
        }
        phmm_batch b{};
        b.n_reads = n_reads;
        b.n_haps = n_haps;
        b.read_bases = rb.data();
        b.read_qual = rq.data();
        b.read_ins = ri.data();
        b.read_del = rd.data();
        b.read_gcp = rc.data();
        b.read_off = roff.data();
        b.read_len = rlen.data();
        b.hap_bases = hb.data();
        b.hap_off = hoff.data();
        b.hap_len = hlen.data();  // pair arrays null: the full cross product
        const size_t n = (size_t)n_reads * n_haps;
        for (int simd = 0; simd < 2; ++simd) {
            std::vector<float> raw(n);
            std::vector<double> raw64(n);
            std::vector<uint8_t> resc(n);
            uint64_t n_resc = 0, cells = 0;
            if (phmm::cpu::compute(b, 1 + rng() % 3, simd != 0, raw.data(), raw64.data(), resc.data(), &n_resc, &cells) != 0) return 4;
            for (size_t p = 0; p < n; ++p) {
                const uint32_t r = (uint32_t)(p / n_haps), h = (uint32_t)(p % n_haps);
                const float want = phmm_oracle_pair_f32(rb.data() + roff[r], rq.data() + roff[r], ri.data() + roff[r], rd.data() + roff[r],
                                                        rc.data() + roff[r], rlen[r], hb.data() + hoff[h], hlen[h]);
                if (memcmp(&want, &raw[p], 4) != 0) {
                    fprintf(stderr, "CPU backend differs from the oracle: pair %zu (simd %d, iteration %d)\n", p, simd, it);
                    return 5;
                }
            }
            checksum += n_resc + cells;
        }
        const uint32_t n_new = 1 + rng() % 4, stride = n_reads + rng() % 3;
        std::vector<double> old((size_t)n_haps * stride), a((size_t)n_new * n_reads), o((size_t)n_new * n_reads);
        std::vector<int32_t> map(n_haps);
        for (auto &v : old) v = (rng() % 11 == 0) ? NAN : -(double)(rng() % 1000) / 7.0;
        for (auto &m : map) m = (int32_t)(rng() % (n_new + 1)) - 1;
        rovaca_amd::marginal_likelihoods_direct(old.data(), n_haps, stride, n_reads, map.data(), n_new, a.data());
        phmm_oracle_marginalize(old.data(), n_haps, stride, n_reads, map.data(), n_new, o.data());
        if (memcmp(a.data(), o.data(), a.size() * sizeof(double)) != 0) return 6;
    }
    printf("sanitizer run clean, checksum %llu\n", (unsigned long long)checksum);
    return 0;
}
