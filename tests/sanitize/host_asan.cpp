// host_asan.cpp -- AddressSanitizer + UBSan run of the CPU-side code (the reference offers the same switch for its targets,
// cmake/asan.cmake): the product's host logic (H3, H4, H8) and the two oracles, on random inputs including the edge shapes
// (single base, repeats up to the caps, all-NaN rows).  Built and run by tests/test_sanitizers.py; exits 0 when clean.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../oracle/phmm_oracle.h"
#include "../../oracle/sw_oracle.h"
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
    printf("sanitizer run clean, checksum %llu\n", (unsigned long long)checksum);
    return 0;
}
