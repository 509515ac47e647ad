/*
 * ref_harness.cpp -- thin extern "C" driver around the REFERENCE's own PairHMM kernels.  TEST INFRASTRUCTURE ONLY.
 *
 * Built only where /root/reference exists (this container): oracle/Makefile compiles the reference's five
 * kernel translation units from where they lie (core/avx2_s.cpp, core/avx2_d.cpp, core/avx512_s.cpp,
 * core/avx512_d.cpp, rovaca/avx_512_float.cc; flags from pairhmm/CMakeLists.txt:3-5 and CMakeLists.txt:48-49)
 * and links them with this file into oracle/_ref/libphmm_ref.so.  No reference source is copied into the repo;
 * the reference headers are found through -I at build time.
 *
 * What this file adds is only what the reference does around its kernels and that cannot be linked without
 * boost/htslib: filling ::TestCase (pairhmm_internal.cpp:569-576) and rovaca::TestCase
 * (pairhmm_internal.cpp:393-411, :450-483, :662-667), ConvertChar::init() (pairhmm_engine.cpp:44) and
 * flush-to-zero (pairhmm_internal.cpp:21-25).
 */
#include <xmmintrin.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "core/avx_impl.h"
#include "core/context.h"
#include "rovaca/avx_512_float.h"
#include "rovaca/common.h"
#include "rovaca/context.h"

namespace
{

inline void ftz_on() { _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON); }

/* pairhmm_internal.cpp:27-41 (static there, so restated): A=1 C=2 T=4 G=8 N=15, else 0 */
inline int32_t one_hot(uint8_t b)
{
    switch (b) {
        case 'A': return 0x1;
        case 'C': return 0x2;
        case 'T': return 0x4;
        case 'G': return 0x8;
        case 'N': return 0xF;
        default: return 0;
    }
}

::TestCase make_tc(const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c, uint32_t R,
                   const uint8_t *hap, uint32_t H)
{
    ::TestCase tc{};
    tc.rslen = R;
    tc.haplen = H;
    tc.hap = hap;
    tc.rs = rs;
    tc.q = q;
    tc.i = i;
    tc.d = d;
    tc.c = c;
    return tc;
}

/* One 16-read group, planes allocated once and reused over haplotypes (as call_new_pairhmm002 does). */
struct Group16
{
    static constexpr uint32_t L = 16;
    rovaca::TestCase tc{};
    std::vector<void *> owned;

    template <class T>
    T *alloc(size_t n)
    {
        void *p = nullptr;
        if (posix_memalign(&p, 64, (n * sizeof(T) + 63) / 64 * 64 + 64) != 0) abort();
        memset(p, 0, (n * sizeof(T) + 63) / 64 * 64 + 64);
        owned.push_back(p);
        return static_cast<T *>(p);
    }

    Group16(uint32_t max_read_len, uint32_t max_hap_len)
    {
        /* pairhmm_internal.cpp:393-411 */
        tc.hap_num = L;
        tc.read_num = L;
        size_t n = (size_t)L * max_read_len;
        tc.mm = alloc<float>(n);
        tc.mi = alloc<float>(n);
        tc.ii = alloc<float>(n);
        tc.md = alloc<float>(n);
        tc.dd = alloc<float>(n);
        tc.gapm = alloc<float>(n);
        tc.distm = alloc<float>(n);
        tc._1_distm = alloc<float>(n);
        tc.hap_len_arr = alloc<uint32_t>(L);
        tc.read_len_arr = alloc<uint32_t>(L);
        tc.haps = alloc<int32_t>(max_hap_len);
        tc.reads = alloc<int32_t>(n);
    }
    ~Group16()
    {
        for (void *p : owned) free(p);
    }
    Group16(const Group16 &) = delete;
    Group16 &operator=(const Group16 &) = delete;

    /* pairhmm_internal.cpp:450-483 (prepare_data) + :429-447 (lengths) */
    void set_reads(const uint8_t *const *rs, const uint8_t *const *q, const uint8_t *const *ins, const uint8_t *const *del,
                   const uint8_t *const *gcp, const uint32_t *len)
    {
        rovaca::Context<float> ctx{};
        uint32_t mn = UINT32_MAX, mx = 0;
        for (uint32_t k = 0; k < L; ++k) {
            tc.read_len_arr[k] = len[k];
            mn = len[k] < mn ? len[k] : mn;
            mx = len[k] > mx ? len[k] : mx;
            for (uint32_t j = 0; j < len[k]; ++j) {
                size_t at = k + (size_t)j * L;
                int _i = ins[k][j] & 127, _d = del[k][j] & 127, _c = gcp[k][j] & 127, _q = q[k][j] & 127;
                tc.reads[at] = one_hot(rs[k][j]);
                tc.mm[at] = ctx.set_mm_prob(_i, _d);
                tc.gapm[at] = 1.0f - rovaca::Context<float>::ph2pr[_c];
                tc.mi[at] = rovaca::Context<float>::ph2pr[_i];
                tc.ii[at] = rovaca::Context<float>::ph2pr[_c];
                tc.md[at] = rovaca::Context<float>::ph2pr[_d];
                tc.dd[at] = rovaca::Context<float>::ph2pr[_c];
                float dist = rovaca::Context<float>::ph2pr[_q];
                tc._1_distm[at] = 1.0f - dist;
                tc.distm[at] = dist / 3.0f;
            }
        }
        tc.max_read_len = mx;
        tc.min_read_len = mn;
    }

    /* pairhmm_internal.cpp:661-669 */
    void run(const uint8_t *hap, uint32_t H, float *out16)
    {
        tc.min_hap_len = tc.max_hap_len = H;
        for (uint32_t j = 0; j < H; ++j) tc.haps[j] = one_hot(hap[j]);
        rovaca::compute_full_prob_avx512_float_1xn(tc, out16);
    }
};

}  // namespace

extern "C" {

void ref_init(void)
{
    ConvertChar::init(); /* pairhmm_engine.cpp:44 */
    Context<float> cf;
    Context<double> cd;
    rovaca::Context<float> rf;
    (void)cf;
    (void)cd;
    (void)rf;
}

/* K2 / K3: the four per-pair kernels, one pair per call */
float ref_pair_avx512_s(const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c, uint32_t R,
                        const uint8_t *hap, uint32_t H)
{
    ftz_on();
    return compute_fp_avx512_s(make_tc(rs, q, i, d, c, R, hap, H));
}
float ref_pair_avx2_s(const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c, uint32_t R,
                      const uint8_t *hap, uint32_t H)
{
    ftz_on();
    return compute_fp_avx2_s(make_tc(rs, q, i, d, c, R, hap, H));
}
double ref_pair_avx512_d(const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c, uint32_t R,
                         const uint8_t *hap, uint32_t H)
{
    ftz_on();
    return compute_fp_avx512_d(make_tc(rs, q, i, d, c, R, hap, H));
}
double ref_pair_avx2_d(const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c, uint32_t R,
                       const uint8_t *hap, uint32_t H)
{
    ftz_on();
    return compute_fp_avx2_d(make_tc(rs, q, i, d, c, R, hap, H));
}

/* K1: 16 reads x 1 haplotype.  Read k is bytes [off[k], off[k]+len[k]) of the five concatenated arrays. */
void ref_group_16x1(const uint8_t *rb, const uint8_t *rq, const uint8_t *ri, const uint8_t *rd, const uint8_t *rc,
                    const uint32_t *off, const uint32_t *len, const uint8_t *hap, uint32_t H, float *out16)
{
    ftz_on();
    uint32_t mx = 0;
    const uint8_t *b[16], *q[16], *i[16], *d[16], *c[16];
    for (int k = 0; k < 16; ++k) {
        mx = len[k] > mx ? len[k] : mx;
        b[k] = rb + off[k];
        q[k] = rq + off[k];
        i[k] = ri + off[k];
        d[k] = rd + off[k];
        c[k] = rc + off[k];
    }
    Group16 g(mx, H);
    g.set_reads(b, q, i, d, c, len);
    g.run(hap, H, out16);
}

/* K4: dump the reference's tables */
void ref_tables(float *ph2pr_f, float *mm_f, double *ph2pr_d, double *mm_d)
{
    ref_init();
    const size_t n = ((MAX_QUAL + 1) * (MAX_QUAL + 2)) >> 1;
    memcpy(ph2pr_f, rovaca::Context<float>::ph2pr, sizeof(float) * 128);
    memcpy(mm_f, rovaca::Context<float>::matchToMatchProb, sizeof(float) * n);
    memcpy(ph2pr_d, Context<double>::ph2pr, sizeof(double) * 128);
    memcpy(mm_d, Context<double>::matchToMatchProb, sizeof(double) * n);
}

/*
 * Region-shaped batch through the 16x1 kernel, the way call_new_pairhmm002 drives it (:649-693): reads of one
 * region are taken 16 at a time (a short last group is padded by repeating its first read) and each group is
 * run against every haplotype of the region.  Regions are split over n_threads std::threads.
 * region r owns reads [rreg[r], rreg[r+1]) and haps [hreg[r], hreg[r+1]).  out is the flat pair list in
 * (region, read, hap) order, i.e. out[pair_base[r] + (read - rreg[r]) * nh + (hap - hreg[r])].
 * Returns wall seconds of the compute section.
 */
double ref_regions_16x1(const uint8_t *rb, const uint8_t *rq, const uint8_t *ri, const uint8_t *rd, const uint8_t *rc,
                        const uint32_t *roff, const uint32_t *rlen, const uint8_t *hb, const uint32_t *hoff,
                        const uint32_t *hlen, const uint32_t *rreg, const uint32_t *hreg, const uint64_t *pair_base,
                        uint32_t n_regions, float *out, int n_threads)
{
    ref_init();
    if (n_threads < 1) n_threads = 1;
    auto work = [&](uint32_t lo, uint32_t hi) {
        ftz_on();
        for (uint32_t r = lo; r < hi; ++r) {
            uint32_t r0 = rreg[r], r1 = rreg[r + 1], h0 = hreg[r], h1 = hreg[r + 1];
            uint32_t nh = h1 - h0, mxr = 0, mxh = 0;
            for (uint32_t k = r0; k < r1; ++k) mxr = rlen[k] > mxr ? rlen[k] : mxr;
            for (uint32_t k = h0; k < h1; ++k) mxh = hlen[k] > mxh ? hlen[k] : mxh;
            if (r1 == r0 || nh == 0) continue;
            Group16 g(mxr, mxh);
            for (uint32_t base = r0; base < r1; base += 16) {
                const uint8_t *b[16], *q[16], *i[16], *d[16], *c[16];
                uint32_t len[16], idx[16];
                for (uint32_t k = 0; k < 16; ++k) {
                    uint32_t rr = base + k < r1 ? base + k : base;
                    idx[k] = rr;
                    len[k] = rlen[rr];
                    b[k] = rb + roff[rr];
                    q[k] = rq + roff[rr];
                    i[k] = ri + roff[rr];
                    d[k] = rd + roff[rr];
                    c[k] = rc + roff[rr];
                }
                g.set_reads(b, q, i, d, c, len);
                for (uint32_t h = h0; h < h1; ++h) {
                    float res[16];
                    g.run(hb + hoff[h], hlen[h], res);
                    for (uint32_t k = 0; k < 16 && base + k < r1; ++k)
                        out[pair_base[r] + (uint64_t)(idx[k] - r0) * nh + (h - h0)] = res[k];
                }
            }
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (n_threads == 1) {
        work(0, n_regions);
    }
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back(work, (uint32_t)((uint64_t)n_regions * t / n_threads),
                            (uint32_t)((uint64_t)n_regions * (t + 1) / n_threads));
        for (auto &x : th) x.join();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

/* Same flat pair list through the per-pair AVX-512 kernel (the leftover-read path, :695-746). */
double ref_pairs_avx512_s(const uint8_t *rb, const uint8_t *rq, const uint8_t *ri, const uint8_t *rd, const uint8_t *rc,
                          const uint32_t *roff, const uint32_t *rlen, const uint8_t *hb, const uint32_t *hoff,
                          const uint32_t *hlen, const uint32_t *pr, const uint32_t *ph, uint64_t n_pairs, float *out,
                          int n_threads)
{
    ref_init();
    if (n_threads < 1) n_threads = 1;
    auto work = [&](uint64_t lo, uint64_t hi) {
        ftz_on();
        for (uint64_t p = lo; p < hi; ++p) {
            uint32_t r = pr[p], h = ph[p], o = roff[r];
            out[p] = compute_fp_avx512_s(make_tc(rb + o, rq + o, ri + o, rd + o, rc + o, rlen[r], hb + hoff[h], hlen[h]));
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (n_threads == 1) {
        work(0, n_pairs);
    }
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back(work, n_pairs * (uint64_t)t / n_threads, n_pairs * (uint64_t)(t + 1) / n_threads);
        for (auto &x : th) x.join();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
