// pairhmm_cpu.cpp -- the CPU backend of the C ABI (BASELINE.json config 1: "CPU AVX2 path, plumbing, no GPU"; SURVEY.md 8b:
// "same entry point must have a CPU backend").  Product code of this repository, written from the recurrence of SURVEY.md 8a;
// it never touches oracle/.  A context made by phmm_create_cpu runs phmm_compute / phmm_upload / phmm_sweep / phmm_download
// here; a device context never falls back to it.
//
// Numerics contract: the reference's AVX-512 build -- core/avx512_s.cpp:229-238, rovaca/avx_512_float.cc:
// 84-88 as GCC contracts them --
//     t = Md*p_mm; t = fma(Xd,p_gapm,t); t = fma(Yd,p_gapm,t); M = t*prior
//     X = fma(Xu,p_xx, Mu*p_mx);  Y = fma(Yl,p_yy, Ml*p_my)
//     result = (sum_j M[R][j]) + (sum_j X[R][j]), both left to right
// with flush-to-zero (pairhmm_internal.cpp:21-25), tables from pairhmm_tables.cpp, fp32 pairs below 1e-28f re-run in fp64
// (scale 2^1020).  Two kernels: a scalar one (any pair, either type) and an AVX2 + FMA inter-pair one (eight pairs of the same
// read and haplotype length per vector, one pair per lane -- the shape of the reference's 16 x 1 kernel at half the width);
// both give the same bits.
#include "pairhmm_cpu.h"

#include <immintrin.h>
#include <xmmintrin.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <thread>
#include <utility>
#include <vector>

#include "pairhmm_tables.h"

namespace phmm
{
namespace cpu
{
namespace
{

// one-hot codes of pairhmm_internal.cpp:27-41: A C T G one bit each, N all four, anything else none
inline uint32_t base_code(uint8_t b)
{
    switch (b) {
    case 'A': return 1u;
    case 'C': return 2u;
    case 'T': return 4u;
    case 'G': return 8u;
    case 'N': return 15u;
    default: return 0u;
    }
}

template <typename T> struct Tab;
template <> struct Tab<float>
{
    static const float *ph2pr() { return host_tables().ph2pr_f32.data(); }
    static const float *mm() { return host_tables().mm_f32.data(); }
    static float init() { return std::ldexp(1.0f, 120); }  // core/context.h:148
    static float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
};
template <> struct Tab<double>
{
    static const double *ph2pr() { return host_tables().ph2pr_f64.data(); }
    static const double *mm() { return host_tables().mm_f64.data(); }
    static double init() { return std::ldexp(1.0, 1020); }  // core/context.h:106
    static double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
};

struct PairView
{
    const uint8_t *rs, *q, *i, *d, *c, *hap;
    uint32_t R, H;
};

// ---- scalar kernel: one pair, row by row over two rows of state ----
template <typename T>
T pair_scalar(const PairView &p, std::vector<T> &buf)
{
    const T *ph2pr = Tab<T>::ph2pr(), *mmt = Tab<T>::mm();
    const uint32_t R = p.R, H = p.H;
    buf.assign(6 * (size_t)(H + 1), T(0));
    T *Mp = buf.data(), *Xp = Mp + (H + 1), *Yp = Xp + (H + 1), *Mc = Yp + (H + 1), *Xc = Mc + (H + 1), *Yc = Xc + (H + 1);
    const T init_y = Tab<T>::init() / static_cast<T>(H);  // rovaca/avx_512_float.cc:32
    for (uint32_t j = 0; j <= H; ++j) Yp[j] = init_y;
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t _i = p.i[r] & 127u, _d = p.d[r] & 127u, _c = p.c[r] & 127u, _q = p.q[r] & 127u;  // core/avx512_s.cpp:90-93
        const uint32_t mx = _i > _d ? _i : _d, mn = _i > _d ? _d : _i;
        const T p_mm = mmt[((mx * (mx + 1)) >> 1) + mn];  // core/context.h:162-173
        const T p_gapm = T(1.0) - ph2pr[_c], p_mx = ph2pr[_i], p_xx = ph2pr[_c], p_my = ph2pr[_d], p_yy = ph2pr[_c];
        const T dist = ph2pr[_q];
        const T hi = T(1.0) - dist, lo = dist / T(3.0);
        const uint32_t rc = base_code(p.rs[r]);
        Mc[0] = Xc[0] = Yc[0] = T(0);
        for (uint32_t j = 1; j <= H; ++j) {
            const T prior = (rc & base_code(p.hap[j - 1])) ? hi : lo;
            T t = Mp[j - 1] * p_mm;
            t = Tab<T>::fma_(Xp[j - 1], p_gapm, t);
            t = Tab<T>::fma_(Yp[j - 1], p_gapm, t);
            Mc[j] = t * prior;
            Xc[j] = Tab<T>::fma_(Xp[j], p_xx, Mp[j] * p_mx);
            Yc[j] = Tab<T>::fma_(Yc[j - 1], p_yy, Mc[j - 1] * p_my);
        }
        std::swap(Mp, Mc);
        std::swap(Xp, Xc);
        std::swap(Yp, Yc);
    }
    T sm = T(0), sx = T(0);
    for (uint32_t j = 1; j <= H; ++j) {
        sm += Mp[j];
        sx += Xp[j];
    }
    return sm + sx;
}

// ---- AVX2 + FMA inter-pair kernel: eight pairs of equal (R, H), one per lane ----
__attribute__((target("avx2,fma"))) void pairs8_avx2(const PairView *p, float *out, std::vector<float> &fbuf, std::vector<int32_t> &ibuf)
{
    const float *ph2pr = Tab<float>::ph2pr(), *mmt = Tab<float>::mm();
    const uint32_t R = p[0].R, H = p[0].H;
    // state: two rows of (M, X, Y), 8 lanes each; haplotype codes [j][lane]
    fbuf.assign(6 * (size_t)(H + 1) * 8, 0.0f);
    ibuf.assign((size_t)(H + 1) * 8, 0);
    float *Mp = fbuf.data(), *Xp = Mp + (H + 1) * 8, *Yp = Xp + (H + 1) * 8, *Mc = Yp + (H + 1) * 8, *Xc = Mc + (H + 1) * 8,
          *Yc = Xc + (H + 1) * 8;
    int32_t *hc = ibuf.data();
    for (uint32_t j = 1; j <= H; ++j)
        for (int l = 0; l < 8; ++l) hc[j * 8 + l] = (int32_t)base_code(p[l].hap[j - 1]);
    const __m256 init_y = _mm256_set1_ps(Tab<float>::init() / static_cast<float>(H));
    for (uint32_t j = 0; j <= H; ++j) _mm256_storeu_ps(Yp + j * 8, init_y);
    const __m256i zero = _mm256_setzero_si256();
    alignas(32) float a_mm[8], a_gapm[8], a_mx[8], a_xx[8], a_my[8], a_yy[8], a_hi[8], a_lo[8];
    alignas(32) int32_t a_rc[8];
    for (uint32_t r = 0; r < R; ++r) {
        for (int l = 0; l < 8; ++l) {
            const uint32_t _i = p[l].i[r] & 127u, _d = p[l].d[r] & 127u, _c = p[l].c[r] & 127u, _q = p[l].q[r] & 127u;
            const uint32_t mx = _i > _d ? _i : _d, mn = _i > _d ? _d : _i;
            a_mm[l] = mmt[((mx * (mx + 1)) >> 1) + mn];
            a_gapm[l] = 1.0f - ph2pr[_c];
            a_mx[l] = ph2pr[_i];
            a_xx[l] = ph2pr[_c];
            a_my[l] = ph2pr[_d];
            a_yy[l] = ph2pr[_c];
            const float dist = ph2pr[_q];
            a_hi[l] = 1.0f - dist;
            a_lo[l] = dist / 3.0f;
            a_rc[l] = (int32_t)base_code(p[l].rs[r]);
        }
        const __m256 p_mm = _mm256_load_ps(a_mm), p_gapm = _mm256_load_ps(a_gapm), p_mx = _mm256_load_ps(a_mx), p_xx = _mm256_load_ps(a_xx),
                     p_my = _mm256_load_ps(a_my), p_yy = _mm256_load_ps(a_yy), hi = _mm256_load_ps(a_hi), lo = _mm256_load_ps(a_lo);
        const __m256i rc = _mm256_load_si256(reinterpret_cast<const __m256i *>(a_rc));
        __m256 m_left = _mm256_setzero_ps(), y_left = _mm256_setzero_ps();  // M[i][j-1], Y[i][j-1]; column 0 of a read row is 0
        _mm256_storeu_ps(Mc, m_left);
        _mm256_storeu_ps(Xc, m_left);
        _mm256_storeu_ps(Yc, m_left);
        for (uint32_t j = 1; j <= H; ++j) {
            const __m256i hit = _mm256_cmpeq_epi32(_mm256_and_si256(rc, _mm256_loadu_si256(reinterpret_cast<const __m256i *>(hc + j * 8))), zero);
            const __m256 prior = _mm256_blendv_ps(hi, lo, _mm256_castsi256_ps(hit));  // hit == all-ones where the codes do NOT meet
            __m256 t = _mm256_mul_ps(_mm256_loadu_ps(Mp + (j - 1) * 8), p_mm);
            t = _mm256_fmadd_ps(_mm256_loadu_ps(Xp + (j - 1) * 8), p_gapm, t);
            t = _mm256_fmadd_ps(_mm256_loadu_ps(Yp + (j - 1) * 8), p_gapm, t);
            const __m256 m = _mm256_mul_ps(t, prior);
            const __m256 x = _mm256_fmadd_ps(_mm256_loadu_ps(Xp + j * 8), p_xx, _mm256_mul_ps(_mm256_loadu_ps(Mp + j * 8), p_mx));
            const __m256 y = _mm256_fmadd_ps(y_left, p_yy, _mm256_mul_ps(m_left, p_my));
            _mm256_storeu_ps(Mc + j * 8, m);
            _mm256_storeu_ps(Xc + j * 8, x);
            _mm256_storeu_ps(Yc + j * 8, y);
            m_left = m;
            y_left = y;
        }
        std::swap(Mp, Mc);
        std::swap(Xp, Xc);
        std::swap(Yp, Yc);
    }
    __m256 sm = _mm256_setzero_ps(), sx = _mm256_setzero_ps();
    for (uint32_t j = 1; j <= H; ++j) {
        sm = _mm256_add_ps(sm, _mm256_loadu_ps(Mp + j * 8));
        sx = _mm256_add_ps(sx, _mm256_loadu_ps(Xp + j * 8));
    }
    _mm256_storeu_ps(out, _mm256_add_ps(sm, sx));
}

bool have_avx2_fma()
{
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return ok;
}

}  // namespace

const char *kernel_name(bool allow_simd) { return (allow_simd && have_avx2_fma()) ? "avx2+fma inter-pair (8 pairs per vector)" : "scalar"; }

int compute(const phmm_batch &b, unsigned n_threads, bool allow_simd, float *out_raw_f32, double *out_raw_f64, uint8_t *rescued,
            uint64_t *n_rescued, uint64_t *cells)
{
    const bool cross = b.pair_read == nullptr;
    const uint64_t n = cross ? (uint64_t)b.n_reads * b.n_haps : b.n_pairs;
    auto view = [&](uint64_t p, PairView &v) -> bool {
        const uint32_t r = cross ? (uint32_t)(p / b.n_haps) : b.pair_read[p], h = cross ? (uint32_t)(p % b.n_haps) : b.pair_hap[p];
        if (r >= b.n_reads || h >= b.n_haps) return false;
        const uint32_t ro = b.read_off[r];
        v = PairView{b.read_bases + ro, b.read_qual + ro, b.read_ins + ro, b.read_del + ro, b.read_gcp + ro, b.hap_bases + b.hap_off[h],
                     b.read_len[r], b.hap_len[h]};
        return v.R != 0 && v.H != 0;
    };
    // group pairs of equal (R, H) for the vector kernel; what does not fill a vector goes to the scalar kernel
    std::vector<uint64_t> vec_items, scalar_items;
    uint64_t total_cells = 0;
    {
        std::map<std::pair<uint32_t, uint32_t>, std::vector<uint64_t>> by_shape;
        PairView v;
        for (uint64_t p = 0; p < n; ++p) {
            if (!view(p, v)) return -1;
            total_cells += (uint64_t)v.R * v.H;
            by_shape[{v.R, v.H}].push_back(p);
        }
        const bool simd = allow_simd && have_avx2_fma();
        for (auto &kv : by_shape) {
            std::vector<uint64_t> &g = kv.second;
            size_t full = simd ? (g.size() / 8) * 8 : 0;
            vec_items.insert(vec_items.end(), g.begin(), g.begin() + full);
            scalar_items.insert(scalar_items.end(), g.begin() + full, g.end());
        }
    }
    if (cells) *cells = total_cells;
    const size_t n_vec = vec_items.size() / 8, n_work = n_vec + scalar_items.size();
    n_threads = std::max(1u, std::min<unsigned>(n_threads, (unsigned)std::max<size_t>(1, n_work)));
    std::vector<uint64_t> resc_t(n_threads, 0);
    auto worker = [&](unsigned t) {
        // the reference sets flush-to-zero on every calling thread (pairhmm_internal.cpp:21-25, :604)
        const unsigned saved = _mm_getcsr();
        _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
        std::vector<float> fbuf, sbuf;
        std::vector<double> dbuf;
        std::vector<int32_t> ibuf;
        auto finish = [&](uint64_t p, float x, const PairView &v) {
            out_raw_f32[p] = x;
            const bool resc = x < 1e-28f;  // rovaca/common.h:9, pairhmm_internal.cpp:578
            if (rescued) rescued[p] = resc;
            if (resc) {
                ++resc_t[t];
                if (out_raw_f64) out_raw_f64[p] = pair_scalar<double>(v, dbuf);
            }
        };
        for (size_t w = t; w < n_work; w += n_threads) {
            if (w < n_vec) {
                PairView v[8];
                float res[8];
                for (int l = 0; l < 8; ++l) view(vec_items[w * 8 + l], v[l]);
                pairs8_avx2(v, res, fbuf, ibuf);
                for (int l = 0; l < 8; ++l) finish(vec_items[w * 8 + l], res[l], v[l]);
            }
            else {
                PairView v;
                const uint64_t p = scalar_items[w - n_vec];
                view(p, v);
                finish(p, pair_scalar<float>(v, sbuf), v);
            }
        }
        _mm_setcsr(saved);
    };
    std::vector<std::thread> th;
    unsigned started = 1;
    try {
        for (; started < n_threads; ++started) th.emplace_back(worker, started);
    }
    catch (...) {
    }
    worker(0);
    for (unsigned t = started; t < n_threads; ++t) worker(t);
    for (auto &x : th) x.join();
    uint64_t nr = 0;
    for (uint64_t x : resc_t) nr += x;
    if (n_rescued) *n_rescued = nr;
    return 0;
}

}  // namespace cpu
}  // namespace phmm
