/*
 * phmm_oracle.c -- CPU restatement of Rovaca's PairHMM path.  TEST INFRASTRUCTURE ONLY (see phmm_oracle.h).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/haplotypecaller/pairhmm/).  Nothing here is copied: the reference is AVX
 * intrinsics over lane-packed data, this is scalar C over one pair at a time.
 *
 * Numerics contract (what makes the fp32 result bit-identical to the reference's AVX-512 build):
 *   - g++ -O3 contracts the reference's  mul(a,b)+mul(c,d)  into  fma(c,d, round(a*b))  in both
 *     AVX-512 translation units (checked in the disassembly of rovaca/avx_512_float.cc and
 *     core/avx512_s.cpp: vmulps + vfmadd231ps/vfmadd132ps).  We write that explicitly with fmaf()
 *     and build with -ffp-contract=off so the compiler adds or removes nothing.
 *   - flush-to-zero is on while a pair is evaluated (pairhmm_internal.cpp:21-25).
 *   - the last row is summed left to right, M and X separately, then added
 *     (rovaca/avx_512_float.cc:166-180, core/avx512_s.cpp:373-393).
 * Build: gcc -O2 -ffp-contract=off -mfma (oracle/Makefile).
 */
#include "phmm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

/* ------------------------------------------------------------------------------------------------
 * K4  lookup tables -- core/context.h:13-189 (rovaca/context.h is the same text in a namespace)
 * ---------------------------------------------------------------------------------------------- */
#define MAX_QUAL 254
#define JAC_TOL 8.0
#define JAC_STEP 0.0001
#define JAC_INV_STEP (1.0 / JAC_STEP)
enum { JAC_SIZE = 80001 }; /* (int)(8.0 / 0.0001) + 1, core/context.h:11 */

static float g_ph2pr_f[128];
static double g_ph2pr_d[128];
static float g_jac_f[JAC_SIZE];
static double g_jac_d[JAC_SIZE];
static float g_mm_f[PHMM_ORACLE_MM_TABLE_SIZE];
static double g_mm_d[PHMM_ORACLE_MM_TABLE_SIZE];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

/* core/context.h:61  fastRound */
static int fast_round_f(float d) { return (d > 0.0f) ? (int)(d + 0.5f) : (int)(d - 0.5f); }
static int fast_round_d(double d) { return (d > 0.0) ? (int)(d + 0.5) : (int)(d - 0.5); }

/* core/context.h:63-85  approximateLog10SumLog10, NUMBER = float */
static float approx_log10_sum_f(float small, float big)
{
    if (small > big) {
        float t = big;
        big = small;
        small = t;
    }
    if (isinf(small) || isinf(big)) return big;
    float diff = big - small;
    if (diff >= ((float)JAC_TOL)) return big;
    int ind = fast_round_f((float)(diff * ((float)JAC_INV_STEP)));
    return big + g_jac_f[ind];
}

/* same, NUMBER = double */
static double approx_log10_sum_d(double small, double big)
{
    if (small > big) {
        double t = big;
        big = small;
        small = t;
    }
    if (isinf(small) || isinf(big)) return big;
    double diff = big - small;
    if (diff >= JAC_TOL) return big;
    int ind = fast_round_d(diff * JAC_INV_STEP);
    return big + g_jac_d[ind];
}

static void build_tables(void)
{
    /* core/context.h:41-46  Jacobian log table, computed in double, stored as NUMBER */
    for (int k = 0; k < JAC_SIZE; k++) {
        double v = log10(1.0 + pow(10.0, -((double)k) * JAC_STEP));
        g_jac_d[k] = v;
        g_jac_f[k] = (float)v;
    }
    /* core/context.h:49-59  matchToMatchProb, triangular, index (max*(max+1))/2 + min */
    double LN10 = log(10);
    double INV_LN10 = 1.0 / LN10;
    for (int i = 0, offset = 0; i <= MAX_QUAL; offset += ++i) {
        for (int j = 0; j <= i; j++) {
            double s_f = approx_log10_sum_f((float)(-0.1 * i), (float)(-0.1 * j));
            double l_f = log1p(-fmin(1.0, pow(10, s_f))) * INV_LN10;
            g_mm_f[offset + j] = (float)(pow(10, l_f));
            double s_d = approx_log10_sum_d(-0.1 * i, -0.1 * j);
            double l_d = log1p(-fmin(1.0, pow(10, s_d))) * INV_LN10;
            g_mm_d[offset + j] = pow(10, l_d);
        }
    }
    /* core/context.h:144-146 (float: powf), :102-104 (double: pow) */
    for (int x = 0; x < 128; x++) {
        g_ph2pr_f[x] = powf(10.f, -((float)x) / 10.f);
        g_ph2pr_d[x] = pow(10.0, -((double)x) / 10.0);
    }
}

static void tables(void) { pthread_once(&g_once, build_tables); }

const float *phmm_oracle_ph2pr_f32(void) { tables(); return g_ph2pr_f; }
const double *phmm_oracle_ph2pr_f64(void) { tables(); return g_ph2pr_d; }
const float *phmm_oracle_mm_table_f32(void) { tables(); return g_mm_f; }
const double *phmm_oracle_mm_table_f64(void) { tables(); return g_mm_d; }

/* core/context.h:162-173 set_mm_prob (float), :120-131 (double).  Qualities here are already & 127,
 * so the MAX_QUAL < maxQual branch is unreachable on the path. */
float phmm_oracle_mm_prob_f32(int ins, int del)
{
    tables();
    int mn = del, mx = ins;
    if (ins <= del) {
        mn = ins;
        mx = del;
    }
    return g_mm_f[((mx * (mx + 1)) >> 1) + mn];
}

double phmm_oracle_mm_prob_f64(int ins, int del)
{
    tables();
    int mn = del, mx = ins;
    if (ins <= del) {
        mn = ins;
        mx = del;
    }
    return g_mm_d[((mx * (mx + 1)) >> 1) + mn];
}

/* ------------------------------------------------------------------------------------------------
 * K5  base codes -- pairhmm_internal.cpp:27-41 (convert_arr): A=1 C=2 T=4 G=8 N=15, anything else 0;
 * a read base matches a haplotype base iff (r & h) != 0 (rovaca/avx_512_float.cc:81).
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t base_code(uint8_t b)
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

/* ------------------------------------------------------------------------------------------------
 * K1/K2  the forward recurrence, fp32 -- rovaca/avx_512_float.cc:29-181 (one lane of it) ==
 * core/avx512_s.cpp:259-397 (one pair).  Row-major sweep over a (H+1)-wide previous row.
 * ---------------------------------------------------------------------------------------------- */
float phmm_oracle_pair_f32(const uint8_t *rs, const uint8_t *q, const uint8_t *ins, const uint8_t *del, const uint8_t *gcp,
                           uint32_t R, const uint8_t *hap, uint32_t H)
{
    tables();
    if (H == 0) return 0.0f;
    unsigned int saved_csr = _mm_getcsr();
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON); /* pairhmm_internal.cpp:21-25 */

    float *buf = (float *)malloc(sizeof(float) * 3 * (size_t)(H + 1));
    uint32_t *hc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(H + 1));
    float *Mp = buf, *Xp = buf + (H + 1), *Yp = buf + 2 * (size_t)(H + 1);
    /* rovaca/avx_512_float.cc:33-42: M[0][*]=X[0][*]=0, Y[0][*]=2^120/H */
    float init_y = ldexpf(1.f, 120) / (float)H;
    for (uint32_t j = 0; j <= H; j++) {
        Mp[j] = 0.0f;
        Xp[j] = 0.0f;
        Yp[j] = init_y;
    }
    for (uint32_t j = 0; j < H; j++) hc[j] = base_code(hap[j]);

    for (uint32_t r = 0; r < R; r++) {
        /* pairhmm_internal.cpp:466-480 (prepare_data) == core/avx512_s.cpp:90-101 */
        int _i = ins[r] & 127, _d = del[r] & 127, _c = gcp[r] & 127, _q = q[r] & 127;
        float p_mm = phmm_oracle_mm_prob_f32(_i, _d);
        float p_gapm = 1.0f - g_ph2pr_f[_c];
        float p_mx = g_ph2pr_f[_i];
        float p_xx = g_ph2pr_f[_c];
        float p_my = g_ph2pr_f[_d];
        float p_yy = g_ph2pr_f[_c];
        float dist = g_ph2pr_f[_q];
        float one_minus = 1.0f - dist;
        float dist3 = dist / 3.0f;
        uint32_t rc = base_code(rs[r]);

        /* column 0 of this row: all zero; the diagonal neighbour of (r, 1) is row r-1 col 0 */
        float Md = Mp[0], Xd = Xp[0], Yd = Yp[0];
        float Ml = 0.0f, Yl = 0.0f; /* M[r][j-1], Y[r][j-1] */
        Mp[0] = 0.0f;
        Xp[0] = 0.0f;
        Yp[0] = 0.0f;
        for (uint32_t j = 1; j <= H; j++) {
            float Mu = Mp[j], Xu = Xp[j], Yu = Yp[j]; /* row r-1, column j */
            float prior = (rc & hc[j - 1]) ? one_minus : dist3;
            /* rovaca/avx_512_float.cc:84-88 with GCC's contraction made explicit */
            float t = Md * p_mm;
            t = fmaf(Xd, p_gapm, t);
            t = fmaf(Yd, p_gapm, t);
            float M = t * prior;
            float X = fmaf(Xu, p_xx, Mu * p_mx);
            float Y = fmaf(Yl, p_yy, Ml * p_my);
            Md = Mu;
            Xd = Xu;
            Yd = Yu;
            Mp[j] = M;
            Xp[j] = X;
            Yp[j] = Y;
            Ml = M;
            Yl = Y;
        }
    }
    /* rovaca/avx_512_float.cc:166-180 */
    float sum_m = 0.0f, sum_x = 0.0f;
    for (uint32_t j = 1; j <= H; j++) {
        sum_m = sum_m + Mp[j];
        sum_x = sum_x + Xp[j];
    }
    float result = sum_m + sum_x;
    free(buf);
    free(hc);
    _mm_setcsr(saved_csr);
    return result;
}

/* ------------------------------------------------------------------------------------------------
 * K3  the same recurrence in double, scale 2^1020.  The reference's compute_fp_avx512_d /
 * compute_fp_avx2_d are defective as shipped, so THIS function is the rescue
 * oracle; it keeps the fp32 kernels' operation order so that a GPU kernel can match it bit for bit.
 * Intended semantics: core/avx512_d.cpp:249-388, core/context.h:98-110.
 * ---------------------------------------------------------------------------------------------- */
double phmm_oracle_pair_f64(const uint8_t *rs, const uint8_t *q, const uint8_t *ins, const uint8_t *del, const uint8_t *gcp,
                            uint32_t R, const uint8_t *hap, uint32_t H)
{
    tables();
    if (H == 0) return 0.0;
    double *buf = (double *)malloc(sizeof(double) * 3 * (size_t)(H + 1));
    uint32_t *hc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(H + 1));
    double *Mp = buf, *Xp = buf + (H + 1), *Yp = buf + 2 * (size_t)(H + 1);
    double init_y = ldexp(1.0, 1020) / (double)H;
    for (uint32_t j = 0; j <= H; j++) {
        Mp[j] = 0.0;
        Xp[j] = 0.0;
        Yp[j] = init_y;
    }
    for (uint32_t j = 0; j < H; j++) hc[j] = base_code(hap[j]);
    for (uint32_t r = 0; r < R; r++) {
        int _i = ins[r] & 127, _d = del[r] & 127, _c = gcp[r] & 127, _q = q[r] & 127;
        double p_mm = phmm_oracle_mm_prob_f64(_i, _d);
        double p_gapm = 1.0 - g_ph2pr_d[_c];
        double p_mx = g_ph2pr_d[_i];
        double p_xx = g_ph2pr_d[_c];
        double p_my = g_ph2pr_d[_d];
        double p_yy = g_ph2pr_d[_c];
        double dist = g_ph2pr_d[_q];
        double one_minus = 1.0 - dist;
        double dist3 = dist / 3.0;
        uint32_t rc = base_code(rs[r]);
        double Md = Mp[0], Xd = Xp[0], Yd = Yp[0];
        double Ml = 0.0, Yl = 0.0;
        Mp[0] = 0.0;
        Xp[0] = 0.0;
        Yp[0] = 0.0;
        for (uint32_t j = 1; j <= H; j++) {
            double Mu = Mp[j], Xu = Xp[j], Yu = Yp[j];
            double prior = (rc & hc[j - 1]) ? one_minus : dist3;
            double t = Md * p_mm;
            t = fma(Xd, p_gapm, t);
            t = fma(Yd, p_gapm, t);
            double M = t * prior;
            double X = fma(Xu, p_xx, Mu * p_mx);
            double Y = fma(Yl, p_yy, Ml * p_my);
            Md = Mu;
            Xd = Xu;
            Yd = Yu;
            Mp[j] = M;
            Xp[j] = X;
            Yp[j] = Y;
            Ml = M;
            Yl = Y;
        }
    }
    double sum_m = 0.0, sum_x = 0.0;
    for (uint32_t j = 1; j <= H; j++) {
        sum_m = sum_m + Mp[j];
        sum_x = sum_x + Xp[j];
    }
    double result = sum_m + sum_x;
    free(buf);
    free(hc);
    return result;
}

/* ------------------------------------------------------------------------------------------------
 * H7  result conversion + rescue -- pairhmm_internal.cpp:577-585; constants adapter.h:25-26,
 * rovaca/common.h:9 (MIN_ACCEPTED 1e-28f).  Note the FLOAT subtraction before widening.
 * ---------------------------------------------------------------------------------------------- */
double phmm_oracle_finalize_f32(float raw) { return (double)(log10f(raw) - 36.1236000061f); }
double phmm_oracle_finalize_f64(double raw) { return log10(raw) - 307.050595577260822; }

double phmm_oracle_pair_log10(const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c,
                              uint32_t R, const uint8_t *hap, uint32_t H, float *raw_f32, int *rescued)
{
    float rf = phmm_oracle_pair_f32(rs, q, i, d, c, R, hap, H);
    if (raw_f32) *raw_f32 = rf;
    if (rf < 1e-28f) {
        if (rescued) *rescued = 1;
        return phmm_oracle_finalize_f64(phmm_oracle_pair_f64(rs, q, i, d, c, R, hap, H));
    }
    if (rescued) *rescued = 0;
    return phmm_oracle_finalize_f32(rf);
}

/* ------------------------------------------------------------------------------------------------
 * flat batch (the layout of include/pairhmm_hip.h), optional pthread split
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *rb, *rq, *ri, *rd, *rc, *hb;
    const uint32_t *roff, *rlen, *hoff, *hlen, *pr, *ph;
    uint64_t lo, hi;
    float *raw;
    double *l10;
    uint8_t *resc;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *jb = (batch_job *)arg;
    for (uint64_t p = jb->lo; p < jb->hi; p++) {
        uint32_t r = jb->pr[p], h = jb->ph[p];
        uint32_t ro = jb->roff[r];
        float rf;
        int rs = 0;
        double v = phmm_oracle_pair_log10(jb->rb + ro, jb->rq + ro, jb->ri + ro, jb->rd + ro, jb->rc + ro, jb->rlen[r],
                                          jb->hb + jb->hoff[h], jb->hlen[h], &rf, &rs);
        if (jb->raw) jb->raw[p] = rf;
        if (jb->l10) jb->l10[p] = v;
        if (jb->resc) jb->resc[p] = (uint8_t)rs;
    }
    return NULL;
}

void phmm_oracle_batch(const uint8_t *read_bases, const uint8_t *read_qual, const uint8_t *read_ins,
                       const uint8_t *read_del, const uint8_t *read_gcp, const uint32_t *read_off,
                       const uint32_t *read_len, const uint8_t *hap_bases, const uint32_t *hap_off,
                       const uint32_t *hap_len, const uint32_t *pair_read, const uint32_t *pair_hap, uint64_t n_pairs,
                       float *out_raw, double *out_log10, uint8_t *out_rescued, int n_threads)
{
    tables();
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n_pairs) n_threads = n_pairs ? (int)n_pairs : 1;
    batch_job *jobs = (batch_job *)calloc((size_t)n_threads, sizeof(batch_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        batch_job jb = {read_bases, read_qual, read_ins, read_del, read_gcp, hap_bases, read_off, read_len, hap_off,
                        hap_len,    pair_read, pair_hap, n_pairs * (uint64_t)t / (uint64_t)n_threads,
                        n_pairs * (uint64_t)(t + 1) / (uint64_t)n_threads, out_raw, out_log10, out_rescued};
        jobs[t] = jb;
        if (n_threads > 1) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    if (n_threads == 1) {
        batch_worker(&jobs[0]);
    }
    else {
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    free(jobs);
    free(th);
}

/* ------------------------------------------------------------------------------------------------
 * H3  pairhmm_internal.cpp:434-439 (== :545-550, :710-715); MIN_QUALITY adapter.h:5
 * ---------------------------------------------------------------------------------------------- */
void phmm_oracle_modify_quals(uint8_t *qual, uint32_t n, uint8_t mq, int32_t min_q)
{
    for (uint32_t x = 0; x < n; x++) {
        uint8_t v = qual[x] < mq ? qual[x] : mq;
        qual[x] = ((int32_t)v < min_q) ? 6 : v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * H4  PCR indel model -- pairhmm_internal.cpp:195-229, :65-119, :121-159; tables adapter.h:13-15
 * ---------------------------------------------------------------------------------------------- */
static const int PCR_HOSTILE[21] = {40, 40, 39, 38, 37, 36, 34, 32, 28, 23, 17, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10};
static const int PCR_AGGRESSIVE[21] = {40, 40, 40, 39, 39, 39, 38, 38, 37, 37, 36, 35, 34, 33, 32, 30, 28, 26, 23, 20, 17};
static const int PCR_CONSERVATIVE[21] = {40, 40, 40, 40, 39, 39, 39, 39, 39, 38, 38, 38, 37, 37, 37, 36, 36, 35, 34, 33, 33};

/* pairhmm_internal.cpp:195-229 */
static int32_t number_of_repetitions(const uint8_t *unit, int32_t unit_off, int32_t unit_len, const uint8_t *test,
                                     int32_t test_off, int32_t test_len, int leading)
{
    if (test_len == 0) return 0;
    int32_t n = 0;
    int32_t diff = test_len - unit_len;
    if (leading) {
        for (int32_t start = 0; start <= diff; start += unit_len) {
            if (memcmp(test + start + test_off, unit + unit_off, (size_t)unit_len) == 0)
                ++n;
            else
                return n;
        }
    }
    else {
        for (int32_t start = diff; start >= 0; start -= unit_len) {
            if (memcmp(test + start + test_off, unit + unit_off, (size_t)unit_len) == 0)
                ++n;
            else
                return n;
        }
    }
    return n;
}

/* pairhmm_internal.cpp:65-119.  Kept quirk for quirk: the backward unit's recorded start is
 * offset-str-1, so the "same unit" test at :103 never fires and the :108 re-count always runs;
 * max_bw keeps the value of the LAST unit length tried when no repeat is found. */
int32_t phmm_oracle_find_tandem_repeat_units(const uint8_t *bases, uint32_t num, uint32_t offset1)
{
    int32_t offset = (int32_t)offset1;
    int32_t base_num = (int32_t)num;
    int32_t max_bw = 0;
    int32_t bw_first = offset, bw_second = 1;
    for (int32_t str = 1; str <= 8; str++) {
        if (offset + 1 - str < 0) break;
        max_bw = number_of_repetitions(bases, offset - str + 1, str, bases, 0, offset + 1, 0);
        if (max_bw > 1) {
            bw_first = offset - str - 1;
            bw_second = str;
            break;
        }
    }
    int32_t max_rl = max_bw;
    if (offset < base_num - 1) {
        int32_t fw_first = offset + 1, fw_second = 1;
        int32_t max_fw = 0;
        for (int32_t str = 1; str <= 8; str++) {
            if (offset + str + 1 > base_num) break;
            max_fw = number_of_repetitions(bases, offset + 1, str, bases, offset + 1, base_num - offset - 1, 1);
            if (max_fw > 1) {
                fw_first = offset + 1;
                fw_second = str;
                break;
            }
        }
        if (fw_first == bw_first && fw_second == bw_second) {
            max_rl = max_bw + max_fw;
        }
        else {
            max_bw = number_of_repetitions(bases, fw_first, fw_second, bases, 0, offset + 1, 0);
            max_rl = max_bw + max_fw;
        }
    }
    if (max_rl > 20) max_rl = 20;
    return max_rl;
}

/* pairhmm_internal.cpp:121-159 */
void phmm_oracle_apply_pcr_error_model(const uint8_t *bases, uint8_t *gap_qual, uint32_t n, int model)
{
    const int *cache = model == 1 ? PCR_HOSTILE : model == 2 ? PCR_AGGRESSIVE : model == 3 ? PCR_CONSERVATIVE : NULL;
    if (!cache) return;
    for (uint32_t i = 1; i < n; i++) {
        int32_t rl = phmm_oracle_find_tandem_repeat_units(bases, n, i - 1);
        if (!((int)gap_qual[i - 1] < cache[rl])) gap_qual[i - 1] = (uint8_t)cache[rl];
    }
}

/* ------------------------------------------------------------------------------------------------
 * H8  pairhmm_internal.cpp:53-63 (normalize), :161-182 (filter); constants adapter.h:20-23
 * ---------------------------------------------------------------------------------------------- */
void phmm_oracle_normalize(double *lik, uint32_t n_reads, uint32_t n_haps)
{
    for (uint32_t r = 0; r < n_reads; r++) {
        double *row = lik + (size_t)r * n_haps;
        double best = row[0];
        for (uint32_t h = 1; h < n_haps; h++)
            if (best < row[h]) best = row[h]; /* std::max_element */
        double cmp = best + (-4.5);
        for (uint32_t h = 0; h < n_haps; h++) row[h] = row[h] < cmp ? cmp : row[h];
    }
}

uint32_t phmm_oracle_filter(const double *lik, const uint32_t *read_len, uint32_t n_reads, uint32_t n_haps, uint8_t *keep)
{
    uint32_t kept = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        const double *row = lik + (size_t)r * n_haps;
        double best = row[0];
        for (uint32_t h = 1; h < n_haps; h++)
            if (best < row[h]) best = row[h];
        double lim = fmin(2.0, ceil((double)(int32_t)read_len[r] * 0.02)) * (-4.0);
        keep[r] = !(best < lim);
        kept += keep[r];
    }
    return kept;
}

/* ------------------------------------------------------------------------------------------------
 * next-3  allele marginalisation -- genotype/allele_likelihoods.hpp:702-732 (marginal_likelihoods_direct), one
 * sample: every new allele's row starts at -inf (:712); each old allele whose map entry is not s_missing_index
 * (-1, :26, :717-719) raises the new row wherever its value is strictly greater (:724-727), old alleles in
 * ascending order.  A NaN never compares greater, so it is skipped.  old is [n_old][stride] with evidence_count
 * valid values per row (the reference's rows may be wider than the evidence list), out is [n_new][evidence_count].
 * ---------------------------------------------------------------------------------------------- */
void phmm_oracle_marginalize(const double *old, uint32_t n_old, uint32_t stride, uint32_t evidence_count, const int32_t *old2new,
                             uint32_t n_new, double *out)
{
    for (size_t k = 0; k < (size_t)n_new * evidence_count; k++) out[k] = -INFINITY;
    for (uint32_t a = 0; a < n_old; a++) {
        int32_t na = old2new[a];
        if (na == -1) continue;
        const double *src = old + (size_t)a * stride;
        double *dst = out + (size_t)na * evidence_count;
        for (uint32_t r = 0; r < evidence_count; r++) {
            double v = src[r];
            if (v > dst[r]) dst[r] = v;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * H1/H2  one region end to end -- pairhmm_internal.cpp:505-594 (call_old_pairhmm); the AVX-512 driver
 * (:596-753) differs only in how reads are packed into lanes, which does not change any per-pair
 * value.
 * ---------------------------------------------------------------------------------------------- */
uint32_t phmm_oracle_call_region(const uint8_t *read_bases, const uint8_t *read_qual, const uint32_t *read_off,
                                 const uint32_t *read_len, const uint8_t *read_mq, uint32_t n_reads,
                                 const uint8_t *hap_bases, const uint32_t *hap_off, const uint32_t *hap_len,
                                 uint32_t n_haps, int32_t min_q, int pcr_model, double *out, uint32_t *survivors)
{
    if (n_reads == 0 || n_haps == 0) return 0; /* :508-510 */
    uint32_t max_len = 0;
    for (uint32_t r = 0; r < n_reads; r++)
        if (read_len[r] > max_len) max_len = read_len[r];
    uint8_t *q = (uint8_t *)malloc(max_len + 1), *gop = (uint8_t *)malloc(max_len + 1), *gcp = (uint8_t *)malloc(max_len + 1);
    double *lik = (double *)malloc(sizeof(double) * (size_t)n_reads * n_haps);
    for (uint32_t r = 0; r < n_reads; r++) {
        uint32_t n = read_len[r];
        const uint8_t *b = read_bases + read_off[r];
        memcpy(q, read_qual + read_off[r], n);
        phmm_oracle_modify_quals(q, n, read_mq[r], min_q); /* :545-550 */
        memset(gop, '-', n);                               /* :552-554 */
        if (pcr_model != 0) phmm_oracle_apply_pcr_error_model(b, gop, n, pcr_model);
        memset(gcp, '+' - 33, n); /* :561-563 */
        for (uint32_t h = 0; h < n_haps; h++) {
            /* del gap-open aliases the ins buffer, :559 */
            lik[(size_t)r * n_haps + h] =
                phmm_oracle_pair_log10(b, q, gop, gop, gcp, n, hap_bases + hap_off[h], hap_len[h], NULL, NULL);
        }
    }
    phmm_oracle_normalize(lik, n_reads, n_haps); /* :589 */
    uint8_t *keep = (uint8_t *)malloc(n_reads);
    uint32_t kept = phmm_oracle_filter(lik, read_len, n_reads, n_haps, keep); /* :590 */
    /* :516 result pre-filled with NaN, :591 transpose of the surviving rows */
    for (size_t k = 0; k < (size_t)n_haps * n_reads; k++) out[k] = NAN;
    uint32_t k = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        if (!keep[r]) continue;
        for (uint32_t h = 0; h < n_haps; h++) out[(size_t)h * n_reads + k] = lik[(size_t)r * n_haps + h];
        if (survivors) survivors[k] = r;
        k++;
    }
    free(q);
    free(gop);
    free(gcp);
    free(lik);
    free(keep);
    return kept;
}
