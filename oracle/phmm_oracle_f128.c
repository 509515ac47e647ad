/*
 * phmm_oracle_f128.c -- an INDEPENDENT high-precision statement of the PairHMM forward recurrence, in __float128 (libquadmath).
 *
 * TEST INFRASTRUCTURE (never linked into the product).  Purpose: the reference's fp64 kernels are defective as shipped
 * (core/avx512_d.cpp:249-388, core/avx2_d.cpp:226-360), so for pairs whose fp32 sum underflows -- the fp64
 * "rescue" -- the only oracle was this repository's own double restatement (oracle/phmm_oracle.c, phmm_oracle_pair_f64), which
 * the device kernels reproduce bit for bit: a mistake common to both would have passed every test.  This file shares no code
 * with either.  It states the INTENDED semantics of the fp64 path -- the recurrence of core/avx512_s.cpp:229-238 /
 * rovaca/avx_512_float.cc:84-88 with the transition and emission probabilities taken from the reference's double tables
 * (core/context.h:98-110: ph2pr[q] = pow(10, -q/10), matchToMatchProb) -- in 113-bit arithmetic, WITHOUT the 2^1020 scale
 * (quad precision has the range), as a plain full-matrix sweep in the textbook order (row by row, three whole rows kept), and
 * returns log10 of the likelihood.  The double path must agree with it to rounding error: tests/test_rescue_f128.py holds it to
 * 1e-9 in log10 on the committed fixture tests/golden/rescue_f128.npz (generator: oracle/gen_golden_f128.py).
 *
 * Tables are passed in (the caller hands over the reference-pinned double tables), so nothing here depends on phmm_oracle.c.
 */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>

typedef __float128 q128;

/* match rule of the kernels: A C G T match themselves, N matches anything, any other byte matches nothing
 * (pairhmm_internal.cpp:27-41: one-hot codes with N = all four bits; avx_512_float.cc:81: match <=> r & h != 0) */
static int sym_bits(uint8_t b)
{
    switch (b) {
    case 'A': return 1;
    case 'C': return 2;
    case 'T': return 4;
    case 'G': return 8;
    case 'N': return 15;
    default: return 0;
    }
}

/* log10 of sum_j (M[R][j] + X[R][j]) with Y[0][j] = 1/H; *scaled_below_denormal (nullable) = 1 when that likelihood times
 * 2^1020 lies below 2^-1075, i.e. where a double computation at the reference's scale has nothing left but zero */
double phmm_f128_pair_log10(const uint8_t *rs, const uint8_t *q, const uint8_t *ins, const uint8_t *del, const uint8_t *gcp, uint32_t R,
                            const uint8_t *hap, uint32_t H, const double *ph2pr /*[128]*/, const double *mm /*[32640]*/,
                            int *scaled_below_denormal)
{
    if (H == 0 || R == 0) return -HUGE_VAL;
    const size_t W = (size_t)H + 1;
    q128 *m = (q128 *)malloc(sizeof(q128) * 6 * W);
    if (!m) return 0.0 / 0.0;
    q128 *Mprev = m, *Xprev = m + W, *Yprev = m + 2 * W, *Mcur = m + 3 * W, *Xcur = m + 4 * W, *Ycur = m + 5 * W;
    const q128 one = 1, three = 3, y0 = one / (q128)H;
    for (size_t j = 0; j <= H; ++j) {  /* row 0: M = X = 0, Y = 1/H at every column (avx512_s.cpp:58-64, :149-160 without the scale) */
        Mprev[j] = 0;
        Xprev[j] = 0;
        Yprev[j] = y0;
    }
    for (uint32_t i = 1; i <= R; ++i) {
        const int qi = q[i - 1] & 127, ii = ins[i - 1] & 127, di = del[i - 1] & 127, ci = gcp[i - 1] & 127;  /* avx512_s.cpp:90-93 */
        const int hi = ii > di ? ii : di, lo = ii > di ? di : ii;
        const q128 p_mm = (q128)mm[((hi * (hi + 1)) >> 1) + lo];  /* context.h:120-131 */
        const q128 p_gapm = one - (q128)ph2pr[ci];                 /* avx512_s.cpp:94-101 */
        const q128 p_mx = (q128)ph2pr[ii], p_my = (q128)ph2pr[di], p_xx = (q128)ph2pr[ci], p_yy = (q128)ph2pr[ci];
        const q128 eps = (q128)ph2pr[qi];
        const q128 match = one - eps, mismatch = eps / three;
        const int rb = sym_bits(rs[i - 1]);
        Mcur[0] = 0;  /* column 0 of every read row is zero */
        Xcur[0] = 0;
        Ycur[0] = 0;
        for (uint32_t j = 1; j <= H; ++j) {
            const q128 prior = (rb & sym_bits(hap[j - 1])) ? match : mismatch;
            Mcur[j] = prior * (Mprev[j - 1] * p_mm + Xprev[j - 1] * p_gapm + Yprev[j - 1] * p_gapm);
            Xcur[j] = Mprev[j] * p_mx + Xprev[j] * p_xx;
            Ycur[j] = Mcur[j - 1] * p_my + Ycur[j - 1] * p_yy;
        }
        q128 *t;
        t = Mprev, Mprev = Mcur, Mcur = t;
        t = Xprev, Xprev = Xcur, Xcur = t;
        t = Yprev, Yprev = Ycur, Ycur = t;
    }
    q128 sum = 0;
    for (uint32_t j = 1; j <= H; ++j) sum += Mprev[j] + Xprev[j];
    free(m);
    if (scaled_below_denormal) *scaled_below_denormal = sum * ldexpq(one, 1020) < ldexpq(one, -1075);
    if (sum <= 0) return -HUGE_VAL;
    return (double)log10q(sum);
}
