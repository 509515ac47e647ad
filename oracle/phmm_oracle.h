/*
 * phmm_oracle.h -- CPU restatement of Rovaca's PairHMM path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle: a plain-C restatement of the reference algorithm, written from the
 * recurrence and host logic in /root/reference/src/haplotypecaller/pairhmm/ (file:line cited at every
 * function in phmm_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product (rovaca_amd/csrc, rovaca_amd/host) never links or calls it.
 *
 * Pinning: the fp32 functions are checked bit-for-bit against the reference's own AVX-512 kernels
 * compiled from /root/reference (oracle/_ref, see oracle/Makefile) and against the golden vectors
 * committed under tests/golden/ that were generated from those kernels (oracle/gen_golden.py).
 * The reference ships no PairHMM test or fixture of its own.
 */
#ifndef PHMM_ORACLE_H_
#define PHMM_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHMM_ORACLE_MM_TABLE_SIZE (((254 + 1) * (254 + 2)) >> 1) /* core/context.h:26 */

/* K4: lookup tables (core/context.h:13-189).  Pointers stay valid for the life of the process. */
const float *phmm_oracle_ph2pr_f32(void);   /* [128]  */
const double *phmm_oracle_ph2pr_f64(void);  /* [128]  */
const float *phmm_oracle_mm_table_f32(void);  /* [PHMM_ORACLE_MM_TABLE_SIZE] */
const double *phmm_oracle_mm_table_f64(void); /* [PHMM_ORACLE_MM_TABLE_SIZE] */
float phmm_oracle_mm_prob_f32(int ins, int del);
double phmm_oracle_mm_prob_f64(int ins, int del);

/* K1/K2: raw forward sum of one (read, haplotype) pair, scaled by 2^120 (fp32) / 2^1020 (fp64).
 * Arrays are raw phred bytes (not +33), as ::TestCase (core/common.h:13-24). */
float phmm_oracle_pair_f32(const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c,
                           uint32_t rslen, const uint8_t *hap, uint32_t haplen);
double phmm_oracle_pair_f64(const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c,
                            uint32_t rslen, const uint8_t *hap, uint32_t haplen);

/* H7: raw fp32 sum -> final log10 likelihood, with the fp64 rescue below 1e-28f
 * (pairhmm_internal.cpp:577-585).  *rescued is set to 1 when the fp64 path was taken. */
double phmm_oracle_pair_log10(const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c,
                              uint32_t rslen, const uint8_t *hap, uint32_t haplen, float *raw_f32, int *rescued);
double phmm_oracle_finalize_f32(float raw);  /* (double)(log10f(raw) - 36.1236000061f) */
double phmm_oracle_finalize_f64(double raw); /* log10(raw) - 307.050595577260822     */

/* Batch over a flat pair list (same layout as include/pairhmm_hip.h); n_threads >= 1 plain pthreads split. */
void phmm_oracle_batch(const uint8_t *read_bases, const uint8_t *read_qual, const uint8_t *read_ins,
                       const uint8_t *read_del, const uint8_t *read_gcp, const uint32_t *read_off,
                       const uint32_t *read_len, const uint8_t *hap_bases, const uint32_t *hap_off,
                       const uint32_t *hap_len, const uint32_t *pair_read, const uint32_t *pair_hap, uint64_t n_pairs,
                       float *out_raw, double *out_log10, uint8_t *out_rescued, int n_threads);

/* H3: base-quality modification, in place (pairhmm_internal.cpp:434-439). */
void phmm_oracle_modify_quals(uint8_t *qual, uint32_t n, uint8_t mapping_quality, int32_t min_quality_threshold);

/* H4: PCR indel error model on the gap-open qualities, in place (pairhmm_internal.cpp:65-159, :195-229).
 * model: 0 NONE, 1 HOSTILE, 2 AGGRESSIVE, 3 CONSERVATIVE. */
int32_t phmm_oracle_find_tandem_repeat_units(const uint8_t *bases, uint32_t n, uint32_t offset);
void phmm_oracle_apply_pcr_error_model(const uint8_t *bases, uint8_t *gap_qual, uint32_t n, int model);

/* H8 on a row-major [n_reads][n_haps] matrix (pairhmm_internal.cpp:53-63, :161-182).
 * filter returns the number of surviving reads and writes keep[i] in {0,1}. */
void phmm_oracle_normalize(double *lik, uint32_t n_reads, uint32_t n_haps);
uint32_t phmm_oracle_filter(const double *lik, const uint32_t *read_len, uint32_t n_reads, uint32_t n_haps, uint8_t *keep);
void phmm_oracle_marginalize(const double *old, uint32_t n_old, uint32_t stride, uint32_t evidence_count, const int32_t *old2new,
                             uint32_t n_new, double *out);

/* H1/H2 end to end for one region: reads x haps -> [hap][read] matrix exactly as call_old_pairhmm /
 * call_new_pairhmm002 return it (row width = n_reads, valid prefix = survivors, tail NaN).
 * read_qual are raw BAM quals; gap-opens are 45 + PCR model, gcp 10 (read_record.cpp:184-210).
 * Returns the number of surviving reads; survivors[k] = original index of the k-th survivor. */
uint32_t phmm_oracle_call_region(const uint8_t *read_bases, const uint8_t *read_qual, const uint32_t *read_off,
                                 const uint32_t *read_len, const uint8_t *read_mq, uint32_t n_reads,
                                 const uint8_t *hap_bases, const uint32_t *hap_off, const uint32_t *hap_len,
                                 uint32_t n_haps, int32_t min_quality_threshold, int pcr_model, double *out_hap_read,
                                 uint32_t *survivors);

#ifdef __cplusplus
}
#endif
#endif
