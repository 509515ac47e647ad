/*
 * pairhmm_hip.h -- C ABI of the MI355X PairHMM likelihood engine (librovaca_pairhmm_hip.so).
 *
 * This is the drop-in boundary for step 3 of Rovaca's HaplotypeCaller.  The reference has no FFI on this path: its seam
 * is a process-global C++ function pointer plus two per-pair kernel pointers
 *     rovaca::call_pairhmm                              src/haplotypecaller/pairhmm/pairhmm_engine.h:14-15
 *     s_compute_full_prob_float_old_pairhmm  (float)    src/haplotypecaller/pairhmm/pairhmm_engine.cpp:8
 *     s_compute_full_prob_double_old_pairhmm (double)   src/haplotypecaller/pairhmm/pairhmm_engine.cpp:9
 * bound once in init_pairhmm_ptr (pairhmm_engine.cpp:21-53).  The entry points below are what a GPU candidate for that
 * seam binds: POD only, caller-owned buffers, int status, no C++ types, no exceptions.  The C++ adapter with the exact
 * signature of rovaca::call_pairhmm lives in rovaca_amd/host/ (see INTEGRATION.md for the reference-side patch).
 *
 * Per-pair semantics are those of the reference's ::TestCase (core/common.h:13-24): read bases and haplotype bases as
 * ASCII, base quality / insertion gap-open / deletion gap-open / gap-continuation as raw phred bytes (not +33), every
 * quality masked & 127 (core/avx512_s.cpp:90-93).  Symbols outside ACGTN never match (pairhmm_internal.cpp:27-41).
 */
#ifndef ROVACA_AMD_PAIRHMM_HIP_H_
#define ROVACA_AMD_PAIRHMM_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHMM_ABI_VERSION 5

/* status codes (0 = ok).  Nothing is thrown across this boundary. */
enum {
    PHMM_OK = 0,
    PHMM_E_INVALID = 1,    /* bad argument: null pointer, zero-length read/haplotype, index out of range, read too long */
    PHMM_E_NO_DEVICE = 2,  /* no usable HIP device: a device context has NO CPU fallback (the CPU backend is asked for by name) */
    PHMM_E_HIP = 3,        /* a HIP runtime call failed; see phmm_last_error() */
    PHMM_E_NOMEM = 4
};

/* result acceptance threshold of the fp32 pass and the two scale constants
 * (rovaca/common.h:9, adapter.h:25-26) */
#define PHMM_MIN_ACCEPTED_F32 1e-28f
#define PHMM_LOG10_INITIAL_CONSTANT_F32 36.1236000061f
#define PHMM_LOG10_INITIAL_CONSTANT_F64 307.050595577260822

typedef struct phmm_context phmm_context; /* device, stream, lookup tables, staging pool */
typedef struct phmm_resident phmm_resident; /* a batch uploaded to HBM, length-binned, ready to sweep */

/* Optional device-side read preparation: what decompression_data does on the host per read
 * (pairhmm_internal.cpp:413-448) -- base-quality modification and the PCR indel model -- done by a GPU kernel instead.
 * When phmm_batch.prep is non-NULL, read_qual holds the RAW base qualities (ReadRecord::qual()), and read_ins / read_del /
 * read_gcp are ignored (may be NULL): the engine derives
 *     qual = min(qual, mapq); qual < min_quality_threshold -> 6                 (pairhmm_internal.cpp:434-439, adapter.h:5)
 *     ins = del = gap_open capped by the PCR table at the tandem-repeat length    (pairhmm_internal.cpp:65-159, adapter.h:13-15)
 *     gcp = gap_cont                                                             (read_record.cpp:205-210) */
typedef struct phmm_read_prep {
    const uint8_t *read_mapq;      /* [n_reads] mapping quality */
    int32_t min_quality_threshold; /* --base-quality-score-threshold, default 18 */
    int32_t pcr_model;             /* 0 NONE, 1 HOSTILE, 2 AGGRESSIVE, 3 CONSERVATIVE (common/enum.h:8) */
    uint8_t gap_open;              /* 45, read_record.cpp:191-196 */
    uint8_t gap_cont;              /* 10, read_record.cpp:205-210 */
} phmm_read_prep;

/* A batch of N (read, haplotype) pairs over a read table and a haplotype table.
 * Replaces the per-pair ::TestCase the reference builds at pairhmm_internal.cpp:569-576 / :675-682. */
typedef struct phmm_batch {
    uint32_t n_reads;
    uint32_t n_haps;
    uint64_t n_pairs;
    /* read table: five concatenated byte arrays; read k is bytes [read_off[k], read_off[k] + read_len[k]) of each */
    const uint8_t *read_bases; /* ::TestCase.rs  */
    const uint8_t *read_qual;  /* ::TestCase.q   */
    const uint8_t *read_ins;   /* ::TestCase.i   */
    const uint8_t *read_del;   /* ::TestCase.d   (may alias read_ins, as the reference does: pairhmm_internal.cpp:425) */
    const uint8_t *read_gcp;   /* ::TestCase.c   */
    const uint32_t *read_off;
    const uint32_t *read_len;
    /* haplotype table */
    const uint8_t *hap_bases;  /* ::TestCase.hap */
    const uint32_t *hap_off;
    const uint32_t *hap_len;
    /* pair list; both NULL means the full cross product in (read, hap) order: pair p = (p / n_haps, p % n_haps) */
    const uint32_t *pair_read;
    const uint32_t *pair_hap;
    /* NULL: the read table above is already prepared (::TestCase semantics).  Non-NULL: see phmm_read_prep. */
    const phmm_read_prep *prep;
} phmm_batch;

typedef struct phmm_stats {
    uint64_t cells;          /* sum over pairs of read_len * hap_len */
    uint64_t n_rescued;      /* pairs re-run in fp64 (fp32 sum < 1e-28f) */
    double kernel_ms;        /* device time of the sweep (fp32 pass + rescue), HIP events on the context's stream */
    double total_ms;         /* host wall time of the call, uploads and downloads included */
    uint32_t n_launches;     /* fp32 sweep launches (one per rows-per-lane bin) */
    uint32_t max_rows_per_lane;
} phmm_stats;

/* ---- life cycle (replaces the kernel binding half of init_pairhmm_ptr, pairhmm_engine.cpp:43-52) ---- */
int phmm_abi_version(void);
/* Which device code this library was linked from: a hash of the sweep kernels' sources and compile flags (rovaca_amd/csrc/Makefile).
   Measurement hygiene only -- profiles/ records it next to the PMC counters, bench.py compares. */
const char *phmm_kernel_build_id(void);
int phmm_device_count(void);
/* device_id >= 0.  stream: a hipStream_t to run on, or NULL to create a private one.  One context drives ONE device (its
 * stream set, staging and tables live there); for several devices see phmm_create_multi below. */
int phmm_create(int device_id, void *stream, phmm_context **out);
/* BASELINE.json config 1 / SURVEY.md 8b ("same entry point must have a CPU backend"): a context that computes on the host --
 * this repository's own scalar and AVX2 + FMA inter-pair kernels (rovaca_amd/csrc/pairhmm_cpu.cpp), the same numerics contract
 * and the same bits as the device path.  n_threads 0 = all hardware threads; allow_simd 0 forces the scalar kernel.  Every
 * entry point that takes a phmm_context works on it (a resident batch is then a host copy).  It is a backend one asks for by
 * name: a device context (phmm_create) never falls back to it. */
int phmm_create_cpu(int n_threads, int allow_simd, phmm_context **out);
const char *phmm_backend_name(const phmm_context *ctx); /* "hip gfx950", "avx2+fma inter-pair (...)", "scalar" */
void phmm_destroy(phmm_context *ctx);
const char *phmm_last_error(void); /* thread-local text of the last failure */

/* ---- one-shot: host buffers in, host buffers out.
 * out_log10[p] = raw >= 1e-28f ? (double)(log10f(raw) - 36.1236000061f) : log10(raw64) - 307.050595577260822
 * (pairhmm_internal.cpp:577-585); out_raw_f32 (nullable) receives the fp32 sums: for a pair decided by fp32 (>= 1e-28f) the sum the
 * reference's fp32 kernel gives, bit for bit; for a pair decided by fp64 a value below 1e-28f -- that same fp32 sum, or 0 where the
 * library established the underflow from the fp64 sum without running the fp32 kernel (long reads that keep underflowing are swept
 * fp64 first; the reference discards the fp32 sum of such a pair, :579-583).
 * Re-entrant per context; use one context per calling thread (the reference calls this path from up to 128 workers,
 * haplotypecaller_engine.cpp:119). */
int phmm_compute(phmm_context *ctx, const phmm_batch *batch, float *out_raw_f32, double *out_log10, phmm_stats *stats);

/* ---- resident batches: upload once, sweep many times (the benchmark's timed region; cross-region batching) ---- */
int phmm_upload(phmm_context *ctx, const phmm_batch *batch, phmm_resident **out);
/* enqueue one full pass (fp32 sweep, rescue collection, fp64 rescue) on the context's stream; asynchronous except for
 * one 4-byte read-back of the rescue count */
int phmm_sweep(phmm_context *ctx, phmm_resident *rb);
/* run `iters` passes bracketed by HIP events on the context's stream; *ms_per_pass is the mean device time of a pass,
 * *fp32_ms_per_pass (nullable) the mean time of the fp32 systolic kernels alone */
int phmm_sweep_timed(phmm_context *ctx, phmm_resident *rb, int iters, double *ms_per_pass, double *fp32_ms_per_pass);
/* copy results of the last pass back and finalize on the host (log10f/log10 in glibc, as the reference does) */
int phmm_download(phmm_context *ctx, phmm_resident *rb, float *out_raw_f32, double *out_log10, phmm_stats *stats);
/* diagnostic: the prepared read table as the kernels see it (after phmm_read_prep, if any); any pointer may be NULL */
int phmm_download_read_table(phmm_context *ctx, phmm_resident *rb, uint8_t *qual, uint8_t *ins, uint8_t *del, uint8_t *gcp);
int phmm_synchronize(phmm_context *ctx);
void phmm_release(phmm_context *ctx, phmm_resident *rb);

/* ---- region-descriptor batches (ABI 5) ----
 * What the reference's call site hands the seam is not a pair list but REGIONS: all reads of an active region against all its
 * haplotypes (haplotypecaller_engine.cpp:250 -> pairhmm_internal.cpp:657-669 loops haplotypes inside reads).  A region is
 * described by four numbers; the engine bins per read (O(reads) host work) and spells the cross products out on the device.
 * Pair ids count region by region, [read][hap] inside one: region g's results start at sum_{g' < g} n_reads * n_haps.
 * Regions list their reads in ascending, non-overlapping runs of the read table; haplotypes may be shared between regions. */
typedef struct phmm_region {
    uint32_t read_first, n_reads; /* reads [read_first, read_first + n_reads) of the read table */
    uint32_t hap_first, n_haps;   /* haplotypes [hap_first, hap_first + n_haps) of the haplotype table */
} phmm_region;
/* one-shot, host buffers in and out: `tables` is a phmm_batch whose pair_read / pair_hap / n_pairs are ignored */
int phmm_compute_regions(phmm_context *ctx, const phmm_batch *tables, const phmm_region *regions, uint32_t n_regions,
                         float *out_raw_f32, double *out_log10, phmm_stats *stats);

/* ---- pinned staging for region batches: the caller writes its tables where the device reads them ----
 * The cross-region batcher of the drop-in adapter (rovaca_amd/host/pairhmm_gpu_engine.cpp) merges the regions of many worker
 * threads into one submission.  With host buffers in and out that costs four copies of every byte (worker -> ticket ->
 * dispatcher -> pinned stage -> HBM) and a pair list spelled out on the host.  A phmm_stage is a block of pinned host memory
 * owned by the engine and laid out as the device wants it: workers write their reads' RAW base qualities
 * (ReadRecord::qual()), bases and mapping qualities and their haplotypes straight into it at ranges the caller hands out (any
 * number of threads, disjoint ranges); one submission is ONE host-to-device copy of the used part of the stage and ONE set-up
 * launch -- quality modification and PCR indel model (pairhmm_internal.cpp:413-448, :65-159), haplotype bit planes and the
 * pair list happen there -- followed by the sweep and the fp64 re-run; the kernels write the sums into the stage's pinned
 * result arrays, where every worker converts its own region (phmm_stage_log10).  No CPU fallback: on a CPU context (phmm_create_cpu) the same calls compute on the host. */
typedef struct phmm_stage phmm_stage;
typedef struct phmm_stage_caps {
    uint32_t max_regions, max_reads, max_haps;
    uint32_t max_bytes; /* the byte arena: read bases + read qualities + haplotype bases of one submission, gaps included */
    uint32_t max_pairs;
} phmm_stage_caps;
typedef struct phmm_stage_view { /* pinned host memory, valid until phmm_stage_destroy; written by the caller between submissions */
    uint8_t *bytes;                /* [max_bytes] ONE arena for all sequence bytes, so that a submission is one contiguous copy */
    uint32_t *read_off;            /* [max_reads] read k's bases are bytes[read_off[k] .. +read_len[k]) */
    uint32_t *read_qual_off;       /* [max_reads] ... its RAW base qualities bytes[read_qual_off[k] .. +read_len[k]) */
    uint32_t *read_len;            /* [max_reads] */
    uint8_t *read_mapq;            /* [max_reads] */
    uint32_t *hap_off, *hap_len;   /* [max_haps] haplotype h is bytes[hap_off[h] .. +hap_len[h]) */
    phmm_region *regions;          /* [max_regions] */
} phmm_stage_view;
typedef struct phmm_stage_params { /* the constants of phmm_read_prep (no per-read array: the mapping qualities are in the stage) */
    int32_t min_quality_threshold, pcr_model;
    uint8_t gap_open, gap_cont;
} phmm_stage_params;
int phmm_stage_create(phmm_context *ctx, const phmm_stage_caps *caps, phmm_stage **out);
void phmm_stage_destroy(phmm_context *ctx, phmm_stage *st);
int phmm_stage_get_view(const phmm_stage *st, phmm_stage_view *view, phmm_stage_caps *caps /* nullable */);
/* asynchronous: validates and bins the first n_regions / n_reads / n_haps entries of the stage and enqueues set-up and the fp32 sweep
 * on the context's stream.  The stage must not be written until phmm_stage_wait has returned.
 * phmm_stage_wait waits for that, and -- only when a pair underflowed fp32 (round 4) -- launches the fp64 re-run of those pairs on `ctx`
 * and waits for it: it is a call that launches work, not a pure wait.  So: `ctx` must be the context the stage was submitted on (the
 * re-run uses that context's tables and streams), and calls that launch work on ONE context -- phmm_stage_submit, phmm_stage_wait,
 * phmm_sweep, phmm_compute* -- must be serialised by the caller (one thread per context, as the in-tree adapter's dispatchers do;
 * different contexts are independent). */
int phmm_stage_submit(phmm_context *ctx, phmm_stage *st, uint32_t n_regions, uint32_t n_reads, uint32_t n_haps,
                      const phmm_stage_params *params);
int phmm_stage_wait(phmm_context *ctx, phmm_stage *st, phmm_stats *stats /* nullable; n_rescued is not counted here */);
/* region g's first pair id in the last submission (the prefix sum above); n_regions gives the total */
uint64_t phmm_stage_pair_first(const phmm_stage *st, uint32_t region);
/* out[i] = log10 likelihood of pair first_pair + i (the conversion of phmm_compute, glibc log10f / log10); any thread, after
 * phmm_stage_wait; *n_rescued (nullable) = how many of them took the fp64 value */
int phmm_stage_log10(const phmm_stage *st, uint64_t first_pair, uint64_t n, double *out, float *out_raw_f32 /* nullable */,
                     uint64_t *n_rescued);

/* the binning alone (host only, no device; tests): launches in processing order.  bin_rows[i] = rows per lane of launch i (0 =
 * striped kernel), bin_first / bin_count = its range of `order` (nullable arrays of capacity max_bins); returns the number of
 * launches, or a negative status */
int phmm_plan_bins(const phmm_batch *batch, uint32_t *order, int32_t *bin_rows, uint32_t *bin_first, uint32_t *bin_count, int max_bins);
int phmm_plan_region_bins(const phmm_batch *tables, const phmm_region *regions, uint32_t n_regions, uint32_t *slot_first,
                          uint32_t *pair_first, int32_t *bin_rows, uint32_t *bin_first, uint32_t *bin_count, int max_bins);

/* ---- several devices behind one handle ("create(device_ids[], n)"; 8e: plain split, no collective) ----
 * The reference's call site hands one region at a time to the seam from up to 128 workers (haplotypecaller_engine.cpp:119);
 * for batch-scale work (BASELINE config 4: ~1e8 pairs over 8 GPUs) the batch is cut instead: pairs sorted by (rows per lane,
 * haplotype length, index) and divided into n_devices contiguous slices of equal DP cells; every device works its slice on
 * its own host thread, with its own context (stream set, pinned staging, tables), and writes results at the pairs' own
 * positions of the caller's arrays.  A phmm_context stays one-device by contract (phmm_create); this handle owns one per
 * listed device.  device_ids must be distinct -- unless PHMM_MULTI_ALLOW_SHARED_DEVICE is set in the environment (a test and
 * rehearsal hook: every entry still gets its own context, host thread and slice, so the N-slice path can be run on a box with
 * fewer than N GPUs; bench.py --oversubscribe uses it and keeps reporting the number of PHYSICAL devices).  Every device receives
 * only the reads and haplotypes its slice refers to. */
typedef struct phmm_multi phmm_multi;
typedef struct phmm_multi_resident phmm_multi_resident;
int phmm_create_multi(const int *device_ids, int n_devices, phmm_multi **out);
/* the same handle over n_slices CPU contexts (phmm_create_cpu): the cut, the per-slice threads, the scatter and the timed gate are
 * the code the device slices run -- the N > 1 path on a machine without GPUs.  phmm_multi_device_id reports -1, -2, ... */
int phmm_create_multi_cpu(int n_slices, int n_threads_each, int allow_simd, phmm_multi **out);
void phmm_destroy_multi(phmm_multi *m);
int phmm_multi_device_count(const phmm_multi *m);
int phmm_multi_device_id(const phmm_multi *m, int i); /* HIP device of slice i, -1 when out of range */
/* one-shot, host buffers in and out, same semantics per pair as phmm_compute; stats are summed (kernel_ms: the slowest device) */
int phmm_multi_compute(phmm_multi *m, const phmm_batch *batch, float *out_raw_f32, double *out_log10, phmm_stats *stats);
/* resident form: cut + one upload per device, then passes on all devices */
int phmm_multi_upload(phmm_multi *m, const phmm_batch *batch, phmm_multi_resident **out);
int phmm_multi_sweep(phmm_multi *m, phmm_multi_resident *rb); /* enqueue one pass on every device (asynchronous) */
int phmm_multi_synchronize(phmm_multi *m);
/* `iters` passes on every device at once, one host thread per device; *wall_ms_per_pass = (slowest device's wall time from
 * the common start) / iters; device_ms_per_pass / fp32_ms_per_pass (nullable, [n_devices]): HIP-event time of a pass */
int phmm_multi_sweep_timed(phmm_multi *m, phmm_multi_resident *rb, int iters, double *wall_ms_per_pass, double *device_ms_per_pass,
                           double *fp32_ms_per_pass);
int phmm_multi_download(phmm_multi *m, phmm_multi_resident *rb, float *out_raw_f32, double *out_log10, phmm_stats *stats);
int phmm_multi_shard_cells(const phmm_multi_resident *rb, uint64_t *cells /*[n_devices], nullable*/, uint64_t *pairs /*nullable*/);
void phmm_multi_release(phmm_multi *m, phmm_multi_resident *rb);
/* the cut alone (host only, no device): slice_of_pair[p] (nullable) = the slice pair p falls into, cells[g] (nullable) = its cells */
int phmm_plan_slices(const phmm_batch *batch, int n_slices, uint32_t *slice_of_pair, uint64_t *cells);

/* ---- allele marginalisation, the step right after the seam ----
 * One sample of AlleleLikelihoods::marginal_likelihoods_direct (genotype/allele_likelihoods.hpp:702-732), called from
 * germline_genotying_engine.cpp:143 through marginalize (:188-217): out[a'][r] = max over the old alleles a with
 * old2new[a] == a' of old[a][r], -inf where nothing maps to a'; entries equal to -1 (s_missing_index, :26) are skipped.
 * old: [n_old][stride] host doubles with evidence_count <= stride valid values per row; out: [n_new][evidence_count].
 * Exact (a maximum of doubles).  device_ms (nullable): HIP-event time of the kernel.  The drop-in adapter does this on the
 * host (rovaca::marginal_likelihoods_direct, rovaca_amd/host): a region's matrix is a few kilobytes -- DESIGN.md section 6. */
int phmm_marginalize(phmm_context *ctx, const double *old, uint32_t n_old, uint32_t stride, uint32_t evidence_count,
                     const int32_t *old2new, uint32_t n_new, double *out, double *device_ms);

/* ---- lookup tables the engine built on the host (core/context.h:13-189), for parity checks ---- */
#define PHMM_MM_TABLE_SIZE 32640 /* ((254 + 1) * (254 + 2)) >> 1 */
int phmm_get_tables(float *ph2pr_f32 /*[128]*/, float *mm_f32 /*[32640]*/, double *ph2pr_f64 /*[128]*/,
                    double *mm_f64 /*[32640]*/);

#ifdef __cplusplus
}
#endif
#endif
