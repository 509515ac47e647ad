// pairhmm_cpu.h -- CPU backend of the C ABI (pairhmm_cpu.cpp): BASELINE.json config 1.
#ifndef ROVACA_AMD_PAIRHMM_CPU_H_
#define ROVACA_AMD_PAIRHMM_CPU_H_
#include <stdint.h>

#include "../../include/pairhmm_hip.h"

namespace phmm
{
namespace cpu
{

// Raw fp32 sums of every pair of a PREPARED batch (::TestCase semantics), fp64 sums (scale 2^1020) of the pairs whose fp32 sum
// fell below 1e-28f.  out_raw_f64 / rescued (one byte per pair) / n_rescued / cells may be null.  allow_simd = false forces the
// scalar kernel.  Returns 0, or -1 for a pair that indexes outside the tables or a zero-length sequence.
int compute(const phmm_batch &b, unsigned n_threads, bool allow_simd, float *out_raw_f32, double *out_raw_f64, uint8_t *rescued,
            uint64_t *n_rescued, uint64_t *cells);
const char *kernel_name(bool allow_simd);

}  // namespace cpu
}  // namespace phmm
#endif
