// pairhmm_marginal.hip -- allele marginalisation on the device, a translation unit of its own so that the
// sweep kernels' object code does not move when this file changes.
//
// genotype/allele_likelihoods.hpp:702-732 (marginal_likelihoods_direct), one sample: out[a'][r] = max over the old alleles a
// with old2new[a] == a' of old[a][r]; -inf where nothing maps to a'; s_missing_index (-1) entries skipped; comparison strict,
// old alleles in ascending order (so a NaN is never taken and the first of two equal values stays).  One thread per
// evidence column: it walks the old alleles in the reference's order and owns column r of every new row, so there is no race
// and no reordering.  HBM-bound by construction: 8 (n_old + n_new) bytes per column, reads coalesced along r.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pairhmm_device.h"

namespace phmm
{

__global__ void __launch_bounds__(256) marginalize_kernel(const double *__restrict__ old, uint32_t n_old, uint32_t stride,
                                                          uint32_t evidence_count, const int32_t *__restrict__ old2new,
                                                          uint32_t n_new, double *__restrict__ out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= evidence_count) return;
    for (uint32_t a = 0; a < n_new; ++a) out[(size_t)a * evidence_count + r] = -__builtin_inf();
    for (uint32_t a = 0; a < n_old; ++a) {
        const int32_t target = old2new[a];  // wave-uniform
        if (target == -1) continue;
        const double v = old[(size_t)a * stride + r];
        double *dst = out + (size_t)target * evidence_count + r;
        if (v > *dst) *dst = v;
    }
}

hipError_t launch_marginalize(const double *old, uint32_t n_old, uint32_t stride, uint32_t evidence_count, const int32_t *old2new,
                              uint32_t n_new, double *out, hipStream_t s)
{
    if (evidence_count == 0 || n_new == 0) return hipSuccess;
    hipLaunchKernelGGL(marginalize_kernel, dim3((evidence_count + 255) / 256), dim3(256), 0, s, old, n_old, stride, evidence_count,
                       old2new, n_new, out);
    return hipGetLastError();
}

}  // namespace phmm
