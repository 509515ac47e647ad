// Compile-time check of the PairHMM seam mirror against the reference's own headers (tests/test_capi_cpu.py builds this
// twice: with the mirror alone, and with -DROVACA_AMD_WITH_REFERENCE_ENUM_AND_STRUCT -I<reference>/src/haplotypecaller, which
// swaps in common/enum.h and genotype/genotype_struct.h -- the two headers of the seam that need neither boost nor htslib).
#include <cstddef>
#include <cstdio>
#include <type_traits>
#include <typeinfo>

#include "pairhmm_gpu_engine.h"

// common/enum.h:8 -- global namespace, uint8_t, these four values in this order
static_assert(std::is_same<std::underlying_type_t<::PcrIndelModel>, uint8_t>::value, "PcrIndelModel underlying type");
static_assert(static_cast<int>(::PcrIndelModel::NONE) == 0 && static_cast<int>(::PcrIndelModel::HOSTILE) == 1 &&
                  static_cast<int>(::PcrIndelModel::AGGRESSIVE) == 2 && static_cast<int>(::PcrIndelModel::CONSERVATIVE) == 3,
              "PcrIndelModel values (the C ABI's phmm_read_prep.pcr_model uses the same numbers)");
// genotype/genotype_struct.h:11-15
static_assert(sizeof(rovaca::Bases) == 4 && offsetof(rovaca::Bases, data) == 4 && offsetof(rovaca::Bases, num) == 0, "Bases layout");
static_assert(std::is_same<decltype(rovaca::Bases::num), uint32_t>::value, "Bases::num");
// pairhmm/pairhmm_engine.h:14-15 -- the candidate has the type of the seam pointer
static_assert(std::is_same<decltype(&rovaca::call_gpu_pairhmm), decltype(rovaca::call_pairhmm)>::value, "seam signature");
static_assert(std::is_same<rovaca::pMemoryPool, std::pmr::memory_resource *>::value, "pMemoryPool (genotype/forward.h:21)");
static_assert(std::is_same<rovaca::DoubleVector2D, std::pmr::vector<std::pmr::vector<double>>>::value, "DoubleVector2D (forward.h:86)");
static_assert(std::is_same<rovaca::ReadVector, std::pmr::vector<rovaca::ReadRecord *>>::value, "ReadVector (forward.h:78)");
static_assert(std::is_same<rovaca::HaplotypeVector, std::pmr::vector<rovaca::Haplotype *>>::value, "HaplotypeVector (forward.h:85)");

int main()
{
    // the mangled type of the seam function: must read the same in both builds
    printf("%s\n", typeid(&rovaca::call_gpu_pairhmm).name());
    return 0;
}
