// pairhmm_tables.h -- host lookup tables of the PairHMM engine (see pairhmm_tables.cpp).
#ifndef ROVACA_AMD_PAIRHMM_TABLES_H_
#define ROVACA_AMD_PAIRHMM_TABLES_H_
#include <vector>

namespace phmm
{

constexpr int kMmTableSize = ((254 + 1) * (254 + 2)) >> 1;  // core/context.h:26

struct HostTables
{
    std::vector<float> ph2pr_f32, mm_f32;
    std::vector<double> ph2pr_f64, mm_f64;
};

const HostTables &host_tables();

}  // namespace phmm
#endif
