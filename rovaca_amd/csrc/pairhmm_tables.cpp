// pairhmm_tables.cpp -- host-side construction of the PairHMM lookup tables (product code, not the oracle).
//
// Follows core/context.h:13-189 of the reference (rovaca/context.h is the same text in a namespace):
//   ph2pr[q]            = 10^(-q/10)                      powf for float (:144-146), pow for double (:102-104)
//   jacobianLogTable[k] = log10(1 + 10^(-k * 1e-4))       computed in double, stored in T (:41-46)
//   matchToMatchProb    = triangular table over (max, min) quality, index (max*(max+1))/2 + min (:49-59, :162-173)
// The tables are built once (std::call_once; the reference's lazy bool flag at core/context.h:99-111 is racy) and uploaded
// to the device by phmm_create.
#include "pairhmm_tables.h"

#include <algorithm>
#include <cmath>
#include <mutex>

namespace phmm
{
namespace
{

constexpr int kMaxQual = 254;
constexpr double kJacTol = 8.0;
constexpr double kJacStep = 0.0001;
constexpr double kJacInvStep = 1.0 / kJacStep;
constexpr int kJacSize = 80001;  // (int)(8.0 / 0.0001) + 1

template <typename T>
struct Builder
{
    std::vector<T> jac;

    static int fast_round(T d) { return (d > T(0)) ? (int)(d + T(0.5)) : (int)(d - T(0.5)); }

    T approx_log10_sum(T small, T big) const
    {
        if (small > big) std::swap(small, big);
        if (std::isinf(small) || std::isinf(big)) return big;
        T diff = big - small;
        if (diff >= T(kJacTol)) return big;
        int ind = fast_round(T(diff * T(kJacInvStep)));
        return big + jac[ind];
    }

    void build(T *ph2pr, T *mm)
    {
        jac.resize(kJacSize);
        for (int k = 0; k < kJacSize; ++k) jac[k] = T(std::log10(1.0 + std::pow(10.0, -double(k) * kJacStep)));
        const double inv_ln10 = 1.0 / std::log(10.0);
        for (int i = 0, offset = 0; i <= kMaxQual; offset += ++i) {
            for (int j = 0; j <= i; ++j) {
                double log10_sum = approx_log10_sum(T(-0.1 * i), T(-0.1 * j));
                double m2m_log10 = std::log1p(-std::min(1.0, std::pow(10.0, log10_sum))) * inv_ln10;
                mm[offset + j] = T(std::pow(10.0, m2m_log10));
            }
        }
        for (int x = 0; x < 128; ++x) {
            if constexpr (sizeof(T) == 4)
                ph2pr[x] = powf(10.f, -float(x) / 10.f);
            else
                ph2pr[x] = std::pow(10.0, -double(x) / 10.0);
        }
    }
};

HostTables g_tables;
std::once_flag g_once;

}  // namespace

const HostTables &host_tables()
{
    std::call_once(g_once, [] {
        g_tables.ph2pr_f32.resize(128);
        g_tables.ph2pr_f64.resize(128);
        g_tables.mm_f32.resize(kMmTableSize);
        g_tables.mm_f64.resize(kMmTableSize);
        Builder<float>().build(g_tables.ph2pr_f32.data(), g_tables.mm_f32.data());
        Builder<double>().build(g_tables.ph2pr_f64.data(), g_tables.mm_f64.data());
    });
    return g_tables;
}

}  // namespace phmm
