// pairhmm_systolic_tab8.h -- EXPERIMENT (round 5): the table sweep with EIGHT lanes per pair (eight pairs per wave, 32 per 256-thread
// workgroup, about twice the rows per lane).  150-base reads are 8 x 19 = 152 row slots instead of 16 x 10 = 160, the array fills and
// drains in 7 steps instead of 15, and the per-step overhead (cross-lane moves, sums, addresses) is carried by 19 rows instead of 10.
// Priors are single-buffered (fetched for the next step right after pass 1 has used them) so that 19 rows fit 128 registers.
#ifndef ROVACA_AMD_PAIRHMM_SYSTOLIC_TAB8_H_
#define ROVACA_AMD_PAIRHMM_SYSTOLIC_TAB8_H_
#include "pairhmm_systolic_tab.h"

namespace phmm
{

__host__ __device__ inline uint32_t tab8_table_bytes(int K) { return (uint32_t)((K + 3) / 4) * 512u; }
__host__ __device__ inline uint32_t tab8_lds_bytes(int K, uint32_t slots, uint32_t max_hap)
{
    return slots * tab8_table_bytes(K) + 32u * tab_hap_bytes(max_hap) + 128u;
}

// lane i of each 8-lane half row receives lane i-1; the first lane of a half row takes `first`
__device__ __forceinline__ float dpp_half_row_shr1(float first, float src, bool is_first)
{
    // (old = src: the lanes without a source are overwritten below anyway, and no zero has to be materialised)
    const float v = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(src), __float_as_int(src), 0x111, 0xF, 0xF, false));
    return is_first ? first : v;
}

// UGO: one gap-open quality in the batch (pairhmm_systolic_ugo_sweep.inc); otherwise per-base gap-opens with deletion == insertion and
// one gap-continuation quality (pairhmm_systolic_sweep.inc with UG = true: the Rovaca case with a PCR indel model).
// n_slots: tables the workgroup's LDS holds (1..16; the launch sized the LDS with tab_lds_bytes for it).
template <int K, bool UGO>
__device__ __forceinline__ void systolic_tab8_sweep(const KernelArgs &a, uint32_t wg, uint32_t pos_first, uint32_t count, uint32_t *rescue_list,
                                                   uint32_t *rescue_count, uint32_t n_slots)
{
    using T = float;
    constexpr int G = 8;
    constexpr int Q = (K + 3) / 4;
    constexpr int KY = K < G - 1 ? K : G - 1;  // rows that can be boundary replicas (fewer than G rows are, the first ones)
    extern __shared__ __attribute__((aligned(256))) uint8_t lds_raw[];
    const uint32_t lane = threadIdx.x & 7u;
    const uint32_t grp = threadIdx.x >> 3;
    const uint32_t gidx = wg * 32u + grp;
    const bool valid = gidx < count;
    const uint32_t pair = a.order[valid ? pos_first + gidx : 0];
    const uint32_t r = a.pair_read[pair], h = a.pair_hap[pair];
    const uint32_t R = a.read_len[r], H = a.hap_len[h];
    const uint32_t roff = a.read_off[r];
    const T *__restrict__ ph2pr = reinterpret_cast<const T *>(a.ph2pr);
    const T *__restrict__ mmtab = reinterpret_cast<const T *>(a.mm_table);

    // ---- LDS of the workgroup: the tables' quads, their tails (interleaved), the 16 pairs' haplotype entries, the 16 read ids ----
    const uint32_t hap_at = n_slots * tab8_table_bytes(K);
    const uint32_t hap_stride = tab_hap_bytes(a.tab_max_hap);
    uint8_t *hapl = lds_raw + hap_at + grp * hap_stride;
    uint32_t *rid = reinterpret_cast<uint32_t *>(lds_raw + hap_at + 32u * hap_stride);

    // ---- the haplotype: 16-bit entries (symbol * 256), entry 16 + d + (j - 1) for column j, d = the haplotype's offset in its dword.
    // 16 lanes x one aligned dword per round; dwords that do not touch the haplotype are not read (symbol 0).  A base outside
    // A, C, G, T makes the pair `dirty` (bases of the neighbours in the first and last dword count too: a false alarm costs speed only). ----
    const uintptr_t hap_at_mem = reinterpret_cast<uintptr_t>(a.hap_bases) + a.hap_off[h];
    const uint32_t d = static_cast<uint32_t>(hap_at_mem & 3u);
    bool dirty = false;
    {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(hap_at_mem - d);
        reinterpret_cast<uint32_t *>(hapl)[lane] = 0;  // the 16 entries in front
        const uint32_t rounds = 2u * tab_hap_rounds(a.tab_max_hap);
        for (uint32_t it = 0; it < rounds; ++it) {
            const uint32_t i = it * 8u + lane;  // dword of the haplotype's span
            uint32_t w = 0x41414141u;
            if (4u * i + 3u >= d && 4u * i < d + H) w = *reinterpret_cast<const uint32_t *>(src + 4u * i);
            const uint32_t x = (w >> 1) & 0x03030303u;  // A 0, C 1, T 2, G 3 in every byte (ASCII bits 1-2)
            dirty = dirty || __builtin_amdgcn_perm(0u, 0x47544341u, x) != w;  // the four codes back to "ACTG": any other byte differs
            uint2 e;
            e.x = ((x & 0xFFu) << 7) | ((x & 0xFF00u) << 15);
            e.y = (((x >> 16) & 0xFFu) << 7) | (((x >> 16) & 0xFF00u) << 15);
            *reinterpret_cast<uint2 *>(hapl + 2u * (16u + 4u * i)) = e;
        }
    }
    if (lane == 0) rid[grp] = valid ? r : 0xFFFFFFFFu;
    (void)dirty;  // (experiment: clean haplotypes only)
    __syncthreads();
    // ---- which table a pair uses: one per run of equal reads among the workgroup's pairs ----
    uint32_t slot = 0, n_tables = 1;
    bool leader = grp == 0;
    {
        uint32_t prev = rid[0];
#pragma unroll
        for (uint32_t g = 1; g < 32; ++g) {
            const uint32_t cur = rid[g];
            const bool chg = cur != 0xFFFFFFFFu && cur != prev;
            n_tables += chg;
            slot += chg && g <= grp;
            leader = leader || (chg && g == grp);
            prev = cur;
        }
    }

    // ---- constants ----
    const T init_y = Num<T>::init_const() / static_cast<T>(H);  // rovaca/avx_512_float.cc:32
    const int pad = G * K - static_cast<int>(R);                  // rows [0, pad) are boundary replicas
    const uint32_t c0 = a.read_gcp[roff] & 127u;
    const T p_gapm = static_cast<T>(1.0) - ph2pr[c0];
    const T p_xx = ph2pr[c0];
    T u_mm = 0, u_mx = 0;  // UGO: the two factors every row shares
    if constexpr (UGO) {
        const uint32_t g0 = a.read_ins[roff] & 127u;
        u_mm = mmtab[((g0 * (g0 + 1)) >> 1) + g0];  // core/context.h:162-173 with ins == del
        u_mx = ph2pr[g0];                            // == p_my
    }
    T p_yy[KY];
    T r_mm[UGO ? 1 : K], r_mx[UGO ? 1 : K];  // per-base gap-opens: per row
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int row = static_cast<int>(lane) * K + k - pad;
        if (k < KY) p_yy[k < KY ? k : 0] = row >= 0 ? p_xx : static_cast<T>(1);  // a boundary replica holds Y with p_yy = 1
        if constexpr (!UGO) {
            r_mm[k] = 0, r_mx[k] = 0;
            if (row >= 0) {
                const uint32_t g = a.read_ins[roff + static_cast<uint32_t>(row)] & 127u;
                r_mm[k] = mmtab[((g * (g + 1)) >> 1) + g];
                r_mx[k] = ph2pr[g];
            }
        }
    }
    auto f_yy = [&](int k) -> T { return k < KY ? p_yy[k < KY ? k : 0] : p_xx; };
    auto f_mm = [&](int k) -> T { return UGO ? u_mm : r_mm[UGO ? 0 : k]; };
    auto f_mx = [&](int k) -> T { return UGO ? u_mx : r_mx[UGO ? 0 : k]; };

    const T y_col0 = (static_cast<int>(lane) * K <= pad) ? init_y : static_cast<T>(0);
    // the entry of this lane's column at step t is hp[t]: entry 16 + d + (t - lane) - 1
    const uint16_t *hp = reinterpret_cast<const uint16_t *>(hapl) + (15u + d - lane);

#pragma unroll 1
    for (uint32_t base = 0; base < n_tables; base += n_slots) {
        const bool act = valid && slot >= base && slot - base < n_slots;
        const uint32_t sl = act ? slot - base : 0u;
        uint8_t *quads = lds_raw + sl * (Q * 512u) + lane * 16u;
        if (base) __syncthreads();  // the round before has read its tables
        if (act && leader) {
            // the read's table: symbol code s (ASCII bits 1-2: A 0, C 1, T 2, G 3) -> read class (A 0, C 1, G 2, T 3); a read N matches all
            T hi[K], lo[K];
            uint32_t cls[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int row = static_cast<int>(lane) * K + k - pad;
                cls[k] = 5, hi[k] = 0, lo[k] = 0;  // boundary replica: prior 0 kills M
                if (row >= 0) {
                    const uint32_t at = roff + static_cast<uint32_t>(row);
                    const T dist = ph2pr[a.read_qual[at] & 127u];
                    hi[k] = static_cast<T>(1.0) - dist;
                    lo[k] = Num<T>::div3(dist);
                    cls[k] = read_class(a.read_bases[at]);
                }
            }
            auto sel = [&](int k, uint32_t s) -> T {
                if (k >= K) return 0;
                const uint32_t c = s == 2u ? 3u : s == 3u ? 2u : s;
                return (cls[k] == c || cls[k] == 4u) ? hi[k] : lo[k];
            };
#pragma unroll
            for (uint32_t s = 0; s < 4; ++s) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    float4 v;
                    v.x = sel(4 * q, s), v.y = sel(4 * q + 1, s), v.z = sel(4 * q + 2, s), v.w = sel(4 * q + 3, s);
                    *reinterpret_cast<float4 *>(quads + q * 512 + s * 128u) = v;
                }
            }
        }
        __syncthreads();  // tables (and, the first time, the haplotype entries) are read by other lanes than wrote them

        // ---- the sweep ----
        const uint32_t Tg = act ? H + (G - 1) : 0;  // last step of this group
        uint32_t Tmax = Tg;
        Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 8));
        Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 16));
        Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 32));
        Tmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmax);
        uint32_t Tmin = Tg;
        Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 8));
        Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 16));
        Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 32));
        Tmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmin);
        if (Tmax == 0) continue;  // none of the wave's four pairs is in this round

        T M[K], X[K], Y[K];
        T PM[UGO && K > 1 ? K - 1 : 1];  // UGO: M[k] * p_mx of the anti-diagonal before, computed in pass 2 and used again in pass 1
#pragma unroll
        for (int k = 0; k < K; ++k) {
            M[k] = 0;
            X[k] = 0;
            Y[k] = static_cast<int>(lane) * K + k < pad ? init_y : static_cast<T>(0);
        }
        if constexpr (UGO) {
#pragma unroll
            for (int k = 0; k + 1 < K; ++k) PM[k] = 0;
        }
        T aM = 0, aX = 0, aY = y_col0, bM = 0, bX = 0, bY = y_col0;
        T sumM = 0, sumX = 0;
        T pr[K];  // priors of the step in hand; re-fetched for the next step as soon as pass 1 has used them
        auto fetch = [&](uint32_t so) {
            const uint8_t *tq = quads + so;
#pragma unroll
            for (int q = Q - 1; q >= 0; --q) {  // pass 1 walks bottom-up: the last quad is needed first
                const float4 v = *reinterpret_cast<const float4 *>(tq + q * 512);
                pr[4 * q] = v.x;
                if (4 * q + 1 < K) pr[4 * q + 1] = v.y;
                if (4 * q + 2 < K) pr[4 * q + 2] = v.z;
                if (4 * q + 3 < K) pr[4 * q + 3] = v.w;
            }
        };
        // one anti-diagonal with the priors `pr`.  (dM, dX, dY): row above slot 0 at the previous column; (uM0, uX0, uY0): receives it
        // at this column.  The arithmetic is the bit-plane sweeps', statement for statement.
        const bool first_lane = lane == 0;
        auto pass1 = [&](T &dM, T &dX, T &dY, T &uM0, T &uX0, T &uY0) {
            uM0 = dpp_half_row_shr1(static_cast<T>(0), M[K - 1], first_lane);
            uX0 = dpp_half_row_shr1(static_cast<T>(0), X[K - 1], first_lane);
            uY0 = dpp_half_row_shr1(y_col0, Y[K - 1], first_lane);
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                if constexpr (UGO)
                    Y[k] = Num<T>::fma_(Y[k], f_yy(k), k == K - 1 ? M[K - 1] * u_mx : PM[k < K - 1 ? k : 0]);
                else
                    Y[k] = Num<T>::fma_(Y[k], f_yy(k), M[k] * f_mx(k));
                const T gM = k ? M[k ? k - 1 : 0] : dM, gX = k ? X[k ? k - 1 : 0] : dX, gY = k ? Y[k ? k - 1 : 0] : dY;
                T tt = gM * f_mm(k);
                tt = Num<T>::fma_(gX, p_gapm, tt);
                tt = Num<T>::fma_(gY, p_gapm, tt);
                M[k] = tt * pr[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto pass2 = [&](T &uM0, T &uX0) {
            T uM = uM0, uX = uX0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const T prod = uM * f_mx(k);
                X[k] = Num<T>::fma_(uX, p_xx, prod);
                if constexpr (UGO)
                    if (k >= 1) PM[k - 1] = prod;
                uM = M[k];
                uX = X[k];
            }
            __builtin_amdgcn_sched_barrier(0);
            sumM = sumM + M[K - 1];  // rovaca/avx_512_float.cc:166-171 (only lane 15 is read back)
            sumX = sumX + X[K - 1];
        };
        // step t: pass 1 with the priors in hand, then the table reads of step t + 1 into the same registers and the haplotype read of
        // step t + 2, then pass 2 (which needs no prior) while they land
        uint32_t so_next;  // the entry of step t + 1
        auto step = [&](uint32_t t, bool odd, bool live) {
            if (odd) {
                if (live) pass1(aM, aX, aY, bM, bX, bY);
                fetch(so_next);
                so_next = hp[t + 2];
                __builtin_amdgcn_sched_barrier(0);
                if (live) pass2(bM, bX);
            }
            else {
                if (live) pass1(bM, bX, bY, aM, aX, aY);
                fetch(so_next);
                so_next = hp[t + 2];
                __builtin_amdgcn_sched_barrier(0);
                if (live) pass2(aM, aX);
            }
        };
        // Steps go in (odd, even) couples everywhere so that the two register sets and the two prior sets swap roles without a move:
        // the fill (steps 1..16, lane l starts at step l + 1), the steady state (every lane of all four pairs has work, no predicate)
        // and the drain (pairs with a shorter haplotype than the wave's longest drop out; may run one dead step past Tmax).
        uint32_t t = 1;
        fetch(hp[1]);
        so_next = hp[2];
        for (; t <= (uint32_t)G; t += 2) {
            step(t, true, (t > lane) && (t <= Tg));
            step(t + 1, false, (t + 1 > lane) && (t + 1 <= Tg));
        }
        for (; t + 1 <= Tmin; t += 2) {
            step(t, true, true);
            step(t + 1, false, true);
        }
        for (; t <= Tmax; t += 2) {
            step(t, true, t <= Tg);
            step(t + 1, false, t + 1 <= Tg);
        }
        if (act && lane == G - 1) {
            const T res = sumM + sumX;
            reinterpret_cast<T *>(a.out_raw)[pair] = res;
            if (rescue_list && res < static_cast<T>(1e-28f))  // pairhmm_internal.cpp:578: queue for fp64
                rescue_list[atomicAdd(rescue_count, 1u)] = pair;
        }
    }
}

}  // namespace phmm
#endif
