// pairhmm_systolic_tab.h -- the single-pass systolic sweep with the prior looked up in LDS ("table" sweep), gfx950: the sweep of ONE
// 256-thread workgroup (16 pairs, 16 lanes each) as a function template of the rows per lane K, for the one-K kernels of
// pairhmm_kernels_tab.hip and the ragged kernels.
//
// What is computed: exactly what the bit-plane sweeps compute (pairhmm_systolic_sweep.inc / pairhmm_systolic_ugo_sweep.inc: the same
// operations in the same order on the same operands -- rovaca/avx_512_float.cc:29-181 == core/avx512_s.cpp:259-397 with the FMA
// contraction written out), same mapping of a pair to a 16-lane array with K rows per lane, same right-aligned rows, same two in-place
// passes per step.
//
// What differs is where `prior` comes from.  The bit-plane sweeps pop one match bit per cell into VCC and select (1 - e) or e/3 with
// it: two VALU instructions per cell of the 9 (one gap-open quality) or 10 the cell costs, in kernels bound by VALU issue.  Here the
// two go away.  prior[i][j] depends on the read row i (its base and quality) and on the haplotype SYMBOL at column j only, so a READ
// gets a table  prior[row][symbol]  in LDS for the four symbols A, C, G, T and a fifth whose priors are all 0 (below) -- 20 bytes per
// row slot, K x 320 bytes per read -- laid out so that one `ds_read_b128` returns the priors of four consecutive rows of a lane for the
// symbol of its column:
//
//      quads:   slot * Q * 1280 + q * 1280 + symbol * 256 + lane * 16       (rows 4q .. 4q+3 of `lane`; five symbols, below)
//      halves:  symbol * 256 + (slot & 1) * 128 + lane * 8  in block slot/2  (K mod 4 == 2: the last two rows; two tables interleave)
//      singles: symbol * 256 + (slot & 3) * 64 + lane * 4   in block slot/4  (K mod 4 == 1: the last row; four tables interleave)
//
// The symbol stride is 256 bytes = the width of the LDS (64 banks x 4 bytes), so whatever symbols the 16 lanes of a service group
// (MI355X_MICROARCH.md, LDS: `ds_read_b128` is served 16 lanes at a time, and every group holds 16 DIFFERENT lane numbers mod 16) look
// at, lane l lands on bank group l: conflict-free by construction, also when two pairs of a wave read the same table.
//
// The pairs of a workgroup that have the same read share its table: the 16 pairs of a workgroup are consecutive in the processing
// order, and the order keeps the pairs of a read together (pairhmm_binning.cpp), so a workgroup needs as many tables as it has reads --
// two or three for regions of 9-25 haplotypes -- instead of 16.  That is what makes the table affordable: LDS for `tab_slots` tables
// (the launch says how many fit next to the haplotypes at four waves per SIMD) instead of 16.  A workgroup with more reads than slots
// sweeps in rounds: the pairs of the first tab_slots reads, then the next ones (correct for any order, slow only for pair lists
// that jump between reads at every pair).
//
// The haplotype travels as one 16-bit entry per column (symbol * 256, ready to add) in LDS as well; a lane reads the entry of ITS
// column each step (`ds_read_u16`), adds its table base (one or two `v_add_u32`) and issues the K / 4 table reads of the NEXT step
// right behind pass 1 -- the only user of the priors -- into the same registers, so that pass 2's 2 K instructions cover the LDS
// latency and 16 rows per lane fit 128 registers (every class runs at four waves per SIMD).  LDS instructions do not take VALU issue
// slots: per step and K = 10 the wave issues 71 arithmetic + 3 DPP + 2 sums + 2-3 address instructions instead of 101.
//
// No predicates: every haplotype entry that is NOT a column of the haplotype names the fifth symbol.  A lane that computes such a
// column leaves the column-0 state of its rows as it is and adds 0 to both sums (see the staging code), so the sweep is one loop from
// step 1 to the wave's last step -- no fill, no drain.
//
// Haplotypes with symbols outside A, C, G, T (N matches everything, anything else nothing: pairhmm_internal.cpp:27-41) are not
// representable in a four-symbol table.  The staging loop notices them (one `v_perm_b32` per four bases), and a workgroup that holds
// such a pair takes the bit-plane sweep instead -- same bits, the speed of the kernels of rounds 1-4.
#ifndef ROVACA_AMD_PAIRHMM_SYSTOLIC_TAB_H_
#define ROVACA_AMD_PAIRHMM_SYSTOLIC_TAB_H_
#include "pairhmm_systolic.h"

namespace phmm
{

template <int K> struct TabShape
{
    static constexpr int rem = K % 4;
    static constexpr int quads = rem == 3 ? (K + 3) / 4 : K / 4;  // three odd rows: a padded quad
    static constexpr int tail = rem == 3 ? 0 : rem;               // rows behind the quads: 0, 1 (b32) or 2 (b64)
};

// entries of the haplotype array of one pair: 16 in front (the columns a lane "sees" before its first step), then whole rounds of 64
// (16 lanes x 4 bases) that cover the haplotype, its offset in its first dword, and the 15 + 2 + 1 columns the sweep looks past its end
__host__ __device__ inline uint32_t tab_hap_rounds(uint32_t max_hap) { return (max_hap + 21u + 63u) / 64u; }
__host__ __device__ inline uint32_t tab_hap_bytes(uint32_t max_hap) { return 2u * (16u + 64u * tab_hap_rounds(max_hap)); }
// LDS of a workgroup: `slots` tables of K rows and FIVE symbols (kTabSymStride bytes apart; the fifth is all zeros, below), the 16
// pairs' haplotype entries, the 16 read ids
constexpr uint32_t kTabSyms = 5, kTabSymStride = 256, kTabQuadStride = kTabSyms * kTabSymStride;
__host__ __device__ inline uint32_t tab_table_region_bytes(int K, uint32_t slots)
{
    const int rem = K % 4, quads = rem == 3 ? (K + 3) / 4 : K / 4, tail = rem == 3 ? 0 : rem;
    return slots * quads * kTabQuadStride + (tail == 2 ? (slots + 1) / 2 * kTabQuadStride : tail == 1 ? (slots + 3) / 4 * kTabQuadStride : 0u);
}
__host__ __device__ inline uint32_t tab_lds_bytes(int K, uint32_t slots, uint32_t max_hap)
{
    return tab_table_region_bytes(K, slots) + 16u * tab_hap_bytes(max_hap) + 64u;
}

// ---- one class, one build ----
// At equal instruction counts the sweep's rate moves by up to 15 % with the order hipcc schedules the instructions in and the registers
// it picks ("Register numbers matter"), differently for every rows-per-lane class.  So the kernel of a class that is
// launched on its own (pairhmm_kernels_tab.hip) is compiled with the instruction scheduler (-mllvm -amdgpu-sched-strategy, three builds
// of that file: csrc/pairhmm_tab_sched.mk) and in the form of the rounds loop (below) that measured best for THAT class:
// tools/sweep_matrix.sh, every class x form x scheduler on one box, profiles/r5_sweep_matrix.txt; a class moves only for 3 % or more.
// The ragged kernels (every class in one kernel, one scheduler for all) keep the default scheduler and tab_one_loop_default.
enum : int { kTabSchedDefault = 0, kTabSchedMaxIlp = 1, kTabSchedIterIlp = 2 };
struct TabPlan
{
    int sched;
    bool one_loop;
};
__host__ __device__ constexpr bool tab_one_loop_default(int K, bool ugo) { return ugo && (K == 5 || K == 6 || K == 9 || K == 10 || K == 11); }
__host__ __device__ constexpr TabPlan tab_plan(int K, bool ugo)
{
#ifdef PHMM_TAB_ONE_BUILD  // tuning builds of the one file (tools/build_tab_variant.sh): every class in it
    return TabPlan{kTabSchedDefault, tab_one_loop_default(K, ugo)};
#else
    if (ugo) {
        switch (K) {
            case 1: return TabPlan{kTabSchedIterIlp, true};    // 3 099 -> 3 551 GCUPS at full slots
            case 2: return TabPlan{kTabSchedMaxIlp, true};     // 4 656 -> 4 978
            case 3: return TabPlan{kTabSchedMaxIlp, false};    // 5 435 -> 5 614
            case 4: return TabPlan{kTabSchedIterIlp, false};   // 5 693 -> 6 213
            case 7: return TabPlan{kTabSchedMaxIlp, false};    // 6 766 -> 7 200
            case 8: return TabPlan{kTabSchedMaxIlp, true};     // 7 146 -> 7 409
            case 9: return TabPlan{kTabSchedMaxIlp, true};     // 6 813 -> 7 289
            case 10: return TabPlan{kTabSchedIterIlp, false};  // 7 460 -> 7 698 (150-base reads: 6 930 -> 7 210)
            case 13: return TabPlan{kTabSchedDefault, true};   // 7 236 -> 7 520
            case 14: return TabPlan{kTabSchedDefault, true};   // 7 191 -> 7 548
            case 15: return TabPlan{kTabSchedMaxIlp, false};   // 7 562 -> 7 921
            case 16: return TabPlan{kTabSchedDefault, true};   // 7 492 -> 7 764
            default: break;
        }
    }
    else {
        switch (K) {
            case 1: return TabPlan{kTabSchedIterIlp, true};    // 3 171 -> 3 396
            case 2: return TabPlan{kTabSchedIterIlp, true};    // 4 545 -> 4 855
            case 3: return TabPlan{kTabSchedIterIlp, false};   // 5 056 -> 5 298
            case 5: return TabPlan{kTabSchedIterIlp, true};    // 5 602 -> 6 007
            case 6: return TabPlan{kTabSchedMaxIlp, true};     // 5 712 -> 6 046
            case 7: return TabPlan{kTabSchedIterIlp, true};    // 6 205 -> 6 578
            case 8: return TabPlan{kTabSchedIterIlp, true};    // 6 204 -> 6 775
            case 9: return TabPlan{kTabSchedMaxIlp, false};    // 6 240 -> 6 599
            case 13: return TabPlan{kTabSchedDefault, true};   // 6 540 -> 6 880
            case 15: return TabPlan{kTabSchedIterIlp, true};   // 6 756 -> 6 998
            default: break;
        }
    }
    return TabPlan{kTabSchedDefault, tab_one_loop_default(K, ugo)};
#endif
}

// UGO: one gap-open quality in the batch (pairhmm_systolic_ugo_sweep.inc); otherwise per-base gap-opens with deletion == insertion and
// one gap-continuation quality (pairhmm_systolic_sweep.inc with UG = true: the Rovaca case with a PCR indel model).
// ONE_LOOP: the form of the rounds loop (at the end of the function).
// n_slots: tables the workgroup's LDS holds (1..16; the launch sized the LDS with tab_lds_bytes for it).
template <int K, bool UGO, bool ONE_LOOP = tab_one_loop_default(K, UGO)>
__device__ __forceinline__ void systolic_tab_sweep(const KernelArgs &a, uint32_t wg, uint32_t pos_first, uint32_t count, uint32_t *rescue_list,
                                                   uint32_t *rescue_count, uint32_t n_slots)
{
    using T = float;
    constexpr int G = 16;
    constexpr int Q = TabShape<K>::quads, TAIL = TabShape<K>::tail;
    extern __shared__ __attribute__((aligned(256))) uint8_t lds_raw[];
    const uint32_t lane = threadIdx.x & 15u;
    const uint32_t grp = threadIdx.x >> 4;
    const uint32_t gidx = wg * 16u + grp;
    const bool valid = gidx < count;
    const uint32_t pair = a.order[valid ? pos_first + gidx : 0];
    const uint32_t r = a.pair_read[pair], h = a.pair_hap[pair];
    const uint32_t R = a.read_len[r], H = a.hap_len[h];
    const uint32_t roff = a.read_off[r];
    const T *__restrict__ ph2pr = reinterpret_cast<const T *>(a.ph2pr);
    const T *__restrict__ mmtab = reinterpret_cast<const T *>(a.mm_table);

    // ---- LDS of the workgroup: the tables' quads, their tails (interleaved), the 16 pairs' haplotype entries, the 16 read ids ----
    const uint32_t tail_at = n_slots * (Q * kTabQuadStride);
    const uint32_t hap_at = tab_table_region_bytes(K, n_slots);
    const uint32_t hap_stride = tab_hap_bytes(a.tab_max_hap);
    uint8_t *hapl = lds_raw + hap_at + grp * hap_stride;
    uint32_t *rid = reinterpret_cast<uint32_t *>(lds_raw + hap_at + 16u * hap_stride);

    // ---- the haplotype: 16-bit entries (symbol * 256), entry 16 + d + (j - 1) for column j, d = the haplotype's offset in its dword.
    // 16 lanes x one aligned dword per round.  Every entry that is NOT a column of the haplotype -- the 16 + d in front, everything
    // behind column H -- names the ZERO symbol, whose priors are 0 for every row: a lane that computes such a column leaves the column-0
    // state of its rows as it is (M = t * 0, X chains down from the boundary's 0 through M = 0, Y = Y * p_yy + 0 keeps 0 and, in a
    // boundary replica with p_yy = 1, init), and adds 0 to both sums.  So the sweep needs no predicate at all: lanes "start" before
    // their first column and run on behind their last one, and pairs with a shorter haplotype than the wave's longest idle exactly.
    // A base outside A, C, G, T makes the pair `dirty`. ----
    const uint32_t hoff = a.hap_off[h];
    const uint32_t d = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(a.hap_bases) + hoff) & 3u);  // (the table may begin at any byte)
    bool dirty = false;
    {
        constexpr uint32_t kZero = 4u * kTabSymStride;
        const uint8_t *__restrict__ src = a.hap_bases + hoff - d;  // dword-aligned (arithmetic on the kernel argument: global loads, not flat)
        if (lane < 8) reinterpret_cast<uint32_t *>(hapl)[lane] = kZero | (kZero << 16);  // the 16 entries in front
        const uint32_t rounds = tab_hap_rounds(a.tab_max_hap);
        for (uint32_t it = 0; it < rounds; ++it) {
            const uint32_t i = it * 16u + lane;  // dword of the haplotype's span: bytes 4 i .. 4 i + 3, of which [d, d + H) are the haplotype's
            const uint32_t b0 = 4u * i;
            uint2 e;
            e.x = e.y = kZero | (kZero << 16);
            if (b0 + 3u >= d && b0 < d + H) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(src + b0);
                const uint32_t x = (w >> 1) & 0x03030303u;                         // A 0, C 1, T 2, G 3 in every byte (ASCII bits 1-2)
                const uint32_t back = __builtin_amdgcn_perm(0u, 0x47544341u, x);  // the four codes back to "ACTG": any other byte differs
                const uint32_t e0 = (x & 0xFFu) << 8, e1 = (x & 0xFF00u), e2 = (x >> 8) & 0xFF00u, e3 = (x >> 16) & 0xFF00u;
                if (b0 >= d && b0 + 3u < d + H) {  // all four bytes are columns
                    dirty = dirty || back != w;
                    e.x = e0 | (e1 << 16);
                    e.y = e2 | (e3 << 16);
                }
                else {
                    const bool v0 = b0 >= d && b0 < d + H, v1 = b0 + 1u >= d && b0 + 1u < d + H, v2 = b0 + 2u >= d && b0 + 2u < d + H,
                               v3 = b0 + 3u >= d && b0 + 3u < d + H;
                    const uint32_t m = (v0 ? 0xFFu : 0u) | (v1 ? 0xFF00u : 0u) | (v2 ? 0xFF0000u : 0u) | (v3 ? 0xFF000000u : 0u);
                    dirty = dirty || ((back ^ w) & m) != 0u;
                    e.x = (v0 ? e0 : kZero) | ((v1 ? e1 : kZero) << 16);
                    e.y = (v2 ? e2 : kZero) | ((v3 ? e3 : kZero) << 16);
                }
            }
            *reinterpret_cast<uint2 *>(hapl + 2u * (16u + 4u * i)) = e;
        }
    }
    if (lane == 0) rid[grp] = valid ? r : 0xFFFFFFFFu;
    if (__syncthreads_or(valid && dirty)) {
        // a haplotype with N (or anything else) in this workgroup: the bit-plane sweep, whose planes know those symbols
        if constexpr (UGO)
            systolic_ugo_sweep<T, K, 16, 256>(a, wg, pos_first, count, rescue_list, rescue_count);
        else
            systolic_sweep<T, K, true, 16, 256>(a, wg, pos_first, count, rescue_list, rescue_count);
        return;
    }
    // ---- which table a pair uses: one per run of equal reads among the workgroup's pairs ----
    uint32_t slot = 0, n_tables = 1;
    bool leader = grp == 0;
    {
        uint32_t prev = rid[0];
#pragma unroll
        for (uint32_t g = 1; g < 16; ++g) {
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
    T p_yy[K];
    T r_mm[UGO ? 1 : K], r_mx[UGO ? 1 : K];  // per-base gap-opens: per row
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int row = static_cast<int>(lane) * K + k - pad;
        p_yy[k] = row >= 0 ? p_xx : static_cast<T>(1);  // a boundary replica holds Y with p_yy = 1
        if constexpr (!UGO) {
            r_mm[k] = 0, r_mx[k] = 0;
            if (row >= 0) {
                const uint32_t g = a.read_ins[roff + static_cast<uint32_t>(row)] & 127u;
                r_mm[k] = mmtab[((g * (g + 1)) >> 1) + g];
                r_mx[k] = ph2pr[g];
            }
        }
    }
    auto f_mm = [&](int k) -> T { return UGO ? u_mm : r_mm[UGO ? 0 : k]; };
    auto f_mx = [&](int k) -> T { return UGO ? u_mx : r_mx[UGO ? 0 : k]; };

    const T y_col0 = (static_cast<int>(lane) * K <= pad) ? init_y : static_cast<T>(0);
    // the entry of this lane's column at step t is hp[t]: entry 16 + d + (t - lane) - 1
    const uint16_t *hp = reinterpret_cast<const uint16_t *>(hapl) + (15u + d - lane);

    // The rounds.  A workgroup sweeps once unless it has more reads than table slots.  Two forms of the same text
    // (pairhmm_systolic_tab_round.inc): the first round as straight-line code and the later ones as a loop of their own, or ONE loop
    // over all rounds.  As one loop, what a round derives from `pad`, the read's offset and the pair id (each row's initial Y, the
    // result's address) is hoisted out of the loop and stays live across the sweep, K + 10 registers: from 13 rows per lane on every
    // lane then spills on its way into the sweep -- 0.3-0.9 GB of scratch stores per launch (WRITE_SIZE), more than the algorithmic
    // bytes -- so those classes take the straight-line form, although the loop form runs 2-5 % faster even so (the stores are 5 % of
    // the HBM's time).  (Making the three values opaque per iteration with an empty asm cured the hoisting too and cost the sweep
    // 15 %.)  Below 13 rows nothing spills either way and the choice is by measurement, `tools/bench_sweep.py` on one box, GCUPS as
    // loop / straight-line: one gap-open quality, reads of 60, 100, 125, 150, 170, 190 bases (K = 4, 7, 8, 10, 11, 12): 5 017 / 5 021,
    // 5 777 / 5 804, 6 434 / 6 883, 7 047 / 6 003, 7 427 / 7 056, 7 656 / 7 721; per-base gap-opens: 4 511 / 4 886, 5 149 / 5 328,
    // 5 767 / 5 974, 5 684 / 6 451, 6 338 / 6 579, 6 818 / 6 694 -- at instruction streams that differ by two moves per couple of steps
    // and by which registers the allocator picked ("Register numbers matter": SQ_INSTS_VALU equal, SQ_WAIT_INST_ANY 7.9e9
    // against 6.1e9).  (A second run, K = 2, 3, 5, 6, 9: 3 938 / 3 986, 4 822 / 4 770, 5 891 / 5 491, 6 319 / 6 259, 6 772 / 6 711; per-base
    // 3 773 / 3 885, 4 495 / 4 451, 5 317 / 5 222, 5 509 / 5 472, 6 358 / 6 216.)  So: the loop form for 5, 6, 9, 10 and 11 rows per lane
    // with one gap-open quality, the straight-line form everywhere else (tab_one_loop_default; since the classes are also built with
    // the scheduler that suits them, the form of a class launched on its own is tab_plan's).
    // -DPHMM_TAB_LOOP_FORM_ALL / _NONE: tuning (tools/build_tab_variant.sh).
#if defined(PHMM_TAB_LOOP_FORM_ALL)
    constexpr bool kOneLoop = true;
#elif defined(PHMM_TAB_LOOP_FORM_NONE)
    constexpr bool kOneLoop = false;
#else
    constexpr bool kOneLoop = ONE_LOOP;
#endif
    if constexpr (!kOneLoop) {
        {
            const uint32_t base = 0;
#include "pairhmm_systolic_tab_round.inc"
        }
#pragma unroll 1
        for (uint32_t base = n_slots; base < n_tables; base += n_slots) {
            __syncthreads();  // the round before has read its tables
#include "pairhmm_systolic_tab_round.inc"
        }
    }
    else {
#pragma unroll 1
        for (uint32_t base = 0; base < n_tables; base += n_slots) {
            if (base) __syncthreads();  // the round before has read its tables
#include "pairhmm_systolic_tab_round.inc"
        }
    }
}

}  // namespace phmm
#endif
