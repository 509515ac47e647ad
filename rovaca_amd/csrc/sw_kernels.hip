// sw_kernels.hip -- gfx950 (MI355X) Smith-Waterman with backtrack: the reference's sw_avx.c, bit for bit.
//
// What is computed (common/base/lib/sw_avx/sw_avx.c): an int32 affine-gap alignment of seq2 ("alt", columns j) against seq1
// ("ref", rows i) with free leading/trailing overhangs (soft clips), the best end cell on the last row / last column with the
// reference's tie rules, the traceback and the merged CIGAR:
//   per cell  (:41-77)   e = max(H[i][j-1] + open, E[i][j-1] + ext)   extension wins ties (flag 4)
//                        f = max(F[i-1][j] + ext, H[i-1][j] + open)   extension wins ties (flag 8)
//                        m = H[i-1][j-1] + (seq1[i] == seq2[j] ? match : mismatch)
//                        h = m; if (e > h) { h = e; op = insert }  if (f > h) { h = f; op = delete }
//                        (the MATRIX_MIN_CUTOFF clamp of :66 can never bind below 2048 bases: h >= -260 * 2047)
//   end cell  (:200-217), traceback (:256-330), merge + element encoding (:333-358, :234-254).
//
// Mapping to CDNA4: the same systolic scheme as the PairHMM kernels, transposed.  An alignment takes G lanes (16: four
// alignments per wave, DPP row_shr; 64: one per wave, wave_shr); lane l owns K consecutive columns of seq2 in VGPRs (H and F of
// the row above, the bases), and at step t computes row t - l of its columns.  E runs left to right inside the lane and
// crosses to the next lane together with H through two DPP moves per step; the lane keeps last step's incoming H as the
// diagonal of its first column.  Backtrack flags are 4 comparison bits per cell, appended with v_addc_co_u32 (the borrow of the
// v_sub_co_u32 that made the comparison), 8 cells per dword.  With 16 lanes the score of a cell comes out of a per-alignment table
// in LDS (below).  The end-cell search is a parallel max followed by an in-order pass over the few cells that attain it; the
// traceback is walked by the alignment's lanes together, a fetch per G diagonal moves (bound by the sectors it touches, overlapped
// with other workgroups' sweeps).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "sw_device.h"

namespace swhip
{

template <int G>
__device__ __forceinline__ int32_t shr1(int32_t old, int32_t src)
{
    // lane i receives lane i-1 of its G-lane array; lane 0 keeps `old`
    if constexpr (G == 64)
        return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xF, 0xF, false);  // wave_shr:1
    else
        return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xF, 0xF, false);  // row_shr:1
}

template <int G>
__device__ __forceinline__ int32_t group_max(int32_t v)
{
#pragma unroll
    for (int d = 1; d < G; d <<= 1) v = max(v, __shfl_xor(v, d));
    return v;
}

// Score table (round 5; up to 16 columns per lane): the scores of a lane's K columns against each of the bases A, C, G, T as packed
// 16-bit values in LDS, [base][half][lane] x 16 bytes per alignment (K <= 8: one half).  A row's step reads the 16 or 32 bytes of its
// seq1 base once (ds_read_b128, no bank conflicts: a lane's entries are its own, 16 bytes apart) and the cell adds the score with a
// word-selected, sign-extended operand (v_add_u32_sdwa ... sext(src1) src1_sel:WORD_n, half rate) instead of compare, select and add:
// 19 -> 17 instructions, 9.1 -> 4.2 cycles for the score.  Equality of arbitrary bytes is not a table lookup, so a workgroup with a byte
// other than ACGT in any of its sequences sweeps with the compare (same results; the scores always fit, the ABI limits them to +-512).
template <int K, int G>
__host__ __device__ constexpr uint32_t score_table_base_stride()
{
    return (uint32_t)G * 16u * (K > 8 ? 2u : 1u);  // bytes per base: G lanes x 16 bytes x (one or two halves)
}

__device__ __forceinline__ bool is_acgt(uint32_t b) { return ((0x47544341u >> (((b >> 1) & 3u) * 8u)) & 0xFFu) == b; }  // "ACTG"[(b >> 1) & 3]

template <int K, int G>
__global__ void __launch_bounds__(kBlockThreads) sw_kernel(SwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_seq1[];  // [alignments per block][lds_stride]: seq1 of each alignment
    constexpr uint32_t W = (K + 7) / 8;  // backtrack words a lane produces per step
    // they are stored as W32 full dwords plus a tail of 1, 2 or 4 bytes for the last K % 8 cells, in two arrays
    // ([step][lane][W32] dwords, then [step][lane] tails): the kernel is bound by these writes, so 10 columns per lane cost
    // 5 bytes per step instead of 8
    constexpr uint32_t W32 = K / 8, TAILB = bt_tail_bytes(K);
    const uint32_t lane = threadIdx.x & (G - 1);
    const uint32_t gidx = blockIdx.x * (blockDim.x / G) + threadIdx.x / G;
    const bool valid = gidx < a.n_pairs;
    const uint32_t pair = a.order[valid ? gidx : 0];
    const uint32_t r = a.pair_ref[pair], s = a.pair_seq[pair];
    const int32_t nrow = (int32_t)a.ref_len[r], ncol = (int32_t)a.seq_len[s];
    const uint8_t *__restrict__ s1 = a.ref_bases + a.ref_off[r];
    const uint8_t *__restrict__ s2 = a.seq_bases + a.seq_off[s];
    // backtrack flags: global memory, or -- for small batches, where the traceback's dependent loads are what the call waits
    // for -- the workgroup's LDS behind seq1 (one alignment per workgroup then)
    uint32_t *bt = a.bt_in_lds ? reinterpret_cast<uint32_t *>(lds_seq1 + (blockDim.x / G) * a.lds_stride) : a.bt + a.bt_off[pair];
    uint8_t *bt_tail = reinterpret_cast<uint8_t *>(bt + (size_t)(nrow + G - 1) * G * (K / 8));
    int32_t *__restrict__ lastcol = a.scratch + a.scratch_off[pair];  // [i-1]: H[i][ncol]
    int32_t *__restrict__ lastrow = lastcol + nrow;                    // [j-1]: H[nrow][j]
    uint32_t *__restrict__ elems = reinterpret_cast<uint32_t *>(lastrow + ncol);
    const uint32_t w_match = (uint32_t)a.w_match, w_mismatch = (uint32_t)a.w_mismatch, w_open = (uint32_t)a.w_open,
                   w_extend = (uint32_t)a.w_extend;

    // ---- per-column state ----
    uint32_t Hp[K], F[K];  // H[i-1][j], F[i-1][j] of this lane's columns, biased by kBias
    uint32_t base2[K];     // seq2[j]; columns past the end never match (their cells feed nothing)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int32_t j = (int32_t)lane * K + k + 1;
        base2[k] = j <= ncol ? (uint32_t)s2[j - 1] : 0x100u;
        Hp[k] = kBias;     // H[0][j] = 0          (sw_avx.c:196)
        F[k] = kLowBiased; // F[0][j] = LOW_INIT   (sw_avx.c:117-119, :198)
    }
    const uint32_t lc = (uint32_t)(ncol - 1) / K, kc = (uint32_t)(ncol - 1) % K;  // where column ncol lives
    // seq1 is read one base per step per lane: park it in LDS (a byte load from LDS instead of a vector memory instruction)
    uint8_t *my1 = lds_seq1 + (threadIdx.x / G) * a.lds_stride;
    bool other_bytes = false;  // a byte that is none of A, C, G, T
    if (valid)
        for (int32_t x = (int32_t)lane; x < nrow; x += G) {
            const uint32_t b = s1[x];
            my1[x] = (uint8_t)b;
            other_bytes |= !is_acgt(b);
        }
    constexpr bool kCanTable = K <= 16;  // (a lane's scores in at most two 16-byte reads; wider lanes keep the compare)
    bool use_table = false;
    constexpr uint32_t BS = score_table_base_stride<K, G>();
    // this lane's entries for base code 0 (behind the workgroup's seq1 copies)
    uint8_t *tab = lds_seq1 + (blockDim.x / G) * a.lds_stride + (threadIdx.x / G) * (4u * BS) + lane * 16u;
    if constexpr (kCanTable) {
        if (a.score_table) {
#pragma unroll
            for (int k = 0; k < K; ++k) other_bytes |= base2[k] != 0x100u && !is_acgt(base2[k]);
            use_table = !__syncthreads_or(other_bytes);
            if (use_table) {
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c) {
                    const uint32_t base_c = (0x47544341u >> (c * 8u)) & 0xFFu;
                    uint32_t packed[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) packed[d] = 0;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const uint32_t sc = (base2[k] == base_c ? w_match : w_mismatch) & 0xFFFFu;
                        packed[k / 2] |= sc << (16 * (k % 2));
                    }
                    *reinterpret_cast<uint4 *>(tab + c * BS) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
                    if constexpr (K > 8) *reinterpret_cast<uint4 *>(tab + c * BS + (uint32_t)G * 16u) = make_uint4(packed[4], packed[5], packed[6], packed[7]);
                }
            }
        }
        else {
            __syncthreads();
        }
    }
    else {
        __syncthreads();
    }

    // what this lane hands to its right neighbour: H and E of its last column in the row it has just finished; before its
    // first row that is column 0 of ... no: row 0 of that column, H = 0 and E = LOW_INIT, which is also what lane 0 keeps
    // receiving from the left edge (H[i][0] = 0, E[i][0] = LOW_INIT; sw_avx.c:120-122, :197, :199)
    uint32_t outH = kBias, outE = kLowBiased;
    uint32_t inH = kBias, inE = kLowBiased;  // received this step: H[i][jL], E[i][jL] with jL the neighbour's last column

    const int32_t Tg = valid ? nrow + (G - 1) : 0;
    int32_t Tmax = Tg;
    if (G == 16) {
        Tmax = max(Tmax, __shfl_xor(Tmax, 16));
        Tmax = max(Tmax, __shfl_xor(Tmax, 32));
    }
    Tmax = __builtin_amdgcn_readfirstlane(Tmax);

    // the scoring constants as VGPRs once (the cell's assembly wants registers; left alone the compiler re-materialises them
    // from SGPRs every step)
    uint32_t vw_match = w_match, vw_mismatch = w_mismatch, vw_open = w_open, vw_extend = w_extend;
    asm volatile("" : "+v"(vw_match), "+v"(vw_mismatch), "+v"(vw_open), "+v"(vw_extend));

    uint32_t next1 = (valid && lane == 0) ? (uint32_t)my1[0] : 0u;  // seq1 base of the row this lane does next
    // with the score table: the packed scores of this lane's columns against the base of the row it does now (cw) -- fetched during the
    // step before, with the base read the step before that (next1 is then the base of the row AFTER the one being done)
    constexpr int KW = K > 8 ? 8 : 4;
    uint32_t cw[KW];
    auto fetch_scores = [&](uint32_t base, uint32_t (&w)[KW]) __attribute__((always_inline)) {
        const uint8_t *src = tab + ((base >> 1) & 3u) * BS;
        const uint4 lo = *reinterpret_cast<const uint4 *>(src);
        w[0] = lo.x, w[1] = lo.y, w[2] = lo.z, w[3] = lo.w;
        if constexpr (KW == 8) {
            const uint4 hi = *reinterpret_cast<const uint4 *>(src + (uint32_t)G * 16u);
            w[4] = hi.x, w[5] = hi.y, w[6] = hi.z, w[7] = hi.w;
        }
    };
    // One step.  `steady` = every lane of the wave is inside its alignment and none is on its last row (with the table: nor on the
    // row before it): no predicate, no last-row capture, the next base is always there.
    auto step = [&](int32_t t, auto steady, auto tabbed) __attribute__((always_inline)) {
        constexpr bool TAB = decltype(tabbed)::value;
        const int32_t i = t - (int32_t)lane;  // this lane's row
        const bool active = steady || (valid && i >= 1 && i <= nrow);
        const uint32_t diag0 = inH;           // H[i-1][jL]: what came in one step ago
        inH = (uint32_t)shr1<G>((int32_t)inH, (int32_t)outH);
        inE = (uint32_t)shr1<G>((int32_t)inE, (int32_t)outE);
        const uint32_t base1 = next1;
        uint32_t nw[KW];
        if constexpr (TAB) {
            fetch_scores(next1, nw);   // row i + 1's scores
            const int32_t ni = i + 2;  // and the base of the row after it
            next1 = (steady || (valid && ni >= 1 && ni <= nrow)) ? (uint32_t)my1[ni - 1] : 0u;
        }
        else {
            const int32_t ni = i + 1;  // prefetch the base of the next row while this one is computed
            next1 = (steady || (valid && ni >= 1 && ni <= nrow)) ? (uint32_t)my1[ni - 1] : 0u;
        }
        if (active) {
            uint32_t hl = inH, e = inE, hd = diag0, h_ncol = 0;
            uint32_t words[W];
#pragma unroll
            for (uint32_t w = 0; w < W; ++w) words[w] = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // One cell = 19 VALU instructions, written out by hand around what tools/ubench_valu.hip measures on gfx950 (cycles per
                // wave instruction and SIMD at 4 waves): v_add_u32, v_sub_co_u32, v_cndmask-from-VCC and the VOP2 logic ops issue at
                // full rate (2.4-2.7); v_cmp to VCC, v_addc_co_u32 (carry in AND out), v_max/v_min, v_alignbit, v_lshl_add, v_and_or
                // and every compare/select through an SGPR pair at HALF rate (4.2-4.4) -- and hipcc picks those.  So each maximum is
                // v_sub_co_u32 (its borrow is "a < b"; the difference is thrown away), a select out of VCC, and v_addc_co_u32 acc, vcc,
                // acc, acc, vcc appends the same bit to the flag word (half rate, but so is every other one-instruction way to append a
                // bit).  Only the base comparison stays a VOPC.  Scores live biased by 2^30 (unsigned order is then signed order).  Two
                // wait states between a VALU write of VCC and a VALU read of it; independent work fills them where there is some,
                // s_nop where there is none (inline assembly is not visible to the hazard recogniser).
                uint32_t e11, f11 = F[k], h11, acc = words[k / 8];  // f11: F[i-1][j] in, F[i][j] out
                uint32_t t_exth, t_openh, t_sc, t_m11, t_extv, t_openv, t_h1, t_d;
                if constexpr (TAB) {
                    // (17 instructions: the score comes out of the row's table words, the low or the high half of cw[k / 2])
                    (void)t_sc;
                    if (k % 2 == 0) {
                    asm volatile(
                            "v_add_u32 %[exth], %[e], %[wext]\n\t"
                            "v_add_u32 %[openh], %[hl], %[wopen]\n\t"
                            "v_add_u32_sdwa %[m11], %[hd], sext(%[scw]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                            "v_sub_co_u32 %[d], vcc, %[exth], %[openh]\n\t"   // E: the gap opens here (extension wins ties)
                            "v_add_u32 %[extv], %[f11], %[wext]\n\t"
                            "v_add_u32 %[openv], %[hp], %[wopen]\n\t"
                            "v_cndmask_b32 %[e11], %[exth], %[openh], vcc\n\t"
                            "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                            "v_sub_co_u32 %[d], vcc, %[extv], %[openv]\n\t"   // F: the gap opens here
                            "s_nop 1\n\t"
                            "v_cndmask_b32 %[f11], %[extv], %[openv], vcc\n\t"
                            "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                            "v_sub_co_u32 %[d], vcc, %[m11], %[e11]\n\t"       // E beats the diagonal: insertion
                            "s_nop 1\n\t"
                            "v_cndmask_b32 %[h1], %[m11], %[e11], vcc\n\t"
                            "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                            "v_sub_co_u32 %[d], vcc, %[h1], %[f11]\n\t"        // F beats both: deletion
                            "s_nop 1\n\t"
                            "v_cndmask_b32 %[h11], %[h1], %[f11], vcc\n\t"
                            "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc"
                            : [acc] "+v"(acc), [e11] "=&v"(e11), [f11] "+v"(f11), [h11] "=&v"(h11), [exth] "=&v"(t_exth),
                              [openh] "=&v"(t_openh), [m11] "=&v"(t_m11), [extv] "=&v"(t_extv), [openv] "=&v"(t_openv),
                              [h1] "=&v"(t_h1), [d] "=&v"(t_d)
                            : [scw] "v"(cw[k / 2]), [e] "v"(e), [hl] "v"(hl), [hd] "v"(hd), [hp] "v"(Hp[k]), [wext] "v"(vw_extend),
                              [wopen] "v"(vw_open)
                            : "vcc");
                    }
                    else {
                    asm volatile(
                            "v_add_u32 %[exth], %[e], %[wext]\n\t"
                            "v_add_u32 %[openh], %[hl], %[wopen]\n\t"
                            "v_add_u32_sdwa %[m11], %[hd], sext(%[scw]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                            "v_sub_co_u32 %[d], vcc, %[exth], %[openh]\n\t"   // E: the gap opens here (extension wins ties)
                            "v_add_u32 %[extv], %[f11], %[wext]\n\t"
                            "v_add_u32 %[openv], %[hp], %[wopen]\n\t"
                            "v_cndmask_b32 %[e11], %[exth], %[openh], vcc\n\t"
                            "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                            "v_sub_co_u32 %[d], vcc, %[extv], %[openv]\n\t"   // F: the gap opens here
                            "s_nop 1\n\t"
                            "v_cndmask_b32 %[f11], %[extv], %[openv], vcc\n\t"
                            "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                            "v_sub_co_u32 %[d], vcc, %[m11], %[e11]\n\t"       // E beats the diagonal: insertion
                            "s_nop 1\n\t"
                            "v_cndmask_b32 %[h1], %[m11], %[e11], vcc\n\t"
                            "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                            "v_sub_co_u32 %[d], vcc, %[h1], %[f11]\n\t"        // F beats both: deletion
                            "s_nop 1\n\t"
                            "v_cndmask_b32 %[h11], %[h1], %[f11], vcc\n\t"
                            "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc"
                            : [acc] "+v"(acc), [e11] "=&v"(e11), [f11] "+v"(f11), [h11] "=&v"(h11), [exth] "=&v"(t_exth),
                              [openh] "=&v"(t_openh), [m11] "=&v"(t_m11), [extv] "=&v"(t_extv), [openv] "=&v"(t_openv),
                              [h1] "=&v"(t_h1), [d] "=&v"(t_d)
                            : [scw] "v"(cw[k / 2]), [e] "v"(e), [hl] "v"(hl), [hd] "v"(hd), [hp] "v"(Hp[k]), [wext] "v"(vw_extend),
                              [wopen] "v"(vw_open)
                            : "vcc");
                    }
                }
                else {
                asm volatile(
                        "v_cmp_eq_u32 vcc, %[b1], %[b2]\n\t"
                        "v_add_u32 %[exth], %[e], %[wext]\n\t"
                        "v_add_u32 %[openh], %[hl], %[wopen]\n\t"
                        "v_cndmask_b32 %[sc], %[wmm], %[wm], vcc\n\t"
                        "v_sub_co_u32 %[d], vcc, %[exth], %[openh]\n\t"   // E: the gap opens here (extension wins ties)
                        "v_add_u32 %[m11], %[hd], %[sc]\n\t"
                        "v_add_u32 %[extv], %[f11], %[wext]\n\t"
                        "v_cndmask_b32 %[e11], %[exth], %[openh], vcc\n\t"
                        "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                        "v_add_u32 %[openv], %[hp], %[wopen]\n\t"
                        "v_sub_co_u32 %[d], vcc, %[extv], %[openv]\n\t"   // F: the gap opens here
                        "s_nop 1\n\t"
                        "v_cndmask_b32 %[f11], %[extv], %[openv], vcc\n\t"
                        "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                        "v_sub_co_u32 %[d], vcc, %[m11], %[e11]\n\t"       // E beats the diagonal: insertion
                        "s_nop 1\n\t"
                        "v_cndmask_b32 %[h1], %[m11], %[e11], vcc\n\t"
                        "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
                        "v_sub_co_u32 %[d], vcc, %[h1], %[f11]\n\t"        // F beats both: deletion
                        "s_nop 1\n\t"
                        "v_cndmask_b32 %[h11], %[h1], %[f11], vcc\n\t"
                        "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc"
                        : [acc] "+v"(acc), [e11] "=&v"(e11), [f11] "+v"(f11), [h11] "=&v"(h11), [exth] "=&v"(t_exth),
                          [openh] "=&v"(t_openh), [sc] "=&v"(t_sc), [m11] "=&v"(t_m11), [extv] "=&v"(t_extv), [openv] "=&v"(t_openv),
                          [h1] "=&v"(t_h1), [d] "=&v"(t_d)
                        : [b1] "v"(base1), [b2] "v"(base2[k]), [e] "v"(e), [hl] "v"(hl), [hd] "v"(hd), [hp] "v"(Hp[k]),
                          [wext] "v"(vw_extend), [wopen] "v"(vw_open), [wm] "v"(vw_match), [wmm] "v"(vw_mismatch)
                        : "vcc");
                }
                words[k / 8] = acc;
                hd = Hp[k];  // the diagonal of the next column
                h_ncol = (uint32_t)k == kc ? h11 : h_ncol;
                Hp[k] = h11;
                F[k] = f11;
                hl = h11;
                e = e11;
            }
            outH = hl;
            outE = e;
            {
                const size_t slot = (size_t)(t - 1) * G + lane;
                uint32_t *dst = bt + slot * W32;
#pragma unroll
                for (uint32_t w = 0; w < W32; ++w) dst[w] = words[w];
                if constexpr (TAILB == 1) bt_tail[slot] = (uint8_t)words[W - 1];
                if constexpr (TAILB == 2) reinterpret_cast<uint16_t *>(bt_tail)[slot] = (uint16_t)words[W - 1];
                if constexpr (TAILB == 4) reinterpret_cast<uint32_t *>(bt_tail)[slot] = words[W - 1];
            }
            if (lane == lc) lastcol[i - 1] = (int32_t)(h_ncol - kBias);  // column ncol, every row
            if (!steady && i == nrow) {  // row nrow, every column
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int32_t j = (int32_t)lane * K + k + 1;
                    if (j <= ncol) lastrow[j - 1] = (int32_t)(Hp[k] - kBias);
                }
            }
        }
        if constexpr (TAB) {
#pragma unroll
            for (int w = 0; w < KW; ++w) cw[w] = nw[w];
        }
    };
    // fill (lanes join one by one), steady, drain (lanes and shorter alignments leave)
    auto sweep = [&](auto tabbed) __attribute__((always_inline)) {  // (forced: a call would put the column state in scratch)
        constexpr bool TAB = decltype(tabbed)::value;
        // last step at which every lane of this alignment is below its last row (with the table: and the one before it)
        int32_t Tsteady = valid ? nrow - 1 - (TAB ? 1 : 0) : 0;
        if (G == 16) {
            Tsteady = min(Tsteady, __shfl_xor(Tsteady, 16));
            Tsteady = min(Tsteady, __shfl_xor(Tsteady, 32));
        }
        Tsteady = __builtin_amdgcn_readfirstlane(Tsteady);
        if constexpr (TAB) {  // before step 1: lane 0 is about to do row 1, lane 1 has row 1 next, lane 0 row 2
            fetch_scores(next1, cw);
            const int32_t r1 = 2 - (int32_t)lane;
            next1 = (valid && r1 >= 1 && r1 <= nrow) ? (uint32_t)my1[r1 - 1] : 0u;
        }
        int32_t t = 1;
        for (; t < G && t <= Tmax; ++t) step(t, std::false_type{}, tabbed);
        for (; t + 1 <= Tsteady; t += 2) {  // two steps per trip: the row above (Hp) alternates between two register sets instead of being copied
            step(t, std::true_type{}, tabbed);
            step(t + 1, std::true_type{}, tabbed);
        }
        for (; t <= Tsteady; ++t) step(t, std::true_type{}, tabbed);
        for (; t <= Tmax; ++t) step(t, std::false_type{}, tabbed);
    };
    if constexpr (kCanTable) {
        if (use_table)
            sweep(std::true_type{});
        else
            sweep(std::false_type{});
    }
    else {
        sweep(std::false_type{});
    }
    __threadfence_block();  // this lane's own stores are visible to the loads below (same work-item or same wave)
    __syncthreads();

    // ---- best end cell (sw_avx.c:200-217) ----
    // Phase 1: the maximum over the last row and the last column.
    int32_t best = INT32_MIN;
    if (valid) {
        for (int32_t j = (int32_t)lane; j < ncol; j += G) best = max(best, lastrow[j]);
        for (int32_t i = (int32_t)lane; i < nrow; i += G) best = max(best, lastcol[i]);
    }
    best = group_max<G>(best);
    // Phase 2: the cells that attain it, in the reference's order (anti-diagonal ascending; on one anti-diagonal the
    // last-row cell before the last-column cell), with its tie rules.  The first of them always replaces what came before.
    int32_t max_i = 0, max_j = 0;
    bool have = false;
    if (valid) {
        const uint32_t gshift = (threadIdx.x & 63u) / G * G;  // where this group's bits sit in a 64-lane ballot
        for (int32_t base = 1; base <= nrow + ncol; base += G) {
            const int32_t ad = base + (int32_t)lane;
            bool eq_a = false, eq_b = false;
            if (ad <= nrow + ncol) {
                if (ad >= nrow + 1) eq_a = lastrow[ad - nrow - 1] == best;                          // cell (nrow, ad - nrow)
                if (ad >= ncol + 1) eq_b = lastcol[ad - ncol - 1] == best;  // cell (ad - ncol, ncol)
            }
            uint64_t ma = (__ballot(eq_a) >> gshift), mb = (__ballot(eq_b) >> gshift);
            if (G < 64) {
                ma &= (1ull << G) - 1;
                mb &= (1ull << G) - 1;
            }
            uint64_t any = ma | mb;
            while (any) {
                const int l = __builtin_ctzll(any);
                any &= any - 1;
                const int32_t d = base + l;
                if ((ma >> l) & 1) {
                    const int32_t ci = nrow, cj = d - nrow;
                    if (!have || abs(ci - cj) < abs(max_i - max_j)) {
                        max_i = ci;
                        max_j = cj;
                        have = true;
                    }
                }
                if ((mb >> l) & 1) {
                    const int32_t ci = d - ncol, cj = ncol;
                    if (!have || max_j == ncol || abs(ci - cj) <= abs(max_i - max_j)) {
                        max_i = ci;
                        max_j = cj;
                        have = true;
                    }
                }
            }
        }
    }

    // ---- traceback and CIGAR (sw_avx.c:256-362) ----
    // The walk is a chain of dependent loads of backtrack words that mostly sit in HBM by now (a launch writes gigabytes of them),
    // and nearly every move is diagonal.  So the G lanes of the alignment walk together: lane q fetches the flags of cell
    // (i - q, j - q) -- the next G cells if the path stays on this diagonal -- and then every lane runs the same state machine over
    // them (values passed with a shuffle; i, j and the state are uniform over the group), stopping at the first move that leaves the
    // diagonal, where the fetch is repeated from the new cell.  One round trip to memory per G diagonal moves instead of one per move.
    // Only lane 0 writes.
    if (valid) {
        int32_t i = max_i, j = max_j;
        uint32_t n_el = 0;
        int32_t cur_state = -1, cur_len = 0;  // the element being built; equal neighbours are merged as they come
        auto emit = [&](int32_t state, int32_t len) {
            if (state == cur_state) {
                cur_len += len;
            }
            else {
                if (cur_state >= 0) {
                    if (lane == 0) elems[n_el] = ((uint32_t)cur_len << 4) | (uint32_t)cur_state;
                    ++n_el;
                }
                cur_state = state;
                cur_len = len;
            }
        };
        auto flags_of = [&](int32_t ci, int32_t cj) -> uint32_t {
            const uint32_t l = (uint32_t)(cj - 1) / K, k = (uint32_t)(cj - 1) % K;
            const uint32_t t = (uint32_t)ci + l;
            const size_t slot = (size_t)(t - 1) * G + l;
            uint32_t word;
            if (k / 8 < W32)
                word = bt[slot * W32 + k / 8];
            else if (TAILB == 1)
                word = bt_tail[slot];
            else if (TAILB == 2)
                word = reinterpret_cast<const uint16_t *>(bt_tail)[slot];
            else
                word = reinterpret_cast<const uint32_t *>(bt_tail)[slot];
            constexpr uint32_t last_cells = K % 8 == 0 ? 8u : (uint32_t)(K % 8);      // cells in the last (partial) word
            const uint32_t in_word = k / 8 + 1 == W ? last_cells : 8u;
            return (word >> (4 * (in_word - 1 - k % 8))) & 15u;                         // first cell of a word sits highest
        };
        if (j < ncol) emit((int32_t)kOpS, ncol - j);
        uint32_t state = 0;
        while (i > 0 && j > 0) {
            const int32_t ci = i - (int32_t)lane, cj = j - (int32_t)lane;
            const uint32_t mine = (ci > 0 && cj > 0) ? flags_of(ci, cj) : 0u;
            bool on_diagonal = true;
            for (uint32_t q = 0; q < (uint32_t)G && on_diagonal && i > 0 && j > 0; ++q) {
                const uint32_t nib = (uint32_t)__shfl((int)mine, (int)q, G);
                // back to the reference's flags (smithwaterman_common.h:41-46)
                const uint32_t btr = ((nib & 8u) ? 0u : (uint32_t)kBtInsertExt) | ((nib & 4u) ? 0u : (uint32_t)kBtDeleteExt) |
                                     ((nib & 1u) ? (uint32_t)kBtDelete : (nib & 2u) ? (uint32_t)kBtInsert : 0u);
                on_diagonal = false;
                if (state == kBtInsertExt) {
                    --j;
                    cur_len += 1;  // lengthens the insertion that is being built
                    state = btr & kBtInsertExt;
                }
                else if (state == kBtDeleteExt) {
                    --i;
                    cur_len += 1;
                    state = btr & kBtDeleteExt;
                }
                else {
                    const uint32_t op = btr & 3u;
                    if (op == kBtInsert) {
                        --j;
                        emit((int32_t)kOpI, 1);
                        state = btr & kBtInsertExt;
                    }
                    else if (op == kBtDelete) {
                        --i;
                        emit((int32_t)kOpD, 1);
                        state = btr & kBtDeleteExt;
                    }
                    else {
                        --i;
                        --j;
                        emit((int32_t)kOpM, 1);
                        state = 0;
                        on_diagonal = true;
                    }
                }
            }
        }
        if (j > 0) emit((int32_t)kOpS, j);
        if (lane != 0) return;
        if (cur_state >= 0) elems[n_el++] = ((uint32_t)cur_len << 4) | (uint32_t)cur_state;
        // reversed into the output; the reference's element buffer holds kMaxSeqLen - 1 of them (sw_avx.c:349-358)
        const uint32_t count = n_el < (uint32_t)(kMaxSeqLen - 1) ? n_el : (uint32_t)(kMaxSeqLen - 1);
        const uint32_t off = atomicAdd(a.cigar_cursor, count);
        for (uint32_t c = 0; c < count; ++c) a.cigars[off + c] = elems[n_el - 1 - c];
        Alignment res;
        res.score = best;
        res.max_i = (int16_t)max_i;
        res.max_j = (int16_t)max_j;
        res.alignment_offset = (int16_t)i;
        res.flags = 0;
        res.cigar_count = count;
        res.cigar_off = off;
        a.out[pair] = res;
    }
}

// ----------------------------------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------------------------------
template <int K, int G>
static hipError_t launch_one(const SwArgs &a, hipStream_t s)
{
    const uint32_t threads = a.bt_in_lds ? (uint32_t)G : (uint32_t)kBlockThreads;  // LDS-resident flags: one alignment per workgroup
    const uint32_t per_block = threads / G;
    const uint32_t blocks = (a.n_pairs + per_block - 1) / per_block;
    const size_t shmem = (size_t)per_block * a.lds_stride + (a.bt_in_lds ? a.lds_bt_bytes : 0u) +
                         (K <= 16 && a.score_table ? (size_t)per_block * score_table_bytes(K, G) : 0u);
    if (shmem > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sw_kernel<K, G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((sw_kernel<K, G>), dim3(blocks), dim3(threads), shmem, s, a);
    return hipGetLastError();
}

int pick_cols_per_lane(uint32_t len2, int lanes_per_pair)
{
    const uint32_t need = (len2 + (uint32_t)lanes_per_pair - 1) / (uint32_t)lanes_per_pair;
    if (lanes_per_pair == 16) {
        for (int k : {2, 4, 6, 8, 10, 12, 14, 16})
            if ((uint32_t)k >= need) return k;
        return 0;
    }
    for (int k : {2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32})
        if ((uint32_t)k >= need) return k;
    return 0;
}

hipError_t launch_sw(int cols_per_lane, int lanes_per_pair, const SwArgs &a, hipStream_t s)
{
    if (a.n_pairs == 0) return hipSuccess;
    if (lanes_per_pair == 16) {
        switch (cols_per_lane) {
            case 2: return launch_one<2, 16>(a, s);
            case 4: return launch_one<4, 16>(a, s);
            case 6: return launch_one<6, 16>(a, s);
            case 8: return launch_one<8, 16>(a, s);
            case 10: return launch_one<10, 16>(a, s);
            case 12: return launch_one<12, 16>(a, s);
            case 14: return launch_one<14, 16>(a, s);
            case 16: return launch_one<16, 16>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    if (lanes_per_pair == 64) {
        switch (cols_per_lane) {
            case 2: return launch_one<2, 64>(a, s);
            case 3: return launch_one<3, 64>(a, s);
            case 4: return launch_one<4, 64>(a, s);
            case 5: return launch_one<5, 64>(a, s);
            case 6: return launch_one<6, 64>(a, s);
            case 8: return launch_one<8, 64>(a, s);
            case 10: return launch_one<10, 64>(a, s);
            case 12: return launch_one<12, 64>(a, s);
            case 16: return launch_one<16, 64>(a, s);
            case 24: return launch_one<24, 64>(a, s);
            case 32: return launch_one<32, 64>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace swhip
