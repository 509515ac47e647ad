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
// diagonal of its first column.  Backtrack flags are 4 bits per cell, packed 8 per dword, stored [step][lane][word].
// The end-cell search is a parallel max followed by an in-order pass over the few cells that attain it; the traceback is
// one lane walking the flags (latency-bound, overlapped with other waves' sweeps).
#include <hip/hip_runtime.h>
#include <stdint.h>

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

template <int K, int G>
__global__ void __launch_bounds__(kBlockThreads) sw_kernel(SwArgs a)
{
    extern __shared__ uint8_t lds_seq1[];  // [alignments per block][lds_stride]: seq1 of each alignment
    constexpr uint32_t W = (K + 7) / 8;  // backtrack words per lane per step
    const uint32_t lane = threadIdx.x & (G - 1);
    const uint32_t gidx = blockIdx.x * (kBlockThreads / G) + threadIdx.x / G;
    const bool valid = gidx < a.n_pairs;
    const uint32_t pair = a.order[valid ? gidx : 0];
    const uint32_t r = a.pair_ref[pair], s = a.pair_seq[pair];
    const int32_t nrow = (int32_t)a.ref_len[r], ncol = (int32_t)a.seq_len[s];
    const uint8_t *__restrict__ s1 = a.ref_bases + a.ref_off[r];
    const uint8_t *__restrict__ s2 = a.seq_bases + a.seq_off[s];
    uint32_t *__restrict__ bt = a.bt + a.bt_off[pair];
    int32_t *__restrict__ lastcol = a.scratch + a.scratch_off[pair];  // [i-1]: H[i][ncol]
    int32_t *__restrict__ lastrow = lastcol + nrow;                    // [j-1]: H[nrow][j]
    uint32_t *__restrict__ elems = reinterpret_cast<uint32_t *>(lastrow + ncol);
    const int32_t w_match = a.w_match, w_mismatch = a.w_mismatch, w_open = a.w_open, w_extend = a.w_extend;

    // ---- per-column state ----
    int32_t Hp[K], F[K];   // H[i-1][j], F[i-1][j] of this lane's columns
    uint32_t base2[K];     // seq2[j]; columns past the end never match (their cells feed nothing)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int32_t j = (int32_t)lane * K + k + 1;
        base2[k] = j <= ncol ? (uint32_t)s2[j - 1] : 0x100u;
        Hp[k] = 0;         // H[0][j] = 0          (sw_avx.c:196)
        F[k] = kLowInit;   // F[0][j] = LOW_INIT   (sw_avx.c:117-119, :198)
    }
    const uint32_t lc = (uint32_t)(ncol - 1) / K, kc = (uint32_t)(ncol - 1) % K;  // where column ncol lives
    // seq1 is read one base per step per lane: park it in LDS (a byte load from LDS instead of a vector memory instruction)
    uint8_t *my1 = lds_seq1 + (threadIdx.x / G) * a.lds_stride;
    if (valid)
        for (int32_t x = (int32_t)lane; x < nrow; x += G) my1[x] = s1[x];
    __syncthreads();

    // what this lane hands to its right neighbour: H and E of its last column in the row it has just finished; before its
    // first row that is column 0 of ... no: row 0 of that column, H = 0 and E = LOW_INIT, which is also what lane 0 keeps
    // receiving from the left edge (H[i][0] = 0, E[i][0] = LOW_INIT; sw_avx.c:120-122, :197, :199)
    int32_t outH = 0, outE = kLowInit;
    int32_t inH = 0, inE = kLowInit;  // received this step: H[i][jL], E[i][jL] with jL the neighbour's last column

    const int32_t Tg = valid ? nrow + (G - 1) : 0;
    int32_t Tmax = Tg;
    if (G == 16) {
        Tmax = max(Tmax, __shfl_xor(Tmax, 16));
        Tmax = max(Tmax, __shfl_xor(Tmax, 32));
    }
    Tmax = __builtin_amdgcn_readfirstlane(Tmax);

    uint32_t next1 = (valid && lane == 0) ? (uint32_t)my1[0] : 0u;  // seq1 base of the row this lane does next
    for (int32_t t = 1; t <= Tmax; ++t) {
        const int32_t i = t - (int32_t)lane;  // this lane's row
        const bool active = valid && i >= 1 && i <= nrow;
        const int32_t diag0 = inH;            // H[i-1][jL]: what came in one step ago
        inH = shr1<G>(inH, outH);
        inE = shr1<G>(inE, outE);
        const uint32_t base1 = next1;
        {
            const int32_t ni = i + 1;  // prefetch the base of the next row while this one is computed
            next1 = (valid && ni >= 1 && ni <= nrow) ? (uint32_t)my1[ni - 1] : 0u;
        }
        if (active) {
            int32_t hl = inH, e = inE, hd = diag0, h_ncol = 0;
            uint32_t words[W];
#pragma unroll
            for (uint32_t w = 0; w < W; ++w) words[w] = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int32_t ext_h = e + w_extend, open_h = hl + w_open;
                const bool open_gt_ext_h = open_h > ext_h;
                const int32_t e11 = open_gt_ext_h ? open_h : ext_h;
                const int32_t ext_v = F[k] + w_extend, open_v = Hp[k] + w_open;
                const bool open_gt_ext_v = open_v > ext_v;
                const int32_t f11 = open_gt_ext_v ? open_v : ext_v;
                const int32_t m11 = hd + (base1 == base2[k] ? w_match : w_mismatch);
                int32_t h11 = m11;
                uint32_t flag = (open_gt_ext_h ? 0u : (uint32_t)kBtInsertExt) | (open_gt_ext_v ? 0u : (uint32_t)kBtDeleteExt);
                const bool ins = e11 > h11;
                h11 = ins ? e11 : h11;
                const bool del = f11 > h11;
                h11 = del ? f11 : h11;
                flag |= del ? (uint32_t)kBtDelete : (ins ? (uint32_t)kBtInsert : 0u);
                words[k / 8] |= flag << (4 * (k % 8));
                hd = Hp[k];  // the diagonal of the next column
                h_ncol = (uint32_t)k == kc ? h11 : h_ncol;
                Hp[k] = h11;
                F[k] = f11;
                hl = h11;
                e = e11;
            }
            outH = hl;
            outE = e;
            uint32_t *dst = bt + ((size_t)(t - 1) * G + lane) * W;
#pragma unroll
            for (uint32_t w = 0; w < W; ++w) dst[w] = words[w];
            if (lane == lc) lastcol[i - 1] = h_ncol;  // column ncol, every row
            if (i == nrow) {  // row nrow, every column
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int32_t j = (int32_t)lane * K + k + 1;
                    if (j <= ncol) lastrow[j - 1] = Hp[k];
                }
            }
        }
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

    // ---- traceback and CIGAR (sw_avx.c:256-362), one lane per alignment ----
    if (valid && lane == 0) {
        int32_t i = max_i, j = max_j;
        uint32_t n_el = 0;
        int32_t cur_state = -1, cur_len = 0;  // the element being built; equal neighbours are merged as they come
        auto emit = [&](int32_t state, int32_t len) {
            if (state == cur_state) {
                cur_len += len;
            }
            else {
                if (cur_state >= 0) elems[n_el++] = ((uint32_t)cur_len << 4) | (uint32_t)cur_state;
                cur_state = state;
                cur_len = len;
            }
        };
        if (j < ncol) emit((int32_t)kOpS, ncol - j);
        uint32_t state = 0;
        while (i > 0 && j > 0) {
            const uint32_t l = (uint32_t)(j - 1) / K, k = (uint32_t)(j - 1) % K;
            const uint32_t t = (uint32_t)i + l;
            const uint32_t btr = (bt[((size_t)(t - 1) * G + l) * W + k / 8] >> (4 * (k % 8))) & 15u;
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
                }
            }
        }
        if (j > 0) emit((int32_t)kOpS, j);
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
    const uint32_t per_block = kBlockThreads / G;
    const uint32_t blocks = (a.n_pairs + per_block - 1) / per_block;
    hipLaunchKernelGGL((sw_kernel<K, G>), dim3(blocks), dim3(kBlockThreads), (size_t)per_block * a.lds_stride, s, a);
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
    for (int k : {4, 8, 16, 32})
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
            case 4: return launch_one<4, 64>(a, s);
            case 8: return launch_one<8, 64>(a, s);
            case 16: return launch_one<16, 64>(a, s);
            case 32: return launch_one<32, 64>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace swhip
