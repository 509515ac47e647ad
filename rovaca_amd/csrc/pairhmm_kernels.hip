// pairhmm_kernels.hip -- gfx950 (MI355X) device code for the PairHMM forward recurrence.
//
// What is computed (per pair, identical to the reference's AVX-512 kernels, bit for bit in fp32):
//   rovaca/avx_512_float.cc:29-181 (compute_full_prob_avx512_float_1xn) == core/avx512_s.cpp:259-397
//   (compute_fp_avx512_s), with GCC's FMA contraction of those translation units made explicit:
//     t = Md*p_mm; t = fma(Xd,p_gapm,t); t = fma(Yd,p_gapm,t); M = t*prior
//     X = fma(Xu,p_xx, Mu*p_mx);  Y = fma(Yl,p_yy, Ml*p_my)
//   result = (sum_j M[R][j]) + (sum_j X[R][j]), both sums left to right.
//
// How it is mapped to CDNA4:
//   * a wave64 is four independent 16-lane systolic arrays (one DPP row each); one (read, haplotype) pair
//     per 16 lanes, so one wave sweeps four pairs at a time;
//   * lane l of a group owns K consecutive read rows (K = ceil(R/16), a template parameter), kept entirely
//     in VGPRs: per-row transition/prior constants and the M/X/Y values of the current anti-diagonal;
//   * time step t: lane l computes column t-l of its K rows; the only cross-lane traffic is the last row of
//     lane l-1 (three `v_mov_b32_dpp row_shr:1` per step, i.e. 3/K per cell);
//   * the read-vs-haplotype comparison is bit-parallel: a prologue kernel turns every haplotype into six
//     bit planes ("does column j match read symbol class c"), staged in LDS per pair; each row keeps a
//     32-column window of its plane in one VGPR and consumes it with `v_add_co_u32` (shift + carry-out) and
//     one `v_cndmask_b32` per cell;
//   * rows are right-aligned in the 16*K slots, so the last read row is always slot K-1 of lane 15 and the
//     final sum is accumulated there in the reference's order.  Padding rows above the read reproduce the
//     boundary row (M = X = 0, Y = 2^120/H) exactly;
//   * the same kernel exists with G = 64 lanes per pair (one pair per wave, `wave_shr:1`) for launches of a region or
//     two, where four pairs per wave would leave most SIMDs empty;
//   * reads above 256 rows, haplotypes above 4096 columns and every fp64 re-run take the striped "team" kernel further
//     down: stripes of G*K rows chained through a carry row, several waves per pair as a software pipeline, and
//     supersteps whose inputs and state are all zero are skipped (far from the alignment band everything is).
//
// No MFMA: this is a first-order recurrence, not a contraction.  No CUDA-isms: wave64, DPP, LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "pairhmm_device.h"
#include "pairhmm_systolic.h"

namespace phmm
{

// ----------------------------------------------------------------------------------------------------
// the single-pass systolic kernels, one launch per rows-per-lane class: the sweep itself is the text of pairhmm_systolic_sweep.inc /
// pairhmm_systolic_ugo_sweep.inc (shared with the ragged kernels' sweep functions, pairhmm_systolic.h)
// ----------------------------------------------------------------------------------------------------
template <typename T, int K, bool UG, int G>
__global__ void __launch_bounds__(systolic_block_threads<G>(), G == 64 ? 4 : PHMM_MIN_WAVES(T, K, UG))
    pairhmm_systolic_kernel(KernelArgs a)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & (uint32_t)(G - 1);                      // lane inside the pair's G-lane array
    const uint32_t grp_in_block = threadIdx.x / (uint32_t)G;                    // 0..15 (G = 16), 0 (G = 64)
    const uint32_t gidx = xcd_local_workgroup(blockIdx.x, a.xcd_run_log2) * (systolic_block_threads<G>() / G) + grp_in_block;
    const bool valid = gidx < a.n_pairs;
    const uint32_t pair = a.order[valid ? gidx : 0];
#define PHMM_SWEEP_RESCUE_LIST a.rescue_list
#define PHMM_SWEEP_RESCUE_COUNT a.rescue_count
#include "pairhmm_systolic_sweep.inc"
#undef PHMM_SWEEP_RESCUE_LIST
#undef PHMM_SWEEP_RESCUE_COUNT
}

template <typename T, int K, int G>
__global__ void __launch_bounds__(systolic_block_threads<G>(), G == 64 ? 4 : PHMM_UGO_MIN_WAVES(T, K))
    pairhmm_systolic_ugo_kernel(KernelArgs a)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & (uint32_t)(G - 1);                      // lane inside the pair's G-lane array
    const uint32_t grp_in_block = threadIdx.x / (uint32_t)G;                    // 0..15 (G = 16), 0 (G = 64)
    const uint32_t gidx = xcd_local_workgroup(blockIdx.x, a.xcd_run_log2) * (systolic_block_threads<G>() / G) + grp_in_block;
    const bool valid = gidx < a.n_pairs;
    const uint32_t pair = a.order[valid ? gidx : 0];
#define PHMM_SWEEP_RESCUE_LIST a.rescue_list
#define PHMM_SWEEP_RESCUE_COUNT a.rescue_count
#include "pairhmm_systolic_ugo_sweep.inc"
#undef PHMM_SWEEP_RESCUE_LIST
#undef PHMM_SWEEP_RESCUE_COUNT
}

// ----------------------------------------------------------------------------------------------------
// the striped ("team") systolic kernel: reads longer than 16 * K rows, very long haplotypes, and every fp64 re-run.
//
// The read is cut into stripes of S = 16 * K rows, right-aligned (the boundary replicas sit on top of stripe 0, the last
// read row is slot K-1 of lane 15 of the last stripe).  A stripe is one sweep of H + 15 steps exactly like the single-pass
// kernel; what changes is the row above lane 0: in stripe 0 it is the boundary row, in later stripes the last row of the
// previous stripe, which lane 15 wrote column by column into a carry row in global memory (3 * (H + 1) values, L2 /
// Infinity-Cache resident) -- the analogue of the reference's shift_out_m/x/y arrays (core/avx512_s.cpp:277, :331-333).
// The carry is read back 32 columns at a time (coalesced, two blocks ahead, agent-scope loads so no stale L1 line is
// served), parked in an LDS ring, and lane 0 picks its column up as the `old` operand of the DPP shift.
//
// Multi-wavefront per pair: a workgroup is a TEAM of WP waves that works on the same four pairs.  Wave w owns the stripes
// v = w, w + WP, w + 2 WP, ... and runs kLag supersteps (of 32 steps) behind wave w-1, so the stripes of one pair are swept
// as a software pipeline along the haplotype; one __syncthreads() per superstep orders carry writes before the reads
// that need them.  WP = 1 is the plain striped sweep (no pipeline, the barrier is trivial).  This is what fills the chip
// when a batch has few, long pairs (BASELINE config 5), and it is how the fp64 re-run is done.
// ----------------------------------------------------------------------------------------------------
// supersteps between consecutive stripes of a pair.  The consumer reads three blocks (96 columns) ahead of its first step and
// the producer's last lane trails its first by G - 1 columns, so 32 * lag >= 95 + G: 4 for 16-lane arrays, 5 (we take 6) for 64.
template <int G> constexpr uint32_t team_lag() { return G == 64 ? 6u : 4u; }

// G = lanes per pair: 16 (four pairs per wave, DPP row_shr) or 64 (one pair per wave, DPP wave_shr -- for batches with
// so few pairs that four-per-wave would leave the chip empty: one long-read pair then keeps up to WP waves busy)
template <typename T, int K, bool UG, int WP, int G>
#ifndef PHMM_TEAM_WAVES_F32
#define PHMM_TEAM_WAVES_F32 4
#endif
#ifndef PHMM_TEAM_WAVES_F64
#define PHMM_TEAM_WAVES_F64 2
#endif
__global__ void __launch_bounds__(64 * WP, sizeof(T) == 4 ? PHMM_TEAM_WAVES_F32 : PHMM_TEAM_WAVES_F64) pairhmm_team_kernel(KernelArgs a, T *__restrict__ carry, uint32_t carry_stride)
{
    extern __shared__ uint32_t lds[];
    constexpr int S = G * K;
    constexpr uint32_t kGroups = 64 / G;  // pairs per wave
    constexpr uint32_t kRing = 64;        // columns parked per value: two blocks of 32
    constexpr uint32_t kLag = team_lag<G>();
    const uint32_t lane = threadIdx.x & (G - 1);
    const uint32_t grp = (threadIdx.x & 63u) / G;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t team = blockIdx.x, n_teams = gridDim.x;
    const T *__restrict__ ph2pr = reinterpret_cast<const T *>(a.ph2pr);
    const T *__restrict__ mmtab = reinterpret_cast<const T *>(a.mm_table);
    // LDS: [kGroups][planes] shared by the team, then [WP waves][kGroups][3 values][kRing] carry ring
    const uint32_t plane_words = a.lds_words_per_group;
    uint32_t *pl = lds + grp * plane_words;
    T *ring = reinterpret_cast<T *>(lds + kGroups * plane_words) + (static_cast<size_t>(wave) * kGroups + grp) * 3u * kRing;
    // carry rows: one slab per (team, wave), laid out [column][group][M, X, Y] so that the slab base is wave-uniform (an
    // SGPR pair) and lane 15 writes its three values with one store; stripe v reads the slab of the wave that ran v - 1
    const uint32_t uwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    const uint32_t prev_wave = (uwave + WP - 1) % WP;
    T *wbase = carry + (static_cast<size_t>(team) * WP + uwave) * 3u * kGroups * carry_stride;
    const T *rbase = carry + (static_cast<size_t>(team) * WP + prev_wave) * 3u * kGroups * carry_stride;
    // uniform over the grid (a list with holes, KernelArgs::sparse_order: all n_pairs positions unless nothing was queued)
    const uint32_t n_pairs = a.n_rescue_queues ? 0u : a.n_pairs_dev ? (a.sparse_order ? (*a.n_pairs_dev ? a.n_pairs : 0u) : *a.n_pairs_dev) : a.n_pairs;
    uint32_t quads = (n_pairs + kGroups - 1) / kGroups;  // pairs are taken kGroups at a time
    uint32_t idle_at = 0;  // where a group without a pair of its own reads a VALID pair id: the first entry of a queue that has one
    for (uint32_t q = a.n_rescue_queues; q-- > 0;) {  // (several queues)
        const uint32_t cnt = a.rescue_count[a.rescue_queue[q]];
        quads += (cnt + kGroups - 1) / kGroups;
        if (cnt) idle_at = a.rescue_base[q];
    }
    const uint32_t iters = (quads + n_teams - 1) / n_teams;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t quad = it * n_teams + team, q_pairs = n_pairs, q_base = 0;
        if (a.n_rescue_queues) {  // which queue this group of pairs comes from (scalar; the queues' groups are numbered one after the other)
            q_pairs = 0;
            for (uint32_t q = 0; q < a.n_rescue_queues; ++q) {
                const uint32_t cnt = a.rescue_count[a.rescue_queue[q]], nq = (cnt + kGroups - 1) / kGroups;
                if (quad < nq) {
                    q_pairs = cnt;
                    q_base = a.rescue_base[q];
                    break;
                }
                quad -= nq;
            }
        }
        const uint32_t gidx = quad * kGroups + grp;
        bool valid = gidx < q_pairs;
        uint32_t pair = a.order[valid ? q_base + gidx : idle_at];  // (one queue: idle_at = 0, its first entry)
        if (a.sparse_order && a.n_pairs_dev) {  // the fp64 re-run over a list with holes: a hole is a group without a pair (pair 0 always exists)
            valid = valid && pair != kNoPair;
            pair = valid ? pair : 0u;
        }
        const uint32_t r = a.pair_read[pair], h = a.pair_hap[pair];
        const uint32_t R = a.read_len[r], H = a.hap_len[h];
        const uint32_t roff = a.read_off[r];
        const uint32_t W = planes_words(H);
        __syncthreads();  // the previous quad's planes are no longer read
        {
            const uint32_t *src = a.planes + a.plane_off[h];
            for (uint32_t i = lane + G * wave; i < kPlanes * W; i += G * WP) pl[i] = src[i];
        }
        const T init_y = Num<T>::init_const() / static_cast<T>(H);
        const uint32_t n_stripes = valid ? (R + S - 1) / S : 0;
        const int pad = static_cast<int>(n_stripes) * S - static_cast<int>(R);
        uint32_t max_stripes = n_stripes;  // team-uniform; a pair with fewer stripes joins late so that all end together
        if (kGroups > 1) {
            max_stripes = max(max_stripes, (uint32_t)__shfl_xor((int)max_stripes, 16));
            max_stripes = max(max_stripes, (uint32_t)__shfl_xor((int)max_stripes, 32));
        }
        max_stripes = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_stripes);
        const uint32_t Tpair = valid ? H + (G - 1) : 0;
        uint32_t Tmax = Tpair;
        if (kGroups > 1) {
            Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 16));
            Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 32));
        }
        Tmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmax);
        // Supersteps per stripe.  With several waves the stripes of a pair chase each other around the team: wave 0 starts
        // its next stripe Bs supersteps after the previous one, and that stripe reads the carry of the one wave WP-1 is still
        // sweeping, which started kLag * (WP - 1) supersteps after wave 0 did.  A haplotype shorter than kLag * WP * 32
        // columns (the team shape is chosen for the longest of the launch) gets idle supersteps, or the carry would be read
        // before it is written.
        const uint32_t Bs = max((Tmax + 31u) / 32u, WP > 1 ? kLag * (uint32_t)WP : 1u);
        const uint32_t n_rounds = (max_stripes + WP - 1) / WP;  // stripes per wave
        const uint32_t n_super = n_rounds * Bs + kLag * (WP - 1);  // supersteps of this batch of pairs
        const uint32_t wmax = W - 2;
        __syncthreads();  // planes staged

        // per-stripe state (lives in registers across supersteps)
        Rows<T, K, UG> P;
        T M[K], X[K], Y[K];
        uint32_t mask[K];
        uint32_t cls_lo = 0, cls_hi = 0;
        T aM = 0, aX = 0, aY = 0, bM = 0, bX = 0, bY = 0;
        T preM[2], preX[2], preY[2];
        bool park_nz[2] = {true, true};  // does the parked carry block (ring half) hold anything but zeros
        T curM = 0, curX = 0, curY = 0;  // the row above lane 0 at the next step's column (prefetched from the ring)
        T sumM = 0, sumX = 0;
        uint32_t Tg = 0;
        bool first = false, last = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            M[k] = 0;
            X[k] = 0;
            Y[k] = 0;
            mask[k] = 0;
            P.p_mm[k] = 0;
            P.p_mx[k] = 0;
            P.p_yy[k] = 0;
            P.lo[k] = 0;
            P.hi[k] = 0;
            if (!UG) {
                P.p_my_r[UG ? 0 : k] = 0;
                P.p_gapm_r[UG ? 0 : k] = 0;
                P.p_xx_r[UG ? 0 : k] = 0;
            }
        }
        if (UG) {
            P.p_gapm_r[0] = 0;
            P.p_xx_r[0] = 0;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            preM[q] = 0;
            preX[q] = 0;
            preY[q] = 0;
        }

        auto carry_load = [&](uint32_t blk) {  // columns 32 * blk + 1 + lane (+ 16) of the producer's row
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint32_t j = 32u * blk + 1u + (lane & 15u) + 16u * q;  // G = 64: lanes 16..63 repeat lanes 0..15
                j = j <= H ? j : H;
                if (first) {
                    preM[q] = 0;
                    preX[q] = 0;
                    preY[q] = init_y;
                }
                else {
                    const T *src = rbase + (j * kGroups + grp) * 3u;
                    preM[q] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    preX[q] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    preY[q] = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };
        auto carry_park = [&](uint32_t blk) {
            const uint32_t base = (blk & 1u) * 32u + (lane & 15u);
            bool nz = false;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ring[base + 16 * q] = preM[q];
                ring[kRing + base + 16 * q] = preX[q];
                ring[2 * kRing + base + 16 * q] = preY[q];
                nz = nz || preM[q] != 0 || preX[q] != 0 || preY[q] != 0;
            }
            park_nz[blk & 1u] = nz;  // this lane's share of the block; the groups' lanes are combined when the block is consumed
        };
        T nxtM = 0, nxtX = 0, nxtY = 0;
        auto ring_fetch = [&](uint32_t t) {  // column t -> the "next" registers (issued a whole step before they are used)
            const uint32_t e = (t - 1) & (kRing - 1);
            nxtM = ring[e];
            nxtX = ring[kRing + e];
            nxtY = ring[2 * kRing + e];
        };
        auto masks_reload = [&](uint32_t t) {
            const uint32_t jn = t > lane ? t - lane : 1u;
            uint32_t w = (jn - 1) >> 5;
            const uint32_t sh = (jn - 1) & 31u;
            w = w < wmax ? w : wmax;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t cls = k < 10 ? (cls_lo >> (3 * k)) & 7u : (cls_hi >> (3 * (k - 10))) & 7u;
                const uint32_t *p = pl + cls * W + w;
                const uint32_t w0 = p[0], w1 = p[1];
                mask[k] = (w0 << sh) | ((w1 >> 1) >> (31u - sh));
            }
        };
        auto stripe_setup = [&](uint32_t v) {
            const bool in_play = v + n_stripes >= max_stripes;
            const uint32_t stripe = in_play ? v + n_stripes - max_stripes : 0;  // this pair's own stripe index
            first = stripe == 0;
            last = stripe + 1 == n_stripes;
            const bool live = valid && in_play;
            Tg = live ? Tpair : 0;
            cls_lo = 0;
            cls_hi = 0;
            if (UG) {
                uint32_t c0 = a.read_gcp[roff] & 127u;
                P.p_gapm_r[0] = static_cast<T>(1.0) - ph2pr[c0];
                P.p_xx_r[0] = ph2pr[c0];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int row = static_cast<int>(stripe) * S + static_cast<int>(lane) * K + k - pad;
                uint32_t cls = 5;
                if (live && row >= 0) {
                    const uint32_t at = roff + static_cast<uint32_t>(row);
                    const uint32_t _i = a.read_ins[at] & 127u, _d = a.read_del[at] & 127u, _c = a.read_gcp[at] & 127u,
                                   _q = a.read_qual[at] & 127u;
                    const uint32_t mx = _i > _d ? _i : _d, mn = _i > _d ? _d : _i;
                    P.p_mm[k] = mmtab[((mx * (mx + 1)) >> 1) + mn];
                    P.p_mx[k] = ph2pr[_i];
                    if (!UG) P.p_my_r[UG ? 0 : k] = ph2pr[_d];
                    P.p_yy[k] = ph2pr[_c];
                    if (!UG) {
                        P.p_gapm_r[UG ? 0 : k] = static_cast<T>(1.0) - ph2pr[_c];
                        P.p_xx_r[UG ? 0 : k] = ph2pr[_c];
                    }
                    const T dist = ph2pr[_q];
                    P.hi[k] = static_cast<T>(1.0) - dist;
                    P.lo[k] = Num<T>::div3(dist);
                    cls = read_class(a.read_bases[at]);
                    M[k] = 0;
                    X[k] = 0;
                    Y[k] = 0;
                }
                else {
                    P.p_mm[k] = 0;
                    P.p_mx[k] = 0;
                    if (!UG) P.p_my_r[UG ? 0 : k] = 0;
                    P.p_yy[k] = 1;
                    if (!UG) {
                        P.p_gapm_r[UG ? 0 : k] = 0;
                        P.p_xx_r[UG ? 0 : k] = 0;
                    }
                    P.hi[k] = 0;
                    P.lo[k] = 0;
                    M[k] = 0;
                    X[k] = 0;
                    Y[k] = init_y;
                }
                if (k < 10)
                    cls_lo |= cls << (3 * k);
                else
                    cls_hi |= cls << (3 * (k - 10));
            }
            // column 0 of the row above slot 0: init for the boundary row and its replicas (stripe 0 only), else 0
            const T y_col0 = (first && static_cast<int>(lane) * K <= pad) ? init_y : static_cast<T>(0);
            aM = 0;
            aX = 0;
            aY = y_col0;
            bM = 0;
            bX = 0;
            bY = y_col0;
            if (first) {
                sumM = 0;
                sumX = 0;
            }
            // carry ring: blocks 0 and 1 parked, block 2 in flight
            carry_load(0);
            carry_park(0);
            carry_load(1);
            carry_park(1);
            carry_load(2);
            ring_fetch(1);
            curM = nxtM;
            curX = nxtX;
            curY = nxtY;
        };
        auto step = [&](uint32_t t, T &dM, T &dX, T &dY, T &uM0, T &uX0, T &uY0, auto is_last) {
            // lane 0 keeps `old` = the row above at column t (boundary row or carry); lanes 1..15 take lane-1's last row
            uM0 = dpp_shr1<G>(curM, M[K - 1]);
            uX0 = dpp_shr1<G>(curX, X[K - 1]);
            uY0 = dpp_shr1<G>(curY, Y[K - 1]);
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                uint32_t nm;
                const bool hit = __builtin_uadd_overflow(mask[k], mask[k], &nm);
                mask[k] = nm;
                const T prior = hit ? P.hi[k] : P.lo[k];
                Y[k] = Num<T>::fma_(Y[k], P.p_yy[k], M[k] * P.p_my(k));
                const T gM = k ? M[k ? k - 1 : 0] : dM, gX = k ? X[k ? k - 1 : 0] : dX, gY = k ? Y[k ? k - 1 : 0] : dY;
                T tt = gM * P.p_mm[k];
                tt = Num<T>::fma_(gX, P.p_gapm(k), tt);
                tt = Num<T>::fma_(gY, P.p_gapm(k), tt);
                M[k] = tt * prior;
                __builtin_amdgcn_sched_barrier(0);  // rows in program order, as in the single-pass kernel
            }
            T uM = uM0, uX = uX0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                X[k] = Num<T>::fma_(uX, P.p_xx(k), uM * P.p_mx[k]);
                uM = M[k];
                uX = X[k];
            }
            if (is_last) {
                sumM = sumM + M[K - 1];
                sumX = sumX + X[K - 1];
            }
            else if (lane == G - 1) {
                const uint32_t j = t - (G - 1);  // the last lane is at column t - (G-1) >= 1 whenever it is active
                struct V3
                {
                    T m, x, y;
                };
                *reinterpret_cast<V3 *>(wbase + (j * kGroups + grp) * 3u) = V3{M[K - 1], X[K - 1], Y[K - 1]};
            }
        };
        // 32 steps with every lane of the wave busy: no per-lane predicate, the two register sets swap roles without a move
        auto superstep_full = [&](uint32_t t0, auto is_last) {
#pragma unroll 1
            for (uint32_t t = t0; t < t0 + 32u; t += 2) {
                ring_fetch(t + 1);
                step(t, aM, aX, aY, bM, bX, bY, is_last);
                curM = nxtM;
                curX = nxtX;
                curY = nxtY;
                ring_fetch(t + 2);
                step(t + 1, bM, bX, bY, aM, aX, aY, is_last);
                curM = nxtM;
                curX = nxtX;
                curY = nxtY;
            }
        };

        for (uint32_t g = 0; g < n_super; ++g) {
            const int32_t tau = static_cast<int32_t>(g) - static_cast<int32_t>(kLag * wave);
            if (tau >= 0 && static_cast<uint32_t>(tau) < n_rounds * Bs) {
                const uint32_t round = static_cast<uint32_t>(tau) / Bs, c = static_cast<uint32_t>(tau) % Bs;
                const uint32_t v = round * WP + wave;
                if (v < max_stripes) {
                    if (c == 0) {
                        stripe_setup(v);
                    }
                    else {
                        carry_park(c + 1);  // block c + 1 joins the ring (block c was parked one superstep ago)
                        carry_load(c + 2);
                    }
                    const uint32_t t0 = 32u * c + 1u;
                    const uint32_t t1 = min(t0 + 31u, Tmax);
                    uint32_t Tmin = Tg;  // wave-uniform: the last step at which every group of the wave is still sweeping
                    if (kGroups > 1) {
                        Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 16));
                        Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 32));
                    }
                    Tmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmin);
                    // A superstep with nothing but zeros going in -- the carry block it consumes and every value the wave
                    // holds -- puts nothing but zeros out (each value is a sum of non-negative products of its neighbours,
                    // and fp32 runs flush-to-zero): write the zero carry columns and move on.  This is where long reads
                    // spend most of their fp32 pass: a 10-kbp read underflows for good after some 2 000 rows.
                    bool lane_nz = park_nz[c & 1u] || aM != 0 || aX != 0 || aY != 0 || bM != 0 || bX != 0 || bY != 0;
#pragma unroll
                    for (int k = 0; k < K; ++k) lane_nz = lane_nz || M[k] != 0 || X[k] != 0 || Y[k] != 0;
                    if (__ballot(lane_nz && Tg != 0) == 0 && !a.no_early_exit) {
                        if (Tg != 0 && v + 1 != max_stripes) {
                            struct V3
                            {
                                T m, x, y;
                            };
                            for (uint32_t t = t0 + lane; t <= t1 && t <= Tg; t += G) {
                                if (t > (uint32_t)(G - 1)) *reinterpret_cast<V3 *>(wbase + ((t - (G - 1)) * kGroups + grp) * 3u) = V3{0, 0, 0};
                            }
                        }
                        ring_fetch(t1 + 1);
                        curM = nxtM;
                        curX = nxtX;
                        curY = nxtY;
                    }
                    else {
                        masks_reload(t0);
                        if (t0 > (uint32_t)(G - 1) && t0 + 31u <= Tmin) {
                            // steady state (t0 is odd).  Every group is live, so `last` is the same for the whole wave.
                            if (v + 1 == max_stripes)
                                superstep_full(t0, std::true_type{});
                            else
                                superstep_full(t0, std::false_type{});
                        }
                        else {
                            for (uint32_t t = t0; t <= t1; ++t) {
                                ring_fetch(t + 1);  // LDS latency hides behind this step's arithmetic
                                if ((t > lane) && (t <= Tg)) {
                                    if (t & 1u)
                                        step(t, aM, aX, aY, bM, bX, bY, last);
                                    else
                                        step(t, bM, bX, bY, aM, aX, aY, last);
                                }
                                curM = nxtM;
                                curX = nxtX;
                                curY = nxtY;
                            }
                        }
                    }
                    if (c + 1 == Bs && v + 1 == max_stripes && valid && lane == G - 1) {
                        const T res = sumM + sumX;
                        reinterpret_cast<T *>(a.out_raw)[pair] = res;
                        // fp32: underflowed, to the fp64 re-run; fp64 (only the fp64-first launch has a list): fp32 might not underflow
                        if (a.rescue_list && (sizeof(T) == 4 ? res < static_cast<T>(1e-28f) : static_cast<double>(res) >= a.need32_above)) {
                            const uint32_t at = atomicAdd(a.rescue_count, 1u);
                            a.rescue_list[a.sparse_order ? gidx : at] = pair;  // (sparse_order: at the pair's own position of the launch)
                        }
                    }
                }
            }
            __syncthreads();  // carry columns written in this superstep are visible to the team before the next one
        }
    }
}

// ----------------------------------------------------------------------------------------------------
// the band kernel: teams of SEVERAL waves for few, long pairs (BASELINE config 5).  Same stripes, same carry rows, same cell
// update as the team kernel above -- what differs is how the waves of a team are driven.  A long read against a long haplotype
// is a serial walk of the alignment band down the stripes: a batch of a thousand such pairs puts one or two busy waves on a
// SIMD, and the pass takes as long as its longest pair.  So (1) the waves of a team run at their own pace, synchronised
// through four words of LDS per wave instead of a barrier per superstep, and (2) a stripe sweeps only its band: it starts at the
// first carry block the stripe above put anything into and stops when that stripe is finished and the state is zero again.
// ----------------------------------------------------------------------------------------------------
// the last carry block a stripe can have put anything into when its last non-skipped superstep was l: the last lane ends that
// superstep at column 32 l + 32 - (G - 1)
template <int G> __device__ __forceinline__ uint32_t team_last_block(uint32_t l) { return G == 64 ? (l > 0 ? l - 1u : 0u) : l; }

// Rows in program order (a scheduling barrier after each row, as in the single-pass and team kernels) pay in the issue-bound regime,
// several waves per SIMD.  Here a wave or two share a SIMD and each row's dependent mul/fma chain sets the pace: the kernel runs
// faster when the compiler may interleave the rows (config 5: fp32 pass 6.6 -> 5.3 ms).
#ifndef PHMM_BAND_ROW_BARRIER
#define PHMM_BAND_ROW_BARRIER 0
#endif
#ifndef PHMM_BAND_PREFETCH_BLOCKS
#define PHMM_BAND_PREFETCH_BLOCKS 2  // 1: the shorter hand-over lag tried in round 4 (tools/build_variant.sh for the A/B): no gain, see kPrefetch
#endif
// TABP (round 5; fp64, 4 rows per lane, one pair per wave): the prior is looked up in a table in LDS instead of selected by a match bit
// -- the single-pass table sweep's idea (pairhmm_systolic_tab.h) where it pays most: an fp64 select is a bit pop and TWO 32-bit
// selects, three VALU instructions per cell beside nine fp64 ones.  Each wave keeps the table of the stripe it sweeps (its own 256
// rows: a lane writes and reads only its own four rows' entries, so no barrier is involved): [half: rows 0-1 / 2-3][symbol][lane] x 16
// bytes, SIX symbols -- A, C, T, G (ASCII bits 1-2), N (matches every base and N), anything else (matches nothing:
// pairhmm_internal.cpp:27-41) -- so haplotypes with N need no other path here.  The haplotype is staged once per pair as one 16-bit
// entry per column (symbol x 1024), 64 entries in front for the columns a lane "sees" before its first.
template <typename T, int K, bool UG, int WP, int G, bool TABP = false>
__global__ void __launch_bounds__(64 * WP, sizeof(T) == 4 ? PHMM_TEAM_WAVES_F32 : PHMM_TEAM_WAVES_F64) pairhmm_band_kernel(KernelArgs a, T *__restrict__ carry, uint32_t carry_stride)
{
    static_assert(!TABP || (sizeof(T) == 8 && K == 4 && G == 64), "the table form is the fp64 band kernel's");
    extern __shared__ uint32_t lds[];
    constexpr int S = G * K;
    constexpr uint32_t kGroups = 64 / G;  // pairs per wave
    constexpr uint32_t kRing = 64;        // columns parked per value: two blocks of 32
    constexpr uint32_t kLag = team_lag<G>();
    // How far ahead of the block a superstep consumes its carry loads are issued.  Two blocks (load block c + 2, park block c + 1,
    // consume block c) hide the L2 round trip twice over, and every block of look-ahead is a superstep of lag between consecutive
    // stripes of a pair.  Round 4 measured the chain on one 10 000 x 12 000 pair (tools/band_timeline.py, profiles/r4_band_timeline_*):
    // consecutive stripes of the band begin 12.9 supersteps apart -- 8.4 because the band moves that far per 256 rows, 4.5 of hand-over
    // -- and tried one block of look-ahead (= 1: loads issued when the superstep begins, parked half way through it, two blocks only
    // when the stripe above is far ahead): the stripes then begin 12.8 supersteps apart, but the superstep itself got 11 % slower
    // (5.64 -> 6.25 us; the loop with the mid-way park allocates worse: 251 registers against 237) and config 5 went 12.4 -> 13.3 ms.
    // Kept as a build variant, not the default.
    constexpr uint32_t kPrefetch = PHMM_BAND_PREFETCH_BLOCKS;
    const uint32_t lane = threadIdx.x & (G - 1);
    const uint32_t grp = (threadIdx.x & 63u) / G;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t team = blockIdx.x, n_teams = gridDim.x;
    const T *__restrict__ ph2pr = reinterpret_cast<const T *>(a.ph2pr);
    const T *__restrict__ mmtab = reinterpret_cast<const T *>(a.mm_table);
    // LDS: [kGroups][planes] shared by the team, then [WP waves][kGroups][3 values][kRing] carry ring
    // (table form: [haplotype entries][WP tables] instead of the planes, sized by band_tab_front_bytes)
    const uint32_t plane_words = TABP ? band_tab_front_bytes(a.tab_max_hap, WP) / 4u : a.lds_words_per_group;
    uint32_t *pl = lds + grp * plane_words;
    uint8_t *const tab_hap = reinterpret_cast<uint8_t *>(lds);
    uint8_t *const tab_mine = reinterpret_cast<uint8_t *>(lds) + band_tab_hap_bytes(a.tab_max_hap) + wave * kBandTabBytes + (threadIdx.x & 63u) * 16u;
    T *ring = reinterpret_cast<T *>(lds + kGroups * plane_words) + (static_cast<size_t>(wave) * kGroups + grp) * 3u * kRing;
    // progress records of the team's waves: [round parity][wave][stripe + 1, supersteps done, first carry block with output, last
    // carry block with output (once the stripe is finished), superstep the sweep began at, "a wait timed out somewhere up the
    // chain" flag, 2 spare words]
    uint32_t *team_rec = reinterpret_cast<uint32_t *>(reinterpret_cast<T *>(lds + kGroups * plane_words) + static_cast<size_t>(WP) * kGroups * 3u * kRing);
    // carry rows: one slab per (team, wave), laid out [column][group][M, X, Y] so that the slab base is wave-uniform (an
    // SGPR pair) and lane 15 writes its three values with one store; stripe v reads the slab of the wave that ran v - 1
    // The wave's ROLE in the team (which stripes it sweeps) is its hardware index rotated by the team's: a workgroup's waves go to the
    // CU's four SIMDs in order, so without the rotation the waves that sweep stripe 0 of every pair -- the only busy ones while the
    // fp32 pass walks its full-width first stripe -- all sat on SIMD 0 of their CUs, four to a SIMD, while SIMDs 1-3 slept (round 4:
    // 350 cycles per 35-instruction step).  PHMM_BAND_NO_ROLE_ROTATION: A/B build hook.
#ifdef PHMM_BAND_NO_ROLE_ROTATION
    const uint32_t uwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
#else
    const uint32_t uwave = ((uint32_t)__builtin_amdgcn_readfirstlane((int)wave) + team) % WP;
#endif
    const uint32_t prev_wave = (uwave + WP - 1) % WP;
    T *wbase = carry + (static_cast<size_t>(team) * WP + uwave) * 3u * kGroups * carry_stride;
    const T *rbase = carry + (static_cast<size_t>(team) * WP + prev_wave) * 3u * kGroups * carry_stride;
    // uniform over the grid (a list with holes, KernelArgs::sparse_order: all n_pairs positions unless nothing was queued)
    const uint32_t n_pairs = a.n_rescue_queues ? 0u : a.n_pairs_dev ? (a.sparse_order ? (*a.n_pairs_dev ? a.n_pairs : 0u) : *a.n_pairs_dev) : a.n_pairs;
    uint32_t quads = (n_pairs + kGroups - 1) / kGroups;  // pairs are taken kGroups at a time
    uint32_t idle_at = 0;  // where a group without a pair of its own reads a VALID pair id: the first entry of a queue that has one
    for (uint32_t q = a.n_rescue_queues; q-- > 0;) {  // (several queues)
        const uint32_t cnt = a.rescue_count[a.rescue_queue[q]];
        quads += (cnt + kGroups - 1) / kGroups;
        if (cnt) idle_at = a.rescue_base[q];
    }
    const uint32_t iters = (quads + n_teams - 1) / n_teams;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t quad = it * n_teams + team, q_pairs = n_pairs, q_base = 0;
        if (a.n_rescue_queues) {  // which queue this group of pairs comes from (scalar; the queues' groups are numbered one after the other)
            q_pairs = 0;
            for (uint32_t q = 0; q < a.n_rescue_queues; ++q) {
                const uint32_t cnt = a.rescue_count[a.rescue_queue[q]], nq = (cnt + kGroups - 1) / kGroups;
                if (quad < nq) {
                    q_pairs = cnt;
                    q_base = a.rescue_base[q];
                    break;
                }
                quad -= nq;
            }
        }
        const uint32_t gidx = quad * kGroups + grp;
        bool valid = gidx < q_pairs;
        uint32_t pair = a.order[valid ? q_base + gidx : idle_at];  // (one queue: idle_at = 0, its first entry)
        if (a.sparse_order && a.n_pairs_dev) {  // the fp64 re-run over a list with holes: a hole is a group without a pair (pair 0 always exists)
            valid = valid && pair != kNoPair;
            pair = valid ? pair : 0u;
        }
        const uint32_t r = a.pair_read[pair], h = a.pair_hap[pair];
        const uint32_t R = a.read_len[r], H = a.hap_len[h];
        const uint32_t roff = a.read_off[r];
        const uint32_t W = planes_words(H);
        __syncthreads();  // the previous quad's planes are no longer read
        if constexpr (TABP) {
            // entry 64 + d + (j - 1) for column j (d = the haplotype's offset in its dword); everything that is not a column: symbol 5
            const uint32_t hoff = a.hap_off[h];
            const uint32_t d = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(a.hap_bases) + hoff) & 3u);
            const uint8_t *__restrict__ src = a.hap_bases + hoff - d;
            constexpr uint32_t kNone = 5u * 1024u;
            const uint32_t n_dwords = band_tab_hap_bytes(a.tab_max_hap) / 8u - 16u;  // entries / 4 behind the 64 in front
            for (uint32_t i = threadIdx.x; i < 16u; i += 64u * WP) reinterpret_cast<uint2 *>(tab_hap)[i] = uint2{kNone | (kNone << 16), kNone | (kNone << 16)};
            for (uint32_t i = threadIdx.x; i < n_dwords; i += 64u * WP) {
                const uint32_t b0 = 4u * i;
                uint32_t e[4] = {kNone, kNone, kNone, kNone};
                if (b0 + 3u >= d && b0 < d + H) {
                    const uint32_t w = *reinterpret_cast<const uint32_t *>(src + b0);
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) {
                        const uint32_t b = (w >> (8u * q)) & 255u;
                        const uint32_t c = (b >> 1) & 3u;  // A 0, C 1, T 2, G 3
                        const uint32_t sym = b == ((0x47544341u >> (8u * c)) & 255u) ? c : b == 'N' ? 4u : 5u;
                        if (b0 + q >= d && b0 + q < d + H) e[q] = sym * 1024u;
                    }
                }
                reinterpret_cast<uint2 *>(tab_hap)[16u + i] = uint2{e[0] | (e[1] << 16), e[2] | (e[3] << 16)};
            }
        }
        else {
            const uint32_t *src = a.planes + a.plane_off[h];
            for (uint32_t i = lane + G * wave; i < kPlanes * W; i += G * WP) pl[i] = src[i];
        }
        const uint32_t tab_d = TABP ? static_cast<uint32_t>((reinterpret_cast<uintptr_t>(a.hap_bases) + a.hap_off[h]) & 3u) : 0u;
        // the entry of this lane's column at step t is tab_hp[t]: entry 64 + d + (t - lane) - 1
        const uint16_t *const tab_hp = reinterpret_cast<const uint16_t *>(tab_hap) + (63u + tab_d - lane);
        if (WP > 1 && threadIdx.x < 2u * WP * 8u) team_rec[threadIdx.x] = 0;  // no wave has begun any stripe of these pairs
        const T init_y = Num<T>::init_const() / static_cast<T>(H);
        const uint32_t n_stripes = valid ? (R + S - 1) / S : 0;
        const int pad = static_cast<int>(n_stripes) * S - static_cast<int>(R);
        uint32_t max_stripes = n_stripes;  // team-uniform; a pair with fewer stripes joins late so that all end together
        if (kGroups > 1) {
            max_stripes = max(max_stripes, (uint32_t)__shfl_xor((int)max_stripes, 16));
            max_stripes = max(max_stripes, (uint32_t)__shfl_xor((int)max_stripes, 32));
        }
        max_stripes = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_stripes);
        const uint32_t Tpair = valid ? H + (G - 1) : 0;
        uint32_t Tmax = Tpair;
        if (kGroups > 1) {
            Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 16));
            Tmax = max(Tmax, (uint32_t)__shfl_xor((int)Tmax, 32));
        }
        Tmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmax);
        // Supersteps per stripe.  With several waves the stripes of a pair chase each other around the team: wave 0 starts
        // its next stripe Bs supersteps after the previous one, and that stripe reads the carry of the one wave WP-1 is still
        // sweeping, which started kLag * (WP - 1) supersteps after wave 0 did.  A haplotype shorter than kLag * WP * 32
        // columns (the team shape is chosen for the longest of the launch) gets idle supersteps, or the carry would be read
        // before it is written.
        const uint32_t Bs = max((Tmax + 31u) / 32u, WP > 1 ? kLag * (uint32_t)WP : 1u);
        const uint32_t n_rounds = (max_stripes + WP - 1) / WP;  // stripes per wave
        const uint32_t n_super = n_rounds * Bs + kLag * (WP - 1);  // supersteps of this batch of pairs
        const uint32_t wmax = W - 2;
        __syncthreads();  // planes staged

        // per-stripe state (lives in registers across supersteps)
        Rows<T, K, UG> P;
        T M[K], X[K], Y[K];
        uint32_t mask[K];
        T pr[TABP ? K : 1];     // table form: the priors of the step in hand ...
        uint32_t so_next = 0;   // ... the haplotype entry of the next step ...
        uint32_t tab_at = 0;    // ... and which step that is (0: none -- a new stripe's table, or supersteps were skipped)
        uint32_t cls_lo = 0, cls_hi = 0;
        T aM = 0, aX = 0, aY = 0, bM = 0, bX = 0, bY = 0;
        T preM[2], preX[2], preY[2];
        bool park_nz[2] = {true, true};  // does the parked carry block (ring half) hold anything but zeros
        T curM = 0, curX = 0, curY = 0;  // the row above lane 0 at the next step's column (prefetched from the ring)
        T sumM = 0, sumX = 0;
        uint32_t Tg = 0;
        bool first = false, last = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            M[k] = 0;
            X[k] = 0;
            Y[k] = 0;
            mask[k] = 0;
            P.p_mm[k] = 0;
            P.p_mx[k] = 0;
            P.p_yy[k] = 0;
            P.lo[k] = 0;
            P.hi[k] = 0;
            if (!UG) {
                P.p_my_r[UG ? 0 : k] = 0;
                P.p_gapm_r[UG ? 0 : k] = 0;
                P.p_xx_r[UG ? 0 : k] = 0;
            }
        }
        if (UG) {
            P.p_gapm_r[0] = 0;
            P.p_xx_r[0] = 0;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            preM[q] = 0;
            preX[q] = 0;
            preY[q] = 0;
        }

        uint32_t carry_last_blk = 0xFFFFFFFFu;  // the producer stripe wrote nothing (i.e. zeros) behind this block; unknown: all ones
        uint32_t carry_first_blk = 0;           // ... and nothing at all, not even zeros, before this one (it began its sweep there)
        auto carry_load = [&](uint32_t blk) {  // columns 32 * blk + 1 + lane (+ 16) of the producer's row
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint32_t j = 32u * blk + 1u + (lane & 15u) + 16u * q;  // G = 64: lanes 16..63 repeat lanes 0..15
                j = j <= H ? j : H;
                if (first) {
                    preM[q] = 0;
                    preX[q] = 0;
                    preY[q] = init_y;
                }
                else if (blk > carry_last_blk || blk < carry_first_blk) {
                    preM[q] = 0;
                    preX[q] = 0;
                    preY[q] = 0;
                }
                else {
                    const T *src = rbase + (j * kGroups + grp) * 3u;
                    preM[q] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    preX[q] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    preY[q] = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };
        auto carry_park = [&](uint32_t blk) {
            const uint32_t base = (blk & 1u) * 32u + (lane & 15u);
            bool nz = false;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ring[base + 16 * q] = preM[q];
                ring[kRing + base + 16 * q] = preX[q];
                ring[2 * kRing + base + 16 * q] = preY[q];
                nz = nz || preM[q] != 0 || preX[q] != 0 || preY[q] != 0;
            }
            park_nz[blk & 1u] = nz;  // this lane's share of the block; the groups' lanes are combined when the block is consumed
        };
        T nxtM = 0, nxtX = 0, nxtY = 0;
        auto ring_fetch = [&](uint32_t t) {  // column t -> the "next" registers (issued a whole step before they are used)
            const uint32_t e = (t - 1) & (kRing - 1);
            nxtM = ring[e];
            nxtX = ring[kRing + e];
            nxtY = ring[2 * kRing + e];
        };
        auto masks_reload = [&](uint32_t t) {
            const uint32_t jn = t > lane ? t - lane : 1u;
            uint32_t w = (jn - 1) >> 5;
            const uint32_t sh = (jn - 1) & 31u;
            w = w < wmax ? w : wmax;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t cls = k < 10 ? (cls_lo >> (3 * k)) & 7u : (cls_hi >> (3 * (k - 10))) & 7u;
                const uint32_t *p = pl + cls * W + w;
                const uint32_t w0 = p[0], w1 = p[1];
                mask[k] = (w0 << sh) | ((w1 >> 1) >> (31u - sh));
            }
        };
        auto stripe_setup = [&](uint32_t v) {
            const bool in_play = v + n_stripes >= max_stripes;
            const uint32_t stripe = in_play ? v + n_stripes - max_stripes : 0;  // this pair's own stripe index
            first = stripe == 0;
            last = stripe + 1 == n_stripes;
            const bool live = valid && in_play;
            Tg = live ? Tpair : 0;
            tab_at = 0;  // (table form: this stripe's table is about to be written)
            cls_lo = 0;
            cls_hi = 0;
            if (UG) {
                uint32_t c0 = a.read_gcp[roff] & 127u;
                P.p_gapm_r[0] = static_cast<T>(1.0) - ph2pr[c0];
                P.p_xx_r[0] = ph2pr[c0];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int row = static_cast<int>(stripe) * S + static_cast<int>(lane) * K + k - pad;
                uint32_t cls = 5;
                if (live && row >= 0) {
                    const uint32_t at = roff + static_cast<uint32_t>(row);
                    const uint32_t _i = a.read_ins[at] & 127u, _d = a.read_del[at] & 127u, _c = a.read_gcp[at] & 127u,
                                   _q = a.read_qual[at] & 127u;
                    const uint32_t mx = _i > _d ? _i : _d, mn = _i > _d ? _d : _i;
                    P.p_mm[k] = mmtab[((mx * (mx + 1)) >> 1) + mn];
                    P.p_mx[k] = ph2pr[_i];
                    if (!UG) P.p_my_r[UG ? 0 : k] = ph2pr[_d];
                    P.p_yy[k] = ph2pr[_c];
                    if (!UG) {
                        P.p_gapm_r[UG ? 0 : k] = static_cast<T>(1.0) - ph2pr[_c];
                        P.p_xx_r[UG ? 0 : k] = ph2pr[_c];
                    }
                    const T dist = ph2pr[_q];
                    P.hi[k] = static_cast<T>(1.0) - dist;
                    P.lo[k] = Num<T>::div3(dist);
                    cls = read_class(a.read_bases[at]);
                    M[k] = 0;
                    X[k] = 0;
                    Y[k] = 0;
                }
                else {
                    P.p_mm[k] = 0;
                    P.p_mx[k] = 0;
                    if (!UG) P.p_my_r[UG ? 0 : k] = 0;
                    P.p_yy[k] = 1;
                    if (!UG) {
                        P.p_gapm_r[UG ? 0 : k] = 0;
                        P.p_xx_r[UG ? 0 : k] = 0;
                    }
                    P.hi[k] = 0;
                    P.lo[k] = 0;
                    M[k] = 0;
                    X[k] = 0;
                    Y[k] = init_y;
                }
                if (k < 10)
                    cls_lo |= cls << (3 * k);
                else
                    cls_hi |= cls << (3 * (k - 10));
                if constexpr (TABP) {
                    // this row's six priors: one-hot semantics (pairhmm_internal.cpp:27-41) -- symbol codes A 0, C 1, T 2, G 3, N 4, other 5
                    // against the read's class A 0, C 1, G 2, T 3, N 4, other 5
#pragma unroll
                    for (uint32_t sy = 0; sy < 6; ++sy) {
                        const uint32_t hc = sy == 2u ? 3u : sy == 3u ? 2u : sy;  // the symbol as a read class
                        const bool m = cls != 5u && sy != 5u && (cls == 4u || sy == 4u || cls == hc);
                        reinterpret_cast<T *>(tab_mine + (k >> 1) * (kBandTabBytes / 2) + sy * 1024u)[k & 1] = m ? P.hi[k] : P.lo[k];
                    }
                }
            }
            // column 0 of the row above slot 0: init for the boundary row and its replicas (stripe 0 only), else 0
            const T y_col0 = (first && static_cast<int>(lane) * K <= pad) ? init_y : static_cast<T>(0);
            aM = 0;
            aX = 0;
            aY = y_col0;
            bM = 0;
            bX = 0;
            bY = y_col0;
            if (first) {
                sumM = 0;
                sumX = 0;
            }
        };
        // carry ring at the first superstep c0 of a stripe: block c0 parked, block c0 + 1 in flight (PHMM_BAND_PREFETCH_BLOCKS = 1), or
        // blocks c0 and c0 + 1 parked and block c0 + 2 in flight (= 2, the form until round 3)
        bool inflight = false;  // kPrefetch == 1: the block after the one being consumed is already in the pre-registers
        auto ring_init = [&](uint32_t c0) {
            carry_load(c0);
            carry_park(c0);
            if (kPrefetch == 2u) {
                carry_load(c0 + 1);
                carry_park(c0 + 1);
                carry_load(c0 + 2);
            }
            inflight = false;
            ring_fetch(32u * c0 + 1u);
            curM = nxtM;
            curX = nxtX;
            curY = nxtY;
        };
        auto tab_fetch = [&](uint32_t so) {
            if constexpr (TABP) {
#pragma unroll
                for (int hf = 0; hf < K / 2; ++hf) {
                    const double2 v = *reinterpret_cast<const double2 *>(tab_mine + hf * (kBandTabBytes / 2) + so);
                    pr[2 * hf] = v.x;
                    pr[2 * hf + 1] = v.y;
                }
            }
        };
        auto tab_prime = [&](uint32_t t) {  // before the first step t of a superstep that is swept
            if constexpr (TABP) {
                if (tab_at == t) return;  // the last step of the superstep before fetched them already
                tab_fetch(tab_hp[t]);
                so_next = tab_hp[t + 1];
                tab_at = t;
            }
        };
        auto tab_next = [&](uint32_t t) {  // behind pass 1 of step t: the priors of step t + 1 into the same registers
            if constexpr (TABP) {
                tab_fetch(so_next);
                so_next = tab_hp[t + 2];
                tab_at = t + 1;
            }
        };
        auto step_a = [&](uint32_t t, T &dM, T &dX, T &dY, T &uM0, T &uX0, T &uY0) {
            // lane 0 keeps `old` = the row above at column t (boundary row or carry); lanes 1..15 take lane-1's last row
            uM0 = dpp_shr1<G>(curM, M[K - 1]);
            uX0 = dpp_shr1<G>(curX, X[K - 1]);
            uY0 = dpp_shr1<G>(curY, Y[K - 1]);
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                T prior;
                if constexpr (TABP)
                    prior = pr[k];
                else {
                    uint32_t nm;
                    const bool hit = __builtin_uadd_overflow(mask[k], mask[k], &nm);
                    mask[k] = nm;
                    prior = hit ? P.hi[k] : P.lo[k];
                }
                Y[k] = Num<T>::fma_(Y[k], P.p_yy[k], M[k] * P.p_my(k));
                const T gM = k ? M[k ? k - 1 : 0] : dM, gX = k ? X[k ? k - 1 : 0] : dX, gY = k ? Y[k ? k - 1 : 0] : dY;
                T tt = gM * P.p_mm[k];
                tt = Num<T>::fma_(gX, P.p_gapm(k), tt);
                tt = Num<T>::fma_(gY, P.p_gapm(k), tt);
                M[k] = tt * prior;
                if (PHMM_BAND_ROW_BARRIER) __builtin_amdgcn_sched_barrier(0);  // rows in program order: see the macro
            }
        };
        auto step_b = [&](uint32_t t, T &uM0, T &uX0, auto is_last) {
            T uM = uM0, uX = uX0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                X[k] = Num<T>::fma_(uX, P.p_xx(k), uM * P.p_mx[k]);
                uM = M[k];
                uX = X[k];
            }
            if (is_last) {
                sumM = sumM + M[K - 1];
                sumX = sumX + X[K - 1];
            }
            else if (lane == G - 1) {
                const uint32_t j = t - (G - 1);  // the last lane is at column t - (G-1) >= 1 whenever it is active
                struct V3
                {
                    T m, x, y;
                };
                *reinterpret_cast<V3 *>(wbase + (j * kGroups + grp) * 3u) = V3{M[K - 1], X[K - 1], Y[K - 1]};
            }
        };
        auto step = [&](uint32_t t, T &dM, T &dX, T &dY, T &uM0, T &uX0, T &uY0, auto is_last) {
            step_a(t, dM, dX, dY, uM0, uX0, uY0);
            tab_next(t);
            step_b(t, uM0, uX0, is_last);
        };
        // 32 steps with every lane of the wave busy: no per-lane predicate, the two register sets swap roles without a move; `mid` runs
        // once, half way (the late park of the next carry block)
        auto superstep_full = [&](uint32_t t0, auto is_last, auto &&mid) {
#pragma unroll 1
            for (uint32_t t = t0; t < t0 + 32u; t += 2) {
                if (t == t0 + 16u) mid();
                ring_fetch(t + 1);
                step(t, aM, aX, aY, bM, bX, bY, is_last);
                curM = nxtM;
                curX = nxtX;
                curY = nxtY;
                ring_fetch(t + 2);
                step(t + 1, bM, bX, bY, aM, aX, aY, is_last);
                curM = nxtM;
                curX = nxtX;
                curY = nxtY;
            }
        };

        // one superstep (32 columns) of stripe v; returns true when it was skipped as all-zero
        // far: the stripe above has block c + 2 complete as well (it is finished, or far enough ahead) -- then that block is loaded a
        // whole superstep before it is parked, as until round 3: a stripe that walks behind a finished one (the rows under the top
        // strip, the skipped supersteps at a band's edges) does not pay an L2 round trip per superstep.  A stripe that follows its
        // producer at the minimum distance -- the steady walk of the band -- never sees `far` and keeps the short lag.
        auto superstep = [&](uint32_t v, uint32_t c, uint32_t c_first, bool far) -> bool {
            bool park_late = false;  // block c + 1 is parked half way through this superstep (it is being loaded now)
            if (c == c_first) ring_init(c);
            if (kPrefetch == 2u) {
                if (c != c_first) {
                    carry_park(c + 1);  // block c + 1 joins the ring (block c was parked one superstep ago)
                    carry_load(c + 2);
                }
            }
            else if (inflight) {
                carry_park(c + 1);  // loaded during the last superstep
                inflight = far;
                if (far) carry_load(c + 2);
            }
            else {
                carry_load(c + 1);
                park_late = true;
            }
            auto late_park = [&]() {
                carry_park(c + 1);
                inflight = far;
                if (far) carry_load(c + 2);
            };
            const uint32_t t0 = 32u * c + 1u;
            const uint32_t t1 = min(t0 + 31u, Tmax);
            uint32_t Tmin = Tg;  // wave-uniform: the last step at which every group of the wave is still sweeping
            if (kGroups > 1) {
                Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 16));
                Tmin = min(Tmin, (uint32_t)__shfl_xor((int)Tmin, 32));
            }
            Tmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tmin);
            // A superstep with nothing but zeros going in -- the carry block it consumes and every value the wave
            // holds -- puts nothing but zeros out (each value is a sum of non-negative products of its neighbours,
            // and fp32 runs flush-to-zero): write the zero carry columns and move on.  This is where long reads
            // spend most of their fp32 pass: a 10-kbp read underflows for good after some 2 000 rows.
            bool lane_nz = park_nz[c & 1u] || aM != 0 || aX != 0 || aY != 0 || bM != 0 || bX != 0 || bY != 0;
#pragma unroll
            for (int k = 0; k < K; ++k) lane_nz = lane_nz || M[k] != 0 || X[k] != 0 || Y[k] != 0;
            bool skipped = false;
            if (__ballot(lane_nz && Tg != 0) == 0 && !a.no_early_exit) {
                skipped = true;
                if (Tg != 0 && v + 1 != max_stripes) {
                    struct V3
                    {
                        T m, x, y;
                    };
                    for (uint32_t t = t0 + lane; t <= t1 && t <= Tg; t += G) {
                        if (t > (uint32_t)(G - 1)) *reinterpret_cast<V3 *>(wbase + ((t - (G - 1)) * kGroups + grp) * 3u) = V3{0, 0, 0};
                    }
                }
                if (park_late) late_park();  // (waits for the load: a stripe this close behind its producer is waiting for it anyway)
                ring_fetch(t1 + 1);
                curM = nxtM;
                curX = nxtX;
                curY = nxtY;
            }
            else {
                if constexpr (TABP)
                    tab_prime(t0);
                else
                    masks_reload(t0);
                if (t0 > (uint32_t)(G - 1) && t0 + 31u <= Tmin) {
                    // steady state (t0 is odd).  Every group is live, so `last` is the same for the whole wave.
                    auto mid = [&]() {
                        if (park_late) late_park();  // its loads were issued 16 steps (3 us) ago; the last step of this superstep already
                    };                               // fetches the block's first column
                    if (v + 1 == max_stripes)
                        superstep_full(t0, std::true_type{}, mid);
                    else
                        superstep_full(t0, std::false_type{}, mid);
                }
                else {
                    bool parked = !park_late;
                    for (uint32_t t = t0; t <= t1; ++t) {
                        if (!parked && t == t0 + 16u) {
                            late_park();
                            parked = true;
                        }
                        ring_fetch(t + 1);  // LDS latency hides behind this step's arithmetic
                        const bool lv = (t > lane) && (t <= Tg);
                        if constexpr (TABP) {  // (the priors of the next step are fetched by every lane, live or not)
                            if (lv) {
                                if (t & 1u)
                                    step_a(t, aM, aX, aY, bM, bX, bY);
                                else
                                    step_a(t, bM, bX, bY, aM, aX, aY);
                            }
                            tab_next(t);
                            if (lv) {
                                if (t & 1u)
                                    step_b(t, bM, bX, last);
                                else
                                    step_b(t, aM, aX, last);
                            }
                        }
                        else if (lv) {
                            if (t & 1u)
                                step(t, aM, aX, aY, bM, bX, bY, last);
                            else
                                step(t, bM, bX, bY, aM, aX, aY, last);
                        }
                        curM = nxtM;
                        curX = nxtX;
                        curY = nxtY;
                    }
                    if (!parked) late_park();  // (a row that ends inside the first half of its last superstep)
                }
            }
            return skipped;
        };
        auto write_result = [&]() {
            if (valid && lane == G - 1) {
                const T res = sumM + sumX;
                reinterpret_cast<T *>(a.out_raw)[pair] = res;
                // fp32: underflowed, to the fp64 re-run; fp64 (only the fp64-first launch has a list): fp32 might not underflow
                if (a.rescue_list && (sizeof(T) == 4 ? res < static_cast<T>(1e-28f) : static_cast<double>(res) >= a.need32_above)) {
                    const uint32_t at = atomicAdd(a.rescue_count, 1u);
                    a.rescue_list[a.sparse_order ? gidx : at] = pair;  // (sparse_order: at the pair's own position of the launch)
                }
            }
        };

        {
            // Several waves per team (few, long pairs: a wave or two per SIMD, latency-bound), each at its own pace.  Wave w sweeps
            // the stripes w, w + WP, ... and only the BAND of each: below the first stripe of a pair everything is exactly zero
            // until the first non-zero column of the carry arrives from the stripe above, and again once the band has passed, so
            // a stripe starts at the first carry block the stripe above put anything into and stops once that stripe is finished,
            // its last such block is consumed and the state is zero again.  Columns outside are never computed (they are zero,
            // and zero columns add +0 to the sums) and never written: the consumer is told where they end.  What it needs from the wave above (stripe v - 1) travels through four words
            // of LDS per wave and round: which stripe the record describes, how many supersteps of it are complete (carry
            // written and visible), where its output begins, and -- once it is finished -- where it ends.  A wave waits only
            // for the carry blocks it is about to load (kNeed supersteps ahead of its own) and never for a wave below it, so
            // the chain of stripes cannot dead-lock; every wait is bounded all the same (a stuck team poisons its results
            // instead of hanging the device).  No barrier inside the sweep.
            constexpr uint32_t kUnknown = 0xFFFFFFFFu;
            // block b of a stripe is complete once its superstep b + 2 (G = 64) / b + 1 (G = 16) is done, and a superstep begins by loading
            // the block kPrefetch ahead of the one it consumes
            constexpr uint32_t kNeed = (G == 64 ? 3u : 2u) + kPrefetch;
            auto rec_of = [&](uint32_t round, uint32_t w) { return team_rec + ((round & 1u) * WP + w) * 8u; };
            auto ld = [&](const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
            auto st = [&](uint32_t *p, uint32_t x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
            bool stuck = false;
            const bool log = a.timeline != nullptr && team < a.timeline_teams;  // tuning aid: see KernelArgs::timeline
            unsigned long long waited = 0;
            auto wait_for = [&](auto &&ready) {
                uint32_t spins = 0;
                const unsigned long long w0 = log ? __builtin_amdgcn_s_memrealtime() : 0ull;
                while (!stuck && !ready()) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 22)) {  // seconds: never reached unless the protocol is broken
                        stuck = true;
                        break;
                    }
                }
                if (log) waited += __builtin_amdgcn_s_memrealtime() - w0;
            };
            for (uint32_t round = 0; round * WP + uwave < max_stripes; ++round) {
                const uint32_t v = round * WP + uwave;
                const unsigned long long t_begin = log ? __builtin_amdgcn_s_memrealtime() : 0ull;
                unsigned long long t_first = 0;
                uint32_t n_run = 0;
                waited = 0;
                stripe_setup(v);
                const bool whole_row = a.no_early_exit || __ballot(first && Tg != 0) != 0;  // (test hook: sweep everything)
                const uint32_t *prec = v > 0 ? rec_of(uwave == 0 ? round - 1 : round, (uwave + WP - 1) % WP) : nullptr;
                uint32_t *mrec = rec_of(round, uwave);
                uint32_t c_first = 0, in_last = kUnknown;
                bool prod_done = v == 0;
                if (v > 0) {
                    wait_for([&] { return ld(prec) == v; });  // the wave above has begun stripe v - 1
                    if (!whole_row) {
                        wait_for([&] { return ld(prec + 2) != kUnknown || ld(prec + 3) != kUnknown; });
                        const uint32_t f = ld(prec + 2);
                        c_first = f == kUnknown ? Bs : min(f, Bs - 1u);  // nothing came out of the stripe above: nothing to do here
                    }
                }
                carry_last_blk = kUnknown;
                // A stripe that sweeps the whole row because one of the wave's pairs begins here still reads, for its other pairs,
                // the carry of a stripe above that may have begun in the middle of the row: the blocks before that point were
                // never written (they are zero).
                carry_first_blk = v > 0 ? ld(prec + 4) : 0u;
                // a wave whose bounded wait expired goes on with unsynchronised carry data: the poison travels down the chain of
                // stripes through word 5 of the records (sticky per wave, inherited from the stripe above at both ends of a
                // stripe), so that whichever wave runs the last stripe writes NaN
                if (v > 0 && ld(prec + 5) != 0u) stuck = true;
                if (lane == 0) {
                    st(mrec + 4, c_first);
                    st(mrec + 1, c_first);
                    st(mrec + 2, kUnknown);
                    st(mrec + 3, kUnknown);
                    st(mrec + 5, stuck ? 1u : 0u);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    st(mrec, v + 1);
                }
                uint32_t out_first = kUnknown, out_last = 0;
                for (uint32_t c = c_first; c < Bs; ++c) {
                    if (!prod_done) {
                        const uint32_t need = c + kNeed;
                        wait_for([&] { return ld(prec + 1) >= need || ld(prec + 3) != kUnknown; });
                        const uint32_t d = ld(prec + 3);
                        if (d != kUnknown) {  // the stripe above is finished: behind block d it wrote nothing (zeros)
                            prod_done = true;
                            in_last = d;
                            carry_last_blk = d;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    // (block c + 2 of the stripe above is complete once its superstep c + 4 (G = 64) / c + 3 is)
                    const bool far = kPrefetch == 2u || prod_done || ld(prec + 1) >= c + kNeed + 1u;
                    const bool skipped = superstep(v, c, c_first, far);
                    if (log && !skipped) {
                        if (n_run == 0) t_first = __builtin_amdgcn_s_memrealtime();  // (after the first non-skipped superstep)
                        ++n_run;
                    }
                    if (!skipped) {
                        if (out_first == kUnknown) {
                            out_first = c;
                            // The first non-skipped superstep f was entered with an all-zero state, so what made it non-zero is
                            // carry block f, which lane 0 meets at column 32 f + 1 at the earliest and the last lane, G - 1 steps
                            // later, at that same column: nothing comes out before block f.
                            if (lane == 0) st(mrec + 2, c);
                        }
                        out_last = c;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this superstep's carry columns before the count
                    if (lane == 0) st(mrec + 1, c + 1u);
                    if (skipped && !whole_row && prod_done && c > in_last + 2u) break;
                }
                if (v > 0 && ld(prec + 5) != 0u) stuck = true;  // (the stripe above is finished by now: its flag is final)
                if (lane == 0 && stuck) st(mrec + 5, 1u);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                // The last non-skipped superstep l leaves the last lane at column 32 l + 32 - (G - 1): block l (G = 16) or l - 1
                // (G = 64); that block is complete once superstep l + 1 has run -- it has, the loop leaves only after a skipped
                // superstep or at the end of the row.
                if (lane == 0) st(mrec + 3, out_first == kUnknown ? 0u : team_last_block<G>(out_last));
                if (v + 1 == max_stripes) {
                    if (stuck) sumM = sumM / static_cast<T>(0) - sumM / static_cast<T>(0);  // NaN: never silently wrong
                    write_result();
                }
                if (log && lane == 0 && v < (uint32_t)kTimelineStripes) {
                    unsigned long long *r = a.timeline + (static_cast<size_t>(team) * kTimelineStripes + v) * kTimelineWords;
                    r[0] = t_begin;
                    r[1] = t_first;
                    r[2] = __builtin_amdgcn_s_memrealtime();
                    r[3] = c_first;
                    r[4] = out_first;
                    r[5] = out_last;
                    r[6] = waited;
                    r[7] = n_run;
                }
            }
            (void)n_super;
        }
    }
}

// ----------------------------------------------------------------------------------------------------
// launchers (one instantiation per rows-per-lane K)
// ----------------------------------------------------------------------------------------------------
// grid of a single-pass launch: with XCD-local ordering it is rounded up to whole rounds of 8 runs (the extra workgroups find no pair)
static uint32_t systolic_grid(uint32_t blocks, uint32_t xcd_run_log2)
{
    if (xcd_run_log2 == 0) return blocks;
    const uint32_t round = 8u << xcd_run_log2;
    return (blocks + round - 1) / round * round;
}

template <typename T, int K, bool UG, int G>
static hipError_t launch_one(const KernelArgs &a, hipStream_t s)
{
    constexpr uint32_t threads = systolic_block_threads<G>();
    const uint32_t groups_per_block = threads / G;
    const uint32_t blocks = systolic_grid((a.n_pairs + groups_per_block - 1) / groups_per_block, a.xcd_run_log2);
    const size_t shmem = static_cast<size_t>(groups_per_block) * a.lds_words_per_group * sizeof(uint32_t);
    hipLaunchKernelGGL((pairhmm_systolic_kernel<T, K, UG, G>), dim3(blocks), dim3(threads), shmem, s, a);
    return hipGetLastError();
}

template <typename T, int K, int G>
static hipError_t launch_one_ugo(const KernelArgs &a, hipStream_t s)
{
    constexpr uint32_t threads = systolic_block_threads<G>();
    const uint32_t groups_per_block = threads / G;
    const uint32_t blocks = systolic_grid((a.n_pairs + groups_per_block - 1) / groups_per_block, a.xcd_run_log2);
    const size_t shmem = static_cast<size_t>(groups_per_block) * a.lds_words_per_group * sizeof(uint32_t);
    hipLaunchKernelGGL((pairhmm_systolic_ugo_kernel<T, K, G>), dim3(blocks), dim3(threads), shmem, s, a);
    return hipGetLastError();
}

template <typename T, int G, int K>
static hipError_t launch_k_ugo(int k, const KernelArgs &a, hipStream_t s)
{
    if (k == K) return launch_one_ugo<T, K, G>(a, s);
    if constexpr (K < (G == 64 ? kMaxRowsPerLaneWide : kMaxRowsPerLane))
        return launch_k_ugo<T, G, K + 1>(k, a, s);
    else
        return hipErrorInvalidValue;
}

template <typename T, bool UG, int G, int K>
static hipError_t launch_k(int k, const KernelArgs &a, hipStream_t s)
{
    if (k == K) return launch_one<T, K, UG, G>(a, s);
    if constexpr (K < (G == 64 ? kMaxRowsPerLaneWide : kMaxRowsPerLane))
        return launch_k<T, UG, G, K + 1>(k, a, s);
    else
        return hipErrorInvalidValue;
}

hipError_t launch_systolic_f32(int rows_per_lane, int lanes_per_pair, bool uniform_gcp, const KernelArgs &a, hipStream_t s, bool uniform_gop)
{
    if (a.n_pairs == 0) return hipSuccess;
    if (uniform_gcp && uniform_gop) {
        if (lanes_per_pair == 64) return launch_k_ugo<float, 64, 1>(rows_per_lane, a, s);
        if (lanes_per_pair == 16) return launch_k_ugo<float, 16, 1>(rows_per_lane, a, s);
        return hipErrorInvalidValue;
    }
    if (lanes_per_pair == 64)
        return uniform_gcp ? launch_k<float, true, 64, 1>(rows_per_lane, a, s) : launch_k<float, false, 64, 1>(rows_per_lane, a, s);
    if (lanes_per_pair != 16) return hipErrorInvalidValue;
    return uniform_gcp ? launch_k<float, true, 16, 1>(rows_per_lane, a, s) : launch_k<float, false, 16, 1>(rows_per_lane, a, s);
}

template <typename T, int K, bool UG, int WP, int G>
static hipError_t launch_team_one(const KernelArgs &a, void *carry, uint32_t carry_stride, uint32_t teams, hipStream_t s)
{
    const size_t shmem = team_lds_bytes(a.lds_words_per_group, WP, sizeof(T), G);
    // one wave per team: the lock-step team kernel; several: the band kernel (free-running waves, band-only sweeps)
    auto launch = [&](auto kern) -> hipError_t {
        if (shmem > 64u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(shmem));
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(teams), dim3(64 * WP), shmem, s, a, static_cast<T *>(carry), carry_stride);
        return hipGetLastError();
    };
    if constexpr (WP == 1)
        return launch(pairhmm_team_kernel<T, K, UG, 1, G>);
    else {
        if constexpr (sizeof(T) == 8 && K == 4 && G == 64 && WP == 4) {
            // the table form (round 5) when the launch asks for it and two teams still fit a CU with its LDS
            const size_t tab_shmem = band_tab_front_bytes(a.tab_max_hap, WP) + team_lds_bytes(0, WP, sizeof(T), G);
            if (a.band_tab && a.hap_bases && a.hap_off && tab_shmem <= 80u * 1024u) {
                auto kern = pairhmm_band_kernel<T, K, UG, WP, G, true>;
                if (tab_shmem > 64u * 1024u) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       static_cast<int>(tab_shmem));
                    if (e != hipSuccess) return e;
                }
                hipLaunchKernelGGL(kern, dim3(teams), dim3(64 * WP), tab_shmem, s, a, static_cast<T *>(carry), carry_stride);
                return hipGetLastError();
            }
        }
        return launch(pairhmm_band_kernel<T, K, UG, WP, G>);
    }
}

// one instantiation per (type, rows per lane, waves per team): one wave per team is the lock-step team kernel at the stripe
// height tuned for throughput, several waves are the band kernel at kBandRows rows per lane (latency-bound: short steps win)
template <typename T, int K, int WP>
static hipError_t launch_team_w(bool ug, int lanes_per_pair, const KernelArgs &a, void *carry, uint32_t carry_stride, uint32_t teams,
                                hipStream_t s)
{
    if (a.n_pairs == 0 && !a.n_pairs_dev && !a.n_rescue_queues) return hipSuccess;
    if (lanes_per_pair == 64)
        return ug ? launch_team_one<T, K, true, WP, 64>(a, carry, carry_stride, teams, s)
                  : launch_team_one<T, K, false, WP, 64>(a, carry, carry_stride, teams, s);
    if (lanes_per_pair == 16)
        return ug ? launch_team_one<T, K, true, WP, 16>(a, carry, carry_stride, teams, s)
                  : launch_team_one<T, K, false, WP, 16>(a, carry, carry_stride, teams, s);
    return hipErrorInvalidValue;
}

hipError_t launch_striped_f32(bool uniform_gcp, int waves_per_team, int lanes_per_pair, const KernelArgs &a, float *carry,
                              uint32_t carry_stride, uint32_t teams, hipStream_t s)
{
    if (waves_per_team == 1) return launch_team_w<float, kStripedRowsF32, 1>(uniform_gcp, lanes_per_pair, a, carry, carry_stride, teams, s);
    if (waves_per_team == 4) return launch_team_w<float, kBandRowsF32, 4>(uniform_gcp, lanes_per_pair, a, carry, carry_stride, teams, s);
    if (waves_per_team == 8) return launch_team_w<float, kBandRowsF32, 8>(uniform_gcp, lanes_per_pair, a, carry, carry_stride, teams, s);
    return hipErrorInvalidValue;
}

hipError_t launch_striped_f64(int rows_per_lane, bool uniform_gcp, int waves_per_team, int lanes_per_pair, const KernelArgs &a,
                              double *carry, uint32_t carry_stride, uint32_t teams, hipStream_t s)
{
    static_assert(kStripedRowsF64Min == 4 && kStripedRowsF64Max == 6, "instantiate every rows-per-lane value in the range");
    // several waves per team: kBandRows rows per lane whatever was asked for (see launch_team_w)
    if (waves_per_team == 4) return launch_team_w<double, kBandRowsF64, 4>(uniform_gcp, lanes_per_pair, a, carry, carry_stride, teams, s);
    if (waves_per_team == 8) return launch_team_w<double, kBandRowsF64, 8>(uniform_gcp, lanes_per_pair, a, carry, carry_stride, teams, s);
    if (waves_per_team != 1) return hipErrorInvalidValue;
    if (rows_per_lane == 4) return launch_team_w<double, 4, 1>(uniform_gcp, lanes_per_pair, a, carry, carry_stride, teams, s);
    if (rows_per_lane == 5) return launch_team_w<double, 5, 1>(uniform_gcp, lanes_per_pair, a, carry, carry_stride, teams, s);
    if (rows_per_lane == 6) return launch_team_w<double, 6, 1>(uniform_gcp, lanes_per_pair, a, carry, carry_stride, teams, s);
    return hipErrorInvalidValue;
}

}  // namespace phmm
