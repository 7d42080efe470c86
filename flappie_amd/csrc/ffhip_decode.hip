// ffhip_decode.hip -- the decode chains of the 8- and 10-state flip-flop models and of the run-length model without a lane gather on the chain.
//
// Reference functions replaced (paths relative to /root/reference/src):
//   k_crf_fb<8|10> + k_post_fb   crf_manystay_partition_function + the subtraction of globalnorm_flipflop (layers.c:1035-1096)
//                                AND transpost_crf_flipflop + log_row_normalise_inplace (decode.c:377-497, flappie_matrix.c:450-467)
//   k_viterbi8x<0>, k_viterbi10x decode_crf_flipflop (decode.c:119-204)
//   k_rle_partition8x            runlengthV2_partition_function (layers.c:1255-1302)
//   k_crf_fb<8, 1> + k_rle_post8 transpost_crf_runlength (decode.c:1037-1159)
//   k_viterbi8x<1>               decode_crf_runlength (decode.c:927-1013)
//
// Both are recursions over the blocks of one read, one wavefront per read and direction: what bounds them is the dependent chain
// of one block, not bytes or flops.  Round 2's kernels in ffhip_kernels.hip (k_crf_chain8, k_transpost8, k_viterbi8) hold the 40
// transition entries of a block one per lane, reduce per destination state with DPP moves and then GATHER the new state vector
// back to the entries' source lanes with a ds_bpermute (an LDS-crossbar round trip) -- every block.  Here the 8 x 8 (to, from)
// square of a block covers the 64 lanes and the lane <-> entry map ALTERNATES between blocks:
//     "lo" step: lane = 8 * a + b, reduce over b = lane bits 0..2 (three DPP moves inside a row of 16 lanes): the result for
//                index a ends up in all eight lanes of group a ("row form");
//     "hi" step: lane = 8 * b + a, the operand in row form is exactly "value of index b = lane >> 3"; reduce over lane bits 3..5
//                (row_ror:8, v_permlane16_swap, v_permlane32_swap): the result for index a = lane & 7 is in every group
//                ("column form"), which is what the next lo step multiplies / adds by.
// No exchange is left between the reductions.  Flop destinations (two sources each) sit in the same square with their six
// other entries masked (0 for the sums, -inf for the maxima).
//
// Posterior: the forward / backward recursions run in LINEAR space in fp64 on the exp(score - block max) values the partition
// function needs anyway (k_crf_exp), so ONE forward chain yields both logZ and the forward vectors; the backward chain runs
// beside it in a second wave.  Per-block powers of two applied to a whole vector cancel in the per-block normalisation of the
// posterior (flappie_matrix.c:450-467), so the vectors are stored as they are, and the rescaling that keeps them in range is
// folded into the next block's E values off the chain.  The reference evaluates these recursions in fp32 log space with pairwise
// logsumexpf; its own rounding noise (<= 1.9e-5 on 2000 blocks, measured against an fp64 evaluation) is the difference that remains.
#include "ffhip_internal.hpp"
#include "ffhip_math.hpp"
#include <stdlib.h>

// a dependent chain: its wave goes first wherever another batch's convolutions share the SIMD
#ifndef FFHIP_CHAIN_PRIO
#define FFHIP_CHAIN_PRIO 3
#endif
#define FFHIP_CHAIN_PRIO_SET() __builtin_amdgcn_s_setprio(FFHIP_CHAIN_PRIO)
#ifndef FFHIP_DECODE_PRIO
#define FFHIP_DECODE_PRIO 2
#endif
#define FFHIP_DECODE_PRIO_SET() __builtin_amdgcn_s_setprio(FFHIP_DECODE_PRIO)
#ifndef FFHIP_FB_CHUNK8
#define FFHIP_FB_CHUNK8 32
#endif

namespace ffhip {

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

// ---- lane exchanges -------------------------------------------------------------------------------------------------------
// DPP controls: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141 (lane j <-> 7 - j of each 8), row_ror:8 = 0x128
template <int CTRL> __device__ __forceinline__ int dpp_all(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) { return __int_as_float(dpp_all<CTRL>(__float_as_int(x))); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double x) {
    return __hiloint2double(dpp_all<CTRL>(__double2hiint(x)), dpp_all<CTRL>(__double2loint(x)));
}
// maximum over lane bits 0..2 (all eight lanes of a group receive it).  Written out: the compiler's fmaxf canonicalises both operands
// of every maximum (v_max_f32 x, x, x) and keeps the DPP move apart from it -- 9 instructions for these 3 on a chain where nothing else
// counts.  v_max_f32 returns the other operand when one is a NaN; there are no NaNs on the path that uses these.
__device__ __forceinline__ float max_lo3(float v) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(r) : "v"(v));
    return r;
}
// maximum over lane bits 3..5 (all lanes with the same lane & 7 receive it): row_ror:8, then v_permlane16_swap / v_permlane32_swap of
// two copies leave { rows 0 0 2 2 | rows 1 1 3 3 } and { low half twice | high half twice }
__device__ __forceinline__ float max_hi3(float v) {
    float r, t;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1\n\t"
        "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1"
        : "=&v"(r), "=&v"(t) : "v"(v));
    return r;
}
// plain v_max_f32 of two values that are no NaNs (fmaxf would canonicalise both operands first: three instructions for one)
__device__ __forceinline__ float max2_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double sum_lo3(double v) {
    v = v + dpp_d<0x141>(v);
    v = v + dpp_d<0xB1>(v);
    v = v + dpp_d<0x4E>(v);
    return v;
}
__device__ __forceinline__ double swap16_sum(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const v2u_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
}
__device__ __forceinline__ double swap32_sum(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const v2u_t a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
}
__device__ __forceinline__ double sum_hi3(double v) {
    v = v + dpp_d<0x128>(v);
    v = swap16_sum(v);
    return swap32_sum(v);
}

// entry of a block's scores that carries the transition from -> to (decode.c:104-114), and whether there is one.  NS = 2 * nbase states:
// flip destinations to < nbase take every source, flop destination to takes itself (stay) and its flip state to - nbase (move).
template <int NS> __device__ __forceinline__ int ff_entry(int to, int from) { return to < NS / 2 ? NS * to + from : (NS / 2) * NS + from; }
template <int NS> __device__ __forceinline__ bool ff_has(int to, int from) { return to < NS / 2 || from == to || from == to - NS / 2; }
__device__ __forceinline__ int ff8_entry(int to, int from) { return ff_entry<8>(to, from); }
__device__ __forceinline__ bool ff8_has(int to, int from) { return ff_has<8>(to, from); }
// TOPO 1: the run-length model's 8 states (4 move + 4 stay) in the same square: move state b1 is entered from every state of another base (entry 8 b1 + from of
// a block's 32 transition scores), stay state 4 + b from the move and the stay state of its own base (entry 8 b + from)
template <int TOPO> __device__ __forceinline__ int vit_entry(int to, int from) { return TOPO == 0 ? ff8_entry(to, from) : (to < 4 ? 8 * to + from : 8 * (to - 4) + from); }
template <int TOPO> __device__ __forceinline__ bool vit_has(int to, int from) { return TOPO == 0 ? ff8_has(to, from) : (to < 4 ? (from & 3) != to : (from & 3) == to - 4); }

// ---- partition function + normalisation + posterior ---------------------------------------------------------------------------
// NS = 8: the square is the whole block.  NS = 10 (the 5-base models): the square carries states 0..7 and a second vector x the states
// 8 and 9, at positions 3 and 4 of whatever form the main vector has (lanes with lane & 7 == 3, 4 in column form, groups 3 and 4 in
// row form).  Per step and lane four coefficients instead of one:
//     main' = reduce(e1 * a + e2 * x)        e2: from the extra source carried at this lane's in-position into the lane's destination
//     x'    = reduce(f1 * x + f2 * a)        in the lanes whose out-position is 3 or 4: f1 extra -> extra (stay), f2 main source -> extra
// and x' lands at positions 3, 4 of the form main' lands in.  Two independent reductions per step; everything else as for NS = 8.
template <int NS, int TOPO = 0> struct FbDims {
    static constexpr int P = TOPO == 1 ? 32 : NS * (NS / 2 + 1);
    static constexpr int Pd = (P + 1 + 7) & ~7;               // crf_exp_stride(P): P entries, the block maximum at [P], zeros behind it
    // blocks of E staged in LDS per chunk and direction.  32 for the 8-state models; 16 for the 10-state ones, whose rows are 64 doubles: with 32
    // the workgroup took 43 KB of LDS, three workgroups a CU, and the 1024 reads of a `c4` batch ran in two rounds -- with 16 (27 KB) all 1024
    // chains are resident at once (four workgroups a CU, what their 176 registers allow)
    static constexpr int kChunk = NS > 8 ? 16 : FFHIP_FB_CHUNK8;
    static constexpr int kStage = kChunk * Pd / 64;           // doubles per lane and chunk
    static constexpr int pad = P + 1;                         // an entry that reads as zero
};

// coefficient index of "in-state -> out-state" for one direction: forwards out = to, in = from; backwards out = from, in = to
template <int NS, bool FWD, int TOPO = 0> __device__ __forceinline__ int fb_coef(int out, int in) {
    const int to = FWD ? out : in, from = FWD ? in : out;
    if constexpr (TOPO == 1) return (to < NS && from < NS && vit_has<1>(to, from)) ? vit_entry<1>(to, from) : FbDims<NS, 1>::pad;
    else return (to < NS && from < NS && ff_has<NS>(to, from)) ? ff_entry<NS>(to, from) : FbDims<NS>::pad;
}

// One direction of the recursion, one wave.  q counts the blocks in the order they are processed (forwards: block q, backwards:
// block Tb - 1 - q); even q are lo steps, odd q hi steps.  `vec` receives the vector after every block: forwards vec[(q + 1) * NS],
// backwards vec[(Tb - 1 - q) * NS]; the caller has written the all-ones start vector.
// TOPO 1 (run-length model: its posterior is NOT normalised per block, decode.c:1037-1159): rows of 10 doubles -- the vector and, in [8], the offset C with
// log(true value) = log(stored) + C (ln2 times the powers of two taken out so far + the block maxima so far), tracked in both directions.
template <int NS, bool FWD, int TOPO = 0>
__device__ __forceinline__ void fb_chain(const double *__restrict__ Er, const int Tb, double *__restrict__ ebuf, double (*__restrict__ stage)[TOPO == 1 ? 10 : NS],
                                         double *__restrict__ vec, const bool store, double *__restrict__ logz) {
    constexpr int Pd = FbDims<NS, TOPO>::Pd, P = FbDims<NS, TOPO>::P, kStage = FbDims<NS, TOPO>::kStage, kFbChunk = FbDims<NS, TOPO>::kChunk;
    constexpr bool ABS = TOPO == 1;
    constexpr int RS = ABS ? 10 : NS;                           // doubles per stored row
    constexpr bool X = NS > 8;                                 // states 8, 9 ride in the second vector
    const int lane = threadIdx.x & 63, g = lane >> 3, j = lane & 7;
    // lo step: out-position g, in-position j (reduce over lane bits 0..2); hi step: out-position j, in-position g (bits 3..5).
    // Entries that do not exist read the row's padding, which k_crf_exp fills with zeros: no select, no predicated load on the chain.
    const int e1_lo = fb_coef<NS, FWD, TOPO>(g, j), e1_hi = fb_coef<NS, FWD, TOPO>(j, g);
    const bool xi_lo = (j == 3 || j == 4), xo_lo = (g == 3 || g == 4), xi_hi = xo_lo, xo_hi = xi_lo;
    const int e2_lo = X && xi_lo ? fb_coef<NS, FWD>(g, j + 5) : FbDims<NS>::pad, e2_hi = X && xi_hi ? fb_coef<NS, FWD>(j, g + 5) : FbDims<NS>::pad;
    const int f1_lo = X && xo_lo && xi_lo ? fb_coef<NS, FWD>(g + 5, j + 5) : FbDims<NS>::pad, f1_hi = X && xo_hi && xi_hi ? fb_coef<NS, FWD>(j + 5, g + 5) : FbDims<NS>::pad;
    const int f2_lo = X && xo_lo ? fb_coef<NS, FWD>(g + 5, j) : FbDims<NS>::pad, f2_hi = X && xo_hi ? fb_coef<NS, FWD>(j + 5, g) : FbDims<NS>::pad;
    const int nchunk = (Tb + kFbChunk - 1) / kFbChunk;
    const int lim = Tb * Pd;             // (< 2^31: Tb <= 2^24 blocks)
    double st[kStage];
    // chunk c: steps [32 c, 32 c + 32) = blocks from `first` upwards in memory.  Clamped addresses and nothing else: the rows of blocks
    // outside the read are never walked, and a select on the loaded value would make every load of the chunk wait for itself.
    auto fetch = [&](int c) {
        const int first = FWD ? c * kFbChunk : Tb - (c + 1) * kFbChunk;
        const int base = first * Pd + lane;
#pragma unroll
        for (int k = 0; k < kStage; k++) st[k] = Er[max(0, min(base + k * 64, lim - 1))];
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < kStage; k++) ebuf[k * 64 + lane] = st[k];
    };
    auto flush = [&](int q) {            // the stage rows 0 .. (q & 63) leave for HBM, NS doubles a lane
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int cnt = (q & 63) + 1, q0 = q & ~63;
        if (lane < cnt) {
            const int qq = q0 + lane;
            double *dst = vec + (size_t)(FWD ? qq + 1 : Tb - 1 - qq) * RS;
            const double2 *src = (const double2 *)&stage[lane][0];
            double2 *d2 = (double2 *)dst;
#pragma unroll
            for (int k = 0; k < RS / 2; k++) d2[k] = src[k];
        }
        __builtin_amdgcn_wave_barrier();
    };
    fetch(0);
    double a = 1.0, x = 1.0, msum = 0.0;
    long long K = 0;
    int sx = 0;                          // power of two taken out of the NEXT pair's first E values (wave-uniform)
    for (int c = 0; c < nchunk; c++) {
        commit();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (c + 1 < nchunk) fetch(c + 1);
        const int q0 = c * kFbChunk, q1 = min(Tb, q0 + kFbChunk);
        // rows of the chunk by step; the E values of the NEXT pair are read from LDS while this pair's reductions run
        auto row_of = [&](int q) { return ebuf + (FWD ? q - q0 : kFbChunk - 1 - (q - q0)) * Pd; };
        double el_n, eh_n, ml_n, mh_n, el2_n = 0.0, eh2_n = 0.0, fl1_n = 0.0, fl2_n = 0.0, fh1_n = 0.0, fh2_n = 0.0;
        auto read_pair = [&](int q) {
            const double *rl = row_of(min(q, q1 - 1)), *rh = row_of(min(q + 1, q1 - 1));
            el_n = rl[e1_lo]; ml_n = rl[P];
            eh_n = rh[e1_hi]; mh_n = rh[P];
            if (X) { el2_n = rl[e2_lo]; fl1_n = rl[f1_lo]; fl2_n = rl[f2_lo]; eh2_n = rh[e2_hi]; fh1_n = rh[f1_hi]; fh2_n = rh[f2_hi]; }
        };
        read_pair(q0);
        for (int q = q0; q < q1; q += 2) {
            // the scaling is exact and off the chain
            const double el = __builtin_ldexp(el_n, -sx), eh = eh_n, ml = ml_n, mh = mh_n;
            const double el2 = X ? __builtin_ldexp(el2_n, -sx) : 0.0, fl1 = X ? __builtin_ldexp(fl1_n, -sx) : 0.0, fl2 = X ? __builtin_ldexp(fl2_n, -sx) : 0.0;
            const double eh2 = eh2_n, fh1 = fh1_n, fh2 = fh2_n;
            read_pair(q + 2);
            {   // lo step
                if (FWD || ABS) { msum = msum + ml; K += sx; }
                if (X) {
                    const double t = __builtin_fma(el2, x, el * a), tx = __builtin_fma(fl1, x, fl2 * a);
                    a = sum_lo3(t); x = sum_lo3(tx);
                } else a = sum_lo3(el * a);
                if (store) {
                    if (j == 0) stage[q & 63][g] = a;
                    if (ABS && lane == 0) stage[q & 63][8] = 0.693147180559945309417232121458 * (double)K + msum;
                    if (X && j == 0 && xo_lo) stage[q & 63][g + 5] = x;
                    if ((q & 63) == 63 || q == Tb - 1) flush(q);
                }
            }
            if (q + 1 < q1) {   // hi step
                const int qh = q + 1;
                if (FWD || ABS) msum = msum + mh;
                if (X) {
                    const double t = __builtin_fma(eh2, x, eh * a), tx = __builtin_fma(fh1, x, fh2 * a);
                    a = sum_hi3(t); x = sum_hi3(tx);
                } else a = sum_hi3(eh * a);
                if (store) {
                    if (lane < 8) stage[qh & 63][lane] = a;
                    if (ABS && lane == 8) stage[qh & 63][8] = 0.693147180559945309417232121458 * (double)K + msum;
                    if (X && (lane == 3 || lane == 4)) stage[qh & 63][lane + 5] = x;
                    if ((qh & 63) == 63 || qh == Tb - 1) flush(qh);
                }
                // the exponent of state 0's value (lane 0; positive: every state is reached from a flip state, every flip state from all)
                // becomes the scaling of the pair after this one -- a scalar side computation
                const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(a));
                const int ex = (hi >> 20) & 0x7ff;
                sx = ex ? ex - 1023 : 0;
            } else sx = 0;
        }
    }
    if (FWD) {
        // logZ = log(sum of the final vector) + ln2 * K + sum of the block maxima (layers.c:1071-1079 in linear space)
        double total;
        if (Tb & 1) total = sum_hi3(X && xo_lo ? a + x : a);          // an odd count ends in row form
        else total = sum_lo3(X && xo_hi ? a + x : a);
        if (lane == 0) *logz = log(total) + 0.693147180559945309417232121458 * (double)K + msum;
    }
}

// The two chains of a read: wave 0 forwards (and logZ), wave 1 backwards (flags & 2: the vectors are wanted).
// `wide` (optional): per-read flag of k_crf_exp "score range too wide for the linear form": such a read is left to the log-space kernels.
template <int NS, int TOPO = 0>
__global__ void __launch_bounds__(128)
k_crf_fb(const double *__restrict__ E, double *__restrict__ fwdbuf, double *__restrict__ bwdbuf, int TbS, double *__restrict__ logz_out,
         const int *__restrict__ tbs, int flags, const int *__restrict__ wide, ReadMap map) {
    FFHIP_CHAIN_PRIO_SET();
    constexpr int Pd = FbDims<NS, TOPO>::Pd, RS = TOPO == 1 ? 10 : NS, kFbChunk = FbDims<NS, TOPO>::kChunk;
    __shared__ double ebuf[2][kFbChunk * Pd];
    __shared__ double stage[2][64][RS];
    __shared__ double s_logz;
    if (wide && wide[blockIdx.x]) return;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;
    if (Tb <= 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double *Er = E + map.row0(blockIdx.x, TbS) * Pd;
    double *F = fwdbuf + map.row1(blockIdx.x, TbS) * RS;
    double *Bw = bwdbuf + map.row1(blockIdx.x, TbS) * RS;
    const bool want_post = (flags & 2) != 0;
    if (wave == 0) {
        if (want_post && lane < RS) F[lane] = lane < NS ? 1.0 : 0.0;              // (run-length rows: offset C = 0 in [8])
        fb_chain<NS, true, TOPO>(Er, Tb, ebuf[0], stage[0], F, want_post, &s_logz);
        if (lane == 0 && logz_out) logz_out[blockIdx.x] = s_logz;
    } else if (want_post) {
        if (lane < RS) Bw[(size_t)Tb * RS + lane] = lane < NS ? 1.0 : 0.0;
        fb_chain<NS, false, TOPO>(Er, Tb, ebuf[1], stage[1], Bw, true, nullptr);
    }
}

// Sixteen lanes per block, four entries a lane (a row of P <= 64 scores is one coalesced float4 load of its group): normalised scores
// (flags & 1: minus (float)(logZ / Tb), layers.c:1089-1096), then (flags & 2) the posterior of entry r = (fwd[from] + bwd[to]) + score
// (decode.c:451-461), log-normalised over the block's P entries (flappie_matrix.c:450-467; the sum over the entries is taken four a
// lane and then across the group, not in index order: fp32 rounding of a sum of at most 60 positive terms).
constexpr int kPostBlocks = 64;          // blocks per workgroup: 4 waves x 4 groups x 4 rounds
template <int CTRL> __device__ __forceinline__ float dpp16_f(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true)); }
template <int CTRL> __device__ __forceinline__ int dpp16_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
// all-to-all reductions inside a row of 16 lanes: xor 1, xor 2, mirror inside 8, mirror inside 16 (DPP row_mirror 0x140)
__device__ __forceinline__ float max16(float v) {
    v = fmaxf(v, dpp16_f<0xB1>(v)); v = fmaxf(v, dpp16_f<0x4E>(v)); v = fmaxf(v, dpp16_f<0x141>(v)); return fmaxf(v, dpp16_f<0x140>(v));
}
__device__ __forceinline__ float sum16(float v) {
    v = v + dpp16_f<0xB1>(v); v = v + dpp16_f<0x4E>(v); v = v + dpp16_f<0x141>(v); return v + dpp16_f<0x140>(v);
}
__device__ __forceinline__ int imax16(int v) {
    v = max(v, dpp16_i<0xB1>(v)); v = max(v, dpp16_i<0x4E>(v)); v = max(v, dpp16_i<0x141>(v)); return max(v, dpp16_i<0x140>(v));
}

template <int NS>
__global__ void __launch_bounds__(256)
k_post_fb(float *__restrict__ trans, float *__restrict__ post, const double *__restrict__ fwdbuf, const double *__restrict__ bwdbuf, int TbS,
          const double *__restrict__ logz, const int *__restrict__ tbs, int flags, const int *__restrict__ wide, ReadMap map) {
    FFHIP_DECODE_PRIO_SET();
    constexpr int P = FbDims<NS>::P, Ps = P, nbase = NS / 2, off = nbase * NS;
    const int read = blockIdx.y;
    const size_t row0 = map.row0(read, TbS), row1 = map.row1(read, TbS);
    if (wide && wide[read]) return;
    const int Tb = tbs ? tbs[read] : TbS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4, sub = lane & 15;
    if ((int)blockIdx.x * kPostBlocks >= Tb) return;
    const float sub_c = (flags & 1) ? (float)(logz[read] / (double)Tb) : 0.0f;
    const bool has = 4 * sub < P;                           // this lane's four entries exist (P is a multiple of 4)
    // lanes of this wave that hold fwd[from] / bwd[to] of my entries
    int lf_src[4], lb_src[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int r = min(4 * sub + e, P - 1);
        const int from = r % NS;
        const int to = (r < off) ? (r / NS) : ((r - off < nbase) ? r - off + nbase : r - off);
        lf_src[e] = 4 * (16 * grp + from); lb_src[e] = 4 * (16 * grp + to);       // byte addresses for ds_bpermute
    }
    for (int round = 0; round < 4; round++) {
        const int blk = blockIdx.x * kPostBlocks + (round * 4 + wave) * 4 + grp;
        const bool live = blk < Tb;
        const int bc = min(blk, Tb - 1);
        float4 *xr = (float4 *)(trans + (row0 + bc) * Ps) + min(sub, P / 4 - 1);
        float4 x = *xr;
        if (flags & 1) { x.x -= sub_c; x.y -= sub_c; x.z -= sub_c; x.w -= sub_c; if (live && has) *xr = x; }
        if (!(flags & 2)) continue;
        // log of the vectors relative to their largest exponent (a common factor per block and direction cancels below); lane sub < NS
        // holds state sub of both.  value = mantissa in [0.5, 1) * 2^e; a zero (unreachable state) gives -inf like the reference's start
        const int st = min(sub, NS - 1);
        const double va = fwdbuf[(row1 + bc) * NS + st], vb = bwdbuf[(row1 + bc + 1) * NS + st];
        const bool pa = sub < NS && va > 0.0, pb = sub < NS && vb > 0.0;
        const int ea = pa ? __builtin_amdgcn_frexp_exp(va) : -100000, eb = pb ? __builtin_amdgcn_frexp_exp(vb) : -100000;
        const int ma = imax16(ea), mb = imax16(eb);
        const float lf = pa ? (float)((double)(ea - ma) * 0.693147180559945309417232121458 + (double)logf((float)__builtin_amdgcn_frexp_mant(va))) : -INFINITY;
        const float lb = pb ? (float)((double)(eb - mb) * 0.693147180559945309417232121458 + (double)logf((float)__builtin_amdgcn_frexp_mant(vb))) : -INFINITY;
        float v[4];
        const float xs[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float f = __int_as_float(__builtin_amdgcn_ds_bpermute(lf_src[e], __float_as_int(lf)));
            const float b = __int_as_float(__builtin_amdgcn_ds_bpermute(lb_src[e], __float_as_int(lb)));
            v[e] = has ? (f + b) + xs[e] : -INFINITY;
        }
        const float m = max16(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        float sum = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) sum += has ? expf(v[e] - m) : 0.0f;
        const float lse = m + logf(sum16(sum));
        if (live && has) *((float4 *)(post + (row0 + blk) * Ps) + sub) = make_float4(v[0] - lse, v[1] - lse, v[2] - lse, v[3] - lse);
    }
}

// transpost_crf_runlength's output (decode.c:1102-1135) from the vectors of k_crf_fb<8, 1>: sixteen lanes a block, four of its 40 values each -- the shape / scale
// rows copied, entry r = 8 tb + from of the transition rows = (fwd[from] + bwd[to]) + score with to = move state tb, or stay state 4 + tb for the two sources of
// its own base; fwd / bwd = log of the stored vector + its offset, formed in fp64 and rounded once
__global__ void __launch_bounds__(256)
k_rle_post8(const float *__restrict__ param, float *__restrict__ post, const double *__restrict__ fwdbuf, const double *__restrict__ bwdbuf, int TbS,
            const int *__restrict__ tbs) {
    FFHIP_DECODE_PRIO_SET();
    constexpr int Ps = 40, RS = 10;
    const int read = blockIdx.y;
    const int Tb = tbs ? tbs[read] : TbS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4, sub = lane & 15;
    if ((int)blockIdx.x * kPostBlocks >= Tb) return;
    const bool has = sub < 10;
    int lf_src[4], lb_src[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int r = max(0, min(4 * sub + e, Ps - 1) - 8), from = r & 7, tb = r >> 3;
        const int to = ((from & 3) != tb) ? tb : tb + 4;
        lf_src[e] = 4 * (16 * grp + from); lb_src[e] = 4 * (16 * grp + to);
    }
    for (int round = 0; round < 4; round++) {
        const int blk = blockIdx.x * kPostBlocks + (round * 4 + wave) * 4 + grp;
        const bool live = blk < Tb;
        const int bc = min(blk, Tb - 1);
        const float4 x = *((const float4 *)(param + ((size_t)read * TbS + bc) * Ps) + min(sub, 9));
        const int st = min(sub, 7);
        const double *fr = fwdbuf + ((size_t)read * (TbS + 1) + bc) * RS, *br = bwdbuf + ((size_t)read * (TbS + 1) + bc + 1) * RS;
        const double va = fr[st], vb = br[st], ca = fr[8], cb = br[8];
        // log(v) = log(mantissa) + e ln2 (a zero: -inf, an unreachable state)
        const float lf = va > 0.0 ? (float)((double)__builtin_amdgcn_frexp_exp(va) * 0.693147180559945309417232121458 + (double)logf((float)__builtin_amdgcn_frexp_mant(va)) + ca) : -INFINITY;
        const float lb = vb > 0.0 ? (float)((double)__builtin_amdgcn_frexp_exp(vb) * 0.693147180559945309417232121458 + (double)logf((float)__builtin_amdgcn_frexp_mant(vb)) + cb) : -INFINITY;
        const float xs[4] = { x.x, x.y, x.z, x.w };
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float f = __int_as_float(__builtin_amdgcn_ds_bpermute(lf_src[e], __float_as_int(lf)));
            const float b = __int_as_float(__builtin_amdgcn_ds_bpermute(lb_src[e], __float_as_int(lb)));
            v[e] = (4 * sub + e < 8) ? xs[e] : (f + b) + xs[e];
        }
        if (live && has) *((float4 *)(post + ((size_t)read * TbS + blk) * Ps) + sub) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// run-length posterior of nbase = 4, stride-40 reads: E (k_crf_exp of the 32 transition rows, stride 40 doubles), the two chains, the assembly
void launch_rle_post8(hipStream_t s, const float *param, float *post, double *E, double *fwd, int nread, int Tb, const int *tbs) {
    launch_crf_exp(s, param, E, nread, Tb, 4, 40, tbs, nullptr, 0.0f, 8, 32);
    double *bwd = fwd + (size_t)nread * (Tb + 1) * 10;
    hipLaunchKernelGGL((k_crf_fb<8, 1>), dim3(nread), dim3(128), 0, s, E, fwd, bwd, Tb, (double *)nullptr, tbs, 2, (const int *)nullptr, ReadMap());
    hipLaunchKernelGGL(k_rle_post8, dim3((Tb + kPostBlocks - 1) / kPostBlocks, nread), dim3(256), 0, s, param, post, fwd, bwd, Tb, tbs);
}

// ---- runlengthV2_partition_function (layers.c:1255-1302) on the alternating layouts, in LOG space -----------------------------------------------
// The reference folds a move state's 2 (nbase - 1) sources with pairwise fp64 logsumexp and rounds every STAY update through the float logsumexpf
// (layers.c:1288-1290); the host-API test holds the operator to 1e-9 of that number, so the stay states keep exactly that float arithmetic and the
// recursion stays in log space -- per block an fp64 max, exp, sum and log for the move states (any association: 1e-16) and the float logsumexpf for
// the stay states, both on the lane layouts of the chains above instead of five LDS round trips per block (k_rle_partition, ffhip_rle.hip).
__device__ __forceinline__ double fmax_lo3_d(double v) { v = fmax(v, dpp_d<0x141>(v)); v = fmax(v, dpp_d<0xB1>(v)); return fmax(v, dpp_d<0x4E>(v)); }
__device__ __forceinline__ double fmax_hi3_d(double v) {
    v = fmax(v, dpp_d<0x128>(v));
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const v2u_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = fmax(__hiloint2double((int)b.x, (int)a.x), __hiloint2double((int)b.y, (int)a.y));
    }
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const v2u_t a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return fmax(__hiloint2double((int)b.x, (int)a.x), __hiloint2double((int)b.y, (int)a.y));
}

__global__ void __launch_bounds__(64)
k_rle_partition8x(const float *__restrict__ param, int TbS, double *__restrict__ logz, const int *__restrict__ tbs) {
    FFHIP_CHAIN_PRIO_SET();
    constexpr int Ps = 40, nbase = 4;
    const int lane = threadIdx.x, g = lane >> 3, j = lane & 7;
    const float *T = param + (size_t)blockIdx.x * TbS * Ps + 2 * nbase;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;
    if (Tb <= 0) return;
    const double NEG = -HUGE_VAL;
    // lo step: (to g, from j), hi step: (to j, from g)
    const int i_lo = vit_entry<1>(g, j), i_hi = vit_entry<1>(j, g);
    const bool v_lo = vit_has<1>(g, j), v_hi = vit_has<1>(j, g);
    double v = 0.0;                                          // all eight states start at 0; column form (state lane & 7)
    auto lse_f = [](float x, float y) { return logsumexpf_ref(x, y); };
    auto step_lo = [&](float x) {                            // v: column form -> row form
        const double cand = v_lo ? v + (double)x : NEG;
        // move destinations (groups 0..3)
        const double m = fmax_lo3_d(cand);
        const double e = exp(cand - m);
        const double mv = m + log(sum_lo3(e));
        // stay destinations (groups 4..7): the two sources sit four lanes apart in the group
        const float xf = (float)cand;
        const float pf = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xf), 0x104, 0xf, 0x5, false) |      // row_shl:4 into banks 0, 2 ..
                                        __builtin_amdgcn_update_dpp(0, __float_as_int(xf), 0x114, 0xf, 0xA, false));      // .. row_shr:4 into banks 1, 3
        const double sv = v_lo ? (double)lse_f(xf, pf) : NEG;
        v = g < nbase ? mv : fmax_lo3_d(sv);
    };
    auto step_hi = [&](float x) {                            // v: row form -> column form
        const double cand = v_hi ? v + (double)x : NEG;
        const double m = fmax_hi3_d(cand);
        const double e = exp(cand - m);
        const double mv = m + log(sum_hi3(e));
        const float xf = (float)cand;
        const v2u_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(xf), __float_as_uint(xf), false, false);       // { low half twice, high half twice }
        const float pf = __uint_as_float(lane < 32 ? r.y : r.x);                 // the lane 32 away: source group g ^ 4
        const double sv = v_hi ? (double)lse_f(xf, pf) : NEG;
        v = j < nbase ? mv : fmax_hi3_d(sv);
    };
    constexpr int kDepth = 8;
    float ring[kDepth];
    auto fetch = [&](int blk) { return T[(size_t)min(blk, Tb - 1) * Ps + ((blk & 1) ? i_hi : i_lo)]; };
#pragma unroll
    for (int k = 0; k < kDepth; k++) ring[k] = fetch(k);
    for (int b0 = 0; b0 < Tb; b0 += kDepth) {
        float cur[kDepth];
#pragma unroll
        for (int k = 0; k < kDepth; k++) cur[k] = ring[k];
#pragma unroll
        for (int k = 0; k < kDepth; k++) ring[k] = fetch(b0 + kDepth + k);
#pragma unroll
        for (int k = 0; k < kDepth; k++) {
            if (b0 + k >= Tb) break;
            if (k & 1) step_hi(cur[k]); else step_lo(cur[k]);
        }
    }
    // final states: an odd block count ends in row form (state s in group s), an even one in column form (state s in lane s)
    double z = 0.0;
#pragma unroll
    for (int s2 = 0; s2 < 2 * nbase; s2++) {
        const double val = __shfl(v, (Tb & 1) ? 8 * s2 : s2);
        z = s2 == 0 ? val : (fmax(z, val) + log1p(exp(-fabs(z - val))));           // layers.c:1296-1299, in order
    }
    if (lane == 0) logz[blockIdx.x] = z;
}

void launch_rle_partition8x(hipStream_t s, const float *param, double *logz, int nread, int Tb, const int *tbs) {
    hipLaunchKernelGGL(k_rle_partition8x, dim3(nread), dim3(64), 0, s, param, Tb, logz, tbs);
}

// flags: 1 = subtract (float)(logZ / Tb) from the scores, 2 = posterior wanted; logz: device doubles per read (required with flags & 1)
void launch_crf_fb(hipStream_t s, int nbase, const double *E, float *trans, float *post, double *fwd, int nread, int Tb, double *logz, const int *tbs,
                   int flags, const int *wide, ReadMap map) {
    const int NS = 2 * nbase;
    double *bwd = fwd + (size_t)(map.nslot > 0 ? map.nslot : nread) * (Tb + 1) * NS;      // (a packed batch: the rows are its slots')
    const dim3 grid((Tb + kPostBlocks - 1) / kPostBlocks, nread);
    if (nbase == 4) {
        hipLaunchKernelGGL(k_crf_fb<8>, dim3(nread), dim3(128), 0, s, E, fwd, bwd, Tb, logz, tbs, flags, wide, map);
        if (flags & 3) hipLaunchKernelGGL(k_post_fb<8>, grid, dim3(256), 0, s, trans, post, fwd, bwd, Tb, logz, tbs, flags, wide, map);
    } else {
        hipLaunchKernelGGL(k_crf_fb<10>, dim3(nread), dim3(128), 0, s, E, fwd, bwd, Tb, logz, tbs, flags, wide, map);
        if (flags & 3) hipLaunchKernelGGL(k_post_fb<10>, grid, dim3(256), 0, s, trans, post, fwd, bwd, Tb, logz, tbs, flags, wide, map);
    }
}

// ---- Viterbi --------------------------------------------------------------------------------------------------------------
// decode.c:119-204 with its tie rules (flip: lowest from-state among equal maxima; flop: stay unless the move is strictly
// greater; final state: first maximum, util.c:17-31).  The chain carries the VALUE only -- max is exact -- and every block leaves a
// 64-bit ballot "this candidate equals its destination's maximum"; winners, traceback bytes and the path come out of those words
// afterwards, in parallel:
//   1. forward: pairs of blocks (lo step, hi step), 8 blocks prefetched three groups ahead; a group of 8 blocks holding a NaN or an
//      infinity (never produced by the network, possible through ffhip_viterbi) takes the reference's scan order literally instead;
//   2. the ballots of a chunk become 8 traceback bytes per block, all lanes a block each;
//   3. traceback by segments: each lane composes the maps of its 1/64 of the chunk for all 8 end states, one lane chains the 64
//      segment maps, each lane then walks its segment from its known end state.
constexpr int kVitChunk = 2048;          // blocks whose traceback words stay in LDS (a multiple of 8)

// TOPO 1: the run-length model's 8 states (4 move + 4 stay, decode.c:927-1013 decode_crf_runlength) on the same chain.  Its 32 transition scores sit behind
// the 8 shape / scale rows of a block; move state b1 is entered from every state of another base (score 8 b1 + from), stay state b from the move and
// the stay state of its own base (score 8 b + from) -- the 8 x 8 square again, with other entries masked.  Its scan visits the sources of a move state
// in the order move b2, stay b2 (b2 ascending) and replaces on a strict > starting from -HUGE_VAL with traceback 0; a stay state takes "stay" only
// if strictly greater than "move".  Its path holds the state AFTER every block (and 0 behind the last), no qpath.

template <int TOPO>
__global__ void __launch_bounds__(64)
k_viterbi8x(const float *__restrict__ M, uint8_t *__restrict__ tbbuf, int *__restrict__ path, float *__restrict__ qpath,
            float *__restrict__ score_out, int TbS, const int *__restrict__ tbs, ReadMap map) {
    FFHIP_CHAIN_PRIO_SET();
    constexpr int Ps = 40, ns = 8, nbase = 4, off = 32;
    __shared__ unsigned long long tbw[kVitChunk];           // per block: first the ballot, then the 8 traceback bytes
    __shared__ unsigned careful[kVitChunk / 8 / 32];        // bit per group of 8 blocks: its words are one-hot in the lo layout
    __shared__ uint8_t path_lds[kVitChunk + 1];
    const int lane = threadIdx.x, g = lane >> 3, j = lane & 7;
    const float *T = M + map.row0(blockIdx.x, TbS) * Ps + (TOPO == 1 ? 2 * nbase : 0);      // (run-length: the transition rows of a block)
    unsigned long long *tbg = (unsigned long long *)(tbbuf + map.row0(blockIdx.x, TbS) * kMaxState);       // 16 bytes a block: room for the 8
    int *pth = path + map.row1(blockIdx.x, TbS);
    float *qp = qpath + map.row1(blockIdx.x, TbS);
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;
    if (Tb <= 0) return;
    const float NEG = -INFINITY;
    // lo step: (to g, from j); hi step: (to j, from g)
    const int i_lo = vit_entry<TOPO>(g, j), i_hi = vit_entry<TOPO>(j, g);
    const bool v_lo = vit_has<TOPO>(g, j), v_hi = vit_has<TOPO>(j, g);
    const int nchunk = (Tb + kVitChunk - 1) / kVitChunk;
    const int ngroup_all = Tb / 8;                          // whole groups of the read; the rest (< 8 blocks) goes the literal way
    float pv = 0.0f;                                        // column form: value of state lane & 7

    // the literal step, lo layout: candidates of block blk from the score x of entry (to g, from j).  (The lane exchanges inside its branches read lanes of
    // the SAME 8-lane group, which take the same branch: every source lane is active.)
    auto literal_step = [&](float x, int slot) {
        const float cand = v_lo ? x + pv : NEG;
        float v;
        int arg;
        if constexpr (TOPO == 1) {
            if (g < nbase) {
                // curr = -HUGE_VAL, traceback 0; candidates in the order move b2, stay b2 (b2 ascending, b2 != g), each on a strict >: the first
                // maximum among those greater than -inf, NaNs skipped
                float mx = fmaxf(cand, dpp_f<0x141>(cand));
                mx = fmaxf(mx, dpp_f<0xB1>(mx));
                mx = fmaxf(mx, dpp_f<0x4E>(mx));
                const unsigned eq = (unsigned)(__ballot(v_lo && cand == mx) >> (8 * g)) & 0xffu;
                v = NEG; arg = 0;
                if (mx > NEG && eq) {
                    v = mx;
                    bool found = false;
#pragma unroll
                    for (int pp = 0; pp < 8; pp++) { const int f = (pp >> 1) + 4 * (pp & 1); if (!found && ((eq >> f) & 1u)) { arg = f; found = true; } }
                }
            } else {
                const float stay = __shfl(cand, 8 * g + g), move = __shfl(cand, 8 * g + g - nbase);
                const bool stayed = stay > move;
                v = stayed ? stay : move;
                arg = stayed ? g : g - nbase;
            }
        } else
        if (g < nbase) {
            // the scan starts from the from-state-0 candidate and replaces on a strict >: a NaN there stays, NaNs elsewhere are skipped
            float mx = fmaxf(cand, dpp_f<0x141>(cand));
            mx = fmaxf(mx, dpp_f<0xB1>(mx));
            mx = fmaxf(mx, dpp_f<0x4E>(mx));
            const float c0 = __shfl(cand, lane & ~7);
            v = (c0 != c0) ? c0 : mx;
            const unsigned eq = (unsigned)(__ballot(cand == v) >> (8 * g)) & 0xffu;
            arg = eq ? __builtin_ctz(eq) : 0;
        } else {
            const float stay = __shfl(cand, 8 * g + g), move = __shfl(cand, 8 * g + g - nbase);
            const bool moved = move > stay;
            v = moved ? move : stay;
            arg = moved ? g - nbase : g;
        }
        const unsigned long long onehot = __ballot(j == arg);
        if (lane == 0) tbw[slot] = onehot;
        pv = __shfl(v, 8 * j);
    };

    for (int c = 0; c < nchunk; c++) {
        const int c0 = c * kVitChunk, n = min(kVitChunk, Tb - c0);
        if (lane < kVitChunk / 8 / 32) careful[lane] = 0u;
        const int ng = min(ngroup_all - c0 / 8, n / 8);      // whole groups in this chunk
        // ---- 1. forward
        float ring[3][8];
        auto fetch_group = [&](float (&r)[8], int gi) {        // group gi of this chunk: blocks c0 + 8 gi ..; clamped, never predicated
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int blk = min(c0 + 8 * gi + k, Tb - 1);
                const float x = T[(size_t)blk * Ps + ((k & 1) ? i_hi : i_lo)];
                r[k] = ((k & 1) ? v_hi : v_lo) ? x : NEG;            // entries that do not exist: -inf, off the chain
            }
        };
        auto run_group = [&](float (&r)[8], int gi) {
            float cur[8];
#pragma unroll
            for (int k = 0; k < 8; k++) cur[k] = r[k];
            fetch_group(r, gi + 3);
            bool bad = TOPO == 1 ? !(fabsf(pv) < INFINITY) : (pv != pv);      // (run-length: an all -inf destination keeps traceback 0 -- only the literal scan knows)
#pragma unroll
            for (int k = 0; k < 8; k++) bad = bad || (((k & 1) ? v_hi : v_lo) && !(fabsf(cur[k]) < INFINITY));
            const int slot0 = 8 * gi;
            if (__builtin_expect(__ballot(bad) != 0ull, 0)) {
                if (lane == 0) careful[gi >> 5] |= 1u << (gi & 31);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    // odd blocks were fetched in the hi layout: lane 8 to + from wants what lane 8 from + to holds
                    const float x = (k & 1) ? __shfl(cur[k], 8 * j + g) : cur[k];         // (masked entries arrive as -inf: literal_step masks them again)
                    literal_step(x, slot0 + k);
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                // (a masked entry is -inf and may "equal" an all -inf maximum of a flop destination: the decoder below only looks at
                // the stay bit of those; flip destinations have no masked entries)
                const float ca = cur[k] + pv;
                const float vr = max_lo3(ca);                                     // row form
                const unsigned long long ea = __ballot(ca == vr);
                const float cb = cur[k + 1] + vr;
                pv = max_hi3(cb);                                                 // column form
                const unsigned long long eb = __ballot(cb == pv);
                if (lane == 0) { tbw[slot0 + k] = ea; tbw[slot0 + k + 1] = eb; }
            }
        };
        fetch_group(ring[0], 0); fetch_group(ring[1], 1); fetch_group(ring[2], 2);
        for (int gi = 0; gi < ng; gi += 3) {
            run_group(ring[0], gi);
            if (gi + 1 < ng) run_group(ring[1], gi + 1);
            if (gi + 2 < ng) run_group(ring[2], gi + 2);
        }
        if (8 * ng < n && lane == 0) careful[ng >> 5] |= 1u << (ng & 31);
        for (int i = 8 * ng; i < n; i++) literal_step(T[(size_t)(c0 + i) * Ps + i_lo], i);          // the last blocks of the read (< 8)
        __syncthreads();
        // ---- 2. ballots -> traceback bytes (byte s of the word = the state before state s)
        for (int i = lane; i < n; i += 64) {
            const unsigned long long w = tbw[i];
            const bool lit = (careful[i >> 8] >> ((i >> 3) & 31)) & 1u;
            const bool hi_layout = (i & 1) && !lit;
            unsigned long long out = 0ull;
#pragma unroll
            for (int s2 = 0; s2 < ns; s2++) {
                int arg;
                // bits of destination s2 by from-state, whichever layout the word has
                unsigned byte;
                if (!hi_layout) byte = (unsigned)(w >> (8 * s2)) & 0xffu;
                else {
                    const unsigned long long col = (w >> s2) & 0x0101010101010101ull;
                    byte = (unsigned)((col * 0x0102040810204080ull) >> 56);           // bit 8 f of col -> bit f
                }
                if constexpr (TOPO == 1) {
                    if (s2 < nbase) {
                        arg = 0;
                        bool found = false;
#pragma unroll
                        for (int pp = 0; pp < 8; pp++) { const int f = (pp >> 1) + 4 * (pp & 1); if (!found && ((byte >> f) & 1u)) { arg = f; found = true; } }
                    } else arg = (((byte >> s2) & 1u) && !((byte >> (s2 - nbase)) & 1u)) ? s2 : s2 - nbase;      // stay only if strictly greater than move
                } else {
                    if (s2 < nbase) arg = byte ? __builtin_ctz(byte) : 0;
                    else arg = ((byte >> s2) & 1u) ? s2 : s2 - nbase;
                }
                out |= (unsigned long long)arg << (8 * s2);
            }
            tbw[i] = out;
        }
        __syncthreads();
        if (c + 1 < nchunk) {                                 // a longer read: this chunk's traceback words leave LDS
            for (int i = lane; i < n; i += 64) tbg[c0 + i] = tbw[i];
            __syncthreads();
        }
    }
    // final score and state: first maximum (pv is in column form: lanes 0..7 hold the states)
    float score = __shfl(pv, 0);
    int last = 0;
    for (int s2 = 1; s2 < ns; s2++) {
        const float v = __shfl(pv, s2);
        if (v > score) { score = v; last = s2; }
    }
    if (lane == 0) { score_out[blockIdx.x] = score; if (TOPO == 0) qp[0] = NAN; }
    // ---- 3. traceback, last chunk first (its words are still in LDS)
    for (int c = nchunk - 1; c >= 0; c--) {
        const int c0 = c * kVitChunk, n = min(kVitChunk, Tb - c0), c1 = c0 + n;
        if (c != nchunk - 1) {
            __syncthreads();
            for (int i = lane; i < n; i += 64) tbw[i] = tbg[c0 + i];
            __syncthreads();
        }
        const int seg = (n + 63) / 64;
        const int s0 = min(lane * seg, n), s1 = min(s0 + seg, n);          // this lane's blocks [s0, s1)
        // map of the segment: state after block s1 - 1 (= path[s1]) -> state before block s0 (= path[s0]), 4 bits a state
        unsigned p[ns];
#pragma unroll
        for (int s2 = 0; s2 < ns; s2++) p[s2] = s2;
        for (int i = s1 - 1; i >= s0; i--) {
            const unsigned long long w = tbw[i];
#pragma unroll
            for (int s2 = 0; s2 < ns; s2++) p[s2] = (unsigned)(w >> (8 * p[s2])) & 7u;
        }
        unsigned map = 0u;
#pragma unroll
        for (int s2 = 0; s2 < ns; s2++) map |= p[s2] << (4 * s2);
        // end state of every segment: lane 63's is `last`, lane k's is what segment k + 1 maps its own end state to -- a scalar chain
        const int last_in = __builtin_amdgcn_readfirstlane(last);
        int myend = 0;
        {
            int e = last_in;
#pragma unroll
            for (int k = 63; k >= 0; k--) {
                myend = (lane == k) ? e : myend;
                e = (int)((unsigned)__builtin_amdgcn_readlane((int)map, k) >> (4 * e)) & 7;
            }
            last = e;                                         // = path[c0]: the end state of the chunk before
        }
        if (lane == 0) path_lds[n] = (uint8_t)last_in;
        {
            int e = myend;
            for (int i = s1 - 1; i >= s0; i--) {
                e = (int)((tbw[i] >> (8 * e)) & 7ull);
                path_lds[i] = (uint8_t)e;
            }
        }
        __syncthreads();
        if constexpr (TOPO == 1) {                             // decode.c:1000-1006: path[blk] = the state after block blk; this engine's extra slot holds 0 / NaN
            if (c1 == Tb && lane == 0) { pth[Tb] = 0; qp[Tb] = NAN; }
            for (int i = lane; i < n; i += 64) { pth[c0 + i] = path_lds[i + 1]; qp[c0 + i] = NAN; }
        } else {
        if (c1 == Tb && lane == 0) pth[Tb] = path_lds[n];
        for (int i = lane; i < n; i += 64) {
            const int from = path_lds[i], to = path_lds[i + 1];
            pth[c0 + i] = from;
            const int idx = (to < nbase) ? (to * ns + from) : (off + from);      // trans_lookup, decode.c:104-114
            qp[c0 + i + 1] = T[(size_t)(c0 + i) * Ps + idx];
        }
        }
    }
}

// ---- the same for the 10-state (5-base) models ---------------------------------------------------------------------------
// States 0..7 in the square, 8 and 9 in a second vector at positions 3 and 4 of the main vector's form, as in fb_chain<10>.  Per block
// and lane up to four candidates -- c1 (square), c2 (from state 8 / 9 into a flip destination), d1 / d2 (stay / move of flop 8, 9) -- two
// reductions, three ballots: c1 == max, c2 == max, d1 == max.  decode.c's scan order makes the winner of a flip destination the lowest
// from-state among equals (all of c1's before c2's), of a flop destination "stay unless the move is strictly greater".
constexpr int kVit10Chunk = 1024;

__global__ void __launch_bounds__(64)
k_viterbi10x(const float *__restrict__ M, uint8_t *__restrict__ tbbuf, int *__restrict__ path, float *__restrict__ qpath,
             float *__restrict__ score_out, int TbS, const int *__restrict__ tbs, ReadMap map) {
    FFHIP_CHAIN_PRIO_SET();
    constexpr int NS = 10, Ps = 60, nbase = 5, off = 50, PAD = 0;
    __shared__ unsigned long long tb0[kVit10Chunk], tb1[kVit10Chunk], tb2[kVit10Chunk];    // ballots, then traceback bytes of states 0..7 | 8, 9
    __shared__ unsigned careful[kVit10Chunk / 8 / 32];
    __shared__ uint8_t path_lds[kVit10Chunk + 1];
    __shared__ float cand[64];
    __shared__ float pvs[16];
    const int lane = threadIdx.x, g = lane >> 3, j = lane & 7;
    const float *T = M + map.row0(blockIdx.x, TbS) * Ps;
    unsigned long long *tbg = (unsigned long long *)(tbbuf + map.row0(blockIdx.x, TbS) * kMaxState);       // 16 bytes a block
    int *pth = path + map.row1(blockIdx.x, TbS);
    float *qp = qpath + map.row1(blockIdx.x, TbS);
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;
    if (Tb <= 0) return;
    const float NEG = -INFINITY;
    // score entry of "from in -> to out", or -1
    auto coef = [&](int out, int in) { return (out < NS && in < NS && ff_has<NS>(out, in)) ? ff_entry<NS>(out, in) : -1; };
    const bool xi_lo = (j == 3 || j == 4), xo_lo = (g == 3 || g == 4);
    // lo step: out-position g, in-position j; hi step: out-position j, in-position g
    const int ie[2][4] = { { coef(g, j), xi_lo ? coef(g, j + 5) : -1, (xo_lo && xi_lo) ? coef(g + 5, j + 5) : -1, xo_lo ? coef(g + 5, j) : -1 },
                           { coef(j, g), xo_lo ? coef(j, g + 5) : -1, (xo_lo && xi_lo) ? coef(j + 5, g + 5) : -1, xi_lo ? coef(j + 5, g) : -1 } };
    const int nchunk = (Tb + kVit10Chunk - 1) / kVit10Chunk;
    const int ngroup_all = Tb / 8;
    float pv = 0.0f, px = 0.0f;                             // column form: state lane & 7; states 8, 9 in lanes with lane & 7 == 3, 4

    // the literal block, decode.c's scan on the values: every entry's candidate through LDS, one lane per state; writes the traceback
    // bytes themselves (its group is marked in `careful`)
    auto literal_step = [&](int blk, int slot) {
        if (lane < 8) pvs[lane] = pv;
        if (lane == 3 || lane == 4) pvs[lane + 5] = px;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
        const float x = T[(size_t)blk * Ps + min(lane, Ps - 1)];
        cand[lane] = lane < Ps ? x + pvs[lane % NS] : NEG;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
        float best = 0.0f;
        int arg = 0;
        if (lane < nbase) {
            best = cand[lane * NS]; arg = 0;
            for (int f = 1; f < NS; f++) { const float sc = cand[lane * NS + f]; if (sc > best) { best = sc; arg = f; } }
        } else if (lane < NS) {
            const float stay = cand[off + lane], move = cand[off + lane - nbase];
            best = stay; arg = lane;
            if (move > stay) { best = move; arg = lane - nbase; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
        if (lane < NS) { pvs[lane] = best; ((uint8_t *)(lane < 8 ? &tb0[slot] : &tb1[slot]))[lane & 7] = (uint8_t)arg; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
        pv = pvs[j];
        px = (j == 3) ? pvs[8] : pvs[9];                    // j = 3 -> state 8, j = 4 -> state 9 (other lanes: unused)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
    };

    for (int c = 0; c < nchunk; c++) {
        const int c0 = c * kVit10Chunk, n = min(kVit10Chunk, Tb - c0);
        if (lane < kVit10Chunk / 8 / 32) careful[lane] = 0u;
        const int ng = min(ngroup_all - c0 / 8, n / 8);
        // ---- 1. forward
        float ring[2][8][4];
        auto fetch_group = [&](float (&r)[8][4], int gi) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int blk = min(c0 + 8 * gi + k, Tb - 1);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int idx = ie[k & 1][e];
                    const float x = T[(size_t)blk * Ps + max(idx, PAD)];
                    r[k][e] = idx >= 0 ? x : NEG;                // entries that do not exist: -inf, off the chain
                }
            }
        };
        auto run_group = [&](float (&r)[8][4], int gi) {
            float cur[8][4];
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int e = 0; e < 4; e++) cur[k][e] = r[k][e];
            fetch_group(r, gi + 2);
            bool bad = pv != pv || (xi_lo && px != px);
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int e = 0; e < 4; e++) bad = bad || (ie[k & 1][e] >= 0 && !(fabsf(cur[k][e]) < INFINITY));
            const int slot0 = 8 * gi;
            if (__builtin_expect(__ballot(bad) != 0ull, 0)) {
                if (lane == 0) careful[gi >> 5] |= 1u << (gi & 31);
                for (int k = 0; k < 8; k++) literal_step(c0 + slot0 + k, slot0 + k);
                return;
            }
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                // lo step: pv, px in column form
                const float c1 = cur[k][0] + pv, c2 = cur[k][1] + px, d1 = cur[k][2] + px, d2 = cur[k][3] + pv;
                const float vr = max_lo3(max2_raw(c1, c2)), xr = max_lo3(max2_raw(d1, d2));      // row form
                const unsigned long long a1 = __ballot(c1 == vr), a2 = __ballot(c2 == vr), a3 = __ballot(d1 == xr);
                // hi step
                const float h1 = cur[k + 1][0] + vr, h2 = cur[k + 1][1] + xr, k1 = cur[k + 1][2] + xr, k2 = cur[k + 1][3] + vr;
                pv = max_hi3(max2_raw(h1, h2)); px = max_hi3(max2_raw(k1, k2));                    // column form
                const unsigned long long b1 = __ballot(h1 == pv), b2 = __ballot(h2 == pv), b3 = __ballot(k1 == px);
                if (lane == 0) {
                    tb0[slot0 + k] = a1; tb1[slot0 + k] = a2; tb2[slot0 + k] = a3;
                    tb0[slot0 + k + 1] = b1; tb1[slot0 + k + 1] = b2; tb2[slot0 + k + 1] = b3;
                }
            }
        };
        fetch_group(ring[0], 0); fetch_group(ring[1], 1);
        for (int gi = 0; gi < ng; gi += 2) {
            run_group(ring[0], gi);
            if (gi + 1 < ng) run_group(ring[1], gi + 1);
        }
        if (8 * ng < n && lane == 0) careful[ng >> 5] |= 1u << (ng & 31);
        for (int i = 8 * ng; i < n; i++) literal_step(c0 + i, i);          // the last blocks of the read (< 8)
        __syncthreads();
        // ---- 2. ballots -> traceback bytes: byte s of (tb0 | tb1) = the state before state s
        for (int i = lane; i < n; i += 64) {
            if ((careful[i >> 8] >> ((i >> 3) & 31)) & 1u) continue;          // literal blocks hold their bytes already
            const unsigned long long w1 = tb0[i], w2 = tb1[i], w3 = tb2[i];
            const bool hi_layout = i & 1;
            // bit of (out-position o, in-position i2) in a ballot
            auto bit = [&](unsigned long long w, int o, int i2) { return (unsigned)(w >> (hi_layout ? 8 * i2 + o : 8 * o + i2)) & 1u; };
            unsigned long long o0 = 0ull, o1 = 0ull;
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++) {
                int arg;
                if (s2 < nbase) {               // flip: lowest from-state among the equal candidates: 0..7 in c1, then 8, 9 in c2
                    arg = 0;
                    bool found = false;
#pragma unroll
                    for (int f = 0; f < 8; f++) if (!found && bit(w1, s2, f)) { arg = f; found = true; }
                    if (!found && bit(w2, s2, 3)) { arg = 8; found = true; }
                    if (!found && bit(w2, s2, 4)) { arg = 9; found = true; }
                } else if (s2 < 8) {            // flop in the square: stay unless the move is strictly greater
                    arg = bit(w1, s2, s2) ? s2 : s2 - nbase;
                } else {                        // flop 8, 9: d1 (stay) at (position s2 - 5, position s2 - 5)
                    arg = bit(w3, s2 - 5, s2 - 5) ? s2 : s2 - nbase;
                }
                if (s2 < 8) o0 |= (unsigned long long)arg << (8 * s2);
                else o1 |= (unsigned long long)arg << (8 * (s2 - 8));
            }
            tb0[i] = o0; tb1[i] = o1;
        }
        __syncthreads();
        if (c + 1 < nchunk) {                                 // a longer read: this chunk's traceback bytes leave LDS
            for (int i = lane; i < n; i += 64) { tbg[2 * (size_t)(c0 + i)] = tb0[i]; tbg[2 * (size_t)(c0 + i) + 1] = tb1[i]; }
            __syncthreads();
        }
    }
    // final score and state: first maximum over states 0..9
    if (lane < 8) pvs[lane] = pv;
    if (lane == 3 || lane == 4) pvs[lane + 5] = px;
    __syncthreads();
    float score = pvs[0];
    int last = 0;
    for (int s2 = 1; s2 < NS; s2++) {
        const float v = pvs[s2];
        if (v > score) { score = v; last = s2; }
    }
    if (lane == 0) { score_out[blockIdx.x] = score; qp[0] = NAN; }
    // ---- 3. traceback, last chunk first
    auto before = [&](int i, int e) {                          // state before block i given the state e after it
        const unsigned long long w = e < 8 ? tb0[i] : tb1[i];
        return (int)(w >> (8 * (e & 7))) & 15;
    };
    for (int c = nchunk - 1; c >= 0; c--) {
        const int c0 = c * kVit10Chunk, n = min(kVit10Chunk, Tb - c0), c1 = c0 + n;
        if (c != nchunk - 1) {
            __syncthreads();
            for (int i = lane; i < n; i += 64) { tb0[i] = tbg[2 * (size_t)(c0 + i)]; tb1[i] = tbg[2 * (size_t)(c0 + i) + 1]; }
            __syncthreads();
        }
        const int seg = (n + 63) / 64;
        const int s0 = min(lane * seg, n), s1 = min(s0 + seg, n);
        int p[NS];
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++) p[s2] = s2;
        for (int i = s1 - 1; i >= s0; i--) {
            const unsigned long long w0 = tb0[i], w1 = tb1[i];
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++) p[s2] = (int)((p[s2] < 8 ? w0 : w1) >> (8 * (p[s2] & 7))) & 15;
        }
        unsigned long long map = 0ull;
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++) map |= (unsigned long long)p[s2] << (4 * s2);
        const int last_in = __builtin_amdgcn_readfirstlane(last);
        int myend = 0;
        {
            int e = last_in;
            const int mlo = (int)(unsigned)map, mhi = (int)(unsigned)(map >> 32);
#pragma unroll
            for (int k = 63; k >= 0; k--) {
                myend = (lane == k) ? e : myend;
                const unsigned long long mk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mhi, k) << 32) | (unsigned)__builtin_amdgcn_readlane(mlo, k);
                e = (int)(mk >> (4 * e)) & 15;
            }
            last = e;
        }
        if (lane == 0) path_lds[n] = (uint8_t)last_in;
        {
            int e = myend;
            for (int i = s1 - 1; i >= s0; i--) { e = before(i, e); path_lds[i] = (uint8_t)e; }
        }
        __syncthreads();
        if (c1 == Tb && lane == 0) pth[Tb] = path_lds[n];
        for (int i = lane; i < n; i += 64) {
            const int from = path_lds[i], to = path_lds[i + 1];
            pth[c0 + i] = from;
            const int idx = (to < nbase) ? (to * NS + from) : (off + from);      // trans_lookup, decode.c:104-114
            qp[c0 + i + 1] = T[(size_t)(c0 + i) * Ps + idx];
        }
    }
}

void launch_viterbi10x(hipStream_t s, const float *score_mat, uint8_t *tb, int *path, float *qpath, float *score, int nread, int Tb, const int *tbs, ReadMap map) {
    hipLaunchKernelGGL(k_viterbi10x, dim3(nread), dim3(64), 0, s, score_mat, tb, path, qpath, score, Tb, tbs, map);
}

void launch_viterbi8x(hipStream_t s, const float *score_mat, uint8_t *tb, int *path, float *qpath, float *score, int nread, int Tb, const int *tbs, ReadMap map) {
    hipLaunchKernelGGL(k_viterbi8x<0>, dim3(nread), dim3(64), 0, s, score_mat, tb, path, qpath, score, Tb, tbs, map);
}
// decode_crf_runlength for nbase = 4, stride 40 (param: shape / scale rows + 32 transition scores a block)
void launch_rle_viterbi8x(hipStream_t s, const float *param, uint8_t *tb, int *path, float *qpath, float *score, int nread, int Tb, const int *tbs) {
    hipLaunchKernelGGL(k_viterbi8x<1>, dim3(nread), dim3(64), 0, s, param, tb, path, qpath, score, Tb, tbs, ReadMap());
}

}  // namespace ffhip
