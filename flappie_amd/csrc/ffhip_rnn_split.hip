// ffhip_rnn_split.hip -- persistent LSTM / GRUmod layer on the 16-bit matrix pipes with fp32-grade products.
//
// Same layer as k_lstm_fused (ffhip_rnn_persist.hip; lstm_forward/lstm_backward + lstm_step, layers.c:877-1026)
// but the two GEMMs of a step, Wi x(t) and sW h(t-1), run as 16-bit MFMAs over a split of BOTH operands (ffhip_split.hpp):
//
//   default       v * 2^e = h0 + h1 in fp16 (round to nearest even), products w0x0 + w0x1 + w1x0 -- three
//                 `v_mfma_f32_16x16x32_f16` per fp32 multiply-add block, 4 B per value; the power-of-two scales (activations
//                 2^12, weights per matrix, one exponent S per layer) keep both slices normal; error of a K = 768 dot product
//                 against fp64 as an fp32 GEMM's (tests/test_split_numerics.py);
//   -DFFHIP_SPLIT_BF16X3   round 1's form: three bf16 slices (exact), six products, 6 B per value -- kept as the comparison
//                 build (profiles/r02_bf16x3_*).  The layout comments below say "NS slices" for both.
//
// Activations travel between layers ALREADY SPLIT, so that the split is computed once by the lane that produces
// a value and not by each of its 32 consumers:
//
//   split layout   A[t][rt][c = k/32][s = 0..NS-1][lane 0..63] 16 B   (H * 2 NS bytes per read and block)
//                  lane l = (kq = l>>4, r = l&15) holds slice s of k = 32c + 8kq + 0..7 of read r: exactly the
//                  B operand of v_mfma_f32_16x16x32_f16 / _bf16, one 1 KiB coalesced load per (chunk, slice).
//   weights        W[mat][ut][c][s][lane] 16 B: A operand, lane l = (row i = l&15 of unit tile ut, kq) holds slice s
//                  of W[16ut + i][32c + 8kq + 0..7]; rows unit-major/gate-minor as everywhere else.
//
// Work decomposition (H = 128 N, N = 1..4; H = 384 is the headline shape; template parameter TS below for the variants):
//   * one GROUP of 32 workgroups per PAIR of read tiles (32 reads); 256 CUs = 8 groups = 256 reads per launch, one
//     workgroup of 8 waves per CU (the weights of a CU, 48 rows x 768 k x 2 NS B = 147 KiB (221 for bf16x3), live in
//     VGPRs).  Member m owns N unit tiles (4N hidden units x 4 gates).  At N <= 2 two workgroups share a CU: one read
//     tile per group (TS = 1) up to 256 reads per launch, the pair form again for full launches of 512 reads;
//   * waves 0-3 ("x waves", low priority) hold the input weights of the member's rows, K split four ways, and compute
//     the projection Wi x(t+1) one step AHEAD, under the hand-off latency of step t; x(t+2) is prefetched into
//     registers right behind those MFMAs (it comes from HBM: a whole step of latency to hide);
//     waves 4-7 ("h waves", high priority) hold the recurrent weights and START their accumulators from the projection
//     partials of their K quarter (LDS), wait for h(t-1), add sW h(t-1), and leave the gate pre-activations in LDS;
//   * after one LDS-only barrier six waves (the h waves and x waves 0, 1) do the gate math of one 16 x 16 tile each
//     (4 gates of one unit of one read per lane, cell state in a register; the gate level GL is a template parameter: 2 = ffhip_math.hpp
//     logistic_hw, v_exp_f32 with a two-word exponent + v_rcp_f32 with a Newton step, the default since round 6 -- DESIGN.md section 3 --, 0 = the
//     *_lean forms, bit-identical to the reference-order arithmetic), split h(t) and store it; a second barrier closes the gate phase, so that no
//     MFMA stream starts next to a gate wave on its SIMD: v_mfma_f32_16x16x32_f16 holds the SIMD's VALU issue port for its four
//     passes, a gate wave beside a back-to-back stream makes NO progress (tools/dev/coissue_probe.cpp, profiles/r05_coissue_probe.txt;
//     round 2's "0.92-0.95 beside a dense stream" came from a stream with a loop branch behind every three MFMAs and is withdrawn,
//     DESIGN.md section 5.1.1); without the barrier the layer takes 2.40 instead of 2.29 ms;
//   * hand-off: the payload is the flag.  A producer lane writes the sentinel 0xFFFFFFFF (two bf16 NaNs -- never a
//     pair of slices of a finite value) to ITS slots of step t+3 when it publishes step t (and of steps 0..2 before
//     the group's start barrier): no host-side fill of the reused buffer.  A consumer wave first polls ONE dword per
//     producing gate wave of its K slice (a full sweep is 18 KiB per wave and the CU's path to L2 takes 64 B/clk),
//     then issues its sweep once; the sweep's chunks feed the MFMAs as they land (counted vmcnt waits) with the
//     sentinel check riding along, and only a failed check (a producer's store became visible line by line) falls
//     back to re-sweeping.  Plain stores when the 32 members verifiably share one XCD (= one L2), write-through
//     otherwise; every spin is bounded (abort word -> FFHIP_ETIMEOUT).
//
// Measured (MI355X, 256 reads x 800 blocks, H = 384): 2.29 ms per layer = 211 TFLOP/s of fp32-equivalent work (round 1's bf16x3
// form 2.84 ms, the f32 MFMA kernel 4.7 ms).  Per step ~6900 cycles: hand-off wait ~1300, MFMAs 1728 per SIMD, gate math ~1150,
// split / transpose / store ~500, barriers -- DESIGN.md section 5.1.1.
#include "ffhip_internal.hpp"
#include "ffhip_math.hpp"
#include "ffhip_split.hpp"
#include <stdlib.h>

namespace ffhip {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
constexpr int NS = kSplitNS;
#ifndef FFHIP_FORCE_SKEW
#define FFHIP_FORCE_SKEW 0       // 1: the test build that delays one wave in thirteen at every phase boundary of the step (see TL below)
#endif
#ifndef FFHIP_FORCE_RETRY
#define FFHIP_FORCE_RETRY 0      // 1: every member re-sweeps h(t-1) once at every 32nd step (tools/test_hooks/libffhip_resweep.so, tests/test_resweep_gpu.py)
#endif

struct SplitArgs {
    const v4u *Wp;            // [2][Ut][Hc][NS][64] 16 B; mat 0 = input weights, 1 = recurrent weights
    const float *bias;        // [16*Ut] permuted (4u + g)
    const unsigned char *xin; // layer input, split layout
    unsigned char *hout;      // layer output, split layout, pre-filled with the sentinel
    float *hout_f32;          // optional fp32 tile-interleaved copy of the output (for the CRF head), or nullptr
    unsigned *flags;          // [ngroup][32] check-in words (zeroed once, when the batch is created)
    unsigned epoch;           // this launch's number (> 0, < 2^27), see the start barrier
    unsigned *abort_word;
    int Tb, B16, H, rt0, nrt, backward, mode;
    float acc_scale;          // 2^S: the exponent both products of this layer carry (ffhip_split.hpp); the bias is added in that space
    int scale_exp;            // S
    int fast_gates;           // FFHIP_RUN_FAST_GATES: hardware exp / reciprocal in the gate phase (ffhip_math.hpp logistic_hw)
    int split_gate;           // LSTM, N = 3, pairs: gate tiles 4 and 5 are each worked by TWO x waves on different SIMDs (front / back)
    const int *tbs, *tbt;     // ragged batch (see PersistArgs)
    const unsigned *live;     // packed batch (several reads one behind the other in a slot, ffhip_batch_set_prepared_packed): word [t][read tile], bit r = slot r of the
                              // tile holds a block of a read at step t; elsewhere -- the gaps between two reads, the tail of a slot -- h and c are forced to zero, which is
                              // the next read's zero start in EITHER direction.  nullptr: one read a slot, dead beyond tbs[]
    unsigned long long *dbg;
};

struct SplitArgsOther {       // what the second batch of a paired launch brings of its own (k_lstm_split_pair)
    const unsigned char *xin; unsigned char *hout; float *hout_f32; unsigned *flags, *abort_word; const int *tbs, *tbt; const unsigned *live; unsigned epoch; int nwg0;
};

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// A flag word in LDS.  Through a plain `volatile int *` the access is a FLAT one -- address-space inference leaves volatile accesses alone --: it travels the vector
// memory pipe, and the compiler follows it with `s_waitcnt vmcnt(0)` (a flat access may alias LDS).  As LDS accesses (`ds_read_b32` / `ds_write_b32`) the flags cost less and
// the x waves' polling no longer slows their h waves down: c2 +1.2 %, h256 +1.9 %, rle +0.6 % (profiles/r05_lds_flags.txt, last block).  Round 5 built this form first, saw ONE
// read tile in ~1000 batches wrong with it -- always the pair's second tile -- and took it back; the cause turned out to be the re-sweep path (below: the faster h waves only took it
// more often), and with that fixed the form is clean in 12 000 batches that showed 12 failures before (tools/dev/front_order_diag.py).  The flat form is in tools/dev/experiments/lstm_split_round5_switches.patch.
#define LDSV(x) (*(volatile __attribute__((address_space(3))) int *)&(x))

// slice `which` of 4 values (each already multiplied by its power of two), packed as two dwords
template <bool CLAMP>
__device__ __forceinline__ v2u split4(v4f v, int which) {
    unsigned s[4];
    const float f[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
    for (int e = 0; e < 4; e++) {
        unsigned sl[NS];
        split_slices<CLAMP>(f[e], sl);
        s[e] = sl[0];
#pragma unroll
        for (int k = 1; k < NS; k++) s[e] = (which == k) ? sl[k] : s[e];
    }
    return (v2u){ s[0] | (s[1] << 16), s[2] | (s[3] << 16) };
}

__device__ __forceinline__ v4f mm(v4u a, v4u b, v4f c) {
#ifdef FFHIP_SPLIT_BF16X3
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
#endif
}

// The kept products of one K chunk (ffhip_split.hpp) for the N row tiles of a wave, smallest terms first.  Issue order is
// term-major, row tile minor: consecutive MFMAs write DIFFERENT accumulators, so the matrix pipe never waits for its own
// result (a dependent 16x16x32 MFMA issues every ~28 cycles, an independent one every 16).
template <int N, int NC = N>
__device__ __forceinline__ void mm6(const v4u (&w)[N][NC][NS], int cc, const v4u (&x)[NS], v4f (&acc)[N]) {
    constexpr int WS[kSplitNT] = FFHIP_SPLIT_TERMS_W, XS[kSplitNT] = FFHIP_SPLIT_TERMS_X;
#pragma unroll
    for (int term = 0; term < kSplitNT; term++)
#pragma unroll
        for (int j = 0; j < N; j++) acc[j] = mm(w[j][cc][WS[term]], x[XS[term]], acc[j]);
}

// the sentinel pair, materialised where it is stored (hoisted out of the step loop it costs two registers for the whole layer)
__device__ __forceinline__ v2u fresh_sentinel() {
    v2u v;
    asm volatile("v_mov_b32 %0, -1\n\tv_mov_b32 %1, -1" : "=v"(v.x), "=v"(v.y));
    return v;
}

// Running maximum (as unsigned) over the dwords of one chunk's operands: it is 0xFFFFFFFF at the end iff one of them was the
// sentinel.  Four v_max3_u32 per chunk and ONE register, evaluated where it stands (a chain of compares whose only use is the
// final test gets sunk there by the compiler -- and keeps every operand register alive until then).
__device__ __forceinline__ unsigned sentinel_max(unsigned seen, const v4u (&r)[kSplitNS]) {
#pragma unroll
    for (int s = 0; s < kSplitNS; s++) {
        seen = max(max(seen, r[s].x), r[s].y);
        seen = max(max(seen, r[s].z), r[s].w);
    }
    asm volatile("" : "+v"(seen));
    return seen;
}

// s_waitcnt vmcnt(n) with n known after unrolling (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// KIND 0 = LSTM (gate rows i, f, g, o), 1 = GRUmod (gate rows z, r, candidate, -; layers.c:571-715)
// TS = read tiles per group.  2: one workgroup per CU, a group serves a PAIR of read tiles (weights read once for 32 reads).
// 1: a group serves ONE read tile and TWO workgroups share a CU (<= 128 VGPRs, launch bound 4 waves per SIMD): two
// independent recurrences per CU, so that one's hand-off wait, sweep latency and gate phase run under the other's matrix
// work -- latency hiding by occupancy instead of by schedule; the two workgroups' gate phases also stop colliding on the same
// two SIMDs in lock step (six gate tiles on four SIMDs).
// DN ("dense", TS = 2 only): the PAIR form in 128 registers and 77 KiB of LDS, so that TWO such workgroups -- of one launch or of the
// launches of two batches in flight -- share a CU: four independent 16-read recurrences per CU.  The sweep of h(t-1) lands in
// LDS (as in the one-tile form at N = 3) one tile after the other through ONE set of accumulators, the recurrent partials are
// written over the landing zone they came from, the projection partials are single-buffered behind a per-(K quarter, tile)
// "consumed" flag, and the x waves load x(t+1) just in time (it is L2-warm) instead of a step ahead across the gate phase.
// PACK (GRUmod, H = 256, dense): the cell has three gates, and a unit tile of 4 units x 4 rows carries an empty row per unit -- a quarter of the
// MFMAs of both products.  Here a member owns 16 units (groups of 16) as THREE gate-major row tiles (z, r, candidate; no empty rows); inside a
// tile row 4 q + c is unit 4 c + q, so that lane (q, read) of a tile's accumulator holds units q, 4 + q, 8 + q, 12 + q in its components: gate wave
// (tile ts, component c) reads component c of the three gates and is, for the arithmetic and for the split / transpose / store of h(t), exactly
// the 4-unit gate tile (units 4 c .. 4 c + 3) of the other forms.  The candidate's projection half travels as a fourth partial tile.
// LIVE: the packed-batch forms (a.live; their own instantiations: the one-read-a-row kernels stay instruction for instruction what they were).
// GL: the gate level as a compile-time constant -- 2 = v_exp_f32 / v_rcp_f32 with a two-word exponent and a Newton step (the default since round 6), 0 = the reference's
// exp_ps and division replayed bit for bit (FFHIP_RUN_EXACT_GATES).  As a run-time branch on a.fast_gates both forms sat in every kernel: 17 spilled scalar registers
// in the paired form instead of 8, and c2 / h256 / c4 1.7 / 1.6 / 0.9 % slower (profiles/r06_gate_speed.txt).  Level 1 (the one-word exponent) runs as level 2: it was no faster.
template <int KIND, int N, int TS, bool DN, bool PACK = false, bool LIVE = false, int GL = 2>
__device__ __forceinline__ void lstm_split_body(const SplitArgs &a, const int block_index) {
    constexpr int fg = GL;
    static_assert(!DN || TS == 2, "the dense form is a pair form");
    static_assert(!PACK || (DN && N == 2), "the packed forms are dense forms at H = 256");
    // PACK with the LSTM (KIND 0): the same regrouping -- 16 members of 16 units, FOUR gate-major row tiles a member -- without a row to save: what
    // it buys there is 1024 reads a launch from two fat workgroups a CU instead of 768 from three (the sweep of h(t-1) enters a CU twice, not thrice)
    constexpr bool PG = PACK && KIND == 1;      // the GRUmod one: three row tiles, the candidate's projection half as a fourth partial tile
    constexpr int NRT = PACK ? (KIND == 1 ? 3 : 4) : N;      // row tiles of a wave
    constexpr int MT = PACK ? 4 : N;            // unit tiles (of 4 units) of a member
    // PACK has registers to spare (two workgroups a CU: 128) and spends 24 of them on the step's latency chain: the h waves take their projection
    // partials into registers at the top of the step and release them at once -- the x waves never wait for the "consumed" flags, and the barrier
    // behind the recurrent pass never waits for the x waves (-4.7 % layer time; profiles/r03_pack_experiments.txt).
    // (Measured and dropped there: the first tile's four gate jobs worked by the x waves UNDER the second tile's MFMAs, as soon as the h waves
    // have its partials in LDS -- a gate chain is twice as long as half a recurrent pass, the h waves then wait at the barrier for it: +11 %.)
    __shared__ v4f px[DN ? 1 : 2][4][TS][NRT][64];     // projection partials, double-buffered (DN: single): [step parity][K quarter][tile of the group][unit tile][lane]
    __shared__ v4f ph_[DN ? 1 : 4][DN ? 1 : TS][DN ? 1 : N][64];        // gate pre-activations by K quarter: projection partial + recurrent partial (DN: in the landing zone)
    __shared__ int pxc[4][2];               // DN: step (+1) whose projection partial of (K quarter, tile) the h wave has consumed
    __shared__ v4f sbias[MT][4];             // bias of my rows: [unit tile][unit in tile] x 4 gates
    __shared__ unsigned short gsl[8][NS][16][4];   // per gate wave: bf16 slices of its tile's h(t), [slice][read][unit]
    __shared__ float gf32[8][16][4];        // per gate wave: fp32 h(t), [read][unit] (last layer's copy for the CRF head)
    __shared__ int lds_abort;
    __shared__ int lds_fast;
    __shared__ float cx[2][64];             // split gate tiles: cell state c(t) from the front wave to the back wave of tiles 4 and 5
    __shared__ float ox[2][64];      // ... and the output gate o(t), evaluated by the front wave as well
    __shared__ int cxflag[2];               // ... and the step it belongs to (+1)
    // HL: the sweep of h(t-1) LANDS IN LDS (buffer_load ... lds: no destination registers) and feeds the MFMAs through ds_read_b128.
    // The one-tile kernel at N = 3 then fits 128 registers: TWO workgroups -- two independent recurrences -- share a CU.
    constexpr bool HL = (TS == 1 && N == 3) || DN;
    __shared__ v4u hland[HL ? 4 : 1][HL ? TS : 1][HL ? N : 1][NS][64];      // per h wave: its K slice of h(t-1), [tile][chunk][slice][lane]
    // partials of K quarter w for gate tile (ts, j): 64 x 16 B
    auto ph_at = [&](int w, int ts, int j) -> v4f * {
        if constexpr (DN) return (v4f *)&hland[w][ts][0][0][0] + j * 64;      // over the (consumed) first chunks of that tile's landing zone
        else return &ph_[w][ts][j][0];
    };
    constexpr int G = PACK ? 16 : 32, Hc = 4 * N, Ut = 32 * N;
    constexpr size_t tileB = (size_t)Hc * NS * 1024;      // bytes of one (t, read tile) in the split layout
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 7);
    const bool xw = wave < 4;
    const int kw = wave & 3;
    const int ngroup = (a.nrt + TS - 1) / TS;
    int g, m;
    {
        const int b = block_index;
        if ((ngroup & 7) == 0) { const int xcd = b & 7, j = b >> 3; g = xcd + 8 * (j / G); m = j % G; }
        else { g = b / G; m = b % G; }
    }
    const int rtA = a.rt0 + TS * g;
    const bool haveB = TS == 2 && (2 * g + 1 < a.nrt);
    const int TbA = a.tbt ? a.tbt[rtA] : a.Tb;
    const int TbB = haveB ? (a.tbt ? a.tbt[rtA + 1] : a.Tb) : 0;
    const int Tb = TbA > TbB ? TbA : TbB;                 // steps of this pair of read tiles
    if (Tb <= 0) return;                                  // empty slots only (uniform for the whole group)
    const int ntl = PACK ? __builtin_amdgcn_readfirstlane((TbB > 0) ? 2 : 1) : ((TbB > 0) ? 2 : 1);      // (PACK: said to be scalar -- a test on it lived in a spilled vector register)
    const int ut0 = m * MT;
    if (threadIdx.x == 0) lds_abort = (__hip_atomic_load(a.abort_word, RLX_AGENT) != 0u) ? 1 : 0;      // an earlier layer of this batch gave up: leave at once
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 4 * MT) sbias[(threadIdx.x - 64) >> 2][threadIdx.x & 3] = *(const v4f *)(a.bias + (size_t)ut0 * 16 + (threadIdx.x - 64) * 4) * (KIND == 1 ? 1.0f : a.acc_scale);      // LSTM: the bias joins the accumulators in their scaled space (exact)
    const int q = lane >> 4, rl = lane & 15;
    auto step_t = [&](int i) { return a.backward ? Tb - 1 - i : i; };
    // Gate tiles: the pair has ntl*N <= 6 of them; tile g6 = ts*N + j.  The h waves take tiles 0..3, x waves 0 and 1 take
    // tiles 4 and 5: at most two gate waves per SIMD, and each is a single dependency chain (two chains interleaved by
    // the hardware on one SIMD run in ~2500 cycles, two tiles back to back in one wave in ~3300).
    // With six tiles (LSTM, H = 384, a pair) two SIMDs would carry two whole gate chains and two SIMDs one -- and the gate
    // phase is bound by VALU issue (~270 instruction slots per tile), so it would last as long as the loaded SIMDs need.  Tiles
    // 4 and 5 are therefore each worked by TWO x waves that sit on different SIMDs: the FRONT wave (x wave 0 / 1, the SIMDs of
    // h waves 0 / 1) evaluates the input, forget and candidate gates and the cell state, hands c(t) over through LDS, and the
    // BACK wave (x wave 2 / 3, the SIMDs of h waves 2 / 3) evaluates the output gate while it waits, then tanh(c), h(t), the
    // split and the store: every SIMD carries ~1.5 tiles' worth.  Same operations on the same values: results are bit-identical.
    constexpr bool SGK = (KIND == 0 && N == 3 && TS == 2);
    const bool sg = SGK && ntl == 2 && a.split_gate != 0;
    const bool sg_front = sg && xw && wave < 2, sg_back = sg && xw && wave >= 2;
    const int g6 = PACK ? wave : (xw ? (sg ? 4 + (wave & 1) : 4 + wave) : kw);      // PACK: 2 tiles x 4 components = the 8 waves
    const bool gate_wave = !(sg && xw) && g6 < ntl * MT;     // works a whole tile
    const bool store_wave = gate_wave || sg_back;            // publishes a tile's h(t)
    const int my_gts = g6 / MT, my_gj = g6 % MT;             // (PACK: my_gj = the component = unit tile of the member)
    if (threadIdx.x < 2) cxflag[threadIdx.x] = 0;
    if (threadIdx.x < 8) pxc[threadIdx.x >> 1][threadIdx.x & 1] = 0;
    // where quarter-wave q of a gate wave stores slice q of its 4 units x 16 reads: 8 bytes at k = 4*ut .. 4*ut+3
    auto out_off = [&](int gj) { const int ut = ut0 + gj; return (unsigned)((((ut >> 3) * NS + q) * 64 + ((ut & 7) >> 1) * 16 + rl) * 16 + (ut & 1) * 8); };
    auto out_tile = [&](int t, int gts) { return a.hout + ((size_t)t * a.B16 + (rtA + gts)) * tileB; };
    auto store_wt = [&](unsigned char *tile, unsigned off, v2u v) {      // write-through: visible to every XCD
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)tile, 0, (int)tileB, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(v, wr, off, 0, 16 /*sc1*/);
    };
    auto store_plain = [&](unsigned char *tile, unsigned off, v2u v) {   // the group shares one L2.  (Scalar base + 32-bit lane offset as
        // above: a flat store keeps a 64-bit address per lane alive across the gate phase, which the 128-register forms cannot afford)
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)tile, 0, (int)tileB, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(v, wr, off, 0, 0);
    };
    // The output doubles as the hand-off flag, and the kernel arms it itself: every producer lane writes the sentinel
    // to ITS slots of steps 0..AHEAD-1 here -- before the group's start barrier below -- and to step i+AHEAD when it
    // publishes step i.  Stores of one lane to one address stay in order, so its sentinel can never overtake or
    // undercut its own data; a consumer looks at step i+AHEAD only after it has seen every producer's step i+AHEAD-1,
    // i.e. at least AHEAD-1 whole steps after that sentinel was written.  No host-side fill of the (reused) buffer.
    constexpr int AHEAD = 3;
    const v2u sentinel2 = { kSplitSentinel, kSplitSentinel };
    if (store_wave && q < NS)
        for (int k = 0; k < AHEAD && k < Tb; k++) store_wt(out_tile(step_t(k), my_gts), out_off(my_gj), sentinel2);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): my sentinels are in L2 ...
    __syncthreads();                          // ... and so are those of the other waves, before this member checks in
    if (threadIdx.x < 64) {
        // start barrier of the group (also tells whether all 32 members share one XCD, i.e. one L2)
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // a check-in word is (launch epoch << 5) | (XCC id + 1): what earlier launches left behind carries an older epoch and
        // counts as "not here yet", so the words need no clearing between launches (one fill kernel less per layer)
        xcc = (a.epoch << 5) | ((xcc & 0xfu) + 1u);
        unsigned *ids = a.flags + (size_t)g * G;
        if (lane == 0) __hip_atomic_store(ids + m, xcc, RLX_AGENT);
        unsigned v = xcc;
        for (unsigned spin = 0; spin < 2000000u; spin++) {
            v = (lane < G) ? __hip_atomic_load(ids + lane, RLX_AGENT) : xcc;
            if (__all((v >> 5) == a.epoch)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!__all((v >> 5) == a.epoch) && lane == 0) { __hip_atomic_store(a.abort_word, 1u, RLX_AGENT); lds_abort = 1; }      // a member never arrived
        const int fast_l = (a.mode == 0 && __all(v == xcc)) ? 1 : 0;
        if (lane == 0) lds_fast = fast_l;
    }
    // K chunks of this wave: quarter (kw + m) & 3 of the 4N chunks, walked from a member-dependent offset.  Any assignment
    // works (all partials are summed); this one spreads the 32 members' simultaneous sweeps of the same 24N KiB per read
    // tile over different lines and L2 channels instead of marching through them in lock step.
    int chunk[N];
#pragma unroll
    for (int cc = 0; cc < N; cc++) chunk[cc] = ((kw + m) & 3) * N + (cc + (PACK ? (m >> 1) : (m >> 2))) % N;      // (PACK: the order of the 8-unit members 2 m, 2 m + 1 of the other forms)
    // resident weights of this wave: rows of my N unit tiles, my N chunks, three slices
    v4u wf[NRT][N][NS];
    {
        // (PACK: a.Wp is the gate-major pack, NRT row tiles a member, NRT * G of them a matrix)
        const v4u *wp = a.Wp + (size_t)(xw ? 0 : 1) * (PACK ? NRT * G : Ut) * Hc * NS * 64;
        const int rt_first = PACK ? NRT * m : ut0;
#pragma unroll
        for (int j = 0; j < NRT; j++)
#pragma unroll
            for (int cc = 0; cc < N; cc++)
#pragma unroll
                for (int s = 0; s < NS; s++)
                    wf[j][cc][s] = wp[(((size_t)(rt_first + j) * Hc + chunk[cc]) * NS + s) * 64 + lane];
    }
    __syncthreads();
    if (lds_abort) return;
    const bool fast = lds_fast != 0;
    // the h waves are the critical path: their MFMAs go first, the projection fills the gaps (without priorities: +9 %)
    if (xw) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);

    const unsigned lane_off = (unsigned)lane * 16u;
    auto tile_ptr = [&](const unsigned char *base, int t, int ts) { return base + ((size_t)t * a.B16 + (rtA + ts)) * tileB; };
    auto raw_barrier = [&]() {               // LDS-only barrier: no vmcnt drain, prefetches stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

#if defined(FFHIP_PHASES)
    // Where every wave's step goes, in the kernel as it runs in production (round 5; tools/dev/phases.py): the time between two stamps is ADDED to
    // a per-wave word in LDS (one ds_add_u32 of lane 0, nobody waits for it) and the sums leave the kernel once, at its end -- 256 bytes of LDS, no
    // global traffic in the loop, both workgroups of a CU still resident.  Phase k = the stretch that ENDS at TL(k): 0 loop turn-around, 1 the h waves'
    // hand-off poll, 2 the wave's matrix work (x: loads + projection + flag wait + px; h: sweep + MFMAs + partials), 3 waiting at barrier 1, 4 the gate
    // phase, 5 waiting at barrier 2.
    __shared__ unsigned tl_acc[8][8];
    if (threadIdx.x < 64) tl_acc[threadIdx.x >> 3][threadIdx.x & 7] = 0u;
    unsigned tl_prev = (unsigned)__builtin_readcyclecounter();
#define TL(k) do { const unsigned now_ = (unsigned)__builtin_readcyclecounter(); if (lane == 0) __hip_atomic_fetch_add(&tl_acc[wave][(k)], now_ - tl_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); tl_prev = now_; } while (0)
#elif FFHIP_FORCE_SKEW
    // The test build that shakes the step's interleavings (tools/test_hooks/libffhip_skew.so, tests/test_resweep_gpu.py; round 6): at every phase boundary of every
    // role's loop -- top of the step, behind the h waves' poll, before / behind barrier 1, behind the gate phase, behind barrier 2 -- one wave in thirteen, rotating
    // with the step, the site, the wave and the group member, sits out ~3000 cycles (a third of a step).  Every flag protocol of the kernel (hand-off sentinels,
    // "consumed" words, the split gate tiles' c(t) flag, LDS landing zones against partial sums) must give the release library's bits with any wave late anywhere.
#define TL(k) do { if ((((unsigned)i * 5u + (unsigned)(k) * 3u + (unsigned)wave + (unsigned)m) % 13u) == 0u) __builtin_amdgcn_s_sleep(48); } while (0)
#else
#define TL(k) do { } while (0)
#endif

    // ---- gate math of one 16 x 16 tile (4 units x 4 gates x 16 reads; layers.c:1005-1025) and the store of its h(t), already
    // split.  ph holds the gate pre-activations Wi x + sW h by K quarter.
    // out of the accumulators' scaled space: x * 2^-S as v_ldexp_f32 with a SCALAR exponent (exact either way; a multiplication by
    // the float 2^-S gets packed, and the packed form keeps the factor splatted in a vector register pair for the whole layer)
    const int neg_exp = -a.scale_exp;
    auto unscale4 = [&](v4f v) -> v4f { return (v4f){ __builtin_ldexpf(v.x, neg_exp), __builtin_ldexpf(v.y, neg_exp), __builtin_ldexpf(v.z, neg_exp), __builtin_ldexpf(v.w, neg_exp) }; };
    // h(t) of one tile -> the split layout (and the hand-off): Split h ONCE, in the lane that owns it, and transpose through a
    // wave-private LDS patch: lane (unit q, read rl) writes its slices to [slice][read][unit]; quarter-wave q then reads the
    // 8 bytes [slice q][read rl][units 0..3] -- the packed operand piece it stores.  (Four ds_bpermute + a 4-value split in
    // every lane cost ~3x the VALU work.)
    auto publish_h = [&](int i, int gts, int gj, float h) {
        const int t = step_t(i);
        // The lane-dependent addresses of this block are recomputed every step from an OPAQUE copy of the lane number: hoisted out
        // of the step loop (where the compiler puts anything loop-invariant) they are six registers held for the whole layer, and
        // the 128-register forms of this kernel spill them; a dozen integer instructions per step cost nothing beside that.
        unsigned lv = (unsigned)lane;
        asm volatile("" : "+v"(lv));
        const unsigned q_ = lv >> 4, rl_ = lv & 15u;
        unsigned sl_[NS];
        split_slices(h * split_pow2(kSplitExpH), sl_);        // |h| <= 1: in range without a clamp
        unsigned short (*gs)[16][4] = gsl[wave];
#pragma unroll
        for (int k = 0; k < NS; k++) gs[k][rl_][q_] = (unsigned short)sl_[k];
        if (a.hout_f32) gf32[wave][rl_][q_] = h;
        asm volatile("" ::: "memory");                        // LDS operations of one wave execute in order
        const int ut = ut0 + gj;
        if (q_ < (unsigned)NS) {
            const unsigned off = (unsigned)((((ut >> 3) * NS) * 64 + ((ut & 7) >> 1) * 16) * 16 + (ut & 1) * 8) + q_ * 1024u + rl_ * 16u;      // = out_off(gj)
            const v2u sl = *(const v2u *)&gs[q_][rl_][0];
            unsigned char *tp_out = out_tile(t, gts);
            if (fast) {                                    // the group shares one L2: plain stores
                store_plain(tp_out, off, sl);              // (the data first: it is what the consumers wait for)
                if (i + AHEAD < Tb) store_plain(out_tile(step_t(i + AHEAD), gts), off, fresh_sentinel());
            } else {
                store_wt(tp_out, off, sl);
                if (i + AHEAD < Tb) store_wt(out_tile(step_t(i + AHEAD), gts), off, fresh_sentinel());
            }
        } else if (q_ == (unsigned)NS && a.hout_f32) {
            const v4f hv = *(const v4f *)&gf32[wave][rl_][0];
            __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.hout_f32 + ((size_t)t * a.B16 + (rtA + gts)) * (size_t)(Ut * 64)), 0, Ut * 256, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, hv), wr, rl_ * 16u, ut * 256, 0);
        }
    };
    auto gate_tile = [&](int i, int gts, int gj, float &c, int my_tb) {
        const int t = step_t(i);
        float h;
        if constexpr (PACK && KIND == 0) {
            // as the LSTM branch below, on component gj of the four partial tiles {i, f, g, o} of read tile gts; quarters in the other forms' order (see below)
            const v4f b = sbias[gj][q];
            float s0 = b.x, s1 = b.y, s2 = b.z, s3 = b.w;
            const int rot = m + (gj >> 1);
            unsigned lv = (unsigned)lane;
            asm volatile("" : "+v"(lv));
            const char *pbase = (const char *)&hland[0][gts][0][0][0] + lv * 16u + (unsigned)gj * 4u;
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) {
                const float *pz = (const float *)(pbase + ((w2 + rot) & 3) * (int)sizeof(hland[0]));
                s0 += pz[0]; s1 += pz[64 * 4]; s2 += pz[128 * 4]; s3 += pz[192 * 4];
            }
            const v4f s = unscale4((v4f){ s0, s1, s2, s3 });
            if (fg) {
                const float forget = logistic_hw(s.y, fg) * c;
                const float update = logistic_hw(s.x, fg) * tanh_hw(s.z, fg);
                c = forget + update;
                h = logistic_hw(s.w, fg) * tanh_hw(c, fg);
            } else {
                const ffv4 L = logistic_ref4_lean((ffv4){ s.x, s.y, s.z + s.z, s.w });
                const float tanh_g = (L.z + L.z) - 1.0f;
                const float forget = L.y * c;
                const float update = L.x * tanh_g;
                c = forget + update;
                h = L.w * tanh_ref_lean(c);
            }
        } else if constexpr (PACK) {
            // as the GRUmod branch below, on component gj of the four partial tiles {z, r, u = (sW h)_c, x_c = (Wi x)_c} of read tile gts
            // The K quarters are added in the order the other forms add them for these units (there they belong to member 2 m + (gj >> 1), whose
            // wave w2 holds quarter (w2 + 2 m + (gj >> 1)) & 3; here wave w holds quarter (w + m) & 3): the results are bit-identical.
            float sz = 0.f, sr = 0.f, su = 0.f, sx = 0.f;
            const int rot = m + (gj >> 1);
            unsigned lv = (unsigned)lane;          // (an opaque copy: the four addresses are recomputed per step instead of living in -- spilled -- registers)
            asm volatile("" : "+v"(lv));
            const char *pbase = (const char *)&hland[0][gts][0][0][0] + lv * 16u + (unsigned)gj * 4u;
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) {
                const float *pz = (const float *)(pbase + ((w2 + rot) & 3) * (int)sizeof(hland[0]));
                sz += pz[0]; sr += pz[64 * 4]; su += pz[128 * 4]; sx += pz[192 * 4];
            }
            sz = __builtin_ldexpf(sz, neg_exp); sr = __builtin_ldexpf(sr, neg_exp); su = __builtin_ldexpf(su, neg_exp); sx = __builtin_ldexpf(sx, neg_exp);
            const v4f b = sbias[gj][q];
            if (fg) {
                const float z = logistic_hw(sz + b.x, fg), r = logistic_hw(sr + b.y, fg);
                const float hbar = tanh_hw(r * su + (sx + b.z), fg);
                h = z * c + (1.0f - z) * hbar;
            } else {
                const ffv2 L = logistic_ref2_lean((ffv2){ sz + b.x, sr + b.y });
                float hbar = L.y * su + (sx + b.z);
                hbar = tanh_ref_lean(hbar);
                h = L.x * c + (1.0f - L.x) * hbar;
            }
            c = h;
        } else if (KIND == 1) {
            // GRUmod (layers.c:690-714).  The accumulator rows are {z: x + h parts, r: x + h parts, u = (sW h)_c, x_c = (Wi x)_c}:
            // the candidate's two halves must stay apart (r multiplies only the recurrent one), and a unit's fourth row is free.
            // `c` carries this lane's own h(t-1).
            v4f s = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) s = s + ph_at(w2, gts, gj)[lane];
            s = unscale4(s);                               // out of the scaled space (a power of two: exact)
            const v4f b = sbias[gj][q];
            if (fg) {
                const float z = logistic_hw(s.x + b.x, fg), r = logistic_hw(s.y + b.y, fg);
                const float hbar = tanh_hw(r * s.z + (s.w + b.z), fg);
                h = z * c + (1.0f - z) * hbar;
            } else {
                const ffv2 L = logistic_ref2_lean((ffv2){ s.x + b.x, s.y + b.y });
                float hbar = L.y * s.z + (s.w + b.z);
                hbar = tanh_ref_lean(hbar);
                h = L.x * c + (1.0f - L.x) * hbar;
            }
            c = h;
        } else {
            v4f s = sbias[gj][q];
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) s = s + ph_at(w2, gts, gj)[lane];
            s = unscale4(s);                               // out of the scaled space (a power of two: exact)
            // (the _lean forms give the bits of logistic_ref4 / tanh_ref with ~50 fewer instructions: ffhip_math.hpp)
            if (fg) {
                const float forget = logistic_hw(s.y, fg) * c;
                const float update = logistic_hw(s.x, fg) * tanh_hw(s.z, fg);
                c = forget + update;
                h = logistic_hw(s.w, fg) * tanh_hw(c, fg);
            } else {
                const ffv4 L = logistic_ref4_lean((ffv4){ s.x, s.y, s.z + s.z, s.w });
                const float tanh_g = (L.z + L.z) - 1.0f;
                const float forget = L.y * c;
                const float update = L.x * tanh_g;
                c = forget + update;
                h = L.w * tanh_ref_lean(c);
            }
        }
        if (t >= my_tb) { h = 0.0f; c = 0.0f; }          // beyond this read's end (ragged batch)
        publish_h(i, gts, gj, h);
    };
    // the two halves of a split gate tile (see `sg` above): the same arithmetic as gate_tile's LSTM branch, operation for operation
    auto gate_front = [&](int i, int gts, int gj, float &c, int my_tb) {
        v4f s = sbias[gj][q];
#pragma unroll
        for (int w2 = 0; w2 < 4; w2++) s = s + ph_at(w2, gts, gj)[lane];
        s = unscale4(s);
        float forget, update;
        // The output gate is evaluated HERE, as the fourth lane of the packed logistic gate_tile uses (round 5): four logistics in packed instructions cost what two
        // packed + one scalar evaluation cost, so this wave's instruction count stays and the back wave loses its own logistic, its four partial sums and their LDS
        // reads -- ~45 VALU instructions a tile off a SIMD whose MFMA and VALU time add (profiles/r05_coissue_probe.txt).  The same operations on the same values.
        float og;
        if (fg) {
            forget = logistic_hw(s.y, fg) * c;
            update = logistic_hw(s.x, fg) * tanh_hw(s.z, fg);
            og = logistic_hw(s.w, fg);
        } else {
            const ffv4 L = logistic_ref4_lean((ffv4){ s.x, s.y, s.z + s.z, s.w });
            const float tanh_g = (L.z + L.z) - 1.0f;
            forget = L.y * c;
            update = L.x * tanh_g;
            og = L.w;
        }
        ox[wave & 1][lane] = og;
        c = forget + update;
        if (step_t(i) >= my_tb) c = 0.0f;
        cx[wave & 1][lane] = c;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (LDS operations of a wave execute in order; the flag follows the data)
        if (lane == 0) LDSV(cxflag[wave & 1]) = i + 1;
    };
    auto gate_back = [&](int i, int gts, int gj, int my_tb) {
        while (LDSV(cxflag[wave & 1]) != i + 1) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const float c = cx[wave & 1][lane];
        const float o = ox[wave & 1][lane];
        float h = o * (fg ? tanh_hw(c, fg) : tanh_ref_lean(c));
        if (step_t(i) >= my_tb) h = 0.0f;
        publish_h(i, gts, gj, h);
    };
    int my_tb = 0;
    if (gate_wave || sg_front || sg_back) my_tb = a.tbs ? a.tbs[(rtA + my_gts) * 16 + rl] : a.Tb;
    float c = 0.0f;
    // Packed batch: this wave's gate tile is live at step i where its bit of live[t][read tile] is set.  One SCALAR load a step and gate wave, issued and waited for
    // at the top of the step (the h waves stand in their hand-off poll there, the x waves have slack) -- a vector load would join the counted vmcnt waits of the sweep.
    // The word becomes the bound the gate functions compare t with: everything (never dead) or nothing (dead), so that their code is the one-read-a-slot code.
    // (the word stays in a scalar register across the matrix phase; the lane's bound is formed in the gate phase)
    auto live_word = [&](int i) -> unsigned {
        unsigned w = 0u;
        if constexpr (LIVE) {
            const unsigned *lp = a.live + ((size_t)step_t(i) * a.B16 + (rtA + my_gts));
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(lp) : "memory");
        }
        return w;
    };
    auto tb_of = [&](unsigned w) -> int {
        if constexpr (!LIVE) return my_tb;
        unsigned lv = (unsigned)lane;
        asm volatile("" : "+v"(lv));
        return ((w >> (lv & 15u)) & 1u) ? 0x7fffffff : 0;
    };

    // The two roles run their own step loop (two barriers per step each), so that the register allocator sees each
    // role's live ranges alone.
    if (xw && DN) {
        // ---- x waves, dense form: per step and tile -- load x(step i+1) of my K quarter (24 registers; an L2 hit: the group touched
        // these lines three steps ago), project, wait until h wave kw has taken the previous partial of that tile (it always has), write
        // the new one.  Nothing is in flight across the gate phase, where x waves 0-3 work gate tiles 4 and 5.
        if constexpr (DN) {
        v4u xb[N][NS];
        auto load_x_tile = [&](int i, int ts) {
            __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)tile_ptr(a.xin, step_t(i), ts), 0, (int)tileB, 0x00020000);
#pragma unroll
            for (int cc = 0; cc < N; cc++) {
#pragma unroll
                for (int s = 0; s < NS; s++) xb[cc][s] = __builtin_amdgcn_raw_buffer_load_b128(rx, lane_off, ((chunk[cc] * NS + s) * 64) * 16, 0);
                if (cc + 1 < N) __builtin_amdgcn_s_sleep(1);
            }
        };
        // Both tiles' partials are computed BEFORE the wait for the h wave's "consumed" flags -- those are raised at the end of its
        // recurrent pass, and a projection of the second tile started only then would stand between the h waves and the barrier.
        // pieces of the NEXT step that leave behind the last MFMAs of this one, in flight across the gate phase: both where the registers
        // are there (GRUmod: 124), the first one otherwise (the LSTM form spills 9 registers with both)
        constexpr int XPRE = KIND == 1 ? 2 : 1;
        v4u xq[2][NS];
        auto ldq = [&](int i, int k, v4u (&dst)[NS]) {      // piece k = (tile, chunk) of x(step i)
            const int ts = (k / N < ntl) ? k / N : 0, cc = k % N;
            __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)tile_ptr(a.xin, step_t(i), ts), 0, (int)tileB, 0x00020000);
#pragma unroll
            for (int s = 0; s < NS; s++) dst[s] = __builtin_amdgcn_raw_buffer_load_b128(rx, lane_off, ((chunk[cc] * NS + s) * 64) * 16, 0);
        };
        auto project_step = [&](int i, int want) {           // x(step i) of both tiles -> px[0][kw][*]; want = the step (+1) whose partials must have been consumed (0: none)
            v4f acc[TS][NRT];
            if constexpr (PACK) {
                // x(t) of the four (tile, chunk) pieces through TWO 8-register buffers, the load of piece k + 2 issued behind the MFMAs of piece k: two exposed L2
                // round trips a step instead of four (the x waves closed every step of this form: the h waves waited 2960 of 11 640 cycles for them,
                // profiles/r05_phases.txt).  An absent second tile re-reads the first and its products are dropped: the outstanding-load count stays static.
                // Pieces 0 and 1 of the NEXT step leave behind the last MFMAs of this one (in flight across the gate phase).
                if constexpr (XPRE < 1) ldq(i, 0, xq[0]);
                if constexpr (XPRE < 2) ldq(i, 1, xq[1]);
#pragma unroll
                for (int ts = 0; ts < TS; ts++)
#pragma unroll
                    for (int j = 0; j < NRT; j++) acc[ts][j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int k = 0; k < TS * N; k++) {
                    mm6<NRT, N>(wf, k % N, xq[k & 1], acc[k / N]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 2 < TS * N) ldq(i, k + 2, xq[k & 1]);
                    else if (k + 2 - TS * N < XPRE) ldq(i + 1 < Tb ? i + 1 : i, k + 2 - TS * N, xq[k & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#pragma unroll
            for (int ts = 0; ts < TS; ts++) {
                if (ts >= ntl) continue;
                if constexpr (!(PACK && KIND == 0)) load_x_tile(i, ts);
#pragma unroll
                for (int j = 0; j < NRT; j++) acc[ts][j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
                if constexpr (PACK && KIND == 0) {
                    // 64 weight + 32 accumulator registers: x(t) comes a chunk at a time (8 registers; the x waves have the slack for the second wait)
                    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)tile_ptr(a.xin, step_t(i), ts), 0, (int)tileB, 0x00020000);
#pragma unroll
                    for (int cc = 0; cc < N; cc++) {
#pragma unroll
                        for (int s = 0; s < NS; s++) xb[0][s] = __builtin_amdgcn_raw_buffer_load_b128(rx, lane_off, ((chunk[cc] * NS + s) * 64) * 16, 0);
                        mm6<NRT, N>(wf, cc, xb[0], acc[ts]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                for (int cc = 0; cc < N; cc++) mm6<NRT, N>(wf, cc, xb[cc], acc[ts]);
                }
            }
#pragma unroll
            for (int ts = 0; ts < TS; ts++) {
                if (ts >= ntl) continue;
                if (want > 0)
                    for (unsigned spin = 0; LDSV(pxc[kw][ts]) != want && 0 == LDSV(lds_abort) && spin < 40000000u; spin++) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int j = 0; j < NRT; j++) px[0][kw][ts][j][lane] = acc[ts][j];
            }
        };
        constexpr int WARM = 3;
        constexpr int LPM = (Hc * NS * 8 * TS + G - 1) / G;  // 128-byte lines of the group's x(step) per member
        unsigned touched = 0, sink = 0;
        auto touch_x = [&](int i) {                          // L2 warming, spread over the group (see the classic loop below)
            unsigned lt = (unsigned)lane;
            if constexpr (PACK && KIND == 0) asm volatile("" : "+v"(lt));      // (an opaque copy: no 64-bit lane address held -- spilled -- across the step)
            const int line = m * LPM + (int)lt;
            unsigned t = 0;
            if (wave == 3 && lane < LPM && line < ntl * Hc * NS * 8 && i < Tb)
                t = *(const unsigned *)(tile_ptr(a.xin, step_t(i), 0) + (size_t)line * 128);
            touched = t;
        };
        if constexpr (PACK) { if constexpr (XPRE >= 1) ldq(0, 0, xq[0]); if constexpr (XPRE >= 2) ldq(0, 1, xq[1]); }
        project_step(0, 0);
        touch_x(1);
        sink ^= touched;
        touch_x(2);
        raw_barrier();                                       // px(0) is in LDS before any h wave starts from it
        for (int i = 0; i < Tb; i++) {
            TL(0);
            const unsigned lw = live_word(i);
            if (i + 1 < Tb) project_step(i + 1, i + 1);
            sink ^= touched;
            touch_x(i + WARM);
            TL(2);
            raw_barrier();
            TL(3);
            const int aborted = LDSV(lds_abort);      // (looked at behind the gate math: see the h waves' loop)
            if (sg_front) {
                __builtin_amdgcn_s_setprio(3);
                gate_front(i, my_gts, my_gj, c, tb_of(lw));
                __builtin_amdgcn_s_setprio(0);
            } else if (sg_back) {
                __builtin_amdgcn_s_setprio(3);           // (the back half publishes the tile's h(t): as much on the step's chain as the front half; -1.3 % launch time at c2)
                gate_back(i, my_gts, my_gj, tb_of(lw));
                __builtin_amdgcn_s_setprio(0);
            } else if (gate_wave) {
                if (PACK) __builtin_amdgcn_s_setprio(3);      // every wave works a gate job: the x waves' at the h waves' priority (-1.7 %)
                gate_tile(i, my_gts, my_gj, c, tb_of(lw));
                if (PACK) __builtin_amdgcn_s_setprio(0);
            }
            if (aborted) return;
            TL(4);
            raw_barrier();                                   // closes the gate phase
            TL(5);
        }
        if (sink == 0x9e3779b9u && a.Tb < 0) a.flags[0] = sink;
        }
    } else if (xw) {
        // ---- x waves: projection of step i+1 under the hand-off latency of step i.  x(step i+2) is prefetched into
        // registers right after the MFMAs that consumed x(step i+1): a whole step ahead of its use (it comes from HBM),
        // and never in the gate phase, where the issue of 18 KiB of loads per wave (the CU's path to L2 takes 64 B/clk)
        // would delay a gating x wave and with it the critical path.
        v4u xb[TS][N][NS];
        auto load_x = [&](int i) {
            const int t = step_t(i);
#pragma unroll
            for (int ts = 0; ts < TS; ts++) {
                if (ts >= ntl) continue;
                const v4u *p = (const v4u *)(tile_ptr(a.xin, t, ts) + lane_off);
#pragma unroll
                for (int cc = 0; cc < N; cc++) {
#pragma unroll
                    for (int s = 0; s < NS; s++) xb[ts][cc][s] = p[(chunk[cc] * NS + s) * 64];
                    __builtin_amdgcn_s_sleep(2);       // the x waves have slack: their prefetch trickles into the memory pipe instead of
                                                       // occupying it with an 18 KiB burst per wave
                }
            }
        };
        auto project = [&](int i) {           // xb holds x(step i): partial Wi x -> px[i & 1]
#pragma unroll
            for (int ts = 0; ts < TS; ts++) {
                if (ts >= ntl) continue;
                v4f acc[N];
#pragma unroll
                for (int j = 0; j < N; j++) acc[j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int cc = 0; cc < N; cc++)
                    if constexpr (!PACK) mm6<N>(wf, cc, xb[ts][cc], acc);
#pragma unroll
                for (int j = 0; j < N; j++) px[DN ? 0 : (i & 1)][kw][ts][j][lane] = acc[j];
            }
        };
        // L2 warming, spread over the group.  x is the previous layer's output, far larger than L2: the first of the 32 members
        // to ask for a line waits for HBM (~2 us) and the other 31 queue behind the same miss, so a prefetch issued one step
        // ahead arrives just in time or late, and while it is outstanding the CU's in-order memory pipe holds up the h waves'
        // sweep and the gate waves' stores.  Each member therefore TOUCHES 1/32 of the lines of x(step i+3) -- one dword
        // load of <= 18 lanes per step, issued behind its own prefetch -- so that the prefetches of step i+3 are L2 hits.
        constexpr int WARM = 3;
        constexpr int LPM = (Hc * NS * 8 * TS + 31) / 32;    // 128-byte lines of the group's x(step) per member
        unsigned touched = 0, sink = 0;
        auto touch_x = [&](int i) {
            const int line = m * LPM + lane;
            unsigned t = 0;                               // (not `touched` itself: keeping the old value would make this a use of the old load)
            if (wave == 3 && lane < LPM && line < ntl * Hc * NS * 8 && i < Tb)
                t = *(const unsigned *)(tile_ptr(a.xin, step_t(i), 0) + (size_t)line * 128);
            touched = t;
        };
        load_x(0);
        project(0);
        if (Tb > 1) load_x(1);
        touch_x(2);
        raw_barrier();                                       // px(0) is in LDS before any h wave starts from it
        for (int i = 0; i < Tb; i++) {
            TL(0);
            const unsigned lw = live_word(i);
            if (i + 1 < Tb) {
                project(i + 1);
                if (i + 2 < Tb) load_x(i + 2);
            }
            sink ^= touched;                                 // (keeps the touch a real load; it landed a step ago)
            touch_x(i + WARM);
            TL(2);
            raw_barrier();
            TL(3);
            const int aborted = LDSV(lds_abort);      // (looked at behind the gate math: see the h waves' loop)
            if constexpr (TS == 2) {                         // (one tile per group: its N <= 4 gate tiles all belong to h waves -- and said at
                                                             // compile time, so that no gate temporaries are live beside the prefetch)
                if (sg_front) {
                    __builtin_amdgcn_s_setprio(3);           // the back wave (and with it the closing barrier) waits for this c(t)
                    gate_front(i, my_gts, my_gj, c, tb_of(lw));
                    __builtin_amdgcn_s_setprio(0);
                } else if (sg_back) gate_back(i, my_gts, my_gj, tb_of(lw));
                else if (gate_wave) gate_tile(i, my_gts, my_gj, c, tb_of(lw));
            }
            if (aborted) return;
            TL(4);
            raw_barrier();                                   // closes the gate phase
            TL(5);
        }
        if (sink == 0x9e3779b9u && a.Tb < 0) a.flags[0] = sink;      // never true: the touches must not be optimised away
    } else {
        // ---- h waves: recurrent half of step i on top of the projection partial of my K quarter, then one gate tile.
        // Hand-off of h(step i-1): (1) a LIGHT poll -- one dword per producing gate wave of my K slice (16N lanes, one
        // load) until none is the sentinel; a full sweep is 24N KiB per wave and 64 B/clk per CU, far too heavy to
        // repeat; (2) ONE full sweep, software-pipelined two chunks ahead of the MFMAs that consume it, with the sentinel
        // check riding along; (3) only if that check fails (a producer's store instruction became visible line by line)
        // the classic re-sweep loop and a recomputation.
        constexpr int NPROD = 8 * N;                      // unit tiles (= producing gate waves' tiles per read tile) in my K slice
        constexpr int NCH = TS * N;                       // (tile, chunk) pairs of my K slice
        raw_barrier();                                        // matches the x waves' prologue barrier
        for (int i = 0; i < Tb; i++) {
            TL(0);
            const unsigned lw = live_word(i);
            v4f acc[TS][N];
            auto init_acc = [&]() {
#pragma unroll
                for (int ts = 0; ts < TS; ts++)
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        const v4f p = px[DN ? 0 : (i & 1)][kw][ts][j][lane];      // (an absent second tile: stale LDS, dropped)
                        acc[ts][j] = KIND == 1 ? (v4f){ p.x, p.y, 0.0f, p.z } : p;      // GRUmod: the projection's candidate row moves to the free row
                    }
            };
            if constexpr (!DN) init_acc();
            v4f pc[PG ? 2 : 1][3];             // PACK (GRUmod): the projection partials move to registers at the top of the step and are released at once
            if constexpr (PG) {
#pragma unroll
                for (int ts = 0; ts < 2; ts++)
#pragma unroll
                    for (int j = 0; j < 3; j++) pc[ts][j] = px[0][kw][ts][j][lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < 2) LDSV(pxc[kw][lane]) = i + 1;
            }
            if (i > 0) {
                const int tp = step_t(i - 1);
                const unsigned char *hp = tile_ptr(a.hout, tp, 0);          // the pair's two tiles are adjacent
                __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)hp, 0, (int)(TS * tileB), 0x00020000);
                bool timed_out = false;
                {
                    const int ul = lane % NPROD, pts = lane / NPROD;
                    const int put = ((kw + m) & 3) * NPROD + ul;
                    const bool act = pts < ntl;
                    const unsigned poff = (unsigned)(pts * (int)tileB + ((put >> 3) * NS * 64 + ((put & 7) >> 1) * 16 + 15) * 16 + (put & 1) * 8);
                    for (unsigned spin = 0; ; spin++) {
                        const unsigned v = act ? __builtin_amdgcn_raw_buffer_load_b32(rs2, poff, 0, 16 /*sc1*/) : 0u;
                        if (__all(v != kSplitSentinel)) break;
                        if (spin > 6000000u || (spin & 511u) == 511u) {
                            const unsigned ab = __hip_atomic_load(a.abort_word, RLX_AGENT);
                            if (ab != 0u || spin > 6000000u) { timed_out = true; break; }
                        }
                    }
                }
                TL(1);
                v4u raw[NCH][NS];
                // Both tiles of the pair are always swept and multiplied (an absent second tile re-reads the first one and
                // its products are dropped): a branch on ntl between the loads makes the outstanding-load count path
                // dependent, and the compiler then waits vmcnt(0) before the first MFMA instead of counting.
                const int offB = (TS > 1 && ntl > 1) ? (int)tileB : 0;
                auto load_chunk = [&](int k) {          // k = ts*N + cc
                    const int ts = k / N, cc = k % N;
#pragma unroll
                    for (int s = 0; s < NS; s++) {
                        raw[k][s] = __builtin_amdgcn_raw_buffer_load_b128(rs2, ts * offB + ((chunk[cc] * NS + s) * 64) * 16 + lane_off, 0, 16 /*sc1*/);
                    }
                };
                // acc += sW h over my K slice; false if a sentinel was among the operands
                auto recur = [&]() -> bool {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < NCH; k++) {
                        load_chunk(k);
                        // a short pause between the chunks: the CU's memory pipe takes the four h waves' requests in issue order,
                        // and a wave that fires its 18 loads back to back gets its data as one burst -- the last wave's first
                        // chunk would wait behind 54 KiB of the others'.  With this and the pause in the x waves' prefetch the
                        // layer takes 14.31 instead of 14.85 ms (same device, interleaved runs; `s_sleep 0` does the same: what
                        // counts is that the burst is broken, not the length of the pause).
                        if (k + 1 < NCH) __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int k = 0; k < NCH; k++) {
#pragma unroll
                        for (int s = 0; s < NS; s++) {
                            const v4u r = raw[k][s];
                            ok = ok && r.x != kSplitSentinel && r.y != kSplitSentinel && r.z != kSplitSentinel && r.w != kSplitSentinel;
                        }
                        if constexpr (!PACK) mm6<N>(wf, k % N, raw[k], acc[k / N]);
                        __builtin_amdgcn_sched_barrier(0);      // keep each chunk's check and MFMAs behind ITS loads only: the sweep streams under the MFMAs
                    }
                    return __all(ok) != 0;
                };
                // HL form of the sweep: 2N one-KiB DMA loads into this wave's landing zone; per chunk a counted wait, two ds_read_b128
                // (issued while the previous chunk's MFMAs are in the pipe) and the same check + products as above.  The waits are
                // written out: the compiler does not see that an LDS read depends on a buffer_load ... lds (left to itself it waits
                // for vmcnt(0) before the first one), and it may move nothing across these statements.
                auto recur_lds = [&]() -> bool {
                    unsigned seen = 0u;
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass rejects the LDS-DMA builtin, and then silently drops the kernel's stub)
                    if constexpr (HL && !DN) {
#pragma unroll
                        for (int k = 0; k < N; k++) {
#pragma unroll
                            for (int s = 0; s < NS; s++)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (__attribute__((address_space(3))) void *)&hland[kw][0][k][s][0], 16,
                                                                         lane_off, ((chunk[k] * NS + s) * 64) * 16, 0, 16 /*sc1*/);
                            if (k + 1 < N) __builtin_amdgcn_s_sleep(1);
                        }
                        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void *)&hland[kw][0][0][0][lane];
                        v4u r[2][NS];
                        auto fetch = [&](int k) {           // chunk k has landed -> issue its two LDS reads
                            wait_vmcnt_upto(NS * (N - 1 - k));
                            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024"
                                         : "=&v"(r[k & 1][0]), "=&v"(r[k & 1][1]) : "v"(la + (unsigned)(k * NS * 1024)) : "memory");
                        };
                        fetch(0);
#pragma unroll
                        for (int k = 0; k < N; k++) {
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[k & 1][0]), "+v"(r[k & 1][1]) :: "memory");
                            if (k + 1 < N) fetch(k + 1);
                            seen = sentinel_max(seen, r[k & 1]);
                            if constexpr (!PACK) mm6<N>(wf, k, r[k & 1], acc[0]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#endif
                    return __all(seen != kSplitSentinel) != 0;
                };
                // Dense form: both tiles' 2N KiB each land in this wave's zone; ONE set of accumulators takes tile A, then tile B (each
                // starting from the projection partial of that tile), and each tile's partials go over the first N KiB of its own,
                // consumed, landing zone.  false: a sentinel was among the operands -- everything is done again from the (still
                // unreleased) projection partials.
                auto recur_dn = [&]() -> bool {
                    unsigned seen = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
                    if constexpr (DN) {
#pragma unroll
                        for (int ts = 0; ts < 2; ts++)
#pragma unroll
                            for (int k = 0; k < N; k++) {
#pragma unroll
                                for (int s = 0; s < NS; s++)
                                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (__attribute__((address_space(3))) void *)&hland[kw][ts][k][s][0], 16,
                                                                             lane_off, ts * offB + ((chunk[k] * NS + s) * 64) * 16, 0, 16 /*sc1*/);
                                if (ts * N + k + 1 < 2 * N) __builtin_amdgcn_s_sleep(1);
                            }
                        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void *)&hland[kw][0][0][0][lane];
                        v4u r[2][NS];
                        auto fetch = [&](int c) {           // (tile, chunk) c = ts * N + k has landed -> issue its two LDS reads
                            wait_vmcnt_upto(NS * (2 * N - 1 - c));
                            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024"
                                         : "=&v"(r[c & 1][0]), "=&v"(r[c & 1][1]) : "v"(la + (unsigned)(c * NS * 1024)) : "memory");
                        };
                        fetch(0);
#pragma unroll
                        for (int ts = 0; ts < 2; ts++) {
                            v4f accd[NRT], xc = { 0.f, 0.f, 0.f, 0.f };
                            if constexpr (PACK && KIND == 0) {
#pragma unroll
                                for (int j = 0; j < NRT; j++) accd[j] = px[0][kw][ts][j][lane];
                            } else if constexpr (PACK) {          // z and r start from their projection partials; the candidate's two halves stay apart
                                accd[0] = pc[ts][0]; accd[1] = pc[ts][1]; accd[2] = (v4f){ 0.f, 0.f, 0.f, 0.f }; xc = pc[ts][2];
                            } else {
#pragma unroll
                                for (int j = 0; j < N; j++) {
                                    const v4f p = px[0][kw][ts][j][lane];      // (an absent second tile: stale LDS, dropped)
                                    accd[j] = KIND == 1 ? (v4f){ p.x, p.y, 0.0f, p.z } : p;
                                }
                            }
#pragma unroll
                            for (int k = 0; k < N; k++) {
                                const int c = ts * N + k;
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[c & 1][0]), "+v"(r[c & 1][1]) :: "memory");
                                if (c + 1 < 2 * N) fetch(c + 1);
                                seen = sentinel_max(seen, r[c & 1]);
                                mm6<NRT, N>(wf, k, r[c & 1], accd);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (ts < ntl) {
#pragma unroll
                                for (int j = 0; j < NRT; j++) ph_at(kw, ts, j)[lane] = accd[j];
                                if constexpr (PG) ph_at(kw, ts, 3)[lane] = xc;
                            }
                        }
                    }
#endif
                    return __all(seen != kSplitSentinel) != 0;
                };
                auto recur_any = [&]() -> bool { if constexpr (DN) return recur_dn(); else if constexpr (HL) return recur_lds(); else return recur(); };
                if (!timed_out) {
                    // gfx9 counts loads and stores on ONE counter and they complete out of order with respect to each other:
                    // while this wave's gate-phase stores of step i-1 may be pending the compiler can only wait vmcnt(0).
                    // They completed long ago (the poll above outlasts them) -- say so, and the sweep gets counted waits.
                    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
                    if (!recur_any() || (FFHIP_FORCE_RETRY && (i & 31) == 5)) {      // (FFHIP_FORCE_RETRY: the test build that takes the re-sweep path on purpose, below)
                        for (unsigned spin = 0;; spin++) {
                            if (spin > 3000000u || (spin & 255u) == 255u) {
                                const unsigned ab = __hip_atomic_load(a.abort_word, RLX_AGENT);
                                if (ab != 0u || spin > 3000000u) { timed_out = true; break; }
                            }
                            __builtin_amdgcn_s_sleep(1);
                            // Round 5: the pass that just failed wrote its partial sums over the landing zones LAST -- the second tile's the very last thing it did -- and
                            // the re-sweep's `buffer_load ... lds` pieces land in those zones again.  LDS writes of the DS path and of the DMA path are not ordered with
                            // each other: a ds_write still queued when the sweep goes out can land AFTER the piece that replaces it, and the pass then multiplies stale
                            // partial sums of the SECOND tile as if they were h(t-1) (no sentinel among them: the check passes).  One read tile wrong from that step
                            // on, once in ~10^4 launches of k_grumod_pack (whose x waves do not slow its h waves down: the sweep meets a half-visible step more often);
                            // taken on purpose (-DFFHIP_FORCE_RETRY) 199 launches of 200.  profiles/r05_pack_repeat.txt; tests/test_resweep_gpu.py keeps it fixed.
                            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                            if constexpr (!DN) init_acc();
                            if (recur_any()) break;
                        }
                    }
                }
                if (timed_out && lane == 0) { __hip_atomic_store(a.abort_word, 1u, RLX_AGENT); lds_abort = 1; }
            }
            if constexpr (DN) {
                if (i == 0) {                                    // h(-1) = 0: the gate pre-activations are the projection alone
                    if constexpr (PACK && KIND == 0) {
#pragma unroll
                        for (int ts = 0; ts < TS; ts++) {
                            if (ts >= ntl) continue;
#pragma unroll
                            for (int j = 0; j < NRT; j++) ph_at(kw, ts, j)[lane] = px[0][kw][ts][j][lane];
                        }
                    } else if constexpr (PACK) {
#pragma unroll
                        for (int ts = 0; ts < TS; ts++) {
                            if (ts >= ntl) continue;
                            ph_at(kw, ts, 0)[lane] = pc[ts][0];
                            ph_at(kw, ts, 1)[lane] = pc[ts][1];
                            ph_at(kw, ts, 2)[lane] = (v4f){ 0.f, 0.f, 0.f, 0.f };
                            ph_at(kw, ts, 3)[lane] = pc[ts][2];
                        }
                    } else {
                    init_acc();
#pragma unroll
                    for (int ts = 0; ts < TS; ts++) {
                        if (ts >= ntl) continue;
#pragma unroll
                        for (int j = 0; j < N; j++) ph_at(kw, ts, j)[lane] = acc[ts][j];
                    }
                    }
                }
                // the projection partials of this step are consumed: the x wave of my K quarter may write the next ones
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!PG && lane < 2) LDSV(pxc[kw][lane]) = i + 1;      // (the packed GRUmod form: released at the top of the step)
            } else {
#pragma unroll
                for (int ts = 0; ts < TS; ts++) {
                    if (ts >= ntl) continue;
#pragma unroll
                    for (int j = 0; j < N; j++) ph_at(kw, ts, j)[lane] = acc[ts][j];
                }
            }
            TL(2);
            raw_barrier();
            TL(3);
            // (the abort word is read here and looked at BEHIND the gate math: a read-wait-branch in front of it is ~100 cycles on the
            // chain of every step; results of an aborted launch are discarded anyway)
            const int aborted = LDSV(lds_abort);
            if (gate_wave) gate_tile(i, my_gts, my_gj, c, tb_of(lw));
            if (aborted) return;
            TL(4);
            // close the gate phase before ph is rewritten (and before the x waves' MFMAs start next to gate VALU work)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            TL(5);
        }
    }
#ifdef FFHIP_PHASES
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (a.dbg && lane < 8) {
        atomicAdd(a.dbg + wave * 8 + lane, (unsigned long long)tl_acc[wave][lane]);
        if (lane == 0) atomicAdd(a.dbg + 64 + wave, (unsigned long long)Tb);
    }
#endif
}

template <int KIND, int N, int TS, bool DN = false, bool LIVE = false, int GL = 2>
__global__ void __launch_bounds__(512, (DN && N <= 2) ? 6 : ((TS == 1 || N <= 2 || DN) ? 4 : 1))
k_lstm_split(SplitArgs a) { lstm_split_body<KIND, N, TS, DN, false, LIVE, GL>(a, (int)blockIdx.x); }

// the packed GRUmod form (H = 256): 128 registers, two workgroups a CU, 16 members a group
template <bool LIVE, int GL>
__global__ void __launch_bounds__(512, 4)
k_grumod_pack(SplitArgs a) { lstm_split_body<1, 2, 2, true, true, LIVE, GL>(a, (int)blockIdx.x); }
template <bool LIVE, int GL>
__global__ void __launch_bounds__(512, 4)
k_lstm_pack(SplitArgs a) { lstm_split_body<0, 2, 2, true, true, LIVE, GL>(a, (int)blockIdx.x); }

// The layer launches of TWO batches as one grid (ffhip_batch_run_pair): workgroups below nwg0 serve the first batch's read tiles, the
// others the second's -- every pointer and count of a batch comes from its own argument block, nothing is shared but the weights.
// For the dense form at H = 384: 2 x 256 reads = 16 groups = two workgroups on every CU, and ONE launch whose duration is the pair's.
template <int KIND, int N, int TS, bool DN, bool LIVE = false, int GL = 2>
__global__ void __launch_bounds__(512, (DN && N <= 2) ? 6 : ((TS == 1 || N <= 2 || DN) ? 4 : 1))
k_lstm_split_pair(SplitArgs a, SplitArgsOther o) {
    int bi = (int)blockIdx.x;
    if (bi >= o.nwg0) {                                           // uniform: the second batch differs in its buffers only (same model, same shape)
        bi -= o.nwg0;
        a.xin = o.xin; a.hout = o.hout; a.hout_f32 = o.hout_f32; a.flags = o.flags; a.abort_word = o.abort_word; a.tbs = o.tbs; a.tbt = o.tbt; a.live = o.live; a.epoch = o.epoch;
    }
    lstm_split_body<KIND, N, TS, DN, false, LIVE, GL>(a, bi);
}

// ---- recurrence only, behind the projection GEMM: shapes whose two weight matrices do not fit a CU's registers ------
// H = 256 * CPW (H = 512: CPW = 2).  The same group / hand-off / gate scheme as k_lstm_split, but all eight waves hold
// recurrent weights (K split eight ways, N = H/128 unit tiles of rows per member: 197 KiB per CU at H = 512) and every wave
// gates one of the pair's 2N <= 8 tiles; the gate pre-activations start from Xa = Wi x + b (k_inproj_split), one D-fragment
// per gate wave and step, prefetched a step ahead.  LSTM only.
struct RnnSplitArgs {
    const v4u *Wp;            // recurrent weights [Ut][Hc][NS][64] 16 B (the second matrix of the split pack)
    const v4f *xa;            // [Tb][B16][Ut][64] float4, D-fragment order, bias included
    unsigned char *hout;      // split layout
    float *hout_f32;
    unsigned *flags, *abort_word;
    int Tb, B16, rt0, nrt, backward, mode;
    float acc_scale;          // 2^S of sW h (ffhip_split.hpp); xa arrives unscaled
    const int *tbs, *tbt;
};

template <int N, int CPW>
__global__ void __launch_bounds__(512, 1)
k_rnn_split(RnnSplitArgs a) {
    constexpr int G = 32, Hc = 8 * CPW, Ut = 32 * N, NCH = 2 * CPW, NPROD = 8 * CPW;
    constexpr size_t tileB = (size_t)Hc * NS * 1024;
    __shared__ v4f ph[8][2][N][64];                  // recurrent partials by K eighth
    __shared__ unsigned short gsl[8][NS][16][4];
    __shared__ float gf32[8][16][4];
    __shared__ int lds_abort;
    __shared__ int lds_fast;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroup = (a.nrt + 1) >> 1;
    int g, m;
    {
        const int b = blockIdx.x;
        if ((ngroup & 7) == 0) { const int xcd = b & 7, j = b >> 3; g = xcd + 8 * (j / G); m = j % G; }
        else { g = b / G; m = b % G; }
    }
    const int rtA = a.rt0 + 2 * g;
    const bool haveB = (2 * g + 1 < a.nrt);
    const int TbA = a.tbt ? a.tbt[rtA] : a.Tb;
    const int TbB = haveB ? (a.tbt ? a.tbt[rtA + 1] : a.Tb) : 0;
    const int Tb = TbA > TbB ? TbA : TbB;
    if (Tb <= 0) return;
    const int ntl = (TbB > 0) ? 2 : 1;
    const int ut0 = m * N;
    if (threadIdx.x == 0) lds_abort = (__hip_atomic_load(a.abort_word, RLX_AGENT) != 0u) ? 1 : 0;      // an earlier layer of this batch gave up: leave at once
    const int q = lane >> 4, rl = lane & 15;
    auto step_t = [&](int i) { return a.backward ? Tb - 1 - i : i; };
    const bool gate_wave = wave < ntl * N;
    const int my_gts = wave / N, my_gj = wave % N;
    auto out_off = [&](int gj) { const int ut = ut0 + gj; return (unsigned)((((ut >> 3) * NS + q) * 64 + ((ut & 7) >> 1) * 16 + rl) * 16 + (ut & 1) * 8); };
    auto out_tile = [&](int t, int gts) { return a.hout + ((size_t)t * a.B16 + (rtA + gts)) * tileB; };
    auto store_wt = [&](unsigned char *tile, unsigned off, v2u v) {
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)tile, 0, (int)tileB, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(v, wr, off, 0, 16 /*sc1*/);
    };
    constexpr int AHEAD = 3;
    const v2u sentinel2 = { kSplitSentinel, kSplitSentinel };
    if (gate_wave && q < NS)
        for (int k = 0; k < AHEAD && k < Tb; k++) store_wt(out_tile(step_t(k), my_gts), out_off(my_gj), sentinel2);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (threadIdx.x < 64) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc = (xcc & 0xfu) + 1u;
        unsigned *ids = a.flags + (size_t)g * G;
        if (lane == 0) __hip_atomic_store(ids + m, xcc, RLX_AGENT);
        unsigned v = xcc;
        for (unsigned spin = 0; spin < 2000000u; spin++) {
            v = (lane < G) ? __hip_atomic_load(ids + lane, RLX_AGENT) : xcc;
            if (__all(v != 0u)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!__all(v != 0u) && lane == 0) { __hip_atomic_store(a.abort_word, 1u, RLX_AGENT); lds_abort = 1; }
        const int fast_l = (a.mode == 0 && __all(v == xcc)) ? 1 : 0;
        if (lane == 0) lds_fast = fast_l;
    }
    const int cbase = ((wave + m) & 7) * CPW;            // my K eighth: chunks cbase .. cbase+CPW-1
    v4u wf[N][CPW][NS];
#pragma unroll
    for (int j = 0; j < N; j++)
#pragma unroll
        for (int cc = 0; cc < CPW; cc++)
#pragma unroll
            for (int s = 0; s < NS; s++)
                wf[j][cc][s] = a.Wp[(((size_t)(ut0 + j) * Hc + (cbase + cc)) * NS + s) * 64 + lane];
    __syncthreads();
    if (lds_abort) return;
    const bool fast = lds_fast != 0;
    const unsigned lane_off = (unsigned)lane * 16u;
    auto raw_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    int my_tb = 0;
    if (gate_wave) my_tb = a.tbs ? a.tbs[(rtA + my_gts) * 16 + rl] : a.Tb;
    float c = 0.0f;
    auto xa_tile = [&](int i) { return a.xa[(((size_t)step_t(i) * a.B16 + (rtA + my_gts)) * Ut + ut0 + my_gj) * 64 + lane]; };
    v4f xa_next = gate_wave ? xa_tile(0) : (v4f){ 0.f, 0.f, 0.f, 0.f };
    for (int i = 0; i < Tb; i++) {
        v4f acc[2][N];
#pragma unroll
        for (int ts = 0; ts < 2; ts++)
#pragma unroll
            for (int j = 0; j < N; j++) acc[ts][j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
        const v4f xa_cur = xa_next;
        if (i > 0) {
            const int tp = step_t(i - 1);
            const unsigned char *hp = a.hout + ((size_t)tp * a.B16 + rtA) * tileB;
            __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)hp, 0, (int)(2 * tileB), 0x00020000);
            bool timed_out = false;
            {
                const int ul = lane % NPROD, pts = lane / NPROD;
                const int put = cbase * 8 + ul;
                const bool act = pts < ntl;
                const unsigned poff = (unsigned)(pts * (int)tileB + ((put >> 3) * NS * 64 + ((put & 7) >> 1) * 16 + 15) * 16 + (put & 1) * 8);
                for (unsigned spin = 0;; spin++) {
                    const unsigned v = act ? __builtin_amdgcn_raw_buffer_load_b32(rs2, poff, 0, 16 /*sc1*/) : 0u;
                    if (__all(v != kSplitSentinel)) break;
                    if (spin > 6000000u || (spin & 511u) == 511u) {
                        const unsigned ab = __hip_atomic_load(a.abort_word, RLX_AGENT);
                        if (ab != 0u || spin > 6000000u) { timed_out = true; break; }
                    }
                }
            }
            v4u raw[NCH][NS];
            const int offB = (ntl > 1) ? (int)tileB : 0;
            auto recur = [&]() -> bool {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NCH; k++) {
#pragma unroll
                    for (int s = 0; s < NS; s++)
                        raw[k][s] = __builtin_amdgcn_raw_buffer_load_b128(rs2, (k / CPW) * offB + (((cbase + k % CPW) * NS + s) * 64) * 16 + lane_off, 0, 16 /*sc1*/);
                    if (k + 1 < NCH) __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int k = 0; k < NCH; k++) {
#pragma unroll
                    for (int s = 0; s < NS; s++) {
                        const v4u r = raw[k][s];
                        ok = ok && r.x != kSplitSentinel && r.y != kSplitSentinel && r.z != kSplitSentinel && r.w != kSplitSentinel;
                    }
                    mm6<N, CPW>(wf, k % CPW, raw[k], acc[k / CPW]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                return __all(ok) != 0;
            };
            if (!timed_out) {
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see k_lstm_split
                if (!recur()) {
                    for (unsigned spin = 0;; spin++) {
                        if (spin > 3000000u || (spin & 255u) == 255u) {
                            const unsigned ab = __hip_atomic_load(a.abort_word, RLX_AGENT);
                            if (ab != 0u || spin > 3000000u) { timed_out = true; break; }
                        }
                        __builtin_amdgcn_s_sleep(1);
#pragma unroll
                        for (int ts = 0; ts < 2; ts++)
#pragma unroll
                            for (int j = 0; j < N; j++) acc[ts][j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
                        if (recur()) break;
                    }
                }
            }
            if (timed_out && lane == 0) { __hip_atomic_store(a.abort_word, 1u, RLX_AGENT); lds_abort = 1; }
        }
        if (gate_wave && i + 1 < Tb) xa_next = xa_tile(i + 1);      // behind the sweep in the memory pipe, a step ahead of its use
#pragma unroll
        for (int ts = 0; ts < 2; ts++) {
            if (ts >= ntl) continue;
#pragma unroll
            for (int j = 0; j < N; j++) ph[wave][ts][j][lane] = acc[ts][j];
        }
        raw_barrier();
        if (lds_abort) return;
        if (gate_wave) {
            const int t = step_t(i);
            v4f s = xa_cur * a.acc_scale;                    // into the accumulators' scaled space and back: powers of two, exact
            if (i > 0) {
#pragma unroll
                for (int w2 = 0; w2 < 8; w2++) s = s + ph[w2][my_gts][my_gj][lane];
            }
            s = s * (1.0f / a.acc_scale);
            const ffv4 L = logistic_ref4_lean((ffv4){ s.x, s.y, s.z + s.z, s.w });
            const float tanh_g = (L.z + L.z) - 1.0f;
            const float forget = L.y * c;
            const float update = L.x * tanh_g;
            c = forget + update;
            float h = L.w * tanh_ref_lean(c);
            if (t >= my_tb) { h = 0.0f; c = 0.0f; }
            {
                unsigned sl[NS];
                split_slices(h * split_pow2(kSplitExpH), sl);
#pragma unroll
                for (int k = 0; k < NS; k++) gsl[wave][k][rl][q] = (unsigned short)sl[k];
            }
            if (a.hout_f32) gf32[wave][rl][q] = h;
            asm volatile("" ::: "memory");
            const int ut = ut0 + my_gj;
            const unsigned off = out_off(my_gj);
            if (q < NS) {
                const v2u sl = *(const v2u *)&gsl[wave][q][rl][0];
                unsigned char *tp_out = out_tile(t, my_gts);
                if (fast) {
                    *(v2u *)(tp_out + off) = sl;
                    if (i + AHEAD < Tb) *(v2u *)(out_tile(step_t(i + AHEAD), my_gts) + off) = sentinel2;
                } else {
                    store_wt(tp_out, off, sl);
                    if (i + AHEAD < Tb) store_wt(out_tile(step_t(i + AHEAD), my_gts), off, sentinel2);
                }
            } else if (a.hout_f32) {
                const v4f hv = *(const v4f *)&gf32[wave][rl][0];
                *(v4f *)(a.hout_f32 + ((size_t)t * a.B16 + (rtA + my_gts)) * (size_t)(Ut * 64) + (size_t)ut * 64 + rl * 4) = hv;
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

bool rnn_split_supported(int kind, int H) { return kind == 0 && (H == 256 || H == 512); }
bool launch_rnn_split(hipStream_t s, const void *Wsplit, const float *xa, void *hout, float *hout_f32, unsigned *flags, unsigned *abort_word,
                      int Tb, int B16, int H, int rt0, int nrt, int backward, int mode, int scale_exp, const int *tbs, const int *tbt) {
    RnnSplitArgs a;
    a.acc_scale = split_pow2(scale_exp);
    const int Ut = H / 4, Hc = H / 32;
    a.Wp = (const v4u *)Wsplit + (size_t)Ut * Hc * NS * 64;      // second matrix of the pack = recurrent weights
    a.xa = (const v4f *)xa; a.hout = (unsigned char *)hout; a.hout_f32 = hout_f32; a.flags = flags; a.abort_word = abort_word;
    a.Tb = Tb; a.B16 = B16; a.rt0 = rt0; a.nrt = nrt; a.backward = backward; a.mode = mode; a.tbs = tbs; a.tbt = tbt;
    const int ngroup = (nrt + 1) / 2;
    if (H == 512) { hipLaunchKernelGGL((k_rnn_split<4, 2>), dim3(ngroup * 32), dim3(512), 0, s, a); return true; }
    if (H == 256) { hipLaunchKernelGGL((k_rnn_split<2, 1>), dim3(ngroup * 32), dim3(512), 0, s, a); return true; }
    return false;
}

// ---- input projection as a plain GEMM on split operands (shapes the layer kernel does not take, e.g. H = 512) -----
// Xa[nt][mt] = Wi[mt] . x[nt] + b, D-fragment order like k_inproj (ffhip_kernels.hip), products as six bf16 MFMA terms.
// A workgroup = 4 waves 2 (M) x 2 (N), each wave 4 x 6 tiles of 16 x 16; per K chunk of 32 a wave loads 12 + 18 KiB and
// issues 144 MFMAs (4 x 4 tiles: 24 KiB per 96 MFMAs sits right at the CU's 64 B/clk load path and measured 10 % slower).
__global__ void __launch_bounds__(256)
k_inproj_split(const unsigned char *__restrict__ in, float *__restrict__ xa, const v4u *__restrict__ Wp, const float *__restrict__ bias,
               int ntile, int Mt, int Hc, float acc_scale) {
    constexpr int TM = 4, TN = 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int nMblk = (Mt + 2 * TM - 1) / (2 * TM);
    const int nNblk = (ntile + 2 * TN - 1) / (2 * TN);
    // neighbouring workgroups share the activation tiles: walk M fastest
    const int L = blockIdx.x;
    const int mblk = L % nMblk, nblk = L / nMblk;
    if (nblk >= nNblk) return;
    const int mt0 = (mblk * 2 + wm) * TM, nt0 = (nblk * 2 + wn) * TN;
    const int kq = lane >> 4;
    v4f acc[TM][TN];
    const v4u *ap[TM], *bp[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = min(mt0 + i, Mt - 1);
        ap[i] = Wp + (size_t)mt * Hc * NS * 64 + lane;
        const v4f bv = *(const v4f *)(bias + mt * 16 + kq * 4) * acc_scale;      // the bias joins the products in their scaled space (exact)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = bv;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int nt = min(nt0 + j, ntile - 1);
        bp[j] = (const v4u *)(in + (size_t)nt * Hc * NS * 1024) + lane;
    }
    constexpr int WS[kSplitNT] = FFHIP_SPLIT_TERMS_W, XS[kSplitNT] = FFHIP_SPLIT_TERMS_X;
    // the operands of chunk c+1 are in flight while the 96 MFMAs of chunk c issue (two register sets, K loop unrolled by two)
    v4u A0[TM][NS], B0[TN][NS], A1[TM][NS], B1[TN][NS];
    auto load = [&](v4u (&A)[TM][NS], v4u (&B)[TN][NS], int c) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int s = 0; s < NS; s++) A[i][s] = ap[i][(size_t)(c * NS + s) * 64];
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int s = 0; s < NS; s++) B[j][s] = bp[j][(size_t)(c * NS + s) * 64];
    };
    auto mma = [&](v4u (&A)[TM][NS], v4u (&B)[TN][NS]) {
#pragma unroll
        for (int term = 0; term < kSplitNT; term++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = mm(A[i][WS[term]], B[j][XS[term]], acc[i][j]);
    };
    const float inv_scale = 1.0f / acc_scale;
    load(A0, B0, 0);
    for (int c = 0; c < Hc; c += 2) {
        if (c + 1 < Hc) load(A1, B1, c + 1);
        mma(A0, B0);
        if (c + 1 < Hc) {
            if (c + 2 < Hc) load(A0, B0, c + 2);
            mma(A1, B1);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = mt0 + i;
        if (mt >= Mt) continue;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nt = nt0 + j;
            if (nt >= ntile) continue;
            *(v4f *)(xa + ((size_t)nt * Mt + mt) * 256 + lane * 4) = acc[i][j] * inv_scale;
        }
    }
}

void launch_inproj_split(hipStream_t s, const void *in_split, float *xa, const void *Wp, const float *bias, int ntile, int H, int scale_exp) {
    const int Mt = H / 4, Hc = H / 32;
    const int nMblk = (Mt + 7) / 8, nNblk = (ntile + 11) / 12;
    hipLaunchKernelGGL(k_inproj_split, dim3(nMblk * nNblk), dim3(256), 0, s, (const unsigned char *)in_split, xa, (const v4u *)Wp, bias, ntile, Mt, Hc, split_pow2(scale_exp));
}

// ---- layout converters -------------------------------------------------------------------------
// fp32 tile-interleaved [tile][Ut][16 reads][4] <-> split [tile][Hc][NS][64][8 bf16]; one thread per (tile, pair of unit tiles, read)
__global__ void __launch_bounds__(256)
k_split_from_f32(const float *__restrict__ in, unsigned char *__restrict__ out, size_t npair, int Ut, float scale, unsigned *__restrict__ sat, int B16) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npair) return;
    const int rl = (int)(idx & 15);
    const size_t pr = idx >> 4;                         // (tile, unit-tile pair)
    const int up = (int)(pr % (Ut / 2));
    const size_t tile = pr / (Ut / 2);
    const float *src = in + (tile * Ut + 2 * up) * 64 + rl * 4;
    const v4f lo = *(const v4f *)src * scale, hi = *(const v4f *)(src + 64) * scale;
    if (sat && (split_overflow(lo) || split_overflow(hi))) sat[(tile % B16) * 16 + rl] = 1u;      // clamped below: the engine re-runs this read on the f32 path
    unsigned char *dst = out + tile * ((size_t)Ut / 8 * NS * 1024);
    const int c = up >> 2, kq = up & 3;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const v2u a = split4<true>(lo, s), b = split4<true>(hi, s);
        *(v4u *)(dst + (size_t)(((c * NS + s) * 64 + kq * 16 + rl) * 16)) = (v4u){ a.x, a.y, b.x, b.y };
    }
}

__global__ void __launch_bounds__(256)
k_f32_from_split(const unsigned char *__restrict__ in, float *__restrict__ out, size_t npair, int Ut, float inv_scale) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npair) return;
    const int rl = (int)(idx & 15);
    const size_t pr = idx >> 4;
    const int up = (int)(pr % (Ut / 2));
    const size_t tile = pr / (Ut / 2);
    const unsigned char *src = in + tile * ((size_t)Ut / 8 * NS * 1024);
    const int c = up >> 2, kq = up & 3;
    float v[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    // smallest slice first (bf16 build: the three slices partition the mantissa, the sum is exact in any order)
#pragma unroll
    for (int s = NS - 1; s >= 0; s--) {
        const v4u w = *(const v4u *)(src + (size_t)(((c * NS + s) * 64 + kq * 16 + rl) * 16));
        const unsigned d[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
        for (int e = 0; e < 4; e++) {
            v[2 * e] += split_slice_value(d[e] & 0xFFFFu);
            v[2 * e + 1] += split_slice_value(d[e] >> 16);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] *= inv_scale;
    float *dst = out + (tile * Ut + 2 * up) * 64 + rl * 4;
    *(v4f *)dst = (v4f){ v[0], v[1], v[2], v[3] };
    *(v4f *)(dst + 64) = (v4f){ v[4], v[5], v[6], v[7] };
}

void launch_split_from_f32(hipStream_t s, const float *in, void *out, size_t ntile, int H, int act_exp, unsigned *sat, int B16) {
    const size_t npair = ntile * (size_t)(H / 8) * 16;
    hipLaunchKernelGGL(k_split_from_f32, dim3((unsigned)((npair + 255) / 256)), dim3(256), 0, s, in, (unsigned char *)out, npair, H / 4, split_pow2(act_exp), sat, B16 > 0 ? B16 : 1);
}
void launch_f32_from_split(hipStream_t s, const void *in, float *out, size_t ntile, int H, int act_exp) {
    const size_t npair = ntile * (size_t)(H / 8) * 16;
    hipLaunchKernelGGL(k_f32_from_split, dim3((unsigned)((npair + 255) / 256)), dim3(256), 0, s, (const unsigned char *)in, out, npair, H / 4, split_pow2(-act_exp));
}

// ---- exhaustive check of the lean gate math (debug entry point) -----------------------------------
// every fp32 mantissa at the given binary exponent: recip_1_to_2p126 against the IEEE division, logistic_ref4_lean against
// logistic_ref4 on the same bit patterns read as gate pre-activations in [-128, 128)
__global__ void __launch_bounds__(256)
k_lean_math_check(int exponent, int steps, unsigned long long *bad) {
    const unsigned mant = blockIdx.x * 256 + threadIdx.x;       // 2^23 threads
    const float d = __uint_as_float(((unsigned)(127 + exponent) << 23) | mant);
    const float want = 1.0f / d;
    const float got = steps == 0 ? recip_1_to_2p126<0>(d) : (steps == 1 ? recip_1_to_2p126<1>(d) : recip_1_to_2p126<2>(d));
    unsigned nbad = (__float_as_uint(want) != __float_as_uint(got)) ? 1u : 0u;
    // gate pre-activations: sign and magnitude patterns derived from the same counter
    const float x = (exponent & 1 ? -1.0f : 1.0f) * __uint_as_float(((unsigned)(127 + (exponent % 7)) << 23) | mant) * 1.0f;
    const ffv4 a = logistic_ref4((ffv4){ x, -x, x * 0.03125f, x * 61.0f });
    const ffv4 b = logistic_ref4_lean((ffv4){ x, -x, x * 0.03125f, x * 61.0f });
    nbad += (__float_as_uint(a.x) != __float_as_uint(b.x)) + (__float_as_uint(a.y) != __float_as_uint(b.y)) +
            (__float_as_uint(a.z) != __float_as_uint(b.z)) + (__float_as_uint(a.w) != __float_as_uint(b.w));
    nbad += (__float_as_uint(tanh_ref(x)) != __float_as_uint(tanh_ref_lean(x)));
    const ffv2 b2 = logistic_ref2_lean((ffv2){ x, x * -0.5f });
    nbad += (__float_as_uint(logistic_ref(x)) != __float_as_uint(b2.x)) + (__float_as_uint(logistic_ref(x * -0.5f)) != __float_as_uint(b2.y));
    if (nbad) atomicAdd(bad, (unsigned long long)nbad);
}
void launch_lean_math_check(hipStream_t s, int exponent, int steps, unsigned long long *bad) {
    hipLaunchKernelGGL(k_lean_math_check, dim3(1u << 15), dim3(256), 0, s, exponent, steps, bad);
}

// ---- host side ---------------------------------------------------------------------------------
// largest H / 128 whose two weight matrices fit one CU's registers: 16N rows x 256N k x 2 B x slices per CU
// (three bf16 slices: 221 KiB at N = 3; two fp16 slices: 147 KiB at N = 3, 256 KiB at N = 4; the file holds 512 KiB)
#ifdef FFHIP_SPLIT_BF16X3
constexpr int kSplitMaxN = 3;
#else
constexpr int kSplitMaxN = 4;
#endif
// tiles per group by [kind][H / 128 - 1] (measured on MI355X)
// 256 reads x 4000 samples, MI355X, Msamples/s TS = 2 -> 1: LSTM H = 256 97.4 -> 106.6, GRUmod H = 256 39.4 -> 44.3; at N = 3 the
// one-tile form needs 161 registers (33 spilled at 128: 77.6 -> 67.5), at N = 4 it is hopeless (102 spilled)
constexpr int kSplitTS[2][4] = { { 1, 1, 1, 2 }, { 1, 1, 2, 2 } };
int split_tiles_per_group(int kind, int H);
bool split_supported(int kind, int H) { return (kind == 0 || kind == 1) && H % 128 == 0 && H >= 128 && H <= 128 * (kind == 0 ? kSplitMaxN : 3); }      // GRUmod at N = 4 spills 169 registers; no GRUmod model is that wide
// read tiles (of 16) one launch takes: 32 workgroups per group; one workgroup per CU and a pair of tiles per group, or two
// workgroups per CU and one tile per group -- 2 * (ncu / 32) tiles either way
// At H <= 256 the pair form also fits two workgroups per CU (<= 128 VGPRs, 53 KiB LDS): a launch then takes 4 * (ncu / 32) tiles --
// 512 reads on 256 CUs, four independent 16-read recurrences per CU.  The step of this kernel is a latency chain (hand-off through
// L2, sweep, gate math), so the reads in flight per launch are what sets its throughput.
// H = 384 (LSTM): the dense pair form k_lstm_split<0, 3, 2, true> -- 128 registers, 77 KiB of LDS, two workgroups per CU: a launch takes
// 512 reads, and the launches of two 256-read batches in flight run BESIDE each other instead of one after the other
// (FFHIP_SPLIT_DENSE=0: the one-tile form)
static bool split_dense3(int kind, int H) {
    const char *e = dbg("split_dense");
    return kind == 0 && H == 384 && kSplitF16 && !(e && e[0] == '0') && !dbg("split_ts");
}
// H = 256 (LSTM and GRUmod): the dense pair form needs 79 registers and 53 KiB there -- THREE workgroups per CU, six 16-read recurrences,
// 768 reads per launch (FFHIP_DENSE256=0: at most the two-workgroup form of round 2, 512 reads)
static bool split_dense256(int H) {
    const char *e = dbg("dense256");
    return H == 256 && kSplitF16 && !(e && e[0] == '0') && !dbg("split_ts") && !dbg("no_dense");
}
// H = 256: the packed forms (lstm_split_body's PACK; k_grumod_pack, k_lstm_pack) -- 16 members a group, 128 registers, two workgroups a CU: a
// FULL launch takes 8 * (ncu / 32) tiles, 1024 reads on 256 CUs; their weights are the second half of the layer's pack (FFHIP_NO_PACK: never)
static bool split_pack256(int kind, int H) { return (kind == 0 || kind == 1) && split_dense256(H) && !dbg("no_pack"); }
static bool split_launch_pack(int kind, int H, int nrt, int ncu) { return split_pack256(kind, H) && nrt == 8 * (ncu / 32); }
int split_max_tiles(int ncu, int H) {
    if (split_dense256(H)) return 6 * (ncu / 32);
    return ((H <= 256 || split_dense3(0, H)) && !dbg("no_dense") ? 4 : 2) * (ncu / 32);
}
// tiles the next launch of a batch takes when `remaining` are left: the dense forms are for FULL launches only (a partly filled
// one has a group count that is no multiple of the 8 XCDs and loses the one-L2 hand-off)
int split_next_launch_tiles(int kind, int H, int remaining, int ncu) {
    const int unit = ncu / 32 > 0 ? ncu / 32 : 1;       // (never 0: the engine's layer loop advances by this, the binary sizes its batches by it)
    if (split_pack256(kind, H) && remaining >= 8 * unit) return 8 * unit;
    if (split_dense256(H) && remaining >= 6 * unit) return 6 * unit;
    if ((H <= 256 || split_dense3(0, H)) && !dbg("no_dense") && remaining >= 4 * unit) return 4 * unit;
    return remaining < 2 * unit ? remaining : 2 * unit;
}
// tiles per group of a launch of nrt read tiles
// `beside`: another batch is between run and finish -- its layer launches are on the chip; the dense form runs BESIDE them
static bool split_launch_dense256(int H, int nrt, int ncu) { return split_dense256(H) && nrt > 4 * (ncu / 32); }
static bool split_launch_dense3(int kind, int H, int nrt, int ncu, int beside) {
    return split_dense3(kind, H) && (nrt > 2 * (ncu / 32) || beside || dbg("split_dense_always"));      // (the variable: development)
}
static int split_launch_ts(int kind, int H, int nrt, int ncu, int beside) {
    // the dense forms (two workgroups per CU, a pair of tiles each) take launches with more tiles than the one-tile form can:
    // FULL launches of 4 * (ncu / 32) tiles in practice (the engine's layer loop cuts a batch that way)
    if (H <= 256 && nrt > 2 * (ncu / 32)) return 2;
    if (split_launch_dense3(kind, H, nrt, ncu, beside)) return 2;
    return split_tiles_per_group(kind, H);
}
// workgroups of such a launch, and how many workgroups of its kernel share a CU: two launches (of two batches in flight) are
// co-resident -- every workgroup of both must be, they wait for their peers -- iff together they fit
int split_launch_workgroups(int kind, int H, int nrt, int ncu, int beside) {
    if (split_launch_pack(kind, H, nrt, ncu)) return nrt / 2 * 16;
    const int ts = split_launch_ts(kind, H, nrt, ncu, beside);
    return (nrt + ts - 1) / ts * 32;
}
int split_workgroups_per_cu(int kind, int H, int nrt, int ncu, int beside) {
    const int ts = split_launch_ts(kind, H, nrt, ncu, beside);
    if (split_launch_pack(kind, H, nrt, ncu)) return 2;
    if (split_launch_dense256(H, nrt, ncu)) return 3;
    return (ts == 1 || H <= 256 || split_launch_dense3(kind, H, nrt, ncu, beside)) ? 2 : 1;
}
size_t split_flag_words(int nrt) { return (size_t)nrt * 32; }
// 16-byte pieces of a layer's classic pack [2 matrices][H / 4 unit tiles][H / 32 chunks][slices][64 lanes]: the gate-major pack of the packed
// GRUmod form ([2][3 H / 16 row tiles][H / 32][slices][64]) follows it
size_t split_pack_offset(int H) { return (size_t)2 * (H / 4) * (H / 32) * NS * 64; }
int split_tiles_per_group(int kind, int H) {
    const char *force = dbg("split_ts");      // development: 1 or 2
    if (force && (force[0] == '1' || force[0] == '2') && H < 512) return force[0] - '0';
    return kSplitTS[kind & 1][H / 128 - 1];
}

unsigned long long *g_split_dbg = nullptr;
#ifdef FFHIP_PHASES
}  // namespace ffhip
// (instrumented variant only: tools/dev/phases.py)  out[0..63] = cycles by [wave][phase] summed over every workgroup and step since the last reset,
// out[64..71] = steps by wave (summed over workgroups)
extern "C" int ffhip_debug_phases(unsigned long long *out, int reset) {
    static unsigned long long *d = nullptr;
    if (!d) { if (hipMalloc(&d, 72 * 8) != hipSuccess) return -1; hipMemset(d, 0, 72 * 8); ffhip::g_split_dbg = d; }
    hipDeviceSynchronize();
    if (out) hipMemcpy(out, d, 72 * 8, hipMemcpyDeviceToHost);
    if (reset) hipMemset(d, 0, 72 * 8);
    return 0;
}
namespace ffhip {
#endif

// one launch for the layers of two batches (the dense form at H = 384 only); false: shapes this does not take -- launch them one by one
bool launch_lstm_split_pair(hipStream_t s, int kind, int H, int ncu, const SplitLaunch &p0, const SplitLaunch &p1) {
#ifdef FFHIP_SPLIT_BF16X3
    return false;
#else
    if (!split_dense3(kind, H) || dbg("no_pair") || p0.nrt > 2 * (ncu / 32) || p1.nrt > 2 * (ncu / 32) || p0.nrt < 1 || p1.nrt < 1) return false;
    auto mk = [&](const SplitLaunch &p) {
        SplitArgs a;
        a.epoch = p.epoch; a.acc_scale = split_pow2(p.scale_exp); a.scale_exp = p.scale_exp; a.fast_gates = p.fast_gates;
        a.split_gate = dbg("no_split_gate") ? 0 : 1;
        a.Wp = (const v4u *)p.Wp; a.bias = p.bias; a.xin = (const unsigned char *)p.xin; a.hout = (unsigned char *)p.hout; a.hout_f32 = p.hout_f32;
        a.flags = p.flags; a.abort_word = p.abort_word;
        a.Tb = p.Tb; a.B16 = p.B16; a.H = H; a.rt0 = p.rt0; a.nrt = p.nrt; a.backward = p.backward; a.mode = p.mode;
        a.tbs = p.tbs; a.tbt = p.tbt; a.live = p.live; a.dbg = g_split_dbg;
        return a;
    };
    const int g0 = (p0.nrt + 1) / 2, g1 = (p1.nrt + 1) / 2;
    if ((g0 & 7) != 0) return false;              // (the second batch's block indices must start at a multiple of the 8 XCDs)
    // the two batches share everything but their buffers: same model (weights, exponents), same capacity and tile count
    if (p0.Wp != p1.Wp || p0.bias != p1.bias || p0.Tb != p1.Tb || p0.B16 != p1.B16 || p0.rt0 != p1.rt0 || p0.nrt != p1.nrt || p0.backward != p1.backward ||
        p0.mode != p1.mode || p0.scale_exp != p1.scale_exp || p0.fast_gates != p1.fast_gates || (p0.hout_f32 == nullptr) != (p1.hout_f32 == nullptr) ||
        (p0.tbs == nullptr) != (p1.tbs == nullptr) || (p0.live == nullptr) != (p1.live == nullptr)) return false;
    const SplitArgs a1 = mk(p1);
    SplitArgsOther o;
    o.xin = a1.xin; o.hout = a1.hout; o.hout_f32 = a1.hout_f32; o.flags = a1.flags; o.abort_word = a1.abort_word; o.tbs = a1.tbs; o.tbt = a1.tbt; o.live = a1.live; o.epoch = a1.epoch;
    o.nwg0 = g0 * 32;
    const dim3 grid((g0 + g1) * 32);
    if (p0.live) { if (p0.fast_gates) hipLaunchKernelGGL((k_lstm_split_pair<0, 3, 2, true, true, 2>), grid, dim3(512), 0, s, mk(p0), o); else hipLaunchKernelGGL((k_lstm_split_pair<0, 3, 2, true, true, 0>), grid, dim3(512), 0, s, mk(p0), o); }
    else { if (p0.fast_gates) hipLaunchKernelGGL((k_lstm_split_pair<0, 3, 2, true, false, 2>), grid, dim3(512), 0, s, mk(p0), o); else hipLaunchKernelGGL((k_lstm_split_pair<0, 3, 2, true, false, 0>), grid, dim3(512), 0, s, mk(p0), o); }
    return true;
#endif
}

bool launch_lstm_split(hipStream_t s, int kind, const void *Wp, const float *bias, const void *xin, void *hout, float *hout_f32,
                       unsigned *flags, unsigned *abort_word, int Tb, int B16, int H, int rt0, int nrt, int backward, int mode,
                       int scale_exp, int fast_gates, const int *tbs, const int *tbt, int ncu, unsigned epoch, int beside, const unsigned *live) {
    SplitArgs a;
    a.epoch = epoch;
    a.acc_scale = split_pow2(scale_exp);
    a.scale_exp = scale_exp;
    a.fast_gates = fast_gates;
    a.split_gate = dbg("no_split_gate") ? 0 : 1;
    a.Wp = (const v4u *)Wp; a.bias = bias; a.xin = (const unsigned char *)xin; a.hout = (unsigned char *)hout; a.hout_f32 = hout_f32;
    a.flags = flags; a.abort_word = abort_word;
    a.Tb = Tb; a.B16 = B16; a.H = H; a.rt0 = rt0; a.nrt = nrt; a.backward = backward; a.mode = mode;
    a.tbs = tbs; a.tbt = tbt; a.live = live; a.dbg = g_split_dbg;
    // tiles per group: 1 (two workgroups per CU, one read tile each) where that is faster, else 2 (kSplitTS)
    // ... and 2 with two workgroups per CU when the launch carries more tiles than the one-tile form can take (H <= 256)
    const int ts = split_launch_ts(kind, H, nrt, ncu, beside);
    const int ngroup_l = (nrt + ts - 1) / ts;
#ifndef FFHIP_SPLIT_BF16X3
    // (a packed batch -- a.live -- takes the LIVE instantiation of the same form; the gate level picks the GL one)
#define SPLIT_GO(KF, GRID) do { if (a.live) { if (a.fast_gates) hipLaunchKernelGGL((KF(true, 2)), dim3(GRID), dim3(512), 0, s, a); else hipLaunchKernelGGL((KF(true, 0)), dim3(GRID), dim3(512), 0, s, a); } \
                                else { if (a.fast_gates) hipLaunchKernelGGL((KF(false, 2)), dim3(GRID), dim3(512), 0, s, a); else hipLaunchKernelGGL((KF(false, 0)), dim3(GRID), dim3(512), 0, s, a); } } while (0)
#define K_DENSE3(L, G) k_lstm_split<0, 3, 2, true, L, G>
#define K_GPACK(L, G) k_grumod_pack<L, G>
#define K_LPACK(L, G) k_lstm_pack<L, G>
#define K_D256L(L, G) k_lstm_split<0, 2, 2, true, L, G>
#define K_D256G(L, G) k_lstm_split<1, 2, 2, true, L, G>
    if (split_launch_dense3(kind, H, nrt, ncu, beside)) { SPLIT_GO(K_DENSE3, ngroup_l * 32); return true; }
    if (split_launch_pack(kind, H, nrt, ncu)) {
        a.Wp += split_pack_offset(H);
        if (kind == 1) SPLIT_GO(K_GPACK, nrt / 2 * 16);
        else SPLIT_GO(K_LPACK, nrt / 2 * 16);
        return true;
    }
    if (split_launch_dense256(H, nrt, ncu)) {
        if (kind == 0) SPLIT_GO(K_D256L, ngroup_l * 32);
        else SPLIT_GO(K_D256G, ngroup_l * 32);
        return true;
    }
#endif
#define K_GEN(KK, NN, TT) [&]() { if (a.live) { if (a.fast_gates) hipLaunchKernelGGL((k_lstm_split<KK, NN, TT, false, true, 2>), dim3(ngroup_l * 32), dim3(512), 0, s, a); else hipLaunchKernelGGL((k_lstm_split<KK, NN, TT, false, true, 0>), dim3(ngroup_l * 32), dim3(512), 0, s, a); } \
                                  else { if (a.fast_gates) hipLaunchKernelGGL((k_lstm_split<KK, NN, TT, false, false, 2>), dim3(ngroup_l * 32), dim3(512), 0, s, a); else hipLaunchKernelGGL((k_lstm_split<KK, NN, TT, false, false, 0>), dim3(ngroup_l * 32), dim3(512), 0, s, a); } }()
#define SPLIT_LAUNCH(K, NN) do { if (ts == 1) K_GEN(K, NN, 1); else K_GEN(K, NN, 2); return true; } while (0)
#ifdef FFHIP_SPLIT_BF16X3
    if (kind == 0) switch (H / 128) { case 1: SPLIT_LAUNCH(0, 1); case 2: SPLIT_LAUNCH(0, 2); case 3: SPLIT_LAUNCH(0, 3); }
    if (kind == 1) switch (H / 128) { case 1: SPLIT_LAUNCH(1, 1); case 2: SPLIT_LAUNCH(1, 2); case 3: SPLIT_LAUNCH(1, 3); }
#else
    if (kind == 0) switch (H / 128) { case 1: SPLIT_LAUNCH(0, 1); case 2: SPLIT_LAUNCH(0, 2); case 3: SPLIT_LAUNCH(0, 3);
                                      case 4: { const dim3 g4((nrt + 1) / 2 * 32);      // (H = 512: the pair form only -- one tile per group would need 122 registers more than a second workgroup leaves)
                                              if (a.live) { if (a.fast_gates) hipLaunchKernelGGL((k_lstm_split<0, 4, 2, false, true, 2>), g4, dim3(512), 0, s, a); else hipLaunchKernelGGL((k_lstm_split<0, 4, 2, false, true, 0>), g4, dim3(512), 0, s, a); }
                                              else { if (a.fast_gates) hipLaunchKernelGGL((k_lstm_split<0, 4, 2, false, false, 2>), g4, dim3(512), 0, s, a); else hipLaunchKernelGGL((k_lstm_split<0, 4, 2, false, false, 0>), g4, dim3(512), 0, s, a); }
                                              return true; } }
    if (kind == 1) switch (H / 128) { case 1: SPLIT_LAUNCH(1, 1); case 2: SPLIT_LAUNCH(1, 2); case 3: SPLIT_LAUNCH(1, 3); }
#endif
#undef SPLIT_LAUNCH
    return false;
}

}  // namespace ffhip
