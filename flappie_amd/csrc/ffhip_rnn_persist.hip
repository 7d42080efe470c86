// ffhip_rnn_persist.hip -- persistent recurrent layer for gfx950: one launch per layer.
//
// Replaces lstm_forward/lstm_backward + lstm_step (layers.c:877-1026) and
// grumod_forward/grumod_backward + grumod_step (layers.c:571-715) for a whole batch.
//
// Why persistent.  A layer is Tb strictly dependent steps; per step the work is the small GEMM
// [16*B16 reads x H] . [H x 4H].  Launch-per-step re-streams the whole recurrent matrix from L2
// every step (8.5 us per step measured, 22 % of the f32 MFMA peak).  Here the recurrent matrix is
// loaded ONCE per layer into VGPRs and stays there for all Tb steps:
//
//   * one GROUP of G workgroups per read tile (16 reads); member m owns UPC unit tiles
//     (4 hidden units x 4 gate rows each), i.e. 16*UPC rows of sW^T;
//   * inside a workgroup the 4 waves split K: wave w holds the [16*UPC x 16*KPW] slice of its rows
//     as UPC*KPW float4 MFMA A-fragments in registers (72 VGPRs at H = 384);
//   * per step a wave waits for the producers of ITS K slice only, loads that slice of h(t-1)
//     straight from L2 in B-fragment order (KPW coalesced 1 KiB loads), issues UPC*KPW*4 MFMAs,
//     and drops its partial tile sums in LDS; after one barrier wave j (j < UPC) adds the four
//     partials (and Xa(t) in k_rnn_persist; in k_lstm_fused the projection Wi x(t) + b was accumulated
//     by the same MFMA stream from a second set of resident weights), does the gate math lane-locally
//     (cell state lives in a register for the whole layer) and stores its 4 units x 16 reads of h(t).
//     The stored values themselves are the hand-off signal (see below): no counters, no flags.
//
// Kernels here: k_rnn_persist (recurrence only, behind a separate projection GEMM) and k_lstm_fused
// (projection + recurrence, the default wherever its 2 x UPC x KPW weight fragments fit two workgroups
// per CU).  Ragged batches: a read tile runs for the block count of its longest read and lanes force
// h = c = 0 beyond their own read's end (PersistArgs::tbs / tbt).
//
// Two workgroups are resident per CU (<= 256 VGPRs, 24 KiB LDS each): while one waits for its
// group's hand-off the other one's MFMAs own the matrix pipes, so the hand-off latency is hidden
// by occupancy rather than by hand-written ping-pong code.
//
// Inter-workgroup protocol (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement &
// inter-workgroup visibility", cdna_hip_programming.md G16 form R2 "the data is the flag").
// The payload is the layer's output h(t) itself.  The host pre-fills the output buffer with a NaN
// sentinel; a producer lane publishes one aligned dword per (unit, read); a consumer wave re-reads
// the 16-byte fragments of its K slice with sc1 loads (served by L2, never by the stale per-CU L1)
// until no dword equals the sentinel.  |h| < 1 always, so a valid h can never be mistaken for the
// sentinel; a NaN produced by NaN inputs shows up as FFHIP_ETIMEOUT instead of garbage.
//   - placement-independent form: stores are write-through (sc1) -> visible to every XCD;
//   - when the G members of a group verify at run time (hardware XCC id, exchanged with the
//     write-through form) that they all sit on ONE XCD they share one L2, and plain stores
//     (which stay in that L2) are sufficient and ~2x faster per hop.
// Nothing depends on dispatch order or on an assumed block -> XCD map.  Every spin is bounded: on
// timeout the workgroup raises the abort word, leaves through its barriers and the host reports
// FFHIP_ETIMEOUT instead of hanging the GPU.
#include "ffhip_internal.hpp"
#include "ffhip_math.hpp"
#include <stdlib.h>

namespace ffhip {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));


struct PersistArgs {
    const v4f *sWp;        // [Ut][K16][64] float4, A-fragment order
    const v4f *iWp;        // fused layers: input weights, same packing
    const float *xin;      // fused layers: layer input, tile-interleaved [Tb][B16][Ut*64]
    const float *bias;     // fused layers: permuted bias [16*Ut]
    const v4f *xa;         // [Tb][B16][Ut][64] float4, D-fragment order
    float *hout;           // [Tb][B16][Ut*64] tile-interleaved
    unsigned *flags;       // [nrt][G] XCC ids (zeroed before launch)
    unsigned *abort_word;  // != 0 -> a wait timed out
    int Tb, B16, Ut, K16, G, rt0, nrt, backward;
    int fast_gates;        // opt-in (FFHIP_FAST_GATES=1): hardware exp2/rcp gate math, NOT bit-compatible with the reference's exp_ps
    const int *tbs;        // ragged batch: blocks of each read [16*B16] (nullptr = all Tb); a read's steps t >= tbs[r] give h = c = 0,
                           // which is a fresh start for backward layers and inert padding for forward ones
    const int *tbt;        // ragged batch: max blocks per read tile [B16]
    int mode;              // 0 = verify placement, use the L2-local hand-off when a group shares an XCD; 1 = always write-through
};

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned kSentinel = 0xFFFFFFFFu;   // a NaN; never a value of h

__device__ __forceinline__ v4f mfma4p(v4f a, v4f b, v4f c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// KIND 0 = LSTM (gate rows i,f,g,o), 1 = GRUmod (gate rows z,r,candidate,-)
template <int KIND, int UPC, int KPW>
__global__ void __launch_bounds__(256, 2)
k_rnn_persist(PersistArgs a) {
    __shared__ v4f part[2][4][UPC][64];
    __shared__ int lds_abort;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = a.G, Ut = a.Ut, K16 = a.K16;
    // block -> (group g, member m).  Hardware places block b on XCD b % 8: when the group count
    // is a multiple of 8 give every XCD whole groups (speed only, never correctness).
    int g, m;
    {
        const int b = blockIdx.x;
        if ((a.nrt & 7) == 0) { const int xcd = b & 7, j = b >> 3; g = xcd + 8 * (j / G); m = j % G; }
        else { g = b / G; m = b % G; }
    }
    const int rt = a.rt0 + g;
    const int Tb = a.tbt ? a.tbt[rt] : a.Tb;                     // steps of this read tile
    if (Tb <= 0) return;                                         // a tile of empty slots (uniform for the whole group)
    const int my_tb = a.tbs ? a.tbs[rt * 16 + (threadIdx.x & 15)] : a.Tb;      // blocks of the read this lane does gate math for
    const int ut0 = m * UPC;
    if (threadIdx.x == 0) lds_abort = 0;

    // ---- placement check.  The write-through (sc1) hand-off is correct for any placement but every
    // hop goes through the fabric.  CUs of ONE XCD share one L2, so when all G members of this group
    // report the same XCC id (read from hardware, exchanged with the write-through protocol) the
    // group may hand off through that L2: stores that stay in L2 (plain / sc0) + L1-bypassing loads.
    __shared__ int lds_fast;
    if (threadIdx.x < 64) {
        int fast_l = 0;
        if (a.mode == 0) {
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
            fast_l = __all(v == xcc) ? 1 : 0;      // a timeout leaves zeros -> not equal -> safe path
        }
        if (lane == 0) lds_fast = fast_l;
    }

    // ---- resident weights: rows of my UPC unit tiles, K slice of my wave
    v4f wreg[UPC][KPW];
#pragma unroll
    for (int j = 0; j < UPC; j++)
#pragma unroll
        for (int kk = 0; kk < KPW; kk++) {
            const int k16 = wave * KPW + kk;
            const int ut = ut0 + j;
            wreg[j][kk] = (k16 < K16 && ut < Ut) ? a.sWp[((size_t)ut * K16 + k16) * 64 + lane] : (v4f){ 0.f, 0.f, 0.f, 0.f };
        }
    const int my_tile = (wave < UPC && ut0 + wave < Ut) ? wave : -1;   // gate role
    float c = 0.0f, hprev_own = 0.0f;
    const int q = lane >> 4, rl = lane & 15;
    const size_t tile_floats = (size_t)Ut * 64;
    __syncthreads();
    const bool fast = lds_fast != 0;
    // This kernel is latency-bound (a dependent chain of Tb hand-offs) and leaves the matrix pipes
    // idle about half of the time; throughput kernels of another stream (input projections of the
    // next batch) are meant to fill those slots.  Raise the wave priority so that they never
    // lengthen the chain.
    __builtin_amdgcn_s_setprio(3);
    v4f x_next = { 0.f, 0.f, 0.f, 0.f };
    if (my_tile >= 0) {
        const int t0 = a.backward ? Tb - 1 : 0;
        x_next = a.xa[(((size_t)t0 * a.B16 + rt) * Ut + ut0 + my_tile) * 64 + lane];
    }

    for (int i = 0; i < Tb; i++) {
        const int t = a.backward ? Tb - 1 - i : i;
        const int tp = a.backward ? t + 1 : t - 1;
        const v4f x = x_next;
        if (my_tile >= 0 && i + 1 < Tb) {      // Xa streams from HBM: fetch one step ahead
            const int tn = a.backward ? t - 1 : t + 1;
            x_next = a.xa[(((size_t)tn * a.B16 + rt) * Ut + ut0 + my_tile) * 64 + lane];
        }
        if (i > 0) {
            v4f acc[UPC];
#pragma unroll
            for (int j = 0; j < UPC; j++) acc[j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
            if (wave * KPW < K16) {
                // B fragments of my K slice of h(t-1), straight from L2 (sc1 loads never hit this CU's
                // L1).  The payload is its own flag: the layer output was pre-filled with a NaN
                // sentinel, every h is a single aligned dword store and |h| < 1, so "no sentinel in my
                // slice" == "every producer of my slice has published step i-1".  No separate counter,
                // no extra round trip.
                const float *hp = a.hout + ((size_t)tp * a.B16 + rt) * tile_floats;
                __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)hp, 0, (int)(tile_floats * 4), 0x00020000);
                v4u raw[KPW];
                bool timed_out = false;
                for (unsigned spin = 0;; spin++) {
                    bool ok = true;
#pragma unroll
                    for (int kk = 0; kk < KPW; kk++) {
                        const int k16 = wave * KPW + kk;
                        raw[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (k16 * 256 + lane * 4) * 4, 0, 16 /*sc1*/);
                    }
#pragma unroll
                    for (int kk = 0; kk < KPW; kk++)
                        ok = ok && raw[kk].x != kSentinel && raw[kk].y != kSentinel && raw[kk].z != kSentinel && raw[kk].w != kSentinel;
                    if (__all(ok)) break;
                    if (spin > 3000000u || (spin & 255u) == 255u) {
                        const unsigned ab = __hip_atomic_load(a.abort_word, RLX_AGENT);
                        if (ab != 0u || spin > 3000000u) { timed_out = true; break; }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (timed_out) {
                    if (lane == 0) { __hip_atomic_store(a.abort_word, 1u, RLX_AGENT); lds_abort = 1; }
                } else {
#pragma unroll
                    for (int kk = 0; kk < KPW; kk++) {
                        const v4f bf = __builtin_bit_cast(v4f, raw[kk]);      // k16 beyond K16 reads 0 (buffer bounds)
#pragma unroll
                        for (int j = 0; j < UPC; j++) acc[j] = mfma4p(wreg[j][kk], bf, acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < UPC; j++) part[i & 1][wave][j][lane] = acc[j];
            // LDS-only barrier: __syncthreads() would also drain vmcnt and put the HBM latency of the
            // Xa prefetch on the critical path of every step.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (lds_abort) return;
        }
        if (my_tile >= 0) {
            v4f s = { 0.f, 0.f, 0.f, 0.f };
            if (i > 0) {
#pragma unroll
                for (int w2 = 0; w2 < 4; w2++) s = s + part[i & 1][w2][my_tile][lane];
            }
            float h;
            if (KIND == 0) {
                s = s + x;
                // layers.c:1014-1025.  sigma(i), sigma(f), sigma(o) and tanh(g) = 2 sigma(2g) - 1 are
                // four independent logistic evaluations: do them as one interleaved vector call.
                const ffv4 L = logistic_ref4_lean((ffv4){ s.x, s.y, s.z + s.z, s.w });      // bit-identical to logistic_ref4, fewer instructions (ffhip_math.hpp)
                const float tanh_g = (L.z + L.z) - 1.0f;
                const float forget = L.y * c;
                const float update = L.x * tanh_g;
                c = forget + update;
                h = L.w * tanh_ref_lean(c);
            } else {
                // layers.c:690-714: x added to z,r before the logistic; candidate = tanh(r*u + x_c)
                const float z = logistic_ref(s.x + x.x);
                const float r = logistic_ref(s.y + x.y);
                float hbar = r * s.z + x.z;
                hbar = tanh_ref(hbar);
                h = z * hprev_own + (1.0f - z) * hbar;
                hprev_own = h;
            }
            // gather the 4 units of a read into one lane: lanes 0..15 then hold 16 contiguous bytes
            // each, and the wave writes its 256 B (two whole 128-B lines) with ONE store instruction,
            // so a consumer never finds a half-written line that L2 would have to complete from HBM.
            if (t >= my_tb) { h = 0.0f; c = 0.0f; hprev_own = 0.0f; }      // beyond this read's end (ragged batch)
            v4f hv;
            hv.x = __shfl(h, rl);
            hv.y = __shfl(h, rl + 16);
            hv.z = __shfl(h, rl + 32);
            hv.w = __shfl(h, rl + 48);
            float *ho = a.hout + ((size_t)t * a.B16 + rt) * tile_floats + (size_t)(ut0 + my_tile) * 64 + rl * 4;
            if (lane < 16) {
                if (fast) *(v4f *)ho = hv;                            // stays in the group's shared L2
                else {
                    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.hout + ((size_t)t * a.B16 + rt) * tile_floats), 0, (int)(tile_floats * 4), 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, hv), wr, ((ut0 + my_tile) * 64 + rl * 4) * 4, 0, 16 /*sc1: write-through*/);
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// Fused LSTM layer: input projection + recurrence in ONE persistent kernel.
//
// The recurrent chain above leaves the matrix pipes idle more than half of the time (the hand-off
// between steps is pure latency).  The input projection Wi x(t) + b has the same FLOP count as the
// recurrence and does not depend on it, so each wave also keeps ITS slice of Wi in registers and
// computes the Wi x(t) half of the gate pre-activations for step i while the producers of h(i-1)
// are still finishing: the projection disappears from the critical path, the separate GEMM kernel
// and the 1.26 GB Xa round trip through HBM disappear altogether.
// Per wave and step: KPW coalesced 1 KiB loads of x(t) (issued one step ahead), UPC*KPW*4 MFMAs on
// x, then the h sweep and another UPC*KPW*4 MFMAs on h; the rest is identical to k_rnn_persist.
template <int KIND, int UPC, int KPW>
__global__ void __launch_bounds__(256, 2)
k_lstm_fused(PersistArgs a) {
    __shared__ v4f part[2][4][UPC][64];
    __shared__ int lds_abort;
    __shared__ int lds_fast;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = a.G, Ut = a.Ut, K16 = a.K16;
    int g, m;
    {
        const int b = blockIdx.x;
        if ((a.nrt & 7) == 0) { const int xcd = b & 7, j = b >> 3; g = xcd + 8 * (j / G); m = j % G; }
        else { g = b / G; m = b % G; }
    }
    const int rt = a.rt0 + g;
    const int Tb = a.tbt ? a.tbt[rt] : a.Tb;                     // steps of this read tile
    if (Tb <= 0) return;                                         // a tile of empty slots (uniform for the whole group)
    const int my_tb = a.tbs ? a.tbs[rt * 16 + (threadIdx.x & 15)] : a.Tb;      // blocks of the read this lane does gate math for
    const int ut0 = m * UPC;
    if (threadIdx.x == 0) lds_abort = 0;
    if (threadIdx.x < 64) {
        int fast_l = 0;
        if (a.mode == 0) {
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
            fast_l = __all(v == xcc) ? 1 : 0;
        }
        if (lane == 0) lds_fast = fast_l;
    }
    // resident weights: recurrent and input slices of my rows / my K range
    v4f wreg[UPC][KPW], wi[UPC][KPW];
#pragma unroll
    for (int j = 0; j < UPC; j++)
#pragma unroll
        for (int kk = 0; kk < KPW; kk++) {
            const int k16 = wave * KPW + kk;
            const int ut = ut0 + j;
            const bool ok = (k16 < K16 && ut < Ut);
            wreg[j][kk] = ok ? a.sWp[((size_t)ut * K16 + k16) * 64 + lane] : (v4f){ 0.f, 0.f, 0.f, 0.f };
            wi[j][kk] = ok ? a.iWp[((size_t)ut * K16 + k16) * 64 + lane] : (v4f){ 0.f, 0.f, 0.f, 0.f };
        }
    const int my_tile = (wave < UPC && ut0 + wave < Ut) ? wave : -1;
    const int q = lane >> 4, rl = lane & 15;
    const size_t tile_floats = (size_t)Ut * 64;
    v4f bias = { 0.f, 0.f, 0.f, 0.f };
    if (my_tile >= 0) bias = *(const v4f *)(a.bias + (size_t)(ut0 + my_tile) * 16 + q * 4);
    float c = 0.0f, hprev_own = 0.0f;
    __syncthreads();
    const bool fast = lds_fast != 0;
    __builtin_amdgcn_s_setprio(3);
    const bool have_k = wave * KPW < K16;

    // x(t) slices are plain data from the previous kernel: ordinary coalesced loads, one step ahead
    v4f xf[KPW];
    {
        const int t0 = a.backward ? Tb - 1 : 0;
        const v4f *xp = (const v4f *)(a.xin + ((size_t)t0 * a.B16 + rt) * tile_floats);
#pragma unroll
        for (int kk = 0; kk < KPW; kk++) {
            const int k16 = wave * KPW + kk;
            xf[kk] = (k16 < K16) ? xp[(size_t)k16 * 64 + lane] : (v4f){ 0.f, 0.f, 0.f, 0.f };
        }
    }

    for (int i = 0; i < Tb; i++) {
        const int t = a.backward ? Tb - 1 - i : i;
        const int tp = a.backward ? t + 1 : t - 1;
        v4f acc[UPC];
        v4f accx[KIND == 1 ? UPC : 1];       // GRUmod keeps the projection apart: its candidate row must not mix with sW h
#pragma unroll
        for (int j = 0; j < UPC; j++) acc[j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
        if (KIND == 1) {
#pragma unroll
            for (int j = 0; j < UPC; j++) accx[j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
        }
        // ---- speculative first sweep of h(t-1): issued BEFORE the projection MFMAs so that its L2 round
        // trip (~1200 cycles even when the data is already there) is covered by them
        v4u raw[KPW];
        __amdgpu_buffer_rsrc_t rsrc;
        const bool do_h = (i > 0 && have_k);
        if (do_h) {
            const float *hp = a.hout + ((size_t)tp * a.B16 + rt) * tile_floats;
            rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)hp, 0, (int)(tile_floats * 4), 0x00020000);
        }
        // ---- projection half: independent of the recurrence, runs under the hand-off latency
        if (have_k) {
#pragma unroll
            for (int kk = 0; kk < KPW; kk++) {
                if (kk == (KPW * 2) / 3 && do_h) {
                    // the sweep's round trip overlaps the last third of the projection MFMAs; issued
                    // earlier it would mostly find the sentinel and cost a second round trip
#pragma unroll
                    for (int k2 = 0; k2 < KPW; k2++) {
                        const int k16 = wave * KPW + k2;
                        raw[k2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (k16 * 256 + lane * 4) * 4, 0, 16 /*sc1*/);
                    }
                }
#pragma unroll
                for (int j = 0; j < UPC; j++) {
                    if (KIND == 1) accx[j] = mfma4p(wi[j][kk], xf[kk], accx[j]);
                    else acc[j] = mfma4p(wi[j][kk], xf[kk], acc[j]);
                }
            }
            if (i + 1 < Tb) {
                const int tn = a.backward ? t - 1 : t + 1;
                const v4f *xp = (const v4f *)(a.xin + ((size_t)tn * a.B16 + rt) * tile_floats);
#pragma unroll
                for (int kk = 0; kk < KPW; kk++) {
                    const int k16 = wave * KPW + kk;
                    if (k16 < K16) xf[kk] = xp[(size_t)k16 * 64 + lane];
                }
            }
        }
        // ---- recurrent half
        if (do_h) {
            bool timed_out = false;
            for (unsigned spin = 0;; spin++) {
                bool ok = true;
#pragma unroll
                for (int kk = 0; kk < KPW; kk++)
                    ok = ok && raw[kk].x != kSentinel && raw[kk].y != kSentinel && raw[kk].z != kSentinel && raw[kk].w != kSentinel;
                if (__all(ok)) break;
                if (spin > 3000000u || (spin & 255u) == 255u) {
                    const unsigned ab = __hip_atomic_load(a.abort_word, RLX_AGENT);
                    if (ab != 0u || spin > 3000000u) { timed_out = true; break; }
                }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int kk = 0; kk < KPW; kk++) {
                    const int k16 = wave * KPW + kk;
                    raw[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (k16 * 256 + lane * 4) * 4, 0, 16 /*sc1*/);
                }
            }
            if (timed_out) {
                if (lane == 0) { __hip_atomic_store(a.abort_word, 1u, RLX_AGENT); lds_abort = 1; }
            } else {
#pragma unroll
                for (int kk = 0; kk < KPW; kk++) {
                    const v4f bf = __builtin_bit_cast(v4f, raw[kk]);
#pragma unroll
                    for (int j = 0; j < UPC; j++) acc[j] = mfma4p(wreg[j][kk], bf, acc[j]);
                }
            }
        }
        if (KIND == 1) {
            // rows (z, r, candidate, -): pack {z: x+h, r: x+h, u = (sW h)_c, x_c = (Wi x)_c}; row 3 is free
#pragma unroll
            for (int j = 0; j < UPC; j++) acc[j] = (v4f){ acc[j].x + accx[j].x, acc[j].y + accx[j].y, acc[j].z, accx[j].z };
        }
#pragma unroll
        for (int j = 0; j < UPC; j++) part[i & 1][wave][j][lane] = acc[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (lds_abort) return;
        if (my_tile >= 0) {
            v4f s = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) s = s + part[i & 1][w2][my_tile][lane];
            float h;
            if (KIND == 1) {
                // layers.c:690-714 with the bias rows (z, r, candidate)
                const float z = logistic_ref(s.x + bias.x);
                const float r = logistic_ref(s.y + bias.y);
                float hbar = r * s.z + (s.w + bias.z);
                hbar = tanh_ref(hbar);
                h = z * hprev_own + (1.0f - z) * hbar;
                hprev_own = h;
            } else if (a.fast_gates) {
                s = s + bias;
                // hardware exp2/rcp (1 ulp each): ~6x fewer VALU instructions than the cephes-exact path
                auto sg = [](float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f)); };
                const float si = sg(s.x), sf = sg(s.y), so = sg(s.w);
                const float tg = 2.0f * sg(s.z + s.z) - 1.0f;
                c = sf * c + si * tg;
                h = so * (2.0f * sg(c + c) - 1.0f);
            } else {
                s = s + bias;
                const ffv4 L = logistic_ref4_lean((ffv4){ s.x, s.y, s.z + s.z, s.w });      // bit-identical to logistic_ref4, fewer instructions (ffhip_math.hpp)
                const float tanh_g = (L.z + L.z) - 1.0f;
                const float forget = L.y * c;
                const float update = L.x * tanh_g;
                c = forget + update;
                h = L.w * tanh_ref_lean(c);
            }
            if (t >= my_tb) { h = 0.0f; c = 0.0f; hprev_own = 0.0f; }      // beyond this read's end (ragged batch)
            v4f hv;
            hv.x = __shfl(h, rl);
            hv.y = __shfl(h, rl + 16);
            hv.z = __shfl(h, rl + 32);
            hv.w = __shfl(h, rl + 48);
            float *ho = a.hout + ((size_t)t * a.B16 + rt) * tile_floats + (size_t)(ut0 + my_tile) * 64 + rl * 4;
            if (lane < 16) {
                if (fast) *(v4f *)ho = hv;
                else {
                    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.hout + ((size_t)t * a.B16 + rt) * tile_floats), 0, (int)(tile_floats * 4), 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, hv), wr, ((ut0 + my_tile) * 64 + rl * 4) * 4, 0, 16 /*sc1*/);
                }
            }
        }
    }
}



// ------------------------------------------------------------------------------------------
static int g_query_blocks = -1;      // >= 0: dispatch answers the occupancy query instead of launching

template <int KIND, int UPC, int KPW>
static void launch_one(hipStream_t s, const PersistArgs &a) {
    if (g_query_blocks >= 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_rnn_persist<KIND, UPC, KPW>, 256, 0) != hipSuccess) n = 0;
        g_query_blocks = n;
        return;
    }
    hipLaunchKernelGGL((k_rnn_persist<KIND, UPC, KPW>), dim3(a.nrt * a.G), dim3(256), 0, s, a);
}

template <int KIND, int UPC>
static bool dispatch_kpw(hipStream_t s, const PersistArgs &a, int kpw) {
    switch (kpw) {
    case 1: launch_one<KIND, UPC, 1>(s, a); return true;
    case 2: launch_one<KIND, UPC, 2>(s, a); return true;
    case 3: launch_one<KIND, UPC, 3>(s, a); return true;
    case 4: launch_one<KIND, UPC, 4>(s, a); return true;
    case 6: launch_one<KIND, UPC, 6>(s, a); return true;
    case 8: launch_one<KIND, UPC, 8>(s, a); return true;
    default: return false;
    }
}

// group size: the largest divisor of Ut that is <= 32 (one XCD's worth of CUs)
int persist_blocks_per_cu(int kind, int H);
static int pick_group(int Ut) {
    for (int g = 32; g >= 1; g--) if (Ut % g == 0) return g;
    return 1;
}

static int pick_kpw(int K16) {
    int kpw = (K16 + 3) / 4;
    if (kpw == 5) kpw = 6;
    if (kpw == 7) kpw = 8;
    return kpw;
}

bool persist_supported(int kind, int H, int ncu) {
    if (kind != 0 && kind != 1) return false;
    if (H % 16 != 0) return false;
    const int Ut = H / 4, K16 = H / 16;
    const int G = pick_group(Ut), UPC = Ut / G, kpw = pick_kpw(K16);
    if (UPC > 4 || kpw > 8) return false;
    if (G > 2 * ncu) return false;
    return true;
}

// words of the per-launch flag area: XCC ids [nrt][G]
size_t persist_flag_words(int H, int nrt) { return (size_t)nrt * pick_group(H / 4); }

// resident workgroups per CU of the kernel instantiation this shape selects (occupancy API; the
// kernels use <= 64 SGPRs, outside the band where the API over-reports -- MI355X_MICROARCH.md)
int persist_blocks_per_cu(int kind, int H) {
    g_query_blocks = 0;
    launch_rnn_persist(nullptr, kind, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, H, 0, 1, 0, 1, nullptr, nullptr);
    const int n = g_query_blocks;
    g_query_blocks = -1;
    return n;
}

int persist_max_tiles(int kind, int H, int ncu, int fused) {
    const int G = pick_group(H / 4);
    if (fused) {
        // k_lstm_fused: launch_bounds(256,2) guarantees two per CU (243 VGPRs at H = 384); smaller shapes need fewer
        // registers and admit three or four, which is what latency-bound layers want
        static int cache_per_cu[2][129];                          // [kind][H/16], 0 = not asked yet (host side, one engine thread per GPU)
        int &cached = cache_per_cu[kind & 1][(H / 16) & 127];
        if (cached == 0) {
            g_query_blocks = 0;
            launch_lstm_fused(nullptr, kind, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, H, 0, 1, 0, 1, nullptr, nullptr);
            cached = g_query_blocks > 0 ? g_query_blocks : 2;
            g_query_blocks = -1;
        }
        int per_cu = cached;
        if (per_cu < 2) per_cu = 2;
        if (per_cu > 4) per_cu = 4;
        return (per_cu * ncu) / G;
    }
    // every workgroup of a launch must be co-resident because groups spin on each other: at least two
    // per CU are guaranteed (launch_bounds(256,2), <= 33 KiB LDS); use what the occupancy query admits.
    int per_cu = persist_blocks_per_cu(kind, H);
    if (per_cu < 2) per_cu = 2;
    if (per_cu > 4) per_cu = 4;
    return (per_cu * ncu) / G;
}

// fused projection+recurrence needs 2*UPC*KPW weight fragments per lane in VGPRs next to the working
// set; beyond 18 fragment pairs (H = 384) two workgroups per CU no longer fit in 256 VGPRs.
bool fused_supported(int kind, int H) {
    if ((kind != 0 && kind != 1) || H % 16 != 0) return false;
    const int Ut = H / 4, K16 = H / 16;
    const int G = pick_group(Ut), UPC = Ut / G, kpw = pick_kpw(K16);
    return UPC <= 4 && kpw <= 8 && UPC * kpw <= 18;
}

template <int KIND, int UPC>
static bool dispatch_fused(hipStream_t s, const PersistArgs &a, int kpw) {
#define FUSED_CASE(K) case K: if (UPC * K <= 18) {                                                                       \
        if (g_query_blocks >= 0) {                                                                                        \
            int n = 0;                                                                                                    \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_lstm_fused<KIND, UPC, (UPC * K <= 18 ? K : 1)>, 256, 0) != hipSuccess) n = 0; \
            g_query_blocks = n;                                                                                           \
            return true;                                                                                                  \
        }                                                                                                                 \
        hipLaunchKernelGGL((k_lstm_fused<KIND, UPC, (UPC * K <= 18 ? K : 1)>), dim3(a.nrt * a.G), dim3(256), 0, s, a); return true; } return false;
    switch (kpw) {
    FUSED_CASE(1) FUSED_CASE(2) FUSED_CASE(3) FUSED_CASE(4) FUSED_CASE(6) FUSED_CASE(8)
    default: return false;
    }
#undef FUSED_CASE
}

bool launch_lstm_fused(hipStream_t s, int kind, const float4 *sWp, const float4 *iWp, const float *bias, const float *xin, float *hout,
                       unsigned *flags, unsigned *abort_word, int Tb, int B16, int H, int rt0, int nrt, int backward, int mode,
                       const int *tbs, const int *tbt) {
    PersistArgs a;
    a.sWp = (const v4f *)sWp; a.iWp = (const v4f *)iWp; a.bias = bias; a.xin = xin; a.xa = nullptr; a.hout = hout;
    a.flags = flags; a.abort_word = abort_word;
    a.Tb = Tb; a.B16 = B16; a.Ut = H / 4; a.K16 = H / 16; a.G = pick_group(a.Ut); a.rt0 = rt0; a.nrt = nrt;
    a.backward = backward; a.mode = mode;
    a.tbs = tbs; a.tbt = tbt;
    a.fast_gates = getenv("FFHIP_FAST_GATES") ? 1 : 0;
    const int UPC = a.Ut / a.G, kpw = pick_kpw(a.K16);
    if (kind == 0) {
        switch (UPC) {
        case 1: return dispatch_fused<0, 1>(s, a, kpw);
        case 2: return dispatch_fused<0, 2>(s, a, kpw);
        case 3: return dispatch_fused<0, 3>(s, a, kpw);
        case 4: return dispatch_fused<0, 4>(s, a, kpw);
        }
    } else {
        switch (UPC) {
        case 1: return dispatch_fused<1, 1>(s, a, kpw);
        case 2: return dispatch_fused<1, 2>(s, a, kpw);
        case 3: return dispatch_fused<1, 3>(s, a, kpw);
        case 4: return dispatch_fused<1, 4>(s, a, kpw);
        }
    }
    return false;
}

// One recurrent layer over read tiles [rt0, rt0+nrt).  nrt <= persist_max_tiles().
bool launch_rnn_persist(hipStream_t s, int kind, const float4 *sWp, const float *xa, float *hout, unsigned *flags,
                        unsigned *abort_word, int Tb, int B16, int H, int rt0, int nrt, int backward, int mode,
                        const int *tbs, const int *tbt) {
    PersistArgs a;
    a.sWp = (const v4f *)sWp; a.iWp = nullptr; a.xin = nullptr; a.bias = nullptr;
    a.xa = (const v4f *)xa; a.hout = hout; a.flags = flags; a.abort_word = abort_word;
    a.Tb = Tb; a.B16 = B16; a.Ut = H / 4; a.K16 = H / 16; a.G = pick_group(a.Ut); a.rt0 = rt0; a.nrt = nrt;
    a.backward = backward;
    a.mode = mode;
    a.tbs = tbs; a.tbt = tbt;
    a.fast_gates = 0;
   
    const int UPC = a.Ut / a.G, kpw = pick_kpw(a.K16);
    if (kind == 0) {
        switch (UPC) {
        case 1: return dispatch_kpw<0, 1>(s, a, kpw);
        case 2: return dispatch_kpw<0, 2>(s, a, kpw);
        case 3: return dispatch_kpw<0, 3>(s, a, kpw);
        case 4: return dispatch_kpw<0, 4>(s, a, kpw);
        }
    } else {
        switch (UPC) {
        case 1: return dispatch_kpw<1, 1>(s, a, kpw);
        case 2: return dispatch_kpw<1, 2>(s, a, kpw);
        case 3: return dispatch_kpw<1, 3>(s, a, kpw);
        case 4: return dispatch_kpw<1, 4>(s, a, kpw);
        }
    }
    return false;
}

}  // namespace ffhip
