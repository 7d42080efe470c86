// ffhip_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the flip-flop basecalling hot path.
//
// Every contraction runs on the matrix cores with the fp32-input MFMA v_mfma_f32_16x16x4_f32
// (exact f32, 64 FLOP/clk/SIMD, 157.3 TFLOP/s chip peak): the parity bar (1e-4 on transition
// scores through five recurrent layers, bit-exact base strings) rules out bf16/fp8 inputs.
// Operands live in HBM in MFMA *fragment order* (see ffhip_internal.hpp), so every operand fetch
// is one fully coalesced 1 KiB wave load of float4 that feeds four MFMAs.
//
// Reference functions replaced (paths relative to /root/reference/src):
//   conv_small / conv_mfma   convolution()            layers.c:189-276 (+ swish/tanh :24-49)
//   inproj                   feedforward_linear()     layers.c:279-283 -> affine_map flappie_matrix.c:361
//   lstm_step / gru_step     lstm_step / grumod_step  layers.c:979-1026 / :664-715
//   head + crf_norm          globalnorm_flipflop()    layers.c:1082-1106, partition fn :1035-1079
//   transpost                transpost_crf_flipflop() decode.c:377-497 + log_row_normalise flappie_matrix.c:450
//   viterbi                  decode_crf_flipflop()    decode.c:119-204
//   assemble                 change_positions + calculate_post loop   decode.c:66-79, flappie.c:284-292
//   trace                    exp_activation_inplace + trace_from_posterior   layers.c:56, decode.c:499-543
#include "ffhip_internal.hpp"
#include "ffhip_math.hpp"
#include <stdlib.h>

// Head and decode kernels run beside the NEXT batch's convolutions (batch_run_impl, FFHIP_DEBUG=front_order=...) and the next layer launches wait for them:
// their waves go first on a shared SIMD (the convolutions stay at priority 0)
#ifndef FFHIP_DECODE_PRIO
#define FFHIP_DECODE_PRIO 2
#endif
#define FFHIP_DECODE_PRIO_SET() __builtin_amdgcn_s_setprio(FFHIP_DECODE_PRIO)

namespace ffhip {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef _Float16 v8h_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ v4f mfma4(v4f a, v4f b, v4f c) {
    // four k-steps of 16x16x4: k = 16*k16 + 4*kq + {0,1,2,3}; the k order inside the 16 is free as
    // long as A and B agree, and both fragments use the same (kq, component) -> k map.
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// XCD-aware bijective remap of a 1-D grid: hardware places block b on XCD b%8; give each XCD a
// contiguous range of logical ids so that neighbours (which share operand panels) share an L2.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

// ------------------------------------------------------------------------------------------
// VALU convolution for the thin front layers (1->4, 4->16 features): one thread per output
// column, all filters.  `x0a/x0b[c]` are the first input samples of the (up to two) windows the
// reference accumulates into column c -- this is how the right-edge behaviour of layers.c:257-271
// is reproduced in index space; kNoWindow = none.  Input pads are zero, so partial windows at
// either edge are plain full-length windows here.
// ------------------------------------------------------------------------------------------
// KT > 0 (round 4): the window length in floats is a compile-time constant and Fout == MAXF -- the window comes as KT / 4 vector loads
// (one dword load per tap before), the weights lie tap-major in LDS and come four filters per ds_read_b128 (one ds_read_b32 per
// multiply-add before: the kernel was bound by LDS issue), the loops are unrolled.  Same multiply-adds in the same order: bit-identical.
template <int MAXF, int KT = 0>
__global__ void __launch_bounds__(256)
k_conv_small(SampleBuf in, SampleBuf out, const float *__restrict__ W, const float *__restrict__ bias,
             const int *__restrict__ x0a, const int *__restrict__ x0b, int Tout, int winlen, int act, int ldp,
             const int *__restrict__ tin, float split_scale, unsigned *__restrict__ sat, const int *__restrict__ seg) {
    extern __shared__ __attribute__((aligned(16))) float w_lds[];          // [Fout][winlen*Fin] (KT > 0: [winlen*Fin][Fout]) then bias [Fout]
    const int Fin = in.F, Fout = KT > 0 ? MAXF : out.F, K = KT > 0 ? KT : winlen * Fin;
    if (KT > 0) for (int i = threadIdx.x; i < Fout * K; i += blockDim.x) w_lds[(i % K) * MAXF + i / K] = W[i];
    else for (int i = threadIdx.x; i < Fout * K; i += blockDim.x) w_lds[i] = W[i];
    for (int i = threadIdx.x; i < Fout; i += blockDim.x) w_lds[Fout * K + i] = bias[i];
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= Tout) return;
    float acc[MAXF];
#pragma unroll
    for (int f = 0; f < MAXF; f++) acc[f] = (f < Fout) ? w_lds[Fout * K + f] : 0.0f;
    // ldp = 0: one window table shared by all reads; otherwise one row of ldp entries per read (ragged batch)
    // tin (stride-1 layers of a ragged batch): per-read input length; the window of column c starts at c - padL
    // (layers.c:216-271 degenerates to the zero-padded "same" convolution for stride 1), zeros beyond the read
    int xs[2];
    if (seg) {
        // stride-1 layer of a PACKED batch (several reads one behind the other in this row): seg[r] .. seg[r + 1] delimit the row's sorted read boundaries
        // {start, end, start, end, ...} behind the table's Bp + 1 offsets; column c belongs to a read iff an odd number of them is <= c
        int lo = seg[r], hi = seg[r + 1];
        const int first = lo;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg[mid] <= c) lo = mid + 1; else hi = mid; }
        xs[0] = ((lo - first) & 1) ? c - (winlen - 1) / 2 : kZeroCol; xs[1] = kNoWindow;
    } else if (tin) { xs[0] = (c < tin[r]) ? c - (winlen - 1) / 2 : kZeroCol; xs[1] = kNoWindow; }
    else { xs[0] = x0a[(size_t)r * ldp + c]; xs[1] = x0b[(size_t)r * ldp + c]; }
    if (xs[0] == kZeroCol) {                     // beyond this read's end: the next layer must see zero padding there
        float *o0 = out.p + (size_t)r * out.rs + (size_t)(kSamplePad + c) * Fout;
        for (int f = 0; f < Fout; f++) o0[f] = 0.0f;
        return;
    }
#pragma unroll
    for (int wdw = 0; wdw < 2; wdw++) {
        if (xs[wdw] == kNoWindow) continue;
        const float *x = in.row(r, xs[wdw]);
        if constexpr (KT > 0) {
            float xv[KT];
            if constexpr (KT % 4 == 0) {              // (Fin = 4: a sample is 16 bytes, every window 16-byte aligned)
#pragma unroll
                for (int k = 0; k < KT; k += 4) { const float4 v = *(const float4 *)(x + k); xv[k] = v.x; xv[k + 1] = v.y; xv[k + 2] = v.z; xv[k + 3] = v.w; }
            } else {
#pragma unroll
                for (int k = 0; k < KT; k++) xv[k] = x[k];
            }
            // (pairs of filters as 2-vectors: v_pk_mul_f32 + v_pk_add_f32 -- this file is built without the SLP vectoriser, DESIGN.md section 5.4;
            // each component is the same multiply and the same add)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a2[MAXF / 2];
#pragma unroll
            for (int f = 0; f < MAXF; f += 2) a2[f / 2] = (f2){ acc[f], acc[f + 1] };
#pragma unroll
            for (int k = 0; k < KT; k++) {
                const f2 xk = { xv[k], xv[k] };
#pragma unroll
                for (int f = 0; f < MAXF; f += 4) {
                    const float4 w = *(const float4 *)__builtin_assume_aligned(&w_lds[k * MAXF + f], 16);
                    a2[f / 2] = a2[f / 2] + (f2){ w.x, w.y } * xk;
                    a2[f / 2 + 1] = a2[f / 2 + 1] + (f2){ w.z, w.w } * xk;
                }
            }
#pragma unroll
            for (int f = 0; f < MAXF; f += 2) { acc[f] = a2[f / 2].x; acc[f + 1] = a2[f / 2].y; }
        } else
        for (int k = 0; k < K; k++) {
            const float xv = x[k];
#pragma unroll
            for (int f = 0; f < MAXF; f++)
                if (f < Fout) acc[f] = acc[f] + w_lds[f * K + k] * xv;
        }
    }
    float *o = out.p + (size_t)r * out.rs + (size_t)(kSamplePad + c) * Fout;
    if (MAXF == 16 && split_scale != 0.0f) {
        // the next layer is the split-operand convolution (k_conv_split): this sample's 16 features leave as kSplitNS 16-bit
        // slices of value * split_scale (ffhip_split.hpp) in the SAME 64-byte row -- [slice][16 features] -- the zero rows of
        // the padding are zeros in either reading
        unsigned pk[2][8];                   // [slice][feature pair]: the row leaves as four 16-byte stores, not 32 two-byte ones
#pragma unroll
        for (int f = 0; f < 16; f += 4) {
            const ffv4 y = apply_act4((ffv4){ acc[f], acc[f + 1], acc[f + 2], acc[f + 3] }, act) * split_scale;
            if (sat && split_overflow(y)) sat[r] = 1u;      // beyond the split format: the engine re-runs this read on the f32 path (ffhip_batch_finish)
            unsigned s0[kSplitNS], s1[kSplitNS], s2[kSplitNS], s3[kSplitNS];
            split_slices<true>(y.x, s0); split_slices<true>(y.y, s1); split_slices<true>(y.z, s2); split_slices<true>(y.w, s3);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                pk[k][f >> 1] = s0[k % kSplitNS] | (s1[k % kSplitNS] << 16);
                pk[k][(f >> 1) + 1] = s2[k % kSplitNS] | (s3[k % kSplitNS] << 16);
            }
        }
        uint4 *o4 = (uint4 *)o;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            o4[2 * k] = make_uint4(pk[k][0], pk[k][1], pk[k][2], pk[k][3]);
            o4[2 * k + 1] = make_uint4(pk[k][4], pk[k][5], pk[k][6], pk[k][7]);
        }
        return;
    }
#pragma unroll
    for (int f = 0; f < MAXF; f += 4) {
        const ffv4 y = apply_act4((ffv4){ acc[f], acc[f + 1], acc[f + 2], acc[f + 3] }, act);
        if (f < Fout) o[f] = y.x;
        if (f + 1 < Fout) o[f + 1] = y.y;
        if (f + 2 < Fout) o[f + 2] = y.z;
        if (f + 3 < Fout) o[f + 3] = y.w;
    }
}

void launch_conv_small(hipStream_t s, SampleBuf in, SampleBuf out, const float *W, const float *bias,
                       const int *x0a, const int *x0b, int Bp, int Tout, int winlen, int act, int ldp, const int *tin, int split_exp, unsigned *sat, const int *seg) {
    dim3 grid((Tout + 255) / 256, Bp), block(256);
    const float split_scale = (split_exp > -1000 && out.F == 16 && kSplitNS == 2) ? split_pow2(split_exp) : 0.0f;
    const size_t lds = (size_t)(out.F * winlen * in.F + out.F) * sizeof(float);
    const char *su_env = dbg("conv_small_u");
    const bool unrolled = !(su_env && su_env[0] == '0');      // (=0: the round-3 loops, for comparison)
    if (unrolled && out.F == 4 && in.F == 1 && winlen == 5)
        hipLaunchKernelGGL((k_conv_small<4, 5>), grid, block, lds, s, in, out, W, bias, x0a, x0b, Tout, winlen, act, ldp, tin, 0.0f, sat, seg);
    else if (unrolled && out.F == 16 && in.F == 4 && winlen == 5)
        hipLaunchKernelGGL((k_conv_small<16, 20>), grid, block, lds, s, in, out, W, bias, x0a, x0b, Tout, winlen, act, ldp, tin, split_scale, sat, seg);
    else if (out.F <= 4)
        hipLaunchKernelGGL(k_conv_small<4>, grid, block, lds, s, in, out, W, bias, x0a, x0b, Tout, winlen, act, ldp, tin, 0.0f, sat, seg);
    else if (out.F <= 16)
        hipLaunchKernelGGL(k_conv_small<16>, grid, block, lds, s, in, out, W, bias, x0a, x0b, Tout, winlen, act, ldp, tin, split_scale, sat, seg);
    else
        hipLaunchKernelGGL(k_conv_small<32>, grid, block, lds, s, in, out, W, bias, x0a, x0b, Tout, winlen, act, ldp, tin, 0.0f, sat, seg);
}

// ------------------------------------------------------------------------------------------
// MFMA tile engine shared by the last convolution, the input projections and the CRF head.
// A workgroup = 4 waves arranged 2 (M) x 2 (N); each wave owns TM x TN tiles of 16x16.
// A fragments: packed weights, `Wp + (mt*K16 + k16)*64 + lane`.
// B fragments: per-lane pointer + k16 * bstep floats (tile-interleaved activations: contiguous
// 1 KiB per k16; convolution windows: 16 consecutive floats of the window per k16).
// Loads are register double-buffered; the f32 MFMA is slow enough (32 cycles per instruction)
// that 8 coalesced 1 KiB loads per 64 MFMAs hide behind the matrix pipe.
// ------------------------------------------------------------------------------------------
template <int TM, int TN, bool BVEC>
__device__ __forceinline__ void mma_tiles(const v4f *(&ap)[TM], const float *(&bp)[TN], size_t bstep, int K16,
                                          v4f (&acc)[TM][TN]) {
    v4f a0[TM], b0[TN], a1[TM], b1[TN];
    auto loadA = [&](v4f(&a)[TM], int k16) {
#pragma unroll
        for (int i = 0; i < TM; i++) a[i] = ap[i][(size_t)k16 * 64];
    };
    auto loadB = [&](v4f(&b)[TN], int k16) {
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const float *p = bp[j] + (size_t)k16 * bstep;
            if (BVEC) b[j] = *(const v4f *)p;
            else { b[j].x = p[0]; b[j].y = p[1]; b[j].z = p[2]; b[j].w = p[3]; }
        }
    };
    auto mma = [&](v4f(&a)[TM], v4f(&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = mfma4(a[i], b[j], acc[i][j]);
    };
    loadA(a0, 0); loadB(b0, 0);
    int k16 = 0;
    for (; k16 + 2 <= K16; k16 += 2) {
        loadA(a1, k16 + 1); loadB(b1, k16 + 1);
        mma(a0, b0);
        if (k16 + 2 < K16) { loadA(a0, k16 + 2); loadB(b0, k16 + 2); }
        mma(a1, b1);
    }
    if (k16 < K16) mma(a0, b0);
}

// ---- last convolution (implicit GEMM over the window) -------------------------------------
// N tile = (output column c, read tile rt); all 16 reads share the window start.  Columns with a
// second window (or none) are irregular and rare: handled by re-running the loop for window b.
// TN (column tiles a wave; round 4).  A one-feature input (the r941_5mC model's only convolution: 19 taps, K16 = 2) has 8 MFMAs a tile against ~130
// VALU instructions of swish + split in the epilogue, and at TN = 4 the kernel holds 230 registers: two waves a SIMD.  FFHIP_DEBUG=conv1_tn=2 (156 registers,
// three waves) is 13 % faster ALONE (1.00 -> 0.88 ms for a 1024-read batch) -- and slower in the pipeline, where this convolution runs beside the previous
// batch's head and decode and the next layer launches wait for THOSE: 100.3 -> 99.0 Msamples/s at `c4` (TN = 1: 98.7).  Default 4; bit-identical all three.
template <bool BVEC, int TN = 4, int WPS = 1>
__global__ void __launch_bounds__(256, WPS)
k_conv_mfma(SampleBuf in, float *__restrict__ out, const v4f *__restrict__ Wp, const float *__restrict__ bias,
            const int *__restrict__ x0a, const int *__restrict__ x0b, int B16, int Tout, int Mt, int K16, int act, int ldp,
            unsigned char *__restrict__ out_split, float split_scale, unsigned *__restrict__ sat) {
    constexpr int TM = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int nMblk = (Mt + 2 * TM - 1) / (2 * TM);
    const int ntile = Tout * B16;
    const int nNblk = (ntile + 2 * TN - 1) / (2 * TN);
    const int L = xcd_remap(blockIdx.x, nMblk * nNblk);
    const int mblk = L % nMblk, nblk = L / nMblk;
    const int mt0 = (mblk * 2 + wm) * TM, nt0 = (nblk * 2 + wn) * TN;
    const int kq = lane >> 4, rl = lane & 15;

    v4f acc[TM][TN];
    const v4f *ap[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = min(mt0 + i, Mt - 1);
        ap[i] = Wp + (size_t)mt * K16 * 64 + lane;
        const v4f bv = *(const v4f *)(bias + mt * 16 + kq * 4);
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = bv;
    }
    for (int pass = 0; pass < 2; pass++) {
        const float *bp[TN];
        bool any = false;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nt = min(nt0 + j, ntile - 1);
            const int c = nt / B16, rt = nt % B16;
            const size_t pi = (size_t)(rt * 16 + rl) * ldp + c;       // ldp = 0: shared table; else this lane's read has its own row
            int x0 = pass == 0 ? x0a[pi] : x0b[pi];
            // a missing window contributes zero: point it at the leading zero pad
            const bool have = (x0 != kNoWindow && x0 != kZeroCol);
            any |= have;
            if (!have) x0 = -kSamplePad;
            bp[j] = in.p + (size_t)(rt * 16 + rl) * in.rs + (size_t)(kSamplePad + x0) * in.F + kq * 4;
        }
        // skip the whole second pass when no lane of the wave has a second window
        if (pass == 1 && !__any(any)) break;
        mma_tiles<TM, TN, BVEC>(ap, bp, 16, K16, acc);
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = mt0 + i;
        if (mt >= Mt) continue;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nt = nt0 + j;
            if (nt >= ntile) continue;
            v4f v = acc[i][j];
            v = apply_act4(v, act);
            if (out_split) {
                // split layout of the recurrent layer kernel (ffhip_rnn_split.hip, ffhip_split.hpp): kSplitNS 16-bit slices of
                // value * 2^split_exp, [k/32][slice][(k%32)/8 * 16 + read][8]; this lane holds k = 16 mt + 4 kq + 0..3 of read rl
                unsigned char *dst = out_split + (size_t)nt * ((size_t)Mt * 16 * 32 * kSplitNS) +
                                     (size_t)(((mt >> 1) * kSplitNS * 64 + ((mt & 1) * 2 + (kq >> 1)) * 16 + rl) * 16 + (kq & 1) * 8);
                const float f[4] = { v.x * split_scale, v.y * split_scale, v.z * split_scale, v.w * split_scale };
                if (sat && split_overflow((v4f){ f[0], f[1], f[2], f[3] })) sat[(nt % B16) * 16 + rl] = 1u;      // see k_conv_small
                unsigned sb[4][kSplitNS];
#pragma unroll
                for (int e = 0; e < 4; e++) split_slices<true>(f[e], sb[e]);
#pragma unroll
                for (int sl = 0; sl < kSplitNS; sl++)
                    *(uint2 *)(dst + (size_t)sl * 1024) = make_uint2(sb[0][sl] | (sb[1][sl] << 16), sb[2][sl] | (sb[3][sl] << 16));
            } else
                *(v4f *)(out + ((size_t)nt * Mt + mt) * 256 + lane * 4) = v;
        }
    }
}

void launch_conv_mfma(hipStream_t s, SampleBuf in, float *out, const float4 *Wp, const float *bias,
                      const int *x0a, const int *x0b, int B16, int Tout, int M, int K16, int act, int ldp, void *out_split, int split_exp, unsigned *sat) {
    const int Mt = M / 16;
    const int nMblk = (Mt + 7) / 8, nNblk = (Tout * B16 + 7) / 8;
    const bool vec = (in.F % 4 == 0);
    const char *tn_env = dbg("conv1_tn");
    const int thin_tn = tn_env ? atoi(tn_env) : 4;
    if (!vec && K16 <= 2 && thin_tn < 4) {
        const int tn = thin_tn <= 1 ? 1 : 2, nNb = (Tout * B16 + 2 * tn - 1) / (2 * tn);
        if (tn == 1)
            hipLaunchKernelGGL((k_conv_mfma<false, 1, 4>), dim3(nMblk * nNb), dim3(256), 0, s, in, out, (const v4f *)Wp, bias,
                               x0a, x0b, B16, Tout, Mt, K16, act, ldp, (unsigned char *)out_split, split_pow2(split_exp), sat);
        else
            hipLaunchKernelGGL((k_conv_mfma<false, 2, 3>), dim3(nMblk * nNb), dim3(256), 0, s, in, out, (const v4f *)Wp, bias,
                               x0a, x0b, B16, Tout, Mt, K16, act, ldp, (unsigned char *)out_split, split_pow2(split_exp), sat);
    } else if (vec)
        hipLaunchKernelGGL(k_conv_mfma<true>, dim3(nMblk * nNblk), dim3(256), 0, s, in, out, (const v4f *)Wp, bias,
                           x0a, x0b, B16, Tout, Mt, K16, act, ldp, (unsigned char *)out_split, split_pow2(split_exp), sat);
    else
        hipLaunchKernelGGL(k_conv_mfma<false>, dim3(nMblk * nNblk), dim3(256), 0, s, in, out, (const v4f *)Wp, bias,
                           x0a, x0b, B16, Tout, Mt, K16, act, ldp, (unsigned char *)out_split, split_pow2(split_exp), sat);
}

// ---- last convolution on split operands ---------------------------------------------------------------------------
// The same implicit GEMM as k_conv_mfma for a 16-feature input (the r941 / r103 / rle models: 16 -> H, 19 taps), on the 16-bit
// matrix pipes over two fp16 slices of both operands (ffhip_split.hpp: three products per fp32 multiply-add, accuracy of an
// fp32 GEMM; 30 MFMAs of 16 cycles per 16 x 16 tile instead of 76 of 32).  K = tap * 16 + feature; a K chunk of 32 is two taps.
//   input    sample-major rows of 64 bytes [slice][16 features] fp16, written by k_conv_small (value * 2^kSplitExpX)
//            B operand of chunk c, lane (kq, read r): 16 bytes at row x0 + 2c + (kq >> 1), slice s, features 8 (kq & 1) ..
//   weights  Wp[mt][c][slice][lane] 16 B: row 16 mt + (lane & 15), k = 32 c + 8 (lane >> 4) .., zero beyond K (the phantom
//            20th tap); a lane whose tap is the phantom one reads a ZERO row instead of the sample behind the window (0 x NaN)
// Accumulators live in the scaled space 2^S (S = weight exponent + kSplitExpX; bias pre-multiplied, result multiplied by 2^-S).
// Two shapes.  <4, 4>: 4 x 4 tiles per wave, 216 VGPRs -- the fast one when the kernel has the chip to itself.  <2, 2>: 2 x 2 tiles,
// <= 128 VGPRs (launch bound 2 waves per SIMD) -- the one that FITS BESIDE another batch's layer launch (two waves of 192 VGPRs per
// SIMD leave 128): with two batches in flight the big shape only ran in the gaps between the other batch's launches and its
// remainder (~0.35 ms) stood between that batch's last layer and this batch's first.
template <int TM, int TN>
__global__ void __launch_bounds__(256, 2)      // (two waves per SIMD for both shapes; <4, 4> without a bound took 216 + 64 registers and ran ONE wave per SIMD.  Measured and dropped: <4, 2> and <2, 4> at three waves per SIMD, 0.39 and 0.52 ms against 0.30)
k_conv_split(SampleBuf in, float *__restrict__ out, const v4u_t *__restrict__ Wp, const float *__restrict__ bias,
             const int *__restrict__ x0a, const int *__restrict__ x0b, int B16, int Tout, int Mt, int NC, int winlen, int act, int ldp,
             unsigned char *__restrict__ out_split, float split_scale, float acc_scale, unsigned *__restrict__ sat) {
    constexpr int NSL = 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: the weight tiles' addresses then are scalar bases + the lane)
    const int wm = wave & 1, wn = wave >> 1;
    const int nMblk = (Mt + 2 * TM - 1) / (2 * TM);
    const int ntile = Tout * B16;
    const int nNblk = (ntile + 2 * TN - 1) / (2 * TN);
    const int L = xcd_remap(blockIdx.x, nMblk * nNblk);
    const int mblk = L % nMblk, nblk = L / nMblk;
    const int mt0 = (mblk * 2 + wm) * TM, nt0 = (nblk * 2 + wn) * TN;
    const int kq = lane >> 4, rl = lane & 15;
    v4f acc[TM][TN];
    const v4u_t *ap[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = min(mt0 + i, Mt - 1);
        ap[i] = Wp + (size_t)mt * NC * NSL * 64 + lane;
        const v4f bv = *(const v4f *)(bias + mt * 16 + kq * 4) * acc_scale;
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = bv;
    }
    const unsigned char *zero_row = (const unsigned char *)in.p;            // the leading pad of read 0: zeros
    for (int pass = 0; pass < 2; pass++) {
        const unsigned char *bp[TN];
        bool any = false;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nt = min(nt0 + j, ntile - 1);
            const int c = nt / B16, rt = nt % B16;
            const size_t pi = (size_t)(rt * 16 + rl) * ldp + c;
            int x0 = pass == 0 ? x0a[pi] : x0b[pi];
            const bool have = (x0 != kNoWindow && x0 != kZeroCol);
            any |= have;
            if (!have) x0 = -kSamplePad;
            bp[j] = (const unsigned char *)(in.p + (size_t)(rt * 16 + rl) * in.rs + (size_t)(kSamplePad + x0) * 16) + (kq >> 1) * 64 + (kq & 1) * 16;
        }
        if (pass == 1 && !__any(any)) break;
        v4u_t A0[TM][NSL], B0[TN][NSL], A1[TM][NSL], B1[TN][NSL];
        auto load = [&](v4u_t (&A)[TM][NSL], v4u_t (&B)[TN][NSL], int c) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int sl = 0; sl < NSL; sl++) A[i][sl] = ap[i][(size_t)(c * NSL + sl) * 64];
            const bool phantom = (2 * c + (kq >> 1) >= winlen);
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const unsigned char *p = phantom ? zero_row : bp[j] + (size_t)c * 128;
#pragma unroll
                for (int sl = 0; sl < NSL; sl++) B[j][sl] = *(const v4u_t *)(p + sl * 32);
            }
        };
        auto mma = [&](v4u_t (&A)[TM][NSL], v4u_t (&B)[TN][NSL]) {
            constexpr int WS[3] = { 1, 0, 0 }, XS[3] = { 0, 1, 0 };
#pragma unroll
            for (int term = 0; term < 3; term++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h_t, A[i][WS[term]]), __builtin_bit_cast(v8h_t, B[j][XS[term]]), acc[i][j], 0, 0, 0);
        };
        load(A0, B0, 0);
        for (int c = 0; c < NC; c += 2) {
            if (c + 1 < NC) load(A1, B1, c + 1);
            mma(A0, B0);
            if (c + 1 < NC) {
                if (c + 2 < NC) load(A0, B0, c + 2);
                mma(A1, B1);
            }
        }
    }
    const float inv_scale = 1.0f / acc_scale;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = mt0 + i;
        if (mt >= Mt) continue;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nt = nt0 + j;
            if (nt >= ntile) continue;
            v4f v = acc[i][j] * inv_scale;
            v = apply_act4(v, act);
            if (out_split) {
                unsigned char *dst = out_split + (size_t)nt * ((size_t)Mt * 16 * 32 * kSplitNS) +
                                     (size_t)(((mt >> 1) * kSplitNS * 64 + ((mt & 1) * 2 + (kq >> 1)) * 16 + rl) * 16 + (kq & 1) * 8);
                const float f[4] = { v.x * split_scale, v.y * split_scale, v.z * split_scale, v.w * split_scale };
                if (sat && split_overflow((v4f){ f[0], f[1], f[2], f[3] })) sat[(nt % B16) * 16 + rl] = 1u;      // see k_conv_small
                unsigned sb[4][kSplitNS];
#pragma unroll
                for (int e = 0; e < 4; e++) split_slices<true>(f[e], sb[e]);
#pragma unroll
                for (int sl = 0; sl < kSplitNS; sl++)
                    *(uint2 *)(dst + (size_t)sl * 1024) = make_uint2(sb[0][sl] | (sb[1][sl] << 16), sb[2][sl] | (sb[3][sl] << 16));
            } else
                *(v4f *)(out + ((size_t)nt * Mt + mt) * 256 + lane * 4) = v;
        }
    }
}

// ---- the same convolution, weights stationary (round 4) -------------------------------------------------------------------------
// k_conv_split<4, 4> spends its time fetching operands: every wave loads its own 4 weight tiles and 4 window tiles per K chunk (16 KiB for 48
// MFMAs), the 491 KB of weight slices do not fit the 32-KiB vector L1 (hit rate 0: TCP_TCC_READ_REQ x 128 B = all 3.07 GB of its loads come from
// L2, profiles/r04_front_pmc.txt), MFMA busy 0.21.  Here a wave KEEPS the weight slices of its 2 row tiles for the whole window in registers
// (160 VGPRs, loaded once), a workgroup of 4 waves covers 8 row tiles, and the window tile of (column, read tile) -- 20 pieces of 1 KiB: [K chunk]
// [slice][lane] -- is gathered ONCE per workgroup straight into LDS (`buffer_load ... lds`, five pieces a wave, the next tile's under this
// tile's MFMAs) and read from there by all four waves: per (column, read tile) a workgroup moves 20 KiB through the L1 instead of 80 and no
// weights at all.  Two workgroups a CU (<= 256 registers): one's 60 MFMAs per tile run under the other's epilogue (the 848 VALU cycles a tile of
// reference-exact swish, split and store costs).  Workgroups of one XCD (blockIdx mod 8) share their window tiles through that XCD's L2: the
// M blocks of a column group sit on the same XCD.  Results bit-identical to k_conv_split (same products in the same order per accumulator).
// -DFFHIP_FORCE_SKEW=1 (tools/test_hooks/libffhip_skew.so, tests/test_resweep_gpu.py; round 6): one wave in seven, rotating with the tile, the site, the wave and the
// workgroup, sits out ~2000 cycles in front of the counted wait, behind the barrier, behind its gather and in front of its epilogue -- the LDS-DMA gather, the DS reads of
// the other three waves and the LDS-only barrier between them must give the release library's bits with any wave late anywhere.
#ifndef FFHIP_FORCE_SKEW
#define FFHIP_FORCE_SKEW 0
#endif
#if FFHIP_FORCE_SKEW
#define WS_SKEW(site) do { if ((((unsigned)(nt / ngroup) * 5u + (unsigned)(site) * 3u + (unsigned)wave + blockIdx.x) % 7u) == 0u) __builtin_amdgcn_s_sleep(32); } while (0)
#else
#define WS_SKEW(site) do { } while (0)
#endif
template <int NC>
__global__ void __launch_bounds__(256, 2)
k_conv_split_ws(SampleBuf in, const v4u_t *__restrict__ Wp, const float *__restrict__ bias, const int *__restrict__ x0a, const int *__restrict__ x0b,
                int B16, int Tout, int Mt, int winlen, int ldp, unsigned char *__restrict__ out_split, float split_scale, float acc_scale,
                unsigned *__restrict__ sat, int ngroup) {
    constexpr int act = 1;                                      // swish: the LSTM models' convolutions (the only ones with 16 input features)
    constexpr int NSL = 2, TM = 2, NPIECE = NC * NSL, PPW = NPIECE / 4;      // K chunks of 32 = two taps; 19 taps -> NC = 10 chunks, 20 pieces, 5 a wave
    static_assert(NPIECE % 4 == 0, "the pieces of a window tile are dealt evenly to the four waves");
    __shared__ v4u_t Bt[2][NPIECE][64];                        // [buffer][piece = chunk * 2 + slice][lane]: 2 x 20 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, rl = lane & 15;
    const int NMB = Mt / (4 * TM);
    const int mb = (blockIdx.x >> 3) % NMB, g = (blockIdx.x & 7) + 8 * (int)(blockIdx.x / (8 * NMB));      // the M blocks of column group g: same XCD
    const int ntile = Tout * B16;
    const int mt0 = mb * 4 * TM + wave * TM;
    // my weights, for the whole layer
    v4u_t A[TM][NC][NSL];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int sl = 0; sl < NSL; sl++) A[i][c][sl] = Wp[((size_t)(mt0 + i) * NC + c) * NSL * 64 + sl * 64 + lane];
    v4f bv[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) bv[i] = *(const v4f *)(bias + (mt0 + i) * 16 + kq * 4) * acc_scale;
    const float inv_scale = 1.0f / acc_scale;
    // window starts of a tile, this lane's read: both passes (x0b: the second window some columns accumulate, layers.c:257-271)
    auto table = [&](int nt, int &xa, int &xb) {
        const int t = nt < ntile ? nt : ntile - 1;
        const size_t pi = (size_t)((t % B16) * 16 + rl) * ldp + t / B16;
        xa = x0a[pi]; xb = x0b[pi];
    };
    // gather my five pieces of (tile nt, window start x0 of this lane's read) into buffer `buf`: a tap beyond the window, or no window at
    // all, reads the leading zero pad of read 0.  Offsets come from registers only: nothing here waits for memory.
    // (the buffer resource starts at the read TILE's first row: offsets are 32 bits, and the rows of a long-read batch -- 512 rows of 200 000 samples are 6.7 GB -- pass
    // that as offsets from the buffer's start; sixteen rows do not)
    auto gather = [&](int nt, int x0, int buf) {
        const bool have = (x0 != kNoWindow && x0 != kZeroCol);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(in.p + (size_t)((nt % B16) * 16) * in.rs), 0, (int)0xFFFFFFFFu, 0x00020000);
        const unsigned row0 = (unsigned)(((size_t)rl * in.rs + (size_t)(kSamplePad + (have ? x0 : 0)) * 16) * 4) + (unsigned)((kq >> 1) * 64 + (kq & 1) * 16);
#pragma unroll
        for (int k = 0; k < PPW; k++) {
            const int piece = wave * PPW + k, ch = piece / NSL, sl = piece % NSL;
            const unsigned off = (!have || 2 * ch + (kq >> 1) >= winlen) ? (unsigned)((kq & 1) * 16 + sl * 32) : row0 + (unsigned)(ch * 128 + sl * 32);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)&Bt[buf][piece][0], 16, off, 0, 0, 0);
#endif
        }
    };
    auto valid = [&](int x0) { return __any(x0 != kNoWindow && x0 != kZeroCol) != 0; };
    // work list of this workgroup: tiles g, g + ngroup, ...; a tile with a second window is two passes over the same accumulators.
    // The window table is read one tile AHEAD of the gather that uses it (cur_*: this tile, nxt_*: the next), the gather one item ahead of
    // the MFMAs that use it: at the top of an iteration everything older than the last epilogue's four stores has arrived.
    // (Measured and dropped: starting every other workgroup of a CU half an iteration late -- are the two in each other's way by running in
    // phase? -- 0.305-0.308 ms per convolution group either way.)
    int nt = g, pass = 0, buf = 0;
    int cur_a, cur_b, nxt_a, nxt_b;
    table(nt, cur_a, cur_b);
    table(nt + ngroup, nxt_a, nxt_b);
    if (nt < ntile) gather(nt, cur_a, 0);
    v4f acc[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) acc[i] = bv[i];
    bool stored = false;                                        // the previous iteration ended with an epilogue: 2 TM stores younger than my gather
    while (nt < ntile) {
        // my pieces have landed: everything but the epilogue's stores, which were issued behind them and drain on their own (vector memory
        // operations of a wave retire in order on this family)
        WS_SKEW(0);
        if (stored) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_assert(TM * kSplitNS == 4, "the counted wait above assumes four stores per epilogue");
        // ... everybody's have, and nobody still reads the other buffer.  An LDS-only barrier: __syncthreads() drains vmcnt -- the stores again
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        WS_SKEW(1);
        const bool two = (pass == 0) && valid(cur_b);
        int nn_a = nxt_a, nn_b = nxt_b;
        if (two) gather(nt, cur_b, buf ^ 1);
        else {
            if (nt + ngroup < ntile) gather(nt + ngroup, nxt_a, buf ^ 1);
            table(nt + 2 * ngroup, nn_a, nn_b);                // (behind the gather: its wait is the next iteration's)
        }
        WS_SKEW(2);
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const v4u_t b0 = Bt[buf][c * NSL][lane], b1 = Bt[buf][c * NSL + 1][lane];
            // smallest terms first, as k_conv_split: w1 x0, w0 x1, w0 x0
#pragma unroll
            for (int i = 0; i < TM; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h_t, A[i][c][1]), __builtin_bit_cast(v8h_t, b0), acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h_t, A[i][c][0]), __builtin_bit_cast(v8h_t, b1), acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h_t, A[i][c][0]), __builtin_bit_cast(v8h_t, b0), acc[i], 0, 0, 0);
        }
        stored = !two;
        WS_SKEW(3);
        if (!two) {
            // epilogue of tile nt (k_conv_split's, value for value).  (Built and measured, not kept: the epilogue one iteration late in the MFMAs'
            // own block, branch-free, for the scheduler to interleave -- it did not, 0.33 against 0.32 ms.)
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int mt = mt0 + i;
                v4f v = acc[i] * inv_scale;
                v = apply_act4(v, act);
                unsigned char *dst = out_split + (size_t)nt * ((size_t)Mt * 16 * 32 * kSplitNS) +
                                     (size_t)(((mt >> 1) * kSplitNS * 64 + ((mt & 1) * 2 + (kq >> 1)) * 16 + rl) * 16 + (kq & 1) * 8);
                const float f[4] = { v.x * split_scale, v.y * split_scale, v.z * split_scale, v.w * split_scale };
                if (sat && split_overflow((v4f){ f[0], f[1], f[2], f[3] })) sat[(nt % B16) * 16 + rl] = 1u;
                unsigned sb[4][kSplitNS];
#pragma unroll
                for (int e = 0; e < 4; e++) split_slices<true>(f[e], sb[e]);
#pragma unroll
                for (int sl = 0; sl < kSplitNS; sl++)
                    *(uint2 *)(dst + (size_t)sl * 1024) = make_uint2(sb[0][sl] | (sb[1][sl] << 16), sb[2][sl] | (sb[3][sl] << 16));
                acc[i] = bv[i];
            }
            nt += ngroup; pass = 0;
            cur_a = nxt_a; cur_b = nxt_b; nxt_a = nn_a; nxt_b = nn_b;
        } else pass = 1;
        buf ^= 1;
    }
}

void launch_conv_split(hipStream_t s, SampleBuf in, float *out, const void *Wp, const float *bias, const int *x0a, const int *x0b,
                       int B16, int Tout, int M, int winlen, int act, int ldp, void *out_split, int split_exp, int acc_exp, int lean, unsigned *sat) {
    const int Mt = M / 16, NC = (winlen + 1) / 2;
    // the weights-stationary form: split output, the chip to itself, shapes it is built for (FFHIP_DEBUG=conv_ws=0: the round-3 kernel)
    const char *ws_txt = dbg("conv_ws");
    const int ws_env = ws_txt ? atoi(ws_txt) : 1;
    if (!lean && out_split && kSplitNS == 2 && ws_env && act == ACT_SWISH && Mt % 8 == 0 && NC == 10 && (size_t)in.rs * 4 * 16 < ((size_t)1 << 32)) {
        // (per call, of the CURRENT device -- the caller's engine has set it: a value cached from the first call would size the groups of an engine on
        // another device of the process by the wrong chip)
        int ncu = 256;
        {
            static int by_dev[32];                                // (0: not asked yet; plain ints, any thread writes the same value)
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 32) {
                if (!by_dev[dev] && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) by_dev[dev] = n;
                if (by_dev[dev]) ncu = by_dev[dev];
            }
        }
        const int NMB = Mt / 8;
        int ngroup = (2 * ncu / NMB) & ~7;                       // two workgroups a CU; whole XCD rounds of column groups
        if (ngroup < 8) ngroup = 8;
        const int ntile = Tout * B16;
        if (ngroup > ((ntile + 7) & ~7)) ngroup = (ntile + 7) & ~7;
        hipLaunchKernelGGL((k_conv_split_ws<10>), dim3(NMB * ngroup), dim3(256), 0, s, in, (const v4u_t *)Wp, bias, x0a, x0b, B16, Tout, Mt, winlen, ldp,
                           (unsigned char *)out_split, split_pow2(split_exp), split_pow2(acc_exp), sat, ngroup);
        return;
    }
    if (lean) {
        const int nMblk = (Mt + 3) / 4, nNblk = (Tout * B16 + 3) / 4;
        hipLaunchKernelGGL((k_conv_split<2, 2>), dim3(nMblk * nNblk), dim3(256), 0, s, in, out, (const v4u_t *)Wp, bias, x0a, x0b, B16, Tout, Mt, NC, winlen, act, ldp,
                           (unsigned char *)out_split, split_pow2(split_exp), split_pow2(acc_exp), sat);
        return;
    }
    const int nMblk = (Mt + 7) / 8, nNblk = (Tout * B16 + 7) / 8;
    hipLaunchKernelGGL((k_conv_split<4, 4>), dim3(nMblk * nNblk), dim3(256), 0, s, in, out, (const v4u_t *)Wp, bias, x0a, x0b, B16, Tout, Mt, NC, winlen, act, ldp,
                       (unsigned char *)out_split, split_pow2(split_exp), split_pow2(acc_exp), sat);
}

// ---- input projection: Xa[nt][mt] = Wp[mt] . act[nt] + b ----------------------------------
__global__ void __launch_bounds__(256)
k_inproj(const float *__restrict__ in, float *__restrict__ xa, const v4f *__restrict__ Wp,
         const float *__restrict__ bias, int ntile, int Mt, int K16) {
    constexpr int TM = 4, TN = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int nMblk = (Mt + 2 * TM - 1) / (2 * TM);
    const int nNblk = (ntile + 2 * TN - 1) / (2 * TN);
    const int L = xcd_remap(blockIdx.x, nMblk * nNblk);
    const int mblk = L % nMblk, nblk = L / nMblk;
    const int mt0 = (mblk * 2 + wm) * TM, nt0 = (nblk * 2 + wn) * TN;
    const int kq = lane >> 4;

    v4f acc[TM][TN];
    const v4f *ap[TM];
    const float *bp[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = min(mt0 + i, Mt - 1);
        ap[i] = Wp + (size_t)mt * K16 * 64 + lane;
        const v4f bv = *(const v4f *)(bias + mt * 16 + kq * 4);
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = bv;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int nt = min(nt0 + j, ntile - 1);
        bp[j] = in + (size_t)nt * K16 * 256 + lane * 4;
    }
    mma_tiles<TM, TN, true>(ap, bp, 256, K16, acc);
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = mt0 + i;
        if (mt >= Mt) continue;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nt = nt0 + j;
            if (nt >= ntile) continue;
            *(v4f *)(xa + ((size_t)nt * Mt + mt) * 256 + lane * 4) = acc[i][j];
        }
    }
}

void launch_inproj(hipStream_t s, const float *in, float *xa, const float4 *Wp, const float *bias,
                   int ntile, int M, int K16) {
    const int Mt = M / 16;
    const int nMblk = (Mt + 7) / 8, nNblk = (ntile + 7) / 8;
    hipLaunchKernelGGL(k_inproj, dim3(nMblk * nNblk), dim3(256), 0, s, in, xa, (const v4f *)Wp, bias, ntile, Mt, K16);
}


// ---- recurrent steps, launch-per-step path --------------------------------------------------
// Ut = H/4 unit tiles, K16 = H/16.  One wave = one unit tile (4 hidden units x 4 gate rows) x one
// read tile.  After the MFMA chain lane l = (q = l>>4, r = l&15) holds the four gate
// pre-activations of hidden unit 4*ut+q for read r: the gate math is lane-local and the cell
// state never leaves its lane's slot.
__global__ void __launch_bounds__(256)
k_lstm_step(const v4f *__restrict__ sWp, const v4f *__restrict__ xa_t, const float *__restrict__ h_prev,
            float *__restrict__ h_out, float *__restrict__ cstate, int Ut, int K16, int first, int t, const int *__restrict__ tbs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = blockIdx.x * 4 + wave, rt = blockIdx.y;
    if (ut >= Ut) return;
    v4f acc = xa_t[((size_t)rt * Ut + ut) * 64 + lane];
    float c = 0.0f;
    if (!first) {
        const v4f *a = sWp + (size_t)ut * K16 * 64 + lane;
        const v4f *b = (const v4f *)h_prev + (size_t)rt * K16 * 64 + lane;
        v4f acc2 = { 0.f, 0.f, 0.f, 0.f };   // two chains hide the 40-cycle dependent MFMA latency
        int k = 0;
        for (; k + 2 <= K16; k += 2) {
            acc = mfma4(a[(size_t)k * 64], b[(size_t)k * 64], acc);
            acc2 = mfma4(a[(size_t)(k + 1) * 64], b[(size_t)(k + 1) * 64], acc2);
        }
        if (k < K16) acc = mfma4(a[(size_t)k * 64], b[(size_t)k * 64], acc);
        acc = acc + acc2;
        c = cstate[((size_t)rt * Ut + ut) * 64 + lane];
    }
    // layers.c:1014-1025, gate rows i,f,g,o
    const float forget = logistic_ref(acc.y) * c;
    const float update = logistic_ref(acc.x) * tanh_ref(acc.z);
    c = forget + update;
    float h = logistic_ref(acc.w) * tanh_ref(c);
    const int q = lane >> 4, rl = lane & 15;
    if (tbs && t >= tbs[rt * 16 + rl]) { h = 0.0f; c = 0.0f; }        // beyond this read's end (ragged batch)
    cstate[((size_t)rt * Ut + ut) * 64 + lane] = c;
    h_out[((size_t)rt * Ut + ut) * 64 + rl * 4 + q] = h;
}

void launch_lstm_step(hipStream_t s, const float4 *sWp, const float *xa_t, const float *h_prev, float *h_out,
                      float *cstate, int B16, int H, int first, int t, const int *tbs) {
    const int Ut = H / 4, K16 = H / 16;
    hipLaunchKernelGGL(k_lstm_step, dim3((Ut + 3) / 4, B16), dim3(256), 0, s, (const v4f *)sWp, (const v4f *)xa_t,
                       h_prev, h_out, cstate, Ut, K16, first, t, tbs);
}

// grumod_step, layers.c:664-715.  Gate rows per unit: z, r, candidate, (unused).  The recurrent
// weights of row 3 are zero; Xa row 2 (candidate input) is kept out of the accumulator because
// the reference zeroes that chunk before the GEMV (:691) and adds x afterwards (:705).
__global__ void __launch_bounds__(256)
k_gru_step(const v4f *__restrict__ sWp, const v4f *__restrict__ xa_t, const float *__restrict__ h_prev,
           float *__restrict__ h_out, int Ut, int K16, int first, int t, const int *__restrict__ tbs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = blockIdx.x * 4 + wave, rt = blockIdx.y;
    if (ut >= Ut) return;
    const int q = lane >> 4, rl = lane & 15;
    const v4f x = xa_t[((size_t)rt * Ut + ut) * 64 + lane];
    v4f acc = { x.x, x.y, 0.f, 0.f };
    float hp = 0.0f;
    if (!first) {
        const v4f *a = sWp + (size_t)ut * K16 * 64 + lane;
        const v4f *b = (const v4f *)h_prev + (size_t)rt * K16 * 64 + lane;
        v4f acc2 = { 0.f, 0.f, 0.f, 0.f };
        int k = 0;
        for (; k + 2 <= K16; k += 2) {
            acc = mfma4(a[(size_t)k * 64], b[(size_t)k * 64], acc);
            acc2 = mfma4(a[(size_t)(k + 1) * 64], b[(size_t)(k + 1) * 64], acc2);
        }
        if (k < K16) acc = mfma4(a[(size_t)k * 64], b[(size_t)k * 64], acc);
        acc = acc + acc2;
        hp = h_prev[((size_t)rt * Ut + ut) * 64 + rl * 4 + q];
    }
    const float z = logistic_ref(acc.x);
    const float r = logistic_ref(acc.y);
    float hbar = r * acc.z + x.z;
    hbar = tanh_ref(hbar);
    float h = z * hp + (1.0f - z) * hbar;
    if (tbs && t >= tbs[rt * 16 + rl]) h = 0.0f;                        // beyond this read's end (ragged batch)
    h_out[((size_t)rt * Ut + ut) * 64 + rl * 4 + q] = h;
}

void launch_gru_step(hipStream_t s, const float4 *sWp, const float *xa_t, const float *h_prev, float *h_out,
                     int B16, int H, int first, int t, const int *tbs) {
    const int Ut = H / 4, K16 = H / 16;
    hipLaunchKernelGGL(k_gru_step, dim3((Ut + 3) / 4, B16), dim3(256), 0, s, (const v4f *)sWp, (const v4f *)xa_t,
                       h_prev, h_out, Ut, K16, first, t, tbs);
}

// ---- CRF head: trans[r][blk][p] = tanh(W^T h + b) / (temperature/5) --------------------------
// layers.c:1084-1087 (+ shift_scale_matrix_inplace flappie_matrix.c:625-633: a true division).
// TM = row tiles of 16 outputs a wave carries: 3 for the 40 scores of the 4-base models, 4 for 60 (5 bases) and the run-length head
template <int TM>
__global__ void __launch_bounds__(256)
k_head(const float *__restrict__ in, float *__restrict__ trans, const v4f *__restrict__ Wp,
       const float *__restrict__ bias, int Tb, int B16, int nread, int P, int Ps, int Mt, int K16, float scale, int raw) {
    constexpr int TN = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntile = Tb * B16;
    const int nt0 = (blockIdx.x * 4 + wave) * TN;
    if (nt0 >= ntile) return;
    const int q = lane >> 4, rl = lane & 15;
    v4f acc[TM][TN];
    const v4f *ap[TM];
    const float *bp[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = min(i, Mt - 1);
        ap[i] = Wp + (size_t)mt * K16 * 64 + lane;
        const v4f bv = *(const v4f *)(bias + mt * 16 + q * 4);
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = bv;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int nt = min(nt0 + j, ntile - 1);
        bp[j] = in + (size_t)nt * K16 * 256 + lane * 4;
    }
    mma_tiles<TM, TN, true>(ap, bp, 256, K16, acc);
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int p = i * 16 + q * 4;
        if (i >= Mt || p >= P) continue;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nt = nt0 + j;
            if (nt >= ntile) continue;
            const int blk = nt / B16, read = (nt % B16) * 16 + rl;
            if (read >= nread) continue;
            v4f v = acc[i][j];
            float *o = trans + ((size_t)read * Tb + blk) * Ps + p;
            const float vv[4] = { v.x, v.y, v.z, v.w };
            float rr[4];
            if (raw) { rr[0] = vv[0]; rr[1] = vv[1]; rr[2] = vv[2]; rr[3] = vv[3]; }
            else {
                const ffv4 t = apply_act4((ffv4){ vv[0], vv[1], vv[2], vv[3] }, 2);      // tanh_ref's bits, four at a time through the lean logistic (ffhip_math.hpp)
                rr[0] = (t.x - 0.0f) / scale; rr[1] = (t.y - 0.0f) / scale; rr[2] = (t.z - 0.0f) / scale; rr[3] = (t.w - 0.0f) / scale;
            }
            if (p + 3 < P && (Ps & 3) == 0) *(float4 *)o = make_float4(rr[0], rr[1], rr[2], rr[3]);      // one 16-byte store a lane (P and Ps are multiples of 4 for every model: 40, 60)
            else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (p + e < P) o[e] = rr[e];
            }
        }
    }
}

// ---- the same head on the last layer's SPLIT output (round 4) ------------------------------------------------------------------
// trans = tanh(W^T h + b) / (temperature / 5) with h read as the recurrent layer kernels leave it -- two fp16 slices of h * 2^12 in the
// B-operand order of v_mfma_f32_16x16x32_f16, 1 KiB per (K chunk, slice) -- and W as fp16 slices of W * 2^sw: three products per chunk
// (w1 h0, w0 h1, w0 h0: fp32-grade, ffhip_split.hpp), accumulators in the scaled space 2^S (bias pre-multiplied, result multiplied by
// 2^-S: powers of two, no rounding).  The last layer then writes no fp32 copy of h.  A wave owns TM row tiles x 4 (block, read tile)
// columns; operands of chunk c + 1 are loaded under the MFMAs of chunk c.  The epilogue is k_head's, value for value.
template <int TM>
__global__ void __launch_bounds__(256)
k_head_split(const unsigned char *__restrict__ in, float *__restrict__ trans, const v4u_t *__restrict__ Wp, const float *__restrict__ bias,
             int Tb, int B16, int nread, int P, int Ps, int Mt, int Hc, float scale, float acc_scale, int raw, double *__restrict__ E, int Pd) {
    FFHIP_DECODE_PRIO_SET();
    constexpr int TN = 4, NSL = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntile = Tb * B16;
    const int nt0 = (blockIdx.x * 4 + wave) * TN;
    if (nt0 >= ntile) return;
    const int q = lane >> 4, rl = lane & 15;
    const size_t tileB = (size_t)Hc * NSL * 1024;
    v4f acc[TM][TN];
    const v4u_t *ap[TM];
    const unsigned char *bp[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = min(i, Mt - 1);
        ap[i] = Wp + (size_t)mt * Hc * NSL * 64 + lane;
        const v4f bv = *(const v4f *)(bias + mt * 16 + q * 4) * acc_scale;
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = bv;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) bp[j] = in + (size_t)min(nt0 + j, ntile - 1) * tileB + (size_t)lane * 16;
    v4u_t A0[TM][NSL], B0[TN][NSL], A1[TM][NSL], B1[TN][NSL];
    auto load = [&](v4u_t (&A)[TM][NSL], v4u_t (&B)[TN][NSL], int c) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int sl = 0; sl < NSL; sl++) A[i][sl] = ap[i][(size_t)(c * NSL + sl) * 64];
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int sl = 0; sl < NSL; sl++) B[j][sl] = *(const v4u_t *)(bp[j] + (size_t)(c * NSL + sl) * 1024);
    };
    auto mma = [&](v4u_t (&A)[TM][NSL], v4u_t (&B)[TN][NSL]) {
        constexpr int WS[3] = { 1, 0, 0 }, XS[3] = { 0, 1, 0 };      // smallest terms first
#pragma unroll
        for (int term = 0; term < 3; term++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h_t, A[i][WS[term]]), __builtin_bit_cast(v8h_t, B[j][XS[term]]), acc[i][j], 0, 0, 0);
    };
    load(A0, B0, 0);
    for (int c = 0; c < Hc; c += 2) {
        if (c + 1 < Hc) load(A1, B1, c + 1);
        mma(A0, B0);
        if (c + 1 < Hc) {
            if (c + 2 < Hc) load(A0, B0, c + 2);
            mma(A1, B1);
        }
    }
    const float inv_scale = 1.0f / acc_scale;
    if (E) {
        // ---- the same epilogue, column by column, and behind each column E = exp(S - max S) of the block for the fp64 chains of ffhip_decode.hip (round 5:
        // k_crf_exp's pass over the scores -- one launch on the path between two pairs' layer launches, 26 MB read again -- is gone).  A block's P scores sit in
        // the four quarter-waves of one read's lanes (rows 16 i + 4 q + e): the maximum is taken inside the lane, then across the quarters; every value is what
        // k_crf_exp computes from the stored float, operation for operation (fmaxf over the same set, exp of the same double difference).
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nt = nt0 + j;
            const bool live_t = nt < ntile;
            const int ntc = live_t ? nt : ntile - 1;
            const int blk = ntc / B16, read = (ntc % B16) * 16 + rl;
            const bool live = live_t && read < nread;
            float rr[TM][4];
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int p = i * 16 + q * 4;
                const v4f v = acc[i][j] * inv_scale;
                const ffv4 t = apply_act4((ffv4){ v.x, v.y, v.z, v.w }, 2);
                rr[i][0] = (t.x - 0.0f) / scale; rr[i][1] = (t.y - 0.0f) / scale; rr[i][2] = (t.z - 0.0f) / scale; rr[i][3] = (t.w - 0.0f) / scale;
                if (i < Mt && p < P) {
                    float *o = trans + ((size_t)read * Tb + blk) * Ps + p;
                    if (live) {
                        if (p + 3 < P && (Ps & 3) == 0) *(float4 *)o = make_float4(rr[i][0], rr[i][1], rr[i][2], rr[i][3]);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (p + e < P) o[e] = rr[i][e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (p + e < P) m = fmaxf(m, rr[i][e]);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            double *Eo = E + ((size_t)read * Tb + blk) * Pd;
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int p = i * 16 + q * 4;
                if (!live || p >= Pd) continue;
                double ev[4];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    ev[e] = (i < Mt && p + e < P) ? exp((double)rr[i][e] - (double)m) : ((p + e == P) ? (double)m : 0.0);      // [P]: the block's maximum; behind it zeros
                *(double2 *)(Eo + p) = make_double2(ev[0], ev[1]);
                *(double2 *)(Eo + p + 2) = make_double2(ev[2], ev[3]);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int p = i * 16 + q * 4;
        if (i >= Mt || p >= P) continue;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nt = nt0 + j;
            if (nt >= ntile) continue;
            const int blk = nt / B16, read = (nt % B16) * 16 + rl;
            if (read >= nread) continue;
            const v4f v = acc[i][j] * inv_scale;
            float *o = trans + ((size_t)read * Tb + blk) * Ps + p;
            const float vv[4] = { v.x, v.y, v.z, v.w };
            float rr[4];
            if (raw) { rr[0] = vv[0]; rr[1] = vv[1]; rr[2] = vv[2]; rr[3] = vv[3]; }
            else {
                const ffv4 t = apply_act4((ffv4){ vv[0], vv[1], vv[2], vv[3] }, 2);
                rr[0] = (t.x - 0.0f) / scale; rr[1] = (t.y - 0.0f) / scale; rr[2] = (t.z - 0.0f) / scale; rr[3] = (t.w - 0.0f) / scale;
            }
            if (p + 3 < P && (Ps & 3) == 0) *(float4 *)o = make_float4(rr[0], rr[1], rr[2], rr[3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (p + e < P) o[e] = rr[e];
            }
        }
    }
}

void launch_head_split(hipStream_t s, const void *in_split, float *trans, const void *Wsplit, const float *bias,
                       int Tb, int B16, int nread, int P, int Ps, int Hc, float scale, int acc_exp, int raw, double *E) {
    const int Mt = (P + 15) / 16;
    const int ntile = Tb * B16;
    const int Pd = crf_exp_stride(P);
    if (E && (raw || Pd > 16 * (Mt <= 3 ? 3 : 4))) E = nullptr;      // (the caller asks head_split_writes_E first)
    if (Mt <= 3)
        hipLaunchKernelGGL(k_head_split<3>, dim3((ntile + 15) / 16), dim3(256), 0, s, (const unsigned char *)in_split, trans, (const v4u_t *)Wsplit, bias, Tb, B16,
                           nread, P, Ps, Mt, Hc, scale, split_pow2(acc_exp), raw, E, Pd);
    else
        hipLaunchKernelGGL(k_head_split<4>, dim3((ntile + 15) / 16), dim3(256), 0, s, (const unsigned char *)in_split, trans, (const v4u_t *)Wsplit, bias, Tb, B16,
                           nread, P, Ps, Mt, Hc, scale, split_pow2(acc_exp), raw, E, Pd);
}
// the head's epilogue can leave E = exp(S - max S) for the chains of ffhip_decode.hip: a block's row of crf_exp_stride(P) doubles must fit the head's row tiles
bool head_split_writes_E(int P) { const int Mt = (P + 15) / 16; return !dbg("no_head_exp") && crf_exp_stride(P) <= 16 * (Mt <= 3 ? 3 : 4) && Mt <= 4; }

void launch_head(hipStream_t s, const float *in, float *trans, const float4 *Wp, const float *bias,
                 int Tb, int B16, int nread, int P, int Ps, int K16, float scale, int raw) {
    const int Mt = (P + 15) / 16;
    const int ntile = Tb * B16;
    if (Mt <= 3)
        hipLaunchKernelGGL(k_head<3>, dim3((ntile + 15) / 16), dim3(256), 0, s, in, trans, (const v4f *)Wp, bias, Tb, B16,
                           nread, P, Ps, Mt, K16, scale, raw);
    else
        hipLaunchKernelGGL(k_head<4>, dim3((ntile + 15) / 16), dim3(256), 0, s, in, trans, (const v4f *)Wp, bias, Tb, B16,
                           nread, P, Ps, Mt, K16, scale, raw);
}

// ---- CRF partition function + global normalisation -------------------------------------------
// layers.c:1035-1096.  One wavefront per read walks the blocks; the fp64 forward vector lives in
// lanes 0..nstate-1.  Each of the P transition scores of a block is combined with its source
// state's value in parallel, the per-destination logsumexp is evaluated as max + log(sum exp),
// i.e. the same quantity as the reference's pairwise chain up to fp64 rounding (the result is
// rounded to fp32 after the division by the block count, layers.c:1089).
__global__ void __launch_bounds__(64)
k_crf_norm(const float *__restrict__ trans, int TbS, int nbase, int P, int Ps, double *__restrict__ logz_out, const int *__restrict__ tbs) {
    __shared__ double term[64];
    __shared__ double smax[kMaxState];
    const int lane = threadIdx.x;
    const int ns = 2 * nbase, off = nbase * ns;
    const float *S = trans + (size_t)blockIdx.x * TbS * Ps;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    // destination state of transition entry `lane`, and its source state
    const int src = lane % ns;
    int dst;
    if (lane < off) dst = lane / ns;
    else { const int idx = lane - off; dst = (idx < nbase) ? idx + nbase : idx; }
    double prev = 0.0;
    float s_next = (lane < P) ? S[lane] : 0.0f;
    for (int blk = 0; blk < Tb; blk++) {
        const float s = s_next;
        if (blk + 1 < Tb) s_next = (lane < P) ? S[(size_t)(blk + 1) * Ps + lane] : 0.0f;
        const double pf = __shfl(prev, src);
        term[lane] = (lane < P) ? pf + (double)s : -INFINITY;
        __syncthreads();
        if (lane < ns) {
            double m;
            if (lane < nbase) {
                m = term[lane * ns];
                for (int f = 1; f < ns; f++) m = fmax(m, term[lane * ns + f]);
            } else {
                m = fmax(term[off + lane], term[off + lane - nbase]);
            }
            smax[lane] = m;
        }
        __syncthreads();
        const double e = (lane < P) ? exp(term[lane] - smax[dst]) : 0.0;
        __syncthreads();
        term[lane] = e;
        __syncthreads();
        if (lane < ns) {
            double sum;
            if (lane < nbase) {
                sum = term[lane * ns];
                for (int f = 1; f < ns; f++) sum += term[lane * ns + f];
            } else {
                sum = term[off + lane] + term[off + lane - nbase];
            }
            prev = smax[lane] + log(sum);
        }
        __syncthreads();
    }
    // logZ = logsumexp over final states (pairwise, layers.c:1071-1074)
    double logZ = __shfl(prev, 0);
    for (int st = 1; st < ns; st++) {
        const double v = __shfl(prev, st);
        logZ = fmax(logZ, v) + log1p(exp(-fabs(logZ - v)));
    }
    if (lane == 0) logz_out[blockIdx.x] = logZ;        // crf_manystay_partition_function's own result
}

// S[r][blk][p] -= (float)(logZ[r] / Tb) for p < P (layers.c:1089-1096), all reads and blocks in parallel
__global__ void __launch_bounds__(256)
k_crf_sub(float *__restrict__ trans, const double *__restrict__ logz, int TbS, int P, int Ps, size_t n /*nread*TbS*Ps*/,
          const int *__restrict__ tbs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if ((int)(i % Ps) >= P) return;
    const size_t r = i / ((size_t)TbS * Ps);
    const int Tb = tbs ? tbs[r] : TbS;
    if ((int)((i / Ps) % TbS) >= Tb) return;
    trans[i] -= (float)(logz[r] / (double)Tb);
}

static void launch_crf_sub(hipStream_t s, float *trans, const double *logz, int nread, int Tb, int P, int Ps, const int *tbs) {
    const size_t n = (size_t)nread * Tb * Ps;
    hipLaunchKernelGGL(k_crf_sub, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, trans, logz, Tb, P, Ps, n, tbs);
}

void launch_crf_norm(hipStream_t s, float *trans, int nread, int Tb, int nbase, int Ps, double *logz, int subtract, const int *tbs) {
    const int P = 2 * nbase * (nbase + 1);
    hipLaunchKernelGGL(k_crf_norm, dim3(nread), dim3(64), 0, s, trans, Tb, nbase, P, Ps, logz, tbs);
    if (subtract) launch_crf_sub(s, trans, logz, nread, Tb, P, Ps, tbs);
}

// ---- lane exchanges inside a row of 16 lanes as DPP moves (a few cycles) instead of ds_bpermute (an LDS-crossbar
// round trip of ~100 cycles): these sit on the Tb-step dependent chains of the decode kernels.
//   quad_perm [1,0,3,2] = xor 1, [2,3,0,1] = xor 2; row_shl:4 / row_shr:4 under bank masks = xor 4; row_ror:8 = xor 8
template <int CTRL, int BANK>
__device__ __forceinline__ int dpp_i(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, BANK, false); }
__device__ __forceinline__ int xor1_i(int x) { return dpp_i<0xB1, 0xf>(x, x); }
__device__ __forceinline__ int xor2_i(int x) { return dpp_i<0x4E, 0xf>(x, x); }
__device__ __forceinline__ int xor4_i(int x) { return dpp_i<0x114, 0xA>(dpp_i<0x104, 0x5>(x, x), x); }
__device__ __forceinline__ int xor8_i(int x) { return dpp_i<0x128, 0xf>(x, x); }
__device__ __forceinline__ float xor1_f(float x) { return __int_as_float(xor1_i(__float_as_int(x))); }
__device__ __forceinline__ float xor2_f(float x) { return __int_as_float(xor2_i(__float_as_int(x))); }
__device__ __forceinline__ float xor4_f(float x) { return __int_as_float(xor4_i(__float_as_int(x))); }
__device__ __forceinline__ float xor8_f(float x) { return __int_as_float(xor8_i(__float_as_int(x))); }

__device__ __forceinline__ double xor1_d(double x) { return __hiloint2double(xor1_i(__double2hiint(x)), xor1_i(__double2loint(x))); }
__device__ __forceinline__ double xor2_d(double x) { return __hiloint2double(xor2_i(__double2hiint(x)), xor2_i(__double2loint(x))); }
__device__ __forceinline__ double xor4_d(double x) { return __hiloint2double(xor4_i(__double2hiint(x)), xor4_i(__double2loint(x))); }
// lane holding the new value of state s after the grouped reductions of the 8-state kernels: flip state s < 4 in
// lanes 8s..8s+7, flop state s in lanes 32+s-4 and 32+s
__device__ __forceinline__ int ff8_src_lane(int s) { return s < 4 ? 8 * s : 32 + s; }
// the same for the 10-state kernels: flip state s < 5 in lanes 8s..8s+7, flop state s in lane 40 + s - 5
__device__ __forceinline__ int ff10_src_lane(int s) { return s < 5 ? 8 * s : 40 + (s - 5); }

// ---- CRF partition function, linear-space form (the pipeline's default) -----------------------------
// The log-space recursion above spends an fp64 exp and log per state per block ON the dependent chain
// (Tb steps x ~3000 cycles).  The same quantity factorises: with m_t = max_p S[t][p] and
// E_t[p] = exp(S[t][p] - m_t) (independent of the chain -> computed for all blocks in parallel),
//     alpha_t[to] = sum_from E_t[to, from] * alpha_{t-1}[from],     alpha_{-1} = 1,
//     logZ = log(sum_s alpha_{Tb-1}[s]) + ln2 * K + sum_t m_t,
// where K collects the exact power-of-two rescalings applied every R blocks.  All in fp64; the chain is an
// 8x8 sparse mat-vec per block (~250 cycles).  The result differs from the pairwise-logsumexp evaluation by
// fp64 rounding only (<= 1e-12 relative, checked against the oracle), and is rounded to fp32 after the
// division by the block count exactly as layers.c:1089 does.
// One workgroup stages 64 blocks of scores in LDS (coalesced), finds each block's maximum once, then every thread
// produces exp(S - max) for its share of the 64 x Pd outputs.
constexpr int kExpBlocks = 64;
__global__ void __launch_bounds__(256)
k_crf_exp(const float *__restrict__ trans, double *__restrict__ E, size_t nblk /*nread*TbS*/, int P, int Ps, int Pd, int TbS,
          const int *__restrict__ tbs, int *__restrict__ wide, float limit, int row_off) {
    FFHIP_DECODE_PRIO_SET();
    __shared__ float sc[kExpBlocks * 64];
    __shared__ float mx[kExpBlocks];
    const size_t b0 = (size_t)blockIdx.x * kExpBlocks;
    const int nb = (int)min((size_t)kExpBlocks, nblk - b0);
    const size_t last = nblk * Ps - 1;
    for (int i = threadIdx.x; i < nb * Ps; i += 256) sc[i] = trans[min(b0 * Ps + i + row_off, last)];      // (row_off: the entries start behind other rows of the block)
    __syncthreads();
    if (threadIdx.x < nb) {
        const float *S = sc + threadIdx.x * Ps;
        float m = S[0], lo = S[0];
        bool finite = true;
        for (int q = 1; q < P; q++) { m = fmaxf(m, S[q]); lo = fminf(lo, S[q]); }
        for (int q = 0; q < P; q++) finite = finite && (fabsf(S[q]) < INFINITY);
        mx[threadIdx.x] = m;
        // a block whose scores span more than `limit` (or are not finite): its read is not for the scaled linear-space recursions of
        // ffhip_decode.hip (their scaling lags a pair of blocks behind the values)
        const size_t blk = b0 + threadIdx.x;
        if (wide && !(finite && m - lo <= limit) && !(tbs && (int)(blk % TbS) >= tbs[blk / TbS])) atomicOr(&wide[blk / TbS], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * Pd; i += 256) {
        const int k = i / Pd, p = i % Pd;
        const size_t blk = b0 + k;
        if (tbs && (int)(blk % TbS) >= tbs[blk / TbS]) continue;
        if (p > P) { E[blk * Pd + p] = 0.0; continue; }          // the row's padding reads as "no such transition" (ffhip_decode.hip)
        E[blk * Pd + p] = (p == P) ? (double)mx[k] : exp((double)sc[k * Ps + p] - (double)mx[k]);
    }
}

// One wave per read.  E is streamed through LDS in chunks of kCrfChunk blocks: the loads of chunk c+1 are
// issued (coalesced, all 64 lanes) before chunk c is walked and land in LDS after it, so the dependent
// chain never waits on HBM.
constexpr int kCrfChunk = 32;
template <int NS>
__global__ void __launch_bounds__(64)
k_crf_chain(const double *__restrict__ E, int TbS, int P, int Pd, int R, double *__restrict__ logz_out, const int *__restrict__ tbs) {
    constexpr int nbase = NS / 2, off = nbase * NS;
    constexpr int kMaxPd = 72;                                  // crf_exp_stride(64)
    __shared__ double ebuf[2][kCrfChunk * kMaxPd];
    __shared__ double al[2][NS];
    const int lane = threadIdx.x;
    const bool active = lane < NS, flip = lane < nbase;
    const double *Er = E + (size_t)blockIdx.x * TbS * Pd;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    const int per_chunk = kCrfChunk * Pd;                       // doubles per chunk (<= 2304)
    constexpr int kStage = (kCrfChunk * kMaxPd + 63) / 64;      // 36 doubles per lane at most
    double stage[kStage];
    const int nchunk = (Tb + kCrfChunk - 1) / kCrfChunk;
    auto fetch = [&](int c) {
        const size_t base = (size_t)c * per_chunk, lim = (size_t)Tb * Pd;
#pragma unroll
        for (int k = 0; k < kStage; k++) {
            const int j = k * 64 + lane;
            stage[k] = (j < per_chunk && base + j < lim) ? Er[base + j] : 0.0;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < kStage; k++) {
            const int j = k * 64 + lane;
            if (j < per_chunk) ebuf[buf][j] = stage[k];
        }
    };
    if (active) al[0][lane] = 1.0;
    fetch(0);
    commit(0);
    __syncthreads();
    double msum = 0.0;
    long long K = 0;
    int cur = 0, since = 0;
    for (int c = 0; c < nchunk; c++) {
        if (c + 1 < nchunk) fetch(c + 1);
        const double *eb = ebuf[c & 1];
        const int t0 = c * kCrfChunk, t1 = min(Tb, t0 + kCrfChunk);
        for (int t = t0; t < t1; t++) {
            const double *row = eb + (t - t0) * Pd;
            double acc = 0.0;
            if (flip) {
                // independent products, pairwise tree: 4 dependent fp64 operations instead of 16
                double pr[NS];
#pragma unroll
                for (int f = 0; f < NS; f++) pr[f] = row[lane * NS + f] * al[cur][f];
#pragma unroll
                for (int w = 1; w < NS; w <<= 1)
#pragma unroll
                    for (int f = 0; f + w < NS; f += 2 * w) pr[f] = pr[f] + pr[f + w];
                acc = pr[0];
            } else if (active) {
                acc = row[off + lane - nbase] * al[cur][lane - nbase] + row[off + lane] * al[cur][lane];
            }
            msum = msum + row[P];
            if (++since == R || t + 1 == Tb) {
                since = 0;
                double mx = active ? acc : 0.0;
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) mx = fmax(mx, __shfl_xor(mx, d));
                const int ex = __builtin_amdgcn_readfirstlane((mx > 0.0) ? ilogb(mx) : 0);      // lane 0's 16-lane group holds the states
                acc = ldexp(acc, -ex);
                K += ex;
            }
            if (active) al[cur ^ 1][lane] = acc;
            // one wave: LDS executes its instructions in order, only the compiler must not reorder them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cur ^= 1;
        }
        if (c + 1 < nchunk) commit((c + 1) & 1);
        __syncthreads();
    }
    double total = 0.0;
    for (int st = 0; st < NS; st++) total = total + al[cur][st];
    const double logZ = log(total) + 0.693147180559945309417232121458 * (double)K + msum;
    if (lane == 0) logz_out[blockIdx.x] = logZ;
}

// nstate = 8 (ACGT): the 40 transitions of a block live one per lane; every lane carries alpha of its source state,
// the per-destination sums are DPP reductions inside groups of 8 lanes (flip) / pairs (flop), and one gather hands
// every lane the new alpha of its source.  No LDS on the chain except the staged E values.
__global__ void __launch_bounds__(64)
k_crf_chain8(const double *__restrict__ E, int TbS, int Pd, int R, double *__restrict__ logz_out, const int *__restrict__ tbs) {
    constexpr int P = 40, kMaxPd = 48;
    __shared__ double ebuf[2][kCrfChunk * kMaxPd];
    const int lane = threadIdx.x;
    const bool valid = lane < P;
    const int gather = ff8_src_lane(lane & 7);
    const double *Er = E + (size_t)blockIdx.x * TbS * Pd;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;
    if (Tb <= 0) return;                                 // an empty slot
    const int per_chunk = kCrfChunk * Pd;
    constexpr int kStage = (kCrfChunk * kMaxPd + 63) / 64;      // 24 doubles per lane
    double stage[kStage];
    const int nchunk = (Tb + kCrfChunk - 1) / kCrfChunk;
    auto fetch = [&](int c) {
        const size_t base = (size_t)c * per_chunk, lim = (size_t)Tb * Pd;
#pragma unroll
        for (int k = 0; k < kStage; k++) {
            const int j = k * 64 + lane;
            stage[k] = (j < per_chunk && base + j < lim) ? Er[base + j] : 0.0;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < kStage; k++) {
            const int j = k * 64 + lane;
            if (j < per_chunk) ebuf[buf][j] = stage[k];
        }
    };
    fetch(0);
    commit(0);
    __syncthreads();
    double a_src = 1.0, msum = 0.0;
    long long K = 0;
    int since = 0;
    for (int c = 0; c < nchunk; c++) {
        if (c + 1 < nchunk) fetch(c + 1);
        const double *eb = ebuf[c & 1];
        const int t0 = c * kCrfChunk, t1 = min(Tb, t0 + kCrfChunk);
        double e_next = valid ? eb[lane] : 0.0, m_next = eb[P];
        for (int t = t0; t < t1; t++) {
            const double e = e_next, mt = m_next;
            if (t + 1 < t1) { const double *row = eb + (t + 1 - t0) * Pd; e_next = valid ? row[lane] : 0.0; m_next = row[P]; }
            const double term = e * a_src;
            const double pair = term + xor4_d(term);                  // flop destinations: entries b - nbase and b
            double grp = pair + xor1_d(pair);
            grp = grp + xor2_d(grp);                                  // flip destinations: all 8 sources
            const double val = (lane < 32) ? grp : pair;
            a_src = __shfl(val, gather);
            msum = msum + mt;
            if (++since == R || t + 1 == Tb) {
                since = 0;
                double mx = fmax(a_src, xor4_d(a_src));               // lanes 0..7 hold alpha of states 0..7
                mx = fmax(mx, xor1_d(mx));
                mx = fmax(mx, xor2_d(mx));
                const int ex = __builtin_amdgcn_readfirstlane((mx > 0.0) ? ilogb(mx) : 0);
                a_src = ldexp(a_src, -ex);
                K += ex;
            }
        }
        if (c + 1 < nchunk) commit((c + 1) & 1);
        __syncthreads();
    }
    double total = a_src + xor4_d(a_src);
    total = total + xor1_d(total);
    total = total + xor2_d(total);
    const double logZ = log(total) + 0.693147180559945309417232121458 * (double)K + msum;
    if (lane == 0) logz_out[blockIdx.x] = logZ;
}

// nstate = 10 (ACGTZ): two entries per lane as in k_viterbi10 / the forward half of k_transpost10 -- flip destination g in
// lanes 8g..8g+7 (lane j: sources j and j+8), flop state 5+j in lane 40+j (stay, move); alpha of the two source states per lane.
__global__ void __launch_bounds__(64)
k_crf_chain10(const double *__restrict__ E, int TbS, int Pd, int R, double *__restrict__ logz_out, const int *__restrict__ tbs) {
    constexpr int P = 60, ns = 10, nbase = 5, off = 50, kMaxPd = 64;
    __shared__ double ebuf[2][kCrfChunk * kMaxPd];
    const int lane = threadIdx.x, g = lane >> 3, j = lane & 7;
    const bool flip = g < nbase, flop = (g == nbase && j < nbase);
    const bool valid0 = flip || flop, valid1 = (flip && j < 2) || flop;
    const int e0 = flip ? g * ns + j : (flop ? off + nbase + j : 0);
    const int e1 = flip ? (j < 2 ? g * ns + 8 + j : e0) : (flop ? off + j : 0);
    const int src0 = flip ? ff10_src_lane(j) : (flop ? ff10_src_lane(nbase + j) : 0);
    const int src1 = flip ? ff10_src_lane(j < 2 ? j + 8 : j) : (flop ? ff10_src_lane(j) : 0);
    const double *Er = E + (size_t)blockIdx.x * TbS * Pd;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;
    if (Tb <= 0) return;                                 // an empty slot
    const int per_chunk = kCrfChunk * Pd;
    constexpr int kStage = (kCrfChunk * kMaxPd + 63) / 64;      // 32 doubles per lane
    double stage[kStage];
    const int nchunk = (Tb + kCrfChunk - 1) / kCrfChunk;
    auto fetch = [&](int c) {
        const size_t base = (size_t)c * per_chunk, lim = (size_t)Tb * Pd;
#pragma unroll
        for (int k = 0; k < kStage; k++) {
            const int i = k * 64 + lane;
            stage[k] = (i < per_chunk && base + i < lim) ? Er[base + i] : 0.0;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < kStage; k++) {
            const int i = k * 64 + lane;
            if (i < per_chunk) ebuf[buf][i] = stage[k];
        }
    };
    fetch(0);
    commit(0);
    __syncthreads();
    double a0 = 1.0, a1 = 1.0, msum = 0.0;
    long long K = 0;
    int since = 0;
    for (int c = 0; c < nchunk; c++) {
        if (c + 1 < nchunk) fetch(c + 1);
        const double *eb = ebuf[c & 1];
        const int t0 = c * kCrfChunk, t1 = min(Tb, t0 + kCrfChunk);
        double x0 = valid0 ? eb[e0] : 0.0, x1 = valid1 ? eb[e1] : 0.0, m_next = eb[P];
        for (int t = t0; t < t1; t++) {
            const double f0 = x0, f1 = x1, mt = m_next;
            if (t + 1 < t1) { const double *row = eb + (t + 1 - t0) * Pd; x0 = valid0 ? row[e0] : 0.0; x1 = valid1 ? row[e1] : 0.0; m_next = row[P]; }
            double val = f0 * a0 + f1 * a1;                       // flop states: stay + move
            if (flip) { val = val + xor4_d(val); val = val + xor1_d(val); val = val + xor2_d(val); }
            a0 = __shfl(val, src0);
            a1 = __shfl(val, src1);
            msum = msum + mt;
            if (++since == R || t + 1 == Tb) {
                since = 0;
                double mx = (lane < 2) ? fmax(a0, a1) : a0;       // lanes 0..7: alpha of states 0..7 (a0) and 8, 9 (a1 of lanes 0, 1)
                mx = fmax(mx, xor4_d(mx));
                mx = fmax(mx, xor1_d(mx));
                mx = fmax(mx, xor2_d(mx));
                const int ex = __builtin_amdgcn_readfirstlane((mx > 0.0) ? ilogb(mx) : 0);
                a0 = ldexp(a0, -ex);
                a1 = ldexp(a1, -ex);
                K += ex;
            }
        }
        if (c + 1 < nchunk) commit((c + 1) & 1);
        __syncthreads();
    }
    double total = (lane < 2) ? a0 + a1 : a0;
    total = total + xor4_d(total);
    total = total + xor1_d(total);
    total = total + xor2_d(total);
    const double logZ = log(total) + 0.693147180559945309417232121458 * (double)K + msum;
    if (lane == 0) logz_out[blockIdx.x] = logZ;
}

void launch_crf_exp(hipStream_t s, const float *trans, double *E, int nread, int Tb, int nbase, int Ps, const int *tbs, int *wide, float limit, int row_off, int P_override) {
    const int P = P_override ? P_override : 2 * nbase * (nbase + 1), Pd = crf_exp_stride(P);
    const size_t nblk = (size_t)nread * Tb;
    hipLaunchKernelGGL(k_crf_exp, dim3((unsigned)((nblk + kExpBlocks - 1) / kExpBlocks)), dim3(256), 0, s, trans, E, nblk, P, Ps, Pd, Tb, tbs, wide, limit, row_off);
}

void launch_crf_norm_linear(hipStream_t s, float *trans, double *E, int nread, int Tb, int nbase, int Ps, int R,
                            double *logz, int subtract, const int *tbs) {
    const int P = 2 * nbase * (nbase + 1), Pd = crf_exp_stride(P);
    const size_t nblk = (size_t)nread * Tb;
    hipLaunchKernelGGL(k_crf_exp, dim3((unsigned)((nblk + kExpBlocks - 1) / kExpBlocks)), dim3(256), 0, s, trans, E, nblk, P, Ps, Pd, Tb, tbs, (int *)nullptr, 0.0f, 0);
    const int Rr = R < 1 ? 1 : R;
    if (nbase == 4 && !dbg("crf_generic")) {
        hipLaunchKernelGGL(k_crf_chain8, dim3(nread), dim3(64), 0, s, E, Tb, Pd, Rr, logz, tbs);
    } else if (nbase == 5 && Pd <= 64 && !dbg("crf_generic")) {
        hipLaunchKernelGGL(k_crf_chain10, dim3(nread), dim3(64), 0, s, E, Tb, Pd, Rr, logz, tbs);
    } else
    switch (2 * nbase) {
#define CHAIN_CASE(NS) case NS: hipLaunchKernelGGL(k_crf_chain<NS>, dim3(nread), dim3(64), 0, s, E, Tb, P, Pd, Rr, logz, tbs); break;
    CHAIN_CASE(2) CHAIN_CASE(4) CHAIN_CASE(6) CHAIN_CASE(8) CHAIN_CASE(10) CHAIN_CASE(12) CHAIN_CASE(14) CHAIN_CASE(16)
#undef CHAIN_CASE
    }
    if (subtract) launch_crf_sub(s, trans, logz, nread, Tb, P, Ps, tbs);
}

// ---- forward/backward transition posteriors ---------------------------------------------------
// decode.c:377-497.  One wavefront per read; lanes 0..nstate-1 carry the fwd / bwd vectors, lanes
// 0..P-1 carry the block's transition scores.  The logsumexpf chains keep the reference's order.
__global__ void __launch_bounds__(64)
k_transpost(const float *__restrict__ trans, float *__restrict__ post, float *__restrict__ fwdbuf,
            int TbS, int nbase, int P, int Ps, const int *__restrict__ tbs) {
    const int lane = threadIdx.x;
    const int ns = 2 * nbase, off = nbase * ns;
    const float *T = trans + (size_t)blockIdx.x * TbS * Ps;
    float *Pp = post + (size_t)blockIdx.x * TbS * Ps;
    float *F = fwdbuf + (size_t)blockIdx.x * (TbS + 1) * kMaxState;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    const bool is_state = lane < ns, is_flip = lane < nbase;

    // forwards (:396-423)
    float prev = 0.0f;
    if (is_state) F[lane] = 0.0f;
    for (int blk = 0; blk < Tb; blk++) {
        const float s = (lane < P) ? T[(size_t)blk * Ps + lane] : 0.0f;
        float acc;
        if (true) {
            // flip candidate chain (valid for lanes < nbase)
            const int base = is_flip ? lane * ns : 0;
            acc = __shfl(s, base) + __shfl(prev, 0);
            for (int f = 1; f < ns; f++) {
                const float sc = __shfl(s, base + f) + __shfl(prev, f);
                acc = logsumexpf_ref(acc, sc);
            }
            // flop: stay, then move from the flip state of the same base
            const int b2 = (is_state && !is_flip) ? lane : nbase;
            const float stay = __shfl(prev, b2) + __shfl(s, off + b2);
            const float move = __shfl(prev, b2 - nbase) + __shfl(s, off + b2 - nbase);
            const float flop = logsumexpf_ref(stay, move);
            if (!is_flip) acc = flop;
        }
        prev = acc;
        if (is_state) F[(size_t)(blk + 1) * kMaxState + lane] = acc;
    }

    // backwards (:434-484); `prev` is the backward vector of block blk
    // source state (st) and destination state (to) of transition entry `lane`
    const int st = lane % ns;
    int to;
    if (lane < off) to = lane / ns;
    else { const int idx = lane - off; to = (idx < nbase) ? idx + nbase : idx; }
    prev = 0.0f;
    for (int blk = Tb; blk > 0; blk--) {
        const float s = (lane < P) ? T[(size_t)(blk - 1) * Ps + lane] : 0.0f;
        const float f = is_state ? F[(size_t)(blk - 1) * kMaxState + lane] : 0.0f;
        // tpost = fwd[st] + bwd[to] + trans   (left to right, :451-461)
        const float tp = (__shfl(f, st) + __shfl(prev, to)) + s;
        if (lane < P) Pp[(size_t)(blk - 1) * Ps + lane] = tp;
        // update of the backward vector for source state `lane`
        const int me = is_state ? lane : 0;
        const int b2 = (me < nbase) ? me + nbase : me;               // flop state reached from `me`
        float curr = __shfl(prev, b2) + __shfl(s, off + me);          // :466-472
        for (int b1 = 0; b1 < nbase; b1++) {                          // :475-483
            const float sc = __shfl(s, b1 * ns + me) + __shfl(prev, b1);
            curr = logsumexpf_ref(curr, sc);
        }
        prev = curr;
    }
    __syncthreads();
    // log_row_normalise_inplace (flappie_matrix.c:450-467): sequential chain over the P rows of a
    // block; blocks are independent -> one block per lane.
    for (int blk = lane; blk < Tb; blk += 64) {
        float *x = Pp + (size_t)blk * Ps;
        float row_logsum = x[0];
        for (int r = 1; r < P; r++) row_logsum = logsumexpf_ref(row_logsum, x[r]);
        for (int r = 0; r < P; r++) x[r] -= row_logsum;
    }
}


// ---- fast path for nstate == 8 (ACGT models) ---------------------------------------------------
// Lane l < 40 owns transition entry l: flip entries l = 8*to + from (l < 32), flop entries
// l = 32 + idx (idx >= 4: stay in flop idx; idx < 4: move flip idx -> flop idx+4).  The source state
// of entry l is always l & 7, so the state vector is kept replicated by (l & 7) and needs no
// shuffle on the operand side.  Every per-destination logsumexp is evaluated as max + log(sum exp)
// with wave butterflies instead of the reference's sequential pairwise chain (decode.c:417-421,
// :478-482): same value up to fp32 rounding of the association, 7x shorter dependent chain.

// Workgroup of 4 waves per read: wave 0 runs the forward recursion, wave 1 the backward recursion at the same
// time (they are independent), then all 256 threads assemble and log-normalise one block each.  Same
// arithmetic per value as a single-wave walk; the dependent chain is halved.
__global__ void __launch_bounds__(256)
k_transpost8(const float *__restrict__ trans, float *__restrict__ post, float *__restrict__ fwdbuf, float *__restrict__ bwdbuf, int TbS,
             const int *__restrict__ tbs, const int *__restrict__ only) {
    constexpr int P = 40, Ps = 40, ns = 8;
    if (only && !only[blockIdx.x]) return;               // the reads k_crf_fb8 has left (score range too wide for its linear form)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *T = trans + (size_t)blockIdx.x * TbS * Ps;
    float *Pp = post + (size_t)blockIdx.x * TbS * Ps;
    float *F = fwdbuf + (size_t)blockIdx.x * (TbS + 1) * kMaxState;
    float *Bw = bwdbuf + (size_t)blockIdx.x * (TbS + 1) * kMaxState;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    const bool valid = lane < P, flip = lane < 32;
    const int st = lane & 7;
    const float NEG = -INFINITY;

    // The two recursions are chains of ~200 cycles of work per block; nothing on them may wait for memory.  Their
    // forward / backward vectors therefore go to an LDS stage and leave in rows of 64 blocks (a global store per block
    // shares the wave's memory counter with the loads: the compiler then waits for the store's acknowledgement, ~700
    // cycles, before it may use the next block's scores), and the scores of the next kDepth blocks are in flight while
    // kDepth blocks are processed (clamped addresses, never predicated loads: counted waits need branch-free queues).
    __shared__ float stage[2][64][kMaxState];
    constexpr int kDepth = 8;
    const int lane_c = valid ? lane : P - 1;
    auto flush = [&](int w, float *dst0, long long dstep, int cnt) {      // row r of the stage -> dst0 + r*dstep (kMaxState floats each)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            const float4 *src = (const float4 *)&stage[w][lane][0];
            float4 *dst = (float4 *)(dst0 + (long long)lane * dstep);
            dst[0] = src[0]; dst[1] = src[1];                              // ns = 8 states
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (wave == 0) {
        // forwards: pv = fwd[blk][lane & 7]
        float pv = 0.0f;
        if (lane < ns) F[lane] = 0.0f;
        float ring[kDepth];
        auto fetch = [&](int blk) { return T[(size_t)min(blk, Tb - 1) * Ps + lane_c]; };
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
                const int blk = b0 + k;
                if (blk >= Tb) break;
                const float term = valid ? cur[k] + pv : NEG;
                float m = fmaxf(term, xor4_f(term));
                if (flip) { m = fmaxf(m, xor1_f(m)); m = fmaxf(m, xor2_f(m)); }
                float e = valid ? expf(term - m) : 0.0f;
                e += xor4_f(e);
                if (flip) { e += xor1_f(e); e += xor2_f(e); }
                const float val = m + logf(e);
                pv = __shfl(val, ff8_src_lane(st));
                if (lane < ns) stage[0][blk & 63][lane] = pv;              // fwd[blk + 1]
                if ((blk & 63) == 63 || blk == Tb - 1) flush(0, F + (size_t)((blk & ~63) + 1) * kMaxState, kMaxState, (blk & 63) + 1);
            }
        }
    } else if (wave == 1) {
        // backwards, lanes grouped by SOURCE state: lane (g = lane >> 3, j = lane & 7) holds the exit of state g to flip state j (j < 4)
        // or into its flop state (j = 4: flip g -> flop g+4, flop g stays) -- the new bwd[g] is an in-group DPP reduction and each
        // lane then needs one value of the new vector (one ds_bpermute per block).
        const int g = lane >> 3, j = lane & 7;
        const bool act = j <= 4;
        const int fdst = g < 4 ? g + 4 : g;                              // flop destination of source g
        const int e = act ? (j < 4 ? j * ns + g : 32 + g) : 0;
        const int srcl = 8 * (j < 4 ? j : fdst);                         // where the new bwd of this exit's destination lives
        float pb_to = 0.0f, nb = 0.0f;                                   // bwd[destination of my exit], bwd[g] (replicated over the group)
        float ring[kDepth];
        auto fetch = [&](int blk) { return T[(size_t)max(blk, 0) * Ps + e]; };           // blk counts down
#pragma unroll
        for (int k = 0; k < kDepth; k++) ring[k] = fetch(Tb - 1 - k);
        for (int j0 = 0; j0 < Tb; j0 += kDepth) {                                          // jj = Tb - blk: 0, 1, ...
            float cur[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) cur[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kDepth; k++) ring[k] = fetch(Tb - 1 - (j0 + kDepth + k));
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                const int jj = j0 + k;
                if (jj >= Tb) break;
                if (j == 0) stage[1][jj & 63][g] = nb;                     // bwd[blk], blk = Tb - jj
                if ((jj & 63) == 63 || jj == Tb - 1) flush(1, Bw + (size_t)(Tb - (jj & ~63)) * kMaxState, -(long long)kMaxState, (jj & 63) + 1);
                const float t2 = act ? cur[k] + pb_to : NEG;
                float m = fmaxf(t2, xor4_f(t2));
                m = fmaxf(m, xor1_f(m));
                m = fmaxf(m, xor2_f(m));
                float x = act ? expf(t2 - m) : 0.0f;
                x += xor4_f(x);
                x += xor1_f(x);
                x += xor2_f(x);
                nb = m + logf(x);
                pb_to = __shfl(nb, srcl);
            }
        }
    }
    __syncthreads();
    // posterior of transition `r` of block blk = (fwd[blk][from] + bwd[blk+1][to]) + trans (decode.c:451-461), then the
    // per-block log-normalisation over the 40 entries (flappie_matrix.c:450-467); one block per thread
    for (int blk = threadIdx.x; blk < Tb; blk += 256) {
        const float *x = T + (size_t)blk * Ps;
        float *o = Pp + (size_t)blk * Ps;
        float f[ns], bb[ns];
#pragma unroll
        for (int k = 0; k < ns; k++) { f[k] = F[(size_t)blk * kMaxState + k]; bb[k] = Bw[(size_t)(blk + 1) * kMaxState + k]; }
        float v[P];
        float m = NEG;
#pragma unroll
        for (int r = 0; r < P; r++) {
            const int from = r & 7;
            const int to = (r < 32) ? (r >> 3) : ((r - 32 < 4) ? r - 32 + 4 : r - 32);
            v[r] = (f[from] + bb[to]) + x[r];
            m = fmaxf(m, v[r]);
        }
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < P; r++) sum += expf(v[r] - m);
        const float lse = m + logf(sum);
#pragma unroll
        for (int r = 0; r < P; r++) o[r] = v[r] - lse;
    }
}




// ---- any nstate: same max + log(sum exp) formulation with the segmented reductions through LDS ----
// Used for the 5-base (ACGTZ, nstate 10) models where the 8-lane butterflies above do not apply.
// State lanes (lane < nstate) own one state each; entry lanes (lane < P) own one transition score.
// Any nstate <= 16 (the 5mC model has 10): the same max + log-sum-exp scheme through LDS.  As in k_transpost8,
// wave 0 runs the forward recursion while wave 1 runs the backward one, then every thread assembles and
// log-normalises whole blocks.
__global__ void __launch_bounds__(256)
k_transpost_lds(const float *__restrict__ trans, float *__restrict__ post, float *__restrict__ fwdbuf, float *__restrict__ bwdbuf,
                int TbS, int nbase, int P, int Ps, const int *__restrict__ tbs) {
    __shared__ float term[2][64];
    __shared__ float texp[2][64];
    __shared__ float smax[2][kMaxState];
    __shared__ float svec[2][kMaxState];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ns = 2 * nbase, off = nbase * ns;
    const float *T = trans + (size_t)blockIdx.x * TbS * Ps;
    float *Pp = post + (size_t)blockIdx.x * TbS * Ps;
    float *F = fwdbuf + (size_t)blockIdx.x * (TbS + 1) * kMaxState;
    float *Bw = bwdbuf + (size_t)blockIdx.x * (TbS + 1) * kMaxState;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    const bool valid = lane < P, is_state = lane < ns, is_flip = lane < nbase;
    const int src = lane % ns;                  // source state of entry `lane` (off is a multiple of ns)
    int dst;                                    // destination state of entry `lane`
    if (lane < off) dst = lane / ns;
    else { const int idx = lane - off; dst = (idx < nbase) ? idx + nbase : idx; }
    if (!valid) dst = 0;
    // one wave: LDS executes its instructions in order; only the compiler has to be kept from reordering
#define WAVE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

    // As in k_transpost8: the forward / backward vectors leave through an LDS stage in rows of 64 blocks (no global store on
    // the per-block chain) and the scores of the next kDepth blocks are in flight (clamped, branch-free loads).
    __shared__ float stage[2][64][kMaxState];
    constexpr int kDepth = 8;
    const int lane_c = valid ? lane : P - 1;
    auto flush = [&](int w, float *dst0, long long dstep, int cnt) {      // row r of the stage -> dst0 + r*dstep (kMaxState floats each)
        WAVE_SYNC();
        if (lane < cnt) {
            const float4 *src4 = (const float4 *)&stage[w][lane][0];
            float4 *dst = (float4 *)(dst0 + (long long)lane * dstep);
            dst[0] = src4[0]; dst[1] = src4[1]; dst[2] = src4[2]; dst[3] = src4[3];
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (wave == 0) {
        // forwards
        float *tm = term[0], *sv = svec[0];
        if (is_state) { F[lane] = 0.0f; sv[lane] = 0.0f; }
        WAVE_SYNC();
        // Per block: every entry's term in parallel, the maximum per destination state (exact, order-free), ONE expf per
        // lane in parallel, then the sums in the reference's order (flip: from-state 0..ns-1; flop: stay, move) -- the same
        // values as a serial logsumexp per state, without ten dependent expf on the chain.
        float *mx = smax[0], *te = texp[0];
        float ring[kDepth];
        auto fetch = [&](int blk) { return T[(size_t)min(blk, Tb - 1) * Ps + lane_c]; };
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
                const int blk = b0 + k;
                if (blk >= Tb) break;
                const float term_l = valid ? cur[k] + sv[src] : -INFINITY;
                tm[lane] = term_l;
                WAVE_SYNC();
                float m = 0.0f;
                if (is_state) {
                    if (is_flip) {
                        m = tm[lane * ns];
                        for (int f = 1; f < ns; f++) m = fmaxf(m, tm[lane * ns + f]);
                    } else {
                        m = fmaxf(tm[off + lane], tm[off + lane - nbase]);
                    }
                    mx[lane] = m;
                }
                WAVE_SYNC();
                te[lane] = valid ? expf(term_l - mx[dst]) : 0.0f;
                WAVE_SYNC();
                if (is_state) {
                    float e;
                    if (is_flip) {
                        e = 0.0f;
                        for (int f = 0; f < ns; f++) e += te[lane * ns + f];
                    } else {
                        e = te[off + lane] + te[off + lane - nbase];
                    }
                    const float val = m + logf(e);
                    stage[0][blk & 63][lane] = val;                        // fwd[blk + 1]
                    sv[lane] = val;
                }
                if ((blk & 63) == 63 || blk == Tb - 1) flush(0, F + (size_t)((blk & ~63) + 1) * kMaxState, kMaxState, (blk & 63) + 1);
                else WAVE_SYNC();
            }
        }
    } else if (wave == 1) {
        // backwards; Bw[blk] is the vector that meets block blk-1's transitions
        float *tm = term[1], *sv = svec[1];
        if (is_state) sv[lane] = 0.0f;
        WAVE_SYNC();
        float *mx = smax[1], *te = texp[1];
        float ring[kDepth];
        auto fetch = [&](int blk) { return T[(size_t)max(blk, 0) * Ps + lane_c]; };       // blk counts down
#pragma unroll
        for (int k = 0; k < kDepth; k++) ring[k] = fetch(Tb - 1 - k);
        for (int j0 = 0; j0 < Tb; j0 += kDepth) {                                          // j = Tb - blk: 0, 1, ...
            float cur[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) cur[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kDepth; k++) ring[k] = fetch(Tb - 1 - (j0 + kDepth + k));
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                const int j = j0 + k;
                if (j >= Tb) break;
                if (is_state) stage[1][j & 63][lane] = sv[lane];           // bwd[blk], blk = Tb - j
                if ((j & 63) == 63 || j == Tb - 1) flush(1, Bw + (size_t)(Tb - (j & ~63)) * kMaxState, -(long long)kMaxState, (j & 63) + 1);
                const float term_l = valid ? cur[k] + sv[dst] : -INFINITY;
                tm[lane] = term_l;
                WAVE_SYNC();
                // state `lane` as a SOURCE: its flop exit (entry off + lane) and its nbase flip exits (entries b1*ns + lane)
                float m = 0.0f;
                if (is_state) {
                    m = tm[off + lane];
                    for (int b1 = 0; b1 < nbase; b1++) m = fmaxf(m, tm[b1 * ns + lane]);
                    mx[lane] = m;
                }
                WAVE_SYNC();
                te[lane] = valid ? expf(term_l - mx[src]) : 0.0f;
                WAVE_SYNC();
                if (is_state) {
                    float e = te[off + lane];
                    for (int b1 = 0; b1 < nbase; b1++) e += te[b1 * ns + lane];
                    sv[lane] = m + logf(e);
                }
                WAVE_SYNC();
            }
        }
    }
#undef WAVE_SYNC
    __syncthreads();
    for (int blk = threadIdx.x; blk < Tb; blk += 256) {
        const float *sc = T + (size_t)blk * Ps;
        float *x = Pp + (size_t)blk * Ps;
        const float *f = F + (size_t)blk * kMaxState, *bb = Bw + (size_t)(blk + 1) * kMaxState;
        float m = -INFINITY;
        for (int r = 0; r < P; r++) {
            const int from = r % ns;
            int to;
            if (r < off) to = r / ns;
            else { const int idx = r - off; to = (idx < nbase) ? idx + nbase : idx; }
            const float v = (f[from] + bb[to]) + sc[r];                      // decode.c:451-461
            x[r] = v;
            m = fmaxf(m, v);
        }
        float sum = 0.0f;
        for (int r = 0; r < P; r++) sum += expf(x[r] - m);
        const float lse = m + logf(sum);
        for (int r = 0; r < P; r++) x[r] -= lse;
    }
}

// ---- nstate == 10 (the 5-base models): the butterfly scheme of k_transpost8 with two entries per lane -------------------
// Forward (wave 0): flip destination g owns lanes 8g..8g+7, lane j holding its entries from states j and j+8 (j < 2); lanes
// 40..44 hold the stay / move entries of flop state 5+j.  Backward (wave 1): SOURCE states g and g+5 own lanes 8g..8g+7, lane j
// holding their exits to flip state j (j < 5) or into the flop state (j = 5; for source g the move to flop g+5, for source g+5
// the stay) -- so each recursion needs only in-group DPP reductions and one or two ds_bpermute per block to hand the new
// vector out.  Every per-state logsumexp is max + log(sum exp) as in k_transpost8 (same value up to fp32 rounding of the
// association); k_transpost_lds keeps the reference's order of the sums (FFHIP_EXACT_ORDER, other nstate).
__global__ void __launch_bounds__(256)
k_transpost10(const float *__restrict__ trans, float *__restrict__ post, float *__restrict__ fwdbuf, float *__restrict__ bwdbuf, int TbS,
              int Ps, const int *__restrict__ tbs, const int *__restrict__ only) {
    constexpr int P = 60, ns = 10, nbase = 5, off = 50;
    if (only && !only[blockIdx.x]) return;               // the reads k_crf_fb<10> has left (score range too wide for its linear form)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *T = trans + (size_t)blockIdx.x * TbS * Ps;
    float *Pp = post + (size_t)blockIdx.x * TbS * Ps;
    float *F = fwdbuf + (size_t)blockIdx.x * (TbS + 1) * kMaxState;
    float *Bw = bwdbuf + (size_t)blockIdx.x * (TbS + 1) * kMaxState;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;
    if (Tb <= 0) return;
    const int g = lane >> 3, j = lane & 7;
    const float NEG = -INFINITY;
    __shared__ float stage[2][64][kMaxState];
    constexpr int kDepth = 8;
    auto flush = [&](int w, float *dst0, long long dstep, int cnt) {      // row r of the stage -> dst0 + r*dstep (kMaxState floats each)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            const float4 *src = (const float4 *)&stage[w][lane][0];
            float4 *dst = (float4 *)(dst0 + (long long)lane * dstep);
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];             // ns = 10 states
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (wave == 0) {
        const bool flip = g < nbase, flop = (g == nbase && j < nbase);
        const bool valid0 = flip || flop, valid1 = (flip && j < 2) || flop;
        const int e0 = flip ? g * ns + j : (flop ? off + nbase + j : 0);                  // flop: stay, from = 5 + j
        const int e1 = flip ? (j < 2 ? g * ns + 8 + j : e0) : (flop ? off + j : 0);      // flop: move, from = j
        const int src0 = flip ? ff10_src_lane(j) : (flop ? ff10_src_lane(nbase + j) : 0);
        const int src1 = flip ? ff10_src_lane(j < 2 ? j + 8 : j) : (flop ? ff10_src_lane(j) : 0);
        float pv0 = 0.0f, pv1 = 0.0f;
        if (lane < ns) F[lane] = 0.0f;
        float ring0[kDepth], ring1[kDepth];
        auto fetch0 = [&](int blk) { return T[(size_t)min(blk, Tb - 1) * Ps + e0]; };
        auto fetch1 = [&](int blk) { return T[(size_t)min(blk, Tb - 1) * Ps + e1]; };
#pragma unroll
        for (int k = 0; k < kDepth; k++) { ring0[k] = fetch0(k); ring1[k] = fetch1(k); }
        for (int b0 = 0; b0 < Tb; b0 += kDepth) {
            float cur0[kDepth], cur1[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) { cur0[k] = ring0[k]; cur1[k] = ring1[k]; }
#pragma unroll
            for (int k = 0; k < kDepth; k++) { ring0[k] = fetch0(b0 + kDepth + k); ring1[k] = fetch1(b0 + kDepth + k); }
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                const int blk = b0 + k;
                if (blk >= Tb) break;
                const float t0 = valid0 ? cur0[k] + pv0 : NEG, t1 = valid1 ? cur1[k] + pv1 : NEG;
                float m = fmaxf(t0, t1);
                if (flip) { m = fmaxf(m, xor4_f(m)); m = fmaxf(m, xor1_f(m)); m = fmaxf(m, xor2_f(m)); }
                float e = (valid0 ? expf(t0 - m) : 0.0f) + (valid1 ? expf(t1 - m) : 0.0f);
                if (flip) { e += xor4_f(e); e += xor1_f(e); e += xor2_f(e); }
                const float val = m + logf(e);
                pv0 = __shfl(val, src0);
                pv1 = __shfl(val, src1);
                if (lane < 8) stage[0][blk & 63][lane] = pv0;              // fwd[blk + 1], states 0..7 (lane j of group 0 reads state j)
                if (lane < 2) stage[0][blk & 63][8 + lane] = pv1;          // states 8, 9
                if ((blk & 63) == 63 || blk == Tb - 1) flush(0, F + (size_t)((blk & ~63) + 1) * kMaxState, kMaxState, (blk & 63) + 1);
            }
        }
    } else if (wave == 1) {
        // lane (g, j), g < 5, j < 6: slot 0 = source state g, slot 1 = source state g + 5; exit j < 5 -> flip state j, j = 5 -> flop state g + 5
        const bool act = g < nbase && j <= nbase;
        const int e0 = act ? (j < nbase ? j * ns + g : off + g) : 0;
        const int e1 = act ? (j < nbase ? j * ns + g + nbase : off + g + nbase) : 0;
        const int srcl = 8 * min(j, nbase - 1);        // new value of flip state j: slot 0 of group j (used by lanes j < 5)
        float pb_to0 = 0.0f;                           // bwd[to] for this lane's exits: flip state j, or the group's own flop state
        float n0 = 0.0f, n1 = 0.0f;                    // bwd of the group's two source states (replicated over the group)
        float ring0[kDepth], ring1[kDepth];
        auto fetch0 = [&](int blk) { return T[(size_t)max(blk, 0) * Ps + e0]; };       // blk counts down
        auto fetch1 = [&](int blk) { return T[(size_t)max(blk, 0) * Ps + e1]; };
#pragma unroll
        for (int k = 0; k < kDepth; k++) { ring0[k] = fetch0(Tb - 1 - k); ring1[k] = fetch1(Tb - 1 - k); }
        for (int j0 = 0; j0 < Tb; j0 += kDepth) {                                          // jj = Tb - blk: 0, 1, ...
            float cur0[kDepth], cur1[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) { cur0[k] = ring0[k]; cur1[k] = ring1[k]; }
#pragma unroll
            for (int k = 0; k < kDepth; k++) { ring0[k] = fetch0(Tb - 1 - (j0 + kDepth + k)); ring1[k] = fetch1(Tb - 1 - (j0 + kDepth + k)); }
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                const int jj = j0 + k;
                if (jj >= Tb) break;
                if (j == 0 && g < nbase) { stage[1][jj & 63][g] = n0; stage[1][jj & 63][g + nbase] = n1; }      // bwd[blk], blk = Tb - jj
                if ((jj & 63) == 63 || jj == Tb - 1) flush(1, Bw + (size_t)(Tb - (jj & ~63)) * kMaxState, -(long long)kMaxState, (jj & 63) + 1);
                const float t0 = act ? cur0[k] + pb_to0 : NEG, t1 = act ? cur1[k] + pb_to0 : NEG;
                float m0 = t0, m1 = t1;
                m0 = fmaxf(m0, xor4_f(m0)); m1 = fmaxf(m1, xor4_f(m1));
                m0 = fmaxf(m0, xor1_f(m0)); m1 = fmaxf(m1, xor1_f(m1));
                m0 = fmaxf(m0, xor2_f(m0)); m1 = fmaxf(m1, xor2_f(m1));
                float x0 = act ? expf(t0 - m0) : 0.0f, x1 = act ? expf(t1 - m1) : 0.0f;
                x0 += xor4_f(x0); x1 += xor4_f(x1);
                x0 += xor1_f(x0); x1 += xor1_f(x1);
                x0 += xor2_f(x0); x1 += xor2_f(x1);
                n0 = m0 + logf(x0);
                n1 = m1 + logf(x1);
                const float flipv = __shfl(n0, srcl);
                pb_to0 = (j < nbase) ? flipv : n1;         // exits into flip state j; or into the group's flop state g + 5
            }
        }
    }
    __syncthreads();
    // posterior of transition r of block blk = (fwd[blk][from] + bwd[blk+1][to]) + trans (decode.c:451-461), then the per-block
    // log-normalisation over the P entries (flappie_matrix.c:450-467); one block per thread
    for (int blk = threadIdx.x; blk < Tb; blk += 256) {
        const float *x = T + (size_t)blk * Ps;
        float *o = Pp + (size_t)blk * Ps;
        float f[ns], bb[ns];
#pragma unroll
        for (int k = 0; k < ns; k++) { f[k] = F[(size_t)blk * kMaxState + k]; bb[k] = Bw[(size_t)(blk + 1) * kMaxState + k]; }
        float v[P];
        float m = NEG;
#pragma unroll
        for (int r = 0; r < P; r++) {
            const int from = r % ns;
            const int to = (r < off) ? (r / ns) : ((r - off < nbase) ? r - off + nbase : r - off);
            v[r] = (f[from] + bb[to]) + x[r];
            m = fmaxf(m, v[r]);
        }
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < P; r++) sum += expf(v[r] - m);
        const float lse = m + logf(sum);
#pragma unroll
        for (int r = 0; r < P; r++) o[r] = v[r] - lse;
    }
}

void launch_transpost(hipStream_t s, const float *trans, float *post, float *fwd, int nread, int Tb, int nbase, int Ps, const int *tbs,
                      double *E, int *wide) {
    const int P = 2 * nbase * (nbase + 1);
    if (((nbase == 4 && Ps == 40) || (nbase == 5 && Ps == 60)) && E && wide && !dbg("exact_order") && !dbg("decode_r2")) {
        // linear-space fp64 recursions on exp(score - block max) (ffhip_decode.hip); reads whose scores span too much keep the log-space kernel
        hipMemsetAsync(wide, 0, (size_t)nread * sizeof(int), s);
        launch_crf_exp(s, trans, E, nread, Tb, nbase, Ps, tbs, wide, kFbRange);
        launch_crf_fb(s, nbase, E, (float *)trans, post, (double *)fwd, nread, Tb, nullptr, tbs, 2, wide);
        if (nbase == 4)
            hipLaunchKernelGGL(k_transpost8, dim3(nread), dim3(256), 0, s, trans, post, fwd, fwd + (size_t)nread * (Tb + 1) * kMaxState, Tb, tbs, (const int *)wide);
        else
            hipLaunchKernelGGL(k_transpost10, dim3(nread), dim3(256), 0, s, trans, post, fwd, fwd + (size_t)nread * (Tb + 1) * kMaxState, Tb, Ps, tbs, (const int *)wide);
    } else if (nbase == 4 && Ps == 40 && !dbg("exact_order"))
        hipLaunchKernelGGL(k_transpost8, dim3(nread), dim3(256), 0, s, trans, post, fwd, fwd + (size_t)nread * (Tb + 1) * kMaxState, Tb, tbs, (const int *)nullptr);
    else if (nbase == 5 && !dbg("exact_order") && !dbg("crf_generic"))
        hipLaunchKernelGGL(k_transpost10, dim3(nread), dim3(256), 0, s, trans, post, fwd, fwd + (size_t)nread * (Tb + 1) * kMaxState, Tb, Ps, tbs, (const int *)nullptr);
    else if (!dbg("exact_order"))
        hipLaunchKernelGGL(k_transpost_lds, dim3(nread), dim3(256), 0, s, trans, post, fwd, fwd + (size_t)nread * (Tb + 1) * kMaxState, Tb, nbase, P, Ps, tbs);
    else
        hipLaunchKernelGGL(k_transpost, dim3(nread), dim3(64), 0, s, trans, post, fwd, Tb, nbase, P, Ps, tbs);
}

// ---- Viterbi ---------------------------------------------------------------------------------
// decode.c:119-204 with its tie rules: a flop state keeps "stay" unless "move" is strictly
// greater; a flip state scans from-states in ascending order and replaces only on strictly
// greater (lowest index wins ties); the final state is the first maximum (util.c:17-31).
// Traceback pointers are bytes; the traceback walks them through LDS in chunks.
constexpr int kTbChunk = 2048;
__global__ void __launch_bounds__(64)
k_viterbi(const float *__restrict__ M, uint8_t *__restrict__ tbbuf, int *__restrict__ path,
          float *__restrict__ qpath, float *__restrict__ score_out, int TbS, int nbase, int P, int Ps, const int *__restrict__ tbs) {
    __shared__ uint8_t tb_lds[kTbChunk * kMaxState];
    __shared__ int path_lds[kTbChunk + 1];
    const int lane = threadIdx.x;
    const int ns = 2 * nbase, off = nbase * ns;
    const float *T = M + (size_t)blockIdx.x * TbS * Ps;
    uint8_t *tb = tbbuf + (size_t)blockIdx.x * TbS * kMaxState;
    int *pth = path + (size_t)blockIdx.x * (TbS + 1);
    float *qp = qpath + (size_t)blockIdx.x * (TbS + 1);
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    const bool is_state = lane < ns, is_flip = lane < nbase;
    const bool valid = lane < P;
    const int src = lane % ns;                      // source state of entry `lane` (off is a multiple of ns)
    __shared__ float cand[64];
    __shared__ float pvs[kMaxState];
#define WAVE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

    // Forward recursion, one wave.  Every entry's candidate s + prev[from] in parallel; each state then scans its entries in
    // the reference's order (flip: from-state 0..ns-1, first maximum; flop: stay unless move is strictly greater) -- the
    // same comparisons on the same values as a lane-exchange chain, without 2*ns+4 dependent exchanges per block.  As in
    // k_viterbi8 the traceback bytes stay in LDS (earlier chunks of longer reads are flushed) and the scores of the next
    // kDepth blocks are in flight: nothing on the chain waits for memory.
    constexpr int kDepth = 8;
    float ring[kDepth];
    const int lane_c = valid ? lane : P - 1;
    auto fetch = [&](int blk) { return T[(size_t)min(blk, Tb - 1) * Ps + lane_c]; };
#pragma unroll
    for (int k = 0; k < kDepth; k++) ring[k] = fetch(k);
    if (is_state) pvs[lane] = 0.0f;
    WAVE_SYNC();
    const int nchunk = (Tb + kTbChunk - 1) / kTbChunk;
    for (int c = 0; c < nchunk; c++) {
        const int c0 = c * kTbChunk, n = min(kTbChunk, Tb - c0);
        for (int b0 = 0; b0 < n; b0 += kDepth) {
            float cur[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) cur[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kDepth; k++) ring[k] = fetch(c0 + b0 + kDepth + k);
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                if (b0 + k >= n) break;
                cand[lane] = valid ? cur[k] + pvs[src] : -INFINITY;
                WAVE_SYNC();
                if (is_state) {
                    float best;
                    int arg;
                    if (is_flip) {
                        best = cand[lane * ns]; arg = 0;
                        for (int f = 1; f < ns; f++) {
                            const float sc = cand[lane * ns + f];
                            if (sc > best) { best = sc; arg = f; }
                        }
                    } else {
                        const float stay = cand[off + lane], move = cand[off + lane - nbase];
                        best = stay; arg = lane;
                        if (move > stay) { best = move; arg = lane - nbase; }
                    }
                    pvs[lane] = best;
                    tb_lds[(b0 + k) * kMaxState + lane] = (uint8_t)arg;
                }
                WAVE_SYNC();
            }
        }
        if (c + 1 < nchunk) {             // a longer read: this chunk's bytes leave LDS (kTbChunk is a multiple of kDepth: the ring stays aligned)
            __syncthreads();
            for (int i = lane; i < n * (kMaxState / 4); i += 64)
                ((uint32_t *)(tb + (size_t)c0 * kMaxState))[i] = ((const uint32_t *)tb_lds)[i];
            __syncthreads();
        }
    }
#undef WAVE_SYNC
    // final score and state: first maximum
    float score = pvs[0];
    int last = 0;
    for (int st = 1; st < ns; st++) {
        const float v = pvs[st];
        if (v > score) { score = v; last = st; }
    }
    if (lane == 0) { score_out[blockIdx.x] = score; qp[0] = NAN; }
    __syncthreads();
    // traceback, last chunk first (it is still in LDS); `last` = path[c1]
    for (int c = nchunk - 1; c >= 0; c--) {
        const int c0 = c * kTbChunk, n = min(kTbChunk, Tb - c0), c1 = c0 + n;
        if (c != nchunk - 1) {
            for (int i = lane; i < n * (kMaxState / 4); i += 64)
                ((uint32_t *)tb_lds)[i] = ((const uint32_t *)(tb + (size_t)c0 * kMaxState))[i];
            __syncthreads();
        }
        if (lane == 0) {
            int p = last;
            path_lds[n] = p;
            for (int i = n; i > 0; i--) { p = tb_lds[(i - 1) * kMaxState + p]; path_lds[i - 1] = p; }
        }
        __syncthreads();
        if (c1 == Tb && lane == 0) pth[Tb] = path_lds[n];
        for (int i = lane; i < n; i += 64) {
            const int from = path_lds[i], to = path_lds[i + 1];
            pth[c0 + i] = from;
            // trans_lookup, decode.c:104-114
            const int idx = (to < nbase) ? (to * ns + from) : (off + from);
            qp[c0 + i + 1] = T[(size_t)(c0 + i) * Ps + idx];
        }
        last = path_lds[0];
        __syncthreads();
    }
}

// Viterbi for nstate == 8: same lane layout; bit-identical to the sequential scan because max is
// exact and the tie rules are reproduced (flip: lowest from-state wins; flop: stay unless move is
// strictly greater).
__global__ void __launch_bounds__(64)
k_viterbi8(const float *__restrict__ M, uint8_t *__restrict__ tbbuf, int *__restrict__ path,
           float *__restrict__ qpath, float *__restrict__ score_out, int TbS, const int *__restrict__ tbs) {
    constexpr int P = 40, Ps = 40, ns = 8, nbase = 4, off = 32;
    __shared__ uint8_t tb_lds[kTbChunk * kMaxState];
    __shared__ int path_lds[kTbChunk + 1];
    const int lane = threadIdx.x;
    const float *T = M + (size_t)blockIdx.x * TbS * Ps;
    uint8_t *tb = tbbuf + (size_t)blockIdx.x * TbS * kMaxState;
    int *pth = path + (size_t)blockIdx.x * (TbS + 1);
    float *qp = qpath + (size_t)blockIdx.x * (TbS + 1);
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    const bool valid = lane < P, flip = lane < 32;
    const int st = lane & 7;
    const float NEG = -INFINITY;

    // Forward recursion.  The chain is ~150 cycles of work per block, so nothing on it may wait for memory:
    //  * the traceback bytes go to LDS (a chunk of kTbChunk blocks; earlier chunks of longer reads are flushed to HBM with wide
    //    stores) -- a global store per block shares the wave's memory counter with the loads, and the compiler then waits for
    //    the store's acknowledgement (~700 cycles) before it may use the next block's scores;
    //  * the scores of the next kDepth blocks are in flight while kDepth blocks are processed.
    constexpr int kDepth = 8;
    float ring[kDepth];
    // (clamped, never predicated: a branch around a load makes the compiler drain the whole queue before the next use)
    const int lane_c = valid ? lane : P - 1;
    auto fetch = [&](int blk) { return T[(size_t)min(blk, Tb - 1) * Ps + lane_c]; };
#pragma unroll
    for (int k = 0; k < kDepth; k++) ring[k] = fetch(k);
    float pv = 0.0f;
    const int nchunk = (Tb + kTbChunk - 1) / kTbChunk;
    for (int c = 0; c < nchunk; c++) {
        const int c0 = c * kTbChunk, n = min(kTbChunk, Tb - c0);
        for (int b0 = 0; b0 < n; b0 += kDepth) {
            float cur[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) cur[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kDepth; k++) ring[k] = fetch(c0 + b0 + kDepth + k);
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                if (b0 + k >= n) break;
                // The chain carries the VALUE only: max is exact, so the winner can be identified afterwards as the lowest
                // from-state whose candidate equals the maximum (= the reference's scan with its strict >), from a ballot, off
                // the critical path.  Chain per block: one add, three max over DPP partners, one ds_bpermute.
                const float cand = valid ? cur[k] + pv : NEG;
                float v = cand;
                const float o4 = xor4_f(v);
                bool moved = false;
                if (!flip) {
                    // flop pair (lanes 32..39): partner = lane ^ 4; the stay entry is the one with from >= 4; stay unless move is strictly greater
                    const bool i_am_stay = (lane & 4) != 0;
                    const float stay = i_am_stay ? v : o4, move = i_am_stay ? o4 : v;
                    moved = move > stay;
                    v = moved ? move : stay;
                } else {
                    v = fmaxf(v, o4);
                    v = fmaxf(v, xor1_f(v));
                    v = fmaxf(v, xor2_f(v));
                }
                // NaN scores (never produced by the network, possible through ffhip_viterbi): the reference's scan starts from
                // the from-state-0 candidate and replaces it only on a strict >, so a NaN there stays the maximum while NaNs
                // elsewhere are skipped; fmaxf skips them all.  Rare and wave-uniform, so the chain pays one scalar branch.
                if (__builtin_expect(__ballot(cand != cand) != 0ull, 0)) {
                    const float c0 = __shfl(cand, lane & ~7);
                    if (flip && c0 != c0) v = c0;
                }
                const unsigned eq = (unsigned)__ballot(flip && cand == v);                  // bit 8*to + from
                const unsigned mv = (unsigned)(__ballot(!flip && valid && moved) >> 32);    // bit (b2 - 4) [and bit b2] of flop state b2
                const int src = ff8_src_lane(st);
                pv = __shfl(v, src);
                int arg = 0;
                if (lane < nbase) { const unsigned m = (eq >> (8 * lane)) & 0xffu; arg = m ? __builtin_ctz(m) : 0; }    // empty (NaN maximum): state 0, as the reference's scan
                else if (lane < ns) arg = ((mv >> (lane - nbase)) & 1u) ? lane - nbase : lane;
                if (lane < ns) tb_lds[(b0 + k) * kMaxState + lane] = (uint8_t)arg;
            }
        }
        if (c + 1 < nchunk) {             // a longer read: this chunk's bytes leave LDS (kTbChunk is a multiple of kDepth: the ring stays aligned)
            __syncthreads();
            for (int i = lane; i < n * (kMaxState / 4); i += 64)
                ((uint32_t *)(tb + (size_t)c0 * kMaxState))[i] = ((const uint32_t *)tb_lds)[i];
            __syncthreads();
        }
    }
    float score = __shfl(pv, 0);
    int last = 0;
    for (int s2 = 1; s2 < ns; s2++) {
        const float v = __shfl(pv, s2);
        if (v > score) { score = v; last = s2; }
    }
    if (lane == 0) { score_out[blockIdx.x] = score; qp[0] = NAN; }
    __syncthreads();
    // traceback, last chunk first (it is still in LDS)
    for (int c = nchunk - 1; c >= 0; c--) {
        const int c0 = c * kTbChunk, n = min(kTbChunk, Tb - c0), c1 = c0 + n;
        if (c != nchunk - 1) {
            for (int i = lane; i < n * (kMaxState / 4); i += 64)
                ((uint32_t *)tb_lds)[i] = ((const uint32_t *)(tb + (size_t)c0 * kMaxState))[i];
            __syncthreads();
        }
        if (lane == 0) {
            int p = last;
            path_lds[n] = p;
            for (int i = n; i > 0; i--) { p = tb_lds[(i - 1) * kMaxState + p]; path_lds[i - 1] = p; }
        }
        __syncthreads();
        if (c1 == Tb && lane == 0) pth[Tb] = path_lds[n];
        for (int i = lane; i < n; i += 64) {
            const int from = path_lds[i], to = path_lds[i + 1];
            pth[c0 + i] = from;
            const int idx = (to < nbase) ? (to * ns + from) : (off + from);
            qp[c0 + i + 1] = T[(size_t)(c0 + i) * Ps + idx];
        }
        last = path_lds[0];
        __syncthreads();
    }
}

// Viterbi for nstate == 10 (the 5-base models, r941_5mC): the scheme of k_viterbi8 with two candidates per lane.  Flip
// destination g (0..4) owns lanes 8g..8g+7, lane j holding its candidates from states j and j+8 (j < 2); lanes 40..44 hold
// the stay / move pair of flop state 5+j.  The chain carries the value only (local max, three DPP exchanges, two
// ds_bpermute to hand the new vector out); the winner -- lowest from-state among equals for a flip, stay unless the move is
// strictly greater for a flop, exactly decode.c:119-204 -- comes from ballots off the chain.

__global__ void __launch_bounds__(64)
k_viterbi10(const float *__restrict__ M, uint8_t *__restrict__ tbbuf, int *__restrict__ path,
            float *__restrict__ qpath, float *__restrict__ score_out, int TbS, int Ps, const int *__restrict__ tbs) {
    constexpr int ns = 10, nbase = 5, off = 50;
    __shared__ uint8_t tb_lds[kTbChunk * kMaxState];
    __shared__ int path_lds[kTbChunk + 1];
    const int lane = threadIdx.x;
    const float *T = M + (size_t)blockIdx.x * TbS * Ps;
    uint8_t *tb = tbbuf + (size_t)blockIdx.x * TbS * kMaxState;
    int *pth = path + (size_t)blockIdx.x * (TbS + 1);
    float *qp = qpath + (size_t)blockIdx.x * (TbS + 1);
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;
    if (Tb <= 0) return;
    const int g = lane >> 3, j = lane & 7;
    const bool flip = g < nbase, flop = (g == nbase && j < nbase);
    const bool valid0 = flip || flop, valid1 = (flip && j < 2) || flop;
    // entries (trans_lookup, decode.c:104-114) and source states of this lane's two candidates
    const int e0 = flip ? g * ns + j : (flop ? off + nbase + j : 0);          // flop: stay, from = 5 + j
    const int e1 = flip ? (j < 2 ? g * ns + 8 + j : e0) : (flop ? off + j : 0);      // flop: move, from = j
    const int src0 = flip ? ff10_src_lane(j) : (flop ? ff10_src_lane(nbase + j) : 0);
    const int src1 = flip ? ff10_src_lane(j < 2 ? j + 8 : j) : (flop ? ff10_src_lane(j) : 0);
    const float NEG = -INFINITY;

    constexpr int kDepth = 8;
    float ring0[kDepth], ring1[kDepth];
    auto fetch0 = [&](int blk) { return T[(size_t)min(blk, Tb - 1) * Ps + e0]; };
    auto fetch1 = [&](int blk) { return T[(size_t)min(blk, Tb - 1) * Ps + e1]; };
#pragma unroll
    for (int k = 0; k < kDepth; k++) { ring0[k] = fetch0(k); ring1[k] = fetch1(k); }
    float pv0 = 0.0f, pv1 = 0.0f;
    const int nchunk = (Tb + kTbChunk - 1) / kTbChunk;
    for (int c = 0; c < nchunk; c++) {
        const int c0 = c * kTbChunk, n = min(kTbChunk, Tb - c0);
        for (int b0 = 0; b0 < n; b0 += kDepth) {
            float cur0[kDepth], cur1[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) { cur0[k] = ring0[k]; cur1[k] = ring1[k]; }
#pragma unroll
            for (int k = 0; k < kDepth; k++) { ring0[k] = fetch0(c0 + b0 + kDepth + k); ring1[k] = fetch1(c0 + b0 + kDepth + k); }
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                if (b0 + k >= n) break;
                const float a0 = valid0 ? cur0[k] + pv0 : NEG, a1 = valid1 ? cur1[k] + pv1 : NEG;
                float v;
                bool moved = false;
                if (flip) {
                    v = fmaxf(a0, a1);
                    v = fmaxf(v, xor4_f(v));
                    v = fmaxf(v, xor1_f(v));
                    v = fmaxf(v, xor2_f(v));
                } else {
                    moved = a1 > a0;                              // stay unless the move is strictly greater
                    v = moved ? a1 : a0;
                }
                if (__builtin_expect(__ballot(a0 != a0 || a1 != a1) != 0ull, 0)) {      // NaN scores: see k_viterbi8
                    const float c0 = __shfl(a0, lane & ~7);
                    if (flip && c0 != c0) v = c0;
                }
                const unsigned long long eq0 = __ballot(flip && a0 == v);                    // bit 8*to + from, from < 8
                const unsigned long long eq1 = __ballot(flip && valid1 && a1 == v);          // bit 8*to + (from - 8)
                const unsigned mv = (unsigned)(__ballot(flop && moved) >> 40);               // bit j: flop state 5 + j moved
                pv0 = __shfl(v, src0);
                pv1 = __shfl(v, src1);
                int arg = 0;
                if (lane < nbase) {
                    const unsigned lo = (unsigned)(eq0 >> (8 * lane)) & 0xffu, hi = (unsigned)(eq1 >> (8 * lane)) & 0x3u;
                    arg = lo ? __builtin_ctz(lo) : (hi ? 8 + __builtin_ctz(hi) : 0);      // none (NaN maximum): state 0, as the reference's scan
                } else if (lane < ns) {
                    arg = ((mv >> (lane - nbase)) & 1u) ? lane - nbase : lane;
                }
                if (lane < ns) tb_lds[(b0 + k) * kMaxState + lane] = (uint8_t)arg;
            }
        }
        if (c + 1 < nchunk) {             // a longer read: this chunk's bytes leave LDS
            __syncthreads();
            for (int i = lane; i < n * (kMaxState / 4); i += 64)
                ((uint32_t *)(tb + (size_t)c0 * kMaxState))[i] = ((const uint32_t *)tb_lds)[i];
            __syncthreads();
        }
    }
    // final score and state: first maximum (state s < 8 is pv0 of lane s, states 8 and 9 are pv1 of lanes 0 and 1)
    float score = __shfl(pv0, 0);
    int last = 0;
    for (int s2 = 1; s2 < ns; s2++) {
        const float v = s2 < 8 ? __shfl(pv0, s2) : __shfl(pv1, s2 - 8);
        if (v > score) { score = v; last = s2; }
    }
    if (lane == 0) { score_out[blockIdx.x] = score; qp[0] = NAN; }
    __syncthreads();
    for (int c = nchunk - 1; c >= 0; c--) {
        const int c0 = c * kTbChunk, n = min(kTbChunk, Tb - c0), c1 = c0 + n;
        if (c != nchunk - 1) {
            for (int i = lane; i < n * (kMaxState / 4); i += 64)
                ((uint32_t *)tb_lds)[i] = ((const uint32_t *)(tb + (size_t)c0 * kMaxState))[i];
            __syncthreads();
        }
        if (lane == 0) {
            int p = last;
            path_lds[n] = p;
            for (int i = n; i > 0; i--) { p = tb_lds[(i - 1) * kMaxState + p]; path_lds[i - 1] = p; }
        }
        __syncthreads();
        if (c1 == Tb && lane == 0) pth[Tb] = path_lds[n];
        for (int i = lane; i < n; i += 64) {
            const int from = path_lds[i], to = path_lds[i + 1];
            pth[c0 + i] = from;
            const int idx = (to < nbase) ? (to * ns + from) : (off + from);
            qp[c0 + i + 1] = T[(size_t)(c0 + i) * Ps + idx];
        }
        last = path_lds[0];
        __syncthreads();
    }
}

void launch_viterbi(hipStream_t s, const float *score_mat, uint8_t *tb, int *path, float *qpath, float *score,
                    int nread, int Tb, int nbase, int Ps, const int *tbs, ReadMap map) {
    const int P = 2 * nbase * (nbase + 1);
    if (nbase == 4 && Ps == 40 && !dbg("decode_r2"))
        launch_viterbi8x(s, score_mat, tb, path, qpath, score, nread, Tb, tbs, map);
    else if (nbase == 4 && Ps == 40)
        hipLaunchKernelGGL(k_viterbi8, dim3(nread), dim3(64), 0, s, score_mat, tb, path, qpath, score, Tb, tbs);
    else if (nbase == 5 && Ps == 60 && !dbg("exact_order") && !dbg("decode_r2"))
        launch_viterbi10x(s, score_mat, tb, path, qpath, score, nread, Tb, tbs, map);
    else if (nbase == 5 && !dbg("exact_order"))
        hipLaunchKernelGGL(k_viterbi10, dim3(nread), dim3(64), 0, s, score_mat, tb, path, qpath, score, Tb, Ps, tbs);
    else
        hipLaunchKernelGGL(k_viterbi, dim3(nread), dim3(64), 0, s, score_mat, tb, path, qpath, score, Tb, nbase, P, Ps, tbs);
}

// ---- change positions -> base and quality strings ---------------------------------------------
// decode.c:66-79 + flappie.c:284-292 + phredf/qscoref (util.h:284-305).  pos runs over [1, nblock):
// path[nblock] and the base of path[0] are never emitted, as in the reference.
__global__ void __launch_bounds__(64)
k_assemble(const int *__restrict__ path, const float *__restrict__ qpath, char *__restrict__ bases,
           char *__restrict__ quals, int *__restrict__ lens, int TbS, int nbase, const int *__restrict__ tbs, ReadMap map) {
    FFHIP_DECODE_PRIO_SET();
    const int lane = threadIdx.x;
    const size_t r1 = map.row1(blockIdx.x, TbS);
    const int *pth = path + r1;
    const float *qp = qpath + r1;
    char *bs = bases + r1;
    char *qs = quals + r1;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    int count = 0;
    for (int p0 = 1; p0 < Tb; p0 += 64) {
        const int pos = p0 + lane;
        bool change = false;
        int st = 0;
        if (pos < Tb) { st = pth[pos]; change = (st != pth[pos - 1]); }
        const unsigned long long mask = __ballot(change);
        if (change) {
            const int idx = count + __popcll(mask & ((1ull << lane) - 1ull));
            const char lut[5] = { 'A', 'C', 'G', 'T', 'Z' };          // decode.h:16
            bs[idx] = lut[st % nbase];
            const float p = expf(qp[pos]);
            const float p_clip = (p < 0.99999) ? p : 0.99999;
            const float q = -(10.0f * 0.43429448190325182765) * log1pf(-p_clip);
            char ph = (char)roundf(33.0f + q);
            qs[idx] = (ph < 126) ? ph : 126;
        }
        count += __popcll(mask);
    }
    if (lane == 0) { bs[count] = 0; qs[count] = 0; lens[blockIdx.x] = count; }
}

void launch_assemble(hipStream_t s, const int *path, const float *qpath, char *bases, char *quals, int *lens,
                     int nread, int Tb, int nbase, const int *tbs, ReadMap map) {
    hipLaunchKernelGGL(k_assemble, dim3(nread), dim3(64), 0, s, path, qpath, bases, quals, lens, Tb, nbase, tbs, map);
}

// ---- trace --------------------------------------------------------------------------------------
// exp_activation_inplace (layers.c:56-66, cephes exp) followed by trace_from_posterior
// (decode.c:499-543): column 0 sums block 0 by from-state, column blk+1 sums block blk by to-state.
// The posterior buffer is left in log space (the reference's in-place exp is folded in here).
// One thread per (column, flip state j): it produces the column's entries j (flip: a sum over the ns from-states) AND nbase + j (flop: two terms) --
// every thread evaluates ns + 2 exponentials.  (One thread per entry, as before round 4, put ns-term and 2-term sums in the same wave: every wave walked
// both loops at full length; same sums in the same order here, 0.31 -> 0.25 ms for a 1024-read 10-state batch beside the next batch's convolution.)
__global__ void __launch_bounds__(256)
k_trace(const float *__restrict__ post, int32_t *__restrict__ trace, int TbS, int nbase, int P, int Ps, int is_log, const int *__restrict__ tbs, ReadMap map) {
    FFHIP_DECODE_PRIO_SET();
    const int ns = 2 * nbase, off = nbase * ns;
    const float *Pp = post + map.row0(blockIdx.y, TbS) * Ps;
    int32_t *tr = trace + map.row1(blockIdx.y, TbS) * ns;
    const int Tb = tbs ? tbs[blockIdx.y] : TbS;          // this read's blocks; TbS is the batch's stride
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (column, flip state)
    if (i >= (Tb + 1) * nbase) return;
    const int col = i / nbase, j = i % nbase;
    auto pr = [&](float v) { return is_log ? exp_cephes(v) : v; };
    float flip, flop;
    if (col == 0) {                                      // block 0 by from-state: states j and nbase + j
        flip = 0.0f; flop = 0.0f;
        for (int to = 0; to < nbase; to++) { flip += pr(Pp[to * ns + j]); flop += pr(Pp[to * ns + nbase + j]); }
        flip += pr(Pp[off + j]);
        flop += pr(Pp[off + nbase + j]);
    } else {
        const float *x = Pp + (size_t)(col - 1) * Ps;
        flip = pr(x[j * ns]);
        for (int f = 1; f < ns; f++) flip += pr(x[j * ns + f]);
        flop = pr(x[off + j]) + pr(x[off + nbase + j]);
    }
    tr[col * ns + j] = (int32_t)roundf(255.0f * flip);
    tr[col * ns + nbase + j] = (int32_t)roundf(255.0f * flop);
}

void launch_trace(hipStream_t s, const float *post, int32_t *trace, int nread, int Tb, int nbase, int Ps, int is_log, const int *tbs, ReadMap map) {
    const int P = 2 * nbase * (nbase + 1);
    const int n = (Tb + 1) * nbase;
    hipLaunchKernelGGL(k_trace, dim3((n + 255) / 256, nread), dim3(256), 0, s, post, trace, Tb, nbase, P, Ps, is_log, tbs, map);
}

// exp_activation_inplace (layers.c:56-66) on the meaningful rows
__global__ void k_exp_inplace(float *x, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = exp_cephes(x[i]);
}
void launch_exp_inplace(hipStream_t s, float *x, size_t n) {
    hipLaunchKernelGGL(k_exp_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n);
}

// ---- debug tap: tile-interleaved -> dense [Tb][H] of one read --------------------------------
__global__ void k_untile(const float *__restrict__ act, float *__restrict__ dense, int read, int Tb, int B16, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Tb * H) return;
    const int t = i / H, f = i % H;
    const int rt = read / 16, r = read % 16;
    dense[i] = act[((size_t)t * B16 + rt) * H * 16 + (size_t)(f / 4) * 64 + r * 4 + (f % 4)];
}

void launch_untile(hipStream_t s, const float *act, float *dense, int read, int Tb, int B16, int H) {
    hipLaunchKernelGGL(k_untile, dim3((Tb * H + 255) / 256), dim3(256), 0, s, act, dense, read, Tb, B16, H);
}

// ---- rows of different lengths from anywhere on the device into the batch's signal buffer (ffhip_batch_set_prepared) ----
// (dst_off, packed batches: where read r's first sample goes, in floats from dst -- its slot's row and its place in it)
__global__ void __launch_bounds__(256) k_gather_rows(const float *const *__restrict__ src, const int *__restrict__ lens, float *__restrict__ dst, size_t row_stride,
                                                     const long long *__restrict__ dst_off) {
    const float *s = src[blockIdx.x];
    const int n = lens[blockIdx.x];
    float *d = dst + (dst_off ? (size_t)dst_off[blockIdx.x] : (size_t)blockIdx.x * row_stride);
    for (int i = threadIdx.x; i < n; i += 256) d[i] = s[i];
}
void launch_gather_rows(hipStream_t s, const float *const *src, const int *lens, float *dst, size_t row_stride, int nrow, const long long *dst_off) {
    hipLaunchKernelGGL(k_gather_rows, dim3(nrow), dim3(256), 0, s, src, lens, dst, row_stride, dst_off);
}

// ---- packed batches: the strided convolution's window table and the layer kernels' live mask, built on the device (a 512-row batch of 200 000-sample rows has
// 2 x 84 MB of table: filled and uploaded by the host it cost 0.4 s a batch) ------------------------------------------------------------------------------------
// build_conv_plan (ffhip_engine.hip: the reference's three regions, layers.c:216-271, in index space) evaluated COLUMN BY COLUMN: the candidates of column c in the
// order the host function adds them -- left edge, the window phases w = 0, s, 2 s ..., right edge -- the first two are the column's windows, a third sets *overflow.
// reads[v] = { row, first column of the read in its row, first input sample of the read in its row, input samples }.
__global__ void __launch_bounds__(256) k_pack_conv_table(const int4 *__restrict__ reads, int winlen, int s, int Tmax, int *__restrict__ x0a, int *__restrict__ x0b,
                                                         unsigned *__restrict__ overflow) {
    const int4 rd = reads[blockIdx.y];
    const int T = rd.w, c = blockIdx.x * 256 + threadIdx.x;
    const int padL = (winlen - 1) / 2, padR = winlen / 2, Tout = (T + s - 1) / s;
    if (c >= Tout) return;
    const int ncolsL = (padL + s - 1) / s, shiftX = ncolsL * s - padL, nstepC = (winlen + s - 1) / s, nstepX = s * nstepC;
    int a = kNoWindow, b = kNoWindow;
    bool over = false;
    auto add = [&](int x0) { if (a == kNoWindow) a = x0; else if (b == kNoWindow) b = x0; else over = true; };
    if (c * s < padL) add(c * s - padL);
    for (int w = 0; w < winlen; w += s) {
        const int d = c - (ncolsL + w / s);
        if (d >= 0 && d % nstepC == 0 && d / nstepC < (T - shiftX - w) / nstepX) add(shiftX + w + nstepX * (d / nstepC));
    }
    const int maxCol = (T - shiftX) / nstepX, rem = (T - shiftX) % nstepX;
    const int colR = ncolsL + nstepC * (maxCol - 1) + rem / s + 1, startR = s - (padL + T - winlen) % s - 1;
    for (int w = startR; w < padR; w += s)
        if (colR + w / s == c) add(T - winlen + 1 + w);
    const size_t at = (size_t)rd.x * Tmax + rd.y + c;
    x0a[at] = (a == kNoWindow) ? kNoWindow : a + rd.z;
    x0b[at] = (b == kNoWindow) ? kNoWindow : b + rd.z;
    if (over) *overflow = 1u;
}
// live[t][read tile] |= bit of the row, for every block of every read.  reads[v] = { row, first block, blocks, - }
__global__ void __launch_bounds__(256) k_pack_live(const int4 *__restrict__ reads, int B16, unsigned *__restrict__ live) {
    const int4 rd = reads[blockIdx.y];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < rd.z) atomicOr(live + (size_t)(rd.y + t) * B16 + (rd.x >> 4), 1u << (rd.x & 15));
}
void launch_pack_conv_table(hipStream_t s, const int4 *reads, int nread, int maxcols, int winlen, int stride, int Tmax, int *x0a, int *x0b, unsigned *overflow) {
    if (nread > 0) hipLaunchKernelGGL(k_pack_conv_table, dim3((maxcols + 255) / 256, nread), dim3(256), 0, s, reads, winlen, stride, Tmax, x0a, x0b, overflow);
}
void launch_pack_live(hipStream_t s, const int4 *reads, int nread, int maxblocks, int B16, unsigned *live) {
    if (nread > 0) hipLaunchKernelGGL(k_pack_live, dim3((maxblocks + 255) / 256, nread), dim3(256), 0, s, reads, B16, live);
}

}  // namespace ffhip
