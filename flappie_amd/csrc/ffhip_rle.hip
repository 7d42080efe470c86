// ffhip_rle.hip -- run-length ("runnie", model rle_r941_native) head and decoders on the GPU.
//
// The network trunk is the LSTM5 one (networks.c:672-725); what differs from the flip-flop path:
//   * head  globalnorm_runlengthV2 (layers.c:1325-1358): rows [0,nbase) shape = 1 + softplus, [nbase,2nbase)
//     scale = 1e-8 + softplus, the 2*nbase*nbase transition rows 5 tanh(x)/temperature, globally normalised
//     with runlengthV2_partition_function (layers.c:1255-1302: fp64 recursion over nbase move + nbase stay states,
//     the stay update through the FLOAT logsumexpf -- reproduced);
//   * transpost_crf_runlength (decode.c:1037-1159): forward/backward transition posteriors, not normalised per
//     block, shape/scale rows copied through;
//   * decode_crf_runlength (decode.c:927-1013): Viterbi over the 2*nbase states.
// One workgroup per read, one lane per state, log-space chains in the reference's order of operations (libm
// expf/log1pf/tanhf are ocml here, <= 1-2 ulp apart).  These are Tb-step dependent chains; forward and backward
// posteriors run on two concurrent waves.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "ffhip_internal.hpp"
#include "ffhip_math.hpp"

namespace ffhip {

namespace {

__device__ __forceinline__ int rle_idx(int base_from, int stay_from, int base_to, int nbase) {      // layers.c:1241-1246
    return base_to * 2 * nbase + base_from + (stay_from ? nbase : 0);
}
__device__ __forceinline__ double lse64(double x, double y) { return fmax(x, y) + log1p(exp(-fabs(x - y))); }
__device__ __forceinline__ float softplus_ref(float x) { return log1pf(expf(-fabsf(x))) + ((x >= 0.0f) ? x : 0.f); }

// rows of the head output after the affine map (k_head writes W^T h + b untouched in mode 1)
__global__ void __launch_bounds__(256)
k_rle_activate(float *__restrict__ param, int TbS, int nbase, int P, int Ps, float temperature) {
    // a workgroup takes 8 consecutive blocks of one read (grid: blocks / 8, read) and walks their rows BY KIND -- the 2 nbase^2 transition rows (tanh) of the 8 blocks,
    // then their 2 nbase shape / scale rows (softplus) -- so that a wave evaluates one of the two functions, not both under a mask; 32-bit indices (round 5: the
    // flat 64-bit index with its `% Ps` and the mixed rows cost 120 us per 256 x 800 blocks, on the decode chain of the run-length model)
    const int blk0 = (int)blockIdx.x * 8, nt = P - 2 * nbase, ns = 2 * nbase;
    float *base = param + ((size_t)blockIdx.y * TbS + blk0) * Ps;
    for (int j = (int)threadIdx.x; j < 8 * nt; j += 256) {
        const int bl = j / nt, pp = ns + j % nt;
        if (blk0 + bl < TbS) { float *v = base + bl * Ps + pp; *v = 5.0f * tanhf(*v) / temperature; }
    }
    for (int j = (int)threadIdx.x; j < 8 * ns; j += 256) {
        const int bl = j / ns, pp = j % ns;
        if (blk0 + bl < TbS) { float *v = base + bl * Ps + pp; *v = (pp < nbase ? 1.0f : 1e-8f) + softplus_ref(*v); }
    }
}

// runlengthV2_partition_function: one wave per read, lane = state
__global__ void __launch_bounds__(64)
k_rle_partition(const float *__restrict__ param, int TbS, int nbase, int Ps, double *__restrict__ logz, const int *__restrict__ tbs) {
    __shared__ double st[2][kMaxState];
    __shared__ double cand[2 * 8 * 8], cmax[8];          // nbase <= 8
    const int lane = threadIdx.x, ns = 2 * nbase;
    const float *C = param + (size_t)blockIdx.x * TbS * Ps + ns;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    if (lane < ns) st[0][lane] = 0.0;
    __syncthreads();
    // The chain is serial by definition (pairwise logsumexp in the reference's order); what it must not do is wait for
    // memory: the transition rows of the next kDepth blocks are in flight (clamped, branch-free loads) and each block's row
    // is handed to the state lanes through LDS.
    __shared__ float srow[64];
    constexpr int kDepth = 8;
    const int nrow = 2 * nbase * nbase, lane_c = lane < nrow ? lane : nrow - 1;
    float ring[kDepth];
    auto fetch = [&](int c) { return C[(size_t)(c < Tb ? c : Tb - 1) * Ps + lane_c]; };
#pragma unroll
    for (int k = 0; k < kDepth; k++) ring[k] = fetch(k);
    int cur = 0;
    for (int c0 = 0; c0 < Tb; c0 += kDepth) {
        float curv[kDepth];
#pragma unroll
        for (int k = 0; k < kDepth; k++) curv[k] = ring[k];
#pragma unroll
        for (int k = 0; k < kDepth; k++) ring[k] = fetch(c0 + kDepth + k);
#pragma unroll
        for (int k = 0; k < kDepth; k++) {
            if (c0 + k >= Tb) break;
            srow[lane] = curv[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const float *S = srow;
            const double *prev = st[cur];
            // layers.c:1271-1290.  A move state's value is the log of a sum of 2 (nbase - 1) exponentials, which the reference folds
            // with pairwise fp64 logsumexp -- ten dependent exp / log1p per block.  Here: all candidates side by side (lanes ns..),
            // their exact maximum, ONE exp per candidate lane, the sum in the reference's order, one log: the same number up to
            // fp64 rounding of the association (1e-16 relative; the result is rounded to fp32 after / nblock).  The stay states
            // keep the reference's FLOAT logsumexpf.
            const int ncand = 2 * nbase * nbase;
            for (int pi = lane - ns; pi >= 0 && pi < ncand; pi += 64 - ns) {
                const int b1 = pi / ns, r = pi % ns, b2 = r >> 1, sf = r & 1;           // candidate order within b1: b2 ascending, move then stay
                cand[pi] = (b1 != b2) ? prev[b2 + sf * nbase] + (double)S[rle_idx(b2, sf, b1, nbase)] : -HUGE_VAL;
            }
            double v = 0.0;
            if (lane >= nbase && lane < ns) {
                const int b = lane - nbase;
                const float x = (float)(prev[b] + (double)S[rle_idx(b, 0, b, nbase)]);
                const float y = (float)(prev[b + nbase] + (double)S[rle_idx(b, 1, b, nbase)]);
                v = (double)logsumexpf_ref(x, y);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            double m = -HUGE_VAL;
            if (lane < nbase) {
                for (int r = 0; r < ns; r++) m = fmax(m, cand[lane * ns + r]);
                cmax[lane] = m;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            for (int pi = lane - ns; pi >= 0 && pi < ncand; pi += 64 - ns) cand[pi] = exp(cand[pi] - cmax[pi / ns]);      // exp(-inf) = 0 for the b2 = b1 slots
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane < nbase) {
                double e = 0.0;
                for (int r = 0; r < ns; r++) e += cand[lane * ns + r];
                v = m + log(e);
            }
            if (lane < ns) st[cur ^ 1][lane] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            cur ^= 1;
        }
    }
    if (lane == 0) {
        double z = st[cur][0];
        for (int s = 1; s < ns; s++) z = lse64(z, st[cur][s]);
        logz[blockIdx.x] = z;
    }
}

// rows [2*nbase, P) -= (float)(logZ / Tb)   (layers.c:1349-1356: `const float logZ = partition / (float)nc`)
__global__ void __launch_bounds__(256)
k_rle_sub(float *__restrict__ param, const double *__restrict__ logz, int TbS, int nbase, int P, int Ps, const int *__restrict__ tbs) {
    const int blk = (int)blockIdx.x * 4 + (int)threadIdx.y, r = (int)blockIdx.y;      // (as k_rle_activate)
    const int Tb = tbs ? tbs[r] : TbS;
    if (blk >= Tb) return;
    const float z = (float)(logz[r] / (double)(float)Tb);
    for (int p = 2 * nbase + (int)threadIdx.x; p < P; p += 64) param[((size_t)r * TbS + blk) * Ps + p] -= z;
}

// transpost_crf_runlength: wave 0 forward, wave 1 backward, then one block per thread
__global__ void __launch_bounds__(256)
k_rle_transpost(const float *__restrict__ param, float *__restrict__ post, float *__restrict__ fwdbuf, float *__restrict__ bwdbuf,
                int TbS, int nbase, int P, int Ps, const int *__restrict__ tbs) {
    __shared__ float fs[2][kMaxState], bs[2][kMaxState];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ns = 2 * nbase;
    const float *T = param + (size_t)blockIdx.x * TbS * Ps;
    float *Pp = post + (size_t)blockIdx.x * TbS * Ps;
    float *F = fwdbuf + (size_t)blockIdx.x * (TbS + 1) * kMaxState;
    float *Bw = bwdbuf + (size_t)blockIdx.x * (TbS + 1) * kMaxState;
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    // As in the flip-flop posterior: transition rows prefetched kDepth blocks ahead and handed over through LDS, forward /
    // backward vectors staged in LDS and written out in rows of 64 blocks -- no global load or store on the per-block chains.
    __shared__ float srow[2][64];
    __shared__ float inner[64];            // forward recursion: lse(stay, move) of every (destination, source) pair of a block
    __shared__ float stage[2][64][kMaxState];
    constexpr int kDepth = 8;
    const int nrow = 2 * nbase * nbase, lane_c = lane < nrow ? lane : nrow - 1;
#define RLE_WAVE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
    auto flush = [&](int w, float *dst0, long long dstep, int cnt) {      // row r of the stage -> dst0 + r*dstep (kMaxState floats each)
        RLE_WAVE_SYNC();
        if (lane < cnt) {
            const float4 *src4 = (const float4 *)&stage[w][lane][0];
            float4 *dst = (float4 *)(dst0 + (long long)lane * dstep);
            dst[0] = src4[0]; dst[1] = src4[1]; dst[2] = src4[2]; dst[3] = src4[3];
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (wave == 0) {
        if (lane < ns) { fs[0][lane] = 0.0f; F[lane] = 0.0f; }
        __builtin_amdgcn_wave_barrier();
        float ring[kDepth];
        auto fetch = [&](int blk) { return T[(size_t)(blk < Tb ? blk : Tb - 1) * Ps + ns + lane_c]; };
#pragma unroll
        for (int k = 0; k < kDepth; k++) ring[k] = fetch(k);
        int cur = 0;
        for (int b0 = 0; b0 < Tb; b0 += kDepth) {
            float curv[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) curv[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kDepth; k++) ring[k] = fetch(b0 + kDepth + k);
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                const int blk = b0 + k;
                if (blk >= Tb) break;
                srow[0][lane] = curv[k];
                RLE_WAVE_SYNC();
                const float *S = srow[0];
                const float *prev = fs[cur];
                // decode.c:1064-1083.  The reference chains, per move state b1, lse(v, lse(stay_b2, move_b2)) over b2 != b1 in ascending b2.
                // The INNER logsumexp of every (b1, b2) pair is independent of the chain: lanes 16.. compute them side by side, the
                // state lanes then run the outer chain in the reference's order on the same values (bit-identical; the first outer
                // step, lse(-inf, x) = x exactly, is a copy) -- three logsumexp latencies per block instead of six.
                for (int pi = lane - 16; pi >= 0 && pi < nbase * nbase; pi += 48) {       // (one pass for nbase <= 6)
                    const int b1 = pi / nbase, b2 = pi % nbase;
                    if (b1 != b2) {
                        const float stay_score = prev[b2 + nbase] + S[rle_idx(b2, 1, b1, nbase)];
                        const float move_score = prev[b2] + S[rle_idx(b2, 0, b1, nbase)];
                        inner[b1 * nbase + b2] = logsumexpf_ref(stay_score, move_score);
                    }
                }
                float v = 0.0f;
                if (lane >= nbase && lane < ns) {
                    const int b = lane - nbase;
                    const float stay_score = prev[b + nbase] + S[rle_idx(b, 1, b, nbase)];
                    const float move_score = prev[b] + S[rle_idx(b, 0, b, nbase)];
                    v = logsumexpf_ref(stay_score, move_score);
                }
                RLE_WAVE_SYNC();
                if (lane < nbase) {
                    const int b1 = lane;
                    bool first = true;
                    for (int b2 = 0; b2 < nbase; b2++) {
                        if (b1 == b2) continue;
                        const float in = inner[b1 * nbase + b2];
                        v = first ? in : logsumexpf_ref(v, in);
                        first = false;
                    }
                    if (first) v = -HUGE_VALF;
                }
                if (lane < ns) { fs[cur ^ 1][lane] = v; stage[0][blk & 63][lane] = v; }      // fwd[blk + 1]
                if ((blk & 63) == 63 || blk == Tb - 1) flush(0, F + (size_t)((blk & ~63) + 1) * kMaxState, kMaxState, (blk & 63) + 1);
                else RLE_WAVE_SYNC();
                cur ^= 1;
            }
        }
    } else if (wave == 1) {
        if (lane < ns) bs[0][lane] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        float ring[kDepth];
        auto fetch = [&](int blk) { return T[(size_t)(blk > 0 ? blk : 0) * Ps + ns + lane_c]; };      // blk counts down
#pragma unroll
        for (int k = 0; k < kDepth; k++) ring[k] = fetch(Tb - 1 - k);
        int cur = 0;
        for (int j0 = 0; j0 < Tb; j0 += kDepth) {                                                    // j = Tb - blk
            float curv[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) curv[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kDepth; k++) ring[k] = fetch(Tb - 1 - (j0 + kDepth + k));
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                const int j = j0 + k;
                if (j >= Tb) break;
                srow[1][lane] = curv[k];
                const float *prev = bs[cur];
                if (lane < ns) stage[1][j & 63][lane] = prev[lane];       // bwd[blk], blk = Tb - j: the vector that meets block blk-1's transitions
                if ((j & 63) == 63 || j == Tb - 1) flush(1, Bw + (size_t)(Tb - (j & ~63)) * kMaxState, -(long long)kMaxState, (j & 63) + 1);
                else RLE_WAVE_SYNC();
                const float *S = srow[1];
                float v = 0.0f;
                if (lane < ns) {
                    // source state (b1, stay = lane >= nbase): chain over its moves to b2 != b1 in ascending b2, then its stay exit
                    // (decode.c:1085-1100; the first step, lse(-inf, x) = x exactly, is a copy)
                    const int b1 = lane < nbase ? lane : lane - nbase, st1 = lane < nbase ? 0 : 1;
                    bool first = true;
                    for (int b2 = 0; b2 < nbase; b2++) {
                        if (b1 == b2) continue;
                        const float c = prev[b2] + S[rle_idx(b1, st1, b2, nbase)];
                        v = first ? c : logsumexpf_ref(v, c);
                        first = false;
                    }
                    const float cs = prev[b1 + nbase] + S[rle_idx(b1, st1, b1, nbase)];
                    v = first ? cs : logsumexpf_ref(v, cs);
                }
                if (lane < ns) bs[cur ^ 1][lane] = v;
                RLE_WAVE_SYNC();
                cur ^= 1;
            }
        }
    }
#undef RLE_WAVE_SYNC
    __syncthreads();
    for (int blk = threadIdx.x; blk < Tb; blk += 256) {
        const float *x = T + (size_t)blk * Ps;
        float *o = Pp + (size_t)blk * Ps;
        const float *f = F + (size_t)blk * kMaxState, *bb = Bw + (size_t)(blk + 1) * kMaxState;
        for (int p = 0; p < ns; p++) o[p] = x[p];                                   // shape and scale rows, decode.c:1131-1135
        const float *S = x + ns;
        float *Q = o + ns;
        for (int b1 = 0; b1 < nbase; b1++)
            for (int b2 = 0; b2 < nbase; b2++) {
                if (b1 == b2) continue;
                const int mi = rle_idx(b1, 0, b2, nbase), si = rle_idx(b1, 1, b2, nbase);
                Q[mi] = f[b1] + bb[b2] + S[mi];                                     // :1106
                Q[si] = f[b1 + nbase] + bb[b2] + S[si];                             // :1110
            }
        for (int b = 0; b < nbase; b++) {
            const int i0 = rle_idx(b, 0, b, nbase), i1 = rle_idx(b, 1, b, nbase);
            Q[i0] = f[b] + S[i0] + bb[b + nbase];                                   // :1118
            Q[i1] = f[b + nbase] + S[i1] + bb[b + nbase];                           // :1124
        }
    }
}

// decode_crf_runlength: Viterbi, traceback bytes in HBM, one wave per read
__global__ void __launch_bounds__(64)
k_rle_viterbi(const float *__restrict__ param, uint8_t *__restrict__ tbbuf, int *__restrict__ path, float *__restrict__ qpath,
              float *__restrict__ score_out, int TbS, int nbase, int Ps, const int *__restrict__ tbs) {
    __shared__ float vs[2][kMaxState];
    const int lane = threadIdx.x, ns = 2 * nbase;
    const float *T = param + (size_t)blockIdx.x * TbS * Ps;
    uint8_t *tb = tbbuf + (size_t)blockIdx.x * TbS * kMaxState;
    int *pth = path + (size_t)blockIdx.x * (TbS + 1);
    float *qp = qpath + (size_t)blockIdx.x * (TbS + 1);
    const int Tb = tbs ? tbs[blockIdx.x] : TbS;          // this read's blocks; TbS is the batch's stride
    if (Tb <= 0) return;                                 // an empty slot: nothing to read, and Tb - 1 must not index
    if (lane < ns) vs[0][lane] = 0.0f;
    __syncthreads();
    // No memory on the chain: transition rows prefetched kDepth blocks ahead and handed over through LDS, traceback bytes kept in
    // LDS (a chunk of kRleChunk blocks; earlier chunks of longer reads are flushed to HBM and read back for the traceback).
    constexpr int kRleChunk = 2048, kDepth = 8;
    __shared__ uint8_t tb_lds[kRleChunk * kMaxState];
    __shared__ float srow[64];
    const int nrow = 2 * nbase * nbase, lane_c = lane < nrow ? lane : nrow - 1;
    float ring[kDepth];
    auto fetch = [&](int blk) { return T[(size_t)(blk < Tb ? blk : Tb - 1) * Ps + ns + lane_c]; };
#pragma unroll
    for (int k = 0; k < kDepth; k++) ring[k] = fetch(k);
    int cur = 0;
    const int nchunk = (Tb + kRleChunk - 1) / kRleChunk;
    for (int c = 0; c < nchunk; c++) {
        const int c0 = c * kRleChunk, n = min(kRleChunk, Tb - c0);
        for (int b0 = 0; b0 < n; b0 += kDepth) {
            float curv[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; k++) curv[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kDepth; k++) ring[k] = fetch(c0 + b0 + kDepth + k);
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                if (b0 + k >= n) break;
                srow[lane] = curv[k];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const float *S = srow;
                const float *prev = vs[cur];
                float v = -HUGE_VALF;
                int arg = 0;
                if (lane < nbase) {
                    const int b1 = lane;
                    for (int b2 = 0; b2 < nbase; b2++) {
                        if (b1 == b2) continue;
                        const float move_score = prev[b2] + S[rle_idx(b2, 0, b1, nbase)];
                        if (move_score > v) { v = move_score; arg = b2; }
                        const float stay_score = prev[b2 + nbase] + S[rle_idx(b2, 1, b1, nbase)];
                        if (stay_score > v) { v = stay_score; arg = b2 + nbase; }
                    }
                } else if (lane < ns) {
                    const int b = lane - nbase;
                    const float stay_score = prev[b + nbase] + S[rle_idx(b, 1, b, nbase)];
                    const float move_score = prev[b] + S[rle_idx(b, 0, b, nbase)];
                    if (stay_score > move_score) { v = stay_score; arg = b + nbase; }
                    else { v = move_score; arg = b; }
                }
                if (lane < ns) { vs[cur ^ 1][lane] = v; tb_lds[(b0 + k) * kMaxState + lane] = (uint8_t)arg; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                cur ^= 1;
            }
        }
        if (c + 1 < nchunk) {             // a longer read: this chunk's bytes leave LDS (kRleChunk is a multiple of kDepth)
            __syncthreads();
            for (int i = lane; i < n * (kMaxState / 4); i += 64)
                ((uint32_t *)(tb + (size_t)c0 * kMaxState))[i] = ((const uint32_t *)tb_lds)[i];
            __syncthreads();
        }
    }
    __syncthreads();
    int last = 0;
    for (int st = 1; st < ns; st++) if (vs[cur][st] > vs[cur][last]) last = st;     // argmaxf: first maximum
    if (lane == 0) { score_out[blockIdx.x] = vs[cur][last]; pth[Tb] = 0; qp[Tb] = NAN; }
    // traceback, last chunk first (it is still in LDS)
    __shared__ int pl[kRleChunk];
    for (int c = nchunk - 1; c >= 0; c--) {
        const int c0 = c * kRleChunk, n = min(kRleChunk, Tb - c0);
        if (c != nchunk - 1) {
            __syncthreads();
            for (int i = lane; i < n * (kMaxState / 4); i += 64)
                ((uint32_t *)tb_lds)[i] = ((const uint32_t *)(tb + (size_t)c0 * kMaxState))[i];
        }
        __syncthreads();
        if (lane == 0) {
            for (int i = n; i > 0; i--) {
                const int state = tb_lds[(i - 1) * kMaxState + last];
                pl[i - 1] = last;
                last = state;
            }
        }
        __syncthreads();
        for (int i = lane; i < n; i += 64) { pth[c0 + i] = pl[i]; qp[c0 + i] = NAN; }
        // `last` for the next (earlier) chunk lives in lane 0: broadcast
        last = __shfl(last, 0);
    }
}


// ---- decoders of the FIRST run-length head (globalnorm_runlength, layers.c:1197-1228: nparam = 4 nbase rows per block -- two run parameters,
// a move weight and a stay weight per base).  No registry entry reaches them (networks.c:86-105); they complete decode.h's surface as
// single-matrix operators, one workgroup a matrix, the param rows prefetched kV1Depth blocks ahead of the chains.
constexpr int kV1Base = 8, kV1Depth = 8, kV1Chunk = 4096;

// decode_runlength (decode.c:694-767).  Per block every state takes the best of the OTHER bases' scores -- the block's maximum, or for the
// maximum's own base the runner-up ("prev[idx] = -HUGE_VAL; idx2 = argmaxf(prev)") -- plus its move weight, unless staying (prev[b] + stay
// weight, strictly greater) wins.  path[blk] = the base that is entered in block blk, -1 while staying.  One wave: every lane scans the
// nbase previous scores literally (first maximum wins, as argmaxf does), lane b then owns state b.
__global__ void __launch_bounds__(64)
k_rl1_viterbi(const float *__restrict__ param, int nblk, int nbase, int Ps, uint8_t *__restrict__ tbbuf, int *__restrict__ path, float *__restrict__ score_out) {
    __shared__ float vs[2][kV1Base];
    __shared__ float srow[2 * kV1Base];
    __shared__ uint8_t tb_lds[kV1Chunk * kV1Base];
    const int lane = threadIdx.x, nrow = 2 * nbase, lane_c = lane < nrow ? lane : nrow - 1;
    if (lane < nbase) { vs[0][lane] = 0.0f; vs[1][lane] = 0.0f; }        // calloc'ed `mem` (decode.c:699)
    __syncthreads();
    float ring[kV1Depth];
    auto fetch = [&](int blk) { return param[(size_t)(blk < nblk ? blk : nblk - 1) * Ps + 2 * nbase + lane_c]; };      // move weights, then stay weights
#pragma unroll
    for (int k = 0; k < kV1Depth; k++) ring[k] = fetch(k);
    int cur = 0;
    const int nchunk = (nblk + kV1Chunk - 1) / kV1Chunk;
    for (int c = 0; c < nchunk; c++) {
        const int c0 = c * kV1Chunk, n = min(kV1Chunk, nblk - c0);
        for (int b0 = 0; b0 < n; b0 += kV1Depth) {
            float curv[kV1Depth];
#pragma unroll
            for (int k = 0; k < kV1Depth; k++) curv[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kV1Depth; k++) ring[k] = fetch(c0 + b0 + kV1Depth + k);
#pragma unroll
            for (int k = 0; k < kV1Depth; k++) {
                if (b0 + k >= n) break;
                if (lane < nrow) srow[lane] = curv[k];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const float *prev = vs[cur];
                int idx = 0;                                   // argmaxf (util.c:17-31): the first maximum
                float vmax = prev[0];
                for (int i = 1; i < nbase; i++) if (prev[i] > vmax) { vmax = prev[i]; idx = i; }
                int idx2 = 0;                                  // ... and again with prev[idx] = -HUGE_VAL
                float v2 = (0 == idx) ? -HUGE_VALF : prev[0];
                for (int i = 1; i < nbase; i++) {
                    const float x = (i == idx) ? -HUGE_VALF : prev[i];
                    if (x > v2) { v2 = x; idx2 = i; }
                }
                if (lane < nbase) {
                    const int b = lane;
                    float v = (b == idx) ? prev[idx2] : vmax;     // (decode.c:731: curr[idx] = prev[idx2], read after prev[idx] was restored)
                    int arg = (b == idx) ? idx2 : idx;
                    v += srow[b];
                    const float stay_score = prev[b] + srow[nbase + b];
                    if (stay_score > v) { v = stay_score; arg = b + nbase; }
                    vs[cur ^ 1][b] = v;
                    tb_lds[(b0 + k) * kV1Base + b] = (uint8_t)arg;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                cur ^= 1;
            }
        }
        if (c + 1 < nchunk) {             // a longer matrix: this chunk's bytes leave LDS
            __syncthreads();
            for (int i = lane; i < n * (kV1Base / 4); i += 64) ((uint32_t *)(tbbuf + (size_t)c0 * kV1Base))[i] = ((const uint32_t *)tb_lds)[i];
            __syncthreads();
        }
    }
    __syncthreads();
    int last = 0;
    for (int st = 1; st < nbase; st++) if (vs[cur][st] > vs[cur][last]) last = st;
    if (lane == 0) score_out[0] = vs[cur][last];
    // traceback, last chunk first (it is still in LDS): lane 0 walks, the path leaves in rows of 64 blocks
    __shared__ int pl[kV1Chunk];
    for (int c = nchunk - 1; c >= 0; c--) {
        const int c0 = c * kV1Chunk, n = min(kV1Chunk, nblk - c0);
        if (c != nchunk - 1) {
            __syncthreads();
            for (int i = lane; i < n * (kV1Base / 4); i += 64) ((uint32_t *)tb_lds)[i] = ((const uint32_t *)(tbbuf + (size_t)c0 * kV1Base))[i];
            __syncthreads();
        }
        if (lane == 0) {
            for (int i = n - 1; i >= 0; i--) {
                const int state = tb_lds[i * kV1Base + last];
                if (state < nbase) { pl[i] = last; last = state; }      // a base was entered here (decode.c:757-760)
                else pl[i] = -1;
            }
        }
        last = __shfl(last, 0);
        __syncthreads();
        for (int i = lane; i < n; i += 64) path[c0 + i] = pl[i];
    }
}

// posterior_runlength (decode.c:793-892): wave 0 runs the forward chain, wave 1 the backward one, then one block per thread.  The chains
// add their terms in the reference's order (the other bases ascending, then the stay), so the values are the oracle's up to the device's
// expf / log1pf.  post is [nparam x (nblk + 1)], zero outside the move and stay rows of blocks 0 .. nblk - 1 (make_flappie_matrix clears).
__global__ void __launch_bounds__(256)
k_rl1_posterior(const float *__restrict__ param, float *__restrict__ post, float *__restrict__ fwdbuf, float *__restrict__ bwdbuf, int nblk, int nbase, int Ps) {
    __shared__ float fs[2][kV1Base], bs[2][kV1Base];
    __shared__ float srow[2][2 * kV1Base];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nrow = 2 * nbase, lane_c = lane < nrow ? lane : nrow - 1;
#define RL1_WAVE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
    if (wave == 0) {
        if (lane < nbase) { fs[0][lane] = 0.0f; fwdbuf[lane] = 0.0f; }      // fwd column 0 (decode.c:803: zeros)
        float ring[kV1Depth];
        auto fetch = [&](int blk) { return param[(size_t)(blk < nblk ? blk : nblk - 1) * Ps + 2 * nbase + lane_c]; };
#pragma unroll
        for (int k = 0; k < kV1Depth; k++) ring[k] = fetch(k);
        int cur = 0;
        for (int b0 = 0; b0 < nblk; b0 += kV1Depth) {
            float curv[kV1Depth];
#pragma unroll
            for (int k = 0; k < kV1Depth; k++) curv[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kV1Depth; k++) ring[k] = fetch(b0 + kV1Depth + k);
#pragma unroll
            for (int k = 0; k < kV1Depth; k++) {
                const int blk = b0 + k;
                if (blk >= nblk) break;
                if (lane < nrow) srow[0][lane] = curv[k];
                RL1_WAVE_SYNC();
                if (lane < nbase) {
                    const float *prev = fs[cur];
                    float v = -HUGE_VALF;                                     // decode.c:822-830
                    for (int b2 = 0; b2 < nbase; b2++) if (b2 != lane) v = logsumexpf_ref(v, prev[b2]);
                    v += srow[0][lane];
                    v = logsumexpf_ref(v, prev[lane] + srow[0][nbase + lane]);      // :833
                    fs[cur ^ 1][lane] = v;
                    fwdbuf[(size_t)(blk + 1) * kV1Base + lane] = v;
                }
                RL1_WAVE_SYNC();
                cur ^= 1;
            }
        }
    } else if (wave == 1) {
        if (lane < nbase) { bs[0][lane] = 0.0f; bwdbuf[(size_t)nblk * kV1Base + lane] = 0.0f; }      // calloc'ed `mem` (decode.c:805)
        float ring[kV1Depth];
        auto fetch = [&](int blk) { return param[(size_t)(blk > 0 ? blk : 0) * Ps + 2 * nbase + lane_c]; };      // blk counts down
#pragma unroll
        for (int k = 0; k < kV1Depth; k++) ring[k] = fetch(nblk - 1 - k);
        int cur = 0;
        for (int j0 = 0; j0 < nblk; j0 += kV1Depth) {
            float curv[kV1Depth];
#pragma unroll
            for (int k = 0; k < kV1Depth; k++) curv[k] = ring[k];
#pragma unroll
            for (int k = 0; k < kV1Depth; k++) ring[k] = fetch(nblk - 1 - (j0 + kV1Depth + k));
#pragma unroll
            for (int k = 0; k < kV1Depth; k++) {
                const int j = j0 + k;                  // block nblk - 1 - j
                if (j >= nblk) break;
                if (lane < nrow) srow[1][lane] = curv[k];
                RL1_WAVE_SYNC();
                if (lane < nbase) {
                    const float *prev = bs[cur];
                    float v = -HUGE_VALF;                                     // decode.c:853-861
                    for (int b2 = 0; b2 < nbase; b2++) if (b2 != lane) v = logsumexpf_ref(v, prev[b2] + srow[1][b2]);
                    v = logsumexpf_ref(v, prev[lane] + srow[1][nbase + lane]);      // :866
                    bs[cur ^ 1][lane] = v;
                    bwdbuf[(size_t)(nblk - 1 - j) * kV1Base + lane] = v;      // the vector in front of block nblk - 1 - j
                }
                RL1_WAVE_SYNC();
                cur ^= 1;
            }
        }
    }
#undef RL1_WAVE_SYNC
    __syncthreads();
    for (int blk = threadIdx.x; blk < nblk; blk += 256) {
        const float *x = param + (size_t)blk * Ps + 2 * nbase;
        float *o = post + (size_t)blk * Ps + 2 * nbase;
        const float *f = fwdbuf + (size_t)blk * kV1Base, *bb = bwdbuf + (size_t)(blk + 1) * kV1Base;      // `prev` of decode.c:839-850 at this block
        for (int b1 = 0; b1 < nbase; b1++) {
            float v = -HUGE_VALF;                                             // :855, :859
            for (int b2 = 0; b2 < nbase; b2++) if (b2 != b1) v = logsumexpf_ref(v, f[b2]);
            o[b1] = v + (bb[b1] + x[b1]);                                     // :862
            o[nbase + b1] = f[b1] + x[nbase + b1] + bb[b1];                   // :867
        }
    }
}

// runlengths_mean (decode.c:576-603) with dwmean (:552-562): per entered base 1 + round(sum_{i=1..100} exp(-(i / scale)^shape)) from that base's
// shape and scale rows.  The reference calls libm's powf / expf; here each is evaluated in double and rounded to float -- the same float
// except where libm itself is not correctly rounded (its stated bound is 0.52 ulp), which can only move a sum that sits on a half-integer.
__global__ void __launch_bounds__(256)
k_rl1_mean(const float *__restrict__ param, const int *__restrict__ path, int *__restrict__ runlength, unsigned long long *__restrict__ seqlen, int nblk, int nbase, int Ps) {
    const int blk = blockIdx.x * 256 + threadIdx.x;
    int rl = 0;
    if (blk < nblk && path[blk] >= 0) {
        const size_t off = (size_t)blk * Ps + (size_t)path[blk];
        const float shape = param[off], scale = param[off + nbase];
        float m = 0.0f;
        for (int i = 1; i <= 100; i++) {
            const float t = (float)i / scale;
            const float p = (float)pow((double)t, (double)shape);
            m += (float)exp(-(double)p);
        }
        rl = (int)(1.0f + roundf(m));
    }
    if (blk < nblk) runlength[blk] = rl;
    unsigned long long tot = (unsigned long long)rl;
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(seqlen, tot);
}

}  // namespace

void launch_rle_head_finish(hipStream_t s, float *param, double *logz, int nread, int Tb, int nbase, int Ps, float temperature, const int *tbs) {
    const int P = 2 * nbase * (nbase + 1);
    hipLaunchKernelGGL(k_rle_activate, dim3((unsigned)((Tb + 7) / 8), (unsigned)nread), dim3(256), 0, s, param, Tb, nbase, P, Ps, temperature);
    if (nbase == 4 && Ps == 40 && !dbg("decode_r2")) launch_rle_partition8x(s, param, logz, nread, Tb, tbs);      // ffhip_decode.hip
    else hipLaunchKernelGGL(k_rle_partition, dim3(nread), dim3(64), 0, s, param, Tb, nbase, Ps, logz, tbs);
    hipLaunchKernelGGL(k_rle_sub, dim3((unsigned)((Tb + 3) / 4), (unsigned)nread), dim3(64, 4), 0, s, param, logz, Tb, nbase, P, Ps, tbs);
}

void launch_rle_partition(hipStream_t s, const float *param, double *logz, int nread, int Tb, int nbase, int Ps, const int *tbs) {
    if (nbase == 4 && Ps == 40 && !dbg("decode_r2")) launch_rle_partition8x(s, param, logz, nread, Tb, tbs);      // ffhip_decode.hip
    else hipLaunchKernelGGL(k_rle_partition, dim3(nread), dim3(64), 0, s, param, Tb, nbase, Ps, logz, tbs);
}

void launch_rle_transpost(hipStream_t s, const float *param, float *post, float *fwd, int nread, int Tb, int nbase, int Ps, const int *tbs) {
    const int P = 2 * nbase * (nbase + 1);
    hipLaunchKernelGGL(k_rle_transpost, dim3(nread), dim3(256), 0, s, param, post, fwd, fwd + (size_t)nread * (Tb + 1) * kMaxState, Tb, nbase, P, Ps, tbs);
}

void launch_rle_viterbi(hipStream_t s, const float *param, uint8_t *tb, int *path, float *qpath, float *score, int nread, int Tb, int nbase, int Ps, const int *tbs) {
    if (nbase == 4 && Ps == 40 && !dbg("decode_r2")) { launch_rle_viterbi8x(s, param, tb, path, qpath, score, nread, Tb, tbs); return; }      // ffhip_decode.hip
    hipLaunchKernelGGL(k_rle_viterbi, dim3(nread), dim3(64), 0, s, param, tb, path, qpath, score, Tb, nbase, Ps, tbs);
}

// first-generation run-length decoders on ONE matrix (nparam = 4 nbase rows, nbase <= 8)
void launch_rl1_viterbi(hipStream_t s, const float *param, uint8_t *tb, int *path, float *score, int nblk, int nbase, int Ps) {
    hipLaunchKernelGGL(k_rl1_viterbi, dim3(1), dim3(64), 0, s, param, nblk, nbase, Ps, tb, path, score);
}
void launch_rl1_posterior(hipStream_t s, const float *param, float *post, float *fwd, float *bwd, int nblk, int nbase, int Ps) {
    hipLaunchKernelGGL(k_rl1_posterior, dim3(1), dim3(256), 0, s, param, post, fwd, bwd, nblk, nbase, Ps);
}
void launch_rl1_mean(hipStream_t s, const float *param, const int *path, int *runlength, unsigned long long *seqlen, int nblk, int nbase, int Ps) {
    hipLaunchKernelGGL(k_rl1_mean, dim3((nblk + 255) / 256), dim3(256), 0, s, param, path, runlength, seqlen, nblk, nbase, Ps);
}

}  // namespace ffhip
