// ffhip_internal.hpp -- shared declarations between the HIP kernels and the host engine.
// gfx950 (MI355X) only.  Not part of the public boundary (include/ffhip.h is).
#pragma once
#include "ffhip_split.hpp"
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ffhip {

// Development switches live in ONE environment variable: FFHIP_DEBUG=token[,token=value ...] (INTEGRATION.md section 6 lists them).
// dbg("token"): nullptr when the token is absent, otherwise its value ("" for a bare token) -- what getenv gave when every switch
// had a variable of its own (rounds 1-4: 41 of them).  Looked up per call: tests flip switches between runs of one process.
const char *dbg(const char *token);

// ---------------------------------------------------------------------------------------------
// Data layouts in HBM (all fp32 unless noted).  B16 = ceil(nread/16) read tiles, Bp = 16*B16.
//
//  sample-major  S[r][pad + t][F]        conv inputs; `pad` zero rows before and after the T real
//                                        rows, read stride rs floats.  A window of `winlen` samples
//                                        is one contiguous vector of winlen*F floats.
//  tile-interleaved  A[t][rt][k/4][r16][k%4]
//                                        recurrent-stack activations (H features).  For one
//                                        (t, read tile) the 16 reads x 16 consecutive features are
//                                        1 KiB in exactly the lane order of an MFMA 16x16x4 f32
//                                        B-fragment quad: lane l=(kq=l>>4, r=l&15) holds float4
//                                        {k = 16*k16 + 4*kq + 0..3}.
//  D-fragment  X[t][rt][mt][lane][4]     gate pre-activations (Wi x + b).  Rows are permuted
//                                        unit-major/gate-minor (m = 4*u + g), so an MFMA output tile
//                                        (lane l: rows 4*(l>>4)+0..3, column l&15) gives every lane
//                                        the 4 gates of ONE hidden unit of ONE read.
//  weights  W[mt][k16][lane] float4      A-fragment order, lane l=(i=l&15, kq=l>>4) holds
//                                        W[16*mt + i][16*k16 + 4*kq + 0..3].
//  trans/post  T[r][blk][Ps]             Ps = 4*ceil(P/4): byte-identical to the reference's
//                                        flappie_matrix image of one read (column = block).
// ---------------------------------------------------------------------------------------------

constexpr int kSamplePad = 64;     // zero rows either side of a sample-major buffer
constexpr int kMaxState = 16;      // nstate <= 16 (nbase <= 8)
constexpr int kNoWindow = INT32_MIN;
constexpr int kZeroCol = INT32_MIN + 1;   // window-table entry of a column beyond a read's end in a ragged batch

struct SampleBuf {                 // sample-major activation buffer
    float *p;
    int F;                         // features per sample (exact, no padding)
    int T;                         // real samples
    size_t rs;                     // read stride in floats
    __host__ __device__ const float *row(int r, int t) const { return p + (size_t)r * rs + (size_t)(kSamplePad + t) * F; }
};

enum Act { ACT_NONE = 0, ACT_SWISH = 1, ACT_TANH = 2 };

// Packed batches (round 6; ffhip_batch_set_prepared_packed): several reads stand one behind the other in a slot (a row of the batch's buffers), so that a batch of reads
// of any lengths costs what its samples cost.  The convolutions, the layer kernels and the CRF head work on SLOTS (window tables / a live mask say where the reads are);
// everything per READ -- chains, Viterbi, assembly, trace -- takes the read's first row in the buffers of Tb rows a slot (b0) and of Tb + 1 rows a slot (b1).
// Both nullptr: one read a slot, read r is row r.
struct ReadMap {
    const int *b0 = nullptr, *b1 = nullptr;
    int nslot = 0;
    __device__ __forceinline__ size_t row0(int read, int TbS) const { return b0 ? (size_t)b0[read] : (size_t)read * (size_t)TbS; }
    __device__ __forceinline__ size_t row1(int read, int TbS) const { return b1 ? (size_t)b1[read] : (size_t)read * (size_t)(TbS + 1); }
};

// ---- kernel launchers (ffhip_kernels.hip) ----------------------------------------------------
void launch_pack_signal(hipStream_t s, const float *src, size_t ld, SampleBuf dst, int nread);

// VALU convolution for the thin front layers; W dense taps [Fout][winlen][Fin]
void launch_conv_small(hipStream_t s, SampleBuf in, SampleBuf out, const float *W, const float *bias,
                       const int *x0a, const int *x0b, int Bp, int Tout, int winlen, int act, int ldp = 0,     // ldp: entries per read of a per-read window table (0 = shared)
                       const int *tin = nullptr,                                  // stride-1 layer of a ragged batch: per-read input lengths instead of a table
                       int split_exp = -100000,                                   // > -1000 (16 output features): write fp16 slices of value * 2^split_exp for launch_conv_split
                       unsigned *sat = nullptr,                                   // per-read word set to 1 when a value leaves the split format's range (ffhip_split.hpp: clamped there; the engine re-runs such a read on the f32 path)
                       const int *seg = nullptr);                                 // stride-1 layer of a packed batch: [Bp + 1] offsets, then every row's sorted read boundaries {start, end, ...} in columns

// MFMA convolution of the last conv layer: sample-major in, tile-interleaved out [Tout][B16][M/4][16][4]
void launch_conv_mfma(hipStream_t s, SampleBuf in, float *out, const float4 *Wp, const float *bias,
                      const int *x0a, const int *x0b, int B16, int Tout, int M, int K16, int act, int ldp = 0,
                      void *out_split = nullptr, int split_exp = 0, unsigned *sat = nullptr);      // != nullptr: write the split layout of ffhip_rnn_split.hip (values * 2^split_exp) INSTEAD of `out` (M % 128 == 0)

// the same convolution on split operands (16 input features): `in` holds fp16 slices (launch_conv_small with split_exp = kSplitExpX),
// Wp the split weight pack [M/16][ceil(winlen/2)][2][64] x 16 B scaled by 2^(acc_exp - kSplitExpX)
void launch_conv_split(hipStream_t s, SampleBuf in, float *out, const void *Wp, const float *bias, const int *x0a, const int *x0b,
                       int B16, int Tout, int M, int winlen, int act, int ldp, void *out_split, int split_exp, int acc_exp, int lean = 0, unsigned *sat = nullptr);      // lean: the <= 128-VGPR shape

// Xa = Wi^T x + b for every (t, read); in tile-interleaved, out D-fragment order
void launch_inproj(hipStream_t s, const float *in, float *xa, const float4 *Wp, const float *bias,
                   int ntile /*Tb*B16*/, int M /*rows, mult of 16*/, int K16);

// one recurrent step for all reads (launch-per-step path)
void launch_lstm_step(hipStream_t s, const float4 *sWp, const float *xa_t, const float *h_prev, float *h_out,
                      float *cstate, int B16, int H, int first, int t = 0, const int *tbs = nullptr);
void launch_gru_step(hipStream_t s, const float4 *sWp, const float *xa_t, const float *h_prev, float *h_out,
                     int B16, int H, int first, int t = 0, const int *tbs = nullptr);

// persistent recurrent layer (ffhip_rnn_persist.hip): one launch per layer and chunk of read tiles
bool persist_supported(int kind, int H, int ncu);
int persist_max_tiles(int kind, int H, int ncu, int fused);      // read tiles one launch can take (all workgroups co-resident)
bool launch_rnn_persist(hipStream_t s, int kind, const float4 *sWp, const float *xa, float *hout, unsigned *flags,
                        unsigned *abort_word, int Tb, int B16, int H, int rt0, int nrt, int backward, int mode,
                        const int *tbs = nullptr, const int *tbt = nullptr);      // ragged batch: blocks per read / max per tile
size_t persist_flag_words(int H, int nrt);
bool fused_supported(int kind, int H);
bool launch_lstm_fused(hipStream_t s, int kind, const float4 *sWp, const float4 *iWp, const float *bias, const float *xin, float *hout,
                       unsigned *flags, unsigned *abort_word, int Tb, int B16, int H, int rt0, int nrt, int backward, int mode,
                       const int *tbs = nullptr, const int *tbt = nullptr);
int persist_blocks_per_cu(int kind, int H);

// persistent LSTM layer on bf16 MFMAs over three-way split operands (ffhip_rnn_split.hip): fp32-exact products at 2.7x
// the f32 MFMA rate.  Activations in the SPLIT layout A[t][rt][k/32][slice 0..2][lane][8 bf16] (6 bytes per value).
void launch_gather_rows(hipStream_t s, const float *const *src, const int *lens, float *dst, size_t row_stride, int nrow, const long long *dst_off = nullptr);
// packed batches: the strided convolution's window table / the layer kernels' live mask from per-read records (ffhip_kernels.hip)
void launch_pack_conv_table(hipStream_t s, const int4 *reads, int nread, int maxcols, int winlen, int stride, int Tmax, int *x0a, int *x0b, unsigned *overflow);
void launch_pack_live(hipStream_t s, const int4 *reads, int nread, int maxblocks, int B16, unsigned *live);
bool split_supported(int kind, int H);
int split_launch_workgroups(int kind, int H, int nrt, int ncu, int beside);      // workgroups of one launch of nrt read tiles ...
int split_workgroups_per_cu(int kind, int H, int nrt, int ncu, int beside);      // ... and how many of them share a CU
int split_next_launch_tiles(int kind, int H, int remaining, int ncu);          // read tiles the next layer launch of a batch takes
int split_max_tiles(int ncu, int H = 512);                // read tiles (of 16) per launch: 32 workgroups per PAIR of tiles, one per CU (two at H <= 256)
size_t split_flag_words(int nrt);
size_t split_pack_offset(int H);           // 16-byte pieces in front of the gate-major weight pack of the packed GRUmod form
inline size_t split_bytes(size_t ntile, int H) { return ntile * (size_t)H * 32 * kSplitNS; }      // 16 reads x H x 2 B x slices
struct SplitLaunch {          // one batch's share of a paired layer launch
    const void *Wp; const float *bias; const void *xin; void *hout; float *hout_f32; unsigned *flags, *abort_word;
    int Tb, B16, rt0, nrt, backward, mode, scale_exp, fast_gates; const int *tbs, *tbt; unsigned epoch;
    const unsigned *live = nullptr;      // packed batch: bit r of word [t][read tile] = slot r holds a block of a read at step t (ffhip_rnn_split.hip SplitArgs)
};
bool launch_lstm_split_pair(hipStream_t s, int kind, int H, int ncu, const SplitLaunch &p0, const SplitLaunch &p1);
bool launch_lstm_split(hipStream_t s, int kind, const void *Wp, const float *bias, const void *xin, void *hout, float *hout_f32,
                       unsigned *flags, unsigned *abort_word, int Tb, int B16, int H, int rt0, int nrt, int backward, int mode,
                       int scale_exp, int fast_gates, const int *tbs, const int *tbt, int ncu, unsigned epoch, int beside, const unsigned *live = nullptr);      // scale_exp: the exponent S both products carry; live: packed batch (SplitLaunch)
// recurrence-only layer kernel on split operands behind launch_inproj_split (LSTM, H = 256 / 512): xa as from launch_inproj_split
bool rnn_split_supported(int kind, int H);
bool launch_rnn_split(hipStream_t s, const void *Wsplit, const float *xa, void *hout, float *hout_f32, unsigned *flags, unsigned *abort_word,
                      int Tb, int B16, int H, int rt0, int nrt, int backward, int mode, int scale_exp, const int *tbs = nullptr, const int *tbt = nullptr);
// input projection GEMM on split operands: in_split = activations in the split layout, Wp = the split weight pack (its first
// matrix is Wi), xa = D-fragment order like launch_inproj
void launch_inproj_split(hipStream_t s, const void *in_split, float *xa, const void *Wp, const float *bias, int ntile, int H, int scale_exp);
void launch_split_from_f32(hipStream_t s, const float *in, void *out, size_t ntile, int H, int act_exp, unsigned *sat = nullptr, int B16 = 1);      // tile-interleaved fp32 -> split of in * 2^act_exp
void launch_f32_from_split(hipStream_t s, const void *in, float *out, size_t ntile, int H, int act_exp);
void launch_lean_math_check(hipStream_t s, int exponent, int steps, unsigned long long *bad);      // adds the mismatch count to *bad

// the same head on the last layer's split output (ffhip_rnn_split.hip layout), weights as fp16 slices scaled by 2^(acc_exp - kSplitExpH)
void launch_head_split(hipStream_t s, const void *in_split, float *trans, const void *Wsplit, const float *bias,
                       int Tb, int B16, int nread, int P, int Ps, int Hc, float scale, int acc_exp, int raw, double *E = nullptr);
bool head_split_writes_E(int P);       // the head can leave exp(S - block max) for the linear-space chains (drops k_crf_exp)
// head: trans = tanh(W^T h + b) / (temperature/5)
void launch_head(hipStream_t s, const float *in, float *trans, const float4 *Wp, const float *bias,
                 int Tb, int B16, int nread, int P, int Ps, int K16, float scale, int raw = 0);      // raw = 1: W^T h + b only
// CRF partition function (fp64) + subtraction of (float)(logZ/Tb)
// logz: device buffer of nread doubles, receives the fp64 partition function per read; subtract = 0 leaves `trans` untouched
// tbs (optional, here and below): device array of the blocks of each read of a ragged batch; Tb is then the stride
void launch_crf_norm(hipStream_t s, float *trans, int nread, int Tb, int nbase, int Ps, double *logz, int subtract = 1, const int *tbs = nullptr);
// the same in linear space (fp64 scaled forward recursion); E = workspace of nread*Tb*crf_exp_stride(P) doubles,
// R = blocks between power-of-two rescalings (see crf_rescale_interval)
inline int crf_exp_stride(int P) { return (P + 1 + 7) & ~7; }
// per block the spread of alpha grows by at most exp(2*bound) (bound = max |score|) times nstate
inline int crf_rescale_interval(float bound) {
    const float bits = 2.0f * bound * 1.4427f + 4.0f;
    const int r = (int)(900.0f / bits);
    return r < 1 ? 0 : (r > 16 ? 16 : r);           // 0: range too wide for the linear form, use launch_crf_norm
}
void launch_crf_norm_linear(hipStream_t s, float *trans, double *E, int nread, int Tb, int nbase, int Ps, int R,
                            double *logz, int subtract = 1, const int *tbs = nullptr);
// forward/backward transition posteriors, log-normalised per block; fwd = workspace of 2*nread*(Tb+1)*kFwdRowBytes bytes
// E (optional, 8-state models): workspace of nread*Tb*crf_exp_stride(P) doubles, wide: nread ints -- with both the
// recursions run in linear space on exp(score - block max) (ffhip_decode.hip); reads whose scores span more than kFbRange per block keep the log-space kernel
void launch_transpost(hipStream_t s, const float *trans, float *post, float *fwd, int nread, int Tb, int nbase, int Ps, const int *tbs = nullptr,
                      double *E = nullptr, int *wide = nullptr);
constexpr size_t kFwdRowBytes = 10 * sizeof(double);   // per block and direction of the forward / backward workspace: kMaxState floats (log-space kernels) or up to 10 doubles (ffhip_decode.hip)
constexpr float kFbRange = 100.0f;       // max - min of a block's scores the scaled linear-space recursions take (fp64 range, scaling one pair of blocks behind)
// row_off / P_override: the run-length model's 32 transition scores sit behind 8 other rows of its 40-float blocks
void launch_crf_exp(hipStream_t s, const float *trans, double *E, int nread, int Tb, int nbase, int Ps, const int *tbs, int *wide, float limit, int row_off = 0, int P_override = 0);
void launch_rle_partition8x(hipStream_t s, const float *param, double *logz, int nread, int Tb, const int *tbs);
void launch_rle_post8(hipStream_t s, const float *param, float *post, double *E, double *fwd, int nread, int Tb, const int *tbs);
// ffhip_decode.hip: partition function (+ subtraction, flags & 1) and posterior (flags & 2) of 8- or 10-state reads from E in one launch; fwd = 2*nread*(Tb+1)*(2*nbase) doubles
void launch_crf_fb(hipStream_t s, int nbase, const double *E, float *trans, float *post, double *fwd, int nread, int Tb, double *logz, const int *tbs,
                    int flags, const int *wide, ReadMap map = ReadMap());
void launch_viterbi10x(hipStream_t s, const float *score_mat, uint8_t *tb, int *path, float *qpath, float *score, int nread, int Tb, const int *tbs, ReadMap map = ReadMap());
void launch_rle_viterbi8x(hipStream_t s, const float *param, uint8_t *tb, int *path, float *qpath, float *score, int nread, int Tb, const int *tbs);
void launch_viterbi8x(hipStream_t s, const float *score_mat, uint8_t *tb, int *path, float *qpath, float *score, int nread, int Tb, const int *tbs, ReadMap map = ReadMap());
// Viterbi + traceback + qpath
void launch_viterbi(hipStream_t s, const float *score_mat, uint8_t *tb, int *path, float *qpath, float *score,
                    int nread, int Tb, int nbase, int Ps, const int *tbs = nullptr, ReadMap map = ReadMap());
// change positions -> base / quality strings
void launch_assemble(hipStream_t s, const int *path, const float *qpath, char *bases, char *quals, int *lens,
                     int nread, int Tb, int nbase, const int *tbs = nullptr, ReadMap map = ReadMap());
// exp + trace_from_posterior
void launch_trace(hipStream_t s, const float *post, int32_t *trace, int nread, int Tb, int nbase, int Ps, int is_log, const int *tbs = nullptr, ReadMap map = ReadMap());
void launch_exp_inplace(hipStream_t s, float *x, size_t n);
// run-length (runnie) head and decoders, ffhip_rle.hip: activation rows + runlengthV2 partition function + subtraction
void launch_rle_head_finish(hipStream_t s, float *param, double *logz, int nread, int Tb, int nbase, int Ps, float temperature, const int *tbs = nullptr);
void launch_rle_partition(hipStream_t s, const float *param, double *logz, int nread, int Tb, int nbase, int Ps, const int *tbs = nullptr);
void launch_rle_transpost(hipStream_t s, const float *param, float *post, float *fwd, int nread, int Tb, int nbase, int Ps, const int *tbs = nullptr);
void launch_rle_viterbi(hipStream_t s, const float *param, uint8_t *tb, int *path, float *qpath, float *score, int nread, int Tb, int nbase, int Ps, const int *tbs = nullptr);
// first-generation run-length decoders (decode.c:552-892) on one matrix of 4 nbase rows; tb: 8 bytes a block, fwd / bwd: 8 floats a block (+1)
void launch_rl1_viterbi(hipStream_t s, const float *param, uint8_t *tb, int *path, float *score, int nblk, int nbase, int Ps);
void launch_rl1_posterior(hipStream_t s, const float *param, float *post, float *fwd, float *bwd, int nblk, int nbase, int Ps);
void launch_rl1_mean(hipStream_t s, const float *param, const int *path, int *runlength, unsigned long long *seqlen, int nblk, int nbase, int Ps);
// tile-interleaved -> dense [Tb][H] of one read (debug tap)
void launch_untile(hipStream_t s, const float *act, float *dense, int read, int Tb, int B16, int H);

}  // namespace ffhip
