// ffhip_layers.hip -- the reference's per-layer operator interface (layers.h / flappie_matrix.h) on the GPU.
//
// Each ffhip_op_* takes host images of flappie matrices (column-major, column stride = 4*ceil(nr/4) floats),
// converts them on the device into the layouts of ffhip_internal.hpp, runs the SAME kernels the batched
// pipeline runs (convolution, MFMA projection, recurrent step / persistent recurrent layer, CRF head,
// partition function) on a batch of one, and writes the result image back.  Activations are separate
// element-wise calls over the whole image, pad lanes included, exactly as layers.c:24-120 applies them.
// These entry points exist for drop-in and unit-test use; throughput comes from the batch API.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "ffhip_host.hpp"
#include "ffhip_math.hpp"

using namespace ffhip;
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------ layout kernels
// image [nc][stride] -> sample-major rows [t][F] (dst points at sample 0; pads are zeroed by the caller)
__global__ void k_img_to_sample(const float *__restrict__ img, size_t stride, int F, int T, float *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * F) return;
    const size_t t = i / F, f = i % F;
    dst[i] = img[t * stride + f];
}

__global__ void k_sample_to_img(const float *__restrict__ src, int F, int T, float *__restrict__ img, size_t stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * F) return;
    const size_t t = i / F, f = i % F;
    img[t * stride + f] = src[i];
}

// image -> tile-interleaved A[tile][k/4][r16][k%4] with K16tot*16 features per tile, this image's
// features starting at quad k4_off.  single = 0: 16 consecutive columns are the 16 "reads" of a tile
// (projection-style work, columns are independent); single = 1: column c is tile c, read 0 (recurrent work).
__global__ void k_img_to_tiles(const float *__restrict__ img, size_t stride, int K, int nc, int K16, int K16tot, int k4_off,
                               int single, int ntile, float *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ntile * K16 * 64) return;
    const int tile = (int)(i / ((size_t)K16 * 64)), rem = (int)(i % ((size_t)K16 * 64));
    const int k4 = rem / 16, r = rem % 16;
    const int col = single ? (r == 0 ? tile : -1) : tile * 16 + r;
    v4f v = { 0.f, 0.f, 0.f, 0.f };
    if (col >= 0 && col < nc) {
        const float *p = img + (size_t)col * stride + 4 * k4;
        if (4 * k4 + 0 < K) v.x = p[0];
        if (4 * k4 + 1 < K) v.y = p[1];
        if (4 * k4 + 2 < K) v.z = p[2];
        if (4 * k4 + 3 < K) v.w = p[3];
    }
    *(v4f *)(dst + ((size_t)tile * K16tot * 64 + (size_t)(k4_off + k4) * 16 + r) * 4) = v;
}

// D-fragment X[tile][mt][lane][4] (row 16*mt + 4*(lane>>4) + e, column 16*tile + (lane&15)) -> image
__global__ void k_dfrag_to_img(const float *__restrict__ xa, int Mt, int M, int nc, int ntile, float *__restrict__ img, size_t stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ntile * Mt * 64) return;
    const int lane = (int)(i % 64), mt = (int)((i / 64) % Mt), tile = (int)(i / ((size_t)64 * Mt));
    const int col = tile * 16 + (lane & 15), row = 16 * mt + 4 * (lane >> 4);
    if (col >= nc) return;
    const v4f v = *(const v4f *)(xa + i * 4);
    float *o = img + (size_t)col * stride + row;
    if (row + 0 < M) o[0] = v.x;
    if (row + 1 < M) o[1] = v.y;
    if (row + 2 < M) o[2] = v.z;
    if (row + 3 < M) o[3] = v.w;
}

// gate pre-activations of one read, image [T][stride] with rows g*H + u -> D-fragment [t][ut][lane][4]
// (rows unit-major: lane (q, r) holds the G gates of unit 4*ut + q of read r; only read 0 is real)
__global__ void k_gates_to_dfrag(const float *__restrict__ img, size_t stride, int H, int G, int Ut, int T, float *__restrict__ xa) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * Ut * 64) return;
    const int lane = (int)(i % 64), ut = (int)((i / 64) % Ut);
    const size_t t = i / ((size_t)64 * Ut);
    const int u = 4 * ut + (lane >> 4);
    v4f v = { 0.f, 0.f, 0.f, 0.f };
    if ((lane & 15) == 0 && u < H) {
        const float *p = img + t * stride + u;
        v.x = p[0];
        v.y = p[(size_t)H];
        v.z = p[(size_t)2 * H];
        if (G > 3) v.w = p[(size_t)3 * H];
    }
    *(v4f *)(xa + i * 4) = v;
}

// tile-interleaved activations (one read tile, read 0) -> image [T][stride]
__global__ void k_tiles_to_img(const float *__restrict__ act, int Hp, int H, int T, float *__restrict__ img, size_t stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * H) return;
    const size_t t = i / H;
    const int f = (int)(i % H);
    img[t * stride + f] = act[t * Hp * 16 + (size_t)(f / 4) * 64 + (f % 4)];
}

// LSTM cell state of read 0: vector [H] <-> the step kernel's slot order [ut][lane = (q, r)]
__global__ void k_vec_to_cstate(const float *__restrict__ v, int H, int Ut, float *__restrict__ c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ut * 64) return;
    const int lane = i % 64, u = 4 * (i / 64) + (lane >> 4);
    c[i] = ((lane & 15) == 0 && u < H) ? v[u] : 0.0f;
}
__global__ void k_cstate_to_vec(const float *__restrict__ c, int H, float *__restrict__ v) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= H) return;
    v[u] = c[(u / 4) * 64 + (u % 4) * 16];
}

// ------------------------------------------------------------------------------------ element-wise
__global__ void k_activation(float *__restrict__ x, size_t n, int act, float p0, float p1, size_t nr, size_t stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (act == FFHIP_ACT_SHIFT_SCALE && i % stride >= nr) return;          // the one loop that skips pad lanes
    const float v = x[i];
    float r;
    switch (act) {
    case FFHIP_ACT_SWISH: r = swish_ref(v); break;                                   // layers.c:24-33
    case FFHIP_ACT_TANH: r = tanh_ref(v); break;                                     // layers.c:40-49
    case FFHIP_ACT_EXP: r = exp_cephes(v); break;                                    // layers.c:56-66
    case FFHIP_ACT_LOG: r = log_cephes(v); break;                                    // layers.c:73-81
    case FFHIP_ACT_ELU: r = (v >= 0.0f) ? v : (exp_cephes(v) - 1.0f); break;         // layers.c:88-96, util.h:339-347
    case FFHIP_ACT_ROBUSTLOG: r = log_cephes(p0 + p1 * v); break;                    // layers.c:109-124 (p0 = min_prob, p1 = 1 - min_prob)
    case FFHIP_ACT_SHIFT_SCALE: r = (v - p0) / p1; break;                            // flappie_matrix.c:625-633
    default: r = v;
    }
    x[i] = r;
}

__global__ void k_add_inplace(float *__restrict__ y, const float *__restrict__ x, size_t n) {      // layers.c:338-353
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = y[i] + x[i];
}

// row_normalise_inplace (flappie_matrix.c:425-447): four lane sums over the quads of a column, the pad
// lanes of the last quad subtracted, ((s0+s1)+(s2+s3)), then a multiply by the reciprocal.
__global__ void k_row_normalise(float *__restrict__ img, int nr, int nrq, int nc) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= nc) return;
    float *c = img + (size_t)col * nrq * 4;
    float s[4] = { c[0], c[1], c[2], c[3] };
    for (int q = 1; q < nrq; q++)
        for (int e = 0; e < 4; e++) s[e] = s[e] + c[4 * q + e];
    const int npad = 4 * nrq - nr;
    for (int e = 1; e < 4; e++)
        if (npad >= 4 - e) s[e] = s[e] - c[4 * (nrq - 1) + e];
    const float tsum = (s[0] + s[1]) + (s[2] + s[3]);
    const float recip = 1.0f / tsum;
    for (int i = 0; i < 4 * nrq; i++) c[i] = c[i] * recip;
}

// log_row_normalise_inplace (flappie_matrix.c:450-467): sequential logsumexpf chain over the nr rows
__global__ void k_log_row_normalise(float *__restrict__ img, int nr, size_t stride, int nc) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= nc) return;
    float *c = img + (size_t)col * stride;
    float ls = c[0];
    for (int r = 1; r < nr; r++) ls = logsumexpf_ref(ls, c[r]);
    for (int r = 0; r < nr; r++) c[r] = c[r] - ls;
}

inline unsigned nblk(size_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

bool view_ok(const ffhip_mat &m) { return m.data && m.nr > 0 && m.nc > 0 && m.stride >= m.nr; }

// image of a whole matrix, device side
float *upload_img(TmpDev &t, const ffhip_mat &m, hipStream_t s) { return mat_in(t, m, s); }      // the matrix's device image if it has one, else an upload

// A-fragment weights of W^T for `W` an [K x M] flappie matrix (column m holds output row m)
std::vector<float> pack_weight_T(const ffhip_mat &W, int Mt, int K16) {
    return pack_afrag(Mt, K16, [&](int row, int k) -> float {
        if ((size_t)row >= W.nc || (size_t)k >= W.nr) return 0.0f;
        return W.data[(size_t)row * W.stride + k];
    });
}

// recurrent weights with gate rows permuted unit-major (m = 4u + g), as ffhip_model_upload does
std::vector<float> pack_recurrent(const ffhip_mat &sW, int H, int G, int Hp) {
    return pack_afrag(Hp / 4, Hp / 16, [&](int row, int k) -> float {
        const int u = row / 4, g = row % 4;
        if (u >= H || g >= G || k >= H) return 0.0f;
        return sW.data[(size_t)(g * H + u) * sW.stride + k];
    });
}

}  // namespace

#define OP_ENTER(eng_)                                                     \
    if (!(eng_)) return set_err(FFHIP_EINVAL, "null engine");              \
    hipSetDevice((eng_)->device);                                          \
    hipStream_t s = (eng_)->streams[0];                                    \
    TmpDev tmp
#define OP_NOMEM() return set_err(FFHIP_ENOMEM, "device allocation failed")

// ------------------------------------------------------------------------------------ element-wise ops
extern "C" int ffhip_op_activation(ffhip_engine *eng, ffhip_mat C, int act, float p0, float p1) {
    OP_ENTER(eng);
    const bool lazy_ = mat_has_dev(C);      // in place on the device image when there is one
    if (!view_ok(C) || act < FFHIP_ACT_NONE || act > FFHIP_ACT_SHIFT_SCALE) return set_err(FFHIP_EINVAL, "bad activation arguments");
    const size_t n = C.nc * C.stride;
    float *d = upload_img(tmp, C, s);
    if (!d) OP_NOMEM();
    hipLaunchKernelGGL(k_activation, dim3(nblk(n)), dim3(256), 0, s, d, n, act, p0, p1, C.nr, C.stride);
    HIP_TRY(mat_done(C, d, lazy_, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

extern "C" int ffhip_op_add_inplace(ffhip_engine *eng, ffhip_mat Y, ffhip_mat X) {
    OP_ENTER(eng);
    const bool lazy_ = mat_has_dev(Y);
    if (!view_ok(Y) || !view_ok(X) || X.nr != Y.nr || X.nc != Y.nc || X.stride != Y.stride) return set_err(FFHIP_EINVAL, "residual: shapes differ");
    const size_t n = Y.nc * Y.stride;
    float *dy = upload_img(tmp, Y, s), *dx = upload_img(tmp, X, s);
    if (!dy || !dx) OP_NOMEM();
    hipLaunchKernelGGL(k_add_inplace, dim3(nblk(n)), dim3(256), 0, s, dy, dx, n);
    HIP_TRY(mat_done(Y, dy, lazy_, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

extern "C" int ffhip_op_row_normalise(ffhip_engine *eng, ffhip_mat C, int log_space) {
    OP_ENTER(eng);
    const bool lazy_ = mat_has_dev(C);      // in place on the device image when there is one
    if (!view_ok(C) || C.stride % 4 != 0) return set_err(FFHIP_EINVAL, "bad row-normalise arguments");
    const size_t n = C.nc * C.stride;
    float *d = upload_img(tmp, C, s);
    if (!d) OP_NOMEM();
    if (log_space) hipLaunchKernelGGL(k_log_row_normalise, dim3(nblk(C.nc, 64)), dim3(64), 0, s, d, (int)C.nr, C.stride, (int)C.nc);
    else hipLaunchKernelGGL(k_row_normalise, dim3(nblk(C.nc, 64)), dim3(64), 0, s, d, (int)C.nr, (int)(C.stride / 4), (int)C.nc);
    HIP_TRY(mat_done(C, d, lazy_, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// ------------------------------------------------------------------------------------ convolution
// layers.c:189-276.  C must be [W.nc x ceil(X.nc/stride)].  Thin layers (<= 32 filters) run the VALU
// kernel, wide ones the MFMA implicit-GEMM kernel, as in the batched pipeline.
extern "C" int ffhip_op_convolution(ffhip_engine *eng, ffhip_mat X, ffhip_mat W, ffhip_mat b, size_t conv_stride, ffhip_mat C) {
    OP_ENTER(eng);
    const bool lazy_ = mat_has_dev(X);      // the inputs live on the device: so does the result
    if (!view_ok(X) || !view_ok(W) || !view_ok(b) || !view_ok(C) || conv_stride < 1) return set_err(FFHIP_EINVAL, "bad convolution arguments");
    const int Fin = (int)X.nr, nf_pad = round_up(Fin, 4), Fout = (int)W.nc, T = (int)X.nc, cs = (int)conv_stride;
    const int wrows = round_up((int)W.nr, 4);
    if (wrows % nf_pad != 0) return set_err(FFHIP_EINVAL, "convolution: filter rows do not match %d input features (layers.c:198)", Fin);
    const int winlen = wrows / nf_pad;
    if (b.nr != W.nc) return set_err(FFHIP_EINVAL, "convolution: bias length (layers.c:195)");
    if (winlen > 48 || winlen * Fin > kSamplePad * Fin) return set_err(FFHIP_EINVAL, "convolution: window of %d samples not supported", winlen);
    std::vector<int> pa, pb;
    const int Tout = build_conv_plan(T, winlen, cs, pa, pb);
    if (Tout < 0) return set_err(FFHIP_EINVAL, "convolution: %d samples is shorter than the window (%d)", T, winlen);
    if ((size_t)Tout != C.nc || C.nr != (size_t)Fout) return set_err(FFHIP_EINVAL, "convolution: output must be %d x %d", Fout, Tout);

    auto tap = [&](int f, int t, int j) -> float {
        const size_t row = (size_t)t * nf_pad + j;
        return row < W.nr ? W.data[(size_t)f * W.stride + row] : 0.0f;
    };
    const size_t in_rows = (size_t)T + 2 * kSamplePad;
    float *d_x = upload_img(tmp, X, s);
    float *d_in = (float *)tmp.get(in_rows * Fin * 4);
    int *d_pa = (int *)tmp.upload(pa.data(), pa.size() * 4, s), *d_pb = (int *)tmp.upload(pb.data(), pb.size() * 4, s);
    float *d_c = mat_out(tmp, C, lazy_);
    if (!d_x || !d_in || !d_pa || !d_pb || !d_c) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_in, 0, in_rows * Fin * 4, s), FFHIP_EHIP);
    HIP_TRY(hipMemsetAsync(d_c, 0, C.nc * C.stride * 4, s), FFHIP_EHIP);
    hipLaunchKernelGGL(k_img_to_sample, dim3(nblk((size_t)T * Fin)), dim3(256), 0, s, d_x, X.stride, Fin, T, d_in + (size_t)kSamplePad * Fin);
    SampleBuf in{ d_in, Fin, T, 0 };             // read stride 0: a batch of one (the MFMA kernel's 16 tile columns alias it)
    const size_t K = (size_t)winlen * Fin;
    if (Fout <= 32 && (Fout * K + Fout) * 4 <= 60 * 1024) {
        std::vector<float> taps((size_t)Fout * K);
        for (int f = 0; f < Fout; f++)
            for (int t = 0; t < winlen; t++)
                for (int j = 0; j < Fin; j++) taps[((size_t)f * winlen + t) * Fin + j] = tap(f, t, j);
        float *d_w = (float *)tmp.upload(taps.data(), taps.size() * 4, s);
        float *d_b = (float *)tmp.upload(b.data, (size_t)Fout * 4, s);
        const size_t out_rows = (size_t)Tout + 2 * kSamplePad;
        float *d_out = (float *)tmp.get(out_rows * Fout * 4);
        if (!d_w || !d_b || !d_out) OP_NOMEM();
        SampleBuf out{ d_out, Fout, Tout, 0 };
        launch_conv_small(s, in, out, d_w, d_b, d_pa, d_pb, 1, Tout, winlen, ACT_NONE);
        hipLaunchKernelGGL(k_sample_to_img, dim3(nblk((size_t)Tout * Fout)), dim3(256), 0, s, d_out + (size_t)kSamplePad * Fout, Fout, Tout, d_c, C.stride);
    } else {
        const int K16 = (int)((K + 15) / 16), Mpad = round_up(Fout, 16);
        if ((size_t)K16 * 16 > (size_t)kSamplePad * Fin) return set_err(FFHIP_EINVAL, "convolution: window too long for the sample pad");
        std::vector<float> wp = pack_afrag(Mpad / 16, K16, [&](int row, int k) -> float {
            if (row >= Fout || (size_t)k >= K) return 0.0f;
            return tap(row, k / Fin, k % Fin);
        });
        std::vector<float> bias(Mpad, 0.0f);
        for (int f = 0; f < Fout; f++) bias[f] = b.data[f];
        float *d_w = (float *)tmp.upload(wp.data(), wp.size() * 4, s);
        float *d_b = (float *)tmp.upload(bias.data(), bias.size() * 4, s);
        float *d_out = (float *)tmp.get((size_t)Tout * Mpad * 16 * 4);
        if (!d_w || !d_b || !d_out) OP_NOMEM();
        launch_conv_mfma(s, in, d_out, (const float4 *)d_w, d_b, d_pa, d_pb, 1, Tout, Mpad, K16, ACT_NONE);
        hipLaunchKernelGGL(k_tiles_to_img, dim3(nblk((size_t)Tout * Fout)), dim3(256), 0, s, d_out, Mpad, Fout, Tout, d_c, C.stride);
    }
    HIP_TRY(mat_done(C, d_c, lazy_, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    return FFHIP_OK;
}

// ------------------------------------------------------------------------------------ affine maps
// affine_map / affine_map2 (flappie_matrix.c:361-419): C = Wf^T Xf (+ Wb^T Xb) + b.  Xb/Wb may be empty.
extern "C" int ffhip_op_affine(ffhip_engine *eng, ffhip_mat Xf, ffhip_mat Wf, ffhip_mat Xb, ffhip_mat Wb, ffhip_mat b, ffhip_mat C) {
    OP_ENTER(eng);
    const bool lazy_ = mat_has_dev(Xf) && (Xb.data == nullptr || mat_has_dev(Xb));      // the inputs live on the device: so does the result
    const bool two = Xb.data != nullptr;
    if (!view_ok(Xf) || !view_ok(Wf) || !view_ok(b) || !view_ok(C) || (two && (!view_ok(Xb) || !view_ok(Wb))))
        return set_err(FFHIP_EINVAL, "bad affine arguments");
    if (Wf.nr != Xf.nr || b.nr != Wf.nc || C.nr != Wf.nc || C.nc != Xf.nc) return set_err(FFHIP_EINVAL, "affine: shapes do not agree (flappie_matrix.c:373)");
    if (two && (Wb.nr != Xb.nr || Xb.nc != Xf.nc || Wb.nc != Wf.nc)) return set_err(FFHIP_EINVAL, "affine2: shapes do not agree (flappie_matrix.c:399-404)");
    const int M = (int)Wf.nc, Mpad = round_up(M, 16), Mt = Mpad / 16, nc = (int)Xf.nc, ntile = (nc + 15) / 16;
    const int K16f = (int)((Xf.nr + 15) / 16), K16b = two ? (int)((Xb.nr + 15) / 16) : 0, K16 = K16f + K16b;
    std::vector<float> wp = pack_afrag(Mt, K16, [&](int row, int k) -> float {
        if (row >= M) return 0.0f;
        if (k < K16f * 16) return (size_t)k < Wf.nr ? Wf.data[(size_t)row * Wf.stride + k] : 0.0f;
        const size_t kb = (size_t)k - (size_t)K16f * 16;
        return kb < Wb.nr ? Wb.data[(size_t)row * Wb.stride + kb] : 0.0f;
    });
    std::vector<float> bias(Mpad, 0.0f);
    for (int m = 0; m < M; m++) bias[m] = b.data[m];
    float *d_xf = upload_img(tmp, Xf, s), *d_xb = two ? upload_img(tmp, Xb, s) : nullptr;
    float *d_w = (float *)tmp.upload(wp.data(), wp.size() * 4, s), *d_b = (float *)tmp.upload(bias.data(), bias.size() * 4, s);
    float *d_in = (float *)tmp.get((size_t)ntile * K16 * 256 * 4), *d_xa = (float *)tmp.get((size_t)ntile * Mt * 256 * 4);
    float *d_c = mat_out(tmp, C, lazy_);
    if (!d_xf || (two && !d_xb) || !d_w || !d_b || !d_in || !d_xa || !d_c) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_c, 0, C.nc * C.stride * 4, s), FFHIP_EHIP);
    hipLaunchKernelGGL(k_img_to_tiles, dim3(nblk((size_t)ntile * K16f * 64)), dim3(256), 0, s, d_xf, Xf.stride, (int)Xf.nr, nc, K16f, K16, 0, 0, ntile, d_in);
    if (two)
        hipLaunchKernelGGL(k_img_to_tiles, dim3(nblk((size_t)ntile * K16b * 64)), dim3(256), 0, s, d_xb, Xb.stride, (int)Xb.nr, nc, K16b, K16, 4 * K16f, 0, ntile, d_in);
    launch_inproj(s, d_in, d_xa, (const float4 *)d_w, d_b, ntile, Mpad, K16);
    hipLaunchKernelGGL(k_dfrag_to_img, dim3(nblk((size_t)ntile * Mt * 64)), dim3(256), 0, s, d_xa, Mt, M, nc, ntile, d_c, C.stride);
    HIP_TRY(mat_done(C, d_c, lazy_, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    return FFHIP_OK;
}

// ------------------------------------------------------------------------------------ recurrent layers
// lstm_forward/backward (layers.c:877-976) and grumod_forward/backward (layers.c:571-660): Xa holds the
// already-projected input [G*H x T], sW is [H x G*H], out is [H x T].  The persistent recurrent kernel
// runs the layer when it supports the shape, the launch-per-step kernels otherwise.
extern "C" int ffhip_op_recurrent(ffhip_engine *eng, int kind, ffhip_mat Xa, ffhip_mat sW, int backward, ffhip_mat out) {
    OP_ENTER(eng);
    const bool lazy_ = mat_has_dev(Xa);      // the inputs live on the device: so does the result
    if (kind != FFHIP_NET_LSTM5 && kind != FFHIP_NET_GRUMOD5) return set_err(FFHIP_EINVAL, "unknown recurrent kind %d", kind);
    if (!view_ok(Xa) || !view_ok(sW) || !view_ok(out)) return set_err(FFHIP_EINVAL, "bad recurrent-layer arguments");
    const int G = (kind == FFHIP_NET_LSTM5) ? 4 : 3, H = (int)sW.nr, T = (int)Xa.nc;
    if (H % 4 != 0 || sW.nc != (size_t)G * H || Xa.nr != (size_t)G * H || out.nr != (size_t)H || out.nc != Xa.nc)
        return set_err(FFHIP_EINVAL, "recurrent layer: shapes do not agree (layers.c:884-889)");
    const int Hp = round_up(H, 16), Ut = Hp / 4;
    std::vector<float> sp = pack_recurrent(sW, H, G, Hp);
    float *d_x = upload_img(tmp, Xa, s);
    float *d_w = (float *)tmp.upload(sp.data(), sp.size() * 4, s);
    float *d_xa = (float *)tmp.get((size_t)T * Ut * 256 * 4), *d_h = (float *)tmp.get((size_t)T * Hp * 16 * 4);
    float *d_o = mat_out(tmp, out, lazy_);
    if (!d_x || !d_w || !d_xa || !d_h || !d_o) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_o, 0, out.nc * out.stride * 4, s), FFHIP_EHIP);
    hipLaunchKernelGGL(k_gates_to_dfrag, dim3(nblk((size_t)T * Ut * 64)), dim3(256), 0, s, d_x, Xa.stride, H, G, Ut, T, d_xa);
    unsigned h_abort = 0;
    if (persist_supported(kind, Hp, eng->prop.multiProcessorCount)) {
        unsigned *d_flags = (unsigned *)tmp.get(persist_flag_words(Hp, 1) * 4), *d_abort = (unsigned *)tmp.get(4);
        if (!d_flags || !d_abort) OP_NOMEM();
        HIP_TRY(hipMemsetAsync(d_flags, 0, persist_flag_words(Hp, 1) * 4, s), FFHIP_EHIP);
        HIP_TRY(hipMemsetAsync(d_abort, 0, 4, s), FFHIP_EHIP);
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_h, (int)0xFFFFFFFF, (size_t)T * Hp * 16, s), FFHIP_EHIP);     // NaN sentinel = "not written yet"
        if (eng->persist_chained) HIP_TRY(hipStreamWaitEvent(s, eng->persist_done, 0), FFHIP_EHIP);
        if (!launch_rnn_persist(s, kind, (const float4 *)d_w, d_xa, d_h, d_flags, d_abort, T, 1, Hp, 0, 1, backward, 0))
            return set_err(FFHIP_EINVAL, "persistent recurrent kernel: unsupported shape");
        HIP_TRY(hipEventRecord(eng->persist_done, s), FFHIP_EHIP);
        eng->persist_chained = 1;
        HIP_TRY(hipMemcpyAsync(&h_abort, d_abort, 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    } else {
        float *d_c = (float *)tmp.get((size_t)Ut * 64 * 4);
        if (!d_c) OP_NOMEM();
        const size_t xa_step = (size_t)Ut * 256, h_step = (size_t)Hp * 16;
        for (int i = 0; i < T; i++) {
            const int t = backward ? T - 1 - i : i, tp = backward ? t + 1 : t - 1;
            const float *hp = (i == 0) ? nullptr : d_h + (size_t)tp * h_step;
            if (kind == FFHIP_NET_LSTM5) launch_lstm_step(s, (const float4 *)d_w, d_xa + (size_t)t * xa_step, hp, d_h + (size_t)t * h_step, d_c, 1, Hp, i == 0);
            else launch_gru_step(s, (const float4 *)d_w, d_xa + (size_t)t * xa_step, hp, d_h + (size_t)t * h_step, 1, Hp, i == 0);
        }
    }
    hipLaunchKernelGGL(k_tiles_to_img, dim3(nblk((size_t)T * H)), dim3(256), 0, s, d_h, Hp, H, T, d_o, out.stride);
    HIP_TRY(mat_done(out, d_o, lazy_, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    if (h_abort) return set_err(FFHIP_ETIMEOUT, "persistent recurrent kernel: an inter-workgroup wait timed out; results are invalid");
    return FFHIP_OK;
}

// lstm_step (layers.c:979-1026) / grumod_step (layers.c:664-715): one step from (x, h_prev[, c]) to h (and c).
// x is the projected input [G*H x 1]; `state` (LSTM cell state, in/out) is ignored for GRUmod.
extern "C" int ffhip_op_recurrent_step(ffhip_engine *eng, int kind, ffhip_mat x, ffhip_mat h_prev, ffhip_mat sW, ffhip_mat state, ffhip_mat h_out) {
    OP_ENTER(eng);
    if (kind != FFHIP_NET_LSTM5 && kind != FFHIP_NET_GRUMOD5) return set_err(FFHIP_EINVAL, "unknown recurrent kind %d", kind);
    const bool lstm = (kind == FFHIP_NET_LSTM5);
    if (!view_ok(x) || !view_ok(h_prev) || !view_ok(sW) || !view_ok(h_out) || (lstm && !view_ok(state))) return set_err(FFHIP_EINVAL, "bad recurrent-step arguments");
    const int G = lstm ? 4 : 3, H = (int)sW.nr;
    if (H % 4 != 0 || sW.nc != (size_t)G * H || x.nr != (size_t)G * H || h_prev.nr != (size_t)H || h_out.nr != (size_t)H || (lstm && state.nr != (size_t)H))
        return set_err(FFHIP_EINVAL, "recurrent step: shapes do not agree (layers.c:985-994)");
    const int Hp = round_up(H, 16), Ut = Hp / 4;
    std::vector<float> sp = pack_recurrent(sW, H, G, Hp);
    float *d_x = (float *)tmp.upload(x.data, x.stride * 4, s), *d_hp = (float *)tmp.upload(h_prev.data, h_prev.stride * 4, s);
    float *d_w = (float *)tmp.upload(sp.data(), sp.size() * 4, s);
    float *d_xa = (float *)tmp.get((size_t)Ut * 256 * 4), *d_hin = (float *)tmp.get((size_t)Hp * 16 * 4), *d_hout = (float *)tmp.get((size_t)Hp * 16 * 4);
    float *d_c = (float *)tmp.get((size_t)Ut * 64 * 4), *d_v = (float *)tmp.get((size_t)Hp * 4);
    if (!d_x || !d_hp || !d_w || !d_xa || !d_hin || !d_hout || !d_c || !d_v) OP_NOMEM();
    hipLaunchKernelGGL(k_gates_to_dfrag, dim3(nblk((size_t)Ut * 64)), dim3(256), 0, s, d_x, x.stride, H, G, Ut, 1, d_xa);
    hipLaunchKernelGGL(k_img_to_tiles, dim3(nblk((size_t)(Hp / 16) * 64)), dim3(256), 0, s, d_hp, h_prev.stride, H, 1, Hp / 16, Hp / 16, 0, 1, 1, d_hin);
    if (lstm) {
        float *d_s = (float *)tmp.upload(state.data, state.stride * 4, s);
        if (!d_s) OP_NOMEM();
        hipLaunchKernelGGL(k_vec_to_cstate, dim3(nblk((size_t)Ut * 64)), dim3(256), 0, s, d_s, H, Ut, d_c);
        launch_lstm_step(s, (const float4 *)d_w, d_xa, d_hin, d_hout, d_c, 1, Hp, 0);
        hipLaunchKernelGGL(k_cstate_to_vec, dim3(nblk(H)), dim3(256), 0, s, d_c, H, d_v);
        HIP_TRY(hipMemcpyAsync(state.data, d_v, (size_t)H * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    } else {
        launch_gru_step(s, (const float4 *)d_w, d_xa, d_hin, d_hout, 1, Hp, 0);
    }
    float *d_o = (float *)tmp.get(h_out.stride * 4);
    if (!d_o) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_o, 0, h_out.stride * 4, s), FFHIP_EHIP);
    hipLaunchKernelGGL(k_tiles_to_img, dim3(nblk(H)), dim3(256), 0, s, d_hout, Hp, H, 1, d_o, h_out.stride);
    HIP_TRY(hipMemcpyAsync(h_out.data, d_o, (size_t)H * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    return FFHIP_OK;
}

// ------------------------------------------------------------------------------------ CRF head
// crf_manystay_partition_function (layers.c:1035-1079): fp64 forward recursion over a [P x nblock] score matrix
extern "C" int ffhip_op_partition_function(ffhip_engine *eng, ffhip_mat S, double *logZ) {
    OP_ENTER(eng);
    int nbase;
    if (!view_ok(S) || !logZ || !flipflop_dims(S.nr, S.stride, &nbase)) return set_err(FFHIP_EINVAL, "bad partition-function arguments");
    float *d = upload_img(tmp, S, s);
    double *d_z = (double *)tmp.get(sizeof(double));
    if (!d || !d_z) OP_NOMEM();
    launch_crf_norm(s, d, 1, (int)S.nc, nbase, (int)S.stride, d_z, 0);
    HIP_TRY(hipMemcpyAsync(logZ, d_z, sizeof(double), hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// the pipeline's linear-space evaluation of the same quantity (ffhip_kernels.hip, k_crf_chain); valid when every
// |score| <= bound, FFHIP_EINVAL when the bound is too wide for it
extern "C" int ffhip_op_partition_function_scaled(ffhip_engine *eng, ffhip_mat S, float bound, double *logZ) {
    OP_ENTER(eng);
    int nbase;
    if (!view_ok(S) || !logZ || !flipflop_dims(S.nr, S.stride, &nbase)) return set_err(FFHIP_EINVAL, "bad partition-function arguments");
    const int R = crf_rescale_interval(bound);
    if (R < 1) return set_err(FFHIP_EINVAL, "score bound %g is too wide for the linear-space partition function", (double)bound);
    float *d = upload_img(tmp, S, s);
    double *d_z = (double *)tmp.get(sizeof(double));
    double *d_e = (double *)tmp.get(S.nc * (size_t)crf_exp_stride((int)S.nr) * sizeof(double));
    if (!d || !d_z || !d_e) OP_NOMEM();
    // what the pipeline runs for the 8- and 10-state models: the forward chain of k_crf_fb (ffhip_decode.hip)
    if (((nbase == 4 && S.stride == 40) || (nbase == 5 && S.stride == 60)) && 2.0f * bound <= kFbRange && !dbg("decode_r2")) {
        launch_crf_exp(s, d, d_e, 1, (int)S.nc, nbase, (int)S.stride, nullptr, nullptr, 0.0f);
        launch_crf_fb(s, nbase, d_e, d, nullptr, nullptr, 1, (int)S.nc, d_z, nullptr, 0, nullptr);
    } else
    launch_crf_norm_linear(s, d, d_e, 1, (int)S.nc, nbase, (int)S.stride, R, d_z, 0);
    HIP_TRY(hipMemcpyAsync(logZ, d_z, sizeof(double), hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// globalnorm_flipflop (layers.c:1082-1106): C = tanh(W^T X + b) / (temperature/5) - logZ/nblock
extern "C" int ffhip_op_globalnorm_flipflop(ffhip_engine *eng, ffhip_mat X, ffhip_mat W, ffhip_mat b, float temperature, ffhip_mat C) {
    OP_ENTER(eng);
    const bool lazy_ = mat_has_dev(X);      // the inputs live on the device: so does the result
    int nbase;
    if (!view_ok(X) || !view_ok(W) || !view_ok(b) || !view_ok(C)) return set_err(FFHIP_EINVAL, "bad globalnorm arguments");
    if (W.nr != X.nr || b.nr != W.nc || C.nr != W.nc || C.nc != X.nc) return set_err(FFHIP_EINVAL, "globalnorm: shapes do not agree");
    if (!flipflop_dims(W.nc, C.stride, &nbase)) return set_err(FFHIP_EINVAL, "globalnorm: %zu rows is not a flip-flop parameterisation this engine supports", W.nc);
    const int H = (int)X.nr, Hp = round_up(H, 16), K16 = Hp / 16, P = (int)W.nc, Mt = (P + 15) / 16, T = (int)X.nc;
    std::vector<float> wp = pack_weight_T(W, Mt, K16);
    std::vector<float> bias((size_t)Mt * 16, 0.0f);
    for (int p = 0; p < P; p++) bias[p] = b.data[p];
    float *d_x = upload_img(tmp, X, s);
    float *d_w = (float *)tmp.upload(wp.data(), wp.size() * 4, s), *d_b = (float *)tmp.upload(bias.data(), bias.size() * 4, s);
    float *d_in = (float *)tmp.get((size_t)T * K16 * 256 * 4), *d_c = mat_out(tmp, C, lazy_);
    if (!d_x || !d_w || !d_b || !d_in || !d_c) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_c, 0, C.nc * C.stride * 4, s), FFHIP_EHIP);
    hipLaunchKernelGGL(k_img_to_tiles, dim3(nblk((size_t)T * K16 * 64)), dim3(256), 0, s, d_x, X.stride, H, T, K16, K16, 0, 1, T, d_in);
    launch_head(s, d_in, d_c, (const float4 *)d_w, d_b, T, 1, 1, P, (int)C.stride, K16, temperature / 5.0f);
    double *d_z = (double *)tmp.get(sizeof(double));
    if (!d_z) OP_NOMEM();
    launch_crf_norm(s, d_c, 1, T, nbase, (int)C.stride, d_z);
    HIP_TRY(mat_done(C, d_c, lazy_, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    return FFHIP_OK;
}

// ------------------------------------------------------------------------------------ run-length (runnie) head and decoders
// globalnorm_runlengthV2 (layers.c:1325-1358)
extern "C" int ffhip_op_globalnorm_runlength(ffhip_engine *eng, ffhip_mat X, ffhip_mat W, ffhip_mat b, float temperature, ffhip_mat C) {
    OP_ENTER(eng);
    const bool lazy_ = mat_has_dev(X);      // the inputs live on the device: so does the result
    int nbase;
    if (!view_ok(X) || !view_ok(W) || !view_ok(b) || !view_ok(C)) return set_err(FFHIP_EINVAL, "bad globalnorm arguments");
    if (W.nr != X.nr || b.nr != W.nc || C.nr != W.nc || C.nc != X.nc) return set_err(FFHIP_EINVAL, "globalnorm: shapes do not agree");
    if (!flipflop_dims(W.nc, C.stride, &nbase)) return set_err(FFHIP_EINVAL, "globalnorm_runlengthV2: %zu rows is not 2*nbase*(nbase+1) with nbase <= 5", W.nc);
    const int H = (int)X.nr, Hp = round_up(H, 16), K16 = Hp / 16, P = (int)W.nc, Mt = (P + 15) / 16, T = (int)X.nc;
    std::vector<float> wp = pack_weight_T(W, Mt, K16);
    std::vector<float> bias((size_t)Mt * 16, 0.0f);
    for (int p = 0; p < P; p++) bias[p] = b.data[p];
    float *d_x = upload_img(tmp, X, s);
    float *d_w = (float *)tmp.upload(wp.data(), wp.size() * 4, s), *d_b = (float *)tmp.upload(bias.data(), bias.size() * 4, s);
    float *d_in = (float *)tmp.get((size_t)T * K16 * 256 * 4), *d_c = mat_out(tmp, C, lazy_);
    double *d_z = (double *)tmp.get(sizeof(double));
    if (!d_x || !d_w || !d_b || !d_in || !d_c || !d_z) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_c, 0, C.nc * C.stride * 4, s), FFHIP_EHIP);
    hipLaunchKernelGGL(k_img_to_tiles, dim3(nblk((size_t)T * K16 * 64)), dim3(256), 0, s, d_x, X.stride, H, T, K16, K16, 0, 1, T, d_in);
    launch_head(s, d_in, d_c, (const float4 *)d_w, d_b, T, 1, 1, P, (int)C.stride, K16, 1.0f, 1);
    launch_rle_head_finish(s, d_c, d_z, 1, T, nbase, (int)C.stride, temperature);
    HIP_TRY(mat_done(C, d_c, lazy_, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    return FFHIP_OK;
}

// runlengthV2_partition_function (layers.c:1255-1302)
extern "C" int ffhip_op_runlength_partition_function(ffhip_engine *eng, ffhip_mat S, double *logZ) {
    OP_ENTER(eng);
    int nbase;
    if (!view_ok(S) || !logZ || !flipflop_dims(S.nr, S.stride, &nbase)) return set_err(FFHIP_EINVAL, "bad partition-function arguments");
    float *d = upload_img(tmp, S, s);
    double *d_z = (double *)tmp.get(sizeof(double));
    if (!d || !d_z) OP_NOMEM();
    launch_rle_partition(s, d, d_z, 1, (int)S.nc, nbase, (int)S.stride);
    HIP_TRY(hipMemcpyAsync(logZ, d_z, sizeof(double), hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// transpost_crf_runlength (decode.c:1037-1159); post has the shape of param
extern "C" int ffhip_runlength_transpost(ffhip_engine *eng, ffhip_mat param, ffhip_mat post) {
    OP_ENTER(eng);
    int nbase;
    if (!view_ok(param) || !view_ok(post) || post.nr != param.nr || post.nc != param.nc || post.stride != param.stride ||
        !flipflop_dims(param.nr, param.stride, &nbase)) return set_err(FFHIP_EINVAL, "bad run-length posterior arguments");
    const size_t n = param.nc * param.stride;
    float *d_p = upload_img(tmp, param, s), *d_o = (float *)tmp.get(n * 4), *d_f = (float *)tmp.get(2 * (param.nc + 1) * kMaxState * 4);
    if (!d_p || !d_o || !d_f) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_o, 0, n * 4, s), FFHIP_EHIP);
    launch_rle_transpost(s, d_p, d_o, d_f, 1, (int)param.nc, nbase, (int)param.stride);
    HIP_TRY(hipMemcpyAsync(post.data, d_o, n * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// decode_crf_runlength (decode.c:927-1013): path[nblock] of states 0..2*nbase-1, returns the best score through *score
extern "C" int ffhip_runlength_viterbi(ffhip_engine *eng, ffhip_mat param, int *path, float *score) {
    OP_ENTER(eng);
    int nbase;
    if (!view_ok(param) || !path || !flipflop_dims(param.nr, param.stride, &nbase)) return set_err(FFHIP_EINVAL, "bad run-length decode arguments");
    const size_t nblock = param.nc;
    float *d_p = upload_img(tmp, param, s), *d_q = (float *)tmp.get((nblock + 1) * 4), *d_s = (float *)tmp.get(4);
    uint8_t *d_tb = (uint8_t *)tmp.get(nblock * kMaxState);
    int *d_path = (int *)tmp.get((nblock + 1) * 4);
    if (!d_p || !d_q || !d_s || !d_tb || !d_path) OP_NOMEM();
    launch_rle_viterbi(s, d_p, d_tb, d_path, d_q, d_s, 1, (int)nblock, nbase, (int)param.stride);
    float sc = NAN;
    HIP_TRY(hipMemcpyAsync(path, d_path, nblock * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(&sc, d_s, 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    if (score) *score = sc;
    return FFHIP_OK;
}

// ---- decoders of the first run-length head (decode.c:552-892): param is [4 nbase x nblock]
static bool rl1_dims(const ffhip_mat &param, int *nbase) {
    if (!view_ok(param) || param.nr % 4 != 0 || param.nr / 4 < 1 || param.nr / 4 > 8) return false;
    *nbase = (int)(param.nr / 4);
    return true;
}

// decode_runlength (decode.c:694-767): path[nblock] = the base entered in a block, -1 while staying; the best score through *score
extern "C" int ffhip_runlength_v1_viterbi(ffhip_engine *eng, ffhip_mat param, int *path, float *score) {
    OP_ENTER(eng);
    int nbase;
    if (!path || !rl1_dims(param, &nbase)) return set_err(FFHIP_EINVAL, "bad run-length (v1) decode arguments");
    const size_t nblock = param.nc;
    float *d_p = upload_img(tmp, param, s), *d_s = (float *)tmp.get(4);
    uint8_t *d_tb = (uint8_t *)tmp.get(nblock * 8);
    int *d_path = (int *)tmp.get(nblock * 4);
    if (!d_p || !d_s || !d_tb || !d_path) OP_NOMEM();
    launch_rl1_viterbi(s, d_p, d_tb, d_path, d_s, (int)nblock, nbase, (int)param.stride);
    float sc = NAN;
    HIP_TRY(hipMemcpyAsync(path, d_path, nblock * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(&sc, d_s, 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    if (score) *score = sc;
    return FFHIP_OK;
}

// posterior_runlength (decode.c:793-892): post is [4 nbase x nblock + 1] with param's stride; everything but the move and stay rows of
// blocks 0 .. nblock - 1 is zero
extern "C" int ffhip_runlength_v1_posterior(ffhip_engine *eng, ffhip_mat param, ffhip_mat post) {
    OP_ENTER(eng);
    int nbase;
    if (!rl1_dims(param, &nbase) || !view_ok(post) || post.nr != param.nr || post.nc != param.nc + 1 || post.stride != param.stride)
        return set_err(FFHIP_EINVAL, "bad run-length (v1) posterior arguments");
    const size_t nblock = param.nc, n = (nblock + 1) * param.stride;
    float *d_p = upload_img(tmp, param, s), *d_o = (float *)tmp.get(n * 4), *d_f = (float *)tmp.get(2 * (nblock + 1) * 8 * 4);
    if (!d_p || !d_o || !d_f) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_o, 0, n * 4, s), FFHIP_EHIP);
    launch_rl1_posterior(s, d_p, d_o, d_f, d_f + (nblock + 1) * 8, (int)nblock, nbase, (int)param.stride);
    HIP_TRY(hipMemcpyAsync(post.data, d_o, n * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// runlengths_mean (decode.c:576-603): runlength[nblock] from the discrete-Weibull rows of the entered base, 0 while staying; the sum through *seqlen
extern "C" int ffhip_runlength_v1_mean(ffhip_engine *eng, ffhip_mat param, const int *path, int *runlength, size_t *seqlen) {
    OP_ENTER(eng);
    int nbase;
    if (!path || !runlength || !rl1_dims(param, &nbase)) return set_err(FFHIP_EINVAL, "bad run-length (v1) mean arguments");
    const size_t nblock = param.nc;
    for (size_t b = 0; b < nblock; b++) if (path[b] >= nbase) return set_err(FFHIP_EINVAL, "run-length (v1) mean: a path entry is not a base");
    float *d_p = upload_img(tmp, param, s);
    int *d_path = (int *)tmp.get(nblock * 4), *d_rl = (int *)tmp.get(nblock * 4);
    unsigned long long *d_n = (unsigned long long *)tmp.get(8);
    if (!d_p || !d_path || !d_rl || !d_n) OP_NOMEM();
    HIP_TRY(hipMemcpyAsync(d_path, path, nblock * 4, hipMemcpyHostToDevice, s), FFHIP_EHIP);
    HIP_TRY(hipMemsetAsync(d_n, 0, 8, s), FFHIP_EHIP);
    launch_rl1_mean(s, d_p, d_path, d_rl, d_n, (int)nblock, nbase, (int)param.stride);
    unsigned long long tot = 0;
    HIP_TRY(hipMemcpyAsync(runlength, d_rl, nblock * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(&tot, d_n, 8, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    if (seqlen) *seqlen = (size_t)tot;
    return FFHIP_OK;
}

// ------------------------------------------------------------------------------------ sloika GRU layers and the first run-length head
// gru_forward/backward/step (layers.c:412-568) and gru_relu_* (layers.c:718-874): the GRU of the sloika networks
// (networks.c:403, :492).  No model in the reference's registry uses them (networks.c:85-99), so this is a
// correctness-level operator, not a throughput path: ONE workgroup walks the steps of the one read, the two weight
// matrices are read from L2 every step, one wave per output row with a lane-strided dot product.
//   z, r = sigma(x[0:2H] + sW^T h) ; hbar = act(x[2H:3H] + sW2^T (r * h)) ; h' = z h + (1 - z) hbar
namespace {

template <int RELU>
__global__ void __launch_bounds__(1024)
k_gru_sloika(const float *__restrict__ X, size_t xs, const float *__restrict__ sW, size_t ws, const float *__restrict__ sW2, size_t w2s,
             const float *__restrict__ h0, int H, int T, int backward, float *__restrict__ out, size_t os) {
    extern __shared__ float gru_sm[];
    float *h = gru_sm, *rh = gru_sm + H, *g = gru_sm + 2 * H;                 // state, r * h, the two gates
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    for (int u = tid; u < H; u += blockDim.x) h[u] = h0 ? h0[u] : 0.0f;       // layers.c:437: the first state is zero
    __syncthreads();
    for (int i = 0; i < T; i++) {
        const int t = backward ? T - 1 - i : i;
        const float *x = X + (size_t)t * xs;
        for (int j = wave; j < 2 * H; j += nw) {                              // layers.c:545-549
            const float *w = sW + (size_t)j * ws;
            float acc = 0.0f;
            for (int k = lane; k < H; k += 64) acc += w[k] * h[k];
            for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
            if (lane == 0) g[j] = logistic_ref(x[j] + acc);
        }
        __syncthreads();
        for (int u = tid; u < H; u += blockDim.x) rh[u] = g[H + u] * h[u];    // layers.c:554-556
        __syncthreads();
        for (int u = wave; u < H; u += nw) {                                  // layers.c:557-566
            const float *w = sW2 + (size_t)u * w2s;
            float acc = 0.0f;
            for (int k = lane; k < H; k += 64) acc += w[k] * rh[k];
            for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
            if (lane == 0) {
                const float pre = x[2 * H + u] + acc;
                const float hbar = RELU ? fmaxf(pre, 0.0f) : tanh_ref(pre);
                const float z = g[u];
                const float hn = z * h[u] + (1.0f - z) * hbar;
                out[(size_t)t * os + u] = hn;
                h[u] = hn;                                                    // every read of h for this step is done
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ double lse64_l(double x, double y) { return fmax(x, y) + log1p(exp(-fabs(x - y))); }
__device__ __forceinline__ float softplus_l(float x) { return log1pf(expf(-fabsf(x))) + ((x >= 0.0f) ? x : 0.f); }

// globalnorm_runlength (layers.c:1197-1228), rows after the affine map: shape, scale, move, stay (nbase each)
__global__ void k_rle1_activate(float *__restrict__ C, size_t n, int nbase, int Ps, float temperature) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = (int)(i % Ps);
    if (p >= 4 * nbase) return;
    const float x = C[i];
    C[i] = p < nbase ? 1.0f + softplus_l(x) : (p < 2 * nbase ? 1e-1f + softplus_l(x) : 5.0f * tanhf(x) / temperature);
}

// runlength_partition_function (layers.c:1127-1174): one wave, lane = base, fp64 chain in the reference's order
__global__ void __launch_bounds__(64)
k_rle1_partition(const float *__restrict__ C, int nc, int nbase, int Ps, double *__restrict__ logz) {
    __shared__ double st[2][64];
    const int lane = threadIdx.x;
    if (lane < nbase) st[0][lane] = 0.0;
    __syncthreads();
    int cur = 0;
    for (int c = 0; c < nc; c++) {
        const float *move = C + (size_t)c * Ps + 2 * nbase, *stay = move + nbase;
        if (lane < nbase) {
            const double *prev = st[cur];
            double v = -HUGE_VAL;
            for (int b2 = 0; b2 < nbase; b2++)
                if (b2 != lane) v = lse64_l(v, prev[b2]);
            v += (double)move[lane];
            v = lse64_l(v, prev[lane] + (double)stay[lane]);
            st[cur ^ 1][lane] = v;
        }
        __syncthreads();
        cur ^= 1;
    }
    if (lane == 0) {
        double z = st[cur][0];
        for (int b = 1; b < nbase; b++) z = lse64_l(z, st[cur][b]);
        *logz = z;
    }
}

__global__ void k_rle1_sub(float *__restrict__ C, size_t n, int nbase, int Ps, int nc, const double *__restrict__ logz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = (int)(i % Ps);
    if (p < 2 * nbase || p >= 4 * nbase) return;
    C[i] -= (float)(*logz / (double)(float)nc);              // layers.c:1216: `float logZ = partition / (float)C->nc`
}

}  // namespace

// gru_forward / gru_backward (relu = 0) and gru_relu_forward / gru_relu_backward (relu = 1): X [3H x T], sW [H x 2H], sW2 [H x H]
extern "C" int ffhip_op_gru(ffhip_engine *eng, int relu, ffhip_mat X, ffhip_mat sW, ffhip_mat sW2, int backward, ffhip_mat out) {
    OP_ENTER(eng);
    if (!view_ok(X) || !view_ok(sW) || !view_ok(sW2) || !view_ok(out)) return set_err(FFHIP_EINVAL, "bad GRU-layer arguments");
    const size_t H = sW2.nc;
    if (H % 4 != 0 || X.nr != 3 * H || sW.nr != H || sW2.nr != H || sW.nc != 2 * H || out.nr != H || out.nc != X.nc)
        return set_err(FFHIP_EINVAL, "GRU layer: shapes do not agree (layers.c:419-425)");
    if (5 * H * sizeof(float) > 64 * 1024) return set_err(FFHIP_EINVAL, "GRU layer: size %zu is beyond this operator (<= 3276)", H);
    float *d_x = upload_img(tmp, X, s), *d_w = upload_img(tmp, sW, s), *d_w2 = upload_img(tmp, sW2, s);
    float *d_o = (float *)tmp.get(out.nc * out.stride * 4);
    if (!d_x || !d_w || !d_w2 || !d_o) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_o, 0, out.nc * out.stride * 4, s), FFHIP_EHIP);
    if (relu) hipLaunchKernelGGL(k_gru_sloika<1>, dim3(1), dim3(1024), 5 * H * sizeof(float), s, d_x, X.stride, d_w, sW.stride, d_w2, sW2.stride, (const float *)nullptr, (int)H, (int)X.nc, backward, d_o, out.stride);
    else hipLaunchKernelGGL(k_gru_sloika<0>, dim3(1), dim3(1024), 5 * H * sizeof(float), s, d_x, X.stride, d_w, sW.stride, d_w2, sW2.stride, (const float *)nullptr, (int)H, (int)X.nc, backward, d_o, out.stride);
    HIP_TRY(hipMemcpyAsync(out.data, d_o, out.nc * out.stride * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    return FFHIP_OK;
}

// gru_step (layers.c:513-568) / gru_relu_step (layers.c:819-874): x [3H x 1], istate and ostate [H x 1]
extern "C" int ffhip_op_gru_step(ffhip_engine *eng, int relu, ffhip_mat x, ffhip_mat istate, ffhip_mat sW, ffhip_mat sW2, ffhip_mat ostate) {
    OP_ENTER(eng);
    if (!view_ok(x) || !view_ok(istate) || !view_ok(sW) || !view_ok(sW2) || !view_ok(ostate)) return set_err(FFHIP_EINVAL, "bad GRU-step arguments");
    const size_t H = istate.nr;
    if (H % 4 != 0 || x.nr != 3 * H || sW.nr != H || sW.nc != 2 * H || sW2.nr != H || sW2.nc != H || ostate.nr != H)
        return set_err(FFHIP_EINVAL, "GRU step: shapes do not agree (layers.c:529-538)");
    if (5 * H * sizeof(float) > 64 * 1024) return set_err(FFHIP_EINVAL, "GRU step: size %zu is beyond this operator (<= 3276)", H);
    float *d_x = (float *)tmp.upload(x.data, x.stride * 4, s), *d_h = (float *)tmp.upload(istate.data, istate.stride * 4, s);
    float *d_w = upload_img(tmp, sW, s), *d_w2 = upload_img(tmp, sW2, s), *d_o = (float *)tmp.get(ostate.stride * 4);
    if (!d_x || !d_h || !d_w || !d_w2 || !d_o) OP_NOMEM();
    if (relu) hipLaunchKernelGGL(k_gru_sloika<1>, dim3(1), dim3(1024), 5 * H * sizeof(float), s, d_x, x.stride, d_w, sW.stride, d_w2, sW2.stride, (const float *)d_h, (int)H, 1, 0, d_o, ostate.stride);
    else hipLaunchKernelGGL(k_gru_sloika<0>, dim3(1), dim3(1024), 5 * H * sizeof(float), s, d_x, x.stride, d_w, sW.stride, d_w2, sW2.stride, (const float *)d_h, (int)H, 1, 0, d_o, ostate.stride);
    HIP_TRY(hipMemcpyAsync(ostate.data, d_o, H * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    return FFHIP_OK;
}

// globalnorm_runlength (layers.c:1197-1228), the first-generation run-length head: C is [4*nbase x T]
extern "C" int ffhip_op_globalnorm_runlength_v1(ffhip_engine *eng, ffhip_mat X, ffhip_mat W, ffhip_mat b, float temperature, ffhip_mat C) {
    OP_ENTER(eng);
    if (!view_ok(X) || !view_ok(W) || !view_ok(b) || !view_ok(C)) return set_err(FFHIP_EINVAL, "bad globalnorm arguments");
    if (W.nr != X.nr || b.nr != W.nc || C.nr != W.nc || C.nc != X.nc) return set_err(FFHIP_EINVAL, "globalnorm: shapes do not agree");
    if (W.nc % 4 != 0 || W.nc / 4 > 64) return set_err(FFHIP_EINVAL, "globalnorm_runlength: %zu rows is not 4*nbase with nbase <= 64 (layers.c:1115-1119)", W.nc);
    const int nbase = (int)(W.nc / 4);
    const int H = (int)X.nr, Hp = round_up(H, 16), K16 = Hp / 16, P = (int)W.nc, Mt = (P + 15) / 16, T = (int)X.nc;
    std::vector<float> wp = pack_weight_T(W, Mt, K16);
    std::vector<float> bias((size_t)Mt * 16, 0.0f);
    for (int p = 0; p < P; p++) bias[p] = b.data[p];
    float *d_x = upload_img(tmp, X, s);
    float *d_w = (float *)tmp.upload(wp.data(), wp.size() * 4, s), *d_b = (float *)tmp.upload(bias.data(), bias.size() * 4, s);
    float *d_in = (float *)tmp.get((size_t)T * K16 * 256 * 4), *d_c = (float *)tmp.get(C.nc * C.stride * 4);
    double *d_z = (double *)tmp.get(sizeof(double));
    if (!d_x || !d_w || !d_b || !d_in || !d_c || !d_z) OP_NOMEM();
    const size_t n = C.nc * C.stride;
    HIP_TRY(hipMemsetAsync(d_c, 0, n * 4, s), FFHIP_EHIP);
    hipLaunchKernelGGL(k_img_to_tiles, dim3(nblk((size_t)T * K16 * 64)), dim3(256), 0, s, d_x, X.stride, H, T, K16, K16, 0, 1, T, d_in);
    launch_head(s, d_in, d_c, (const float4 *)d_w, d_b, T, 1, 1, P, (int)C.stride, K16, 1.0f, 1);
    hipLaunchKernelGGL(k_rle1_activate, dim3(nblk(n)), dim3(256), 0, s, d_c, n, nbase, (int)C.stride, temperature);
    hipLaunchKernelGGL(k_rle1_partition, dim3(1), dim3(64), 0, s, (const float *)d_c, T, nbase, (int)C.stride, d_z);
    hipLaunchKernelGGL(k_rle1_sub, dim3(nblk(n)), dim3(256), 0, s, d_c, n, nbase, (int)C.stride, T, (const double *)d_z);
    HIP_TRY(hipMemcpyAsync(C.data, d_c, n * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    return FFHIP_OK;
}

// runlength_partition_function (layers.c:1127-1174)
extern "C" int ffhip_op_runlength_partition_function_v1(ffhip_engine *eng, ffhip_mat S, double *logZ) {
    OP_ENTER(eng);
    if (!view_ok(S) || !logZ || S.nr % 4 != 0 || S.nr / 4 > 64) return set_err(FFHIP_EINVAL, "bad partition-function arguments");
    float *d = upload_img(tmp, S, s);
    double *d_z = (double *)tmp.get(sizeof(double));
    if (!d || !d_z) OP_NOMEM();
    hipLaunchKernelGGL(k_rle1_partition, dim3(1), dim3(64), 0, s, (const float *)d, (int)S.nc, (int)(S.nr / 4), (int)S.stride, d_z);
    HIP_TRY(hipMemcpyAsync(logZ, d_z, sizeof(double), hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// ------------------------------------------------------------------------------------ the other flip-flop decoders of decode.h
// argmax_decoder (decode.c:17-36), constrained_crf_flipflop (:209-270) and posterior_crf_flipflop (:275-372): entry points of
// decode.h that flappie.c never calls (its path is transpost_crf_flipflop + decode_crf_flipflop).  They are here so that a
// caller of the whole header links and gets the reference's results; one wave per call, the recursions in the
// reference's order -- correctness-level operators, not throughput paths.
namespace {

// argmaxf (util.c:17-34): first maximum, a NaN is never greater
__global__ void __launch_bounds__(256)
k_argmax_cols(const float *__restrict__ X, size_t stride, int nr, int nc, int *__restrict__ seq, float *__restrict__ val) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    const float *x = X + (size_t)c * stride;
    int imax = 0;
    float vmax = x[0];
    for (int i = 1; i < nr; i++) if (x[i] > vmax) { vmax = x[i]; imax = i; }
    seq[c] = (imax == nr - 1) ? -1 : imax;              // decode.c:32: the last state is the blank
    val[c] = vmax;
}
// the reference adds the block maxima one by one in block order (decode.c:31): a float sum, so the order is the result
__global__ void k_sum_in_order(const float *__restrict__ v, int n, float *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float acc = 0.0f;
    for (int i = 0; i < n; i++) acc += v[i];
    out[0] = acc;
}

// constrained_crf_flipflop (decode.c:209-270): Viterbi over per-state posteriors with the flip-flop transition constraint.
// One wave, lane st < nstate holds prev[st].  Flop b2 comes from itself if prev[b2] > prev[b2 - nbase], else from the flip of
// its base (:239-243); every flip comes from the best state of all (argmaxf: first maximum, :246-250).
__global__ void __launch_bounds__(64)
k_constrained_flipflop(const float *__restrict__ post, size_t stride, int nstate, int nblk, int *__restrict__ tb /*[nblk][16]*/,
                       int *__restrict__ path, float *__restrict__ score) {
    const int lane = threadIdx.x, nbase = nstate / 2;
    const bool on = lane < nstate;
    float prev = 0.0f;                                  // calloc'ed (:219)
    for (int blk = 0; blk < nblk; blk++) {
        int best = 0;
        float vbest = __shfl(prev, 0);
        for (int st = 1; st < nstate; st++) {
            const float v = __shfl(prev, st);
            if (v > vbest) { vbest = v; best = st; }
        }
        const float flip_of_base = __shfl(prev, lane >= nbase ? lane - nbase : lane);
        int from;
        float cur;
        if (lane >= nbase) { from = (prev > flip_of_base) ? lane : lane - nbase; cur = (prev > flip_of_base) ? prev : flip_of_base; }
        else { from = best; cur = vbest; }
        if (on) {
            tb[(size_t)blk * 16 + lane] = from;
            cur += post[(size_t)blk * stride + lane];
        }
        prev = cur;
    }
    // :256-262 valmaxf / argmaxf over the final vector, then the traceback
    int last = 0;
    float vlast = __shfl(prev, 0);
    for (int st = 1; st < nstate; st++) {
        const float v = __shfl(prev, st);
        if (v > vlast) { vlast = v; last = st; }
    }
    __syncthreads();
    if (lane == 0) {
        score[0] = vlast;
        path[nblk] = last;
        for (int blk = nblk; blk > 0; blk--) path[blk - 1] = tb[(size_t)(blk - 1) * 16 + path[blk]];
    }
}

// posterior_crf_flipflop (decode.c:275-372): per-STATE log posteriors [nstate x nblk+1] = forward + backward, every
// log-sum-exp chain in the reference's order (flip: over from-states ascending; backward: flop terms first, then the flips
// ascending).  One wave; lane st owns state st in both passes.
__global__ void __launch_bounds__(64)
k_posterior_flipflop(const float *__restrict__ trans, size_t tstride, int nbase, int nblk, float *__restrict__ fwd, size_t fstride) {
    const int lane = threadIdx.x, nstate = 2 * nbase, off = nstate * nbase;
    const bool on = lane < nstate;
    __shared__ float pv[16];
    if (lane < 16) pv[lane] = 0.0f;                     // make_flappie_matrix zero-fills column 0
    if (on) fwd[lane] = 0.0f;
    __syncthreads();
    for (int blk = 0; blk < nblk; blk++) {              // forwards, :288-315
        const float *t = trans + (size_t)blk * tstride;
        float cur = 0.0f;
        if (on) {
            if (lane >= nbase) {
                cur = pv[lane] + t[off + lane];                                           // stay in flop
                cur = logsumexpf_ref(cur, pv[lane - nbase] + t[off + lane - nbase]);      // flip -> flop
            } else {
                cur = t[lane * nstate] + pv[0];
                for (int from = 1; from < nstate; from++) cur = logsumexpf_ref(cur, t[lane * nstate + from] + pv[from]);
            }
        }
        __syncthreads();
        if (on) { pv[lane] = cur; fwd[(size_t)(blk + 1) * fstride + lane] = cur; }
        __syncthreads();
    }
    if (lane < 16) pv[lane] = 0.0f;                     // backwards from zeros (calloc, :317)
    __syncthreads();
    for (int blk = nblk; blk > 0; blk--) {              // :326-366
        const float *t = trans + (size_t)(blk - 1) * tstride;
        float cur = 0.0f;
        if (on) {
            // :339-345  source lane: a flop state stays, a flip state moves to the flop of its base
            cur = (lane >= nbase) ? pv[lane] + t[off + lane] : pv[lane + nbase] + t[off + lane];
            for (int b1 = 0; b1 < nbase; b1++) cur = logsumexpf_ref(cur, t[b1 * nstate + lane] + pv[b1]);      // :348-356
        }
        __syncthreads();
        if (on) { pv[lane] = cur; fwd[(size_t)(blk - 1) * fstride + lane] += cur; }      // :358-361
        __syncthreads();
    }
}

}  // namespace

// argmax_decoder (decode.c:17-36): seq[nc] (index of the column maximum, -1 for the last row), *score = sum of the maxima
extern "C" int ffhip_op_argmax_decoder(ffhip_engine *eng, ffhip_mat logpost, int *seq, float *score) {
    OP_ENTER(eng);
    if (!view_ok(logpost) || !seq || !score) return set_err(FFHIP_EINVAL, "bad argmax_decoder arguments");
    const int nc = (int)logpost.nc;
    float *d = upload_img(tmp, logpost, s), *d_v = (float *)tmp.get((size_t)nc * 4), *d_s = (float *)tmp.get(4);
    int *d_q = (int *)tmp.get((size_t)nc * 4);
    if (!d || !d_v || !d_s || !d_q) OP_NOMEM();
    hipLaunchKernelGGL(k_argmax_cols, dim3((nc + 255) / 256), dim3(256), 0, s, d, logpost.stride, (int)logpost.nr, nc, d_q, d_v);
    hipLaunchKernelGGL(k_sum_in_order, dim3(1), dim3(64), 0, s, d_v, nc, d_s);
    HIP_TRY(hipMemcpyAsync(seq, d_q, (size_t)nc * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(score, d_s, 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// constrained_crf_flipflop (decode.c:209-270): post [nstate x nblk] per-state scores; path[nblk + 1]
extern "C" int ffhip_op_constrained_flipflop(ffhip_engine *eng, ffhip_mat post, int *path, float *score) {
    OP_ENTER(eng);
    if (!view_ok(post) || !path || !score || post.nr % 2 != 0 || post.nr > 16) return set_err(FFHIP_EINVAL, "bad constrained_crf_flipflop arguments");
    const size_t nblk = post.nc;
    float *d = upload_img(tmp, post, s), *d_s = (float *)tmp.get(4);
    int *d_tb = (int *)tmp.get(nblk * 16 * 4), *d_p = (int *)tmp.get((nblk + 1) * 4);
    if (!d || !d_s || !d_tb || !d_p) OP_NOMEM();
    hipLaunchKernelGGL(k_constrained_flipflop, dim3(1), dim3(64), 0, s, d, post.stride, (int)post.nr, (int)nblk, d_tb, d_p, d_s);
    HIP_TRY(hipMemcpyAsync(path, d_p, (nblk + 1) * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(score, d_s, 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// posterior_crf_flipflop (decode.c:275-372), log space: out is [nstate x nblk + 1] (column stride out.stride)
extern "C" int ffhip_op_posterior_flipflop(ffhip_engine *eng, ffhip_mat trans, ffhip_mat out) {
    OP_ENTER(eng);
    const bool lazy_ = mat_has_dev(trans);
    int nbase;
    if (!view_ok(trans) || !view_ok(out) || !flipflop_dims(trans.nr, trans.stride, &nbase) || out.nr != (size_t)(2 * nbase) || out.nc != trans.nc + 1)
        return set_err(FFHIP_EINVAL, "bad posterior_crf_flipflop arguments");
    const size_t n = out.nc * out.stride;
    float *d = upload_img(tmp, trans, s), *d_o = mat_out(tmp, out, lazy_);
    if (!d || !d_o) OP_NOMEM();
    HIP_TRY(hipMemsetAsync(d_o, 0, n * 4, s), FFHIP_EHIP);
    hipLaunchKernelGGL(k_posterior_flipflop, dim3(1), dim3(64), 0, s, d, trans.stride, nbase, (int)trans.nc, d_o, out.stride);
    HIP_TRY(mat_done(out, d_o, lazy_, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

// ---- development probe: packed-fp32 VALU instructions next to another kernel's MFMAs ----------------
// Every op_sel / op_sel_hi form of v_pk_add_f32, v_pk_mul_f32 and v_pk_fma_f32 against the scalar instruction on the selected
// halves, in a loop, on its own stream; mismatches are counted by (instruction, form, result half, quarter of the wave).
namespace ffhip {
typedef float pkf2 __attribute__((ext_vector_type(2)));
#define PK_FORM(OPI, MNE, F, S0, S1, H0, H1)                                                                                                  \
    {                                                                                                                                       \
        pkf2 r;                                                                                                                             \
        if (OPI == 2) asm volatile(MNE " %0, %1, %2, %3 op_sel:[" #S0 "," #S1 ",0] op_sel_hi:[" #H0 "," #H1 ",1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); \
        else asm volatile(MNE " %0, %1, %2 op_sel:[" #S0 "," #S1 "] op_sel_hi:[" #H0 "," #H1 "]" : "=v"(r) : "v"(a), "v"(b));                 \
        const float elo = ref(OPI, S0 ? a.y : a.x, S1 ? b.y : b.x, c.x), ehi = ref(OPI, H0 ? a.y : a.x, H1 ? b.y : b.x, c.y);                  \
        if (__float_as_uint(r.x) != __float_as_uint(elo)) atomicAdd(&counts[(((OPI) * 16 + (F)) * 2 + 0) * 4 + quarter], 1u);                 \
        if (__float_as_uint(r.y) != __float_as_uint(ehi)) atomicAdd(&counts[(((OPI) * 16 + (F)) * 2 + 1) * 4 + quarter], 1u);                 \
    }
#define PK_ALL(OPI, MNE)                                                                                                                     \
    PK_FORM(OPI, MNE, 0, 0, 0, 0, 0) PK_FORM(OPI, MNE, 1, 0, 0, 0, 1) PK_FORM(OPI, MNE, 2, 0, 0, 1, 0) PK_FORM(OPI, MNE, 3, 0, 0, 1, 1)           \
    PK_FORM(OPI, MNE, 4, 0, 1, 0, 0) PK_FORM(OPI, MNE, 5, 0, 1, 0, 1) PK_FORM(OPI, MNE, 6, 0, 1, 1, 0) PK_FORM(OPI, MNE, 7, 0, 1, 1, 1)           \
    PK_FORM(OPI, MNE, 8, 1, 0, 0, 0) PK_FORM(OPI, MNE, 9, 1, 0, 0, 1) PK_FORM(OPI, MNE, 10, 1, 0, 1, 0) PK_FORM(OPI, MNE, 11, 1, 0, 1, 1)         \
    PK_FORM(OPI, MNE, 12, 1, 1, 0, 0) PK_FORM(OPI, MNE, 13, 1, 1, 0, 1) PK_FORM(OPI, MNE, 14, 1, 1, 1, 0) PK_FORM(OPI, MNE, 15, 1, 1, 1, 1)

template <int NV>
__global__ void __launch_bounds__(256) k_pk_probe(unsigned *counts, int iters) {
    auto ref = [](int op, float x, float y, float z) {
        float r;
        if (op == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
        else if (op == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
        else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
        return r;
    };
    const int lane = threadIdx.x & 63, quarter = lane >> 4;
    float pad[NV];                    // register ballast: the probe's VGPR allocation decides which waves it can sit next to
#pragma unroll
    for (int k = 0; k < NV; k++) pad[k] = (float)(threadIdx.x * (k + 1));
    for (int it = 0; it < iters; it++) {
        const float t = (float)(it & 1023);
        const pkf2 a = { 1.25f * lane + t, 1000.0f + 0.5f * lane - t }, b = { 3.0f + 0.125f * lane, -77.0f + t + 2.0f * lane }, c = { 0.5f * t, 9.0f - lane };
        PK_ALL(0, "v_pk_add_f32")
        PK_ALL(1, "v_pk_mul_f32")
        PK_ALL(2, "v_pk_fma_f32")
#define PK_MOV(F, S0, S1)                                                                                                                    \
        {                                                                                                                                   \
            pkf2 r;                                                                                                                         \
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[" #S0 "," #S1 "]" : "=v"(r) : "v"(a), "v"(b));                                      \
            if (__float_as_uint(r.x) != __float_as_uint(S0 ? a.y : a.x)) atomicAdd(&counts[((3 * 16 + (F)) * 2 + 0) * 4 + quarter], 1u);      \
            if (__float_as_uint(r.y) != __float_as_uint(S1 ? b.y : b.x)) atomicAdd(&counts[((3 * 16 + (F)) * 2 + 1) * 4 + quarter], 1u);      \
        }
        PK_MOV(0, 0, 0) PK_MOV(4, 0, 1) PK_MOV(8, 1, 0) PK_MOV(12, 1, 1)
        // v_pk_fma_f32 with the LOW result taking the HIGH half of source 2 (counted under instruction 3, forms 1 and 2)
#define PK_FMA2(F, S0, S1)                                                                                                                   \
        {                                                                                                                                   \
            pkf2 r;                                                                                                                         \
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[" #S0 "," #S1 ",1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));       \
            if (__float_as_uint(r.x) != __float_as_uint(ref(2, S0 ? a.y : a.x, S1 ? b.y : b.x, c.y))) atomicAdd(&counts[((3 * 16 + (F)) * 2 + 0) * 4 + quarter], 1u); \
            if (__float_as_uint(r.y) != __float_as_uint(ref(2, a.y, b.y, c.x))) atomicAdd(&counts[((3 * 16 + (F)) * 2 + 1) * 4 + quarter], 1u); \
        }
        PK_FMA2(1, 0, 0) PK_FMA2(2, 1, 1)
#pragma unroll
        for (int k = 0; k < NV; k++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(pad[k]) : "v"(t));
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; k++) s += pad[k];
    if (s == 12345.678f) counts[0] = 0xFFFFFFFFu;       // keeps the ballast alive
}
}  // namespace ffhip

extern "C" int ffhip_debug_pk_probe(ffhip_engine *eng, int iters, int nwg, int ballast, unsigned *counts) {
    using namespace ffhip;
    if (!eng || !counts || iters <= 0 || nwg <= 0) return FFHIP_EINVAL;
    hipSetDevice(eng->device);
    unsigned *d = nullptr;
    hipStream_t s = nullptr;
    if (hipMalloc(&d, 4 * 16 * 2 * 4 * 4) != hipSuccess) return FFHIP_ENOMEM;
    hipMemset(d, 0, 4 * 16 * 2 * 4 * 4);
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (ballast >= 96) hipLaunchKernelGGL(k_pk_probe<96>, dim3(nwg), dim3(256), 0, s, d, iters);
    else if (ballast >= 48) hipLaunchKernelGGL(k_pk_probe<48>, dim3(nwg), dim3(256), 0, s, d, iters);
    else hipLaunchKernelGGL(k_pk_probe<1>, dim3(nwg), dim3(256), 0, s, d, iters);
    hipStreamSynchronize(s);
    hipMemcpy(counts, d, 4 * 16 * 2 * 4 * 4, hipMemcpyDeviceToHost);
    hipStreamDestroy(s);
    hipFree(d);
    return FFHIP_OK;
}
