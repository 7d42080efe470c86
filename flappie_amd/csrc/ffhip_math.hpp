// ffhip_math.hpp -- element-wise device math with the reference's semantics.
//
// The reference evaluates exp through the cephes polynomial `exp_ps` (sse_mathfun.h:225-301) and
// builds logistic / tanh / swish from it (util.h:329-337, layers.c:24-49), compiled WITHOUT FMA
// (-march=ivybridge, CMakeLists.txt:115).  These functions restate that arithmetic operation for
// operation so that, for equal inputs, the GPU produces bit-identical activations.  The translation
// unit is compiled with -ffp-contract=off; fp32 division is IEEE (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>

namespace ffhip {

__device__ __forceinline__ float exp_cephes(float x) {
    x = (x < 88.3762626647949f) ? x : 88.3762626647949f;
    x = (x > -88.3762626647949f) ? x : -88.3762626647949f;
    float fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    int n = __float2int_rz(fx);
    float tmp = (float)n;
    fx = tmp - ((tmp > fx) ? 1.0f : 0.0f);
    tmp = fx * 0.693359375f;
    float z = fx * -2.12194440e-4f;
    x = x - tmp;
    x = x - z;
    z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x; y = y + 1.3981999507E-3f;
    y = y * x; y = y + 8.3334519073E-3f;
    y = y * x; y = y + 4.1665795894E-2f;
    y = y * x; y = y + 1.6666665459E-1f;
    y = y * x; y = y + 5.0000001201E-1f;
    y = y * z;
    y = y + x;
    y = y + 1.0f;
    n = __float2int_rz(fx);
    return y * __int_as_float((n + 0x7f) << 23);
}

// log_ps, sse_mathfun.h:123-208: x <= 0 gives NaN (all-ones mask), denormals are clamped to FLT_MIN
__device__ __forceinline__ float log_cephes(float x) {
    const bool invalid = (x <= 0.0f);
    x = (x > 1.17549435e-38f) ? x : 1.17549435e-38f;
    const int bits = __float_as_int(x);
    float e = (float)((int)((unsigned)bits >> 23) - 0x7f);
    x = __int_as_float((bits & ~0x7f800000) | 0x3f000000);     // mantissa in [0.5, 1)
    e = e + 1.0f;
    const bool small = (x < 0.707106781186547524f);
    const float tmp0 = small ? x : 0.0f;
    x = x - 1.0f;
    e = e - (small ? 1.0f : 0.0f);
    x = x + tmp0;
    const float z = x * x;
    float y = 7.0376836292E-2f;
    y = y * x; y = y + -1.1514610310E-1f;
    y = y * x; y = y + 1.1676998740E-1f;
    y = y * x; y = y + -1.2420140846E-1f;
    y = y * x; y = y + 1.4249322787E-1f;
    y = y * x; y = y + -1.6668057665E-1f;
    y = y * x; y = y + 2.0000714765E-1f;
    y = y * x; y = y + -2.4999993993E-1f;
    y = y * x; y = y + 3.3333331174E-1f;
    y = y * x;
    y = y * z;
    float tmp = e * -2.12194440e-4f;
    y = y + tmp;
    tmp = z * 0.5f;
    y = y - tmp;
    tmp = e * 0.693359375f;
    x = x + y;
    x = x + tmp;
    return invalid ? __int_as_float(-1) : x;
}

__device__ __forceinline__ float logistic_ref(float x) { return 1.0f / (1.0f + exp_cephes(-x)); }

__device__ __forceinline__ float tanh_ref(float x) {
    const float y = logistic_ref(x + x);
    return (y + y) - 1.0f;
}

__device__ __forceinline__ float swish_ref(float x) { return x * logistic_ref(x); }

// util.h:276-282 (libm expf/log1pf on the host; ocml expf/log1pf here, <= 1-2 ulp apart)
__device__ __forceinline__ float logsumexpf_ref(float x, float y) {
    return fmaxf(x, y) + log1pf(expf(-fabsf(x - y)));
}

// Four independent logistic evaluations written as one straight-line vector computation: the same
// operations in the same order per component as logistic_ref (bit-identical results), but the
// four dependency chains are interleaved by construction, which is what a single wave on the
// recurrent critical path needs (latency-bound, one element per lane).
typedef float ffv4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ ffv4 exp_cephes4(ffv4 x) {
    const ffv4 hi = { 88.3762626647949f, 88.3762626647949f, 88.3762626647949f, 88.3762626647949f };
    x = __builtin_elementwise_min(x, hi);
    x = __builtin_elementwise_max(x, -hi);
    ffv4 fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    ffv4 tmp;
    tmp.x = (float)__float2int_rz(fx.x); tmp.y = (float)__float2int_rz(fx.y);
    tmp.z = (float)__float2int_rz(fx.z); tmp.w = (float)__float2int_rz(fx.w);
    ffv4 one_if_gt;
    one_if_gt.x = (tmp.x > fx.x) ? 1.0f : 0.0f; one_if_gt.y = (tmp.y > fx.y) ? 1.0f : 0.0f;
    one_if_gt.z = (tmp.z > fx.z) ? 1.0f : 0.0f; one_if_gt.w = (tmp.w > fx.w) ? 1.0f : 0.0f;
    fx = tmp - one_if_gt;
    tmp = fx * 0.693359375f;
    ffv4 z = fx * -2.12194440e-4f;
    x = x - tmp;
    x = x - z;
    z = x * x;
    ffv4 y = { 1.9875691500E-4f, 1.9875691500E-4f, 1.9875691500E-4f, 1.9875691500E-4f };
    y = y * x; y = y + 1.3981999507E-3f;
    y = y * x; y = y + 8.3334519073E-3f;
    y = y * x; y = y + 4.1665795894E-2f;
    y = y * x; y = y + 1.6666665459E-1f;
    y = y * x; y = y + 5.0000001201E-1f;
    y = y * z;
    y = y + x;
    y = y + 1.0f;
    ffv4 p2;
    p2.x = __int_as_float((__float2int_rz(fx.x) + 0x7f) << 23);
    p2.y = __int_as_float((__float2int_rz(fx.y) + 0x7f) << 23);
    p2.z = __int_as_float((__float2int_rz(fx.z) + 0x7f) << 23);
    p2.w = __int_as_float((__float2int_rz(fx.w) + 0x7f) << 23);
    return y * p2;
}

__device__ __forceinline__ ffv4 logistic_ref4(ffv4 x) {
    const ffv4 e = exp_cephes4(-x);
    ffv4 r;
    r.x = 1.0f / (1.0f + e.x); r.y = 1.0f / (1.0f + e.y); r.z = 1.0f / (1.0f + e.z); r.w = 1.0f / (1.0f + e.w);
    return r;
}

// ---- the same results with fewer instructions, for the gate phase of the persistent layer kernels -------------------
// 1 / d for 1 <= d <= 2^126, correctly rounded (= the IEEE quotient 1.0f / d, bit for bit): hardware reciprocal, one
// Newton step, and the closing fused step q + (1 - d q) q of the standard division expansion.  What the compiler's
// expansion adds on top -- v_div_scale x2, v_div_fmas, v_div_fixup -- only serves operands outside this range.
// Checked exhaustively over every mantissa at several exponents by ffhip_debug_recip_check (tests/test_split_gpu.py).
template <int STEPS = 1>
__device__ __forceinline__ float recip_1_to_2p126(float d) {
    float r = __builtin_amdgcn_rcpf(d);
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
        const float e = __builtin_fmaf(-d, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    const float e2 = __builtin_fmaf(-d, r, 1.0f);      // exact: r is within an ulp of 1/d
    return __builtin_fmaf(e2, r, r);
}

// exp_cephes4 with the reference's truncate / compare / subtract floor written as floor (the same value for every
// input: |fx| <= 128 here) -- 1 instruction instead of 5 per component
__device__ __forceinline__ ffv4 exp_cephes4_floor(ffv4 x) {
    const ffv4 hi = { 88.3762626647949f, 88.3762626647949f, 88.3762626647949f, 88.3762626647949f };
    x = __builtin_elementwise_min(x, hi);
    x = __builtin_elementwise_max(x, -hi);
    ffv4 fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    fx = __builtin_elementwise_floor(fx);
    ffv4 tmp = fx * 0.693359375f;
    ffv4 z = fx * -2.12194440e-4f;
    x = x - tmp;
    x = x - z;
    z = x * x;
    ffv4 y = { 1.9875691500E-4f, 1.9875691500E-4f, 1.9875691500E-4f, 1.9875691500E-4f };
    y = y * x; y = y + 1.3981999507E-3f;
    y = y * x; y = y + 8.3334519073E-3f;
    y = y * x; y = y + 4.1665795894E-2f;
    y = y * x; y = y + 1.6666665459E-1f;
    y = y * x; y = y + 5.0000001201E-1f;
    y = y * z;
    y = y + x;
    y = y + 1.0f;
    ffv4 p2;
    p2.x = __int_as_float((__float2int_rz(fx.x) + 0x7f) << 23);
    p2.y = __int_as_float((__float2int_rz(fx.y) + 0x7f) << 23);
    p2.z = __int_as_float((__float2int_rz(fx.z) + 0x7f) << 23);
    p2.w = __int_as_float((__float2int_rz(fx.w) + 0x7f) << 23);
    return y * p2;
}

// logistic_ref4, bit for bit, through the two functions above; a lane whose denominator 1 + exp(-x) exceeds 2^126
// (x < -87.3) sends the wave through the general division
__device__ __forceinline__ ffv4 logistic_ref4_lean(ffv4 x) {
    const ffv4 e = exp_cephes4_floor(-x);
    const ffv4 d = e + 1.0f;
    ffv4 r;
    const float big = 8.5070592e37f;      // 2^126
    if (__builtin_expect(__any(d.x > big || d.y > big || d.z > big || d.w > big), 0)) {
        r.x = 1.0f / d.x; r.y = 1.0f / d.y; r.z = 1.0f / d.z; r.w = 1.0f / d.w;
    } else {
        r.x = recip_1_to_2p126<>(d.x); r.y = recip_1_to_2p126<>(d.y); r.z = recip_1_to_2p126<>(d.z); r.w = recip_1_to_2p126<>(d.w);
    }
    return r;
}

// the same for one value and for two (a 4-vector with idle components costs its full instruction count)
__device__ __forceinline__ float exp_cephes_floor1(float x) {
    x = __builtin_fminf(x, 88.3762626647949f);
    x = __builtin_fmaxf(x, -88.3762626647949f);
    float fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    fx = __builtin_floorf(fx);
    const float tmp = fx * 0.693359375f;
    float z = fx * -2.12194440e-4f;
    x = x - tmp;
    x = x - z;
    z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x; y = y + 1.3981999507E-3f;
    y = y * x; y = y + 8.3334519073E-3f;
    y = y * x; y = y + 4.1665795894E-2f;
    y = y * x; y = y + 1.6666665459E-1f;
    y = y * x; y = y + 5.0000001201E-1f;
    y = y * z;
    y = y + x;
    y = y + 1.0f;
    return y * __int_as_float((__float2int_rz(fx) + 0x7f) << 23);
}

__device__ __forceinline__ float logistic_ref_lean(float x) {
    const float d = exp_cephes_floor1(-x) + 1.0f;
    if (__builtin_expect(__any(d > 8.5070592e37f), 0)) return 1.0f / d;
    return recip_1_to_2p126<>(d);
}

typedef float ffv2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ffv2 logistic_ref2_lean(ffv2 x) {
    ffv2 r;
    // two independent scalar chains: the compiler packs the multiplies and adds of the pair (v_pk_mul_f32 / v_pk_add_f32)
    const float d0 = exp_cephes_floor1(-x.x) + 1.0f, d1 = exp_cephes_floor1(-x.y) + 1.0f;
    if (__builtin_expect(__any(d0 > 8.5070592e37f || d1 > 8.5070592e37f), 0)) { r.x = 1.0f / d0; r.y = 1.0f / d1; }
    else { r.x = recip_1_to_2p126<>(d0); r.y = recip_1_to_2p126<>(d1); }
    return r;
}

__device__ __forceinline__ float tanh_ref_lean(float x) {
    const float y = logistic_ref_lean(x + x);
    return (y + y) - 1.0f;
}

// ---- hardware gate math (opt-in in the layer kernels: FFHIP_RUN_FAST_GATES) -------------------------------------------
// logistic through v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the cephes replay: 6 instructions instead of ~35.  NOT
// bit-compatible with exp_ps -- results move by ~1e-7 per activation, a tenth of what the summation order of a dot product
// moves them.  The clamp keeps the reference's NaN behaviour: min/max return the finite bound for a NaN input, as
// _mm_min_ps / _mm_max_ps do in exp_ps (sse_mathfun.h:228-229), so a NaN pre-activation gives a finite gate here too.
// `level` (wave-uniform): 1 = the six instructions above; 2 = the argument of v_exp_f32 carried in two words and one Newton step behind v_rcp_f32
// (round 6).  Level 1 rounds t = -x log2(e) once, an error of |t| 2^-24 in the EXPONENT, i.e. a relative error of |x| 6e-8 in exp(-x) on top of the
// instruction's own ulp; level 2 keeps the product's residual (one fma recovers it exactly) and the low word of log2(e) and multiplies exp2(t_hi) by
// 1 + t_lo ln 2: exp(-x) to ~1 ulp at every |x|, the reciprocal to ~0.5 ulp -- the error budget of the reference's own cephes polynomial + division.
__device__ __forceinline__ float logistic_hw(float x, int level = 1) {
    float v = __builtin_fminf(-x, 88.3762626647949f);
    v = __builtin_fmaxf(v, -88.3762626647949f);
    const float t = v * 1.44269504088896341f;
    float e = __builtin_amdgcn_exp2f(t);
    if (level >= 2) {
        const float lo = __builtin_fmaf(v, 1.925963033500138e-08f, __builtin_fmaf(v, 1.44269504088896341f, -t));      // log2(e) = 0x1.715476p+0 + 1.92596e-8
        e = __builtin_fmaf(e, lo * 0.693147180559945309f, e);
    }
    const float d = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(d);
    if (level >= 2) r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return r;
}
__device__ __forceinline__ float tanh_hw(float x, int level = 1) {
    const float y = logistic_hw(x + x, level);
    return (y + y) - 1.0f;
}

// four activations at once through the lean logistic (bit-identical to swish_ref / tanh_ref, ~50 fewer instructions per four:
// the epilogues of the convolution kernels evaluate 80 million of them per headline batch)
__device__ __forceinline__ ffv4 apply_act4(ffv4 x, int act) {
    if (act == 1) { const ffv4 L = logistic_ref4_lean(x); return x * L; }
    if (act == 2) { const ffv4 L = logistic_ref4_lean(x + x); return (L + L) - 1.0f; }
    return x;
}

__device__ __forceinline__ float apply_act(float x, int act) {
    return act == 1 ? swish_ref(x) : (act == 2 ? tanh_ref(x) : x);
}

}  // namespace ffhip
