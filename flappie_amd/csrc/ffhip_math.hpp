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

__device__ __forceinline__ float apply_act(float x, int act) {
    return act == 1 ? swish_ref(x) : (act == 2 ? tanh_ref(x) : x);
}

}  // namespace ffhip
