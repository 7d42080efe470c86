// ffhip_split.hpp -- operand format of the split-precision recurrent layer kernels (ffhip_rnn_split.hip), shared with the
// producers of that format (last convolution's epilogue in ffhip_kernels.hip, weight packer in ffhip_engine.hip).
//
// An fp32 operand v travels as kSplitNS 16-bit slices of v * 2^e whose sum is v * 2^e (to 2^-22 relative, or exactly):
//
//   default       two fp16 slices   h0 = f16(v'), h1 = f16(v' - h0), v' = v * 2^e           4 bytes per value
//                 products kept: w0 x0, w0 x1, w1 x0 (each exact in the f32 accumulator of v_mfma_f32_16x16x32_f16); the
//                 dropped w1 x1 and the two representation errors are <= 2^-21 of a product -- K = 768 dot products of
//                 this network's magnitudes come out at 2.1e-6 of float64, a plain fp32 GEMM at 2.4e-6, the reference's
//                 sequential sgemv order at 3.0e-6 (tests/test_split_numerics.py): THREE matrix instructions per fp32
//                 multiply-add block with the accuracy of fp32 arithmetic.
//   FFHIP_SPLIT_BF16X3   three bf16 slices (8 + 8 + 8 mantissa bits hold any fp32 exactly), six products kept: every
//                 product fp32-exact, 8.3e-7 on the same test -- the round-1 formulation, twice the matrix work and 1.5x
//                 the operand bytes; kept buildable (-DFFHIP_SPLIT_BF16X3) as the cross-check of the default.
//
// fp16 has a 5-bit exponent, so operands are scaled by powers of two (exact) into its range before they are split:
//   activations bounded by 1 (LSTM / GRUmod outputs, tanh convolution output)   2^kSplitExpH = 4096: second slice normal
//                 down to |v| = 2^-15, absolute error floor 2^-37;
//   swish convolution output (unbounded above)                                   2^kSplitExpX = 16, clamped at +-65504/16 =
//                 +-4094 (NaN stays NaN); error floor 2^-29;
//   each weight matrix                                                           2^sw with max |w| 2^sw in [2^14, 2^15).
// A layer's two products Wi x and sW h carry the same total exponent S = swi + e_x = sws + e_h (ffhip_engine.hip picks
// the weight exponents that way), the accumulators live in the scaled space (bias pre-multiplied by 2^S, exact) and
// the gate phase multiplies by 2^-S (exact) -- so scaling changes no rounding anywhere.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FFHIP_HD __host__ __device__
#else
#define FFHIP_HD              // the host-side helpers also compile with a plain C++ compiler (tests/test_cabi_and_model.py)
#endif
#include <stdint.h>

namespace ffhip {

#ifdef FFHIP_SPLIT_BF16X3
constexpr int kSplitNS = 3;          // slices per value
constexpr int kSplitNT = 6;          // matrix products per K chunk and accumulator
constexpr int kSplitExpH = 0, kSplitExpX = 0;
constexpr bool kSplitF16 = false;
#else
constexpr int kSplitNS = 2;
constexpr int kSplitNT = 3;
constexpr int kSplitExpH = 12, kSplitExpX = 4;
constexpr bool kSplitF16 = true;
#endif
constexpr unsigned kSplitSentinel = 0xFFFFFFFFu;      // two 16-bit NaNs in either format: never a pair of slices of a finite value
// products of one K chunk, smallest first: slice of w, slice of x
#ifdef FFHIP_SPLIT_BF16X3
#define FFHIP_SPLIT_TERMS_W { 2, 0, 1, 1, 0, 0 }
#define FFHIP_SPLIT_TERMS_X { 0, 2, 1, 0, 1, 0 }
#else
#define FFHIP_SPLIT_TERMS_W { 1, 0, 0 }
#define FFHIP_SPLIT_TERMS_X { 0, 1, 0 }
#endif

FFHIP_HD inline float split_pow2(int e) {          // 2^e, |e| <= 126
    union { uint32_t u; float f; } c;
    c.u = (uint32_t)(127 + e) << 23;
    return c.f;
}

#if defined(__HIPCC__)
// the kSplitNS 16-bit patterns of v (already multiplied by its power of two).  CLAMP: v may exceed the fp16 range
template <bool CLAMP = false>
__device__ __forceinline__ void split_slices(float v, unsigned (&s)[kSplitNS]) {
#ifdef FFHIP_SPLIT_BF16X3
    const unsigned b0 = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v);
    const float r1 = v - __uint_as_float(b0 << 16);
    const unsigned b1 = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)r1);
    const float r2 = r1 - __uint_as_float(b1 << 16);
    s[0] = b0; s[1] = b1; s[2] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)r2);
#else
    if (CLAMP) {                                               // compare-selects, not min/max: a NaN must stay a NaN
        v = (v > 65504.0f) ? 65504.0f : v;
        v = (v < -65504.0f) ? -65504.0f : v;
    }
    const _Float16 h0 = (_Float16)v;                           // round to nearest even
    const float r1 = v - (float)h0;                            // exact
    const _Float16 h1 = (_Float16)r1;
    s[0] = (unsigned)__builtin_bit_cast(unsigned short, h0);
    s[1] = (unsigned)__builtin_bit_cast(unsigned short, h1);
#endif
}

// does split_slices<true> clamp one of these four (already scaled) values?  A NaN is not clamped (it stays a NaN, as in the reference)
template <typename V4>
__device__ __forceinline__ bool split_overflow(V4 v) {
#ifdef FFHIP_SPLIT_BF16X3
    return false;
#else
    return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) > 65504.0f;
#endif
}

__device__ __forceinline__ float split_slice_value(unsigned bits16) {
#ifdef FFHIP_SPLIT_BF16X3
    return __uint_as_float(bits16 << 16);
#else
    return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16);
#endif
}
#endif

// ---- host side: the same slices for the weight packer ------------------------------------------------------------
inline uint16_t split_host_f16_rne(float f) {                  // fp32 -> fp16 bits, round to nearest even, subnormals kept
    union { float f; uint32_t u; } c;
    c.f = f;
    const uint32_t sign = (c.u >> 16) & 0x8000u;
    const uint32_t a = c.u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | (a > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                  // rounds to >= 65520: infinity
    if (a < 0x33000001u) return (uint16_t)sign;                                                // <= 2^-25: zero
    const int e = (int)(a >> 23) - 127;
    uint32_t man = (a & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                                             // bits dropped from the 24-bit significand
    const uint32_t halfway = 1u << (shift - 1), mask = (1u << shift) - 1u;
    uint32_t q = man >> shift;
    const uint32_t rem = man & mask;
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    uint32_t out;
    if (e < -14) out = q;                                                                      // subnormal (q may carry into the normal range: still right)
    else out = ((uint32_t)(e + 15) << 10) + (q - 0x400u);                                      // q in [0x400, 0x800]; a carry bumps the exponent
    return (uint16_t)(sign | out);
}
inline float split_host_f16_value(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 0x1f;
    const uint32_t m = h & 0x3ffu;
    union { float f; uint32_t u; } c;
    if (e == 0) { c.f = (float)m * 5.9604644775390625e-8f; c.u |= sign; return c.f; }         // m * 2^-24
    if (e == 31) { c.u = sign | 0x7f800000u | (m << 13); return c.f; }
    c.u = sign | ((uint32_t)(e - 15 + 127) << 23) | (m << 13);
    return c.f;
}
inline uint16_t split_host_bf16_rne(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    c.u += 0x7fffu + ((c.u >> 16) & 1u);
    return (uint16_t)(c.u >> 16);
}
// slices of w * 2^exp2
inline void split_host_slices(float w, int exp2, uint16_t (&s)[kSplitNS]) {
#ifdef FFHIP_SPLIT_BF16X3
    (void)exp2;
    auto val = [](uint16_t h) { union { float f; uint32_t u; } c; c.u = (uint32_t)h << 16; return c.f; };
    s[0] = split_host_bf16_rne(w);
    const float r1 = w - val(s[0]);
    s[1] = split_host_bf16_rne(r1);
    s[2] = split_host_bf16_rne(r1 - val(s[1]));
#else
    const float v = w * split_pow2(exp2);
    s[0] = split_host_f16_rne(v);
    s[1] = split_host_f16_rne(v - split_host_f16_value(s[0]));
#endif
}
// largest exponent e with max|w| * 2^e < 2^15 (fp16 tops out at 65504 = 2^16 - 32); 0 for the bf16 build and for an all-zero matrix
inline int split_weight_exp(float maxabs) {
#ifdef FFHIP_SPLIT_BF16X3
    (void)maxabs;
    return 0;
#else
    if (!(maxabs > 0.0f) || !(maxabs < 3.0e38f)) return 0;
    union { float f; uint32_t u; } c;
    c.f = maxabs;
    const int e = (int)((c.u >> 23) & 0xff) - 127;             // maxabs in [2^e, 2^(e+1))
    int sw = 14 - e;                                           // -> [2^14, 2^15)
    if (sw > 40) sw = 40;
    if (sw < -40) sw = -40;
    return sw;
#endif
}

}  // namespace ffhip
