// Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs on gfx950, and does the f32 -> f16 conversion produce them?
// (The two-slice fp16 operand split of ffhip_rnn_split.hip relies on neither, but its error floor for small values is lower when both hold.)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/f16_denorm_probe tools/dev/f16_denorm_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
__global__ void k(float a_val, float b_val, float *out, unsigned *bits) {
    v8h a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    v4f acc = { 0.f, 0.f, 0.f, 0.f };
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc.x; bits[0] = __builtin_bit_cast(unsigned short, (_Float16)a_val); }
}
int main() {
    float *d; unsigned *db; hipMalloc(&d, 4); hipMalloc(&db, 4);
    const float vals[] = { 1.0f, 6.1035156e-5f /* 2^-14, smallest normal */, 9.5367432e-7f /* 2^-20, subnormal */, 5.9604645e-8f /* 2^-24, smallest subnormal */ };
    for (float v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, v, 1.0f, d, db);
        float h; unsigned hb; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, db, 4, hipMemcpyDeviceToHost);
        printf("a = %.8e (f16 bits 0x%04x) x b = 1, K = 32: mfma = %.8e, expected %.8e -> %s\n", v, hb, h, 32.0f * v, h == 32.0f * v ? "kept" : "FLUSHED/other");
    }
    return 0;
}
