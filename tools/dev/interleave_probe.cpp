// Probe: can ONE wave hide its gate math (VALU, cephes-exact logistic/tanh) under its own independent f32 MFMAs
// on gfx950?  Per iteration: 72 MFMAs (3 accumulator chains) + one LSTM gate evaluation per lane.
//   mode 0: MFMAs only      mode 1: gate math only      mode 2: MFMAs then gates (program order)
//   mode 3: one basic block, sched_group_barrier pattern {1 MFMA, NV VALU} repeated
// `other` = 1 adds a second wave per SIMD that streams MFMAs (the co-resident workgroup).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../flappie_amd/csrc/ffhip_math.hpp"
using namespace ffhip;
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ void __launch_bounds__(512, 1) probe(float *out, unsigned long long *cyc, int iters, int other) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4 && !other) return;
    float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
    v4f a0 = {0,0,0,0}, a1 = {0,0,0,0}, a2 = {0,0,0,0};
    float c = 0.1f * lane, h = 0.0f;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave >= 4) {
        for (int i = 0; i < iters; i++)
#pragma unroll
            for (int k = 0; k < 24; k++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            }
    } else {
        for (int i = 0; i < iters; i++) {
            v4f s = { h + 0.3f, c * 0.01f, h - 0.2f, 0.5f + h };
            if (MODE != 1) {
#pragma unroll
                for (int k = 0; k < 24; k++) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                }
            }
            if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
            if (MODE != 0) {
                const ffv4 L = logistic_ref4((ffv4){ s.x, s.y, s.z + s.z, s.w });
                const float tanh_g = (L.z + L.z) - 1.0f;
                c = L.y * c + L.x * tanh_g;
                h = L.w * tanh_ref(c);
            }
            if (MODE == 3) {
#pragma unroll
                for (int k = 0; k < 72; k++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0.x + a1.y + a2.z + h + c;
    if (lane == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int NV>
static void run(const char *name, int other) {
    float *out; unsigned long long *cyc, hc[8];
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 400;
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(cyc, 0, 64);
        hipLaunchKernelGGL((probe<MODE, NV>), dim3(1), dim3(512), 0, 0, out, cyc, iters, other);
        hipDeviceSynchronize();
    }
    hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-46s other=%d  cycles/iter: gate waves %6.0f %6.0f %6.0f %6.0f   stream waves %6.0f %6.0f\n", name, other,
           hc[0] / (double)iters, hc[1] / (double)iters, hc[2] / (double)iters, hc[3] / (double)iters, hc[4] / (double)iters, hc[5] / (double)iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int other = 0; other < 2; other++) {
        run<0, 1>("72 MFMA only", other);
        run<1, 1>("gate math only", other);
        run<2, 1>("72 MFMA then gate math", other);
        run<3, 2>("interleaved {1 MFMA, 2 VALU}", other);
        run<3, 3>("interleaved {1 MFMA, 3 VALU}", other);
        run<3, 4>("interleaved {1 MFMA, 4 VALU}", other);
        run<3, 6>("interleaved {1 MFMA, 6 VALU}", other);
    }
    return 0;
}
