// Probe: how do an f32-MFMA stream and a dependent VALU chain on the SAME SIMD interact on gfx950?
// One workgroup of 8 waves on one CU: waves 0-3 (one per SIMD) stream MFMAs, waves 4-7 run a dependent
// VALU chain (or IEEE divisions).  Report cycles for each role alone and together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512, 2) probe(float *out, unsigned long long *cyc, int do_mfma, int do_valu, int nm, int nv, int prio_valu) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ int go;
    if (threadIdx.x == 0) go = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (do_mfma) {
            v4f a0 = {0,0,0,0}, a1 = {0,0,0,0}, a2 = {0,0,0,0};
            float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
            for (int i = 0; i < nm; i++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            }
            out[threadIdx.x] = a0.x + a1.y + a2.z;
        }
    } else {
        if (do_valu) {
            if (prio_valu) __builtin_amdgcn_s_setprio(3);
            float v = 1.0f + lane * 1e-3f;
            if (do_valu == 1) { for (int i = 0; i < nv; i++) v = v * 1.0001f + 0.5f; }            // dependent mul+add (2 VALU)
            else if (do_valu == 2) { for (int i = 0; i < nv; i++) v = 1.0f / (1.0f + v); }            // IEEE division chain
            else { for (int i = 0; i < nv; i++) v = __builtin_amdgcn_rcpf(1.0f + v); }                // rcp chain
            out[threadIdx.x] = v;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[wave] = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc, h[8];
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 64);
    const int nm = 2000, nv = 2000;
    struct { int m, v, p; const char *name; } cases[] = {
        {1,0,0,"mfma only (6000 MFMA per wave)"}, {0,1,0,"valu mul+add chain only (4000 VALU)"}, {1,1,0,"both, equal prio"}, {1,1,1,"both, valu prio 3"},
        {0,2,0,"IEEE div chain only (2000 div)"}, {1,2,1,"mfma + IEEE div, valu prio 3"}, {0,3,0,"rcp chain only"}, {1,3,1,"mfma + rcp chain, valu prio 3"} };
    for (auto &c : cases) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(cyc, 0, 64);
            hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, out, cyc, c.m, c.v, nm, nv, c.p);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("%-40s mfma waves: %llu %llu %llu %llu   valu waves: %llu %llu %llu %llu\n", c.name, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    return 0;
}
