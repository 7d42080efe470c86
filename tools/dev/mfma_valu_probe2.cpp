// Probe 2 (round 2): a dependent VALU chain (the shape of the gate math: mul / add / fma, some transcendental-free integer ops)
// beside v_mfma_f32_16x16x32_f16 streams of different duty on the SAME SIMD.  One workgroup of 12 waves on one CU: waves 0-3 and 4-7
// (two per SIMD) stream MFMAs with `gap` s_nop 15 between groups of 3; waves 8-11 (one per SIMD) run the VALU chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(768, 1) probe(float *out, unsigned long long *cyc, int mfma_waves, int do_valu, int nm, int nv, int gap, int prio_valu) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 8) {
        if (wave < mfma_waves) {
            v4f a0 = {0,0,0,0}, a1 = {0,0,0,0}, a2 = {0,0,0,0};
            v8h x, y;
            for (int k = 0; k < 8; k++) { x[k] = (_Float16)(lane * 0.001f + k); y[k] = (_Float16)(1.0f + lane * 0.002f); }
            for (int i = 0; i < nm; i++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a2, 0, 0, 0);
                for (int g = 0; g < gap; g++) asm volatile("s_nop 15");
            }
            out[threadIdx.x] = a0.x + a1.y + a2.z;
        }
    } else if (do_valu) {
        if (prio_valu) __builtin_amdgcn_s_setprio(3);
        float v = 1.0f + lane * 1e-3f, w = 0.5f;
        for (int i = 0; i < nv; i++) {          // 6 dependent VALU ops per iteration, like a polynomial step
            v = v * 1.0001f + 0.5f;
            w = __builtin_fmaf(v, w, -0.25f);
            v = v - w * 0.125f;
        }
        out[threadIdx.x] = v + w;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[wave] = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc, h[12];
    hipMalloc(&out, 768 * 4); hipMalloc(&cyc, 128);
    const int nm = 1500, nv = 1500;
    struct { int mw, v, gap, p; const char *name; } cases[] = {
        {0,1,0,0,"valu chain alone"}, {4,0,0,0,"1 mfma wave per SIMD, dense"}, {8,0,0,0,"2 mfma waves per SIMD, dense"},
        {4,1,0,0,"valu + 1 dense mfma wave"}, {4,1,0,1,"valu (prio 3) + 1 dense mfma wave"}, {8,1,0,1,"valu (prio 3) + 2 dense mfma waves"},
        {4,1,1,1,"valu (prio 3) + 1 mfma wave, gap 1 (duty ~75%)"}, {4,1,3,1,"valu (prio 3) + 1 mfma wave, gap 3 (duty ~50%)"},
        {8,1,3,1,"valu (prio 3) + 2 mfma waves, gap 3 each"}, {8,1,9,1,"valu (prio 3) + 2 mfma waves, gap 9 each (~25% each)"},
        {4,1,9,1,"valu (prio 3) + 1 mfma wave, gap 9 (~25%)"} };
    for (auto &c : cases) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(cyc, 0, 128);
            hipLaunchKernelGGL(probe, dim3(1), dim3(768), 0, 0, out, cyc, c.mw, c.v, nm, nv, c.gap, c.p);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, 96, hipMemcpyDeviceToHost);
        printf("%-58s mfma wave0 %7llu wave4 %7llu | valu wave8 %7llu wave9 %7llu\n", c.name, h[0], h[4], h[8], h[9]);
    }
    return 0;
}
