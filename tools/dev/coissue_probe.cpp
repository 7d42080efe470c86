// Probe 3 (round 5): how much of the layer kernel's two kinds of work does one SIMD of gfx950 carry AT THE SAME TIME?
// The layer kernels' SIMDs are "busy" two thirds of a step if the matrix pipe's and the VALU's cycles are simply added (216 MFMAs = 3456 cycles + ~830 VALU
// instructions = ~3300 cycles of a 10 250-cycle step, profiles/r04_c2_sq_pmc.csv), and three schedules that tried to put the projection's MFMAs under something
// else did not shorten the step (profiles/r05_lds_flags.txt).  This measures the two instruction mixes of that kernel against each other on ONE SIMD, with
// nothing else in the way: waves of role M issue the kernel's MFMA pattern (`v_mfma_f32_16x16x32_f16`, three independent accumulators, 18 distinct weight quads
// and 2 operand quads -- the register footprint of an h wave), waves of role G evaluate the kernel's gate arithmetic for one tile per iteration (four partial sums
// from LDS, ffhip_math.hpp's logistic_ref4_lean + tanh_ref_lean, the split of h and the LDS transpose of publish_h: the real functions).  One workgroup on one CU,
// W waves per SIMD; per case the time of each role alone and together.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iflappie_amd/csrc -Iinclude tools/dev/coissue_probe.cpp -o tools/variants/coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "ffhip_math.hpp"
#include "ffhip_split.hpp"
using namespace ffhip;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int NOP>
__device__ __forceinline__ v4f mm(v4u a, v4u b, v4f c) {
    v4f d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
    if constexpr (NOP >= 0) { asm volatile("s_nop %1" : "+v"(d) : "n"(NOP)); }      // (tied to the result's name only: the wave idles NOP + 1 cycles behind the issue)
    return d;
}
template <int NOP>
__device__ __forceinline__ float mfma_stream(int itersM, int lane) {
    v4u w[18], x[2];
    for (int k = 0; k < 18; k++) w[k] = (v4u){ 0x3c003c00u + k, 0x3c003800u + lane, 0x38003c00u, 0x3c003c00u };
    for (int k = 0; k < 2; k++) x[k] = (v4u){ 0x34003400u + lane, 0x34003000u, 0x30003400u + k, 0x34003400u };
    v4f a0 = { 0, 0, 0, 0 }, a1 = a0, a2 = a0;
    for (int i = 0; i < itersM; i++) {
        // one K chunk of one tile as mm6 issues it: 3 terms x 3 row tiles, consecutive MFMAs on different accumulators
#pragma unroll
        for (int c = 0; c < 3; c++) {
            a0 = mm<NOP>(w[6 * c + 1], x[0], a0); a1 = mm<NOP>(w[6 * c + 3], x[0], a1); a2 = mm<NOP>(w[6 * c + 5], x[0], a2);
            a0 = mm<NOP>(w[6 * c + 0], x[1], a0); a1 = mm<NOP>(w[6 * c + 2], x[1], a1); a2 = mm<NOP>(w[6 * c + 4], x[1], a2);
            a0 = mm<NOP>(w[6 * c + 0], x[0], a0); a1 = mm<NOP>(w[6 * c + 2], x[0], a1); a2 = mm<NOP>(w[6 * c + 4], x[0], a2);
        }
        asm volatile("" : "+v"(x[0]), "+v"(x[1]));
    }
    return a0.x + a1.y + a2.z;
}

// role of wave w (0..15, SIMD = w & 3, slot = w >> 2): roles[slot] = 0 idle, 1 MFMA stream, 2 gate arithmetic; prio[slot] = its s_setprio
__global__ void __launch_bounds__(1024, 1) probe(float *out, unsigned long long *cyc, const int *roles, const int *prio, int itersM, int itersG) {
    __shared__ v4f part[16][4][64];
    __shared__ unsigned short gsl[16][2][16][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, slot = wave >> 2;
    const int role = roles[slot];
    for (int k = 0; k < 4; k++) part[wave][k][lane] = (v4f){ 0.01f * lane, -0.02f * lane, 0.03f * (lane & 7), 0.5f - 0.01f * lane };
    __syncthreads();
    if (prio[slot]) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    float sink = 0.f;
    if (role == 1) sink = mfma_stream<-1>(itersM, lane);
    else if (role == 13) sink = mfma_stream<3>(itersM, lane);
    else if (role == 17) sink = mfma_stream<7>(itersM, lane);
    else if (role == 19) sink = mfma_stream<9>(itersM, lane);
    else if (role == 21) sink = mfma_stream<11>(itersM, lane);
    else if (role == 23) sink = mfma_stream<13>(itersM, lane);
    else if (role == 2) {
        float c = 0.1f;
        const int q = lane >> 4, rl = lane & 15;
        for (int i = 0; i < itersG; i++) {
            v4f s = { 0.01f, 0.02f, 0.03f, 0.04f };
#pragma unroll
            for (int k = 0; k < 4; k++) s = s + part[wave][k][lane];
            s = (v4f){ __builtin_ldexpf(s.x, -3), __builtin_ldexpf(s.y, -3), __builtin_ldexpf(s.z, -3), __builtin_ldexpf(s.w, -3) };
            const ffv4 L = logistic_ref4_lean((ffv4){ s.x, s.y, s.z + s.z, s.w });
            const float tanh_g = (L.z + L.z) - 1.0f;
            c = L.y * c + L.x * tanh_g;
            const float h = L.w * tanh_ref_lean(c);
            unsigned sl[2];
            split_slices(h * split_pow2(kSplitExpH), sl);
            gsl[wave][0][rl][q] = (unsigned short)sl[0];
            gsl[wave][1][rl][q] = (unsigned short)sl[1];
            asm volatile("" ::: "memory");
            const unsigned long long v = *(const unsigned long long *)&gsl[wave][q & 1][rl][0];
            part[wave][i & 3][lane].x = __uint_as_float((unsigned)v) * 1e-30f + h;      // (keeps the chain: next iteration's sums depend on this one)
        }
        sink = c;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[wave] = t1 - t0;
    out[threadIdx.x] = sink;
}

int main() {
    float *out; unsigned long long *cyc, h[16]; int *roles, *prio;
    hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 16 * 8); hipMalloc(&roles, 16); hipMalloc(&prio, 16);
    struct { int r[4], p[4]; const char *name; } cases[] = {
        { { 1, 0, 0, 0 }, { 1, 0, 0, 0 }, "M alone (1 wave a SIMD)" },
        { { 1, 1, 0, 0 }, { 1, 1, 0, 0 }, "M + M" },
        { { 2, 0, 0, 0 }, { 1, 0, 0, 0 }, "G alone (1 wave a SIMD)" },
        { { 2, 2, 0, 0 }, { 1, 1, 0, 0 }, "G + G" },
        { { 1, 2, 0, 0 }, { 1, 1, 0, 0 }, "M + G, same priority" },
        { { 1, 2, 0, 0 }, { 0, 1, 0, 0 }, "M (priority 0) + G (priority 3)" },
        { { 1, 2, 0, 0 }, { 1, 0, 0, 0 }, "M (priority 3) + G (priority 0)" },
        { { 1, 1, 2, 2 }, { 1, 1, 1, 1 }, "M + M + G + G (the kernel's four waves a SIMD)" },
        { { 1, 1, 2, 2 }, { 0, 1, 1, 1 }, "M (0) + M (3) + G (3) + G (3)" },
        { { 1, 1, 0, 0 }, { 0, 1, 0, 0 }, "M (older, priority 0) + M (younger, priority 3)" },
        { { 2, 1, 0, 0 }, { 1, 1, 0, 0 }, "G (older) + M (younger)" },
        { { 13, 0, 0, 0 }, { 1, 0, 0, 0 }, "Mn3 alone: s_nop 3 behind every MFMA" },
        { { 17, 0, 0, 0 }, { 1, 0, 0, 0 }, "Mn7 alone" },
        { { 19, 0, 0, 0 }, { 1, 0, 0, 0 }, "Mn9 alone" },
        { { 21, 0, 0, 0 }, { 1, 0, 0, 0 }, "Mn11 alone" },
        { { 23, 0, 0, 0 }, { 1, 0, 0, 0 }, "Mn13 alone" },
        { { 13, 2, 0, 0 }, { 1, 1, 0, 0 }, "Mn3 + G" },
        { { 17, 2, 0, 0 }, { 1, 1, 0, 0 }, "Mn7 + G" },
        { { 19, 2, 0, 0 }, { 1, 1, 0, 0 }, "Mn9 + G" },
        { { 21, 2, 0, 0 }, { 1, 1, 0, 0 }, "Mn11 + G" },
        { { 23, 2, 0, 0 }, { 1, 1, 0, 0 }, "Mn13 + G" },
        { { 21, 2, 0, 0 }, { 0, 1, 0, 0 }, "Mn11 (priority 0) + G (priority 3)" },
        { { 21, 2, 2, 0 }, { 1, 1, 1, 0 }, "Mn11 + G + G" },
        { { 17, 17, 2, 2 }, { 1, 1, 1, 1 }, "Mn7 + Mn7 + G + G" },
        { { 21, 21, 2, 2 }, { 1, 1, 1, 1 }, "Mn11 + Mn11 + G + G" },
    };
    printf("one workgroup on one CU; per case: cycles per 27-MFMA chunk group (M) and per gate tile (G), by wave slot; ideal M = 27 x 16 = 432\n");
    for (auto &c : cases) {
        hipMemcpy(roles, c.r, 16, hipMemcpyHostToDevice); hipMemcpy(prio, c.p, 16, hipMemcpyHostToDevice);
        // M iterates three times as often as G (a chunk group is ~1/3 of a gate tile's time): both roles run about as long, and the one that ends first
        // has run beside the other for all of its time
        const int itersG = 1000, itersM = 3 * itersG;
        hipLaunchKernelGGL(probe, dim3(1), dim3(1024), 0, 0, out, cyc, roles, prio, 10, 10);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe, dim3(1), dim3(1024), 0, 0, out, cyc, roles, prio, itersM, itersG);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
        printf("%-52s", c.name);
        for (int s = 0; s < 4; s++) if (c.r[s]) printf("  %s %7.0f (total %8llu)", c.r[s] != 2 ? "M" : "G", (double)h[4 * s] / (c.r[s] != 2 ? itersM : itersG), h[4 * s]);
        printf("\n");
    }
    return 0;
}
