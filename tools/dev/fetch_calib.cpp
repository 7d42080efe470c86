// What does rocprofv3's FETCH_SIZE count on gfx950 for the access patterns of the layer kernel?  
// Streams a buffer of known size once per launch with
//   k_global16   global_load_dwordx4, 16 B per lane, coalesced                              (the guide's calibration pattern)
//   k_buf16_sc1  buffer_load_dwordx4 ... sc1, 16 B per lane                                  (the x waves' loads of the layer input)
//   k_lds16_sc1  buffer_load_dwordx4 ... lds sc1, 16 B per lane straight into LDS            (the h waves' sweep)
//   k_touch      one dword per 128-byte line                                                 (the x waves' L2 warming)
// Run under   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- fetch_calib   and   --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
// and divide by the bytes printed here.   build: hipcc --offload-arch=gfx950 -O3 tools/dev/fetch_calib.cpp -o tools/variants/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_global16(const v4u *__restrict__ p, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const v4u v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_buf16_sc1(const v4u *__restrict__ p, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    const size_t per = (size_t)gridDim.x * 256;
    // a buffer resource per 1 GiB window keeps the 32-bit offsets in range
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += per) {
        const size_t base = (i * 16) & ~(((size_t)1 << 30) - 1);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p + base), 0, (int)0x40000000, 0x00020000);
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(i * 16 - base), 0, 16 /*sc1*/);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_lds16_sc1(const v4u *__restrict__ p, size_t n16, unsigned *sink) {
    __shared__ v4u land[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    const size_t per = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + wave * 64; i0 < n16; i0 += per) {      // a wave's 64 lanes land 1 KiB
        const size_t base = (i0 * 16) & ~(((size_t)1 << 30) - 1);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p + base), 0, (int)0x40000000, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)&land[wave][0], 16, lane * 16, (int)(i0 * 16 - base), 0, 16 /*sc1*/);
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const v4u v = land[wave][lane];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_touch(const unsigned *__restrict__ p, size_t nline, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nline; i += (size_t)gridDim.x * 256) acc ^= p[i * 32];
    if (acc == 0x12345678u) *sink = acc;
}

// writes: 16 B per lane coalesced (the gate waves' stores of h are 8 B per lane into 16-byte slots; the fp32 copy 16 B)
__global__ void __launch_bounds__(256) k_store16(v4u *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = (v4u){ (unsigned)i, 1u, 2u, 3u };
}
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_store8(v2u *__restrict__ p, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) p[i] = (v2u){ (unsigned)i, 1u };
}

int main(int argc, char **argv) {
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)314572800);      // 256 reads x 800 blocks x 384 x 4 B: one layer's input
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    // a fresh region per launch (4 of them, round robin, > the 256-MiB Infinity Cache in between)
    char *buf; unsigned *sink;
    CK(hipMalloc(&buf, bytes * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes * 4)); CK(hipDeviceSynchronize());
    printf("bytes per launch: %zu (%zu lines of 128 B, %zu sectors of 64 B)\n", bytes, bytes / 128, bytes / 64);
    int k = 0;
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(k_global16, dim3(2048), dim3(256), 0, 0, (const v4u *)(buf + (k++ % 4) * bytes), bytes / 16, sink);
        hipLaunchKernelGGL(k_buf16_sc1, dim3(2048), dim3(256), 0, 0, (const v4u *)(buf + (k++ % 4) * bytes), bytes / 16, sink);
        hipLaunchKernelGGL(k_lds16_sc1, dim3(2048), dim3(256), 0, 0, (const v4u *)(buf + (k++ % 4) * bytes), bytes / 16, sink);
        hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, 0, (const unsigned *)(buf + (k++ % 4) * bytes), bytes / 128, sink);
        hipLaunchKernelGGL(k_store16, dim3(2048), dim3(256), 0, 0, (v4u *)(buf + (k++ % 4) * bytes), bytes / 16);
        hipLaunchKernelGGL(k_store8, dim3(2048), dim3(256), 0, 0, (v2u *)(buf + (k++ % 4) * bytes), bytes / 8);
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
