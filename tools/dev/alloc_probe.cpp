// What ~100 GB of device memory cost a process (round 6, third session): hipMalloc, the first touch, a second object, free + malloc again,
// and the same from a second thread while a kernel runs.   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/dev/alloc_probe.cpp -o tools/bin/alloc_probe -lpthread && tools/bin/alloc_probe [GB=100]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <pthread.h>
static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
__global__ void spin(unsigned long long cycles, int *out) { const unsigned long long t0 = clock64(); while (clock64() - t0 < cycles) { } if (out) out[0] = 1; }
__global__ void touch(char *p, size_t n, size_t stride) { const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }
static size_t BYTES;
static void *alloc_thread(void *arg) {
    hipSetDevice(0);
    double t0 = now(); void *p = nullptr;
    hipError_t e = hipMalloc(&p, BYTES);
    double t1 = now();
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipMemsetAsync(p, 0, BYTES, s); hipStreamSynchronize(s);
    double t2 = now();
    printf("  second thread: hipMalloc %.3f s (%s), memset %.3f s\n", t1 - t0, hipGetErrorString(e), t2 - t1);
    *(void **)arg = p;
    return nullptr;
}
int main(int argc, char **argv) {
    BYTES = (size_t)(argc > 1 ? atof(argv[1]) : 100.0) * (1ull << 30);
    hipSetDevice(0);
    hipFree(0);
    double t0 = now(); char *a = nullptr; hipError_t e = hipMalloc((void **)&a, BYTES); double t1 = now();
    printf("hipMalloc of %.0f GB: %.3f s (%s)\n", BYTES / 1073741824.0, t1 - t0, hipGetErrorString(e));
    t0 = now(); hipLaunchKernelGGL(touch, dim3((unsigned)((BYTES / 4096 + 255) / 256)), dim3(256), 0, 0, a, BYTES, (size_t)4096); hipDeviceSynchronize(); t1 = now();
    printf("first touch (one byte a 4 KiB page, a kernel): %.3f s\n", t1 - t0);
    t0 = now(); hipLaunchKernelGGL(touch, dim3((unsigned)((BYTES / 4096 + 255) / 256)), dim3(256), 0, 0, a, BYTES, (size_t)4096); hipDeviceSynchronize(); t1 = now();
    printf("second touch: %.3f s\n", t1 - t0);
    t0 = now(); hipMemset(a, 0, BYTES); hipDeviceSynchronize(); t1 = now();
    printf("hipMemset of all of it: %.3f s\n", t1 - t0);
    t0 = now(); char *b = nullptr; e = hipMalloc((void **)&b, BYTES); t1 = now();
    printf("second object, hipMalloc: %.3f s (%s)\n", t1 - t0, hipGetErrorString(e));
    t0 = now(); hipMemset(b, 0, BYTES); hipDeviceSynchronize(); t1 = now();
    printf("second object, hipMemset: %.3f s\n", t1 - t0);
    t0 = now(); hipFree(b); t1 = now(); printf("hipFree of the second: %.3f s\n", t1 - t0);
    t0 = now(); e = hipMalloc((void **)&b, BYTES); t1 = now(); printf("hipMalloc again (memory this process just gave back): %.3f s\n", t1 - t0);
    t0 = now(); hipMemset(b, 0, BYTES); hipDeviceSynchronize(); t1 = now(); printf("  its hipMemset: %.3f s\n", t1 - t0);
    hipFree(b);
    // a kernel that keeps the device busy for ~3 s while another thread allocates and clears an object: does the allocation wait for the kernel, does the kernel finish on time?
    int *flag; hipMalloc((void **)&flag, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    t0 = now();
    hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, s, 3ull * 100000000ull, flag);      // clock64 counts at 100 MHz: ~3 s
    pthread_t th; void *p2 = nullptr; pthread_create(&th, nullptr, alloc_thread, &p2);
    hipStreamSynchronize(s); t1 = now();
    printf("  the 3 s kernel beside it finished after %.3f s\n", t1 - t0);
    pthread_join(th, nullptr);
    return 0;
}
