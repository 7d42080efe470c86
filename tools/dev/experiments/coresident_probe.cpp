// dev probe: which workgroups of a 512-workgroup launch (512 threads, 77 KB of LDS, two per CU) share a CU on MI355X?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ void __launch_bounds__(512, 4) probe(unsigned *out, int spin) {
    extern __shared__ unsigned char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; lds[0] = 1; }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100);      // stay resident until every workgroup has been placed
}
int main() {
    const int nwg = 512;
    unsigned *d; hipMalloc(&d, nwg * 8);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 78576);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), 78576, 0, d, 200);
        hipDeviceSynchronize();
        std::vector<unsigned> h(nwg * 2);
        hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> cu;
        for (int b = 0; b < nwg; b++) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;      // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
            cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
        }
        printf("rep %d: %zu distinct (xcc, se, sh, cu) slots\n", rep, cu.size());
        int shown = 0;
        std::map<int, int> delta;
        for (auto &kv : cu) {
            if (shown++ < 12) { printf("  slot %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
            if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
        }
        for (auto &kv : delta) printf("  pairs with block distance %d: %d\n", kv.first, kv.second);
    }
    return 0;
}
