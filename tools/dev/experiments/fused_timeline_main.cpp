// dev harness: fused LSTM layer with per-wave phase timestamps for every workgroup; prints the
// timeline of the workgroup pairs that share a CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#include <algorithm>
#include "ffhip_internal.hpp"
namespace ffhip { extern unsigned long long *g_persist_dbg; }
using namespace ffhip;
int main(int argc, char **argv) {
    const int H = 384, B16 = argc > 1 ? atoi(argv[1]) : 16, Tb = 400, Bp = 16 * B16;
    float4 *sWp, *iWp; float *bias, *xin, *hout; unsigned *flags, *ab; unsigned long long *dbg;
    hipMalloc(&sWp, (size_t)4*H*H*4); hipMemset(sWp, 0, (size_t)4*H*H*4);
    hipMalloc(&iWp, (size_t)4*H*H*4); hipMemset(iWp, 0, (size_t)4*H*H*4);
    hipMalloc(&bias, 4*H*4); hipMemset(bias, 0, 4*H*4);
    hipMalloc(&xin, (size_t)Tb*Bp*H*4); hipMemset(xin, 0, (size_t)Tb*Bp*H*4);
    hipMalloc(&hout, (size_t)Tb*Bp*H*4);
    hipMalloc(&flags, 4096*4); hipMalloc(&ab, 4); hipMemset(ab, 0, 4);
    const size_t ndbg = 64*512 + (size_t)512*4*32*6;
    hipMalloc(&dbg, ndbg*8); hipMemset(dbg, 0, ndbg*8);
    g_persist_dbg = dbg;
    for (int rep = 0; rep < 2; rep++) {
        hipMemsetD32(hout, 0xFFFFFFFF, (size_t)Tb*Bp*H);
        hipMemset(flags, 0, 4096*4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        launch_lstm_fused(0, 0, sWp, iWp, bias, xin, hout, flags, ab, Tb, B16, H, 0, B16, 1, argc > 2 ? 100 + atoi(argv[2]) : 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("layer %.3f ms = %.3f us/step\n", ms, ms*1e3/Tb);
    }
    std::vector<unsigned long long> h(ndbg);
    hipMemcpy(h.data(), dbg, ndbg*8, hipMemcpyDeviceToHost);
    const int nwg = B16 * 32;
    std::map<unsigned long long, std::vector<int>> bycu;
    for (int b = 0; b < nwg; b++) {
        const unsigned hw = (unsigned)h[b], xc = (unsigned)(h[b] >> 32) & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        bycu[((unsigned long long)xc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    std::map<int,int> hist;
    for (auto &kv : bycu) hist[(int)kv.second.size()]++;
    for (auto &kv : hist) printf("CUs hosting %d workgroups: %d\n", kv.first, kv.second);
    // print the timeline of the first CU that hosts exactly 2 workgroups
    for (auto &kv : bycu) {
        if (kv.second.size() != 2) continue;
        const int b0 = kv.second[0], b1 = kv.second[1];
        printf("CU key %llx hosts blocks %d and %d\n", kv.first, b0, b1);
        auto T = [&](int b, int w, int st, int k) { return h[64*512 + (((size_t)b*4 + w)*32 + st)*6 + k]; };
        const unsigned long long base = T(b0, 0, 0, 0);
        for (int st = 0; st < 4; st++)
            for (int bi = 0; bi < 2; bi++) {
                const int b = kv.second[bi];
                for (int w = 0; w < 4; w++) {
                    printf("step %d blk %4d wave %d:", 100 + st, b, w);
                    for (int k = 0; k < 6; k++) { unsigned long long v = T(b, w, st, k); printf(" %7lld", v ? (long long)(v - base) : -1LL); }
                    printf("   | x %5lld wait %5lld h %5lld bar %5lld gate %5lld\n", (long long)(T(b,w,st,1)-T(b,w,st,0)), (long long)(T(b,w,st,2)-T(b,w,st,1)),
                           (long long)(T(b,w,st,3)-T(b,w,st,2)), (long long)(T(b,w,st,4)-T(b,w,st,3)), T(b,w,st,5) ? (long long)(T(b,w,st,5)-T(b,w,st,4)) : 0LL);
                }
            }
        break;
    }
    return 0;
}
