// dev harness: split-bf16 LSTM layer (ffhip_rnn_split.hip built with -DFFHIP_TIMELINE) with per-wave phase timestamps.
// stamps per step: 0 loop top, 1 (h waves) sweep complete, 2 before barrier 1 (MFMAs done, partials in LDS), 3 after
// barrier 1, 4 gate phase done (stores issued), 5 after barrier 2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "ffhip_internal.hpp"
namespace ffhip { extern unsigned long long *g_split_dbg; }
using namespace ffhip;
int main(int argc, char **argv) {
    const int H = argc > 2 ? atoi(argv[2]) : 384, B16 = argc > 1 ? atoi(argv[1]) : 16, Tb = 400;
    void *Wp, *xin, *hout; float *bias; unsigned *flags, *ab; unsigned long long *dbg;
    const size_t wbytes = (size_t)2 * 4 * H * H * 8, abytes = split_bytes((size_t)Tb * B16, H);
    hipMalloc(&Wp, wbytes); hipMemset(Wp, 0, wbytes);
    hipMalloc(&bias, 4 * H * 4); hipMemset(bias, 0, 4 * H * 4);
    hipMalloc(&xin, abytes); hipMemset(xin, 0, abytes);
    hipMalloc(&hout, abytes);
    hipMalloc(&flags, 4096 * 4); hipMalloc(&ab, 4); hipMemset(ab, 0, 4);
    const int nwg = ((B16 + 1) / 2) * 32;
    const size_t ndbg = (size_t)nwg * 8 * 32 * 16;
    hipMalloc(&dbg, ndbg * 8); hipMemset(dbg, 0, ndbg * 8);
    g_split_dbg = dbg;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const int nrep = argc > 4 ? atoi(argv[4]) : 40;      // enough launches for the clocks to settle: the first ones run ~40 % slower
    const int kind = argc > 5 ? atoi(argv[5]) : 0;          // 1: GRUmod (H = 256 and B16 = 64: the packed form)
    for (int rep = 0; rep < nrep; rep++) {
        hipMemsetD32((hipDeviceptr_t)hout, 0xFFFFFFFF, abytes / 4);
        hipMemset(flags, 0, 4096 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        launch_lstm_split(0, kind, Wp, bias, xin, hout, nullptr, flags, ab, Tb, B16, H, 0, B16, 1, mode, 0, 0, nullptr, nullptr, 256, (unsigned)(rep + 1), 0);   // B16 = 32 at H = 384: the dense form, two workgroups per CU
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep < 2 || rep >= nrep - 2) printf("rep %d: layer %.3f ms = %.3f us/step = %.0f cycles/step\n", rep, ms, ms * 1e3 / Tb, ms * 1e3 / Tb * 2400);
    }
    unsigned abv = 0; hipMemcpy(&abv, ab, 4, hipMemcpyDeviceToHost);
    printf("abort word %u\n", abv);
    std::vector<unsigned long long> h(ndbg);
    hipMemcpy(h.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
    auto T = [&](int b, int w, int st, int k) { return h[(((size_t)b * 8 + w) * 32 + st) * 16 + k]; };
    for (int b : { 0, 8, 256 }) {
        if (b >= nwg) continue;
        const unsigned long long base = T(b, 0, 0, 0);
        for (int st = 0; st < 3; st++)
            for (int w = 0; w < 8; w++) {
                printf("blk %3d step %d wave %d:", b, 100 + st, w);
                for (int k = 0; k < 6; k++) { unsigned long long v = T(b, w, st, k); printf(" %7lld", v ? (long long)(v - base) : -1LL); }
                if (w < 4) printf("  | proj %5lld bar1 %5lld gate %5lld bar2 %5lld\n", (long long)(T(b,w,st,2)-T(b,w,st,0)), (long long)(T(b,w,st,3)-T(b,w,st,2)),
                                  (long long)(T(b,w,st,4)-T(b,w,st,3)), (long long)(T(b,w,st,5)-T(b,w,st,4)));
                else printf("  | poll %5lld mfma %5lld bar1 %5lld gate %5lld bar2 %5lld | sweep issue %5lld land %5lld\n", (long long)(T(b,w,st,1)-T(b,w,st,0)), (long long)(T(b,w,st,2)-T(b,w,st,1)),
                            (long long)(T(b,w,st,3)-T(b,w,st,2)), (long long)(T(b,w,st,4)-T(b,w,st,3)), (long long)(T(b,w,st,5)-T(b,w,st,4)),
                            T(b,w,st,6) ? (long long)(T(b,w,st,6)-T(b,w,st,1)) : -1LL, T(b,w,st,7) ? (long long)(T(b,w,st,7)-T(b,w,st,6)) : -1LL);
                if (w >= 4) { printf("        chunk stamps after poll:"); for (int k = 8; k < 14; k++) printf(" %5lld", T(b,w,st,k) ? (long long)(T(b,w,st,k)-T(b,w,st,1)) : -1LL); printf("\n"); }
            }
    }
    return 0;
}
