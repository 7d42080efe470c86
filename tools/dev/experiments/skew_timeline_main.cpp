// dev harness: k_lstm_skew (ffhip_rnn_split.hip built with -DFFHIP_TIMELINE) with per-wave phase timestamps of half-steps 200..231.
// x waves: 0 start, 1 projection done, 2 prefetch issued, 3 after barrier 1, 4 back gate done, 5 after barrier 2
// h waves: 0 start, 1 MFMAs done, 2 partials in LDS, 3 after barrier 1, 6 sweep issued, 4 front gate done, 5 after barrier 2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "ffhip_internal.hpp"
namespace ffhip { extern unsigned long long *g_split_dbg; }
using namespace ffhip;
int main(int argc, char **argv) {
    setenv("FFHIP_SKEW", "1", 1);
    const int H = argc > 2 ? atoi(argv[2]) : 384, B16 = argc > 1 ? atoi(argv[1]) : 16, Tb = 400;
    void *Wp, *xin, *hout; float *bias; unsigned *flags, *ab; unsigned long long *dbg;
    const size_t wbytes = (size_t)2 * 4 * H * H * 2 * kSplitNS, abytes = split_bytes((size_t)Tb * B16, H);
    hipMalloc(&Wp, wbytes); hipMemset(Wp, 0, wbytes);
    hipMalloc(&bias, 4 * H * 4); hipMemset(bias, 0, 4 * H * 4);
    hipMalloc(&xin, abytes); hipMemset(xin, 0, abytes);
    hipMalloc(&hout, abytes);
    hipMalloc(&flags, 4096 * 4); hipMemset(flags, 0, 4096 * 4); hipMalloc(&ab, 8); hipMemset(ab, 0, 8);
    const int nwg = ((B16 + 1) / 2) * 32;
    const size_t ndbg = (size_t)nwg * 8 * 32 * 16;
    hipMalloc(&dbg, ndbg * 8); hipMemset(dbg, 0, ndbg * 8);
    g_split_dbg = dbg;
    const int nrep = argc > 3 ? atoi(argv[3]) : 30;
    for (int rep = 0; rep < nrep; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        launch_lstm_split(0, 0, Wp, bias, xin, hout, nullptr, flags, ab, Tb, B16, H, 0, B16, 0, 0, 0, 0, nullptr, nullptr, 256, (unsigned)rep + 1);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep < 2 || rep >= nrep - 2) printf("rep %d: layer %.3f ms = %.0f cycles/step\n", rep, ms, ms * 1e3 / Tb * 2400);
    }
    unsigned abv = 0; hipMemcpy(&abv, ab, 4, hipMemcpyDeviceToHost);
    printf("abort word %u\n", abv);
    std::vector<unsigned long long> h(ndbg);
    hipMemcpy(h.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
    auto T = [&](int b, int w, int st, int k) { return h[(((size_t)b * 8 + w) * 32 + st) * 16 + k]; };
    for (int b : { 0, 9 }) {
        if (b >= nwg) continue;
        const unsigned long long base = T(b, 0, 0, 0);
        for (int st = 0; st < 4; st++)
            for (int w = 0; w < 8; w++) {
                printf("blk %3d half-step %d wave %d:", b, 200 + st, w);
                for (int k = 0; k < 7; k++) { unsigned long long v = T(b, w, st, k); printf(" %7lld", v ? (long long)(v - base) : -1LL); }
                if (w < 4) printf("  | proj %5lld prefetch %5lld bar1 %5lld back %5lld bar2 %5lld\n", (long long)(T(b,w,st,1)-T(b,w,st,0)), (long long)(T(b,w,st,2)-T(b,w,st,1)), (long long)(T(b,w,st,3)-T(b,w,st,2)),
                                  (long long)(T(b,w,st,4)-T(b,w,st,3)), (long long)(T(b,w,st,5)-T(b,w,st,4)));
                else printf("  | mfma %5lld lds %5lld bar1 %5lld sweep-issue %5lld front %5lld bar2 %5lld\n", (long long)(T(b,w,st,1)-T(b,w,st,0)), (long long)(T(b,w,st,2)-T(b,w,st,1)),
                            (long long)(T(b,w,st,3)-T(b,w,st,2)), (long long)(T(b,w,st,6)-T(b,w,st,3)), (long long)(T(b,w,st,4)-T(b,w,st,6)), (long long)(T(b,w,st,5)-T(b,w,st,4)));
            }
    }
    return 0;
}
