// dev harness: k_lstm_flow (ffhip_rnn_split.hip built with -DFFHIP_TIMELINE): per-wave stamps of half-steps 200..215
// x waves: 0 start, 1 px slot free, 2 projection stored, 3 prefetch issued
// h waves: 0 start, 1 projection there, 2 MFMAs done (6 = fell back to polling), 3 ph free, 4 ph stored
// g waves: 0 start, 1 pre-activations there, 2 gate math done, 3 published
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "ffhip_internal.hpp"
namespace ffhip { extern unsigned long long *g_split_dbg; }
using namespace ffhip;
int main(int argc, char **argv) {
    setenv("FFHIP_FLOW", "1", 1);
    const int H = 384, B16 = argc > 1 ? atoi(argv[1]) : 16, Tb = 400;
    void *Wp, *xin, *hout; float *bias; unsigned *flags, *ab; unsigned long long *dbg;
    const size_t wbytes = (size_t)2 * 4 * H * H * 2 * kSplitNS, abytes = split_bytes((size_t)Tb * B16, H);
    hipMalloc(&Wp, wbytes); hipMemset(Wp, 0, wbytes);
    hipMalloc(&bias, 4 * H * 4); hipMemset(bias, 0, 4 * H * 4);
    hipMalloc(&xin, abytes); hipMemset(xin, 0, abytes);
    hipMalloc(&hout, abytes);
    hipMalloc(&flags, 4096 * 4); hipMemset(flags, 0, 4096 * 4); hipMalloc(&ab, 8); hipMemset(ab, 0, 8);
    const int nwg = ((B16 + 1) / 2) * 32;
    const size_t ndbg = (size_t)nwg * 12 * 16 * 8;
    hipMalloc(&dbg, ndbg * 8); hipMemset(dbg, 0, ndbg * 8);
    g_split_dbg = dbg;
    const int nrep = 30;
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
    auto T = [&](int b, int w, int st, int k) { return h[(((size_t)b * 12 + w) * 16 + st) * 8 + k]; };
    for (int b : { 0, 9 }) {
        if (b >= nwg) continue;
        const unsigned long long base = T(b, 4, 0, 0);
        for (int st = 0; st < 5; st++)
            for (int w = 0; w < 11; w++) {
                printf("blk %3d half-step %d %s wave %2d:", b, 200 + st, w < 4 ? "x" : w < 8 ? "h" : "g", w);
                for (int k = 0; k < 7; k++) { unsigned long long v = T(b, w, st, k); printf(" %7lld", v ? (long long)(v - base) : -1LL); }
                printf("\n");
            }
    }
    return 0;
}
