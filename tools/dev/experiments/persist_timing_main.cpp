// dev harness: run one persistent LSTM layer with timing stamps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "ffhip_internal.hpp"
namespace ffhip { extern unsigned long long *g_persist_dbg; }
using namespace ffhip;
int main(int argc, char** argv) {
    const int H = 384, B16 = argc > 1 ? atoi(argv[1]) : 16, Tb = 800, Bp = 16 * B16;
    float4 *sWp; float *xa, *hout; unsigned *flags, *ab; unsigned long long *dbg;
    hipMalloc(&sWp, (size_t)4*H*H*4); hipMemset(sWp, 0, (size_t)4*H*H*4);
    hipMalloc(&xa, (size_t)Tb*Bp*H*4*4); hipMemset(xa, 0, (size_t)Tb*Bp*H*4*4);
    hipMalloc(&hout, (size_t)Tb*Bp*H*4);
    hipMalloc(&flags, 4096*4); hipMalloc(&ab, 4); hipMemset(ab, 0, 4);
    hipMalloc(&dbg, (size_t)Tb*4*6*8); hipMemset(dbg, 0, (size_t)Tb*4*6*8);
    g_persist_dbg = dbg;
    for (int rep = 0; rep < 3; rep++) {
        hipMemsetD32(hout, 0xFFFFFFFF, (size_t)Tb*Bp*H);
        hipMemset(flags, 0, 4096*4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        launch_rnn_persist(0, 0, sWp, xa, hout, flags, ab, Tb, B16, H, 0, B16, 1, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("layer %.3f ms = %.3f us/step\n", ms, ms*1e3/Tb);
    }
    std::vector<unsigned long long> h((size_t)Tb*4*6);
    hipMemcpy(h.data(), dbg, h.size()*8, hipMemcpyDeviceToHost);
    // average phase durations over steps 100..700 per wave
    for (int w = 0; w < 4; w++) {
        double d[6] = {0}; int n = 0;
        for (int i = 100; i < 700; i++) {
            unsigned long long *p = &h[((size_t)i*4 + w)*6], *pn = &h[((size_t)(i+1)*4 + w)*6];
            d[0] += (double)(p[1]-p[0]); d[1] += (double)(p[2]-p[1]); d[2] += (double)(p[3]-p[2]);
            if (p[4]) { d[3] += (double)(p[4]-p[3]); d[4] += (double)(pn[0]-p[4]); } else d[4] += (double)(pn[0]-p[3]);
            d[5] += (double)(pn[0]-p[0]); n++;
        }
        printf("wave %d: wait+load %.0f  mfma %.0f  part+barrier %.0f  reduce+gates %.0f  store->next %.0f  step %.0f (cycles of readcyclecounter)\n",
               w, d[0]/n, d[1]/n, d[2]/n, d[3]/n, d[4]/n, d[5]/n);
    }
    return 0;
}
