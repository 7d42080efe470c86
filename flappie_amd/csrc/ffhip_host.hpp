// ffhip_host.hpp -- host-side helpers shared by ffhip_engine.hip and ffhip_layers.hip (not public).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <vector>

#include "../../include/ffhip.h"
#include "ffhip_internal.hpp"

struct ffhip_engine {
    int device = 0;
    hipDeviceProp_t prop;
    hipStream_t streams[4] = { nullptr, nullptr, nullptr, nullptr };      // batches take them in turn (FFHIP_DEBUG=streams=2..4)
    int nstreams = 4, next_stream = 0;
    int profiling = 0;
    // Persistent recurrent kernels spin on their peers: every workgroup of a launch must be resident.
    // Two batches (streams) may run such kernels at the same time only if both fit; otherwise the
    // launches are chained through this event.
    hipEvent_t persist_done = nullptr;
    hipEvent_t head_done = nullptr; int head_done_rec = 0;      // behind the CRF head of the last batch run: a packed batch's set-up waits for it (ffhip_engine.hip apply_packed)
    int persist_chained = 0;
    hipEvent_t batch_done = nullptr;     // end of the last submitted batch (any stream): see batch_run_impl, "whole batches one after the other"
    int batch_done_rec = 0;
    hipEvent_t done_ring[4] = { nullptr, nullptr, nullptr, nullptr };      // end of the decode of the last four submitted batches (before their copies to the host)
    unsigned done_head = 0;
    int persist_last_half = 0;  // the engine's last persistent launch left room for a twin of its size: only then may the next half-chip launch run beside it
    // A persistent layer kernel whose workgroups are not all resident (another tenant on the GPU, e.g. a second flappie
    // process) gives up through its bounded waits (abort word).  ffhip_batch_finish then re-runs the batch on the
    // launch-per-step kernels, which need no co-residency, and the next `stepwise_batches` runs go there directly:
    // co-tenancy becomes a slowdown, not FFHIP_ETIMEOUT.
    int stepwise_batches = 0;
    int fallbacks = 0;          // how often that happened (ffhip_debug_fallback_count)
    unsigned long long f32_reruns = 0;      // reads that left the split format's range and were run again on the f32 path (ffhip_engine_f32_reruns)
    int in_flight = 0;          // batches between ffhip_batch_run and ffhip_batch_finish: kernels of a batch submitted beside another
                                // take the shapes that fit next to a resident layer launch (k_conv_split<2, 2>)
    // Signal preparation (ffhip_prep.hip) runs on a stream of its own -- beside the batches, not queued behind one of them -- and
    // keeps its buffers: a pinned staging area (one packed upload per chunk instead of one per read), the kernel's scratch, and a
    // pool of output buffers handed to ffhip_prep objects (hipMalloc / hipFree per chunk would synchronise the device each time).
    hipStream_t prep_stream = nullptr;
    void *prep_pin = nullptr;
    size_t prep_pin_cap = 0;
    void *prep_scratch[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t prep_scratch_cap[4] = { 0, 0, 0, 0 };
    std::vector<std::pair<void *, size_t>> prep_pool;       // free output buffers (pointer, bytes)
    double rehearsal_busy_until = 0.0;                      // FFHIP_DEBUG_HOST_REHEARSAL_MSPS (ffhip_engine.hip): when the emulated GPU is free again
};

struct ffhip_prep;

namespace ffhip {

int set_err(int code, const char *fmt, ...);        // records the thread's last error text, returns `code`

#define HIP_TRY(expr, ret)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            ffhip::set_err(FFHIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ret;                                                                             \
        }                                                                                           \
    } while (0)

// W(row m, k) accessor -> A-fragment order [Mt][K16][64 lanes][4]; rows/cols beyond the matrix are 0
template <class F>
static std::vector<float> pack_afrag(int Mt, int K16, F w) {
    std::vector<float> out((size_t)Mt * K16 * 256, 0.0f);
    for (int mt = 0; mt < Mt; mt++)
        for (int k16 = 0; k16 < K16; k16++)
            for (int lane = 0; lane < 64; lane++) {
                const int i = lane & 15, kq = lane >> 4;
                for (int e = 0; e < 4; e++)
                    out[(((size_t)mt * K16 + k16) * 64 + lane) * 4 + e] = w(mt * 16 + i, k16 * 16 + kq * 4 + e);
            }
    return out;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// strided-convolution column plan (layers.c:216-271 restated in index space); returns Tout, <0 on failure
int build_conv_plan(int T, int winlen, int s, std::vector<int> &a, std::vector<int> &bq);

// Process-wide pool of device buffers (power-of-two size classes from 4 KiB): the device images that flappie matrices own and
// the scratch of the single-matrix calls.  hipMalloc / hipFree per call cost more than the kernels of a one-read call, and hipFree
// synchronises the device.  A buffer goes back only when no enqueued work uses it (the single-matrix calls are synchronous).
void *pool_get(size_t bytes);
void pool_put(void *p);
void pool_trim(int device);           // hipFree every unused buffer of that device (its last engine is destroyed)
void pool_engine_born(int device);
bool pool_engine_gone(int device);    // true: it was the device's last engine
int pool_default_device();
int pool_device_of(const void *p);    // -1: not a pool buffer
void pool_state(unsigned long long out[4]);
#ifdef FFHIP_TEST_HOOKS
double rehearsal_rate();              // test hook FFHIP_DEBUG_HOST_REHEARSAL_MSPS (ffhip_engine.hip; the `make hooks` build only); <= 0: off
bool rehearsal_nogpu();
#else
static inline double rehearsal_rate() { return 0.0; }      // the release library: no hook, the code behind it folds away
static inline bool rehearsal_nogpu() { return false; }
#endif
void image_remember(const void *owner, void *dev);
void *image_forget(const void *owner);

// every host<->device copy of the library goes through these: ffhip_copy_counts (include/ffhip.h)
hipError_t counted_memcpy_async(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s);
hipError_t counted_memcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind);
hipError_t counted_memcpy_2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s);
#define hipMemcpyAsync(...) ::ffhip::counted_memcpy_async(__VA_ARGS__)
#define hipMemcpy(...) ::ffhip::counted_memcpy(__VA_ARGS__)
#define hipMemcpy2DAsync(...) ::ffhip::counted_memcpy_2d_async(__VA_ARGS__)

// scratch device allocations of one single-matrix call
struct TmpDev {
    std::vector<void *> p;
    ~TmpDev() { for (void *q : p) pool_put(q); }
    void *get(size_t bytes) {
        void *d = pool_get(bytes ? bytes : 4);
        if (d) p.push_back(d);
        return d;
    }
    void *upload(const void *host, size_t bytes, hipStream_t s) {
        void *d = get(bytes);
        if (d && hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return nullptr;
        return d;
    }
};

// ---- device images of flappie matrices (include/ffhip.h, "flappie matrices with a device image")
int matrix_policy();
static inline bool mat_has_dev(const ffhip_mat &m) { return m.dev && m.dev_state && *m.dev && *m.dev_state >= 1 && matrix_policy() != 0; }
// the image an operator READS: the matrix's device image when it has one, an uploaded copy of the host image otherwise
static inline float *mat_in(TmpDev &t, const ffhip_mat &m, hipStream_t s) {
    if (mat_has_dev(m)) return (float *)*m.dev;
    return (float *)t.upload(m.data, m.nc * m.stride * sizeof(float), s);
}
// the image an operator WRITES: `lazy` (the result stays on the device, the matrix owns the buffer) or scratch to be downloaded
static inline float *mat_out(TmpDev &t, const ffhip_mat &m, bool lazy) {
    const size_t bytes = m.nc * m.stride * sizeof(float);
    if (lazy && m.dev && m.dev_state) {
        if (!*m.dev) *m.dev = pool_get(bytes);
        return (float *)*m.dev;
    }
    return (float *)t.get(bytes);
}
// after the kernels that wrote `d` were enqueued on s: mark the device image current (lazy) or download it
static inline hipError_t mat_done(const ffhip_mat &m, float *d, bool lazy, hipStream_t s) {
    if (lazy && m.dev && m.dev_state && d == (float *)*m.dev) { *m.dev_state = 2; return hipSuccess; }
    if (m.dev && m.dev_state && *m.dev && d != (float *)*m.dev) { pool_put(*m.dev); *m.dev = nullptr; *m.dev_state = 0; }      // a reused output: its old device image is stale
    return hipMemcpyAsync(m.data, d, m.nc * m.stride * sizeof(float), hipMemcpyDeviceToHost, s);
}

static inline bool flipflop_dims(size_t nparam, size_t stride, int *nbase) {
    const int nb = (int)roundf((-1.0f + sqrtf(1.0f + 2.0f * (float)nparam)) / 2.0f);
    if (nb < 1 || (size_t)(2 * nb * (nb + 1)) != nparam || 2 * nb > kMaxState || nparam > 64 || stride < nparam) return false;
    *nbase = nb;
    return true;
}

// device address and length of a prepared read's kept samples (ffhip_prep.hip); nullptr if rejected
const float *prep_device_signal(const struct ::ffhip_prep *p, int read, size_t *len);
void prep_mark_used(const struct ::ffhip_prep *p, hipStream_t s);      // an asynchronous reader of the prepared signals was enqueued on s

}  // namespace ffhip
