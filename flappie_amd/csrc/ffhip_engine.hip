// ffhip_engine.hip -- host side of the C-ABI in include/ffhip.h: engine, resident model, batch.
// One engine per GPU; weights are re-packed once into MFMA fragment order and stay in HBM; a batch
// owns all of its workspace (sized for the whole pipeline, no allocation on the run path).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <ctype.h>
#include <queue>
#include <vector>
#include <map>
#include <algorithm>
#include <string>
#include <atomic>
#include <mutex>
#include <unordered_map>

#include "../../include/ffhip.h"
#include "ffhip_internal.hpp"
#include "ffhip_host.hpp"

using namespace ffhip;

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
int ffhip::set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *ffhip_last_error(void) { return g_err; }
extern "C" const char *ffhip_version(void) { return "ffhip 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------ engine
extern "C" int ffhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" ffhip_engine *ffhip_engine_create(int device) {
    int n = ffhip_device_count();
    if (n <= 0 || device < 0 || device >= n) { set_err(FFHIP_ENODEV, "no HIP device %d (found %d)", device, n); return nullptr; }
    ffhip_engine *e = new ffhip_engine();
    e->device = device;
    HIP_TRY(hipSetDevice(device), (delete e, nullptr));
    // how a host thread waits for the device: HIP's default spins (a CPU per waiting thread -- eight ranks of a node, a CPU each); FFHIP_DEBUG=blocking_sync
    // sleeps on the interrupt instead (bench.py asks for it when it runs more than one rank; profiles/r06_blocking_sync.txt)
    if (dbg("blocking_sync")) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    HIP_TRY(hipGetDeviceProperties(&e->prop, device), (delete e, nullptr));
    if (strncmp(e->prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(FFHIP_ENODEV, "device %d is %s; this library is built for gfx950 only", device, e->prop.gcnArchName);
        delete e;
        return nullptr;
    }
    if (const char *ns = dbg("streams")) e->nstreams = atoi(ns) < 2 ? 2 : (atoi(ns) > 4 ? 4 : atoi(ns));
    for (int i = 0; i < e->nstreams; i++) HIP_TRY(hipStreamCreateWithFlags(&e->streams[i], hipStreamNonBlocking), (delete e, nullptr));
    HIP_TRY(hipStreamCreateWithFlags(&e->prep_stream, hipStreamNonBlocking), (delete e, nullptr));
    HIP_TRY(hipEventCreateWithFlags(&e->persist_done, hipEventDisableTiming), (delete e, nullptr));
    HIP_TRY(hipEventCreateWithFlags(&e->head_done, hipEventDisableTiming), (delete e, nullptr));
    HIP_TRY(hipEventCreateWithFlags(&e->batch_done, hipEventDisableTiming), (delete e, nullptr));
    for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreateWithFlags(&e->done_ring[i], hipEventDisableTiming), (delete e, nullptr));
    ffhip::pool_engine_born(device);
    return e;
}

// gate arithmetic of the split layer kernels: 0 = the reference's exp_ps / division replayed bit for bit, 1 = v_exp_f32 / v_rcp_f32, 2 = those with a
// two-word exponent and a Newton step (ffhip_math.hpp logistic_hw) -- the default since round 6 (include/ffhip.h, profiles/r06_gates_*.txt); run flags first,
// then FFHIP_FAST_GATES=0|1|2 for a whole process
static int gate_level(unsigned flags) {
    if (flags & FFHIP_RUN_EXACT_GATES) return 0;
    if (flags & FFHIP_RUN_FAST_GATES2) return 2;
    if (flags & FFHIP_RUN_FAST_GATES) return 1;
    const char *e = getenv("FFHIP_FAST_GATES");
    if (e && e[0]) return atoi(e) >= 2 ? 2 : (e[0] == '0' ? 0 : 1);
    return 2;
}

// ---- development switches: FFHIP_DEBUG=token[,token=value ...] (ffhip_internal.hpp) ---------------------------
namespace ffhip {
const char *dbg(const char *token) {
    static std::mutex mu;
    static std::string seen;                                  // the variable's text the table below was built from
    static std::vector<std::pair<std::string, std::string>> table;
    std::lock_guard<std::mutex> lock(mu);                     // (getenv inside: a test's setenv in another thread must not race the read)
    static bool legacy_checked = false;
    if (!legacy_checked) {
        // rounds 1-4 had one variable per switch; a script that still sets one now compares the default with itself -- say so, once
        legacy_checked = true;
        static const char *const legacy[] = { "FFHIP_NO_SPLIT_HEAD", "FFHIP_CONV_WS", "FFHIP_CONV_SMALL_U", "FFHIP_STREAMS", "FFHIP_FRONT_ORDER", "FFHIP_NO_DECODE_WAIT",
            "FFHIP_CONV1_TN", "FFHIP_NO_PACK", "FFHIP_NO_SPLIT", "FFHIP_NO_FUSE", "FFHIP_NO_PAIR", "FFHIP_NO_DENSE", "FFHIP_PERSIST_MODE", "FFHIP_LEAN_CONV",
            "FFHIP_NO_SPLIT_CONV", "FFHIP_NO_BATCH_ORDER", "FFHIP_CRF_LOGSPACE", "FFHIP_DECODE_R2", "FFHIP_EXACT_ORDER", "FFHIP_CRF_GENERIC", "FFHIP_SPLIT_TS",
            "FFHIP_SPLIT_DENSE", "FFHIP_DENSE256", "FFHIP_NO_SPLIT_GATE", "FFHIP_NO_HEAD_EXP", "FFHIP_FORCE_ABORT" };
        for (const char *name : legacy)
            if (getenv(name)) {
                std::string tok(name + 6);
                for (auto &c : tok) c = (char)tolower((unsigned char)c);
                fprintf(stderr, "libffhip: %s is no longer read (one variable since round 5: FFHIP_DEBUG=%s[=value], INTEGRATION.md section 6)\n", name, tok.c_str());
            }
    }
    const char *e = getenv("FFHIP_DEBUG");
    if (!e || !e[0]) return nullptr;
    if (seen != e) {                                          // (tests change the variable between runs of one process)
        seen = e;
        std::vector<std::pair<std::string, std::string>> t;
        size_t i = 0;
        while (i <= seen.size()) {
            size_t j = seen.find_first_of(", ;", i);
            if (j == std::string::npos) j = seen.size();
            if (j > i) {
                const std::string item = seen.substr(i, j - i);
                const size_t eq = item.find('=');
                t.emplace_back(eq == std::string::npos ? item : item.substr(0, eq), eq == std::string::npos ? std::string() : item.substr(eq + 1));
            }
            i = j + 1;
        }
        // (values handed out earlier stay valid: the old strings are kept, a few bytes per change of the variable)
        static std::vector<std::vector<std::pair<std::string, std::string>>> retired;
        retired.push_back(std::move(table));
        table = std::move(t);
    }
    for (const auto &kv : table) if (kv.first == token) return kv.second.c_str();
    return nullptr;
}
}  // namespace ffhip

#ifdef FFHIP_TEST_HOOKS
// ---- host-load rehearsal (TEST HOOK of the -DFFHIP_TEST_HOOKS build only: `make hooks` -> tools/test_hooks/libffhip.so, loaded by
// tools/host_scaling.py's emulated processes through LD_LIBRARY_PATH; the release library has none of this -- VERDICT r4 weak 12) ----
// FFHIP_DEBUG_HOST_REHEARSAL_MSPS=<rate>: this process evaluates NO network.  A run produces placeholder results (calls of 0.4 bases per
// block, all 'A') and is "busy" for (samples of the batch) / rate on an emulated GPU that works its batches one after the other;
// ffhip_batch_finish sleeps until then.  With FFHIP_DEBUG_HOST_REHEARSAL_NOGPU=1 beside it the steady state touches the GPU not at all:
// the signal preparation packs the chunk into its pinned staging buffer (the host's share of it) and stops there -- fixed trims instead
// of the segmentation, no upload -- and the placeholder results are written on the host.  Eight processes that SHARE one physical GPU
// otherwise measure that GPU's scheduler (32+ queues of 8 contexts, time-sliced), which a node with a GPU per process does not have.
// Everything else -- fast5 reader processes, pipes, staging copies, batching, FASTQ formatting and writing -- is the real thing.
// Announced on stderr; never a fallback, never set by the library itself.
namespace ffhip {
double rehearsal_rate() {
    static const double v = [] {                              // (a function-local static: initialised once, thread-safe)
        const char *e = getenv("FFHIP_DEBUG_HOST_REHEARSAL_MSPS");
        const double r = e ? atof(e) : -1.0;
        if (r > 0) fprintf(stderr, "ffhip: FFHIP_DEBUG_HOST_REHEARSAL_MSPS=%g%s -- NO NETWORK IS EVALUATED in this process: every batch returns placeholder calls "
                                   "after (its samples) / %g us (host-side load rehearsal, tools/host_scaling.py)\n", r,
                           getenv("FFHIP_DEBUG_HOST_REHEARSAL_NOGPU") ? " without the GPU" : "", r);
        return r;
    }();
    return v;
}
bool rehearsal_nogpu() { return rehearsal_rate() > 0 && getenv("FFHIP_DEBUG_HOST_REHEARSAL_NOGPU") != nullptr; }
}  // namespace ffhip
#endif

// ---- device buffer pool and copy accounting (ffhip_host.hpp) ------------------------------------------------------
#undef hipMemcpyAsync
#undef hipMemcpy
#undef hipMemcpy2DAsync
namespace ffhip {
namespace {
std::mutex g_pool_mu;
// every buffer the pool handed out or holds -> its rounded size and the device it lives on; free lists per (device, size)
struct PoolBuf { size_t bytes; int device; };
std::unordered_map<void *, PoolBuf> g_pool_buf;
std::map<std::pair<int, size_t>, std::vector<void *>> g_pool_free;
std::unordered_map<int, int> g_pool_engines;                  // live engines per device: the last one to go trims that device's lists
int g_pool_default_device = 0;                                // device of the most recently created engine (entry points without an engine argument)
// matrices whose device image may be orphaned by a plain free() of the struct (flappie_matrix.c): struct address -> image
std::unordered_map<const void *, void *> g_image_owner;
std::atomic<unsigned long long> g_copy[5];
std::atomic<int> g_matrix_policy{ -1 };
// powers of two up to 64 MiB (few classes, quick reuse); beyond that multiples of 16 MiB -- a power of two wastes up to half of a multi-GB scratch
size_t round_size(size_t bytes) {
    if (bytes <= ((size_t)64 << 20)) { size_t c = 4096; while (c < bytes) c <<= 1; return c; }
    const size_t q = (size_t)16 << 20;
    return (bytes + q - 1) / q * q;
}
void count_copy(size_t n, hipMemcpyKind kind) {
    if (kind == hipMemcpyHostToDevice) { g_copy[0]++; g_copy[1] += n; }
    else if (kind == hipMemcpyDeviceToHost) {
        g_copy[2]++; g_copy[3] += n;
        unsigned long long cur = g_copy[4].load();
        while (n > cur && !g_copy[4].compare_exchange_weak(cur, n)) { }
    }
}
}  // namespace
void *pool_get(size_t bytes) {
    const size_t sz = round_size(bytes);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_free.find({ dev, sz });
        if (it != g_pool_free.end() && !it->second.empty()) { void *p = it->second.back(); it->second.pop_back(); return p; }
    }
    void *d = nullptr;
    if (hipMalloc(&d, sz) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_buf[d] = PoolBuf{ sz, dev };
    return d;
}
void pool_put(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    // whoever releases an image, the owner table must not name it any more (a later address reuse would release it a second time)
    for (auto it = g_image_owner.begin(); it != g_image_owner.end(); ) { if (it->second == p) it = g_image_owner.erase(it); else ++it; }
    auto it = g_pool_buf.find(p);
    if (it == g_pool_buf.end()) { hipFree(p); return; }      // not ours (never happens): free it the plain way
    g_pool_free[{ it->second.device, it->second.bytes }].push_back(p);
}
// free what the pool holds for `device` (its last engine is going); buffers still with their owners stay recorded
void pool_trim(int device) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &kv : g_pool_free) {
        if (kv.first.first != device) continue;
        for (void *p : kv.second) { g_pool_buf.erase(p); hipFree(p); }
        kv.second.clear();
    }
}
int pool_device_of(const void *p) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_buf.find((void *)p);
    return it == g_pool_buf.end() ? -1 : it->second.device;
}
void pool_engine_born(int device) { std::lock_guard<std::mutex> lk(g_pool_mu); g_pool_engines[device]++; g_pool_default_device = device; }
bool pool_engine_gone(int device) { std::lock_guard<std::mutex> lk(g_pool_mu); return --g_pool_engines[device] <= 0; }
int pool_default_device() { std::lock_guard<std::mutex> lk(g_pool_mu); return g_pool_default_device; }
void pool_state(unsigned long long out[4]) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    std::unordered_map<void *, int> seen;
    unsigned long long nfree = 0, dup = 0;
    for (auto &kv : g_pool_free) for (void *p : kv.second) { nfree++; if (seen[p]++) dup++; }
    out[0] = g_pool_buf.size(); out[1] = nfree; out[2] = g_image_owner.size(); out[3] = dup;
}
void image_remember(const void *owner, void *dev) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_image_owner[owner] = dev;
    static bool warned = false;
    if (g_image_owner.size() > 4096 && !warned) {
        warned = true;
        fprintf(stderr, "ffhip: more than 4096 transition matrices are alive with a device image, or were released with a plain free() whose addresses were never "
                        "reused: their device buffers stay allocated (free_flappie_matrix releases them; FLAPPIE_HOST_MATRICES=1 avoids device images)\n");
    }
}
void *image_forget(const void *owner) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_image_owner.find(owner);
    if (it == g_image_owner.end()) return nullptr;
    void *d = it->second;
    g_image_owner.erase(it);
    return d;
}
hipError_t counted_memcpy_async(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s) { count_copy(n, kind); return ::hipMemcpyAsync(dst, src, n, kind, s); }
hipError_t counted_memcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind) { count_copy(n, kind); return ::hipMemcpy(dst, src, n, kind); }
hipError_t counted_memcpy_2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s) {
    count_copy(width * height, kind);
    return ::hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, s);
}
int matrix_policy() {
    int p = g_matrix_policy.load();
    if (p < 0) { const char *e = getenv("FLAPPIE_HOST_MATRICES"); p = (e && e[0] && e[0] != '0') ? 0 : 1; g_matrix_policy.store(p); }
    return p;
}
}  // namespace ffhip
#define hipMemcpyAsync(...) ::ffhip::counted_memcpy_async(__VA_ARGS__)
#define hipMemcpy(...) ::ffhip::counted_memcpy(__VA_ARGS__)
#define hipMemcpy2DAsync(...) ::ffhip::counted_memcpy_2d_async(__VA_ARGS__)

extern "C" void ffhip_set_matrix_policy(int device_images) { ffhip::g_matrix_policy.store(device_images ? 1 : 0); }
extern "C" int ffhip_matrix_policy(void) { return ffhip::matrix_policy(); }
extern "C" void ffhip_copy_counts(unsigned long long out[5], int reset) {
    for (int i = 0; i < 5; i++) { if (out) out[i] = ffhip::g_copy[i].load(); if (reset) ffhip::g_copy[i].store(0); }
}
extern "C" void ffhip_dev_release(void *dev) { ffhip::pool_put(dev); }
extern "C" void ffhip_debug_pool_state(unsigned long long out[4]) { if (out) ffhip::pool_state(out); }
extern "C" void ffhip_dev_remember(const void *owner, void *dev) { if (owner && dev) ffhip::image_remember(owner, dev); }
extern "C" void *ffhip_dev_forget(const void *owner) { return owner ? ffhip::image_forget(owner) : nullptr; }
extern "C" int ffhip_dev_download(const void *dev, float *host, size_t nfloat) {
    if (!dev || !host) return set_err(FFHIP_EINVAL, "null image");
    const int on = ffhip::pool_device_of(dev);
    if (on >= 0) hipSetDevice(on);
    HIP_TRY(hipMemcpy(host, dev, nfloat * sizeof(float), hipMemcpyDeviceToHost), FFHIP_EHIP);
    return FFHIP_OK;
}
extern "C" void *ffhip_dev_upload(const float *host, size_t nfloat) {
    if (!host || !nfloat) return nullptr;
    // (no engine argument: the device of the engine created last -- the host layer has one engine --; the calling thread's own device is put back,
    // so that an upload does not move a thread that drives another GPU: ADVICE r4)
    int prev = -1;
    hipGetDevice(&prev);
    hipSetDevice(ffhip::pool_default_device());
    void *d = ffhip::pool_get(nfloat * sizeof(float));
    if (d && hipMemcpy(d, host, nfloat * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { ffhip::pool_put(d); d = nullptr; }
    if (prev >= 0 && prev != ffhip::pool_default_device()) hipSetDevice(prev);
    return d;
}

extern "C" void ffhip_engine_destroy(ffhip_engine *e) {
    if (!e) return;
    hipSetDevice(e->device);
    hipDeviceSynchronize();
    if (ffhip::pool_engine_gone(e->device)) ffhip::pool_trim(e->device);
    for (int i = 0; i < 4; i++) if (e->streams[i]) hipStreamDestroy(e->streams[i]);
    if (e->prep_stream) hipStreamDestroy(e->prep_stream);
    if (e->prep_pin) hipHostFree(e->prep_pin);
    for (int i = 0; i < 4; i++) if (e->prep_scratch[i]) hipFree(e->prep_scratch[i]);
    for (auto &b : e->prep_pool) hipFree(b.first);
    if (e->persist_done) hipEventDestroy(e->persist_done);
    if (e->head_done) hipEventDestroy(e->head_done);
    if (e->batch_done) hipEventDestroy(e->batch_done);
    for (int i = 0; i < 4; i++) if (e->done_ring[i]) hipEventDestroy(e->done_ring[i]);
    delete e;
}

extern "C" int ffhip_engine_synchronize(ffhip_engine *e) {
    if (!e) return set_err(FFHIP_EINVAL, "null engine");
    for (int i = 0; i < e->nstreams; i++) HIP_TRY(hipStreamSynchronize(e->streams[i]), FFHIP_EHIP);
    return FFHIP_OK;
}

extern "C" int ffhip_engine_info(const ffhip_engine *e, char *name, size_t name_len, int *ncu, int *clock_khz) {
    if (!e) return set_err(FFHIP_EINVAL, "null engine");
    if (name && name_len) { strncpy(name, e->prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (ncu) *ncu = e->prop.multiProcessorCount;
    if (clock_khz) *clock_khz = e->prop.clockRate;
    return FFHIP_OK;
}

extern "C" int ffhip_engine_set_profiling(ffhip_engine *e, int on) {
    if (!e) return set_err(FFHIP_EINVAL, "null engine");
    e->profiling = on;
    return FFHIP_OK;
}

// ------------------------------------------------------------------------------------ model
struct ConvDev {
    int Fin = 0, Fout = 0, winlen = 0, stride = 1;
    float *taps = nullptr;      // [Fout][winlen][Fin]   (VALU layers)
    float *bias = nullptr;      // [Fout] or [Mpad] for the MFMA layer
    float4 *Wp = nullptr;       // A-fragment order     (last layer)
    int K16 = 0, Mpad = 0;
    void *Wsplit = nullptr;     // last layer with 16 input features: fp16 slices in 16x16x32 A order for k_conv_split (ffhip_split.hpp)
    int split_S = 0;            // exponent its accumulators carry: weight exponent + kSplitExpX
};
struct RnnDev {
    float4 *iWp = nullptr, *sWp = nullptr;
    float *bias = nullptr;      // [4*Hp] permuted rows
    int Kin16 = 0;
    void *Wsplit = nullptr;     // both matrices as 16-bit slices in MFMA 16x16x32 A order (ffhip_rnn_split.hip, ffhip_split.hpp), or nullptr
    int split_S = 0;            // the power-of-two exponent both products of this layer carry: sw(Wi) + e(x) = sw(sW) + e(h)
};

struct ffhip_model {
    ffhip_engine *eng = nullptr;
    int kind = 0, cell = 0, nconv = 0, G = 4;
    int H = 0, Hp = 0;          // hidden units, padded to a multiple of 16
    int P = 0, Ps = 0, nbase = 0, nstate = 0;
    int act = ACT_SWISH;
    ConvDev conv[3];
    RnnDev rnn[5];
    float4 *FFp = nullptr;
    float *FFb = nullptr;
    void *FFsplit = nullptr;    // the head's weights as fp16 slices in 16x16x32 A order (k_head_split: reads the last layer's split output)
    int FF_split_S = 0;         // exponent its accumulators carry: weight exponent + kSplitExpH
    std::vector<void *> owned;
};

static void *dev_upload(ffhip_model *m, const void *host, size_t bytes) {
    void *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
    m->owned.push_back(d);
    return d;
}

extern "C" void ffhip_model_free(ffhip_model *m) {
    if (!m) return;
    for (void *p : m->owned) hipFree(p);
    delete m;
}

extern "C" ffhip_model *ffhip_model_upload(ffhip_engine *eng, const ffhip_model_desc *d) {
    if (!eng || !d) { set_err(FFHIP_EINVAL, "null engine or descriptor"); return nullptr; }
    if (d->kind != FFHIP_NET_LSTM5 && d->kind != FFHIP_NET_GRUMOD5 && d->kind != FFHIP_NET_LSTM5_RLE) { set_err(FFHIP_EINVAL, "unknown network kind %d", d->kind); return nullptr; }
    if (d->nconv < 1 || d->nconv > 3) { set_err(FFHIP_EINVAL, "nconv must be 1..3"); return nullptr; }
    hipSetDevice(eng->device);
    ffhip_model *m = new ffhip_model();
    m->eng = eng;
    m->kind = d->kind;
    m->nconv = d->nconv;
    m->cell = (d->kind == FFHIP_NET_GRUMOD5) ? 1 : 0;            // recurrent cell: 0 LSTM, 1 GRUmod
    m->G = (m->cell == 0) ? 4 : 3;
    m->act = (m->cell == 0) ? ACT_SWISH : ACT_TANH;
#define FAIL(...) do { set_err(FFHIP_EINVAL, __VA_ARGS__); ffhip_model_free(m); return nullptr; } while (0)
    for (int l = 0; l < 5; l++)
        if (!d->rnn_iW[l] || !d->rnn_sW[l] || !d->rnn_b[l]) FAIL("missing recurrent layer %d", l);
    if (!d->FF_W || !d->FF_b) FAIL("missing output layer");
    const int H = (int)d->rnn_sW[0]->nr, G = m->G;
    if (H <= 0 || H % 4 != 0) FAIL("hidden size %d must be a positive multiple of 4 (layers.c:1012)", H);
    m->H = H;
    m->Hp = round_up(H, 16);
    const int Hp = m->Hp;

    // ---- convolutions
    int Fin = 1;
    for (int l = 0; l < d->nconv; l++) {
        const_flappie_matrix W = d->conv_W[l], b = d->conv_b[l];
        if (!W || !b) FAIL("missing convolution %d", l);
        ConvDev &c = m->conv[l];
        const int nf_pad = round_up(Fin, 4);
        if ((int)W->nrq * 4 % nf_pad != 0) FAIL("conv %d: filter rows do not match %d input features", l, Fin);
        c.Fin = Fin;
        c.Fout = (int)W->nc;
        c.winlen = (int)(W->nrq * 4) / nf_pad;                 // layers.c:199
        c.stride = d->conv_stride[l];
        if (c.stride < 1 || c.winlen < 1 || c.winlen > 48) FAIL("conv %d: unsupported winlen %d / stride %d", l, c.winlen, c.stride);
        if (b->nr != W->nc) FAIL("conv %d: bias length", l);
        auto tap = [&](int f, int t, int j) -> float {
            const size_t row = (size_t)t * nf_pad + j;
            return row < W->nr ? W->data.f[(size_t)f * W->stride + row] : 0.0f;
        };
        const bool last = (l == d->nconv - 1);
        if (!last) {
            if (c.Fout > 32) FAIL("conv %d: more than 32 filters in a front layer", l);
            std::vector<float> taps((size_t)c.Fout * c.winlen * Fin);
            for (int f = 0; f < c.Fout; f++)
                for (int t = 0; t < c.winlen; t++)
                    for (int j = 0; j < Fin; j++) taps[((size_t)f * c.winlen + t) * Fin + j] = tap(f, t, j);
            c.taps = (float *)dev_upload(m, taps.data(), taps.size() * 4);
            c.bias = (float *)dev_upload(m, b->data.f, (size_t)c.Fout * 4);
            if (!c.taps || !c.bias) { set_err(FFHIP_ENOMEM, "device allocation failed"); ffhip_model_free(m); return nullptr; }
        } else {
            if (c.Fout != H) FAIL("last convolution has %d filters, recurrent stack expects %d", c.Fout, H);
            const int K = c.winlen * Fin;
            c.K16 = (K + 15) / 16;
            c.Mpad = Hp;
            if (c.K16 * 16 > kSamplePad * Fin) FAIL("conv %d: window too long for the sample pad", l);
            auto w = [&](int row, int k) -> float {
                if (row >= c.Fout || k >= K) return 0.0f;
                return tap(row, k / Fin, k % Fin);
            };
            std::vector<float> wp = pack_afrag(Hp / 16, c.K16, w);
            std::vector<float> bias(Hp, 0.0f);
            for (int f = 0; f < c.Fout; f++) bias[f] = b->data.f[f];
            c.Wp = (float4 *)dev_upload(m, wp.data(), wp.size() * 4);
            c.bias = (float *)dev_upload(m, bias.data(), bias.size() * 4);
            if (!c.Wp || !c.bias) { set_err(FFHIP_ENOMEM, "device allocation failed"); ffhip_model_free(m); return nullptr; }
            if (kSplitF16 && Fin == 16 && l > 0 && m->act == ACT_SWISH) {
                // the same weights for the split-operand kernel: [mt][chunk of two taps][slice][lane][8 halves], k = 16 tap + feature
                const int Mt = Hp / 16, NC = (c.winlen + 1) / 2;
                float mx = 0.0f;
                for (int row = 0; row < c.Fout; row++)
                    for (int k = 0; k < K; k++) mx = fmaxf(mx, fabsf(w(row, k)));
                const int sw = split_weight_exp(mx);
                c.split_S = sw + kSplitExpX;
                std::vector<uint16_t> sp((size_t)Mt * NC * kSplitNS * 64 * 8);
                for (int mt = 0; mt < Mt; mt++)
                    for (int ch = 0; ch < NC; ch++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int e = 0; e < 8; e++) {
                                uint16_t sl[kSplitNS];
                                split_host_slices(w(mt * 16 + (lane & 15), ch * 32 + (lane >> 4) * 8 + e), sw, sl);
                                const size_t base = (((size_t)mt * NC + ch) * kSplitNS) * 64 * 8 + (size_t)lane * 8 + e;
                                for (int k2 = 0; k2 < kSplitNS; k2++) sp[base + (size_t)k2 * 64 * 8] = sl[k2];
                            }
                c.Wsplit = dev_upload(m, sp.data(), sp.size() * 2);
                if (!c.Wsplit) { set_err(FFHIP_ENOMEM, "device allocation failed"); ffhip_model_free(m); return nullptr; }
            }
        }
        Fin = c.Fout;
    }

    // ---- recurrent layers: gate rows permuted unit-major (m = 4u + g), padded units are all-zero
    for (int l = 0; l < 5; l++) {
        const_flappie_matrix iW = d->rnn_iW[l], sW = d->rnn_sW[l], b = d->rnn_b[l];
        if ((int)iW->nr != H || (int)iW->nc != G * H) FAIL("rnn %d: iW is %zux%zu, expected %dx%d", l, iW->nr, iW->nc, H, G * H);
        if ((int)sW->nr != H || (int)sW->nc != G * H) FAIL("rnn %d: sW is %zux%zu, expected %dx%d", l, sW->nr, sW->nc, H, G * H);
        if ((int)b->nr != G * H) FAIL("rnn %d: bias length %zu, expected %d", l, b->nr, G * H);
        auto rowcol = [&](const_flappie_matrix Wm, int row, int k) -> float {
            const int u = row / 4, g = row % 4;
            if (u >= H || g >= G || k >= H) return 0.0f;
            return Wm->data.f[(size_t)(g * H + u) * Wm->stride + k];      // column g*H+u of the [H x G*H] matrix
        };
        std::vector<float> ip = pack_afrag(Hp / 4, Hp / 16, [&](int r, int k) { return rowcol(iW, r, k); });
        std::vector<float> sp = pack_afrag(Hp / 4, Hp / 16, [&](int r, int k) { return rowcol(sW, r, k); });
        std::vector<float> bias((size_t)4 * Hp, 0.0f);
        for (int u = 0; u < H; u++)
            for (int g = 0; g < G; g++) bias[(size_t)4 * u + g] = b->data.f[g * H + u];
        RnnDev &r = m->rnn[l];
        r.Kin16 = Hp / 16;
        r.iWp = (float4 *)dev_upload(m, ip.data(), ip.size() * 4);
        r.sWp = (float4 *)dev_upload(m, sp.data(), sp.size() * 4);
        r.bias = (float *)dev_upload(m, bias.data(), bias.size() * 4);
        if (!r.iWp || !r.sWp || !r.bias) { set_err(FFHIP_ENOMEM, "device allocation failed"); ffhip_model_free(m); return nullptr; }
        if (H == Hp && Hp % 128 == 0) {      // layer kernel shapes (H <= 384) and the split projection GEMM of the unfused path
            // [mat][ut][k/32][slice][lane][8] 16-bit slices of w * 2^sw (ffhip_split.hpp).  The layer's input arrives scaled by
            // 2^ex (the swish convolution's output by 2^kSplitExpX, everything bounded by 1 by 2^kSplitExpH), h by
            // 2^kSplitExpH; each matrix could take the exponent that brings its largest entry to [2^14, 2^15), and the two
            // products must carry the same total S: the matrix with headroom gives some of it up.
            const int Ut = Hp / 4, Hc = Hp / 32;
            const int ex = (l == 0 && m->act == ACT_SWISH) ? kSplitExpX : kSplitExpH, eh = kSplitExpH;
            float mx[2] = { 0.0f, 0.0f };
            for (int mat = 0; mat < 2; mat++)
                for (int row = 0; row < 4 * Hp; row++)
                    for (int k = 0; k < Hp; k++) mx[mat] = fmaxf(mx[mat], fabsf(rowcol(mat == 0 ? iW : sW, row, k)));
            const int S = (split_weight_exp(mx[0]) + ex < split_weight_exp(mx[1]) + eh) ? split_weight_exp(mx[0]) + ex : split_weight_exp(mx[1]) + eh;
            const int sw[2] = { S - ex, S - eh };
            r.split_S = kSplitF16 ? S : 0;
            // H = 256: the gate-major pack of the packed layer forms follows -- row tile G mb + gate of member mb (16 units; G = 3 gates for GRUmod,
            // 4 for the LSTM), row 4 q + c of a tile = unit 16 mb + 4 c + q (ffhip_rnn_split.hip, PACK)
            const bool gpack = ((G == 3 || G == 4) && Hp == 256);
            const size_t classic = (size_t)2 * Ut * Hc * kSplitNS * 64 * 8, Vt = G * (Hp / 16);
            std::vector<uint16_t> sp3(classic + (gpack ? (size_t)2 * Vt * Hc * kSplitNS * 64 * 8 : 0));
            if (gpack)
                for (int mat = 0; mat < 2; mat++)
                    for (int vt = 0; vt < (int)Vt; vt++)
                        for (int c = 0; c < Hc; c++)
                            for (int lane = 0; lane < 64; lane++)
                                for (int e = 0; e < 8; e++) {
                                    const int row = lane & 15, unit = 16 * (vt / G) + 4 * (row & 3) + (row >> 2);
                                    const float w = rowcol(mat == 0 ? iW : sW, 4 * unit + vt % G, c * 32 + (lane >> 4) * 8 + e);
                                    uint16_t sl[kSplitNS];
                                    split_host_slices(w, sw[mat], sl);
                                    const size_t base = classic + ((((size_t)mat * Vt + vt) * Hc + c) * kSplitNS) * 64 * 8 + (size_t)lane * 8 + e;
                                    for (int k = 0; k < kSplitNS; k++) sp3[base + (size_t)k * 64 * 8] = sl[k];
                                }
            for (int mat = 0; mat < 2; mat++)
                for (int ut = 0; ut < Ut; ut++)
                    for (int c = 0; c < Hc; c++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int e = 0; e < 8; e++) {
                                const float w = rowcol(mat == 0 ? iW : sW, ut * 16 + (lane & 15), c * 32 + (lane >> 4) * 8 + e);
                                uint16_t sl[kSplitNS];
                                split_host_slices(w, sw[mat], sl);
                                const size_t base = ((((size_t)mat * Ut + ut) * Hc + c) * kSplitNS) * 64 * 8 + (size_t)lane * 8 + e;
                                for (int k = 0; k < kSplitNS; k++) sp3[base + (size_t)k * 64 * 8] = sl[k];
                            }
            r.Wsplit = dev_upload(m, sp3.data(), sp3.size() * 2);
            if (!r.Wsplit) { set_err(FFHIP_ENOMEM, "device allocation failed"); ffhip_model_free(m); return nullptr; }
        }
    }

    // ---- output layer
    {
        const_flappie_matrix W = d->FF_W, b = d->FF_b;
        if ((int)W->nr != H) FAIL("FF_W has %zu rows, expected %d", W->nr, H);
        const int P = (int)W->nc;
        const int nbase = (int)roundf((-1.0f + sqrtf(1 + 2 * P)) / 2.0f);        // layers.c:1029-1032
        if (2 * nbase * (nbase + 1) != P || nbase < 1 || 2 * nbase > kMaxState || P > 64)
            FAIL("output layer has %d rows: not a flip-flop parameterisation this engine supports", P);
        if ((int)b->nr != P) FAIL("FF_b length");
        m->P = P; m->Ps = round_up(P, 4); m->nbase = nbase; m->nstate = 2 * nbase;
        const int Mt = (P + 15) / 16;
        if (Mt > 4) FAIL("output layer too wide");
        std::vector<float> wp = pack_afrag(Mt, Hp / 16, [&](int row, int k) -> float {
            if (row >= P || k >= H) return 0.0f;
            return W->data.f[(size_t)row * W->stride + k];
        });
        std::vector<float> bias((size_t)Mt * 16, 0.0f);
        for (int p = 0; p < P; p++) bias[p] = b->data.f[p];
        m->FFp = (float4 *)dev_upload(m, wp.data(), wp.size() * 4);
        m->FFb = (float *)dev_upload(m, bias.data(), bias.size() * 4);
        if (!m->FFp || !m->FFb) { set_err(FFHIP_ENOMEM, "device allocation failed"); ffhip_model_free(m); return nullptr; }
        if (kSplitF16 && Hp % 32 == 0 && split_supported(m->cell, Hp)) {
            // the same weights for the split-operand head: [mt][K chunk of 32][slice][lane][8 halves], row 16 mt + (lane & 15), k = 32 c + 8 (lane >> 4) + e
            const int Hc = Hp / 32;
            float mx = 0.0f;
            for (int p = 0; p < P; p++)
                for (int k = 0; k < H; k++) mx = fmaxf(mx, fabsf(W->data.f[(size_t)p * W->stride + k]));
            const int sw = split_weight_exp(mx);
            m->FF_split_S = sw + kSplitExpH;
            std::vector<uint16_t> sp((size_t)Mt * Hc * kSplitNS * 64 * 8);
            for (int mt = 0; mt < Mt; mt++)
                for (int c = 0; c < Hc; c++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int e = 0; e < 8; e++) {
                            const int row = mt * 16 + (lane & 15), k = c * 32 + (lane >> 4) * 8 + e;
                            const float w = (row < P && k < H) ? W->data.f[(size_t)row * W->stride + k] : 0.0f;
                            uint16_t sl[kSplitNS];
                            split_host_slices(w, sw, sl);
                            const size_t base = (((size_t)mt * Hc + c) * kSplitNS) * 64 * 8 + (size_t)lane * 8 + e;
                            for (int q = 0; q < kSplitNS; q++) sp[base + (size_t)q * 64 * 8] = sl[q];
                        }
            m->FFsplit = dev_upload(m, sp.data(), sp.size() * 2);
            if (!m->FFsplit) { set_err(FFHIP_ENOMEM, "device allocation failed"); ffhip_model_free(m); return nullptr; }
        }
    }
#undef FAIL
    return m;
}

extern "C" size_t ffhip_model_hidden(const ffhip_model *m) { return m ? (size_t)m->H : 0; }
extern "C" size_t ffhip_model_nparam(const ffhip_model *m) { return m ? (size_t)m->P : 0; }
extern "C" size_t ffhip_model_nbase(const ffhip_model *m) { return m ? (size_t)m->nbase : 0; }
// reads one FULL layer launch of this model takes on this device (the batch size that keeps every launch full): 1024 at 256 hidden units (the
// packed forms), 512 at 384, else 256 -- on 256 CUs (ffhip_rnn_split.hip, split_next_launch_tiles)
// (a device with fewer than 32 CUs -- none exists in this family -- still gets a unit of 1: the layer loop must advance)
extern "C" size_t ffhip_model_launch_reads(const ffhip_model *m) {
    if (!m) return 0;
    const int ncu = m->eng->prop.multiProcessorCount;
    if (m->H == m->Hp && split_supported(m->cell, m->Hp) && m->rnn[0].Wsplit != nullptr) return (size_t)16 * split_next_launch_tiles(m->cell, m->Hp, 1 << 20, ncu);
    return (size_t)16 * 2 * std::max(1, ncu / 32);
}
extern "C" size_t ffhip_model_nblock(const ffhip_model *m, size_t nsample) {
    if (!m) return 0;
    size_t n = nsample;
    for (int l = 0; l < m->nconv; l++) n = (n + m->conv[l].stride - 1) / m->conv[l].stride;
    return n;
}

// ------------------------------------------------------------------------------------ batch
struct ConvPlan { int Tin = 0, Tout = 0; int *x0a = nullptr, *x0b = nullptr; };

struct ffhip_batch {
    ffhip_engine *eng = nullptr;
    const ffhip_model *mdl = nullptr;
    hipStream_t stream = nullptr;
    int nread = 0, B16 = 0, Bp = 0;
    int T = 0, Tb = 0;                  // capacity: samples / blocks of the longest read the batch can take
    ConvPlan plan[3];
    // ragged batch (reads of different lengths, all <= T): per-read window tables, block counts and tile maxima
    bool ragged = false;
    std::vector<int> hT, hTb;           // samples / blocks of each read
    // host images of the tables below: they stay alive here so that their uploads can be asynchronous on the batch's stream (a
    // synchronous copy or a stream synchronisation in the set-up path waits for compute slots the other batch's layer launches hold)
    // (pinned: an asynchronous copy from pageable memory is staged by the runtime and may block the caller)
    struct Pinned {
        void *p = nullptr; size_t cap = 0;
        void *get(size_t bytes) {
            if (cap >= bytes && p) return p;
            if (p) hipHostFree(p);
            p = nullptr; cap = 0;
            if (hipHostMalloc(&p, bytes + bytes / 4 + 64, hipHostMallocDefault) != hipSuccess) return nullptr;
            cap = bytes + bytes / 4 + 64;
            return p;
        }
        ~Pinned() { if (p) hipHostFree(p); }
    };
    Pinned h_tin[3], h_ta[3], h_tq[3], h_tbs, h_tbt, h_glen, h_gsrc, h_rehearsal;
    int *d_tbs = nullptr, *d_tbt = nullptr;
    int *rag_x0a[3] = { nullptr, nullptr, nullptr }, *rag_x0b[3] = { nullptr, nullptr, nullptr };
    int *rag_tin[3] = { nullptr, nullptr, nullptr };       // stride-1 thin layers: per-read input lengths replace the table
    // Packed batch (ffhip_batch_set_*_packed): the `nread` rows of the buffers are SLOTS, each holding one or more reads one behind the other with a gap of
    // ffhip_model_pack_gap() blocks; results are indexed by READ (0 .. nvirt - 1, the order of the set call).  hT / hTb hold the reads' lengths then.
    int res_copied = 0;                 // the result block's copy to the host is already enqueued (a packed batch: behind its decode, in ffhip_batch_run)
    int cap_reads = 0;                  // the per-read result arrays (lens, score, logZ) take this many reads (>= nread)
    bool packed = false;
    int nvirt = 0;
    std::vector<int> v_slot, v_off;     // slot and first block of every read
    int *d_vb0 = nullptr, *d_vb1 = nullptr, *d_vtb = nullptr;      // per read: first row in Tb-row / (Tb + 1)-row buffers, blocks
    unsigned *d_live = nullptr;         // [Tb][B16]: bit r = slot 16 rt + r holds a block of a read at step t (the layer kernels' reset mask)
    int *rag_seg[3] = { nullptr, nullptr, nullptr };       // stride-1 thin layers of a packed batch: read boundaries per row (k_conv_small `seg`)
    size_t rag_seg_cap[3] = { 0, 0, 0 };
    long long *d_goff = nullptr;        // destination of every read's signal (floats from sbuf[0].p + kSamplePad)
    int4 *d_prd = nullptr;             // per read records for the device-side table / mask builders: [conv table | live mask] x cap_reads
    Pinned h_vb0, h_vb1, h_vtb, h_prd, h_seg[3], h_goff;
    SampleBuf sbuf[3];                  // sbuf[0] = signal, sbuf[l] = output of conv l-1
    float *act[2] = { nullptr, nullptr };
    void *actS[2] = { nullptr, nullptr };      // the same two buffers in the split-operand layout (ffhip_rnn_split.hip), allocated on first use
    float *keep[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    float *xa = nullptr, *cstate = nullptr;
    float *trans = nullptr, *post = nullptr, *fwd = nullptr;
    double *crf_logz = nullptr;         // fp64 partition function per read
    double *crf_e = nullptr;            // exp(score - block max), workspace of the linear-space partition function
    uint8_t *tb = nullptr;
    int *path = nullptr; float *qpath = nullptr; float *score = nullptr;
    char *bases = nullptr, *quals = nullptr; int *lens = nullptr;
    int32_t *trace = nullptr;
    unsigned split_epoch = 0;           // launch counter of the split layer kernel (its check-in words are never cleared)
    int counted = 0;                    // this batch is in the engine's in_flight count (between run and finish)
    const float **d_gsrc = nullptr; int *d_glen = nullptr;              // ffhip_batch_set_prepared: source rows of the gather
    unsigned *pflags = nullptr, *pabort = nullptr, *h_abort = nullptr;   // persistent-kernel XCC ids / abort word
    // reads with a value beyond the split format's range (ffhip_split.hpp: the swish convolutions' outputs are clamped at +-4094 there, the
    // reference's are unbounded, layers.c:24-33): one word per read, set by the producers of that format, looked at by ffhip_batch_finish,
    // which runs such reads again on the f32 path (`side`) and puts their results in place
    unsigned *sat = nullptr, *h_sat = nullptr;
    double rehearsal_done_at = 0.0;     // FFHIP_DEBUG_HOST_REHEARSAL_MSPS: when the emulated GPU is done with this batch
    ffhip_batch *side = nullptr;        // 16 slots of this batch's capacity, created when the first read needs it
    bool is_side = false;
    int reruns = 0;                     // reads of the last run that took that way
    int persist_concurrent_ok = 0;      // two such batches fit on the chip at once
    float *scratch = nullptr;           // dense [Tb][H] for debug taps
    // pinned host mirrors of the small results
    char *h_bases = nullptr, *h_quals = nullptr; int *h_lens = nullptr; float *h_score = nullptr;
    // what ffhip_batch_finish brings down is ONE block on the device and one pinned block on the host, [sat | abort | lens | score | bases | quals]: one copy
    // instead of six (round 5; a batch that was not decoded takes the first two parts only)
    unsigned char *res_dev = nullptr, *res_host = nullptr; size_t res_bytes = 0, res_head = 0;
    std::vector<void *> owned;
    unsigned last_flags = 0;
    float last_temperature = 1.0f;
    int ran = 0, finished = 0;
    int final_act = 0;                  // which act[] holds the last recurrent layer's output
    int rnn_path = 0;                   // what the last run used: 0 launch per step, 1 persistent recurrence behind a projection GEMM, 2 fused f32 layer kernel, 3 split-operand layer kernel, 4 split-operand projection GEMM + recurrence-only layer kernel
    hipEvent_t ev[FFHIP_NGROUP + 1];
    int have_ev = 0;
    int launches[FFHIP_NGROUP];
    hipEvent_t lev[5][3];
    hipEvent_t pair_ev = nullptr;       // ffhip_batch_run_pair: orders the two streams around the paired layer launches
    int run_cur = 0;                    // which of act[] / actS[] holds the current activations between the phases of a run
    unsigned run_flags = 0;
    int paired_last = 0;                // the last run's layers were one launch with another batch's
    ffhip_batch *prof_mate = nullptr;   // second batch of a profiled pair: the batch whose lev[][] events bracket the paired launches (two event records
                                        // per launch instead of six: each is a packet between two layer launches, ~5 us of idle chip)
    ffhip_batch *prof_ref = nullptr;    // ... and the first batch's way back, so that either can go first
    int pair_front = 0;                 // set by ffhip_batch_run_pair around the front phase: the layers to come are a paired (chip-filling) launch
    int profiled = 0;
};

static void *dalloc(ffhip_batch *b, size_t bytes, bool zero) {
    void *d = nullptr;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) { set_err(FFHIP_ENOMEM, "hipMalloc of %zu bytes failed", bytes); return nullptr; }
    // zero on the BATCH's stream: everything that touches the buffer afterwards is enqueued there.  (A hipMemset on the null stream
    // is not ordered against the batch's non-blocking stream and may complete after the call returns: an asynchronous upload into
    // a freshly allocated table could be overtaken by its own zero-fill -- seen as a wrong first ragged batch of a batch object.)
    if (zero) {
        const hipError_t e = b->stream ? hipMemsetAsync(d, 0, bytes, b->stream) : hipMemset(d, 0, bytes);
        if (e != hipSuccess || (!b->stream && hipDeviceSynchronize() != hipSuccess)) { hipFree(d); set_err(FFHIP_EHIP, "hipMemset failed"); return nullptr; }
    }
    b->owned.push_back(d);
    return d;
}

// Column -> window-start table of one convolution: the reference's three regions (layers.c:220-271)
// restated in index space.  With zero pads either side of the input a
// partial edge window is a full window starting at x0 (possibly negative).
int ffhip::build_conv_plan(int T, int winlen, int s, std::vector<int> &a, std::vector<int> &bq) {
    const int padL = (winlen - 1) / 2, padR = winlen / 2;
    const int Tout = (T + s - 1) / s;
    const int ncolsL = (padL + s - 1) / s, shiftX = ncolsL * s - padL;
    const int nstepC = (winlen + s - 1) / s, nstepX = s * nstepC;
    if (T < winlen || T - shiftX - (winlen - 1) < 0) return -1;
    a.assign(Tout, kNoWindow);
    bq.assign(Tout, kNoWindow);
    bool overflow = false;
    auto add = [&](int c, int x0) {
        if (c < 0 || c >= Tout) return;
        if (a[c] == kNoWindow) a[c] = x0;
        else if (bq[c] == kNoWindow) bq[c] = x0;
        else overflow = true;
    };
    for (int w = 0; w < padL; w += s) add(w / s, w - padL);
    for (int w = 0; w < winlen; w += s) {
        const int ncol = (T - shiftX - w) / nstepX, col0 = ncolsL + w / s;
        for (int k = 0; k < ncol; k++) add(col0 + nstepC * k, shiftX + w + nstepX * k);
    }
    const int maxCol = (T - shiftX) / nstepX, rem = (T - shiftX) % nstepX;
    const int colR = ncolsL + nstepC * (maxCol - 1) + rem / s + 1;
    const int startR = s - (padL + T - winlen) % s - 1;
    for (int w = startR; w < padR; w += s) add(colR + w / s, T - winlen + 1 + w);
    return overflow ? -2 : Tout;
}

// a profiled pair's links (ffhip_batch::prof_mate / prof_ref), taken apart from either end
static void prof_unlink(ffhip_batch *b) {
    if (b->prof_ref && b->prof_ref->prof_mate == b) b->prof_ref->prof_mate = nullptr;
    if (b->prof_mate && b->prof_mate->prof_ref == b) b->prof_mate->prof_ref = nullptr;
    b->prof_ref = b->prof_mate = nullptr;
}

extern "C" void ffhip_batch_destroy(ffhip_batch *b) {
    if (!b) return;
    hipSetDevice(b->eng->device);
    hipStreamSynchronize(b->stream);
    if (b->counted) { b->counted = 0; b->eng->in_flight--; }
    for (void *p : b->owned) hipFree(p);
    if (b->res_host) hipHostFree(b->res_host);      // (h_sat, h_abort, h_lens, h_score, h_bases, h_quals point into it)
    if (b->side) ffhip_batch_destroy(b->side);
    prof_unlink(b);
    if (b->have_ev) {
        for (int i = 0; i <= FFHIP_NGROUP; i++) hipEventDestroy(b->ev[i]);
        for (int l = 0; l < 5; l++) for (int i = 0; i < 3; i++) hipEventDestroy(b->lev[l][i]);
        if (b->pair_ev) hipEventDestroy(b->pair_ev);
    }
    delete b;
}

static ffhip_batch *batch_create_impl(ffhip_engine *eng, const ffhip_model *m, int nread, size_t nsample, int cap_reads) {
    if (!eng || !m || nread <= 0 || nsample == 0 || nsample > (1u << 30) || cap_reads < 0 || cap_reads > (1 << 22)) { set_err(FFHIP_EINVAL, "bad batch arguments"); return nullptr; }
    hipSetDevice(eng->device);
    ffhip_batch *b = new ffhip_batch();
    b->eng = eng; b->mdl = m;
    b->cap_reads = cap_reads > nread ? cap_reads : nread;
    b->stream = eng->streams[eng->next_stream];
    eng->next_stream = (eng->next_stream + 1) % eng->nstreams;
    b->nread = nread; b->B16 = (nread + 15) / 16; b->Bp = b->B16 * 16;
    b->T = (int)nsample;
#define BFAIL() do { ffhip_batch_destroy(b); return nullptr; } while (0)
    int Tin = b->T;
    for (int l = 0; l < m->nconv; l++) {
        std::vector<int> a, bq;
        const int Tout = build_conv_plan(Tin, m->conv[l].winlen, m->conv[l].stride, a, bq);
        if (Tout < 0) { set_err(FFHIP_EINVAL, "conv %d: %d samples is outside the domain of the reference's convolution (winlen %d, stride %d)", l, Tin, m->conv[l].winlen, m->conv[l].stride); BFAIL(); }
        b->plan[l].Tin = Tin; b->plan[l].Tout = Tout;
        b->plan[l].x0a = (int *)dalloc(b, (size_t)Tout * 4, false);
        b->plan[l].x0b = (int *)dalloc(b, (size_t)Tout * 4, false);
        if (!b->plan[l].x0a || !b->plan[l].x0b) BFAIL();
        if (hipMemcpy(b->plan[l].x0a, a.data(), (size_t)Tout * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(b->plan[l].x0b, bq.data(), (size_t)Tout * 4, hipMemcpyHostToDevice) != hipSuccess) {
            set_err(FFHIP_EHIP, "upload of the convolution window table failed");
            BFAIL();
        }
        // input buffer of this conv
        SampleBuf &sb = b->sbuf[l];
        sb.F = m->conv[l].Fin; sb.T = Tin;
        sb.rs = (size_t)round_up((Tin + 2 * kSamplePad) * sb.F, 64);
        sb.p = (float *)dalloc(b, (size_t)b->Bp * sb.rs * 4, true);
        if (!sb.p) BFAIL();
        Tin = Tout;
    }
    b->Tb = Tin;
    b->hT.assign(nread, b->T);
    b->hTb.assign(nread, b->Tb);
    const size_t Tb = b->Tb, Bp = b->Bp, Hp = m->Hp, Ps = m->Ps, ns = m->nstate;
    // b->act[0 / 1] (fp32 activations, Tb * Bp * Hp * 4 bytes each: 10.5 GB for 256 reads of 100 000 samples at H = 512) are allocated by the
    // first run that needs them: the default path writes the last convolution and the layers in the split layout and only the last layer's fp32
    // copy (act[1], what the head reads)
    // b->xa (gate pre-activations, 4x the size of an activation buffer) exists only on the unfused path: allocated on first use
    if (!(b->cstate = (float *)dalloc(b, Bp * Hp * 4, true))) BFAIL();
    if (!(b->trans = (float *)dalloc(b, (size_t)nread * Tb * Ps * 4, true))) BFAIL();
    const size_t nres = (size_t)b->cap_reads;      // per-READ results (a packed batch holds more reads than rows)
    if (!(b->crf_logz = (double *)dalloc(b, nres * sizeof(double), true))) BFAIL();
    if (!(b->crf_e = (double *)dalloc(b, (size_t)nread * Tb * crf_exp_stride(m->P) * sizeof(double), false))) BFAIL();
    if (!(b->post = (float *)dalloc(b, (size_t)nread * Tb * Ps * 4, true))) BFAIL();
    if (!(b->fwd = (float *)dalloc(b, (size_t)2 * nread * (Tb + 1) * kFwdRowBytes, false))) BFAIL();      // forward + backward vectors
    if (!(b->tb = (uint8_t *)dalloc(b, (size_t)nread * Tb * kMaxState, false))) BFAIL();
    if (!(b->path = (int *)dalloc(b, (size_t)nread * (Tb + 1) * 4, true))) BFAIL();
    if (!(b->qpath = (float *)dalloc(b, (size_t)nread * (Tb + 1) * 4, true))) BFAIL();
    {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_sat = 0, o_abort = up((size_t)b->Bp * 4), o_lens = o_abort + 256, o_score = o_lens + up(nres * 4),
                     o_bases = o_score + up(nres * 4), o_quals = o_bases + up((size_t)nread * (Tb + 1));
        b->res_head = o_lens;
        b->res_bytes = o_quals + up((size_t)nread * (Tb + 1));
        if (!(b->res_dev = (unsigned char *)dalloc(b, b->res_bytes, true))) BFAIL();
        if (hipHostMalloc((void **)&b->res_host, b->res_bytes) != hipSuccess) { set_err(FFHIP_ENOMEM, "pinned host allocation failed"); BFAIL(); }
        memset(b->res_host, 0, b->res_bytes);
        b->sat = (unsigned *)(b->res_dev + o_sat); b->pabort = (unsigned *)(b->res_dev + o_abort); b->lens = (int *)(b->res_dev + o_lens);
        b->score = (float *)(b->res_dev + o_score); b->bases = (char *)(b->res_dev + o_bases); b->quals = (char *)(b->res_dev + o_quals);
        b->h_sat = (unsigned *)(b->res_host + o_sat); b->h_abort = (unsigned *)(b->res_host + o_abort); b->h_lens = (int *)(b->res_host + o_lens);
        b->h_score = (float *)(b->res_host + o_score); b->h_bases = (char *)(b->res_host + o_bases); b->h_quals = (char *)(b->res_host + o_quals);
    }
    if (!(b->trace = (int32_t *)dalloc(b, (size_t)nread * (Tb + 1) * ns * 4, true))) BFAIL();
    if (!(b->pflags = (unsigned *)dalloc(b, persist_flag_words((int)Hp, b->B16) * sizeof(unsigned), true))) BFAIL();
    // (pabort -- [0] abort word, [1] development counter -- and sat live in the result block above)
    if (persist_supported(m->cell, (int)Hp, eng->prop.multiProcessorCount)) {
        const int maxt = persist_max_tiles(m->cell, (int)Hp, eng->prop.multiProcessorCount, fused_supported(m->cell, (int)Hp));
        b->persist_concurrent_ok = 2 * b->B16 <= maxt;      // two such launches fit on the chip together
    }
    for (int i = 0; i <= FFHIP_NGROUP; i++)
        if (hipEventCreate(&b->ev[i]) != hipSuccess) { set_err(FFHIP_EHIP, "hipEventCreate failed"); BFAIL(); }
    for (int l = 0; l < 5; l++)
        for (int i = 0; i < 3; i++)
            if (hipEventCreate(&b->lev[l][i]) != hipSuccess) { set_err(FFHIP_EHIP, "hipEventCreate failed"); BFAIL(); }
    if (hipEventCreateWithFlags(&b->pair_ev, hipEventDisableTiming) != hipSuccess) { set_err(FFHIP_EHIP, "hipEventCreate failed"); BFAIL(); }
    b->have_ev = 1;
#undef BFAIL
    return b;
}
extern "C" ffhip_batch *ffhip_batch_create(ffhip_engine *eng, const ffhip_model *m, int nread, size_t nsample) { return batch_create_impl(eng, m, nread, nsample, 0); }
// a batch of `nslot` rows of `nsample` samples that takes up to max_reads reads, several to a row (ffhip_batch_set_prepared_packed / _signals_packed)
extern "C" ffhip_batch *ffhip_batch_create_packed(ffhip_engine *eng, const ffhip_model *m, int nslot, size_t nsample, int max_reads) {
    return batch_create_impl(eng, m, nslot, nsample, max_reads);
}
static inline int batch_nreads(const ffhip_batch *b) { return b->packed ? b->nvirt : b->nread; }
extern "C" int ffhip_batch_nreads(const ffhip_batch *b) { return b ? batch_nreads(b) : 0; }
// first row of a read in the buffers of Tb / Tb + 1 rows a slot
static inline size_t read_row0(const ffhip_batch *b, int read) { return b->packed ? (size_t)b->v_slot[read] * b->Tb + b->v_off[read] : (size_t)read * b->Tb; }
static inline size_t read_row1(const ffhip_batch *b, int read) { return b->packed ? (size_t)b->v_slot[read] * (b->Tb + 1) + b->v_off[read] : (size_t)read * (b->Tb + 1); }
static int total_stride(const ffhip_model *m) { int st = 1; for (int l = 0; l < m->nconv; l++) st *= m->conv[l].stride; return st; }
// does this model's default path take packed batches?  (what batch_run_impl asks of a packed batch, on the model's side)
extern "C" int ffhip_model_packable(const ffhip_model *m) {
    if (!m || m->kind == FFHIP_NET_LSTM5_RLE || !((m->nbase == 4 && m->Ps == 40) || (m->nbase == 5 && m->Ps == 60))) return 0;
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    return (persist_supported(m->cell, m->Hp, ncu) && split_supported(m->cell, m->Hp) && m->rnn[0].Wsplit != nullptr && m->conv[m->nconv - 1].Mpad == m->Hp &&
            !dbg("no_split") && !dbg("no_fuse")) ? 1 : 0;
}
// Blocks that stay free behind every read of a packed slot: the widest convolution window of the model either side of a read must see the zero padding the
// reference gives it (layers.c:216-271), in every layer's coordinates, and the dead block behind a read is the next one's zero recurrent state.
extern "C" size_t ffhip_model_pack_gap(const ffhip_model *m) {
    if (!m) return 0;
    int wmax = 1;
    for (int l = 0; l < m->nconv; l++) wmax = std::max(wmax, m->conv[l].winlen);
    const int st = total_stride(m);
    return (size_t)((2 * wmax + st - 1) / st + 1);
}
// Rows (a multiple of 16, at most want_rows) of a packed batch of `nsample`-sample rows whose workspace takes at most 72 % / nobjects of the device's memory (512 rows of 228 352 samples at 384 hidden units: 103 GB of 288):
// `nobjects` such objects are alive in a pipeline (two: one runs, one is set up; one: the caller collects a batch before it sets up the next) beside the prepared signals.  What a row costs is what batch_create_impl and the default path of
// batch_run_impl allocate per block and per sample.
extern "C" int ffhip_pack_rows_for(const ffhip_model *m, int want_rows, size_t nsample, int nobjects) {
    if (!m || want_rows <= 0 || nsample == 0 || nobjects < 1) return 0;
    const size_t nb = ffhip_model_nblock(m, nsample) + 1;
    size_t per_block = (size_t)m->Ps * 8 + (size_t)crf_exp_stride(m->P) * 8 + 2 * kFwdRowBytes + kMaxState + 4 + 4 + 2 + (size_t)m->nstate * 4 + 8      // scores, E, chains, traceback, path, strings, trace, tables
                       + 2 * (size_t)m->Hp * 2 * kSplitNS;                                                                                            // two activation buffers in the split layout
    size_t per_sample = 0;
    for (int l = 0; l < m->nconv; l++) per_sample += (size_t)m->conv[l].Fin * 4;       // the convolutions' inputs (sample-major; the last one's as fp16 slices: the same 4 bytes a value)
    const double row = (double)nb * (double)per_block + (double)(nsample + 2 * kSamplePad) * (double)per_sample;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return want_rows;
    long rows = (long)((double)total_b * 0.72 / (double)nobjects / row);      // (of the device's TOTAL memory: the answer must not change while the first object is alive)
    rows = rows / 16 * 16;
    if (rows > want_rows) rows = want_rows;
    return rows < 16 ? 16 : (int)rows;
}
extern "C" int ffhip_pack_rows(const ffhip_model *m, int want_rows, size_t nsample) { return ffhip_pack_rows_for(m, want_rows, nsample, 2); }
// Plan of `nread` reads into nslot rows of nsample_cap samples: slot[] / block_off[] of every read (slot -1: it did not fit); returns the number placed.
// Longest read first, each into the row that holds LEAST so far (round 6, third session; rounds before: into the first row it fits).  A launch runs as long as its longest
// row, whatever the others hold: first fit fills row after row to the brim -- the longest row IS the capacity, and the slack the caller plans with (5 %) is paid as empty
// steps of every layer -- while the least-loaded rule ends with all rows within a short read of the mean (the classic longest-processing-time bound), so the launch
// lasts total / rows.  A read that does not fit the emptiest row fits none.  FFHIP_DEBUG=pack_first_fit: the old rule.
extern "C" int ffhip_pack_plan(const ffhip_model *m, int nslot, size_t nsample_cap, int nread, const size_t *nsample, int *slot, int *block_off) {
    if (!m || nslot <= 0 || nread < 0 || !nsample || !slot || !block_off) { set_err(FFHIP_EINVAL, "bad pack plan arguments"); return -1; }
    const long cap = (long)ffhip_model_nblock(m, nsample_cap), gap = (long)ffhip_model_pack_gap(m);
    std::vector<int> order(nread);
    for (int i = 0; i < nread; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return nsample[x] > nsample[y]; });
    std::vector<long> used(nslot, 0);
    int placed = 0;
    if (dbg("pack_first_fit")) {
        // rows by free blocks: the row count is small (<= 1024) and reads come longest first, so a scan from a moving start is short
        int first_open = 0;
        for (int k = 0; k < nread; k++) {
            const int i = order[k];
            slot[i] = -1; block_off[i] = 0;
            if (nsample[i] == 0 || nsample[i] > nsample_cap) continue;
            const long nb = (long)ffhip_model_nblock(m, nsample[i]);
            for (int r = first_open; r < nslot; r++) {
                if (used[r] + nb + 1 <= cap) {      // (a dead block behind the last read of a row too)
                    slot[i] = r; block_off[i] = (int)used[r];
                    used[r] += nb + gap;
                    placed++;
                    break;
                }
            }
            while (first_open < nslot && used[first_open] + gap + 8 >= cap) first_open++;      // (rows with no room for even a short read)
        }
        return placed;
    }
    // a min-heap of (blocks in use, row): ties go to the lower row, so a plan is a function of its arguments
    typedef std::pair<long, int> Row;
    std::priority_queue<Row, std::vector<Row>, std::greater<Row>> rows;
    for (int r = 0; r < nslot; r++) rows.push(Row(0, r));
    for (int k = 0; k < nread; k++) {
        const int i = order[k];
        slot[i] = -1; block_off[i] = 0;
        if (nsample[i] == 0 || nsample[i] > nsample_cap) continue;
        const long nb = (long)ffhip_model_nblock(m, nsample[i]);
        const Row top = rows.top();
        if (top.first + nb + 1 > cap) continue;      // (a dead block behind the last read of a row too) -- not in the emptiest row: in none
        rows.pop();
        slot[i] = top.second; block_off[i] = (int)top.first;
        rows.push(Row(top.first + nb + gap, top.second));
        placed++;
    }
    return placed;
}

extern "C" size_t ffhip_batch_nblock(const ffhip_batch *b) { return b ? (size_t)b->Tb : 0; }
extern "C" size_t ffhip_batch_read_nblock(const ffhip_batch *b, int read) {
    return (b && read >= 0 && read < batch_nreads(b)) ? (size_t)b->hTb[read] : 0;
}

// Records the reads' lengths.  All equal to the capacity: the uniform fast path (shared window tables, no
// masks).  Otherwise the batch becomes ragged: every read gets its own column -> window-start rows (its right
// edge is where the reference's quirk lives), its block count, and every read tile its maximum.
static int apply_lengths(ffhip_batch *b, const std::vector<int> &lens) {
    const ffhip_model *m = b->mdl;
    bool uniform = true;
    for (int r = 0; r < b->nread; r++) {
        if (lens[r] < 0 || lens[r] > b->T) return set_err(FFHIP_EINVAL, "read %d: %d samples, the batch takes up to %d", r, lens[r], b->T);
        uniform = uniform && lens[r] == b->T;          // 0 = an empty slot: no work, no results
    }
    b->hT = lens;
    b->packed = false; b->nvirt = 0;
    if (uniform) { b->ragged = false; b->hTb.assign(b->nread, b->Tb); return FFHIP_OK; }
    std::vector<int> cur(lens);
    for (int l = 0; l < m->nconv; l++) {
        const int Tmax = b->plan[l].Tout;
        if (m->conv[l].stride == 1 && l < m->nconv - 1) {
            // stride 1: column c's window starts at c - padL for every length; the kernel only needs the lengths
            if (!b->rag_tin[l] && !(b->rag_tin[l] = (int *)dalloc(b, (size_t)b->Bp * 4, true))) return FFHIP_ENOMEM;
            int *tin = (int *)b->h_tin[l].get((size_t)b->Bp * 4);
            if (!tin) return set_err(FFHIP_ENOMEM, "pinned host allocation failed");
            memset(tin, 0, (size_t)b->Bp * 4);
            for (int r = 0; r < b->nread; r++) {
                if (cur[r] != 0 && cur[r] < m->conv[l].winlen)
                    return set_err(FFHIP_EINVAL, "read %d: %d samples at convolution %d is outside the domain of the reference's convolution (winlen %d)", r, cur[r], l, m->conv[l].winlen);
                tin[r] = cur[r];
            }
            HIP_TRY(hipMemcpyAsync(b->rag_tin[l], tin, (size_t)b->Bp * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
            continue;                                                  // output length = input length
        }
        const size_t n = (size_t)b->Bp * Tmax;
        if (!b->rag_x0a[l]) {
            b->rag_x0a[l] = (int *)dalloc(b, n * 4, false);
            b->rag_x0b[l] = (int *)dalloc(b, n * 4, false);
            if (!b->rag_x0a[l] || !b->rag_x0b[l]) return FFHIP_ENOMEM;
        }
        int *ta = (int *)b->h_ta[l].get(n * 4), *tq = (int *)b->h_tq[l].get(n * 4);
        if (!ta || !tq) return set_err(FFHIP_ENOMEM, "pinned host allocation failed");
        std::fill(ta, ta + n, kZeroCol); std::fill(tq, tq + n, kZeroCol);
        std::map<int, std::pair<std::vector<int>, std::vector<int>>> cache;
        for (int r = 0; r < b->nread; r++) {
            if (cur[r] == 0) continue;                                 // empty slot: its row stays all kZeroCol
            auto it = cache.find(cur[r]);
            if (it == cache.end()) {
                std::vector<int> a, bq;
                if (build_conv_plan(cur[r], m->conv[l].winlen, m->conv[l].stride, a, bq) < 0)
                    return set_err(FFHIP_EINVAL, "read %d: %d samples at convolution %d is outside the domain of the reference's convolution (winlen %d)", r, cur[r], l, m->conv[l].winlen);
                it = cache.emplace(cur[r], std::make_pair(std::move(a), std::move(bq))).first;
            }
            const std::vector<int> &a = it->second.first, &bq = it->second.second;
            memcpy(ta + (size_t)r * Tmax, a.data(), a.size() * 4);
            memcpy(tq + (size_t)r * Tmax, bq.data(), bq.size() * 4);
            cur[r] = (int)a.size();
        }
        HIP_TRY(hipMemcpyAsync(b->rag_x0a[l], ta, n * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
        HIP_TRY(hipMemcpyAsync(b->rag_x0b[l], tq, n * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    }
    b->hTb = cur;
    if (!b->d_tbs) {
        b->d_tbs = (int *)dalloc(b, (size_t)b->Bp * 4, true);
        b->d_tbt = (int *)dalloc(b, (size_t)b->B16 * 4, true);
        if (!b->d_tbs || !b->d_tbt) return FFHIP_ENOMEM;
    }
    int *tbs = (int *)b->h_tbs.get((size_t)b->Bp * 4), *tbt = (int *)b->h_tbt.get((size_t)b->B16 * 4);
    if (!tbs || !tbt) return set_err(FFHIP_ENOMEM, "pinned host allocation failed");
    memset(tbs, 0, (size_t)b->Bp * 4); memset(tbt, 0, (size_t)b->B16 * 4);
    for (int r = 0; r < b->nread; r++) { tbs[r] = cur[r]; tbt[r / 16] = std::max(tbt[r / 16], cur[r]); }
    HIP_TRY(hipMemcpyAsync(b->d_tbs, tbs, (size_t)b->Bp * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(b->d_tbt, tbt, (size_t)b->B16 * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    b->ragged = true;
    return FFHIP_OK;
}

// Packed batch: read v (lens[v] samples) stands in row slot[v] from block off[v] on.  Builds what makes every kernel see each read as if it were alone:
//   * the strided convolution's column -> window-start tables per ROW, every read's own table (its right edge is where the reference's quirk lives, layers.c:257-271)
//     shifted to the read's place, kZeroCol between the reads;
//   * the stride-1 thin layers' read boundaries per row (k_conv_small `seg`): columns outside a read are written as zeros -- the next layer's padding;
//   * the layer kernels' live mask (bit per step and row) and the rows' extents (tbs / tbt: how many steps a read tile runs);
//   * per read: its first row in the buffers of Tb and of Tb + 1 rows a slot, and its blocks.
static int apply_packed(ffhip_batch *b, int nv, const std::vector<int> &lens, const int *slot, const int *off) {
    const ffhip_model *m = b->mdl;
    if (nv < 0 || nv > b->cap_reads) return set_err(FFHIP_EINVAL, "%d reads, the batch was created for %d", nv, b->cap_reads);
    const int nconv = m->nconv, gap = (int)ffhip_model_pack_gap(m);
    std::vector<int> sfrom(nconv + 1, 1);                       // sfrom[l]: samples of layer l's input per block
    for (int l = nconv - 1; l >= 0; l--) sfrom[l] = sfrom[l + 1] * m->conv[l].stride;
    std::vector<int> vtb(nv, 0);
    std::vector<std::vector<int>> rows(b->nread);              // reads of every row
    for (int v = 0; v < nv; v++) {
        if (slot[v] < 0 || slot[v] >= b->nread || off[v] < 0 || lens[v] <= 0 || lens[v] > b->T) return set_err(FFHIP_EINVAL, "packed read %d: bad slot, offset or length", v);
        vtb[v] = (int)ffhip_model_nblock(m, (size_t)lens[v]);
        if (off[v] + vtb[v] + 1 > b->Tb) return set_err(FFHIP_EINVAL, "packed read %d: blocks %d .. %d do not fit a row of %d (one free block behind the last read)", v, off[v], off[v] + vtb[v], b->Tb);
        rows[slot[v]].push_back(v);
    }
    for (auto &rv : rows) {
        std::sort(rv.begin(), rv.end(), [&](int x, int y) { return off[x] < off[y]; });
        for (size_t k = 1; k < rv.size(); k++)
            if (off[rv[k]] < off[rv[k - 1]] + vtb[rv[k - 1]] + gap) return set_err(FFHIP_EINVAL, "packed reads %d and %d of row %d are closer than %d blocks", rv[k - 1], rv[k], slot[rv[k]], gap);
    }
    // The set-up of a packed batch is hundreds of megabytes of fills (tables, signal rows) and a few small kernels on the batch's stream.  Beside another batch's
    // layer launch -- which holds every CU -- a fill crawls AND keeps the next layer launch from becoming resident (kernel trace of a mixed directory: a 0.4 GB fill
    // 170 ms long, the layer launch beside it 300 ms instead of 180): it goes behind the engine's last layer launch, beside that batch's head and decode.
    if (b->eng->persist_chained) HIP_TRY(hipStreamWaitEvent(b->stream, b->eng->persist_done, 0), FFHIP_EHIP);
    // ... and behind the CRF head of the packed batch that ran last (round 6, third session): the head is a throughput kernel (12 ms alone for a 100 M-sample batch) that k_conv_split_ws
    // beside it stretches to 47 ms, and the chains wait for it -- with the head first they run under the convolutions instead of behind them (profiles/r06_pack_trace.txt).
    // FFHIP_DEBUG=no_pack_behind_head: both start with the gap.
    if (b->eng->head_done_rec && !dbg("no_pack_behind_head")) HIP_TRY(hipStreamWaitEvent(b->stream, b->eng->head_done, 0), FFHIP_EHIP);
    std::vector<int> cur(lens);
    int nstrided = 0;
    for (int l = 0; l < nconv; l++) {
        const int Tmax = b->plan[l].Tout;
        if (m->conv[l].stride == 1 && l < nconv - 1) {
            // offsets [Bp + 1], then the boundaries (columns of this layer = samples of its input)
            const size_t n = (size_t)b->Bp + 1 + 2 * (size_t)nv;
            if (b->rag_seg_cap[l] < n) {
                b->rag_seg_cap[l] = n + n / 2 + 64;
                if (!(b->rag_seg[l] = (int *)dalloc(b, b->rag_seg_cap[l] * 4, false))) return FFHIP_ENOMEM;      // (an outgrown table stays owned until the batch goes)
            }
            int *sg = (int *)b->h_seg[l].get(n * 4);
            if (!sg) return set_err(FFHIP_ENOMEM, "pinned host allocation failed");
            int at = b->Bp + 1;
            for (int r = 0; r < b->Bp; r++) {
                sg[r] = at;
                if (r < b->nread)
                    for (int v : rows[r]) {
                        if (cur[v] < m->conv[l].winlen) return set_err(FFHIP_EINVAL, "read %d: %d samples at convolution %d is outside the domain of the reference's convolution (winlen %d)", v, cur[v], l, m->conv[l].winlen);
                        sg[at++] = off[v] * sfrom[l]; sg[at++] = off[v] * sfrom[l] + cur[v];
                    }
            }
            sg[b->Bp] = at;
            HIP_TRY(hipMemcpyAsync(b->rag_seg[l], sg, n * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
            continue;                                           // output length = input length
        }
        const size_t n = (size_t)b->Bp * Tmax;
        if (!b->rag_x0a[l]) {
            b->rag_x0a[l] = (int *)dalloc(b, n * 4, false);
            b->rag_x0b[l] = (int *)dalloc(b, n * 4, false);
            if (!b->rag_x0a[l] || !b->rag_x0b[l]) return FFHIP_ENOMEM;
        }
        // the table is built on the device (k_pack_conv_table: build_conv_plan column by column), kZeroCol wherever no read stands
        if (!b->d_prd && !(b->d_prd = (int4 *)dalloc(b, (size_t)2 * b->cap_reads * sizeof(int4), false))) return FFHIP_ENOMEM;
        if (nstrided++ > 0) HIP_TRY(hipStreamSynchronize(b->stream), FFHIP_EHIP);      // (a second strided layer: the records' host image is still being copied)
        int4 *prd = (int4 *)b->h_prd.get((size_t)2 * b->cap_reads * sizeof(int4));
        if (!prd) return set_err(FFHIP_ENOMEM, "pinned host allocation failed");
        const int winlen = m->conv[l].winlen, st = m->conv[l].stride, padL = (winlen - 1) / 2, ncolsL = (padL + st - 1) / st, shiftX = ncolsL * st - padL;
        int maxcols = 0;
        for (int v = 0; v < nv; v++) {
            const int T = cur[v], Tout = (T + st - 1) / st;
            if (T < winlen || T - shiftX - (winlen - 1) < 0)      // (build_conv_plan's domain check)
                return set_err(FFHIP_EINVAL, "read %d: %d samples at convolution %d is outside the domain of the reference's convolution (winlen %d)", v, T, l, winlen);
            if (off[v] * sfrom[l + 1] + Tout > Tmax) return set_err(FFHIP_EINVAL, "packed read %d: convolution %d runs past the row", v, l);
            prd[v] = make_int4(slot[v], off[v] * sfrom[l + 1], off[v] * sfrom[l], T);
            maxcols = std::max(maxcols, Tout);
            cur[v] = Tout;
        }
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)b->rag_x0a[l], kZeroCol, n, b->stream), FFHIP_EHIP);
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)b->rag_x0b[l], kZeroCol, n, b->stream), FFHIP_EHIP);
        HIP_TRY(hipMemsetAsync(b->pabort + 2, 0, sizeof(unsigned), b->stream), FFHIP_EHIP);
        if (nv > 0) {
            HIP_TRY(hipMemcpyAsync(b->d_prd, prd, (size_t)nv * sizeof(int4), hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
            launch_pack_conv_table(b->stream, b->d_prd, nv, maxcols, winlen, st, Tmax, b->rag_x0a[l], b->rag_x0b[l], b->pabort + 2);
        }
    }
    for (int v = 0; v < nv; v++) if (cur[v] != vtb[v]) return set_err(FFHIP_EINVAL, "internal error: block count of packed read %d", v);
    if (!b->d_tbs) {
        b->d_tbs = (int *)dalloc(b, (size_t)b->Bp * 4, true);
        b->d_tbt = (int *)dalloc(b, (size_t)b->B16 * 4, true);
        if (!b->d_tbs || !b->d_tbt) return FFHIP_ENOMEM;
    }
    if (!b->d_live) {
        b->d_live = (unsigned *)dalloc(b, (size_t)b->Tb * b->B16 * 4, false);
        if (!b->d_prd && !(b->d_prd = (int4 *)dalloc(b, (size_t)2 * b->cap_reads * sizeof(int4), false))) return FFHIP_ENOMEM;
        b->d_vb0 = (int *)dalloc(b, (size_t)b->cap_reads * 4, false);
        b->d_vb1 = (int *)dalloc(b, (size_t)b->cap_reads * 4, false);
        b->d_vtb = (int *)dalloc(b, (size_t)b->cap_reads * 4, false);
        if (!b->d_live || !b->d_vb0 || !b->d_vb1 || !b->d_vtb) return FFHIP_ENOMEM;
    }
    int *tbs = (int *)b->h_tbs.get((size_t)b->Bp * 4), *tbt = (int *)b->h_tbt.get((size_t)b->B16 * 4);
    int4 *prl = (int4 *)b->h_prd.get((size_t)2 * b->cap_reads * sizeof(int4));
    if (prl) prl += b->cap_reads;                              // (second half: the first may still be waiting for its copy)
    int *vb0 = (int *)b->h_vb0.get((size_t)std::max(1, nv) * 4), *vb1 = (int *)b->h_vb1.get((size_t)std::max(1, nv) * 4), *vt = (int *)b->h_vtb.get((size_t)std::max(1, nv) * 4);
    if (!tbs || !tbt || !prl || !vb0 || !vb1 || !vt) return set_err(FFHIP_ENOMEM, "pinned host allocation failed");
    memset(tbs, 0, (size_t)b->Bp * 4); memset(tbt, 0, (size_t)b->B16 * 4);
    if ((size_t)b->nread * (size_t)(b->Tb + 1) >= ((size_t)1 << 31)) return set_err(FFHIP_EINVAL, "packed batch: rows x blocks beyond 2^31");
    int maxb = 0;
    for (int v = 0; v < nv; v++) {
        const int r = slot[v], rt = r / 16;
        prl[v] = make_int4(r, off[v], vtb[v], 0);
        maxb = std::max(maxb, vtb[v]);
        tbs[r] = std::max(tbs[r], off[v] + vtb[v] + 1);         // (the dead block behind the row's last read is a step of the layer: the reverse layers start from it)
        tbt[rt] = std::max(tbt[rt], tbs[r]);
        vb0[v] = r * b->Tb + off[v]; vb1[v] = r * (b->Tb + 1) + off[v]; vt[v] = vtb[v];
    }
    HIP_TRY(hipMemcpyAsync(b->d_tbs, tbs, (size_t)b->Bp * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(b->d_tbt, tbt, (size_t)b->B16 * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    HIP_TRY(hipMemsetAsync(b->d_live, 0, (size_t)b->Tb * b->B16 * 4, b->stream), FFHIP_EHIP);
    if (nv > 0) {
        HIP_TRY(hipMemcpyAsync(b->d_prd + b->cap_reads, prl, (size_t)nv * sizeof(int4), hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
        launch_pack_live(b->stream, b->d_prd + b->cap_reads, nv, maxb, b->B16, b->d_live);
        HIP_TRY(hipMemcpyAsync(b->d_vb0, vb0, (size_t)nv * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
        HIP_TRY(hipMemcpyAsync(b->d_vb1, vb1, (size_t)nv * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
        HIP_TRY(hipMemcpyAsync(b->d_vtb, vt, (size_t)nv * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    }
    b->hT = lens; b->hTb = vtb;
    b->v_slot.assign(slot, slot + nv); b->v_off.assign(off, off + nv);
    b->nvirt = nv; b->packed = true; b->ragged = true;
    return FFHIP_OK;
}

// the signal rows must be zero beyond each read's end (they are the convolution's right padding)
static int clear_signals(ffhip_batch *b) {
    SampleBuf &sb = b->sbuf[0];
    HIP_TRY(hipMemsetAsync(sb.p, 0, (size_t)b->Bp * sb.rs * 4, b->stream), FFHIP_EHIP);
    return FFHIP_OK;
}

extern "C" int ffhip_batch_set_signals(ffhip_batch *b, const float *signals, size_t ld) {
    if (!b || !signals || ld < (size_t)b->T) return set_err(FFHIP_EINVAL, "bad signal arguments");
    hipSetDevice(b->eng->device);
    if (int rc = apply_lengths(b, std::vector<int>(b->nread, b->T))) return rc;
    SampleBuf &sb = b->sbuf[0];
    HIP_TRY(hipMemcpy2DAsync((void *)(sb.p + kSamplePad), sb.rs * 4, signals, ld * 4, (size_t)b->T * 4, b->nread,
                             hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(b->stream), FFHIP_EHIP);
    b->ran = b->finished = 0;
    return FFHIP_OK;
}

// rows of `ld` floats, read r uses its first nsample[r] <= capacity samples
extern "C" int ffhip_batch_set_signals_ragged(ffhip_batch *b, const float *signals, size_t ld, const size_t *nsample) {
    if (!b || !signals || !nsample) return set_err(FFHIP_EINVAL, "bad signal arguments");
    hipSetDevice(b->eng->device);
    std::vector<int> lens(b->nread);
    for (int r = 0; r < b->nread; r++) {
        if (nsample[r] > ld || nsample[r] > (size_t)b->T) return set_err(FFHIP_EINVAL, "read %d: %zu samples exceed the row / the batch's %d", r, nsample[r], b->T);
        lens[r] = (int)nsample[r];
    }
    if (int rc = apply_lengths(b, lens)) return rc;
    if (int rc = clear_signals(b)) return rc;
    SampleBuf &sb = b->sbuf[0];
    for (int r = 0; r < b->nread; r++)
        if (lens[r] > 0)
            HIP_TRY(hipMemcpyAsync((void *)(sb.p + (size_t)r * sb.rs + kSamplePad), signals + (size_t)r * ld, (size_t)lens[r] * 4,
                                   hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(b->stream), FFHIP_EHIP);
    b->ran = b->finished = 0;
    return FFHIP_OK;
}

extern "C" int ffhip_batch_set_reads(ffhip_batch *b, const raw_table *reads) {
    if (!b || !reads) return set_err(FFHIP_EINVAL, "bad read arguments");
    hipSetDevice(b->eng->device);
    std::vector<int> lens(b->nread);
    for (int r = 0; r < b->nread; r++) {
        const raw_table &rt = reads[r];
        if (!rt.raw || rt.end <= rt.start || rt.end - rt.start > (size_t)b->T)
            return set_err(FFHIP_EINVAL, "read %d: end-start must be within the batch's %d samples", r, b->T);
        lens[r] = (int)(rt.end - rt.start);
    }
    if (int rc = apply_lengths(b, lens)) return rc;
    if (b->ragged) if (int rc = clear_signals(b)) return rc;
    SampleBuf &sb = b->sbuf[0];
    for (int r = 0; r < b->nread; r++)
        HIP_TRY(hipMemcpyAsync((void *)(sb.p + (size_t)r * sb.rs + kSamplePad), reads[r].raw + reads[r].start, (size_t)lens[r] * 4,
                               hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(b->stream), FFHIP_EHIP);
    b->ran = b->finished = 0;
    return FFHIP_OK;
}

static int gather_tables(ffhip_batch *b) {      // device side of ffhip_batch_set_prepared(_packed)'s gather: one entry per read the batch can take
    if (b->d_gsrc) return FFHIP_OK;
    b->d_gsrc = (const float **)dalloc(b, (size_t)b->cap_reads * sizeof(float *), false);
    b->d_glen = (int *)dalloc(b, (size_t)b->cap_reads * 4, false);
    return (b->d_gsrc && b->d_glen) ? FFHIP_OK : FFHIP_ENOMEM;
}

// reads already trimmed and normalised on the device (ffhip_prep_create): device-to-device, no host copy
extern "C" int ffhip_batch_set_prepared(ffhip_batch *b, const ffhip_prep *prep, const int *reads) {
    if (!b || !prep || !reads) return set_err(FFHIP_EINVAL, "bad prepared-read arguments");
    hipSetDevice(b->eng->device);
    std::vector<int> lens(b->nread, 0);
    int *plen = (int *)b->h_glen.get((size_t)b->nread * 4);
    const float **src = (const float **)b->h_gsrc.get((size_t)b->nread * sizeof(float *));
    if (!plen || !src) return set_err(FFHIP_ENOMEM, "pinned host allocation failed");
    for (int r = 0; r < b->nread; r++) {
        size_t len = 0;
        if (reads[r] < 0) { src[r] = nullptr; lens[r] = 0; continue; }            // empty slot
        src[r] = prep_device_signal(prep, reads[r], &len);
        if (!src[r] || len > (size_t)b->T) return set_err(FFHIP_EINVAL, "prepared read %d: rejected by trimming, or longer than the batch's %d samples", reads[r], b->T);
        lens[r] = (int)len;
    }
    if (rehearsal_nogpu()) {                                 // test hook (top of this file): the lengths, and nothing on the device
        b->hT = lens;
        b->hTb.assign(b->nread, 0);
        for (int r = 0; r < b->nread; r++) b->hTb[r] = lens[r] > 0 ? (int)ffhip_model_nblock(b->mdl, (size_t)lens[r]) : 0;
        b->ran = b->finished = 0;
        return FFHIP_OK;
    }
    if (int rc = apply_lengths(b, lens)) return rc;
    if (b->ragged) if (int rc = clear_signals(b)) return rc;
    SampleBuf &sb = b->sbuf[0];
    // one gather launch for the whole batch (a device-to-device copy per read is a launch per read: 512 of them cost more than the copies)
    if (int rc = gather_tables(b)) return rc;
    memcpy(plen, lens.data(), (size_t)b->nread * 4);
    HIP_TRY(hipMemcpyAsync((void *)b->d_gsrc, src, (size_t)b->nread * sizeof(float *), hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(b->d_glen, plen, (size_t)b->nread * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    launch_gather_rows(b->stream, b->d_gsrc, b->d_glen, sb.p + kSamplePad, sb.rs, b->nread);      // (no wait here: the host images are members)
    prep_mark_used(prep, b->stream);                // ffhip_prep_destroy waits for this gather, not for the batch
    b->ran = b->finished = 0;
    return FFHIP_OK;
}

// ---- packed batches: reads of any lengths, several to a row (ffhip.h "packed batches")
extern "C" int ffhip_batch_set_signals_packed(ffhip_batch *b, int nread, const float *const *signals, const size_t *nsample, const int *slot, const int *block_off) {
    if (!b || nread < 0 || (nread > 0 && (!signals || !nsample || !slot || !block_off))) return set_err(FFHIP_EINVAL, "bad packed-signal arguments");
    hipSetDevice(b->eng->device);
    std::vector<int> lens(nread);
    for (int v = 0; v < nread; v++) {
        if (!signals[v] || nsample[v] == 0 || nsample[v] > (size_t)b->T) return set_err(FFHIP_EINVAL, "packed read %d: no signal, or longer than a row's %d samples", v, b->T);
        lens[v] = (int)nsample[v];
    }
    if (int rc = apply_packed(b, nread, lens, slot, block_off)) return rc;
    if (int rc = clear_signals(b)) return rc;
    SampleBuf &sb = b->sbuf[0];
    const int st = total_stride(b->mdl);
    for (int v = 0; v < nread; v++)
        HIP_TRY(hipMemcpyAsync((void *)(sb.p + (size_t)slot[v] * sb.rs + kSamplePad + (size_t)block_off[v] * st), signals[v], (size_t)lens[v] * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(b->stream), FFHIP_EHIP);
    b->ran = b->finished = 0;
    return FFHIP_OK;
}

extern "C" int ffhip_batch_set_prepared_packed(ffhip_batch *b, const ffhip_prep *prep, int nread, const int *reads, const int *slot, const int *block_off) {
    if (!b || !prep || nread < 0 || (nread > 0 && (!reads || !slot || !block_off))) return set_err(FFHIP_EINVAL, "bad packed prepared-read arguments");
    if (rehearsal_rate() > 0) return set_err(FFHIP_EINVAL, "packed batches are not part of the host-load rehearsal");
    hipSetDevice(b->eng->device);
    const size_t nv = (size_t)std::max(1, nread);
    std::vector<int> lens(nread, 0);
    int *plen = (int *)b->h_glen.get(nv * 4);
    const float **src = (const float **)b->h_gsrc.get(nv * sizeof(float *));
    long long *goff = (long long *)b->h_goff.get(nv * sizeof(long long));
    if (!plen || !src || !goff) return set_err(FFHIP_ENOMEM, "pinned host allocation failed");
    SampleBuf &sb = b->sbuf[0];
    const int st = total_stride(b->mdl);
    for (int v = 0; v < nread; v++) {
        size_t len = 0;
        src[v] = reads[v] >= 0 ? prep_device_signal(prep, reads[v], &len) : nullptr;
        if (!src[v] || len == 0 || len > (size_t)b->T) return set_err(FFHIP_EINVAL, "prepared read %d: rejected by trimming, or longer than a row's %d samples", reads[v], b->T);
        lens[v] = plen[v] = (int)len;
    }
    if (int rc = apply_packed(b, nread, lens, slot, block_off)) return rc;
    if (int rc = clear_signals(b)) return rc;
    for (int v = 0; v < nread; v++) goff[v] = (long long)((size_t)slot[v] * sb.rs + (size_t)block_off[v] * st);
    if (!b->d_goff) {      // (sized for the batch's read capacity, like the other per-read tables of a packed batch)
        b->d_goff = (long long *)dalloc(b, (size_t)b->cap_reads * sizeof(long long), false);
        if (!b->d_goff) return FFHIP_ENOMEM;
    }
    if (int rc = gather_tables(b)) return rc;
    if (nread > 0) {
        HIP_TRY(hipMemcpyAsync((void *)b->d_gsrc, src, (size_t)nread * sizeof(float *), hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
        HIP_TRY(hipMemcpyAsync(b->d_glen, plen, (size_t)nread * 4, hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
        HIP_TRY(hipMemcpyAsync(b->d_goff, goff, (size_t)nread * sizeof(long long), hipMemcpyHostToDevice, b->stream), FFHIP_EHIP);
        launch_gather_rows(b->stream, b->d_gsrc, b->d_glen, sb.p + kSamplePad, sb.rs, nread, b->d_goff);
        prep_mark_used(prep, b->stream);
    }
    b->ran = b->finished = 0;
    return FFHIP_OK;
}

static void mark(ffhip_batch *b, int i) {
    if (b->eng->profiling) hipEventRecord(b->ev[i], b->stream);
}

// One run of a batch is enqueued in three phases -- front (convolutions), the recurrent stack, back (head, CRF, decode) -- so that
// ffhip_batch_run_pair can put the layer launches of TWO batches into one grid between their fronts and backs.
enum { PH_FRONT = 1, PH_LAYERS = 2, PH_BACK = 4, PH_ALL = 7 };
static int batch_run_impl(ffhip_batch *b, float temperature, unsigned flags, int phases) {
    if (!b) return set_err(FFHIP_EINVAL, "null batch");
    const ffhip_model *m = b->mdl;
    hipSetDevice(b->eng->device);
    hipStream_t s = b->stream;
    const int Tb = b->Tb, B16 = b->B16, Bp = b->Bp, Hp = m->Hp;
    if (phases & PH_FRONT) {
        if (b->eng->stepwise_batches > 0 && !(flags & FFHIP_RUN_STEPWISE_RNN)) {      // a co-tenant was seen recently (ffhip_batch_finish)
            flags |= FFHIP_RUN_STEPWISE_RNN;
            b->eng->stepwise_batches--;
        }
        b->run_flags = flags;
        b->last_temperature = temperature;
        memset(b->launches, 0, sizeof(b->launches));
    } else {
        flags = b->run_flags;
        temperature = b->last_temperature;
    }
    const bool keep = (flags & FFHIP_RUN_KEEP_ACTS) != 0;
    const int *tbs = b->ragged ? b->d_tbs : nullptr, *tbt = b->ragged ? b->d_tbt : nullptr;      // ragged batch: per-read / per-tile block counts
    // packed batch: tbs / tbt above are the ROWS' extents (what the layer kernels walk); the per-read kernels take the reads' own tables
    const int nR = b->packed ? b->nvirt : b->nread;
    const int *tbr = b->packed ? b->d_vtb : tbs;
    ReadMap rmap;
    if (b->packed) { rmap.b0 = b->d_vb0; rmap.b1 = b->d_vb1; rmap.nslot = b->nread; }
    const unsigned *live = b->packed ? b->d_live : nullptr;
    auto keep_copy = [&](int slot, const float *src) -> int {
        if (!keep) return FFHIP_OK;
        if (!b->keep[slot] && !(b->keep[slot] = (float *)dalloc(b, (size_t)Tb * Bp * Hp * 4, false))) return FFHIP_ENOMEM;
        HIP_TRY(hipMemcpyAsync(b->keep[slot], src, (size_t)Tb * Bp * Hp * 4, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
        return FFHIP_OK;
    };

    // split-operand layer kernel (ffhip_rnn_split.hip): the default wherever it exists (LSTM H = 128..512, GRUmod H = 128..384)
    const bool use_persist = !(flags & FFHIP_RUN_STEPWISE_RNN) && persist_supported(m->cell, Hp, b->eng->prop.multiProcessorCount);
    const bool use_fused = !(flags & FFHIP_RUN_UNFUSED_RNN) && !dbg("no_fuse") && fused_supported(m->cell, Hp);
    const bool want_fused = !(flags & FFHIP_RUN_UNFUSED_RNN) && !dbg("no_fuse");      // (use_fused also asks whether the f32 layer kernel takes the shape)
    const bool use_split = use_persist && want_fused && !(flags & FFHIP_RUN_F32_RNN) && !dbg("no_split") &&
                           split_supported(m->cell, Hp) && m->rnn[0].Wsplit != nullptr;
    // shapes whose two weight matrices do not fit a CU's registers (H = 512): projection GEMM + recurrence-only layer kernel, both
    // on split operands (also what FFHIP_RUN_UNFUSED_RNN selects at H = 256)
    const bool use_split2 = !use_split && use_persist && !(flags & FFHIP_RUN_F32_RNN) && !dbg("no_split") &&
                            rnn_split_supported(m->cell, Hp) && m->rnn[0].Wsplit != nullptr;
    // the last convolution writes the split layout directly unless the fp32 activations are wanted as well
    const bool conv_split = (use_split || use_split2) && !keep && m->conv[m->nconv - 1].Mpad == Hp;
    // the CRF head reads the last layer's SPLIT output (k_head_split): that layer then writes no fp32 copy (315 MB per headline batch, ~65 us of
    // its launch) and the batch needs no fp32 activation buffer at all (FFHIP_NO_SPLIT_HEAD: the f32-MFMA head on the fp32 copy)
    const bool split_head = use_split && !keep && m->FFsplit != nullptr && !dbg("no_split_head");
    const bool prof = b->eng->profiling != 0;
    const int fast_gates = gate_level(flags);
    const char *pm_env = dbg("persist_mode");      // 1 = always use the write-through hand-off
    const int persist_mode = pm_env ? atoi(pm_env) : 0;
    // A packed batch runs on the default path only: the split layer kernels know the live mask, the chain / Viterbi / assembly / trace kernels the read map
    if (b->packed && !(use_split && conv_split && !keep && m->kind != FFHIP_NET_LSTM5_RLE && ((m->nbase == 4 && m->Ps == 40) || (m->nbase == 5 && m->Ps == 60)) &&
                       !dbg("decode_r2") && !dbg("crf_logspace") && 10.0f / temperature <= kFbRange && crf_rescale_interval(5.0f / temperature) > 0))
        return set_err(FFHIP_EINVAL, "packed batches take the default path only (flip-flop model with 128 .. 512 hidden units, no kept activations, no f32 / stepwise / unfused flags, ordinary temperature)");
    int cur = b->run_cur;
    {
        const bool need0 = !conv_split;           // the convolution's fp32 output (every path but split layers behind a split-writing convolution)
        for (int i = need0 ? 0 : 1; i < ((split_head && !need0) ? 1 : 2); i++)      // (act[1]: the last layer's fp32 copy for the f32 head)
            if (!b->act[i] && !(b->act[i] = (float *)dalloc(b, (size_t)Tb * Bp * Hp * 4, false))) return FFHIP_ENOMEM;
    }
  if (phases & PH_FRONT) {
    if (use_split || use_split2) {
        const size_t bytes = split_bytes((size_t)Tb * B16, Hp);
        for (int i = 0; i < 2; i++)
            if (!b->actS[i] && !(b->actS[i] = dalloc(b, bytes, false))) return FFHIP_ENOMEM;
    }
    // ---- convolutions (layers.c:189-276, activations :24-49)
    // the last convolution runs on split operands when the model has them (16 input features): its predecessor then writes fp16 slices
    const bool conv_f16 = m->conv[m->nconv - 1].Wsplit != nullptr && !dbg("no_split_conv") && !(flags & FFHIP_RUN_F32_RNN);
    // another batch is between run and finish: its layer launches hold 384 of every SIMD's 512 registers, so this batch's last
    // convolution takes the shape that fits in what is left (FFHIP_LEAN_CONV=0 / 1 forces one)
    const char *lean_env = dbg("lean_conv");
    // ... unless this batch's own layer launches fill the chip (a paired launch; a full launch of the dense forms; H = 512): the layer
    // launches of whatever else is in flight do too, the convolution only ever shares the chip with other batches' convolutions and
    // decodes, and the fat shape is the faster one there (in pairs at H = 384: 0.98 -> 0.6 ms per batch)
    bool full_chip = b->pair_front != 0 || use_split2;
    if (use_split && !full_chip) {
        const int ncu_ = b->eng->prop.multiProcessorCount, beside_ = (b->eng->in_flight - (b->counted ? 1 : 0) > 0) ? 1 : 0;
        const int nrt_ = split_next_launch_tiles(m->cell, Hp, B16, ncu_);
        full_chip = 2 * split_launch_workgroups(m->cell, Hp, nrt_, ncu_, beside_) > ncu_ * split_workgroups_per_cu(m->cell, Hp, nrt_, ncu_, beside_);
    }
    const int lean_conv = lean_env ? (lean_env[0] == '1') : (!full_chip && b->eng->in_flight - (b->counted ? 1 : 0) > 0);
    // Whole batches one after the other when this batch's layer launches fill the chip and it is not half of a pair: its convolutions
    // would otherwise run beside the other batch's layer launches, whose workgroups then wait for slots (h256 with two 768-read batches
    // in flight: 160 against 179 Msamples/s one at a time; c4: 65.5 against 75).  The host side still overlaps: this only orders the GPU.
    // Round 4: the order is taken from the other batch's LAST LAYER LAUNCH instead of its last kernel -- this batch's convolutions then run
    // beside that batch's head and decode (dependent chains of one wave a read: they leave the chip nearly empty), and a pair's front, on
    // streams of its own (the engine hands out four), no longer queues behind the previous pair's decode: between two pairs' layer launches
    // 1.33 ms of head + decode + copies + convolutions one after the other became max(...) of the two sides.  FFHIP_FRONT_ORDER=batch: round 3's.
    const char *fo = dbg("front_order");
    if (dbg("no_batch_order")) fo = "none";
    const bool by_layers = !fo || fo[0] == 'l';
    {
        if (full_chip && by_layers && b->eng->persist_chained) HIP_TRY(hipStreamWaitEvent(s, b->eng->persist_done, 0), FFHIP_EHIP);
        else if (full_chip && !by_layers && fo[0] == 'b' && !b->pair_front && b->eng->batch_done_rec) HIP_TRY(hipStreamWaitEvent(s, b->eng->batch_done, 0), FFHIP_EHIP);
    }
    mark(b, 0);                                        // (behind the wait: the convolution group's time is its kernels')
    HIP_TRY(hipMemsetAsync(b->sat, 0, (size_t)Bp * sizeof(unsigned), s), FFHIP_EHIP);
    for (int l = 0; l < m->nconv; l++) {
        const ConvDev &c = m->conv[l];
        if (l < m->nconv - 1) {
            launch_conv_small(s, b->sbuf[l], b->sbuf[l + 1], c.taps, c.bias, b->ragged ? b->rag_x0a[l] : b->plan[l].x0a,
                              b->ragged ? b->rag_x0b[l] : b->plan[l].x0b, Bp, b->plan[l].Tout, c.winlen, m->act, b->ragged ? b->plan[l].Tout : 0,
                              (b->ragged && !b->packed) ? b->rag_tin[l] : nullptr, (conv_f16 && l == m->nconv - 2) ? kSplitExpX : -100000, b->sat,
                              (b->packed && m->conv[l].stride == 1) ? b->rag_seg[l] : nullptr);
        } else if (conv_f16) {
            launch_conv_split(s, b->sbuf[l], b->act[0], c.Wsplit, c.bias, b->ragged ? b->rag_x0a[l] : b->plan[l].x0a,
                              b->ragged ? b->rag_x0b[l] : b->plan[l].x0b, B16, Tb, c.Mpad, c.winlen, m->act, b->ragged ? b->plan[l].Tout : 0,
                              conv_split ? b->actS[0] : nullptr, kSplitExpX, c.split_S, lean_conv, b->sat);
        } else {
            launch_conv_mfma(s, b->sbuf[l], b->act[0], c.Wp, c.bias, b->ragged ? b->rag_x0a[l] : b->plan[l].x0a,
                             b->ragged ? b->rag_x0b[l] : b->plan[l].x0b, B16, Tb, c.Mpad, c.K16, m->act, b->ragged ? b->plan[l].Tout : 0,
                             conv_split ? b->actS[0] : nullptr, m->act == ACT_SWISH ? kSplitExpX : kSplitExpH, b->sat);
        }
        b->launches[0]++;
    }
    if (int rc = keep_copy(0, b->act[0])) return rc;
    mark(b, 1);
    // ---- recurrent stack: B,F,B,F,B (networks.c:556-580 / :459-483)
    // profiling groups 1 (in-projection) and 2 (recurrent) interleave; their events bracket the
    // whole stack and the split is measured with per-layer events when profiling is on.
    cur = 0;
    HIP_TRY(hipMemsetAsync(b->pabort, (use_persist && dbg("force_abort")) ? 1 : 0, sizeof(unsigned), s), FFHIP_EHIP);      // (debug: pretend a wait timed out)
    if ((use_split || use_split2) && !conv_split) {
        launch_split_from_f32(s, b->act[0], b->actS[0], (size_t)Tb * B16, Hp, m->act == ACT_SWISH ? kSplitExpX : kSplitExpH, b->sat, B16);
        b->launches[0]++;
    }
    b->run_cur = cur;
    // ... and this batch's LAYER launches follow the decode of the batches before it (the last two: a pair): a persistent launch that becomes
    // resident piecemeal beside a running chain of decode kernels squeezes those onto the CUs it has not taken yet and cannot start before
    // they are through (the run-length shape, whose head and decode are the longer side: 88 against 98 Msamples/s without this wait)
    if (full_chip && by_layers && !dbg("no_decode_wait"))
        for (unsigned k = 1; k <= 2 && k <= b->eng->done_head; k++) HIP_TRY(hipStreamWaitEvent(s, b->eng->done_ring[(b->eng->done_head - k) & 3u], 0), FFHIP_EHIP);
  }      // PH_FRONT
  if (phases & PH_LAYERS) {
    for (int l = 0; l < 5; l++) {
        const RnnDev &r = m->rnn[l];
        const bool backward = (l % 2 == 0);
        float *in = b->act[cur], *out = b->act[cur ^ 1];
        const bool fuse = use_persist && use_fused;
        if (prof) hipEventRecord(b->lev[l][0], s);
        if (use_split2) {
            if (!b->xa && !(b->xa = (float *)dalloc(b, (size_t)Tb * Bp * Hp * 4 * 4, false))) return FFHIP_ENOMEM;
            launch_inproj_split(s, b->actS[cur], b->xa, r.Wsplit, r.bias, Tb * B16, Hp, r.split_S);
            b->launches[1]++;
            if (prof) hipEventRecord(b->lev[l][1], s);
            const int maxt = split_max_tiles(b->eng->prop.multiProcessorCount);
            float *out_f32 = (l == 4 || keep) ? out : nullptr;
            for (int rt0 = 0; rt0 < B16; rt0 += maxt) {
                const int nrt = (B16 - rt0 < maxt) ? B16 - rt0 : maxt;
                HIP_TRY(hipMemsetAsync(b->pflags, 0, split_flag_words(nrt) * sizeof(unsigned), s), FFHIP_EHIP);
                const bool chain = 2 * ((B16 + 1) / 2) * 32 > b->eng->prop.multiProcessorCount;
                if (b->eng->persist_chained && (chain || !b->eng->persist_last_half)) HIP_TRY(hipStreamWaitEvent(s, b->eng->persist_done, 0), FFHIP_EHIP);      // beside another launch only if BOTH are half-chip ones
                if (!launch_rnn_split(s, r.Wsplit, b->xa, b->actS[cur ^ 1], out_f32, b->pflags, b->pabort, Tb, B16, Hp, rt0, nrt,
                                      backward, persist_mode, r.split_S, tbs, tbt))
                    return set_err(FFHIP_EINVAL, "split recurrent kernel: unsupported shape");
                { HIP_TRY(hipEventRecord(b->eng->persist_done, s), FFHIP_EHIP); b->eng->persist_chained = 1; b->eng->persist_last_half = chain ? 0 : 1; }
                b->launches[2]++;
            }
            if (prof) hipEventRecord(b->lev[l][2], s);
            cur ^= 1;
            if (int rc = keep_copy(l + 1, b->act[cur])) return rc;
            continue;
        }
        if (use_split) {
            // H <= 256: a launch of the dense form (pairs of tiles, two workgroups per CU) takes twice the tiles; it is used for FULL
            // launches only -- a partly filled one has a group count that is no multiple of 8 XCDs and loses the one-L2 hand-off
            const int maxt1 = split_max_tiles(b->eng->prop.multiProcessorCount), maxt2 = split_max_tiles(b->eng->prop.multiProcessorCount, Hp);
            void *outS = b->actS[cur ^ 1];
            // (the output doubles as the hand-off flag; the kernel arms it itself a few steps ahead of its stores -- no fill)
            // the fp32 copy of a layer's output is needed by the CRF head (last layer) and by FFHIP_RUN_KEEP_ACTS
            float *out_f32 = ((l == 4 && !split_head) || keep) ? out : nullptr;
            for (int rt0 = 0, nrt = 0; rt0 < B16; rt0 += nrt) {
                nrt = split_next_launch_tiles(m->cell, Hp, B16 - rt0, b->eng->prop.multiProcessorCount);
                (void)maxt1; (void)maxt2;
                // (the check-in words carry the launch's epoch: no fill between launches)
                b->split_epoch = (b->split_epoch % 0x3FFFFFFu) + 1u;
                // two such launches (this batch's and another's in flight) run beside each other only if ALL their workgroups fit on the chip together
                const int ncu_ = b->eng->prop.multiProcessorCount;
                // another batch is between run and finish: at H = 384 this batch's launches take the dense form, which fits beside that batch's
                const int beside = (b->eng->in_flight - (b->counted ? 1 : 0) > 0) ? 1 : 0;
                const bool chain = 2 * split_launch_workgroups(m->cell, Hp, nrt, ncu_, beside) > ncu_ * split_workgroups_per_cu(m->cell, Hp, nrt, ncu_, beside);
                if (b->eng->persist_chained && (chain || !b->eng->persist_last_half)) HIP_TRY(hipStreamWaitEvent(s, b->eng->persist_done, 0), FFHIP_EHIP);      // beside another launch only if BOTH are half-chip ones
                if (prof && rt0 == 0) hipEventRecord(b->lev[l][1], s);      // behind the wait: the layer's time is its kernels', not the other batch's
                if (!launch_lstm_split(s, m->cell, r.Wsplit, r.bias, b->actS[cur], outS, out_f32, b->pflags, b->pabort, Tb, B16, Hp, rt0, nrt,
                                       backward, persist_mode, r.split_S, fast_gates, tbs, tbt, b->eng->prop.multiProcessorCount, b->split_epoch, beside, live))
                    return set_err(FFHIP_EINVAL, "split recurrent kernel: unsupported shape");
                { HIP_TRY(hipEventRecord(b->eng->persist_done, s), FFHIP_EHIP); b->eng->persist_chained = 1; b->eng->persist_last_half = chain ? 0 : 1; }
                b->launches[2]++;
            }
            if (prof) hipEventRecord(b->lev[l][2], s);
            cur ^= 1;
            if (int rc = keep_copy(l + 1, b->act[cur])) return rc;
            continue;
        }
        if (!fuse) {
            if (!b->xa && !(b->xa = (float *)dalloc(b, (size_t)Tb * Bp * Hp * 4 * 4, false))) return FFHIP_ENOMEM;
            if (r.Wsplit && !(flags & FFHIP_RUN_F32_RNN) && !dbg("no_split")) {
                // projection on the bf16 pipes over split operands (fp32-exact products): the layer input is
                // converted to the split layout first
                if (!b->actS[0] && !(b->actS[0] = dalloc(b, split_bytes((size_t)Tb * B16, Hp), false))) return FFHIP_ENOMEM;
                launch_split_from_f32(s, in, b->actS[0], (size_t)Tb * B16, Hp, (l == 0 && m->act == ACT_SWISH) ? kSplitExpX : kSplitExpH, b->sat, B16);
                launch_inproj_split(s, b->actS[0], b->xa, r.Wsplit, r.bias, Tb * B16, Hp, r.split_S);
                b->launches[1] += 2;
            } else {
                launch_inproj(s, in, b->xa, r.iWp, r.bias, Tb * B16, 4 * Hp, r.Kin16);
                b->launches[1]++;
            }
        }
        if (prof) hipEventRecord(b->lev[l][1], s);
        const size_t xa_step = (size_t)Bp * Hp * 4, h_step = (size_t)Bp * Hp;
        if (use_persist) {
            // one launch per layer (and per chunk of read tiles that fits co-resident on the chip)
            const int maxt = persist_max_tiles(m->cell, Hp, b->eng->prop.multiProcessorCount, fuse);
            // the output doubles as the hand-off flag: pre-fill with the NaN sentinel
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)out, (int)0xFFFFFFFF, (size_t)Tb * Bp * Hp, s), FFHIP_EHIP);
            for (int rt0 = 0; rt0 < B16; rt0 += maxt) {
                const int nrt = (B16 - rt0 < maxt) ? B16 - rt0 : maxt;
                HIP_TRY(hipMemsetAsync(b->pflags, 0, persist_flag_words(Hp, nrt) * sizeof(unsigned), s), FFHIP_EHIP);
                const bool chain = !b->persist_concurrent_ok;
                if (b->eng->persist_chained && (chain || !b->eng->persist_last_half)) HIP_TRY(hipStreamWaitEvent(s, b->eng->persist_done, 0), FFHIP_EHIP);      // beside another launch only if BOTH are half-chip ones
                const bool okl = fuse
                    ? launch_lstm_fused(s, m->cell, r.sWp, r.iWp, r.bias, in, out, b->pflags, b->pabort, Tb, B16, Hp, rt0, nrt, backward, persist_mode, tbs, tbt)
                    : launch_rnn_persist(s, m->cell, r.sWp, b->xa, out, b->pflags, b->pabort, Tb, B16, Hp, rt0, nrt, backward, persist_mode, tbs, tbt);
                if (!okl) return set_err(FFHIP_EINVAL, "persistent recurrent kernel: unsupported shape");
                { HIP_TRY(hipEventRecord(b->eng->persist_done, s), FFHIP_EHIP); b->eng->persist_chained = 1; b->eng->persist_last_half = chain ? 0 : 1; }
                b->launches[2]++;
            }
        } else
        for (int i = 0; i < Tb; i++) {
            const int t = backward ? Tb - 1 - i : i;
            const int tp = backward ? t + 1 : t - 1;
            const float *hp = (i == 0) ? nullptr : out + (size_t)tp * h_step;
            if (m->cell == 0)
                launch_lstm_step(s, r.sWp, b->xa + (size_t)t * xa_step, hp, out + (size_t)t * h_step, b->cstate, B16, Hp, i == 0, t, tbs);
            else
                launch_gru_step(s, r.sWp, b->xa + (size_t)t * xa_step, hp, out + (size_t)t * h_step, B16, Hp, i == 0, t, tbs);
        }
        if (!use_persist) b->launches[2] += Tb;
        if (prof) hipEventRecord(b->lev[l][2], s);
        cur ^= 1;
        if (int rc = keep_copy(l + 1, b->act[cur])) return rc;
    }
    b->run_cur = cur;
  }      // PH_LAYERS
  if (!(phases & PH_BACK)) return FFHIP_OK;
    b->profiled = prof;
    b->final_act = cur;
    b->rnn_path = use_split ? 3 : (use_split2 ? 4 : (use_persist ? (use_fused ? 2 : 1) : 0));
    mark(b, 3);
    const bool rle = (m->kind == FFHIP_NET_LSTM5_RLE);
    bool post_done = false;                       // the posterior came out of the partition function's launch
    if (rle) {
        // ---- globalnorm_runlengthV2 (layers.c:1325-1358)
        if (split_head) launch_head_split(s, b->actS[cur], b->trans, m->FFsplit, m->FFb, Tb, B16, b->nread, m->P, m->Ps, Hp / 32, 1.0f, m->FF_split_S, 1);
        else launch_head(s, b->act[cur], b->trans, m->FFp, m->FFb, Tb, B16, b->nread, m->P, m->Ps, Hp / 16, 1.0f, 1);
        launch_rle_head_finish(s, b->trans, b->crf_logz, b->nread, Tb, m->nbase, m->Ps, temperature, tbs);
        b->launches[3] += 4;
    } else {
        // ---- globalnorm_flipflop (layers.c:1082-1106)
        // |score| <= 5/temperature (tanh bounded by 1): picks the rescaling interval of the linear-space form;
        // extreme temperatures (or FFHIP_CRF_LOGSPACE=1) take the log-space recursion
        const int R = dbg("crf_logspace") ? 0 : crf_rescale_interval(5.0f / temperature);
        // 8-state models, a block's scores spanning at most kFbRange: ONE pair of fp64 linear-space chains per read gives logZ, the
        // normalised scores and (when asked for) the posterior (k_crf_fb8, ffhip_decode.hip)
        post_done = R > 0 && ((m->nbase == 4 && m->Ps == 40) || (m->nbase == 5 && m->Ps == 60)) && 10.0f / temperature <= kFbRange && !dbg("decode_r2");
        // ... whose input E = exp(S - block max) the split head leaves behind from its own epilogue (round 5: k_crf_exp's launch and its pass over the scores are gone)
        const bool head_e = split_head && post_done && head_split_writes_E(m->P);
        if (split_head) launch_head_split(s, b->actS[cur], b->trans, m->FFsplit, m->FFb, Tb, B16, b->nread, m->P, m->Ps, Hp / 32, temperature / 5.0f, m->FF_split_S, 0, head_e ? b->crf_e : nullptr);
        else launch_head(s, b->act[cur], b->trans, m->FFp, m->FFb, Tb, B16, b->nread, m->P, m->Ps, Hp / 16, temperature / 5.0f);
        if (b->packed) { HIP_TRY(hipEventRecord(b->eng->head_done, s), FFHIP_EHIP); b->eng->head_done_rec = 1; }      // (the next packed batch's set-up and convolutions start behind it: apply_packed)
        if (post_done) {
            const bool want_post = !(flags & FFHIP_RUN_NO_DECODE) && !(flags & FFHIP_RUN_VITERBI_ONLY);
            if (!head_e) launch_crf_exp(s, b->trans, b->crf_e, b->nread, Tb, m->nbase, m->Ps, tbs, nullptr, 0.0f);
            mark(b, 4);       // the profile's "posterior" slot times the chain launch: partition function + normalisation + posterior together
            launch_crf_fb(s, m->nbase, b->crf_e, b->trans, b->post, (double *)b->fwd, nR, Tb, b->crf_logz, tbr, want_post ? 3 : 1, nullptr, rmap);
            b->launches[3] += want_post ? 4 : 3;      // head, exp, chains, assembly (the subtraction alone: three)
        } else {
            if (R > 0) launch_crf_norm_linear(s, b->trans, b->crf_e, b->nread, Tb, m->nbase, m->Ps, R, b->crf_logz, 1, tbs);
            else launch_crf_norm(s, b->trans, b->nread, Tb, m->nbase, m->Ps, b->crf_logz, 1, tbs);
            b->launches[3] += 3;
        }
    }
    if (!post_done) mark(b, 4);
    b->last_flags = flags;
    if (!(flags & FFHIP_RUN_NO_DECODE)) {
        const float *scores = b->trans;
        if (!(flags & FFHIP_RUN_VITERBI_ONLY)) {
            if (rle && m->nbase == 4 && m->Ps == 40 && 10.0f / temperature <= kFbRange && !dbg("decode_r2"))
                launch_rle_post8(s, b->trans, b->post, b->crf_e, (double *)b->fwd, b->nread, Tb, tbs);              // fp64 linear-space chains (ffhip_decode.hip)
            else if (rle) launch_rle_transpost(s, b->trans, b->post, b->fwd, b->nread, Tb, m->nbase, m->Ps, tbs);       // decode.c:1037-1159
            else if (!post_done) launch_transpost(s, b->trans, b->post, b->fwd, b->nread, Tb, m->nbase, m->Ps, tbs);
            scores = b->post;
            b->launches[4]++;
        }
        mark(b, 5);
        if (rle) {
            // decode_crf_runlength (decode.c:927-1013); the run records are formed from the path by the caller
            // (runnie.c:282-313), there are no base/quality strings or trace for this model
            launch_rle_viterbi(s, scores, b->tb, b->path, b->qpath, b->score, b->nread, Tb, m->nbase, m->Ps, tbs);
            HIP_TRY(hipMemsetAsync(b->lens, 0, (size_t)b->nread * 4, s), FFHIP_EHIP);
            HIP_TRY(hipMemsetAsync(b->bases, 0, (size_t)b->nread * (Tb + 1), s), FFHIP_EHIP);
            HIP_TRY(hipMemsetAsync(b->quals, 0, (size_t)b->nread * (Tb + 1), s), FFHIP_EHIP);
            b->launches[5]++;
        } else {
            launch_viterbi(s, scores, b->tb, b->path, b->qpath, b->score, nR, Tb, m->nbase, m->Ps, tbr, rmap);
            launch_assemble(s, b->path, b->qpath, b->bases, b->quals, b->lens, nR, Tb, m->nbase, tbr, rmap);
            b->launches[5] += 2;
            if (!(flags & FFHIP_RUN_NO_TRACE)) {
                launch_trace(s, scores, b->trace, nR, Tb, m->nbase, m->Ps, 1, tbr, rmap);
                b->launches[5]++;
            }
        }
    } else {
        mark(b, 5);
    }
    mark(b, 6);
    // A packed batch's strings are tens of megabytes (512 rows of 45 000 blocks: 47 MB), and ffhip_batch_finish is called when the NEXT batch's layer launches are
    // already running: the copy's blit kernel then crawls beside them (193 ms in a kernel trace) and holds a layer launch up as long.  It goes here, behind the decode
    // and in front of the events the next batch's layer launches wait for.
    b->res_copied = 0;
    if (b->packed && (phases & PH_BACK)) {
        HIP_TRY(hipMemcpyAsync(b->res_host, b->res_dev, (flags & FFHIP_RUN_NO_DECODE) ? b->res_head : b->res_bytes, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
        b->res_copied = 1;
    }
    HIP_TRY(hipEventRecord(b->eng->batch_done, s), FFHIP_EHIP);
    b->eng->batch_done_rec = 1;
    if (phases & PH_BACK) { HIP_TRY(hipEventRecord(b->eng->done_ring[b->eng->done_head & 3u], s), FFHIP_EHIP); b->eng->done_head++; }
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    b->ran = 1; b->finished = 0;
    if (!b->counted) { b->counted = 1; b->eng->in_flight++; }
    return FFHIP_OK;
}

// ---- host-load rehearsal: the run side (see rehearsal_rate() at the top of this file)
static double now_seconds() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static int rehearsal_run(ffhip_batch *b, float temperature, unsigned flags) {
    hipSetDevice(b->eng->device);
    hipStream_t s = b->stream;
    const size_t n = (size_t)b->nread, L = (size_t)b->Tb + 1;
    int *lens = (int *)b->h_rehearsal.get(n * 4);       // (a buffer of its own: set_prepared's pinned tables may still be waiting for their copies)
    if (!lens) return set_err(FFHIP_ENOMEM, "pinned host allocation failed");
    double samples = 0;
    int w = b->Tb;
    for (size_t r = 0; r < n; r++) if (b->hTb[r] > 0 && b->hTb[r] < w) w = b->hTb[r];
    w = w * 2 / 5;                                        // placeholder calls: 0.4 'A' per block of the batch's shortest read, NUL-terminated rows
    for (size_t r = 0; r < n; r++) { lens[r] = b->hTb[r] > 0 ? w : 0; samples += b->hT[r]; }
    if (rehearsal_nogpu()) {                              // the host mirrors ffhip_batch_finish would have filled
        memset(b->h_bases, 0, n * L); memset(b->h_quals, 0, n * L);
        for (size_t r = 0; r < n; r++) { memset(b->h_bases + r * L, 'A', (size_t)lens[r]); memset(b->h_quals + r * L, '5', (size_t)lens[r]); b->h_lens[r] = lens[r]; b->h_score[r] = 0.0f; }
        *b->h_abort = 0; memset(b->h_sat, 0, (size_t)b->Bp * sizeof(unsigned));
        const double t = now_seconds(), start = t > b->eng->rehearsal_busy_until ? t : b->eng->rehearsal_busy_until;
        b->eng->rehearsal_busy_until = b->rehearsal_done_at = start + samples / (rehearsal_rate() * 1e6);
        b->last_flags = b->run_flags = flags; b->last_temperature = temperature;
        b->ran = 1; b->finished = 0; b->paired_last = 0;
        return FFHIP_OK;
    }
    HIP_TRY(hipMemsetAsync(b->bases, 0, n * L, s), FFHIP_EHIP);
    HIP_TRY(hipMemsetAsync(b->quals, 0, n * L, s), FFHIP_EHIP);
    if (w > 0) {
        HIP_TRY(hipMemset2DAsync(b->bases, L, 'A', (size_t)w, n, s), FFHIP_EHIP);
        HIP_TRY(hipMemset2DAsync(b->quals, L, '5', (size_t)w, n, s), FFHIP_EHIP);
    }
    HIP_TRY(hipMemsetAsync(b->score, 0, n * 4, s), FFHIP_EHIP);
    HIP_TRY(hipMemsetAsync(b->pabort, 0, sizeof(unsigned), s), FFHIP_EHIP);
    HIP_TRY(hipMemsetAsync(b->sat, 0, (size_t)b->Bp * sizeof(unsigned), s), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(b->lens, lens, n * 4, hipMemcpyHostToDevice, s), FFHIP_EHIP);
    const double t = now_seconds(), start = t > b->eng->rehearsal_busy_until ? t : b->eng->rehearsal_busy_until;
    b->eng->rehearsal_busy_until = b->rehearsal_done_at = start + samples / (rehearsal_rate() * 1e6);
    b->last_flags = b->run_flags = flags; b->last_temperature = temperature;
    b->ran = 1; b->finished = 0; b->paired_last = 0;
    return FFHIP_OK;
}

extern "C" int ffhip_batch_run(ffhip_batch *b, float temperature, unsigned flags) {
    if (b) b->paired_last = 0;
    if (b && rehearsal_rate() > 0) return rehearsal_run(b, temperature, flags);
    return batch_run_impl(b, temperature, flags, PH_ALL);
}

// Two batches of the same model and shape, their recurrent layers as ONE launch per layer (k_lstm_split_pair: the dense form of the
// H = 384 layer kernel, two workgroups per CU -- 512 reads in flight as with one 512-read batch).  Everything else of a run is
// enqueued per batch on its own stream as ffhip_batch_run does; shapes this does not apply to simply run one after the other.
extern "C" int ffhip_batch_run_pair(ffhip_batch *b0, ffhip_batch *b1, float temperature, unsigned flags) {
    if (b0 && b1 && rehearsal_rate() > 0) { if (int rc = rehearsal_run(b0, temperature, flags)) return rc; return rehearsal_run(b1, temperature, flags); }
    if (!b0 || !b1 || b0 == b1) return set_err(FFHIP_EINVAL, "two distinct batches are needed");
    const ffhip_model *m = b0->mdl;
    ffhip_engine *eng = b0->eng;
    const int ncu = eng->prop.multiProcessorCount;
    const bool pairable = b1->mdl == m && b1->eng == eng && b0->Tb == b1->Tb && b0->B16 == b1->B16 && b0->packed == b1->packed && b0->B16 <= 2 * (ncu / 32) && (((b0->B16 + 1) / 2) & 7) == 0 &&
                          !(flags & (FFHIP_RUN_KEEP_ACTS | FFHIP_RUN_STEPWISE_RNN | FFHIP_RUN_F32_RNN | FFHIP_RUN_UNFUSED_RNN)) && eng->stepwise_batches == 0 &&
                          m->cell == 0 && m->Hp == 384 && split_supported(m->cell, m->Hp) && m->rnn[0].Wsplit != nullptr && !dbg("no_split") &&
                          !dbg("no_fuse") && !dbg("no_pair") && persist_supported(m->cell, m->Hp, ncu);
    if (!pairable) {
        if (int rc = ffhip_batch_run(b0, temperature, flags)) return rc;
        return ffhip_batch_run(b1, temperature, flags);
    }
    hipSetDevice(eng->device);
    b0->pair_front = b1->pair_front = 1;
    const int rc0 = batch_run_impl(b0, temperature, flags, PH_FRONT), rc1 = rc0 ? rc0 : batch_run_impl(b1, temperature, flags, PH_FRONT);
    b0->pair_front = b1->pair_front = 0;
    if (rc1) return rc1;
    hipStream_t s = b0->stream;
    HIP_TRY(hipEventRecord(b1->pair_ev, b1->stream), FFHIP_EHIP);
    HIP_TRY(hipStreamWaitEvent(s, b1->pair_ev, 0), FFHIP_EHIP);              // the second batch's convolutions are done before the first paired layer
    const bool prof = eng->profiling != 0;
    const int fast_gates = gate_level(flags);
    const char *pm_env = dbg("persist_mode");
    const int persist_mode = pm_env ? atoi(pm_env) : 0;
    ffhip_batch *bb[2] = { b0, b1 };
    const bool split_head_pair = m->FFsplit != nullptr && !dbg("no_split_head");      // as batch_run_impl's split_head (a pair never keeps activations)
    bool paired = true;
    for (int l = 0; l < 5 && paired; l++) {
        const RnnDev &r = m->rnn[l];
        SplitLaunch p[2];
        for (int k = 0; k < 2; k++) {
            ffhip_batch *b = bb[k];
            const int cur = b->run_cur;
            b->split_epoch = (b->split_epoch % 0x3FFFFFFu) + 1u;
            p[k] = SplitLaunch{ r.Wsplit, r.bias, b->actS[cur], b->actS[cur ^ 1], (l == 4 && !split_head_pair) ? b->act[cur ^ 1] : nullptr, b->pflags, b->pabort,
                                b->Tb, b->B16, 0, b->B16, (l % 2 == 0) ? 1 : 0, persist_mode, r.split_S, fast_gates,
                                b->ragged ? b->d_tbs : nullptr, b->ragged ? b->d_tbt : nullptr, b->split_epoch, b->packed ? b->d_live : nullptr };
        }
        if (prof) hipEventRecord(b0->lev[l][1], s);
        if (eng->persist_chained) HIP_TRY(hipStreamWaitEvent(s, eng->persist_done, 0), FFHIP_EHIP);      // a paired launch fills the chip: after any other layer launch
        if (!launch_lstm_split_pair(s, m->cell, m->Hp, ncu, p[0], p[1])) { paired = false; break; }
        HIP_TRY(hipEventRecord(eng->persist_done, s), FFHIP_EHIP);
        eng->persist_chained = 1;
        eng->persist_last_half = 0;
        if (prof) hipEventRecord(b0->lev[l][2], s);
        for (int k = 0; k < 2; k++) {
            bb[k]->launches[2]++;
            bb[k]->run_cur ^= 1;
        }
    }
    if (!paired) return set_err(FFHIP_EINVAL, "paired layer launch refused a shape ffhip_batch_run_pair had accepted");
    HIP_TRY(hipEventRecord(b0->pair_ev, s), FFHIP_EHIP);
    HIP_TRY(hipStreamWaitEvent(b1->stream, b0->pair_ev, 0), FFHIP_EHIP);     // the second batch's head and decode follow the paired layers
    b0->paired_last = b1->paired_last = 1;
    prof_unlink(b0); prof_unlink(b1);                  // (whatever pairs they were part of before, in either role)
    if (prof) { b1->prof_mate = b0; b0->prof_ref = b1; }
    if (int rc = batch_run_impl(b0, temperature, flags, PH_BACK)) return rc;
    return batch_run_impl(b1, temperature, flags, PH_BACK);
}

extern "C" int ffhip_batch_paired(const ffhip_batch *b) { return (b && b->paired_last) ? 1 : 0; }

// Reads of a finished batch that left the split format's range (b->h_sat) again, 16 at a time, through the all-f32 kernels, which
// have no bound (the reference has none: layers.c:24-33); their results replace the clamped ones in the batch's buffers.  Rare by
// construction -- a normalised sample in the hundreds -- so this path is written for clarity: a side batch of 16 slots with the
// same capacity (hence the same strides: a read's results are contiguous device-to-device copies), created on first use.
static int rerun_on_f32_path(ffhip_batch *b, const std::vector<int> &reads) {
    const ffhip_model *m = b->mdl;
    if (!b->side) {
        b->side = ffhip_batch_create(b->eng, m, 16, (size_t)b->T);
        if (!b->side) return FFHIP_ENOMEM;
        b->side->is_side = true;
    }
    ffhip_batch *sd = b->side;
    const size_t Tb = b->Tb, L = Tb + 1, Ps = m->Ps, ns = m->nstate;
    const unsigned fl = b->last_flags;
    for (size_t k0 = 0; k0 < reads.size(); k0 += 16) {
        const int n = (int)std::min<size_t>(16, reads.size() - k0);
        std::vector<int> lens(16, 0);
        for (int k = 0; k < n; k++) lens[k] = b->hT[reads[k0 + k]];
        if (int rc = apply_lengths(sd, lens)) return rc;
        if (int rc = clear_signals(sd)) return rc;
        for (int k = 0; k < n; k++) {
            const int rd = reads[k0 + k];
            const float *from = b->packed ? b->sbuf[0].p + (size_t)b->v_slot[rd] * b->sbuf[0].rs + kSamplePad + (size_t)b->v_off[rd] * total_stride(m)
                                          : b->sbuf[0].p + (size_t)rd * b->sbuf[0].rs + kSamplePad;
            HIP_TRY(hipMemcpyAsync(sd->sbuf[0].p + (size_t)k * sd->sbuf[0].rs + kSamplePad, from, (size_t)lens[k] * 4, hipMemcpyDeviceToDevice, sd->stream), FFHIP_EHIP);
        }
        sd->ran = sd->finished = 0;
        if (int rc = ffhip_batch_run(sd, b->last_temperature, (fl & ~(unsigned)FFHIP_RUN_KEEP_ACTS) | FFHIP_RUN_F32_RNN)) return rc;
        if (int rc = ffhip_batch_finish(sd)) return rc;
        hipStream_t s = b->stream;
        for (int k = 0; k < n; k++) {
            // this read's rows only (a packed batch: the rows behind them are the next read's), at its place in the batch's buffers
            const size_t r = (size_t)reads[k0 + k], r0 = read_row0(b, (int)r), r1 = read_row1(b, (int)r), nb = (size_t)b->hTb[r], nb1 = nb + 1;
            HIP_TRY(hipMemcpyAsync(b->trans + r0 * Ps, sd->trans + (size_t)k * Tb * Ps, nb * Ps * 4, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
            if (fl & FFHIP_RUN_NO_DECODE) continue;
            if (!(fl & FFHIP_RUN_VITERBI_ONLY)) HIP_TRY(hipMemcpyAsync(b->post + r0 * Ps, sd->post + (size_t)k * Tb * Ps, nb * Ps * 4, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
            HIP_TRY(hipMemcpyAsync(b->path + r1, sd->path + (size_t)k * L, nb1 * 4, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
            HIP_TRY(hipMemcpyAsync(b->qpath + r1, sd->qpath + (size_t)k * L, nb1 * 4, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
            HIP_TRY(hipMemcpyAsync(b->score + r, sd->score + k, 4, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
            HIP_TRY(hipMemcpyAsync(b->bases + r1, sd->bases + (size_t)k * L, nb1, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
            HIP_TRY(hipMemcpyAsync(b->quals + r1, sd->quals + (size_t)k * L, nb1, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
            HIP_TRY(hipMemcpyAsync(b->lens + r, sd->lens + k, 4, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
            if (!(fl & FFHIP_RUN_NO_TRACE) && m->kind != FFHIP_NET_LSTM5_RLE)
                HIP_TRY(hipMemcpyAsync(b->trace + r1 * ns, sd->trace + (size_t)k * L * ns, nb1 * ns * 4, hipMemcpyDeviceToDevice, s), FFHIP_EHIP);
            memcpy(b->h_bases + r1, sd->h_bases + (size_t)k * L, nb1);
            memcpy(b->h_quals + r1, sd->h_quals + (size_t)k * L, nb1);
            b->h_lens[r] = sd->h_lens[k];
            b->h_score[r] = sd->h_score[k];
        }
        HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    }
    b->reruns = (int)reads.size();
    b->eng->f32_reruns += reads.size();
    return FFHIP_OK;
}

extern "C" int ffhip_batch_finish(ffhip_batch *b) {
    if (!b) return set_err(FFHIP_EINVAL, "null batch");
    if (!b->ran) return set_err(FFHIP_EINVAL, "ffhip_batch_run has not been called");
    hipSetDevice(b->eng->device);
    const size_t n = (size_t)b->nread, L = (size_t)b->Tb + 1;
    if (rehearsal_nogpu()) {                                 // test hook (top of this file): the results are on the host already
        const double left = b->rehearsal_done_at - now_seconds();
        if (left > 0) { struct timespec ts = { (time_t)left, (long)((left - (double)(time_t)left) * 1e9) }; nanosleep(&ts, nullptr); }
        if (b->counted) { b->counted = 0; b->eng->in_flight--; }
        b->finished = 1; b->reruns = 0;
        return FFHIP_OK;
    }
    // one copy: [sat | abort] and, when the batch was decoded, [lens | score | bases | quals] behind them (the block of ffhip_batch_create)
    if (!b->res_copied) HIP_TRY(hipMemcpyAsync(b->res_host, b->res_dev, (b->last_flags & FFHIP_RUN_NO_DECODE) ? b->res_head : b->res_bytes, hipMemcpyDeviceToHost, b->stream), FFHIP_EHIP);
    b->res_copied = 0;
    (void)n; (void)L;
    HIP_TRY(hipStreamSynchronize(b->stream), FFHIP_EHIP);
    if (rehearsal_rate() > 0) {                              // (test hook above: the emulated GPU finishes this batch at rehearsal_done_at)
        const double left = b->rehearsal_done_at - now_seconds();
        if (left > 0) { struct timespec ts = { (time_t)left, (long)((left - (double)(time_t)left) * 1e9) }; nanosleep(&ts, nullptr); }
    }
    if (b->counted) { b->counted = 0; b->eng->in_flight--; }
    HIP_TRY(hipGetLastError(), FFHIP_EHIP);
    if (*b->h_abort != 0) {
        if ((b->last_flags & FFHIP_RUN_STEPWISE_RNN) || getenv("FFHIP_NO_FALLBACK") || b->packed)      // (a packed batch has no launch-per-step form: the caller sets its reads again, one to a row)
            return set_err(FFHIP_ETIMEOUT, "persistent recurrent kernel: an inter-workgroup wait timed out; results are invalid");
        // Not every workgroup of a persistent layer launch became resident -- something else holds part of the GPU.  The
        // launch-per-step kernels have no such requirement: run this batch again on them, and stay there for a while.
        static int warned = 0;
        if (!warned++) fprintf(stderr, "ffhip: a persistent recurrent kernel timed out waiting for its peer workgroups (is the GPU shared?); "
                                       "falling back to the launch-per-step kernels\n");
        b->eng->fallbacks++;
        b->eng->stepwise_batches = 64;
        const int rc = ffhip_batch_run(b, b->last_temperature, b->last_flags | FFHIP_RUN_STEPWISE_RNN);
        if (rc != FFHIP_OK) return rc;
        return ffhip_batch_finish(b);
    }
    if (b->packed && b->h_abort[2] != 0) return set_err(FFHIP_EINVAL, "packed batch: a read's convolution columns take more than two windows (not a shape of the reference's models)");
    b->finished = 1;
    b->reruns = 0;
    if (!b->is_side) {
        std::vector<int> over;
        if (b->packed) { for (int v = 0; v < b->nvirt; v++) if (b->h_sat[b->v_slot[v]]) over.push_back(v); }      // (the flag is the row's: every read of it goes again)
        else for (int r = 0; r < b->nread; r++) if (b->h_sat[r] && b->hT[r] > 0) over.push_back(r);
        if (!over.empty()) if (int rc = rerun_on_f32_path(b, over)) { b->finished = 0; return rc; }
    }
    return FFHIP_OK;
}

extern "C" int ffhip_batch_f32_reruns(const ffhip_batch *b) { return b ? b->reruns : -1; }
extern "C" unsigned long long ffhip_engine_f32_reruns(const ffhip_engine *eng) { return eng ? eng->f32_reruns : 0; }

static bool results_ok(const ffhip_batch *b, int read) {
    if (!b || !b->finished || read < 0 || read >= batch_nreads(b)) { set_err(FFHIP_EINVAL, "results not available (finish the batch, check the read index)"); return false; }
    if (b->hTb[read] == 0) { set_err(FFHIP_EINVAL, "slot %d of the batch is empty", read); return false; }
    return true;
}

extern "C" const char *ffhip_batch_basecall(const ffhip_batch *b, int read, size_t *length) {
    if (!results_ok(b, read) || (b->last_flags & FFHIP_RUN_NO_DECODE)) return nullptr;
    if (length) *length = (size_t)b->h_lens[read];
    return b->h_bases + read_row1(b, read);
}
extern "C" const char *ffhip_batch_quality(const ffhip_batch *b, int read) {
    if (!results_ok(b, read) || (b->last_flags & FFHIP_RUN_NO_DECODE)) return nullptr;
    return b->h_quals + read_row1(b, read);
}
extern "C" float ffhip_batch_score(const ffhip_batch *b, int read) {
    if (!results_ok(b, read) || (b->last_flags & FFHIP_RUN_NO_DECODE)) return NAN;
    return b->h_score[read];
}

static int d2h(ffhip_batch *b, void *dst, const void *src, size_t bytes) {
    hipSetDevice(b->eng->device);
    HIP_TRY(hipStreamSynchronize(b->stream), FFHIP_EHIP);
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), FFHIP_EHIP);
    return FFHIP_OK;
}

extern "C" int ffhip_batch_get_path(ffhip_batch *b, int read, int *path, float *qpath) {
    if (!results_ok(b, read)) return FFHIP_EINVAL;
    const size_t n = (size_t)b->hTb[read] + 1;      // this read's entries
    if (path) if (int rc = d2h(b, path, b->path + read_row1(b, read), n * 4)) return rc;
    if (qpath) if (int rc = d2h(b, qpath, b->qpath + read_row1(b, read), n * 4)) return rc;
    return FFHIP_OK;
}

static int get_scores(ffhip_batch *b, const float *src, int read, float *out) {
    if (!results_ok(b, read) || !out) return FFHIP_EINVAL;
    const ffhip_model *m = b->mdl;
    const size_t nb = b->hTb[read];             // this read's blocks
    if (m->Ps == m->P) return d2h(b, out, src + read_row0(b, read) * m->Ps, nb * m->P * 4);
    std::vector<float> tmp(nb * m->Ps);
    if (int rc = d2h(b, tmp.data(), src + read_row0(b, read) * m->Ps, tmp.size() * 4)) return rc;
    for (size_t c = 0; c < nb; c++) memcpy(out + c * m->P, tmp.data() + c * m->Ps, (size_t)m->P * 4);
    return FFHIP_OK;
}
extern "C" int ffhip_batch_get_transitions(ffhip_batch *b, int read, float *out) { return b ? get_scores(b, b->trans, read, out) : FFHIP_EINVAL; }
extern "C" int ffhip_batch_transitions_to(ffhip_batch *b, int read, ffhip_mat out) {
    if (!results_ok(b, read) || !out.dev || !out.dev_state) return b ? set_err(FFHIP_EINVAL, "bad arguments") : FFHIP_EINVAL;
    const ffhip_model *m = b->mdl;
    const size_t nb = b->hTb[read];
    if (out.nr != (size_t)m->P || out.nc != nb || out.stride != (size_t)m->Ps) return set_err(FFHIP_EINVAL, "transition matrix must be %d x %zu", m->P, nb);
    hipSetDevice(b->eng->device);
    if (!*out.dev && !(*out.dev = pool_get(nb * m->Ps * 4))) return set_err(FFHIP_ENOMEM, "device allocation failed");
    // the batch holds read r's scores as [block][Ps]: the matrix image itself
    HIP_TRY(hipMemcpyAsync(*out.dev, b->trans + read_row0(b, read) * m->Ps, nb * m->Ps * 4, hipMemcpyDeviceToDevice, b->stream), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(b->stream), FFHIP_EHIP);
    *out.dev_state = 2;
    return FFHIP_OK;
}
extern "C" int ffhip_batch_get_posterior(ffhip_batch *b, int read, float *out) {
    if (b && (b->last_flags & (FFHIP_RUN_VITERBI_ONLY | FFHIP_RUN_NO_DECODE))) return set_err(FFHIP_EINVAL, "posterior was not computed in this run");
    return b ? get_scores(b, b->post, read, out) : FFHIP_EINVAL;
}
extern "C" int ffhip_batch_get_trace(ffhip_batch *b, int read, int32_t *out) {
    if (!results_ok(b, read) || !out) return FFHIP_EINVAL;
    if ((b->last_flags & (FFHIP_RUN_NO_TRACE | FFHIP_RUN_NO_DECODE)) || b->mdl->kind == FFHIP_NET_LSTM5_RLE) return set_err(FFHIP_EINVAL, "trace was not computed in this run");
    return d2h(b, out, b->trace + read_row1(b, read) * b->mdl->nstate, ((size_t)b->hTb[read] + 1) * b->mdl->nstate * 4);
}

extern "C" int ffhip_batch_get_activation(ffhip_batch *b, int layer, int read, float *out) {
    if (!results_ok(b, read) || !out || layer < -1 || layer > 4) return FFHIP_EINVAL;
    if (b->packed) return set_err(FFHIP_EINVAL, "a packed batch keeps no activations");
    const ffhip_model *m = b->mdl;
    const float *src = b->keep[layer + 1];
    if (!src) {
        if (layer == 4 && b->act[b->final_act] && !(b->rnn_path == 3 && b->mdl->FFsplit && !dbg("no_split_head"))) src = b->act[b->final_act];      // (the default path keeps no fp32 copy of the last layer: k_head_split)
        else return set_err(FFHIP_EINVAL, "activation of layer %d was not kept (run with flag 16)", layer);
    }
    hipSetDevice(b->eng->device);
    if (!b->scratch && !(b->scratch = (float *)dalloc(b, (size_t)b->Tb * m->Hp * 4, false))) return FFHIP_ENOMEM;
    launch_untile(b->stream, src, b->scratch, read, b->Tb, b->B16, m->Hp);
    std::vector<float> tmp((size_t)b->Tb * m->Hp);
    if (int rc = d2h(b, tmp.data(), b->scratch, tmp.size() * 4)) return rc;
    for (int t = 0; t < b->Tb; t++) memcpy(out + (size_t)t * m->H, tmp.data() + (size_t)t * m->Hp, (size_t)m->H * 4);
    return FFHIP_OK;
}

extern "C" int ffhip_batch_rnn_path(const ffhip_batch *b) { return b ? b->rnn_path : -1; }

extern "C" int ffhip_debug_fallback_count(const ffhip_engine *eng) { return eng ? eng->fallbacks : -1; }

// development counter next to the abort word (e.g. re-sweeps of the split layer kernel in builds that count them)
extern "C" unsigned ffhip_debug_batch_counter(ffhip_batch *b) {
    unsigned v = 0;
    if (b) { hipSetDevice(b->eng->device); hipMemcpy(&v, b->pabort + 1, 4, hipMemcpyDeviceToHost); }
    return v;
}

extern "C" int ffhip_debug_lean_math_check(ffhip_engine *eng, int exponent, int steps, unsigned long long *mismatches) {
    if (!eng || !mismatches || exponent < 0 || exponent > 125 || steps < 0 || steps > 2) return set_err(FFHIP_EINVAL, "lean math check: bad arguments");
    hipSetDevice(eng->device);
    TmpDev tmp;
    unsigned long long *d = (unsigned long long *)tmp.get(8);
    if (!d) return set_err(FFHIP_ENOMEM, "device allocation failed");
    HIP_TRY(hipMemset(d, 0, 8), FFHIP_EHIP);
    launch_lean_math_check(nullptr, exponent, steps, d);
    HIP_TRY(hipDeviceSynchronize(), FFHIP_EHIP);
    HIP_TRY(hipMemcpy(mismatches, d, 8, hipMemcpyDeviceToHost), FFHIP_EHIP);
    return FFHIP_OK;
}

extern "C" int ffhip_debug_split_round_trip(ffhip_engine *eng, const float *in, float *out, size_t ntile, int hidden) {
    if (!eng || !in || !out || ntile == 0 || hidden <= 0 || hidden % 128 != 0) return set_err(FFHIP_EINVAL, "split round trip: bad arguments");
    hipSetDevice(eng->device);
    TmpDev tmp;
    const size_t nf = ntile * 16 * (size_t)hidden;
    float *d_in = (float *)tmp.get(nf * 4), *d_out = (float *)tmp.get(nf * 4);
    void *d_split = tmp.get(split_bytes(ntile, hidden));
    if (!d_in || !d_out || !d_split) return set_err(FFHIP_ENOMEM, "device allocation failed");
    HIP_TRY(hipMemcpy(d_in, in, nf * 4, hipMemcpyHostToDevice), FFHIP_EHIP);
    launch_split_from_f32(nullptr, d_in, d_split, ntile, hidden, kSplitExpH);
    launch_f32_from_split(nullptr, d_split, d_out, ntile, hidden, kSplitExpH);
    HIP_TRY(hipDeviceSynchronize(), FFHIP_EHIP);
    HIP_TRY(hipMemcpy(out, d_out, nf * 4, hipMemcpyDeviceToHost), FFHIP_EHIP);
    return FFHIP_OK;
}

extern "C" int ffhip_batch_profile(const ffhip_batch *b, float ms[FFHIP_NGROUP], int launches[FFHIP_NGROUP]) {
    if (!b || !b->profiled || !b->finished) return set_err(FFHIP_EINVAL, "the last run was not profiled or the batch is not finished");
    float t01 = 0, t34 = 0, t45 = 0, t56 = 0;
    hipEventElapsedTime(&t01, b->ev[0], b->ev[1]);
    hipEventElapsedTime(&t34, b->ev[3], b->ev[4]);
    hipEventElapsedTime(&t45, b->ev[4], b->ev[5]);
    hipEventElapsedTime(&t56, b->ev[5], b->ev[6]);
    float ms_inproj = 0.f, ms_rnn = 0.f;
    const ffhip_batch *pb = (b->paired_last && b->prof_mate) ? b->prof_mate : b;      // a pair's launches are bracketed once, on the first batch (gone before its mate is asked: the layer time reads 0)
    for (int l = 0; l < 5; l++) {
        float a = 0.f, c = 0.f;
        if (!b->paired_last) hipEventElapsedTime(&a, pb->lev[l][0], pb->lev[l][1]);
        hipEventElapsedTime(&c, pb->lev[l][1], pb->lev[l][2]);
        if (b->launches[1] > 0) ms_inproj += a;      // a fused layer has no projection launch: that interval is the wait for the other batch's layer launches
        ms_rnn += c;
    }
    ms[0] = t01; ms[1] = ms_inproj; ms[2] = ms_rnn; ms[3] = t34; ms[4] = t45; ms[5] = t56;
    for (int i = 0; i < FFHIP_NGROUP; i++) launches[i] = b->launches[i];
    return FFHIP_OK;
}


// ------------------------------------------------------------------------------------ single-matrix decode

extern "C" int ffhip_op_transpost(ffhip_engine *eng, ffhip_mat trans, int return_log, ffhip_mat post) {
    int nbase;
    const size_t nblock = trans.nc, nparam = trans.nr, stride = trans.stride;
    if (!eng || !trans.data || !post.data || nblock == 0 || !flipflop_dims(nparam, stride, &nbase) || post.nr != nparam || post.nc != nblock || post.stride != stride)
        return set_err(FFHIP_EINVAL, "bad transpost arguments");
    hipSetDevice(eng->device);
    hipStream_t s = eng->streams[0];
    TmpDev t;
    const size_t n = nblock * stride;
    const bool lazy = mat_has_dev(trans);                  // the scores live on the device: so does the posterior
    float *d_tr = mat_in(t, trans, s), *d_po = mat_out(t, post, lazy), *d_fw = (float *)t.get(2 * (nblock + 1) * kFwdRowBytes);
    double *d_e = ((nbase == 4 && stride == 40) || (nbase == 5 && stride == 60)) ? (double *)t.get(nblock * crf_exp_stride((int)nparam) * sizeof(double)) : nullptr;
    int *d_wide = d_e ? (int *)t.get(sizeof(int)) : nullptr;
    if (!d_tr || !d_po || !d_fw) return set_err(FFHIP_ENOMEM, "device allocation failed");
    HIP_TRY(hipMemsetAsync(d_po, 0, n * 4, s), FFHIP_EHIP);
    launch_transpost(s, d_tr, d_po, d_fw, 1, (int)nblock, nbase, (int)stride, nullptr, d_e, d_wide);
    if (!return_log) launch_exp_inplace(s, d_po, n);        // (the reference's exp touches pad lanes too: exp(0) = 1, as this one does)
    HIP_TRY(mat_done(post, d_po, lazy, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}
extern "C" int ffhip_transpost(ffhip_engine *eng, const float *trans, size_t nblock, size_t nparam, size_t stride,
                               int return_log, float *post_out) {
    if (!trans || !post_out) return set_err(FFHIP_EINVAL, "bad transpost arguments");
    const ffhip_mat tv = { (float *)trans, nparam, nblock, stride, nullptr, nullptr }, pv = { post_out, nparam, nblock, stride, nullptr, nullptr };
    return ffhip_op_transpost(eng, tv, return_log, pv);
}

extern "C" int ffhip_op_viterbi(ffhip_engine *eng, ffhip_mat scores, int combine_stays, int *path, float *qpath, float *score) {
    int nbase;
    const size_t nblock = scores.nc, nparam = scores.nr, stride = scores.stride;
    if (!eng || !scores.data || !path || !qpath || nblock == 0 || !flipflop_dims(nparam, stride, &nbase)) return set_err(FFHIP_EINVAL, "bad viterbi arguments");
    hipSetDevice(eng->device);
    hipStream_t s = eng->streams[0];
    TmpDev t;
    float *d_sc = mat_in(t, scores, s), *d_q = (float *)t.get((nblock + 1) * 4), *d_s = (float *)t.get(4);
    uint8_t *d_tb = (uint8_t *)t.get(nblock * kMaxState);
    int *d_p = (int *)t.get((nblock + 1) * 4);
    if (!d_sc || !d_q || !d_s || !d_tb || !d_p) return set_err(FFHIP_ENOMEM, "device allocation failed");
    launch_viterbi(s, d_sc, d_tb, d_p, d_q, d_s, 1, (int)nblock, nbase, (int)stride);
    HIP_TRY(hipMemcpyAsync(path, d_p, (nblock + 1) * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipMemcpyAsync(qpath, d_q, (nblock + 1) * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    float sc = NAN;
    HIP_TRY(hipMemcpyAsync(&sc, d_s, 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    if (score) *score = sc;
    if (combine_stays)          // decode.c:194-198
        for (size_t b = 0; b <= nblock; b++) path[b] = (path[b] < nbase) ? path[b] : -1;
    return FFHIP_OK;
}
extern "C" int ffhip_viterbi(ffhip_engine *eng, const float *scores, size_t nblock, size_t nparam, size_t stride,
                             int combine_stays, int *path, float *qpath, float *score) {
    if (!scores) return set_err(FFHIP_EINVAL, "bad viterbi arguments");
    const ffhip_mat v = { (float *)scores, nparam, nblock, stride, nullptr, nullptr };
    return ffhip_op_viterbi(eng, v, combine_stays, path, qpath, score);
}

extern "C" int ffhip_op_trace(ffhip_engine *eng, ffhip_mat post, int32_t *out) {
    int nbase;
    const size_t nblock = post.nc, nparam = post.nr, stride = post.stride;
    if (!eng || !post.data || !out || nblock == 0 || !flipflop_dims(nparam, stride, &nbase)) return set_err(FFHIP_EINVAL, "bad trace arguments");
    hipSetDevice(eng->device);
    hipStream_t s = eng->streams[0];
    TmpDev t;
    const size_t nt = (nblock + 1) * 2 * nbase;
    float *d_po = mat_in(t, post, s);
    int32_t *d_tr = (int32_t *)t.get(nt * 4);
    if (!d_po || !d_tr) return set_err(FFHIP_ENOMEM, "device allocation failed");
    launch_trace(s, d_po, d_tr, 1, (int)nblock, nbase, (int)stride, 0);
    HIP_TRY(hipMemcpyAsync(out, d_tr, nt * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}
extern "C" int ffhip_trace(ffhip_engine *eng, const float *post, size_t nblock, size_t nparam, size_t stride, int32_t *out) {
    if (!post) return set_err(FFHIP_EINVAL, "bad trace arguments");
    const ffhip_mat v = { (float *)post, nparam, nblock, stride, nullptr, nullptr };
    return ffhip_op_trace(eng, v, out);
}
