// ffhip_prep.hip -- signal preparation on the GPU: trim_and_segment_raw (flappie_common.c:13-81) and
// medmad_normalise_array / the --delta transform (util.c:100-223, flappie.c:255-262).
//
// One workgroup per read.  Every order statistic the reference takes from a qsort'ed copy
// (quantilef, util.c:100-139) is an exact selection here:
//   * chunk medians / MADs (chunk_size samples, default 100): rank counting inside one wavefront, the
//     chunk staged in LDS -- four chunks in flight per workgroup;
//   * the threshold quantile over the chunk MADs and the median / MAD of the trimmed read (any length):
//     a 4-pass, 8-bit radix selection over order-preserving integer keys, histogram in LDS, data read
//     from HBM/L2.  The element after the selected one (for the interpolation of util.c:128-133) is the
//     same value if it has duplicates left, else the smallest larger key (one more pass).
// The interpolation and the scale/shift arithmetic are the reference's expressions, operation for
// operation (-ffp-contract=off), so equal inputs give bit-identical trimmed ranges and signals.
// HBM-bound work: 11 passes over 4 bytes/sample, a few hundred microseconds for a 256 x 4000 batch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "ffhip_host.hpp"

using namespace ffhip;

namespace {

constexpr int kMaxChunk = 1024;        // samples per variance-segmentation chunk held in LDS per wave

struct PrepArgs {
    const float *raw;                  // all reads back to back
    float *out;                        // same indexing as raw
    float *madarr;                     // chunk MADs, indexed by the read's sample offset
    const size_t *off;                 // [nread] first sample of each read in raw/out
    const size_t *n_in, *start_in, *end_in;   // the raw_table fields
    size_t *start_out, *end_out;
    float *stats;                      // [nread][2] median, MAD (MEDMAD mode)
    size_t trim_start, trim_end, chunk;
    float perc, delta, shift;
    int mode, do_trim;
};

__device__ __forceinline__ unsigned fkey(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// quantilef's interpolation, util.c:123-133.  a = sorted[idx], b = sorted[idx+1].
__device__ __forceinline__ void quantile_index(float p, size_t nx, size_t *idx, float *remf) {
    const float pos = p * (float)(nx - 1);
    *idx = (size_t)pos;
    *remf = pos - (float)*idx;
}
__device__ __forceinline__ float quantile_mix(float a, float b, float remf, size_t idx, size_t nx) {
    if (idx < nx - 1) return (float)((1.0 - (double)remf) * (double)a + (double)(remf * b));     // `remf * space[idx+1]` is a float product in the reference
    return a;
}

// ---- workgroup-wide exact selection: sorted[k] and sorted[k+1] of get(0..n) ------------------------
// hist: 256 words of LDS, sh: 4 words of LDS.  All 256 threads must call it.
template <class F>
__device__ void wg_select2(F get, size_t n, size_t k, float *vk, float *vk1, unsigned *hist, unsigned *sh) {
    const int tid = threadIdx.x;
    unsigned prefix = 0, mask = 0;
    size_t kk = k;
    unsigned nequal = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        for (size_t i = tid; i < n; i += 256) {
            const unsigned key = fkey(get(i));
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            size_t cum = 0;
            int b = 0;
            for (; b < 255; b++) {
                if (kk < cum + hist[b]) break;
                cum += hist[b];
            }
            sh[0] = (unsigned)b;
            sh[1] = (unsigned)(kk - cum);
            sh[2] = hist[b];
        }
        __syncthreads();
        prefix |= sh[0] << shift;
        mask |= 0xffu << shift;
        kk = sh[1];
        nequal = sh[2];
        __syncthreads();
    }
    *vk = funkey(prefix);
    if (kk + 1 < nequal || k + 1 >= n) { *vk1 = *vk; return; }      // (uniform across the workgroup)
    if (tid == 0) sh[3] = 0xffffffffu;
    __syncthreads();
    unsigned best = 0xffffffffu;
    for (size_t i = tid; i < n; i += 256) {
        const unsigned key = fkey(get(i));
        if (key > prefix && key < best) best = key;
    }
    atomicMin(&sh[3], best);
    __syncthreads();
    *vk1 = funkey(sh[3]);
    __syncthreads();
}

template <class F>
__device__ float wg_quantile(F get, size_t n, float p, unsigned *hist, unsigned *sh) {
    size_t idx; float remf, a, b;
    quantile_index(p, n, &idx, &remf);
    wg_select2(get, n, idx, &a, &b, hist, sh);
    return quantile_mix(a, b, remf, idx, n);
}

// ---- wave-wide exact selection by rank counting; c[0..m) in LDS, result broadcast through res[2] ----
__device__ void wave_select2(const float *c, int m, int k, float *res) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < m; i += 64) {
        const float v = c[i];
        int rank = 0;
        for (int j = 0; j < m; j++) {
            const float w = c[j];
            rank += (w < v || (w == v && j < i)) ? 1 : 0;
        }
        if (rank == k) res[0] = v;
        if (rank == k + 1) res[1] = v;
    }
}

__global__ void __launch_bounds__(256) k_prep(PrepArgs a) {
    __shared__ float chunk_lds[4][kMaxChunk];
    __shared__ float sel[4][2];
    __shared__ unsigned hist[256];
    __shared__ unsigned sh[4];
    __shared__ unsigned lohi[2];
    const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float *x = a.raw + a.off[r];
    float *out = a.out + a.off[r];
    const size_t n_total = a.n_in[r];
    size_t start = a.start_in[r], end = a.end_in[r];

    if (a.do_trim) {
        // ---- trim_raw_by_mad, flappie_common.c:47-81
        const size_t nsample = end - start, chunk = a.chunk;
        const size_t nchunk = nsample / chunk;
        float *madarr = a.madarr + a.off[r];
        const int m = (int)chunk;
        size_t idx; float remf;
        quantile_index(0.5f, chunk, &idx, &remf);
        for (size_t c0 = 0; c0 < nchunk; c0 += 4) {
            const size_t c = c0 + wave;
            const bool live = c < nchunk;
            float *cl = chunk_lds[wave];
            if (live) for (int i = lane; i < m; i += 64) cl[i] = x[start + c * chunk + i];
            __syncthreads();
            if (live) wave_select2(cl, m, (int)idx, sel[wave]);
            __syncthreads();
            float med = 0.0f;
            if (live) {
                med = quantile_mix(sel[wave][0], (idx < chunk - 1) ? sel[wave][1] : sel[wave][0], remf, idx, chunk);
                for (int i = lane; i < m; i += 64) cl[i] = fabsf(cl[i] - med);        // madf, util.c:173-175
            }
            __syncthreads();
            if (live) wave_select2(cl, m, (int)idx, sel[wave]);
            __syncthreads();
            if (live && lane == 0) {
                const float mad = quantile_mix(sel[wave][0], (idx < chunk - 1) ? sel[wave][1] : sel[wave][0], remf, idx, chunk);
                madarr[c] = (1 == chunk) ? 0.0f : mad * 1.4826f;                     // util.c:162-180
            }
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        end = nchunk * chunk;                                                          // flappie_common.c:54 (absolute, as in the reference)
        if (nchunk > 0) {
            const float thresh = wg_quantile([&](size_t i) { return madarr[i]; }, nchunk, a.perc, hist, sh);
            if (tid == 0) { lohi[0] = 0xffffffffu; lohi[1] = 0; }
            __syncthreads();
            unsigned lo = 0xffffffffu, hi = 0;
            for (size_t i = tid; i < nchunk; i += 256)
                if (madarr[i] > thresh) { lo = min(lo, (unsigned)i); hi = max(hi, (unsigned)i + 1); }
            atomicMin(&lohi[0], lo);
            atomicMax(&lohi[1], hi);
            __syncthreads();
            const size_t first = (lohi[0] == 0xffffffffu) ? nchunk : lohi[0];        // chunks dropped at the front (:68-73)
            const size_t last = lohi[1];                                               // chunks kept at the back (:74-79)
            start += first * chunk;
            end -= (nchunk - last) * chunk;
        }
        // ---- trim_and_segment_raw, flappie_common.c:19-26
        start = (n_total - start) > a.trim_start ? start + a.trim_start : n_total;
        end = (end > a.trim_end) ? end - a.trim_end : 0;
    }
    if (tid == 0) { a.start_out[r] = start; a.end_out[r] = end; }
    if (start >= end) return;                                                          // rejected read (uniform)

    const size_t n = end - start;
    const float *y = x + start;
    float *o = out + start;
    if (a.mode == FFHIP_PREP_MEDMAD) {
        // ---- medmad_normalise_array, util.c:198-212
        if (n == 1) { if (tid == 0) { o[0] = 0.0f; a.stats[2 * r] = y[0]; a.stats[2 * r + 1] = 0.0f; } return; }
        const float med = wg_quantile([&](size_t i) { return y[i]; }, n, 0.5f, hist, sh);
        const float mad = wg_quantile([&](size_t i) { return fabsf(y[i] - med); }, n, 0.5f, hist, sh) * 1.4826f;
        for (size_t i = tid; i < n; i += 256) o[i] = (y[i] - med) / mad;
        if (tid == 0) { a.stats[2 * r] = med; a.stats[2 * r + 1] = mad; }
    } else if (a.mode == FFHIP_PREP_DELTA) {
        // ---- difference_array + shift_scale_array(0, delta), util.c:214-223,416-427, flappie.c:261-262
        for (size_t i = tid; i < n; i += 256) {
            const float d = (i + 1 < n) ? (y[i + 1] - y[i]) : 0.0f;
            o[i] = (d - 0.0f) / a.delta;
        }
    } else if (a.mode == FFHIP_PREP_DIFFERENCE) {                                          // util.c:416-427
        for (size_t i = tid; i < n; i += 256) o[i] = (i + 1 < n) ? (y[i + 1] - y[i]) : 0.0f;
    } else if (a.mode == FFHIP_PREP_SHIFT_SCALE) {                                        // util.c:214-223
        for (size_t i = tid; i < n; i += 256) o[i] = (y[i] - a.shift) / a.delta;
    } else {
        for (size_t i = tid; i < n; i += 256) o[i] = y[i];
    }
}

// quantiles of one array (quantilef, util.c:100-139)
__global__ void __launch_bounds__(256) k_quantiles(const float *x, size_t n, float *p, int np) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh[4];
    for (int q = 0; q < np; q++) {
        const float v = wg_quantile([&](size_t i) { return x[i]; }, n, p[q], hist, sh);
        __syncthreads();
        if (threadIdx.x == 0) p[q] = v;
        __syncthreads();
    }
}

// madf (util.c:164-187): MAD about `*med_in`, or about the median when med_in is NULL
__global__ void __launch_bounds__(256) k_mad(const float *x, size_t n, const float *med_in, float *mad_out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh[4];
    if (n == 1) { if (threadIdx.x == 0) *mad_out = 0.0f; return; }
    const float med = med_in ? *med_in : wg_quantile([&](size_t i) { return x[i]; }, n, 0.5f, hist, sh);
    const float mad = wg_quantile([&](size_t i) { return fabsf(x[i] - med); }, n, 0.5f, hist, sh) * 1.4826f;
    if (threadIdx.x == 0) *mad_out = mad;
}

}  // namespace

struct ffhip_prep {
    ffhip_engine *eng = nullptr;
    int nread = 0;
    float *d_out = nullptr;
    size_t d_out_cap = 0;                // bytes, as taken from the engine's pool
    mutable bool used_recorded = false;
    mutable hipEvent_t used = nullptr;   // recorded behind the last asynchronous read of d_out (ffhip_batch_set_prepared's gather)
    std::vector<size_t> off, start, end;
    std::vector<float> stats;
    // ffhip_prep_begin: enqueued and not yet waited for (the tables the uploads read live here until then; `start` / `end` / `stats` are being written)
    bool pending = false;
    std::vector<size_t> n_in, s_in, e_in;
    // pinned landing place of the ranges and statistics, in the engine's staging buffer (a copy into pageable memory is not asynchronous: the call would wait for the kernel)
    void *h_res = nullptr;
};

// ---- buffers kept by the engine (ffhip_host.hpp): no hipMalloc / hipFree -- i.e. no device-wide synchronisation -- per chunk ----
static void *prep_scratch(ffhip_engine *e, int slot, size_t bytes) {
    if (e->prep_scratch_cap[slot] >= bytes && e->prep_scratch[slot]) return e->prep_scratch[slot];
    if (e->prep_scratch[slot]) hipFree(e->prep_scratch[slot]);
    e->prep_scratch[slot] = nullptr; e->prep_scratch_cap[slot] = 0;
    const size_t cap = bytes + bytes / 4 + 256;
    if (hipMalloc(&e->prep_scratch[slot], cap) != hipSuccess) return nullptr;
    e->prep_scratch_cap[slot] = cap;
    return e->prep_scratch[slot];
}
static void *prep_pinned(ffhip_engine *e, size_t bytes) {
    if (e->prep_pin_cap >= bytes && e->prep_pin) return e->prep_pin;
    if (e->prep_pin) hipHostFree(e->prep_pin);
    e->prep_pin = nullptr; e->prep_pin_cap = 0;
    const size_t cap = bytes + bytes / 4 + 256;
    if (hipHostMalloc(&e->prep_pin, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    e->prep_pin_cap = cap;
    return e->prep_pin;
}
static float *prep_pool_take(ffhip_engine *e, size_t bytes, size_t *cap_out) {
    int best = -1;
    for (int i = 0; i < (int)e->prep_pool.size(); i++)
        if (e->prep_pool[i].second >= bytes && (best < 0 || e->prep_pool[i].second < e->prep_pool[best].second)) best = i;
    if (best >= 0) {
        void *q = e->prep_pool[best].first;
        *cap_out = e->prep_pool[best].second;
        e->prep_pool.erase(e->prep_pool.begin() + best);
        return (float *)q;
    }
    void *q = nullptr;
    const size_t cap = bytes + bytes / 4 + 256;
    if (hipMalloc(&q, cap) != hipSuccess) return nullptr;
    *cap_out = cap;
    return (float *)q;
}

// ffhip_batch_set_prepared reads d_out asynchronously on the batch's stream: the buffer may go back to the pool (and be written
// by the next chunk's preparation) only when that read is done
void ffhip::prep_mark_used(const ffhip_prep *p, hipStream_t s) {
    if (!p) return;
    if (!p->used && hipEventCreateWithFlags(&p->used, hipEventDisableTiming) != hipSuccess) { p->used = nullptr; hipStreamSynchronize(s); return; }
    // readers on different streams: the new record must imply the earlier ones (an event remembers its last record only)
    if (p->used_recorded) hipStreamWaitEvent(s, p->used, 0);
    hipEventRecord(p->used, s);
    p->used_recorded = true;
}

extern "C" void ffhip_prep_destroy(ffhip_prep *p) {
    if (!p) return;
    if (p->pending) { hipStreamSynchronize(p->eng->prep_stream); p->pending = false; }
    if (p->used) { hipEventSynchronize(p->used); hipEventDestroy(p->used); }
    if (p->d_out) {
        // back to the engine's pool (at most four buffers wait there; the smallest goes when a fifth arrives)
        auto &pool = p->eng->prep_pool;
        pool.emplace_back((void *)p->d_out, p->d_out_cap);
        if (pool.size() > 4) {
            size_t k = 0;
            for (size_t i = 1; i < pool.size(); i++) if (pool[i].second < pool[k].second) k = i;
            hipFree(pool[k].first);
            pool.erase(pool.begin() + k);
        }
    }
    delete p;
}

static ffhip_prep *prep_run(ffhip_engine *eng, const raw_table *reads, int nread, size_t trim_start, size_t trim_end,
                            size_t chunk, float perc, int mode, float delta, int do_trim, float shift = 0.0f, bool wait = true) {
    if (!eng || !reads || nread <= 0) { set_err(FFHIP_EINVAL, "bad signal-preparation arguments"); return nullptr; }
    if (mode < FFHIP_PREP_MEDMAD || mode > FFHIP_PREP_SHIFT_SCALE) { set_err(FFHIP_EINVAL, "unknown preparation mode %d", mode); return nullptr; }
    if (do_trim && (chunk < 2 || chunk > (size_t)kMaxChunk || !(perc >= 0.0f && perc <= 1.0f))) {
        set_err(FFHIP_EINVAL, "segmentation chunk must be 2..%d samples and the quantile within [0,1] (flappie_common.c:48-49)", kMaxChunk);
        return nullptr;
    }
    if ((mode == FFHIP_PREP_DELTA || mode == FFHIP_PREP_SHIFT_SCALE) && !(delta != 0.0f)) { set_err(FFHIP_EINVAL, "delta scaling factor must be non-zero"); return nullptr; }
    hipSetDevice(eng->device);
    hipStream_t s = eng->prep_stream;      // beside the batches' streams: a chunk is prepared while the previous chunk's last batch runs
    ffhip_prep *p = new ffhip_prep();
    p->eng = eng;
    p->nread = nread;
    p->off.resize(nread); p->start.resize(nread); p->end.resize(nread); p->stats.assign((size_t)2 * nread, 0.0f);
    std::vector<size_t> &n_in = p->n_in, &s_in = p->s_in, &e_in = p->e_in;
    n_in.resize(nread); s_in.resize(nread); e_in.resize(nread);
    size_t total = 0;
    for (int r = 0; r < nread; r++) {
        const raw_table &rt = reads[r];
        if (!rt.raw || rt.n == 0 || rt.end > rt.n || rt.start > rt.end) {
            set_err(FFHIP_EINVAL, "read %d: empty signal or start/end outside [0, n]", r);
            ffhip_prep_destroy(p);
            return nullptr;
        }
        p->off[r] = total; n_in[r] = rt.n; s_in[r] = rt.start; e_in[r] = rt.end;
        total += (rt.n + 3) & ~(size_t)3;
    }
    float *d_raw = (float *)prep_scratch(eng, 0, total * 4), *d_mad = (float *)prep_scratch(eng, 1, total * 4);
    size_t *d_sz = (size_t *)prep_scratch(eng, 2, (size_t)6 * nread * sizeof(size_t));
    float *d_stats = (float *)prep_scratch(eng, 3, (size_t)2 * nread * 4);
    // (behind the raw samples: the landing place of ffhip_prep_begin's results -- ranges [2 nread] size_t, statistics [2 nread] float -- in the same pinned buffer)
    const size_t res_off = (total * 4 + 63) & ~(size_t)63, res_bytes = (size_t)2 * nread * sizeof(size_t) + (size_t)2 * nread * 4;
    float *pin = (float *)prep_pinned(eng, res_off + res_bytes);
#define PFAIL(code, msg) do { set_err(code, msg); ffhip_prep_destroy(p); return nullptr; } while (0)
    if (!d_raw || !d_mad || !d_sz || !d_stats || !pin || !(p->d_out = prep_pool_take(eng, total * 4, &p->d_out_cap))) PFAIL(FFHIP_ENOMEM, "device allocation failed");
    // one packed upload for the chunk: the reads are gathered in pinned memory first (a copy per read from pageable memory costs
    // 10-20 us of launch and staging each -- 30 ms for 2048 reads, during which nothing else was submitted)
    for (int r = 0; r < nread; r++) memcpy(pin + p->off[r], reads[r].raw, reads[r].n * 4);
    if (rehearsal_nogpu()) {             // test hook (ffhip_engine.hip, "host-load rehearsal"): the host's share is done; fixed trims stand for the segmentation
        for (int r = 0; r < nread; r++) {
            p->start[r] = std::min(e_in[r], s_in[r] + trim_start);
            p->end[r] = std::max(p->start[r], e_in[r] > trim_end ? e_in[r] - trim_end : (size_t)0);
            p->stats[2 * (size_t)r] = 0.0f; p->stats[2 * (size_t)r + 1] = 1.0f;
        }
        return p;
    }
    if (hipMemcpyAsync(d_raw, pin, total * 4, hipMemcpyHostToDevice, s) != hipSuccess) PFAIL(FFHIP_EHIP, "upload of raw signal failed");
    size_t *d_off = d_sz, *d_n = d_sz + nread, *d_s = d_sz + 2 * (size_t)nread, *d_e = d_sz + 3 * (size_t)nread, *d_so = d_sz + 4 * (size_t)nread, *d_eo = d_sz + 5 * (size_t)nread;
    bool ok = hipMemcpyAsync(d_off, p->off.data(), nread * sizeof(size_t), hipMemcpyHostToDevice, s) == hipSuccess;
    ok = ok && hipMemcpyAsync(d_n, n_in.data(), nread * sizeof(size_t), hipMemcpyHostToDevice, s) == hipSuccess;
    ok = ok && hipMemcpyAsync(d_s, s_in.data(), nread * sizeof(size_t), hipMemcpyHostToDevice, s) == hipSuccess;
    ok = ok && hipMemcpyAsync(d_e, e_in.data(), nread * sizeof(size_t), hipMemcpyHostToDevice, s) == hipSuccess;
    if (eng->persist_chained && total >= ((size_t)48 << 20)) hipStreamWaitEvent(s, eng->persist_done, 0);      // (see below: the fill and the kernel of a LARGE chunk)
    ok = ok && hipMemsetAsync(p->d_out, 0, total * 4, s) == hipSuccess;
    if (!ok) PFAIL(FFHIP_EHIP, "upload of read table failed");
    PrepArgs a{ d_raw, p->d_out, d_mad, d_off, d_n, d_s, d_e, d_so, d_eo, d_stats, trim_start, trim_end, chunk, perc, delta, shift, mode, do_trim };
    // A LARGE chunk (the flappie binary widens its window when long reads turn up: 100 M samples and more) is tens of milliseconds of this kernel alone -- and 190 ms
    // beside a resident layer launch of the batch in flight, which it slows down as much (kernel trace, tools/dev/mixed_trace.sh): such a chunk is prepared behind
    // the engine's last layer launch (the wait stands in front of the output's fill above; the uploads before it are not held up: copies do not wait for compute).
    // The usual chunks (a millisecond or two) stay where they were.
    hipLaunchKernelGGL(k_prep, dim3(nread), dim3(256), 0, s, a);
    if (!wait) {
        // ffhip_prep_begin: the results land in pinned memory (ffhip_prep_finish waits and takes them from there); d_so and d_eo stand one behind the other
        const size_t nb = (size_t)2 * nread * sizeof(size_t);
        p->h_res = (char *)pin + res_off;
        ok = hipMemcpyAsync(p->h_res, d_so, nb, hipMemcpyDeviceToHost, s) == hipSuccess;
        ok = ok && hipMemcpyAsync((char *)p->h_res + nb, d_stats, (size_t)2 * nread * 4, hipMemcpyDeviceToHost, s) == hipSuccess;
        if (!ok) PFAIL(FFHIP_EHIP, "signal-preparation kernel failed");
        p->pending = true;
        return p;
    }
    ok = hipMemcpyAsync(p->start.data(), d_so, nread * sizeof(size_t), hipMemcpyDeviceToHost, s) == hipSuccess;
    ok = ok && hipMemcpyAsync(p->end.data(), d_eo, nread * sizeof(size_t), hipMemcpyDeviceToHost, s) == hipSuccess;
    ok = ok && hipMemcpyAsync(p->stats.data(), d_stats, (size_t)2 * nread * 4, hipMemcpyDeviceToHost, s) == hipSuccess;
    ok = ok && hipStreamSynchronize(s) == hipSuccess && hipGetLastError() == hipSuccess;
    if (!ok) PFAIL(FFHIP_EHIP, "signal-preparation kernel failed");
#undef PFAIL
    return p;
}

// The two halves of ffhip_prep_create (round 6): begin enqueues the uploads, the kernel and the copies of the ranges on the engine's preparation stream and returns;
// finish waits for them.  A caller with a pipeline of chunks begins chunk k + 1 before it submits chunk k's batches: the preparation -- one workgroup a read, as long as
// its longest read's selection passes -- then runs BESIDE those batches' convolutions instead of in front of the next ones' (profiles/r06_pack_trace.txt).
// ONE preparation may be pending at a time (the engine's staging buffers are shared); ranges, statistics and signals are there after finish.
extern "C" ffhip_prep *ffhip_prep_begin(ffhip_engine *eng, const raw_table *reads, int nread, size_t trim_start, size_t trim_end,
                                        size_t varseg_chunk, float varseg_thresh, int mode, float delta) {
    return prep_run(eng, reads, nread, trim_start, trim_end, varseg_chunk, varseg_thresh, mode, delta, 1, 0.0f, false);
}
extern "C" int ffhip_prep_finish(ffhip_prep *p) {
    if (!p) return set_err(FFHIP_EINVAL, "no preparation");
    if (!p->pending) return FFHIP_OK;
    hipSetDevice(p->eng->device);
    p->pending = false;
    if (hipStreamSynchronize(p->eng->prep_stream) != hipSuccess || hipGetLastError() != hipSuccess) return set_err(FFHIP_EHIP, "signal-preparation kernel failed");
    if (p->h_res) {
        const size_t *r = (const size_t *)p->h_res;
        const float *st = (const float *)((const char *)p->h_res + (size_t)2 * p->nread * sizeof(size_t));
        for (int i = 0; i < p->nread; i++) { p->start[i] = r[i]; p->end[i] = r[p->nread + i]; }
        for (size_t i = 0; i < (size_t)2 * p->nread; i++) p->stats[i] = st[i];
        p->h_res = nullptr;
    }
    return FFHIP_OK;
}

extern "C" ffhip_prep *ffhip_prep_create(ffhip_engine *eng, const raw_table *reads, int nread, size_t trim_start, size_t trim_end,
                                         size_t varseg_chunk, float varseg_thresh, int mode, float delta) {
    return prep_run(eng, reads, nread, trim_start, trim_end, varseg_chunk, varseg_thresh, mode, delta, 1);
}

extern "C" int ffhip_prep_range(const ffhip_prep *p, int read, size_t *start, size_t *end) {
    if (!p || read < 0 || read >= p->nread) return set_err(FFHIP_EINVAL, "bad prepared-read index");
    if (start) *start = p->start[read];
    if (end) *end = p->end[read];
    return FFHIP_OK;
}

extern "C" int ffhip_prep_stats(const ffhip_prep *p, int read, float *median, float *mad) {
    if (!p || read < 0 || read >= p->nread) return set_err(FFHIP_EINVAL, "bad prepared-read index");
    if (median) *median = p->stats[2 * (size_t)read];
    if (mad) *mad = p->stats[2 * (size_t)read + 1];
    return FFHIP_OK;
}

extern "C" int ffhip_prep_get_signal(const ffhip_prep *p, int read, float *out) {
    if (!p || !out || read < 0 || read >= p->nread) return set_err(FFHIP_EINVAL, "bad prepared-read index");
    if (p->start[read] >= p->end[read]) return set_err(FFHIP_EINVAL, "read %d was rejected by trimming", read);
    hipSetDevice(p->eng->device);
    HIP_TRY(hipMemcpy(out, p->d_out + p->off[read] + p->start[read], (p->end[read] - p->start[read]) * 4, hipMemcpyDeviceToHost), FFHIP_EHIP);
    return FFHIP_OK;
}

// device address of a prepared read's first kept sample (for ffhip_batch_set_prepared, ffhip_engine.hip)
const float *ffhip::prep_device_signal(const ffhip_prep *p, int read, size_t *len) {
    if (!p || read < 0 || read >= p->nread || p->start[read] >= p->end[read]) return nullptr;
    *len = p->end[read] - p->start[read];
    return p->d_out + p->off[read] + p->start[read];
}

// ---- single-array entry points behind util.h's names -------------------------------------------------
extern "C" int ffhip_quantiles(ffhip_engine *eng, const float *x, size_t n, float *p, size_t np) {
    if (!eng || !x || !p || n == 0 || np == 0 || np > 1024) return set_err(FFHIP_EINVAL, "bad quantile arguments");
    for (size_t i = 0; i < np; i++) if (!(p[i] >= 0.0f && p[i] <= 1.0f)) return set_err(FFHIP_EINVAL, "quantile outside [0,1] (util.c:104-106)");
    hipSetDevice(eng->device);
    hipStream_t s = eng->streams[0];
    TmpDev tmp;
    float *d_x = (float *)tmp.upload(x, n * 4, s), *d_p = (float *)tmp.upload(p, np * 4, s);
    if (!d_x || !d_p) return set_err(FFHIP_ENOMEM, "device allocation failed");
    hipLaunchKernelGGL(k_quantiles, dim3(1), dim3(256), 0, s, d_x, n, d_p, (int)np);
    HIP_TRY(hipMemcpyAsync(p, d_p, np * 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}

extern "C" int ffhip_medmad_normalise(ffhip_engine *eng, float *x, size_t n, float *median, float *mad) {
    if (!x || n == 0) return set_err(FFHIP_EINVAL, "bad normalisation arguments");
    raw_table rt;
    rt.uuid = nullptr; rt.n = n; rt.start = 0; rt.end = n; rt.raw = x;
    ffhip_prep *p = prep_run(eng, &rt, 1, 0, 0, 0, 0.0f, FFHIP_PREP_MEDMAD, 0.0f, 0);
    if (!p) return FFHIP_EINVAL;
    const int rc = ffhip_prep_get_signal(p, 0, x);
    if (median) *median = p->stats[0];
    if (mad) *mad = p->stats[1];
    ffhip_prep_destroy(p);
    return rc;
}

// difference_array / shift_scale_array (util.c:214-223,416-427) on one array, in place
extern "C" int ffhip_array_transform(ffhip_engine *eng, float *x, size_t n, int mode, float shift, float scale) {
    if (!x || n == 0 || (mode != FFHIP_PREP_DIFFERENCE && mode != FFHIP_PREP_SHIFT_SCALE && mode != FFHIP_PREP_DELTA)) return set_err(FFHIP_EINVAL, "bad array-transform arguments");
    raw_table rt;
    rt.uuid = nullptr; rt.n = n; rt.start = 0; rt.end = n; rt.raw = x;
    ffhip_prep *p = prep_run(eng, &rt, 1, 0, 0, 0, 0.0f, mode, scale, 0, shift);
    if (!p) return FFHIP_EINVAL;
    const int rc = ffhip_prep_get_signal(p, 0, x);
    ffhip_prep_destroy(p);
    return rc;
}

extern "C" int ffhip_mad(ffhip_engine *eng, const float *x, size_t n, const float *med, float *mad) {
    if (!eng || !x || !mad || n == 0) return set_err(FFHIP_EINVAL, "bad MAD arguments");
    hipSetDevice(eng->device);
    hipStream_t s = eng->streams[0];
    TmpDev tmp;
    float *d_x = (float *)tmp.upload(x, n * 4, s), *d_m = (float *)tmp.get(8);
    if (!d_x || !d_m) return set_err(FFHIP_ENOMEM, "device allocation failed");
    if (med) HIP_TRY(hipMemcpyAsync(d_m + 1, med, 4, hipMemcpyHostToDevice, s), FFHIP_EHIP);
    hipLaunchKernelGGL(k_mad, dim3(1), dim3(256), 0, s, d_x, n, med ? d_m + 1 : nullptr, d_m);
    HIP_TRY(hipMemcpyAsync(mad, d_m, 4, hipMemcpyDeviceToHost, s), FFHIP_EHIP);
    HIP_TRY(hipStreamSynchronize(s), FFHIP_EHIP);
    return FFHIP_OK;
}
