/*
 * wm_api.hip -- host pipeline behind the C ABI of include/wmbus_hip.h.
 *
 * One context owns one HIP stream, the HBM-resident buffers of wm_dev.h, pinned host staging
 * for bursts, and the persistent host packet decoders (4 per capture: chain x framer).
 * There is NO CPU fallback: without a HIP device wmbus_open() fails with WMBUS_ENODEVICE.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wmbus_hip.h"
#include "wm_decoder.h"
#include "wm_dev.h"
#include "wm_kernels.hip"

namespace {

struct LineRec {
    uint64_t sample; uint32_t stream; uint8_t chain, algo, crc_ok; uint32_t seq; std::string text;
};

/* Persistent host worker pool of a context (packet decoders): run(n, f) executes f(0..n-1) on the
 * workers and the caller; threads are created once, not per push. */
class WorkerPool {
public:
    explicit WorkerPool(unsigned n_workers)
    {
        for (unsigned i = 0; i < n_workers; i++) th_.emplace_back([this] { loop(); });
    }
    ~WorkerPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    unsigned size() const { return (unsigned)th_.size(); }
    template <typename F> void run(unsigned n, F &&f)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = f; next_ = 0; total_ = n; done_ = 0; gen_++;      /* the job lives in the pool, not on this stack frame */
        }
        cv_.notify_all();
        work();                                              /* the caller helps */
        std::unique_lock<std::mutex> lk(m_);
        /* every item done AND every worker that picked this generation up has left work(): nothing of
         * this run can still be executing when the caller's captures go out of scope */
        cv_done_.wait(lk, [&] { return done_ == total_ && active_ == 0; });
        job_ = nullptr;
    }
private:
    void work()
    {
        for (;;) {
            unsigned i;
            { std::lock_guard<std::mutex> lk(m_); if (next_ >= total_) return; i = next_++; }
            job_(i);                                         /* job_ only changes while no item is outstanding */
            { std::lock_guard<std::mutex> lk(m_); if (++done_ == total_) cv_done_.notify_all(); }
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (gen_ != seen && job_ != nullptr); });
                if (stop_) return;
                seen = gen_; active_++;
            }
            work();
            { std::lock_guard<std::mutex> lk(m_); if (--active_ == 0) cv_done_.notify_all(); }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    std::function<void(unsigned)> job_;
    unsigned next_ = 0, total_ = 0, done_ = 0, active_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

struct HostDecoder {           /* persistent across pushes */
    wm_decoder dec;
    uint32_t owed;             /* chips to request at the start of the next push */
};

}  // namespace

struct wmbus_ctx {
    wmbus_cfg cfg{};
    char err[256] = {0};
    hipStream_t stream = nullptr;
    hipEvent_t ev[9] = {};
    /* geometry */
    uint32_t d = 2, S = 1, C[2] = {8192, 32768}, Mcap = 0, nseg_cap[2] = {0, 0}, ntiles_cap = 0, T = WM_K1_TILE2;
    uint32_t cap[2] = {0, 0}, flags = 0;
    uint64_t in_stride = 0, n0 = 0;
    size_t staged = 0;
    /* device buffers */
    uint8_t *d_in = nullptr; float *d_dphi = nullptr; uint8_t *d_rssi = nullptr; uint32_t *d_bits = nullptr;
    float *d_lut = nullptr;
    float *d_ema_head = nullptr, *d_ema_tail = nullptr, *d_ema_carry = nullptr;
    uint32_t *d_chips[2] = {}, *d_counts[2] = {};
    void *d_st_start[2] = {}, *d_st_final[2] = {}, *d_st_carry[2] = {};
    uint32_t *d_list2 = nullptr;                        /* run-length re-run list of the fused framer launches */
    uint32_t *d_list = nullptr, *d_scalars = nullptr;   /* scalars: err, n_list, n_hits, n_hdr, n_words */
    uint32_t *d_sync_seen[2] = {};                      /* per framer: [2][S][nseg_cap] access-code chip seen in region */
    uint32_t *d_spill = nullptr, *d_chain = nullptr, *d_nchain = nullptr; uint32_t spill_words = 0;   /* WmSpill (wm_dev.h) */
    bool poisoned = false;                              /* an internal error left the carried state undefined */
    uint32_t *d_first_bad = nullptr;                    /* [2][S] first uncertified EMA tile of a row, or ~0 */
    uint32_t *d_ckpt = nullptr; uint32_t nck = 0;       /* clock kernel checkpoints [2][S][nseg_cap][nck][16] */
    uint2 *d_hits = nullptr; uint32_t hits_cap = 0;
    uint32_t *d_pending = nullptr;
    WmBurstHdr *d_hdr = nullptr; uint32_t hdr_cap = 0;
    uint32_t *d_words = nullptr; uint32_t words_cap = 0;
    /* host */
    uint32_t *h_scalars = nullptr;                      /* pinned */
    WmBurstHdr *h_hdr = nullptr; uint32_t *h_words = nullptr; uint32_t *h_pending = nullptr;
    uint32_t n_hdr = 0, n_words = 0;
    std::vector<HostDecoder> decs;                      /* [stream][chain][algo] */
    std::unique_ptr<WorkerPool> pool;                   /* packet-decoder workers, created on first use */
    std::vector<wmbus_line> lines; std::string text;
    WmPush last{}; bool have_last = false, in_flight = false;
    std::atomic<uint32_t> short_burst{0};               /* set by the decoder threads of one collect: 1 + header index */
    wmbus_timing tim{};
};

namespace {

int fail(wmbus_ctx *c, int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return fail((c), WMBUS_EDEVICE, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

template <typename T> hipError_t dalloc(T **p, size_t n) { return hipMalloc((void **)p, n * sizeof(T)); }

enum { SC_ERR = 0, SC_NLIST = 1, SC_NHITS = 2, SC_NHDR = 3, SC_NWORDS = 4, SC_NLIST2 = 5, SC_CHIPS = 8 /* [algo][chain] */, SC_COUNT = 16 };

__global__ void k_roll_history(uint8_t *in, uint64_t stride, uint32_t nbytes)
{
    /* new history = the 4096 bytes that end at the end of the staged data */
    uint8_t *row = in + (uint64_t)blockIdx.x * stride;
    const uint4 v = *(const uint4 *)(row + nbytes + 16u * threadIdx.x);
    __syncthreads();
    *(uint4 *)(row + 16u * threadIdx.x) = v;
}

__global__ void k_fill(uint8_t *p, uint64_t stride, uint32_t n, uint8_t v)
{
    uint8_t *row = p + (uint64_t)blockIdx.x * stride;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) row[i] = v;
}

__global__ void k_carry(const uint32_t *fin, uint32_t *carry, uint32_t words, uint32_t rows, uint32_t nseg_cap, uint32_t nseg)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    for (uint32_t k = 0; k < words; k++) carry[(uint64_t)r * words + k] = fin[((uint64_t)r * nseg_cap + nseg - 1) * words + k];
}

/* wmbus_timing.chips: chips produced per (framer, chain) in this push = sum of the settled segment counts.  One
 * atomic per wave and sum (82 000 threads adding to four words directly cost the whole GPU 15 % while they ran). */
__global__ __launch_bounds__(256) void k_sum_counts(WmPush g, const uint32_t *counts0, const uint32_t *counts1, uint32_t *sums)
{
    const uint32_t n0 = 2u * g.nseg_cap[0] * g.S, n1 = 2u * g.nseg_cap[1] * g.S;
    uint32_t part[4] = {0u, 0u, 0u, 0u};
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n0 + n1; t += gridDim.x * blockDim.x) {
        const uint32_t algo = t >= n0, i = algo ? t - n0 : t;
        const uint32_t seg = i % g.nseg_cap[algo], ch = i / g.nseg_cap[algo] / g.S;
        if (seg < g.nseg[algo] && (g.flags & (algo ? WM_F_T2A : WM_F_RLA))) part[algo * 2u + ch] += (algo ? counts1 : counts0)[i];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t v = part[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63u) == 0u && v) atomicAdd(sums + k, v);
    }
}

template <int D, bool SHIFT, bool GEN> int launch_k1v3(wmbus_ctx *c, const K1Args &a, dim3 grid)
{
    const size_t sm = K1Geo::smem(D ? D : (int)c->d, SHIFT);
    HIPCHK(c, hipFuncSetAttribute((const void *)k1_demod2<D, SHIFT, GEN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    hipLaunchKernelGGL((k1_demod2<D, SHIFT, GEN>), grid, dim3(256), sm, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return 0;
}

/* the default switches' first pass runs the kernel without the option paths (k1_demod2<.., GEN = false>) */
template <int D, bool SHIFT> int launch_k1v2(wmbus_ctx *c, const K1Args &a, dim3 grid)
{
    const uint32_t need = WM_F_ACCURATE | WM_F_T1C1 | WM_F_S1, never = WM_F_APPROX1 | WM_F_APPROX2;
    if (D != 0 && a.relist == nullptr && (c->flags & need) == need && !(c->flags & never)) return launch_k1v3<D, SHIFT, false>(c, a, grid);
    return launch_k1v3<D, SHIFT, true>(c, a, grid);
}

int launch_k1_ppf(wmbus_ctx *c, const K1Args &a, dim3 grid)
{
    const size_t sm = K1PpfGeo::smem();
    HIPCHK(c, hipFuncSetAttribute((const void *)k1_demod_ppf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    hipLaunchKernelGGL(k1_demod_ppf, grid, dim3(256), sm, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return 0;
}

/* The kernels' arithmetic (table-driven atan2 included) on arbitrary operand pairs. */
__global__ void k_selftest(const float *a, const float *b, float *o_sqrt, float *o_div, float *o_atan2, float *o_disc, uint32_t n)
{
    __shared__ float tab[WM_ATAN_TAB_WORDS];
    for (int k = threadIdx.x; k < WM_ATAN_TAB_WORDS; k += blockDim.x) wm_atan_tab_word(k, tab);
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_sqrt[i] = wm_sqrt_dom(fabsf(a[i]));
    o_div[i] = wm_div_dom(a[i], b[i]);
    o_atan2[i] = wm_atan2f_tab(a[i], b[i], tab);
    o_disc[i] = wm_discriminator_tab(a[i], b[i], b[(i + 1) % n], a[(i + 1) % n], tab);
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

void wmbus_default_cfg(wmbus_cfg *cfg)
{
    memset(cfg, 0, sizeof *cfg);
    cfg->decimation = 2; cfg->accurate_atan = 1; cfg->t1c1_enabled = 1; cfg->s1_enabled = 1;
    cfg->rla_enabled = 1; cfg->time2_enabled = 1; cfg->n_streams = 1;
    cfg->max_push_bytes = 4u << 20;
}

int wmbus_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *wmbus_last_error(const wmbus_ctx *ctx) { return ctx ? ctx->err : "null context"; }

void wmbus_close(wmbus_ctx *c)
{
    if (!c) return;
    if (c->stream) hipStreamSynchronize(c->stream);
    void *dev[] = {c->d_spill, c->d_chain, c->d_nchain, c->d_list2, c->d_sync_seen[0], c->d_sync_seen[1], c->d_first_bad, c->d_ckpt, c->d_in, c->d_dphi, c->d_rssi, c->d_bits, c->d_lut, c->d_ema_head, c->d_ema_tail, c->d_ema_carry,
                   c->d_chips[0], c->d_chips[1], c->d_counts[0], c->d_counts[1], c->d_st_start[0], c->d_st_start[1],
                   c->d_st_final[0], c->d_st_final[1], c->d_st_carry[0], c->d_st_carry[1], c->d_list, c->d_scalars,
                   c->d_hits, c->d_pending, c->d_hdr, c->d_words};
    for (void *p : dev) if (p) hipFree(p);
    void *host[] = {c->h_scalars, c->h_hdr, c->h_words, c->h_pending};
    for (void *p : host) if (p) hipHostFree(p);
    for (auto &e : c->ev) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int wmbus_open(const wmbus_cfg *cfg, wmbus_ctx **out)
{
    if (!cfg || !out) return WMBUS_EINVAL;
    *out = nullptr;
    wmbus_ctx *c = new wmbus_ctx();
    c->cfg = *cfg;
    auto bail = [&](int code) { *out = c; return code; };   /* caller reads the message, then closes */

    if (cfg->decimation < 1 || cfg->decimation > WM_MAX_DECIM) return bail(fail(c, WMBUS_EINVAL, "decimation must be 1..%u", WM_MAX_DECIM));
    if (cfg->n_streams < 1) return bail(fail(c, WMBUS_EINVAL, "n_streams must be >= 1"));
    if (cfg->max_push_bytes < WMBUS_BLOCK_BYTES || cfg->max_push_bytes % WMBUS_BLOCK_BYTES)
        return bail(fail(c, WMBUS_EINVAL, "max_push_bytes must be a positive multiple of 4096"));
    if (cfg->prefilter != WMBUS_PREFILTER_BOXCAR && cfg->prefilter != WMBUS_PREFILTER_POLYPHASE)
        return bail(fail(c, WMBUS_EINVAL, "prefilter must be WMBUS_PREFILTER_BOXCAR or WMBUS_PREFILTER_POLYPHASE"));
    if (cfg->atan_mode < WMBUS_ATAN_LIBM || cfg->atan_mode > WMBUS_ATAN_APPROX2)
        return bail(fail(c, WMBUS_EINVAL, "atan_mode must be WMBUS_ATAN_LIBM, WMBUS_ATAN_APPROX1 or WMBUS_ATAN_APPROX2"));
    if (cfg->prefilter == WMBUS_PREFILTER_POLYPHASE && (cfg->decimation != 2 || cfg->simultaneous))
        return bail(fail(c, WMBUS_EINVAL, "the polyphase pre-filter is the 1.6 MS/s design of rtl_wmbus.c:258-294: decimation 2, no -s"));
    if (wmbus_device_count() <= cfg->device) return bail(fail(c, WMBUS_ENODEVICE, "no HIP device %d (this library has no CPU fallback)", cfg->device));
    if (hipSetDevice(cfg->device) != hipSuccess) return bail(fail(c, WMBUS_EDEVICE, "hipSetDevice(%d) failed", cfg->device));

    c->d = cfg->decimation; c->S = cfg->n_streams;
    c->C[1] = cfg->seg_len ? cfg->seg_len : 32768u;    /* in-box A/B with the fused framer launches: 65536 -3 %, 16384 -4 % */
    c->C[0] = cfg->rla_seg_len ? cfg->rla_seg_len : 8192u;
    for (int a = 0; a < 2; a++)
        if (c->C[a] < 1024u || c->C[a] > (1u << 20) || (c->C[a] & (c->C[a] - 1)))
            return bail(fail(c, WMBUS_EINVAL, "seg_len / rla_seg_len must be powers of two in [1024, 1048576]"));
    c->cfg.warmup_t1c1 = cfg->warmup_t1c1 ? (cfg->warmup_t1c1 + 31u) & ~31u : 12288u;   /* whole 32-sample blocks */
    c->cfg.warmup_s1 = cfg->warmup_s1 ? (cfg->warmup_s1 + 31u) & ~31u : 24576u;
    c->cfg.rla_lookback = cfg->rla_lookback ? (cfg->rla_lookback + 31u) & ~31u : 1024u;
    c->flags = (cfg->atan_mode == WMBUS_ATAN_APPROX1 ? WM_F_APPROX1 : cfg->atan_mode == WMBUS_ATAN_APPROX2 ? WM_F_APPROX2 : 0) |
               (cfg->simultaneous ? WM_F_SHIFT : 0) | (cfg->accurate_atan ? WM_F_ACCURATE : 0) | (cfg->remove_dc ? WM_F_DC : 0) |
               (cfg->t1c1_enabled ? WM_F_T1C1 : 0) | (cfg->s1_enabled ? WM_F_S1 : 0) | (cfg->rla_enabled ? WM_F_RLA : 0) |
               (cfg->time2_enabled ? WM_F_T2A : 0);
    c->T = (uint32_t)WM_K1_TILE2;
    const uint32_t T = c->T;
    const uint64_t max_samples = cfg->max_push_bytes / 2;
    c->ntiles_cap = (uint32_t)((max_samples / c->d + 1 + 8 + T - 1) / T);
    c->Mcap = (c->ntiles_cap * T + 255) / 256 * 256;     /* whole tiles (partial tiles still store full runs); slicer-word rows 32-byte aligned */
    for (int a = 0; a < 2; a++) c->nseg_cap[a] = (c->Mcap + c->C[a] - 1) / c->C[a];
    c->cap[1] = c->C[1] / 4 + 8;   /* time2: the lock logic needs >= 4 samples per chip */
    /* run-length: a primary region of half a chip per sample (four times the nominal eight samples per chip); the
     * reference's bit-length tracker has no floor -- switch combinations (-d 3 -s -o -a on a capture with both modes)
     * and interferers drag it to a third of a sample per chip for a while (3.1 chips per sample seen, found by the host
     * emulation campaigns), and exact silence ends in one long run -- such segments continue in the spill arena
     * (WmSpill, wm_dev.h); a push never fails for want of chip storage */
    c->cap[0] = (c->C[0] / 2 + 8 + 7) / 8 * 8;
    /* K1 stages whole tiles: the partial last tile of a push reads up to (tile + halo) x d input samples past the
     * staged bytes (never used: they only feed outputs beyond M) -- the row must hold them */
    c->in_stride = (WM_HIST_BYTES + cfg->max_push_bytes + 2ull * (WM_K1_TILE2 + WM_K1_HALO + 16) * WM_MAX_DECIM + WM_IN_SLACK + 255) / 256 * 256;

    const uint64_t rows = 2ull * c->S;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto &ev : c->ev) A(hipEventCreate(&ev));
    A(dalloc(&c->d_in, (size_t)c->in_stride * c->S));
    A(dalloc(&c->d_dphi, (size_t)rows * c->Mcap));
    A(dalloc(&c->d_rssi, (size_t)rows * c->Mcap));
    A(dalloc(&c->d_bits, (size_t)rows * (c->Mcap / 32)));
    A(dalloc(&c->d_lut, (size_t)2 * 32 * WM_MAX_DECIM));
    A(dalloc(&c->d_ema_head, (size_t)rows * c->ntiles_cap));
    A(dalloc(&c->d_ema_tail, (size_t)rows * c->ntiles_cap));
    A(dalloc(&c->d_ema_carry, (size_t)rows));
    A(dalloc(&c->d_first_bad, (size_t)rows));
    const size_t stw[2] = {sizeof(WmRlaState), sizeof(WmClkState)};
    for (int a = 0; a < 2; a++) {
        A(dalloc(&c->d_chips[a], (size_t)rows * c->nseg_cap[a] * c->cap[a]));
        A(dalloc(&c->d_counts[a], (size_t)rows * c->nseg_cap[a]));
        A(dalloc(&c->d_sync_seen[a], (size_t)rows * c->nseg_cap[a]));
        A(hipMalloc(&c->d_st_start[a], (size_t)rows * c->nseg_cap[a] * stw[a]));
        A(hipMalloc(&c->d_st_final[a], (size_t)rows * c->nseg_cap[a] * stw[a]));
        A(hipMalloc(&c->d_st_carry[a], (size_t)rows * stw[a]));
    }
    {
        const uint64_t dec_total_ = (uint64_t)c->S * c->Mcap;
        const uint64_t want = cfg->spill_words ? cfg->spill_words : std::max<uint64_t>(1u << 20, dec_total_ / 16);
        c->spill_words = (uint32_t)std::min<uint64_t>((want + WM_SPILL_CHUNK - 1) / WM_SPILL_CHUNK * WM_SPILL_CHUNK, 0xFFFF0000u);
        A(dalloc(&c->d_spill, (size_t)c->spill_words));
        A(dalloc(&c->d_chain, (size_t)rows * c->nseg_cap[0] * WM_SPILL_LEVELS));
        A(dalloc(&c->d_nchain, (size_t)rows * c->nseg_cap[0] + 1));          /* + the arena's bump counter */
    }
    A(dalloc(&c->d_list, (size_t)rows * std::max(c->nseg_cap[0], c->nseg_cap[1])));
    A(dalloc(&c->d_list2, (size_t)rows * c->nseg_cap[0]));
    c->nck = c->C[1] / WM_CK_SAMPLES ? c->C[1] / WM_CK_SAMPLES - 1 : 0;
    A(dalloc(&c->d_ckpt, std::max<size_t>(16, (size_t)rows * c->nseg_cap[1] * c->nck * 16)));
    A(dalloc(&c->d_scalars, (size_t)SC_COUNT));
    const uint64_t dec_total = (uint64_t)c->S * c->Mcap;
    c->hdr_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(65536, dec_total / 1024), 1u << 24);
    c->hits_cap = c->hdr_cap;
    c->words_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 20, dec_total / 8), 1u << 29);
    A(dalloc(&c->d_hits, (size_t)c->hits_cap));
    A(dalloc(&c->d_pending, (size_t)4 * c->S));
    A(dalloc(&c->d_hdr, (size_t)c->hdr_cap));
    A(dalloc(&c->d_words, (size_t)c->words_cap));
    A(hipHostMalloc((void **)&c->h_scalars, SC_COUNT * sizeof(uint32_t)));
    A(hipHostMalloc((void **)&c->h_hdr, (size_t)c->hdr_cap * sizeof(WmBurstHdr)));
    A(hipHostMalloc((void **)&c->h_words, (size_t)c->words_cap * sizeof(uint32_t)));
    A(hipHostMalloc((void **)&c->h_pending, (size_t)4 * c->S * sizeof(uint32_t)));
    if (e != hipSuccess) return bail(fail(c, e == hipErrorOutOfMemory ? WMBUS_ENOMEM : WMBUS_EDEVICE, "allocation failed: %s", hipGetErrorString(e)));

    /* initial state = the reference's zero-initialised statics */
    A(hipMemsetAsync(c->d_ema_carry, 0, rows * sizeof(float), c->stream));
    A(hipMemsetAsync(c->d_first_bad, 0xFF, rows * sizeof(uint32_t), c->stream));
    A(hipMemsetAsync(c->d_st_carry[1], 0, rows * sizeof(WmClkState), c->stream));
    {
        std::vector<WmRlaState> init(rows, WmRlaState{0, 8 * 256, 0, 0u, 0u, 0u, 24, 24});   /* rtl_wmbus.c:628-637,717-726 */
        A(hipMemcpyAsync(c->d_st_carry[0], init.data(), rows * sizeof(WmRlaState), hipMemcpyHostToDevice, c->stream));
        A(hipStreamSynchronize(c->stream));
    }
    A(hipMemsetAsync(c->d_scalars, 0, SC_COUNT * sizeof(uint32_t), c->stream));
    A(hipMemsetAsync(c->d_pending, 0, 4 * c->S * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_fill, dim3(c->S), dim3(256), 0, c->stream, c->d_in, c->in_stride, (uint32_t)c->in_stride, (uint8_t)128);
    /* frequency-translation LUT, built with the host libm exactly like rtl_wmbus.c:974-993 */
    {
        const int fs_khz = (int)c->d * 800;
        const size_t n_max = (size_t)(fs_khz / 25);
        std::vector<float> lut(2 * 32 * WM_MAX_DECIM, 0.f);
        for (size_t n = 0; n < n_max; n++) {
            const double phi = (2. * M_PI * (25 * (double)n)) / fs_khz;
            lut[n] = cosf(phi);
            lut[32 * WM_MAX_DECIM + n] = -sinf(phi);
        }
        A(hipMemcpyAsync(c->d_lut, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        A(hipStreamSynchronize(c->stream));
    }
    if (e != hipSuccess) return bail(fail(c, WMBUS_EDEVICE, "initialisation failed: %s", hipGetErrorString(e)));

    c->decs.resize((size_t)c->S * 4);
    for (uint32_t s = 0; s < c->S; s++)
        for (int ch = 0; ch < 2; ch++)
            for (int al = 0; al < 2; al++) {
                HostDecoder &hd = c->decs[((size_t)s * 2 + ch) * 2 + al];
                wm_decoder_init(&hd.dec, ch ? WM_MODE_S1 : WM_MODE_T1C1);
                hd.owed = 0;
            }
    *out = c;
    return WMBUS_OK;
}

void *wmbus_alloc_pinned(size_t nbytes)
{
    void *p = nullptr;
    return hipHostMalloc(&p, nbytes) == hipSuccess ? p : nullptr;
}

void wmbus_free_pinned(void *p) { if (p) hipHostFree(p); }

void *wmbus_device_input(wmbus_ctx *c, unsigned stream)
{
    if (!c || stream >= c->S) return nullptr;
    return c->d_in + (size_t)stream * c->in_stride + WM_HIST_BYTES;
}

int wmbus_stage(wmbus_ctx *c, unsigned stream, const uint8_t *cu8, size_t nbytes)
{
    if (!c || stream >= c->S || !cu8) return WMBUS_EINVAL;
    if (nbytes > c->cfg.max_push_bytes || nbytes % WMBUS_BLOCK_BYTES) return fail(c, WMBUS_EINVAL, "stage: nbytes must be a multiple of 4096 and <= max_push_bytes");
    HIPCHK(c, hipMemcpyAsync(wmbus_device_input(c, stream), cu8, nbytes, hipMemcpyHostToDevice, c->stream));
    return WMBUS_OK;
}

static int run_segments(wmbus_ctx *c, int algo, K2Args a, float *ms, unsigned *reruns)
{
    const WmPush &g = a.g;
    const uint32_t lanes = 2u * g.nseg[algo] * g.S;
    const uint32_t words = (algo == WMBUS_ALGO_RLA ? sizeof(WmRlaState) : sizeof(WmClkState)) / 4;
    auto launch = [&](const uint32_t *list, uint32_t n) {
        a.list = list; a.n_lanes = n;
        if (algo == WMBUS_ALGO_RLA) hipLaunchKernelGGL(k2_rla, dim3((n + 64 * WM_RLA_WPB - 1) / (64 * WM_RLA_WPB)), dim3(64 * WM_RLA_WPB), 0, c->stream, a);
        else if (c->flags & WM_F_DC) hipLaunchKernelGGL(k2_clock<true>, dim3((n + 64 * WM_CLK_WPB - 1) / (64 * WM_CLK_WPB)), dim3(64 * WM_CLK_WPB), 0, c->stream, a);
        else hipLaunchKernelGGL(k2_clock<false>, dim3((n + 64 * WM_CLK_WPB - 1) / (64 * WM_CLK_WPB)), dim3(64 * WM_CLK_WPB), 0, c->stream, a);
    };
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    launch(nullptr, lanes);
    for (unsigned round = 0;; round++) {
        HIPCHK(c, hipMemsetAsync(c->d_scalars + SC_NLIST, 0, sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(k2_verify, dim3((lanes + 255) / 256), dim3(256), 0, c->stream, g, (uint32_t)algo,
                           (const uint32_t *)a.st_start, (const uint32_t *)a.st_final, words, c->d_list, c->d_scalars + SC_NLIST);
        HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, SC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint32_t n = c->h_scalars[SC_NLIST];
        if (n == 0) break;
        if (round > g.nseg[algo] + 1) return fail(c, WMBUS_EDEVICE, "segment verification did not converge");
        *reruns += n;
        launch(c->d_list, n);
    }
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev[0], c->ev[1]));
    /* carry the exact end state into the next push */
    const uint32_t rows = 2u * g.S;
    hipLaunchKernelGGL(k_carry, dim3((rows + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)a.st_final,
                       (uint32_t *)a.st_carry, words, rows, g.nseg_cap[algo], g.nseg[algo]);
    HIPCHK(c, hipGetLastError());
    return 0;
}

/* Both framers when the slicer words do not depend on filter state (no DC remover): the clock
 * kernel's first pass, then launches that carry the clock re-run lanes AND the run-length framer
 * (main pass first, its own re-run lists afterwards) until both have converged. */
static int run_framers_fused(wmbus_ctx *c, K2Args clk, K2Args rla)
{
    const WmPush &g = clk.g;
    const uint32_t lanes_c = 2u * g.nseg[1] * g.S, lanes_r = 2u * g.nseg[0] * g.S;
    const uint32_t wc = sizeof(WmClkState) / 4, wr = sizeof(WmRlaState) / 4, B = 64 * WM_CLK_WPB, Br = 64 * WM_RLA_WPB;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    clk.list = nullptr; clk.n_lanes = lanes_c;
    hipLaunchKernelGGL(k2_clock<false>, dim3((lanes_c + B - 1) / B), dim3(B), 0, c->stream, clk);
    bool clk_done = false, rla_started = false, rla_done = false;
    for (unsigned round = 0;; round++) {
        HIPCHK(c, hipMemsetAsync(c->d_scalars + SC_NLIST, 0, sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_scalars + SC_NLIST2, 0, sizeof(uint32_t), c->stream));
        if (!clk_done)
            hipLaunchKernelGGL(k2_verify, dim3((lanes_c + 255) / 256), dim3(256), 0, c->stream, g, (uint32_t)WMBUS_ALGO_T2A,
                               (const uint32_t *)clk.st_start, (const uint32_t *)clk.st_final, wc, c->d_list, c->d_scalars + SC_NLIST);
        if (rla_started && !rla_done)
            hipLaunchKernelGGL(k2_verify, dim3((lanes_r + 255) / 256), dim3(256), 0, c->stream, g, (uint32_t)WMBUS_ALGO_RLA,
                               (const uint32_t *)rla.st_start, (const uint32_t *)rla.st_final, wr, c->d_list2, c->d_scalars + SC_NLIST2);
        HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, SC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint32_t n_c = clk_done ? 0u : c->h_scalars[SC_NLIST], n_r = c->h_scalars[SC_NLIST2];
        if (!clk_done && n_c == 0) { clk_done = true; HIPCHK(c, hipEventRecord(c->ev[8], c->stream)); }
        if (rla_started && n_r == 0) rla_done = true;
        if (clk_done && rla_done) break;
        if (round > std::max(g.nseg[0], g.nseg[1]) + 2) return fail(c, WMBUS_EDEVICE, "segment verification did not converge");
        K2Args ca = clk, ra = rla;
        ca.list = c->d_list; ca.n_lanes = n_c;
        c->tim.clock_reruns += n_c;
        if (!rla_started) { ra.list = nullptr; ra.n_lanes = lanes_r; rla_started = true; }
        else { ra.list = c->d_list2; ra.n_lanes = n_r; c->tim.rla_reruns += n_r; }
        const uint32_t cb = (n_c + 63u) / 64u, rb = (ra.n_lanes + Br - 1) / Br;     /* one clock wave per block here */
        hipLaunchKernelGGL(k2_clock_rla, dim3(cb + rb), dim3(Br), 0, c->stream, ca, ra, cb);
    }
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    HIPCHK(c, hipEventElapsedTime(&c->tim.clock_ms, c->ev[0], c->ev[8]));      /* until the clock kernel had converged */
    HIPCHK(c, hipEventElapsedTime(&c->tim.rla_ms, c->ev[8], c->ev[1]));        /* what the run-length framer added   */
    const uint32_t rows = 2u * g.S;
    hipLaunchKernelGGL(k_carry, dim3((rows + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)clk.st_final,
                       (uint32_t *)clk.st_carry, wc, rows, g.nseg_cap[1], g.nseg[1]);
    hipLaunchKernelGGL(k_carry, dim3((rows + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)rla.st_final,
                       (uint32_t *)rla.st_carry, wr, rows, g.nseg_cap[0], g.nseg[0]);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int wmbus_process(wmbus_ctx *c, size_t nbytes)
{
    if (!c) return WMBUS_EINVAL;
    if (nbytes == 0 || nbytes > c->cfg.max_push_bytes || nbytes % WMBUS_BLOCK_BYTES)
        return fail(c, WMBUS_EINVAL, "process: nbytes must be a positive multiple of 4096 and <= max_push_bytes");
    if (c->in_flight) return fail(c, WMBUS_EINVAL, "process: previous push not collected");
    if (c->poisoned) return fail(c, WMBUS_EDEVICE, "process: an earlier internal error left this context unusable; close it");
    c->tim = wmbus_timing{};
    const uint32_t n_new = (uint32_t)(nbytes / 2);
    WmPush g{};
    g.in = c->d_in; g.in_stride = c->in_stride; g.n0 = c->n0; g.m0 = c->n0 / c->d;
    g.n_new = n_new; g.M = (uint32_t)((c->n0 + n_new) / c->d - g.m0);
    g.Mcap = c->Mcap; g.d = c->d; g.S = c->S; g.lut_n = 32 * c->d;
    g.lut_phase0 = (uint32_t)((13ull * (c->n0 % g.lut_n)) % g.lut_n);
    g.flags = c->flags;
    for (int al = 0; al < 2; al++) {
        g.seg_len[al] = c->C[al]; g.nseg[al] = (g.M + c->C[al] - 1) / c->C[al]; g.nseg_cap[al] = c->nseg_cap[al]; g.cap[al] = c->cap[al];
    }
    g.warm[0] = c->cfg.warmup_t1c1; g.warm[1] = c->cfg.warmup_s1; g.lookback = c->cfg.rla_lookback;
    g.sp.arena = c->d_spill; g.sp.arena_words = c->spill_words; g.sp.chain = c->d_chain; g.sp.nchain = c->d_nchain;
    g.sp.used = c->d_nchain + (size_t)2 * c->S * c->nseg_cap[0];
    c->last = g; c->have_last = true;
    c->n_hdr = c->n_words = 0;

    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    if (g.M > 0) {
        HIPCHK(c, hipMemsetAsync(c->d_scalars, 0, SC_COUNT * sizeof(uint32_t), c->stream));
        /* K1 */
        const uint32_t T = c->T, ntiles = (g.M + T - 1) / T;
        K1Args k1{g, c->d_dphi, c->d_rssi, c->d_lut, c->d_lut + 32 * WM_MAX_DECIM, c->d_ema_head, c->d_ema_tail, ntiles, c->d_scalars + SC_ERR,
                  nullptr, c->d_ema_carry};
        /* Contexts of one process take turns in the demodulation kernel: it fills the GPU on its own
         * (VALU bound), so two of them side by side only time-slice, while one of them beside the
         * other contexts' latency- and memory-bound framer kernels is complementary.  The turn also
         * makes the event pair below measure the kernel rather than its share of the GPU. */
        static std::mutex k1_turn[16];
        std::unique_lock<std::mutex> turn(k1_turn[c->cfg.device & 15], std::defer_lock);
        static const bool take_turns = !(getenv("WMBUS_K1_TURNS") && atoi(getenv("WMBUS_K1_TURNS")) == 0);
        if (take_turns) {
            const auto t_wait = std::chrono::steady_clock::now();
            turn.lock();
            c->tim.turn_wait_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_wait).count();
        }
        HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
        auto launch = [&](uint32_t n_list) -> int {                  /* n_list = 0: all tiles; else the repair list */
            const bool sh = c->flags & WM_F_SHIFT;
            k1.relist = n_list ? c->d_list : nullptr;
            const dim3 grid = n_list ? dim3(n_list, 1) : dim3(ntiles, c->S);
            if (c->cfg.prefilter == WMBUS_PREFILTER_POLYPHASE) return launch_k1_ppf(c, k1, grid);
            if (c->d == 2) return sh ? launch_k1v2<2, true>(c, k1, grid) : launch_k1v2<2, false>(c, k1, grid);
            if (c->d == 3) return sh ? launch_k1v2<3, true>(c, k1, grid) : launch_k1v2<3, false>(c, k1, grid);
            if (c->d == 4) return sh ? launch_k1v2<4, true>(c, k1, grid) : launch_k1v2<4, false>(c, k1, grid);
            if (c->d == 5) return sh ? launch_k1v2<5, true>(c, k1, grid) : launch_k1v2<5, false>(c, k1, grid);
            return sh ? launch_k1v2<0, true>(c, k1, grid) : launch_k1v2<0, false>(c, k1, grid);      /* any other rate */
        };
        int rc = launch(0);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
        /* the turn ends with the kernel: the step time of n contexts is n times the time the
         * turn is held */
        if (take_turns) { HIPCHK(c, hipEventSynchronize(c->ev[4])); turn.unlock(); }
        /* EMA hand-offs between tiles; an uncertified tile is re-run sequentially from its
         * predecessor's exact tail (which may uncover the next one): exact by construction */
        for (unsigned round = 0;; round++) {
            hipLaunchKernelGGL(k1_verify, dim3((2 * c->S + 63) / 64, ntiles), dim3(64), 0, c->stream, c->d_ema_head, c->d_ema_tail,
                               c->d_ema_carry, ntiles, 2 * c->S, c->d_first_bad);
            hipLaunchKernelGGL(k1_collect, dim3((2 * c->S + 63) / 64), dim3(64), 0, c->stream, c->d_first_bad, ntiles, 2 * c->S, c->S,
                               c->d_list, c->d_scalars + SC_NLIST);
            HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, SC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const uint32_t n = c->h_scalars[SC_NLIST];
            if (n == 0) break;
            if (round > ntiles + 1) return fail(c, WMBUS_EDEVICE, "RSSI filter hand-off repair did not converge");
            c->tim.ema_retries += n;
            rc = launch(n);
            if (rc) return rc;
            HIPCHK(c, hipMemsetAsync(c->d_scalars + SC_NLIST, 0, sizeof(uint32_t), c->stream));
        }
        hipLaunchKernelGGL(k1_commit, dim3((2 * c->S + 63) / 64), dim3(64), 0, c->stream, c->d_ema_tail, c->d_ema_carry, ntiles, 2 * c->S);

        /* K2: clock recovery + time2 framer (also produces the slicer bits the RLA needs) */
        K2Args k2{};
        k2.g = g; k2.dphi = c->d_dphi; k2.rssi = c->d_rssi; k2.bits = c->d_bits;
        k2.err = c->d_scalars + SC_ERR;
        for (int al = 0; al < 2; al++)
            HIPCHK(c, hipMemsetAsync(c->d_sync_seen[al], 0, (size_t)2 * c->S * c->nseg_cap[al] * sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_nchain, 0, ((size_t)2 * c->S * c->nseg_cap[0] + 1) * sizeof(uint32_t), c->stream));   /* spill chains + bump counter */
        k2.ckpt = c->d_ckpt; k2.nck = c->nck;
        K2Args ka = k2, kr = k2;
        ka.algo = WMBUS_ALGO_T2A;
        ka.chips = c->d_chips[1]; ka.counts = c->d_counts[1]; ka.sync_seen = c->d_sync_seen[1];
        ka.st_start = c->d_st_start[1]; ka.st_final = c->d_st_final[1]; ka.st_carry = c->d_st_carry[1];
        kr.algo = WMBUS_ALGO_RLA;
        kr.chips = c->d_chips[0]; kr.counts = c->d_counts[0]; kr.sync_seen = c->d_sync_seen[0];
        kr.st_start = c->d_st_start[0]; kr.st_final = c->d_st_final[0]; kr.st_carry = c->d_st_carry[0];
        static const bool fuse = !(getenv("WMBUS_FUSE_FRAMERS") && atoi(getenv("WMBUS_FUSE_FRAMERS")) == 0);   /* tuning aid */
        if ((c->flags & WM_F_RLA) && !(c->flags & WM_F_DC) && fuse) {
            rc = run_framers_fused(c, ka, kr);
            if (rc) return rc;
        } else {
            rc = run_segments(c, WMBUS_ALGO_T2A, ka, &c->tim.clock_ms, &c->tim.clock_reruns);
            if (rc) return rc;
            if (c->flags & WM_F_RLA) {
                rc = run_segments(c, WMBUS_ALGO_RLA, kr, &c->tim.rla_ms, &c->tim.rla_reruns);
                if (rc) return rc;
            } else {
                HIPCHK(c, hipMemsetAsync(c->d_counts[0], 0, (size_t)2 * c->S * c->nseg_cap[0] * sizeof(uint32_t), c->stream));
            }
        }

        /* K3: access-code hits of the settled chip streams, then bursts */
        HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
        hipLaunchKernelGGL(k_sum_counts, dim3(std::min(64u, (2u * (c->nseg_cap[0] + c->nseg_cap[1]) * c->S + 255u) / 256u)), dim3(256), 0, c->stream, g,
                           c->d_counts[0], c->d_counts[1], c->d_scalars + SC_CHIPS);
        hipLaunchKernelGGL(k3_scan, dim3((2u * (g.nseg[0] + g.nseg[1]) * g.S + 255u) / 256u), dim3(256), 0, c->stream, g, c->d_chips[0], c->d_chips[1],
                           c->d_counts[0], c->d_counts[1], c->d_sync_seen[0], c->d_sync_seen[1], c->d_hits, c->d_scalars + SC_NHITS,
                           c->hits_cap, c->d_scalars + SC_ERR);
        HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, SC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint32_t n_hits = std::min(c->h_scalars[SC_NHITS], c->hits_cap);
        for (uint32_t s = 0; s < c->S; s++)
            for (int ch = 0; ch < 2; ch++)
                for (int al = 0; al < 2; al++)
                    c->h_pending[(al * 2 + ch) * c->S + s] = c->decs[((size_t)s * 2 + ch) * 2 + al].owed;
        HIPCHK(c, hipMemcpyAsync(c->d_pending, c->h_pending, 4 * c->S * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        K3Args k3{};
        k3.g = g; k3.rssi = c->d_rssi;
        k3.chips[0] = c->d_chips[0]; k3.chips[1] = c->d_chips[1]; k3.counts[0] = c->d_counts[0]; k3.counts[1] = c->d_counts[1];
        k3.hits = c->d_hits; k3.n_hits = c->d_scalars + SC_NHITS; k3.hits_cap = c->hits_cap; k3.pending = c->d_pending;
        k3.hdr = c->d_hdr; k3.hdr_cap = c->hdr_cap; k3.words = c->d_words; k3.words_cap = c->words_cap;
        k3.n_hdr = c->d_scalars + SC_NHDR; k3.n_words = c->d_scalars + SC_NWORDS; k3.err = c->d_scalars + SC_ERR;
        {
            static const uint32_t max_blocks = getenv("WMBUS_K3_BLOCKS") ? (uint32_t)atoi(getenv("WMBUS_K3_BLOCKS")) : 256u;   /* tuning aid; 128 ... 512 are within 2 % of each other */
            const uint32_t n_items = 4 * c->S + n_hits;
            hipLaunchKernelGGL(k3_bursts, dim3(std::max(1u, std::min((n_items + 3u) / 4u, max_blocks))), dim3(256), 0, c->stream, k3, n_items);
        }
        HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, SC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint32_t err = c->h_scalars[SC_ERR];
        if (err & WM_ERR_CHIP_OVERFLOW) {            /* a time2 region over its proven bound: a defect, not an input property */
            c->poisoned = true;
            return fail(c, WMBUS_EDEVICE, "internal error: time2 chip region overflow (err=%u)", err);
        }
        /* storage exhausted (spill arena / burst arena): chips or candidate bursts were dropped; every carried state
         * is exact, so the stream goes on -- reported, never fatal (the reference never gives up either) */
        c->tim.warnings = ((err & WM_ERR_CHIP_TRUNC) ? WMBUS_WARN_CHIPS_DROPPED : 0u) | ((err & WM_ERR_BURST_OVERFLOW) ? WMBUS_WARN_BURSTS_DROPPED : 0u);
        c->n_hdr = c->h_scalars[SC_NHDR]; c->n_words = c->h_scalars[SC_NWORDS];
        for (int al = 0; al < 2; al++) for (int ch = 0; ch < 2; ch++) c->tim.chips[ch][al] = c->h_scalars[SC_CHIPS + al * 2 + ch];
        if (c->n_hdr) HIPCHK(c, hipMemcpyAsync(c->h_hdr, c->d_hdr, (size_t)c->n_hdr * sizeof(WmBurstHdr), hipMemcpyDeviceToHost, c->stream));
        if (c->n_words) HIPCHK(c, hipMemcpyAsync(c->h_words, c->d_words, (size_t)c->n_words * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    }
    /* slide the input history: the next push sees the last 4096 staged bytes in front of it */
    hipLaunchKernelGGL(k_roll_history, dim3(c->S), dim3(256), 0, c->stream, c->d_in, c->in_stride, (uint32_t)nbytes);
    HIPCHK(c, hipGetLastError());
    c->n0 += n_new;
    c->in_flight = true;
    return WMBUS_OK;
}

/* Run the persistent packet decoders of the (stream, chain, framer) groups in order[lo, hi). */
static void decode_stream_range(wmbus_ctx *c, const std::vector<uint32_t> &order, size_t lo, size_t hi,
                                std::vector<LineRec> &out, const char *ts_fixed)
{
    char line[1024], ts[64];
    uint32_t seq = 0;
    size_t i = lo;
    while (i < hi) {
        const WmBurstHdr &h0 = c->h_hdr[order[i]];
        HostDecoder &hd = c->decs[((size_t)h0.stream * 2 + h0.chain) * 2 + h0.algo];
        const char *tag = c->cfg.show_algorithm ? (h0.algo == WMBUS_ALGO_RLA ? "rla;" : "t2a;") : "";
        uint64_t next_free = 0;                         /* first chip the decoder has not consumed */
        bool cut = false;
        size_t j = i;
        for (; j < hi; j++) {
            const WmBurstHdr &h = c->h_hdr[order[j]];
            if (h.stream != h0.stream || h.chain != h0.chain || h.algo != h0.algo) break;
            const bool cont = h.flags & 1u;
            if (cont ? hd.owed == 0 : h.chip0 < next_free) continue;   /* the access code passed while the decoder was busy */
            const uint32_t *w = c->h_words + h.word_off;
            int st = cont ? WM_DEC_RECEIVING : WM_DEC_IDLE;
            uint32_t k = 0;
            for (; k < h.n_chips; k++) {
                const uint32_t word = w[k];
                const unsigned val = word & 7u, rssi = (word >> 3) & 0xFFu;
                if ((val & 4u) && st == WM_DEC_RECEIVING) {   /* the run-length framer reset itself: telegram lost */
                    wm_decoder_abort(&hd.dec);
                    st = WM_DEC_IDLE;
                    break;                                     /* this chip may start a burst of its own */
                }
                st = wm_decoder_chip(&hd.dec, val & 3u, rssi);
                if (st == WM_DEC_DONE) {
                    int ok = 0;
                    if (ts_fixed) snprintf(ts, sizeof ts, "%s", ts_fixed); else wm_timestamp(ts, sizeof ts);
                    const size_t n = wm_decoder_format(&hd.dec, tag, ts, rssi, line, sizeof line, &ok);
                    LineRec r; r.sample = h.pos0 + (word >> 11); r.stream = h.stream; r.chain = h.chain; r.algo = h.algo;
                    r.crc_ok = (uint8_t)ok; r.seq = seq++; r.text.assign(line, n);
                    out.push_back(std::move(r));
                    st = WM_DEC_IDLE;
                }
                if (st == WM_DEC_IDLE) { k++; break; }
            }
            next_free = (uint64_t)h.chip0 + k;
            cut = st == WM_DEC_RECEIVING;
            if (cut && h.n_chips != h.avail) {                 /* device under-estimated the burst: a bug */
                uint32_t none = 0;
                c->short_burst.compare_exchange_strong(none, 1u + order[j]);
            }
            hd.owed = cut ? std::max(1u, wm_decoder_chips_owed(&hd.dec)) : 0u;
        }
        i = j;
    }
}

int wmbus_collect(wmbus_ctx *c)
{
    if (!c) return WMBUS_EINVAL;
    c->lines.clear(); c->text.clear();
    c->short_burst.store(0);
    if (!c->in_flight) return WMBUS_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->in_flight = false;
    if (c->last.M > 0) {
        float ms = 0;
        hipEventElapsedTime(&ms, c->ev[3], c->ev[4]); c->tim.demod_ms = ms;
        hipEventElapsedTime(&ms, c->ev[5], c->ev[6]); c->tim.gather_ms = ms;
        hipEventElapsedTime(&ms, c->ev[6], c->ev[7]); c->tim.d2h_ms = ms;
        hipEventElapsedTime(&ms, c->ev[2], c->ev[7]); c->tim.gpu_total_ms = ms;
    }
    c->tim.bursts = c->n_hdr;
    const double t0 = now_ms();

    std::vector<uint32_t> order(c->n_hdr);
    for (uint32_t i = 0; i < c->n_hdr; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const WmBurstHdr &a = c->h_hdr[x], &b = c->h_hdr[y];
        if (a.stream != b.stream) return a.stream < b.stream;
        if (a.chain != b.chain) return a.chain < b.chain;
        if (a.algo != b.algo) return a.algo < b.algo;
        if ((a.flags & 1u) != (b.flags & 1u)) return (a.flags & 1u) > (b.flags & 1u);   /* continuation first */
        return a.chip0 < b.chip0;
    });
    /* drop exact duplicates (a re-run segment may have re-recorded a hit) */
    order.erase(std::unique(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const WmBurstHdr &a = c->h_hdr[x], &b = c->h_hdr[y];
        return a.stream == b.stream && a.chain == b.chain && a.algo == b.algo && a.flags == b.flags && a.chip0 == b.chip0;
    }), order.end());
    unsigned nt = c->cfg.host_threads ? c->cfg.host_threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (order.size() < 4096) nt = 1;
    const char *tsf = c->cfg.fixed_timestamp ? "TS" : nullptr;
    /* more pieces than threads (cut at stream boundaries): the decoders' cost per stream is uneven */
    const unsigned np = nt == 1 ? 1 : 4 * nt;
    std::vector<std::vector<LineRec>> parts(np);
    if (np == 1) decode_stream_range(c, order, 0, order.size(), parts[0], tsf);
    else {
        std::vector<size_t> cut(np + 1, order.size());
        cut[0] = 0;
        for (unsigned t = 1; t < np; t++) {
            size_t p = order.size() * t / np;
            while (p < order.size() && p > 0 && c->h_hdr[order[p]].stream == c->h_hdr[order[p - 1]].stream) p++;
            cut[t] = std::max(p, cut[t - 1]);
        }
        if (!c->pool || c->pool->size() + 1 != nt) c->pool.reset(new WorkerPool(nt - 1));
        c->pool->run(np, [&](unsigned t) { decode_stream_range(c, order, cut[t], cut[t + 1], parts[t], tsf); });
    }
    /* stdout order of the reference: by completing sample, then T1/C1 before S1, run-length before time2 */
    std::vector<LineRec> all;
    for (auto &p : parts) for (auto &r : p) all.push_back(std::move(r));
    std::stable_sort(all.begin(), all.end(), [](const LineRec &a, const LineRec &b) {
        if (a.stream != b.stream) return a.stream < b.stream;
        if (a.sample != b.sample) return a.sample < b.sample;
        if (a.chain != b.chain) return a.chain < b.chain;
        if (a.algo != b.algo) return a.algo < b.algo;
        return a.seq < b.seq;
    });
    for (auto &r : all) {
        wmbus_line l{};
        l.stream = r.stream; l.chain = r.chain; l.algo = r.algo; l.crc_ok = r.crc_ok; l.sample = r.sample;
        l.text_off = (uint32_t)c->text.size(); l.text_len = (uint32_t)r.text.size();
        c->text += r.text;
        c->lines.push_back(l);
    }
    c->tim.host_decode_ms = (float)(now_ms() - t0);
    if (const uint32_t sb = c->short_burst.load()) {       /* formatted once, after the pool has joined */
        const WmBurstHdr &h = c->h_hdr[sb - 1u];
        return fail(c, WMBUS_EDEVICE, "burst too short: stream %u chain %u algo %u chip %u", h.stream, h.chain, h.algo, h.chip0);
    }
    return WMBUS_OK;
}

size_t wmbus_lines(const wmbus_ctx *c, const wmbus_line **lines)
{
    if (!c) return 0;
    if (lines) *lines = c->lines.data();
    return c->lines.size();
}

const char *wmbus_lines_text(const wmbus_ctx *c, size_t *len)
{
    if (!c) return "";
    if (len) *len = c->text.size();
    return c->text.c_str();
}

int wmbus_get_timing(const wmbus_ctx *c, wmbus_timing *t)
{
    if (!c || !t) return WMBUS_EINVAL;
    *t = c->tim;
    return WMBUS_OK;
}

long wmbus_read_tap(wmbus_ctx *c, const char *what, int chain, unsigned stream, void *dst, size_t max_elems)
{
    if (!c || !what || !dst || chain < 0 || chain > 1 || stream >= c->S || !c->have_last) return WMBUS_EINVAL;
    if (!c->cfg.keep_taps) return fail(c, WMBUS_EINVAL, "read_tap: the context was opened without cfg.keep_taps");
    hipStreamSynchronize(c->stream);
    const size_t n = std::min<size_t>(max_elems, c->last.M);
    const size_t row = (size_t)chain * c->S + stream;
    if (!strcmp(what, "dphi")) {
        if (hipMemcpy(dst, c->d_dphi + row * c->Mcap, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return WMBUS_EDEVICE;
    } else if (!strcmp(what, "rssi")) {
        if (hipMemcpy(dst, c->d_rssi + row * c->Mcap, n, hipMemcpyDeviceToHost) != hipSuccess) return WMBUS_EDEVICE;
    } else if (!strcmp(what, "bits")) {
        std::vector<uint32_t> w((n + 31) / 32);
        if (hipMemcpy(w.data(), c->d_bits + row * (c->Mcap / 32), w.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return WMBUS_EDEVICE;
        for (size_t k = 0; k < n; k++) ((uint8_t *)dst)[k] = (w[k >> 5] >> (k & 31)) & 1u;
    } else return WMBUS_EINVAL;
    return (long)n;
}

long wmbus_read_chips(wmbus_ctx *c, int chain, int algo, unsigned stream, uint32_t *dst, uint64_t *pos, size_t max_elems)
{
    if (!c || chain < 0 || chain > 1 || algo < 0 || algo > 1 || stream >= c->S || !c->have_last || !dst) return WMBUS_EINVAL;
    hipStreamSynchronize(c->stream);
    uint32_t *d_dst = nullptr, *d_n = nullptr; uint64_t *d_pos = nullptr;
    if (hipMalloc((void **)&d_dst, max_elems * 4) != hipSuccess || hipMalloc((void **)&d_pos, max_elems * 8) != hipSuccess ||
        hipMalloc((void **)&d_n, 4) != hipSuccess) {
        hipFree(d_dst); hipFree(d_pos); hipFree(d_n);            /* hipFree(nullptr) is a no-op */
        return WMBUS_ENOMEM;
    }
    hipLaunchKernelGGL(k4_flatten, dim3(1), dim3(1), 0, c->stream, c->last, (uint32_t)algo, c->d_chips[algo], c->d_counts[algo],
                       c->d_rssi, c->cap[algo], (uint32_t)chain, stream, d_dst, d_pos, (uint32_t)max_elems, d_n);
    uint32_t n = 0;
    hipStreamSynchronize(c->stream);
    hipMemcpy(&n, d_n, 4, hipMemcpyDeviceToHost);
    const size_t m = std::min<size_t>(n, max_elems);
    hipMemcpy(dst, d_dst, m * 4, hipMemcpyDeviceToHost);
    if (pos) hipMemcpy(pos, d_pos, m * 8, hipMemcpyDeviceToHost);
    hipFree(d_dst); hipFree(d_pos); hipFree(d_n);
    return (long)n;
}

/* Runs the device versions of the exact scalar helpers (wm_exact.h) over n operand pairs so a
 * test can compare them with the host's IEEE / libm results bit for bit. */
int wmbus_selftest_math(int device, const float *a, const float *b, float *o_sqrt, float *o_div, float *o_atan2,
                        float *o_disc, size_t n)
{
    if (wmbus_device_count() <= device || hipSetDevice(device) != hipSuccess) return WMBUS_ENODEVICE;
    float *d[6] = {};
    for (auto &p : d) if (hipMalloc((void **)&p, n * sizeof(float)) != hipSuccess) return WMBUS_ENOMEM;
    hipMemcpy(d[0], a, n * sizeof(float), hipMemcpyHostToDevice);
    hipMemcpy(d[1], b, n * sizeof(float), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], (uint32_t)n);
    float *outs[4] = {o_sqrt, o_div, o_atan2, o_disc};
    int rc = hipDeviceSynchronize() == hipSuccess ? WMBUS_OK : WMBUS_EDEVICE;
    for (int k = 0; k < 4; k++) if (hipMemcpy(outs[k], d[2 + k], n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = WMBUS_EDEVICE;
    for (auto &p : d) hipFree(p);
    return rc;
}

}  // extern "C"
