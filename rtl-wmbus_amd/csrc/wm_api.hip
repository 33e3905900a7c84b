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

#include <sys/time.h>

#include "../../include/wmbus_hip.h"
#include "wm_decoder.h"
#include "wm_dev.h"
#include "wm_kernels.hip"

namespace {

/* one formatted line of a push: where its text lies in its part's buffer (round 6: a std::string per line was a heap allocation per
 * line -- with eight ranks' contexts decoding at once on one host, 64 x 4 threads in malloc, the decode of a context-push took
 * 18 ms instead of 2.8: tools/host_replay.py) */
struct LineRec {
    uint64_t sample; uint32_t stream; uint8_t chain, algo, crc_ok; uint32_t seq; uint32_t part, off, len;
};
struct LinePart { std::vector<LineRec> recs; std::string text; };

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
    uint64_t fed;              /* push in which the decoder last took chips (see the dropped-burst clean-up in wmbus_collect) */
};

}  // namespace

struct wmbus_ctx {
    wmbus_cfg cfg{};
    char err[256] = {0};
    hipStream_t stream = nullptr;                      /* every kernel of the context */
    hipStream_t copy_stream = nullptr;                 /* wmbus_stage's H2D copies (north_star: "pinned hipMemcpyAsync on a side stream") */
    hipEvent_t ev[11] = {};                            /* [9], [10]: the RSSI launch over the listed tiles (timing) */
    hipEvent_t ev_staged = nullptr;                    /* recorded on copy_stream at process(): the push's input is in HBM */
    hipEvent_t ev_turn = nullptr;                       /* behind the main part of this context's K1: the next context's K1 waits for it */
    uint32_t n_win = 1, fill = 0;                      /* input windows (cfg.input_windows) and the one wmbus_stage fills now */
    /* geometry */
    uint32_t d = 2, S = 1, C[2] = {8192, 32768}, Mcap = 0, nseg_cap[2] = {0, 0}, ntiles_cap = 0, T = WM_K1_TILE2;
    uint32_t cap[2] = {0, 0}, flags = 0;
    uint64_t in_stride = 0, n0 = 0;
    size_t staged = 0;
    /* device buffers */
    uint8_t *d_in = nullptr;                           /* [n_win][S][in_stride] */
    uint8_t *d_hist = nullptr;                         /* [S][4096] the input history of the next push (see k_copy_hist) */
    float *d_dphi = nullptr; uint8_t *d_rssi = nullptr; uint32_t *d_bits = nullptr;
    float *d_lut = nullptr;
    float *d_ema_head = nullptr, *d_ema_tail = nullptr, *d_ema_carry = nullptr;   /* carry: [2][rows], see carry_in */
    /* Carried state is double-buffered: a push reads the exact end state of the previous one from half `carry_in` and
     * commits its own into the other half.  The commits are enqueued before the host knows whether every hand-off was
     * certified; if wmbus_collect has to finish rounds, the re-runs of first tiles / segments still find the state
     * they must start from, and the commit is simply made again. */
    uint32_t carry_in = 0; bool committed = false;
    uint32_t *d_chips[2] = {}, *d_counts[2] = {};
    void *d_st_start[2] = {}, *d_st_final[2] = {}, *d_st_carry[2] = {};
    uint32_t *d_list2 = nullptr;                        /* run-length re-run list */
    uint32_t *d_list_ema = nullptr;                     /* RSSI repair list (one entry per row at most); every verifier has its own list:
                                                           collect's slow path starts from what the LAST verify of each kind left */
    uint32_t *d_list = nullptr;
    uint32_t *d_scalars = nullptr;                      /* head of ONE allocation zeroed once per push: scalars | sync_seen[0] | sync_seen[1] |
                                                           nchain + bump counter (separate fills cost a context's chain of launches 40 us each) */
    size_t zero_words = 0;
    uint32_t *d_sync_seen[2] = {};                      /* per framer: [2][S][nseg_cap] access-code chip seen in region */
    uint32_t *d_spill = nullptr, *d_chain = nullptr, *d_nchain = nullptr; uint32_t spill_words = 0;   /* WmSpill (wm_dev.h) */
    unsigned ema_rounds = 1, fr_rounds = 2, rla_rounds = 3;   /* hand-off rounds enqueued unattended, see WM_MAX_ROUNDS; the run-length framer's
                                                          counters start at index 1 (its list rounds: 1 .. rla_rounds - 1) */
    unsigned rla_fin = 3;                               /* this push: index of the run-length framer's last (unattended) verification */
    bool opt_rounds = true;                             /* hand-off rounds are enqueued unattended (cfg.rounds_on_host = 0) */
    bool poisoned = false, gpu_decode = true;                              /* an internal error left the carried state undefined */
    uint32_t *d_first_bad = nullptr;                    /* [2][S] first uncertified EMA tile of a row, or ~0 */
    /* RSSI on demand (wm_k1_demod.h): the first pass leaves the RSSI out, k3_spans lists the tiles whose RSSI is read, an
     * RS = 2 launch computes those.  Off for contexts with debug taps (they show every sample's RSSI) and for the option
     * kernels; cfg.rssi_full turns it off for A/Bs. */
    bool rs_od = false, rs_full_now = false;            /* rs_full_now: this push has fallen back to the full pass */
    bool rs_this = false;                               /* this push runs on demand (a context whose bursts cover most of its tiles takes the full pass for a while) */
    unsigned rs_pause = 0;                              /* pushes left before on demand is tried again */
    bool k1_big = false;                                /* the first pass without the RSSI runs on 2000-sample tiles of 512 threads (decimation 2, no -s) */
    uint32_t k1_tpb = 1;                                /* tiles per block of the first pass without the RSSI (cfg.k1_tiles_per_block) */
    unsigned clk_form = 4;                              /* cfg.clock_waves: 4 systolic (wm_k2_clock_sys.h), 1 rounds 1-5's one wave per lane group (wm_k2_clock.h) */
    uint32_t k1_tail_pm = 60;                           /* per mille of a push's tiles behind the early hand-over of the K1 turn (enqueue_front_impl) */
    uint32_t *d_rs_flags = nullptr, *d_rs_list = nullptr;   /* [ntiles_cap][S] chains read per (tile, capture); the tiles listed */
    WmItemRec *d_plans = nullptr;                          /* [4 S + hits_cap] what k3_spans leaves k3_bursts about every item */
    uint32_t *d_ckpt = nullptr; uint32_t nck = 0;       /* clock kernel checkpoints [2][S][nseg_cap][nck][16] */
    uint32_t *d_bad = nullptr;                          /* [2][nseg_cap[0]][S] the run-length verifier's verdict per segment (K2Args.bad) */
    uint32_t *d_bad_clk = nullptr;                      /* [2][nseg_cap[1]][S] the clock verifier's */
    bool rla_chains = true;                             /* the run-length re-run lanes walk chains from their second list round on (K2Args.bad) */
    uint2 *d_hits = nullptr; uint32_t hits_cap = 0;
    uint32_t *d_pending = nullptr;
    uint32_t hdr_cap = 0, words_cap = 0, pkts_cap = 0, bytes_cap = 0;
    /* host (pinned).  The burst kernel writes its results straight into h_hdr / h_words / h_pkts / h_bytes (zero-copy over
     * PCIe: ~1 MB per push once the bursts are decoded on the GPU), so no copy has to wait for a size. */
    uint32_t *h_scalars = nullptr;
    WmBurstHdr *h_hdr = nullptr; uint32_t *h_words = nullptr; uint32_t *h_pending = nullptr;
    WmPkt *h_pkts = nullptr; uint8_t *h_bytes = nullptr;
    void *dv_hdr = nullptr, *dv_words = nullptr, *dv_pkts = nullptr, *dv_bytes = nullptr;    /* their device views */
    uint32_t n_hdr = 0, n_words = 0, n_pkts = 0;
    K1Args k1a{}; K2Args k2clk{}, k2rla{}; uint32_t ntiles = 0;    /* this push's launch arguments (collect's slow path re-uses them) */
    std::vector<HostDecoder> decs;                      /* [stream][chain][algo] */
    std::vector<wm_twin> twins;                         /* [stream][chain][2]: the last lines printed (cfg.dedup_twins) */
    std::unique_ptr<WorkerPool> pool;                   /* packet-decoder workers, created on first use */
    std::vector<wmbus_line> lines; std::string text;
    WmPush last{}; bool have_last = false, in_flight = false;
    uint64_t push_seq = 0;                              /* pushes enqueued so far */
    /* A push is enqueued in two parts: FRONT (input history, K1, hand-off rounds, both framers, carried state) and BACK
     * (the chips half-received telegrams are owed, K3, the scalars).  Only the back needs what the host packet decoders
     * left behind, so a caller that pipelines (wmbus_batch) enqueues the next push's front BEFORE it decodes the previous
     * one on the host: 2 ms less in every context's chain of dependent work.  wmbus_process = front + back. */
    bool back_due = false, enqueued = false;            /* front enqueued, back not yet; something of this push reached the stream */
    struct timeval arrival{};                           /* wall clock when the push was handed over (its last byte had arrived) */
    /* what wait_gpu() found, kept apart from `last` & co. because the next push's front may be enqueued before decode_host() */
    struct { uint32_t n_hdr = 0, n_pkts = 0; uint64_t m_end = 0, seq = 0; struct timeval arrival{}; bool valid = false; } done;
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

/* Hand-off verification runs a few rounds WITHOUT asking the host: verify -> re-run list on the device -> list launch
 * with a fixed grid -> verify ..., each round with its own counter; the host only looks at the last counters when it
 * collects the push and finishes the (rare) leftovers round by round. */
enum { WM_MAX_DEVICES = 64 };                    /* per-device tables (K1 order, kernel attributes); wmbus_open refuses ordinals beyond */
/* How many: wmbus_ctx.ema_rounds / fr_rounds.  A batch of whole waves enqueues one RSSI round and two framer rounds (bench
 * workload: clock re-runs 1400, then < 10, then 0; run-length 2300, then 0; a third costs every push 0.2 ms of empty launches).
 * A small batch is bound by its chain of dependent launches and by every host round trip in it, and its short segments
 * (wmbus_open) cascade further: it enqueues more rounds, so that the host-driven path (0.5 ms per round) stays the exception. */
enum { WM_MAX_ROUNDS = 6 };                      /* counters per kind of round: rounds + 1 <= 8 slots of the SC_* layout below (the run-length framer's
                                                    index runs to rla_rounds <= WM_MAX_ROUNDS + 1) */
/* bounded grids of the burst kernels and of the RSSI launch over the listed tiles (launch_k3) */
#ifndef WM_K3_BLOCKS
#define WM_K3_BLOCKS 64           /* build-time A/B (tools/build_variant.sh): r02 256 -> 64 blocks + 6 %, 16 blocks - 9 % */
#endif
enum { WM_RS_BLOCKS = 2048 };
enum { WM_K1_TPB_DEFAULT = 2 };                /* tiles per block of the demodulation kernel's first pass (RSSI on demand), see enqueue_front_impl */
enum { SC_ERR = 0, SC_NHITS = 1, SC_NHDR = 2, SC_NWORDS = 3, SC_NPKTS = 4, SC_NBYTES = 5, SC_SLOW = 6, SC_CHIPS = 8 /* [algo][chain] */,
       SC_RS_N = 12 /* tiles listed for the RSSI-on-demand launch */, SC_RS_FAIL = 13 /* a lane that is read could not prove its value */,
       SC_EMA = 16 /* [ema_rounds + 1] */, SC_CLK = 24 /* [fr_rounds + 1] */, SC_RLA = 32 /* [rla_rounds + 1] */, SC_COUNT = 40 };
static_assert(WM_MAX_ROUNDS + 2 <= 8 && SC_RLA + WM_MAX_ROUNDS + 2 <= SC_COUNT, "every kind of round keeps its counters inside its eight SC_* slots");

/* 4096 bytes per capture from src + row * sstride + soff to dst + row * dstride (256 threads x 16 bytes).  The input
 * history: the last 4096 staged bytes of a push are put aside (d_hist) when the push is enqueued and placed in front of
 * the window the next push is staged into when THAT push is enqueued -- in between the window stays as it was, so
 * that wmbus_collect's slow path can still re-run the first tile of a capture. */
__global__ void k_copy_hist(const uint8_t *src, uint64_t sstride, uint64_t soff, uint8_t *dst, uint64_t dstride)
{
    const uint4 v = *(const uint4 *)(src + (uint64_t)blockIdx.x * sstride + soff + 16u * threadIdx.x);
    *(uint4 *)(dst + (uint64_t)blockIdx.x * dstride + 16u * threadIdx.x) = v;
}

__global__ void k_copy_word(uint32_t *dst, uint32_t v) { *dst = v; }
__global__ void k_and_word(uint32_t *dst, uint32_t v) { *dst &= v; }

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

template <int D, bool SHIFT, bool GEN, bool FAST = false, int RS = 0, int NT = 256> int launch_k1v3(wmbus_ctx *c, const K1Args &a, dim3 grid, hipStream_t st)
{
    const size_t sm = K1GeoT<NT>::smem(D ? D : (int)c->d, SHIFT);
    static std::atomic<size_t> set_for[WM_MAX_DEVICES];     /* per device: the attribute call is not free, a push makes several launches */
    if (set_for[c->cfg.device] < sm) {
        HIPCHK(c, hipFuncSetAttribute((const void *)k1_demod2<D, SHIFT, GEN, FAST, RS, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        set_for[c->cfg.device] = sm;
    }
    hipLaunchKernelGGL((k1_demod2<D, SHIFT, GEN, FAST, RS, NT>), grid, dim3(NT), sm, st, a);
    HIPCHK(c, hipGetLastError());
    return 0;
}

/* the default switches' first pass runs the kernel without the option paths (k1_demod2<.., GEN = false>) */
template <int D, bool SHIFT> int launch_k1v2(wmbus_ctx *c, const K1Args &a, dim3 grid, hipStream_t st, int rs, bool big)
{
    const uint32_t need = WM_F_ACCURATE | WM_F_T1C1 | WM_F_S1, never = WM_F_APPROX1 | WM_F_APPROX2;
    if (D == 2 && !SHIFT && rs == 1 && big)      /* ... on the 2000-sample tile of 512 threads (the caller's grid counts THOSE tiles; wmbus_ctx.k1_big) */
        return c->cfg.tolerance_mode ? launch_k1v3<2, false, false, true, 1, 512>(c, a, grid, st) : launch_k1v3<2, false, false, false, 1, 512>(c, a, grid, st);
    if (D != 0 && rs == 1)                       /* RSSI on demand (wmbus_open has checked the switches): the first pass without it */
        return c->cfg.tolerance_mode ? launch_k1v3<D, SHIFT, false, true, 1>(c, a, grid, st) : launch_k1v3<D, SHIFT, false, false, 1>(c, a, grid, st);
    if (D != 0 && rs == 2) return launch_k1v3<D, SHIFT, false, false, 2>(c, a, grid, st);     /* ... and the listed tiles' RSSI (the same in tolerance mode) */
    if (D != 0 && a.relist == nullptr && (c->flags & need) == need && !(c->flags & never))
        /* tolerance mode (an option of the default switches' kernel only; everything else stays exact) */
        return c->cfg.tolerance_mode ? launch_k1v3<D, SHIFT, false, true>(c, a, grid, st) : launch_k1v3<D, SHIFT, false>(c, a, grid, st);
    return launch_k1v3<D, SHIFT, true>(c, a, grid, st);
}

int launch_k1_ppf(wmbus_ctx *c, const K1Args &a, dim3 grid, hipStream_t st)
{
    const size_t sm = K1PpfGeo::smem();
    static std::atomic<size_t> set_for[WM_MAX_DEVICES];
    if (set_for[c->cfg.device] < sm) {
        HIPCHK(c, hipFuncSetAttribute((const void *)k1_demod_ppf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        set_for[c->cfg.device] = sm;
    }
    hipLaunchKernelGGL(k1_demod_ppf, grid, dim3(256), sm, st, a);
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

/* The demodulation kernel's two low-pass filters (k1_fir_t / k1_fir_s, the very functions its stage B inlines) over one row of
 * arbitrary floats: x[0 .. n + 48) with the 48 samples of history in front, n a multiple of 4 up to 976 (one tile of the kernel); out[0 .. n) the 11-tap
 * filter, out[n .. 2n) the 46-tap one.  For the operands a capture cannot easily produce: signed zeros above all (a soft symbol
 * must never be -0, the clock kernel takes the slicer's bit from its sign). */
__global__ void __launch_bounds__(256) k_selftest_fir(const float *x, float *out, uint32_t n)
{
    __shared__ __attribute__((aligned(16))) float row[K1Geo::YD];
    for (uint32_t w = threadIdx.x; w < (uint32_t)K1Geo::YD; w += blockDim.x) row[w] = w >= 4u && w - 4u < n + 48u ? x[w - 4u] : 0.0f;
    __syncthreads();
    K1Args a{};
    a.g.S = 1; a.g.Mcap = n; a.dphi = out;
    k1_fir_t<false>(a, row, (int)threadIdx.x, 0, 0, (int)n);
    k1_fir_s<false>(a, row, (int)threadIdx.x, 0, 0, (int)n);
}

/* The demodulation kernels of one device's contexts run ONE AFTER THE OTHER (each fills the GPU on its own -- VALU
 * bound -- so two of them side by side only time-slice, while one of them beside the other contexts' latency- and
 * memory-bound framer kernels is complementary).  The order is kept on the GPU: a context's K1 waits for the event
 * behind the K1 launched before it, so no host thread sits in the way (a host-side turn held across the launches of
 * everything behind K1 made the turn 2.9 ms longer than the kernel). */
struct K1Chain { std::mutex m; hipEvent_t last = nullptr; const wmbus_ctx *owner = nullptr; };
K1Chain k1_chain[WM_MAX_DEVICES];

/* One HIP stream per receiver context (two with cfg.input_windows = 2); ROCm maps streams onto GPU_MAX_HW_QUEUES hardware
 * queues (4 by default) round robin, and streams that share a queue serialise against each other: eight contexts on four
 * queues run at 81 instead of 143 Gsamples/s, ten on eight at 107 (r03 A/B).  The runtime reads the variable when it
 * initialises (the first HIP call of the process), so wmbus_runtime_init() (include/wmbus_hip.h) must run before that:
 * wmbus_batch_open calls it, the CLI calls it first thing, an embedder that uses HIP itself calls it (or sets the
 * variable) before its own first HIP call -- a caller's own setting wins.  (Rounds 2-3 did this in a load-time constructor:
 * a setenv behind the host application's back)  16 covers the default batch (8 contexts, 12 in tolerance mode) with and without copy streams; with 16
 * queues and 8 contexts the rate is the one with 8 queues.  (bench.py and INTEGRATION.md used to ask the caller for this.) */
double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

void wmbus_runtime_init(void)
{
    /* setenv is not safe against concurrent getenv: call this before the process starts threads that read the
     * environment (the CLI: first statement of main) */
    if (!getenv("WMBUS_KEEP_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "16", 0);
}

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
    /* an open that was refused before it had a stream (bad device ordinal, bad geometry: the caller closes the handle it reads
     * the message from) never took part in the K1 order, and its cfg.device may lie outside the table */
    if (c->stream && c->cfg.device >= 0 && c->cfg.device < WM_MAX_DEVICES) {
        K1Chain &kc = k1_chain[c->cfg.device];
        std::lock_guard<std::mutex> lk(kc.m);
        if (kc.owner == c) { kc.last = nullptr; kc.owner = nullptr; }      /* nobody may wait on an event that is about to go */
    }
    void *dev[] = {c->d_plans, c->d_rs_flags, c->d_rs_list, c->d_bad, c->d_bad_clk, c->d_hist, c->d_list_ema, c->d_spill, c->d_chain, c->d_list2, c->d_first_bad, c->d_ckpt, c->d_in, c->d_dphi, c->d_rssi, c->d_bits, c->d_lut, c->d_ema_head, c->d_ema_tail, c->d_ema_carry,
                   c->d_chips[0], c->d_chips[1], c->d_counts[0], c->d_counts[1], c->d_st_start[0], c->d_st_start[1],
                   c->d_st_final[0], c->d_st_final[1], c->d_st_carry[0], c->d_st_carry[1], c->d_list, c->d_scalars,
                   c->d_hits, c->d_pending};
    for (void *p : dev) if (p) hipFree(p);
    void *host[] = {c->h_scalars, c->h_hdr, c->h_words, c->h_pending, c->h_pkts, c->h_bytes};
    for (void *p : host) if (p) hipHostFree(p);
    for (auto &e : c->ev) if (e) hipEventDestroy(e);
    for (hipEvent_t e : {c->ev_staged, c->ev_turn}) if (e) hipEventDestroy(e);
    if (c->copy_stream && c->copy_stream != c->stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
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
    if (cfg->device < 0 || cfg->device >= WM_MAX_DEVICES) return bail(fail(c, WMBUS_EINVAL, "device must be 0..%d", WM_MAX_DEVICES - 1));
    if (wmbus_device_count() <= cfg->device) return bail(fail(c, WMBUS_ENODEVICE, "no HIP device %d (this library has no CPU fallback)", cfg->device));
    if (hipSetDevice(cfg->device) != hipSuccess) return bail(fail(c, WMBUS_EDEVICE, "hipSetDevice(%d) failed", cfg->device));

    c->d = cfg->decimation; c->S = cfg->n_streams;
    /* Time segments: 32768 / 8192 decimated samples for batches of whole waves (r02 / r03 A/Bs: clock 65536 -8 %, 16384 -25 %;
     * run-length 16384 -9 %, 4096 -20 %).  A batch with fewer captures than a wave has lanes is bound by how long ONE lane
     * walks (5 us per 32 samples whatever runs beside it), not by the work: its segments shrink with the capture count, down
     * to 8192 / 2048 for a single capture (one 2^22-sample push of one capture: 29 -> 9 ms; the result does not depend on the
     * segmentation, test_result_independent_of_segmentation). */
    {
        uint32_t k = 1;
        while (k < 8u && c->S * k < 64u) k *= 2u;          /* 1 for >= 64 captures ... 8 for fewer than 16 */
        c->clk_form = cfg->clock_waves == 1u ? 1u : 4u;
        if (c->clk_form != 4u)
            c->C[1] = 32768u / std::min(k, 4u);            /* one capture, r04: 9.3 / 8.8 / 7.7 ms per configs[1] push with 4096 / 8192 / 16384, 9.5 / 9.0 / 10.5 for configs[2]: 8192 */
        else
            /* the systolic form walks a segment in 0.4 of the time (round 6, one capture of configs[1]: 5.6 / 4.9 / 4.25 / 5.8 ms per push with
             * 8192 / 16384 / 32768 / 65536, against 7.6 with the one-wave form's 8192): 32768 whatever the batch size -- and 65536 where the
             * demodulation kernel is the heavier side (exact arithmetic, 1.6 MS/s, no -s: the warm-ups of 12288 / 24576 samples are a quarter
             * less work per sample; bench workload 181-187 against 174-181 Gsamples/s, tolerance mode 199 against 202, configs[2] 161 against 171) */
            c->C[1] = c->S >= 64u && !cfg->tolerance_mode && cfg->decimation == 2u && !cfg->simultaneous ? 65536u : 32768u;
        /* a live stream's pushes are short (the CLI's 1 MiB = 2^18 decimated samples at 1.6 MS/s): a push of two to eight 32768-sample
         * segments is as long as its one lane with a full warm-up, 57 344 samples; with 16384 that lane is 40 960 (one capture, ms per push of
         * 2^17 / 2^18 / 2^19 input samples: 5.1 / 4.6 / 3.9 -> 2.8 / 3.4 / 3.4; 2^22: 4.3 against 4.8, so long pushes keep 32768) */
        {
            const uint64_t mp = cfg->max_push_bytes / 2u / std::max(1u, cfg->decimation);      /* a push's decimated samples at most */
            if (c->clk_form == 4u && c->S < 64u && mp > 32768u && mp <= (1u << 19)) c->C[1] = 16384u;      /* <= 32768: one segment from its exact start */
        }
        if (cfg->seg_len) c->C[1] = cfg->seg_len;
        /* run-length segments: 4096 (r04 A/B on the bench workload: 151.0 / 151.1 against 148.9 / 150.6 with round 3's 8192 and
         * 136.3 / 134.1 with 2048); 2048 with -s, where S1 telegrams -- 30-100 ms, several segments long -- are expected in both
         * chains and every segment inside one is re-run in a chain walk whose length is the telegram's whatever the segment
         * length, while first pass and first list round scale with it (configs[2] batch: 92.8 / 110.3 / 133.3 Gsamples/s with
         * 8192 / 4096 / 2048, 124.9 with 1024) */
        c->C[0] = cfg->rla_seg_len ? cfg->rla_seg_len : std::max(2048u, (cfg->simultaneous ? 2048u : 4096u) / k);
    }
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
    if (c->S < 64u) { c->ema_rounds = 2; c->fr_rounds = 5; }
    c->opt_rounds = !cfg->rounds_on_host;
    /* the run-length framer gets as many list rounds as the clock kernel (round 3: one fewer -- enough for 1.6 MS/s captures,
     * "2300 re-runs, then 0", but configs[2] (-d 5 -s) leaves 750 lanes after the first round and fell to the host-driven
     * path on EVERY push: 64 ms per step instead of 25) */
    c->rla_rounds = std::min<unsigned>(c->fr_rounds + 1u, WM_MAX_ROUNDS + 1u);
    /* ... and one more with -s: S1 telegrams (30-100 ms) lie in BOTH chains' bands there, a telegram is a chain of twenty to forty
     * run-length segments that only a walk from its first one settles, and two list rounds left a remainder in every tenth push of
     * configs[2] (round 5: rla_round = 10 994, 5 088, then the host-driven path; an empty round costs 0.2 ms) */
    if (cfg->simultaneous) c->rla_rounds = std::min<unsigned>(c->rla_rounds + 1u, WM_MAX_ROUNDS + 1u);
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
    c->n_win = cfg->input_windows == 2 ? 2u : 1u;
    /* the side stream for wmbus_stage's copies exists where it can overlap something: with two input windows.  (Every
     * stream takes a hardware queue -- ROCm maps streams onto GPU_MAX_HW_QUEUES of them round robin -- and two contexts'
     * compute streams that end up on one queue serialise: an idle copy stream per context halved the 8-context rate.) */
    if (c->n_win == 2) A(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)); else c->copy_stream = c->stream;
    for (auto &ev : c->ev) A(hipEventCreate(&ev));
    A(hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming));
    A(hipEventCreateWithFlags(&c->ev_turn, hipEventDisableTiming));
    A(dalloc(&c->d_in, (size_t)c->in_stride * c->S * c->n_win));
    A(dalloc(&c->d_hist, (size_t)WM_HIST_BYTES * c->S));
    A(dalloc(&c->d_dphi, (size_t)rows * c->Mcap));
    A(dalloc(&c->d_rssi, (size_t)rows * c->Mcap));
    A(dalloc(&c->d_bits, (size_t)rows * (c->Mcap / 32)));
    A(dalloc(&c->d_lut, (size_t)2 * 32 * WM_MAX_DECIM));
    A(dalloc(&c->d_ema_head, (size_t)rows * c->ntiles_cap));
    A(dalloc(&c->d_ema_tail, (size_t)rows * c->ntiles_cap));
    A(dalloc(&c->d_ema_carry, (size_t)2 * rows));
    A(dalloc(&c->d_first_bad, (size_t)rows));
    {
        const uint32_t need = WM_F_ACCURATE | WM_F_T1C1 | WM_F_S1, never = WM_F_APPROX1 | WM_F_APPROX2;
        c->rs_od = !cfg->rssi_full && !cfg->keep_taps && cfg->prefilter != WMBUS_PREFILTER_POLYPHASE && c->d >= 2 && c->d <= 5 &&
                   (c->flags & need) == need && !(c->flags & never);
        /* the larger tile of the first pass: decimation 2 without -s (the LDS of a 512-thread block at d >= 3 or with the -s staging
         * would cost more occupancy than the halo saves) */
        c->k1_big = c->rs_od && c->d == 2 && !(c->flags & WM_F_SHIFT) && !cfg->k1_small_tile;
        c->k1_tail_pm = cfg->tolerance_mode ? 0u : 60u;
        c->k1_tpb = cfg->k1_tiles_per_block ? std::min(cfg->k1_tiles_per_block, 64u) : WM_K1_TPB_DEFAULT;
        c->clk_form = cfg->clock_waves == 1u ? 1u : 4u;
        if (c->rs_od) { A(dalloc(&c->d_rs_flags, (size_t)c->ntiles_cap * c->S)); A(dalloc(&c->d_rs_list, (size_t)c->ntiles_cap * c->S)); }
    }
    const size_t stw[2] = {sizeof(WmRlaState), sizeof(WmClkState)};
    for (int a = 0; a < 2; a++) {
        A(dalloc(&c->d_chips[a], (size_t)rows * c->nseg_cap[a] * c->cap[a]));
        A(dalloc(&c->d_counts[a], (size_t)rows * c->nseg_cap[a]));
        A(hipMalloc(&c->d_st_start[a], (size_t)rows * c->nseg_cap[a] * stw[a]));
        A(hipMalloc(&c->d_st_final[a], (size_t)rows * c->nseg_cap[a] * stw[a]));
        A(hipMalloc(&c->d_st_carry[a], (size_t)2 * rows * stw[a]));
    }
    {
        const uint64_t dec_total_ = (uint64_t)c->S * c->Mcap;
        const uint64_t want = cfg->spill_words ? cfg->spill_words : std::max<uint64_t>(1u << 20, dec_total_ / 16);
        c->spill_words = (uint32_t)std::min<uint64_t>((want + WM_SPILL_CHUNK - 1) / WM_SPILL_CHUNK * WM_SPILL_CHUNK, 0xFFFF0000u);
        A(dalloc(&c->d_spill, (size_t)c->spill_words));
        A(dalloc(&c->d_chain, (size_t)rows * c->nseg_cap[0] * WM_SPILL_LEVELS));
    }
    A(dalloc(&c->d_list, (size_t)rows * std::max(c->nseg_cap[0], c->nseg_cap[1])));
    A(dalloc(&c->d_list2, (size_t)rows * c->nseg_cap[0]));
    A(dalloc(&c->d_list_ema, (size_t)rows));
    c->nck = c->C[1] / WM_CK_SAMPLES ? c->C[1] / WM_CK_SAMPLES - 1 : 0;
    A(dalloc(&c->d_ckpt, std::max<size_t>(16, (size_t)rows * c->nseg_cap[1] * c->nck * 16)));
    {
        const size_t n_seen0 = (size_t)rows * c->nseg_cap[0], n_seen1 = (size_t)rows * c->nseg_cap[1], n_chain = (size_t)rows * c->nseg_cap[0] + 1;
        c->zero_words = SC_COUNT + n_seen0 + n_seen1 + n_chain;
        A(dalloc(&c->d_scalars, c->zero_words));
        c->d_sync_seen[0] = c->d_scalars + SC_COUNT; c->d_sync_seen[1] = c->d_sync_seen[0] + n_seen0; c->d_nchain = c->d_sync_seen[1] + n_seen1;
    }
    const uint64_t dec_total = (uint64_t)c->S * c->Mcap;
    c->hdr_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(65536, dec_total / 1024), 1u << 24);
    c->hits_cap = c->hdr_cap;
    /* bursts that still travel as chips are the ones whose plan reaches past the end of the push (every such hit is
     * shipped with the chips up to the end, at most 16 x 290 + 1 of them) and the continuations of those the decoders
     * took: room for two longest bursts per (capture, chain, framer); beyond that bursts are dropped with a warning */
    c->gpu_decode = !cfg->bursts_to_host;                /* else: every burst to the host decoders as chips (A/B, tests) */
    c->words_cap = c->gpu_decode ? (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 20, 4ull * c->S * 2 * WM_MAXCHIPS_S1), 1u << 28)
                                 : (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 20, dec_total / 8), 1u << 29);
    c->pkts_cap = c->hdr_cap;
    c->bytes_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 20, dec_total / 64), 1u << 30);
    if (cfg->burst_caps[0]) { c->hdr_cap = cfg->burst_caps[0]; c->hits_cap = std::max(c->hits_cap, c->hdr_cap); }      /* tests: tiny burst storage, to reach the overflow paths */
    if (cfg->burst_caps[1]) c->words_cap = cfg->burst_caps[1];
    if (cfg->burst_caps[2]) c->pkts_cap = cfg->burst_caps[2];
    if (cfg->burst_caps[3]) c->bytes_cap = cfg->burst_caps[3];
    A(dalloc(&c->d_hits, (size_t)c->hits_cap));
    if (c->rs_od) A(dalloc(&c->d_plans, (size_t)4 * c->S + c->hits_cap));
    A(dalloc(&c->d_pending, (size_t)4 * c->S));
    /* the verifiers' verdict per segment, for the chain walk of the re-run lanes (K2Args.bad).  Zeroed on the context's own stream: a
     * synchronous hipMemset runs on the NULL stream, whose hardware queue then takes part in the round-robin of streams onto
     * queues -- eight contexts lost 5 % to that in round 4's A/Bs. */
    A(dalloc(&c->d_bad, (size_t)rows * c->nseg_cap[0]));
    if (e == hipSuccess) A(hipMemsetAsync(c->d_bad, 0, (size_t)rows * c->nseg_cap[0] * sizeof(uint32_t), c->stream));
    A(dalloc(&c->d_bad_clk, (size_t)rows * c->nseg_cap[1]));
    if (e == hipSuccess) A(hipMemsetAsync(c->d_bad_clk, 0, (size_t)rows * c->nseg_cap[1] * sizeof(uint32_t), c->stream));
    A(hipHostMalloc((void **)&c->h_scalars, SC_COUNT * sizeof(uint32_t)));
    A(hipHostMalloc((void **)&c->h_hdr, (size_t)c->hdr_cap * sizeof(WmBurstHdr)));
    A(hipHostMalloc((void **)&c->h_words, (size_t)c->words_cap * sizeof(uint32_t)));
    A(hipHostMalloc((void **)&c->h_pkts, (size_t)c->pkts_cap * sizeof(WmPkt)));
    A(hipHostMalloc((void **)&c->h_bytes, (size_t)c->bytes_cap));
    A(hipHostMalloc((void **)&c->h_pending, (size_t)4 * c->S * sizeof(uint32_t)));
    if (e == hipSuccess) {
        A(hipHostGetDevicePointer(&c->dv_hdr, c->h_hdr, 0)); A(hipHostGetDevicePointer(&c->dv_words, c->h_words, 0));
        A(hipHostGetDevicePointer(&c->dv_pkts, c->h_pkts, 0)); A(hipHostGetDevicePointer(&c->dv_bytes, c->h_bytes, 0));
    }
    if (e != hipSuccess) return bail(fail(c, e == hipErrorOutOfMemory ? WMBUS_ENOMEM : WMBUS_EDEVICE, "allocation failed: %s", hipGetErrorString(e)));

    /* initial state = the reference's zero-initialised statics */
    A(hipMemsetAsync(c->d_ema_carry, 0, 2 * rows * sizeof(float), c->stream));
    /* a disabled chain (-p T / -p S) never writes its hand-off records: they must compare equal, not hold what an
     * earlier context left in the recycled allocation */
    A(hipMemsetAsync(c->d_ema_head, 0, (size_t)rows * c->ntiles_cap * sizeof(float), c->stream));
    A(hipMemsetAsync(c->d_ema_tail, 0, (size_t)rows * c->ntiles_cap * sizeof(float), c->stream));
    A(hipMemsetAsync(c->d_first_bad, 0xFF, rows * sizeof(uint32_t), c->stream));
    A(hipMemsetAsync(c->d_st_carry[1], 0, 2 * rows * sizeof(WmClkState), c->stream));
    {
        std::vector<WmRlaState> init(2 * rows, WmRlaState{0, 8 * 256, 0, 0u, 0u, 0u, 24, 24});   /* rtl_wmbus.c:628-637,717-726 */
        A(hipMemcpyAsync(c->d_st_carry[0], init.data(), 2 * rows * sizeof(WmRlaState), hipMemcpyHostToDevice, c->stream));
        A(hipStreamSynchronize(c->stream));
    }
    A(hipMemsetAsync(c->d_scalars, 0, c->zero_words * sizeof(uint32_t), c->stream));
    A(hipMemsetAsync(c->d_pending, 0, 4 * c->S * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_fill, dim3(c->S * c->n_win), dim3(256), 0, c->stream, c->d_in, c->in_stride, (uint32_t)c->in_stride, (uint8_t)128);
    hipLaunchKernelGGL(k_fill, dim3(c->S), dim3(256), 0, c->stream, c->d_hist, (uint64_t)WM_HIST_BYTES, (uint32_t)WM_HIST_BYTES, (uint8_t)128);
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
                hd.owed = 0; hd.fed = 0;
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

/* Every stream's bytes of the next push in ONE copy (the batch's host-sourced path: a context's streams lie `pitch` apart in
 * its page-locked slab and in_stride apart in HBM -- a pitched copy; 128 copies of 1-2 MiB per context-push were 1024
 * copy commands per step of the batch CLI).  Same rules as wmbus_stage. */
static int wm_stage_all(wmbus_ctx *c, const uint8_t *slab, size_t pitch, size_t nbytes)
{
    if (!c || !slab) return WMBUS_EINVAL;
    if (nbytes > c->cfg.max_push_bytes || nbytes % WMBUS_BLOCK_BYTES || pitch < nbytes) return fail(c, WMBUS_EINVAL, "stage: nbytes must be a multiple of 4096 and <= max_push_bytes");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->in_flight && c->n_win == 1) return fail(c, WMBUS_EINVAL, "stage: a push is in flight and the context has one input window (cfg.input_windows = 2 overlaps them)");
    HIPCHK(c, hipMemcpy2DAsync(wmbus_device_input(c, 0), c->in_stride, slab, pitch, nbytes, c->S, hipMemcpyHostToDevice, c->copy_stream));
    return WMBUS_OK;
}

void *wmbus_device_input(wmbus_ctx *c, unsigned stream)
{
    if (!c || stream >= c->S) return nullptr;
    return c->d_in + ((size_t)c->fill * c->S + stream) * c->in_stride + WM_HIST_BYTES;
}

int wmbus_stage(wmbus_ctx *c, unsigned stream, const uint8_t *cu8, size_t nbytes)
{
    if (!c || stream >= c->S || !cu8) return WMBUS_EINVAL;
    if (nbytes > c->cfg.max_push_bytes || nbytes % WMBUS_BLOCK_BYTES) return fail(c, WMBUS_EINVAL, "stage: nbytes must be a multiple of 4096 and <= max_push_bytes");
    /* with one input window the kernels of a push in flight still read it; with two (cfg.input_windows = 2) the next
     * push is staged into the other one while they run */
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->in_flight && c->n_win == 1) return fail(c, WMBUS_EINVAL, "stage: a push is in flight and the context has one input window (cfg.input_windows = 2 overlaps them)");
    HIPCHK(c, hipMemcpyAsync(wmbus_device_input(c, stream), cu8, nbytes, hipMemcpyHostToDevice, c->copy_stream));
    return WMBUS_OK;
}

/* ---- launches --------------------------------------------------------------------------------- */
static float *ema_carry(wmbus_ctx *c, bool out) { return c->d_ema_carry + (size_t)(c->carry_in ^ (out ? 1u : 0u)) * 2 * c->S; }
static void *st_carry(wmbus_ctx *c, int algo, bool out)
{
    const size_t w = algo == WMBUS_ALGO_RLA ? sizeof(WmRlaState) : sizeof(WmClkState);
    return (char *)c->d_st_carry[algo] + (size_t)(c->carry_in ^ (out ? 1u : 0u)) * 2 * c->S * w;
}

static int launch_k1_any(wmbus_ctx *c, const K1Args &k1, dim3 grid, hipStream_t st = nullptr, int rs = 0, bool big = false)    /* rs: 0 everything, 1 no RSSI, 2 RSSI of the listed tiles; big: grid.x counts 2000-sample tiles (rs = 1 only) */
{
    const bool sh = c->flags & WM_F_SHIFT;
    if (!st) st = c->stream;
    if (c->cfg.prefilter == WMBUS_PREFILTER_POLYPHASE) return launch_k1_ppf(c, k1, grid, st);
    if (c->d == 2) return sh ? launch_k1v2<2, true>(c, k1, grid, st, rs, false) : launch_k1v2<2, false>(c, k1, grid, st, rs, big);
    if (c->d == 3) return sh ? launch_k1v2<3, true>(c, k1, grid, st, rs, false) : launch_k1v2<3, false>(c, k1, grid, st, rs, false);
    if (c->d == 4) return sh ? launch_k1v2<4, true>(c, k1, grid, st, rs, false) : launch_k1v2<4, false>(c, k1, grid, st, rs, false);
    if (c->d == 5) return sh ? launch_k1v2<5, true>(c, k1, grid, st, rs, false) : launch_k1v2<5, false>(c, k1, grid, st, rs, false);
    return sh ? launch_k1v2<0, true>(c, k1, grid, st, rs, false) : launch_k1v2<0, false>(c, k1, grid, st, rs, false);      /* any other rate */
}

/* EMA hand-offs between tiles (k1_verify), the first uncertified tile of every row into the repair list (k1_collect),
 * its length into scalar `cnt`.  A listed tile is re-run sequentially from its predecessor's exact tail, which may
 * uncover the next one: exact by construction. */
static void ema_verify(wmbus_ctx *c, uint32_t cnt)
{
    const uint32_t rows = 2 * c->S, ntiles = c->ntiles;
    hipLaunchKernelGGL(k1_verify, dim3((rows + 63) / 64, ntiles), dim3(64), 0, c->stream, c->d_ema_head, c->d_ema_tail, ema_carry(c, false), ntiles, rows, c->d_first_bad);
    hipLaunchKernelGGL(k1_collect, dim3((rows + 63) / 64), dim3(64), 0, c->stream, c->d_first_bad, ntiles, rows, c->S, c->d_list_ema, c->d_scalars + cnt);
}

static int ema_repair(wmbus_ctx *c, uint32_t cnt)          /* list launch: a fixed grid walks d_list[0 .. scalar cnt) */
{
    K1Args k1 = c->k1a;
    k1.relist = c->d_list_ema; k1.n_relist = c->d_scalars + cnt;
    return launch_k1_any(c, k1, dim3(16, 1));
}

static void ema_commit(wmbus_ctx *c)
{
    hipLaunchKernelGGL(k1_commit, dim3((2 * c->S + 63) / 64), dim3(64), 0, c->stream, c->d_ema_tail, ema_carry(c, true), c->ntiles, 2 * c->S);
}

/* Does the launch that reads the re-run list whose length is scalar `cnt` walk chains (K2Args.bad), or re-run every listed segment on its own?
 * The framers' FIRST list round is the parallel one -- except the run-length framer's with -s (fr_launch has the measurements). */
static bool list_walks_chains(const wmbus_ctx *c, int algo, uint32_t cnt)
{
    if (algo == WMBUS_ALGO_RLA) return c->rla_chains && (c->cfg.simultaneous || cnt != SC_RLA + 1u);
    return cnt != (uint32_t)SC_CLK;
}

static void fr_verify(wmbus_ctx *c, int algo, uint32_t cnt, hipStream_t st = nullptr)
{
    if (!st) st = c->stream;
    const K2Args &a = algo == WMBUS_ALGO_RLA ? c->k2rla : c->k2clk;
    const uint32_t lanes = 2u * a.g.nseg[algo] * a.g.S, words = (algo == WMBUS_ALGO_RLA ? sizeof(WmRlaState) : sizeof(WmClkState)) / 4;
    hipLaunchKernelGGL(k2_verify, dim3((lanes + 255) / 256), dim3(256), 0, st, a.g, (uint32_t)algo, (const uint32_t *)a.st_start,
                       (const uint32_t *)a.st_final, words, algo == WMBUS_ALGO_RLA ? c->d_list2 : c->d_list, c->d_scalars + cnt,
                       algo == WMBUS_ALGO_RLA ? (c->rla_chains ? c->d_bad : (uint32_t *)nullptr) : c->d_bad_clk, list_walks_chains(c, algo, cnt) ? 1u : 0u);
}

/* one framer's kernel alone: every lane (cnt == ~0) or the re-run list whose length is scalar `cnt` */
static void fr_launch(wmbus_ctx *c, int algo, uint32_t cnt, hipStream_t st = nullptr)
{
    if (!st) st = c->stream;
    K2Args a = algo == WMBUS_ALGO_RLA ? c->k2rla : c->k2clk;
    const uint32_t lanes = 2u * a.g.nseg[algo] * a.g.S;
    const bool all = cnt == 0xFFFFFFFFu;
    a.list = all ? nullptr : (algo == WMBUS_ALGO_RLA ? c->d_list2 : c->d_list);
    a.n_lanes = lanes; a.n_ptr = all ? nullptr : c->d_scalars + cnt;
    /* The run-length framer's FIRST list round re-runs every listed segment on its own, in parallel, from its predecessor's
     * end state as recorded: a neighbour that is listed too usually comes out of its re-run with the end state it had (the
     * framer resets inside the segment), so that state was right all along -- walking such runs serially made the round twice
     * as long on the bench workload (r04 A/B: 133 against 147 Gsamples/s).  What is STILL listed after that round is a true
     * cascade (a burst longer than a segment): from the second list round on a listed lane walks its chain (rla_lanes). */
    /* ... except with -s (round 6): S1 telegrams lie in both chains' bands there, a telegram is a run of twenty to forty listed segments, and
     * re-running each of them on its own from a predecessor that is wrong as well is work in vain: 10 994 lanes in the first round, 5 088
     * walking their chains in the second.  With the walk in the FIRST round the second finds 2 lanes (configs[2] at batch size: 174-177 ->
     * 181-182 Gsamples/s, run-length stage 9.9 -> 8.3 ms). */
    if (!all && !list_walks_chains(c, algo, cnt)) a.bad = nullptr;      /* (the clock kernel's first list round is parallel too: clock_lanes) */
    /* (Round 5 tried a middle way for the clock kernel's first round: parallel, but a lane whose end state came out new carries it
     * on into an UNLISTED successor.  The second round shrank from 7 to 4 lanes and the job lost 1.8 %: 170.0 against 173.1.) */
    /* list launches: blocks for 3/16 of the lanes (a re-run list is a few percent of them; the blocks walk whatever is more) */
#ifdef WM_DBG_SKIP_CLOCK                                      /* timing experiment only (tools/build_variant.sh): from the fourth push on no clock-recovery launch at all --
                                                             * what the clock kernels' place on the GPU costs the rest of the job.  The output is wrong. */
    if (algo == WMBUS_ALGO_T2A && c->push_seq > 3) return;
#endif
#ifdef WM_DBG_SKIP_RLA
    if (algo == WMBUS_ALGO_RLA && c->push_seq > 3) return;
#endif
    if (algo == WMBUS_ALGO_T2A && c->clk_form == 4u) {
        /* the systolic form: a block of four waves per 64 lanes (wm_k2_clock_sys.h); list launches as below: blocks for 3/16 of the lanes */
        const uint32_t grid = all ? (lanes + 63u) / 64u : std::max(64u, (lanes / 64u) * 3u / 16u);
        if (all) {
            const bool dc = c->flags & WM_F_DC, coop = a.g.S % 64u == 0u;        /* whole waves of captures load cooperatively */
            if (dc && coop) hipLaunchKernelGGL((k2_clock_sys<true, true>), dim3(grid), dim3(256), sizeof(ClkSysLds), st, a);
            else if (dc) hipLaunchKernelGGL((k2_clock_sys<true, false>), dim3(grid), dim3(256), sizeof(ClkSysLds), st, a);
            else if (coop) hipLaunchKernelGGL((k2_clock_sys<false, true>), dim3(grid), dim3(256), sizeof(ClkSysLds), st, a);
            else hipLaunchKernelGGL((k2_clock_sys<false, false>), dim3(grid), dim3(256), sizeof(ClkSysLds), st, a);
        } else if (c->flags & WM_F_DC) hipLaunchKernelGGL(k2_clock_sys_list<true>, dim3(grid), dim3(256), sizeof(ClkSysLds), st, a);
        else hipLaunchKernelGGL(k2_clock_sys_list<false>, dim3(grid), dim3(256), sizeof(ClkSysLds), st, a);
        return;
    }
    const uint32_t B = 64 * (algo == WMBUS_ALGO_RLA ? WM_RLA_WPB : WM_CLK_WPB), grid = all ? (lanes + B - 1) / B : std::max(16u, (lanes / B) * 3u / 16u);
    if (algo == WMBUS_ALGO_RLA) {
        if (all) hipLaunchKernelGGL(k2_rla, dim3(grid), dim3(B), 0, st, a);
        else hipLaunchKernelGGL(k2_rla_list, dim3(grid), dim3(B), 0, st, a);
    }
    else if (all) {
        if (c->flags & WM_F_DC) hipLaunchKernelGGL(k2_clock<true>, dim3(grid), dim3(B), 0, st, a);
        else hipLaunchKernelGGL(k2_clock<false>, dim3(grid), dim3(B), 0, st, a);
    } else if (c->flags & WM_F_DC) hipLaunchKernelGGL(k2_clock_list<true>, dim3(grid), dim3(B), 0, st, a);
    else hipLaunchKernelGGL(k2_clock_list<false>, dim3(grid), dim3(B), 0, st, a);
}

static void fr_carry(wmbus_ctx *c)
{
    const uint32_t rows = 2u * c->S;
    const WmPush &g = c->k2clk.g;
    hipLaunchKernelGGL(k_carry, dim3((rows + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)c->k2clk.st_final,
                       (uint32_t *)st_carry(c, 1, true), (uint32_t)(sizeof(WmClkState) / 4), rows, g.nseg_cap[1], g.nseg[1]);
    if (c->flags & WM_F_RLA)
        hipLaunchKernelGGL(k_carry, dim3((rows + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)c->k2rla.st_final,
                           (uint32_t *)st_carry(c, 0, true), (uint32_t)(sizeof(WmRlaState) / 4), rows, g.nseg_cap[0], g.nseg[0]);
    else   /* a disabled run-length framer keeps its state */
        hipMemcpyAsync(st_carry(c, 0, true), st_carry(c, 0, false), (size_t)rows * sizeof(WmRlaState), hipMemcpyDeviceToDevice, c->stream);
}

/* K3 on the settled chip streams: chips-per-framer sums, access-code hits, bursts (decoded on the GPU where they are
 * complete), then the scalars for the host.  Nothing here needs a number from the host. */
static int launch_k3(wmbus_ctx *c, bool again)
{
    const WmPush &g = c->last;
    if (again) {                                             /* collect's slow path: the counters of the first attempt go */
        HIPCHK(c, hipMemsetAsync(c->d_scalars + SC_NHITS, 0, 5 * sizeof(uint32_t), c->stream));          /* NHITS .. NBYTES */
        HIPCHK(c, hipMemsetAsync(c->d_scalars + SC_CHIPS, 0, 6 * sizeof(uint32_t), c->stream));          /* CHIPS, RS_N, RS_FAIL */
        /* ... and so does its burst-storage warning (chips dropped by the framers stay dropped: that bit is kept) */
        hipLaunchKernelGGL(k_and_word, dim3(1), dim3(1), 0, c->stream, c->d_scalars + SC_ERR, ~(uint32_t)WM_ERR_BURST_OVERFLOW);
    }
    hipLaunchKernelGGL(k_sum_counts, dim3(std::min(64u, (2u * (c->nseg_cap[0] + c->nseg_cap[1]) * c->S + 255u) / 256u)), dim3(256), 0, c->stream, g,
                       c->d_counts[0], c->d_counts[1], c->d_scalars + SC_CHIPS);
    /* grid.y: the parts a region is scanned in -- the run-length framer's regions can be longer than their capacity (spill chunks) */
    const uint32_t scan_parts = (std::max(g.cap[0] + WM_SPILL_LEVELS * WM_SPILL_CHUNK, g.cap[1]) + WM_K3_SCAN_PART - 1u) / WM_K3_SCAN_PART;
    hipLaunchKernelGGL(k3_scan, dim3((2u * (g.nseg[0] + g.nseg[1]) * g.S + 255u) / 256u, scan_parts), dim3(256), 0, c->stream, g, c->d_chips[0], c->d_chips[1],
                       c->d_counts[0], c->d_counts[1], c->d_sync_seen[0], c->d_sync_seen[1], c->d_hits, c->d_scalars + SC_NHITS,
                       c->hits_cap, c->d_scalars + SC_ERR);
    K3Args k3{};
    k3.g = g; k3.rssi = c->d_rssi;
    k3.chips[0] = c->d_chips[0]; k3.chips[1] = c->d_chips[1]; k3.counts[0] = c->d_counts[0]; k3.counts[1] = c->d_counts[1];
    k3.hits = c->d_hits; k3.n_hits = c->d_scalars + SC_NHITS; k3.hits_cap = c->hits_cap; k3.pending = c->d_pending;
    k3.hdr = (WmBurstHdr *)c->dv_hdr; k3.hdr_cap = c->hdr_cap; k3.words = (uint32_t *)c->dv_words; k3.words_cap = c->words_cap;
    k3.n_hdr = c->d_scalars + SC_NHDR; k3.n_words = c->d_scalars + SC_NWORDS; k3.err = c->d_scalars + SC_ERR;
    if (c->gpu_decode) {
        k3.pkts = (WmPkt *)c->dv_pkts; k3.pkts_cap = c->pkts_cap; k3.bytes = (uint8_t *)c->dv_bytes; k3.bytes_cap = c->bytes_cap;
        k3.n_pkts = c->d_scalars + SC_NPKTS; k3.n_bytes = c->d_scalars + SC_NBYTES;
    }
    k3.plans = nullptr;
    if (c->rs_this && !c->rs_full_now) {
        k3.plans = c->d_plans;
        /* RSSI on demand: which tiles do the bursts touch (k3_spans), then their RSSI (an RS = 2 launch of the demodulation
         * kernel over the list, a fixed grid), then the bursts */
        HIPCHK(c, hipMemsetAsync(c->d_rs_flags, 0, (size_t)c->ntiles * c->S * sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(k3_spans, dim3(std::max(1u, std::min((4 * c->S + c->hits_cap + 3u) / 4u, (uint32_t)WM_K3_BLOCKS))), dim3(256), 0, c->stream, k3, c->T, c->ntiles,
                           c->d_rs_flags, c->d_rs_list, c->d_scalars + SC_RS_N);
        K1Args k1 = c->k1a;
        k1.relist = c->d_rs_list; k1.n_relist = c->d_scalars + SC_RS_N;
        HIPCHK(c, hipEventRecord(c->ev[9], c->stream));
        const int rc = launch_k1_any(c, k1, dim3(std::max(1u, std::min((uint32_t)WM_RS_BLOCKS, c->ntiles * c->S)), 1), nullptr, 2);    /* r04 A/B: 512 / 1024 / 2048 / 4096 blocks: no difference */
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->ev[10], c->stream));
    }
    /* Few blocks: a latency-bound wave parked on a SIMD costs the demodulation kernel one of its eight wave slots there
     * for as long as it lives; 256 blocks put one on every SIMD of the chip (r02 sweep: 256 -> 64 blocks + 6 %, 16 blocks - 9 %: then the kernel itself becomes the longest link of the chain) */
    const uint32_t most = 4 * c->S + c->hits_cap;
    hipLaunchKernelGGL(k3_bursts, dim3(std::max(1u, std::min((most + 3u) / 4u, (uint32_t)WM_K3_BLOCKS))), dim3(256), 0, c->stream, k3, 0xFFFFFFFFu);
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int enqueue_front_impl(wmbus_ctx *c, size_t nbytes)
{
    if (nbytes == 0 || nbytes > c->cfg.max_push_bytes || nbytes % WMBUS_BLOCK_BYTES)
        return fail(c, WMBUS_EINVAL, "process: nbytes must be a positive multiple of 4096 and <= max_push_bytes");
    if (c->in_flight) return fail(c, WMBUS_EINVAL, "process: previous push not collected");
    HIPCHK(c, hipSetDevice(c->cfg.device));            /* HIP's current device is per host thread; a context may be driven from any */
    if (c->poisoned) return fail(c, WMBUS_EDEVICE, "process: an earlier internal error left this context unusable; close it");
    gettimeofday(&c->arrival, nullptr);
    if (c->committed) { c->carry_in ^= 1u; c->committed = false; }     /* the previous push's end state is this one's start state */
    const uint32_t n_new = (uint32_t)(nbytes / 2);
    uint8_t *win = c->d_in + (size_t)c->fill * c->S * c->in_stride;      /* the window wmbus_stage has been filling */
    WmPush g{};
    g.in = win; g.in_stride = c->in_stride; g.n0 = c->n0; g.m0 = c->n0 / c->d;
    g.n_new = n_new; g.M = (uint32_t)((c->n0 + n_new) / c->d - g.m0);
    g.Mcap = c->Mcap; g.d = c->d; g.S = c->S; g.lut_n = 32 * c->d;
    g.lut_phase0 = (uint32_t)((13ull * (c->n0 % g.lut_n)) % g.lut_n);
    g.flags = c->flags;
    for (int al = 0; al < 2; al++) {
        g.seg_len[al] = c->C[al]; g.nseg[al] = (g.M + c->C[al] - 1) / c->C[al]; g.nseg_cap[al] = c->nseg_cap[al]; g.cap[al] = c->cap[al];
    }
    g.warm[0] = c->cfg.warmup_t1c1; g.warm[1] = c->cfg.warmup_s1; g.lookback = c->cfg.rla_lookback;
    g.s1_span = 1u;                                     /* (2: S1 clock lanes over two segments -- built and measured in rounds 2-4, -2 %: the kernels keep the form) */
    g.sp.arena = c->d_spill; g.sp.arena_words = c->spill_words; g.sp.chain = c->d_chain; g.sp.nchain = c->d_nchain;
    g.sp.used = c->d_nchain + (size_t)2 * c->S * c->nseg_cap[0];
    c->last = g; c->have_last = true;
    c->n_hdr = c->n_words = c->n_pkts = 0;
    c->enqueued = true;                                 /* from here on a failure leaves half a push on the stream: the context is poisoned */

    /* the staged bytes arrive on the copy stream */
    if (c->copy_stream != c->stream) {
        HIPCHK(c, hipEventRecord(c->ev_staged, c->copy_stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_staged, 0));
    }
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    hipLaunchKernelGGL(k_copy_hist, dim3(c->S), dim3(256), 0, c->stream, c->d_hist, (uint64_t)WM_HIST_BYTES, (uint64_t)0, win, c->in_stride);
    if (g.M > 0) {
        HIPCHK(c, hipMemsetAsync(c->d_scalars, 0, c->zero_words * sizeof(uint32_t), c->stream));     /* scalars, region flags, spill chains */
        /* K1 */
        const uint32_t T = c->T, ntiles = (g.M + T - 1) / T;
        c->ntiles = ntiles;
        /* on demand pays while the bursts touch a minority of the tiles (an RS = 2 tile costs a third of a full one, and the
         * first pass saves a seventh): a context that has just listed more than cfg.rssi_dense_pm per mille of them (configs[2]: S1
         * telegrams of 30-100 ms in both chains) takes the full pass for the next sixteen pushes, then looks again
         * (r04 visit: configs[2] lists 31 % of its tiles and loses 7 % on demand -- its front end, five input samples per decimated one
         * through the -s rotation, is most of the kernel; the bench workload lists 11 % and gains 5-7 %) */
        c->rs_this = c->rs_od && c->rs_pause == 0;
        if (c->rs_pause) c->rs_pause--;
        c->k1a = K1Args{g, c->d_dphi, c->d_rssi, c->d_lut, c->d_lut + 32 * WM_MAX_DECIM, c->d_ema_head, c->d_ema_tail, ntiles, c->d_scalars + SC_ERR,
                        nullptr, ema_carry(c, false), nullptr, 0u, c->d_rs_flags, ema_carry(c, true), c->d_scalars + SC_RS_FAIL};
        int rc;
        {
            K1Chain &kc = k1_chain[c->cfg.device];
            std::lock_guard<std::mutex> lk(kc.m);
            if (kc.last && kc.owner != c) HIPCHK(c, hipStreamWaitEvent(c->stream, kc.last, 0));
            HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
            /* The hand-over of the turn costs 0.15-0.2 ms (the next context's stream sits in an event wait on another
             * hardware queue; r03 trace: median 128 us between one K1 and the next, eight times per step = 5 % of it).
             * So the turn is handed over EARLY: a push's tiles leave in two launches, the event the next context waits for
             * lies behind the first, and the last k1_tail_pm per mille of the tiles run while the hand-over is under
             * way -- beside the head of the next context's K1 at most.
             * r04 A/B: exact 152.5 / 153.2 against 149.1 / 150.1 Gsamples/s with 60 per mille (30: 151.6 / 152.0, 120: 151.0 / 152.3, 250:
             * 148.9 / 147.4); tolerance mode, whose K1 is a third of the length, 170.1 / 171.9 against 171.0 / 173.0: off there */
            const bool big = c->rs_this && c->k1_big;            /* the first pass without the RSSI on 2000-sample tiles (wm_k1_demod.h K1GeoT) */
            const uint32_t T1 = big ? (uint32_t)WM_K1_TILE_BIG : T, nt1 = (g.M + T1 - 1) / T1;
            const uint32_t n_tail = std::min(nt1 - 1u, (uint32_t)((uint64_t)nt1 * c->k1_tail_pm / 1000u));
            K1Args k1 = c->k1a;
            const int rs = c->rs_this ? 1 : 0;
            /* the first pass without the RSSI takes several consecutive tiles per block: the input of a block's next tile is on
             * its way while the current one is computed (wm_k1_demod.h, K1Args.tpb) */
            const uint32_t tpb = c->rs_this && c->d >= 2 && c->d <= 5 ? c->k1_tpb : 1u;
            k1.tpb = tpb; k1.tile_end = nt1 - n_tail;
#ifdef WM_DBG_SKIP_K1                                       /* timing experiment only (tools/build_variant.sh): from the fourth push on the framers run on the
                                                             * soft symbols the third push left -- what a context's chain costs without any demodulation kernel
                                                             * beside it.  The output is wrong. */
            const bool dbg_skip = c->push_seq > 3;
#else
            const bool dbg_skip = false;
#endif
            rc = dbg_skip ? 0 : launch_k1_any(c, k1, dim3((nt1 - n_tail + tpb - 1u) / tpb, c->S), nullptr, rs, big);
            if (rc) return rc;
            if (n_tail && !dbg_skip) {
                HIPCHK(c, hipEventRecord(c->ev_turn, c->stream));
                k1.tile0 = nt1 - n_tail; k1.tile_end = nt1;
                rc = launch_k1_any(c, k1, dim3((n_tail + tpb - 1u) / tpb, c->S), nullptr, rs, big);
                if (rc) return rc;
            }
            HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
            kc.last = n_tail ? c->ev_turn : c->ev[4]; kc.owner = c;
        }
        /* Everything behind K1 is enqueued while it runs -- no step of a push waits for the host any more:
         * hand-off verification makes its first rounds on the device (counters per round), K3 reads its item
         * count there, the results land in pinned host memory.  wmbus_collect looks at the last counters. */
        c->rs_full_now = false;
        if (!c->rs_this) {                                      /* (on demand: the RSSI comes behind the framers, launch_k3) */
            for (unsigned r = 0; r < c->ema_rounds && c->opt_rounds; r++) {
                ema_verify(c, SC_EMA + r);
                rc = ema_repair(c, SC_EMA + r);
                if (rc) return rc;
            }
            ema_verify(c, SC_EMA + c->ema_rounds);
            ema_commit(c);
        }

        /* K2: clock recovery + time2 framer (also produces the slicer bits the RLA needs) */
        K2Args k2{};
        k2.g = g; k2.dphi = c->d_dphi; k2.rssi = c->d_rssi; k2.bits = c->d_bits;
        k2.err = c->d_scalars + SC_ERR;
        k2.ckpt = c->d_ckpt; k2.nck = c->nck;
        c->k2clk = k2; c->k2rla = k2;
        K2Args &ka = c->k2clk, &kr = c->k2rla;
        ka.algo = WMBUS_ALGO_T2A; ka.bad = c->d_bad_clk;
        ka.chips = c->d_chips[1]; ka.counts = c->d_counts[1]; ka.sync_seen = c->d_sync_seen[1];
        ka.st_start = c->d_st_start[1]; ka.st_final = c->d_st_final[1]; ka.st_carry = st_carry(c, 1, false);
        kr.algo = WMBUS_ALGO_RLA; kr.bad = c->rla_chains ? c->d_bad : nullptr;
        kr.chips = c->d_chips[0]; kr.counts = c->d_counts[0]; kr.sync_seen = c->d_sync_seen[0];
        kr.st_start = c->d_st_start[0]; kr.st_final = c->d_st_final[0]; kr.st_carry = st_carry(c, 0, false);
        /* One stream, one launch after the other: clock first pass, its unattended re-run rounds, then the run-length framer and its
         * rounds.  (Rounds 2-4 built and measured the alternatives -- the clock re-run lanes and the run-length framer fused into one
         * launch; the run-length framer on a side stream beside the clock rounds -- both shorten a context's chain of launches by
         * 4-5 ms and both LOSE: the job is bound by the ring of demodulation kernels, and more streams or fatter launches only
         * get in its way; DESIGN_HISTORY.md.  Round 5 brought the fused launch back once more, with the demodulation kernel 13 % shorter and
         * a context's chain the bound: 155-161 against 172 Gsamples/s, the demodulation kernel in region 3.8-4.0 instead of 3.3 ms.) */
        const bool rla = c->flags & WM_F_RLA;
        HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
        fr_launch(c, WMBUS_ALGO_T2A, 0xFFFFFFFFu);                 /* the clock kernel's first pass */
        /* (a push of one segment per row has no hand-off inside it: its only lane starts from the carried state -- no rounds to enqueue;
         * a live stream's 2^16-sample pushes are such) */
        for (unsigned r = 0; r < c->fr_rounds && c->opt_rounds && g.nseg[1] > 1u; r++) { fr_verify(c, WMBUS_ALGO_T2A, SC_CLK + r); fr_launch(c, WMBUS_ALGO_T2A, SC_CLK + r); }
        HIPCHK(c, hipEventRecord(c->ev[8], c->stream));
        if (rla) {
            fr_launch(c, WMBUS_ALGO_RLA, 0xFFFFFFFFu);
            for (unsigned r = 1; r < c->rla_rounds && c->opt_rounds && g.nseg[0] > 1u; r++) { fr_verify(c, WMBUS_ALGO_RLA, SC_RLA + r); fr_launch(c, WMBUS_ALGO_RLA, SC_RLA + r); }
        } else HIPCHK(c, hipMemsetAsync(c->d_counts[0], 0, (size_t)2 * c->S * c->nseg_cap[0] * sizeof(uint32_t), c->stream));
        c->rla_fin = c->rla_rounds;
        if (rla) {
            /* the last verification of either framer and both carries in one launch (k2_finish) */
            K2Finish f{};
            const K2Args *ka2[2] = {&c->k2rla, &c->k2clk};
            for (int al = 0; al < 2; al++) {
                f.st_start[al] = (const uint32_t *)ka2[al]->st_start; f.st_final[al] = (const uint32_t *)ka2[al]->st_final;
                f.words[al] = (uint32_t)((al ? sizeof(WmClkState) : sizeof(WmRlaState)) / 4);
                f.carry[al] = (uint32_t *)st_carry(c, al, true); f.verify[al] = 1u; f.do_carry[al] = 1u;
            }
            f.list[0] = c->d_list2; f.list[1] = c->d_list; f.n_list[0] = c->d_scalars + SC_RLA + c->rla_fin; f.n_list[1] = c->d_scalars + SC_CLK + c->fr_rounds;
            f.bad[0] = c->rla_chains ? c->d_bad : nullptr; f.bad[1] = c->d_bad_clk;
            f.heads[0] = list_walks_chains(c, WMBUS_ALGO_RLA, SC_RLA + c->rla_fin); f.heads[1] = list_walks_chains(c, WMBUS_ALGO_T2A, SC_CLK + c->fr_rounds);
            const uint32_t most = std::max(2u * g.S, 2u * std::max(g.nseg[0], g.nseg[1]) * g.S);
            hipLaunchKernelGGL(k2_finish, dim3((most + 255u) / 256u), dim3(256), 0, c->stream, g, f);
            HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        } else {
            fr_verify(c, WMBUS_ALGO_T2A, SC_CLK + c->fr_rounds);
            HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
            fr_carry(c);
        }
        c->committed = true;
    }
    /* the next push sees the last 4096 staged bytes in front of it */
    hipLaunchKernelGGL(k_copy_hist, dim3(c->S), dim3(256), 0, c->stream, win, c->in_stride, (uint64_t)nbytes, c->d_hist, (uint64_t)WM_HIST_BYTES);
    HIPCHK(c, hipGetLastError());
    c->fill = (c->fill + 1) % c->n_win;
    c->n0 += n_new;
    c->push_seq++;
    c->in_flight = true; c->back_due = true;
    return WMBUS_OK;
}

/* BACK of a push: what needs the host decoders' state after the previous push -- the chips half-received telegrams are
 * still owed -- then K3 on the settled chip streams and the scalars for the host. */
static int enqueue_back_impl(wmbus_ctx *c)
{
    if (!c->back_due) return WMBUS_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->back_due = false;
    if (c->last.M > 0) {
        for (uint32_t s = 0; s < c->S; s++)
            for (int ch = 0; ch < 2; ch++)
                for (int al = 0; al < 2; al++)
                    c->h_pending[(al * 2 + ch) * c->S + s] = c->decs[((size_t)s * 2 + ch) * 2 + al].owed;
        HIPCHK(c, hipMemcpyAsync(c->d_pending, c->h_pending, 4 * c->S * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
        const int rc = launch_k3(c, false);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, SC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    }
    return WMBUS_OK;
}

/* a failure after the first enqueue leaves half a push on the stream and the carried state undefined */
static int enqueue_front(wmbus_ctx *c, size_t nbytes)
{
    c->enqueued = false;
    const int rc = enqueue_front_impl(c, nbytes);
    if (rc && c->enqueued) c->poisoned = true;
    return rc;
}
static int enqueue_back(wmbus_ctx *c)
{
    const int rc = enqueue_back_impl(c);
    if (rc) c->poisoned = true;
    return rc;
}

int wmbus_process(wmbus_ctx *c, size_t nbytes)
{
    if (!c) return WMBUS_EINVAL;
    const int rc = enqueue_front(c, nbytes);
    return rc ? rc : enqueue_back(c);
}

/* The optimistic rounds of wmbus_process left work: finish it round by round with the host in the loop (rare: a
 * cascade of hand-off failures longer than the rounds enqueued), then redo what depended on it. */
static int finish_slowly(wmbus_ctx *c, bool ema_left, bool clk_left, bool rla_left, bool rs_left = false)
{
    auto leftovers = [&](uint32_t *n) -> int {
        HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, SC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *n = c->h_scalars[SC_SLOW];
        return 0;
    };
    const uint32_t max_rounds = std::max({c->ntiles, c->last.nseg[0], c->last.nseg[1]}) + 4;
    if (rs_left) {
        /* RSSI on demand, and a lane whose value is read could not prove it (wm_k1_demod.h: constant input can keep the bracket
         * open): this push's RSSI the long way -- the full kernel over every tile, its hand-offs verified against the carried
         * state and repaired like any other context's */
        uint32_t n = 0;
        int rc = launch_k1_any(c, c->k1a, dim3(c->ntiles, c->S));
        if (rc) return rc;
        hipLaunchKernelGGL(k_copy_word, dim3(1), dim3(1), 0, c->stream, c->d_scalars + SC_SLOW, 0u);
        ema_verify(c, SC_SLOW);
        if ((rc = leftovers(&n))) return rc;
        ema_left = n != 0;
        if (!ema_left) ema_commit(c);
        c->rs_full_now = true;
    }
    if (ema_left) {
        uint32_t cnt = rs_left ? (uint32_t)SC_SLOW : SC_EMA + c->ema_rounds, n = 0;          /* the list of the last verify is still in d_list_ema */
        for (uint32_t round = 0;; round++) {
            c->tim.ema_retries += c->h_scalars[cnt];
            int rc = ema_repair(c, cnt);
            if (rc) return rc;
            hipLaunchKernelGGL(k_copy_word, dim3(1), dim3(1), 0, c->stream, c->d_scalars + SC_SLOW, 0u);    /* after the repair has read it */
            ema_verify(c, SC_SLOW);
            if ((rc = leftovers(&n))) return rc;
            if (n == 0) break;
            if (round > max_rounds) return fail(c, WMBUS_EDEVICE, "RSSI filter hand-off repair did not converge");
            cnt = SC_SLOW;
        }
        ema_commit(c);
    }
    for (int algo = 1; algo >= 0; algo--) {                    /* clock first: with -o the slicer words depend on it */
        if (!(algo ? clk_left : rla_left)) continue;
        uint32_t cnt = algo ? SC_CLK + c->fr_rounds : SC_RLA + c->rla_fin, n = 0;
        for (uint32_t round = 0;; round++) {
            (algo ? c->tim.clock_reruns : c->tim.rla_reruns) += c->h_scalars[cnt];
            fr_launch(c, algo, cnt);
            hipLaunchKernelGGL(k_copy_word, dim3(1), dim3(1), 0, c->stream, c->d_scalars + SC_SLOW, 0u);
            fr_verify(c, algo, SC_SLOW);
            int rc = leftovers(&n);
            if (rc) return rc;
            if (n == 0) break;
            if (round > max_rounds) return fail(c, WMBUS_EDEVICE, "segment verification did not converge");
            cnt = SC_SLOW;
        }
        if (algo == 1 && (c->flags & WM_F_DC) && (c->flags & WM_F_RLA)) {
            /* the run-length framer ran on slicer words that the clock re-runs have just replaced: all of it again */
            fr_launch(c, WMBUS_ALGO_RLA, 0xFFFFFFFFu);
            hipLaunchKernelGGL(k_copy_word, dim3(1), dim3(1), 0, c->stream, c->d_scalars + SC_RLA + c->rla_fin, 0u);
            fr_verify(c, WMBUS_ALGO_RLA, SC_RLA + c->rla_fin);
            HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, SC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            rla_left = c->h_scalars[SC_RLA + c->rla_fin] != 0;
        }
    }
    fr_carry(c);
    int rc = launch_k3(c, true);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, SC_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    /* the chips have changed, so have the bursts and the tiles they touch: one of those may be unprovable in turn */
    if (c->rs_this && !c->rs_full_now && c->h_scalars[SC_RS_FAIL]) return finish_slowly(c, false, false, false, true);
    return 0;
}

namespace {
struct Entry { uint32_t idx; uint8_t raw; };          /* one candidate telegram of a push: a WmPkt (decoded on the GPU) or a WmBurstHdr (chips) */
struct EntryKey { uint32_t stream, chip0; uint8_t chain, algo, cont; };
}

static EntryKey entry_key(const wmbus_ctx *c, const Entry &e)
{
    if (e.raw) { const WmBurstHdr &h = c->h_hdr[e.idx]; return EntryKey{h.stream, h.chip0, h.chain, h.algo, (uint8_t)(h.flags & 1u)}; }
    const WmPkt &p = c->h_pkts[e.idx];
    return EntryKey{p.stream, p.chip0, p.chain, p.algo, 0};
}

/* Candidate telegrams of the (stream, chain, framer) groups in order[lo, hi), each group in chip order: what the
 * reference's decoder would do with them -- an access code that passes while the decoder is busy is ignored
 * (t1_c1_packet_decoder.h:272-278), a telegram the GPU has assembled is stripped and formatted, a burst cut by the end
 * of the push goes chip by chip through the persistent host decoder. */
/* TIMESTAMP field of a line.  The reference stamps a telegram when its last chip is processed (t1_c1_packet_decoder.h:
 * 390,458, s1_packet_decoder.h:229), which behind a live SDR is the moment that sample arrived.  A push is handed over
 * when its LAST sample has arrived, so a telegram completed by decimated sample m of the push was on the air
 * (m_end - 1 - m) / 800 kHz earlier (the decimated rate is 800 kS/s whatever -d is, rtl_wmbus.c:1296). */
static void line_timestamp(const wmbus_ctx *c, uint64_t sample, const char *ts_fixed, char *ts, size_t cap)
{
    if (ts_fixed) { snprintf(ts, cap, "%s", ts_fixed); return; }
    const uint64_t back = c->done.m_end > sample ? c->done.m_end - 1u - sample : 0u;
    const int64_t us = (int64_t)c->done.arrival.tv_sec * 1000000 + c->done.arrival.tv_usec - (int64_t)(back * 5u / 4u);   /* 1.25 us per decimated sample */
    wm_timestamp_at(ts, cap, (long)(us / 1000000), (long)(us % 1000000));
}

static void decode_stream_range(wmbus_ctx *c, const std::vector<Entry> &order, size_t lo, size_t hi,
                                LinePart &part, uint32_t part_no, const char *ts_fixed)
{
    std::vector<LineRec> &out = part.recs;
    auto keep = [&](LineRec &r, const char *line, size_t n) {
        r.part = part_no; r.off = (uint32_t)part.text.size(); r.len = (uint32_t)n;
        part.text.append(line, n);
        out.push_back(r);
    };
    char line[1024], ts[64];
    uint8_t pkt[WM_PKT_MAXBYTES + 4];
    uint32_t seq = 0;
    size_t i = lo;
    while (i < hi) {
        const EntryKey k0 = entry_key(c, order[i]);
        HostDecoder &hd = c->decs[((size_t)k0.stream * 2 + k0.chain) * 2 + k0.algo];
        const char *tag = c->cfg.show_algorithm ? (k0.algo == WMBUS_ALGO_RLA ? "rla;" : "t2a;") : "";
        uint64_t next_free = 0;                         /* first chip the decoder has not consumed */
        size_t j = i;
        for (; j < hi; j++) {
            const EntryKey kj = entry_key(c, order[j]);
            if (kj.stream != k0.stream || kj.chain != k0.chain || kj.algo != k0.algo) break;
            if (!order[j].raw) {
                /* ---- the whole burst was inside the push: the GPU has run the decoder over it ---- */
                const WmPkt &p = c->h_pkts[order[j].idx];
                if (hd.owed != 0 || p.chip0 < next_free) continue;     /* the access code passed while the decoder was busy */
                next_free = (uint64_t)p.chip0 + p.consumed;
                if (p.status != WM_PKT_DONE) continue;
                const unsigned nb = std::min<unsigned>(std::max<unsigned>(p.L, 2u), WM_PKT_MAXBYTES);
                /* the reference's decoder is memset on reset (t1_c1_packet_decoder.h:268,276): bytes a short telegram never
                 * stored read as zero in the ident field of its line (get_serial looks at bytes 4..7 whatever L is) */
                memset(pkt, 0, sizeof pkt);
                memcpy(pkt, c->h_bytes + p.off, nb);
                line_timestamp(c, p.sample, ts_fixed, ts, sizeof ts);
                const int ok = (p.flags & WM_PKTF_CRC_OK) != 0;
                const size_t n = wm_packet_format(p.chain ? WM_MODE_S1 : WM_MODE_T1C1, (p.flags & WM_PKTF_C1) != 0, (p.flags & WM_PKTF_FRAME_B) != 0,
                                                  (p.flags & WM_PKTF_ERR3OF6) != 0, ok, p.L, pkt, p.pkt_rssi, p.rssi_now, tag, ts, line, sizeof line);
                LineRec r; r.sample = p.sample; r.stream = p.stream; r.chain = p.chain; r.algo = p.algo;
                r.crc_ok = (uint8_t)ok; r.seq = seq++;
                keep(r, line, n);
                continue;
            }
            const WmBurstHdr &h = c->h_hdr[order[j].idx];
            const bool cont = h.flags & 1u;
            if (cont ? hd.owed == 0 : (hd.owed != 0 || h.chip0 < next_free)) continue;   /* the access code passed while the decoder was busy */
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
                    line_timestamp(c, h.pos0 + (word >> 11), ts_fixed, ts, sizeof ts);
                    const size_t n = wm_decoder_format(&hd.dec, tag, ts, rssi, line, sizeof line, &ok);
                    LineRec r; r.sample = h.pos0 + (word >> 11); r.stream = h.stream; r.chain = h.chain; r.algo = h.algo;
                    r.crc_ok = (uint8_t)ok; r.seq = seq++;
                    keep(r, line, n);
                    st = WM_DEC_IDLE;
                }
                if (st == WM_DEC_IDLE) { k++; break; }
            }
            next_free = (uint64_t)h.chip0 + k;
            const bool cut = st == WM_DEC_RECEIVING;
            if (cut && h.n_chips != h.avail) {                 /* device under-estimated the burst: a bug */
                uint32_t none = 0;
                c->short_burst.compare_exchange_strong(none, 1u + order[j].idx);
            }
            hd.owed = cut ? std::max(1u, wm_decoder_chips_owed(&hd.dec)) : 0u;
            hd.fed = c->done.seq;
        }
        i = j;
    }
}

/* First half of wmbus_collect: wait for the push, finish leftover hand-off rounds, read the counters.  Afterwards the
 * candidate telegrams sit in pinned host memory and nothing of this push is left on the GPU. */
static int wait_gpu(wmbus_ctx *c)
{
    c->done.valid = false;
    if (!c->in_flight) return WMBUS_OK;
    if (c->back_due) { const int rc = enqueue_back(c); if (rc) return rc; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->in_flight = false;
    c->tim = wmbus_timing{};
    if (c->last.M > 0) {
        float ms = 0;
        hipEventElapsedTime(&ms, c->ev[3], c->ev[4]); c->tim.demod_ms = ms;
        hipEventElapsedTime(&ms, c->ev[2], c->ev[3]); c->tim.turn_wait_ms = ms;  /* on the GPU: behind the other contexts' demodulation kernels */
        hipEventElapsedTime(&ms, c->ev[0], c->ev[8]); c->tim.clock_ms = ms;      /* the clock kernel's launches */
        hipEventElapsedTime(&ms, c->ev[8], c->ev[1]); c->tim.rla_ms = ms;        /* the run-length framer's */
        if (c->rs_this) { hipEventElapsedTime(&ms, c->ev[9], c->ev[10]); c->tim.rssi_ms = ms; }
        hipEventElapsedTime(&ms, c->ev[5], c->ev[6]); c->tim.gather_ms = ms;
        hipEventElapsedTime(&ms, c->ev[6], c->ev[7]); c->tim.d2h_ms = ms;
        hipEventElapsedTime(&ms, c->ev[2], c->ev[7]); c->tim.gpu_total_ms = ms;
        const uint32_t *hs = c->h_scalars;
#ifdef WM_DBG_WALK_STATS
        if (c->push_seq == 4) {
            unsigned h[2][32];
            hipMemcpyFromSymbol(h, HIP_SYMBOL(wm_walk_hist), sizeof h);
            for (int ch = 0; ch < 2; ch++) {
                fprintf(stderr, "run-length walks so far, %s chain, by segments walked:", ch ? "S1" : "T1/C1");
                for (int i = 1; i < 32; i++) fprintf(stderr, " %u", h[ch][i]);
                fprintf(stderr, "\n");
            }
        }
#endif
        for (unsigned r = 0; r < c->ema_rounds; r++) c->tim.ema_retries += hs[SC_EMA + r];
        for (unsigned r = 0; r < c->fr_rounds; r++) c->tim.clock_reruns += hs[SC_CLK + r];
        for (unsigned r = 0; r < c->rla_fin; r++) c->tim.rla_reruns += hs[SC_RLA + r];
        const bool ema_left = hs[SC_EMA + c->ema_rounds], clk_left = hs[SC_CLK + c->fr_rounds], rla_left = hs[SC_RLA + c->rla_fin];
        const bool rs_left = c->rs_this && hs[SC_RS_FAIL] != 0;
        for (unsigned r = 0; r < 4; r++) {
            c->tim.clock_round[r] = r < c->fr_rounds ? hs[SC_CLK + r] : 0u;
            c->tim.rla_round[r] = r + 1u < c->rla_fin ? hs[SC_RLA + 1u + r] : 0u;     /* the run-length framer's counters start at index 1 */
        }
        c->tim.rssi_mode = c->rs_this ? WMBUS_RSSI_ON_DEMAND : c->rs_od ? WMBUS_RSSI_PAUSED : WMBUS_RSSI_EVERY_SAMPLE;
        if (c->rs_this) {
            c->tim.rssi_tiles = hs[SC_RS_N];
            const uint32_t dense_pm = c->cfg.rssi_dense_pm ? c->cfg.rssi_dense_pm : 200u;
            if ((uint64_t)hs[SC_RS_N] * 1000u > (uint64_t)dense_pm * c->ntiles * c->S) c->rs_pause = 16;
        }
        if (ema_left || clk_left || rla_left || rs_left) {
            const int rc = finish_slowly(c, ema_left, clk_left, rla_left, rs_left);
            if (rc) { c->poisoned = true; return rc; }
            c->tim.slow_path = 1;
            if (c->rs_full_now) c->tim.rssi_mode = WMBUS_RSSI_FELL_BACK;
        }
        const uint32_t err = hs[SC_ERR];
        if (err & WM_ERR_CHIP_OVERFLOW) {            /* a time2 region over its proven bound: a defect, not an input property */
            c->poisoned = true;
            return fail(c, WMBUS_EDEVICE, "internal error: time2 chip region overflow (err=%u)", err);
        }
        /* storage exhausted (spill arena / burst arena): chips or candidate bursts were dropped; every carried state
         * is exact, so the stream goes on -- reported, never fatal (the reference never gives up either) */
        c->tim.warnings = ((err & WM_ERR_CHIP_TRUNC) ? WMBUS_WARN_CHIPS_DROPPED : 0u) | ((err & WM_ERR_BURST_OVERFLOW) ? WMBUS_WARN_BURSTS_DROPPED : 0u);
        c->n_hdr = std::min(hs[SC_NHDR], c->hdr_cap); c->n_words = hs[SC_NWORDS]; c->n_pkts = std::min(hs[SC_NPKTS], c->pkts_cap);
        for (int al = 0; al < 2; al++) for (int ch = 0; ch < 2; ch++) c->tim.chips[ch][al] = hs[SC_CHIPS + al * 2 + ch];
    }
    c->tim.bursts = c->n_hdr + c->n_pkts;
    c->done.n_hdr = c->n_hdr; c->done.n_pkts = c->n_pkts; c->done.m_end = c->last.m0 + c->last.M; c->done.arrival = c->arrival; c->done.seq = c->push_seq;
    c->done.valid = true;
    return WMBUS_OK;
}

/* Second half: candidate telegrams -> lines (host packet decoders, strip, format, merge).  Touches pinned host memory
 * and the decoders only, so the NEXT push's front may already be on the GPU. */
static int decode_host(wmbus_ctx *c)
{
    c->lines.clear(); c->text.clear();
    c->short_burst.store(0);
    if (!c->done.valid) return WMBUS_OK;
    c->done.valid = false;
    const double t0 = now_ms();
    const uint32_t n_hdr = c->done.n_hdr, n_pkts = c->done.n_pkts;

    /* candidates in the order the decoders take them: (capture, chain, framer), a continuation first, then by chip.  The key is made
     * ONCE per candidate (the comparator used to look every candidate's record up in the page-locked result area at every comparison) */
    std::vector<Entry> order(n_hdr + n_pkts);
    {
        struct Keyed { uint64_t hi; uint32_t lo; Entry e; };
        std::vector<Keyed> ks(n_hdr + n_pkts);
        for (uint32_t i = 0; i < n_hdr + n_pkts; i++) {
            const Entry e = i < n_hdr ? Entry{i, 1} : Entry{i - n_hdr, 0};
            const EntryKey k = entry_key(c, e);
            ks[i] = Keyed{((uint64_t)k.stream << 8) | ((uint64_t)k.chain << 4) | ((uint64_t)k.algo << 1) | (k.cont ? 0u : 1u), k.chip0, e};
        }
        std::stable_sort(ks.begin(), ks.end(), [](const Keyed &x, const Keyed &y) { return x.hi != y.hi ? x.hi < y.hi : x.lo < y.lo; });
        for (size_t i = 0; i < ks.size(); i++) order[i] = ks[i].e;
    }
    unsigned nt = c->cfg.host_threads ? c->cfg.host_threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (order.size() < 4096) nt = 1;
    const char *tsf = c->cfg.fixed_timestamp ? "TS" : nullptr;
    /* more pieces than threads (cut at stream boundaries): the decoders' cost per stream is uneven */
    const unsigned np = nt == 1 ? 1 : 4 * nt;
    std::vector<LinePart> parts(np);
    if (np == 1) decode_stream_range(c, order, 0, order.size(), parts[0], 0u, tsf);
    else {
        std::vector<size_t> cut(np + 1, order.size());
        cut[0] = 0;
        for (unsigned t = 1; t < np; t++) {
            size_t p = order.size() * t / np;
            while (p < order.size() && p > 0 && entry_key(c, order[p]).stream == entry_key(c, order[p - 1]).stream) p++;
            cut[t] = std::max(p, cut[t - 1]);
        }
        if (!c->pool || c->pool->size() + 1 != nt) c->pool.reset(new WorkerPool(nt - 1));
        c->pool->run(np, [&](unsigned t) { decode_stream_range(c, order, cut[t], cut[t + 1], parts[t], t, tsf); });
    }
    /* Burst storage ran out (a warning): a half-received telegram whose continuation was among the dropped bursts would
     * otherwise wait for it for ever and take the FIRST chips of the next push for its own -- it is lost, like the
     * bursts that were dropped. */
    if (c->tim.warnings & WMBUS_WARN_BURSTS_DROPPED)
        for (auto &hd : c->decs)
            if (hd.owed != 0 && hd.fed != c->done.seq) { wm_decoder_abort(&hd.dec); hd.owed = 0; }
    /* stdout order of the reference: by completing sample, then T1/C1 before S1, run-length before time2 */
    std::vector<LineRec> all;
    {
        size_t n_all = 0, n_text = 0;
        for (auto &p : parts) { n_all += p.recs.size(); n_text += p.text.size(); }
        all.reserve(n_all); c->text.reserve(n_text); c->lines.reserve(n_all);
        for (auto &p : parts) all.insert(all.end(), p.recs.begin(), p.recs.end());
    }
    std::stable_sort(all.begin(), all.end(), [](const LineRec &a, const LineRec &b) {
        if (a.stream != b.stream) return a.stream < b.stream;
        if (a.sample != b.sample) return a.sample < b.sample;
        if (a.chain != b.chain) return a.chain < b.chain;
        if (a.algo != b.algo) return a.algo < b.algo;
        return a.seq < b.seq;
    });
    /* Options (off by default: the drop-in prints what the reference prints).  Both framers work on every burst, so a
     * clean telegram is printed twice, once per framer (README.md:105-108 "You will eventually get two identical
     * datagrams"): dedup_twins drops the later of two lines of one capture and mode that carry the same payload, come
     * from different framers and complete within one longest-telegram time of each other.  only_crc_ok drops what a
     * consumer like wmbusmeters would discard anyway. */
    if (c->cfg.dedup_twins || c->cfg.only_crc_ok) {
        if (c->twins.empty()) c->twins.assign((size_t)c->S * 2 * 2, wm_twin{0, 0, 0, 0});
        std::vector<LineRec> kept;
        kept.reserve(all.size());
        for (auto &r : all) {
            if (c->cfg.only_crc_ok && !r.crc_ok) continue;
            if (c->cfg.dedup_twins && wm_twin_check(&c->twins[((size_t)r.stream * 2 + r.chain) * 2], r.chain, r.algo, r.sample, parts[r.part].text.data() + r.off, r.len)) continue;
            kept.push_back(r);
        }
        all.swap(kept);
    }
    for (auto &r : all) {
        wmbus_line l{};
        l.stream = r.stream; l.chain = r.chain; l.algo = r.algo; l.crc_ok = r.crc_ok; l.sample = r.sample;
        l.text_off = (uint32_t)c->text.size(); l.text_len = r.len;
        c->text.append(parts[r.part].text, r.off, r.len);
        c->lines.push_back(l);
    }
    c->tim.host_decode_ms = (float)(now_ms() - t0);
    if (const uint32_t sb = c->short_burst.load()) {       /* formatted once, after the pool has joined */
        const WmBurstHdr &h = c->h_hdr[sb - 1u];
        return fail(c, WMBUS_EDEVICE, "burst too short: stream %u chain %u algo %u chip %u", h.stream, h.chain, h.algo, h.chip0);
    }
    return WMBUS_OK;
}

int wmbus_collect(wmbus_ctx *c)
{
    if (!c) return WMBUS_EINVAL;
    const int rc = wait_gpu(c);
    if (rc) { c->lines.clear(); c->text.clear(); return rc; }
    return decode_host(c);
}

/* The host half of wmbus_collect -- sort, strip / format, merge -- over the records of the context's LAST push again, `reps` times, with no
 * GPU work at all: how many lines per second the host side sustains when several ranks' contexts decode at once (eight
 * GPUs' worth of lines through one host; tools/host_replay.py).  The persistent packet decoders are put back after every repetition; the
 * lines of the last one stay readable through wmbus_lines.  Returns the number of lines of one repetition, or a negative code. */
long wmbus_debug_replay_decode(wmbus_ctx *c, unsigned reps, double *seconds)
{
    if (!c) return WMBUS_EINVAL;
    const std::vector<HostDecoder> saved = c->decs;
    const double t0 = now_ms();
    long n = 0;
    for (unsigned r = 0; r < reps; r++) {
        c->decs = saved;
        c->done.valid = true;
        const int rc = decode_host(c);
        if (rc) return rc;
        n = (long)c->lines.size();
    }
    if (seconds) *seconds = (now_ms() - t0) * 1e-3;
    return n;
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
    /* a context without debug views computes the RSSI only in the tiles a packet decoder reads (RSSI on demand): elsewhere its rows hold
     * nothing, so this view shows 0 there instead of what an earlier push left */
    hipLaunchKernelGGL(k4_flatten, dim3(1), dim3(1), 0, c->stream, c->last, (uint32_t)algo, c->d_chips[algo], c->d_counts[algo],
                       /* no RSSI row where the last push computed it on demand only (a paused or fallen-back push ran the full pass: the row is valid) */
                       c->rs_this && !c->rs_full_now ? (const uint8_t *)nullptr : c->d_rssi, c->cap[algo], (uint32_t)chain, stream, d_dst, d_pos, (uint32_t)max_elems, d_n);
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

int wmbus_selftest_fir(int device, const float *x, float *out, size_t n)
{
    if (wmbus_device_count() <= device || hipSetDevice(device) != hipSuccess) return WMBUS_ENODEVICE;
    if (!x || !out || n == 0 || n > (size_t)K1Geo::T || (n & 3)) return WMBUS_EINVAL;
    float *d_x = nullptr, *d_o = nullptr;
    if (hipMalloc((void **)&d_x, (n + 48) * sizeof(float)) != hipSuccess || hipMalloc((void **)&d_o, 2 * n * sizeof(float)) != hipSuccess) {
        hipFree(d_x); hipFree(d_o);
        return WMBUS_ENOMEM;
    }
    int rc = hipMemcpy(d_x, x, (n + 48) * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? WMBUS_OK : WMBUS_EDEVICE;
    if (rc == WMBUS_OK) {
        hipLaunchKernelGGL(k_selftest_fir, dim3(1), dim3(256), 0, 0, d_x, d_o, (uint32_t)n);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d_o, 2 * n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = WMBUS_EDEVICE;
    }
    hipFree(d_x); hipFree(d_o);
    return rc;
}

#ifdef WM_K1_STAMPS
/* -DWM_K1_STAMPS builds only (tools/gpu_k1_stamps.py): the stage intervals the demodulation kernel's waves have left
 * (wm_k1_demod.h), summed over the slots into out8[0 .. 6], optionally cleared. */
int wmbus_debug_k1_stamps(unsigned long long *out8, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return WMBUS_EDEVICE;
    const size_t n = (size_t)WM_K1_STAMP_SLOTS * 8u;
    if (out8) {
        std::vector<unsigned int> h(n);
        if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(wm_k1_stamp_buf), n * sizeof(unsigned int)) != hipSuccess) return WMBUS_EDEVICE;
        for (int k = 0; k < 8; k++) out8[k] = 0;
        for (size_t i = 0; i < n; i++) out8[i & 7u] += h[i];
    }
    if (reset) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(wm_k1_stamp_buf)) != hipSuccess || hipMemset(p, 0, n * sizeof(unsigned int)) != hipSuccess) return WMBUS_EDEVICE;
    }
    return WMBUS_OK;
}
#endif

}  // extern "C"

#include "wm_batch.h"
