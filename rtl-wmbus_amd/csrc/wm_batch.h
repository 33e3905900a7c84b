/* wm_batch.h -- wmbus_batch_*: many captures on one device (include/wmbus_hip.h).  Part of wm_api.hip's translation unit
 * (it drives the two halves of a push, enqueue_front / enqueue_back and wait_gpu / decode_host, directly).
 *
 * What the headline rate needs lived in bench.py for two rounds: several receiver contexts per GPU, each on its own host
 * thread, free-running through their pushes so that one context's latency-bound framer kernels and host decoding are
 * covered by the other contexts' demodulation kernels (which the library already orders one after the other on the GPU,
 * K1Chain).  It is product code now -- the batch CLI (rtl_wmbus_hip FILE...) and bench.py both go through it -- and it
 * adds one thing a caller of the plain context API cannot do: the NEXT push's front (demodulation, framers) is enqueued
 * BEFORE the previous push is decoded on the host, so host decoding leaves every context's chain of dependent work. */
#ifndef WM_BATCH_H
#define WM_BATCH_H

struct wmbus_batch {
    wmbus_cfg cfg{};
    char err[256] = {0};
    std::vector<wmbus_ctx *> ctx;
    std::vector<unsigned> first, count;                 /* stream range of every context */
    std::vector<uint8_t *> slab;                        /* per context: its page-locked staging slab (host-sourced runs), allocated on first use */
    std::mutex out_lock, err_lock;
    std::atomic<bool> stop{false};
    int rc = 0;
    bool opened = false;                                /* wmbus_batch_open succeeded: every planned context exists */
};

namespace {

int batch_fail(wmbus_batch *b, int code, const char *fmt, ...)
{
    std::lock_guard<std::mutex> lk(b->err_lock);
    if (b->rc == 0) {                                       /* the first error is the one reported */
        va_list ap; va_start(ap, fmt);
        vsnprintf(b->err, sizeof b->err, fmt, ap);
        va_end(ap);
        b->rc = code;
    }
    b->stop.store(true);
    return code;
}

struct BatchTotals { std::atomic<uint64_t> samples{0}, lines{0}; std::atomic<unsigned> pushes{0}, warnings{0}; };

/* One context's run: pushes until its source ends.  GPU work of push k+1 (front) overlaps the host decode of push k;
 * in host-sourced runs the bytes of push k+1 are read and staged into the context's other input window while push k is
 * in flight. */
void batch_worker(wmbus_batch *b, unsigned i, const wmbus_batch_io *io, BatchTotals *tot)
{
    wmbus_ctx *c = b->ctx[i];
    const unsigned S = b->count[i], s0 = b->first[i];
    const size_t pitch = b->cfg.max_push_bytes;
    unsigned passes_left = io->fill ? 0u : io->passes;
    int cur = 0;
    auto source = [&](int k) -> size_t {                    /* bytes of the next push (0: the input has ended), staged if host-sourced */
        if (b->stop.load()) return 0;
        if (!io->fill) { if (passes_left == 0) return 0; passes_left--; return io->resident_bytes; }
        (void)k;
        /* ONE page-locked slab per context (round 5; two until then): the copies of a push leave the slab long before the next
         * push is read into it -- they were queued an iteration ago and the demodulation kernel behind them has been enqueued
         * since -- so the refill only has to make sure (one event, normally signalled already), and the staging the driver has
         * to pin, at about 5 GB/s and inside the decode time, is files x push bytes instead of twice that. */
        if (!io->self_staged && !b->slab[i]) {
            /* Page-locked staging is allocated HERE, by the context's own thread when it first needs the slab: the driver
             * pins one allocation at a time (16 GB for 1024 files of 8 MiB pushes: 2.4-3 s), so a batch that pinned
             * everything before its first push spent longer setting up than decoding; now the first contexts decode while
             * the others' slabs are still being pinned. */
            hipSetDevice(b->cfg.device);
            if (hipHostMalloc((void **)&b->slab[i], (size_t)S * pitch) != hipSuccess) {
                b->slab[i] = nullptr;
                batch_fail(b, WMBUS_ENOMEM, "batch: cannot allocate %zu bytes of page-locked staging", (size_t)S * pitch);
                return 0;
            }
        }
        if (!io->self_staged && hipEventSynchronize(c->ev_staged) != hipSuccess) { batch_fail(b, WMBUS_EDEVICE, "batch: context %u: waiting for the staging copies failed", i); return 0; }
        const size_t n = io->fill(io->user, s0, S, io->self_staged ? nullptr : b->slab[i], pitch, pitch);
        if (n == 0) return 0;
        if (n > pitch || n % WMBUS_BLOCK_BYTES) { batch_fail(b, WMBUS_EINVAL, "batch: the source returned %zu bytes (multiple of 4096, at most %zu)", n, pitch); return 0; }
        if (!io->self_staged) {
#ifndef WM_STAGE_PER_STREAM
            { const int src = wm_stage_all(c, b->slab[i], pitch, n); if (src) { batch_fail(b, src, "batch: context %u: %s", i, c->err); return 0; } }      /* the code as it came (an argument error is not a device error) */
#else                                                       /* one copy per stream (until round 5; A/B) */
            for (unsigned s = 0; s < S; s++)
                if (wmbus_stage(c, s, b->slab[i] + (size_t)s * pitch, n)) { batch_fail(b, WMBUS_EDEVICE, "batch: context %u: %s", i, c->err); return 0; }
#endif
        }
        return n;
    };
    size_t n_cur = source(cur);
    bool have_prev = false;
    while (n_cur || have_prev) {                            /* after an error elsewhere source() returns 0: the loop winds down */
        int rc = 0;
        if (n_cur) rc = enqueue_front(c, n_cur);
        if (!rc && have_prev) {
            rc = decode_host(c);
            if (!rc) {
                tot->lines += c->lines.size();
                tot->warnings |= c->tim.warnings;
                if (io->lines) {
                    /* the lines carry batch-wide stream numbers for the sink (and are put back: the context's own view stays local) */
                    for (auto &l : c->lines) l.stream += s0;
                    {
                        std::lock_guard<std::mutex> lk(b->out_lock);
                        io->lines(io->user, s0, S, c->lines.data(), c->lines.size(), c->text.c_str(), &c->tim);
                    }
                    for (auto &l : c->lines) l.stream -= s0;
                }
            }
        }
        if (!rc && n_cur) rc = enqueue_back(c);
        if (rc) { batch_fail(b, rc, "batch: context %u: %s", i, c->err); break; }
        const size_t n_next = n_cur ? source(cur ^ 1) : 0;
        if (n_cur) {
            rc = wait_gpu(c);
            if (rc) { batch_fail(b, rc, "batch: context %u: %s", i, c->err); break; }
            tot->samples += (uint64_t)S * (n_cur / 2);
            tot->pushes++;
            have_prev = true;
        } else have_prev = false;
        n_cur = n_next; cur ^= 1;
    }
    if (c->in_flight) wait_gpu(c);                          /* an aborted run leaves nothing on the stream */
}

}  // namespace

extern "C" {

const char *wmbus_batch_last_error(const wmbus_batch *b) { return b ? b->err : "null batch"; }
unsigned wmbus_batch_contexts(const wmbus_batch *b) { return b ? (unsigned)b->ctx.size() : 0u; }

wmbus_ctx *wmbus_batch_context(wmbus_batch *b, unsigned i, unsigned *first_stream, unsigned *n_streams)
{
    if (!b || i >= b->ctx.size()) return nullptr;
    if (first_stream) *first_stream = b->first[i];
    if (n_streams) *n_streams = b->count[i];
    return b->ctx[i];
}

void wmbus_batch_close(wmbus_batch *b)
{
    if (!b) return;
    for (auto *c : b->ctx) wmbus_close(c);
    for (auto *p : b->slab) if (p) hipHostFree(p);
    delete b;
}

/* How a batch is split (no device needed): the number of contexts and, if `counts` is given, the captures of each. */
unsigned wmbus_batch_plan(const wmbus_cfg *cfg, unsigned contexts, unsigned *counts, unsigned cap)
{
    if (!cfg || cfg->n_streams < 1) return 0;
    const unsigned S = cfg->n_streams;
    /* Contexts of whole 64-capture waves where the batch allows it (the clock kernel's cooperative loads need that); by
     * default 8 of them, at most one per 64 captures: 8 x 128 for the 1024 captures of the headline configuration
     * (4 / 6 / 10 / 12 / 16 contexts measured 96 / 111 / 141 / 120 / 117 against 144 Gsamples/s with 8, DESIGN_HISTORY.md section 8).
     * (rounds 4-5 gave tolerance mode twelve -- the demodulation kernel is a third shorter there and a context's chain of framer launches
     * was the bound: 167 against 162 Gsamples/s.  With the clock recovery in its systolic form the chain is 4 ms shorter and eight are
     * ahead again: 211-212 against 206-209, round 6.) */
    unsigned nctx = contexts ? contexts : std::min(8u, std::max(1u, S / 64u));
    nctx = std::max(1u, std::min(nctx, S));
    /* whole groups of 64 wherever the batch has that many captures per context; a remainder (S not a multiple of 64) rides
     * with the last context, which alone then takes the clock kernel's lane-private load path */
    const unsigned gran = S / 64u >= nctx ? 64u : 1u, units = S / gran, rest = S - units * gran;
    for (unsigned i = 0; i < nctx && counts && i < cap; i++)
        counts[i] = gran * (units / nctx + (i < units % nctx ? 1u : 0u)) + (i + 1 == nctx ? rest : 0u);
    return nctx;
}

int wmbus_batch_open(const wmbus_cfg *cfg, unsigned contexts, wmbus_batch **out)
{
    if (!cfg || !out) return WMBUS_EINVAL;
    *out = nullptr;
    wmbus_runtime_init();                                   /* hardware queues for the contexts' streams, if HIP has not started yet */
    wmbus_batch *b = new wmbus_batch();
    b->cfg = *cfg;
    *out = b;                                               /* the caller reads the message, then closes */
    const unsigned S = cfg->n_streams;
    if (S < 1) return batch_fail(b, WMBUS_EINVAL, "batch: n_streams must be >= 1");
    std::vector<unsigned> plan(S);
    const unsigned nctx = wmbus_batch_plan(cfg, contexts, plan.data(), S);
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 16;
    /* the contexts are opened side by side (round 6: one after the other it was 0.06 s each -- allocations, page-locked result areas,
     * initial fills --, 0.3 s of a 320-file batch that decodes in 0.23 s) */
    std::vector<wmbus_ctx *> opened(nctx, nullptr);
    std::vector<int> rcs(nctx, 0);
    std::vector<std::thread> openers;
    for (unsigned i = 0; i < nctx; i++)
        openers.emplace_back([&, i] {
            wmbus_cfg cc = *cfg;
            cc.n_streams = plan[i];
            /* host decoder threads: the contexts decode at different times, so the box is shared 2 x oversubscribed */
            if (cc.host_threads == 0) cc.host_threads = std::max(2u, std::min(16u, 2u * hw / std::max(1u, nctx)));
            rcs[i] = wmbus_open(&cc, &opened[i]);
        });
    for (auto &t : openers) t.join();
    for (unsigned i = 0; i < nctx; i++)
        if (rcs[i]) {
            batch_fail(b, rcs[i], "batch: context %u of %u (%u captures): %s", i, nctx, plan[i], opened[i] ? opened[i]->err : "out of memory");
            for (wmbus_ctx *c : opened) wmbus_close(c);
            return rcs[i];
        }
    unsigned at = 0;
    for (unsigned i = 0; i < nctx; i++) {
        b->ctx.push_back(opened[i]); b->first.push_back(at); b->count.push_back(plan[i]);
        at += plan[i];
    }
    b->slab.assign(nctx, nullptr);
    b->opened = true;
    return WMBUS_OK;
}

static int batch_locate(wmbus_batch *b, unsigned stream, unsigned *i)
{
    if (!b || !b->opened || stream >= b->cfg.n_streams) return WMBUS_EINVAL;      /* a handle whose open failed has no contexts */
    unsigned k = 0;
    while (k + 1 < b->ctx.size() && stream >= b->first[k + 1]) k++;
    *i = k;
    return WMBUS_OK;
}

int wmbus_batch_stage(wmbus_batch *b, unsigned stream, const uint8_t *cu8, size_t nbytes)
{
    unsigned i;
    if (batch_locate(b, stream, &i)) return WMBUS_EINVAL;
    const int rc = wmbus_stage(b->ctx[i], stream - b->first[i], cu8, nbytes);
    if (rc) { std::lock_guard<std::mutex> lk(b->err_lock); snprintf(b->err, sizeof b->err, "batch: context %u: %s", i, b->ctx[i]->err); }
    return rc;
}

void *wmbus_batch_device_input(wmbus_batch *b, unsigned stream)
{
    unsigned i;
    if (batch_locate(b, stream, &i)) return nullptr;
    return wmbus_device_input(b->ctx[i], stream - b->first[i]);
}

int wmbus_batch_run(wmbus_batch *b, const wmbus_batch_io *io, wmbus_batch_stats *stats)
{
    if (!b || !io) return WMBUS_EINVAL;
    if (stats) memset(stats, 0, sizeof *stats);
    if (!b->opened) return WMBUS_EINVAL;                    /* wmbus_batch_open failed: the handle only carries its message */
    { std::lock_guard<std::mutex> lk(b->err_lock); b->rc = 0; b->err[0] = 0; }
    b->stop.store(false);
    if (!io->fill && (io->resident_bytes == 0 || io->resident_bytes > b->cfg.max_push_bytes || io->resident_bytes % WMBUS_BLOCK_BYTES))
        return batch_fail(b, WMBUS_EINVAL, "batch: resident_bytes must be a positive multiple of 4096 and <= max_push_bytes");
    if (io->fill && b->cfg.input_windows != 2) return batch_fail(b, WMBUS_EINVAL, "batch: a host-sourced run needs cfg.input_windows = 2");
    BatchTotals tot;
    const double t0 = now_ms();
    std::vector<std::thread> th;
    for (unsigned i = 1; i < b->ctx.size(); i++) th.emplace_back(batch_worker, b, i, io, &tot);
    batch_worker(b, 0, io, &tot);                           /* the caller's thread drives context 0 */
    for (auto &t : th) t.join();
    if (stats) {
        stats->seconds = (now_ms() - t0) * 1e-3;
        stats->samples = tot.samples.load(); stats->lines = tot.lines.load(); stats->pushes = tot.pushes.load(); stats->warnings = tot.warnings.load();
    }
    return b->rc;
}

}  // extern "C"

#endif /* WM_BATCH_H */
