/* wm_reader.c -- see wm_reader.h. */
#include "wm_reader.h"

#include <errno.h>
#include <poll.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define WM_BLOCK 4096u

static long long now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}

int wm_reader_run(const wm_reader_cfg *cfg, wm_reader_push_fn push, void *user)
{
    if (!cfg || !push || cfg->max_push < WM_BLOCK || cfg->max_push % WM_BLOCK) return WM_READER_ERROR;
    unsigned char *buf = malloc(cfg->max_push);
    if (!buf) return WM_READER_ERROR;
    size_t have = 0;
    unsigned long long total = 0;                  /* bytes read so far */
    /* t_first: arrival of the oldest staged byte.  t_data: completion of the last whole 4096-byte block -- the reference
     * arms alarm(2) around the fread of ONE block (rtl_wmbus.c:1300-1302), so a source that dribbles single bytes trips
     * -f as soon as a block takes longer than the time-out, however regularly the bytes come */
    long long t_first = 0, t_data = now_ms();
    int rc = WM_READER_EOF;

#define PUSH_WHOLE_BLOCKS() do { \
        const size_t k_ = have / WM_BLOCK * WM_BLOCK; \
        if (k_) { \
            const long long t0_ = now_ms(); \
            if (push(user, buf, k_)) { rc = WM_READER_PUSH_FAILED; goto out; } \
            memmove(buf, buf + k_, have - k_); have -= k_; t_first = now_ms(); \
            t_data += t_first - t0_;          /* -f measures how long the INPUT has been quiet: time inside push() (GPU push, line printing) does not count */ \
        } } while (0)

    for (;;) {
        long long wait = -1;
        const long long now = now_ms();
        if (cfg->max_latency_ms && have >= WM_BLOCK) { wait = t_first + cfg->max_latency_ms - now; if (wait < 0) wait = 0; }
        if (cfg->flow_timeout_ms) {
            long long w = t_data + cfg->flow_timeout_ms - now;
            if (w < 0) w = 0;
            if (wait < 0 || w < wait) wait = w;
        }
        struct pollfd pf = {cfg->fd, POLLIN, 0};
        const int pr = poll(&pf, 1, (int)wait);
        if (pr < 0) { if (errno == EINTR) continue; rc = WM_READER_ERROR; break; }
        if (pr == 0) {
            const long long t = now_ms();
            if (cfg->flow_timeout_ms && t - t_data >= (long long)cfg->flow_timeout_ms) {
                PUSH_WHOLE_BLOCKS();               /* what has been read is decoded before giving up (rtl_wmbus.c:1300-1308) */
                rc = WM_READER_FLOW_STOPPED;
                break;
            }
            if (cfg->max_latency_ms && have >= WM_BLOCK && t - t_first >= (long long)cfg->max_latency_ms) PUSH_WHOLE_BLOCKS();
            continue;
        }
        const ssize_t n = read(cfg->fd, buf + have, cfg->max_push - have);
        if (n < 0) { if (errno == EINTR || errno == EAGAIN) continue; rc = WM_READER_ERROR; break; }
        if (n == 0) { PUSH_WHOLE_BLOCKS(); break; }            /* end of input: the partial tail is dropped */
        const long long t_now = now_ms();
        const int block_done = (total + (unsigned long long)n) / WM_BLOCK != total / WM_BLOCK;
        if (block_done) t_data = t_now;                        /* a block has been completed */
        total += (unsigned long long)n;
        if (have == 0) t_first = t_now;
        have += (size_t)n;
        /* a source that always has a byte ready never lets poll() time out: the -f test has to be made here too (the reference's
         * alarm(2) fires around ONE fread however steadily single bytes arrive) */
        if (!block_done && cfg->flow_timeout_ms && t_now - t_data >= (long long)cfg->flow_timeout_ms) {
            PUSH_WHOLE_BLOCKS();
            rc = WM_READER_FLOW_STOPPED;
            break;
        }
        if (have == cfg->max_push) PUSH_WHOLE_BLOCKS();
        else if (cfg->max_latency_ms && have >= WM_BLOCK && t_now - t_first >= (long long)cfg->max_latency_ms) PUSH_WHOLE_BLOCKS();
    }
out:
    free(buf);
    return rc;
}
