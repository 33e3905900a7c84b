/*
 * wm_reader.h -- the drop-in CLI's input side for a LIVE stream (stdin / TCP): plain C, no GPU in sight, so that
 * tests/test_reader.py can drive it through a paced pipe on any box.
 *
 * The reference consumes 4096-byte blocks as they arrive and prints a telegram the moment its last chip is processed
 * (/root/reference/rtl_wmbus.c:1298-1308, t1_c1_packet_decoder.h:671-699); with -f it gives up when no block arrives
 * for two seconds, AFTER having processed every block it has read (:1300-1302, :71-78).  A GPU push wants more than one
 * block, so bytes are staged -- but never for longer than `max_latency_ms`: a push leaves when it is full, when its
 * oldest byte has waited that long, at end of input, and before a flow time-out is reported.  Only whole blocks are
 * pushed; the partial tail at end of input is dropped like the reference drops it (:1304-1308).
 */
#ifndef WM_READER_H
#define WM_READER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int (*wm_reader_push_fn)(void *user, const unsigned char *buf, size_t nbytes);   /* nbytes: a positive multiple of 4096; non-zero return stops the reader */

typedef struct wm_reader_cfg {
    int fd;                      /* input descriptor (blocking or not) */
    size_t max_push;             /* multiple of 4096 */
    unsigned max_latency_ms;     /* 0: pushes leave only when full (and at the end) */
    unsigned flow_timeout_ms;    /* 0: wait for ever; the reference's -f uses 2000 */
} wm_reader_cfg;

enum { WM_READER_EOF = 0, WM_READER_FLOW_STOPPED = 1, WM_READER_ERROR = -1, WM_READER_PUSH_FAILED = -2 };

/* Runs until end of input (WM_READER_EOF), until no whole 4096-byte block has been completed for flow_timeout_ms
 * (WM_READER_FLOW_STOPPED: the staged blocks have been pushed first), or until an error. */
int wm_reader_run(const wm_reader_cfg *cfg, wm_reader_push_fn push, void *user);

#ifdef __cplusplus
}
#endif
#endif
