/*
 * wm_main.c -- `rtl_wmbus_hip`: drop-in for the reference's command line.
 *
 *   rtl_sdr -f 868.95M -s 1600000 - 2>/dev/null | rtl_wmbus_hip [-o] [-a] [-d N] [-p T|S] [-r 0] [-t 0] [-v] [-s] [-f] [-V]
 *
 * Same switches, usage text layout, exit codes and stdout line format as
 * /root/reference/rtl_wmbus.c:869-967,1217-1372; the per-sample loop (:1298-1357) is replaced by
 * wmbus_stage()/wmbus_process()/wmbus_collect() from libwmbus_hip.so.  Plain C; the GPU is only
 * reached through the C ABI in include/wmbus_hip.h.
 * Extensions (letters the reference does not use): -B bytes per GPU push, -L ms (a live stream's bytes never wait longer
 * than this for their push to fill; wm_reader.c), -G HIP device(s), -P polyphase pre-filter, -A 1|2 fast arctangents,
 * -U / -W de-duplication, -T host:port (cu8 over TCP; the role of the reference's unused net_support.h:15-44 /
 * rtl_wmbus.c:1281), -S statistics, and
 *   rtl_wmbus_hip [switches] a.cu8 b.cu8 ...      batch mode: one capture per file, lines prefixed "a.cu8: ".
 * A batch is a wmbus_batch of the library per device: the files are split over several receiver contexts (whole groups
 * of 64, eight contexts at most) that run on their own threads and overlap each other, 2 MiB per file and push, the next
 * push read into page-locked memory and staged into a context's second input window while the previous one is in
 * flight.  With -G all (or -G 0,2,5) the files are sharded over the GPUs of the node, file i on device list[i mod n]
 * (file-per-GPU, no collective), each file's lines in its own order.  -M prints that map and exits.
 */
#include <arpa/inet.h>
#include <netdb.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <time.h>

#include "wm_reader.h"
#include "wmbus_hip.h"

#define VERSION "rtl_wmbus_hip 0.4 (MI355X/gfx950 back end)"
#ifndef COMMIT
#define COMMIT "unknown"          /* the Makefile passes the repository's HEAD, like the reference's does (Makefile:16,38) */
#endif

static void print_usage(const char *prog)
{
    fprintf(stdout, "rtl_wmbus: " VERSION "\n\n");
    fprintf(stdout, "Usage %s:\n", prog);
    fprintf(stdout, "\t-o remove DC offset\n");
    fprintf(stdout, "\t-a accelerate (use an inaccurate atan version)\n");
    fprintf(stdout, "\t-r 0 to disable run length algorithm\n");
    fprintf(stdout, "\t-t 0 to disable time2 algorithm\n");
    fprintf(stdout, "\t-d 2 set decimation rate to 2 (defaults to 2 if omitted)\n");
    fprintf(stdout, "\t-v show used algorithm in the output\n");
    fprintf(stdout, "\t-V show version\n");
    fprintf(stdout, "\t-s receive S1 and T1/C1 datagrams simultaneously. rtl_sdr _MUST_ be set to 868.625MHz (-f 868.625M)\n");
    fprintf(stdout, "\t-p [T,S] to disable processing T1/C1 or S1 mode\n");
    fprintf(stdout, "\t-f exit if flow of incoming data stops\n");
    fprintf(stdout, "\t   (this back end decimates by 1 ... 16: -d 17 and beyond, which the reference accepts, are refused)\n");
    fprintf(stdout, "\t-B bytes per GPU push (multiple of 4096; default 1048576 for a live stream; per file in batch mode 2097152, 1048576 from 384 files per GPU on)\n");
    fprintf(stdout, "\t-L ms a live stream's bytes wait at most this long for their push to fill (default 50; 0: only full pushes)\n");
    fprintf(stdout, "\t-S batch mode: print samples, seconds and Msamples/s to stderr\n");
    fprintf(stdout, "\t-G HIP device ordinal (default 0); batch mode: 'all' or a list '0,2,5' shards the files, file i on device list[i mod n]\n");
    fprintf(stdout, "\t-M batch mode: print the file -> device map and exit\n");
    fprintf(stdout, "\t-U unique: print a datagram that both the time2 and the run length method decoded only once\n");
    fprintf(stdout, "\t-W like -U, and only datagrams with a correct CRC (what wmbusmeters keeps)\n");
    fprintf(stdout, "\t-A 1|2 atan2_approximation / atan2_approximation2 (atan2.h) instead of cargf in the discriminator\n");
    fprintf(stdout, "\t-F tolerance mode: soft symbols within 2e-6 of the reference's instead of bit-identical (faster; default switches only)\n");
    fprintf(stdout, "\t-P polyphase low-pass (ppf.h) instead of the moving average before decimation (1.6 MS/s, -d 2, no -s)\n");
    fprintf(stdout, "\t-T host:port read the cu8 stream from a TCP server instead of stdin\n");
    fprintf(stdout, "\tFILE... batch mode: decode several cu8 files at once (lines prefixed with the file name)\n");
    fprintf(stdout, "\t-h print this help\n");
}

/* cu8 over TCP: connect and hand back the descriptor (what net_support.h:15-44 offers the reference). */
static int open_tcp(const char *hostport)
{
    char host[256];
    const char *colon = strrchr(hostport, ':');
    if (!colon || colon == hostport || (size_t)(colon - hostport) >= sizeof host) { fprintf(stderr, "rtl_wmbus_hip: -T needs host:port\n"); return -1; }
    memcpy(host, hostport, (size_t)(colon - hostport)); host[colon - hostport] = 0;
    struct addrinfo hints, *res = NULL;
    memset(&hints, 0, sizeof hints);
    hints.ai_family = AF_UNSPEC; hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(host, colon + 1, &hints, &res) || !res) { fprintf(stderr, "rtl_wmbus_hip: cannot resolve %s\n", hostport); return -1; }
    int fd = -1;
    for (struct addrinfo *a = res; a; a = a->ai_next) {
        fd = socket(a->ai_family, a->ai_socktype, a->ai_protocol);
        if (fd < 0) continue;
        if (!connect(fd, a->ai_addr, a->ai_addrlen)) break;
        close(fd); fd = -1;
    }
    freeaddrinfo(res);
    if (fd < 0) fprintf(stderr, "rtl_wmbus_hip: cannot connect to %s\n", hostport);
    return fd;
}

/* ---- batch mode ------------------------------------------------------------------------------
 * One wmbus_batch per device (include/wmbus_hip.h): the library splits the device's files over several receiver
 * contexts, drives each on its own thread and overlaps them; this file only supplies the bytes (fread into the page-locked
 * slab the library hands out) and prints the lines. */
struct batch_job {
    wmbus_cfg cfg; int n; char **names; FILE **f; int *live; int rc;
    unsigned fill_threads;                               /* readers per fill call (a context's files are read side by side) */
    unsigned told;                                       /* warnings already reported for this device */
    int stats, fast_exit;
    wmbus_batch_stats st;
};

static pthread_mutex_t out_lock = PTHREAD_MUTEX_INITIALIZER;      /* one push's lines leave as a unit, whichever device they come from */

/* The reference never stops on an input; neither do we: exhausted chip / burst storage is reported once per device and
 * the stream goes on (wmbus_hip.h, WMBUS_WARN_*).  Callers hold out_lock or are single-threaded. */
static void report_warnings(unsigned warnings, unsigned *told)
{
    if (!(warnings & ~*told)) return;
    if (warnings & ~*told & WMBUS_WARN_CHIPS_DROPPED) fprintf(stderr, "rtl_wmbus_hip: warning: run-length chip storage exhausted (interferer?), some chips dropped; continuing\n");
    if (warnings & ~*told & WMBUS_WARN_BURSTS_DROPPED) fprintf(stderr, "rtl_wmbus_hip: warning: burst storage exhausted, some candidate telegrams dropped; continuing\n");
    *told |= warnings;
}

/* wmbus_batch_io.fill: up to `cap` bytes (whole 4096-byte blocks) of every file of the group; a file that has ended
 * contributes mid-scale bytes (no signal).  Returns the longest contribution; 0 when all have ended.
 * The files of a group are read by several threads side by side: one thread copies page-cache bytes into the page-locked slab
 * at 5-6 GB/s, a context's push of 128 x 2 MiB then takes 43 ms to read against 12 ms on the GPU, and the whole program ran
 * at the readers' 4-8 x 6 GB/s instead of the link's 50 (round 4: 12.9 Gsamples/s over 256 files, 16-17 over 1024). */
struct fill_part { struct batch_job *j; unsigned first, k0, k1; uint8_t *slab; size_t pitch, cap; size_t *got; struct fill_part *next; unsigned *left; };

static void *fill_part_run(void *p)
{
    struct fill_part *q = p;
    for (unsigned k = q->k0; k < q->k1; k++) {
        const unsigned s = q->first + k;
        if (q->j->live[s]) {
            q->got[k] = fread(q->slab + (size_t)k * q->pitch, WMBUS_BLOCK_BYTES, q->cap / WMBUS_BLOCK_BYTES, q->j->f[s]) * WMBUS_BLOCK_BYTES;
            if (q->got[k] < q->cap) q->j->live[s] = 0;
        }
    }
    return NULL;
}

/* The readers are a pool that lives as long as the process (up to 15 threads were created and joined in EVERY fill
 * call, thousands of times per batch, inside the out-of-GPU critical path).  A fill call -- several may run at once, one per context --
 * queues its parts, works on the queue itself and waits for its own parts' count to reach zero. */
static struct { pthread_mutex_t m; pthread_cond_t work, done; struct fill_part *head; unsigned threads; } pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0};

static struct fill_part *pool_take(void) { struct fill_part *q = pool.head; if (q) pool.head = q->next; return q; }      /* pool.m held */
static void pool_finish(struct fill_part *q)                                                                               /* pool.m held */
{
    if (--*q->left == 0) pthread_cond_broadcast(&pool.done);
}
static void *pool_worker(void *unused)
{
    (void)unused;
    pthread_mutex_lock(&pool.m);
    for (;;) {
        struct fill_part *q = pool_take();
        if (!q) { pthread_cond_wait(&pool.work, &pool.m); continue; }
        pthread_mutex_unlock(&pool.m);
        fill_part_run(q);
        pthread_mutex_lock(&pool.m);
        pool_finish(q);
    }
    return NULL;
}
static void pool_grow(unsigned want)       /* detached workers, started on first use; fewer than asked for is fine (the caller works too) */
{
    pthread_mutex_lock(&pool.m);
    while (pool.threads < want && pool.threads < 64u) {
        pthread_t th;
        if (pthread_create(&th, NULL, pool_worker, NULL)) break;
        pthread_detach(th);
        pool.threads++;
    }
    pthread_mutex_unlock(&pool.m);
}

static size_t batch_fill_padded(void *user, unsigned first, unsigned n, uint8_t *slab, size_t pitch, size_t cap)
{
    struct batch_job *j = user;
    size_t most = 0;
    size_t *got = calloc(n, sizeof *got);
    if (!got) { fprintf(stderr, "rtl_wmbus_hip: out of memory\n"); return 0; }
    unsigned T = j->fill_threads ? j->fill_threads : 1u;
    if (T > 16u) T = 16u;
    if (n < 2u * T) T = 1u;
    struct fill_part part[16];
    unsigned left = T;
    for (unsigned t = 0; t < T; t++) part[t] = (struct fill_part){j, first, (unsigned)((uint64_t)n * t / T), (unsigned)((uint64_t)n * (t + 1) / T), slab, pitch, cap, got, NULL, &left};
    if (T > 1u) {
        pool_grow(8u * (T - 1u));                            /* readers for the contexts that fill at the same time */
        pthread_mutex_lock(&pool.m);
        for (unsigned t = T - 1u; t >= 1u; t--) { part[t].next = pool.head; pool.head = &part[t]; }
        pthread_cond_broadcast(&pool.work);
        pthread_mutex_unlock(&pool.m);
    }
    fill_part_run(&part[0]);
    pthread_mutex_lock(&pool.m);
    pool_finish(&part[0]);
    for (;;) {                                               /* my own parts nobody has taken yet: here; those in other hands: wait */
        struct fill_part *q = NULL, **pp = &pool.head;
        for (; *pp; pp = &(*pp)->next) if ((*pp)->left == &left) { q = *pp; *pp = q->next; break; }
        if (q) { pthread_mutex_unlock(&pool.m); fill_part_run(q); pthread_mutex_lock(&pool.m); pool_finish(q); continue; }
        if (left == 0) break;
        pthread_cond_wait(&pool.done, &pool.m);
    }
    pthread_mutex_unlock(&pool.m);
    for (unsigned k = 0; k < n; k++) if (got[k] > most) most = got[k];
    for (unsigned k = 0; k < n; k++) if (got[k] < most) memset(slab + (size_t)k * pitch + got[k], 128, most - got[k]);
    free(got);
    return most;
}

static void batch_lines(void *user, unsigned first, unsigned n, const wmbus_line *ln, size_t nl, const char *text, const wmbus_timing *t)
{
    struct batch_job *j = user;
    (void)first; (void)n;
    pthread_mutex_lock(&out_lock);
    report_warnings(t->warnings, &j->told);
    for (size_t i = 0; i < nl; i++) {
        fputs(j->names[ln[i].stream], stdout); fputs(": ", stdout);
        fwrite(text + ln[i].text_off, 1, ln[i].text_len, stdout);
    }
    if (nl) fflush(stdout);
    pthread_mutex_unlock(&out_lock);
}

static int run_batch(struct batch_job *j)
{
    int rc = EXIT_FAILURE;
    wmbus_batch *b = NULL;
    j->f = calloc((size_t)j->n, sizeof *j->f); j->live = calloc((size_t)j->n, sizeof *j->live);
    if (!j->f || !j->live) goto out;
    for (int s = 0; s < j->n; s++) {
        j->f[s] = fopen(j->names[s], "rb");
        if (!j->f[s]) { fprintf(stderr, "rtl_wmbus_hip: cannot open %s\n", j->names[s]); goto out; }
        j->live[s] = 1;
    }
    j->cfg.n_streams = (unsigned)j->n;
    j->cfg.input_windows = 2;                                /* the next push crosses PCIe while the previous one is in flight */
    struct timespec t_open0, t_open1;
    clock_gettime(CLOCK_MONOTONIC, &t_open0);
    if (wmbus_batch_open(&j->cfg, 0, &b)) {
        fprintf(stderr, "rtl_wmbus_hip: cannot open GPU back end: %s\n", b ? wmbus_batch_last_error(b) : "out of memory");
        goto out;
    }
    clock_gettime(CLOCK_MONOTONIC, &t_open1);
    if (j->stats) fprintf(stderr, "rtl_wmbus_hip: device %d: contexts open after %.3f s\n", j->cfg.device, (double)(t_open1.tv_sec - t_open0.tv_sec) + (t_open1.tv_nsec - t_open0.tv_nsec) * 1e-9);
    wmbus_batch_io io;
    memset(&io, 0, sizeof io);
    io.fill = batch_fill_padded; io.lines = batch_lines; io.user = j;
    if (wmbus_batch_run(b, &io, &j->st)) { fprintf(stderr, "rtl_wmbus_hip: %s\n", wmbus_batch_last_error(b)); goto out; }
    if (j->stats)
        fprintf(stderr, "rtl_wmbus_hip: device %d: %d files, %u contexts, %llu samples, %llu lines, %u pushes in %.3f s = %.1f Msamples/s\n", j->cfg.device, j->n,
                wmbus_batch_contexts(b), (unsigned long long)j->st.samples, (unsigned long long)j->st.lines, j->st.pushes, j->st.seconds,
                j->st.seconds > 0 ? (double)j->st.samples / j->st.seconds / 1e6 : 0.0);
    rc = EXIT_SUCCESS;
out:
    if (!j->fast_exit || rc != EXIT_SUCCESS) wmbus_batch_close(b);      /* every exit path closes what it opened, unless the process is about to end anyway */
    for (int s = 0; j->f && s < j->n; s++) if (j->f[s]) fclose(j->f[s]);
    free(j->f); free(j->live);
    return rc;
}

/* File i of a batch -> position in the device list (stream s -> GPU s mod n).  The only place
 * the map is defined; -M prints it, tests/test_multi_gloo.py holds it against rtl-wmbus_amd/shard.py. */
static int shard_slot(int file_index, int n_devices) { return file_index % n_devices; }

/* ---- batch mode over several GPUs: one batch per device, each on its own thread ---------- */
static void *shard_thread(void *p) { struct batch_job *j = p; j->rc = j->n ? run_batch(j) : EXIT_SUCCESS; return NULL; }

static int run_sharded(wmbus_cfg cfg, int n, char **names, const int *devs, int n_devs, int map_only, int stats)
{
    struct batch_job *jobs = calloc((size_t)n_devs, sizeof *jobs);
    pthread_t *th = calloc((size_t)n_devs, sizeof *th);
    /* host decoder threads per context: devices x contexts (8 per batch) x threads within the host's hardware threads */
    long cpus = sysconf(_SC_NPROCESSORS_ONLN);
    if (cpus < 1) cpus = 16;
    unsigned per_ctx = (unsigned)(cpus / (8L * n_devs));
    if (per_ctx < 1) per_ctx = 1;
    if (per_ctx > 16) per_ctx = 16;
    if (cfg.host_threads == 0 && n_devs > 1) cfg.host_threads = per_ctx;
    /* file readers per context: the host's threads over the devices' contexts, half of them (the decoder threads want the rest) */
    unsigned readers = (unsigned)(cpus / (16L * n_devs));
    if (readers < 1) readers = 1;
    if (readers > 8) readers = 8;
    for (int k = 0; k < n_devs; k++) jobs[k].fill_threads = readers;
    for (int k = 0; k < n_devs; k++) { jobs[k].cfg = cfg; jobs[k].cfg.device = devs[k]; jobs[k].names = calloc((size_t)n, sizeof(char *)); jobs[k].stats = stats; jobs[k].fast_exit = !getenv("WMBUS_SLOW_EXIT"); }
    for (int i = 0; i < n; i++) {
        const int k = shard_slot(i, n_devs);
        jobs[k].names[jobs[k].n++] = names[i];
        if (map_only) fprintf(stdout, "%s -> device %d\n", names[i], devs[k]);
    }
    int rc = EXIT_SUCCESS;
    if (!map_only) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int k = 1; k < n_devs; k++) pthread_create(&th[k], NULL, shard_thread, &jobs[k]);
        shard_thread(&jobs[0]);
        if (jobs[0].rc) rc = jobs[0].rc;
        for (int k = 1; k < n_devs; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = jobs[k].rc; }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (stats) {
            unsigned long long samples = 0; double run_s = 0;
            for (int k = 0; k < n_devs; k++) { samples += jobs[k].st.samples; if (jobs[k].st.seconds > run_s) run_s = jobs[k].st.seconds; }
            const double wall = (double)(t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
            fprintf(stderr, "rtl_wmbus_hip: total: %d files on %d device(s), %llu samples; decode %.3f s = %.1f Msamples/s; with set-up (contexts, page-locked staging) %.3f s = %.1f Msamples/s\n",
                    n, n_devs, samples, run_s, run_s > 0 ? samples / run_s / 1e6 : 0.0, wall, wall > 0 ? samples / wall / 1e6 : 0.0);
        }
    }
    for (int k = 0; k < n_devs; k++) free(jobs[k].names);
    free(jobs); free(th);
    return rc;
}

/* The last thing a batch run does: every line is out, and the process ends HERE.  An orderly teardown (unpinning 16 GB of
 * staging, freeing 44 GB of HBM, the HIP runtime's own exit handlers) took 1.5 s of a 1024-file job's 5 s; the driver
 * reclaims all of it when the process is gone.  Library callers that live on use wmbus_batch_close. */
static void finish(int rc)
{
    fflush(stdout); fflush(stderr);
    if (getenv("WMBUS_SLOW_EXIT")) exit(rc);             /* tests / leak checkers: every batch has been closed (run_batch), the runtime's exit handlers run */
    _exit(rc);
}

/* -G: "3" | "all" | "0,2,5" -> device list; returns the count or -1 */
static int parse_devices(const char *arg, int *devs, int cap)
{
    if (!strcmp(arg, "all")) {
        int n = wmbus_device_count();
        if (n > cap) n = cap;
        for (int k = 0; k < n; k++) devs[k] = k;
        return n;                                            /* 0: no HIP device */
    }
    int n = 0;
    const char *p = arg;
    while (*p) {
        char *end;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || n == cap) return -1;
        devs[n++] = (int)v;
        if (*end == ',') end++; else if (*end) return -1;
        p = end;
    }
    return n ? n : -1;
}

/* ---- live stream (stdin / TCP): wm_reader.c stages the bytes, this is where a push leaves -------- */
struct live { wmbus_ctx *ctx; unsigned told; };

static int live_push(void *user, const unsigned char *buf, size_t n)
{
    struct live *lv = user;
    int rc = wmbus_stage(lv->ctx, 0, buf, n);
    if (!rc) rc = wmbus_process(lv->ctx, n);
    if (!rc) rc = wmbus_collect(lv->ctx);
    if (rc) { fprintf(stderr, "rtl_wmbus_hip: %s\n", wmbus_last_error(lv->ctx)); return rc; }
    wmbus_timing t;
    if (!wmbus_get_timing(lv->ctx, &t)) report_warnings(t.warnings, &lv->told);
    size_t len = 0;
    const char *text = wmbus_lines_text(lv->ctx, &len);
    if (len) { fwrite(text, 1, len, stdout); fflush(stdout); }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 1 && isatty(0)) { print_usage(argv[0]); return 0; }
    wmbus_runtime_init();                                    /* before the first HIP call, before any thread exists */

    wmbus_cfg cfg;
    wmbus_default_cfg(&cfg);
    cfg.max_push_bytes = 0;                                  /* 0: the mode's default, see below */
    int check_flow = 0, opt, map_only = 0, devs[64], n_devs = 0, stats = 0;
    unsigned max_latency_ms = 50;
    const char *tcp = NULL;
    while ((opt = getopt(argc, argv, "ofad:p:r:vVst:B:G:PT:A:MUWL:SF")) != -1) {
        switch (opt) {
        case 'o': cfg.remove_dc = 1; break;
        case 'f': check_flow = 1; break;
        case 'a': cfg.accurate_atan = 0; break;
        case 'p':
            if (!strcmp(optarg, "T") || !strcmp(optarg, "t")) cfg.t1c1_enabled = 0;
            else if (!strcmp(optarg, "S") || !strcmp(optarg, "s")) cfg.s1_enabled = 0;
            else { print_usage(argv[0]); return EXIT_FAILURE; }
            break;
        case 'r': if (strcmp(optarg, "0")) { print_usage(argv[0]); return EXIT_FAILURE; } cfg.rla_enabled = 0; break;
        case 't': if (strcmp(optarg, "0")) { print_usage(argv[0]); return EXIT_FAILURE; } cfg.time2_enabled = 0; break;
        case 'd': cfg.decimation = (unsigned)strtoul(optarg, NULL, 10); break;
        case 's': cfg.simultaneous = 1; break;
        case 'v': cfg.show_algorithm = 1; break;
        case 'V': fprintf(stdout, "rtl_wmbus: " VERSION "\n"); fprintf(stdout, COMMIT "\n"); return EXIT_SUCCESS;    /* two lines: version, commit (rtl_wmbus.c:886-890) */
        case 'B': cfg.max_push_bytes = (size_t)strtoull(optarg, NULL, 10) / WMBUS_BLOCK_BYTES * WMBUS_BLOCK_BYTES; break;
        case 'G':
            n_devs = parse_devices(optarg, devs, 64);
            if (n_devs < 0) { print_usage(argv[0]); return EXIT_FAILURE; }
            if (n_devs == 0) { fprintf(stderr, "rtl_wmbus_hip: -G all: no HIP device (this program has no CPU fallback)\n"); return EXIT_FAILURE; }
            cfg.device = devs[0];
            break;
        case 'M': map_only = 1; break;
        case 'U': cfg.dedup_twins = 1; break;
        case 'W': cfg.dedup_twins = 1; cfg.only_crc_ok = 1; break;
        case 'P': cfg.prefilter = WMBUS_PREFILTER_POLYPHASE; break;
        case 'A': cfg.atan_mode = atoi(optarg); break;
        case 'T': tcp = optarg; break;
        case 'L': max_latency_ms = (unsigned)strtoul(optarg, NULL, 10); break;
        case 'S': stats = 1; break;
        case 'F': cfg.tolerance_mode = 1; break;
        default: print_usage(argv[0]); return EXIT_FAILURE;
        }
    }
    if (getenv("WMBUS_FIXED_TS")) cfg.fixed_timestamp = 1;
    /* -d is taken unchecked like the reference takes it (rtl_wmbus.c:950).  There `-d 0` keeps every sample, exactly like
     * `-d 1` (the counter test at :1350-1352 is `++index < rate`), so that is what it means here; with -s it indexes a
     * zero-length table in the reference (undefined behaviour) and is refused.  Rates above 16 (12.8 MS/s) do not fit the
     * demodulation kernel's staging and are refused with a message instead of the usage text. */
    if (cfg.decimation == 0 && !cfg.simultaneous) cfg.decimation = 1;
    if (cfg.decimation == 0 || cfg.decimation > 16) {
        fprintf(stderr, "rtl_wmbus_hip: -d %u: this back end decimates by 1..16 (-d 0 without -s is -d 1)\n", cfg.decimation);
        return EXIT_FAILURE;
    }

    if (optind < argc) {
        /* batch mode: 2 MiB per file and push (-B overrides).  The staging is page-locked -- files x push bytes (twice that until round 5), pinned by
         * the driver one allocation at a time at about 6 GB/s -- and a job whose files are a few pushes long spends longer
         * pinning than decoding: 1024 files of 64 MiB took 3.9-4.3 s end to end with 8 MiB pushes (16 GB of staging),
         * 2.5-2.7 s with 2 MiB (4 GB), 2.9-3.1 s with 4 MiB; r03, which pinned everything before its first push: 6.6-7.4 s.
         * Round 5 (parallel readers; ONE slab per context, wm_batch.h: the staging is files x push bytes), Gsamples/s decode:
         * 1024 files 20.5-21.3 with 1 MiB, 17-18 with 512-768 KiB, 14.2-15.2 with 2 MiB, 11.5-13.9 with 4 MiB; 512 files 17.5-18.6
         * with 1 MiB, 15.4-15.9 with 2 MiB; 256 files 15.0-15.5 with 1.5-2 MiB, 11.9 with 1 MiB, 14.4-15.7 with 4 MiB
         * (profiles/r05_cli_push_bytes.txt): a context wants 64-128 MiB per push -- less starves the GPU, more makes the
         * pipeline's first and last push, and the pinning, long.  So: 1 MiB per file and push from 384 files per device on, else 2 MiB */
        if (n_devs == 0) { devs[0] = cfg.device; n_devs = 1; }
        if (cfg.max_push_bytes == 0) cfg.max_push_bytes = (argc - optind + n_devs - 1) / n_devs >= 384 ? 1u << 20 : 2u << 20;
        finish(run_sharded(cfg, argc - optind, argv + optind, devs, n_devs, map_only, stats));
    }
    if (cfg.max_push_bytes == 0) cfg.max_push_bytes = 1u << 20;

    if (check_flow) fprintf(stderr, "rtl_wmbus: monitoring flow\n");
    int fd = 0;
    if (tcp && (fd = open_tcp(tcp)) < 0) return EXIT_FAILURE;

    struct live lv = {NULL, 0};
    int rc = wmbus_open(&cfg, &lv.ctx);
    if (rc) {
        fprintf(stderr, "rtl_wmbus_hip: cannot open GPU back end: %s\n", lv.ctx ? wmbus_last_error(lv.ctx) : "out of memory");
        wmbus_close(lv.ctx);
        return EXIT_FAILURE;
    }
    /* -f: the reference arms alarm(2) around every fread (rtl_wmbus.c:1300-1302) */
    const wm_reader_cfg rc_cfg = {fd, cfg.max_push_bytes, max_latency_ms, check_flow ? 2000u : 0u};
    const int how = wm_reader_run(&rc_cfg, live_push, &lv);
    wmbus_close(lv.ctx);
    if (how == WM_READER_FLOW_STOPPED) {
        fprintf(stderr, "rtl_wmbus: exiting since incoming data stopped flowing!\n");      /* rtl_wmbus.c:76 */
        return EXIT_FAILURE;
    }
    return how == WM_READER_EOF ? EXIT_SUCCESS : EXIT_FAILURE;
}
