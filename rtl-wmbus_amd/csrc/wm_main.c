/*
 * wm_main.c -- `rtl_wmbus_hip`: drop-in for the reference's command line.
 *
 *   rtl_sdr -f 868.95M -s 1600000 - 2>/dev/null | rtl_wmbus_hip [-o] [-a] [-d N] [-p T|S] [-r 0] [-t 0] [-v] [-s] [-f] [-V]
 *
 * Same switches, usage text layout, exit codes and stdout line format as
 * /root/reference/rtl_wmbus.c:869-967,1217-1372; the per-sample loop (:1298-1357) is replaced by
 * wmbus_stage()/wmbus_process()/wmbus_collect() from libwmbus_hip.so.  Plain C; the GPU is only
 * reached through the C ABI in include/wmbus_hip.h.
 * Extensions (letters the reference does not use): -B bytes per GPU push (default 1 MiB),
 * -G HIP device ordinal, -P polyphase pre-filter, -T host:port (cu8 over TCP, e.g. a raw IQ server;
 * the role of the reference's unused net_support.h:15-44 / rtl_wmbus.c:1281), and
 *   rtl_wmbus_hip [switches] a.cu8 b.cu8 ...      batch mode: one capture per file, all of them in
 * lock step on one GPU, lines prefixed "a.cu8: "; while the GPU works on one pinned slab the next one is read and
 * staged into the context's second input window (double-buffered H2D on the copy stream).  With -G all (or -G 0,2,5) batch mode shards the files over the
 * GPUs of the node, file i on device list[i mod n] (file-per-GPU, no collective): one receiver
 * context and one worker thread per device, each file's lines in its own order.  -M prints that map and exits.
 */
#include <arpa/inet.h>
#include <netdb.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include "wmbus_hip.h"

#define VERSION "rtl_wmbus_hip 0.1 (MI355X/gfx950 back end)"

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
    fprintf(stdout, "\t-B bytes per GPU push (multiple of 4096, default 1048576)\n");
    fprintf(stdout, "\t-G HIP device ordinal (default 0); batch mode: 'all' or a list '0,2,5' shards the files, file i on device list[i mod n]\n");
    fprintf(stdout, "\t-M batch mode: print the file -> device map and exit\n");
    fprintf(stdout, "\t-U unique: print a datagram that both the time2 and the run length method decoded only once\n");
    fprintf(stdout, "\t-W like -U, and only datagrams with a correct CRC (what wmbusmeters keeps)\n");
    fprintf(stdout, "\t-A 1|2 atan2_approximation / atan2_approximation2 (atan2.h) instead of cargf in the discriminator\n");
    fprintf(stdout, "\t-P polyphase low-pass (ppf.h) instead of the moving average before decimation (1.6 MS/s, -d 2, no -s)\n");
    fprintf(stdout, "\t-T host:port read the cu8 stream from a TCP server instead of stdin\n");
    fprintf(stdout, "\tFILE... batch mode: decode several cu8 files at once (lines prefixed with the file name)\n");
    fprintf(stdout, "\t-h print this help\n");
}

/* cu8 over TCP: connect and hand back a stdio stream (what net_support.h:15-44 offers the reference). */
static FILE *open_tcp(const char *hostport)
{
    char host[256];
    const char *colon = strrchr(hostport, ':');
    if (!colon || colon == hostport || (size_t)(colon - hostport) >= sizeof host) { fprintf(stderr, "rtl_wmbus_hip: -T needs host:port\n"); return NULL; }
    memcpy(host, hostport, (size_t)(colon - hostport)); host[colon - hostport] = 0;
    struct addrinfo hints, *res = NULL;
    memset(&hints, 0, sizeof hints);
    hints.ai_family = AF_UNSPEC; hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(host, colon + 1, &hints, &res) || !res) { fprintf(stderr, "rtl_wmbus_hip: cannot resolve %s\n", hostport); return NULL; }
    int fd = -1;
    for (struct addrinfo *a = res; a; a = a->ai_next) {
        fd = socket(a->ai_family, a->ai_socktype, a->ai_protocol);
        if (fd < 0) continue;
        if (!connect(fd, a->ai_addr, a->ai_addrlen)) break;
        close(fd); fd = -1;
    }
    freeaddrinfo(res);
    if (fd < 0) { fprintf(stderr, "rtl_wmbus_hip: cannot connect to %s\n", hostport); return NULL; }
    return fdopen(fd, "rb");
}

/* ---- batch mode ------------------------------------------------------------------------------ */
struct batch {
    int n; FILE **f; int *live; size_t push; unsigned char *slab[2]; size_t filled[2];   /* bytes per stream in slab k */
};

/* Fill slab k: up to `push` bytes (whole 4096-byte blocks) of every file; a file that has ended
 * contributes mid-scale bytes (no signal).  filled = longest contribution. */
static void batch_fill(struct batch *b, int k)
{
    size_t most = 0;
    for (int s = 0; s < b->n; s++) {
        unsigned char *dst = b->slab[k] + (size_t)s * b->push;
        size_t got = 0;
        if (b->live[s]) {
            got = fread(dst, WMBUS_BLOCK_BYTES, b->push / WMBUS_BLOCK_BYTES, b->f[s]) * WMBUS_BLOCK_BYTES;
            if (got < b->push) b->live[s] = 0;
        }
        memset(dst + got, 128, b->push - got);
        if (got > most) most = got;
    }
    b->filled[k] = most;
}

static pthread_mutex_t out_lock = PTHREAD_MUTEX_INITIALIZER;      /* one push's lines leave as a unit */

/* The reference never stops on an input; neither do we: exhausted chip / burst storage is reported once and the
 * stream goes on (wmbus_hip.h, WMBUS_WARN_*). */
static void report_warnings(wmbus_ctx *ctx)
{
    static unsigned told = 0;
    wmbus_timing t;
    if (wmbus_get_timing(ctx, &t) || !(t.warnings & ~told)) return;
    if (t.warnings & ~told & WMBUS_WARN_CHIPS_DROPPED) fprintf(stderr, "rtl_wmbus_hip: warning: run-length chip storage exhausted (interferer?), some chips dropped; continuing\n");
    if (t.warnings & ~told & WMBUS_WARN_BURSTS_DROPPED) fprintf(stderr, "rtl_wmbus_hip: warning: burst storage exhausted, some candidate telegrams dropped; continuing\n");
    told |= t.warnings;
}

/* File i of a batch -> position in the device list (stream s -> GPU s mod n).  The only place
 * the map is defined; -M prints it, tests/test_multi_gloo.py holds it against rtl-wmbus_amd/shard.py. */
static int shard_slot(int file_index, int n_devices) { return file_index % n_devices; }

static int run_batch(wmbus_cfg cfg, int n, char **names)
{
    struct batch b;
    memset(&b, 0, sizeof b);
    b.n = n; b.push = cfg.max_push_bytes;
    b.f = calloc((size_t)n, sizeof *b.f); b.live = calloc((size_t)n, sizeof *b.live);
    for (int s = 0; s < n; s++) {
        b.f[s] = fopen(names[s], "rb");
        if (!b.f[s]) { fprintf(stderr, "rtl_wmbus_hip: cannot open %s\n", names[s]); return EXIT_FAILURE; }
        b.live[s] = 1;
    }
    cfg.n_streams = (unsigned)n;
    cfg.input_windows = 2;
    wmbus_ctx *ctx = NULL;
    if (wmbus_open(&cfg, &ctx)) {
        fprintf(stderr, "rtl_wmbus_hip: cannot open GPU back end: %s\n", ctx ? wmbus_last_error(ctx) : "out of memory");
        wmbus_close(ctx);
        return EXIT_FAILURE;
    }
    for (int k = 0; k < 2; k++) {
        b.slab[k] = wmbus_alloc_pinned((size_t)n * b.push);
        if (!b.slab[k]) { fprintf(stderr, "rtl_wmbus_hip: cannot allocate pinned staging\n"); return EXIT_FAILURE; }
    }
    /* wmbus_process only enqueues: while the GPU works on one slab this thread reads the next one from the files and
     * stages it into the context's other input window (cfg.input_windows = 2: the copies run on their own stream) */
    int rc = 0, cur = 0, staged = 0;
    batch_fill(&b, cur);
    while (b.filled[cur] && !rc) {
        const size_t nbytes = b.filled[cur];
        for (int s = 0; s < n && !rc && !staged; s++) rc = wmbus_stage(ctx, (unsigned)s, b.slab[cur] + (size_t)s * b.push, nbytes);
        if (!rc) rc = wmbus_process(ctx, nbytes);
        staged = 0;
        if (!rc) {
            batch_fill(&b, cur ^ 1);
            for (int s = 0; s < n && !rc && b.filled[cur ^ 1]; s++)
                rc = wmbus_stage(ctx, (unsigned)s, b.slab[cur ^ 1] + (size_t)s * b.push, b.filled[cur ^ 1]);
            staged = b.filled[cur ^ 1] != 0;
        }
        if (!rc) rc = wmbus_collect(ctx);
        if (rc) fprintf(stderr, "rtl_wmbus_hip: %s\n", wmbus_last_error(ctx));
        else {
            const wmbus_line *ln = NULL;
            const size_t nl = wmbus_lines(ctx, &ln);
            size_t len = 0;
            const char *text = wmbus_lines_text(ctx, &len);
            pthread_mutex_lock(&out_lock);
            report_warnings(ctx);
            for (size_t i = 0; i < nl; i++) {
                fputs(names[ln[i].stream], stdout); fputs(": ", stdout);
                fwrite(text + ln[i].text_off, 1, ln[i].text_len, stdout);
            }
            fflush(stdout);
            pthread_mutex_unlock(&out_lock);
        }
        cur ^= 1;
    }
    for (int k = 0; k < 2; k++) wmbus_free_pinned(b.slab[k]);
    for (int s = 0; s < n; s++) fclose(b.f[s]);
    free(b.f); free(b.live);
    wmbus_close(ctx);
    return rc ? EXIT_FAILURE : EXIT_SUCCESS;
}

/* ---- batch mode over several GPUs: one run_batch per device, each on its own thread ---------- */
struct shard_job { wmbus_cfg cfg; int n; char **names; int rc; };
static void *shard_thread(void *p) { struct shard_job *j = p; j->rc = j->n ? run_batch(j->cfg, j->n, j->names) : EXIT_SUCCESS; return NULL; }

static int run_sharded(wmbus_cfg cfg, int n, char **names, const int *devs, int n_devs, int map_only)
{
    struct shard_job *jobs = calloc((size_t)n_devs, sizeof *jobs);
    pthread_t *th = calloc((size_t)n_devs, sizeof *th);
    for (int k = 0; k < n_devs; k++) { jobs[k].cfg = cfg; jobs[k].cfg.device = devs[k]; jobs[k].names = calloc((size_t)n, sizeof(char *)); }
    for (int i = 0; i < n; i++) {
        const int k = shard_slot(i, n_devs);
        jobs[k].names[jobs[k].n++] = names[i];
        if (map_only) fprintf(stdout, "%s -> device %d\n", names[i], devs[k]);
    }
    int rc = EXIT_SUCCESS;
    if (!map_only) {
        for (int k = 0; k < n_devs; k++) pthread_create(&th[k], NULL, shard_thread, &jobs[k]);
        for (int k = 0; k < n_devs; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = jobs[k].rc; }
    }
    for (int k = 0; k < n_devs; k++) free(jobs[k].names);
    free(jobs); free(th);
    return rc;
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

static void on_alarm(int signo)
{
    (void)signo;
    static const char msg[] = "rtl_wmbus: exiting since incoming data stopped flowing!\n";
    if (write(2, msg, sizeof msg - 1) < 0) _exit(EXIT_FAILURE);
    _exit(EXIT_FAILURE);
}

static int flush_push(wmbus_ctx *ctx, const unsigned char *buf, size_t n)
{
    int rc = wmbus_stage(ctx, 0, buf, n);
    if (!rc) rc = wmbus_process(ctx, n);
    if (!rc) rc = wmbus_collect(ctx);
    if (rc) { fprintf(stderr, "rtl_wmbus_hip: %s\n", wmbus_last_error(ctx)); return rc; }
    report_warnings(ctx);
    size_t len = 0;
    const char *text = wmbus_lines_text(ctx, &len);
    if (len) { fwrite(text, 1, len, stdout); fflush(stdout); }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 1 && isatty(0)) { print_usage(argv[0]); return 0; }

    wmbus_cfg cfg;
    wmbus_default_cfg(&cfg);
    cfg.max_push_bytes = 1u << 20;
    int check_flow = 0, opt, map_only = 0, devs[64], n_devs = 0;
    const char *tcp = NULL;
    while ((opt = getopt(argc, argv, "ofad:p:r:vVst:B:G:PT:A:MUW")) != -1) {
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
        case 'V': fprintf(stdout, "rtl_wmbus: " VERSION "\n"); return EXIT_SUCCESS;
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
        default: print_usage(argv[0]); return EXIT_FAILURE;
        }
    }
    if (getenv("WMBUS_FIXED_TS")) cfg.fixed_timestamp = 1;

    if (check_flow) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = on_alarm;
        sigemptyset(&sa.sa_mask);
        fprintf(stderr, "rtl_wmbus: monitoring flow\n");
        sigaction(SIGALRM, &sa, NULL);
    }

    if (optind < argc) {
        if (n_devs == 0) { devs[0] = cfg.device; n_devs = 1; }
        if (n_devs == 1 && !map_only) return run_batch(cfg, argc - optind, argv + optind);
        return run_sharded(cfg, argc - optind, argv + optind, devs, n_devs, map_only);
    }

    FILE *input = stdin;
    if (tcp && !(input = open_tcp(tcp))) return EXIT_FAILURE;

    wmbus_ctx *ctx = NULL;
    int rc = wmbus_open(&cfg, &ctx);
    if (rc) {
        fprintf(stderr, "rtl_wmbus_hip: cannot open GPU back end: %s\n", ctx ? wmbus_last_error(ctx) : "out of memory");
        wmbus_close(ctx);
        return EXIT_FAILURE;
    }

    unsigned char *buf = malloc(cfg.max_push_bytes);
    size_t fill = 0;
    if (!buf) return EXIT_FAILURE;
    for (;;) {
        if (check_flow) alarm(2);
        const size_t got = fread(buf + fill, WMBUS_BLOCK_BYTES, 1, input);   /* whole blocks only */
        if (check_flow) alarm(0);
        if (got != 1) break;                                                 /* EOF: partial tail dropped */
        fill += WMBUS_BLOCK_BYTES;
        if (fill == cfg.max_push_bytes) { if (flush_push(ctx, buf, fill)) { rc = 1; break; } fill = 0; }
    }
    if (!rc && fill) rc = flush_push(ctx, buf, fill) ? 1 : 0;
    free(buf);
    wmbus_close(ctx);
    return rc ? EXIT_FAILURE : EXIT_SUCCESS;
}
