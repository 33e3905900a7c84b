/* wmbus_oracle_cli.c -- stdin cu8 -> stdout datagram lines through the oracle restatement.
 * TEST INFRASTRUCTURE ONLY.  Same switches as the reference (rtl_wmbus.c:892-967) plus
 * -T (print the literal "TS" instead of the wall-clock time). */
#include "wmbus_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int main(int argc, char **argv)
{
    wmo_opts o;
    wmo_default_opts(&o);
    int c;
    while ((c = getopt(argc, argv, "ofad:p:r:vVst:T")) != -1) {
        switch (c) {
        case 'o': o.remove_dc = 1; break;
        case 'f': break;
        case 'a': o.accurate_atan = 0; break;
        case 'd': o.decimation = (unsigned)strtoul(optarg, NULL, 10); break;
        case 'p':
            if (!strcmp(optarg, "T") || !strcmp(optarg, "t")) o.t1c1_enabled = 0;
            else if (!strcmp(optarg, "S") || !strcmp(optarg, "s")) o.s1_enabled = 0;
            else return 1;
            break;
        case 'r': if (strcmp(optarg, "0")) return 1; o.rla_enabled = 0; break;
        case 't': if (strcmp(optarg, "0")) return 1; o.time2_enabled = 0; break;
        case 'v': o.show_algorithm = 1; break;
        case 's': o.simultaneous = 1; break;
        case 'T': o.fixed_timestamp = 1; break;
        default: return 1;
        }
    }
    wmo_ctx *ctx = wmo_new(&o);
    static unsigned char block[4096];
    while (fread(block, sizeof block, 1, stdin) == 1) {
        wmo_feed(ctx, block, sizeof block);
        size_t n;
        const char *t = wmo_output(ctx, &n);
        if (n) { fwrite(t, 1, n, stdout); fflush(stdout); wmo_clear_output(ctx); }
    }
    wmo_free(ctx);
    return 0;
}
