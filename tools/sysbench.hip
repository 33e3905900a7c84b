// sysbench.hip -- the product's clock-recovery kernels on their own (first pass of one context-push: 128 captures x 2 chains x 64
// segments of 32 768 samples, synthetic soft symbols with a chip clock in them), one-wave form against systolic form, with the
// systolic form's cycle accounting per role (-DWM_SYS_STAMPS: control word | input reads + first barrier | work | second barrier).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DWM_SYS_STAMPS -I rtl-wmbus_amd/csrc -o tools/sysbench tools/sysbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
using std::min;
#include "wm_dev.h"
#include "wm_exact.h"
#include "wm_k2_common.h"
#include "wm_k2_clock.h"
#include "wm_k2_clock_sys.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const uint32_t S = 128, seg_len = argc > 1 ? (uint32_t)atoi(argv[1]) : 32768u, M = 1u << 21, Mcap = M + 256;
    const uint32_t nseg = M / seg_len, cap = seg_len / 4 + 8, nck = seg_len / WM_CK_SAMPLES - 1;
    const uint64_t rows = 2ull * S;
    WmPush g{};
    g.M = M; g.Mcap = Mcap; g.S = S; g.d = 2; g.flags = WM_F_T1C1 | WM_F_S1 | WM_F_T2A | WM_F_RLA | WM_F_ACCURATE;
    g.seg_len[1] = seg_len; g.nseg[1] = nseg; g.nseg_cap[1] = nseg; g.cap[1] = cap; g.warm[0] = 12288; g.warm[1] = 24576; g.s1_span = 1;
    float *d_x; uint32_t *d_bits, *d_chips, *d_counts, *d_seen, *d_ckpt, *d_err; WmClkState *d_s, *d_f, *d_c; unsigned long long *d_st;
    CK(hipMalloc(&d_x, rows * Mcap * 4)); CK(hipMalloc(&d_bits, rows * (Mcap / 32) * 4)); CK(hipMalloc(&d_chips, rows * nseg * cap * 4));
    CK(hipMalloc(&d_counts, rows * nseg * 4)); CK(hipMalloc(&d_seen, rows * nseg * 4)); CK(hipMalloc(&d_ckpt, rows * nseg * (size_t)nck * 64)); CK(hipMalloc(&d_err, 4));
    CK(hipMalloc(&d_s, rows * nseg * sizeof(WmClkState))); CK(hipMalloc(&d_f, rows * nseg * sizeof(WmClkState))); CK(hipMalloc(&d_c, rows * sizeof(WmClkState)));
    CK(hipMalloc(&d_st, 4096 * 16 * 8));
    CK(hipMemset(d_c, 0, rows * sizeof(WmClkState))); CK(hipMemset(d_err, 0, 4)); CK(hipMemset(d_seen, 0, rows * nseg * 4));
    {
        std::vector<float> h((size_t)rows * Mcap);
        uint32_t r = 99u;
        for (uint64_t row = 0; row < rows; row++) {
            const double per = row < S ? 16.0 : 48.0, ph = 0.37 * (double)row;          /* T1/C1: a chip every 8 samples; S1: every 24 */
            for (uint32_t m = 0; m < Mcap; m++) {
                r = r * 1664525u + 1013904223u;
                const float n = ((int)(r >> 8) - (1 << 23)) * (0.2f / (1 << 23));
                const bool burst = ((m / 4096u) % 5u) == (uint32_t)(row % 5u);      /* a fifth of the time: signal */
                h[row * Mcap + m] = (burst ? 0.6f * (float)(std::sin(2 * M_PI * m / per + ph) > 0 ? 1 : -1) : 0.0f) + n;
            }
        }
        CK(hipMemcpy(d_x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
#ifdef WM_SYS_STAMPS
    CK(hipMemcpyToSymbol(HIP_SYMBOL(wm_sys_stamps), &d_st, sizeof(d_st)));
#endif
    K2Args a{};
    a.g = g; a.dphi = d_x; a.bits = d_bits; a.chips = d_chips; a.counts = d_counts; a.st_start = d_s; a.st_final = d_f; a.st_carry = d_c;
    a.n_lanes = (uint32_t)(rows * nseg); a.algo = 1; a.err = d_err; a.sync_seen = d_seen; a.ckpt = d_ckpt; a.nck = nck;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t lanes = a.n_lanes;
    std::vector<uint32_t> chips_one((size_t)rows * nseg), counts(rows * nseg);
    for (int form = 0; form < 3; form++) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            if (form == 0) hipLaunchKernelGGL(k2_clock<false>, dim3(lanes / 256), dim3(256), 0, 0, a);
            else if (form == 1) hipLaunchKernelGGL((k2_clock_sys<false, true>), dim3(lanes / 64), dim3(256), sizeof(ClkSysLds), 0, a);
            else hipLaunchKernelGGL((k2_clock_sys<false, false>), dim3(lanes / 64), dim3(256), sizeof(ClkSysLds), 0, a);      /* the lane-private form (odd batch sizes, a live stream) */
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(counts.data(), d_counts, counts.size() * 4, hipMemcpyDeviceToHost));
        uint64_t total = 0; for (uint32_t v : counts) total += v;
        const uint32_t blocks_t1 = (seg_len + 12288) / 32, blocks_s1 = (seg_len + 24576) / 32;
        printf("%s: %.3f ms for %u lanes (segments of %u): %.3f us per block of the longest lane (%u blocks; T1/C1 lanes: %u); %llu chips\n", form == 0 ? "one wave" : form == 1 ? "systolic" : "systolic, lane-private",
               ms, lanes, seg_len, ms * 1e3 / blocks_s1, blocks_s1, blocks_t1, (unsigned long long)total);
        if (form == 0) chips_one = counts;
        else if (chips_one != counts) printf("  !! the two forms' chip counts differ\n");
#ifdef WM_SYS_STAMPS
        if (form >= 1) {
            const uint32_t nb = lanes / 64;
            std::vector<unsigned long long> h((size_t)nb * 16); CK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
            for (int chain = 0; chain < 2; chain++) {
                double acc[4][4] = {}; uint32_t n = 0;
                for (uint32_t b = 0; b < nb; b++) {
                    if ((b * 64 / S / nseg) != (uint32_t)chain) continue;       /* lane = (ch * nseg + seg) * S + stream */
                    if ((b * 64 / S) % nseg < 2) continue;                      /* full warm-ups only */
                    n++;
                    for (int r = 0; r < 4; r++) for (int i = 0; i < 4; i++) acc[r][i] += (double)h[((size_t)b * 4 + r) * 4 + i];
                }
                const double steps = (chain ? blocks_s1 : blocks_t1) + 3.0;
                printf("  %s blocks, cycles per step: control | reads + barrier A | work | barrier B\n", chain ? "S1   " : "T1/C1");
                for (int r = 0; r < 4; r++)
                    printf("    role %d: %6.0f | %6.0f | %6.0f | %6.0f   = %6.0f\n", r, acc[r][0] / n / steps, acc[r][1] / n / steps, acc[r][2] / n / steps, acc[r][3] / n / steps,
                           (acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3]) / n / steps);
            }
        }
#endif
    }
    return 0;
}
