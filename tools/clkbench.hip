// clkbench.hip -- one lane-group (64 lanes) of the clock-recovery cascade: how long does a 32-sample block take
//   ONE   as one wave carrying the whole cascade (clk_block32, wm_k2_clock.h: 27 instructions per sample; blocks of 4 independent waves), or
//   SYS   as four waves of a block, one per SIMD, each a quarter of the cascade, hops through LDS, one s_barrier per block
//         (wm_k2_sys_blocks.h; round 6)
// at 1 / 2 / 4 lane-groups per SIMD-quad of the chip (256 / 512 / 1024 lane-groups), alone and beside a background kernel shaped like the
// demodulation kernel's first pass (512-thread blocks, 64 VGPRs, 35 KB LDS, dense VALU): the clock launch's own duration and what
// it costs the background.  Soft symbols are read the way the product's re-run lanes read them (lane-private 16-byte loads, two
// blocks in flight, through the wave's LDS rows); the chip loop is the product's shift-register upkeep loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I rtl-wmbus_amd/csrc -o tools/clkbench tools/clkbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
using std::min;
#include "wm_dev.h"
#include "wm_exact.h"
#include "wm_k2_common.h"
#include "wm_k2_clock.h"
#include "wm_k2_sys_blocks.h"

__device__ __forceinline__ uint32_t chip_loop(uint32_t smask, uint32_t bitw, uint32_t sr)
{
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const bool has = smask != 0u;
        if (__ballot(has) == 0ull) break;
        const uint32_t k = has ? (uint32_t)__ffs((int)smask) - 1u : 0u;
        smask &= smask - 1u;
        const uint32_t sr_new = ((sr << 1) | ((bitw >> k) & 1u)) & 0xFFFFu;
        sr = has ? sr_new : sr;
    }
    return sr;
}

// ---- ONE: the product's form ---------------------------------------------------------------------------------------
template <int PRIO> __global__ __launch_bounds__(256) void k_one(const float *x, uint64_t stride, uint32_t nblk, uint32_t *out)
{
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    __shared__ __attribute__((aligned(16))) float s_xall[4][64 * WM_CLK_XROW];
    const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6, group = blockIdx.x * 4 + wv;
    float *s_x = s_xall[wv];
    const float *row = x + ((uint64_t)group * 64 + ln) * stride;
    WmClkState s = {};
    const IirCoef c = iir_coef(0);
    wm_f4 gxA[8], gxB[8];
    auto fetch = [&](wm_f4 (&gx)[8], uint32_t b) WM_LAMBDA_INLINE {
#pragma unroll
        for (int i = 0; i < 8; i++) gx[i] = *(const wm_f4 *)(row + (uint64_t)min(b, nblk - 1u) * 32 + 4 * i);
    };
    const float *xrow = s_x + ln * WM_CLK_XROW;
    auto put = [&](const wm_f4 (&gx)[8]) WM_LAMBDA_INLINE {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; i++) *(wm_f4 *)(s_x + ln * WM_CLK_XROW + 4 * i) = gx[i];
        __builtin_amdgcn_wave_barrier();
    };
    uint32_t acc = 0, b = 0;
    auto block = [&](wm_f4 (&gx)[8]) WM_LAMBDA_INLINE {
        put(gx); fetch(gx, b + 2);
        uint32_t bitw, smask;
        clk_block32<false>(s, c, xrow, bitw, smask);
        s.sr = chip_loop(smask, bitw, s.sr);
        acc += bitw; b++;
    };
    fetch(gxA, 0); fetch(gxB, 1);
    while (b < nblk) { block(gxA); if (b < nblk) block(gxB); }
    out[group * 64 + ln] = acc + s.sr + wm_f2u(s.h[4]);
}

// ---- SYS: four roles ---------------------------------------------------------------------------------------------
struct SysLds {
    float x[64 * WM_CLK_XROW];
    float hop[3][WM_SYS_HOP_WORDS];
    uint32_t bitw[4][64];
    uint32_t rest[3 * 1024 + 512];        /* what the product's block carries besides: chip and slicer-word staging, state snapshots, control words (47.9 KB in all) */
};

// stats (optional): per wave {HW_ID, cycles before the first barrier of a step (input reads), between the barriers (work), at the second barrier}
template <int PRIO> __global__ __launch_bounds__(256) void k_sys(const float *x, uint64_t stride, uint32_t nblk, uint32_t *out, uint64_t *stats)
{
    uint64_t cw = 0, cb = 0;
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    __shared__ __attribute__((aligned(16))) SysLds lds;
    const uint32_t ln = threadIdx.x & 63u, role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), group = blockIdx.x;
    const float *row = x + ((uint64_t)group * 64 + ln) * stride;
    const IirCoef c = iir_coef(0);
    uint32_t acc = 0;
    if (role == 3) lds.rest[ln] = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    auto mark = [&](uint64_t &sum) WM_LAMBDA_INLINE { const uint64_t t = __builtin_readcyclecounter(); sum += t - t0; t0 = t; };
    if (role == 0) {
        float h1 = 0, h2 = 0, dcx = 0, dcy = 0;
        wm_f4 gxA[8], gxB[8];
        auto fetch = [&](wm_f4 (&gx)[8], uint32_t b) WM_LAMBDA_INLINE {
#pragma unroll
            for (int i = 0; i < 8; i++) gx[i] = *(const wm_f4 *)(row + (uint64_t)min(b, nblk - 1u) * 32 + 4 * i);
        };
        const float *xrow = lds.x + ln * WM_CLK_XROW;
        auto put = [&](const wm_f4 (&gx)[8]) WM_LAMBDA_INLINE {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; i++) *(wm_f4 *)(lds.x + ln * WM_CLK_XROW + 4 * i) = gx[i];
            __builtin_amdgcn_wave_barrier();
        };
        uint32_t b = 0;
        auto block = [&](wm_f4 (&gx)[8]) WM_LAMBDA_INLINE {
            wm_f4 in[8];
            if (b < nblk) {
                put(gx); fetch(gx, b + 2);
#pragma unroll
                for (int q = 0; q < 8; q++) in[q] = *(const wm_f4 *)(xrow + 4 * q);
            }
            wm_sys_barrier(); mark(cb);
            if (b < nblk) {
                uint32_t sgn = 0;
                const wm_f4 (&lo)[4] = *(const wm_f4 (*)[4])&in[0], (&hi)[4] = *(const wm_f4 (*)[4])&in[4];
                sys_r0_half16<false, false, 0>(h1, h2, dcx, dcy, c, lo, lds.hop[0] + 4u * ln, sgn);
                sys_r0_half16<false, false, 1>(h1, h2, dcx, dcy, c, hi, lds.hop[0] + 4u * ln, sgn);
                lds.bitw[b & 3u][ln] = ~__builtin_bitreverse32(sgn);
            }
            mark(cw); wm_sys_barrier(); mark(cb); b++;
        };
        fetch(gxA, 0); fetch(gxB, 1);
        while (b < nblk + 3u) { block(gxA); if (b < nblk + 3u) block(gxB); }
        acc = wm_f2u(h1);
    } else if (role == 1) {
        float g1 = 0, g2 = 0, h1 = 0, h2 = 0;
        for (uint32_t s = 0; s < nblk + 3u; s++) {
            const uint32_t b = s - 1u;
            wm_f4 in[8];
            if (b < nblk) sys_hop_read(lds.hop[0] + 4u * ln, in);
            wm_sys_barrier(); mark(cb);
            if (b < nblk) { sys_mid_block32<1>(g1, g2, h1, h2, c, in, lds.hop[1] + 4u * ln); acc += lds.bitw[b & 3u][ln]; }
            mark(cw); wm_sys_barrier(); mark(cb);
        }
        acc += wm_f2u(h1);
    } else if (role == 2) {
        float g1 = 0, g2 = 0, h1 = 0, h2 = 0;
        for (uint32_t s = 0; s < nblk + 3u; s++) {
            const uint32_t b = s - 2u;
            wm_f4 in[8];
            if (b < nblk) sys_hop_read(lds.hop[1] + 4u * ln, in);
            wm_sys_barrier(); mark(cb);
            if (b < nblk) sys_mid_block32<2>(g1, g2, h1, h2, c, in, lds.hop[2] + 4u * ln);
            mark(cw); wm_sys_barrier(); mark(cb);
        }
        acc = wm_f2u(h1);
    } else {
        float g1 = 0, g2 = 0; uint32_t clk = 0, sr = 0;
        for (uint32_t s = 0; s < nblk + 3u; s++) {
            const uint32_t b = s - 3u;
            wm_f4 in[8];
            if (b < nblk) sys_hop_read(lds.hop[2] + 4u * ln, in);
            wm_sys_barrier(); mark(cb);
            if (b < nblk) {
                uint32_t smask;
                sys_r3_block32(g1, g2, clk, c, in, smask);
                sr = chip_loop(smask, lds.bitw[b & 3u][ln], sr);
            }
            mark(cw); wm_sys_barrier(); mark(cb);
        }
        acc = sr + clk + lds.rest[ln];
    }
    out[(group * 4 + role) * 64 + ln] = acc;
    if (stats && ln == 0) {
        uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        uint64_t *q = stats + (group * 4 + role) * 3; q[0] = hw; q[1] = cw; q[2] = cb;
    }
}

// ---- background: the demodulation kernel's shape ---------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_bg(float *out, uint32_t iters)
{
    __shared__ __attribute__((aligned(16))) float pad[35 * 256];                      // 35 KB: four blocks per CU
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (uint32_t i = threadIdx.x; i < 35 * 256; i += 512) pad[i] = (float)i * 1e-4f;
    __syncthreads();
    for (uint32_t it = 0; it < iters; it++) {            // per trip: one 16-byte LDS read, 32 VALU (the demodulation kernel: ~1050 VALU, ~60 LDS accesses per thread-tile)
        const wm_f4 v = *(const wm_f4 *)(pad + 4u * ((threadIdx.x + 33u * it) & 2047u));
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = __fadd_rn(__fmul_rn(a[i], 0.999f), v[i & 3]);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    if (s == 123.456f) out[0] = s;
}

int main(int argc, char **argv)
{
    const uint32_t nblk = argc > 1 ? (uint32_t)atoi(argv[1]) : 1408u;      /* 45 056 samples: a T1/C1 lane with its warm-up */
    const uint32_t max_groups = 1024;
    const uint64_t stride = (uint64_t)nblk * 32 + 256;
    const size_t n = (size_t)max_groups * 64 * stride;
    float *d_x, *d_bg; uint32_t *d_out;
    if (hipMalloc(&d_x, n * 4) != hipSuccess || hipMalloc(&d_out, max_groups * 4 * 64 * 4) != hipSuccess || hipMalloc(&d_bg, 64) != hipSuccess) { puts("alloc failed"); return 1; }
    {
        std::vector<float> h(n);
        uint32_t r = 12345u;
        for (size_t i = 0; i < n; i++) { r = r * 1664525u + 1013904223u; h[i] = ((int)(r >> 8) - (1 << 23)) * (1.0f / (1 << 24)); }
        hipMemcpy(d_x, h.data(), n * 4, hipMemcpyHostToDevice);
    }
    uint64_t *d_stats; hipMalloc(&d_stats, max_groups * 4 * 3 * 8);
    hipStream_t s_clk, s_bg; hipStreamCreateWithFlags(&s_clk, hipStreamNonBlocking); hipStreamCreateWithFlags(&s_bg, hipStreamNonBlocking);
    hipEvent_t e0, e1, b0, b1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&b0); hipEventCreate(&b1);
    /* the two forms must agree on role 3's / the one wave's registers: a cheap sanity check of the split (the real one is the emulator suite) */
    const uint32_t bg_blocks = 65536, bg_iters = 100;       /* ~ a first pass's worth of 512-thread blocks, each a few tens of microseconds */
    auto bg = [&]() { hipLaunchKernelGGL(k_bg, dim3(bg_blocks), dim3(512), 0, s_bg, d_bg, bg_iters); };
    auto bg_time = [&]() { float t = 0; for (int rep = 0; rep < 3; rep++) { hipEventRecord(b0, s_bg); bg(); hipEventRecord(b1, s_bg); hipEventSynchronize(b1); hipEventElapsedTime(&t, b0, b1); } return t; };
    float bg_alone = bg_time();
    printf("background alone: %.3f ms (%u blocks x 512 threads)\n", bg_alone, bg_blocks);
    printf("%-5s %4s %7s | %10s %12s | %10s %12s %12s\n", "form", "prio", "groups", "alone ms", "us/block", "beside ms", "us/block", "bg ms (+%)");
    for (uint32_t groups : {64u, 256u, 512u, 1024u})
        for (int form = 0; form < 4; form++) {
            auto clk = [&]() {
                if (form == 0) hipLaunchKernelGGL(k_one<0>, dim3(groups / 4), dim3(256), 0, s_clk, d_x, stride, nblk, d_out);
                else if (form == 1) hipLaunchKernelGGL(k_one<3>, dim3(groups / 4), dim3(256), 0, s_clk, d_x, stride, nblk, d_out);
                else if (form == 2) hipLaunchKernelGGL(k_sys<0>, dim3(groups), dim3(256), 0, s_clk, d_x, stride, nblk, d_out, d_stats);
                else hipLaunchKernelGGL(k_sys<3>, dim3(groups), dim3(256), 0, s_clk, d_x, stride, nblk, d_out, d_stats);
            };
            float alone = 0, beside = 0, bgt = 0;
            auto show = [&](const char *when) {
                if (form != 2) return;
                std::vector<uint64_t> h((size_t)groups * 12); hipMemcpy(h.data(), d_stats, h.size() * 8, hipMemcpyDeviceToHost);
                double w[4] = {0, 0, 0, 0}, bw[4] = {0, 0, 0, 0}; uint32_t distinct[5] = {0, 0, 0, 0, 0};
                for (uint32_t g = 0; g < groups; g++) {
                    uint32_t m = 0;
                    for (int r = 0; r < 4; r++) { m |= 1u << ((h[(g * 4 + r) * 3] >> 4) & 3u); w[r] += (double)h[(g * 4 + r) * 3 + 1]; bw[r] += (double)h[(g * 4 + r) * 3 + 2]; }
                    distinct[__builtin_popcount(m)]++;
                }
                printf("      %s: blocks on 1/2/3/4 distinct SIMDs: %u %u %u %u; cycles per step at work | at the barrier, role 0..3:", when, distinct[1], distinct[2], distinct[3], distinct[4]);
                for (int r = 0; r < 4; r++) printf("  %.0f | %.0f", w[r] / groups / (nblk + 3), bw[r] / groups / (nblk + 3));
                printf("\n");
            };
            for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0, s_clk); clk(); hipEventRecord(e1, s_clk); hipEventSynchronize(e1); hipEventElapsedTime(&alone, e0, e1); }
            show("alone");
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(b0, s_bg); bg(); bg(); hipEventRecord(b1, s_bg);       /* two launches: the clock launch never outlives its company */
                hipEventRecord(e0, s_clk); clk(); hipEventRecord(e1, s_clk);
                hipEventSynchronize(e1); hipEventSynchronize(b1);
                hipEventElapsedTime(&beside, e0, e1); hipEventElapsedTime(&bgt, b0, b1);
            }
            show("beside");
            printf("%-5s %4d %7u | %10.3f %12.3f | %10.3f %12.3f %8.3f (%+.1f)\n", form >= 2 ? "SYS" : "ONE", (form & 1) * 3, groups, alone, alone * 1e3 / nblk, beside, beside * 1e3 / nblk,
                   bgt, 100.0 * (bgt - 2 * bg_alone) / (2 * bg_alone));
        }
    printf("background alone, again: %.3f ms\n", bg_time());
    return 0;
}
