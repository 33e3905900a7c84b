// clkbench.hip -- how long does one 32-sample block of the clock kernel's biquad pipeline (clk_block32, wm_k2_clock.h) take,
// depending on how its soft symbols reach the registers?  MODE 0: no loads (the arithmetic alone); 1: the product's in-place
// queue (two 16-byte loads at every 8th tick into the registers just vacated, consumed a block later); 2: the same, but a second register set: the whole
// next block asked for at the top of a block (round 3's depth); 3: in-place, every 16-byte load issued as soon as its four
// registers retire.  Rows are `stride` floats apart; `active` lanes of each wave work.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I rtl-wmbus_amd/csrc -o tools/clkbench tools/clkbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "wm_dev.h"
#include "wm_exact.h"
#include "wm_k2_common.h"
#include "wm_k2_clock.h"

template <int MODE>
__global__ __launch_bounds__(256) void kb(const float *x, uint64_t stride, uint32_t nblk, uint32_t active, uint32_t *out)
{
    const uint32_t ln = threadIdx.x & 63u, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ln >= active) return;
    const float *row = x + ((uint64_t)wave * 64 + ln) * stride;
    WmClkState s = {};
    const IirCoef c = iir_coef(0);
    wm_f4 X[8], Y[8];
    const float *pn = row;
    auto fill = [&](wm_f4 (&Z)[8], const float *p) {
#pragma unroll
        for (int j = 0; j < 8; j++) { Z[j] = *(const wm_f4 *)(p + 4 * j); __builtin_amdgcn_sched_barrier(0); }
    };
    auto none = [&](int) {};
    auto refill = [&](int j) { X[2 * j] = *(const wm_f4 *)(pn + 8 * j); X[2 * j + 1] = *(const wm_f4 *)(pn + 8 * j + 4); };
    fill(X, row);
    uint32_t acc = 0;
    for (uint32_t b = 0; b < nblk; b++) {
        pn = row + (uint64_t)(b + 1 < nblk ? b + 1 : b) * 32;
        uint32_t bitw, smask;
        if (MODE == 0) clk_block32<false>(s, c, X, none, bitw, smask);
        else if (MODE == 1) clk_block32<false>(s, c, X, refill, bitw, smask);
        else if (MODE == 2) {
            fill(Y, pn);
            clk_block32<false>(s, c, X, none, bitw, smask);
#pragma unroll
            for (int j = 0; j < 8; j++) X[j] = Y[j];
        }
        acc += bitw ^ smask;
    }
    out[wave * 64 + ln] = acc + wm_f2u(s.h[0]);
}

template <int MODE> static void run(const char *what, const float *d_x, uint64_t stride, uint32_t nblk, uint32_t blocks, uint32_t active, uint32_t *d_out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kb<MODE>, dim3(blocks), dim3(256), 0, 0, d_x, stride, nblk, active, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s mode %d blocks %4u active %2u: %8.3f ms = %6.2f us per 32-sample block\n", what, MODE, blocks, active, ms, ms * 1e3 / nblk);
}

int main()
{
    const uint32_t nblk = 2048, waves = 1024;                 /* 65536 samples per lane */
    const uint64_t stride = (uint64_t)nblk * 32 + 256;        /* rows ~256 KB apart */
    const size_t n = (size_t)waves * 64 * stride;
    float *d_x; uint32_t *d_out;
    if (hipMalloc(&d_x, n * 4) != hipSuccess || hipMalloc(&d_out, waves * 64 * 4) != hipSuccess) { puts("alloc failed"); return 1; }
    hipMemset(d_x, 0x3c, n * 4);
    for (uint32_t blocks : {1u, 64u, 256u})
        for (uint32_t active : {1u, 64u}) {
            run<0>("arithmetic only", d_x, stride, nblk, blocks, active, d_out);
            run<1>("in-place queue, refilled at claim time", d_x, stride, nblk, blocks, active, d_out);
            run<2>("second register set, next block at the top", d_x, stride, nblk, blocks, active, d_out);
        }
    return 0;
}
