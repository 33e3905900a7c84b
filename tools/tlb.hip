// tools/tlb.hip -- does a wave whose 64 lanes stream 64 rows far apart pay for address translation?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
// each lane reads `blocks` consecutive 128-byte blocks of its own row (8 x float4), rows `stride` floats apart;
// coop = 8 lanes per row read one line (same bytes per wave, 8 rows per instruction)
__global__ void walk(const float *base, size_t stride, int blocks, int coop, float *out, int depth)
{
    const int ln = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const float *row = base + (size_t)(wave * 64 + ln) * stride;
    const float *xc = base + (size_t)(wave * 64 + (ln >> 3)) * stride + 4 * (ln & 7);
    float acc = 0;
    float4 g[2][8];
    auto fetch = [&](float4 (&gx)[8], int b) {
        if (coop) { for (int i = 0; i < 8; i++) gx[i] = *(const float4 *)(xc + (size_t)i * 8 * stride + 32 * b); }
        else { for (int i = 0; i < 8; i++) gx[i] = *(const float4 *)(row + 32 * b + 4 * i); }
    };
    fetch(g[0], 0); if (depth > 1) fetch(g[1], 1);
    for (int b = 0; b < blocks; b += 2) {
        for (int i = 0; i < 8; i++) acc += g[0][i].x + g[0][i].w;
        fetch(g[0], min(b + 2, blocks - 1));
        if (depth == 1) { for (int i = 0; i < 8; i++) acc += g[0][i].y; fetch(g[0], min(b + 3, blocks - 1)); }
        else { for (int i = 0; i < 8; i++) acc += g[1][i].x + g[1][i].w; fetch(g[1], min(b + 3, blocks - 1)); }
        // ~ some compute per block
        for (int k = 0; k < 200; k++) acc = acc * 1.0000001f + 0.5f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    const size_t rows = 131072 + 64, rowlen = 1 << 16;       // floats per row when packed densely (256 KB)
    float *d; float *o;
    const size_t total = (size_t)17 << 30;                   // 17 GB arena
    if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, total); hipMalloc(&o, rows * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 1024;
    struct { const char *name; size_t stride; int waves; } cfg[] = {
        {"3 waves, rows 4 KB apart   ", 1024, 3}, {"3 waves, rows 256 KB apart ", 65536, 3}, {"3 waves, rows 8 MB apart   ", 2097152, 3},
        {"2048 waves, rows 128KB apart", 32768, 2048}, {"2048 waves, rows 8 MB/64... ", 0, 2048}};
    for (auto &c : cfg) {
        for (int coop = 0; coop < 2; coop++) for (int depth = 1; depth <= 2; depth++) {
            size_t stride = c.stride;
            if (stride == 0) stride = total / 4 / ((size_t)c.waves * 64 + 64) & ~(size_t)1023;   // spread rows over the whole arena
            if ((size_t)c.waves * 64 * stride * 4 + (size_t)blocks * 128 > total) { printf("skip\n"); continue; }
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(walk, dim3(c.waves), dim3(64), 0, 0, d, stride, blocks, coop, o, depth);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%s stride %8zu KB coop %d depth %d: %8.3f ms  = %6.2f us per 32-sample block per wave, %7.1f GB/s\n", c.name, stride * 4 / 1024, coop, depth, best,
                   best * 1e3 / blocks, (double)c.waves * 64 * blocks * 128 / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
