// tools/mixed_issue.hip -- what does ONE latency-bound wave cost the throughput waves it shares a SIMD with?
// Every block is 8 (or N+1) waves per SIMD x 4 SIMDs; wave slot 0 of each SIMD runs the "guest" stream (a clock-recovery wave: one
// dependent chain / three interleaved chains / absent), the others the "host" stream (demodulation-like: 8 independent chains each).
// Reports cycles per instruction of guest and hosts and the SIMD's aggregate issue rate.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mixed_issue tools/mixed_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP8(x) x x x x x x x x
__global__ void k(uint64_t *out, uint64_t T, int guest_kind, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.999f + 1.0f;
    const int wave = threadIdx.x >> 6;
    const bool guest = wave < 4 && guest_kind >= 0;          /* the first four waves of a block land on the four SIMDs */
    uint64_t n = 0;
    const uint64_t c0 = __builtin_readcyclecounter();
    if (guest && guest_kind == 0) {          // one dependent chain
        for (; __builtin_readcyclecounter() - c0 < T; n++) {
            REP8(asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\t"
                              "v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b));)
        }
    } else if (guest && guest_kind == 1) {   // three interleaved dependent chains
        for (; __builtin_readcyclecounter() - c0 < T; n++) {
            REP8(asm volatile("v_mul_f32 %0, %0, %3\n\tv_mul_f32 %1, %1, %3\n\tv_mul_f32 %2, %2, %3\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %3\n\t"
                              "v_mul_f32 %0, %0, %3\n\tv_mul_f32 %1, %1, %3" : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(b));)
        }
    } else {                                 // host: 8 independent chains
        for (; __builtin_readcyclecounter() - c0 < T; n++) {
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                              "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) { uint64_t *q = out + 1 + 2 * ((size_t)blockIdx.x * (blockDim.x >> 6) + wave); q[0] = c1 - c0; q[1] = n * 64; }
}
int main()
{
    uint64_t *d; hipMalloc(&d, 8 * (1 << 18));
    const uint64_t T = 3000000;
    printf("%-46s %10s %10s %12s\n", "guest | hosts per SIMD", "guest c/i", "host c/i", "SIMD i/cycle");
    for (int hosts : {3, 5, 7})
        for (int gk : {-1, 0, 1}) {
            const int wps = hosts + 1, threads = 64 * 4 * wps;        /* waves per block = 4 SIMDs x wps: one block per CU */
            if (threads > 1024) {                                       /* two blocks per CU would break the placement; use 1024 = 4 per SIMD, else skip */
            }
            const int blocks = 256;
            // guest runs 1/4 of the hosts' instruction count per iteration budget (it issues ~4x slower): same wall time
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads > 1024 ? 1024 : threads), 0, 0, d, T, gk, 1.25f);
            hipDeviceSynchronize();
            const int wpb = (threads > 1024 ? 1024 : threads) / 64, waves = blocks * wpb;
            std::vector<uint64_t> h(1 + 2 * waves); hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
            double gc = 0, gi = 0, hc = 0, hi = 0, tmax = 0;
            for (int w = 0; w < waves; w++) {
                const bool guest = (w % wpb) < 4 && gk >= 0;
                const double c = (double)h[1 + 2 * w], i = (double)h[2 + 2 * w];
                if (guest) { gc += c; gi += i; } else { hc += c; hi += i; }
                if (c > tmax) tmax = c;
            }
            // aggregate issue rate per SIMD while everybody runs: approximate with total instructions / (SIMDs x mean host time)
            const double total_i = gi + hi;
            const double mean_host_t = (gc + hc) / waves;
            char name[96]; snprintf(name, sizeof name, "%s | %d", gk < 0 ? "none (all hosts)" : gk == 0 ? "one dependent chain" : "three interleaved chains", gk < 0 ? wpb : wpb - 1);
            printf("%-46s %10.2f %10.2f %12.3f\n", name, gi > 0 ? gc / gi : 0.0, hc / hi, total_i / (blocks * 4.0 * mean_host_t));
        }
    return 0;
}
