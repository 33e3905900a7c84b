// tools/lone_wave.hip -- what ONE wave on a SIMD can issue (the clock-recovery kernel's situation: 256 waves on 1024 SIMDs):
// cycles per instruction (s_memtime) of independent and dependent streams of plain and PACKED f32 operations, for 1 wave on the
// chip, one wave per SIMD (1024 waves) and two per SIMD.   hipcc --offload-arch=gfx950 -O2 -o tools/lone_wave tools/lone_wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(uint64_t *out, int iters, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.999f + 1.0f;
    f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f, q = p0 * 0.999f + 1.0f;
    const uint64_t c0 = __builtin_readcyclecounter();
    for (int l = 0; l < iters; l++) {
        if (KIND == 0) {          // 8 independent plain VOP2 (x 8)
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                              "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 1) {   // 8 independent packed (x 8)
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n\tv_pk_mul_f32 %1, %1, %8\n\tv_pk_mul_f32 %2, %2, %8\n\tv_pk_mul_f32 %3, %3, %8\n\t"
                              "v_pk_add_f32 %4, %4, %8\n\tv_pk_add_f32 %5, %5, %8\n\tv_pk_add_f32 %6, %6, %8\n\tv_pk_add_f32 %7, %7, %8"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));)
        } else if (KIND == 2) {   // one dependent plain chain
            REP8(asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\t"
                              "v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b));)
        } else if (KIND == 3) {   // one dependent packed chain
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\t"
                              "v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(q));)
        } else if (KIND == 4) {   // three interleaved dependent plain chains (the clock block's shape: three biquad sections)
            REP8(asm volatile("v_mul_f32 %0, %0, %3\n\tv_mul_f32 %1, %1, %3\n\tv_mul_f32 %2, %2, %3\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %3\n\t"
                              "v_mul_f32 %0, %0, %3\n\tv_mul_f32 %1, %1, %3" : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(b));)
        } else if (KIND == 5) {   // the same work as 4 with two of the chains packed into one: packed + plain interleaved
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3\n\tv_pk_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"
                              "v_pk_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3\n\tv_pk_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(p0), "+v"(a0) : "v"(q), "v"(b));)
        }
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x == 123.456f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0;
}
template <int KIND> static void run(const char *name, uint64_t *d)
{
    const int iters = 2000;
    const struct { int blocks, threads; const char *what; } cfg[] = {{1, 64, "1 wave on the chip"}, {256, 256, "1 wave per SIMD"}, {256, 512, "2 waves per SIMD"}, {256, 1024, "4 waves per SIMD"}};
    printf("%-62s", name);
    for (auto &c : cfg) {
        hipLaunchKernelGGL(k<KIND>, dim3(c.blocks), dim3(c.threads), 0, 0, d, iters, 1.25f);
        hipDeviceSynchronize();
        const int waves = c.blocks * c.threads / 64;
        std::vector<uint64_t> h(waves + 1); hipMemcpy(h.data(), d, 8 * (waves + 1), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < waves; i++) s += (double)h[1 + i];
        printf("  %6.2f", s / waves / ((double)iters * 64));
    }
    printf("\n");
}
int main()
{
    uint64_t *d; hipMalloc(&d, 8 * (1 << 16));
    printf("shader cycles (s_memtime) per instruction of one wave; columns: 1 wave on the chip | 1 wave per SIMD | 2 per SIMD | 4 per SIMD\n");
    run<0>("8 independent streams, plain v_mul/v_add_f32", d);
    run<1>("8 independent streams, v_pk_mul/v_pk_add_f32", d);
    run<2>("1 dependent chain, plain", d);
    run<3>("1 dependent chain, packed", d);
    run<4>("3 interleaved dependent chains, plain (8 instr)", d);
    run<5>("the same flops: 1 packed chain + 1 plain chain (8 instr = 12 ops)", d);
    return 0;
}
