// tools/clk.hip -- effective shader clock under load: s_memtime (shader cycles) vs s_memrealtime (100 MHz)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void burn(uint64_t *out, int iters, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.999f + 1.0f;
    uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int l = 0; l < iters; l++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                         "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        }
    }
    uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[0] = 1;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void burn_pk(uint64_t *out, int iters, float seed)
{
    f2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b = a0 * 0.999f + 1.0f;
    for (int l = 0; l < iters; l++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            asm volatile("v_pk_mul_f32 %0, %0, %8\n\tv_pk_mul_f32 %1, %1, %8\n\tv_pk_mul_f32 %2, %2, %8\n\tv_pk_mul_f32 %3, %3, %8\n\t"
                         "v_pk_add_f32 %4, %4, %8\n\tv_pk_add_f32 %5, %5, %8\n\tv_pk_add_f32 %6, %6, %8\n\tv_pk_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        }
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 123.456f) out[0] = 1;
}
__global__ void burn_fma(uint64_t *out, int iters, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.999f + 1.0f;
    for (int l = 0; l < iters; l++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                         "v_cvt_f32_i32 %4, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_max_f32 %6, %6, %8\n\tv_bfi_b32 %7, %7, %8, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[0] = 1;
}
typedef void (*kern_t)(uint64_t *, int, float);
static void chip(const char *name, kern_t k, uint64_t *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 8192, iters = 20000;
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.25f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double instr = (double)blocks * 4 * iters * 64;
    printf("full chip %-28s %.2f ms => nominal cycles (2.4 GHz) per wave-instruction and SIMD = %.2f\n", name, best, 2.4e9 / (instr / (best * 1e-3) / 1024));
}

int main()
{
    uint64_t *d; hipMalloc(&d, 16 * 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 256, 2048, 2048 * 4}) {
        for (int rep = 0; rep < 2; rep++) {
            const int iters = 20000;
            hipEventRecord(e0); hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, d, iters, 1.25f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> h(2 * blocks); hipMemcpy(h.data(), d, 16 * blocks, hipMemcpyDeviceToHost);
            double c = 0, r = 0; for (int i = 0; i < blocks; i++) { c += h[2 * i]; r += h[2 * i + 1]; }
            const double instr = (double)blocks * 4 * iters * 64;
            printf("blocks %5d: %.2f ms  memtime/realtime = %.3f (x100 MHz = shader MHz if memtime counts shader cycles)  wave-instr/s/SIMD(1024) = %.3e  => cycles/instr at 2.4GHz = %.2f\n",
                   blocks, ms, c / r, instr / (ms * 1e-3) / 1024, 2.4e9 / (instr / (ms * 1e-3) / 1024));
        }
    }
    chip("v_pk_mul_f32/v_pk_add_f32", burn_pk, d);
    chip("fma/cvt/lshl/max/bfi mix", burn_fma, d);
    return 0;
}
