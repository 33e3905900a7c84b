// Is v_cvt_pk_u8_f32 the same as (uint32_t)x & 0xFF (truncation) for every float in [0, 256)?  (gfx950; candidate for K1's RSSI byte packing)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long *bad, uint32_t *first)
{
    const uint32_t n = 0x43800000u;                       // bits of 256.0f: every non-negative float below it
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const uint32_t a = (uint32_t)x & 0xFFu;
        const uint32_t b = __builtin_amdgcn_cvt_pk_u8_f32(x, 0u, 0u) & 0xFFu;
        const uint32_t c = (__builtin_amdgcn_cvt_pk_u8_f32(x, 2u, 0xAABBCCDDu));
        if (a != b || c != ((0xAABBCCDDu & ~0x00FF0000u) | (a << 16))) { if (atomicAdd(bad, 1ull) == 0) *first = (uint32_t)i; }
    }
}
int main()
{
    unsigned long long *bad, hb = 0; uint32_t *first, hf = 0;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, first);
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    float f; __builtin_memcpy(&f, &hf, 4);
    printf("cvt_pk_u8_f32 vs truncation on [0,256): %llu mismatches, first at bits 0x%08x = %.9g\n", hb, hf, f);
    return 0;
}
