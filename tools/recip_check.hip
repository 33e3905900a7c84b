// tools/recip_check.hip -- can the domain division (wm_exact.h wm_div_dom) do with ONE correction step instead of two?
// Markstein: if y = RN(1/b) and q0 is a faithful a/b, then q1 = fma(fma(-b, q0, a), y, q0) = RN(a/b).  gfx950's v_rcp_f32 is good to
// 1 ulp; this program (1) checks EXHAUSTIVELY, over all 2^23 significands b in [1, 2), that one Newton step y1 = fma(fma(-b, y0, 1), y0, y0)
// is the correctly rounded reciprocal, and (2) compares the 6-instruction quotient with the compiler's correctly rounded `/` on
// 2^23 x 96 operand pairs (every significand of b against random significands and against integers a < 2^24 -- the discriminator's domain).
//   hipcc --offload-arch=gfx950 -O2 -o tools/recip_check tools/recip_check.hip && tools/recip_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float y1_of(float b)
{
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ float div6(float a, float b)
{
    const float y = y1_of(b);
    const float q = __fmul_rn(a, y);
    const float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ uint32_t rnd(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__global__ void check(unsigned long long *bad, uint32_t *first)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;             // significand of b
    const float b = __uint_as_float(0x3f800000u | m);
    const float y1 = y1_of(b), yr = 1.0f / b;
    if (__float_as_uint(y1) != __float_as_uint(yr)) { if (atomicAdd(&bad[0], 1ull) < 16ull) first[atomicAdd(&first[0], 1u) + 1u] = m; }
    uint32_t s = m * 2654435761u + 12345u;
    for (int k = 0; k < 96; k++) {
        float a;
        const uint32_t r = rnd(s);
        if (k < 32) a = __uint_as_float(0x3f800000u | (r & 0x7fffffu));                         // random significand, same binade
        else if (k < 64) a = (float)(int)(r & 0xffffffu) - (float)(1 << 23);                    // integers in (-2^23, 2^23)
        else a = __uint_as_float(((r >> 23 & 0x1f) + 112u) << 23 | (rnd(s) & 0x7fffffu));       // exponents 2^-15 .. 2^16
        const float bb = k & 1 ? b : -b * (float)(1 << (k & 14));
        const float q6 = div6(a, bb), q = a / bb;
        if (__float_as_uint(q6) != __float_as_uint(q)) atomicAdd(&bad[1], 1ull);
    }
    if (m == 0) bad[2] = 96ull << 23;
}
int main()
{
    unsigned long long *bad; uint32_t *first;
    hipMalloc(&bad, 32); hipMalloc(&first, 4 * 32); hipMemset(bad, 0, 32); hipMemset(first, 0, 4 * 32);
    hipLaunchKernelGGL(check, dim3((1u << 23) / 256), dim3(256), 0, 0, bad, first);
    unsigned long long h[3]; uint32_t f[17];
    hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost); hipMemcpy(f, first, 4 * 17, hipMemcpyDeviceToHost);
    printf("significands b in [1, 2) whose one-step reciprocal is NOT the correctly rounded one: %llu of %u\n", h[0], 1u << 23);
    for (uint32_t i = 0; i < f[0] && i < 16; i++) printf("  b = 0x%08x\n", 0x3f800000u | f[1 + i]);
    printf("6-instruction quotients differing from the correctly rounded `/`: %llu of %llu\n", h[1], h[2]);
    return h[0] || h[1];
}
