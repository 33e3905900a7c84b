/* wm_k2_sys_blocks.h -- the clock-recovery cascade of wm_k2_clock.h cut into four ROLES, one wave each (round 6).
 * Device code, included by wm_k2_clock_sys.h (and by tools/clkbench.hip, tests/emu/clock_sys_emu.cpp).
 *
 * One wave that carries [DC remover] -> x^2 -> three biquads -> level for its 64 lanes issues 27 VALU instructions per sample,
 * and a wave alone on its SIMD issues one instruction every 4.8 cycles whatever their dependences (tools/lone_wave.hip,
 * profiles/r05_lone_wave_issue.txt): a lane's walk over a 32 768-sample segment and its warm-up is 45-57 k samples x 27 x 4.8
 * cycles, and a context's chain of launches waits for that walk three times (first pass, two list rounds).  The recurrence cannot
 * be re-associated (bit-exactness), but it can be cut ALONG the cascade: a biquad in direct form II is
 *     h0 = in - (a1 h1 + a2 h2)          feedback half: the recurrence proper
 *     out = (h0 + b1 h1) + b2 h2         feed-forward half: a function of h0 and its two predecessors only
 * so a wave that receives section k's h0 stream can compute section k's output (it keeps h1, h2 as the two values it saw last) and
 * run section k+1's feedback half on it.  Four waves of a block, on the CU's four SIMDs:
 *     role 0: soft symbols -> [DC remover] -> slicer sign bits -> square -> feedback of section 0              -> hop 0
 *     role 1: feed-forward of section 0, feedback of section 1 (+ the slicer words' way to memory)             -> hop 1
 *     role 2: feed-forward of section 1, feedback of section 2                                                  -> hop 2
 *     role 3: feed-forward of section 2, level, clock lock, chips
 * Role r works on 32-sample block b - r while role 0 works on block b (a systolic pipeline); a hop is one f32 per lane and sample
 * through an LDS row, two s_barriers per block.  Every value is produced by exactly the operations of iir.h:57-74
 * and rtl_wmbus.c:497-515 in their order: bit-identical to the one-wave form, at 6-8 instead of 27 instructions per sample and wave. */
#ifndef WM_K2_SYS_BLOCKS_H
#define WM_K2_SYS_BLOCKS_H

/* A hop buffer holds 32 samples of 64 lanes, sample-quad major: the four samples 4q .. 4q+3 of lane l are the 16 bytes at word
 * (q * 64 + l) * 4 -- a wave's ds_write_b128 / ds_read_b128 covers 1024 consecutive bytes, bank-conflict free.  ONE buffer per hop:
 * a step of the pipeline is  [every role reads its whole input block into registers] barrier [compute, write the output block] barrier,
 * so the producer overwrites what its consumer has just taken (double buffers and one barrier per step cost 24 KB more LDS, and
 * the block has to fit beside the demodulation kernel's: wm_k2_clock_sys.h). */
#define WM_SYS_HOP_WORDS (32 * 64)
#define WM_SYS_HOP_Q     (64 * 4)          /* words between a lane's consecutive quads */

/* this lane's 32 values of a hop (hop: already offset by 4 * lane) */
__device__ __forceinline__ void sys_hop_read(const float *hop, wm_f4 (&in)[8])
{
#pragma unroll
    for (int q = 0; q < 8; q++) in[q] = *(const wm_f4 *)(hop + WM_SYS_HOP_Q * q);
}

/* role 0: 16 soft symbols of this lane (half HALF of a block) -> h0 of section 0 (hop, already offset by 4 * lane); sgn collects the
 * slicer's sign bits MSB-first over the two halves.  Role 0 holds two blocks of loads in flight in a wave of 128 VGPRs: it takes its
 * row in two halves of 16 registers.  WARM: a warm-up block whose slicer bits nobody reads (clk_block32's WARM). */
template <bool DC, bool WARM, int HALF>
__device__ __forceinline__ void sys_r0_half16(float &h1, float &h2, float &dcx, float &dcy, const IirCoef &c, const wm_f4 (&x)[4], float *hop, uint32_t &sgn)
{
    const float al = 0.999f, kk = wm_div(wm_add(1.0f, al), 2.0f);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        wm_f4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float v = x[q][k];
            if (DC) { const float y = wm_add(wm_mul(kk, wm_sub(v, dcx)), wm_mul(al, dcy)); dcx = v; dcy = y; v = y; }   /* rtl_wmbus.c:501/511 */
            if (!WARM) sgn = __builtin_amdgcn_alignbit(sgn, wm_f2u(v), 31);                                      /* (sgn << 1) | signbit */
            const float h0 = wm_sub(wm_mul(v, v), wm_add(wm_mul(c.a1[0], h1), wm_mul(c.a2[0], h2)));
            h2 = h1; h1 = h0; o[k] = h0;
        }
        *(wm_f4 *)(hop + WM_SYS_HOP_Q * (4 * HALF + q)) = o;
    }
}

/* roles 1 and 2: h0 stream of section K-1 (g1, g2: its two newest values so far) -> h0 stream of section K */
template <int K>
__device__ __forceinline__ void sys_mid_block32(float &g1, float &g2, float &h1, float &h2, const IirCoef &c, const wm_f4 (&in)[8], float *hout)
{
#pragma unroll
    for (int q = 0; q < 8; q++) {
        wm_f4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float p = in[q][k];
            const float v = wm_add(wm_add(p, wm_mul(c.b1[K - 1], g1)), wm_mul(c.b2[K - 1], g2));       /* b0 == 1 */
            g2 = g1; g1 = p;
            const float h0 = wm_sub(v, wm_add(wm_mul(c.a1[K], h1), wm_mul(c.a2[K], h2)));
            h2 = h1; h1 = h0; o[k] = h0;
        }
        *(wm_f4 *)(hout + WM_SYS_HOP_Q * q) = o;
    }
}

/* role 3: h0 stream of section 2 -> clock levels of the block -> sample mask of the clock lock (rtl_wmbus.c:1092-1111: take the bit
 * at n iff the levels at n-3 .. n are L,H,H,H); clk = the last three levels, newest in bit 0 (WmClkState.clk) */
__device__ __forceinline__ void sys_r3_block32(float &g1, float &g2, uint32_t &clk, const IirCoef &c, const wm_f4 (&in)[8], uint32_t &smask)
{
    uint32_t low = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float p = in[q][k];
            const float v = wm_add(wm_add(p, wm_mul(c.b1[2], g1)), wm_mul(c.b2[2], g2));
            g2 = g1; g1 = p;
            low = wm_shift_in_level_low(low, wm_f2u(v));
        }
    }
    const uint32_t prev3 = ((clk & 1u) << 2) | (clk & 2u) | ((clk >> 2) & 1u);
    const uint64_t H = ((uint64_t)(~__builtin_bitreverse32(low)) << 3) | prev3;           /* bit n+3 = level at n */
    smask = (uint32_t)((~H) & (H >> 1) & (H >> 2) & (H >> 3));
    const uint32_t last3 = (uint32_t)(H >> 32) & 7u;                                        /* levels at 29, 30, 31 */
    clk = ((last3 & 1u) << 2) | (last3 & 2u) | ((last3 >> 2) & 1u);
}

/* the block-wide meeting point between two steps of the pipeline: LDS traffic of this wave done, NOT its global loads (role 0 keeps
 * two blocks of soft symbols in flight across it; __syncthreads() would drain them) */
__device__ __forceinline__ void wm_sys_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    __syncthreads();
#endif
}

#endif /* WM_K2_SYS_BLOCKS_H */
