/* wm_k2_sys_blocks.h -- the clock-recovery cascade of wm_k2_clock.h cut into ROLES, one wave each (round 6).
 * Device code, included by wm_k2_clock_sys.h (and by tools/sysbench.hip, tests/emu/clock_emu.cpp).
 *
 * One wave that carries [DC remover] -> x^2 -> three biquads -> level for its 64 lanes issues 27 VALU instructions per sample,
 * and a wave alone on its SIMD issues one instruction every 4.8 cycles whatever their dependences (tools/lone_wave.hip,
 * profiles/r05_lone_wave_issue.txt): a lane's walk over a 32 768-sample segment and its warm-up is 45-57 k samples x 27 x 4.8
 * cycles, and a context's chain of launches waits for that walk three times (first pass, two list rounds).  The recurrence cannot
 * be re-associated (bit-exactness), but the cascade can be cut between its sections.  Four waves of a block, on the CU's four SIMDs:
 *     role 0: soft symbols -> [DC remover] -> slicer sign bits -> square -> section 0      (10 instructions per sample)   -> hop 0
 *     role 1: section 1 (8), and the LOADS: soft symbols from memory into the block's LDS rows, two blocks ahead              -> hop 1
 *     role 2: section 2 -> level -> clock lock: the block's sample mask                     (10)                            -> a word per block
 *     role 3: no arithmetic: the STORES -- time2 chips, slicer words, state records -- and the lanes' control
 * Role r works on 32-sample block b - r while role 0 works on block b (a systolic pipeline); a hop is one f32 per lane and sample
 * through an LDS row, two s_barriers per block.  Every value is produced by exactly the operations of iir.h:57-74 and
 * rtl_wmbus.c:497-515 in their order: bit-identical to the one-wave form.
 * (The first cut of this round gave every wave half of two sections -- feed-forward of one, feedback of the next, 6-8 instructions per
 * sample -- with the loads in role 0 and the chips behind role 3's arithmetic: 3 200 / 2 700 cycles per step of a T1/C1 / S1 block
 * (tools/sysbench.hip), bound by role 3's chips and role 0's staging.) */
#ifndef WM_K2_SYS_BLOCKS_H
#define WM_K2_SYS_BLOCKS_H

/* A hop buffer holds 32 samples of 64 lanes, sample-quad major: the four samples 4q .. 4q+3 of lane l are the 16 bytes at word
 * (q * 64 + l) * 4 -- a wave's ds_write_b128 / ds_read_b128 covers 1024 consecutive bytes, bank-conflict free.  ONE buffer per hop:
 * a step of the pipeline is  [every role reads its whole input block into registers] barrier [compute, write the output block] barrier,
 * so the producer overwrites what its consumer has just taken (double buffers and one barrier per step cost 16 KB more LDS, and
 * the block has to fit beside the demodulation kernel's: wm_k2_clock_sys.h). */
#define WM_SYS_HOP_WORDS (32 * 64)
#define WM_SYS_HOP_Q     (64 * 4)          /* words between a lane's consecutive quads */

/* this lane's 32 values of a hop (hop: already offset by 4 * lane) */
__device__ __forceinline__ void sys_hop_read(const float *hop, wm_f4 (&in)[8])
{
#pragma unroll
    for (int q = 0; q < 8; q++) in[q] = *(const wm_f4 *)(hop + WM_SYS_HOP_Q * q);
}

/* one sample through section K (iir.h:57-74; b0 == 1) */
template <int K>
__device__ __forceinline__ float sys_biquad(float v, float &h1, float &h2, const IirCoef &c)
{
    const float h0 = wm_sub(v, wm_add(wm_mul(c.a1[K], h1), wm_mul(c.a2[K], h2)));
    const float o = wm_add(wm_add(h0, wm_mul(c.b1[K], h1)), wm_mul(c.b2[K], h2));
    h2 = h1; h1 = h0;
    return o;
}

/* role 0: 32 soft symbols of this lane -> output of section 0 (hop, already offset by 4 * lane).  WARM: a warm-up block whose slicer
 * bits nobody reads (clk_block32's WARM). */
template <bool DC, bool WARM>
__device__ __forceinline__ void sys_r0_block32(float &h1, float &h2, float &dcx, float &dcy, const IirCoef &c, const wm_f4 (&x)[8], float *hop, uint32_t &bitw)
{
    uint32_t sgn = 0;                                      /* MSB-first: sample n ends up in bit 31 - n */
    const float al = 0.999f, kk = wm_div(wm_add(1.0f, al), 2.0f);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        wm_f4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float v = x[q][k];
            if (DC) { const float y = wm_add(wm_mul(kk, wm_sub(v, dcx)), wm_mul(al, dcy)); dcx = v; dcy = y; v = y; }   /* rtl_wmbus.c:501/511 */
            if (!WARM) sgn = __builtin_amdgcn_alignbit(sgn, wm_f2u(v), 31);                                      /* (sgn << 1) | signbit */
            o[k] = sys_biquad<0>(wm_mul(v, v), h1, h2, c);
        }
        *(wm_f4 *)(hop + WM_SYS_HOP_Q * q) = o;
    }
    bitw = WARM ? 0u : ~__builtin_bitreverse32(sgn);
}

/* role 1: output of section 0 -> output of section 1 */
__device__ __forceinline__ void sys_r1_block32(float &h1, float &h2, const IirCoef &c, const wm_f4 (&in)[8], float *hout)
{
#pragma unroll
    for (int q = 0; q < 8; q++) {
        wm_f4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = sys_biquad<1>(in[q][k], h1, h2, c);
        *(wm_f4 *)(hout + WM_SYS_HOP_Q * q) = o;
    }
}

/* role 2: output of section 1 -> section 2 -> clock levels of the block -> sample mask of the clock lock (rtl_wmbus.c:1092-1111: take the
 * bit at n iff the levels at n-3 .. n are L,H,H,H); clk = the last three levels, newest in bit 0 (WmClkState.clk).  WARM: a warm-up block
 * whose chips nobody looks at: the section's recurrence only (no output, no level; clk is not touched), as clk_block32's WARM. */
template <bool WARM>
__device__ __forceinline__ void sys_r2_block32(float &h1, float &h2, uint32_t &clk, const IirCoef &c, const wm_f4 (&in)[8], uint32_t &smask)
{
    uint32_t low = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (WARM) {
                const float h0 = wm_sub(in[q][k], wm_add(wm_mul(c.a1[2], h1), wm_mul(c.a2[2], h2)));
                h2 = h1; h1 = h0;
            } else low = wm_shift_in_level_low(low, wm_f2u(sys_biquad<2>(in[q][k], h1, h2, c)));
        }
    }
    if (WARM) { smask = 0u; return; }
    const uint32_t prev3 = ((clk & 1u) << 2) | (clk & 2u) | ((clk >> 2) & 1u);
    const uint64_t H = ((uint64_t)(~__builtin_bitreverse32(low)) << 3) | prev3;           /* bit n+3 = level at n */
    smask = (uint32_t)((~H) & (H >> 1) & (H >> 2) & (H >> 3));
    const uint32_t last3 = (uint32_t)(H >> 32) & 7u;                                        /* levels at 29, 30, 31 */
    clk = ((last3 & 1u) << 2) | (last3 & 2u) | ((last3 >> 2) & 1u);
}

/* the block-wide meeting point between two steps of the pipeline: LDS traffic of this wave done, NOT its global loads (role 1 keeps
 * two blocks of soft symbols in flight across it; __syncthreads() would drain them) */
__device__ __forceinline__ void wm_sys_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(WM_SYS_SLEEP)
    __builtin_amdgcn_s_sleep(WM_SYS_SLEEP);               /* timing experiment (tools/build_variant.sh): a longer step with no more instructions -- what does a clock block's RESIDENCE cost the job? */
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(WM_SYS_BARRIER_ASM)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#elif defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    __syncthreads();
#endif
}

#endif /* WM_K2_SYS_BLOCKS_H */
