/*
 * wm_kernels.hip -- gfx950 kernels for the rtl-wmbus hot path.
 *
 *   k1_demod2    time-parallel front end, one 976-sample tile per block: cu8 -> [+-325 kHz
 *                shift] -> integer boxcars -> decimate -> polar discriminator (exact fdlibm
 *                atan2f) -> FIR low-pass -> soft symbol; |s| -> EMA -> RSSI byte.
 *                                  (rtl_wmbus.c:1310-1352, 517-586, 369-392, 475-495, 1066-1067)
 *   k1_demod_ppf the same behind the polyphase pre-filter of ppf.h (option; rtl_wmbus.c:258-294).
 *   k1_verify / k1_collect / k1_commit   certify the EMA hand-offs between tiles; an uncertified
 *                tile is repaired by a list launch of k1 (one lane per chain, sequential).
 *   k2_clock     lane = (chain, stream, time segment): DC remover, slicer, squared-signal IIR
 *                band-pass, clock lock, time2 framer.            (rtl_wmbus.c:497-515, 1059,
 *                336-365, 1089-1111, 806-852)
 *   k2_rla       lane = (chain, stream, time segment): run-length framer with deglitch and
 *                bit-length tracking.                            (rtl_wmbus.c:617-803)
 *   k2_verify    compares each segment's start state with its predecessor's end state; a
 *                segment whose speculative start was wrong is re-run from the true state (and
 *                stops at the first checkpoint of the speculative pass it reproduces).
 *   k3_bursts    access-code hits -> the chips (with their RSSI bytes) a host packet decoder
 *                consumes.   k4_flatten: debug/parity view of a whole chip stream.
 *
 * Exactness: every float operation the reference performs is performed here in the same order
 * with separate roundings (wm_exact.h; the file is also built with -ffp-contract=off).  The
 * recurrences (EMA, DC, IIR, run-length state) are evaluated sequentially inside a lane; time
 * parallelism comes from starting lanes early (warm-up / look-back) and PROVING, by bitwise state
 * comparison at the hand-off, that the lane had converged onto the sequential trajectory.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wm_dev.h"
#include "wm_exact.h"

typedef short wm_s2 __attribute__((ext_vector_type(2)));

/* ---------------------------------------------------------------------------------------------
 * Filter constants (rtl_wmbus.c:372, 384, 338-341, 353-356) as decimal literals, converted by the
 * compiler to the same floats the reference's arrays hold.
 * ------------------------------------------------------------------------------------------- */
__device__ static constexpr float FIR_T[11] = {
    -0.00456638213f, -0.002571450348f, 0.02689425925f, 0.1141330398f, 0.2264456422f, 0.2793297826f,
    0.2264456422f, 0.1141330398f, 0.02689425925f, -0.002571450348f, -0.00456638213f};
__device__ static constexpr float FIR_S[46] = {
    -0.000649081282f, -0.0009491938209f, -0.001361601657f, -0.001910785234f, -0.002570133495f,
    -0.003251218426f, -0.003801634695f, -0.004012672882f, -0.003636803575f, -0.002413585945f,
    -0.0001013597693f, 0.003488892085f, 0.008461671287f, 0.01481127545f, 0.02240598045f,
    0.03098477999f, 0.0401679839f, 0.04948137286f, 0.05839197924f, 0.06635211627f, 0.07284719662f,
    0.07744230649f, 0.07982251613f, 0.07982251613f, 0.07744230649f, 0.07284719662f, 0.06635211627f,
    0.05839197924f, 0.04948137286f, 0.0401679839f, 0.03098477999f, 0.02240598045f, 0.01481127545f,
    0.008461671287f, 0.003488892085f, -0.0001013597693f, -0.002413585945f, -0.003636803575f,
    -0.004012672882f, -0.003801634695f, -0.003251218426f, -0.002570133495f, -0.001910785234f,
    -0.001361601657f, -0.0009491938209f, -0.000649081282f};

/* =============================================================================================
 * K1: demodulation tile kernels
 * ===========================================================================================*/
struct K1Args {
    WmPush g;
    float *dphi;             /* [2][S][Mcap] */
    uint8_t *rssi;           /* [2][S][Mcap] */
    const float *lut_cos;    /* [lut_n]  cosf table, built on the host with the host libm */
    const float *lut_msin;   /* [lut_n]  -sinf table                                      */
    float *ema_head;         /* [ntiles][2][S] EMA after warm-up (= value at tile_start-1); tile-major so that */
    float *ema_tail;         /* [ntiles][2][S] EMA after the tile's last valid sample       k1_verify reads coalesced */
    uint32_t ntiles;
    uint32_t *err;
    /* repair launches: grid.x walks `relist` (stream * ntiles + tile);
     * the tile's EMA is then run sequentially from its predecessor's exact tail */
    const uint32_t *relist;
    const float *ema_carry;  /* [2][S] exact EMA carried in from the previous push */
};

/* =============================================================================================
 * K1, moving-average front end (second generation of this kernel; the first one -- 1024-sample
 * tiles, five samples per thread, sliding sums, select-based arctangent, 30.7 ms -- is in the git
 * history): same arithmetic, far fewer instructions, balanced waves.
 *
 * Tile = 976 decimated samples, so that tile + 48-sample halo = 1024 = 256 threads x 4: every
 * thread owns exactly one chunk of 4 consecutive samples in stage A (the first generation spent
 * 20 wave-iterations on 1072 samples; this one 16 on 1024).
 *   stage 0  cu8 -> packed int16 (i,q) in LDS, stored so that LDS word 0 is the oldest sample the
 *            tile needs; quantisation with byte-permute + packed-int16 arithmetic.
 *   stage A  a chunk needs the 4D+16 staged samples [4cD, 4cD+4D+16): aligned ds_read_b128, packed
 *            int16 prefix sums, every boxcar (8 and 16 taps, 5 positions) one packed subtract.
 *            Discriminator on the table-driven atan2 (wm_exact.h) fed with the unscaled sums.
 *   stage B1 FIR from unskewed rows with aligned ds_read_b128 windows (13 + 4 loads instead of
 *            49 + 14 dword loads).
 *   stage B2 RSSI EMA: one wave per chain, 16 samples per lane behind a 48-sample warm-up
 *            (first generation: all four waves, 8 samples per lane behind the same warm-up).
 * LDS (words): U[max(staging, 2 magnitude rows)] | yDrT[YD] yDrS[YD] | sFin[128] sHead[128] |
 *              atan table[64] = 18.3 KB at d = 2 (8 workgroups per CU).  The magnitude rows
 *              overlay the staging area: stage A keeps its 8 magnitudes in registers until the
 *              barrier that retires the staging data.
 * ===========================================================================================*/
/* D = the decimation as a compile-time constant (2..5: the rates rtl-wmbus documents) or 0: read it
 * from the push at run time (any 1..WM_MAX_DECIM; same code with loops instead of unrolled runs). */
struct K1Geo {
    static constexpr int T = WM_K1_TILE2, NA = T + WM_K1_HALO;
    static constexpr int YD = NA + 8, YM = NA + NA / 16 + 4;
    __host__ __device__ static constexpr int nstg(int d) { return (NA * d + 16 + 8 + 7) / 8 * 8 + 8; }   /* 8 slack words in front */
    __host__ __device__ static constexpr int U(int d, bool shift)
    {
        return nstg(d) * (shift ? 2 : 1) > 2 * YM ? nstg(d) * (shift ? 2 : 1) : 2 * YM;
    }
    static constexpr size_t smem(int d, bool shift) { return (size_t)(U(d, shift) + 2 * YD + 256 + WM_ATAN_TAB_WORDS) * 4; }
};
static_assert(K1Geo::NA == 1024, "stage A maps one 4-sample chunk to each of the 256 threads");

/* 8- and 16-tap boxcar sums at the five positions a0-1 .. a0+3 of one chunk from the staged
 * samples w[0 .. 4D+16) (w[15] is the newest input of position a0-1). */
template <int D>
__device__ __forceinline__ void k1_boxcars(const uint32_t *w, int d_rt, wm_s2 s8[5], wm_s2 s16[5])
{
    if (D == 0) {                                             /* run-time decimation: direct sums */
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int n = j * d_rt + 15;
            wm_s2 lo = {0, 0}, hi = {0, 0};
            for (int k = 0; k < 8; k++) { lo += __builtin_bit_cast(wm_s2, w[n - k]); hi += __builtin_bit_cast(wm_s2, w[n - 8 - k]); }
            s8[j] = lo; s16[j] = lo + hi;
        }
        return;
    }
    constexpr int N = 4 * (D ? D : 1) + 16;
    uint32_t x[N];
#pragma unroll
    for (int k = 0; k < N / 4; k++) {
        const uint4 v = *(const uint4 *)(w + 4 * k);
        x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
    }
    wm_s2 P[N];
    P[0] = __builtin_bit_cast(wm_s2, x[0]);
#pragma unroll
    for (int k = 1; k < N; k++) P[k] = P[k - 1] + __builtin_bit_cast(wm_s2, x[k]);
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int n = j * D + 15;
        s8[j] = P[n] - P[n - 8];
        s16[j] = n >= 16 ? P[n] - P[n - 16] : P[n];
    }
}

/* Stages B1 (FIR) and B2 (RSSI EMA + hand-off certification) of a 976-sample tile; shared by the
 * moving-average and the polyphase front ends.  Rows: element a of a discriminator row at word
 * a + 4, of a magnitude row at a + a/16 (the two chains' rows may alias when they carry the same
 * data).  Ends with the magnitude rows' barrier already passed by every thread. */
__device__ __forceinline__ void k1_fir_t(const K1Args &a, const float *yDrT, const int slot, const int stream, const int ts, const int tn)
{
    const WmPush &g = a.g;
    const int m0l = 4 * slot;
    if (m0l >= tn) return;
    float w[16];                                              /* w[i] = element 4 slot + 36 + i */
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float4 v = *(const float4 *)(yDrT + 4 * slot + 40 + 4 * k);
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 11; k++) s = wm_add(s, wm_mul(FIR_T[k], w[12 + j - k]));
        acc[j] = s;
    }
    *(float4 *)(a.dphi + (uint64_t)stream * g.Mcap + (uint64_t)ts + m0l) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

__device__ __forceinline__ void k1_fir_s(const K1Args &a, const float *yDrS, const int slot, const int stream, const int ts, const int tn)
{
    const WmPush &g = a.g;
    const int m0l = 4 * slot;
    if (m0l >= tn) return;
    float w[52];                                              /* w[i] = element 4 slot + i */
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const float4 v = *(const float4 *)(yDrS + 4 * slot + 4 + 4 * k);
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 46; k++) s = wm_add(s, wm_mul(FIR_S[k], w[48 + j - k]));
        acc[j] = s;
    }
    *(float4 *)(a.dphi + ((uint64_t)g.S + stream) * g.Mcap + (uint64_t)ts + m0l) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

/* Stages B1 (FIR low-pass, y[n] = sum_k b[k] x[n-k], k ascending, fir.h:48-72) and B2 (RSSI EMA,
 * rtl_wmbus.c:475-495, + hand-off certification) of a 976-sample tile; shared by the moving-average
 * and the polyphase front ends.  Rows: element a of a discriminator row at word a + 4, of a
 * magnitude row at a + a/16 (the two chains' rows may alias when they carry the same data).
 * Work split: waves 0 and 1 run one chain's EMA each (16 samples per lane behind the warm-up) and
 * the 11-tap FIR of half the tile; waves 2 and 3 the 46-tap FIR of half the tile each -- 630 against
 * 860 instructions, instead of 960 on the EMA waves and 530 on the others. */
__device__ __forceinline__ void k1_stage_b(const K1Args &a, const int tid, const int tile, const int stream, const int ts, const int tn,
                                           const bool chT, const bool chS, const float *yDrT, const float *yDrS,
                                           const float *yMgT, const float *yMgS, float *sFin, float *sHead)
{
    constexpr int T = WM_K1_TILE2;
    const WmPush &g = a.g;
    __syncthreads();                                          /* magnitude rows complete */
    const int wv = tid >> 6, e = tid & 63;
    const int ch = wv & 1;                                    /* EMA chain of waves 0, 1 */
    const bool on = wv < 2 && (ch ? chS : chT) && 16 * e < T;
    const float al = 0.6789f, be = wm_sub(1.0f, 0.6789f);
    const int m0l = 16 * e;
    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint64_t rows = 2ull * g.S, ti = (uint64_t)tile * rows + row;
    if (a.relist != nullptr) {
        /* REPAIR: a hand-off of this tile could not be certified (e.g. exact-zero input after a
         * signal: the true state decays through 90 more samples while a warm-up from zero is already
         * at zero).  One lane per chain runs the whole tile sequentially from the predecessor's exact
         * tail -- slow, exact, and only for the listed tiles. */
        if (chT) k1_fir_t(a, yDrT, tid, stream, ts, tn);
        if (chS) k1_fir_s(a, yDrS, tid, stream, ts, tn);
        if (on && e == 0) {
            const float *mrow = ch ? yMgS : yMgT;
            float ema = tile ? a.ema_tail[ti - rows] : a.ema_carry[row];
            const float head = ema;
            uint8_t *o = a.rssi + row * g.Mcap + ts;
            for (int m = 0; m < tn; m++) {
                const int el = WM_K1_HALO + m;
                ema = wm_add(wm_mul(al, mrow[el + (el >> 4)]), wm_mul(be, ema));
                o[m] = (uint8_t)((uint32_t)ema & 0xFFu);
            }
            a.ema_head[ti] = head; a.ema_tail[ti] = ema;
        }
        return;
    }
    float ema = 0.0f, tail = 0.0f, head = 0.0f;
    if (wv < 2) {
        if (on) {
            const float *mg = (ch ? yMgS : yMgT) + 17 * e;    /* element 16 e + kk at 17 e + kk + kk/16 */
#pragma unroll
            for (int k = WM_K1_HALO - WM_EMA_WARMUP; k < WM_K1_HALO; k++)     /* the last WM_EMA_WARMUP halo samples */
                ema = wm_add(wm_mul(al, mg[k + (k >> 4)]), wm_mul(be, ema));
            head = ema;
            uint32_t pk[4] = {0u, 0u, 0u, 0u};
            if (tn == T) {                                    /* full tile: the tail is lane 60's last value */
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    ema = wm_add(wm_mul(al, mg[WM_K1_HALO + k + ((WM_K1_HALO + k) >> 4)]), wm_mul(be, ema));
                    pk[k >> 2] |= ((uint32_t)ema & 0xFFu) << (8 * (k & 3));
                }
                tail = ema;
            } else {                                          /* last tile of a push */
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    ema = wm_add(wm_mul(al, mg[WM_K1_HALO + k + ((WM_K1_HALO + k) >> 4)]), wm_mul(be, ema));
                    pk[k >> 2] |= ((uint32_t)ema & 0xFFu) << (8 * (k & 3));
                    if (m0l + k == tn - 1) tail = ema;
                }
            }
            if (m0l < tn)
                *(uint4 *)(a.rssi + row * g.Mcap + ts + m0l) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            sFin[tid] = ema; sHead[tid] = head;
        }
        if (chT) { k1_fir_t(a, yDrT, 128 * wv + e, stream, ts, tn); k1_fir_t(a, yDrT, 128 * wv + 64 + e, stream, ts, tn); }
    } else if (chS) {
        k1_fir_s(a, yDrS, 128 * (wv - 2) + e, stream, ts, tn);
        k1_fir_s(a, yDrS, 128 * (wv - 2) + 64 + e, stream, ts, tn);
    }
    __syncthreads();
    /* certify: a lane's warm-up must have landed exactly on its predecessor's trajectory; a tile
     * with an uncertified lane publishes a head that cannot match (NaN) and is repaired */
    const bool bad = on && e > 0 && m0l < tn && wm_f2u(head) != wm_f2u(sFin[tid - 1]);
    const unsigned long long badT = __ballot(bad);            /* waves 0 and 1 are the two chains */
    if (on) {
        if (e == 0) a.ema_head[ti] = badT ? wm_u2f(0x7FC00000u) : head;
        if (m0l <= tn - 1 && tn - 1 < m0l + 16) a.ema_tail[ti] = tail;
    }
}

template <int D, bool SHIFT>
__global__ __launch_bounds__(256) void k1_demod2(K1Args a)
{
    using G = K1Geo;
    constexpr int T = G::T, NA = G::NA, YD = G::YD, YM = G::YM;
    const WmPush &g = a.g;
    const int d = D ? D : (int)g.d;
    const int NSTG = G::nstg(d);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *stgT = (uint32_t *)smem + 8;                /* word 0 of a row = oldest sample of the tile */
    uint32_t *stgS = SHIFT ? stgT + NSTG : stgT;
    float *yMgT = (float *)smem, *yMgS = yMgT + YM;       /* overlay the staging rows (see stage A) */
    float *yDrT = (float *)smem + G::U(d, SHIFT), *yDrS = yDrT + YD;
    float *sFin = yDrS + YD, *sHead = sFin + 128, *tab = sHead + 128;

    const int tid = threadIdx.x;
    const int tile = a.relist ? (int)(a.relist[blockIdx.x] % a.ntiles) : (int)blockIdx.x;
    const int stream = a.relist ? (int)(a.relist[blockIdx.x] / a.ntiles) : (int)blockIdx.y;
    const int ts = tile * T;
    const int tn = min(T, (int)g.M - ts);
    const bool chT = g.flags & WM_F_T1C1, chS = g.flags & WM_F_S1;
    const bool accurate = g.flags & WM_F_ACCURATE;

    if (tid < WM_ATAN_TAB_WORDS) wm_atan_tab_word(tid, tab);   /* 5 range rows + range LUT */

    /* ---- stage 0: one dword (two IQ samples) per lane and pass: coalesced loads, LDS stores at a
     * two-word lane stride (the 16-byte-per-lane variant stored at an 8-word stride: 8-way bank
     * conflicts) ------------------------------------------------------------------------------- */
    {
        const long r_lo = ((long)(g.m0 + (uint64_t)ts) - WM_K1_HALO) * d - 16 - (long)g.n0;   /* LDS word 0 */
        const long r_al = r_lo & ~1L;
        const int off = (int)(r_lo - r_al);                   /* 0 or 1 */
        const int NDW = (NA * d + 16 + 1 + 1) / 2;            /* dwords covering the staged samples */
        const uint8_t *base = g.in + (uint64_t)stream * g.in_stride + WM_HIST_BYTES;
        const uint32_t *src = (const uint32_t *)(base + 2 * r_al);
        constexpr int NP = D ? ((NA * D + 16 + 1 + 1) / 2 + 255) / 256 : 1;     /* loads in flight per lane */
        const int passes = D ? 1 : (NDW + 255) / 256;
        for (int ps = 0; ps < passes; ps++) {
        uint32_t wv[NP];
#pragma unroll
        for (int it = 0; it < NP; it++) {
            const int u = tid + 256 * (it + ps);
            wv[it] = u < NDW ? src[u] : 0u;
        }
#pragma unroll
        for (int it = 0; it < NP; it++) {
            const int u = tid + 256 * (it + ps);
            if (u < NDW) {
                const int p = 2 * u - off;
                if (!SHIFT) {
                    /* bytes (i,q) -> halfwords, then u - 127 - (u >> 7) per halfword
                     * (= (int)((float)u - 127.5f), rtl_wmbus.c:1312-1313 + moving_average_filter.h:47) */
                    const wm_s2 c127 = {127, 127};
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t h = __builtin_amdgcn_perm(0u, wv[it], k ? 0x0c030c02u : 0x0c010c00u);
                        const wm_s2 q = __builtin_bit_cast(wm_s2, h) - c127 - __builtin_bit_cast(wm_s2, (h >> 7) & 0x00010001u);
                        stgT[p + k] = __builtin_bit_cast(uint32_t, q);
                    }
                } else {
                    /* LUT index of global sample n: (13 n) mod lut_n, rtl_wmbus.c:1006-1010 */
                    const int L = (int)g.lut_n;
                    int rm = (int)((r_al + 2L * u) % L); if (rm < 0) rm += L;
                    uint32_t li = (g.lut_phase0 + 13u * (uint32_t)rm) % (uint32_t)L;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t iq = (wv[it] >> (16 * k)) & 0xFFFFu;
                        const float fi = wm_sub((float)(iq & 0xFFu), 127.5f), fq = wm_sub((float)(iq >> 8), 127.5f);
                        const float x = a.lut_cos[li], z = a.lut_msin[li];
                        li += 13u; if (li >= g.lut_n) li -= g.lut_n;
                        const float ix = wm_mul(fi, x), qx = wm_mul(fq, x), iz = wm_mul(fi, z), qz = wm_mul(fq, z);
                        wm_s2 t, sv;
                        t.x = (short)(int)wm_sub(ix, qz); t.y = (short)(int)wm_add(qx, iz);
                        sv.x = (short)(int)wm_add(ix, qz); sv.y = (short)(int)wm_sub(qx, iz);
                        stgT[p + k] = __builtin_bit_cast(uint32_t, t); stgS[p + k] = __builtin_bit_cast(uint32_t, sv);
                    }
                }
            }
        }
        }
    }
    __syncthreads();

    /* ---- stage A: thread = chunk ------------------------------------------------------------- */
    float mgT[4], mgS[4];
    {
        const int c = tid;
        wm_s2 s8[5], s16[5], u8[5], u16[5];
        k1_boxcars<D>(stgT + 4 * c * d, d, s8, SHIFT ? u16 : s16);
        if (SHIFT) k1_boxcars<D>(stgS + 4 * c * d, d, u8, s16);
        /* the eight arctangents of a thread (4 samples x 2 chains) are independent: computed in one
         * straight-line region (the accurate / -a choice hoisted out of the loops), the scheduler
         * interleaves their dependent chains */
        float drT[4], drS[4], fT[5][2], fS[5][2];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            fT[j][0] = (float)s8[j].x; fT[j][1] = (float)s8[j].y;          /* 8 x the reference's i, q */
            fS[j][0] = (float)s16[j].x; fS[j][1] = (float)s16[j].y;       /* 16 x */
        }
        if (accurate && chT && chS) {                        /* default switches: no branch between the eight */
#pragma unroll
            for (int j = 0; j < 4; j++) {
                drT[j] = wm_discriminator_tab(fT[j + 1][0], fT[j + 1][1], fT[j][0], fT[j][1], tab);
                drS[j] = wm_discriminator_tab(fS[j + 1][0], fS[j + 1][1], fS[j][0], fS[j][1], tab);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float iT = fT[j + 1][0], qT = fT[j + 1][1], iS = fS[j + 1][0], qS = fS[j + 1][1];
                mgT[j] = wm_mul(wm_sqrt_dom(wm_add(wm_mul(iT, iT), wm_mul(qT, qT))), 0.125f);
                mgS[j] = wm_mul(wm_sqrt_dom(wm_add(wm_mul(iS, iS), wm_mul(qS, qS))), 0.0625f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float iT = fT[j + 1][0], qT = fT[j + 1][1], iS = fS[j + 1][0], qS = fS[j + 1][1];
                drT[j] = !chT ? 0.0f : accurate ? wm_discriminator_tab(iT, qT, fT[j][0], fT[j][1], tab)
                                                : wm_mul(wm_discriminator_fast(iT, qT, fT[j][0], fT[j][1]), 0.015625f);
                drS[j] = !chS ? 0.0f : accurate ? wm_discriminator_tab(iS, qS, fS[j][0], fS[j][1], tab)
                                                : wm_mul(wm_discriminator_fast(iS, qS, fS[j][0], fS[j][1]), 0.00390625f);
                mgT[j] = chT ? wm_mul(wm_sqrt_dom(wm_add(wm_mul(iT, iT), wm_mul(qT, qT))), 0.125f) : 0.0f;
                mgS[j] = chS ? wm_mul(wm_sqrt_dom(wm_add(wm_mul(iS, iS), wm_mul(qS, qS))), 0.0625f) : 0.0f;
            }
        }
        /* element a of a discriminator row lives at word a + 4 */
        *(float4 *)(yDrT + 4 * c + 4) = make_float4(drT[0], drT[1], drT[2], drT[3]);
        *(float4 *)(yDrS + 4 * c + 4) = make_float4(drS[0], drS[1], drS[2], drS[3]);
    }
    __syncthreads();                                          /* staging data retired */
    {   /* element a of a magnitude row lives at word a + a/16 (conflict-free 17-word lane stride in B2) */
        const int qb = 4 * tid + (tid >> 2);
#pragma unroll
        for (int j = 0; j < 4; j++) { yMgT[qb + j] = mgT[j]; yMgS[qb + j] = mgS[j]; }
    }

    k1_stage_b(a, tid, tile, stream, ts, tn, chT, chS, yDrT, yDrS, yMgT, yMgS, sFin, sHead);
}

/* =============================================================================================
 * K1 with the POLYPHASE pre-filter: ppf.h:46-59 driven as the reference's
 * lp_ppf_butter_1600kHz_160kHz_200kHz does (rtl_wmbus.c:258-294): even input samples through the
 * 12 taps b[1], odd ones through b[0], y[m] = (0 + F_even[m]) + F_odd[m] taken after the odd
 * sample -- in place of the two moving averages.  The reference defines this filter but never calls
 * it, so it is an OPTION here (cfg.prefilter = 1, d = 2, no -s) and is pinned at stage level: the
 * reference's own function, driven by oracle/ref_probe.c, against the oracle, and the oracle against
 * this kernel.  One filtered (i,q) pair feeds both chains, so discriminator and magnitude are
 * computed once; the operands are arbitrary floats, hence the general wm_atan2f / wm_sqrt.
 * LDS (words): float2 staging[2 NA + 24] (the magnitude row overlays it) | yDr[YD] | sFin, sHead.
 * ===========================================================================================*/
__device__ static constexpr float PPF_EVEN[12] = {1.102280392e-05f, 0.001356012537f, 0.01499414005f, 0.05525973093f,
    0.1099887688f, 0.1366692652f, 0.1099887688f, 0.05525973093f, 0.01499414005f, 0.001356012537f, 1.102280392e-05f, 0.0f};
__device__ static constexpr float PPF_ODD[12] = {0.000140535927f, 0.0001309279731f, 0.00551787474f, 0.03160167988f,
    0.08315031015f, 0.1295143636f, 0.1295143636f, 0.08315031015f, 0.03160167988f, 0.00551787474f, 0.0001309279731f,
    0.000140535927f};

struct K1PpfGeo {
    static constexpr int T = WM_K1_TILE2, NA = T + WM_K1_HALO;
    static constexpr int NSTG = 2 * (2 * NA + 24);                    /* words: float2 per input sample */
    static constexpr int YD = NA + 8, YM = NA + NA / 16 + 4;
    static constexpr size_t smem() { return (size_t)(NSTG + YD + 256) * 4; }
};

__global__ __launch_bounds__(256) void k1_demod_ppf(K1Args a)
{
    using G = K1PpfGeo;
    constexpr int T = G::T, NA = G::NA;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *stg = (float2 *)smem;                            /* element 0 = input sample 2 (m_first - 12) */
    float *yMg = (float *)smem;                              /* overlays the staging after stage A */
    float *yDr = (float *)smem + G::NSTG;
    float *sFin = yDr + G::YD, *sHead = sFin + 128;

    const WmPush &g = a.g;
    const int tid = threadIdx.x;
    const int tile = a.relist ? (int)(a.relist[blockIdx.x] % a.ntiles) : (int)blockIdx.x;
    const int stream = a.relist ? (int)(a.relist[blockIdx.x] / a.ntiles) : (int)blockIdx.y;
    const int ts = tile * T;
    const int tn = min(T, (int)g.M - ts);
    const bool chT = g.flags & WM_F_T1C1, chS = g.flags & WM_F_S1;
    const bool accurate = g.flags & WM_F_ACCURATE;

    /* ---- stage 0: bytes -> floats (rtl_wmbus.c:1312-1313), no truncation on this path; samples
     * before the start of the stream are the filters' zero history, not the byte the input window
     * was pre-filled with ---------------------------------------------------------------------- */
    {
        const long n_first = 2L * ((long)(g.m0 + (uint64_t)ts) - WM_K1_HALO - 1 - 11);     /* global input index of element 0 */
        const long r_lo = n_first - (long)g.n0;                                            /* even: n0 is a multiple of 2048 */
        const uint8_t *base = g.in + (uint64_t)stream * g.in_stride + WM_HIST_BYTES;
        const uint32_t *src = (const uint32_t *)(base + 2 * r_lo);
        constexpr int NDW = (2 * NA + 24) / 2;
        for (int u = tid; u < NDW; u += 256) {
            const uint32_t w = src[u];
            const bool live = n_first + 2L * u >= 0;
            float4 v;
            v.x = live ? wm_sub((float)(w & 0xFFu), 127.5f) : 0.0f;
            v.y = live ? wm_sub((float)((w >> 8) & 0xFFu), 127.5f) : 0.0f;
            v.z = live ? wm_sub((float)((w >> 16) & 0xFFu), 127.5f) : 0.0f;
            v.w = live ? wm_sub((float)(w >> 24), 127.5f) : 0.0f;
            *(float4 *)(stg + 2 * u) = v;
        }
    }
    __syncthreads();

    /* ---- stage A: thread = 4 consecutive decimated samples (+ the one before, for the
     * discriminator); output j (a = 4 tid - 1 + j) uses staged samples 2 (j + 11 - k) [+ 1] of the
     * thread's 32-sample window ---------------------------------------------------------------- */
    float mg[4];
    {
        float2 x[32];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float4 v = *(const float4 *)(stg + 8 * tid + 2 * k);
            x[2 * k] = make_float2(v.x, v.y); x[2 * k + 1] = make_float2(v.z, v.w);
        }
        float fi[5], fq[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            float ei = 0.0f, eq = 0.0f, oi = 0.0f, oq = 0.0f;        /* fir.h:58-67: accumulate from 0, taps ascending */
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const float2 e = x[2 * (j + 11 - k)], o = x[2 * (j + 11 - k) + 1];
                ei = wm_add(ei, wm_mul(PPF_EVEN[k], e.x)); eq = wm_add(eq, wm_mul(PPF_EVEN[k], e.y));
                oi = wm_add(oi, wm_mul(PPF_ODD[k], o.x)); oq = wm_add(oq, wm_mul(PPF_ODD[k], o.y));
            }
            fi[j] = wm_add(wm_add(0.0f, ei), oi);                     /* ppf.h:49-54: sum = 0; sum += even; sum += odd */
            fq[j] = wm_add(wm_add(0.0f, eq), oq);
        }
        float dr[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float i = fi[j + 1], q = fq[j + 1], pi_ = fi[j], pq_ = fq[j];
            dr[j] = accurate ? wm_discriminator(i, q, pi_, pq_) : wm_discriminator_fast(i, q, pi_, pq_);
            mg[j] = wm_sqrt(wm_add(wm_mul(i, i), wm_mul(q, q)));
        }
        *(float4 *)(yDr + 4 * tid + 4) = make_float4(dr[0], dr[1], dr[2], dr[3]);
    }
    __syncthreads();                                          /* staging data retired */
    {
        const int qb = 4 * tid + (tid >> 2);
#pragma unroll
        for (int j = 0; j < 4; j++) yMg[qb + j] = mg[j];
    }
    k1_stage_b(a, tid, tile, stream, ts, tn, chT, chS, yDr, yDr, yMg, yMg, sFin, sHead);
}

/* head[tile] must equal tail[tile-1] (or the value carried from the previous push).  One thread
 * per (tile, row) -- a per-row scan over 2150 tiles is a millisecond of dependent latency -- keeps
 * the FIRST tile of each row that does not in first_bad[row]; k1_collect turns those into the
 * repair list (tiles after a bad one cannot be judged before it is repaired). */
__global__ void k1_verify(const float *head, const float *tail, const float *carry, uint32_t ntiles,
                          uint32_t rows, uint32_t *first_bad)
{
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;   /* row = chain*S + stream */
    if (row >= rows) return;
    const float prev = t ? tail[(uint64_t)(t - 1) * rows + row] : carry[row];
    if (wm_f2u(head[(uint64_t)t * rows + row]) != wm_f2u(prev)) atomicMin(first_bad + row, t);
}

__global__ void k1_collect(uint32_t *first_bad, uint32_t ntiles, uint32_t rows, uint32_t S, uint32_t *relist, uint32_t *n_relist)
{
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const uint32_t t = first_bad[row];
    if (t < ntiles) { relist[atomicAdd(n_relist, 1u)] = (row % S) * ntiles + t; first_bad[row] = 0xFFFFFFFFu; }
}

/* every hand-off certified: the last tile's tail becomes the carry of the next push */
__global__ void k1_commit(const float *tail, float *carry, uint32_t ntiles, uint32_t rows)
{
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < rows) carry[row] = tail[(uint64_t)(ntiles - 1) * rows + row];
}

/* =============================================================================================
 * K2: sequential lanes over time segments
 * ===========================================================================================*/
struct K2Args {
    WmPush g;
    const float *dphi;
    const uint8_t *rssi;
    uint32_t *bits;            /* [2][S][Mcap/32]                                          */
    uint32_t *chips;           /* base of this algo's regions: [2][S][nseg_cap][cap]       */
    uint32_t *counts;          /* [2][S][nseg_cap]                                          */
    void *st_start;            /* state each segment's main loop started from               */
    void *st_final;            /* state after the segment's last sample                     */
    void *st_carry;            /* [2][S] exact state carried from the previous push         */
    const uint32_t *list;      /* re-run list of lane ids, or nullptr                       */
    uint32_t n_lanes;
    uint32_t algo;             /* WMBUS_ALGO_* of this launch                              */
    uint32_t *err;
    uint32_t *sync_seen;       /* [2][S][nseg_cap]: set when a pass emitted an access-code chip into the region */
    /* checkpoints of the speculative pass, every WM_CK_SAMPLES inside a segment: lane state + chips so
     * far (16 words each).  A re-run stops at the first checkpoint it reproduces: from there on the
     * speculative pass had already been on the exact trajectory. */
    uint32_t *ckpt; uint32_t nck;
};

__device__ __forceinline__ void lane_decode(const WmPush &g, uint32_t algo, uint32_t lane, uint32_t &ch, uint32_t &stream, uint32_t &seg)
{
    /* lane = (ch * nseg + seg) * S + stream : neighbouring lanes = neighbouring streams */
    stream = lane % g.S;
    const uint32_t r = lane / g.S;
    seg = r % g.nseg[algo];
    ch = r / g.nseg[algo];
}

struct IirCoef { float a1[3], a2[3], b1[3], b2[3]; };

__device__ __forceinline__ IirCoef iir_coef(uint32_t ch)
{
    IirCoef c;
    if (ch == 0) { /* rtl_wmbus.c:340-341 */
        c.b1[0] = 1.999994649f; c.b2[0] = 0.9999946492f; c.b1[1] = -1.99999482f; c.b2[1] = 0.9999948196f;
        c.b1[2] = 1.703868036e-07f; c.b2[2] = -1.000010531f;
        c.a1[0] = -1.387139203f; c.a2[0] = 0.9921518712f; c.a1[1] = -1.403492665f; c.a2[1] = 0.9845934971f;
        c.a1[2] = -1.430055639f; c.a2[2] = 0.9923856172f;
    } else {       /* rtl_wmbus.c:355-356 */
        c.b1[0] = 1.999994187f; c.b2[0] = 0.9999941867f; c.b1[1] = -1.999994026f; c.b2[1] = 0.9999940262f;
        c.b1[2] = -1.605750097e-07f; c.b2[2] = -1.000011787f;
        c.a1[0] = -1.92151475f; c.a2[0] = 0.9918135499f; c.a1[1] = -1.922481015f; c.a2[1] = 0.984593497f;
        c.a1[2] = -1.937432099f; c.a2[2] = 0.9927241336f;
    }
    return c;
}

/* One sample through DC remover + squarer + 3 biquads; returns the clock level (iir.h:57-74). */
__device__ __forceinline__ bool clk_step(WmClkState &s, const IirCoef &c, bool dc, float x, float &soft)
{
    if (dc) { /* rtl_wmbus.c:501/511: (1+a)/2 * (x - x_old) + a * y_old, a = 0.999f */
        const float al = 0.999f, k = wm_div(wm_add(1.0f, al), 2.0f);
        const float y = wm_add(wm_mul(k, wm_sub(x, s.dc_x)), wm_mul(al, s.dc_y));
        s.dc_x = x; s.dc_y = y; x = y;
    }
    soft = x;
    float v = wm_mul(x, x);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float h1 = s.h[2 * k], h2 = s.h[2 * k + 1];
        const float h0 = wm_sub(v, wm_add(wm_mul(c.a1[k], h1), wm_mul(c.a2[k], h2)));
        v = wm_add(wm_add(h0, wm_mul(c.b1[k], h1)), wm_mul(c.b2[k], h2));   /* b0 == 1 */
        s.h[2 * k + 1] = h1; s.h[2 * k] = h0;
    }
    return wm_mul(v, 1.874981046e-06f) >= 0.0f;
}

/* Exact truncating signed division for |a| < 2^24, 0 < b < 2^12 via one float reciprocal and a
 * +-1 fix-up (the hardware integer divide is ~40 instructions and sits on the serial path). */
__device__ __forceinline__ int wm_sdiv(int a, int b)
{
    const unsigned ua = (unsigned)(a < 0 ? -a : a);
    if (ua >= (1u << 24) || (unsigned)b >= (1u << 12)) return a / b;
    unsigned q = (unsigned)((float)ua * __frcp_rn((float)b));
    const int r = (int)ua - (int)(q * (unsigned)b);
    if (r < 0) q--; else if (r >= b) q++;
    return a < 0 ? -(int)q : (int)q;
}

#define WM_RLA_CROW 17           /* words per lane in the run-length kernel's chip staging (16 + 1: conflict-free) */
#define WM_CLK_XROW 36           /* words per lane in the clock kernel's soft-symbol buffer: 32 + 4 (rows stay 16-byte
                                    aligned; a lane's 8 ds_read_b128 are bank-conflict free: 9 L mod 16 is a permutation) */
#define WM_CLK_CROW 17           /* words per lane in its chip staging (16 + 1) */
#define WM_CLK_BROW 9            /* words per lane in its slicer-word staging (8 + 1) */

/* 32 samples through [DC remover] -> x^2 -> 3 biquads -> clock level, SOFTWARE-PIPELINED across the
 * filter sections: at tick t section k works on sample t - k, so the three (four with -o) recurrences
 * of a tick are independent instruction streams; a lone wave issues a dependent VALU operation only
 * every ~8.5 cycles on gfx950, and the straight per-sample order is one 36-deep dependent chain.
 * The compiler's scheduler would undo the interleaving (it sinks each section's recurrence into one
 * serial run over the block), so the levels of a tick are fenced with sched_barrier.  The pipeline
 * drains at the end of the block: the lane state at block boundaries is the plain sequential
 * state.  Every value is produced by exactly the operations of iir.h:57-74 / rtl_wmbus.c:497-515.
 *
 * Bits: the slicer output (soft >= 0, rtl_wmbus.c:1059) is the inverted sign bit -- a soft symbol
 * is never -0 (the FIR accumulates from +0, and +0 + -0 = +0; the DC remover's x - x_old is never
 * -0 either) -- shifted into a word with one v_alignbit; clock levels via WM_LEVEL_CARRY. */
template <bool DC>
__device__ __forceinline__ void clk_block32(WmClkState &s, const IirCoef &c, const float *xrow, uint32_t &bitw, uint32_t &smask)
{
    /* xrow: this lane's 32 soft symbols in LDS; four are fetched every fourth tick, so the block in
     * flight and the one after it can stay in registers (two blocks of loads outstanding per lane) */
    float4 xq = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int P = DC ? 1 : 0;                          /* pipeline depth before the first biquad */
    float h1[3] = {s.h[0], s.h[2], s.h[4]}, h2[3] = {s.h[1], s.h[3], s.h[5]};
    float dcx = s.dc_x, dcy = s.dc_y;
    float in[3] = {0.0f, 0.0f, 0.0f};                      /* input of section k at the coming tick */
    float soft = 0.0f;                                     /* DC stage output waiting for section 0 */
    uint32_t sgn = 0, low = 0;                             /* MSB-first: sample n ends up in bit 31 - n */
    const float al = 0.999f, kk = wm_div(wm_add(1.0f, al), 2.0f);
#pragma unroll
    for (int t = 0; t < 32 + P + 2; t++) {
        float m1[3], m2[3], p1[3], p2[3], tt[3], h0[3], u[3], o[3];
        float d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
        if (t < 32 && (t & 3) == 0) xq = *(const float4 *)(xrow + t);
        const float xt = (t & 3) == 0 ? xq.x : (t & 3) == 1 ? xq.y : (t & 3) == 2 ? xq.z : xq.w;   /* sample t (t < 32) */
        /* level 1: every product that only needs last tick's state */
        if (DC && t < 32) { d1 = wm_sub(xt, dcx); d2 = wm_mul(al, dcy); }
        {   /* section 0's input: the (DC-filtered) soft symbol, squared */
            const int n0 = t - P;
            if (n0 >= 0 && n0 < 32) {
                const float sf = DC ? soft : xt;
                sgn = __builtin_amdgcn_alignbit(sgn, wm_f2u(sf), 31);      /* (sgn << 1) | signbit */
                in[0] = wm_mul(sf, sf);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int n = t - P - k;
            if (n >= 0 && n < 32) {
                m1[k] = wm_mul(c.a1[k], h1[k]); m2[k] = wm_mul(c.a2[k], h2[k]);
                p1[k] = wm_mul(c.b1[k], h1[k]); p2[k] = wm_mul(c.b2[k], h2[k]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        /* level 2 */
        if (DC && t < 32) d3 = wm_mul(kk, d1);
#pragma unroll
        for (int k = 0; k < 3; k++) { const int n = t - P - k; if (n >= 0 && n < 32) tt[k] = wm_add(m1[k], m2[k]); }
        __builtin_amdgcn_sched_barrier(0);
        /* level 3 */
        if (DC && t < 32) { const float y = wm_add(d3, d2); dcx = xt; dcy = y; soft = y; }
#pragma unroll
        for (int k = 0; k < 3; k++) { const int n = t - P - k; if (n >= 0 && n < 32) h0[k] = wm_sub(in[k], tt[k]); }
        __builtin_amdgcn_sched_barrier(0);
        /* level 4, 5 */
#pragma unroll
        for (int k = 0; k < 3; k++) { const int n = t - P - k; if (n >= 0 && n < 32) u[k] = wm_add(h0[k], p1[k]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 3; k++) { const int n = t - P - k; if (n >= 0 && n < 32) o[k] = wm_add(u[k], p2[k]); }
        /* hand over: section k's output is section k+1's input at the next tick */
#pragma unroll
        for (int k = 2; k >= 0; k--) {
            const int n = t - P - k;
            if (n >= 0 && n < 32) {
                h2[k] = h1[k]; h1[k] = h0[k];
                if (k < 2) in[k + 1] = o[k];
                else {
                    uint32_t tmp;
                    asm("v_add_co_u32 %1, vcc, %3, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                        : "+v"(low), "=&v"(tmp) : "v"(wm_f2u(o[2])), "s"(WM_LEVEL_CARRY) : "vcc");
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    s.h[0] = h1[0]; s.h[1] = h2[0]; s.h[2] = h1[1]; s.h[3] = h2[1]; s.h[4] = h1[2]; s.h[5] = h2[2];
    s.dc_x = dcx; s.dc_y = dcy;
    bitw = ~__builtin_bitreverse32(sgn);
    /* clock lock (rtl_wmbus.c:1092-1111): take the bit at n iff the levels at n-3..n are L,H,H,H */
    /* WmClkState.clk keeps the last three levels with the NEWEST in bit 0; here time runs upwards */
    const uint32_t prev3 = ((s.clk & 1u) << 2) | (s.clk & 2u) | ((s.clk >> 2) & 1u);
    const uint64_t H = ((uint64_t)(~__builtin_bitreverse32(low)) << 3) | prev3;           /* bit n+3 = level at n */
    smask = (uint32_t)((~H) & (H >> 1) & (H >> 2) & (H >> 3));
    const uint32_t last3 = (uint32_t)(H >> 32) & 7u;                                        /* levels at 29, 30, 31 */
    s.clk = ((last3 & 1u) << 2) | (last3 & 2u) | ((last3 >> 2) & 1u);
}

/* Clock-recovery lane.  The reference's lock counter (rtl_wmbus.c:1092-1111: rising edge -> 1,
 * still high -> 2, third high sample -> take the bit) is equivalent to "sample at n iff the clock
 * levels at n-3..n are L,H,H,H" (checked exhaustively over all level sequences);
 * the lane state keeps the last three levels.
 *
 * Memory: a lane walks its own row (stream, chain) of soft symbols, 128 bytes per 32-sample block.
 * When the 64 lanes of the wave are 64 consecutive streams of one (chain, segment) -- n_streams a
 * multiple of 64, first pass -- the wave fetches the 64 rows' blocks COOPERATIVELY: 8 lanes per
 * row read one whole 128-byte line, and the block is transposed through LDS (conflict-free, see
 * WM_CLK_XROW).  Lane-private 16-byte loads of the same data touch 64 lines per instruction and
 * re-fetch each line from L2 several times.  Re-run launches and odd stream counts take the
 * lane-private path. */
template <int W> struct ClkLds {         /* per block: W independent waves */
    float x[W][64 * WM_CLK_XROW];
    uint32_t chip[W][64 * WM_CLK_CROW];
    uint32_t bits[W][64 * WM_CLK_BROW];
};

/* WM_CLK_WPB independent waves per block (no block-wide barrier anywhere): a block's waves land on
 * the CU's four SIMDs, so the framer loads every SIMD of the CUs it is on equally.  A lone
 * long-running wave on ONE SIMD slows every 4-wave K1 block of that CU down to the pace of the K1
 * wave that shares the SIMD with it (measured: two clock launches in flight, one wave per CU, cost
 * K1 60 %). */
template <bool DC, int W>
__device__ __forceinline__ void clock_lanes(const K2Args &a, const uint32_t block, ClkLds<W> &lds)
{
    const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (wv >= (uint32_t)W) return;           /* W < waves of the block: the fused launch (see k2_clock_rla) */
    float *s_x = lds.x[wv];
    uint32_t *s_chip = lds.chip[wv], *s_bits = lds.bits[wv];
    uint32_t lane = (block * W + wv) * 64 + ln;
    const bool rerun = a.list != nullptr;
    const WmPush &g = a.g;
    const bool coop = !rerun && (g.S % 64u) == 0u;         /* wave = 64 consecutive streams, lock step */
    if (lane >= a.n_lanes) return;
    if (rerun) lane = a.list[lane];
    uint32_t ch, stream, seg;
    lane_decode(g, 1, lane, ch, stream, seg);
    if (!(g.flags & (ch ? WM_F_S1 : WM_F_T1C1))) return;

    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint64_t sidx = row * g.nseg_cap[1] + seg;
    const uint32_t mb = seg * g.seg_len[1], me = min(g.M, mb + g.seg_len[1]);
    const uint32_t cap_t2 = g.cap[1];
    WmClkState *stS = (WmClkState *)a.st_start, *stF = (WmClkState *)a.st_final, *stC = (WmClkState *)a.st_carry;

    WmClkState s;
    uint32_t m;
    if (rerun) { s = seg ? stF[sidx - 1] : stC[row]; m = mb; }
    else {
        const uint32_t w = g.warm[ch];
        if (mb <= w) { s = stC[row]; m = 0; }            /* exact: run from the push start  */
        else { s = WmClkState{}; m = mb - w; }           /* speculative cold start          */
    }
    const IirCoef c = iir_coef(ch);
    const bool t2a = g.flags & WM_F_T2A;
    const float *x = a.dphi + row * g.Mcap;
    /* cooperative view: lane ln fetches piece ln%8 of row (8 i + ln/8), i = 0..7; rows of the wave
     * are consecutive */
    const uint64_t row0 = row - ln;
    const float *xc = a.dphi + (row0 + (ln >> 3)) * g.Mcap + 4u * (ln & 7u);
    const uint64_t xc_step = 8ull * g.Mcap;
    const uint32_t syncw = ch ? WM_SYNC_S1 : WM_SYNC_T1C1, syncm = ch ? WM_SYNC_S1_MASK : WM_SYNC_T1C1_MASK;
    uint32_t *out = a.chips + sidx * cap_t2;
    uint32_t *bw = a.bits + row * (g.Mcap / 32);
    uint32_t n_out = 0, saw_sync = 0;

    /* chips of one 32-sample block: walk the set bits of the sample mask (ragged tail, shift
     * register upkeep during warm-up) */
    auto emit_block = [&](uint32_t m0, uint32_t smask, uint32_t bitw, bool emit) {
        while (smask) {
            const uint32_t k = (uint32_t)__ffs((int)smask) - 1u;
            smask &= smask - 1u;
            const uint32_t bit = (bitw >> k) & 1u;
            s.sr = ((s.sr << 1) | bit) & syncm;                       /* rtl_wmbus.c:818-828 */
            if (emit && t2a) {
                const uint32_t val = bit | (s.sr == syncw ? 2u : 0u);
                saw_sync |= val & 2u;
                if (n_out < cap_t2) out[n_out] = WM_CHIP_WORD(m0 + k - mb, val);
                n_out++;
            }
        }
    };

    const uint32_t me_full = mb + ((me - mb) & ~31u);
    /* Two blocks of loads are kept in flight per lane (register sets A and B, used alternately):
     * with one, the kernel ran at the latency of a single 10 KB request per wave (2.8 TB/s). */
    float4 gxA[8], gxB[8];
    auto fetch_x = [&](float4 (&gx)[8], uint32_t mm) {
        if (coop) {
#pragma unroll
            for (int i = 0; i < 8; i++) gx[i] = *(const float4 *)(xc + i * xc_step + mm);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) gx[i] = *(const float4 *)(x + mm + 4 * i);
        }
    };
    /* registers -> LDS rows (coop: the pieces I fetched for other lanes' rows; else my own row) */
    const uint32_t xw = coop ? (ln >> 3) * WM_CLK_XROW + 4u * (ln & 7u) : ln * WM_CLK_XROW;
    const uint32_t xw_step = coop ? 8u * WM_CLK_XROW : 4u;
    const float *xrow = s_x + ln * WM_CLK_XROW;
    auto put_x = [&](const float4 (&gx)[8]) {
        __builtin_amdgcn_wave_barrier();                     /* the previous block's reads are done */
#pragma unroll
        for (int i = 0; i < 8; i++) *(float4 *)(s_x + xw + i * xw_step) = gx[i];
        __builtin_amdgcn_wave_barrier();
    };
    const uint32_t m_last = me_full >= 32u ? me_full - 32u : 0u;      /* clamp for prefetches past the end */

    /* ---- phase 1: warm-up blocks [m, mb): soft symbols only; no store is issued in this loop, so
     * waiting for a block in flight never waits for anything else (gfx950's vmcnt counts loads
     * and stores in one in-order queue) --------------------------------------------------------- */
    auto warm_block = [&](float4 (&gx)[8]) {
        put_x(gx);
        fetch_x(gx, min(m + 64u, m_last));
        uint32_t bitw, smask;
        clk_block32<DC>(s, c, xrow, bitw, smask);
        /* shift-register upkeep, loop-free: at most 8 chips per block, oldest first */
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool has = smask != 0u;
            const uint32_t k = has ? (uint32_t)__ffs((int)smask) - 1u : 0u;
            smask &= smask - 1u;
            const uint32_t sr_new = ((s.sr << 1) | ((bitw >> k) & 1u)) & syncm;
            s.sr = has ? sr_new : s.sr;
        }
        m += 32;
    };
    if (m < me_full) { fetch_x(gxA, m); fetch_x(gxB, min(m + 32u, m_last)); }
    while (m < mb) {
        warm_block(gxA);
        if (m < mb) warm_block(gxB);
        else {                                               /* keep "A = next block" for phase 2 */
#pragma unroll
            for (int i = 0; i < 8; i++) { const float4 t = gxA[i]; gxA[i] = gxB[i]; gxB[i] = t; }
        }
    }
    stS[sidx] = s;                                       /* state the main loop starts from */
    /* ---- phase 2: blocks of the segment proper.  Exactly three stores per block (slicer word and
     * two 16-byte chip stores; a block holds at most 8 chips because the lock pattern L,H,H,H needs 4
     * samples, and slots beyond the block's chips are overwritten by the next block), so the
     * compiler can wait for a prefetched block with a counted vmcnt instead of draining the stores. */
    /* chips leave in whole, 32-byte aligned groups of 8 (see k2_rla: partial-sector stores from
     * 131 072 lanes with private output regions become read-modify-write traffic) */
    uint32_t *my_chip = s_chip + ln * WM_CLK_CROW;
    uint32_t pend = 0, n_fl = 0;
    auto flush8 = [&]() {
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = my_chip[i];
        *(uint4 *)(out + n_fl) = make_uint4(w[0], w[1], w[2], w[3]);
        *(uint4 *)(out + n_fl + 4) = make_uint4(w[4], w[5], w[6], w[7]);
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t v = my_chip[8 + i]; if (8u + i < pend) my_chip[i] = v; }
        n_fl += 8u; pend = pend > 8u ? pend - 8u : 0u;
    };
    /* slicer words leave in aligned groups of 8 as well (one word per 32 samples and lane) */
    uint32_t *my_bits = s_bits + ln * WM_CLK_BROW;
    auto main_block = [&](float4 (&gx)[8]) {
        put_x(gx);
        fetch_x(gx, min(m + 64u, m_last));
        uint32_t bitw, smask;
        clk_block32<DC>(s, c, xrow, bitw, smask);
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool has = smask != 0u;
            const uint32_t k = has ? (uint32_t)__ffs((int)smask) - 1u : 0u;
            smask &= smask - 1u;
            const uint32_t bit = (bitw >> k) & 1u;
            const uint32_t sr_new = ((s.sr << 1) | bit) & syncm;          /* rtl_wmbus.c:818-828 */
            s.sr = has ? sr_new : s.sr;
            const uint32_t val = bit | (sr_new == syncw ? 2u : 0u);
            saw_sync |= has ? (val & 2u) : 0u;
            my_chip[pend + i] = WM_CHIP_WORD(m + k - mb, val);                /* slots beyond the block's chips are rewritten */
            cnt += has;
        }
        pend += t2a ? cnt : 0u;
        const uint32_t bi = m >> 5;
        my_bits[bi & 7u] = bitw;
        if ((bi & 7u) == 7u) {
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = my_bits[i];
            *(uint4 *)(bw + (bi - 7u)) = make_uint4(w[0], w[1], w[2], w[3]);
            *(uint4 *)(bw + (bi - 3u)) = make_uint4(w[4], w[5], w[6], w[7]);
        }
        if (pend >= 8u) flush8();
        m += 32;
    };
    uint32_t *ck = a.ckpt + sidx * (uint64_t)a.nck * 16u;
    for (uint32_t j = 0; m < me_full; j++) {
        const uint32_t stop = min(me_full, m + (uint32_t)WM_CK_SAMPLES);     /* an even number of blocks, or the end */
        while (m < stop) {
            main_block(gxA);
            if (m < stop) main_block(gxB);
        }
        if (m < me_full && j < a.nck) {                  /* interior checkpoint j */
            uint32_t *q = ck + 16u * j;
            const uint32_t *sw = (const uint32_t *)&s;
            if (!rerun) {
                *(uint4 *)(q) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
                *(uint4 *)(q + 4) = make_uint4(sw[4], sw[5], sw[6], sw[7]);
                *(uint4 *)(q + 8) = make_uint4(sw[8], sw[9], sw[10], sw[11]);
                q[12] = n_fl + pend;
            } else {
                bool same = true;
#pragma unroll
                for (int i = 0; i < 12; i++) same &= q[i] == sw[i];
                const uint32_t n1 = n_fl + pend, n0 = q[12];
                if (same && n1 <= n0) {
                    /* Back on the speculative pass's trajectory: everything it produced from here on is
                     * exact already.  My chips replace its first n0; if they are fewer, its tail moves
                     * down.  (More chips
                     * than it had: its tail is partly overwritten -- run on to the segment's end.) */
                    for (uint32_t i = 0; i < pend; i++) out[n_fl + i] = my_chip[i];
                    if (n1 < n0) {
                        const uint32_t total0 = min(a.counts[sidx], cap_t2);
                        for (uint32_t i = n0; i < total0; i++) {
                            const uint32_t w = out[i];
                            out[n1 + (i - n0)] = w;
                        }
                        a.counts[sidx] = n1 + (total0 - n0);
                        /* this and the later checkpoints describe the tail, which has moved: a later
                         * round may re-run this segment again and meet them */
                        for (uint32_t jj = j; jj < a.nck; jj++) ck[16u * jj + 12u] -= n0 - n1;
                    }
                    if (saw_sync) a.sync_seen[sidx] = 1u;       /* the tail's flag, if any, is already set */
                    return;
                }
                /* Not on the recorded trajectory: from here on the region holds MY chips (and all of it
                 * if I run to the end), so the checkpoint must describe me -- a later round that re-runs
                 * this segment once more compares against what is in memory, not against the
                 * speculative pass.  (Found by the randomised tests: two chips lost after a second
                 * round met a checkpoint whose chip count predated the first round's move.) */
                *(uint4 *)(q) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
                *(uint4 *)(q + 4) = make_uint4(sw[4], sw[5], sw[6], sw[7]);
                *(uint4 *)(q + 8) = make_uint4(sw[8], sw[9], sw[10], sw[11]);
                q[12] = n1;
            }
        }
    }
    for (uint32_t bi = (m >> 5) & ~7u; bi < (m >> 5); bi++) bw[bi] = my_bits[bi & 7u];   /* incomplete last group */
    n_out = n_fl + pend;
    if (pend) flush8();                                  /* last group; slots beyond n_out are never read */
    if (m < me) {                                        /* ragged tail of the last segment */
        uint32_t bitw = 0, smask = 0, hist = s.clk;
        for (uint32_t k = 0; m + k < me; k++) {
            float soft;
            const uint32_t high = clk_step(s, c, DC, x[m + k], soft);
            hist = ((hist << 1) | high) & 0xFu;
            bitw |= (uint32_t)(soft >= 0.0f) << k;
            smask |= (uint32_t)(hist == 7u) << k;
        }
        s.clk = hist & 7u;
        bw[m >> 5] = bitw;
        emit_block(m, smask, bitw, true);
    }
    stF[sidx] = s;
    a.counts[sidx] = n_out;
    if (saw_sync) a.sync_seen[sidx] = 1u;
    if (n_out > cap_t2) atomicOr(a.err, WM_ERR_CHIP_OVERFLOW);
}

template <bool DC>
__global__ __launch_bounds__(64 * WM_CLK_WPB) void k2_clock(K2Args a)
{
    __shared__ __attribute__((aligned(16))) ClkLds<WM_CLK_WPB> lds;
    clock_lanes<DC, WM_CLK_WPB>(a, blockIdx.x, lds);
}

/* Deglitch filter for a whole 32-sample block, bit-parallel.  W holds raw slicer bits in time
 * order: bit 5+k = sample k of the block, bits 0..4 = the five samples before it.
 *   T1/C1 (rtl_wmbus.c:126-144,733): level = popcount(last 6 raw bits) >= 3, by a bit-sliced adder;
 *   S1    (rtl_wmbus.c:149-154,644): LUT 0101011101111111 = newest | majority(previous three).
 * Returns bit k = deglitched level at sample k. */
__device__ __forceinline__ uint32_t deglitch_block(uint64_t W, bool s1)
{
    const uint64_t a0 = W, a1 = W << 1, a2 = W << 2, a3 = W << 3;
    uint64_t D;
    if (s1) D = a0 | (a1 & a2) | (a1 & a3) | (a2 & a3);
    else {
        const uint64_t a4 = W << 4, a5 = W << 5;
        const uint64_t x1 = a0 ^ a1, s_1 = x1 ^ a2, c_1 = (a0 & a1) | (a2 & x1);
        const uint64_t x2 = a3 ^ a4, s_2 = x2 ^ a5, c_2 = (a3 & a4) | (a5 & x2);
        D = (c_1 & c_2) | ((c_1 ^ c_2) & (s_1 | s_2));          /* s1+s2+2(c1+c2) >= 3 */
    }
    return (uint32_t)(D >> 5);
}

/* Run-length framer lane (rtl_wmbus.c:640-702 S1, :729-803 T1/C1), edge-driven: the per-sample
 * work (shift, deglitch, compare, count) is done for 32 samples at once with bit operations and
 * the lane only iterates over the EDGES of the deglitched signal.  A framer reset clears the raw
 * history (rtl_wmbus.c:632,723), so after one the remaining levels of the block are recomputed
 * from the masked history.  WmRlaState.raw keeps the last five raw bits in time order. */
struct RlaLds { uint32_t chip[64 * WM_RLA_WPB * WM_RLA_CROW]; };     /* lane-private staging; the block's waves are independent */

__device__ __forceinline__ void rla_lanes(const K2Args &a, const uint32_t block_id, RlaLds &lds)
{
    uint32_t *s_chip = lds.chip;
    uint32_t lane = block_id * (64 * WM_RLA_WPB) + threadIdx.x;
    if (lane >= a.n_lanes) return;
    const bool rerun = a.list != nullptr;
    if (rerun) lane = a.list[lane];
    const WmPush &g = a.g;
    uint32_t ch, stream, seg;
    lane_decode(g, 0, lane, ch, stream, seg);
    if (!(g.flags & (ch ? WM_F_S1 : WM_F_T1C1))) return;

    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint64_t sidx = row * g.nseg_cap[0] + seg;
    const uint32_t mb = seg * g.seg_len[0], me = min(g.M, mb + g.seg_len[0]);
    const uint32_t cap_rl = g.cap[0];
    WmRlaState *stS = (WmRlaState *)a.st_start, *stF = (WmRlaState *)a.st_final, *stC = (WmRlaState *)a.st_carry;
    const WmRlaState reset = {0, 8 * 256, 0, 2u, 0u, 0u, 24, 24};   /* :628-637 / :717-726, reset pending */

    WmRlaState s;
    uint32_t m;
    if (rerun) { s = seg ? stF[sidx - 1] : stC[row]; m = mb; }
    else if (mb <= g.lookback) { s = stC[row]; m = 0; }
    else { s = reset; m = mb - g.lookback; }

    const uint32_t *bw = a.bits + row * (g.Mcap / 32);
    uint32_t *out = a.chips + sidx * cap_rl;
    const bool s1 = ch != 0;
    const uint32_t syncw = s1 ? WM_SYNC_S1 : WM_SYNC_T1C1, syncm = s1 ? WM_SYNC_S1_MASK : WM_SYNC_T1C1_MASK;
    const uint32_t hist_mask = s1 ? 0x1Cu : 0x1Fu;       /* S1 looks back 3 samples, T1/C1 5      */

    /* Chips are staged in LDS (16 words per lane) and leave in whole, 32-byte aligned groups of 8:
     * every lane appends to its own region of HBM, so with half a million lanes in flight the
     * partially written lines do not stay in L2; 4-byte stores (or unaligned 16-byte ones) turn
     * into read-modify-write traffic at the memory side and cost 3 of the kernel's 8.4 ms. */
    uint32_t *my_chip = s_chip + threadIdx.x * WM_RLA_CROW;
    uint32_t pend = 0, n_fl = 0, saw_sync = 0;               /* staged chips; chips already in HBM (multiple of 8) */
    auto flush8 = [&]() {                                    /* the oldest 8 staged words -> HBM */
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = my_chip[i];
        if (n_fl + 8u <= cap_rl) {
            *(uint4 *)(out + n_fl) = make_uint4(w[0], w[1], w[2], w[3]);
            *(uint4 *)(out + n_fl + 4) = make_uint4(w[4], w[5], w[6], w[7]);
        }
        for (uint32_t i = 8; i < pend; i++) my_chip[i - 8] = my_chip[i];
        n_fl += 8u; pend = pend > 8u ? pend - 8u : 0u;
    };

    /* slicer words arrive 8 at a time (one aligned 32-byte sector per lane and 256 samples; single
     * words cost a sector of HBM traffic each); the next group is in flight while this one is used */
    uint32_t grp = m >> 8;                                   /* group (256 samples) the lane is in */
    uint4 wq0 = *(const uint4 *)(bw + 8u * grp), wq1 = *(const uint4 *)(bw + 8u * grp + 4), nq0 = {}, nq1 = {};
    auto fetch_group = [&](uint32_t gq) {                    /* rows hold whole groups (Mcap is a multiple of 256) */
        if (gq * 256u < g.Mcap) { nq0 = *(const uint4 *)(bw + 8u * gq); nq1 = *(const uint4 *)(bw + 8u * gq + 4); }
    };
    fetch_group(grp + 1u);
    auto block = [&](const bool emit) {
        const uint32_t sub = (m >> 5) & 7u;
        const uint32_t wsel[8] = {wq0.x, wq0.y, wq0.z, wq0.w, wq1.x, wq1.y, wq1.z, wq1.w};
        uint32_t word = wsel[0];
#pragma unroll
        for (int i = 1; i < 8; i++) word = sub == (uint32_t)i ? wsel[i] : word;
        const uint32_t kend = min(32u, me - m);
        const uint32_t valid = kend == 32u ? 0xFFFFFFFFu : ((1u << kend) - 1u);
        uint64_t W = ((uint64_t)(word & valid) << 5) | (s.raw & hist_mask);
        uint32_t D = deglitch_block(W, s1);
        uint32_t k0 = 0;
        while (k0 < kend) {
            const uint32_t level = s.state & 1u;
            const uint32_t x = (level ? ~D : D) & valid & (0xFFFFFFFFu << k0);
            if (!x) { s.run += (int)(kend - k0); break; }
            const uint32_t k = (uint32_t)__ffs((int)x) - 1u;          /* first sample whose level differs */
            s.run += (int)(k - k0);
            int unit = 0, half = 0;
            const int run0 = s.run;
            bool rst;
            if (!s1) {
                rst = s.run < 5;                                                             /* :742 */
                if (!rst) { s.run *= 256; unit = s.bitlen; half = unit / 2; rst = s.run <= half; }   /* :752-756 */
            } else {
                unit = (s.spb0 + s.spb1) / 2;
                rst = unit <= 12 || unit >= 36;                                              /* :659 */
                if (!rst) { half = unit / 2; rst = run0 <= half; }                           /* :671 */
            }
            if (rst) {
                s = reset;
                W &= ~((2ull << (5u + k)) - 1ull);       /* raw history cleared, incl. sample k */
                D = deglitch_block(W, s1);
            } else {
                int n = 0;
                while (s.run > half && n < (int)WM_RLA_RUN_LIMIT) {                          /* :765-779 / :680-694 */
                    s.run -= unit;
                    s.sr = ((s.sr << 1) | level) & syncm;
                    if (emit) {
                        const uint32_t val = level | (s.sr == syncw ? 2u : 0u) | ((s.state & 2u) ? 4u : 0u);
                        saw_sync |= val & 2u;
                        my_chip[pend] = WM_CHIP_WORD(m + k - mb, val);
                        if (++pend == 16u) flush8();         /* a long run can emit many chips at one edge */
                    }
                    s.state &= ~2u;                        /* reset marker travels with the first chip */
                    n++;
                }
                if (s.run > half) {
                    /* A run of more than WM_RLA_RUN_LIMIT chips (exact silence, then an edge): a packet
                     * decoder consumes at most 16*290 chips after an access code, and identical chips
                     * cannot complete one, so the rest of the run need not be materialised -- only
                     * counted, as the reference's loop would. */
                    const int k = (s.run - half + unit - 1) / unit;
                    s.run -= k * unit; n += k;
                    s.sr = level ? syncm : 0u;
                }
                if (!s1) {
                    s.cum += s.run;
                    s.bitlen += wm_sdiv(s.run + s.cum / 16, 32 * n);                         /* :792-796 */
                } else {
                    const int v = wm_sdiv(run0, n);                                          /* :698 */
                    if (level) s.spb1 = v; else s.spb0 = v;
                }
            }
            s.state = (s.state & 2u) | (level ^ 1u);
            s.run = 1;
            k0 = k + 1u;
        }
        s.raw = (uint32_t)(W >> kend) & hist_mask;        /* the five newest raw bits, time order */
        if (sub == 7u) { wq0 = nq0; wq1 = nq1; grp++; fetch_group(grp + 1u); }
        if (emit && pend >= 8u) flush8();
        m += 32;
    };
    while (m < mb) block(false);                         /* speculative look-back: no stores at all */
    stS[sidx] = s;
    while (m < me) block(true);
    const uint32_t n_out = n_fl + pend;
    while (pend) flush8();                               /* last group: the slots beyond n_out are never read */
    stF[sidx] = s;
    a.counts[sidx] = n_out;
    if (saw_sync) a.sync_seen[sidx] = 1u;
    if (n_out > cap_rl) atomicOr(a.err, WM_ERR_CHIP_OVERFLOW);
}

__global__ __launch_bounds__(64 * WM_RLA_WPB) void k2_rla(K2Args a)
{
    __shared__ RlaLds lds;
    rla_lanes(a, blockIdx.x, lds);
}

/* One launch for two independent pieces of work: the clock kernel's re-run lanes (few, long) and
 * the run-length framer (its main pass or its own re-run list).  Without the DC remover the slicer
 * words are final after the clock kernel's FIRST pass (sign of the soft symbol, no state), so the
 * run-length framer need not wait for the clock re-runs; sharing a launch keeps both on the
 * context's one stream (more streams than hardware queues serialise against each other).
 * Every block of a launch gets the same LDS allocation, and a block that needs a quarter of a CU's
 * LDS cannot be placed while K1 refills the CU with its small blocks (and, once placed, costs K1
 * three of its eight blocks): the few clock blocks of this launch therefore run ONE wave each, in
 * the footprint of a run-length block (17 KB instead of 63 KB). */
__global__ __launch_bounds__(64 * WM_RLA_WPB) void k2_clock_rla(K2Args clk, K2Args rla, uint32_t clk_blocks)
{
    __shared__ __attribute__((aligned(16))) union { ClkLds<1> c; RlaLds r; } lds;
    if (blockIdx.x < clk_blocks) clock_lanes<false, 1>(clk, blockIdx.x, lds.c);
    else rla_lanes(rla, blockIdx.x - clk_blocks, lds.r);
}

/* start[seg] must equal final[seg-1]; mismatching lanes are appended to `list`. */
__global__ void k2_verify(WmPush g, uint32_t algo, const uint32_t *st_start, const uint32_t *st_final, uint32_t words,
                          uint32_t *list, uint32_t *n_list)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= 2u * g.nseg[algo] * g.S) return;
    uint32_t ch, stream, seg;
    lane_decode(g, algo, lane, ch, stream, seg);
    if (!(g.flags & (ch ? WM_F_S1 : WM_F_T1C1)) || seg == 0) return;
    const uint64_t sidx = ((uint64_t)ch * g.S + stream) * g.nseg_cap[algo] + seg;
    const uint32_t *p = st_start + sidx * words, *q = st_final + (sidx - 1) * words;
    bool same = true;
    for (uint32_t k = 0; k < words; k++) same &= p[k] == q[k];
    if (!same) list[atomicAdd(n_list, 1u)] = lane;
}

/* =============================================================================================
 * K3: access-code hits -> bursts for the host packet decoders
 * ===========================================================================================*/
struct K3Args {
    WmPush g;
    const uint8_t *rssi;         /* [2][S][Mcap]: the framers no longer copy the RSSI byte into every chip */
    const uint32_t *chips[2];    /* per algo: [2][S][nseg_cap][cap]                          */
    const uint32_t *counts[2];   /* per algo: [2][S][nseg_cap]                               */
    const uint2 *hits; const uint32_t *n_hits; uint32_t hits_cap;
    const uint32_t *pending;     /* [2 algo][2 chain][S]: chips still owed to a busy decoder  */
    WmBurstHdr *hdr; uint32_t hdr_cap;
    uint32_t *words; uint32_t words_cap;
    uint32_t *n_hdr, *n_words;
    uint32_t *err;
};

__device__ static const uint8_t D3OF6[64] = {
    255,255,255,255,255,255,255,255,255,255,255,3,255,1,2,255,255,255,255,7,255,255,0,255,255,5,6,255,4,255,255,255,
    255,255,255,11,255,9,10,255,255,15,255,255,8,255,255,255,255,13,14,255,12,255,255,255,255,255,255,255,255,255,255,255};

__device__ __forceinline__ uint32_t full_len_a(uint32_t L) { return 1u + L + 2u * (1u + (L > 9u ? (L - 9u + 15u) / 16u : 0u)); }

/* Chips after the access-code chip that a decoder consumes before it returns to idle, ignoring
 * RSSI aborts and framer resets (those only shorten it).  hb = the next 24 chips, first chip in
 * bit 23; nb = how many of them exist.  Mirrors the length logic of
 * t1_c1_packet_decoder.h:298-349,399-438 and s1_packet_decoder.h:152-197. */
__device__ uint32_t burst_need(uint32_t chain, uint32_t hb, uint32_t nb)
{
    if (chain == 0) {
        if (nb < 12) return WM_MAXCHIPS_T1C1;
        const uint32_t hi = D3OF6[(hb >> 18) & 63u], lo = D3OF6[(hb >> 12) & 63u];
        if (hi != 255u && lo != 255u) return 12u * full_len_a((hi << 4) | lo);
        const uint32_t mode = hb >> 12;
        if (mode != 0x54Cu && mode != 0x543u) return 12u;
        if (nb < 24) return WM_MAXCHIPS_T1C1;
        if (((hb >> 8) & 15u) != 0xDu) return 16u;
        const uint32_t L = hb & 255u;
        return 24u + 8u * ((mode == 0x543u ? 1u + L : full_len_a(L)) - 1u);
    }
    if (nb < 16) return WM_MAXCHIPS_S1;
    uint32_t L = 0;
    for (int j = 0; j < 8; j++) {
        const uint32_t pair = (hb >> (22 - 2 * j)) & 3u;
        if (pair == 0u || pair == 3u) return 2u * (uint32_t)j + 2u;
        L = (L << 1) | (pair == 1u ? 1u : 0u);
    }
    return 16u * full_len_a(L);
}

/* Access-code hits = chips with the sync flag, collected AFTER both framers have settled (re-runs
 * included).  (The framer kernels used to append hits as they went; every re-run then left stale
 * and duplicate records behind, each of which cost a burst copy.)  The framers only leave a
 * per-region flag "an access-code chip was emitted here by some pass"; each lane of a wave looks at
 * the flag of one (framer, chain, capture, segment) region, and the wave then scans the flagged
 * regions (a minority) together, appending {lane | algo << 31, chip index}. */
__global__ __launch_bounds__(256) void k3_scan(WmPush g, const uint32_t *chips0, const uint32_t *chips1, const uint32_t *counts0,
                                               const uint32_t *counts1, const uint32_t *seen0, const uint32_t *seen1,
                                               uint2 *hits, uint32_t *n_hits, uint32_t hits_cap, uint32_t *err)
{
    const uint32_t n0 = 2u * g.nseg[0] * g.S;                /* run-length lanes first */
    const uint32_t ln = threadIdx.x & 63u;
    uint32_t lane = blockIdx.x * 256u + threadIdx.x, algo = 0;
    if (lane >= n0) { lane -= n0; algo = 1; }
    uint32_t my_cnt = 0, my_sidx = 0;
    if (lane < 2u * g.nseg[algo] * g.S) {
        uint32_t ch, stream, seg;
        lane_decode(g, algo, lane, ch, stream, seg);
        const uint32_t sidx = (ch * g.S + stream) * g.nseg_cap[algo] + seg;
        if ((g.flags & (ch ? WM_F_S1 : WM_F_T1C1)) && (g.flags & (algo ? WM_F_T2A : WM_F_RLA)) && (algo ? seen1 : seen0)[sidx]) {
            my_sidx = sidx;
            my_cnt = min((algo ? counts1 : counts0)[sidx], g.cap[algo]);
        }
    }
    for (uint64_t todo = __ballot(my_cnt != 0u); todo; todo &= todo - 1ull) {
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t cnt = __shfl(my_cnt, src), sidx = __shfl(my_sidx, src), r_lane = __shfl(lane, src), r_algo = __shfl(algo, src);
        const uint32_t *w = (r_algo ? chips1 : chips0) + (uint64_t)sidx * g.cap[r_algo];
        for (uint32_t k4 = 4u * ln; k4 < cnt; k4 += 256u) {             /* regions are 32-byte aligned, cap % 8 == 0 */
            const uint4 v = *(const uint4 *)(w + k4);
            const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (k4 + j < cnt && (q[j] & 2u)) {
                    const uint32_t i = atomicAdd(n_hits, 1u);
                    if (i < hits_cap) hits[i] = make_uint2(r_lane | (r_algo << 31), k4 + j);
                    else atomicOr(err, WM_ERR_BURST_OVERFLOW);
                }
        }
    }
}

/* One access-code hit (or pending continuation), handled by one wave. */
__device__ void burst_item(const K3Args &a, const uint32_t item, const uint32_t ln)
{
    const WmPush &g = a.g;
    const uint32_t n_hits = min(*a.n_hits, a.hits_cap);
    uint32_t algo, ch, stream, seg, k, cont = 0, want;
    if (item < 4u * g.S) {                       /* continuation slots come first            */
        algo = item / (2u * g.S); ch = (item / g.S) & 1u; stream = item % g.S;
        want = a.pending[item];
        if (want == 0u) return;
        seg = 0; k = 0; cont = 1;
    } else {
        if (item - 4u * g.S >= n_hits) return;
        const uint2 h = a.hits[item - 4u * g.S];
        algo = h.x >> 31;
        lane_decode(g, algo, h.x & 0x7FFFFFFFu, ch, stream, seg);
        k = h.y; want = 0;
    }
    const uint32_t cap = g.cap[algo], nseg = g.nseg[algo], seg_len = g.seg_len[algo];
    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint32_t *cnt = a.counts[algo] + row * g.nseg_cap[algo];
    const uint32_t *base = a.chips[algo] + row * g.nseg_cap[algo] * (uint64_t)cap;
    if (!cont) {                                 /* stale record of a re-run segment?         */
        if (k >= min(cnt[seg], cap) || !(base[(uint64_t)seg * cap + k] & 2u)) return;
    }
    /* chips before / from the hit in this push's chip stream: the wave sums the segment counts in
     * parallel (a serial scan of up to 256 dependent loads per wave was most of this kernel's time) */
    uint32_t before = 0, total = 0;
    for (uint32_t s = ln; s < nseg; s += 64u) { const uint32_t c = min(cnt[s], cap); if (s < seg) before += c; total += c; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { before += __shfl_xor(before, off); total += __shfl_xor(total, off); }
    const uint32_t chip0 = before + k;
    if (chip0 >= total) return;
    const uint32_t avail = total - chip0;        /* chips from the hit to the end of the push */

    auto locate = [&](uint32_t j, uint32_t &sg, uint32_t &kk) {   /* chip0 + j -> (segment, index) */
        sg = seg; kk = k + j;
        while (sg < nseg) { const uint32_t c = min(cnt[sg], cap); if (kk < c) break; kk -= c; sg++; }
    };

    uint32_t n;
    if (cont) n = min(want, avail);
    else {
        uint32_t bit = 0;
        if (ln < 24u && 1u + ln < avail) { uint32_t sg, kk; locate(1u + ln, sg, kk); bit = base[(uint64_t)sg * cap + kk] & 1u; }
        const unsigned long long m = __ballot(bit);
        uint32_t hb = 0;
        for (int j = 0; j < 24; j++) hb |= (uint32_t)((m >> j) & 1ull) << (23 - j);
        n = min(burst_need(ch, hb, min(24u, avail - 1u)) + 1u, avail);
    }
    uint32_t hslot = 0, woff = 0;
    if (ln == 0) { hslot = atomicAdd(a.n_hdr, 1u); woff = atomicAdd(a.n_words, n); }
    hslot = __shfl(hslot, 0); woff = __shfl(woff, 0);
    if (hslot >= a.hdr_cap || woff + n > a.words_cap) { if (ln == 0) atomicOr(a.err, WM_ERR_BURST_OVERFLOW); return; }
    uint32_t sg0, k0; locate(0, sg0, k0);
    const uint64_t pos0 = g.m0 + (uint64_t)sg0 * seg_len + WM_CHIP_POS(base[(uint64_t)sg0 * cap + k0]);
    for (uint32_t j = ln; j < n; j += 64u) {
        uint32_t sg, kk; locate(j, sg, kk);
        const uint32_t w = base[(uint64_t)sg * cap + kk];
        const uint32_t pm = sg * seg_len + WM_CHIP_POS(w);               /* push-relative decimated sample */
        const uint64_t pos = g.m0 + pm;
        const uint32_t rssi = a.rssi[row * g.Mcap + pm];                 /* (unsigned)EMA at the chip's sample */
        a.words[woff + j] = ((uint32_t)(pos - pos0) << 11) | (rssi << 3) | (WM_CHIP_VAL(w) & 7u);
    }
    if (ln == 0) {
        WmBurstHdr h;
        h.stream = stream; h.chain = (uint8_t)ch; h.algo = (uint8_t)algo; h.flags = (uint16_t)cont;
        h.chip0 = chip0; h.n_chips = n; h.pos0 = pos0; h.word_off = woff; h.avail = avail;
        a.hdr[hslot] = h;
    }
}

/* A bounded number of waves walks the items (continuation slots, then hits).  One wave per item
 * -- 14 000 single-wave blocks per 128 captures, each a chain of dependent loads -- took every wave
 * slot of the chip for the kernel's duration and stalled the demodulation kernel of the next
 * context (measured: K1 ran at a quarter of its speed while this kernel was resident). */
__global__ __launch_bounds__(256) void k3_bursts(K3Args a, uint32_t n_items)
{
    const uint32_t ln = threadIdx.x & 63u;
    for (uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6); item < n_items; item += gridDim.x * 4u) burst_item(a, item, ln);
}

/* Debug/parity helper: flatten one (chain, algo, stream) chip stream. */
__global__ void k4_flatten(WmPush g, uint32_t algo, const uint32_t *chips, const uint32_t *counts, const uint8_t *rssi, uint32_t cap,
                           uint32_t ch, uint32_t stream, uint32_t *dst, uint64_t *pos, uint32_t max_out, uint32_t *n_out)
{
    if (blockIdx.x || threadIdx.x) return;
    const uint64_t row = (uint64_t)ch * g.S + stream;
    uint32_t n = 0;
    for (uint32_t s = 0; s < g.nseg[algo]; s++) {
        const uint32_t c = min(counts[row * g.nseg_cap[algo] + s], cap);
        for (uint32_t k = 0; k < c; k++, n++)
            if (n < max_out) {
                const uint32_t w = chips[(row * g.nseg_cap[algo] + s) * (uint64_t)cap + k];
                dst[n] = WM_CHIP_VAL(w) | ((uint32_t)rssi[row * g.Mcap + s * g.seg_len[algo] + WM_CHIP_POS(w)] << 8);
                if (pos) pos[n] = g.m0 + (uint64_t)s * g.seg_len[algo] + WM_CHIP_POS(w);
            }
    }
    *n_out = n;
}
