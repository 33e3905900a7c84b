/* wm_k1_demod.h -- K1: demodulation tile kernels (front end, discriminator, FIR, RSSI) and their hand-off checks.
 * Device code, included by wm_kernels.hip (one translation unit, see the overview there). */
#ifndef WM_K1_DEMOD_H
#define WM_K1_DEMOD_H

typedef short wm_s2 __attribute__((ext_vector_type(2)));
#if defined(__HIPCC__)
#define WM_LDS_STATIC __shared__
#else
#define WM_LDS_STATIC static              /* host emulation: one block at a time */
#endif

/* -DWM_K1_STAMPS (tools/gpu_k1_stamps.py, never in the product build): every wave of the first pass without the RSSI reads
 * the shader clock (s_memtime) at its stage boundaries and leaves the five intervals -- stage 0 (input loads + conversion), the
 * wait at the first barrier, stage A, the wait at the second barrier, stage B -- in a slot of its own (the wave's number in the
 * launch, modulo the buffer: a first version added everything into seven words, and 30 million atomics on seven addresses
 * made the kernel forty times slower).  [5] counts the waves that met in a slot, [6] is the whole tile as its first wave saw
 * it.  profiles/r05_k1_stage_cycles.txt is made from it. */
#if defined(WM_K1_STAMPS) && defined(__HIPCC__)
#define WM_K1_STAMP_SLOTS (1u << 21)
__device__ unsigned int wm_k1_stamp_buf[WM_K1_STAMP_SLOTS * 8u];
#define WM_K1_STAMP_DECL unsigned long long k1_t[6] = {0, 0, 0, 0, 0, 0}
#define WM_K1_STAMP(i) do { if (RS == 1) k1_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define WM_K1_STAMP_END(tid) do { if (RS == 1 && ((tid) & 63) == 0) { \
        const unsigned wv_ = ((blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + ((tid) >> 6)) & (WM_K1_STAMP_SLOTS - 1u); \
        unsigned int *q_ = wm_k1_stamp_buf + 8u * wv_; \
        for (int i_ = 0; i_ < 5; i_++) atomicAdd(q_ + i_, (unsigned int)(k1_t[i_ + 1] - k1_t[i_])); \
        atomicAdd(q_ + 5, 1u); if ((tid) == 0) atomicAdd(q_ + 6, (unsigned int)(k1_t[5] - k1_t[0])); } } while (0)
#else
#define WM_K1_STAMP_DECL
#define WM_K1_STAMP(i)
#define WM_K1_STAMP_END(tid)
#endif

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
    const uint32_t *n_relist;/* repair launches: entries in `relist` (on the device: k1_collect has just written it) */
    uint32_t tile0;          /* first pass: the launch's first tile (a push's tiles may leave in two launches, see enqueue_front_impl) */
    /* RSSI on demand (RS = 2 launches; `relist` / `n_relist` then name the tiles k3_spans has listed) */
    const uint32_t *rs_flags;/* [ntiles][S] bit 0: the T1/C1 chain's RSSI is read in this tile, bit 1: the S1 chain's */
    float *ema_out;          /* [2][S] the EMA after the push's last sample (the last tile of every capture is always listed) */
    uint32_t *rs_fail;       /* set when a lane that is read could not prove its value: the push falls back to the full pass */
    /* first-pass launches whose blocks take SEVERAL consecutive tiles (round 5): block x of the launch takes tiles tile0 + x tpb ...
     * (below tile_end), and the input of a block's next tile is already on its way while the current one is computed */
    uint32_t tpb, tile_end;  /* tpb <= 1: one tile per block, tile_end unused */
};

/* RSSI on demand.  The RSSI of a sample (rtl_wmbus.c:475-495: EMA of the filtered magnitude) is READ only at the chips of
 * bursts a packet decoder is looking at (t1_c1_packet_decoder.h:651,707; s1_packet_decoder.h:235,277) -- a sixth of the
 * samples of the bench workload -- while computing it for every sample was 14 % of the demodulation kernel (eight exact
 * square roots per thread, the EMA roles of stage B2).  With RS = 1 the first pass leaves it out; when the framers and
 * k3_spans have found the bursts, an RS = 2 launch computes it for the 976-sample tiles they touch.
 * The EMA is recursive, so a lane must PROVE the state it starts from without the chain of hand-offs the full pass
 * certifies against (k1_verify): it runs its 32-sample warm-up twice, from 0 and from WM_EMA_UPPER.  The filter step
 * y -> fl(fl(al x) + fl(be y)) is monotone in y and the true state lies in [0, WM_EMA_UPPER) (magnitudes are below 256:
 * |i|, |q| <= 127.5 sqrt 2 after the -s rotation), so both trajectories bracket the true one at every sample, and where they
 * meet bit for bit the true one is there too.  A lane whose bracket is still open takes its predecessor's end value if
 * that one is proven and equals its own lower trajectory (the full pass's certificate).  If a lane that is read has neither
 * (constant input can keep a bracket one ulp open for ever: the map has neighbouring fixed points), rs_fail is set and the
 * host runs the full pass for this push -- slow, exact, rare. */
#define WM_EMA_UPPER 256.0f

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
/* NT = threads per block: a tile is 4 NT - 48 decimated samples (976 on 256 threads; 2000 on 512, round 5: the halo --
 * 48 samples of discriminators computed again by the next tile -- is 2.4 % of the tile instead of 4.9 %, and the blocks
 * of a launch are half as many).  Only the first pass without the RSSI (RS = 1) knows the larger tile: it writes nothing
 * per tile, so the tile size of the RSSI launches, the repair walk and k3_spans stays WM_K1_TILE2. */
template <int NT> struct K1GeoT {
    static constexpr int T = 4 * NT - WM_K1_HALO, NA = T + WM_K1_HALO;
    static constexpr int YD = NA + 8, YM = NA + NA / 16 + 4;
    __host__ __device__ static constexpr int nstg(int d) { return (NA * d + 16 + 8 + 7) / 8 * 8 + 8; }   /* 8 slack words in front */
    __host__ __device__ static constexpr int U(int d, bool shift)
    {
        return nstg(d) * (shift ? 2 : 1) > 2 * YM ? nstg(d) * (shift ? 2 : 1) : 2 * YM;
    }
    static constexpr size_t smem(int d, bool shift) { return (size_t)U(d, shift) * 4; }     /* the DYNAMIC part: staging / magnitude rows */
};
/* The rest of a block's LDS is static: discriminator rows, the EMA hand-off scratch, the arctangent's table.  A static array has
 * its address at compile time -- the address of the dynamic (extern) one is only resolved after instruction selection, and the
 * byte fetch of the range LUT, eight per thread, carried an add of that "unknown" base (round 5, read off the ISA).  One struct,
 * so that the table sits BEHIND the rows: the LUT index arrives biased by bits(7/16) >> 18 = 4024, and only a non-negative
 * remainder fits the instruction's offset field. */
template <int NT> struct K1LdsT { float yDr[2 * K1GeoT<NT>::YD]; float sFin[128], sHead[128]; float tab[WM_ATAN_TAB_WORDS]; };
static_assert(sizeof(K1LdsT<256>) - sizeof(float) * WM_ATAN_TAB_WORDS >= 4096, "the LUT's offset must absorb the index bias");
using K1Geo = K1GeoT<256>;
#define WM_K1_TILE_BIG (4 * 512 - WM_K1_HALO)      /* the tile of the 512-thread first pass */
static_assert(K1Geo::T == WM_K1_TILE2, "the 256-thread tile is the tile of every per-tile record");
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

/* The first tap of a low-pass: the reference's sum starts from +0 (fir.h:56), so its first partial sum is 0 + b[0] x -- never
 * -0 -- and neither is any soft symbol (+0 + -0 = +0; an exact cancellation is +0), which the clock kernel relies on when it
 * takes the slicer's decision from the sign bit.  Written as an addition the compiler loses this: b[0] < 0, so 0 + b[0] x
 * becomes 0 - |b[0]| x, and the backend folds "0 - y" into a negated operand of the next subtraction: -(|b[0]| x) - |b[1]| x',
 * which is -0 where the reference has +0 when both samples are +0 (round 5, read off the ISA; reaching the filter's output
 * would take a crafted input, -0 under every positive tap and +0 under every negative one).  fma(b[0], x, +0) is the same
 * single rounding of the product with the +0 added, in one instruction like the multiplication it replaces. */
template <bool FAST>
__device__ __forceinline__ float k1_fir_first(const float b0, const float x)
{
#ifdef WM_FIR_FIRST_AS_ADD                                    /* the form before (A/B and the self-test's evidence: tools/build_variant.sh firadd -DWM_FIR_FIRST_AS_ADD) */
    return FAST ? __builtin_fmaf(b0, x, 0.0f) : wm_add(0.0f, wm_mul(b0, x));
#else
    return FAST ? __builtin_fmaf(b0, x, 0.0f) : wm_fma_exact(b0, x, 0.0f);
#endif
}

/* A 16-byte LDS read that stays ONE ds_read_b128: where a window's first or last vector is only partly used the compiler
 * narrows the load to the used words and then re-pairs the scalars from the odd start -- the 46-tap window arrived as 24
 * ds_read2_b32, each with its own address add (round 5, read off the ISA).  The empty asm makes all four words "used". */
typedef float wm_k1v4 __attribute__((vector_size(16)));
__device__ __forceinline__ wm_k1v4 k1_lds_v4(const float *p)
{
    wm_k1v4 v = *(const wm_k1v4 *)p;
#ifdef __HIP_DEVICE_COMPILE__
    asm("" : "+v"(v));
#endif
    return v;
}

/* Stages B1 (FIR) and B2 (RSSI EMA + hand-off certification) of a 976-sample tile; shared by the
 * moving-average and the polyphase front ends.  Rows: element a of a discriminator row at word
 * a + 4, of a magnitude row at a + a/16 (the two chains' rows may alias when they carry the same
 * data).  Ends with the magnitude rows' barrier already passed by every thread. */
/* FAST (wmbus_cfg.tolerance_mode, never the default): one fused multiply-add per tap instead of the reference's separately
 * rounded product and sum -- the soft symbol then differs from the reference's by rounding noise (DESIGN_HISTORY.md section 12). */
template <bool FAST = false>
__device__ __forceinline__ void k1_fir_t(const K1Args &a, const float *yDrT, const int slot, const int stream, const int ts, const int tn)
{
    const WmPush &g = a.g;
    const int m0l = 4 * slot;
    if (m0l >= tn) return;
    float w[16];                                              /* w[i] = element 4 slot + 36 + i */
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const wm_k1v4 v = k1_lds_v4(yDrT + 4 * slot + 40 + 4 * k);
        w[4 * k] = v[0]; w[4 * k + 1] = v[1]; w[4 * k + 2] = v[2]; w[4 * k + 3] = v[3];
    }
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float s = k1_fir_first<FAST>(FIR_T[0], w[12 + j]);
#pragma unroll
        for (int k = 1; k < 11; k++) s = FAST ? __builtin_fmaf(FIR_T[k], w[12 + j - k], s) : wm_add(s, wm_mul(FIR_T[k], w[12 + j - k]));
        acc[j] = s;
    }
    *(float4 *)(a.dphi + (uint64_t)stream * g.Mcap + (uint64_t)ts + m0l) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

template <bool FAST = false>
__device__ __forceinline__ void k1_fir_s(const K1Args &a, const float *yDrS, const int slot, const int stream, const int ts, const int tn)
{
    const WmPush &g = a.g;
    const int m0l = 4 * slot;
    if (m0l >= tn) return;
    /* y[n] = sum_k b[k] x[n - k], k ascending (fir.h:58-67), for the four outputs n = 4 slot + j; element i of the thread's
     * 52-sample window is x[4 slot - 48 + i], so tap k of output j reads element 48 + j - k.  The window passes through the
     * registers in TWO halves, newest first: taps 0 .. 23 read elements 25 .. 51, taps 24 .. 45 elements 3 .. 27 (seven
     * vectors each) -- 28 registers at a time instead of 52.  (Round 5: the registers are what a block's NEXT tile
     * needs for its input words, which are in flight during this stage; with the whole window resident the kernel spilled.) */
    float acc[4];
    {
        float w[28];                                          /* w[i] = element 24 + i */
#pragma unroll
        for (int k = 6; k >= 0; k--) {
            const wm_k1v4 v = k1_lds_v4(yDrS + 4 * slot + 4 + 24 + 4 * k);
            w[4 * k] = v[0]; w[4 * k + 1] = v[1]; w[4 * k + 2] = v[2]; w[4 * k + 3] = v[3];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s_ = k1_fir_first<FAST>(FIR_S[0], w[24 + j]);
#pragma unroll
            for (int k = 1; k < 24; k++) s_ = FAST ? __builtin_fmaf(FIR_S[k], w[24 + j - k], s_) : wm_add(s_, wm_mul(FIR_S[k], w[24 + j - k]));
            acc[j] = s_;
        }
    }
    {
        float w[28];                                          /* w[i] = element i */
#pragma unroll
        for (int k = 6; k >= 0; k--) {
            const wm_k1v4 v = k1_lds_v4(yDrS + 4 * slot + 4 + 4 * k);
            w[4 * k] = v[0]; w[4 * k + 1] = v[1]; w[4 * k + 2] = v[2]; w[4 * k + 3] = v[3];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s_ = acc[j];
#pragma unroll
            for (int k = 24; k < 46; k++) s_ = FAST ? __builtin_fmaf(FIR_S[k], w[48 + j - k], s_) : wm_add(s_, wm_mul(FIR_S[k], w[48 + j - k]));
            acc[j] = s_;
        }
    }
    *(float4 *)(a.dphi + ((uint64_t)g.S + stream) * g.Mcap + (uint64_t)ts + m0l) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

/* Stages B1 (FIR low-pass, y[n] = sum_k b[k] x[n-k], k ascending, fir.h:48-72) and B2 (RSSI EMA,
 * rtl_wmbus.c:475-495, + hand-off certification) of a 976-sample tile; shared by the moving-average
 * and the polyphase front ends.  Rows: element a of a discriminator row at word a + 4, of a
 * magnitude row at a + a/16 (the two chains' rows may alias when they carry the same data).
 * Work split (WM_K1_BALANCED, the default): every wave runs ONE quarter of the 46-tap FIR (the largest item: 4 x 92
 * operations per lane); waves 0 and 1 add one chain's EMA each (16 samples per lane behind the warm-up), waves 2 and 3
 * half of the 11-tap FIR each -- about 570 against 545 instructions.  (Second generation: EMA + half of the 11-tap
 * FIR against half of the 46-tap FIR, 380 against 740: the light waves sat at the barrier for a quarter of the
 * tile's time, holding their wave slots; first generation 960 against 530.) */
#ifndef WM_K1_BALANCED
#define WM_K1_BALANCED 1
#endif
template <bool GEN, bool FAST = false, int RS = 0, int NT = 256>
__device__ __forceinline__ void k1_stage_b(const K1Args &a, const int tid, const int tile, const int stream, const int ts, const int tn,
                                           const bool chT, const bool chS, const float *yDrT, const float *yDrS,
                                           const float *yMgT, const float *yMgS, float *sFin, float *sHead)
{
    constexpr int T = K1GeoT<NT>::T;
    static_assert(RS == 1 || NT == 256, "only the first pass without the RSSI runs on the larger tile");
    const WmPush &g = a.g;
    if (RS != 1) __syncthreads();                             /* magnitude rows complete (RS = 1 writes none: the caller's barrier is the only one) */
    /* The four waves have different jobs here (EMA + the 11-tap FIR: ~520 instructions; the 46-tap FIR:
     * ~870): the jobs rotate with the tile, so that no SIMD of a CU can end up with the heavy job block
     * after block whatever order the hardware places a workgroup's waves in (two instructions). */
    const int e = tid & 63, wv = ((tid >> 6) + tile) & (NT / 64 - 1), rt = 64 * wv + e;      /* rt: thread id within the rotated roles */
    if (RS == 1) {                                            /* no RSSI here: every wave a quarter of each low-pass (4 x 92 + 2 x 44) */
        /* two waves start with the short filter: all four reading their 13 + 4 window vectors at once, right behind the barrier,
         * queued on the CU's one LDS pipe (SQ_WAIT_INST_LDS doubled against the full kernel, whose roles differ by wave; r04 A/B:
         * K1 alone 2.32 -> 2.15 ms, the job 159.2 -> 161.2) */
        if ((wv & 2) && chT) k1_fir_t<FAST>(a, yDrT, 64 * wv + e, stream, ts, tn);
        if (chS) k1_fir_s<FAST>(a, yDrS, 64 * wv + e, stream, ts, tn);
        if (!(wv & 2) && chT) k1_fir_t<FAST>(a, yDrT, 64 * wv + e, stream, ts, tn);
        return;
    }
    const int ch = wv & 1;                                    /* EMA chain of waves 0, 1 */
    const bool on = wv < 2 && (ch ? chS : chT) && 16 * e < T;
    const float al = 0.6789f, be = wm_sub(1.0f, 0.6789f);
    const int m0l = 16 * e;
    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint64_t rows = 2ull * g.S, ti = (uint64_t)tile * rows + row;
    if (GEN && a.relist != nullptr) {
        /* REPAIR: a hand-off of this tile could not be certified (e.g. exact-zero input after a
         * signal: the true state decays through 90 more samples while a warm-up from zero is already
         * at zero).  One lane per chain runs the whole tile sequentially from the predecessor's exact
         * tail -- slow, exact, and only for the listed tiles. */
        /* the soft symbols of the tile stay as the first pass wrote them: a repair concerns the RSSI filter only (and in
         * tolerance mode a re-computed tile would mix the exact arithmetic of this kernel into the FMA low-pass's output:
         * which tiles are repaired depends on the push size, the soft symbols must not -- ADVICE r3) */
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
            sFin[rt] = ema; sHead[rt] = head;
        }
        if (WM_K1_BALANCED) { if (chS) k1_fir_s<FAST>(a, yDrS, 64 * wv + e, stream, ts, tn); }
        else if (chT) { k1_fir_t<FAST>(a, yDrT, 128 * wv + e, stream, ts, tn); k1_fir_t<FAST>(a, yDrT, 128 * wv + 64 + e, stream, ts, tn); }
    } else if (WM_K1_BALANCED) {
        if (chS) k1_fir_s<FAST>(a, yDrS, 64 * wv + e, stream, ts, tn);
        if (chT) { k1_fir_t<FAST>(a, yDrT, 128 * (wv - 2) + e, stream, ts, tn); k1_fir_t<FAST>(a, yDrT, 128 * (wv - 2) + 64 + e, stream, ts, tn); }
    } else if (chS) {
        k1_fir_s<FAST>(a, yDrS, 128 * (wv - 2) + e, stream, ts, tn);
        k1_fir_s<FAST>(a, yDrS, 128 * (wv - 2) + 64 + e, stream, ts, tn);
    }
    __syncthreads();
    /* certify: a lane's warm-up must have landed exactly on its predecessor's trajectory; a tile
     * with an uncertified lane publishes a head that cannot match (NaN) and is repaired */
    const bool bad = on && e > 0 && m0l < tn && wm_f2u(head) != wm_f2u(sFin[rt - 1]);
    const unsigned long long badT = __ballot(bad);            /* waves 0 and 1 are the two chains */
    if (on) {
        if (e == 0) a.ema_head[ti] = badT ? wm_u2f(0x7FC00000u) : head;
        if (m0l <= tn - 1 && tn - 1 < m0l + 16) a.ema_tail[ti] = tail;
    }
}

/* Stage B2 of an RS = 2 launch: the RSSI of one listed tile, every lane proving its own start (see WM_EMA_UPPER above).
 * Waves 0 and 1 take one chain each, 16 samples per lane behind the 32-sample warm-up, as in the full pass. */
__device__ __forceinline__ void k1_stage_rssi(const K1Args &a, const int tid, const int tile, const int stream, const int ts, const int tn,
                                              const uint32_t mask, const float *yMgT, const float *yMgS, float *sFin)
{
    constexpr int T = WM_K1_TILE2;
    const WmPush &g = a.g;
    __syncthreads();                                          /* magnitude rows complete */
    const int e = tid & 63, wv = tid >> 6, ch = wv & 1, rt = 64 * wv + e;
    const bool on = wv < 2 && ((mask >> ch) & 1u) && 16 * e < T;
    const float al = 0.6789f, be = wm_sub(1.0f, 0.6789f);
    const int m0l = 16 * e;
    const uint64_t row = (uint64_t)ch * g.S + stream;
    float lo = 0.0f, hi = WM_EMA_UPPER, head = 0.0f, tail = 0.0f;
    uint32_t pk[4] = {0u, 0u, 0u, 0u};
    bool proven = false;
    if (on) {
        const float *mg = (ch ? yMgS : yMgT) + 17 * e;        /* element 16 e + kk at 17 e + kk + kk/16 */
#pragma unroll
        for (int k = WM_K1_HALO - WM_EMA_WARMUP; k < WM_K1_HALO; k++) {
            const float x = wm_mul(al, mg[k + (k >> 4)]);
            lo = wm_add(x, wm_mul(be, lo)); hi = wm_add(x, wm_mul(be, hi));
        }
        head = lo;
        /* a warm-up that begins at or before the stream's first sample starts from the true state: zero, over zero input */
        proven = wm_f2u(lo) == wm_f2u(hi) || (long)(g.m0 + (uint64_t)ts) + m0l - WM_EMA_WARMUP <= 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            lo = wm_add(wm_mul(al, mg[WM_K1_HALO + k + ((WM_K1_HALO + k) >> 4)]), wm_mul(be, lo));
            pk[k >> 2] |= ((uint32_t)lo & 0xFFu) << (8 * (k & 3));
            if (m0l + k == tn - 1) tail = lo;
        }
        sFin[rt] = lo;
    }
    __syncthreads();
    if (wv >= 2) return;
    const bool read = on && m0l < tn;                         /* lanes whose bytes exist */
    const bool link = read && e > 0 && wm_f2u(head) == wm_f2u(sFin[rt - 1]);
    unsigned long long ok = __ballot(on && proven);
    const unsigned long long lk = __ballot(link), rd = __ballot(read);
    for (;;) {                                                /* a proven predecessor whose end value I started from proves me */
        const unsigned long long nx = ok | ((ok << 1) & lk);
        if (nx == ok) break;
        ok = nx;
    }
    if ((rd & ~ok) != 0ull && e == 0) atomicOr(a.rs_fail, 1u);
    if (read) {
        *(uint4 *)(a.rssi + row * g.Mcap + ts + m0l) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        if (tile == (int)a.ntiles - 1 && m0l <= tn - 1 && tn - 1 < m0l + 16) a.ema_out[row] = tail;
    }
}

/* How many of a lane's NP input words per tile are PREFETCHED across the tile before (first pass, PRE): all of them where they fit the
 * 64 registers of a wave that shares its SIMD with seven others -- decimation 2 and 3 without -s --, else the first few: the rest is
 * asked for at the top of the tile's stage 0 and arrives while the prefetched words are converted.  (Round 5 prefetched all eleven
 * words of -d 5 -s: 31 VGPRs spilled, 84 bytes of scratch per lane in the product's first pass; VERDICT r5 #3.) */
#ifndef WM_K1_PF_PLAIN
#define WM_K1_PF_PLAIN 5
#endif
#ifndef WM_K1_PF_SHIFT
#define WM_K1_PF_SHIFT 16
#endif
__host__ __device__ constexpr int k1_prefetched(int np, bool shift) { return np < (shift ? WM_K1_PF_SHIFT : WM_K1_PF_PLAIN) ? np : (shift ? WM_K1_PF_SHIFT : WM_K1_PF_PLAIN); }

/* The staged input of a tile: dword u of the tile's window (two IQ samples) at src[u], u < NDW; LDS word 2 u - off. */
template <int NT> struct K1Src { const uint32_t *src; long r_al; int off, NDW; };
template <int D, int NT>
__device__ __forceinline__ K1Src<NT> k1_src(const K1Args &a, const int tile, const int stream)
{
    const WmPush &g = a.g;
    const int d = D ? D : (int)g.d, ts = tile * K1GeoT<NT>::T;
    K1Src<NT> s;
    const long r_lo = ((long)(g.m0 + (uint64_t)ts) - WM_K1_HALO) * d - 16 - (long)g.n0;   /* LDS word 0 */
    s.r_al = r_lo & ~1L;
    s.off = (int)(r_lo - s.r_al);                             /* 0 or 1 */
    s.NDW = (K1GeoT<NT>::NA * d + 16 + 1 + 1) / 2;            /* dwords covering the staged samples */
    const uint8_t *base = g.in + (uint64_t)stream * g.in_stride + WM_HIST_BYTES;
    s.src = (const uint32_t *)(base + 2 * s.r_al);
    return s;
}

/* GEN = false: the kernel of the DEFAULT switches (both chains, cargf arctangent, first pass): the switch tests, the
 * -a / -A / -p paths and the RSSI repair walk are not compiled in, so the default path carries no cost for the options
 * (any other configuration, and every repair launch, runs the GEN = true kernel: same arithmetic, same results). */
/* PRE (first pass, compiled decimations): the tile's input words are in `pre` already (loads the caller or the previous tile
 * of this block issued), and once they are converted the same registers receive the words of tile `next` (< 0: none) -- the
 * 4.5 microseconds a block waited for its input before it could do anything (41 % of a wave's life, profiles/r05_k1_stage_cycles.txt)
 * then pass while the block computes.  `first`: the block's first tile (the arctangent's table is loaded once per block). */
template <int D, bool SHIFT, bool GEN, bool FAST = false, int RS = 0, int NT = 256, bool PRE = false>
__device__ __forceinline__ void k1_tile(const K1Args &a, const int tile, const int stream, const int tid,
                                        uint32_t (&pre)[D ? ((K1GeoT<NT>::NA * D + 16 + 1 + 1) / 2 + NT - 1) / NT : 1], const int next, const bool first)
{
    using G = K1GeoT<NT>;
    constexpr int T = G::T, NA = G::NA, YD = G::YD, YM = G::YM;
    const WmPush &g = a.g;
    const int d = D ? D : (int)g.d;
    const int NSTG = G::nstg(d);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *stgT = (uint32_t *)smem + 8;                /* word 0 of a row = oldest sample of the tile */
    uint32_t *stgS = SHIFT ? stgT + NSTG : stgT;
    float *yMgT = (float *)smem, *yMgS = yMgT + YM;       /* overlay the staging rows (see stage A) */
    WM_LDS_STATIC __attribute__((aligned(16))) K1LdsT<NT> lds;        /* see K1LdsT */
    float *yDrT = lds.yDr, *yDrS = yDrT + YD;
    float *sFin = lds.sFin, *sHead = lds.sHead, *tab = lds.tab;

    const int ts = tile * T;
    const int tn = min(T, (int)g.M - ts);
    const bool chT = !GEN || (g.flags & WM_F_T1C1), chS = !GEN || (g.flags & WM_F_S1);
    const bool accurate = !GEN || (g.flags & WM_F_ACCURATE);
    const int approx = !GEN ? 0 : (g.flags & WM_F_APPROX1) ? 1 : (g.flags & WM_F_APPROX2) ? 2 : 0;   /* option: atan2.h's approximations */

    /* ---- stage 0: one dword (two IQ samples) per lane and pass: coalesced loads, LDS stores at a
     * two-word lane stride (the 16-byte-per-lane variant stored at an 8-word stride: 8-way bank
     * conflicts) ------------------------------------------------------------------------------- */
    WM_K1_STAMP_DECL;
    WM_K1_STAMP(0);
    {
        const K1Src<NT> in = k1_src<D, NT>(a, tile, stream);
        const long r_al = in.r_al;
        const int off = in.off, NDW = in.NDW;
        const uint32_t *src = in.src;
        constexpr int NP = D ? ((NA * D + 16 + 1 + 1) / 2 + NT - 1) / NT : 1;     /* loads in flight per lane */
        static_assert(!PRE || D != 0, "the prefetch is for the compiled decimations (one pass of NP loads per lane)");
        const int passes = D ? 1 : (NDW + NT - 1) / NT;
        for (int ps = 0; ps < passes; ps++) {
        uint32_t wv[NP];
#pragma unroll
        for (int it = 0; it < NP; it++) {
            const int u = tid + NT * (it + ps);
            wv[it] = PRE && it < k1_prefetched(NP, SHIFT) ? pre[it] : u < NDW ? src[u] : 0u;
        }
        /* the arctangent's table (5 range rows + range LUT, wm_exact.h): word k by lane k, ONE load issued behind the
         * input loads and waited for with them (round 4 computed it in place: six dependent global loads in the first
         * wave of every block before its input loads went out) */
        uint32_t tabw = 0u;
        if (RS != 2 && ps == 0 && first && tid < WM_ATAN_TAB_WORDS) {
            int k = tid;
#ifdef __HIP_DEVICE_COMPILE__
            asm volatile("" : "+v"(k));                       /* the address is made HERE: hoisted in front of the tile loop it was a register pair spilled */
#endif
            tabw = WM_ATAN_TAB_BITS[k];
        }
#pragma unroll
        for (int it = 0; it < NP; it++) {
            const int u = tid + NT * (it + ps);
            if (u < NDW) {
                const int p = 2 * u - off;
                if (!SHIFT) {
                    /* (int)((float)u - 127.5f) per byte (rtl_wmbus.c:1312-1313 + moving_average_filter.h:47) = u - 127 - (u >> 7):
                     * the (u >> 7) of all four bytes leaves in ONE subtraction (a byte never borrows: u - (u >> 7) >= 0),
                     * then bytes (i,q) -> halfwords and - 127 per halfword: 7 instead of 10 instructions per dword */
                    const wm_s2 c127 = {127, 127};
                    const uint32_t w4 = wv[it] - ((wv[it] >> 7) & 0x01010101u);
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t h = __builtin_amdgcn_perm(0u, w4, k ? 0x0c030c02u : 0x0c010c00u);
                        const wm_s2 q = __builtin_bit_cast(wm_s2, h) - c127;
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
        if (RS != 2 && ps == 0 && first && tid < WM_ATAN_TAB_WORDS) tab[tid] = wm_u2f(tabw);
        }
        if (PRE && next >= 0) {                               /* the next tile's input into the registers this tile's has just left */
            const K1Src<NT> nx = k1_src<D, NT>(a, next, stream);
#pragma unroll
            for (int it = 0; it < k1_prefetched(NP, SHIFT); it++) { const int u = tid + NT * it; pre[it] = u < nx.NDW ? nx.src[u] : 0u; }
        }
    }
    WM_K1_STAMP(1);
    __syncthreads();
    WM_K1_STAMP(2);

    /* ---- stage A: thread = chunk ------------------------------------------------------------- */
    float mgT[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mgS[4] = {0.0f, 0.0f, 0.0f, 0.0f};
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
        if (RS == 2) {                                       /* RSSI only: the magnitudes of the chunk's four samples, of the chains that are read */
            const uint32_t mask = a.rs_flags[(uint64_t)tile * g.S + stream];
#pragma unroll
            for (int j = 0; j < 4; j++) { drT[j] = 0.0f; drS[j] = 0.0f; }
            if (mask & 1u) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float iT = fT[j + 1][0], qT = fT[j + 1][1];
                    mgT[j] = wm_mul(wm_sqrt_dom(wm_fma_exact(iT, iT, wm_mul(qT, qT))), 0.125f);
                }
            }
            if (mask & 2u) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float iS = fS[j + 1][0], qS = fS[j + 1][1];
                    mgS[j] = wm_mul(wm_sqrt_dom(wm_fma_exact(iS, iS, wm_mul(qS, qS))), 0.0625f);
                }
            }
        } else if (accurate && chT && chS && !approx) {      /* default switches: no branch between the eight */
#pragma unroll
            for (int j = 0; j < 4; j++) {
                drT[j] = FAST ? wm_discriminator_tol(fT[j + 1][0], fT[j + 1][1], fT[j][0], fT[j][1]) : wm_discriminator_tab(fT[j + 1][0], fT[j + 1][1], fT[j][0], fT[j][1], tab);
                drS[j] = FAST ? wm_discriminator_tol(fS[j + 1][0], fS[j + 1][1], fS[j][0], fS[j][1]) : wm_discriminator_tab(fS[j + 1][0], fS[j + 1][1], fS[j][0], fS[j][1], tab);
            }
#pragma unroll
            for (int j = 0; j < 4 && RS == 0; j++) {
                const float iT = fT[j + 1][0], qT = fT[j + 1][1], iS = fS[j + 1][0], qS = fS[j + 1][1];
                mgT[j] = wm_mul(wm_sqrt_dom(wm_fma_exact(iT, iT, wm_mul(qT, qT))), 0.125f);
                mgS[j] = wm_mul(wm_sqrt_dom(wm_fma_exact(iS, iS, wm_mul(qS, qS))), 0.0625f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float iT = fT[j + 1][0], qT = fT[j + 1][1], iS = fS[j + 1][0], qS = fS[j + 1][1];
                drT[j] = !chT ? 0.0f : accurate ? (approx ? wm_discriminator_approx(iT, qT, fT[j][0], fT[j][1], approx) : wm_discriminator_tab(iT, qT, fT[j][0], fT[j][1], tab))
                                                : wm_mul(wm_discriminator_fast(iT, qT, fT[j][0], fT[j][1]), 0.015625f);
                drS[j] = !chS ? 0.0f : accurate ? (approx ? wm_discriminator_approx(iS, qS, fS[j][0], fS[j][1], approx) : wm_discriminator_tab(iS, qS, fS[j][0], fS[j][1], tab))
                                                : wm_mul(wm_discriminator_fast(iS, qS, fS[j][0], fS[j][1]), 0.00390625f);
                mgT[j] = chT ? wm_mul(wm_sqrt_dom(wm_fma_exact(iT, iT, wm_mul(qT, qT))), 0.125f) : 0.0f;
                mgS[j] = chS ? wm_mul(wm_sqrt_dom(wm_fma_exact(iS, iS, wm_mul(qS, qS))), 0.0625f) : 0.0f;
                /* before the first sample of the stream the FIR's delay line holds zeros, not the discriminator
                 * of zero input -- the same thing for cargf and -a, but atan2_approximation(0, 0) is not 0 */
                if (approx && (long)(g.m0 + (uint64_t)ts) + 4 * c + j < (long)WM_K1_HALO) { drT[j] = 0.0f; drS[j] = 0.0f; }
            }
        }
        if (RS != 2) {                                       /* element a of a discriminator row lives at word a + 4 */
            *(float4 *)(yDrT + 4 * c + 4) = make_float4(drT[0], drT[1], drT[2], drT[3]);
            *(float4 *)(yDrS + 4 * c + 4) = make_float4(drS[0], drS[1], drS[2], drS[3]);
        }
    }
    WM_K1_STAMP(3);
    __syncthreads();                                          /* staging data retired, discriminator rows complete */
    WM_K1_STAMP(4);
    if (RS != 1) {   /* element a of a magnitude row lives at word a + a/16 (conflict-free 17-word lane stride in B2) */
        const int qb = 4 * tid + (tid >> 2);
#pragma unroll
        for (int j = 0; j < 4; j++) { yMgT[qb + j] = mgT[j]; yMgS[qb + j] = mgS[j]; }
    }

    if (RS == 2) k1_stage_rssi(a, tid, tile, stream, ts, tn, a.rs_flags[(uint64_t)tile * g.S + stream], yMgT, yMgS, sFin);
    else k1_stage_b<GEN, FAST, RS, NT>(a, tid, tile, stream, ts, tn, chT, chS, yDrT, yDrS, yMgT, yMgS, sFin, sHead);
    WM_K1_STAMP(5);
    WM_K1_STAMP_END(tid);
}

/* Waves per SIMD the register budget is cut for: what the block's LDS lets onto a CU anyway (160 KB; a 256-thread block is one wave per
 * SIMD, a 512-thread block two).  Decimation 2 without -s: eight -- 64 VGPRs.  With -s the staging is two rows and at -d 5 a block has
 * 50 KB: three blocks per CU whatever the registers, and a 64-register budget only bought 31 spilled VGPRs. */
template <int D, bool SHIFT, int NT>
__host__ __device__ constexpr int k1_waves_per_simd()
{
    const size_t lds = sizeof(K1LdsT<NT>) + K1GeoT<NT>::smem(D ? D : 2, SHIFT);
    const int blocks = (int)(160u * 1024u / lds), waves = blocks * (NT / 256);
    return waves > 8 ? 8 : waves < 1 ? 1 : waves;
}

template <int D, bool SHIFT, bool GEN = true, bool FAST = false, int RS = 0, int NT = 256>
__global__ __launch_bounds__(NT, (GEN ? 1 : RS == 2 ? 4 : k1_waves_per_simd<D, SHIFT, NT>())) void k1_demod2(K1Args a)          /* the RSSI launch over the listed tiles is small and latency-bound: 128 VGPRs (its -s forms spilled up to 48 registers at 64) */
{
    static_assert(NT == 256 || (NT == 512 && RS == 1 && !GEN), "the 512-thread tile is the first pass's without the RSSI");
    /* One tile per block.  (A bounded grid whose blocks walk several tiles made the kernel itself 6 % faster -- fewer block
     * launches, per-thread addresses kept across tiles -- and the whole job 10 % slower: the framer kernels of the other
     * contexts get onto a CU when demodulation blocks retire, and blocks that live twice as long halve their chances; r03
     * A/B in DESIGN_HISTORY.md section 10; round 5 measured it again with the input prefetched: DESIGN.md section 6.) */
    if (RS == 2) {                                            /* the tiles k3_spans has listed, walked by a fixed grid */
        const uint32_t n = *a.n_relist;
        for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
            uint32_t none[D ? ((K1Geo::NA * D + 16 + 1 + 1) / 2 + 255) / 256 : 1];
            k1_tile<D, SHIFT, GEN, FAST, RS>(a, (int)(a.relist[e] % a.ntiles), (int)(a.relist[e] / a.ntiles), (int)threadIdx.x, none, -1, true);
            __syncthreads();                                  /* the tile's LDS is reused by the next entry */
        }
        return;
    }
    constexpr int NP = D ? ((K1GeoT<NT>::NA * D + 16 + 1 + 1) / 2 + NT - 1) / NT : 1;
    uint32_t pre[NP];
    if constexpr (D != 0 && RS == 1) {
        /* the first pass without the RSSI: a.tpb consecutive tiles per block, each tile's input loaded while the one before is computed */
        const int tpb = a.tpb > 1u ? (int)a.tpb : 1;
        const int t0 = (int)a.tile0 + (int)blockIdx.x * tpb, t1 = a.tpb > 1u ? min(t0 + tpb, (int)a.tile_end) : t0 + 1;
        const int stream = (int)blockIdx.y, tid = (int)threadIdx.x;
        {
            const K1Src<NT> in = k1_src<D, NT>(a, t0, stream);
#pragma unroll
            for (int it = 0; it < k1_prefetched(NP, SHIFT); it++) { const int u = tid + NT * it; pre[it] = u < in.NDW ? in.src[u] : 0u; }
        }
        /* no barrier between two tiles of a block: a tile's stage 0 writes the staging area, which nobody reads behind the
         * tile's second barrier (RS = 1 has no magnitude rows), and the discriminator rows are written behind the NEXT first barrier */
        for (int t = t0; t < t1; t++) k1_tile<D, SHIFT, GEN, FAST, RS, NT, true>(a, t, stream, tid, pre, t + 1 < t1 ? t + 1 : -1, t == t0);
        return;
    }
    if (!GEN || a.relist == nullptr) { k1_tile<D, SHIFT, GEN, FAST, RS, NT>(a, (int)(blockIdx.x + a.tile0), (int)blockIdx.y, (int)threadIdx.x, pre, -1, true); return; }
    /* repair launch: a fixed grid walks the list k1_collect has just written (no host round trip in between) */
    const uint32_t n = *a.n_relist;
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        k1_tile<D, SHIFT, GEN, FAST>(a, (int)(a.relist[e] % a.ntiles), (int)(a.relist[e] / a.ntiles), (int)threadIdx.x, pre, -1, true);
        __syncthreads();                                      /* the tile's LDS is reused by the next entry */
    }
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

__device__ __forceinline__ void k1_tile_ppf(const K1Args &a, const int tile, const int stream)
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
    const int ts = tile * T;
    const int tn = min(T, (int)g.M - ts);
    const bool chT = g.flags & WM_F_T1C1, chS = g.flags & WM_F_S1;
    const bool accurate = g.flags & WM_F_ACCURATE;
    const int approx = (g.flags & WM_F_APPROX1) ? 1 : (g.flags & WM_F_APPROX2) ? 2 : 0;   /* option: atan2.h's approximations */

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
            dr[j] = !accurate ? wm_discriminator_fast(i, q, pi_, pq_) : approx ? wm_discriminator_approx(i, q, pi_, pq_, approx) : wm_discriminator(i, q, pi_, pq_);
            if (approx && (long)(g.m0 + (uint64_t)ts) + 4 * tid + j < (long)WM_K1_HALO) dr[j] = 0.0f;   /* the FIR's delay line starts from zeros */
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
    k1_stage_b<true>(a, tid, tile, stream, ts, tn, chT, chS, yDr, yDr, yMg, yMg, sFin, sHead);
}

__global__ __launch_bounds__(256) void k1_demod_ppf(K1Args a)
{
    if (a.relist == nullptr) { k1_tile_ppf(a, (int)(blockIdx.x + a.tile0), (int)blockIdx.y); return; }
    const uint32_t n = *a.n_relist;
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        k1_tile_ppf(a, (int)(a.relist[e] % a.ntiles), (int)(a.relist[e] / a.ntiles));
        __syncthreads();
    }
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

#endif /* WM_K1_DEMOD_H */
