/* k1_emu.cpp -- TEST INFRASTRUCTURE: the demodulation kernels' DEVICE SOURCE
 * (rtl-wmbus_amd/csrc/wm_k1_demod.h: front end, boxcars, table-driven atan2f, FIR, RSSI filter with its
 * in-block certification) compiled for the host with clang++ and run block by block on the coroutine
 * block emulator (block_emu.h), with the hand-off verification / repair loop of wm_api.hip around it. */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "block_emu.h"

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
using std::max;
using std::min;
static inline uint32_t atomicMin(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = std::min(o, v); return o; }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p += v; return o; }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p |= v; return o; }
/* v_perm_b32: byte i of the result is picked by selector byte i from {s0 (bytes 4..7), s1 (bytes 0..3)};
 * 0x0c gives 0x00, 0x0d and above 0xff, 8..11 the sign of bytes 1, 3, 5, 7 */
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (sel >> (8 * i)) & 0xFFu;
        uint32_t b;
        if (s <= 7u) b = (uint32_t)(src >> (8 * s)) & 0xFFu;
        else if (s <= 11u) b = ((src >> (8 * (2 * (s - 8u) + 1) + 7)) & 1u) ? 0xFFu : 0x00u;
        else if (s == 12u) b = 0x00u;
        else b = 0xFFu;
        out |= b << (8 * i);
    }
    return out;
}

alignas(16) char smem[160 * 1024];          /* what `extern __shared__ char smem[]` in the kernels refers to */

#include "wm_dev.h"
#include "wm_exact.h"
#include "wm_k1_demod.h"

template <int D, bool SHIFT> static void run_k1(const K1Args &a, uint32_t gx, uint32_t gy)
{
    for (uint32_t y = 0; y < gy; y++)
        for (uint32_t x = 0; x < gx; x++) {
            blockIdx = {x, y, 0};
            /* the host's selection (launch_k1v2 in wm_api.hip): default switches, first pass -> the kernel without the option paths */
            const uint32_t need = WM_F_ACCURATE | WM_F_T1C1 | WM_F_S1, never = WM_F_APPROX1 | WM_F_APPROX2;
            if (D != 0 && a.relist == nullptr && (a.g.flags & need) == need && !(a.g.flags & never))
                block_emu::run_block(256, [&] { k1_demod2<D, SHIFT, false>(a); });
            else
                block_emu::run_block(256, [&] { k1_demod2<D, SHIFT, true>(a); });
        }
}

extern "C" { int wm_emu_k1_big = 0; int wm_emu_k1_tpb = 1; }      /* tpb: consecutive tiles per block of the first pass (K1Args.tpb) */      /* 1: the first pass without the RSSI on 2000-sample tiles of 512 threads (wmbus_ctx.k1_big: d = 2, no -s) */

template <int D, bool SHIFT> static void run_k1_od(K1Args a, uint32_t ntiles, uint32_t S, const std::vector<uint32_t> &list, uint32_t *n_list)
{
    const uint32_t tpb = (uint32_t)(wm_emu_k1_tpb > 1 ? wm_emu_k1_tpb : 1);
    a.tpb = tpb;
    if (D == 2 && !SHIFT && wm_emu_k1_big) {
        const uint32_t nt1 = (a.g.M + WM_K1_TILE_BIG - 1) / WM_K1_TILE_BIG, nb = (nt1 + tpb - 1) / tpb;
        a.tile_end = nt1;
        gridDim = {nb, S, 1};
        for (uint32_t y = 0; y < S; y++)
            for (uint32_t x = 0; x < nb; x++) { blockIdx = {x, y, 0}; block_emu::run_block(512, [&] { k1_demod2<2, false, false, false, 1, 512>(a); }); }
    } else {
    /* like the product: the push's tiles leave in two launches (the turn of the demodulation kernel is handed over early) */
    const uint32_t n_tail = ntiles > 3 ? ntiles / 3 : 0, n_main = ntiles - n_tail;
    for (int part = 0; part < 2; part++) {
        const uint32_t lo = part ? n_main : 0, hi = part ? ntiles : n_main, nb = (hi - lo + tpb - 1) / tpb;
        if (hi == lo) continue;
        a.tile0 = lo; a.tile_end = hi;
        gridDim = {nb, S, 1};
        for (uint32_t y = 0; y < S; y++)
            for (uint32_t x = 0; x < nb; x++) { blockIdx = {x, y, 0}; block_emu::run_block(256, [&] { k1_demod2<D, SHIFT, false, false, 1>(a); }); }
    }
    a.tile0 = 0;
    }
    a.tpb = 0; a.tile_end = 0;
    a.relist = list.data(); a.n_relist = n_list;
    gridDim = {5, 1, 1};                                                    /* a fixed grid walks the list */
    for (uint32_t x = 0; x < 5; x++) { blockIdx = {x, 0, 0}; block_emu::run_block(256, [&] { k1_demod2<D, SHIFT, false, false, 2>(a); }); }
}

extern "C" {

/* One push.  in: S rows of in_stride bytes (4096 history bytes first); dphi: [2][S][Mcap]; rssi:
 * [2][S][Mcap]; ema_carry: [2][S] in/out.  Returns the number of repaired tiles, -1 on no convergence. */
long wm_emu_k1(const uint8_t *in, uint64_t in_stride, uint32_t S, uint32_t d, uint32_t flags, uint64_t n0, uint32_t n_new,
               uint32_t Mcap, float *dphi, uint8_t *rssi, float *ema_carry, uint32_t *err_out, int polyphase)
{
    WmPush g{};
    g.in = in; g.in_stride = in_stride; g.n0 = n0; g.m0 = n0 / d; g.n_new = n_new;
    g.M = (uint32_t)((n0 + n_new) / d - g.m0); g.Mcap = Mcap; g.d = d; g.S = S; g.lut_n = 32 * d;
    g.lut_phase0 = (uint32_t)((13ull * (n0 % g.lut_n)) % g.lut_n); g.flags = flags;
    if (g.M == 0) return 0;
    const uint32_t T = WM_K1_TILE2, ntiles = (g.M + T - 1) / T, rows = 2 * S;
    std::vector<float> lut(2 * 32 * WM_MAX_DECIM, 0.f), head((size_t)ntiles * rows), tail((size_t)ntiles * rows);
    {   /* wm_api.hip wmbus_open: rtl_wmbus.c:974-993 */
        const int fs_khz = (int)d * 800;
        for (size_t n = 0; n < (size_t)(fs_khz / 25); n++) {
            const double phi = (2. * M_PI * (25 * (double)n)) / fs_khz;
            lut[n] = cosf(phi); lut[32 * WM_MAX_DECIM + n] = -sinf(phi);
        }
    }
    std::vector<uint32_t> first_bad(rows, 0xFFFFFFFFu), relist((size_t)rows * ntiles + 1);
    uint32_t err = 0, n_relist = 0;
    K1Args a{g, dphi, rssi, lut.data(), lut.data() + 32 * WM_MAX_DECIM, head.data(), tail.data(), ntiles, &err, nullptr, ema_carry, nullptr, 0u};
    const bool sh = flags & WM_F_SHIFT;
    auto launch = [&](uint32_t n_list) {
        a.relist = n_list ? relist.data() : nullptr;
        a.n_relist = &n_relist;                                             /* repair launches: a fixed grid walks the list (3 blocks here) */
        const uint32_t gx = n_list ? 3u : ntiles, gy = n_list ? 1 : S;
        gridDim = {gx, gy, 1};
        if (polyphase) {                                                    /* ppf.h pre-filter (d = 2, no shift) */
            for (uint32_t y = 0; y < gy; y++)
                for (uint32_t x = 0; x < gx; x++) { blockIdx = {x, y, 0}; block_emu::run_block(256, [&] { k1_demod_ppf(a); }); }
            return;
        }
        switch (d) {
        case 2: sh ? run_k1<2, true>(a, gx, gy) : run_k1<2, false>(a, gx, gy); break;
        case 3: sh ? run_k1<3, true>(a, gx, gy) : run_k1<3, false>(a, gx, gy); break;
        case 4: sh ? run_k1<4, true>(a, gx, gy) : run_k1<4, false>(a, gx, gy); break;
        case 5: sh ? run_k1<5, true>(a, gx, gy) : run_k1<5, false>(a, gx, gy); break;
        default: sh ? run_k1<0, true>(a, gx, gy) : run_k1<0, false>(a, gx, gy); break;
        }
    };
    launch(0);
    long repaired = 0;
    for (uint32_t round = 0;; round++) {
        blockDim = {64, 1, 1};
        for (uint32_t t = 0; t < ntiles; t++)                               /* k1_verify */
            for (uint32_t r = 0; r < rows; r++) { blockIdx = {r / 64, t, 0}; threadIdx = {r % 64, 0, 0}; k1_verify(head.data(), tail.data(), ema_carry, ntiles, rows, first_bad.data()); }
        n_relist = 0;
        for (uint32_t r = 0; r < rows; r++) { blockIdx = {r / 64, 0, 0}; threadIdx = {r % 64, 0, 0}; k1_collect(first_bad.data(), ntiles, rows, S, relist.data(), &n_relist); }
        if (n_relist == 0) break;
        if (round > ntiles + 1) return -1;
        repaired += n_relist;
        launch(n_relist);
    }
    for (uint32_t r = 0; r < rows; r++) { blockIdx = {r / 64, 0, 0}; threadIdx = {r % 64, 0, 0}; k1_commit(tail.data(), ema_carry, ntiles, rows); }
    if (err_out) *err_out = err;
    return repaired;
}

/* RSSI on demand (wm_k1_demod.h): the first pass without the RSSI (RS = 1), then the RSSI of the tiles flagged in
 * tile_flags [ntiles][S] (bit 0: T1/C1 chain, bit 1: S1; the last tile of every capture is added, as k3_spans does) by an
 * RS = 2 launch over the list.  Default switches, d = 2..5 only.  ema_out: [2][S] the filter's state after the push.
 * Returns 1 when a lane that is read could not prove its value (the product then runs the full pass), else 0. */
long wm_emu_k1_od(const uint8_t *in, uint64_t in_stride, uint32_t S, uint32_t d, uint32_t flags, uint64_t n0, uint32_t n_new,
                  uint32_t Mcap, float *dphi, uint8_t *rssi, float *ema_out, uint32_t *tile_flags)
{
    WmPush g{};
    g.in = in; g.in_stride = in_stride; g.n0 = n0; g.m0 = n0 / d; g.n_new = n_new;
    g.M = (uint32_t)((n0 + n_new) / d - g.m0); g.Mcap = Mcap; g.d = d; g.S = S; g.lut_n = 32 * d;
    g.lut_phase0 = (uint32_t)((13ull * (n0 % g.lut_n)) % g.lut_n); g.flags = flags;
    if (g.M == 0 || d < 2 || d > 5) return -1;
    const uint32_t T = WM_K1_TILE2, ntiles = (g.M + T - 1) / T;
    std::vector<float> lut(2 * 32 * WM_MAX_DECIM, 0.f);
    const int fs_khz = (int)d * 800;
    for (size_t n = 0; n < (size_t)(fs_khz / 25); n++) {
        const double phi = (2. * M_PI * (25 * (double)n)) / fs_khz;
        lut[n] = cosf(phi); lut[32 * WM_MAX_DECIM + n] = -sinf(phi);
    }
    std::vector<uint32_t> list;
    for (uint32_t s = 0; s < S; s++) tile_flags[(size_t)(ntiles - 1) * S + s] |= 3u;
    for (uint32_t tl = 0; tl < ntiles; tl++)
        for (uint32_t s = 0; s < S; s++) if (tile_flags[(size_t)tl * S + s]) list.push_back(s * ntiles + tl);
    uint32_t err = 0, fail = 0, n_list = (uint32_t)list.size();
    K1Args a{g, dphi, rssi, lut.data(), lut.data() + 32 * WM_MAX_DECIM, nullptr, nullptr, ntiles, &err, nullptr, nullptr, nullptr, 0u, tile_flags, ema_out, &fail};
    const bool sh = flags & WM_F_SHIFT;
    switch (d) {
    case 2: sh ? run_k1_od<2, true>(a, ntiles, S, list, &n_list) : run_k1_od<2, false>(a, ntiles, S, list, &n_list); break;
    case 3: sh ? run_k1_od<3, true>(a, ntiles, S, list, &n_list) : run_k1_od<3, false>(a, ntiles, S, list, &n_list); break;
    case 4: sh ? run_k1_od<4, true>(a, ntiles, S, list, &n_list) : run_k1_od<4, false>(a, ntiles, S, list, &n_list); break;
    default: sh ? run_k1_od<5, true>(a, ntiles, S, list, &n_list) : run_k1_od<5, false>(a, ntiles, S, list, &n_list); break;
    }
    return err ? -2 : (long)fail;
}

}
