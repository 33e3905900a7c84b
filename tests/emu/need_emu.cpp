/* need_emu.cpp -- TEST INFRASTRUCTURE: the device function burst_need (wm_k3_bursts.h: how many chips
 * after an access code the burst copy kernel ships to the host) compiled for the host and checked
 * against what the host packet decoder (wm_decoder.c) really consumes.  If burst_need were ever too
 * small, a telegram would be cut short. */
#include <algorithm>
#include <cstdint>
#include <cstring>

#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
struct Idx3 { uint32_t x, y, z; };
static Idx3 threadIdx, blockIdx, gridDim;
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline uint32_t __brev(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
#define WM_WAVE_SYNC() ((void)0)
#define WM_PEEK(p) (*(p))
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }       /* one-lane "wave": only to let the header compile */
template <typename T> static inline T __shfl(T v, int) { return v; }
template <typename T> static inline T __shfl_xor(T v, int) { return v; }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p += v; return o; }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p |= v; return o; }
using std::min;

#include "wm_dev.h"
#include "wm_k2_common.h"
#include "wm_k3_bursts.h"
extern "C" {
#include "wm_decoder.h"
}

extern "C" {

unsigned wm_emu_burst_need(unsigned chain, unsigned hb, unsigned nb) { return burst_need(chain, hb, nb); }
unsigned wm_emu_decoder_bytes(void) { return sizeof(wm_decoder); }

/* Feed `n` chips (bit0 of each byte) to a decoder that has just seen the access code; returns how many
 * it consumed before it went back to idle (or finished a telegram), or n + 1 if it is still receiving. */
unsigned wm_emu_decoder_consumes(int mode, const uint8_t *chips, unsigned n)
{
    wm_decoder d;
    wm_decoder_init(&d, mode);
    int st = wm_decoder_chip(&d, 2u, 100u);                 /* the access-code chip */
    if (st != WM_DEC_RECEIVING) return 0;
    for (unsigned k = 0; k < n; k++) {
        st = wm_decoder_chip(&d, chips[k] & 1u, 100u);
        if (st == WM_DEC_DONE || st == WM_DEC_IDLE) return k + 1;
    }
    return n + 1;
}

}
