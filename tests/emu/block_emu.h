/* block_emu.h -- TEST INFRASTRUCTURE: run a HIP kernel body for one thread block on the host.
 * Every thread of the block is a coroutine (ucontext); __syncthreads() and the wave-level __ballot
 * are scheduling points: a coroutine parks there and the scheduler releases a block (or a wave) when
 * all of its live members have arrived.  Deterministic, single OS thread.  Enough of the HIP language
 * for the kernels of this repository; not a general emulator. */
#ifndef BLOCK_EMU_H
#define BLOCK_EMU_H
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

struct EmuIdx3 { uint32_t x, y, z; };
static EmuIdx3 threadIdx, blockIdx, blockDim, gridDim;

namespace block_emu {
enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };
struct Co { ucontext_t ctx; State st; std::vector<char> stack; int pred; unsigned long long ballot, val; };
static unsigned long long snap[16][64];     /* values the lanes of a wave posted at their last wave operation */
static std::vector<Co> cos;
static ucontext_t sched_ctx;
static int cur = -1;
static std::function<void()> body;

static void trampoline() { body(); cos[cur].st = DONE; swapcontext(&cos[cur].ctx, &sched_ctx); }

static inline void yield(State s) { cos[cur].st = s; swapcontext(&cos[cur].ctx, &sched_ctx); }

/* Run `fn` as a block of `nthreads` threads (a multiple of 64 or less than 64). */
static void run_block(uint32_t nthreads, const std::function<void()> &fn, size_t stack_bytes = 256 * 1024)
{
    body = fn;
    blockDim = {nthreads, 1, 1};
    if (cos.size() < nthreads) cos.resize(nthreads);
    for (uint32_t t = 0; t < nthreads; t++) {
        Co &c = cos[t];
        if (c.stack.size() < stack_bytes) c.stack.resize(stack_bytes);
        getcontext(&c.ctx);
        c.ctx.uc_stack.ss_sp = c.stack.data();
        c.ctx.uc_stack.ss_size = c.stack.size();
        c.ctx.uc_link = &sched_ctx;
        makecontext(&c.ctx, trampoline, 0);
        c.st = RUNNABLE;
    }
    for (;;) {
        bool progressed = false;
        for (uint32_t t = 0; t < nthreads; t++)
            if (cos[t].st == RUNNABLE) {
                cur = (int)t; threadIdx = {t, 0, 0};
                swapcontext(&sched_ctx, &cos[t].ctx);
                progressed = true;
            }
        uint32_t live = 0, at_block = 0;
        for (uint32_t t = 0; t < nthreads; t++) { live += cos[t].st != DONE; at_block += cos[t].st == WAIT_BLOCK; }
        if (live == 0) break;
        bool released = false;
        for (uint32_t w = 0; w * 64 < nthreads; w++) {              /* waves whose live lanes all wait for a wave op */
            uint32_t lv = 0, ww = 0; unsigned long long mask = 0;
            for (uint32_t l = 0; l < 64 && w * 64 + l < nthreads; l++) {
                const Co &c = cos[w * 64 + l];
                lv += c.st != DONE; ww += c.st == WAIT_WAVE;
                if (c.st == WAIT_WAVE && c.pred) mask |= 1ull << l;
            }
            if (lv && ww == lv) {
                for (uint32_t l = 0; l < 64 && w * 64 + l < nthreads; l++)
                    if (cos[w * 64 + l].st == WAIT_WAVE) { snap[w][l] = cos[w * 64 + l].val; cos[w * 64 + l].ballot = mask; cos[w * 64 + l].st = RUNNABLE; }
                released = true;
            }
        }
        if (!released && at_block == live) {
            for (uint32_t t = 0; t < nthreads; t++) if (cos[t].st == WAIT_BLOCK) cos[t].st = RUNNABLE;
            released = true;
        }
        if (!released && !progressed) { std::fprintf(stderr, "block_emu: deadlock (divergent barrier?)\n"); std::abort(); }
    }
    cur = -1;
}
}  // namespace block_emu

static inline void __syncthreads() { block_emu::yield(block_emu::WAIT_BLOCK); }
static inline unsigned long long __ballot(int pred)
{
    if (block_emu::cur < 0) return pred ? 1ull : 0ull;       /* outside run_block: a lane on its own is a wave of one */
    block_emu::cos[block_emu::cur].pred = pred != 0;
    block_emu::yield(block_emu::WAIT_WAVE);
    return block_emu::cos[block_emu::cur].ballot;
}
/* a wave-level barrier (all live lanes of the wave arrive); a no-op when code runs outside run_block */
static inline void emu_wave_barrier()
{
    if (block_emu::cur < 0) return;
    block_emu::cos[block_emu::cur].pred = 0;
    block_emu::yield(block_emu::WAIT_WAVE);
}
/* wave shuffles of 32-bit values: every lane posts its value, the wave meets, everyone reads the snapshot */
static inline uint32_t emu_wave_exchange(uint32_t v, int src_lane)
{
    block_emu::cos[block_emu::cur].val = v; block_emu::cos[block_emu::cur].pred = 0;
    block_emu::yield(block_emu::WAIT_WAVE);
    return (uint32_t)block_emu::snap[block_emu::cur / 64][src_lane & 63];
}
static inline uint32_t __shfl(uint32_t v, int src) { return emu_wave_exchange(v, src); }
static inline int __shfl(int v, int src) { return (int)emu_wave_exchange((uint32_t)v, src); }
static inline uint32_t __shfl_xor(uint32_t v, int m) { return emu_wave_exchange(v, (block_emu::cur & 63) ^ m); }
#endif
