// tools/ubench.hip -- VALU issue-rate / latency micro-benchmarks for gfx950 (development aid).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
// Prints cycles per wave-instruction (s_memtime ticks) for throughput (8 independent chains)
// and latency (1 dependent chain) at 1 wave / SIMD and at 4 waves / SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP 64
#define LOOPS 64

#define DEF_BENCH(NAME, DECL, INIT, BODY_T, BODY_L, SINK)                                         \
    __global__ void NAME##_thr(uint64_t *out, float seed) {                                       \
        DECL; INIT;                                                                               \
        uint64_t t0 = __builtin_readcyclecounter();                                               \
        for (int l = 0; l < LOOPS; l++) {                                                         \
            _Pragma("unroll") for (int r = 0; r < REP / 8; r++) { BODY_T }                        \
        }                                                                                         \
        uint64_t t1 = __builtin_readcyclecounter();                                               \
        SINK;                                                                                     \
        if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;   \
    }                                                                                             \
    __global__ void NAME##_lat(uint64_t *out, float seed) {                                       \
        DECL; INIT;                                                                               \
        uint64_t t0 = __builtin_readcyclecounter();                                               \
        for (int l = 0; l < LOOPS; l++) {                                                         \
            _Pragma("unroll") for (int r = 0; r < REP; r++) { BODY_L }                            \
        }                                                                                         \
        uint64_t t1 = __builtin_readcyclecounter();                                               \
        SINK;                                                                                     \
        if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;   \
    }

#define F8 float a0, a1, a2, a3, a4, a5, a6, a7, b
#define F8I a0 = seed; a1 = seed + 1; a2 = seed + 2; a3 = seed + 3; a4 = seed + 4; a5 = seed + 5; a6 = seed + 6; a7 = seed + 7; b = seed * 0.999f + 1.0f
#define F8S if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[1000000] = 1
#define OP1(op, x) asm volatile(op " %0, %0, %1" : "+v"(x) : "v"(b));
#define T8(op) OP1(op, a0) OP1(op, a1) OP1(op, a2) OP1(op, a3) OP1(op, a4) OP1(op, a5) OP1(op, a6) OP1(op, a7)

DEF_BENCH(mul, F8, F8I, T8("v_mul_f32"), OP1("v_mul_f32", a0), F8S)
DEF_BENCH(add, F8, F8I, T8("v_add_f32"), OP1("v_add_f32", a0), F8S)
#define OPF(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
DEF_BENCH(fma, F8, F8I, OPF(a0) OPF(a1) OPF(a2) OPF(a3) OPF(a4) OPF(a5) OPF(a6) OPF(a7), OPF(a0), F8S)
#define OPU(op, x) asm volatile(op " %0, %0" : "+v"(x));
#define U8(op) OPU(op, a0) OPU(op, a1) OPU(op, a2) OPU(op, a3) OPU(op, a4) OPU(op, a5) OPU(op, a6) OPU(op, a7)
DEF_BENCH(rcp, F8, F8I, U8("v_rcp_f32"), OPU("v_rcp_f32", a0), F8S)
DEF_BENCH(sqrt, F8, F8I, U8("v_sqrt_f32"), OPU("v_sqrt_f32", a0), F8S)
DEF_BENCH(cvt, F8, F8I, U8("v_cvt_f32_i32"), OPU("v_cvt_f32_i32", a0), F8S)
#define OPC(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : );
DEF_BENCH(cnd, F8, F8I, OPC(a0) OPC(a1) OPC(a2) OPC(a3) OPC(a4) OPC(a5) OPC(a6) OPC(a7), OPC(a0), F8S)
#define OPX(x) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
DEF_BENCH(fixup, F8, F8I, OPX(a0) OPX(a1) OPX(a2) OPX(a3) OPX(a4) OPX(a5) OPX(a6) OPX(a7), OPX(a0), F8S)
#define OPM(x) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b) : );
DEF_BENCH(fmas, F8, F8I, OPM(a0) OPM(a1) OPM(a2) OPM(a3) OPM(a4) OPM(a5) OPM(a6) OPM(a7), OPM(a0), F8S)
#define OPI(x) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(x) : "v"(b));
DEF_BENCH(pki16, F8, F8I, OPI(a0) OPI(a1) OPI(a2) OPI(a3) OPI(a4) OPI(a5) OPI(a6) OPI(a7), OPI(a0), F8S)

#define P8 f2 a0, a1, a2, a3, a4, a5, a6, a7, b
#define P8I a0 = f2{seed, seed}; a1 = a0 + 1.f; a2 = a0 + 2.f; a3 = a0 + 3.f; a4 = a0 + 4.f; a5 = a0 + 5.f; a6 = a0 + 6.f; a7 = a0 + 7.f; b = a0 * 0.999f + 1.0f
#define P8S { f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; if (s.x + s.y == 123.456f) out[1000000] = 1; }
DEF_BENCH(pkmul, P8, P8I, T8("v_pk_mul_f32"), OP1("v_pk_mul_f32", a0), P8S)
DEF_BENCH(pkadd, P8, P8I, T8("v_pk_add_f32"), OP1("v_pk_add_f32", a0), P8S)
#define OPPF(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
DEF_BENCH(pkfma, P8, P8I, OPPF(a0) OPPF(a1) OPPF(a2) OPPF(a3) OPPF(a4) OPPF(a5) OPPF(a6) OPPF(a7), OPPF(a0), P8S)


// ---- second batch: integer / select / compare forms --------------------------------------------
DEF_BENCH(and_, F8, F8I, T8("v_and_b32"), OP1("v_and_b32", a0), F8S)
DEF_BENCH(lshl, F8, F8I, T8("v_lshlrev_b32"), OP1("v_lshlrev_b32", a0), F8S)
DEF_BENCH(addu, F8, F8I, T8("v_add_u32"), OP1("v_add_u32", a0), F8S)
DEF_BENCH(maxf, F8, F8I, T8("v_max_f32"), OP1("v_max_f32", a0), F8S)
#define OP3(op, x) asm volatile(op " %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define T83(op) OP3(op, a0) OP3(op, a1) OP3(op, a2) OP3(op, a3) OP3(op, a4) OP3(op, a5) OP3(op, a6) OP3(op, a7)
DEF_BENCH(bfi, F8, F8I, T83("v_bfi_b32"), OP3("v_bfi_b32", a0), F8S)
DEF_BENCH(med3, F8, F8I, T83("v_med3_f32"), OP3("v_med3_f32", a0), F8S)
DEF_BENCH(perm, F8, F8I, T83("v_perm_b32"), OP3("v_perm_b32", a0), F8S)
DEF_BENCH(mad24, F8, F8I, T83("v_mad_u32_u24"), OP3("v_mad_u32_u24", a0), F8S)
DEF_BENCH(and_or, F8, F8I, T83("v_and_or_b32"), OP3("v_and_or_b32", a0), F8S)
// cndmask with an SGPR-pair condition (VOP3) instead of VCC
#define OPCS(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x) : "v"(b) : "s10", "s11");
DEF_BENCH(cnd64, F8, F8I, OPCS(a0) OPCS(a1) OPCS(a2) OPCS(a3) OPCS(a4) OPCS(a5) OPCS(a6) OPCS(a7), OPCS(a0), F8S)
// compare only (writes vcc), and compare+select pairs as the compiler emits them
#define OPCMP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
DEF_BENCH(cmp, F8, F8I, OPCMP(a0) OPCMP(a1) OPCMP(a2) OPCMP(a3) OPCMP(a4) OPCMP(a5) OPCMP(a6) OPCMP(a7), OPCMP(a0), F8S)
#define OPCMPS(x) asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" : : "v"(x), "v"(b) : "s10", "s11");
DEF_BENCH(cmp64, F8, F8I, OPCMPS(a0) OPCMPS(a1) OPCMPS(a2) OPCMPS(a3) OPCMPS(a4) OPCMPS(a5) OPCMPS(a6) OPCMPS(a7), OPCMPS(a0), F8S)
#define OPCC(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
DEF_BENCH(cmpcnd, F8, F8I, OPCC(a0) OPCC(a1) OPCC(a2) OPCC(a3) OPCC(a4) OPCC(a5) OPCC(a6) OPCC(a7), OPCC(a0), F8S)
// mixed stream: 3 muls per cndmask (does the select stall the following VALU?)
#define OPMIX(x, y) asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_mul_f32 %0, %0, %2\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x), "+v"(y) : "v"(b));
DEF_BENCH(mix31, F8, F8I, OPMIX(a0, a1) OPMIX(a2, a3) OPMIX(a4, a5) OPMIX(a6, a7) OPMIX(a0, a1) OPMIX(a2, a3) OPMIX(a4, a5) OPMIX(a6, a7), OPMIX(a0, a1), F8S)
#define OPCVTU(x) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x));
DEF_BENCH(cvtub, F8, F8I, OPCVTU(a0) OPCVTU(a1) OPCVTU(a2) OPCVTU(a3) OPCVTU(a4) OPCVTU(a5) OPCVTU(a6) OPCVTU(a7), OPCVTU(a0), F8S)
#define OPDS(x) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %1" : "+v"(x) : "v"(b) : "vcc");
DEF_BENCH(dscale, F8, F8I, OPDS(a0) OPDS(a1) OPDS(a2) OPDS(a3) OPDS(a4) OPDS(a5) OPDS(a6) OPDS(a7), OPDS(a0), F8S)


// ---- third batch: operand kinds ------------------------------------------------------------------
#define OPLIT(x) asm volatile("v_mul_f32 %0, 0x3f7c0e52, %0" : "+v"(x));
DEF_BENCH(mullit, F8, F8I, OPLIT(a0) OPLIT(a1) OPLIT(a2) OPLIT(a3) OPLIT(a4) OPLIT(a5) OPLIT(a6) OPLIT(a7), OPLIT(a0), F8S)
#define OPSG(x) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(x) : : "s20");
DEF_BENCH(mulsgpr, F8, F8I, OPSG(a0) OPSG(a1) OPSG(a2) OPSG(a3) OPSG(a4) OPSG(a5) OPSG(a6) OPSG(a7), OPSG(a0), F8S)
#define OPINL(x) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x));
DEF_BENCH(mulinl, F8, F8I, OPINL(a0) OPINL(a1) OPINL(a2) OPINL(a3) OPINL(a4) OPINL(a5) OPINL(a6) OPINL(a7), OPINL(a0), F8S)
// FIR-like: t = lit * w ; acc = acc + t   (two instructions per OP; 4 accumulators in flight)
#define OPFIRL(acc, w) asm volatile("v_mul_f32 %1, 0x3f7c0e52, %2\n\tv_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(tmpf) : "v"(w));
#define OPFIRS(acc, w) asm volatile("v_mul_f32 %1, s20, %2\n\tv_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(tmpf) : "v"(w) : "s20");
#define F8T float a0, a1, a2, a3, a4, a5, a6, a7, b, tmpf
DEF_BENCH(firlit, F8T, F8I, OPFIRL(a0, a4) OPFIRL(a1, a5) OPFIRL(a2, a6) OPFIRL(a3, a7) OPFIRL(a0, a5) OPFIRL(a1, a6) OPFIRL(a2, a7) OPFIRL(a3, a4), OPFIRL(a0, a4), F8S)
DEF_BENCH(firsgpr, F8T, F8I, OPFIRS(a0, a4) OPFIRS(a1, a5) OPFIRS(a2, a6) OPFIRS(a3, a7) OPFIRS(a0, a5) OPFIRS(a1, a6) OPFIRS(a2, a7) OPFIRS(a3, a4), OPFIRS(a0, a4), F8S)
// three distinct VGPR operands
#define OP3R(x, y, z) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
DEF_BENCH(mul3r, F8, F8I, OP3R(a0, a1, a2) OP3R(a1, a2, a3) OP3R(a2, a3, a4) OP3R(a3, a4, a5) OP3R(a4, a5, a6) OP3R(a5, a6, a7) OP3R(a6, a7, a0) OP3R(a7, a0, a1), OP3R(a0, a0, a1), F8S)
#define OPSUB(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b));
DEF_BENCH(sub, F8, F8I, OPSUB(a0) OPSUB(a1) OPSUB(a2) OPSUB(a3) OPSUB(a4) OPSUB(a5) OPSUB(a6) OPSUB(a7), OPSUB(a0), F8S)
#define OPCVTI(x) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x));
DEF_BENCH(cvtu32, F8, F8I, OPCVTI(a0) OPCVTI(a1) OPCVTI(a2) OPCVTI(a3) OPCVTI(a4) OPCVTI(a5) OPCVTI(a6) OPCVTI(a7), OPCVTI(a0), F8S)
#define OPMOV(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(b));
DEF_BENCH(mov, F8, F8I, OPMOV(a0) OPMOV(a1) OPMOV(a2) OPMOV(a3) OPMOV(a4) OPMOV(a5) OPMOV(a6) OPMOV(a7), OPMOV(a0), F8S)
#define OPXOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
DEF_BENCH(xor_, F8, F8I, OPXOR(a0) OPXOR(a1) OPXOR(a2) OPXOR(a3) OPXOR(a4) OPXOR(a5) OPXOR(a6) OPXOR(a7), OPXOR(a0), F8S)
#define OPMULLO(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
DEF_BENCH(mulu24, F8, F8I, OPMULLO(a0) OPMULLO(a1) OPMULLO(a2) OPMULLO(a3) OPMULLO(a4) OPMULLO(a5) OPMULLO(a6) OPMULLO(a7), OPMULLO(a0), F8S)

// LDS read throughput / latency
__global__ void lds_b32(uint64_t *out, float seed) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = seed + i;
    __syncthreads();
    float acc = 0; int idx = threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int l = 0; l < LOOPS; l++) {
#pragma unroll
        for (int r = 0; r < REP; r++) acc += s[(idx + r * 64) & 4095];
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (acc == 123.456f) out[1000000] = 1;
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

typedef void (*kern_t)(uint64_t *, float);
struct Entry { const char *name; kern_t thr, lat; };

static double run(kern_t k, int blocks, int threads, uint64_t *d_out, int per)
{
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 1.25f);
    hipDeviceSynchronize();
    int waves = blocks * threads / 64;
    std::vector<uint64_t> h(waves);
    hipMemcpy(h.data(), d_out, waves * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    return s / waves / (double)(LOOPS * per);
}

int main()
{
    uint64_t *d_out; hipMalloc(&d_out, 8 * 1000001 + 64);
    Entry e[] = {
#define E(n) {#n, n##_thr, n##_lat}
        E(mul), E(add), E(fma), E(rcp), E(sqrt), E(cvt), E(cnd), E(fixup), E(fmas), E(pki16), E(pkmul), E(pkadd), E(pkfma), E(and_), E(lshl), E(addu), E(maxf), E(bfi), E(med3), E(perm), E(mad24), E(and_or), E(cnd64), E(cmp), E(cmp64), E(cmpcnd), E(mix31), E(cvtub), E(dscale), E(mullit), E(mulsgpr), E(mulinl), E(firlit), E(firsgpr), E(mul3r), E(sub), E(cvtu32), E(mov), E(xor_), E(mulu24)};
    // s_memtime / readcyclecounter ticks: report raw ticks per wave-instruction
    printf("%-8s %12s %12s %12s %12s   (ticks per wave-instruction; thr = 8 indep chains, lat = 1 chain)\n", "op", "thr 1w/SIMD", "thr 4w/SIMD",
           "thr 8w/SIMD", "lat 1w/SIMD");
    for (auto &x : e) {
        for (int w = 0; w < 2; w++) { run(x.thr, 1, 256, d_out, REP); }
        double a = run(x.thr, 1, 256, d_out, REP);     // 4 waves on one CU = 1 per SIMD
        double b = run(x.thr, 1, 1024, d_out, REP);    // 16 waves = 4 per SIMD
        double c2 = run(x.thr, 2, 1024, d_out, REP);   // 2 blocks may land on different CUs: informational
        double c = run(x.lat, 1, 256, d_out, REP);
        printf("%-8s %12.2f %12.2f %12.2f %12.2f\n", x.name, a, b, c2, c);
    }
    double l1 = run(lds_b32, 1, 256, d_out, REP), l4 = run(lds_b32, 1, 1024, d_out, REP);
    printf("lds_b32+add  1w/SIMD %.2f  4w/SIMD %.2f ticks per (ds_read_b32 + v_add) pair\n", l1, l4);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wall = 0; hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
    printf("clockRate kHz %d wallClockRate kHz %d\n", clk, wall);
    return 0;
}
