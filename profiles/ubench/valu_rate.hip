// SUPERSEDED (round 3) by profiles/ubench/fp64_peak.hip, which times whole chip-filling launches with HIP events.  This file
// reads s_memtime around the loop of ONE wave per block; under the SIMD's oldest-first issue that wave runs ahead of its three
// neighbours (fp64_peak.hip measures it: wave 0 is done after 32 % of the launch at 4 waves per SIMD, 16 % at 8), so its "ticks
// per instruction" are not a unit of the SIMD's issue rate and cannot be calibrated into one.  Kept for the relative order it
// showed in round 1 (profiles/r1/ubench_valu_rate.txt).
//
// valu_rate.hip — issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the
// render kernel is made of, measured with s_memtime on gfx950.  Build & run (GPU box):
//   hipcc -O2 --offload-arch=gfx950 profiles/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Each test: W waves per SIMD, every wave runs ITER x 32 independent copies of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, int iters) {
    double a0 = threadIdx.x * 1.0000001 + 1.5, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double b = 1.0000001, c = 0.5;
    float f0 = (float)a0, f1 = (float)a1;
    unsigned u0 = threadIdx.x * 2654435761u + 1, u1 = u0 * 3 + 1;
    int e0 = 0;
    unsigned long long t0, t1;
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) { REP32(asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b));) }
        if constexpr (OP == 1) { REP32(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b));) }
        if constexpr (OP == 2) { REP32(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));) }
        if constexpr (OP == 3) { REP32(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1" : "+v"(a0), "+v"(a1));) }
        if constexpr (OP == 4) { REP32(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1" : "+v"(a0), "+v"(a1));) }
        if constexpr (OP == 5) { REP32(asm volatile("v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %3, %2, vcc" : "=v"(u0), "=v"(u1) : "v"(e0), "v"(f0) : "vcc");) }
        if constexpr (OP == 25) { REP32(asm volatile("v_cmp_lt_f64 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc" : "=v"(u0), "=v"(u1) : "v"(a0), "v"(a1), "v"(e0), "v"(f0) : "vcc");) }
        if constexpr (OP == 26) { REP32(asm volatile("v_min_f64 %0, %0, %2\n v_max_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b));) }
        if constexpr (OP == 27) { REP32(asm volatile("v_cmp_lt_f64 s[20:21], %2, %3\n v_cndmask_b32 %0, %4, %5, s[20:21]\n v_cndmask_b32 %1, %5, %4, s[20:21]" : "=v"(u0), "=v"(u1) : "v"(a0), "v"(a1), "v"(e0), "v"(f0) : "s20", "s21");) }
        if constexpr (OP == 31) { REP32(asm volatile("v_cndmask_b32_e32 %0, %2, %3, vcc\n v_cndmask_b32_e32 %1, %3, %2, vcc" : "=v"(u0), "=v"(u1) : "v"(e0), "v"(f0) : );) }
        if constexpr (OP == 32) { REP32(asm volatile("v_cndmask_b32_e64 %0, %2, %3, vcc\n v_cndmask_b32_e64 %1, %3, %2, vcc" : "=v"(u0), "=v"(u1) : "v"(e0), "v"(f0) : );) }
        if constexpr (OP == 33) { REP32(asm volatile("v_cndmask_b32_e64 %0, %2, %3, s[20:21]\n v_cndmask_b32_e64 %1, %3, %2, s[20:21]" : "=v"(u0), "=v"(u1) : "v"(e0), "v"(f0) : );) }
        if constexpr (OP == 34) { REP32(asm volatile("v_cmp_lt_f64_e32 vcc, %2, %3\n v_cndmask_b32_e32 %0, %4, %5, vcc\n v_cndmask_b32_e32 %1, %5, %4, vcc" : "=v"(u0), "=v"(u1) : "v"(a0), "v"(a1), "v"(e0), "v"(f0) : "vcc");) }
        if constexpr (OP == 35) { REP32(asm volatile("v_cmp_lt_f64_e64 s[20:21], %2, %3\n v_cmp_lt_f64_e64 s[22:23], %3, %2\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %5, %4, s[20:21]" : "=v"(u0), "=v"(u1) : "v"(a0), "v"(a1), "v"(e0), "v"(f0) : "s20", "s21", "s22", "s23", "scc");) }
        if constexpr (OP == 36) { REP32(asm volatile("v_cmp_lt_f64_e32 vcc, %2, %3\n s_nop 1\n v_cndmask_b32_e32 %0, %4, %5, vcc\n v_cndmask_b32_e32 %1, %5, %4, vcc\n v_cndmask_b32_e32 %0, %4, %5, vcc\n v_cndmask_b32_e32 %1, %5, %4, vcc\n v_cndmask_b32_e32 %0, %4, %5, vcc\n v_cndmask_b32_e32 %1, %5, %4, vcc" : "=v"(u0), "=v"(u1) : "v"(a0), "v"(a1), "v"(e0), "v"(f0) : "vcc");) }
        if constexpr (OP == 37) { REP32(asm volatile("v_cmp_lt_f64_e64 s[20:21], %2, %3\n v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %5, %4, s[20:21]\n v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %5, %4, s[20:21]\n v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %5, %4, s[20:21]" : "=v"(u0), "=v"(u1) : "v"(a0), "v"(a1), "v"(e0), "v"(f0) : "s20", "s21");) }
        if constexpr (OP == 38) { REP32(asm volatile("v_add_f64 %0, %0, %2\n v_cndmask_b32_e32 %1, %3, %4, vcc" : "+v"(a0), "=v"(u1) : "v"(b), "v"(e0), "v"(f0) : );) }
        if constexpr (OP == 39) { REP32(asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %0, %0, %2" : "+v"(a0) : "v"(b));) }
        if constexpr (OP == 28) { REP32(asm volatile("s_nop 0\n s_nop 0" ::);) }
        if constexpr (OP == 29) { REP32(asm volatile("s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], exec" ::: "s20", "s21", "s22", "s23", "scc");) }
        if constexpr (OP == 30) { REP32(asm volatile("v_add_f64 %0, %0, %2\n s_and_b64 s[20:21], s[20:21], exec" : "+v"(a0) : "v"(a1), "v"(b) : "s20", "s21", "scc");) }
        if constexpr (OP == 6) { REP32(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %0, %1" ::"v"(a0), "v"(a1) : "vcc");) }
        if constexpr (OP == 7) { REP32(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %0" : "+v"(u0), "+v"(u1));) }
        if constexpr (OP == 8) { REP32(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %0" : "+v"(u0), "+v"(u1));) }
        if constexpr (OP == 9) { REP32(asm volatile("v_div_fixup_f64 %0, %0, %2, %3\n v_div_fixup_f64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));) }
        if constexpr (OP == 10) { REP32(asm volatile("v_frexp_exp_i32_f64 %0, %1\n v_frexp_exp_i32_f64 %0, %2" : "+v"(e0) : "v"(a0), "v"(a1));) }
        if constexpr (OP == 11) { REP32(asm volatile("v_cvt_f32_f64 %0, %2\n v_cvt_f64_f32 %1, %0" : "+v"(f0), "+v"(a1) : "v"(a0));) }
        if constexpr (OP == 12) { REP32(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1" : "+v"(f0), "+v"(f1));) }
        if constexpr (OP == 13) { REP32(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %0, %0" : "+v"(f0), "+v"(f1));) }
        if constexpr (OP == 14) { REP32(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %0" : "+v"(u0), "+v"(u1));) }
        if constexpr (OP == 15) { REP32(asm volatile("v_ldexp_f64 %0, %0, %2\n v_ldexp_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(e0));) }
        if constexpr (OP == 16) { REP32(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %0" : "+v"(u0), "+v"(u1));) }
        if constexpr (OP == 17) { REP32(asm volatile("v_div_scale_f64 %0, vcc, %0, %2, %3\n v_div_scale_f64 %1, vcc, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c) : "vcc");) }
        if constexpr (OP == 18) { REP32(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %0" : "+v"(a0), "+v"(a1));) }
        if constexpr (OP == 19) { REP32(asm volatile("v_cmp_class_f64 vcc, %0, %2\n v_cmp_class_f64 vcc, %1, %2" ::"v"(a0), "v"(a1), "v"(e0) : "vcc");) }
        if constexpr (OP == 20) { REP32(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5" ::"v"(u0), "v"(u1) : "s20", "s21");) }
        if constexpr (OP == 21) { REP32(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1" : "+v"(a0), "+v"(a1) : "v"(u0), "v"(u1) : "vcc");) }
        if constexpr (OP == 22) { REP32(asm volatile("v_cvt_f64_u32 %0, %2\n v_cvt_f64_u32 %1, %3" : "+v"(a0), "+v"(a1) : "v"(u0), "v"(u1));) }
        if constexpr (OP == 23) { REP32(asm volatile("v_add_f64 %0, %0, %2\n v_add_u32 %1, %1, %1" : "+v"(a0), "+v"(u1) : "v"(b));) }
        if constexpr (OP == 24) { REP32(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0" : "+v"(a0), "+v"(a1));) }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + u0 + u1 + e0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, double* out, unsigned long long* cyc) {
    const int iters = 2000, blocks = 256 * 4;  // 4 blocks of 256 threads per CU -> 4 waves per SIMD
    k<OP><<<blocks, 256>>>(out, cyc, 10);
    k<OP><<<blocks, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= blocks;
    // per SIMD: 4 waves x iters x 64 instructions in avg cycles
    printf("%-22s %6.2f cycles / wave-instruction / SIMD (4 waves/SIMD)\n", name, avg / (4.0 * iters * 64));
}

int main() {
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 256 * 8);
    hipMalloc(&cyc, 1024 * 8);
    run<0>("v_add_f64", out, cyc);
    run<1>("v_mul_f64", out, cyc);
    run<2>("v_fma_f64", out, cyc);
    run<3>("v_rcp_f64", out, cyc);
    run<4>("v_rsq_f64", out, cyc);
    run<5>("v_cndmask_b32", out, cyc);
    run<6>("v_cmp_f64", out, cyc);
    run<25>("cmp+nop1+2cndmask (x2)", out, cyc);
    run<27>("cmp sgpr+2cndmask (x1.5)", out, cyc);
    run<26>("v_min/max_f64", out, cyc);
    run<31>("cndmask e32 vcc (no write)", out, cyc);
    run<32>("cndmask e64 vcc", out, cyc);
    run<33>("cndmask e64 sgpr", out, cyc);
    run<34>("cmp,2cnd e32 no nop (x1.5)", out, cyc);
    run<35>("2cmp,s_and,2cnd e64 (x2.5)", out, cyc);
    run<36>("cmp,nop,6cnd e32 (x4)", out, cyc);
    run<37>("cmp,6cnd e64 (x3.5)", out, cyc);
    run<38>("add_f64 + cnd e32 vcc", out, cyc);
    run<39>("dependent add_f64 chain", out, cyc);
    run<28>("s_nop 0", out, cyc);
    run<29>("s_and/or_b64", out, cyc);
    run<30>("v_add_f64 + s_and_b64", out, cyc);
    run<7>("v_mul_lo_u32", out, cyc);
    run<16>("v_mul_hi_u32", out, cyc);
    run<21>("v_mad_u64_u32", out, cyc);
    run<8>("v_add_u32/v_xor_b32", out, cyc);
    run<9>("v_div_fixup_f64", out, cyc);
    run<17>("v_div_scale_f64", out, cyc);
    run<10>("v_frexp_exp_i32_f64", out, cyc);
    run<15>("v_ldexp_f64", out, cyc);
    run<19>("v_cmp_class_f64", out, cyc);
    run<11>("v_cvt f32<->f64", out, cyc);
    run<22>("v_cvt_f64_u32", out, cyc);
    run<12>("v_sqrt_f32", out, cyc);
    run<13>("v_fma_f32", out, cyc);
    run<24>("v_pk_mul_f32", out, cyc);
    run<14>("v_mov_b32", out, cyc);
    run<18>("v_mov_b64", out, cyc);
    run<20>("v_readlane_b32", out, cyc);
    run<23>("v_add_f64 + v_add_u32", out, cyc);
    return 0;
}
