// fp64_peak.hip — the fp64 vector peak of this MI355X by WALL CLOCK ("FP64 vector 78.6 TF, AMD
// datasheet; verify by microbenchmark"), and a wall-clock price list of the VALU instruction classes the render
// kernel is made of (the s_memtime list of valu_rate.hip had an uncalibrated tick; this one calibrates it too).
//
//   hipcc -O2 --offload-arch=gfx950 profiles/ubench/fp64_peak.hip -o profiles/ubench/fp64_peak
//   profiles/ubench/fp64_peak [json-file]
//
// Every test fills the chip: 256 CUs x 4 SIMDs x W waves (W = 4 and 8), each wave running ITER x 64 instructions
// over 8 independent accumulators (no dependent issue closer than 8 instructions), timed with HIP events over a
// launch of >= 50 ms so that the clock has settled under the load it will see (DVFS).  Output per class:
// wave-instructions per second per SIMD, the same as ns, relative to v_fma_f64, and — for the FMA classes —
// TFLOP/s counting an FMA as 2 flops (the vendor's convention).  s_memtime and s_memrealtime are read at both ends
// of wave 0 of the timed launch: their rates against the event time give the tick units.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));              \
            std::exit(1);                                                            \
        }                                                                            \
    } while (0)

#define REP8(x) x x x x x x x x

enum Op { FMA64, ADD64, MUL64, FMA32, PKFMA32, CND64E, CMP64, MOV32, MOV64, ADDU32, XOR32, MULLO, MULHI, MADU64, RCP64, RSQ64,
          CVT6432, MIN64, LDEXP64, SQRT32, NOPS, CND32VCC, CMP64VCC, CMPU32, CMPF32, MIN3U32, BFI32, LSHLADD, FFBL, MAXF32, ADD64S,
          RSQ32, CNDCMP };

template <int OP>
__global__ __launch_bounds__(256) void peak_kernel(double* out, unsigned long long* ticks, int iters) {
    double a0 = threadIdx.x * 1.0000001 + 1.5, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
           a7 = a0 + 7;
    double b = 1.0000001, c = 0.5;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6,
          f7 = (float)a7, g = 1.0000001f, h = 0.5f;
    unsigned u0 = threadIdx.x * 2654435761u + 1, u1 = u0 * 3 + 1, u2 = u0 + 7, u3 = u1 + 9, u4 = u0 ^ 5, u5 = u1 ^ 3, u6 = u0 + 11,
             u7 = u1 + 13, m = 0x9E3779B9u;
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    asm volatile("s_mov_b32 s20, 0x33333333\n s_mov_b32 s21, 0x33333333\n s_mov_b64 vcc, s[20:21]" ::: "s20", "s21", "vcc");
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#define A8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define F8 "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
#define U8 "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
#define I8(ins, tail) ins " %0, %0" tail "\n" ins " %1, %1" tail "\n" ins " %2, %2" tail "\n" ins " %3, %3" tail "\n" \
                      ins " %4, %4" tail "\n" ins " %5, %5" tail "\n" ins " %6, %6" tail "\n" ins " %7, %7" tail
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == FMA64) { REP8(asm volatile(I8("v_fma_f64", ", %8, %9") : A8 : "v"(b), "v"(c));) }
        if constexpr (OP == ADD64) { REP8(asm volatile(I8("v_add_f64", ", %8") : A8 : "v"(b));) }
        if constexpr (OP == MUL64) { REP8(asm volatile(I8("v_mul_f64", ", %8") : A8 : "v"(b));) }
        if constexpr (OP == MIN64) { REP8(asm volatile(I8("v_min_f64", ", %8") : A8 : "v"(b));) }
        if constexpr (OP == LDEXP64) { REP8(asm volatile(I8("v_ldexp_f64", ", %8") : A8 : "v"(u0));) }
        if constexpr (OP == FMA32) { REP8(asm volatile(I8("v_fma_f32", ", %8, %9") : F8 : "v"(g), "v"(h));) }
        if constexpr (OP == PKFMA32) { REP8(asm volatile(I8("v_pk_fma_f32", ", %8, %9") : A8 : "v"(b), "v"(c));) }
        if constexpr (OP == SQRT32) { REP8(asm volatile(I8("v_sqrt_f32", "") : F8);) }
        if constexpr (OP == RCP64) { REP8(asm volatile(I8("v_rcp_f64", "") : A8);) }
        if constexpr (OP == RSQ64) { REP8(asm volatile(I8("v_rsq_f64", "") : A8);) }
        if constexpr (OP == MOV64) { REP8(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n"
                                                       "v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0" : A8);) }
        if constexpr (OP == MOV32) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                                                       "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : U8);) }
        if constexpr (OP == ADDU32) { REP8(asm volatile(I8("v_add_u32", ", %8") : U8 : "v"(m));) }
        if constexpr (OP == XOR32) { REP8(asm volatile(I8("v_xor_b32", ", %8") : U8 : "v"(m));) }
        if constexpr (OP == MULLO) { REP8(asm volatile(I8("v_mul_lo_u32", ", %8") : U8 : "v"(m));) }
        if constexpr (OP == MULHI) { REP8(asm volatile(I8("v_mul_hi_u32", ", %8") : U8 : "v"(m));) }
        if constexpr (OP == MADU64) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %8, %0\n v_mad_u64_u32 %1, vcc, %8, %8, %1\n"
                                                        "v_mad_u64_u32 %2, vcc, %8, %8, %2\n v_mad_u64_u32 %3, vcc, %8, %8, %3\n"
                                                        "v_mad_u64_u32 %4, vcc, %8, %8, %4\n v_mad_u64_u32 %5, vcc, %8, %8, %5\n"
                                                        "v_mad_u64_u32 %6, vcc, %8, %8, %6\n v_mad_u64_u32 %7, vcc, %8, %8, %7"
                                                        : A8 : "v"(m) : "vcc");) }
        // selects with an SGPR-pair mask (VOP3), the form the hit logic uses; compares into SGPR pairs
        if constexpr (OP == CND64E) { REP8(asm volatile(I8("v_cndmask_b32_e64", ", %8, s[20:21]") : U8 : "v"(m));) }
        if constexpr (OP == CMP64) { REP8(asm volatile("v_cmp_lt_f64_e64 s[22:23], %0, %1\n v_cmp_lt_f64_e64 s[24:25], %1, %2\n"
                                                       "v_cmp_lt_f64_e64 s[22:23], %2, %3\n v_cmp_lt_f64_e64 s[24:25], %3, %4\n"
                                                       "v_cmp_lt_f64_e64 s[22:23], %4, %5\n v_cmp_lt_f64_e64 s[24:25], %5, %6\n"
                                                       "v_cmp_lt_f64_e64 s[22:23], %6, %7\n v_cmp_lt_f64_e64 s[24:25], %7, %0"
                                                       :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)
                                                       : "s22", "s23", "s24", "s25");) }
        if constexpr (OP == CVT6432) { REP8(asm volatile("v_cvt_f32_f64 %8, %0\n v_cvt_f64_f32 %1, %9\n v_cvt_f32_f64 %10, %2\n v_cvt_f64_f32 %3, %11\n"
                                                         "v_cvt_f32_f64 %9, %4\n v_cvt_f64_f32 %5, %8\n v_cvt_f32_f64 %11, %6\n v_cvt_f64_f32 %7, %10"
                                                         : A8, "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        // selects in the VOP2 form (mask in VCC, set once), compares into VCC, 32-bit compares, three-operand integer ops
        if constexpr (OP == CND32VCC) { REP8(asm volatile(I8("v_cndmask_b32_e32", ", %8, vcc") : U8 : "v"(m) : );) }
        if constexpr (OP == CMP64VCC) { REP8(asm volatile("v_cmp_lt_f64_e32 vcc, %0, %1\n v_cmp_lt_f64_e32 vcc, %1, %2\n"
                                                          "v_cmp_lt_f64_e32 vcc, %2, %3\n v_cmp_lt_f64_e32 vcc, %3, %4\n"
                                                          "v_cmp_lt_f64_e32 vcc, %4, %5\n v_cmp_lt_f64_e32 vcc, %5, %6\n"
                                                          "v_cmp_lt_f64_e32 vcc, %6, %7\n v_cmp_lt_f64_e32 vcc, %7, %0"
                                                          :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");) }
        if constexpr (OP == CMPU32) { REP8(asm volatile("v_cmp_lt_u32_e64 s[22:23], %0, %1\n v_cmp_lt_u32_e64 s[24:25], %1, %2\n"
                                                        "v_cmp_lt_u32_e64 s[22:23], %2, %3\n v_cmp_lt_u32_e64 s[24:25], %3, %4\n"
                                                        "v_cmp_lt_u32_e64 s[22:23], %4, %5\n v_cmp_lt_u32_e64 s[24:25], %5, %6\n"
                                                        "v_cmp_lt_u32_e64 s[22:23], %6, %7\n v_cmp_lt_u32_e64 s[24:25], %7, %0"
                                                        :: "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7)
                                                        : "s22", "s23", "s24", "s25");) }
        if constexpr (OP == CMPF32) { REP8(asm volatile("v_cmp_lt_f32_e64 s[22:23], %0, %1\n v_cmp_lt_f32_e64 s[24:25], %1, %2\n"
                                                        "v_cmp_lt_f32_e64 s[22:23], %2, %3\n v_cmp_lt_f32_e64 s[24:25], %3, %4\n"
                                                        "v_cmp_lt_f32_e64 s[22:23], %4, %5\n v_cmp_lt_f32_e64 s[24:25], %5, %6\n"
                                                        "v_cmp_lt_f32_e64 s[22:23], %6, %7\n v_cmp_lt_f32_e64 s[24:25], %7, %0"
                                                        :: "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7)
                                                        : "s22", "s23", "s24", "s25");) }
        if constexpr (OP == MIN3U32) { REP8(asm volatile(I8("v_min3_u32", ", %8, %8") : U8 : "v"(m));) }
        if constexpr (OP == BFI32) { REP8(asm volatile(I8("v_bfi_b32", ", %8, %8") : U8 : "v"(m));) }
        if constexpr (OP == LSHLADD) { REP8(asm volatile(I8("v_lshl_add_u32", ", 3, %8") : U8 : "v"(m));) }
        if constexpr (OP == FFBL) { REP8(asm volatile(I8("v_ffbl_b32", "") : U8);) }
        if constexpr (OP == MAXF32) { REP8(asm volatile(I8("v_max_f32", ", %8") : F8 : "v"(g));) }
        if constexpr (OP == RSQ32) { REP8(asm volatile(I8("v_rsq_f32", "") : F8);) }
        // fp64 add with a scalar (SGPR pair) operand, the form the sphere chunk's p = c - o takes
        if constexpr (OP == ADD64S) { REP8(asm volatile(I8("v_add_f64", ", s[20:21]") : A8);) }
        // the hit logic's pattern: compare into an SGPR pair, two selects on it
        if constexpr (OP == CNDCMP) { REP8(asm volatile("v_cmp_lt_f64_e64 s[22:23], %8, %9\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]\n"
                                                        "v_cndmask_b32_e64 %2, %2, %3, s[22:23]\n v_cmp_lt_f64_e64 s[24:25], %9, %8\n"
                                                        "v_cndmask_b32_e64 %4, %4, %5, s[24:25]\n v_cndmask_b32_e64 %6, %6, %7, s[24:25]\n"
                                                        "v_cndmask_b32_e64 %1, %1, %0, s[22:23]\n v_cndmask_b32_e64 %3, %3, %2, s[24:25]"
                                                        : U8 : "v"(a0), "v"(a1) : "s22", "s23", "s24", "s25");) }
        if constexpr (OP == NOPS) { REP8(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");) }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] =
        a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ticks[0] = t1 - t0;
        ticks[1] = r1 - r0;
    }
}

struct Result {
    std::string name;
    int waves;
    double ms, inst_per_s_per_simd, ns_per_inst, tflops, memtime_hz, memrealtime_hz;
};

static int g_cus = 256;

template <int OP>
Result run(const char* name, int waves_per_simd, double flops_per_lane_inst, int lanes_x, double* out, unsigned long long* ticks) {
    const int blocks = g_cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int iters = 2000;
    float ms = 0;
    for (int attempt = 0; attempt < 6; ++attempt) {  // grow until the launch is >= 50 ms
        peak_kernel<OP><<<blocks, 256>>>(out, ticks, iters / 10 + 1);  // warm
        CHECK(hipEventRecord(e0));
        peak_kernel<OP><<<blocks, 256>>>(out, ticks, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms >= 50.f) break;
        iters = (int)(iters * (60.0 / (ms > 0.05f ? ms : 0.05f))) + 1;
    }
    unsigned long long h[2];
    CHECK(hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost));
    const double s = ms * 1e-3;
    const double inst_per_wave = (double)iters * 64.0;
    const double simds = g_cus * 4.0;
    Result r;
    r.name = name;
    r.waves = waves_per_simd;
    r.ms = ms;
    r.inst_per_s_per_simd = inst_per_wave * waves_per_simd / s;
    r.ns_per_inst = 1e9 / r.inst_per_s_per_simd;
    r.tflops = r.inst_per_s_per_simd * simds * 64.0 * lanes_x * flops_per_lane_inst / 1e12;
    // wave 0's own interval is (nearly) the kernel's: every wave runs the same loop
    r.memtime_hz = h[0] / s;
    r.memrealtime_hz = h[1] / s;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return r;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    double* out;
    unsigned long long* ticks;
    CHECK(hipMalloc(&out, (size_t)g_cus * 8 * 256 * sizeof(double)));
    CHECK(hipMalloc(&ticks, 16));
    std::printf("device: %s, %d CUs, clockRate %.0f MHz (reported max)\n", prop.name, g_cus, prop.clockRate / 1e3);
    std::vector<Result> rs;
    for (int w : {4, 8}) {
        rs.push_back(run<FMA64>("v_fma_f64", w, 2, 1, out, ticks));
        rs.push_back(run<ADD64>("v_add_f64", w, 1, 1, out, ticks));
        rs.push_back(run<MUL64>("v_mul_f64", w, 1, 1, out, ticks));
        rs.push_back(run<FMA32>("v_fma_f32", w, 2, 1, out, ticks));
        rs.push_back(run<PKFMA32>("v_pk_fma_f32", w, 2, 2, out, ticks));
    }
    const int w = 4;  // the render kernel's occupancy
    rs.push_back(run<MIN64>("v_min_f64", w, 0, 1, out, ticks));
    rs.push_back(run<LDEXP64>("v_ldexp_f64", w, 0, 1, out, ticks));
    rs.push_back(run<RCP64>("v_rcp_f64", w, 0, 1, out, ticks));
    rs.push_back(run<RSQ64>("v_rsq_f64", w, 0, 1, out, ticks));
    rs.push_back(run<CMP64>("v_cmp_lt_f64_e64 (sgpr pair)", w, 0, 1, out, ticks));
    rs.push_back(run<CND64E>("v_cndmask_b32_e64 (sgpr mask)", w, 0, 1, out, ticks));
    rs.push_back(run<CVT6432>("v_cvt f32<->f64", w, 0, 1, out, ticks));
    rs.push_back(run<MOV32>("v_mov_b32", w, 0, 1, out, ticks));
    rs.push_back(run<MOV64>("v_mov_b64", w, 0, 1, out, ticks));
    rs.push_back(run<ADDU32>("v_add_u32", w, 0, 1, out, ticks));
    rs.push_back(run<XOR32>("v_xor_b32", w, 0, 1, out, ticks));
    rs.push_back(run<MULLO>("v_mul_lo_u32", w, 0, 1, out, ticks));
    rs.push_back(run<MULHI>("v_mul_hi_u32", w, 0, 1, out, ticks));
    rs.push_back(run<MADU64>("v_mad_u64_u32", w, 0, 1, out, ticks));
    rs.push_back(run<SQRT32>("v_sqrt_f32", w, 0, 1, out, ticks));
    rs.push_back(run<CND32VCC>("v_cndmask_b32_e32 (vcc mask)", w, 0, 1, out, ticks));
    rs.push_back(run<CNDCMP>("2 v_cmp_f64 + 6 v_cndmask_e64 (per 8)", w, 0, 1, out, ticks));
    rs.push_back(run<CMP64VCC>("v_cmp_lt_f64_e32 (vcc)", w, 0, 1, out, ticks));
    rs.push_back(run<CMPU32>("v_cmp_lt_u32_e64 (sgpr pair)", w, 0, 1, out, ticks));
    rs.push_back(run<CMPF32>("v_cmp_lt_f32_e64 (sgpr pair)", w, 0, 1, out, ticks));
    rs.push_back(run<MIN3U32>("v_min3_u32", w, 0, 1, out, ticks));
    rs.push_back(run<BFI32>("v_bfi_b32", w, 0, 1, out, ticks));
    rs.push_back(run<LSHLADD>("v_lshl_add_u32", w, 0, 1, out, ticks));
    rs.push_back(run<FFBL>("v_ffbl_b32", w, 0, 1, out, ticks));
    rs.push_back(run<MAXF32>("v_max_f32", w, 0, 1, out, ticks));
    rs.push_back(run<RSQ32>("v_rsq_f32", w, 0, 1, out, ticks));
    rs.push_back(run<ADD64S>("v_add_f64 (sgpr pair operand)", w, 0, 1, out, ticks));
    rs.push_back(run<NOPS>("s_nop 0", w, 0, 1, out, ticks));
    const double fma_ns = rs[0].ns_per_inst;
    std::printf("%-32s %5s %8s %14s %9s %8s %9s %12s %12s\n", "instruction", "w/SIMD", "ms", "inst/s/SIMD", "ns/inst", "rel fma64",
                "TFLOP/s", "s_memtime Hz", "realtime Hz");
    for (const Result& r : rs)
        std::printf("%-32s %5d %8.2f %14.4e %9.4f %8.3f %9.2f %12.4e %12.4e\n", r.name.c_str(), r.waves, r.ms, r.inst_per_s_per_simd,
                    r.ns_per_inst, r.ns_per_inst / fma_ns, r.tflops, r.memtime_hz, r.memrealtime_hz);
    std::printf("\nfp64 FMA peak by wall clock: %.2f TFLOP/s at 4 waves/SIMD, %.2f at 8 (datasheet 78.6); if a wave64 fp64 FMA\n"
                "occupies its SIMD for 4 cycles, the sustained clock under this load is %.3f GHz; s_memtime ticks at %.4f GHz,\n"
                "i.e. %.3f ticks per such instruction.\n",
                rs[0].tflops, rs[5].tflops, rs[0].inst_per_s_per_simd * 4 / 1e9, rs[0].memtime_hz / 1e9,
                rs[0].memtime_hz / rs[0].inst_per_s_per_simd);
    if (argc > 1) {
        FILE* f = std::fopen(argv[1], "w");
        if (!f) return 1;
        std::fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"fp64_fma_tflops_4_waves\": %.4f, \"fp64_fma_tflops_8_waves\": %.4f,\n"
                        " \"s_memtime_hz\": %.6e, \"s_memrealtime_hz\": %.6e, \"rows\": [\n", prop.name, g_cus, rs[0].tflops, rs[5].tflops,
                     rs[0].memtime_hz, rs[0].memrealtime_hz);
        for (size_t i = 0; i < rs.size(); ++i)
            std::fprintf(f, "  {\"instruction\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"inst_per_s_per_simd\": %.6e, "
                            "\"ns_per_inst\": %.5f, \"rel_fma64\": %.4f, \"tflops\": %.3f}%s\n",
                         rs[i].name.c_str(), rs[i].waves, rs[i].ms, rs[i].inst_per_s_per_simd, rs[i].ns_per_inst,
                         rs[i].ns_per_inst / fma_ns, rs[i].tflops, i + 1 < rs.size() ? "," : "");
        std::fprintf(f, "]}\n");
        std::fclose(f);
    }
    return 0;
}
