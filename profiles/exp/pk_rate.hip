// Issue rate of v_pk_fma_f32 by operand form on gfx950 (hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned long long* sp, f2* out, int iters) {
    typedef const __attribute__((address_space(4))) unsigned long long* CP;
    CP q = (CP)(unsigned long long)sp;
    const unsigned long long s0 = q[0], s1 = q[1];
    f2 a = {1.f + threadIdx.x, 2.f}, b = {0.5f, 0.25f}, c = {3.f, 4.f}, d = {5.f, 6.f}, v = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // VGPR operands only
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %0, %0\n v_pk_fma_f32 %1, %4, %1, %1\n v_pk_fma_f32 %2, %4, %2, %2\n v_pk_fma_f32 %3, %4, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(v));)
        } else if (MODE == 1) {  // SGPR pair, plain halves (two spheres per instruction)
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %0, %0\n v_pk_fma_f32 %1, %5, %1, %1\n v_pk_fma_f32 %2, %4, %2, %2\n v_pk_fma_f32 %3, %5, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0), "s"(s1));)
        } else if (MODE == 2) {  // SGPR pair, low half broadcast
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %0, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %5, %1, %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %4, %2, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %5, %3, %3 op_sel_hi:[0,1,1]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0), "s"(s1));)
        } else if (MODE == 3) {  // SGPR pair, high half broadcast
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %0, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %5, %1, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %2, %4, %2, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %5, %3, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0), "s"(s1));)
        } else if (MODE == 4) {  // the same SGPR pair as multiplicand (low) and addend (high)
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %1, %4 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %5, %2, %5 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %4, %3, %4 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %5, %0, %5 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0), "s"(s1));)
        } else if (MODE == 5) {  // non-packed v_fma_f32 with an SGPR
            REP8(asm volatile("v_fma_f32 %0, %4, %0, %0\n v_fma_f32 %1, %4, %1, %1\n v_fma_f32 %2, %4, %2, %2\n v_fma_f32 %3, %4, %3, %3" : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "s"((unsigned)s0));)
        } else if (MODE == 6) {  // v_mov_b64 from an SGPR pair
            REP8(asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %5\n v_mov_b64 %2, %4\n v_mov_b64 %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0), "s"(s1));)
        } else if (MODE == 7) {  // dependent chain: every instruction needs the one before it
            REP8(asm volatile("v_pk_fma_f32 %0, %1, %0, %0\n v_pk_fma_f32 %0, %1, %0, %0\n v_pk_fma_f32 %0, %1, %0, %0\n v_pk_fma_f32 %0, %1, %0, %0" : "+v"(a) : "v"(v));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
template <int MODE>
static void run(const char* name, const unsigned long long* sp, f2* out) {
    const int blocks = 256 * 8, iters = 4000;  // 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(sp, out, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(sp, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters * 32;  // wave-instructions
    // 1024 SIMDs; cycles per wave-instruction per SIMD at 2.4 GHz
    printf("%-64s %8.3f ms  %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / instr);
}
int main() {
    unsigned long long h[2] = {0x3f8000013f7fffffull, 0x3f8000023f7ffffeull}, *sp; f2* out;
    hipMalloc(&sp, 16); hipMemcpy(sp, h, 16, hipMemcpyHostToDevice); hipMalloc(&out, 256 * 8 * 256 * sizeof(f2));
    run<0>("v_pk_fma_f32 v, v, v, v", sp, out);
    run<1>("v_pk_fma_f32 v, s[pair], v, v", sp, out);
    run<2>("v_pk_fma_f32 v, s[pair] (low half to both), v, v", sp, out);
    run<3>("v_pk_fma_f32 v, s[pair] (high half to both), v, v", sp, out);
    run<4>("v_pk_fma_f32 v, s[pair].lo, v, s[same pair].hi", sp, out);
    run<5>("v_fma_f32 v, s, v, v", sp, out);
    run<6>("v_mov_b64 v, s[pair]", sp, out);
    run<7>("v_pk_fma_f32 dependent chain", sp, out);
    return 0;
}
