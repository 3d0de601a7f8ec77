// rtm_device.h — device-side math of the path-tracing hot path, fp64 with the reference's float
// islands.  Compiled with -ffp-contract=off: every a*b+c below is a separate multiply and add, in
// the operand order of the reference (citations: file:line in the reference checkout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The kernel headers (this one, rtm_path.h, rtm_render_kernel.h) are compiled twice: into namespace rtm by rtm_kernels.hip
// (-ffp-contract=off, correctly rounded division and square root sequences: the bit-exact kernels) and into namespace
// rtm_tol by rtm_kernels_tol.hip (-DRTM_TOL=1, FMA contraction on, division and square root to about one ulp: the
// LABELLED tolerance row, variant 18 — same loop nest, RNG, thresholds, float islands and order of additions).
#ifndef RTM_NS
#define RTM_NS rtm
#endif
#ifndef RTM_TOL
#define RTM_TOL 0
#endif

namespace RTM_NS {

struct D3 {
    double x, y, z;
};

__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
// src/Ray.h:15-35
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) { return D3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ D3 operator/(D3 a, double s) { return D3{a.x / s, a.y / s, a.z / s}; }

// One level of the recursion's unwinding, L = colorKD * L_next + emission (src/Renderer.cpp:109), as a separate multiply
// and add WHATEVER the translation unit's contraction mode: a sample's radiance — and with it the image — is then a
// function of the path's hit ids alone in the tolerance build too (rtm_kernels_tol.hip), bit for bit.
__device__ __forceinline__ D3 fold_step(const D3 c, const D3 L, const D3 e) {
#pragma clang fp contract(off)
    return D3{c.x * L.x + e.x, c.y * L.y + e.y, c.z * L.z + e.z};
}
// ... and the same level where the emission added is known to be (+0, +0, +0) and the product cannot be -0 (SceneView::
// fold_flags, kFoldNoLevelEmission): x + (+0) == x bit for bit for every x but -0, so the addition is left out
__device__ __forceinline__ D3 fold_step_mul(const D3 c, const D3 L) { return D3{c.x * L.x, c.y * L.y, c.z * L.z}; }
// src/Ray.h:61-63
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// src/Ray.h:64-66 (middle component is (-a.x)*b.z + a.z*b.x)
__device__ __forceinline__ D3 cross(D3 a, D3 b) {
    return D3{a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x};
}
// src/Ray.h:67-69: std::sqrtf of a double => round to float, correctly rounded float sqrt, widen.
// hipcc's sqrtf is the correctly rounded expansion by default
// (-fhip-fp32-correctly-rounded-divide-sqrt); checked exhaustively by tests/test_device_math.py.
__device__ __forceinline__ double magnitude(D3 a) {
    const float len2 = (float)(a.x * a.x + a.y * a.y + a.z * a.z);
    return (double)__builtin_sqrtf(len2);
}
// The same float square root without hipcc's denormal pre-scaling and zero/inf fix-up: exact
// (correctly rounded) for x in [2^-96, FLT_MAX]; callers flag anything else (sqrtf_fast_ok).
// v_sqrt_f32 is within 1 ulp; the two fma residuals pick the neighbour when it is the rounded one —
// the compiler's own correction step.  Checked against sqrtf for EVERY float in that range
// (rtm_debug_selfcheck, tests/test_parity_gpu.py::test_fast_sqrtf_exhaustive).
__device__ __forceinline__ bool sqrtf_fast_ok(float x) {
    return (__float_as_uint(x) - 0x0F800000u) < (0x7F800000u - 0x0F800000u);  // 2^-96 <= x < +inf
}
__device__ __forceinline__ float sqrtf_fast(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float r = (r_dn <= 0.0f) ? s_dn : s;
    r = (r_up > 0.0f) ? s_up : r;
    return r;
}
// src/Ray.h:70-72: three true divisions
__device__ __forceinline__ D3 normalize(D3 a) { return a / magnitude(a); }

// ---- selects ----
// gfx950 issues back-to-back v_cndmask_b32_e32 (the VCC form hipcc prefers for a?b:c on doubles)
// at ~11-13 cycles each (profiles/ubench/valu_rate.hip; an isolated one, or the e64 form, costs ~3).
// A per-lane all-ones/zero mask made once per condition and v_bfi_b32 per dword is full rate.
// The empty asm keeps LLVM from folding the bit-select back into a select.
struct LaneMask {
    uint32_t m;
};
__device__ __forceinline__ LaneMask lane_mask(bool c) {
    uint32_t m = c ? 0xFFFFFFFFu : 0u;
    asm("" : "+v"(m));
    return LaneMask{m};
}
__device__ __forceinline__ uint32_t sel_u32(LaneMask k, uint32_t a, uint32_t b) {
    return (a & k.m) | (b & ~k.m);  // v_bfi_b32
}
__device__ __forceinline__ int sel_i32(LaneMask k, int a, int b) {
    return (int)sel_u32(k, (uint32_t)a, (uint32_t)b);
}
__device__ __forceinline__ double sel_f64(LaneMask k, double a, double b) {  // k ? a : b, bit-exact
    const uint32_t hi = sel_u32(k, (uint32_t)__double2hiint(a), (uint32_t)__double2hiint(b));
    const uint32_t lo = sel_u32(k, (uint32_t)__double2loint(a), (uint32_t)__double2loint(b));
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ D3 sel_d3(LaneMask k, D3 a, D3 b) {
    return D3{sel_f64(k, a.x, b.x), sel_f64(k, a.y, b.y), sel_f64(k, a.z, b.z)};
}
// x * -1.0 for the lanes of k, x elsewhere: IEEE multiplication by -1 is a sign-bit flip
__device__ __forceinline__ double negate_where(LaneMask k, double x) {
    return __hiloint2double(__double2hiint(x) ^ (int)(k.m & 0x80000000u), __double2loint(x));
}

// k ? a : b with the condition as a wave mask in an SGPR pair (what v_cmp / ballot produce) and the VOP3
// encoding spelled out: hipcc's own a ? b : c on doubles becomes back-to-back v_cndmask_b32_e32 reading VCC,
// the slow form above (13 vs 3 cycles each in the microbenchmark).
__device__ __forceinline__ uint32_t sel_u32_mask(unsigned long long m, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ double sel_f64_mask(unsigned long long m, double a, double b) {
    const uint32_t hi = sel_u32_mask(m, (uint32_t)__double2hiint(a), (uint32_t)__double2hiint(b));
    const uint32_t lo = sel_u32_mask(m, (uint32_t)__double2loint(a), (uint32_t)__double2loint(b));
    return __hiloint2double((int)hi, (int)lo);
}

// k ? index : old for a small non-negative index: as an inline constant of the VOP3 encoding when the
// compiler knows it (the unrolled tail chunk of a scene under 8 spheres), else from a register.
__device__ __forceinline__ int sel_index_mask(unsigned long long m, int index, int old) {
    int r;
#define RTM_SEL_INDEX_CASE(i) \
    case i: asm("v_cndmask_b32_e64 %0, %1, " #i ", %2" : "=v"(r) : "v"(old), "s"(m)); return r;
    if (__builtin_constant_p(index)) {
        switch (index) {
            RTM_SEL_INDEX_CASE(0) RTM_SEL_INDEX_CASE(1) RTM_SEL_INDEX_CASE(2) RTM_SEL_INDEX_CASE(3)
            RTM_SEL_INDEX_CASE(4) RTM_SEL_INDEX_CASE(5) RTM_SEL_INDEX_CASE(6) RTM_SEL_INDEX_CASE(7)
            default: break;
        }
    }
#undef RTM_SEL_INDEX_CASE
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(old), "v"(index), "s"(m));
    return r;
}

// ---- sin and cos of r in [0, 2^30) without branches ----
// The same operation sequence as the small-argument path of ROCm's ocml sincos (three-part pi/2
// Cody-Waite reduction with FMA, then the (hi, lo) sin/cos kernels), written out so that the
// shading block stays one basic block (ocml's entry point branches to a Payne-Hanek path the
// renderer's r1 = 2*pi*u can never take).  Bit-identical to ::sincos on that range
// (tests/test_parity_gpu.py::test_sincos_small_matches_ocml, all 2^23 possible r1).
// Constants of sincos_small in the order the LDS copy (TrigTable) holds them.
struct TrigConsts {
    double two_over_pi, pio2_h, pio2_m, pio2_t;
    double c5, c4, c3, c2, c1, c0;   // cos kernel, highest degree first
    double s4, s3, s2, s1, s0, third;  // sin kernel, highest degree first; third = -1/6
};
constexpr int kTrigConstCount = 16;
// Behind them in the same LDS block: the Normalize of a vector whose squared length, as a float, is within a few ulps of 1
// (a bounce direction: src/Renderer.cpp:103-107) — kUnitWindow float bit patterns from kUnitWindowFirst on, for each the
// magnitude (double)sqrtf(len2) and its refined reciprocal as the policies' own sequences give them (fill_shade_consts).
constexpr int kUnitWindow = 16;
constexpr uint32_t kUnitWindowFirst = 0x3F7FFFF8u;  // 1.0f - 8 ulps ... 1.0f + 7 ulps (observed over 4e6 bounces: -4 .. +2)
constexpr int kShadeConstCount = kTrigConstCount + 2 * kUnitWindow;  // doubles of LDS a shading kernel sets aside
__host__ __device__ __forceinline__ constexpr TrigConsts trig_consts() {
    return TrigConsts{0x1.45f306dc9c883p-1, 0x1.921fb54442d18p+0, 0x1.1a62633145c00p-54, 0x1.b839a252049c0p-104,
                      -0x1.907db46cc5e42p-37, 0x1.1eeb69037ab78p-29, -0x1.27e4fa17f65f6p-22, 0x1.a01a019f4ec90p-16,
                      -0x1.6c16c16c16967p-10, 0x1.5555555555555p-5,
                      0x1.5e0b2f9a43bb8p-33, -0x1.ae600b42fdfa7p-26, 0x1.71de3796cde01p-19, -0x1.a01a019e83e5cp-13,
                      0x1.1111111110bb3p-7, -0x1.5555555555555p-3};
}
// K provides the constants: TrigConsts by value (immediates / registers) or a pointer-backed reader
// (LDS copy).  hipcc hoists fp64 literals out of the render loop into VGPRs and then spills them
// to scratch; reading them from LDS where they are used keeps them out of the register file.
struct TrigFromRegs {
    __device__ __forceinline__ double operator[](int i) const {
        constexpr TrigConsts k = trig_consts();
        const double t[kTrigConstCount] = {k.two_over_pi, k.pio2_h, k.pio2_m, k.pio2_t, k.c5, k.c4, k.c3, k.c2,
                                           k.c1, k.c0, k.s4, k.s3, k.s2, k.s1, k.s0, k.third};
        return t[i];
    }
};
// What a shading kernel hands the math policies: its LDS copy of the sincos constants and, behind it when the kernel has
// room for one (the any-depth kernels run at 10 072 bytes of LDS per wave: 16 waves per CU end at 10 240), the near-unit
// Normalize table.  A bare pointer converts to "constants only".
struct ShadeLds {
    const double* trig = nullptr;
    const double* unit = nullptr;
    __device__ __forceinline__ ShadeLds() {}
    __device__ __forceinline__ ShadeLds(const double* t, bool with_unit = false) : trig(t), unit(with_unit && t ? t + kTrigConstCount : nullptr) {}
};
struct TrigFromLds {
    const double* p;
    __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};
template <class K>
__device__ __forceinline__ void sincos_small_k(const K& k, const double x, double& sn, double& cs) {
    // trigredsmall
    const double pio2_m = k[2];
    const double dn = __builtin_rint(x * k[0]);
    const double xt = __builtin_fma(dn, -k[1], x);
    const double yt = __builtin_fma(dn, -pio2_m, xt);
    const double ph = dn * pio2_m;
    const double pt = __builtin_fma(dn, pio2_m, -ph);
    const double th = xt - ph;
    const double tt = (xt - th) - ph;
    const double c = ((th - yt) + tt) - pt;
    const double d = __builtin_fma(dn, -k[3], c);
    const double hi = yt + d;
    const double lo = d - (hi - yt);
    const int q = (int)dn;
    // sincosred2(hi, lo)
    const double t = hi * hi;
    const double h = t * 0.5;
    const double c1 = 1.0 - h;
    const double c3 = (1.0 - c1) - h;
    const double t2 = t * t;
    double p = __builtin_fma(t, k[4], k[5]);
    p = __builtin_fma(t, p, k[6]);
    p = __builtin_fma(t, p, k[7]);
    p = __builtin_fma(t, p, k[8]);
    p = __builtin_fma(t, p, k[9]);
    const double nlo = -lo;
    const double cq = __builtin_fma(t2, p, __builtin_fma(hi, nlo, c3));
    const double cosv = c1 + cq;
    double s = __builtin_fma(t, k[10], k[11]);
    s = __builtin_fma(t, s, k[12]);
    s = __builtin_fma(t, s, k[13]);
    s = __builtin_fma(t, s, k[14]);
    const double v = hi * -t;
    const double w = __builtin_fma(t, __builtin_fma(v, s, lo * 0.5), nlo);
    const double sinv = hi - __builtin_fma(v, k[15], w);
    // quadrant: sin <- (q odd ? cos : sin), cos <- (q odd ? -sin : cos); both negated for q & 2
    // (bit selects: four v_cndmask_b32_e32 back to back, what hipcc makes of ?: on doubles, cost three times as much — LaneMask)
    const LaneMask odd = lane_mask((q & 1) != 0);
    const int flip = (q & 2) ? (int)0x80000000 : 0;
    const double so = sel_f64(odd, cosv, sinv);
    const double co = sel_f64(odd, -sinv, cosv);
    sn = __hiloint2double(__double2hiint(so) ^ flip, __double2loint(so));
    cs = __hiloint2double(__double2hiint(co) ^ flip, __double2loint(co));
}
__device__ __forceinline__ void sincos_small(const double x, double& sn, double& cs) {
    sincos_small_k(TrigFromRegs{}, x, sn, cs);
}
#if RTM_TOL
// The tolerance unit's sin / cos of r1 = 2 pi (m 2^-24), m the draw's 24-bit integer (src/Renderer.cpp:88,93-94), without the
// Cody-Waite reduction: the quadrant and the offset from it are EXACT in the draw's own units — q = rint(m 2^-22),
// f = m 2^-22 - q in [-1/2, 1/2] — so the reduced argument is one rounded product f (pi/2) (relative error 2^-53; the
// reference's own r1 carries 2^-53 of up to 2 pi) and the kernels run without their low word.  Same polynomials as above:
// within 8e-16 of the exact unit's sin / cos of the ROUNDED r1 over all 2^23 draws (tests/test_tolerance_gpu.py; half an ulp
// of r1 in [4, 8) is 4.4e-16 of that); 14 instructions fewer.
template <class K>
__device__ __forceinline__ void sincos_turn24_k(const K& k, const double m24, double& sn, double& cs) {
    const double x4 = m24 * 0x1p-22;
    const double dn = __builtin_rint(x4);
    const double hi = (x4 - dn) * k[1];
    const int q = (int)dn;
    const double t = hi * hi;
    const double h = t * 0.5;
    const double c1 = 1.0 - h;
    const double c3 = (1.0 - c1) - h;
    double p = __builtin_fma(t, k[4], k[5]);
    p = __builtin_fma(t, p, k[6]);
    p = __builtin_fma(t, p, k[7]);
    p = __builtin_fma(t, p, k[8]);
    p = __builtin_fma(t, p, k[9]);
    const double cosv = c1 + __builtin_fma(t * t, p, c3);
    double s = __builtin_fma(t, k[10], k[11]);
    s = __builtin_fma(t, s, k[12]);
    s = __builtin_fma(t, s, k[13]);
    s = __builtin_fma(t, s, k[14]);
    const double v = hi * t;  // sin = hi - v (1/6 - t s)
    const double sinv = __builtin_fma(v, __builtin_fma(t, s, k[15]), hi);
    // (bit selects: four v_cndmask_b32_e32 back to back, what hipcc makes of ?: on doubles, cost three times as much — LaneMask)
    const LaneMask odd = lane_mask((q & 1) != 0);
    const int flip = (q & 2) ? (int)0x80000000 : 0;
    const double so = sel_f64(odd, cosv, sinv);
    const double co = sel_f64(odd, -sinv, cosv);
    sn = __hiloint2double(__double2hiint(so) ^ flip, __double2loint(so));
    cs = __hiloint2double(__double2hiint(co) ^ flip, __double2loint(co));
}
#endif

// ---- build-defined counter RNG; must agree with rtm_rng_u01 on the host ----
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
__host__ __device__ __forceinline__ uint64_t smfin64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t seed_multiplier(uint64_t seed) {
    return smfin64(seed + 0x9E3779B97F4A7C15ull) | 1ull;
}
// Keying: one 64-bit mix per PIXEL, one 32-bit mix per SAMPLE, one 32-bit mix per DRAW.
//   pixel key   (p0, p1) = halves of smfin64((pixel + 1) * seed_mult)
//   sample key  k0 = mix32(p0 + sample * 0x9E3779B9), k1 = mix32(p1 ^ (sample * 0x85EBCA6B))
//   draw j      bits = mix32((k0 + j * 0x9E3779B9) ^ k1);  u = (2 * (bits >> 9) + 1) * 2^-24
struct RngPixelKey {
    uint32_t p0, p1;
};
struct RngStream {
    uint32_t ctr, k1;  // ctr = k0 + index * 0x9E3779B9
};
__host__ __device__ __forceinline__ RngPixelKey rng_pixel_key(uint64_t seed_mult, uint32_t pixel) {
    const uint64_t z = smfin64(((uint64_t)pixel + 1ull) * seed_mult);
    return RngPixelKey{(uint32_t)z, (uint32_t)(z >> 32)};
}
__host__ __device__ __forceinline__ RngStream rng_open(RngPixelKey pk, uint32_t sample) {
    return RngStream{mix32(pk.p0 + sample * 0x9E3779B9u), mix32(pk.p1 ^ (sample * 0x85EBCA6Bu))};
}
__host__ __device__ __forceinline__ double rng_bits_to_u01(uint32_t x) {
    // 23 random bits -> odd multiple of 2^-24: never 0 or 1, exact in fp32 and fp64
    return (double)(2u * (x >> 9) + 1u) * (1.0 / 16777216.0);
}
__host__ __device__ __forceinline__ double rng_next(RngStream& s) {
    const uint32_t x = mix32(s.ctr ^ s.k1);
    s.ctr += 0x9E3779B9u;
    return rng_bits_to_u01(x);
}
// The draw as the odd integer m = 2 (bits >> 9) + 1 < 2^24 in a double: u = m * 2^-24.  Callers that only compare u
// with a constant, or scale it by one, fold the 2^-24 into the constant (exact: a power of two) and save the ldexp.
__host__ __device__ __forceinline__ double rng_next_m(RngStream& s) {
    const uint32_t x = mix32(s.ctr ^ s.k1);
    s.ctr += 0x9E3779B9u;
    return (double)(2u * (x >> 9) + 1u);
}
__host__ __device__ __forceinline__ double rng_u01_at(uint64_t seed_mult, uint32_t pixel, uint32_t sample,
                                                      uint32_t index) {
    RngStream s = rng_open(rng_pixel_key(seed_mult, pixel), sample);
    s.ctr += index * 0x9E3779B9u;
    return rng_next(s);
}

}  // namespace RTM_NS
