// rtm_path.h — the device-side path: SphereObject::Intersect, the nearest-hit loop, one
// PathTracing invocation and the back-to-front fold, written once and parameterised by
//   * a Math policy  (how sqrt / x/y are evaluated — every policy returns IEEE-correct bits), and
//   * a Scene policy (where sphere geometry and materials are read from).
// Citations are file:line in the reference checkout.
#pragma once
#include <cfloat>
#include <type_traits>

// A/B switches of the round-2 instruction trimming (profiles/r2/ab_*.txt); all on by default.
#ifndef RTM_OPT_GUARD
#define RTM_OPT_GUARD 1
#endif
#ifndef RTM_OPT_SEL64
#define RTM_OPT_SEL64 1
#endif
#ifndef RTM_OPT_TRIGLOAD
#define RTM_OPT_TRIGLOAD 1
#endif
#ifndef RTM_OPT_ONB
#define RTM_OPT_ONB RTM_OPT_GUARD  // rides on the guards' "no zero component" guarantee
#endif

#ifndef RTM_OPT_AXIS
#define RTM_OPT_AXIS 1  // axis-signature instantiations of the exact-n kernels (sphere_disc; A/B switch)
#endif
#ifndef RTM_OPT_CTMODE
#define RTM_OPT_CTMODE 1  // what an instantiation knows at compile time: its mode (axis signatures), "no planes" (SceneLds) ...
#endif
#ifndef RTM_TOL_LIGHT_ROOTS
#define RTM_TOL_LIGHT_ROOTS 1  // the tolerance unit's search roots without the residual step in compact scenes (A/B switch)
#endif
#ifndef RTM_TOL_PACKID
#define RTM_TOL_PACKID 1  // (with the light roots: the sphere's index in the near root's last three bits — accept_batch; A/B switch)
#endif
#ifndef RTM_OPT_FOLDMUL
#define RTM_OPT_FOLDMUL 1  // the packed folds leave a bounce level's "+ (+0, +0, +0)" out (SceneView::fold_flags; A/B switch)
#endif
#ifndef RTM_OPT_CTN
#define RTM_OPT_CTN 0     // ... and (NOT kept: profiles/r4/ctn_ab.txt — nothing for the tolerance row, +3 % for the exact kernel,
#endif                    // whose register allocation it upsets) its sphere count in the exact-n instantiations; A/B switches

#include "../../include/rtm.h"
#include "rtm_device.h"

namespace RTM_NS {

// ------------------------------------------------------------------------------------------------
// Math policies
// ------------------------------------------------------------------------------------------------

// The UNSCALED division and square-root sequences every fast policy below is made of.
// RTM_TOL == 0 (the bit-exact kernels): exactly the instructions hipcc's correctly rounded expansions keep when no
// operand needs rescaling — v_rcp_f64 + two Newton steps, quotient, remainder, correction; v_rsq_f64 + one coupled
// iteration + two residual corrections.
// RTM_TOL == 1 (the labelled tolerance row, rtm_kernels_tol.hip): v_rcp_f64 / v_rsq_f64 are good to 2^-23, so
//   1/y  = r0 (1 + e + e^2), e = 1 - y r0         relative error e^3 = 2^-69 + two roundings: within one ulp;
//   x/y  = x * (1/y)                               one more rounding: within about 1.5 ulp (no remainder step);
//   sqrt = s1 + (x - s1^2) h0, s1 = s0 + s0 r0     s1 is good to 1.5 e^2 = 2^-45, the residual step to 2^-68 + half an ulp
// — 4 + 1 instead of 5 + 3 instructions for a quotient, 7 instead of 10 for a root.
__device__ __forceinline__ double seq_rcp(const double y) {
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
#if RTM_TOL
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
#else
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    return __builtin_fma(r, e, r);
#endif
}
// x / y given r = seq_rcp(y), before any v_div_fixup
__device__ __forceinline__ double seq_quot(const double x, const double y, const double r) {
#if RTM_TOL
    (void)y;
    return x * r;
#else
    const double q = x * r;
    const double rem = __builtin_fma(-y, q, x);
    return __builtin_fma(rem, r, q);
#endif
}
__device__ __forceinline__ double seq_sqrt(const double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y, h0 = 0.5 * y;
    const double r0 = __builtin_fma(-h0, s0, 0.5);
#if RTM_TOL
    const double s1 = __builtin_fma(s0, r0, s0);
    const double d0 = __builtin_fma(-s1, s1, x);
    return __builtin_fma(d0, h0, s1);
#else
    const double s1 = __builtin_fma(s0, r0, s0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-s1, s1, x);
    const double s2 = __builtin_fma(d0, h1, s1);
    const double d1 = __builtin_fma(-s2, s2, x);
    return __builtin_fma(d1, h1, s2);
#endif
}
// K independent roots, stage by stage (K independent dependency chains for the scheduler to interleave)
template <int K, bool LIGHT = false>
__device__ __forceinline__ void seq_sqrt_batch(const double (&x)[K], double (&out)[K]) {
#if RTM_TOL
    double y[K], s0[K], h0[K], r0[K], s1[K], d0[K];
#pragma unroll
    for (int k = 0; k < K; ++k) y[k] = __builtin_amdgcn_rsq(x[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) { s0[k] = x[k] * y[k]; h0[k] = 0.5 * y[k]; }
#pragma unroll
    for (int k = 0; k < K; ++k) r0[k] = __builtin_fma(-h0[k], s0[k], 0.5);
#pragma unroll
    for (int k = 0; k < K; ++k) s1[k] = __builtin_fma(s0[k], r0[k], s0[k]);
    if constexpr (LIGHT) {
        // The nearest-hit search of a COMPACT scene (SceneView::fold_flags, kSceneCompact), tolerance unit only: the roots
        // without their residual step — s1 is good to 1.5 e^2 = 2^-45 (e = 2^-23: v_rsq_f64).  What a root decides there: which
        // sphere is nearest (a tie within 3e-14 of a distance: as rare as the ties contraction already moves), t1 against
        // 0.001, and the far root t2 = b + sqrt(D4) of the sphere a bounce ray starts on against 1e-5f — t2 is 0 in real
        // arithmetic and 3e-14 |b| here, five orders below the threshold while |b| <= 1e7, which is what "compact" promises
        // (a sphere of radius 1e9 would start hitting itself: such scenes keep the full roots).  The winner's distance carries
        // the same 3e-14: 2e-10 of absolute error on the shipped box's walls, against hit ids that need 1e-3 to change.
        (void)d0;
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = s1[k];
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) d0[k] = __builtin_fma(-s1[k], s1[k], x[k]);
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = __builtin_fma(d0[k], h0[k], s1[k]);
    }
#else
    static_assert(!LIGHT, "light roots are the tolerance unit's");
    double y[K], s0[K], h0[K], r0[K], s1[K], h1[K], d0[K], s2[K], d1[K];
#pragma unroll
    for (int k = 0; k < K; ++k) y[k] = __builtin_amdgcn_rsq(x[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) { s0[k] = x[k] * y[k]; h0[k] = 0.5 * y[k]; }
#pragma unroll
    for (int k = 0; k < K; ++k) r0[k] = __builtin_fma(-h0[k], s0[k], 0.5);
#pragma unroll
    for (int k = 0; k < K; ++k) { s1[k] = __builtin_fma(s0[k], r0[k], s0[k]); h1[k] = __builtin_fma(h0[k], r0[k], h0[k]); }
#pragma unroll
    for (int k = 0; k < K; ++k) d0[k] = __builtin_fma(-s1[k], s1[k], x[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) s2[k] = __builtin_fma(d0[k], h1[k], s1[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) d1[k] = __builtin_fma(-s2[k], s2[k], x[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = __builtin_fma(d1[k], h1[k], s2[k]);
#endif
}

// The LDS constants of a shading kernel: the 16 sincos constants and the near-unit Normalize table (rtm_device.h); lanes
// 0 .. 31 of a wave each write their share.  The table's entries are what div3_by_magnitude computes for those squared
// lengths, instruction for instruction.
__device__ __forceinline__ void fill_shade_consts(double* consts, const int lane, const bool unit_table) {
    if (lane < kTrigConstCount) {
        consts[lane] = TrigFromRegs{}[lane];
    } else if (unit_table && lane < kTrigConstCount + kUnitWindow) {
        const int k = lane - kTrigConstCount;
        const double y = (double)sqrtf_fast(__uint_as_float(kUnitWindowFirst + (uint32_t)k));
        consts[kTrigConstCount + 2 * k] = y;
        consts[kTrigConstCount + 2 * k + 1] = seq_rcp(y);
    }
}

// MathRef: the compiler's IEEE expansions (v_div_scale/v_div_fmas/v_div_fixup, scaled rsq+NR sqrt).
struct MathRef {
    static __device__ __forceinline__ double sqrt64(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ D3 div3(D3 a, double y) { return a / y; }
    static __device__ __forceinline__ double div(double x, double y) { return x / y; }
    template <int K>
    static __device__ __forceinline__ void sqrt64_batch(const double (&x)[K], double (&out)[K]) {
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = ::sqrt(x[k]);
    }
    template <int K>
    static __device__ __forceinline__ void sqrt64_batch_hit(const double (&x)[K], double (&out)[K]) {
        sqrt64_batch<K>(x, out);
    }
};

// MathFast: the same instruction sequences with the range-scaling steps hoisted into one
// wave-uniform guard.  hipcc expands every fp64 division into div_scale, div_scale, rcp, 4 fma,
// mul, fma, div_fmas, div_fixup and every fp64 sqrt into (scale) rsq, 2 mul, 7 fma, (unscale),
// class fix-up.  div_scale / the sqrt pre-scale are the identity unless an operand's exponent is
// extreme, so when no lane of the wave has such an operand the scaling instructions are dropped and
// — for the three divisions of a Normalize (src/Ray.h:70-72) — the reciprocal refinement of the
// common denominator is computed once.  The remaining instructions are exactly the compiler's, so
// results are bit-identical; any lane outside the guard sends the whole wave down MathRef's path.
// tests/test_parity_gpu.py::test_fast_math_is_bit_identical checks both against each other.
struct MathFast {
    static __device__ __forceinline__ bool moderate(double v) {
        // exponent within +-400 (zero, inf and nan report exponent 0 and are settled by
        // v_div_fixup exactly like in the compiler's sequence)
        const int e = __builtin_amdgcn_frexp_exp(v);
        return (unsigned)(e + 400) <= 800u;
    }
    static __device__ __forceinline__ double sqrt64(double x) {
        // Fast lanes: positive, finite, >= 2^-767 (no pre-scaling, no 0/inf fix-up needed) or
        // negative (rsq gives the NaN IEEE asks for).  +-0, tiny, +inf, +NaN take the full path.
        const unsigned h = (unsigned)__double2hiint(x);
        const bool fast = (h - 0x10000000u < 0x6FF00000u) || (h > 0x80000000u);
        if (__builtin_amdgcn_ballot_w64(!fast) != 0) return ::sqrt(x);
        return seq_sqrt(x);
    }
    static __device__ __forceinline__ bool sqrt_fast_ok(double x) {
        const unsigned h = (unsigned)__double2hiint(x);
        return (h - 0x10000000u < 0x6FF00000u) || (h > 0x80000000u);
    }
    // K independent square roots behind ONE wave-uniform guard (one branch per batch, and K
    // independent dependency chains for the scheduler to interleave)
    template <int K>
    static __device__ __forceinline__ void sqrt64_batch(const double (&x)[K], double (&out)[K]) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < K; ++k) ok = ok && sqrt_fast_ok(x[k]);
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) out[k] = ::sqrt(x[k]);
            return;
        }
        seq_sqrt_batch<K>(x, out);
    }
    // The square roots of a chunk of DISCRIMINANTS (src/SettingData.cpp:205), for the nearest-hit search
    // only.  There the guard can be one unsigned compare per value — "not in [+0, 2^-767)" — because
    // every other operand the unscaled sequence gets wrong yields the same decision as the reference:
    //   D4 < 0      rsq gives NaN, t is NaN, nothing is accepted (the reference returns false);
    //   D4 = +inf   the sequence gives NaN where sqrt gives +inf; with sq = +inf, t1 = b - inf is -inf or
    //               NaN, so t = t2 = b + inf is +inf or NaN and "t < dis" is false either way;
    //   D4 = NaN    NaN both ways;
    //   D4 = -0     cannot occur: D4 = (b*b - p.p) + r2 with r2 = (double)(r*r) >= +0, and x + (+0 or a
    //               positive number) is never -0.
    // +0, subnormals and tiny normals are the cases that need ocml's pre-scaling (a tangent ray with
    // D4 = +0 has a real hit at t = b).
    template <int K, bool LIGHT = false>
    static __device__ __forceinline__ void sqrt64_batch_hit(const double (&x)[K], double (&out)[K]) {
#if RTM_OPT_GUARD
        // "every high word >= 0x10000000" as ONE compare of their unsigned minimum (v_min3_u32): the K
        // compare / select / shift / or steps hipcc makes of the && chain were ~25 instructions per cast
        unsigned lowest = (unsigned)__double2hiint(x[0]);
#pragma unroll
        for (int k = 1; k < K; ++k) {
            const unsigned h = (unsigned)__double2hiint(x[k]);
            lowest = h < lowest ? h : lowest;
        }
        const bool ok = lowest >= 0x10000000u;
#else
        bool ok = true;
#pragma unroll
        for (int k = 0; k < K; ++k) ok = ok && ((unsigned)__double2hiint(x[k]) >= 0x10000000u);
#endif
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) out[k] = ::sqrt(x[k]);
            return;
        }
        seq_sqrt_batch<K, LIGHT>(x, out);
    }
    // one division (the plane test's t): the same sequence for a single numerator
    static __device__ __forceinline__ double div(double x, double y) {
        const bool ok = moderate(y) && moderate(x);
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) return x / y;
        const double r = seq_rcp(y);
        return __builtin_amdgcn_div_fixup(seq_quot(x, y, r), y, x);
    }
    static __device__ __forceinline__ D3 div3(D3 a, double y) {
        const bool ok = moderate(y) && moderate(a.x) && moderate(a.y) && moderate(a.z);
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) return a / y;
        const double r = seq_rcp(y);
        auto one = [&](double x) {
            return __builtin_amdgcn_div_fixup(seq_quot(x, y, r), y, x);
        };
        return D3{one(a.x), one(a.y), one(a.z)};
    }
};

// MathSpec: speculate.  Every sqrt / division runs MathFast's unscaled sequence unconditionally and
// records, per lane, whether an operand was outside the range in which that sequence is the
// compiler's own (bit-identical) expansion.  The caller checks `bad` once per block with a ballot
// and re-runs the block with MathRef if any lane tripped it — one branch per block instead of one
// per operation, and one long basic block for the scheduler.
// GUARD (the default flavour, MathSpec): the guards of a Normalize reduced to ONE compare of the smallest squared
// component, no v_div_fixup behind them — which costs the ability to pass an exactly zero component through (it trips
// `bad`).  Fine for sphere normals and random bounce directions, where a zero component is a measure-zero event;
// useless for PLANE normals, which are axis-aligned more often than not.  MathSpecZ = MathSpecT<false> keeps the
// v_div_fixup and the per-operand exponent checks (zeros, infinities and NaNs are settled by the fix-up exactly as in
// the compiler's sequence) and serves the scenes that hold planes.
template <bool GUARD>
struct MathSpecT {
    static constexpr bool kGuard = GUARD;
    bool bad = false;
    const double* trig_lds = nullptr;  // LDS copy of the sincos constants (optional)
    const double* unit_lds = nullptr;  // ... and the near-unit Normalize table (optional): set_lds
    __device__ __forceinline__ void set_lds(const ShadeLds& l) {
        trig_lds = l.trig;
        unit_lds = l.unit;
    }
    __device__ __forceinline__ double sqrt64(double x) {
        bad = bad || !MathFast::sqrt_fast_ok(x);
        return seq_sqrt(x);
    }
    // sqrt of an operand KNOWN to be a positive normal number of moderate size (r2 and 1 - r2 of a 23-bit draw:
    // [2^-24, 1)): the unscaled sequence without its range check
    __device__ __forceinline__ double sqrt64_unit(double x) {
        return seq_sqrt(x);
    }
    __device__ __forceinline__ D3 div3(D3 a, double y) {
        bad = bad || !(MathFast::moderate(y) && MathFast::moderate(a.x) && MathFast::moderate(a.y) &&
                       MathFast::moderate(a.z));
        const double r = seq_rcp(y);
        auto one = [&](double x) {
            return __builtin_amdgcn_div_fixup(seq_quot(x, y, r), y, x);
        };
        return D3{one(a.x), one(a.y), one(a.z)};
    }
    __device__ __forceinline__ void sincos_r1(double x, double& sn, double& cs) {
        if (trig_lds)
            sincos_small_k(TrigFromLds{trig_lds}, x, sn, cs);
        else
            sincos_small(x, sn, cs);
    }
    // sin / cos of r1 = 2 pi u for the draw u = m24 2^-24 (src/Renderer.cpp:88: (2 pi 2^-24) m is the same correctly
    // rounded product as 2 pi (m 2^-24))
    __device__ __forceinline__ void sincos_draw(double m24, double& sn, double& cs) {
#if RTM_TOL
        if (trig_lds)
            sincos_turn24_k(TrigFromLds{trig_lds}, m24, sn, cs);
        else
            sincos_turn24_k(TrigFromRegs{}, m24, sn, cs);
#else
        sincos_r1((6.283185307179586 * 0x1p-24) * m24, sn, cs);
#endif
    }
    // Magnitude (src/Ray.h:67-69) with the unscaled float sqrt
    __device__ __forceinline__ double magnitude_spec(D3 a) {
        const float len2 = (float)(a.x * a.x + a.y * a.y + a.z * a.z);
        bad = bad || !sqrtf_fast_ok(len2);
        return (double)sqrtf_fast(len2);
    }
    // The three divisions x / m, m = (double)sqrtf((float)(x.x)) of a Normalize, with the guard reduced to what
    // can actually go wrong.  sqrtf_fast_ok(len2) already pins m to [2^-48, 2^64) — always "moderate" — and
    // every |x_i| to < 2^64, and a NaN or infinite component makes len2 NaN or infinite, which it rejects.
    // What is left is a component so small that v_div_scale would have rescaled the division (|x_i| below
    // about 2^-900): caught by ONE unsigned compare of the smallest high word of the SQUARES, which the
    // caller has at hand (x_i^2 < 2^-800 means |x_i| < 2^-400, the bound `moderate` uses).  An exactly
    // zero component (square +0) trips it too and sends the wave down the compiler's path: correct, and
    // for directions that went through a random bounce a measure-zero event.
    __device__ __forceinline__ D3 div3_by_magnitude(D3 a, double sx, double sy, double sz, float len2) {
#if RTM_TOL
        // (the tolerance unit's quotient is x * (1 / y): no remainder step that a tiny component could break, no guard for one)
        (void)sx; (void)sy; (void)sz;
        bad = bad | !sqrtf_fast_ok(len2);
#else
        unsigned lo = (unsigned)__double2hiint(sx);
        const unsigned hy = (unsigned)__double2hiint(sy), hz = (unsigned)__double2hiint(sz);
        lo = hy < lo ? hy : lo;
        lo = hz < lo ? hz : lo;
        bad = bad | !sqrtf_fast_ok(len2) | (lo < 0x0DF00000u);  // 2^-800
#endif
        const double y = (double)sqrtf_fast(len2);
        const double r = seq_rcp(y);
        // no v_div_fixup: with a non-zero, non-tiny, finite numerator and a denominator in [2^-48, 2^64) it returns
        // its first operand (it only rewrites zeros, infinities, NaNs and results outside the exponent range)
        auto one = [&](double x) {
            return seq_quot(x, y, r);
        };
        return D3{one(a.x), one(a.y), one(a.z)};
    }
    __device__ __forceinline__ D3 normalize(D3 a) {
        if constexpr (GUARD) {
            const double sx = a.x * a.x, sy = a.y * a.y, sz = a.z * a.z;
            return div3_by_magnitude(a, sx, sy, sz, (float)(sx + sy + sz));
        } else {
            return div3(a, magnitude_spec(a));
        }
    }
    // Normalize of a bounce direction (src/Renderer.cpp:103-107): a combination of an orthonormal basis whose vectors went
    // through the float-rounded Normalize themselves, so its squared length, as a float, is one of a handful of values
    // around 1 (-4 .. +2 ulps over 4e6 bounces) — the float square root, its correction and the reciprocal's refinement are
    // a 16-byte LDS read instead of a dozen instructions, three and a half of them transcendental-rate.  Same values:
    // the table holds what div3_by_magnitude computes (fill_shade_consts).  A length outside the window trips `bad`.
    __device__ __forceinline__ D3 normalize_near_unit(D3 a) {
        if constexpr (GUARD) {
            if (unit_lds) {
                const double sx = a.x * a.x, sy = a.y * a.y, sz = a.z * a.z;
                const unsigned k = __float_as_uint((float)(sx + sy + sz)) - kUnitWindowFirst;
#if RTM_TOL
                bad = bad | (k >= (unsigned)kUnitWindow);
#else
                unsigned lo = (unsigned)__double2hiint(sx);
                const unsigned hy = (unsigned)__double2hiint(sy), hz = (unsigned)__double2hiint(sz);
                lo = hy < lo ? hy : lo;
                lo = hz < lo ? hz : lo;
                bad = bad | (k >= (unsigned)kUnitWindow) | (lo < 0x0DF00000u);  // see div3_by_magnitude
#endif
                const double2 yr = reinterpret_cast<const double2*>(unit_lds)[k & (unsigned)(kUnitWindow - 1)];
                auto one = [&](double x) {
                    return seq_quot(x, yr.x, yr.y);
                };
                return D3{one(a.x), one(a.y), one(a.z)};
            }
        }
        return normalize(a);
    }
    // Normalize of a vector whose y component is a (signed) zero — Cross((0,1,0), w) for finite w:
    // y*y adds +0 to the squared length and +-0 / m is the same +-0, so only x and z are divided.
    // Anything else in y (NaN from a non-finite w) trips `bad`.
    __device__ __forceinline__ D3 normalize_y0(D3 a) {
        double y;
        if constexpr (GUARD) {
            const double sx = a.x * a.x, sz = a.z * a.z;
            const float len2 = (float)(sx + sz);
            const unsigned hx = (unsigned)__double2hiint(sx), hz = (unsigned)__double2hiint(sz);
            bad = bad | !sqrtf_fast_ok(len2) | !(a.y == 0.0) | ((hx < hz ? hx : hz) < 0x0DF00000u);  // see div3_by_magnitude
            y = (double)sqrtf_fast(len2);
        } else {
            const float len2 = (float)(a.x * a.x + a.z * a.z);
            bad = bad || !sqrtf_fast_ok(len2) || !(a.y == 0.0);
            y = (double)sqrtf_fast(len2);
            bad = bad || !(MathFast::moderate(a.x) && MathFast::moderate(a.z));  // y: a normal float here
        }
        const double r = seq_rcp(y);
        auto one = [&](double x) {
            return __builtin_amdgcn_div_fixup(seq_quot(x, y, r), y, x);
        };
        return D3{one(a.x), a.y, one(a.z)};
    }
    // Normalize of (cx, +-0, cz): the two non-zero components of normalize_y0, for the caller that has
    // established the zero structurally and does not need it back (GUARD flavour only).
    __device__ __forceinline__ void normalize_xz(double cx, double cz, double& ux, double& uz) {
        const double sx = cx * cx, sz = cz * cz;
        const float len2 = (float)(sx + sz);
#if RTM_TOL
        bad = bad | !sqrtf_fast_ok(len2);
#else
        const unsigned hx = (unsigned)__double2hiint(sx), hz = (unsigned)__double2hiint(sz);
        bad = bad | !sqrtf_fast_ok(len2) | ((hx < hz ? hx : hz) < 0x0DF00000u);  // see div3_by_magnitude
#endif
        const double y = (double)sqrtf_fast(len2);
        const double r = seq_rcp(y);
        auto one = [&](double x) {  // no v_div_fixup: see div3_by_magnitude
            return seq_quot(x, y, r);
        };
        ux = one(cx);
        uz = one(cz);
    }
    // Normalize(hit - centre) (src/SettingData.cpp:214-215) for a hit point that is ON its sphere to
    // float precision: then (float)|dv|^2 is exactly (float)(r*r), so Magnitude returns the
    // per-sphere constant `ms` and the reciprocal refinement of the three divisions is the
    // per-sphere constant `rinv` (both precomputed with the same instruction sequence).  Taken only
    // when every active lane is in that case; otherwise the general Normalize runs.
    __device__ __forceinline__ D3 normalize_on_sphere(D3 dv, double ms, double rinv, float r2f) {
        if constexpr (GUARD) {
            const double sx = dv.x * dv.x, sy = dv.y * dv.y, sz = dv.z * dv.z;
            const float len2f = (float)(sx + sy + sz);
            const bool canon = (len2f == r2f) && (rinv == rinv);
            if (__builtin_amdgcn_ballot_w64(!canon) != 0) return div3_by_magnitude(dv, sx, sy, sz, len2f);
            // canonical: |dv|^2 rounds to the float r*r of a sphere whose refined reciprocal exists (rinv is NaN when
            // ms is outside the exact range), so only a tiny component remains to be excluded
#if !RTM_TOL
            unsigned lo = (unsigned)__double2hiint(sx);
            const unsigned hy = (unsigned)__double2hiint(sy), hz = (unsigned)__double2hiint(sz);
            lo = hy < lo ? hy : lo;
            lo = hz < lo ? hz : lo;
            bad = bad | (lo < 0x0DF00000u);
#endif
            auto one = [&](double x) {
                return seq_quot(x, ms, rinv);  // no v_div_fixup: see div3_by_magnitude
            };
            return D3{one(dv.x), one(dv.y), one(dv.z)};
        } else {
            const float len2f = (float)(dv.x * dv.x + dv.y * dv.y + dv.z * dv.z);
            const bool canon = (len2f == r2f) && (rinv == rinv);
            if (__builtin_amdgcn_ballot_w64(!canon) != 0) {
                bad = bad || !sqrtf_fast_ok(len2f);
                return div3(dv, (double)sqrtf_fast(len2f));
            }
            bad = bad || !(MathFast::moderate(dv.x) && MathFast::moderate(dv.y) && MathFast::moderate(dv.z));
            auto one = [&](double x) {
                return __builtin_amdgcn_div_fixup(seq_quot(x, ms, rinv), ms, x);
            };
            return D3{one(dv.x), one(dv.y), one(dv.z)};
        }
    }
    // One correctly rounded division x / y (the plane test's t = n.(p - o) / n.d): the unscaled sequence with the fix-up
    __device__ __forceinline__ double div(double x, double y) {
        bad = bad || !(MathFast::moderate(y) && MathFast::moderate(x));
        const double r = seq_rcp(y);
        return __builtin_amdgcn_div_fixup(seq_quot(x, y, r), y, x);
    }
};
using MathSpec = MathSpecT<RTM_OPT_GUARD != 0>;
using MathSpecZ = MathSpecT<false>;
template <class T>
struct is_spec : std::false_type {};
template <bool G>
struct is_spec<MathSpecT<G>> : std::true_type {};
// the reciprocal refinement MathFast/MathSpec::div3 apply to a denominator (v_rcp_f64 + two
// Newton steps), or NaN when the denominator is outside their exact range
__device__ __forceinline__ double refined_rcp_or_nan(double y) {
    if (!MathFast::moderate(y) || y == 0.0) return __builtin_nan("");
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    return __builtin_fma(r, e, r);
}
// instance adaptors so that path_shade can take any policy as an object
struct MathRefI {
    static constexpr bool bad = false;
    const double* trig_lds = nullptr;
    __device__ __forceinline__ void set_lds(const ShadeLds& l) { trig_lds = l.trig; }
    __device__ __forceinline__ D3 normalize(D3 a) { return MathRef::div3(a, magnitude(a)); }
    __device__ __forceinline__ D3 normalize_y0(D3 a) { return normalize(a); }
    __device__ __forceinline__ D3 normalize_near_unit(D3 a) { return normalize(a); }
    __device__ __forceinline__ double sqrt64(double x) { return MathRef::sqrt64(x); }
    __device__ __forceinline__ double sqrt64_unit(double x) { return MathRef::sqrt64(x); }
    __device__ __forceinline__ D3 div3(D3 a, double y) { return MathRef::div3(a, y); }
    // r1 = 2*pi*u < 2^30 always: ocml's small-argument sequence IS ::sincos there (bit-identical)
    __device__ __forceinline__ void sincos_r1(double x, double& sn, double& cs) {
        if (trig_lds)
            sincos_small_k(TrigFromLds{trig_lds}, x, sn, cs);
        else
            sincos(x, &sn, &cs);
    }
    __device__ __forceinline__ void sincos_draw(double m24, double& sn, double& cs) { sincos_r1((6.283185307179586 * 0x1p-24) * m24, sn, cs); }
};
struct MathFastI {
    static constexpr bool bad = false;
    __device__ __forceinline__ D3 normalize(D3 a) { return MathFast::div3(a, magnitude(a)); }
    __device__ __forceinline__ D3 normalize_y0(D3 a) { return normalize(a); }
    __device__ __forceinline__ D3 normalize_near_unit(D3 a) { return normalize(a); }
    __device__ __forceinline__ double sqrt64(double x) { return MathFast::sqrt64(x); }
    __device__ __forceinline__ double sqrt64_unit(double x) { return MathFast::sqrt64(x); }
    __device__ __forceinline__ D3 div3(D3 a, double y) { return MathFast::div3(a, y); }
    __device__ __forceinline__ void sincos_r1(double x, double& sn, double& cs) { sincos(x, &sn, &cs); }
    __device__ __forceinline__ void sincos_draw(double m24, double& sn, double& cs) { sincos_r1((6.283185307179586 * 0x1p-24) * m24, sn, cs); }
};

// src/Ray.h:67-72 through a policy
template <class M>
__device__ __forceinline__ D3 normalize_m(D3 a) {
    return M::div3(a, magnitude(a));
}
template <class MI>
__device__ __forceinline__ D3 normalize_i(MI& m, D3 a) {
    return m.normalize(a);
}

// ------------------------------------------------------------------------------------------------
// Scene policies.  geom[i] = (cx, cy, cz, (double)(float)(r*r)); mat[i*8..] = colorKD.xyz,
// emission.xyz, kd, pad (the fold reads the first six doubles of a row as three 16-byte loads).
// ------------------------------------------------------------------------------------------------
// Uniform grid over a large scene's spheres (rtm_kernels.hip: build_scene_grid; nearest_hit_grid below).  Device memory,
// read through wave-uniform scalar loads.  A sphere is listed in every cell its box — centre +- (radius + pad_i) —
// overlaps; the few spheres whose box would cover too many cells are in `big` instead and are tested by every ray.
struct GridHeader {
    double lo[3], hi[3];  // the grid's box: hi = lo + dim * h
    double h, inv_h;      // cell edge
    double cb[3], reach2; // a ray whose origin is farther than sqrt(reach2) from the box's centre cb, ...
    double dd_tol;        // ... or whose |dir.dir - 1| is larger, takes the exhaustive loop (the pads do not cover it)
    double t_ok;          // the largest hit parameter the pads were sized for (host side; reach2 follows from it)
    int dim[3];
    int n_big;
    unsigned n_recs, pad_;
    const uint2* cell_range;     // per cell (dim[0]*dim[1]*dim[2] of them): first and one-past-last entry of its list in `recs`
    const double4* recs;         // a cell's spheres, self-contained: (cx, cy, cz, r*r | index), ascending index within a cell
    const int* big;              // n_big sphere indices, ascending
};

struct SceneView {
    const double4* __restrict__ geom;
    const double* __restrict__ mat;
    int n;
    // large-scene rejection test (rtm_wavefront.h): wprime[i] = r2_i - |c_i|^2 (padded to a multiple of
    // 8 entries), bounds = {max |c|^2, max r2}
    const double* __restrict__ wprime = nullptr;
    const double* __restrict__ bounds = nullptr;
    const float4* __restrict__ geom32 = nullptr;  // (float)cx, cy, cz, (float)w' (padded like wprime)
    const float4* __restrict__ geom32s = nullptr;  // the same numbers, one float4 (cx, cy, cz, w') per sphere
    // RTM_MODE_HOST_TRIG: 4 bits per possible r1 (2^23 of them): the differences, in units of the last
    // place, between the host libm's sin/cos and the device's (rtm_kernels.hip, ensure_trig_fix)
    const uint32_t* __restrict__ trig_fix = nullptr;
    // scenes with png::PlaneObject entries (include/rtm.h: rtm_object): 16 doubles per object — position, normal,
    // right, upv, right.right, upv.upv, 0, 0 — valid where geom[i].w < 0 (a sphere's r*r is never negative); null
    // for all-sphere scenes.  Served by the per-object loop only (variant 1).
    const double* __restrict__ plane = nullptr;
    // large all-sphere scenes held by an rtm_scene: the uniform grid of the grid kernel (variant 17), else null
    const GridHeader* __restrict__ grid = nullptr;
    // png::SurfaeSample (rtm_surface.h) reads what PathTracing does not: per object the RAW material colour (3 doubles) and
    // the radius as the float it is, widened (SphereObject::ComputeSurfacePoint); null where the tables were made without
    const double* __restrict__ surf = nullptr;
    // two bits per sphere for the first 32 (sphere i: bits 2i, 2i+1; set by the host where it holds the geometry, else 0):
    // 1 / 2 / 3 = the centre's only non-zero coordinate is x / y / z (the other two are +-0), 0 = anything else.  The
    // chunked search of the LDS-table kernels evaluates such a sphere's discriminant from per-ray shared products
    // (sphere_disc below): the reference's own roundings, fewer instructions.
    unsigned long long axis_pat = 0ull;
    // kFoldNoLevelEmission (set by the host where it holds the material rows, else 0): every object a path can bounce off
    // (kd > 0) has emission (+0, +0, +0) and a colorKD without sign bits, and no emission has a sign bit — the fold's
    // "+ emission" of a bounce level (src/Renderer.cpp:109) then adds +0 to a product that is never -0: an identity, bit for
    // bit, which the packed folds leave out (and with it the levels' emission reads).  The shipped Cornell box: the walls.
    unsigned fold_flags = 0u;
};
constexpr unsigned kFoldNoLevelEmission = 1u;
// ... and bit 1, kSceneCompact: every |centre| + radius is finite and at most 1e7 (the launcher adds the camera): what the
// tolerance unit's light search roots ask for (seq_sqrt_batch)
constexpr unsigned kSceneCompact = 2u;
constexpr double kCompactExtent = 1e7;

// png::PlaneObject::Intersect as this build completes it (include/rtm.h): the reference's first line
// (src/SettingData.cpp:244), the sphere's near threshold, then the square's extent.
__device__ __forceinline__ bool plane_test(const double* __restrict__ pl, const D3 org, const D3 dir, double& t) {
    const D3 pos = d3(pl[0], pl[1], pl[2]), nrm = d3(pl[3], pl[4], pl[5]);
    const double dn = dot(nrm, dir);
    if (fabs(dn) < (double)FLT_EPSILON) return false;
    const double tt = dot(nrm, pos - org) / dn;
    if (!(tt > 0.001)) return false;
    const D3 d = (org + dir * tt) - pos;
    const D3 right = d3(pl[6], pl[7], pl[8]), upv = d3(pl[9], pl[10], pl[11]);
    if (fabs(dot(d, right)) > pl[12] || fabs(dot(d, upv)) > pl[13]) return false;
    t = tt;
    return true;
}

// sin(r1), cos(r1) as the HOST's libm returns them (src/Renderer.cpp:93-94 call std::sin / std::cos):
// r1 = 2*pi*u takes one of 2^23 values, so the device's results are compared once with the host's over
// the whole domain and the +-1 ulp differences kept in a table (two signed 2-bit fields per r1).
// `rng` is the stream right after the draws of r1 and r2: r1's draw is recomputed from it rather than
// kept live, so the default path (no table) carries nothing extra.
// Two halves, so that the table word can be requested as soon as r1's draw is known and used ~150
// instructions later, after the sincos it corrects (the load's latency then overlaps the block's arithmetic).
struct TrigFixWord {
    uint32_t word;
    int off;
};
__device__ __forceinline__ TrigFixWord trig_fix_load(const uint32_t* __restrict__ fix, const RngStream& rng) {
    const uint32_t k = mix32((rng.ctr - 2u * 0x9E3779B9u) ^ rng.k1) >> 9;
    return TrigFixWord{fix[k >> 3], (int)(k & 7u) * 4};
}
__device__ __forceinline__ void trig_fix_apply(const TrigFixWord w, double& sn, double& cs) {
    const long long ds = (long long)(int)__builtin_amdgcn_sbfe(w.word, (unsigned)w.off, 2u);      // sign-extended
    const long long dc = (long long)(int)__builtin_amdgcn_sbfe(w.word, (unsigned)w.off + 2u, 2u);
    sn = __longlong_as_double(__double_as_longlong(sn) + ds);
    cs = __longlong_as_double(__double_as_longlong(cs) + dc);
}
__device__ __forceinline__ void apply_trig_fix(const uint32_t* __restrict__ fix, const RngStream& rng, double& sn,
                                               double& cs) {
    trig_fix_apply(trig_fix_load(fix, rng), sn, cs);
}

// Wave-uniform geometry fetch.  The tables are never written while a render kernel runs, but the
// kernel also stores through unrelated pointers (pixels, pooled records), which makes hipcc fall
// back to per-lane vector loads; reading through the constant address space states the invariance
// and brings back scalar loads (s_load_dwordx8 into SGPRs, one per sphere per wave).
__device__ __forceinline__ double4 load_geom_uniform(const double4* geom, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) double* ConstF64Ptr;
    ConstF64Ptr p = (ConstF64Ptr)(unsigned long long)(geom + i);
    return double4{p[0], p[1], p[2], p[3]};
#else
    return geom[i];
#endif
}

// Everything from global memory (uniform geometry loads become scalar loads; per-lane material and
// centre look-ups are vector gathers served by L1/L2).  Works for any n.
struct SceneGlobal {
    SceneView v;
    static constexpr bool kHasNormTable = false;
    static constexpr bool kPlanes = false;  // (planes, if any, are found at run time: v.plane, the per-object loop)
    static constexpr bool kNeverPlanes = false;
    __device__ __forceinline__ int n() const { return v.n; }
    __device__ __forceinline__ double4 geom_uniform(int i) const { return load_geom_uniform(v.geom, i); }
    __device__ __forceinline__ D3 center(int id) const {
        const double4 g = v.geom[id];
        return D3{g.x, g.y, g.z};
    }
    __device__ __forceinline__ double kd(int id) const { return v.mat[(size_t)id * 8 + 6]; }
    __device__ __forceinline__ double kd24(int id) const { return v.mat[(size_t)id * 8 + 7]; }  // kd * 2^24
    __device__ __forceinline__ D3 emission(int id) const {
        const double* m = v.mat + (size_t)id * 8;
        return D3{m[3], m[4], m[5]};
    }
    __device__ __forceinline__ D3 color_kd(int id) const {
        const double* m = v.mat + (size_t)id * 8;
        return D3{m[0], m[1], m[2]};
    }
};

// Small scenes: geometry still arrives through wave-uniform (scalar) loads, but the per-lane
// look-ups (hit sphere's centre, kd, emission, colorKD) come from an LDS copy of the tables.
struct SceneLds {
    SceneView v;
    const double* lgeom;  // LDS, 4 doubles per sphere
    const double* lmat;   // LDS, 8 doubles per sphere
    const double* lnrm;   // LDS, 3 doubles per sphere: |hit - centre| as Magnitude returns it for a point
                          // on the sphere, (double)sqrtf((float)(r*r)), its refined reciprocal, the float r*r
    static constexpr bool kHasNormTable = true;
    static constexpr bool kPlanes = false;
    static constexpr bool kNeverPlanes = RTM_OPT_CTMODE != 0;  // the launchers hand scenes that hold planes to SceneLdsObjects (v.plane is null here)
    __device__ __forceinline__ double norm_m(int id) const { return lnrm[id * 3]; }
    __device__ __forceinline__ double norm_rinv(int id) const { return lnrm[id * 3 + 1]; }
    __device__ __forceinline__ float norm_r2f(int id) const { return reinterpret_cast<const float*>(lnrm + id * 3 + 2)[0]; }
    __device__ __forceinline__ int n() const { return v.n; }
    __device__ __forceinline__ double4 geom_uniform(int i) const { return load_geom_uniform(v.geom, i); }
    __device__ __forceinline__ D3 center(int id) const {
        const double* g = lgeom + id * 4;
        return D3{g[0], g[1], g[2]};
    }
    __device__ __forceinline__ double kd(int id) const { return lmat[id * 8 + 6]; }
    __device__ __forceinline__ double kd24(int id) const { return lmat[id * 8 + 7]; }
    __device__ __forceinline__ D3 emission(int id) const {
        const double* m = lmat + id * 8;
        return D3{m[3], m[4], m[5]};
    }
    __device__ __forceinline__ D3 color_kd(int id) const {
        const double* m = lmat + id * 8;
        return D3{m[0], m[1], m[2]};
    }
};

// Small scenes that hold png::PlaneObject entries (include/rtm.h: rtm_object), for the chunked fast kernels: the same
// LDS tables, where a plane's geometry row is (position, negative "r*r") and its 3-double row of the normal-length
// table holds the plane's NORMAL instead (a sphere's normal-length shortcut is not used in such scenes).  A plane's
// 14 test operands are wave-uniform like a sphere's geometry and come by scalar loads from v.plane.
struct SceneLdsObjects {
    SceneView v;
    const double* lgeom;  // LDS, 4 doubles per object
    const double* lmat;   // LDS, 8 doubles per object
    const double* lnrm;   // LDS, 3 doubles per object: a plane's m_normal (unused for spheres)
    static constexpr bool kHasNormTable = false;
    static constexpr bool kPlanes = true;
    static constexpr bool kNeverPlanes = false;
    __device__ __forceinline__ int n() const { return v.n; }
    __device__ __forceinline__ double4 geom_uniform(int i) const { return load_geom_uniform(v.geom, i); }
    __device__ __forceinline__ D3 center(int id) const {
        const double* g = lgeom + id * 4;
        return D3{g[0], g[1], g[2]};
    }
    __device__ __forceinline__ bool is_plane(int id) const { return __double2hiint(lgeom[id * 4 + 3]) < 0; }
    __device__ __forceinline__ D3 plane_normal(int id) const {
        const double* q = lnrm + id * 3;
        return D3{q[0], q[1], q[2]};
    }
    __device__ __forceinline__ double kd(int id) const { return lmat[id * 8 + 6]; }
    __device__ __forceinline__ double kd24(int id) const { return lmat[id * 8 + 7]; }
    __device__ __forceinline__ D3 emission(int id) const {
        const double* m = lmat + id * 8;
        return D3{m[3], m[4], m[5]};
    }
    __device__ __forceinline__ D3 color_kd(int id) const {
        const double* m = lmat + id * 8;
        return D3{m[0], m[1], m[2]};
    }
};

// ------------------------------------------------------------------------------------------------
// src/SettingData.cpp:197-226 without the normal: returns hit and t (the literal statement order).
template <class M>
__device__ __forceinline__ bool sphere_test(const double4 g, const D3 org, const D3 dir, double& t) {
    const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);  // :198
    const double b = dot(p_o, dir);                            // :199
    const double D4 = b * b - dot(p_o, p_o) + g.w;             // :200 (g.w = float r*r widened)
    if (D4 < 0.0) return false;                                // :202
    const double sq = M::sqrt64(D4);                           // :205
    const double t1 = b - sq, t2 = b + sq;
    const double min_value = (double)1e-5f;                    // :208
    if (t1 < min_value && t2 < min_value) return false;        // :209
    t = (t1 > 0.001) ? t1 : t2;                                // :212-223
    return true;
}

// Intersect (src/SettingData.cpp:197-226) fused with the caller's acceptance test
// (src/Renderer.cpp:67: hit && t < dis && t > 0), select-only form.  Equivalence with the
// statement order above, with m = 1e-5f and sq >= 0 or NaN (so t2 >= t1):
//   t1 > 0.001          : Intersect returns t = t1 (> m > 0); accepted iff t1 < dis.
//   t1 <= 0.001         : Intersect returns t = t2 unless (t1 < m && t2 < m); since t2 >= t1 that is
//                         "t2 >= m", and then t2 > 0 holds; accepted iff m <= t2 < dis.
//   D4 < 0              : the policy sqrt yields NaN, t is NaN, "t < dis" is false: no hit.
//   NaN ray (literal)   : D4 NaN, Intersect returns true with t NaN, the caller's t < dis fails.
// Hence accept == (t < dis) && !(t < m) with t = t1 > 0.001 ? t1 : t2, in every case.
template <class M>
__device__ __forceinline__ void sphere_update(const double4 g, const D3 org, const D3 dir, const int i,
                                              double& dis, int& hit_object) {
    const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);  // :198
    const double b = dot(p_o, dir);                            // :199
    const double D4 = b * b - dot(p_o, p_o) + g.w;             // :200
    if (__builtin_amdgcn_ballot_w64(D4 >= 0.0) == 0) return;   // whole wave misses (or NaN rays)
    const double sq = M::sqrt64(D4);                           // :205
    const double t1 = b - sq, t2 = b + sq;
    const double t = (t1 > 0.001) ? t1 : t2;
    const bool accept = (t < dis) && !(t < (double)1e-5f);
    dis = accept ? t : dis;
    hit_object = accept ? i : hit_object;
}

// The acceptance update of one sphere (select-only form, see sphere_update): t = t1 > 0.001 ? t1 : t2;
// accept = t < dis && !(t < 1e-5f); the nearest hit and its index follow.
__device__ __forceinline__ void accept_update(const double b, const double sq, const int index, double& dis,
                                              int& hit_object) {
    const double t1 = b - sq, t2 = b + sq;
#if RTM_OPT_SEL64
    // masks straight from the compares (SGPR pairs); NaN t: "t < dis" is false, so nothing is accepted
    const double t = sel_f64_mask(__builtin_amdgcn_ballot_w64(t1 > 0.001), t1, t2);
    const unsigned long long accept =
        __builtin_amdgcn_ballot_w64(t < dis) & ~__builtin_amdgcn_ballot_w64(t < (double)1e-5f);
    dis = sel_f64_mask(accept, t, dis);
    hit_object = sel_index_mask(accept, index, hit_object);
#else
    const double t = (t1 > 0.001) ? t1 : t2;
    const bool accept = (t < dis) && !(t < (double)1e-5f);
    dis = accept ? t : dis;
    hit_object = accept ? index : hit_object;
#endif
}

// The K acceptance updates of a chunk, NEAR ROOTS FIRST.  Intersect returns t1 when t1 > 0.001 and otherwise the far root
// t2 — unless t2 < 1e-5f (or NaN), a miss.  A far root is a rare thing: the ray starts inside the sphere or within a
// thousandth of its surface, heading in.  When NO lane of the wave has one for ANY sphere of the chunk (one scalar test:
// some t1 <= 0.001 with t2 >= 1e-5f?) the updates are those of the near roots alone,
//     accept = t1 > 0.001 && t1 < dis        (t1 > 0.001 implies !(t1 < 1e-5f); a NaN t1 fails the first compare)
// and with t1 replaced by a QUIET NaN where !(t1 > 0.001) — one select on the high word — the new distance is
// v_min_f64(dis, t1v) (the instruction returns the other operand for a quiet NaN; equal operands: the same bits, and the
// strict compare keeps the lower index) and the index follows the one compare t1v < dis: 8 vector instructions per
// sphere instead of 10, the same (dis, index) in every case.  Otherwise: the general updates, from the same b and sq.
#ifndef RTM_OPT_NEARFIRST
#define RTM_OPT_NEARFIRST 1
#endif
//   PACKID  (the tolerance unit's light-root searches: ONE chunk of at most 7 spheres from index 0, dis = DBL_MAX on entry) the
//           sphere's index rides in the last three bits of its near root — a perturbation of at most 7 ulps, under the light
//           root's 35 — so that v_min_f64 alone carries distance AND index (ties between equal upper bits go to the lower
//           index, as the strict compare did; DBL_MAX's own last bits are 7: "none"): one instruction per sphere instead of
//           three, and three integer instructions to unpack
template <int K, bool PACKID = false>
__device__ __forceinline__ void accept_batch(const double (&b)[K], const double (&sq)[K], const int i0, double& dis,
                                             int& hit_object) {
#if RTM_OPT_NEARFIRST && defined(__HIP_DEVICE_COMPILE__)
    static_assert(!PACKID || K <= 7, "three bits of index, 7 = none");
    double t1v[K];
    unsigned long long far = 0ull;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double t1 = b[k] - sq[k], t2 = b[k] + sq[k];
        const unsigned long long near = __builtin_amdgcn_ballot_w64(t1 > 0.001);
        far |= __builtin_amdgcn_ballot_w64(t2 >= (double)1e-5f) & ~near;
        const uint32_t lo = PACKID ? (((uint32_t)__double2loint(t1) & ~7u) | (uint32_t)k) : (uint32_t)__double2loint(t1);
        t1v[k] = __hiloint2double((int)sel_u32_mask(near, (uint32_t)__double2hiint(t1), 0x7FF80000u), (int)lo);
    }
    if constexpr (PACKID) {
        if (far == 0ull) {
            double best = dis;  // DBL_MAX: index field 7
#pragma unroll
            for (int k = 0; k < K; ++k) {
                double m;
                asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(best), "v"(t1v[k]));
                best = m;
            }
            dis = best;
            hit_object = (int)((((uint32_t)__double2loint(best)) + 1u) & 7u) - 1;  // 7 -> -1
            return;
        }
    }
    if (!PACKID && far == 0ull) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long accept = __builtin_amdgcn_ballot_w64(t1v[k] < dis);
            hit_object = sel_index_mask(accept, i0 + k, hit_object);
            double m;  // (spelled out: the builtin's lowering would first canonicalise an operand it cannot prove quiet)
            asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(dis), "v"(t1v[k]));
            dis = m;
        }
        return;
    }
#endif
#pragma unroll
    for (int k = 0; k < K; ++k) accept_update(b[k], sq[k], i0 + k, dis, hit_object);
}

// Discriminants of spheres whose centre lies ON A COORDINATE AXIS (SceneView::axis_pat; the six walls and the light of the
// shipped Cornell box, every sphere of simpleSetting1.json): with c = (cx, 0, 0) the reference's p_o = c - org is
// (cx - ox, -oy, -oz) exactly, its products with dir are -(oy dy), -(oz dz) exactly and their squares oy oy, oz oz, so
//   b  = ((px dx) + (py dy)) + (pz dz) = ((px dx) - oy dy) - oz dz           (src/SettingData.cpp:199, src/Ray.h Dot)
//   pp = ((px px) + (py py)) + (pz pz) = ((px px) + oy oy) + oz oz           (:200)
// with every rounding where the reference has it — the per-ray products o_k d_k, o_k o_k are shared by all such spheres of
// the scene.  Likewise (0, cy, 0): b = ((py dy) - ox dx) - oz dz (the first sum commutes), pp = (ox ox + py py) + oz oz; and
// (0, 0, cz): b = (pz dz) - (ox dx + oy dy) since (-a) + (-b) = -(a + b) in round-to-nearest, pp = (ox ox + oy oy) + pz pz.
// The only bits that can differ are the SIGNS OF ZEROS (0 - (+0) = +0 where -(+0) = -0): a zero product changes no sum it
// is added to unless the sum is zero itself; b = +-0 gives the same b b, t1, t2 up to a zero's sign, and a zero t is
// rejected by t < 1e-5f either way.  16 instructions -> 10 (x, y) / 8 (z) without contraction.
// The tolerance unit (RTM_TOL) also shares the SUMS of the two foreign products (one rounding placed differently, the same
// error size as the contraction it already has): 11 -> 5.
struct AxisShared {
#if RTM_TOL
    double Pyz, Pxz, Pxy, Qyz, Qxz, Qxy;
#else
    double Px, Py, Pz, Pxy, Qx, Qy, Qz, Qxy;
#endif
    __device__ __forceinline__ AxisShared(const D3 org, const D3 dir) {
        const double px = org.x * dir.x, py = org.y * dir.y, pz = org.z * dir.z;
        const double qx = org.x * org.x, qy = org.y * org.y, qz = org.z * org.z;
#if RTM_TOL
        Pyz = py + pz; Pxz = px + pz; Pxy = px + py;
        Qyz = qy + qz; Qxz = qx + qz; Qxy = qx + qy;
#else
        Px = px; Py = py; Pz = pz; Pxy = px + py;
        Qx = qx; Qy = qy; Qz = qz; Qxy = qx + qy;
#endif
    }
};
// Axis signatures that have an instantiation of the exact-n kernels (nearest_hit: UNROLL <= -1000; the launchers compare
// SceneView::axis_pat and the sphere count), RTM_AXIS_SIGNATURES below: the shipped scenes whose spheres sit on the axes.
// ExampleScene/cornellBoxSetting.json — the light on the y axis and the six wall spheres on +x, -x, +y, -y, +z, -z
// (BASELINE configs[1..3]).  A run-time pattern per sphere (a scalar branch in front of
// every discriminant) was measured first and lost: 130.8 -> 139.4 ms for the tolerance row, 160.8 -> 161.6 ms for the exact
// kernel (profiles/r4/axis_ab.txt) — the patterns' SGPRs spill to VGPR lanes and the chunk stops being one basic block.
constexpr unsigned kAxisSigCornell7 = 2u | (1u << 2) | (1u << 4) | (2u << 6) | (2u << 8) | (3u << 10) | (3u << 12);
// ExampleScene/simpleSetting1.json (BASELINE configs[0]): a sphere at the origin (no pattern), two on the x axis, two on the y axis
constexpr unsigned kAxisSigSimple5 = (1u << 2) | (1u << 4) | (2u << 6) | (2u << 8);
// settingData.json: the light on the y axis, a sphere at the origin, one on the x axis
constexpr unsigned kAxisSigSetting3 = 2u | (1u << 4);
constexpr int axis_unroll(const int n, const unsigned sig) { return -(1000 + n + 8 * (int)sig); }
// (n, signature) pairs with an instantiation: X(n, sig) for each
#define RTM_AXIS_SIGNATURES(X) X(7, kAxisSigCornell7) X(5, kAxisSigSimple5) X(3, kAxisSigSetting3)
// b and D4 of sphere g (src/SettingData.cpp:198-200); pat: 0 general, 1 / 2 / 3 the centre is on the x / y / z axis
__device__ __forceinline__ void sphere_disc(const double4 g, const unsigned pat, const D3 org, const D3 dir,
                                            const AxisShared& A, double& b, double& D4) {
    if (pat == 1u) {
        const double px = g.x - org.x;
#if RTM_TOL
        b = px * dir.x - A.Pyz;
        D4 = b * b - (px * px + A.Qyz) + g.w;
#else
        b = px * dir.x - A.Py - A.Pz;
        D4 = b * b - (px * px + A.Qy + A.Qz) + g.w;
#endif
    } else if (pat == 2u) {
        const double py = g.y - org.y;
#if RTM_TOL
        b = py * dir.y - A.Pxz;
        D4 = b * b - (py * py + A.Qxz) + g.w;
#else
        b = py * dir.y - A.Px - A.Pz;
        D4 = b * b - (A.Qx + py * py + A.Qz) + g.w;
#endif
    } else if (pat == 3u) {
        const double pz = g.z - org.z;
        b = pz * dir.z - A.Pxy;
        D4 = b * b - (A.Qxy + pz * pz) + g.w;
    } else {
        const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);  // src/SettingData.cpp:198
        b = dot(p_o, dir);                                          // :199
        D4 = b * b - dot(p_o, p_o) + g.w;                           // :200
    }
}

// K consecutive spheres starting at i0 as ONE basic block: K independent Intersect evaluations
// (independent dependency chains the scheduler interleaves), their square roots behind a single
// wave-uniform guard, then the K acceptance updates in index order (strict <: the lowest index still
// wins ties).  Select-only form, see sphere_update for the equivalence with the statement order.
// profiles/ubench/nearest_variants.hip: 36 % fewer cycles per cast than the per-sphere loop.
// EARLY_OUT (scenes of many small spheres): when no lane of the wave has D4 >= 0 for any sphere of
// the chunk — the common case there — the square roots and updates are skipped altogether.
template <class M, int K, bool EARLY_OUT = false>
__device__ __forceinline__ void sphere_chunk_g(const double4 (&g)[K], const int i0, const D3 org, const D3 dir,
                                               double& dis, int& hit_object) {
    double b[K], D4[K], sq[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const D3 p_o = d3(g[k].x - org.x, g[k].y - org.y, g[k].z - org.z);  // src/SettingData.cpp:198
        b[k] = dot(p_o, dir);                                               // :199
        D4[k] = b[k] * b[k] - dot(p_o, p_o) + g[k].w;                       // :200
    }
    if constexpr (EARLY_OUT) {
        // "some D4 >= 0" as one compare of the largest discriminant (v_max_f64 skips NaN operands; a
        // NaN discriminant, literal mode, is "not >= 0": it can never be accepted; -0 counts as >= 0)
        double top = D4[0];
#pragma unroll
        for (int k = 1; k < K; ++k) top = __builtin_fmax(top, D4[k]);
        if (__builtin_amdgcn_ballot_w64(top >= 0.0) == 0) return;
        // rare from here on (scenes of many small spheres): one sphere at a time keeps the register
        // footprint of this path — and with it the occupancy of the whole kernel — small
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (__builtin_amdgcn_ballot_w64(D4[k] >= 0.0) == 0) continue;
            const double sqk = M::sqrt64(D4[k]);  // :205 (D4 < 0 gives NaN: no hit)
            const double t1 = b[k] - sqk, t2 = b[k] + sqk;
            const double t = (t1 > 0.001) ? t1 : t2;
            const bool accept = (t < dis) && !(t < (double)1e-5f);
            dis = accept ? t : dis;
            hit_object = accept ? i0 + k : hit_object;
        }
        return;
    }
    M::template sqrt64_batch_hit<K>(D4, sq);                       // :205 (D4 < 0 gives NaN: no hit)
#pragma unroll
    for (int k = 0; k < K; ++k) accept_update(b[k], sq[k], i0 + k, dis, hit_object);
}
// spheres whose geometry is fetched together inside a chunk (profiles/r1: 7 at once 202.7 ms, 4: 195.6,
// 3: 194.5, 2: 194.2 ms on the same box; SGPR spills 44 -> 10)
constexpr int kGeomPhase = 2;
template <class M, int K, class Scene, bool EARLY_OUT = false, unsigned SIG = 0u>
__device__ __forceinline__ void sphere_chunk(const Scene& sc, const int i0, const D3 org, const D3 dir,
                                             double& dis, int& hit_object) {
    static_assert(SIG == 0u || (K > kGeomPhase && K < 8 && !EARLY_OUT), "an axis signature names the spheres of ONE exact chunk");
    constexpr int PH = kGeomPhase;
    if constexpr (K > PH && !EARLY_OUT) {
        // The geometry of a chunk arrives by scalar loads, 8 SGPRs per sphere; fetching all K spheres at
        // once pins 8K SGPRs across the discriminants and makes hipcc spill the kernel's long-lived
        // scalars to VGPR lanes (v_writelane / v_readlane in the hot loop).  Phases of PH spheres keep
        // the batch structure of the rest (K independent chains into ONE sqrt guard and K selects).
        double b[K], D4[K], sq[K];
        if constexpr (SIG != 0u) {  // the scene's axis signature is a compile-time constant: sphere_disc's branches fold
            const AxisShared A(org, dir);
#ifndef RTM_AXIS_PHASE
#define RTM_AXIS_PHASE 2  // spheres whose geometry is fetched together in the axis form (A/B knob; an axis sphere needs 4 SGPRs, not 8)
#endif
            constexpr int PA = RTM_AXIS_PHASE;
#pragma unroll
            for (int k0 = 0; k0 < K; k0 += PA) {
                double4 g[PA];
#pragma unroll
                for (int k = 0; k < PA; ++k)
                    if (k0 + k < K) g[k] = sc.geom_uniform(i0 + k0 + k);
#pragma unroll
                for (int k = 0; k < PA; ++k)
                    if (k0 + k < K) sphere_disc(g[k], (SIG >> (2 * (k0 + k))) & 3u, org, dir, A, b[k0 + k], D4[k0 + k]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // (the tolerance unit's axis-signature instantiations are launched for compact scenes only: light roots, above)
            M::template sqrt64_batch_hit<K, (RTM_TOL != 0) && (RTM_TOL_LIGHT_ROOTS != 0)>(D4, sq);
            accept_batch<K, (RTM_TOL != 0) && (RTM_TOL_LIGHT_ROOTS != 0) && (RTM_TOL_PACKID != 0)>(b, sq, i0, dis, hit_object);
            return;
        }
#pragma unroll
        for (int k0 = 0; k0 < K; k0 += PH) {
            double4 g[PH];
#pragma unroll
            for (int k = 0; k < PH; ++k)
                if (k0 + k < K) g[k] = sc.geom_uniform(i0 + k0 + k);
#pragma unroll
            for (int k = 0; k < PH; ++k)
                if (k0 + k < K) {
                    const D3 p_o = d3(g[k].x - org.x, g[k].y - org.y, g[k].z - org.z);  // src/SettingData.cpp:198
                    b[k0 + k] = dot(p_o, dir);                                          // :199
                    D4[k0 + k] = b[k0 + k] * b[k0 + k] - dot(p_o, p_o) + g[k].w;        // :200
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (K >= 8)
            M::template sqrt64_batch<K>(D4, sq);  // :205 (D4 < 0 gives NaN: no hit)
        else
            M::template sqrt64_batch_hit<K>(D4, sq);
        accept_batch<K>(b, sq, i0, dis, hit_object);
        return;
    }
    double4 g[K];
#pragma unroll
    for (int k = 0; k < K; ++k) g[k] = sc.geom_uniform(i0 + k);
    sphere_chunk_g<M, K, EARLY_OUT>(g, i0, org, dir, dis, hit_object);
}

// png::PlaneObject::Intersect as this build completes it (plane_test above) fused with the caller's acceptance test
// (src/Renderer.cpp:67), select-only.  plane_test returns false on  |n.d| < FLT_EPSILON,  !(t > 0.001),  |d.right| >
// right.right  and  |d.upv| > upv.upv  in that order; none of the tests has a side effect, so
//   hit == !(|n.d| < eps) && (t > 0.001) && !(|d.right| > rr) && !(|d.upv| > uu)
// with every comparison false for a NaN operand exactly as in the statement form, and the caller's "t < dis && t > 0"
// adds "t < dis" (t > 0.001 already holds).  `pl` is wave-uniform: its 14 doubles arrive by scalar loads.
template <class M>
__device__ __forceinline__ void plane_update(const double* pl_global, const D3 org, const D3 dir, const int index,
                                             double& dis, int& hit_object) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) double* ConstF64Ptr;
    ConstF64Ptr pl = (ConstF64Ptr)(unsigned long long)pl_global;
#else
    const double* pl = pl_global;
#endif
    const D3 pos = d3(pl[0], pl[1], pl[2]), nrm = d3(pl[3], pl[4], pl[5]);
    const double dn = dot(nrm, dir);
    const double tt = M::div(dot(nrm, pos - org), dn);
    const D3 d = (org + dir * tt) - pos;
    const D3 right = d3(pl[6], pl[7], pl[8]), upv = d3(pl[9], pl[10], pl[11]);
    const double dr = dot(d, right), du = dot(d, upv);
    const bool hit = !(fabs(dn) < (double)FLT_EPSILON) && (tt > 0.001) && !(fabs(dr) > pl[12]) && !(fabs(du) > pl[13]);
    const bool accept = hit && (tt < dis);
    dis = accept ? tt : dis;
    hit_object = accept ? index : hit_object;
}

// K consecutive OBJECTS of a scene that holds planes.  The type of object i is wave-uniform (the sign of its geometry
// row's fourth entry), so every per-object decision below is a scalar branch: spheres keep the batch structure of
// sphere_chunk (independent discriminant chains, ONE guard for the K square roots — a plane's slot carries NaN, which
// the unscaled sequence passes through and no compare accepts), and the acceptance updates run in index order with a
// plane's own test in its slot, so strict < still lets the lowest index win ties across both types.
template <class M, int K, class Scene>
__device__ __forceinline__ void object_chunk(const Scene& sc, const int i0, const D3 org, const D3 dir, double& dis,
                                             int& hit_object) {
    double b[K], D4[K], sq[K];
    bool plane[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double4 g = sc.geom_uniform(i0 + k);
        plane[k] = __double2hiint(g.w) < 0;  // wave-uniform
        if (plane[k]) {
            b[k] = 0.0;
            D4[k] = __builtin_nan("");
        } else {
            const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);  // src/SettingData.cpp:198
            b[k] = dot(p_o, dir);                                      // :199
            D4[k] = b[k] * b[k] - dot(p_o, p_o) + g.w;                 // :200
        }
    }
    M::template sqrt64_batch_hit<K>(D4, sq);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (plane[k])
            plane_update<M>(sc.v.plane + (size_t)(i0 + k) * 16, org, dir, i0 + k, dis, hit_object);
        else
            accept_update(b[k], sq[k], i0 + k, dis, hit_object);
    }
}

// ------------------------------------------------------------------------------------------------
// The nearest-hit loop of src/Renderer.cpp:58-73 for LARGE scenes, through a uniform grid: the same (hit object, dis)
// as the loop over all objects, from a fraction of its Intersect calls.
//
// Why it is the same.  The loop's result is the lexicographic minimum of (t, index) over the spheres whose Intersect
// reports a hit with t > 0 (strict `t < dis` in index order = the lowest index among equal t), so any order of testing
// gives it as long as ties are broken by index and every sphere that could win IS tested.  Intersect's t is a root of
// t^2 - 2bt + (|c - o|^2 - r^2), so the point o + t d lies at distance sqrt(r^2 + t^2 (d.d - 1)) of the centre:
// on the sphere but for the direction's float-normalised length (|d.d - 1| <= 1.8e-7, src/Ray.h:67-72) and rounding.
// (The reference's loop in effect sees every sphere inflated to that radius: a far, tiny sphere is "hit" by rays that
// pass it at up to sqrt(|d.d - 1|) t.)  Each sphere is listed in every cell that lies within r_i + pad_i of its centre,
// pad_i = 0.05 h + (sqrt(r_i^2 + dd_tol t_ok^2) - r_i) + 1e-6 t_ok, which covers that excess for every t <= t_ok and
// |d.d - 1| <= dd_tol = 4e-7, the rounding of t itself (<= 1e-7 t for a grazing hit), the DDA's own rounding (1e-12)
// by ten orders of magnitude, and a sphere hit in a cell the DDA cuts at a corner (it is listed in the neighbours too).
// So when the walk stops at a cell whose exit parameter is >= dis, every sphere with a hit before that exit has been
// tested, and a ray that misses the grid's box (the padded boxes' union) hits nothing.  A hit's parameter is at most
// |c - o| |d| + sqrt(|c - o|^2 (d.d - 1) + r^2) <= 1.001 x the distance from o to the farthest point of the box, so
// t <= t_ok holds for every origin within reach = t_ok / 1.001 - half the box's diagonal of the box's centre.  Rays
// outside those bounds (a far-away origin; a direction that is not unit length: the reference's arithmetic then "hits"
// spheres the geometric ray passes at a distance) take the exhaustive loop; rays with a non-finite component hit
// nothing in the reference loop either (every t is NaN or infinite).
// Spheres in `big` (padded box over too many cells) are tested for every ray.
#ifndef RTM_GRID_K
#define RTM_GRID_K 4
#endif
#ifndef RTM_GRID_SPILL
#define RTM_GRID_SPILL 1
#endif
constexpr int kGridBatch = RTM_GRID_K;  // records in flight per trip of the walk (profiles/r3/grid_crossover.txt)
#ifndef RTM_GRID_SHADE_AT
#define RTM_GRID_SHADE_AT 5
#endif
constexpr int kGridShadeAt8 = RTM_GRID_SHADE_AT;  // the render loop shades when this many eighths of a wave's lanes have finished their walks

// EXPERIMENT build (-DRTM_GRID_EXP_OCC): how often each region of the grid kernel runs and with how many lanes
// (profiles/r3/grid_occupancy.txt); the counters are printed and cleared by grid_finalize_kernel.
#ifdef RTM_GRID_EXP_OCC
__device__ unsigned long long g_grid_occ[32];
#define RTM_GRID_OCC(region)                                                                          \
    do {                                                                                              \
        const unsigned long long occ_m = __builtin_amdgcn_ballot_w64(true);                           \
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(occ_m)) {                                      \
            atomicAdd(&g_grid_occ[2 * (region)], 1ull);                                               \
            atomicAdd(&g_grid_occ[2 * (region) + 1], (unsigned long long)__builtin_popcountll(occ_m)); \
        }                                                                                             \
    } while (0)
#else
#define RTM_GRID_OCC(region) do { } while (0)
#endif

// One ray's walk through the grid, as per-lane state that can be advanced a trip at a time: the render kernel's lanes
// walk independently and are shaded in groups (rtm_grid_kernel.h), the probe runs it to the end.
template <class M, class Scene, bool COUNT = false>
struct GridWalk {
    double dis;  // the nearest accepted hit so far (DBL_MAX: none) and its object
    int best;
    double tmx, tmy, tmz;  // parameter at which the ray leaves the NEXT cell's predecessor planes (Amanatides & Woo's tMax)
    double tdx, tdy, tdz;  // parameter per cell along each axis, signed like the direction (inf: never)
    double t_exit;         // parameter at which the ray leaves the current cell
    int ix, iy, iz;        // the NEXT cell
    unsigned j, jend, nj, nje;  // records left in the current cell; the next cell's list
    bool next_ok;               // the next cell is inside the grid
    unsigned tests, steps;      // (COUNT: the probe's diagnostics, rtm_debug_grid_nearest)
    // Candidates — records whose discriminant came out >= 0 — wait in LDS for their square root: one lane in seven per
    // test, so the root taken on the spot was paid by the whole wave in 73 % of the test slots for 5 lanes
    // (profiles/r3/grid_occupancy.txt); queued, every trip takes ONE root for all lanes that have a candidate.
    // kGridBatch entries per lane, [entry][lane]: b, D4 (doubles) and the sphere's index.
    unsigned pending;  // this lane's queued candidates
    double* q_b;       // wave-uniform LDS pointers, already offset by the lane; entries `q_stride` elements apart
    double* q_d;
    unsigned* q_i;
    int q_stride;
    static constexpr size_t queue_bytes(int lanes) { return (size_t)kGridBatch * lanes * (2 * sizeof(double) + sizeof(unsigned)); }
    __device__ __forceinline__ void attach_queue(unsigned char* lds, int lanes, int lane) {
        q_b = reinterpret_cast<double*>(lds) + lane;
        q_d = q_b + (size_t)kGridBatch * lanes;
        q_i = reinterpret_cast<unsigned*>(reinterpret_cast<double*>(lds) + (size_t)2 * kGridBatch * lanes) + lane;
        q_stride = lanes;
    }

#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) GridHeader* HdrPtr;
    static __device__ __forceinline__ HdrPtr header(const Scene& sc) { return (HdrPtr)(unsigned long long)sc.v.grid; }
#else
    typedef const GridHeader* HdrPtr;
    static HdrPtr header(const Scene& sc) { return sc.v.grid; }
#endif

    __device__ __forceinline__ void consider(const int i, const double4 g, const D3 org, const D3 dir) {
        if constexpr (COUNT) ++tests;
        const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);  // src/SettingData.cpp:198
        const double b = dot(p_o, dir);                            // :199
        const double D4 = b * b - dot(p_o, p_o) + g.w;             // :200
        if (D4 >= 0.0) {                                           // :202 (a NaN D4 ends in a NaN t: never accepted)
            RTM_GRID_OCC(7);
            const double sq = M::sqrt64(D4);
            const double t1 = b - sq, t2 = b + sq;
            const double t = (t1 > 0.001) ? t1 : t2;  // :212-223; accepted as in sphere_update, ties to the lower index
            const bool accept = !(t < (double)1e-5f) && (t < dis || (t == dis && i < best));
            dis = accept ? t : dis;
            best = accept ? i : best;
        }
    }
    // A png::PlaneObject among the objects every ray tests (this build's completion of its Intersect: plane_test, literal
    // statement order, the compiler's division) with the caller's acceptance (src/Renderer.cpp:67) and ties to the lower index
    __device__ __forceinline__ void consider_plane(const double* __restrict__ pl, const int i, const D3 org, const D3 dir) {
        if constexpr (COUNT) ++tests;
        double t;
        if (plane_test(pl, org, dir, t)) {
            const bool accept = (t > 0.0) && (t < dis || (t == dis && i < best));
            dis = accept ? t : dis;
            best = accept ? i : best;
        }
    }
    // The step out of the current cell: the axis whose boundary comes first (a NaN among the three falls through to z:
    // the walk still ends, each index moves one way only), the cell's exit parameter, and the next cell's offsets.
    __device__ __forceinline__ void plan_next(HdrPtr G) {
        const int nx = G->dim[0], ny = G->dim[1], nz = G->dim[2];
        const bool ax = tmx <= tmy && tmx <= tmz, ay = !ax && tmy <= tmz, az = !ax && !ay;
        t_exit = ax ? tmx : (ay ? tmy : tmz);
        ix += ax ? (tdx < 0.0 ? -1 : 1) : 0;
        iy += ay ? (tdy < 0.0 ? -1 : 1) : 0;
        iz += az ? (tdz < 0.0 ? -1 : 1) : 0;
        tmx += ax ? fabs(tdx) : 0.0;
        tmy += ay ? fabs(tdy) : 0.0;
        tmz += az ? fabs(tdz) : 0.0;
        next_ok = (unsigned)ix < (unsigned)nx && (unsigned)iy < (unsigned)ny && (unsigned)iz < (unsigned)nz;
        const unsigned cell = next_ok ? ((unsigned)iz * ny + iy) * nx + ix : 0u;
        const uint2 range = G->cell_range[cell];
        nj = range.x;
        nje = range.y;
    }
    // Start the walk of (org, dir): the big list, the box, the guards.  Returns whether there is a walk to advance
    // (false: dis / best are final — a miss of the box, a non-finite ray, or the exhaustive loop has run).
    __device__ __forceinline__ bool begin(const Scene& sc, const D3 org, const D3 dir) {
        HdrPtr G = header(sc);
        RTM_GRID_OCC(0);
        best = -1;
        dis = DBL_MAX;
        pending = 0u;
        if constexpr (COUNT) tests = steps = 0;
        const int n_big = G->n_big;
        const int* big = G->big;
        for (int k = 0; k < n_big; ++k) {  // the objects every ray tests: spheres that span the scene, and planes
            const int i = __builtin_amdgcn_readfirstlane(big[k]);
            const double4 g = sc.geom_uniform(i);
            if (sc.v.plane != nullptr && __double2hiint(g.w) < 0)  // wave-uniform: object i is a png::PlaneObject
                consider_plane(sc.v.plane + (size_t)i * 16, i, org, dir);
            else
                consider(i, g, org, dir);
        }
        bool live = __builtin_isfinite(org.x) && __builtin_isfinite(org.y) && __builtin_isfinite(org.z) &&
                    __builtin_isfinite(dir.x) && __builtin_isfinite(dir.y) && __builtin_isfinite(dir.z);
        // the ray's parameter interval inside the grid's box
        double t0 = 0.0, t_out = DBL_MAX;
        const double inf = __builtin_huge_val();
        // 1 / d for the walk's set-up: v_rcp_f64 + two Newton steps (2^-52 relative; the compiler's correctly rounded
        // division is three times the instructions, and nothing here needs the last bit: the pads have ten orders of
        // magnitude of room), the true division where the exponent is extreme
        const bool tame = M::moderate(dir.x) && M::moderate(dir.y) && M::moderate(dir.z);
        const bool slow_div = __builtin_amdgcn_ballot_w64(!tame) != 0;
        auto slab = [&](const double o, const double d, const double lo, const double hi, double& inv) {
            const bool flat = d == 0.0;
            if (slow_div) {
                inv = 1.0 / d;
            } else {
                double r = __builtin_amdgcn_rcp(d);
                double e = __builtin_fma(-d, r, 1.0);
                r = __builtin_fma(r, e, r);
                e = __builtin_fma(-d, r, 1.0);
                inv = __builtin_fma(r, e, r);
            }
            const double ta = (lo - o) * inv, tb = (hi - o) * inv;
            const double tn = flat ? ((o >= lo && o <= hi) ? -inf : inf) : (ta < tb ? ta : tb);
            const double tf = flat ? inf : (ta < tb ? tb : ta);
            t0 = tn > t0 ? tn : t0;
            t_out = tf < t_out ? tf : t_out;
        };
        const double lox = G->lo[0], loy = G->lo[1], loz = G->lo[2];
        double invx, invy, invz;
        slab(org.x, dir.x, lox, G->hi[0], invx);
        slab(org.y, dir.y, loy, G->hi[1], invy);
        slab(org.z, dir.z, loz, G->hi[2], invz);
        const D3 oc = d3(org.x - G->cb[0], org.y - G->cb[1], org.z - G->cb[2]);
        const bool exhaustive = live && !(dot(oc, oc) <= G->reach2 && fabs(dot(dir, dir) - 1.0) <= G->dd_tol);
        live = live && (t0 <= t_out);
        if (__builtin_amdgcn_ballot_w64(exhaustive) != 0) {  // (never, for a camera within two scene diagonals and unit directions)
            const int n = sc.n();
            for (int i = 0; i < n; ++i) {
                const double4 g = sc.geom_uniform(i);
                if (!exhaustive) continue;
                if (sc.v.plane != nullptr && __double2hiint(g.w) < 0) {  // (in the big list too: tested twice, same result)
                    consider_plane(sc.v.plane + (size_t)i * 16, i, org, dir);
                    continue;
                }
                consider(i, g, org, dir);
            }
        }
        live = live && !exhaustive;
        // 3D-DDA over the cells from the entry point on (Amanatides & Woo).  The search is bound by memory latency as much
        // as by arithmetic (a cell's list is a dependent load behind the cell's offsets), so every trip has ONE round of
        // loads in flight: up to kGridBatch records of the current cell (self-contained: centre, r*r and the sphere's index
        // in r*r's 29 zero mantissa bits) and the list offsets of the NEXT cell on the ray, which do not depend on the tests.
        const double h = G->h, inv_h = G->inv_h;
        const int nx = G->dim[0], ny = G->dim[1], nz = G->dim[2];
        auto cell_of = [&](const double p, const double lo, const int dim) {
            const int c = (int)floor((p - lo) * inv_h);
            return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
        };
        ix = cell_of(org.x + dir.x * t0, lox, nx);
        iy = cell_of(org.y + dir.y * t0, loy, ny);
        iz = cell_of(org.z + dir.z * t0, loz, nz);
        tmx = dir.x == 0.0 ? inf : ((lox + (double)(ix + (dir.x > 0.0 ? 1 : 0)) * h) - org.x) * invx;
        tmy = dir.y == 0.0 ? inf : ((loy + (double)(iy + (dir.y > 0.0 ? 1 : 0)) * h) - org.y) * invy;
        tmz = dir.z == 0.0 ? inf : ((loz + (double)(iz + (dir.z > 0.0 ? 1 : 0)) * h) - org.z) * invz;
        tdx = dir.x == 0.0 ? inf : h * invx;  // (signed: the direction of the index step rides on the sign)
        tdy = dir.y == 0.0 ? inf : h * invy;
        tdz = dir.z == 0.0 ? inf : h * invz;
        j = jend = nj = nje = 0u;
        t_exit = 0.0;
        next_ok = false;
        if (live) {
            const unsigned cell = ((unsigned)iz * ny + iy) * nx + ix;
            const uint2 range = G->cell_range[cell];
            j = range.x;
            jend = range.y;
            plan_next(G);  // (ix, iy, iz, tm* now describe the NEXT cell; t_exit is the current one's)
        }
        return live;
    }
    // One trip: move on when the current cell is done, then test up to kGridBatch of the cell's records.
    // Returns whether the walk goes on (false: dis / best are final).
    __device__ __forceinline__ bool advance(const Scene& sc, const D3 org, const D3 dir) {
        HdrPtr G = header(sc);
        RTM_GRID_OCC(1);
        if (pending == 0u) {
            if (j >= jend) {
                RTM_GRID_OCC(2);
                // this cell is done and its candidates are settled: every sphere that can be hit before its exit has been
                // tested (see above)
                if (dis <= t_exit || !next_ok) return false;
                if constexpr (COUNT) ++steps;
                j = nj;
                jend = nje;
                plan_next(G);
            }
            // Up to kGridBatch of the cell's records, all loads first.  The loads and tests are per lane (a lane with two
            // records left issues two loads): the walk is bound by the vector memory pipeline, which pays per lane and
            // distinct line (profiles/r3/grid_perturbation.txt: the same loads issued twice cost +31 %, the tests'
            // arithmetic twice +8 %), so nothing is fetched that is not needed.
#if RTM_GRID_SPILL
            // (round 4, profiles/r4/grid_spill.txt: 201.0 -> 197.1 ms for the configs[4] frame; RTM_GRID_SPILL=0 is the twin)
            // A lane whose cell has fewer than kGridBatch records left fills its free slots with the first records of the
            // NEXT cell on its ray (their range is known already), unless the walk is known to end in this cell.  Testing
            // a sphere early, or one the walk would never have reached, changes nothing (see above: any order, ties by
            // index); it makes the slots fuller and the next cell shorter.
            const unsigned left1 = jend - j < (unsigned)kGridBatch ? jend - j : (unsigned)kGridBatch;
            const bool spill = left1 < (unsigned)kGridBatch && next_ok && !(dis <= t_exit);
            const unsigned room = (unsigned)kGridBatch - left1, avail = nje - nj;
            const unsigned left2 = spill ? (avail < room ? avail : room) : 0u;
            const unsigned left = left1 + left2;
            double4 r[kGridBatch];
#pragma unroll
            for (unsigned k = 0; k < (unsigned)kGridBatch; ++k)
                if (left > k) r[k] = G->recs[k < left1 ? j + k : nj + (k - left1)];
            nj += left2;
#pragma unroll
            for (unsigned k = 0; k < (unsigned)kGridBatch; ++k) {
                if (left > k) {
#else
            const double4* recs = G->recs + j;
            const unsigned left = jend - j;
            double4 r[kGridBatch];
#pragma unroll
            for (unsigned k = 0; k < (unsigned)kGridBatch; ++k)
                if (left > k) r[k] = recs[k];
#pragma unroll
            for (unsigned k = 0; k < (unsigned)kGridBatch; ++k) {
                if (left > k) {
#endif
                    RTM_GRID_OCC(3 + k);
                    if constexpr (COUNT) ++tests;
                    // src/SettingData.cpp:198-200; the sphere's index rides in the low 29 mantissa bits of the float-valued r*r
                    const unsigned long long w = (unsigned long long)__double_as_longlong(r[k].w);
                    const D3 p_o = d3(r[k].x - org.x, r[k].y - org.y, r[k].z - org.z);
                    const double b = dot(p_o, dir);
                    const double D4 = b * b - dot(p_o, p_o) + __longlong_as_double((long long)(w & ~0x1FFFFFFFull));
                    if (D4 >= 0.0) {  // :202 (a NaN D4 would end in a NaN t: never accepted)
                        const unsigned at = pending * (unsigned)q_stride;
                        q_b[at] = b;
                        q_d[at] = D4;
                        q_i[at] = (unsigned)w & 0x1FFFFFFFu;
                        ++pending;
                    }
                }
            }
#if RTM_GRID_SPILL
            j += left1;
#else
            j = left > (unsigned)kGridBatch ? j + (unsigned)kGridBatch : jend;
#endif
        }
        if (pending != 0u) {  // one candidate per lane and trip: :205-223 and the caller's acceptance (src/Renderer.cpp:67)
            RTM_GRID_OCC(7);
            --pending;
            const unsigned at = pending * (unsigned)q_stride;
            const double b = q_b[at], D4 = q_d[at];
            const int i = (int)q_i[at];
            const double sq = M::sqrt64(D4);
            const double t1 = b - sq, t2 = b + sq;
            const double t = (t1 > 0.001) ? t1 : t2;  // accepted as in sphere_update, ties to the lower index
            const bool accept = !(t < (double)1e-5f) && (t < dis || (t == dis && i < best));
            dis = accept ? t : dis;
            best = accept ? i : best;
        }
        // the walk's end seen in the trip that settles it, not by the test at the top of the next one: a trip per cast less
        // (round 4, profiles/r4/grid_early_end.txt: 192.1 -> 183.9 ms for the configs[4] frame)
        if (pending == 0u && j >= jend && (dis <= t_exit || !next_ok)) return false;
        return true;
    }
};

template <class M, class Scene, bool COUNT = false>
__device__ __forceinline__ int nearest_hit_grid(const Scene& sc, const D3 org, const D3 dir, double& dis,
                                                unsigned char* queue_lds, int queue_lanes, int queue_lane,
                                                unsigned* n_tests = nullptr, unsigned* n_steps = nullptr) {
    GridWalk<M, Scene, COUNT> W;
    W.attach_queue(queue_lds, queue_lanes, queue_lane);
    bool walking = W.begin(sc, org, dir);
    while (walking) walking = W.advance(sc, org, dir);
    if constexpr (COUNT) {
        *n_tests = W.tests;
        *n_steps = W.steps;
    }
    dis = W.dis;
    return W.best;
}

// src/Renderer.cpp:58-73: brute-force nearest hit; strict < keeps the lowest index on ties.
// UNROLL == 1: the literal loop.  UNROLL > 1: batches of UNROLL spheres (sphere_batch).
template <class M, int UNROLL, class Scene>
__device__ __forceinline__ int nearest_hit(const Scene& sc, const D3 org, const D3 dir, double& dis) {
    int hit_object = -1;
    dis = DBL_MAX;
    const int n = sc.n();
    if constexpr (UNROLL == 0 || UNROLL == 1) {
        for (int i = 0; i < n; ++i) {
            const double4 g = sc.geom_uniform(i);
            double t;
            bool hit;
            if (sc.v.plane != nullptr && g.w < 0.0)  // wave-uniform: object i is a plane
                hit = plane_test(sc.v.plane + (size_t)i * 16, org, dir, t);
            else
                hit = sphere_test<M>(g, org, dir, t);
            if (hit && t < dis && t > 0) {
                dis = t;
                hit_object = i;
            }
        }
    } else {
        // UNROLL < 0: the caller guarantees n < -UNROLL, so there is no full chunk and the loop is not even
        // compiled (it is what sets the kernel's register high-water mark)
        if constexpr (Scene::kPlanes) {  // scenes with planes: chunks of 4 objects plus an exact tail
            int i0 = 0;
            for (; i0 + 4 <= n; i0 += 4) object_chunk<M, 4>(sc, i0, org, dir, dis, hit_object);
            switch (n - i0) {  // wave-uniform
                case 1: object_chunk<M, 1>(sc, i0, org, dir, dis, hit_object); break;
                case 2: object_chunk<M, 2>(sc, i0, org, dir, dis, hit_object); break;
                case 3: object_chunk<M, 3>(sc, i0, org, dir, dis, hit_object); break;
                default: break;
            }
            return hit_object;
        }
        if constexpr (UNROLL <= -1000) {
            // the caller guarantees n == code & 7 and SceneView::axis_pat == code >> 3 (code = -UNROLL - 1000): one exact
            // chunk whose spheres' axis patterns are compile-time constants (sphere_disc; the launcher picks it)
            constexpr int code = -UNROLL - 1000;
            sphere_chunk<M, (code & 7), Scene, false, (unsigned)(code >> 3)>(sc, 0, org, dir, dis, hit_object);
            return hit_object;
        }
        if constexpr (UNROLL <= -101 && UNROLL >= -107) {
            // the caller guarantees n == -UNROLL - 100: the scene is ONE exact chunk, no switch, no other sizes
            // compiled in (scenes under 8 spheres; the launcher picks the instantiation)
            sphere_chunk<M, -UNROLL - 100>(sc, 0, org, dir, dis, hit_object);
            return hit_object;
        }
        constexpr int U = UNROLL <= -101 ? 8 : (UNROLL < 0 ? -UNROLL : UNROLL);  // (UNROLL <= -1000 returned above)
        static_assert(U == 8 || U == 4, "chunks of 8 or 4 plus an exact tail");
        int i0 = 0;
        if constexpr (UNROLL > 0) {
            for (; i0 + U <= n; i0 += U) sphere_chunk<M, U>(sc, i0, org, dir, dis, hit_object);
        }
        switch (n - i0) {  // wave-uniform: exactly one tail chunk, sized to the remainder
            case 1: sphere_chunk<M, 1>(sc, i0, org, dir, dis, hit_object); break;
            case 2: sphere_chunk<M, 2>(sc, i0, org, dir, dis, hit_object); break;
            case 3: sphere_chunk<M, 3>(sc, i0, org, dir, dis, hit_object); break;
            case 4: sphere_chunk<M, 4>(sc, i0, org, dir, dis, hit_object); break;
            case 5: sphere_chunk<M, 5>(sc, i0, org, dir, dis, hit_object); break;
            case 6: sphere_chunk<M, 6>(sc, i0, org, dir, dis, hit_object); break;
            case 7: sphere_chunk<M, 7>(sc, i0, org, dir, dis, hit_object); break;
            default: break;
        }
    }
    return hit_object;
}

#if RTM_TOL
// The reference's nearest-hit loop (src/Renderer.cpp:58-73 over src/SettingData.cpp:197-226) in the reference's OWN
// arithmetic inside the tolerance translation unit: every multiply and add separately rounded (the pragma; nothing here
// goes through the vector helpers, whose operations carry the unit's contraction), IEEE square root.  For PRIMARY rays
// only: the S samples of a sub-pixel share one primary ray (no jitter, src/Renderer.cpp:224-232), and the shipped Cornell
// box is symmetric about a camera that looks down its axis — the seams between two wall spheres project exactly onto the
// image's diagonals, where a primary ray meets two spheres at the SAME distance in real arithmetic and rounding noise
// decides which one wins for a quarter of the pixel's samples at once.  There the contracted arithmetic would differ from
// the reference's by whole fractions of a pixel (146 pixels of the headline frame, up to 0.18), so the kernel settles
// those rays — the ones whose contracted search disagrees with this loop, found once per sub-pixel in the prologue — with
// this loop (rtm_render_kernel.h, prim_fix).
template <class Scene>
__device__ __forceinline__ int nearest_hit_exactfp(const Scene& sc, const D3 org, const D3 dir, double& dis) {
#pragma clang fp contract(off)
    int hit_object = -1;
    dis = DBL_MAX;
    const int n = sc.n();
    for (int i = 0; i < n; ++i) {
        const double4 g = sc.geom_uniform(i);
        const double px = g.x - org.x, py = g.y - org.y, pz = g.z - org.z;  // :198
        const double b = px * dir.x + py * dir.y + pz * dir.z;              // :199
        const double D4 = b * b - (px * px + py * py + pz * pz) + g.w;      // :200
        if (D4 < 0.0) continue;                                             // :202
        const double sq = __builtin_sqrt(D4);                               // :205
        const double t1 = b - sq, t2 = b + sq;
        const double min_value = (double)1e-5f;                             // :208
        if (t1 < min_value && t2 < min_value) continue;                     // :209
        const double t = (t1 > 0.001) ? t1 : t2;                            // :212-223
        if (t < dis && t > 0) {                                             // src/Renderer.cpp:67
            dis = t;
            hit_object = i;
        }
    }
    return hit_object;
}
#endif

#if RTM_TOL
// Could last-bit differences in the arithmetic change what the reference's nearest-hit loop returns for this (primary) ray?
// The loop again in the reference's arithmetic, with a margin on every decision it makes: a discriminant whose sign, a root
// whose place against the 0.001 and 1e-5f thresholds, or a nearest hit whose lead over the runner-up is within 1e-11 of the
// operands' magnitude — five orders above what contraction and one-ulp roots can move them by (~1e-16 of the same
// magnitudes) — is "at risk".  prim_prepass_kernel (rtm_render_kernel.h) evaluates it once per sub-pixel.
template <class Scene>
__device__ __forceinline__ bool primary_tie_risk(const Scene& sc, const D3 org, const D3 dir) {
#pragma clang fp contract(off)
    const double kMargin = 1e-11;
    double best = DBL_MAX, best_m = 0.0, second = DBL_MAX, second_m = 0.0;
    bool risk = !(__builtin_isfinite(dir.x) && __builtin_isfinite(dir.y) && __builtin_isfinite(dir.z));
    const int n = sc.n();
    for (int i = 0; i < n; ++i) {
        const double4 g = sc.geom_uniform(i);
        const double px = g.x - org.x, py = g.y - org.y, pz = g.z - org.z;
        const double b = px * dir.x + py * dir.y + pz * dir.z;
        const double pp = px * px + py * py + pz * pz;
        const double D4 = b * b - pp + g.w;
        const double mag = b * b + pp + g.w;  // what D4's rounding errors scale with
        if (fabs(D4) <= kMargin * mag) risk = true;
        if (!(D4 >= 0.0)) continue;
        const double sq = __builtin_sqrt(D4);
        const double t1 = b - sq, t2 = b + sq;
        // a root moves by at most dD4 / (2 sq) + rounding of b and sq
        const double m = kMargin * (mag / (sq > 1e-300 ? sq : 1e-300) + fabs(b) + sq);
        if (fabs(t1 - 0.001) <= m || fabs(t1 - (double)1e-5f) <= m || fabs(t2 - (double)1e-5f) <= m) risk = true;
        if (t1 < (double)1e-5f && t2 < (double)1e-5f) continue;
        const double t = (t1 > 0.001) ? t1 : t2;
        if (!(t > 0)) continue;
        if (t < best) {
            second = best;
            second_m = best_m;
            best = t;
            best_m = m;
        } else if (t < second) {
            second = t;
            second_m = m;
        }
    }
    if (second < DBL_MAX && second - best <= best_m + second_m) risk = true;
    return risk;
}
#endif

struct PathCounters {
    unsigned casts, bounces, draws;
};

// The part of one PathTracing invocation (src/Renderer.cpp:57-117) after the nearest-hit loop:
// `id`/`dis` are that loop's result.  PURE: the caller's ray, depth and RNG stream come in by value and the
// bounce goes out through `out`, which is written only when the path continues (returns true); when it ended,
// `term` is the value the deepest invocation returned.  Keeping the inputs untouched is what lets the
// speculative form below re-run the block from the same operands without saving them first — and it keeps
// hipcc from copying the whole ray at the entry of every nested region (24 v_mov per trip before).
// `m` is a math policy object (MathRefI, MathFastI or MathSpec).
struct ShadeOut {
    D3 org, dir;     // the bounced ray (src/Renderer.cpp:108)
    uint32_t ctr;    // RNG counter after this invocation's draws
    int draws;       // 0, 1 or 3
};
// The bounce itself (src/Renderer.cpp:79-108): everything after the Russian-roulette test has passed.  `rng` is
// the stream right after the RR draw; always continues.
template <class MI, class Scene>
__device__ __forceinline__ void path_bounce_core(MI& m, const Scene& sc, const int id, const double dis, const int mode,
                                                 const D3 org, const D3 dir, RngStream rng, ShadeOut& out);

template <class MI, class Scene>
__device__ __forceinline__ bool path_shade_core(MI& m, const Scene& sc, const int id, const double dis,
                                                const int mode, const int max_bounces, const D3 org, const D3 dir,
                                                const int depth, RngStream rng, D3& term, ShadeOut& out) {
    out.draws = 0;
    if (id < 0) {  // :116
        term = d3(0, 0, 0);
        return false;
    }
    const D3 emission = sc.emission(id);
    if (max_bounces >= 0 && depth >= max_bounces) {  // build extension: no draw
        term = emission;
        return false;
    }
    out.draws = 1;
    // :78, kd() is a float widened to double.  u <= kd  <=>  m <= kd * 2^24 with u = m * 2^-24 (both sides scaled by
    // an exact power of two)
    if (!(rng_next_m(rng) <= sc.kd24(id))) {
        term = emission;                  // :112
        return false;
    }
    path_bounce_core(m, sc, id, dis, mode, org, dir, rng, out);
    return true;
}

template <class MI, class Scene>
__device__ __forceinline__ void path_bounce_core(MI& m, const Scene& sc, const int id, const double dis, const int mode,
                                                 const D3 org, const D3 dir, RngStream rng, ShadeOut& out) {
    const D3 hit_point = dir * dis + org;  // :79
    // D2: in literal mode the caller's normal stays (0,0,0); repaired: src/SettingData.cpp:214-215
    D3 normal = d3(0, 0, 0);
    if (mode != RTM_MODE_LITERAL) {
        if constexpr (Scene::kPlanes) {  // the fast kernels' instantiation for scenes that hold planes (MathSpecZ)
            const bool is_plane = sc.is_plane(id);
            // a plane lane's "hit - position" is no sphere radius: whatever its Normalize trips must not count
            [[maybe_unused]] bool bad_before = false;
            if constexpr (is_spec<MI>::value) bad_before = m.bad;
            const D3 nsphere = normalize_i(m, hit_point - sc.center(id));
            if constexpr (is_spec<MI>::value) m.bad = bad_before || (m.bad && !is_plane);
            const D3 pn = sc.plane_normal(id);
            normal = is_plane ? pn : nsphere;  // PlaneObject: out_normal = m_normal
        } else if (!Scene::kNeverPlanes && sc.v.plane != nullptr) {  // wave-uniform: the scene holds planes (per-object kernel, MathRefI)
            const bool is_plane = sc.v.geom[id].w < 0.0;
            const double* pl = sc.v.plane + (size_t)id * 16;
            const D3 nsphere = normalize_i(m, hit_point - sc.center(id));
            normal = is_plane ? d3(pl[3], pl[4], pl[5]) : nsphere;  // PlaneObject: out_normal = m_normal
            if constexpr (is_spec<MI>::value) m.bad = m.bad | is_plane;  // its shortcuts assume spheres
        } else {
            const D3 dv = hit_point - sc.center(id);
            if constexpr (Scene::kHasNormTable && is_spec<MI>::value)
                normal = m.normalize_on_sphere(dv, sc.norm_m(id), sc.norm_rinv(id), sc.norm_r2f(id));
            else
                normal = normalize_i(m, dv);
        }
    }
    // :82-83  w = Dot(n, d) < 0 ? n : n * -1.0  (multiplying by -1.0 flips the sign bit, exactly)
#ifndef RTM_OPT_NOFLIP
#define RTM_OPT_NOFLIP 1  // (A/B switch)
#endif
    const bool turn = !(dot(normal, dir) < 0.0);
    D3 w = normal;
    // (a sphere hit from outside — every hit in a room seen from inside — never turns its normal: one scalar test for the wave,
    // and the mask and the three sign flips are skipped)
    if (!RTM_OPT_NOFLIP || __builtin_amdgcn_ballot_w64(turn) != 0) {
        const LaneMask flip = lane_mask(turn);
        w = d3(negate_where(flip, normal.x), negate_where(flip, normal.y), negate_where(flip, normal.z));
    }
    out.draws = 3;
    // :88 r1 = 2 PI u with u = m * 2^-24: (2 PI * 2^-24) * m is the same correctly rounded product as 2 PI * (m * 2^-24)
    const double m24 = rng_next_m(rng);  // (r1 is formed where its sin / cos are taken: sincos_draw)
    const double r2 = rng_next(rng);                      // :89
#if RTM_OPT_TRIGLOAD && !RTM_TOL  // (the tolerance unit takes the device's sin / cos: launch_tol)
    TrigFixWord fixw{0u, 0};
    if (sc.v.trig_fix) fixw = trig_fix_load(sc.v.trig_fix, rng);  // wave-uniform; consumed after the sincos
#endif
    const double r2s = m.sqrt64_unit(r2);                 // :90
    // :96-101 — one Normalize on the selected cross product (same values as the two-armed if)
    const bool use_y = fabs(w.x) > (double)FLT_MIN;
    const bool some_x_axis = __builtin_amdgcn_ballot_w64(!use_y) != 0;  // some lane has |w.x| <= FLT_MIN (e.g. literal mode)
    double sn, cs;
#if RTM_OPT_ONB
    if constexpr (std::is_same<MI, MathSpecT<true>>::value) {  // the guarded flavour only: it has excluded zero components
        if (!some_x_axis) {
            // The orthonormal basis with the structural zeros of Cross((0,1,0), w) taken out — exact, not
            // approximate, GIVEN that no component of w is zero, which is what the guard of the normal's
            // Normalize established (a lane where it is not so has m.bad set and the block is re-run):
            //   c = (1*w.z - 0*w.y, -0*w.z + 0*w.x, 0*w.y - 1*w.x) = (w.z, +-0, -w.x)    [x - (+-0) = x for x != 0]
            //   u = Normalize(c) = (c.x / L, +-0, c.z / L)
            //   v = Cross(w, u) = (w.y*u.z - w.z*(+-0), -w.x*u.z + w.z*u.x, w.x*(+-0) - w.y*u.x)
            //                   = (w.y*u.z, -w.x*u.z + w.z*u.x, -(w.y*u.x))             [products are never 0]
            //   nd.y = ((+-0)*cos*r2s + (v.y*sin)*r2s) + w.y*s = (v.y*sin)*r2s + w.y*s   [v.y = L > 0, sin != 0]
            // 12 fp64 instructions fewer than the literal form.
            double ux, uz;
            m.normalize_xz(w.z, -w.x, ux, uz);
            const double vx = w.y * uz, vy = -w.x * uz + w.z * ux, vz = -(w.y * ux);
            m.sincos_draw(m24, sn, cs);
#if RTM_TOL
#elif RTM_OPT_TRIGLOAD
            if (sc.v.trig_fix) trig_fix_apply(fixw, sn, cs);  // wave-uniform
#else
            if (sc.v.trig_fix) apply_trig_fix(sc.v.trig_fix, rng, sn, cs);
#endif
            const double s1 = m.sqrt64_unit(1.0 - r2);
            out.dir = m.normalize_near_unit(d3(((ux * cs) * r2s + (vx * sn) * r2s) + w.x * s1, (vy * sn) * r2s + w.y * s1,
                                               ((uz * cs) * r2s + (vz * sn) * r2s) + w.z * s1));  // :103-107
            out.org = hit_point;
            out.ctr = rng.ctr;
            return;
        }
    }
#endif
    D3 c = cross(d3(0, 1, 0), w);
    D3 u;
    if (some_x_axis) {
        const D3 cx = cross(d3(1, 0, 0), w);
        c = d3(use_y ? c.x : cx.x, use_y ? c.y : cx.y, use_y ? c.z : cx.z);
        u = normalize_i(m, c);
    } else {
        u = m.normalize_y0(c);  // c.y = (-0)*w.z + 0*w.x is a signed zero for finite w
    }
    const D3 v = cross(w, u);  // :102
    m.sincos_draw(m24, sn, cs);
#if RTM_TOL
#elif RTM_OPT_TRIGLOAD
    if (sc.v.trig_fix) trig_fix_apply(fixw, sn, cs);  // wave-uniform
#else
    if (sc.v.trig_fix) apply_trig_fix(sc.v.trig_fix, rng, sn, cs);  // wave-uniform
#endif
    out.dir = m.normalize_near_unit((u * cs) * r2s + (v * sn) * r2s + w * m.sqrt64_unit(1.0 - r2));  // :103-107
    out.org = hit_point;
    out.ctr = rng.ctr;
}

// The classic interface on top of it: returns true when the path continues (org/dir/depth/rng updated, hit id
// pushed through `push`); false when it ended with `term`.  Counters as the reference would count them.
template <class MI, class Scene, typename PushFn>
__device__ __forceinline__ bool path_shade_with(MI& m, const Scene& sc, const int id, const double dis,
                                                const int mode, const int max_bounces, D3& org, D3& dir,
                                                int& depth, RngStream& rng, D3& term, PathCounters& pc,
                                                PushFn push) {
    ShadeOut o;
    const bool cont = path_shade_core(m, sc, id, dis, mode, max_bounces, org, dir, depth, rng, term, o);
    pc.casts++;
    pc.draws += (unsigned)o.draws;
    if (cont) {
        push(depth, id);
        depth++;
        pc.bounces++;
        org = o.org;
        dir = o.dir;
        rng.ctr = o.ctr;
    } else {
        rng.ctr += (unsigned)o.draws * 0x9E3779B9u;
    }
    return cont;
}

// Static-policy form (M = MathRef or MathFast).
template <class M, class Scene, typename PushFn>
__device__ __forceinline__ bool path_shade(const Scene& sc, const int id, const double dis, const int mode,
                                           const int max_bounces, D3& org, D3& dir, int& depth,
                                           RngStream& rng, D3& term, PathCounters& pc, PushFn push) {
    typename std::conditional<std::is_same<M, MathRef>::value, MathRefI, MathFastI>::type m;
    return path_shade_with(m, sc, id, dis, mode, max_bounces, org, dir, depth, rng, term, pc, push);
}

// Speculative form: the whole shading block with MathSpec (one basic block), then ONE check; if any
// lane's operand was outside MathSpec's exact range the block is re-run with the compiler's math
// from the same inputs (path_shade_core leaves them alone).  The hit record is pushed once, after the
// attempt that counts.
template <class Scene, typename PushFn>
__device__ __forceinline__ bool path_shade_spec(const Scene& sc, const int id, const double dis, const int mode,
                                                const int max_bounces, D3& org, D3& dir, int& depth,
                                                RngStream& rng, D3& term, PathCounters& pc, PushFn push,
                                                const ShadeLds lds = ShadeLds()) {
    ShadeOut o;
    typename std::conditional<Scene::kPlanes, MathSpecZ, MathSpec>::type m;  // plane normals have exact zeros: see MathSpecT
    m.set_lds(lds);
    bool cont = path_shade_core(m, sc, id, dis, mode, max_bounces, org, dir, depth, rng, term, o);
    if (__builtin_amdgcn_ballot_w64(m.bad) != 0) {
        MathRefI r;
        r.set_lds(lds);
        cont = path_shade_core(r, sc, id, dis, mode, max_bounces, org, dir, depth, rng, term, o);
    }
    pc.casts++;
    pc.draws += (unsigned)o.draws;
    if (cont) {
        push(depth, id);
        depth++;
        pc.bounces++;
        org = o.org;
        dir = o.dir;
        rng.ctr = o.ctr;
    } else {
        rng.ctr += (unsigned)o.draws * 0x9E3779B9u;
    }
    return cont;
}

#if RTM_TOL
// path_shade_spec for the tolerance translation unit's render loop: a lane whose (primary) ray is flagged — its nearest hit
// must be the one the reference's arithmetic finds, rtm_render_kernel.h: prim_fix — rides on the block's EXISTING fallback:
// the one check behind the speculative block also asks "any flagged lane?", and the fallback first lets `fix` settle those
// lanes' (id, dis) and then shades again with the compiler's math.  Nothing is added between the search and the shading
// block (a branch there, tried first, cost 4 % of the frame: the two blocks no longer overlapped).
template <class Scene, typename PushFn, typename FixFn>
__device__ __forceinline__ bool path_shade_spec_fix(const Scene& sc, int& id, double& dis, const int mode, const int max_bounces,
                                                    D3& org, D3& dir, int& depth, RngStream& rng, D3& term, PathCounters& pc,
                                                    PushFn push, const ShadeLds lds, const bool flagged, FixFn fix) {
    ShadeOut o;
    typename std::conditional<Scene::kPlanes, MathSpecZ, MathSpec>::type m;
    m.set_lds(lds);
    bool cont = path_shade_core(m, sc, id, dis, mode, max_bounces, org, dir, depth, rng, term, o);
    if (__builtin_amdgcn_ballot_w64(m.bad || flagged) != 0) {
        fix(id, dis);
        MathRefI r;
        r.set_lds(lds);
        cont = path_shade_core(r, sc, id, dis, mode, max_bounces, org, dir, depth, rng, term, o);
    }
    pc.casts++;
    pc.draws += (unsigned)o.draws;
    if (cont) {
        push(depth, id);
        depth++;
        pc.bounces++;
        org = o.org;
        dir = o.dir;
        rng.ctr = o.ctr;
    } else {
        rng.ctr += (unsigned)o.draws * 0x9E3779B9u;
    }
    return cont;
}
#endif

// One PathTracing invocation: nearest-hit loop + shading.
template <class M, int UNROLL, class Scene, typename PushFn>
__device__ __forceinline__ bool path_step(const Scene& sc, const int mode, const int max_bounces,
                                          D3& org, D3& dir, int& depth, RngStream& rng, D3& term,
                                          PathCounters& pc, PushFn push, const ShadeLds lds = ShadeLds(),
                                          int* hit_id = nullptr) {
    double dis;
    const int id = nearest_hit<M, UNROLL>(sc, org, dir, dis);
    if (hit_id) *hit_id = id;
    if constexpr (std::is_same<M, MathFast>::value)
        return path_shade_spec(sc, id, dis, mode, max_bounces, org, dir, depth, rng, term, pc, push, lds);
    else
        return path_shade<M>(sc, id, dis, mode, max_bounces, org, dir, depth, rng, term, pc, push);
}

// Fold the recursion back to front: L = colorKD * L_next + emission (src/Renderer.cpp:109).
template <class Scene, typename PopFn>
__device__ __forceinline__ D3 path_fold(const Scene& sc, const D3 term, const int depth, PopFn pop) {
    D3 L = term;
    for (int d = depth - 1; d >= 0; --d) {
        const int id = pop(d);
        L = fold_step(sc.color_kd(id), L, sc.emission(id));
    }
    return L;
}

// The same fold, four levels per trip.  The literal loop is a chain of dependent LDS round trips
// (record -> material row -> 6 flops, per level) and cost 17 % of the frame; here the four records
// and then the four material rows of a trip are fetched together, and only the 4 x (mul, add) steps
// remain serial.  Levels a lane does not have use the identity row the material table carries at
// index n (colorKD = 1, emission = +0): L*1 + 0 is L (only a -0 would become +0, which no later
// operation can tell apart).  `pop_lds(d)` must be valid for 0 <= d < depth.
template <class Scene, typename PopFn>
__device__ __forceinline__ D3 path_fold_blocked(const Scene& sc, const D3 term, const int depth, PopFn pop_lds) {
    D3 L = term;
    const int identity = sc.n();
    int d = depth - 1;
    while (__builtin_amdgcn_ballot_w64(d >= 0) != 0) {
        int id[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int raw = pop_lds(d - k > 0 ? d - k : 0);
            id[k] = (d - k >= 0) ? raw : identity;
        }
        D3 c[4], e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c[k] = sc.color_kd(id[k]);
            e[k] = sc.emission(id[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) L = fold_step(c[k], L, e[k]);
        d -= 4;
    }
    return L;
}

// Fold for paths of at most eight bounces whose records are packed in one 64-bit register, the most
// recent bounce in the low byte (n_spheres <= 256): no LDS traffic for the records at all.  Two
// groups of four levels; the second is skipped when no ending lane is deeper than four.
template <class Scene>
__device__ __forceinline__ D3 path_fold_packed8(const Scene& sc, const D3 term, const int depth,
                                                const unsigned long long rec) {
    // `rec` starts every path filled with the identity row's index, so bytes beyond `depth` already
    // select the identity material: no per-level compare.
    D3 L = term;
    const unsigned lo = (unsigned)rec, hi = (unsigned)(rec >> 32);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (g == 1 && __builtin_amdgcn_ballot_w64(depth > 4) == 0) break;
        const unsigned word = g ? hi : lo;
        D3 c[4], e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = (int)((word >> (8 * k)) & 0xFFu);
            c[k] = sc.color_kd(id);
            e[k] = sc.emission(id);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) L = fold_step(c[k], L, e[k]);
    }
    return L;
}
// One queued path end -> its clamped sample term (src/Renderer.cpp:240): the terminal value is the
// emission row of `term_id` (the identity row, emission +0, for a miss), then the fold.
template <class Scene>
__device__ __forceinline__ D3 path_fold_packed8_all(const Scene& sc, const int term_id, const unsigned long long rec) {
    D3 L = sc.emission(term_id);
    const unsigned lo = (unsigned)rec, hi = (unsigned)(rec >> 32);
#if RTM_OPT_FOLDMUL
    if (sc.v.fold_flags & kFoldNoLevelEmission) {  // wave-uniform: a bounce level adds (+0, +0, +0) — SceneView::fold_flags
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const unsigned word = g ? hi : lo;
            D3 c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = sc.color_kd((int)((word >> (8 * k)) & 0xFFu));
#pragma unroll
            for (int k = 0; k < 4; ++k) L = fold_step_mul(c[k], L);
        }
        return L;
    }
#endif
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const unsigned word = g ? hi : lo;
        D3 c[4], e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = (int)((word >> (8 * k)) & 0xFFu);
            c[k] = sc.color_kd(id);
            e[k] = sc.emission(id);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) L = fold_step(c[k], L, e[k]);
    }
    return L;
}

// The same for paths of any depth (PACKL records): level d sits in byte d & 7 of word d >> 3 (w0: levels
// 0-7, w1: 8-15), levels from 16 up in this path's pooled stack `deep` (deep[0] = level 16).  Bytes a
// path never reached hold the identity row's index, so whole words are folded without compares;
// deepest level first, like the recursion unwinds.
template <class Scene>
__device__ __forceinline__ D3 path_fold_packed16(const Scene& sc, const int term_id, const int depth,
                                                 const unsigned long long w0, const unsigned long long w1,
                                                 const unsigned char* deep) {
    D3 L = sc.emission(term_id);
    if (__builtin_amdgcn_ballot_w64(depth > 16) != 0) {  // rare: P(depth > 16) ~ kd^16
        for (int d = depth - 1; d >= 16; --d) {
            const int id = (int)deep[d - 16];
            L = fold_step(sc.color_kd(id), L, sc.emission(id));
        }
    }
    auto word = [&](const unsigned long long w) {
        const unsigned half[2] = {(unsigned)(w >> 32), (unsigned)w};  // levels 7..4, then 3..0
#if RTM_OPT_FOLDMUL
        if (sc.v.fold_flags & kFoldNoLevelEmission) {  // wave-uniform: a bounce level adds (+0, +0, +0)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                D3 c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = sc.color_kd((int)((half[g] >> (8 * (3 - k))) & 0xFFu));
#pragma unroll
                for (int k = 0; k < 4; ++k) L = fold_step_mul(c[k], L);
            }
            return;
        }
#endif
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            D3 c[4], e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int id = (int)((half[g] >> (8 * (3 - k))) & 0xFFu);
                c[k] = sc.color_kd(id);
                e[k] = sc.emission(id);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) L = fold_step(c[k], L, e[k]);
        }
    };
    if (__builtin_amdgcn_ballot_w64(depth > 8) != 0) word(w1);
    word(w0);
    return L;
}

// the 64-bit record register of a path that has not bounced yet: identity index in every byte
__device__ __forceinline__ unsigned long long packed8_empty(int identity) {
    return 0x0101010101010101ull * (unsigned long long)(unsigned)identity;
}

// src/Renderer.cpp:43-49: std::min<double>(std::max<double>(v, 0), 1.0f)
__device__ __forceinline__ double clamp01(double v) {
    const double lo = (v < 0.0) ? 0.0 : v;
    return (1.0 < lo) ? 1.0 : lo;
}
// clampColor on a vec3.  The clamp is the identity unless a channel is < 0 or > 1 (NaN passes
// through both comparisons unchanged), so the selects run only when some lane needs them.
__device__ __forceinline__ D3 clamp01_d3(D3 c) {
    // v in [+0, 1] <=> its bit pattern, as an unsigned integer, is <= that of 1.0 (negative values,
    // -0 and NaN have larger patterns and take the exact path below)
    const unsigned long long one = 0x3FF0000000000000ull;
    const bool out = ((unsigned long long)__double_as_longlong(c.x) > one) ||
                     ((unsigned long long)__double_as_longlong(c.y) > one) ||
                     ((unsigned long long)__double_as_longlong(c.z) > one);
    if (__builtin_amdgcn_ballot_w64(out) == 0) return c;
    return D3{clamp01(c.x), clamp01(c.y), clamp01(c.z)};
}

}  // namespace RTM_NS
