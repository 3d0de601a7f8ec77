// rtm_path.h — the device-side path: SphereObject::Intersect, the nearest-hit loop, one
// PathTracing invocation and the back-to-front fold, written once and parameterised by
//   * a Math policy  (how sqrt / x/y are evaluated — every policy returns IEEE-correct bits), and
//   * a Scene policy (where sphere geometry and materials are read from).
// Citations are file:line in the reference checkout.
#pragma once
#include <cfloat>

#include "../../include/rtm.h"
#include "rtm_device.h"

namespace rtm {

// ------------------------------------------------------------------------------------------------
// Math policies
// ------------------------------------------------------------------------------------------------

// MathRef: the compiler's IEEE expansions (v_div_scale/v_div_fmas/v_div_fixup, scaled rsq+NR sqrt).
struct MathRef {
    static __device__ __forceinline__ double sqrt64(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ D3 div3(D3 a, double y) { return a / y; }
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
        const double y = __builtin_amdgcn_rsq(x);
        const double s0 = x * y, h0 = 0.5 * y;
        const double r0 = __builtin_fma(-h0, s0, 0.5);
        const double s1 = __builtin_fma(s0, r0, s0), h1 = __builtin_fma(h0, r0, h0);
        const double d0 = __builtin_fma(-s1, s1, x);
        const double s2 = __builtin_fma(d0, h1, s1);
        const double d1 = __builtin_fma(-s2, s2, x);
        return __builtin_fma(d1, h1, s2);
    }
    static __device__ __forceinline__ D3 div3(D3 a, double y) {
        const bool ok = moderate(y) && moderate(a.x) && moderate(a.y) && moderate(a.z);
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) return a / y;
        double r = __builtin_amdgcn_rcp(y);
        double e = __builtin_fma(-y, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-y, r, 1.0);
        r = __builtin_fma(r, e, r);
        auto one = [&](double x) {
            const double q = x * r;
            const double rem = __builtin_fma(-y, q, x);
            return __builtin_amdgcn_div_fixup(__builtin_fma(rem, r, q), y, x);
        };
        return D3{one(a.x), one(a.y), one(a.z)};
    }
};

// src/Ray.h:67-72 through a policy
template <class M>
__device__ __forceinline__ D3 normalize_m(D3 a) {
    return M::div3(a, magnitude(a));
}

// ------------------------------------------------------------------------------------------------
// Scene policies.  geom[i] = (cx, cy, cz, (double)(float)(r*r)); mat[i*8..] = colorKD.xyz, kd,
// emission.xyz, pad.
// ------------------------------------------------------------------------------------------------
struct SceneView {
    const double4* __restrict__ geom;
    const double* __restrict__ mat;
    int n;
};

// Everything from global memory (uniform geometry loads become scalar loads; per-lane material and
// centre look-ups are vector gathers served by L1/L2).  Works for any n.
struct SceneGlobal {
    SceneView v;
    __device__ __forceinline__ int n() const { return v.n; }
    __device__ __forceinline__ double4 geom_uniform(int i) const { return v.geom[i]; }
    __device__ __forceinline__ D3 center(int id) const {
        const double4 g = v.geom[id];
        return D3{g.x, g.y, g.z};
    }
    __device__ __forceinline__ double kd(int id) const { return v.mat[(size_t)id * 8 + 3]; }
    __device__ __forceinline__ D3 emission(int id) const {
        const double* m = v.mat + (size_t)id * 8;
        return D3{m[4], m[5], m[6]};
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
    __device__ __forceinline__ int n() const { return v.n; }
    __device__ __forceinline__ double4 geom_uniform(int i) const { return v.geom[i]; }
    __device__ __forceinline__ D3 center(int id) const {
        const double* g = lgeom + id * 4;
        return D3{g[0], g[1], g[2]};
    }
    __device__ __forceinline__ double kd(int id) const { return lmat[id * 8 + 3]; }
    __device__ __forceinline__ D3 emission(int id) const {
        const double* m = lmat + id * 8;
        return D3{m[4], m[5], m[6]};
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

// src/Renderer.cpp:58-73: brute-force nearest hit; strict < keeps the lowest index on ties.
// UNROLL > 1 fetches the geometry of UNROLL spheres (wave-uniform loads) before testing them.
template <class M, int UNROLL, class Scene>
__device__ __forceinline__ int nearest_hit(const Scene& sc, const D3 org, const D3 dir, double& dis) {
    int hit_object = -1;
    dis = DBL_MAX;
    const int n = sc.n();
    if constexpr (UNROLL <= 1) {
        for (int i = 0; i < n; ++i) {
            const double4 g = sc.geom_uniform(i);
            double t;
            if (sphere_test<M>(g, org, dir, t) && t < dis && t > 0) {
                dis = t;
                hit_object = i;
            }
        }
    } else {
        for (int i0 = 0; i0 < n; i0 += UNROLL) {
            double4 g[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) g[k] = sc.geom_uniform(i0 + k < n ? i0 + k : n - 1);
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
                if (i0 + k < n) sphere_update<M>(g[k], org, dir, i0 + k, dis, hit_object);  // uniform
        }
    }
    return hit_object;
}

struct PathCounters {
    unsigned casts, bounces, draws;
};

// The part of one PathTracing invocation (src/Renderer.cpp:57-117) after the nearest-hit loop:
// `id`/`dis` are that loop's result.  Returns true when the path continues (org/dir/depth updated,
// hit id pushed through `push`); false when it ended with `term` = the value the deepest invocation
// returned.
template <class M, class Scene, typename PushFn>
__device__ __forceinline__ bool path_shade(const Scene& sc, const int id, const double dis, const int mode,
                                           const int max_bounces, D3& org, D3& dir, int& depth,
                                           RngStream& rng, D3& term, PathCounters& pc, PushFn push) {
    pc.casts++;
    if (id < 0) {  // :116
        term = d3(0, 0, 0);
        return false;
    }
    const D3 emission = sc.emission(id);
    if (max_bounces >= 0 && depth >= max_bounces) {  // build extension: no draw
        term = emission;
        return false;
    }
    pc.draws++;
    if (!(rng_next(rng) <= sc.kd(id))) {  // :78, kd() is a float widened to double
        term = emission;                  // :112
        return false;
    }
    const D3 hit_point = dir * dis + org;  // :79
    // D2: in literal mode the caller's normal stays (0,0,0); repaired: src/SettingData.cpp:214-215
    const D3 normal = (mode == RTM_MODE_LITERAL) ? d3(0, 0, 0)
                                                 : normalize_m<M>(hit_point - sc.center(id));
    const D3 w = dot(normal, dir) < 0.0 ? normal : normal * -1.0;  // :82-83
    pc.draws += 2;
    const double r1 = 6.283185307179586 * rng_next(rng);  // :88  (2*PI folded)
    const double r2 = rng_next(rng);                      // :89
    const double r2s = M::sqrt64(r2);                     // :90
    // :96-101 — one Normalize on the selected cross product (same values as the two-armed if)
    const bool use_y = fabs(w.x) > (double)FLT_MIN;
    const D3 cy = cross(d3(0, 1, 0), w), cx = cross(d3(1, 0, 0), w);
    const D3 u = normalize_m<M>(use_y ? cy : cx);
    const D3 v = cross(w, u);  // :102
    double sn, cs;
    sincos(r1, &sn, &cs);
    const D3 nd =
        normalize_m<M>((u * cs) * r2s + (v * sn) * r2s + w * M::sqrt64(1.0 - r2));  // :103-107
    push(depth, id);
    depth++;
    pc.bounces++;
    org = hit_point;
    dir = nd;
    return true;
}

// One PathTracing invocation: nearest-hit loop + shading.
template <class M, int UNROLL, class Scene, typename PushFn>
__device__ __forceinline__ bool path_step(const Scene& sc, const int mode, const int max_bounces,
                                          D3& org, D3& dir, int& depth, RngStream& rng, D3& term,
                                          PathCounters& pc, PushFn push) {
    double dis;
    const int id = nearest_hit<M, UNROLL>(sc, org, dir, dis);
    return path_shade<M>(sc, id, dis, mode, max_bounces, org, dir, depth, rng, term, pc, push);
}

// Fold the recursion back to front: L = colorKD * L_next + emission (src/Renderer.cpp:109).
template <class Scene, typename PopFn>
__device__ __forceinline__ D3 path_fold(const Scene& sc, const D3 term, const int depth, PopFn pop) {
    D3 L = term;
    for (int d = depth - 1; d >= 0; --d) {
        const int id = pop(d);
        L = sc.color_kd(id) * L + sc.emission(id);
    }
    return L;
}

// src/Renderer.cpp:43-49: std::min<double>(std::max<double>(v, 0), 1.0f)
__device__ __forceinline__ double clamp01(double v) {
    const double lo = (v < 0.0) ? 0.0 : v;
    return (1.0 < lo) ? 1.0 : lo;
}

}  // namespace rtm
