// rtm_surface.h — png::SurfaeSample (src/Renderer.cpp:119-198) with SphereObject::ComputeSurfacePoint
// (src/SettingData.cpp:227-233) / PlaneObject::ComputeSurfacePoint (:247-249): the reference's second, experimental
// integrator.  Renderer::Render would select it at :234-236 if a U[0,1) draw were >= 1.0 — never — so it is dead code in
// the reference; it is DEFINED code all the same (external linkage, the same injectable generator as PathTracing), and
// SURVEY.md §8(f) row 4 lists it.  Restated in the test oracle first, then here:
//   * a per-ray seam (surface_sample_rays_kernel, rtm_surface_sample_batch) mirroring rtm_path_trace_batch, and
//   * an integrator switch (RTM_MODE_SURFACE_SAMPLE) served by render_surface_kernel — a general per-object kernel in the
//     class of variant 1 (the reference's loop as written, the compiler's correctly rounded math), not by the tuned ones.
// The recursion is flattened: level 0 records its hit object, every deeper level that recurses records (object, div), and the
// value is folded back to front in the recursion's own order:  level k > 0: L = (L * colorKD_k) * div_k + emission_k  (:192),
// level 0: L = L * color_0 + emission_0  (:147).  div = std::min(Magnitude(..), 1.0) is a float-valued double, kept as a float.
// Build extension, as for PathTracing: max_bounces >= 0 makes an invocation at depth > max_bounces return 0 without drawing.
#pragma once

namespace rtm {

struct SurfaceConsts {
    double lx, ly, lz;  // ComputeSurfacePoint's localPoint: sin(2pi) sin(pi/2), sin(2pi) cos(pi/2), cos(2pi) as the HOST libm evaluates them
};

struct SurfOut {
    D3 L;
    unsigned casts, bounces, draws;
    bool overflow;
};

// One nearest-hit loop (src/Renderer.cpp:126-137 / :163-174: the loop of PathTracing, written out again) with the normal
// Intersect delivers: Normalize(hitPoint - position) for a sphere (src/SettingData.cpp:214-215), m_normal for a plane,
// nothing in literal mode (D2: the caller's normal stays (0,0,0)).
__device__ __forceinline__ int surf_nearest(const SceneGlobal& sc, const int mode, const D3 org, const D3 dir, double& dis, D3& normal) {
    const int id = nearest_hit<MathRef, 1>(sc, org, dir, dis);
    normal = d3(0, 0, 0);
    if (id >= 0 && mode != RTM_MODE_LITERAL) {
        const double4 g = sc.v.geom[id];
        if (sc.v.plane != nullptr && g.w < 0.0) {
            const double* pl = sc.v.plane + (size_t)id * 16;
            normal = d3(pl[3], pl[4], pl[5]);
        } else {
            normal = normalize((org + dir * dis) - d3(g.x, g.y, g.z));
        }
    }
    return id;
}

// push(level >= 1, id, div), pop(level) -> (id, div); capacity: levels the records hold
template <typename PushFn, typename PopFn>
__device__ __forceinline__ SurfOut surface_sample_core(const SceneGlobal& sc, const SurfaceConsts& K, const int mode,
                                                       const int max_bounces, D3 org, D3 dir, RngStream& rng, const int capacity,
                                                       PushFn push, PopFn pop) {
    SurfOut o{d3(0, 0, 0), 0u, 0u, 0u, false};
    const double* surf = sc.v.surf;  // per object: raw color (3), (double)radius
    double dis;
    D3 normal;
    o.casts++;
    const int id0 = surf_nearest(sc, mode, org, dir, dis, normal);  // depth 0, :122-137
    if (id0 < 0) return o;                                         // :138-140
    org = org + dir * dis;                                         // :133 hitpoint; the next ray is (hitpoint, normal), :146
    dir = normal;
    int depth = 1;
    D3 term = d3(0, 0, 0);
    const int n = sc.n();
    for (;;) {
        if (max_bounces >= 0 && depth > max_bounces) break;  // build extension: returns 0 without drawing
        if (depth > capacity) {
            o.overflow = true;
            break;
        }
        o.draws++;
        const int oi = (int)(rng_next(rng) * (double)n);  // :150
        const double4 g = sc.v.geom[oi];
        const bool plane = sc.v.plane != nullptr && g.w < 0.0;
        const double radius = surf[(size_t)oi * 4 + 3];
        const D3 sp = plane ? d3(0, 0, 0) : (d3(K.lx, K.ly, K.lz) * radius + d3(g.x, g.y, g.z));  // :152 ComputeSurfacePoint
        const D3 cdir = normalize(sp - org);                                                      // :160 (and :179: the same value)
        o.casts++;
        const int hit = surf_nearest(sc, mode, org, cdir, dis, normal);                           // :163-174
        if (hit < 0 || hit != oi) break;                                                          // :175-177 -> 0
        const D3 hitpoint = org + cdir * dis;                                                     // :171
        const double dot1 = dot(dir, cdir);                                                       // :180
        const double dot2 = dot(d3(-cdir.x, -cdir.y, -cdir.z), normal);                           // :181
        if (dot1 <= 0 || dot2 <= 0) break;                                                        // :182-184 -> 0
        const double distance = magnitude(org - hitpoint);                                        // :185
        const double probability = (distance < 1.0) ? 1.0 : distance;                             // :186 std::max
        const double div = (1.0 < distance) ? 1.0 : distance;                                     // :187 std::min
        const D3 emission = sc.emission(oi);
        o.draws++;
        if (dot1 * dot2 * sc.kd(oi) * probability < rng_next(rng)) {                              // :188-190
            term = emission;
            break;
        }
        o.bounces++;
        push(depth, oi, (float)div);  // (div: a float-valued double in (0, 1], or NaN)
        org = hitpoint;               // :191 Ray(hitpoint, normal)
        dir = normal;
        ++depth;
    }
    // the recursion unwinds: levels depth-1 .. 1, then level 0
    D3 L = term;
    for (int d = depth - 1; d >= 1; --d) {
        int id;
        float div;
        pop(d, id, div);
        L = (L * sc.color_kd(id)) * (double)div + sc.emission(id);  // :192
    }
    L = L * d3(surf[(size_t)id0 * 4], surf[(size_t)id0 * 4 + 1], surf[(size_t)id0 * 4 + 2]) + sc.emission(id0);  // :147
    o.L = L;
    return o;
}

// ---- per-ray seam -------------------------------------------------------------------------------------------------
struct SurfBatchParams {
    SceneView scene;
    SurfaceConsts K;
    int mode, max_bounces;
    uint64_t seed_mult;
    const double* __restrict__ org;
    const double* __restrict__ dir;
    size_t n_rays;
    double* __restrict__ out;
    uint32_t* __restrict__ out_draws;
    uint32_t* __restrict__ out_casts;
    uint2* __restrict__ scratch;  // [level][ray] (object, div as float bits), SURF_MAX_DEPTH levels
    unsigned long long* __restrict__ counters;
};
constexpr int SURF_MAX_DEPTH = 1024;

__global__ __launch_bounds__(64) void surface_sample_rays_kernel(const SurfBatchParams P) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= P.n_rays) return;
    SceneGlobal sc;
    sc.v = P.scene;
    RngStream rng = rng_open(rng_pixel_key(P.seed_mult, (uint32_t)i), 0u);
    const SurfOut o = surface_sample_core(
        sc, P.K, P.mode, P.max_bounces, d3(P.org[i * 3], P.org[i * 3 + 1], P.org[i * 3 + 2]),
        d3(P.dir[i * 3], P.dir[i * 3 + 1], P.dir[i * 3 + 2]), rng, SURF_MAX_DEPTH,
        [&](int d, int id, float div) { P.scratch[(size_t)(d - 1) * P.n_rays + i] = uint2{(unsigned)id, __float_as_uint(div)}; },
        [&](int d, int& id, float& div) {
            const uint2 r = P.scratch[(size_t)(d - 1) * P.n_rays + i];
            id = (int)r.x;
            div = __uint_as_float(r.y);
        });
    P.out[i * 3] = o.L.x;
    P.out[i * 3 + 1] = o.L.y;
    P.out[i * 3 + 2] = o.L.z;
    if (P.out_draws) P.out_draws[i] = o.draws;
    if (P.out_casts) P.out_casts[i] = o.casts;
    if (o.overflow) atomicOr(P.counters + 3, 1ull);
}

// ---- the integrator switch: Renderer::Render's loop nest (src/Renderer.cpp:215-250) with SurfaeSample underneath ----------
// One wave per 8x8 tile, lane = pixel, samples in the reference's order; records: 16 levels in LDS ([level][lane], 8 bytes),
// deeper ones in a slot of the pooled stack (kPoolLevels levels of 8 bytes, taken once per lane); a path deeper than both
// raises the overflow flag and the call fails loudly.
constexpr int kSurfLdsLevels = 16;
__global__ __launch_bounds__(64) void render_surface_kernel(const RenderParams P, const SurfaceConsts K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint2* rec = reinterpret_cast<uint2*>(lds_raw);  // [kSurfLdsLevels][64]
    const int lane = threadIdx.x;
    const unsigned tile = blockIdx.x;
    const int x = (int)(tile % (unsigned)P.tiles_x) * 8 + (lane & 7);
    const int y = band_row(P, (int)(tile / (unsigned)P.tiles_x), lane >> 3);
    const bool valid = x < P.W && y < P.row_end;
    SceneGlobal sc;
    sc.v = P.scene;
    int slot = -1;
    bool overflow = false;
    auto slot_ptr = [&]() -> uint2* {
        if (slot < 0) {
            const unsigned got = P.pool ? atomicAdd(P.pool_next, 1u) : 0xFFFFFFFFu;
            if (got >= P.pool_slots) return nullptr;
            slot = (int)got;
        }
        return reinterpret_cast<uint2*>(P.pool) + (size_t)slot * kPoolLevels;
    };
    auto push = [&](int d, int id, float div) {
        const uint2 r{(unsigned)id, __float_as_uint(div)};
        if (d <= kSurfLdsLevels) {
            rec[(d - 1) * 64 + lane] = r;
        } else if (uint2* p = slot_ptr()) {
            p[d - 1 - kSurfLdsLevels] = r;
        } else {
            overflow = true;
        }
    };
    auto pop = [&](int d, int& id, float& div) {
        uint2 r{0u, 0u};
        if (d <= kSurfLdsLevels) r = rec[(d - 1) * 64 + lane];
        else if (slot >= 0) r = (reinterpret_cast<const uint2*>(P.pool) + (size_t)slot * kPoolLevels)[d - 1 - kSurfLdsLevels];
        id = (int)r.x;
        div = __uint_as_float(r.y);
    };
    const int capacity = P.pool ? kSurfLdsLevels + kPoolLevels : kSurfLdsLevels;
    unsigned casts = 0, bounces = 0, draws = 0;
    D3 acc = d3(0, 0, 0);
    if (valid) {
        const RngPixelKey pkey = rng_pixel_key(P.seed_mult, (uint32_t)y * (uint32_t)P.W + (uint32_t)x);
        for (int sx = 1; sx <= P.SS; ++sx)
            for (int sy = 1; sy <= P.SS; ++sy) {
                const D3 pdir = primary_dir(P, x, y, sx, sy);
                for (int s = 0; s < P.S; ++s) {
                    RngStream rng = rng_open(pkey, (uint32_t)(((sx - 1) * P.SS + (sy - 1)) * P.S + s));
                    const SurfOut o = surface_sample_core(sc, K, P.mode, P.max_bounces, P.cam_org, pdir, rng, capacity, push, pop);
                    overflow = overflow || o.overflow;
                    casts += o.casts;
                    bounces += o.bounces;
                    draws += o.draws;
                    const D3 cal = ((o.L / P.dSS) / P.dSS) / P.dS;                      // :240
                    acc = acc + d3(clamp01(cal.x), clamp01(cal.y), clamp01(cal.z));  // :241-242
                }
            }
    }
    store_pixel(P, valid, x, y, acc);
    if (P.counters) {
        wave_add_counter(P.counters + 0, casts);
        wave_add_counter(P.counters + 1, bounces);
        wave_add_counter(P.counters + 2, draws);
        if (overflow) atomicOr(P.counters + 3, 1ull);
    }
}

}  // namespace rtm
