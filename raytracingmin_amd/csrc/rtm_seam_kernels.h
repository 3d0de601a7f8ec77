// rtm_seam_kernels.h — parity instruments and diagnostics, never timed as the product: the per-ray seam
// (png::PathTracing, src/Renderer.cpp:57-117), the per-call seam (Object::Intersect, src/SettingData.cpp:197-226, :243-246),
// the RNG batch, the reference-loop probe, the isolated-loop benches, the fp64 peak kernel, the exhaustive self-checks and
// the math probe.  Included by rtm_kernels.hip after rtm_render_kernel.h.
#pragma once

namespace rtm {

// ------------------------------------------------------------------------------------------------
// Per-ray seam: png::PathTracing for a batch of rays (one lane per ray).
struct RayBatchParams {
    SceneView scene;
    int mode, max_bounces;
    uint64_t seed_mult;
    const double* __restrict__ org;
    const double* __restrict__ dir;
    size_t n_rays;
    double* __restrict__ out;
    uint32_t* __restrict__ out_draws;
    uint32_t* __restrict__ out_casts;
    uint32_t* __restrict__ scratch;  // [depth][ray] hit records, RAY_MAX_DEPTH deep
    unsigned long long* __restrict__ counters;
    // debugging aid (RTM_DEBUG_SEAM_KEY="pixel,sample"): every ray draws from that stream instead of (i, 0),
    // so one sample of a rendered frame can be replayed through the per-ray seam
    int key_override;
    uint32_t key_pixel, key_sample;
    double* __restrict__ trace;  // debugging aid: (org, dir) of ray 0 at every cast, 6 doubles each
    int trace_cap;
};
constexpr int RAY_MAX_DEPTH = 4096;

__global__ __launch_bounds__(64) void path_trace_rays_kernel(const RayBatchParams P) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= P.n_rays) return;
    D3 org = d3(P.org[i * 3], P.org[i * 3 + 1], P.org[i * 3 + 2]);
    D3 dir = d3(P.dir[i * 3], P.dir[i * 3 + 1], P.dir[i * 3 + 2]);
    RngStream rng = P.key_override ? rng_open(rng_pixel_key(P.seed_mult, P.key_pixel), P.key_sample)
                                   : rng_open(rng_pixel_key(P.seed_mult, (uint32_t)i), 0u);
    PathCounters pc = {0, 0, 0};
    int depth = 0;
    bool overflow = false;
    auto push = [&](int d, int id) {
        if (d < RAY_MAX_DEPTH)
            P.scratch[(size_t)d * P.n_rays + i] = (uint32_t)id;
        else
            overflow = true;
    };
    auto pop = [&](int d) -> int { return (int)P.scratch[(size_t)d * P.n_rays + i]; };
    D3 term;
    SceneGlobal sc;
    sc.v = P.scene;
    auto log_ray = [&]() {
        if (P.trace && i == 0 && depth < P.trace_cap) {
            double* t = P.trace + (size_t)depth * 6;
            t[0] = org.x; t[1] = org.y; t[2] = org.z; t[3] = dir.x; t[4] = dir.y; t[5] = dir.z;
        }
    };
    log_ray();
    while (path_step<MathRef, 1>(sc, P.mode, P.max_bounces, org, dir, depth, rng, term, pc, push)) {
        log_ray();
        if (depth >= RAY_MAX_DEPTH) {
            overflow = true;
            term = d3(0, 0, 0);
            break;
        }
    }
    const D3 L = path_fold(sc, term, depth, pop);
    P.out[i * 3] = L.x;
    P.out[i * 3 + 1] = L.y;
    P.out[i * 3 + 2] = L.z;
    if (P.out_draws) P.out_draws[i] = pc.draws;
    if (P.out_casts) P.out_casts[i] = pc.casts;
    if (overflow) atomicOr(P.counters + 3, 1ull);
}

// Per-call seam: SphereObject::Intersect, pair i = (ray i, sphere i).
__global__ __launch_bounds__(64) void intersect_pairs_kernel(const double4* __restrict__ geom,
                                                             const double* __restrict__ org,
                                                             const double* __restrict__ dir,
                                                             size_t n, int mode,
                                                             int32_t* __restrict__ out_hit,
                                                             double* __restrict__ out_t,
                                                             double* __restrict__ out_normal,
                                                             const double* __restrict__ plane = nullptr) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const D3 o = d3(org[i * 3], org[i * 3 + 1], org[i * 3 + 2]);
    const D3 d = d3(dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2]);
    const double4 g = geom[i];
    double t;
    const bool is_plane = plane != nullptr && g.w < 0.0;
    const bool hit = is_plane ? plane_test(plane + i * 16, o, d, t) : sphere_test<MathRef>(g, o, d, t);
    out_hit[i] = hit ? 1 : 0;
    if (hit) {
        out_t[i] = t;
        if (mode != RTM_MODE_LITERAL) {  // D2: literal mode never delivers the normal
            const D3 nrm = is_plane ? d3(plane[i * 16 + 3], plane[i * 16 + 4], plane[i * 16 + 5])
                                    : normalize((o + d * t) - d3(g.x, g.y, g.z));  // src/SettingData.cpp:214-215
            out_normal[i * 3] = nrm.x;
            out_normal[i * 3 + 1] = nrm.y;
            out_normal[i * 3 + 2] = nrm.z;
        }
    }
}

__global__ void rng_batch_kernel(uint64_t seed_mult, uint32_t pixel0, uint32_t n_pixels,
                                 uint32_t sample, uint32_t n_draws, double* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    RngStream s = rng_open(rng_pixel_key(seed_mult, pixel0 + p), sample);
    for (uint32_t k = 0; k < n_draws; ++k) out[(size_t)p * n_draws + k] = rng_next(s);
}

// ------------------------------------------------------------------------------------------------
// Component micro-benchmarks (profiles/component_bench.py): the nearest-hit loop and the shading
// block in isolation, timed with s_memtime, same policies as the render kernel.
// src/Renderer.cpp:58-73 as written, one lane per ray (SoA rays): the yardstick of rtm_debug_wf_nearest
__global__ __launch_bounds__(256) void nearest_probe_kernel(SceneView scene, const double* __restrict__ org,
                                                            const double* __restrict__ dir, unsigned n_rays,
                                                            int* __restrict__ out_id, double* __restrict__ out_t) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const unsigned r = i < n_rays ? i : n_rays - 1u;
    SceneGlobal sc;
    sc.v = scene;
    double dis;
    const int id = nearest_hit<MathRef, 1>(sc, d3(org[r], org[n_rays + r], org[2 * (size_t)n_rays + r]),
                                           d3(dir[r], dir[n_rays + r], dir[2 * (size_t)n_rays + r]), dis);
    if (i < n_rays) {
        out_id[i] = id;
        out_t[i] = dis;
    }
}

// nearest_hit_grid on caller-supplied rays, with its per-ray work counts (rtm_debug_grid_nearest)
__global__ __launch_bounds__(256) void grid_probe_kernel(SceneView scene, const double* __restrict__ org,
                                                         const double* __restrict__ dir, unsigned n_rays,
                                                         int* __restrict__ out_id, double* __restrict__ out_t,
                                                         unsigned* __restrict__ out_tests, unsigned* __restrict__ out_steps) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const unsigned r = i < n_rays ? i : n_rays - 1u;
    SceneGlobal sc;
    sc.v = scene;
    double dis;
    unsigned tests, steps;
    __shared__ __attribute__((aligned(16))) unsigned char queue[GridWalk<MathFast, SceneGlobal, true>::queue_bytes(256)];
    const int id = nearest_hit_grid<MathFast, SceneGlobal, true>(sc, d3(org[r], org[n_rays + r], org[2 * (size_t)n_rays + r]),
                                                                 d3(dir[r], dir[n_rays + r], dir[2 * (size_t)n_rays + r]), dis,
                                                                 queue, 256, (int)threadIdx.x, &tests, &steps);
    if (i < n_rays) {
        out_id[i] = id;
        out_t[i] = dis;
        if (out_tests) out_tests[i] = tests;
        if (out_steps) out_steps[i] = steps;
    }
}

template <class M, int UNROLL>
__global__ __launch_bounds__(64) void nearest_bench_kernel(SceneView scene, D3 org0, int reps, double* out,
                                                           unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    SceneGlobal sc;
    sc.v = scene;
    const int lane = threadIdx.x;
    D3 dir = normalize(d3(-0.8 + 0.025 * (lane & 7) + 1e-3 * blockIdx.x, -0.5 + 0.12 * (lane >> 3), 1.0));
    D3 org = org0;
    double acc = 0.0;
    int ids = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r) {
        double dis;
        const int id = nearest_hit<M, UNROLL>(sc, org, dir, dis);
        acc += dis;
        ids += id;
        org.x += 1e-7 * (id + 2);  // the next cast depends on this one, like a path
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[(size_t)blockIdx.x * 64 + lane] = acc + ids;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    if (lds_raw[0] == 77 && reps < 0) out[0] = 1;  // keeps the dynamic LDS allocation alive
}

template <class M>
__global__ __launch_bounds__(64) void shade_bench_kernel(SceneView scene, D3 org0, int reps, double* out,
                                                         unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    SceneGlobal sc;
    sc.v = scene;
    const int lane = threadIdx.x;
    D3 dir = normalize(d3(-0.8 + 0.025 * (lane & 7) + 1e-3 * blockIdx.x, -0.5 + 0.12 * (lane >> 3), 1.0));
    D3 org = org0;
    RngStream rng = rng_open(rng_pixel_key(12345, blockIdx.x * 64 + lane), 0);
    PathCounters pc = {0, 0, 0};
    double acc = 0.0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r) {
        int depth = 0;
        D3 term;
        // hit sphere 1 (a wall, kd 0.9) at a fixed distance: ~90 % of the lanes continue
        bool cont;
        if constexpr (std::is_same<M, MathFast>::value)
            cont = path_shade_spec(sc, 1, 9.5 + 1e-3 * lane, RTM_MODE_REPAIRED, -1, org, dir, depth, rng, term, pc,
                                   [](int, int) {});
        else
            cont = path_shade<M>(sc, 1, 9.5 + 1e-3 * lane, RTM_MODE_REPAIRED, -1, org, dir, depth, rng, term, pc,
                                 [](int, int) {});
        if (!cont) {
            acc += term.x;
            org = org0;
        }
        org.x *= 0.5;
        org.y *= 0.5;
        org.z = org.z * 0.5 - 5.0;
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[(size_t)blockIdx.x * 64 + lane] = acc + dir.x + org.y + pc.draws;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    if (lds_raw[0] == 77 && reps < 0) out[0] = 1;
}

// Exhaustive device-side self-checks (return the number of mismatching inputs).
//   kind 0: sqrtf_fast == sqrtf for every float in [2^-96, FLT_MAX]
// fp64 vector peak by wall clock (rtm_debug_fp64_peak; the stand-alone profiles/ubench/fp64_peak.hip has the whole
// price list): every wave runs `iters` x 64 v_fma_f64 over 8 independent accumulators, `waves_per_simd` waves per SIMD
// on every CU, no memory traffic in the loop.
__global__ __launch_bounds__(256) void fp64_peak_kernel(double* out, int iters) {
    double a0 = threadIdx.x * 1.0000001 + 1.5, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
           a7 = a0 + 7;
    const double b = 1.0000001, c = 0.5;
    for (int i = 0; i < iters; ++i) {
#define RTM_FMA8                                                                                                      \
    asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"  \
                 "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"    \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                     \
                 : "v"(b), "v"(c));
        RTM_FMA8 RTM_FMA8 RTM_FMA8 RTM_FMA8 RTM_FMA8 RTM_FMA8 RTM_FMA8 RTM_FMA8
#undef RTM_FMA8
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void selfcheck_kernel(int kind, unsigned long long* mismatches) {
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    if (kind == 0) {
        for (unsigned long long b = 0x0F800000ull + blockIdx.x * blockDim.x + threadIdx.x; b < 0x7F800000ull; b += stride) {
            const float x = __uint_as_float((unsigned)b);
            if (!sqrtf_fast_ok(x) || __float_as_uint(sqrtf_fast(x)) != __float_as_uint(__builtin_sqrtf(x))) ++bad;
        }
        // outside the range the guard must say so
        const float outside[6] = {0.0f, -1.0f, 1e-30f, __uint_as_float(0x7F800000u), __uint_as_float(0x7FC00000u), 1e-38f};
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int i = 0; i < 6; ++i)
                if (sqrtf_fast_ok(outside[i])) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Device primitives exposed for parity tests of the building blocks (tests/test_device_math.py).
__global__ void math_probe_kernel(int op, const double* __restrict__ a, const double* __restrict__ b,
                                  size_t n, double* __restrict__ out, const uint32_t* __restrict__ fix = nullptr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b ? b[i] : 0.0;
    double r = 0.0, s, c;
    if (op >= 16 && op <= 18) {
        // the draws of one bounce (RR, r1, r2) from the stream (ctr, k1) = (a, b), then the shading
        // block's sincos of r1 corrected to the host libm: 16 -> r1, 17 -> sin, 18 -> cos
        RngStream st{(uint32_t)x, (uint32_t)y};
        (void)rng_next(st);
        const double r1 = 6.283185307179586 * rng_next(st);
        (void)rng_next(st);
        sincos_small(r1, s, c);
        apply_trig_fix(fix, st, s, c);
        out[i] = op == 16 ? r1 : (op == 17 ? s : c);
        return;
    }
    switch (op) {
        case 0: r = sqrt(x); break;
        case 1: r = (double)__builtin_sqrtf((float)x); break;
        case 2: r = x / y; break;
        case 3: r = sin(x); break;
        case 4: r = cos(x); break;
        case 5: sincos(x, &s, &c); r = s; break;
        case 6: sincos(x, &s, &c); r = c; break;
        case 7: r = x * y + 1.0; break;  // must NOT be contracted to an fma
        case 8: r = MathFast::sqrt64(x); break;
        case 12: sincos_small(x, s, c); r = s; break;
        case 13: sincos_small(x, s, c); r = c; break;
        case 14: { MathSpec m; r = m.sqrt64(x); if (m.bad) r = ::sqrt(x); } break;
        case 15: { MathSpec m; r = m.div3(d3(x, 1.0, -x), y).x; if (m.bad) r = x / y; } break;
        case 9: r = MathFast::div3(d3(x, x * 0.5, -x), y).x; break;
        case 10: r = MathFast::div3(d3(1.0, x, 0.0), y).y; break;
        case 11: r = MathFast::div3(d3(y, -0.0, x), y).z; break;
    }
    out[i] = r;
}

}  // namespace rtm
