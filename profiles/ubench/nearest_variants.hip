// nearest_variants.hip — code-shape experiments for the nearest-hit loop (7-sphere Cornell box) at a
// fixed 4 waves/SIMD.  Standalone: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I. ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../raytracingmin_amd/csrc/rtm_path.h"
using namespace rtm;

// V: 0 literal loop (sphere_test), 1 sphere_update loop w/ ballot skips, 2 select-only no skips,
//    3 batch4, 4 prefetched scalar loads + select-only, 5 LDS broadcast geometry + select-only,
//    6 fully unrolled N=7 select-only
template <class M>
__device__ __forceinline__ void upd_noskip(const double4 g, const D3 org, const D3 dir, const int i, double& dis, int& id) {
    const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);
    const double b = dot(p_o, dir);
    const double D4 = b * b - dot(p_o, p_o) + g.w;
    const double sq = M::sqrt64(D4);
    const double t1 = b - sq, t2 = b + sq;
    const double t = sel_f64(lane_mask(t1 > 0.001), t1, t2);
    const LaneMask k = lane_mask((t < dis) && !(t < (double)1e-5f));
    dis = sel_f64(k, t, dis);
    id = sel_i32(k, i, id);
}
template <class M>
__device__ __forceinline__ void upd_cnd(const double4 g, const D3 org, const D3 dir, const int i, double& dis, int& id) {
    const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);
    const double b = dot(p_o, dir);
    const double D4 = b * b - dot(p_o, p_o) + g.w;
    const double sq = M::sqrt64(D4);
    const double t1 = b - sq, t2 = b + sq;
    const double t = (t1 > 0.001) ? t1 : t2;
    const bool k = (t < dis) && !(t < (double)1e-5f);
    dis = k ? t : dis;
    id = k ? i : id;
}

template <int V>
__global__ __launch_bounds__(64) void bench(SceneView scene, int reps, double* out, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* lg = reinterpret_cast<double*>(lds_raw);
    SceneGlobal sc;
    sc.v = scene;
    const int lane = threadIdx.x, n = scene.n;
    if (V == 5) {
        for (int i = lane; i < n * 4; i += 64) lg[i] = reinterpret_cast<const double*>(scene.geom)[i];
        __syncthreads();
    }
    D3 dir = normalize(d3(-0.8 + 0.025 * (lane & 7) + 1e-3 * blockIdx.x, -0.5 + 0.12 * (lane >> 3), 1.0));
    D3 org = d3(0, 0, -10);
    double acc = 0.0;
    int ids = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r) {
        double dis = DBL_MAX;
        int id = -1;
        if constexpr (V == 0) id = nearest_hit<MathFast, 1>(sc, org, dir, dis);
        if constexpr (V == 1) { for (int i = 0; i < n; ++i) sphere_update<MathFast>(sc.geom_uniform(i), org, dir, i, dis, id); }
        if constexpr (V == 2) { for (int i = 0; i < n; ++i) upd_noskip<MathFast>(sc.geom_uniform(i), org, dir, i, dis, id); }
        if constexpr (V == 3) id = nearest_hit<MathFast, 4>(sc, org, dir, dis);
        if constexpr (V == 4) {
            double4 g = sc.geom_uniform(0);
            for (int i = 0; i < n; ++i) {
                const double4 gn = sc.geom_uniform(i + 1 < n ? i + 1 : i);
                upd_noskip<MathFast>(g, org, dir, i, dis, id);
                g = gn;
            }
        }
        if constexpr (V == 5) {
            for (int i = 0; i < n; ++i) {
                const double4 g = double4{lg[i * 4], lg[i * 4 + 1], lg[i * 4 + 2], lg[i * 4 + 3]};
                upd_noskip<MathFast>(g, org, dir, i, dis, id);
            }
        }
        if constexpr (V == 6) {
#pragma unroll
            for (int i = 0; i < 7; ++i) upd_noskip<MathFast>(sc.geom_uniform(i), org, dir, i, dis, id);
        }
        if constexpr (V == 7) { for (int i = 0; i < n; ++i) upd_cnd<MathFast>(sc.geom_uniform(i), org, dir, i, dis, id); }
        if constexpr (V == 8) {
#pragma unroll
            for (int i = 0; i < 7; ++i) upd_cnd<MathFast>(sc.geom_uniform(i), org, dir, i, dis, id);
        }
        if constexpr (V == 9) {
#pragma unroll
            for (int i = 0; i < 7; ++i) upd_cnd<MathRef>(sc.geom_uniform(i), org, dir, i, dis, id);
        }
        if constexpr (V == 10) {  // unrolled x7, one sqrt guard for all seven, ?: selects
            double b[7], D4[7], sq[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const double4 g = sc.geom_uniform(i);
                const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);
                b[i] = dot(p_o, dir);
                D4[i] = b[i] * b[i] - dot(p_o, p_o) + g.w;
            }
            MathFast::sqrt64_batch<7>(D4, sq);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const double t1 = b[i] - sq[i], t2 = b[i] + sq[i];
                const double t = (t1 > 0.001) ? t1 : t2;
                const bool k = (t < dis) && !(t < (double)1e-5f);
                dis = k ? t : dis;
                id = k ? i : id;
            }
        }
        if constexpr (V == 11) {  // unrolled x7, literal sphere_test (branches) 
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                double t;
                if (sphere_test<MathFast>(sc.geom_uniform(i), org, dir, t) && t < dis && t > 0) { dis = t; id = i; }
            }
        }
        if constexpr (V == 12) {  // unrolled x7, ?: selects with per-sphere whole-wave-miss skip
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const double4 g = sc.geom_uniform(i);
                const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);
                const double b = dot(p_o, dir);
                const double D4 = b * b - dot(p_o, p_o) + g.w;
                if (__builtin_amdgcn_ballot_w64(D4 >= 0.0) != 0) {
                    const double sq = MathFast::sqrt64(D4);
                    const double t1 = b - sq, t2 = b + sq;
                    const double t = (t1 > 0.001) ? t1 : t2;
                    const bool k = (t < dis) && !(t < (double)1e-5f);
                    dis = k ? t : dis;
                    id = k ? i : id;
                }
            }
        }
        acc += dis;
        ids += id;
        org.x += 1e-7 * (id + 2);
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[(size_t)blockIdx.x * 64 + lane] = acc + ids;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, SceneView sv, double* out, unsigned long long* cyc, size_t lds) {
    const int blocks = 256 * 32, reps = 2000;
    bench<V><<<blocks, 64, lds>>>(sv, 4, out, cyc);
    bench<V><<<blocks, 64, lds>>>(sv, reps, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::vector<double> ho(64);
    hipMemcpy(ho.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    printf("%-44s %8.1f cycles/cast/wave  (%7.1f per SIMD at 4 waves)  check %.6f\n", name, avg / blocks / reps,
           avg / blocks / reps / 4, ho[5]);
}

int main() {
    // Cornell box geometry (tests/golden/scenes/cornellBoxSetting.json): light + six walls
    const double c[7][4] = {{0, 10, 0, 25}, {10010, 0, 0, 1e8}, {-10010, 0, 0, 1e8}, {0, 10010, 0, 1e8},
                            {0, -10010, 0, 1e8}, {0, 0, 10010, 1e8}, {0, 0, -10010, 1e8}};
    double* dg;
    hipMalloc(&dg, sizeof c);
    hipMemcpy(dg, c, sizeof c, hipMemcpyHostToDevice);
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 32 * 64 * 8);
    hipMalloc(&cyc, 256 * 32 * 8);
    SceneView sv{(const double4*)dg, nullptr, 7};
    const size_t lds = 9500;  // 16 waves per CU = 4 per SIMD
    run<0>("0 literal sphere_test loop", sv, out, cyc, lds);
    run<1>("1 sphere_update (bfi selects, ballot skips)", sv, out, cyc, lds);
    run<2>("2 bfi selects, no skips", sv, out, cyc, lds);
    run<7>("7 ?: selects (cndmask), no skips", sv, out, cyc, lds);
    run<3>("3 batch of 4 (8 tests)", sv, out, cyc, lds);
    run<4>("4 prefetched s_load, bfi, no skips", sv, out, cyc, lds);
    run<5>("5 LDS-broadcast geometry, bfi, no skips", sv, out, cyc, lds);
    run<6>("6 unrolled x7, bfi, no skips", sv, out, cyc, lds);
    run<8>("8 unrolled x7, ?: selects", sv, out, cyc, lds);
    run<9>("9 unrolled x7, ?: selects, MathRef sqrt", sv, out, cyc, lds);
    run<10>("10 unrolled x7, one sqrt guard, ?: selects", sv, out, cyc, lds);
    run<11>("11 unrolled x7, literal sphere_test branches", sv, out, cyc, lds);
    run<12>("12 unrolled x7, ?: selects, wave-miss skip", sv, out, cyc, lds);
    return 0;
}
