// rtm_wavefront.h — large scenes (BASELINE configs[4], 100 k spheres) as a wavefront pipeline.
// Included by rtm_kernels.hip after RenderParams / primary_dir / store_pixel / wave_add_counter.
//
// With 10^5 spheres a ray cast is ~2 M instructions of brute-force intersection against ~500 of
// shading, so the two are split into kernels with their own register budgets:
//   nearest hit  one lane per ACTIVE pixel (compacted index list), nothing live but the ray and the
//                running nearest hit: wf_nearest_f32_kernel — the sphere list read through the scalar
//                cache into SGPRs, a conservative rejection test in packed fp32 in front of the
//                reference's arithmetic, per-lane candidate lists for what it cannot reject.  (Its three
//                predecessors — LDS tiles, scalar stream without a test, fp64 test — and the two-rays-per-
//                instruction twin were A/B-ed in rounds 1-2 and retired from the product:
//                profiles/r3/retired_variants.patch)
//                Every ray still meets every sphere, in index order, and every possible hit is
//                decided by the reference's own arithmetic.
//   wf_shade     the rest of PathTracing (src/Renderer.cpp:75-117), the back-to-front fold, the
//                per-sample accumulate and path regeneration, with the per-pixel state in HBM (SoA),
//                then wave-ballot/prefix COMPACTION of the pixels that still have samples into the
//                next active list.
// One path per pixel at a time and samples in the reference's order, so every pixel sees exactly the
// arithmetic of render_tiles_kernel: the image is bit-identical to the other variants.
// Per-pixel state traffic (~300 B read + written per cast) is noise next to the intersection work.
#pragma once

namespace rtm {

struct WfState {
    double* org;    // [3][npix]
    double* dir;    // [3][npix]
    double* pdir;   // [3][npix] cached primary direction of the current sub-pixel
    double* acc;    // [3][npix]
    double* hit_t;  // [npix]
    int* hit_id;    // [npix]
    unsigned* rng_ctr;
    unsigned* rng_k1;
    unsigned* n;     // sample index
    int* left;       // samples left in the current sub-pixel
    int* depth;
    unsigned* rec;   // [levels][npix] hit records
    unsigned* active[2];
    unsigned* n_active;  // [2]
    unsigned npix;
    int levels;
    // Few active rays (a small frame, a strip, the tail of a frame): the sphere list is cut into wf_parts(active count)
    // ranges and every range gets its own blocks, so that the chip is filled by rays x ranges instead of idling at one
    // or two waves per SIMD on a latency-bound scalar stream.  Part 0 writes hit_t / hit_id as always; part q >= 1
    // writes slot i of the active list into part_t / part_id[(q - 1) * part_slots + i]; the shade kernel takes the
    // smallest t with strict < in part order — parts ascend in sphere index, so the lowest index still wins ties.
    // Both kernels derive the number of parts from the DEVICE's active count (the host never has to know it).
    unsigned part_slots;   // capacity of one part's slot arrays
    double* part_t;
    int* part_id;
};
constexpr int kWfCandCap = 16;               // candidate slots per lane of wf_nearest_f32_kernel (LDS: 4 B x BLOCK each)
constexpr unsigned kWfPartSlots = 1u << 19;  // the split is used only while the active list is this short
constexpr int kWfMaxParts = 8;
// parts of the sphere list for `na` active rays and `n` spheres: doubled while the list is long enough (>= 2048 spheres
// per part), the rays few enough for the slot arrays, and rays x parts short of ~16 waves per SIMD's worth
__host__ __device__ inline int wf_parts(unsigned na, int n) {
    int parts = 1;
    while (parts < kWfMaxParts && na <= kWfPartSlots && (unsigned long long)na * (unsigned)parts * 2ull <= (1ull << 20) &&
           n / (parts * 2) >= 2048)
        parts *= 2;
    return parts;
}
// blocks of 256 rays x parts a nearest-hit launch needs for an active count of AT MOST `na` (the product is not
// monotonic in the count: fewer rays may mean more parts)
inline unsigned wf_nearest_grid(unsigned na, int n) {
    unsigned best = 1;
    const unsigned cand[4] = {na, na < kWfPartSlots ? na : kWfPartSlots, na < (1u << 19) ? na : (1u << 19), na < (1u << 18) ? na : (1u << 18)};
    const unsigned cand2 = na < (1u << 17) ? na : (1u << 17);
    for (unsigned c : {cand[0], cand[1], cand[2], cand[3], cand2}) {
        if (c == 0) continue;
        const unsigned g = ((c + 255u) / 256u) * (unsigned)wf_parts(c, n);
        best = g > best ? g : best;
    }
    return best;
}

__device__ __forceinline__ void wf_pixel_xy(const RenderParams& P, unsigned p, int& x, int& y) {
    x = (int)(p % (unsigned)P.W);
    const int lr = (int)(p / (unsigned)P.W);  // row of this call's compact output
    y = band_row(P, lr >> 3, lr & 7);
}

__global__ __launch_bounds__(256) void wf_init_kernel(const RenderParams P, const WfState S) {
    const unsigned p = blockIdx.x * 256 + threadIdx.x;
    if (p >= S.npix) return;
    if (p == 0) {  // every pixel starts active; the other list is empty
        S.n_active[0] = S.npix;
        S.n_active[1] = 0u;
    }
    int x, y;
    wf_pixel_xy(P, p, x, y);
    const D3 pd = primary_dir(P, x, y, 1, 1);
    const unsigned N = S.npix;
    S.org[p] = P.cam_org.x; S.org[N + p] = P.cam_org.y; S.org[2 * N + p] = P.cam_org.z;
    S.dir[p] = pd.x; S.dir[N + p] = pd.y; S.dir[2 * N + p] = pd.z;
    S.pdir[p] = pd.x; S.pdir[N + p] = pd.y; S.pdir[2 * N + p] = pd.z;
    S.acc[p] = 0.0; S.acc[N + p] = 0.0; S.acc[2 * N + p] = 0.0;
    const RngStream r = rng_open(rng_pixel_key(P.seed_mult, (uint32_t)y * (uint32_t)P.W + (uint32_t)x), 0u);
    S.rng_ctr[p] = r.ctr;
    S.rng_k1[p] = r.k1;
    S.n[p] = 0;
    S.left[p] = P.S;
    S.depth[p] = 0;
    S.active[0][p] = p;
}

// Auxiliary per-sphere data for the rejection tests.  aux = [R2][r2max][exponent sum][count] then
// wprime[n_pad] then the float list.  The margins of the tests scale with R2 = max c.c and max r2, so a
// few spheres far larger or farther than the rest (the wall spheres of a Cornell-style box: r = 1e5
// around objects of size 1) would make them useless for everything else.  Such OUTLIERS — squared
// magnitude max(c.c, r2) more than 2^10 above the scene's geometric mean — are left out of R2 / r2max
// and get w' = +inf (and a zero float centre): their test value is +inf, so they always take the
// reference arithmetic, which costs a handful of exact tests per ray.
__global__ __launch_bounds__(256) void wf_scene_scale_kernel(const double4* __restrict__ geom, const int n,
                                                             long long* __restrict__ acc /* [exponent sum, count] */) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double4 g = geom[i];
    const double cc = g.x * g.x + g.y * g.y + g.z * g.z;
    const double m = cc < g.w ? g.w : cc;
    if (m > 0.0 && m < HUGE_VAL) {
        atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)(long long)__builtin_amdgcn_frexp_exp(m));
        atomicAdd(reinterpret_cast<unsigned long long*>(acc + 1), 1ull);
    }
}

__global__ __launch_bounds__(256) void wf_scene_aux_kernel(const double4* __restrict__ geom, const int n, const int n_pad,
                                                           double* __restrict__ wprime, unsigned long long* __restrict__ bounds,
                                                           float4* __restrict__ geom32, float4* __restrict__ geom32s) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    float* q = geom32 ? reinterpret_cast<float*>(geom32) + (size_t)(i >> 1) * 8 + (i & 1) : nullptr;
    if (i >= n) {  // padding: a sphere that can never pass the test
        wprime[i] = -HUGE_VAL;
        if (q) {
            q[0] = q[2] = q[4] = 0.f;
            q[6] = -HUGE_VALF;
        }
        if (geom32s) geom32s[i] = float4{0.f, 0.f, 0.f, -HUGE_VALF};
        return;
    }
    const long long* acc = reinterpret_cast<const long long*>(bounds + 2);
    const long long count = acc[1];
    const int limit = count > 0 ? (int)(acc[0] / count) + 10 : 0x7FFFFFFF;
    const double4 g = geom[i];
    const double cc = g.x * g.x + g.y * g.y + g.z * g.z;
    const double m = cc < g.w ? g.w : cc;
    const bool outlier = !(m < HUGE_VAL) || (m > 0.0 && __builtin_amdgcn_frexp_exp(m) > limit);
    wprime[i] = outlier ? HUGE_VAL : g.w - cc;
    if (q) {  // spheres 2m, 2m+1 interleaved: cx0 cx1 cy0 cy1 cz0 cz1 w0 w1 (operands of packed fp32 math)
        q[0] = outlier ? 0.f : (float)g.x;
        q[2] = outlier ? 0.f : (float)g.y;
        q[4] = outlier ? 0.f : (float)g.z;
        q[6] = outlier ? HUGE_VALF : (float)(g.w - cc);
    }
    if (geom32s)
        geom32s[i] = outlier ? float4{0.f, 0.f, 0.f, HUGE_VALF} : float4{(float)g.x, (float)g.y, (float)g.z, (float)(g.w - cc)};
    if (outlier) return;
    // max over non-negative doubles == max over their bit patterns; NaN / negative values stay out
    if (cc >= 0.0) atomicMax(bounds + 0, (unsigned long long)__double_as_longlong(cc));
    if (g.w >= 0.0) atomicMax(bounds + 1, (unsigned long long)__double_as_longlong(g.w));
}

// The nearest-hit pass of the large-scene pipeline.
//  * The walk over the sphere list is wave-uniform, so a chunk of K spheres is one or two s_loads into SGPRs
//    (two SGPR buffers, one in flight while the other is tested) and the VALU takes its sphere operands
//    straight from SGPRs: no LDS round trip, no tile staging, no barriers.  All waves stream the same list,
//    so the scalar caches and L2 absorb the re-reads.
//  * 99.99 % of (ray, sphere) pairs of such a scene have a negative discriminant and are only ever REJECTED;
//    for that decision the reference's 16 non-fused operations are not needed, only a guarantee.  With
//    P = c - o:  D4 = (P.d)^2 - P.P + r2 = (c.d - o.d)^2 + (r2 - c.c) + 2 c.o - o.o.  Both an FMA evaluation of
//    that identity in fp64 and the reference's own are within  27 u64 (1 + d.d)(max c.c + o.o + max r2)  of the
//    exact discriminant (standard dot-product bounds): the "fp64 margin" below.
// The test runs in SINGLE precision (packed: two spheres per instruction), in front of the reference arithmetic.
// Same identity, arranged so that the per-sphere data is (cx, cy, cz, w') as floats:
//   D4 = (c.d)^2 + c.e + w' + k,   e = 2o - 2(o.d)d,  k = (o.d)^2 - o.o   (per-ray, computed in fp64)
// = 1 mul + 6 fma + 1 add in fp32.  With u = 2^-24, rounding every input to float and every operation
// once gives (Higham's gamma_n bounds; sum |c_i||x_i| <= R |x|_1, R^2 = max c.c):
//   |D4_f32 - D4| <= u [ 11.7 R^2 d.d + 7.2 R |e|_1 + 6.1 |k| + 6.1 (R^2 + max r2) ]
// plus underflow (<= 17 x 2^-126 x max(1, |d|, |e|, 2R|d|)) and the fp64 margin above.  The kernel uses 16 u (...) + 2^-110 (1 + d.d + |e|_1 + R^2) + the fp64
// margin, rounded up; a chunk none of whose spheres reaches -margin in any lane is skipped, and in any
// other chunk the spheres some lane could not reject get the reference arithmetic, in index order.
// When the magnitude sum is >= 2^100 (a product could overflow a float) or not finite, the margin is
// infinite and nothing is rejected; below that every fp32 intermediate is finite, so no NaN can hide
// a sphere from the max.
template <class M, int BLOCK, int K, bool PACKED = true>
__global__ __launch_bounds__(BLOCK) void wf_nearest_f32_kernel(const RenderParams P, const WfState S, const int cur) {
    static_assert(K == 4 || K == 8, "one or two 64-byte scalar loads per chunk");
    static_assert(BLOCK == 256, "wf_nearest_grid counts blocks of 256 rays");
    const unsigned na = S.n_active[cur];
    // the NEXT list's count starts at zero: the shade launch behind this one appends to it (the slot was this trip's
    // predecessor's `cur`, which nothing reads any more)
    if (blockIdx.x == 0 && threadIdx.x == 0) S.n_active[cur ^ 1] = 0u;
    if (na == 0u) return;  // the frame is finished: the remaining launches of a fixed trip budget fall through
    const int parts = wf_parts(na, P.scene.n);
    const unsigned part_blocks = (na + (unsigned)BLOCK - 1u) / (unsigned)BLOCK;
    const unsigned part = blockIdx.x / part_blocks;       // wave-uniform
    const unsigned ray_block = blockIdx.x % part_blocks;
    if (part >= (unsigned)parts) return;  // the grid is sized for the worst (count, parts) pair
    const unsigned i = ray_block * (unsigned)BLOCK + threadIdx.x;
    const bool live = i < na;
    const unsigned N = S.npix;
    const unsigned p = S.active[cur][live ? i : na - 1];
    const D3 org = d3(S.org[p], S.org[N + p], S.org[2 * N + p]);
    const D3 dir = d3(S.dir[p], S.dir[N + p], S.dir[2 * N + p]);
    double dis = DBL_MAX;
    int id = -1;
    const int n = P.scene.n;

    typedef const __attribute__((address_space(4))) double* ConstF64Ptr;
    typedef const __attribute__((address_space(4))) float* ConstF32Ptr;
    ConstF64Ptr bnd = (ConstF64Ptr)(unsigned long long)P.scene.bounds;
    const double R2 = bnd[0], r2max = bnd[1];
    const double od = __builtin_fma(org.z, dir.z, __builtin_fma(org.y, dir.y, org.x * dir.x));
    const double oo = __builtin_fma(org.z, org.z, __builtin_fma(org.y, org.y, org.x * org.x));
    const double dd = __builtin_fma(dir.z, dir.z, __builtin_fma(dir.y, dir.y, dir.x * dir.x));
    const double od2 = od + od;
    const D3 e = d3(__builtin_fma(-od2, dir.x, org.x + org.x), __builtin_fma(-od2, dir.y, org.y + org.y),
                    __builtin_fma(-od2, dir.z, org.z + org.z));
    const double kq = __builtin_fma(od, od, -oo);
    const double R = __builtin_sqrt(R2) * (1.0 + 0x1p-40);
    const double e1 = __builtin_fabs(e.x) + __builtin_fabs(e.y) + __builtin_fabs(e.z);
    const double m64 = 0x1p-42 * ((1.0 + dd) * (R2 + oo + r2max));
    const double mag = R2 * dd + R * e1 + __builtin_fabs(kq) + R2 + r2max;  // bounds every fp32 intermediate
    // rounding (u = 2^-24; the derived constants are 11.7, 6.0 and 5.0 for the form with k outside the sum — DESIGN.md,
    // appendix — rounded up; round 1 used 16 for every term: twice the margin.  Halving it did not change the time: the
    // chunks that escalate do so because some lane's ray really has D4 >= 0 for one of their spheres) + underflow
    const double m32 = 0x1p-24 * (12.0 * (R2 * dd) + 7.0 * (R * e1) + 6.0 * (R2 + r2max)) + 0x1p-110 * (1.0 + dd + e1 + R2);
    // The per-ray constant k = (o.d)^2 - o.o never enters the per-sphere arithmetic: D4 >= -margin is tested as
    //   (c.d)^2 + (c.e + w') >= -margin - k =: thr,
    // one packed instruction per sphere pair fewer than adding k inside (8 -> 7).  thr is formed in fp64 and
    // rounded DOWN to float (a threshold that is too low only lets more spheres through to the exact test);
    // the margin carries 2 u |k| more for the value the sum would have had with k inside.
    // Magnitudes that could overflow single precision (or a non-finite ray): reject nothing.
    float neg_margin = -HUGE_VALF;
    if (mag < 0x1p100) {
        const double thr64 = -((m32 + m64 + 0x1p-23 * __builtin_fabs(kq)) * (1.0 + 0x1p-20)) - kq;
        float thr = (float)thr64;
        if ((double)thr > thr64) thr = __uint_as_float(thr > 0.f ? __float_as_uint(thr) - 1u :
                                                     thr < 0.f ? __float_as_uint(thr) + 1u : 0x80000001u);  // next float below
        neg_margin = thr;
    }
    const float dx = (float)dir.x, dy = (float)dir.y, dz = (float)dir.z;
    const float ex = (float)e.x, ey = (float)e.y, ez = (float)e.z;

    typedef float f2 __attribute__((ext_vector_type(2)));
    struct Pair {
        f2 x, y, z, w;  // two spheres
    };
    auto load32 = [&](int j, Pair (&g)[K / 2]) {
        ConstF32Ptr q = (ConstF32Ptr)(unsigned long long)(reinterpret_cast<const float*>(P.scene.geom32) + (size_t)j * 4);
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {
            g[k].x = f2{q[8 * k], q[8 * k + 1]};
            g[k].y = f2{q[8 * k + 2], q[8 * k + 3]};
            g[k].z = f2{q[8 * k + 4], q[8 * k + 5]};
            g[k].w = f2{q[8 * k + 6], q[8 * k + 7]};
        }
    };
    const f2 dx2 = f2{dx, dx}, dy2 = f2{dy, dy}, dz2 = f2{dz, dz}, ex2 = f2{ex, ex}, ey2 = f2{ey, ey}, ez2 = f2{ez, ez};
    // per-lane candidate lists in LDS, [slot][thread] (conflict-free: a slot is one dword per thread)
    extern __shared__ unsigned cand[];
    int cnt = 0;
    // settle every lane's candidates with the reference's arithmetic (src/SettingData.cpp:197-226 + the caller's
    // acceptance, select form — sphere_update), each lane walking ITS list in the order it was noted = index order, so
    // strict < still lets the lowest index win ties
    auto drain = [&]() {
        for (int e = 0; __builtin_amdgcn_ballot_w64(e < cnt) != 0; ++e) {
            if (e < cnt) {
                const int jj = (int)cand[e * BLOCK + threadIdx.x];
                const double4 g = P.scene.geom[jj];
                const D3 p_o = d3(g.x - org.x, g.y - org.y, g.z - org.z);  // :198
                const double b = dot(p_o, dir);                            // :199
                const double D4 = b * b - dot(p_o, p_o) + g.w;             // :200
                const double sq = M::sqrt64(D4);                           // :205 (D4 < 0: NaN, nothing is accepted)
                const double t1 = b - sq, t2 = b + sq;
                const double t = (t1 > 0.001) ? t1 : t2;
                const bool accept = (t < dis) && !(t < (double)1e-5f);
                dis = accept ? t : dis;
                id = accept ? jj : id;
            }
        }
        cnt = 0;
    };
    auto test = [&](const Pair (&g)[K / 2], int j) {
        f2 tq[K / 2];
        f2 top2;
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {
            if constexpr (PACKED) {
                const f2 uq = __builtin_elementwise_fma(g[k].z, dz2, __builtin_elementwise_fma(g[k].y, dy2, g[k].x * dx2));
                const f2 vq = __builtin_elementwise_fma(g[k].z, ez2, __builtin_elementwise_fma(g[k].y, ey2,
                                  __builtin_elementwise_fma(g[k].x, ex2, g[k].w)));
                tq[k] = __builtin_elementwise_fma(uq, uq, vq);
            } else {
                const float u0 = __builtin_fmaf(g[k].z.x, dz, __builtin_fmaf(g[k].y.x, dy, g[k].x.x * dx));
                const float u1 = __builtin_fmaf(g[k].z.y, dz, __builtin_fmaf(g[k].y.y, dy, g[k].x.y * dx));
                const float v0 = __builtin_fmaf(g[k].z.x, ez, __builtin_fmaf(g[k].y.x, ey, __builtin_fmaf(g[k].x.x, ex, g[k].w.x)));
                const float v1 = __builtin_fmaf(g[k].z.y, ez, __builtin_fmaf(g[k].y.y, ey, __builtin_fmaf(g[k].x.y, ex, g[k].w.y)));
                tq[k] = f2{__builtin_fmaf(u0, u0, v0), __builtin_fmaf(u1, u1, v1)};
            }
            top2 = k ? __builtin_elementwise_max(top2, tq[k]) : tq[k];  // v_pk_max_f32 skips NaN operands like v_max_f32
        }
        const float top = __builtin_fmaxf(top2.x, top2.y);
        if (__builtin_amdgcn_ballot_w64(top >= neg_margin) == 0) return;
        // rare (~6 % of the chunks): a sphere some lane could not reject.  It is not settled here, where the exact
        // arithmetic would run in all 64 lanes for the one that needs it (a quarter of the kernel's VALU work before),
        // but noted in THAT lane's candidate list; the lists are settled 64 lanes at a time, in index order (drain).
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float t = (k & 1) ? tq[k / 2].y : tq[k / 2].x;
            const bool pass = !(t < neg_margin);
            if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;
            if (pass) {
                cand[cnt * BLOCK + threadIdx.x] = (unsigned)(j + k);
                ++cnt;
            }
        }
        if (__builtin_amdgcn_ballot_w64(cnt > kWfCandCap - K) != 0) drain();  // the next chunk might not fit
    };

    // this block's range of the sphere list: all of it, or part `part` of `parts` (cut on chunk-pair boundaries)
    const int n_pairs_all = n / (2 * K);
    const int pairs_per = (n_pairs_all + parts - 1) / parts;
    const int j_begin = (int)part * pairs_per * 2 * K;
    const int j_end_full = ((int)part + 1 == parts) ? n_pairs_all * 2 * K
                                                   : (((int)part + 1) * pairs_per < n_pairs_all ? ((int)part + 1) * pairs_per
                                                                                              : n_pairs_all) * 2 * K;
    Pair a[K / 2], b[K / 2];
    if (j_begin < j_end_full) load32(j_begin, a);
    for (int j = j_begin; j < j_end_full; j += 2 * K) {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): A has arrived
        __builtin_amdgcn_sched_barrier(0);
        load32(j + K, b);
        __builtin_amdgcn_sched_barrier(0);
        test(a, j);
        const int jn = (j + 2 * K < j_end_full) ? j + 2 * K : j;  // the last trip re-reads its own chunk
        __builtin_amdgcn_s_waitcnt(0xC07F);  // B has arrived
        __builtin_amdgcn_sched_barrier(0);
        load32(jn, a);
        __builtin_amdgcn_sched_barrier(0);
        test(b, j + K);
    }
    drain();
    if ((int)part + 1 == parts) {  // the spheres after the last full chunk pair belong to the last part
        for (int j = n_pairs_all * 2 * K; j < n; ++j) {
            double4 g1[1] = {load_geom_uniform(P.scene.geom, j)};
            sphere_chunk_g<M, 1, true>(g1, j, org, dir, dis, id);
        }
    }
    if (live) {
        if (part == 0u) {
            S.hit_id[p] = id;
            S.hit_t[p] = dis;
        } else {
            S.part_id[(size_t)(part - 1u) * S.part_slots + i] = id;
            S.part_t[(size_t)(part - 1u) * S.part_slots + i] = dis;
        }
    }
}

// Shade / fold / accumulate / regenerate for the active pixels, then compaction of the survivors.
__global__ __launch_bounds__(256) void wf_shade_kernel(const RenderParams P, const WfState S, const int cur) {
    const unsigned na = S.n_active[cur];
    if (blockIdx.x * 256u >= na) return;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const bool live = i < na;
    const unsigned N = S.npix;
    PathCounters pc = {0, 0, 0};
    bool overflow = false;
    bool alive = false;
    unsigned p = 0;
    if (live) {
        p = S.active[cur][i];
        SceneGlobal sc;
        sc.v = P.scene;
        D3 org = d3(S.org[p], S.org[N + p], S.org[2 * N + p]);
        D3 dir = d3(S.dir[p], S.dir[N + p], S.dir[2 * N + p]);
        RngStream rng{S.rng_ctr[p], S.rng_k1[p]};
        int depth = S.depth[p];
        unsigned n = S.n[p];
        auto push = [&](int d, int id) {
            if (d < S.levels)
                S.rec[(size_t)d * N + p] = (unsigned)id;
            else
                overflow = true;
        };
        // nearest hit over the parts of the sphere list (one part: what the search wrote)
        int hit_id = S.hit_id[p];
        double hit_t = S.hit_t[p];
        const int parts = wf_parts(na, P.scene.n);
        for (int q = 1; q < parts; ++q) {  // wave-uniform count; ascending sphere index, strict <: lowest index wins ties
            const int idq = S.part_id[(size_t)(q - 1) * S.part_slots + i];
            const double tq = S.part_t[(size_t)(q - 1) * S.part_slots + i];
            const bool better = idq >= 0 && tq < hit_t;
            hit_id = better ? idq : hit_id;
            hit_t = better ? tq : hit_t;
        }
        D3 term;
        bool cont = path_shade_spec(sc, hit_id, hit_t, P.mode, P.max_bounces, org, dir, depth, rng, term,
                                    pc, push);
        if (cont && overflow) {
            cont = false;
            term = d3(0, 0, 0);
            depth = 0;
        }
        if (!cont) {
            const D3 L = path_fold(sc, term, depth, [&](int d) { return (int)S.rec[(size_t)d * N + p]; });
            const bool pow2 = P.inv_s != 0.0;
            const D3 cal = pow2 ? ((L * P.inv_ss) * P.inv_ss) * P.inv_s : ((L / P.dSS) / P.dSS) / P.dS;
            const D3 add = clamp01_d3(cal);
            const D3 acc = d3(S.acc[p], S.acc[N + p], S.acc[2 * N + p]) + add;
            S.acc[p] = acc.x; S.acc[N + p] = acc.y; S.acc[2 * N + p] = acc.z;
            ++n;
            int left = S.left[p] - 1;
            int x, y;
            wf_pixel_xy(P, p, x, y);
            if (n < P.total_samples) {
                D3 pd;
                if (left == 0) {
                    left = P.S;
                    const int sub = (int)(n / (unsigned)P.S);
                    pd = primary_dir(P, x, y, sub / P.SS + 1, sub % P.SS + 1);
                    S.pdir[p] = pd.x; S.pdir[N + p] = pd.y; S.pdir[2 * N + p] = pd.z;
                } else {
                    pd = d3(S.pdir[p], S.pdir[N + p], S.pdir[2 * N + p]);
                }
                org = P.cam_org;
                dir = pd;
                depth = 0;
                rng = rng_open(rng_pixel_key(P.seed_mult, (uint32_t)y * (uint32_t)P.W + (uint32_t)x), n);
            } else {
                store_pixel(P, true, x, y, acc);
            }
            S.left[p] = left;
            S.n[p] = n;
        }
        alive = n < P.total_samples;
        if (alive) {
            S.org[p] = org.x; S.org[N + p] = org.y; S.org[2 * N + p] = org.z;
            S.dir[p] = dir.x; S.dir[N + p] = dir.y; S.dir[2 * N + p] = dir.z;
            S.rng_ctr[p] = rng.ctr;
            S.rng_k1[p] = rng.k1;
            S.depth[p] = depth;
        }
    }
    // ---- compaction: wave ballot + prefix, one atomic per wave ----
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(alive);
    if (mask) {
        const unsigned lane = threadIdx.x & 63;
        const unsigned count = (unsigned)__builtin_popcountll(mask);
        unsigned base = 0;
        if (lane == (unsigned)__builtin_ctzll(mask)) base = atomicAdd(&S.n_active[cur ^ 1], count);
        base = (unsigned)__shfl((int)base, (int)__builtin_ctzll(mask), 64);
        if (alive) {
            const unsigned before = (unsigned)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            S.active[cur ^ 1][base + before] = p;
        }
    }
    if (P.counters) {
        wave_add_counter(P.counters + 0, pc.casts);
        wave_add_counter(P.counters + 1, pc.bounces);
        wave_add_counter(P.counters + 2, pc.draws);
        if (overflow) atomicOr(P.counters + 3, 1ull);
    }
}

}  // namespace rtm
