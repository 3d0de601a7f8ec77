// Large scenes through the uniform grid (variant 17): the render kernel and its finalize pass.
//
// The pixel/sample loop nest of src/Renderer.cpp:215-250 with PathTracing (:57-117) underneath, like render_tiles_kernel,
// but shaped by what a grid walk costs: the walk of one ray (rtm_path.h: GridWalk) takes a number of trips that is
// exponentially distributed (the free path), and pixels differ by 2-3x in casts per sample (a pixel whose primary ray
// ends on a light takes one cast per sample, its neighbour three).  A wave that keeps lane = pixel and shades its 64
// lanes in lockstep therefore runs at a quarter of its lanes (measured: 24 % active lanes, profiles/r3/README.md).  So:
//
//   * one wave per 8x8 tile, but a lane is not tied to a pixel: the tile's (pixel, sample) pairs — 64 x spp UNITS, in
//     sample-major order — are dealt to lanes as they become free (an LDS counter).  The counter RNG is keyed by
//     (seed, global pixel, sample, draw), so any lane can trace any unit; all 64 lanes finish within one unit of each
//     other whatever the pixels cost.
//   * lanes walk independently and are shaded in groups: each trip of the loop advances the walks until at least
//     kGridShadeAt8 / 8 (five eighths) of the busy lanes have finished theirs, shades those (they start their next ray, or fold their
//     path and take the next unit) and leaves the others walking.
//   * the sum over a pixel's samples must be added in sample order (src/Renderer.cpp:241-242; fp64 addition does not
//     commute bitwise), and the samples of a pixel now finish on different lanes in any order: every sample's term
//     (cal / SS / SS / S, clamped — :240) goes to memory, [tile][sample][pixel] slots of 32 bytes, and
//     grid_finalize_kernel adds them per pixel in sample order and stores the pixel.  A sector per sample written once
//     and read once, coalesced on the read side: 2 x 17 GB for the 5.3e8 samples of BASELINE configs[4], 1 % of the
//     frame's time in the finalize.
#ifndef RTM_GRID_KERNEL_H
#define RTM_GRID_KERNEL_H

#include "rtm_render_kernel.h"

namespace rtm {

#ifndef RTM_GRID_WPE
#define RTM_GRID_WPE 4
#endif
constexpr int kGridWavesPerSimd = RTM_GRID_WPE;  // launch bound of the grid kernel (profiles/r3/grid_variants.txt)
// a term's slot: three doubles padded to 32 bytes — one aligned 32-byte sector, written whole by two 16-byte stores (24-byte
// slots straddle sectors: L2 then FETCHES around every store, 55 GB per configs[4] frame, profiles/r3/README.md)
constexpr int kGridTermDoubles = 4;
typedef double grid_dbl2 __attribute__((ext_vector_type(2)));
__host__ __device__ inline size_t grid_tile_term_bytes(unsigned total_samples) { return (size_t)total_samples * 64 * kGridTermDoubles * sizeof(double); }
// ... and its "term stored" bits behind the launch's slots: 64 bits per sample of a tile
__host__ __device__ inline size_t grid_tile_bit_bytes(unsigned total_samples) { return (size_t)total_samples * 2 * sizeof(unsigned); }

// tile of block b: blocks are dealt round-robin to the 8 XCDs, each with its own L2 (RenderParams::xcd_on)
__device__ __forceinline__ unsigned grid_tile_of_block(const RenderParams& P, unsigned b) {
    if (P.xcd_on) {
        const unsigned xcd = b & 7u;
        b = xcd * P.xcd_q + (xcd < P.xcd_rem ? xcd : P.xcd_rem) + (b >> 3);
    }
    return b;
}

// blockIdx.x: index into this launch's tiles [tile_base, tile_base + gridDim.x); terms: P.contrib, gridDim.x tiles
//   COUNT   (RTM_MODE_COUNT_TESTS) the walks count their Intersect evaluations into P.counters[4]: the measurement
//           bench.py's rows of this kernel carry (sphere tests per cast); the timed steps run the plain instantiation
template <typename RecT, int LDS_D, bool COUNT = false>
__global__ __launch_bounds__(64, kGridWavesPerSimd) void render_grid_kernel(const RenderParams P, const unsigned tile_base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x;
    double* cam = reinterpret_cast<double*>(lds_raw);  // 9 camera doubles + pad
    double* trig = cam + 10;                           // the shading constants (sincos, near-unit Normalize)
    const bool unit_tab = P.unit_tab != 0u;            // wave-uniform: the launcher found the LDS for the near-unit Normalize table
    RecT* rec = reinterpret_cast<RecT*>(trig + (unit_tab ? kShadeConstCount : kTrigConstCount));
    unsigned* next_unit = reinterpret_cast<unsigned*>(rec + LDS_D * 64);
    unsigned char* queue = reinterpret_cast<unsigned char*>(next_unit + 4);  // the walks' candidate queue (16-byte aligned)
    fill_shade_consts(trig, lane, unit_tab);
    if (lane < 9) {
        const double v9[9] = {P.ax.x, P.ax.y, P.ax.z, P.by.x, P.by.y, P.by.z, P.cz.x, P.cz.y, P.cz.z};
        double pick = v9[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) pick = (lane == k) ? v9[k] : pick;
        cam[lane] = pick;
    }
    if (lane == 0) *next_unit = 0u;
    __syncthreads();

    SceneGlobal sc;
    sc.v = P.scene;
    const unsigned local_tile = grid_tile_of_block(P, blockIdx.x);
    const unsigned tile = tile_base + local_tile;
    const int tile_x8 = (int)(tile % (unsigned)P.tiles_x) * 8, tile_y = (int)(tile / (unsigned)P.tiles_x);
    const unsigned total_units = P.total_samples * 64u;
    double* const terms = reinterpret_cast<double*>(P.contrib) + (size_t)local_tile * P.total_samples * (64 * kGridTermDoubles);
    unsigned* const nz_bits = P.nz_bits + (size_t)local_tile * P.total_samples * 2;  // 64 bits per sample of the tile

    PathCounters pc = {0, 0, 0};
    RecordStack<RecT, LDS_D> stack{rec, lane, &P};
    auto push = [&](int d, int id) { stack.push(d, id); };
    auto pop = [&](int d) -> int { return stack.pop(d); };
    const bool pow2 = P.inv_s != 0.0;  // wave-uniform

    // this lane's unit and path
    unsigned unit = 0;
    D3 org = P.cam_org, dir = d3(0, 0, 1);
    int depth = 0;
    RngStream rng = rng_open(rng_pixel_key(P.seed_mult, 0u), 0u);
    // Take the next unit of the tile whose pixel is inside the frame and set its primary ray up (:224-232).
    auto take_unit = [&]() -> bool {
        int px, py;
        for (;;) {
            unit = atomicAdd(next_unit, 1u);
            if (unit >= total_units) return false;
            px = tile_x8 + (int)(unit & 7u);
            py = band_row(P, tile_y, (int)((unit >> 3) & 7u));
            if (px < P.W && py < P.row_end) break;
        }
        const unsigned s = unit >> 6;  // sample index ((sx-1)*SS + (sy-1))*S + s
        const int sub = (int)(s / (unsigned)P.S);
        org = P.cam_org;
        dir = primary_dir_lds(P, cam, px, py, sub / P.SS + 1, sub % P.SS + 1);
        depth = 0;
        rng = rng_open(rng_pixel_key(P.seed_mult, (uint32_t)py * (uint32_t)P.W + (uint32_t)px), s);
        return true;
    };
    bool busy = take_unit();
    GridWalk<MathFast, SceneGlobal, COUNT> walk;
    walk.attach_queue(queue, 64, lane);
    bool walking = false;
    [[maybe_unused]] unsigned long long n_tests = 0;
    while (__builtin_amdgcn_ballot_w64(busy) != 0) {  // wave-uniform
        if (busy && !walking) walking = walk.begin(sc, org, dir);
        for (;;) {
            const unsigned n_busy = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(busy));
            const unsigned n_walk = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(walking));
            if (n_walk * 8u <= n_busy * (8u - (unsigned)kGridShadeAt8)) break;
            if (walking) walking = walk.advance(sc, org, dir);
        }
        if (busy && !walking) {
            RTM_GRID_OCC(8);
            if constexpr (COUNT) n_tests += walk.tests;
            D3 term;
            bool cont = path_shade_spec(sc, walk.best, walk.dis, P.mode, P.max_bounces, org, dir, depth, rng, term, pc, push, ShadeLds(trig, unit_tab));
            if (cont && stack.overflow) {  // records exhausted: stop the path; the call fails loudly
                cont = false;
                term = d3(0, 0, 0);
                depth = 0;
            }
            if (!cont) {
                RTM_GRID_OCC(9);
                const bool deep = depth > LDS_D;
                const D3 L = (__builtin_amdgcn_ballot_w64(deep) == 0)
                                 ? path_fold_blocked(sc, term, depth, [&](int d) { return (int)rec[d * 64 + lane]; })
                                 : path_fold(sc, term, depth, pop);
                // :240 cal / SS / SS / S (power-of-two divisors as exact multiplications), :241 the clamp
                const D3 cal = pow2 ? ((L * P.inv_ss) * P.inv_ss) * P.inv_s : ((L / P.dSS) / P.dSS) / P.dS;
                const D3 add = clamp01_d3(cal);
                // x + (+-0) == x: a term of zeros is neither stored nor read back (P.nz_bits; see RenderParams)
                if (!(add.x == 0.0 && add.y == 0.0 && add.z == 0.0)) {
                    grid_dbl2* t = reinterpret_cast<grid_dbl2*>(terms + (size_t)unit * kGridTermDoubles);  // unit = sample * 64 + pixel
                    const grid_dbl2 lo = {add.x, add.y}, hi = {add.z, 0.0};
                    __builtin_nontemporal_store(lo, t);  // (system-scope write-through stores, sc0 sc1 nt, were 5 % slower)
                    __builtin_nontemporal_store(hi, t + 1);
                    atomicOr(nz_bits + (unit >> 5), 1u << (unit & 31u));
                }
                busy = take_unit();
            }
        }
    }
    if (P.counters) {
        wave_add_counter(P.counters + 0, pc.casts);
        wave_add_counter(P.counters + 1, pc.bounces);
        wave_add_counter(P.counters + 2, pc.draws);
        if (stack.overflow) atomicOr(P.counters + 3, 1ull);
        if constexpr (COUNT) {
            unsigned long long t = n_tests;
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0) atomicAdd(P.counters + 4, t);
        }
    }
}

// image[pixel] = ((0 + term[0]) + term[1]) + ... in sample order (src/Renderer.cpp:241-248).  One wave per tile, lane =
// pixel; a sample's 64 terms are one contiguous 2 048-byte row.
__global__ __launch_bounds__(64) void grid_finalize_kernel(const RenderParams P, const unsigned tile_base) {
    const int lane = threadIdx.x;
    const unsigned tile = tile_base + blockIdx.x;
#ifdef RTM_GRID_EXP_OCC
    if (blockIdx.x == 0 && lane == 0) {
        const char* names[10] = {"begin", "advance", "step", "test0", "test1", "test2", "test3", "sqrt", "shade", "path end"};
        for (int r = 0; r < 10; ++r) {
            // (one value per call: device printf mangled three 64-bit arguments in one)
            printf("[grid occ] %s:", names[r]);
            printf(" executions %llu", g_grid_occ[2 * r]);
            printf(" lanes %llu\n", g_grid_occ[2 * r + 1]);
            g_grid_occ[2 * r] = g_grid_occ[2 * r + 1] = 0ull;
        }
    }
#endif
    const int px = (int)(tile % (unsigned)P.tiles_x) * 8 + (lane & 7);
    const int py = band_row(P, (int)(tile / (unsigned)P.tiles_x), lane >> 3);
    const bool valid = px < P.W && py < P.row_end;
    if (!valid) return;  // (its units were never traced: the rows hold nothing for it)
    constexpr int kRow = 64 * kGridTermDoubles;  // doubles per sample row
    const double* t = reinterpret_cast<const double*>(P.contrib) + (size_t)blockIdx.x * P.total_samples * kRow + lane * kGridTermDoubles;
    // this pixel's column of the tile's "term stored" bits: word (sample, half) holds 32 pixels
    const unsigned* bits = P.nz_bits + (size_t)blockIdx.x * P.total_samples * 2 + (lane >> 5);
    const unsigned my_bit = 1u << (lane & 31);
    D3 acc = d3(0, 0, 0);
    unsigned s = 0;
    for (; s + 4 <= P.total_samples; s += 4) {  // four rows in flight; the additions stay in order
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = bits[(size_t)(s + k) * 2];
        if (__builtin_amdgcn_ballot_w64(((w[0] | w[1] | w[2] | w[3]) & my_bit) != 0u) == 0) continue;  // (no pixel of the tile has a term here)
        D3 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = d3(0, 0, 0);
            if (w[k] & my_bit) {
                const grid_dbl2* r = reinterpret_cast<const grid_dbl2*>(t + (size_t)(s + k) * kRow);
                const grid_dbl2 lo = __builtin_nontemporal_load(r), hi = __builtin_nontemporal_load(r + 1);
                v[k] = d3(lo.x, lo.y, hi.x);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (w[k] & my_bit) acc = acc + v[k];
    }
    for (; s < P.total_samples; ++s) {
        if (!(bits[(size_t)s * 2] & my_bit)) continue;
        const double* r = t + (size_t)s * kRow;
        acc = acc + d3(r[0], r[1], r[2]);
    }
    store_pixel(P, true, px, py, acc);
}

}  // namespace rtm
#endif
