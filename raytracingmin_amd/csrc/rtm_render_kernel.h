// rtm_render_kernel.h — the hot kernel of the RaytracingMin path on gfx950 (render_tiles_kernel) and the second half of
// its sample split (split_finalize_kernel).  Included by rtm_kernels.hip, which holds the host side (scene lifetime,
// per-stream contexts, variant selection, launches).
//
// Path (reference file:line): Renderer::Render's pixel/sample loop src/Renderer.cpp:215-250, png::PathTracing
// src/Renderer.cpp:57-117 (through rtm_path.h), the per-sample normalise + clamp + accumulate src/Renderer.cpp:240-248,
// the quantiser src/Renderer.cpp:251-254.
#pragma once
#include <type_traits>

#include "rtm_path.h"

namespace RTM_NS {

struct RenderParams {
    SceneView scene;
    int W, H, S, SS;
    int row_begin, row_end;
    int band_count, band_index;  // interleaved 8-row bands (rtm_options); 1, 0 = every band
    int tiles_x;
    int mode, max_bounces;
    unsigned total_samples;  // SS*SS*S per pixel
    float rate;              // 1.0 / (1 + SS) as float, src/Renderer.cpp:227
    double dSS, dS;          // divisors of src/Renderer.cpp:240
    double inv_ss, inv_s;    // exact reciprocals when SS and S are powers of two, else 0
    D3 cam_org, ax, by, cz;  // origin, camX*fovx, camY*fovy, camZ (src/Renderer.cpp:202-208)
    uint64_t seed_mult;
    double* __restrict__ out64;
    float* __restrict__ out32;
    uint8_t* __restrict__ out8;
    unsigned long long* __restrict__ counters;  // casts, bounces, draws, overflow flag
    // deep-path record pool (only when a path may run deeper than the LDS record stack)
    unsigned char* __restrict__ pool;       // pool_slots x kPoolLevels records
    unsigned* __restrict__ pool_next;       // bump allocator
    unsigned pool_slots;
    unsigned long long* __restrict__ stamps;  // STAMP builds: per block {nearest, shade, end, iterations}
    // sample split (SPLIT kernels): `split` waves per tile; wave 0 traces samples [0, split_head), wave f >= 1
    // samples [split_head + (f-1)*split_len, split_head + f*split_len)
    // Tiles [0, split_first) are NOT split (one wave traces all their samples and stores the pixel itself): a
    // long launch splits only its last tiles, whose short waves fill the SIMDs that the last whole tiles leave
    // idle one after another.  n_tiles counts the split tiles; their blocks follow the whole tiles' in the grid.
    unsigned n_tiles, split, split_head, split_len, split_first;
    double* __restrict__ partial;  // [tile][3][64]: accumulator of wave 0 after its split_head samples
    // per-sample terms of waves 1..: one block of split_len rows per (tile, small wave), a row = 64 terms in the order
    // the wave FOLDED them ([3][64] doubles, whole lines from one store instruction) + 64 tags (owner lane, sample)
    unsigned char* __restrict__ contrib;
    // in-wave sample stealing (STEAL kernels, whole tiles): per tile a block of kStealHdrBytes (terms stored, every
    // pixel's final own-sample end) + 3 x 64 doubles (the accumulators) + steal_rows rows of kStealRowBytes (64 terms in
    // fold order + 64 four-byte tags (pixel lane, sample)); steal_finalize_kernel turns the block into pixels
    unsigned char* __restrict__ steal_ws;
    unsigned steal_rows, steal_depth;  // capacity in rows per tile; how far below the last sample a pixel may be stolen from
    unsigned magic_S, magic_SS;        // floor(2^32 / d) + 1: x / d == umulhi(x, magic) for x, d < 2^16 (a stolen sample's sub-pixel)
    // grid kernel (rtm_grid_kernel.h): blocks are dealt round-robin to the 8 XCDs, each with its own L2; with xcd_on, block
    // b of a launch renders its tile (b % 8) * xcd_q + min(b % 8, xcd_rem) + b / 8, so that an XCD's blocks cover one
    // contiguous part of the frame and its L2 holds that part's cells (xcd_q = tiles / 8, xcd_rem = tiles % 8)
    unsigned xcd_on, xcd_q, xcd_rem;
    // the launch's LDS holds the near-unit Normalize table behind the sincos constants (rtm_device.h; set by the launcher
    // where the 256 bytes do not cost a wave per CU: unit_table_fits)
    unsigned unit_tab;
    // the tolerance translation unit (RTM_TOL): per tile of the launch 64 words — bit s of word p: "sub-pixel s of pixel p
    // has a primary ray whose nearest hit last-bit differences could change" (prim_prepass_kernel); null elsewhere
    const unsigned long long* __restrict__ prim_masks;
    // ... and, from the same pre-pass, every sub-pixel's primary direction ([tile][pixel][sub-pixel][component] doubles):
    // a lane that moves on to its next sub-pixel LOADS its direction instead of running the five correctly rounded
    // divisions of src/Renderer.cpp:228-232 alone in its wave (some lane of a wave does about every 4.4 trips: 4 % of the
    // frame).  Null: the directions are computed where they are needed, as before.
    const double* __restrict__ prim_dirs;
    // grid kernel: one bit per (tile, sample, pixel) of the launch — "this sample's term is not all zeros and was stored";
    // zeroed before the launch, set with atomicOr, read by grid_finalize_kernel.  A term of (+-0, +-0, +-0) changes no sum
    // it is added to (the accumulator starts at +0 and no term is negative zero's only partner), so it is neither stored nor
    // read: 94 % of the samples of BASELINE configs[4] (a path that reaches no emitter returns 0).
    unsigned* __restrict__ nz_bits;
};
constexpr size_t kTermRowBytes = 3 * 64 * sizeof(double) + 64 * sizeof(unsigned short);  // 1664 = 13 lines of 128 B
constexpr size_t kStealRowBytes = 3 * 64 * sizeof(double) + 64 * sizeof(unsigned);       // 1792 = 14 lines of 128 B
constexpr size_t kStealHdrBytes = 256;                                                    // unsigned n_terms; ...; u16 end[64] at +128
__host__ __device__ inline size_t steal_tile_bytes(unsigned rows) { return kStealHdrBytes + 3 * 64 * sizeof(double) + (size_t)rows * kStealRowBytes; }

__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

constexpr int kPoolLevels = 960;  // records per pool slot beyond the LDS levels

// Per-lane stack of hit records: the first LDS_D levels live in LDS ([depth][lane]); a lane whose
// path runs deeper takes, once, a slot of kPoolLevels records from a global pool.  Exhausting the
// pool or the slot raises the overflow flag and the call fails with RTM_ERR_UNSUPPORTED.
template <typename RecT, int LDS_D>
struct RecordStack {
    RecT* lds;  // this wave's [LDS_D][64]
    int lane;
    const RenderParams* P;
    int slot = -1;
    bool overflow = false;
    __device__ __forceinline__ RecT* slot_ptr() {
        if (slot < 0) {
            const unsigned got = P->pool ? atomicAdd(P->pool_next, 1u) : 0xFFFFFFFFu;
            if (got >= P->pool_slots) {
                overflow = true;
                return nullptr;
            }
            slot = (int)got;
        }
        return reinterpret_cast<RecT*>(P->pool) + (size_t)slot * kPoolLevels;
    }
    __device__ __forceinline__ void push(int d, int id) {
        if (d < LDS_D) {
            lds[d * 64 + lane] = (RecT)id;
        } else if (d - LDS_D < kPoolLevels) {
            if (RecT* p = slot_ptr()) p[d - LDS_D] = (RecT)id;
        } else {
            overflow = true;
        }
    }
    __device__ __forceinline__ int pop(int d) {
        if (d < LDS_D) return (int)lds[d * 64 + lane];
        if (slot < 0 || d - LDS_D >= kPoolLevels) return 0;  // only after an overflow: image is discarded
        return (int)(reinterpret_cast<const RecT*>(P->pool) + (size_t)slot * kPoolLevels)[d - LDS_D];
    }
    static constexpr int kCapacity = LDS_D + kPoolLevels;
};

// LDS copy of the scene tables: n geometry rows (4 doubles) + n+1 material rows (8 doubles, the last
// one is the identity row) + n normal-length rows (3 doubles: |hit - centre| as Magnitude returns it, its refined
// reciprocal, and the float r*r in the low word of the third)
// Does a launch whose workgroups (one wave each) take `lds` bytes keep its waves per CU with the near-unit Normalize table's
// 256 bytes on top?  160 KB of LDS per CU, at most 16 waves (4 per SIMD at the kernels' 128 registers).
inline bool unit_table_fits(size_t lds) {
    auto waves = [](size_t b) { const size_t w = b ? (size_t)163840 / b : 16; return w > 16 ? (size_t)16 : w; };
    return waves(lds) == waves(lds + (size_t)(kShadeConstCount - kTrigConstCount) * sizeof(double));
}
__host__ __device__ inline size_t lds_table_bytes(int n) { return ((size_t)n * 7 + ((size_t)n + 1) * 8) * sizeof(double); }

// src/Renderer.cpp:227-232; sx, sy in 1..SS
__device__ __forceinline__ D3 primary_dir(const RenderParams& P, int x, int y, int sx, int sy) {
    const double px = 2.0 * ((double)x + (double)(P.rate * (float)sx)) / P.W - 1.0;
    const double py = 2.0 * ((double)y + (double)(P.rate * (float)sy)) / P.H - 1.0;
    return normalize((P.ax * px + P.by * py) + P.cz);
}

// The same, with the nine camera-basis doubles read from an LDS block (cam[0..8] = ax, by, cz).  The
// block is read once per sub-pixel; keeping those doubles in registers across the render loop made
// hipcc spill them to scratch per lane (200 MB of stray HBM writes per 1080p frame).
__device__ __forceinline__ D3 primary_dir_lds(const RenderParams& P, const double* cam, int x, int y, int sx,
                                              int sy) {
#if RTM_TOL
    // the tolerance translation unit: the primary ray stays the reference's bit for bit (separately rounded multiplies and
    // adds, true divisions) — once per S samples, and what rtm_path.h: nearest_hit_exactfp settles exact ties on
#pragma clang fp contract(off)
    const double px = 2.0 * ((double)x + (double)(P.rate * (float)sx)) / P.W - 1.0;
    const double py = 2.0 * ((double)y + (double)(P.rate * (float)sy)) / P.H - 1.0;
    const double vx = (cam[0] * px + cam[3] * py) + cam[6];
    const double vy = (cam[1] * px + cam[4] * py) + cam[7];
    const double vz = (cam[2] * px + cam[5] * py) + cam[8];
    const float len2 = (float)(vx * vx + vy * vy + vz * vz);  // src/Ray.h:67-69
    const double m = (double)__builtin_sqrtf(len2);
    return D3{vx / m, vy / m, vz / m};                          // src/Ray.h:70-72
#else
    int w = P.W, h = P.H;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(w), "+s"(h));  // (opaque: their conversions to double are not worth a register pair each across the render loop)
#endif
    const double px = 2.0 * ((double)x + (double)(P.rate * (float)sx)) / w - 1.0;
    const double py = 2.0 * ((double)y + (double)(P.rate * (float)sy)) / h - 1.0;
    const D3 ax = d3(cam[0], cam[1], cam[2]), by = d3(cam[3], cam[4], cam[5]), cz = d3(cam[6], cam[7], cam[8]);
    return normalize((ax * px + by * py) + cz);
#endif
}

__device__ __forceinline__ void wave_add_counter(unsigned long long* dst, unsigned v) {
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(dst, s);
}

// src/Renderer.cpp:246-254: image[] += acc; optional float3 / 8-bit views of the same pixel
// Image row of row `sub` (0..7) of this call's band `local_band`: the call renders bands
// band_index, band_index + band_count, ... of [row_begin, row_end) and stores them back to back.
__device__ __forceinline__ int band_row(const RenderParams& P, int local_band, int sub) {
    return P.row_begin + (local_band * P.band_count + P.band_index) * 8 + sub;
}

__device__ __forceinline__ void store_pixel(const RenderParams& P, bool valid, int x, int y, D3 acc) {
    if (!valid) return;
    const int rel = y - P.row_begin;
    const int out_row = (P.band_count > 1) ? ((rel >> 3) / P.band_count) * 8 + (rel & 7) : rel;
    const size_t o = ((size_t)out_row * P.W + x) * 3;
    const double r = 0.0 + acc.x, g = 0.0 + acc.y, b = 0.0 + acc.z;  // :246-248
    if (P.out64) {
        P.out64[o] = r;
        P.out64[o + 1] = g;
        P.out64[o + 2] = b;
    }
    if (P.out32) {
        P.out32[o] = (float)r;
        P.out32[o + 1] = (float)g;
        P.out32[o + 2] = (float)b;
    }
    if (P.out8) {  // :253: (unsigned char)255 * std::min(v, 1.0), truncation
        const double q[3] = {r, g, b};
        for (int c = 0; c < 3; ++c) {
            const double v = 255 * ((1.0 < q[c]) ? 1.0 : q[c]);
            P.out8[o + c] = (v >= 0.0 && v < 256.0) ? (uint8_t)v : (uint8_t)0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The render kernel.
//   M       Math policy (rtm_path.h)
//   LDS_TAB scene tables (centres + materials) copied to LDS for the per-lane look-ups
//   UNROLL  spheres whose geometry is fetched together (wave-uniform loads)
//   RecT    hit-record type (u8 when n_spheres <= 256), LDS_D: record depth staged in LDS per lane
// Dynamic LDS layout (16-byte aligned base): [geom n*4 doubles][mat (n+1)*8][norm n*2] (LDS_TAB), [camera 10]
// [sincos constants 16][accumulator + primary direction 6 x 64] (PARK), then the record stack
// [LDS_D][64] or the fold queue (DEFER).
//   PARK    the pixel accumulator and the cached primary direction live in LDS ([component][lane]),
//           not in VGPRs: they are touched once per sample, and the 12 registers they would pin
//           are what the unrolled sphere chunk needs to stay under 128 VGPRs without scratch spills
//   STAMP   diagnostic build: s_memtime around the three segments of an iteration (never timed itself)
//   PACK8   max_bounces <= 8 and n < 256: hit records packed in a 64-bit register, no LDS stack
//   SPLIT   P.split waves per tile, each tracing a contiguous range of the pixel's samples.  Wave 0
//           accumulates as usual and leaves its accumulator in P.partial; the others store every
//           sample's term (src/Renderer.cpp:240-242, after the clamp) to P.contrib, and
//           split_finalize_kernel adds them IN THE REFERENCE'S ORDER, so the image does not change by
//           a bit.  Used when a strip has too few tiles to keep every SIMD busy to the end (few rows
//           per GPU, small images): the launch's tail is then the spread of per-tile cost.
//   DEFER   (with PACK8 + PARK) path ends are queued and folded 64 at a time.  The fold, the /SS/SS/S and
//           the clamp of a finished path are a pure function of its hit ids, but only ~22 % of the
//           lanes finish a path in a given iteration, so doing that work on the spot runs ~100
//           instructions with a fifth of the lanes (13 % of the frame, measured).  Instead the ending
//           lanes append (records, terminal id) to a wave-shared ring in LDS and start their next
//           sample at once; whenever 64 entries are waiting, all 64 lanes fold one entry each and leave
//           the term in LDS, and every owner then adds its terms to its accumulator in the order its
//           paths ended (a per-lane FIFO of ring positions keeps that order) — the accumulation order
//           of src/Renderer.cpp:241-242 is untouched.
//   PACKL   (with DEFER) packed records for paths of ANY depth (max_bounces < 0 or > 8): level d in byte
//           d & 7 of word d >> 3 — word 0 in a register, word 1 in LDS per lane, levels from 16 up (about
//           kd^16 of the paths) in the pooled global stack.  A queue entry carries both words and the pool
//           slot; a lane that queues a path deeper than 16 forces a pass, so its slot is free again
//           before its next path can reach level 16.
//   REUSE   (with DEFER + PACK8; variant 15, a SEPARATELY LABELLED row, never the default) the S samples of a
//           sub-pixel share one primary ray (no jitter, src/Renderer.cpp:224-232), so its nearest hit is computed
//           once per sub-pixel and reused: a lane whose path ended restarts at "primary hit known" in the SAME
//           trip and joins the shading block with the lanes that bounced.  Every trip is then one nearest-hit
//           search and one bounce for every lane (no idle lanes in either block) and a sample costs C - 1 trips
//           instead of C.  Same image, same counters (a reused primary hit still counts as the cast the reference
//           performs); different work per sample than the reference, hence the label.
constexpr int kFoldRing = 128;  // entries of the ring: < 64 waiting + <= 64 appended per iteration
constexpr size_t kFoldQueueBytes = kFoldRing * 16 + 3 * 64 * 8 + 64 * 8 + 64 * 4;  // the FIFO form: ring, terms, FIFO, counts
constexpr size_t kFoldQueueBytesL = kFoldQueueBytes + kFoldRing * 8 + 64 * 8;  // + word 1 of the entries and of the lanes
#ifndef RTM_OPT_SCATTER
#define RTM_OPT_SCATTER 1  // the folding lane adds its term to the entry's pixel (render_tiles_kernel: kScatter; A/B switch)
#endif
// kScatter kernels (every deferred-fold kernel but the primary-hit-reuse row): no staged terms, no FIFO — ring, the pixels'
// counts of added entries (a small wave's tags in their place), the lanes' counts: 1 792 bytes less per wave, which is what lets
// the any-depth kernels keep the near-unit Normalize table at 16 waves per CU
constexpr size_t kFoldQueueBytesS = RTM_OPT_SCATTER ? kFoldRing * 16 + 64 * 4 + 64 * 4 : kFoldQueueBytes;
constexpr size_t kFoldQueueBytesLS = RTM_OPT_SCATTER ? kFoldQueueBytesS + kFoldRing * 8 + 64 * 8 : kFoldQueueBytesL;
// (SPLIT: a small wave's ring entries carry a 2-byte tag (owner lane, sample).  The tags live in the per-lane FIFO array,
// which a small wave does not use: 256 of its 512 bytes.  A separate array put the any-depth kernel at 10 328 bytes of LDS per
// wave — 15 instead of 16 waves per CU, 5 % on every unlimited-depth frame, profiles/r3/ab_r2_vs_r3.txt.)
constexpr size_t kFoldTagBytes = 0;

//   STEAL   (with DEFER + PACK8) in-wave sample stealing for WHOLE tiles.  All 64 lanes of a wave trace the same number of
//           samples but not the same number of casts, so a wave runs until its slowest lane is done: 4.4 % of the lane-trips
//           of the headline frame are spent by lanes that have finished (8.9 % at 256 spp, 15 % at 64 spp:
//           profiles/r3/ragged_end.txt).  When the first lane runs out of samples the wave leaves the main loop for a
//           TAIL loop in which a lane without own samples takes the LAST unstarted sample of the pixel that has most left
//           (wave-uniform choice, no atomics: one victim per trip), traces it — the RNG is keyed by (pixel, sample) — and
//           leaves its clamped term, tagged (pixel, sample), in the tile's row buffer in HBM; a pixel's own lane
//           accumulates samples [0, end) in order as before, and steal_finalize_kernel adds the stolen terms [end, total)
//           in order on top.  Only the ORDER of the additions is observable, and it is the reference's.
//   PLANES  (with LDS_TAB) the scene holds png::PlaneObject entries: SceneLdsObjects / object_chunk / MathSpecZ
//           (rtm_path.h) — a plane's test in its index slot of the chunk, its normal from the LDS table
template <class M, bool LDS_TAB, int UNROLL, typename RecT, int LDS_D, int WPE = 1, bool PARK = false,
          bool STAMP = false, bool PACK8 = false, bool SPLIT = false, bool DEFER = false, bool PACKL = false,
          bool REUSE = false, bool PLANES = false, bool STEAL = false>
__global__ __launch_bounds__(64, WPE) void render_tiles_kernel(const RenderParams P) {
    static_assert(!STEAL || (DEFER && PACK8 && !REUSE && !PACKL), "sample stealing rides on the deferred fold with packed records");
    static_assert(!PLANES || (LDS_TAB && !REUSE), "plane scenes ride on the LDS tables");
    static_assert(!SPLIT || DEFER, "the sample split's small waves store their terms from the fold queue");
    static_assert(!DEFER || ((PACK8 || PACKL) && PARK && !STAMP), "the fold queue rides on packed records and the LDS accumulator");
    static_assert(!REUSE || (DEFER && PACK8 && !SPLIT), "primary-hit reuse rides on the deferred fold with packed records");
    static_assert(!PACKL || (DEFER && !PACK8 && LDS_D == 0 && sizeof(RecT) == 1), "PACKL: deferred fold, byte records, pooled stack only");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x;
    // The exact-n instantiations (UNROLL in -101 .. -107, or an axis signature: UNROLL <= -1000) are launched for scenes of
    // exactly that many spheres, so the count COULD be a compile-time constant there (the LDS tables' offsets, the identity
    // row's index and the empty record word as immediates).  Measured and not kept (RTM_OPT_CTN, profiles/r4/ctn_ab.txt): the
    // tolerance row does not move, the exact kernel loses 3 %.
    const int scene_n = (RTM_OPT_CTN && UNROLL <= -1000) ? ((-UNROLL - 1000) & 7)
                        : (RTM_OPT_CTN && UNROLL <= -101 && UNROLL >= -107) ? (-UNROLL - 100) : P.scene.n;
    double* lgeom = reinterpret_cast<double*>(lds_raw);
    double* lmat = lgeom + (LDS_TAB ? scene_n * 4 : 0);
    double* lnrm = lmat + (LDS_TAB ? (scene_n + 1) * 8 : 0);
    double* cam = reinterpret_cast<double*>(lds_raw + (LDS_TAB ? lds_table_bytes(scene_n) : 0));
    double* trig = cam + 10;                 // 9 camera doubles + pad
    // the shading constants: sincos and — where the launcher found the LDS for it — the near-unit Normalize table
    // wave-uniform; the axis-signature instantiations are launched only when the table is there (the launchers check), so it is
    // a compile-time constant for them
#ifndef RTM_OPT_CTTAB
#define RTM_OPT_CTTAB 1  // (A/B switch)
#endif
    const bool unit_tab = (RTM_OPT_CTTAB && UNROLL <= -1000) ? true : (P.unit_tab != 0u);
    double* park = trig + (unit_tab ? kShadeConstCount : kTrigConstCount);
    const ShadeLds shade_lds(trig, unit_tab);
    RecT* rec = reinterpret_cast<RecT*>(park + (PARK ? 6 * 64 : 0));
    // DEFER: [ring 128 x uint4][terms 3 x 64 doubles][per-lane FIFO of ring positions][per-lane count]; kScatter (below):
    // [ring][per pixel: entries added so far, 64 words — a small wave's 128 two-byte tags in their place][per-lane count]
    constexpr bool kScatter = (RTM_OPT_SCATTER != 0) && DEFER && (PACK8 || PACKL) && !REUSE;
    uint4* fq_in = reinterpret_cast<uint4*>(park + 6 * 64);
    double* fq_out = reinterpret_cast<double*>(fq_in + kFoldRing);  // (the FIFO form only)
    unsigned long long* fq_fifo = kScatter ? reinterpret_cast<unsigned long long*>(fq_in + kFoldRing)
                                           : reinterpret_cast<unsigned long long*>(fq_out + 3 * 64);
    unsigned* fq_pend = kScatter ? reinterpret_cast<unsigned*>(fq_fifo) + 64 : reinterpret_cast<unsigned*>(fq_fifo + 64);
    unsigned long long* fq_in1 = reinterpret_cast<unsigned long long*>(fq_pend + 64);  // PACKL: word 1 of the entries
    unsigned long long* rec_w1 = fq_in1 + kFoldRing;                                   // PACKL: word 1 of the lanes
    // SPLIT, small waves only: tag of every ring entry, in the FIFO array they have no other use for
    unsigned short* fq_tag = reinterpret_cast<unsigned short*>(fq_fifo);
    static_assert(kFoldRing * sizeof(unsigned short) <= 64 * sizeof(unsigned), "the tags fit the FIFO array, and the counts' that takes its place");
    // Who adds a folded term to its pixel (fold_pass).  Round 1's form: the OWNER finds its entries through a per-lane FIFO of
    // ring positions and reads the terms back from LDS — as many rounds as the busiest pixel has entries in the pass, each with
    // the FIFO's decoding (23 vector instructions and 10 LDS operations a round, ~4.5 rounds a pass).  kScatter (round 4, the
    // packed-record kernels): the entry carries (owner lane, the owner's running count of path ends) and the FOLDING lane adds
    // its term to the owner's accumulator itself, in the round that is the entry's turn — rank = count - the owner's count of
    // entries already added (fq_done) — straight from its registers: a compare, three LDS read-add-writes and the count per
    // round, no FIFO, no term staging, and no forced passes (a FIFO held 8 positions).  The order of a pixel's additions is the
    // order of its path ends either way.
    // (PACKL, the any-depth kernels: owner and count ride in the entry's third word above term id and depth, and the lane's
    // "may a deep entry of mine still be waiting" is the count it had after its last deep path against fq_done — fq_pend bits
    // 8..15 and 16)
    unsigned* fq_done = reinterpret_cast<unsigned*>(fq_fifo);  // kScatter: per lane, entries of its pixel added so far (the FIFO's place)
    if constexpr (DEFER) fq_pend[lane] = 0u;  // entries waiting (FIFO form) / path ends so far (kScatter)
    if constexpr (kScatter) fq_done[lane] = 0u;
    if constexpr (PACKL) rec_w1[lane] = packed8_empty(scene_n);
    fill_shade_consts(trig, lane, unit_tab);
    if (lane < 9) {
        const double v9[9] = {P.ax.x, P.ax.y, P.ax.z, P.by.x, P.by.y, P.by.z, P.cz.x, P.cz.y, P.cz.z};
        double pick = v9[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) pick = (lane == k) ? v9[k] : pick;
        cam[lane] = pick;
    }
    {  // the spare slot: this tile's first entry of the pre-pass's direction table (primary_of below)
        const unsigned tile_of_block = (SPLIT && blockIdx.x >= P.split_first) ? P.split_first + (blockIdx.x - P.split_first) % P.n_tiles
                                                                               : blockIdx.x;
        const double first = __longlong_as_double((long long)((unsigned long long)tile_of_block * (unsigned long long)(unsigned)(P.SS * P.SS)));
        if (lane == 9) cam[9] = first;
    }
    __syncthreads();
    if constexpr (LDS_TAB) {
        const double* gsrc = reinterpret_cast<const double*>(P.scene.geom);
        for (int i = lane; i < scene_n * 4; i += 64) lgeom[i] = gsrc[i];
        for (int i = lane; i < (scene_n + 1) * 8; i += 64) lmat[i] = P.scene.mat[i];
        for (int i = lane; i < scene_n; i += 64) {
            if (PLANES && gsrc[i * 4 + 3] < 0.0) {  // a plane's row: its normal (SceneLdsObjects::plane_normal)
                lnrm[i * 3] = P.scene.plane[(size_t)i * 16 + 3];
                lnrm[i * 3 + 1] = P.scene.plane[(size_t)i * 16 + 4];
                lnrm[i * 3 + 2] = P.scene.plane[(size_t)i * 16 + 5];
                continue;
            }
            const double ms = (double)__builtin_sqrtf((float)gsrc[i * 4 + 3]);
            lnrm[i * 3] = ms;
            lnrm[i * 3 + 1] = refined_rcp_or_nan(ms);
            reinterpret_cast<float*>(lnrm + i * 3 + 2)[0] = (float)gsrc[i * 4 + 3];  // r*r as the float it is
        }
        __syncthreads();  // one wave per block: orders the LDS writes before the reads
    }
    using Scene = typename std::conditional<PLANES, SceneLdsObjects,
                                            typename std::conditional<LDS_TAB, SceneLds, SceneGlobal>::type>::type;
    Scene sc;
    sc.v = P.scene;
    sc.v.n = scene_n;
    // The axis-signature instantiations (UNROLL <= -1000) are launched for RTM_MODE_REPAIRED only: the mode is a compile-time
    // constant there (the shading block's "literal mode: the normal stays 0" arm and its wave-uniform test go away; a literal-mode
    // render of such a scene takes the plain exact-n kernel)
    const int mode = (RTM_OPT_CTMODE && UNROLL <= -1000) ? (int)RTM_MODE_REPAIRED : P.mode;
    if constexpr (LDS_TAB) {
        sc.lgeom = lgeom;
        sc.lmat = lmat;
        sc.lnrm = lnrm;
    }

    // Pixel coordinates are recomputed where they are needed (sub-pixel change, final store) from an
    // opaque copy of the lane id, so that nothing derived from them stays live across the render loop.
    auto pixel_xy = [&](int& px, int& py) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const unsigned tile = (SPLIT && blockIdx.x >= P.split_first)
                                  ? P.split_first + (blockIdx.x - P.split_first) % P.n_tiles : blockIdx.x;
        const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
        px = tx * 8 + (l & 7);
        py = band_row(P, ty, l >> 3);
    };
    int x, y;
    pixel_xy(x, y);
    const bool valid = (x < P.W) && (y < P.row_end);
    const uint32_t pixel = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;  // GLOBAL pixel index

    PathCounters pc = {0, 0, 0};
    RecordStack<RecT, LDS_D> stack{rec, lane, &P};
    // The tolerance translation unit (RTM_TOL, rtm_kernels_tol.hip; STEAL instantiations only): per pixel a bit per sub-pixel —
    // "the contracted search does not find the reference's hit for this sub-pixel's primary ray" (an exact tie in real
    // arithmetic: rtm_path.h, nearest_hit_exactfp) —, in LDS so that a lane that steals a sample of another pixel finds it.
    // Sub-pixels from 64 up have no bit and always take the exact loop.
#ifndef RTM_DIR_PIXEL_MAJOR
#define RTM_DIR_PIXEL_MAJOR 0  // layout of RenderParams::prim_dirs (A/B: profiles/r4/dir_table_ab.txt)
#endif
#ifndef RTM_DIR_NT
#define RTM_DIR_NT 0           // non-temporal reads of it
#endif
#ifndef RTM_TOL_PRIMFIX
#define RTM_TOL_PRIMFIX 1  // (A/B switch: 0 compiles the exact-tie handling out — NOT within tolerance on the Cornell diagonals)
#endif
    // (round 4, later: the any-depth kernels too — PACKL has no stealing, a lane only ever asks for its OWN pixel's word and
    // reads it from the pre-pass's table when it moves on to a sub-pixel: no LDS, no register pair across the loop)
    constexpr bool kPrimFix = (RTM_TOL != 0) && (STEAL || PACKL) && (RTM_TOL_PRIMFIX != 0);
    unsigned long long* prim_mask = reinterpret_cast<unsigned long long*>(fq_pend + 3 * 64);  // STEAL: behind its two arrays
    [[maybe_unused]] auto own_prim_mask = [&]() -> unsigned long long {
        if constexpr (STEAL) {
            return prim_mask[lane];
        } else {
            unsigned b = blockIdx.x;
            asm volatile("" : "+s"(b));  // (opaque, like split_wave's: nothing of this stays live across the render loop)
            const unsigned tile_id = (SPLIT && b >= P.split_first) ? P.split_first + (b - P.split_first) % P.n_tiles : b;
            return P.prim_masks[(size_t)tile_id * 64 + (unsigned)lane];
        }
    };
    [[maybe_unused]] bool prim_fix = false;  // this lane's current sample is of a flagged sub-pixel
    [[maybe_unused]] auto prim_flag_of = [&](const unsigned long long mask, const unsigned sub) {
        return sub >= 64u || ((mask >> sub) & 1ull) != 0ull;
    };
    if constexpr (PACKL) {
        if (P.pool) stack.slot = (int)((blockIdx.x * 64u + (unsigned)lane) * 2u);  // two pooled stacks per lane, no allocator
    }
    D3 acc = d3(0, 0, 0);

    // this wave's samples [n_first, n_end) of the pixel (wave-uniform)
    const bool whole = !SPLIT || blockIdx.x < P.split_first;  // this wave traces all samples of its tile
    // Which wave of a split tile this is, recomputed from an OPAQUE copy of the block index wherever it is needed (a
    // small wave's fold passes, the head wave's final store): kept live across the render loop, these scalars pushed
    // hipcc over its SGPR budget and it spilled loop-carried scalars to VGPR lanes (27 v_readlane / v_writelane more in
    // the kernel: +1 % on the headline frame, profiles/r3/ab_r2_vs_r3.txt).
    auto split_wave = [&](unsigned& f, unsigned& tile) {
        unsigned b = blockIdx.x;
        asm volatile("" : "+s"(b));
        f = (b - P.split_first) / P.n_tiles;
        tile = (b - P.split_first) % P.n_tiles;  // index into partial / contrib
    };
    unsigned n_first = 0u, n_end = P.total_samples;
    if (!whole) {
        unsigned f, tile;
        split_wave(f, tile);
        n_first = f != 0u ? P.split_head + (f - 1u) * P.split_len : 0u;
        n_end = f != 0u ? n_first + P.split_len : P.split_head;
    }
    // a SMALL wave of the sample split (wave-uniform; its first sample is never 0: wave 0 keeps at least one share):
    // it accumulates nothing — every term goes to P.contrib in the order the wave folds them, 64 to a row, and
    // split_finalize_kernel puts each pixel's terms back in sample order
    // The render loop is compiled TWICE into a SPLIT kernel (`trace` below, generic over a compile-time tag) and a wave
    // picks its copy once: a whole or head wave runs exactly the loop of the unsplit kernel, a small wave a loop without
    // the per-lane FIFO.  One copy with wave-uniform "am I small" branches cost every whole wave of a split launch 1 %
    // (profiles/r3/ab_r2_vs_r3.txt: 181.1 against round 2's 179.3 ms with 64 split tiles).
    unsigned terms_out = 0;  // terms this small wave has stored (wave-uniform)
    unsigned n = valid ? n_first : n_end;  // sample index ((sx-1)*SS + (sy-1))*S + s
    int left_in_sub = P.S - (int)(n_first % (unsigned)P.S);  // samples left before the sub-pixel changes
    const int sub_first = (int)(n_first / (unsigned)P.S);
    if constexpr (kPrimFix) {
        const unsigned tile_id = (SPLIT && blockIdx.x >= P.split_first) ? P.split_first + (blockIdx.x - P.split_first) % P.n_tiles
                                                                         : blockIdx.x;
        const unsigned long long mask = P.prim_masks[(size_t)tile_id * 64 + (unsigned)lane];  // prim_prepass_kernel, the same launch
        if constexpr (STEAL) prim_mask[lane] = mask;
        prim_fix = prim_flag_of(mask, (unsigned)sub_first);
    }
    // The primary direction of sub-pixel `sub` of the tile's pixel `pl` (its lane number; px, py: its coordinates): from the
    // pre-pass's table where there is one (wave-uniform choice), else computed here.
    constexpr bool kPrimDirs = DEFER && !REUSE;  // the deferred-fold kernels (both translation units)
    [[maybe_unused]] auto primary_of = [&](const unsigned pl, const int px, const int py, const unsigned sub) -> D3 {
        if constexpr (kPrimDirs) {
            if (P.prim_dirs != nullptr) {
                // (the tile's first entry, parked in the camera block's spare LDS slot by the prologue: an integer
                // division to recompute, and a register pair too many to keep across the loop)
                // (a lane that is past its range keeps tracing dummies and keeps asking: its "next sub-pixel" may be one
                // past the pixel's last — clamped, the table ends with the last tile's last sub-pixel)
                const unsigned all = (unsigned)(P.SS * P.SS);
                const unsigned long long first = (unsigned long long)__double_as_longlong(cam[9]);
                unsigned col = pl & 63u;
                asm volatile("" : "+v"(col));  // (opaque: hipcc otherwise hoists the lane's table address out of the render
                                               // loop into a register pair it then spills — 28 GB of scratch reloads per frame)
#if RTM_DIR_PIXEL_MAJOR
                // [tile][pixel][sub-pixel][component]: a lane's three doubles share a line
                const double* q = P.prim_dirs + ((first * 64u + (unsigned long long)col * all) + (sub < all ? sub : all - 1u)) * 3;
                constexpr int kStep = 1;
#else
                // [tile][sub-pixel][component][pixel]
                const double* q = P.prim_dirs + (first + (sub < all ? sub : all - 1u)) * 192 + col;
                constexpr int kStep = 64;
#endif
#if RTM_DIR_NT
                return D3{__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + kStep), __builtin_nontemporal_load(q + 2 * kStep)};
#else
                return D3{q[0], q[kStep], q[2 * kStep]};
#endif
            }
        }
        return primary_dir_lds(P, cam, px, py, (int)(sub / (unsigned)P.SS) + 1, (int)(sub % (unsigned)P.SS) + 1);
    };
    D3 pdir = kPrimDirs ? primary_of((unsigned)lane, x, y, (unsigned)sub_first)
                        : primary_dir_lds(P, cam, x, y, sub_first / P.SS + 1, sub_first % P.SS + 1);
    D3 org = P.cam_org, dir = pdir;
    int depth = 0;
    const RngPixelKey pkey = rng_pixel_key(P.seed_mult, pixel);
    RngStream rng = rng_open(pkey, n_first);
    if constexpr (PARK) {
        park[0 * 64 + lane] = 0.0;
        park[1 * 64 + lane] = 0.0;
        park[2 * 64 + lane] = 0.0;
        park[3 * 64 + lane] = pdir.x;
        park[4 * 64 + lane] = pdir.y;
        park[5 * 64 + lane] = pdir.z;
    }

    unsigned long long recq = packed8_empty(scene_n);  // PACK8 records, most recent bounce in the low byte
    auto push = [&](int d, int id) {
        if constexpr (PACK8) {
            recq = (recq << 8) | (unsigned long long)(unsigned)id;
        } else if constexpr (PACKL) {
            // the byte still holds the identity index: xor turns it into id
            const unsigned long long flip = (unsigned long long)((unsigned)id ^ (unsigned)scene_n) << (8 * (d & 7));
            if (d < 8)
                recq ^= flip;
            else if (d < 16)
                rec_w1[lane] ^= flip;
            else
                stack.push(d - 16, id);  // LDS_D == 0: straight to the pooled stack
        } else {
            stack.push(d, id);
        }
    };
    auto pop = [&](int d) -> int { return stack.pop(d); };
    const bool pow2 = P.inv_s != 0.0;  // wave-uniform

    // DEFER: ring indices (wave-uniform) and one pass of the queue
    unsigned fq_head = 0, fq_tail = 0, fq_count = 0;
    // what a lane of the STEAL tail loop folded in this pass, kept for the export behind the per-lane part
    D3 tail_add = d3(0, 0, 0);
    unsigned tail_tag = 0u;
    unsigned stolen_out = 0u;  // STEAL: terms this wave has exported (wave-uniform)
    // MODE 0: whole / head waves, 1: small waves of the sample split, 2: the STEAL tail loop
    auto fold_pass = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool SMALL = SPLIT && MODE == 1;
        constexpr bool TAIL = STEAL && MODE == 2;
        if constexpr (TAIL) tail_tag = 0u;
        [[maybe_unused]] D3 scatter_add = d3(0, 0, 0);
        [[maybe_unused]] unsigned scatter_w = 0x80000000u;  // (a lane without an entry: nobody's)
        const unsigned m = fq_count < 64u ? fq_count : 64u;
        if ((unsigned)lane < m) {
            const unsigned at = (fq_head + (unsigned)lane) & (kFoldRing - 1);
            const uint4 e = fq_in[at];
            const unsigned long long w0 = (unsigned long long)e.x | ((unsigned long long)e.y << 32);
            D3 L;
            if constexpr (PACKL)
                L = path_fold_packed16(sc, (int)(e.z & 0xFFu), (int)((e.z >> 8) & 0x3FFu), w0, fq_in1[at],
                                       P.pool + (size_t)e.w * kPoolLevels);
            else
                L = path_fold_packed8_all(sc, (int)e.z, w0);
            const D3 cal = pow2 ? ((L * P.inv_ss) * P.inv_ss) * P.inv_s : ((L / P.dSS) / P.dSS) / P.dS;  // :240
            const D3 add = clamp01_d3(cal);
            if constexpr (SMALL) {
                // Whole lines from one instruction: the 64 folding lanes write their terms back to back (three 512-byte
                // rows) with the entry's tag behind them.  Round 2 let every OWNER store its term into a [sample][lane]
                // cell: 32-byte stores scattered over the tile's buffer, each completed to the memory's 64-byte granule by
                // a read (PMC: 4.5 GB fetched + 1.6 GB written per headline launch for 1.2 GB of terms).
                unsigned sf, stile;
                split_wave(sf, stile);
                unsigned char* term_rows = P.contrib + ((size_t)stile * (P.split - 1u) + (sf - 1u)) * P.split_len * kTermRowBytes;
                const unsigned pos = terms_out + (unsigned)lane;
                unsigned char* row = term_rows + (size_t)(pos >> 6) * kTermRowBytes;
                double* v = reinterpret_cast<double*>(row) + (pos & 63u);
                // NON-TEMPORAL stores: a plain store of a whole 512-byte row still made L2 fetch the lines it was about to
                // overwrite (PMC, profiles/r3/term_store_modes.txt: 1.25 GB fetched in the render kernel against 0.48 with
                // `nt`; 16-byte-per-lane stores through LDS changed nothing)
                __builtin_nontemporal_store(add.x, v);
                __builtin_nontemporal_store(add.y, v + 64);
                __builtin_nontemporal_store(add.z, v + 128);
                const unsigned tag = fq_tag[at];
                __builtin_nontemporal_store((unsigned short)tag, reinterpret_cast<unsigned short*>(row + 3 * 64 * sizeof(double)) + (pos & 63u));
                if constexpr (PACKL) {
                    // this entry's levels from 16 up have just been read from its owner's pooled stack: one deep entry fewer
                    // of that lane is waiting (two folding lanes may serve the same owner in one pass: an LDS atomic)
                    if (((e.z >> 8) & 0x3FFu) > 16u) atomicSub(&fq_pend[tag & 63u], 0x100u);
                }
            } else if constexpr (kScatter) {
                scatter_add = add;
                // own sample: owner lane | count << 8; stolen (TAIL): 0x80000000 | pixel lane << 16 | sample
                scatter_w = PACKL ? (((e.z >> 18) & 63u) | ((e.z >> 24) << 8)) : e.w;
                if constexpr (TAIL) {
                    tail_add = add;
                    tail_tag = (e.w >> 31) != 0u ? e.w : 0u;
                }
            } else {
                if constexpr (TAIL) {
                    tail_add = add;
                    tail_tag = e.w;  // 0: the owner's own sample; else 0x80000000 | pixel lane << 16 | sample
                }
                if (!TAIL || e.w == 0u) {
                    fq_out[0 * 64 + lane] = add.x;
                    fq_out[1 * 64 + lane] = add.y;
                    fq_out[2 * 64 + lane] = add.z;
                }
            }
        }
        if constexpr (TAIL) {
            // stolen samples' terms leave the chip in fold order, densely packed into the tile's rows (non-temporal)
            const bool out = tail_tag != 0u;
            const unsigned long long m_out = __builtin_amdgcn_ballot_w64(out);
            if (m_out != 0ull) {
                if (out) {
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m_out >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_out, 0u));
                    const unsigned pos = stolen_out + rank;
                    unsigned char* row = P.steal_ws + (size_t)blockIdx.x * steal_tile_bytes(P.steal_rows) + kStealHdrBytes +
                                         3 * 64 * sizeof(double) + (size_t)(pos >> 6) * kStealRowBytes;
                    double* v = reinterpret_cast<double*>(row) + (pos & 63u);
                    __builtin_nontemporal_store(tail_add.x, v);
                    __builtin_nontemporal_store(tail_add.y, v + 64);
                    __builtin_nontemporal_store(tail_add.z, v + 128);
                    __builtin_nontemporal_store(tail_tag & 0x7FFFFFFFu, reinterpret_cast<unsigned*>(row + 3 * 64 * sizeof(double)) + (pos & 63u));
                }
                stolen_out += (unsigned)__builtin_popcountll(m_out);
            }
        }
        if constexpr (SMALL) {
            terms_out += m;
        } else if constexpr (kScatter) {
            // every folding lane adds its term to its entry's pixel when it is the entry's turn (:241-242 on the LDS accumulator)
            const bool own_entry = (scatter_w >> 31) == 0u;
            const unsigned owner = scatter_w & 63u, count = (scatter_w >> 8) & 0xFFu;
            const unsigned rank = own_entry ? ((count - fq_done[owner]) & 0xFFu) : 0xFFFFFFFFu;  // 0: the pixel's next term
            for (unsigned round = 0u;; ++round) {
                if (__builtin_amdgcn_ballot_w64(own_entry && rank >= round) == 0) break;
                if (own_entry && rank == round) {
                    park[0 * 64 + owner] += scatter_add.x;
                    park[1 * 64 + owner] += scatter_add.y;
                    park[2 * 64 + owner] += scatter_add.z;
                    fq_done[owner] = count + 1u;  // (rounds go in order: the last write is the pixel's last entry of the pass)
                }
            }
        } else {
            // every owner adds its terms of this pass, oldest first (:241-242 on the LDS accumulator)
            for (;;) {
                const unsigned pend = fq_pend[lane] & 0xFFu;  // (PACKL: bit 8 = "a deep entry of this lane may be waiting")
                const unsigned oldest = (unsigned)(fq_fifo[lane] >> (8u * ((pend ? pend : 1u) - 1u))) & 0xFFu;
                const unsigned rel = (oldest - fq_head) & (kFoldRing - 1);
                const bool mine = pend != 0u && rel < m;
                if (__builtin_amdgcn_ballot_w64(mine) == 0) break;
                if (mine) {
                    park[0 * 64 + lane] += fq_out[0 * 64 + rel];
                    park[1 * 64 + lane] += fq_out[1 * 64 + rel];
                    park[2 * 64 + lane] += fq_out[2 * 64 + rel];
                    // (the last of the lane's entries: the deep flag goes too; otherwise it stays)
                    fq_pend[lane] = pend == 1u ? 0u : (fq_pend[lane] - 1u);
                }
            }
        }
        fq_head = (fq_head + m) & (kFoldRing - 1);
        fq_count -= m;
        (void)mode_tag;
    };
    using ModeWhole = std::integral_constant<int, 0>;
    using ModeSmall = std::integral_constant<int, 1>;
    using ModeTail = std::integral_constant<int, 2>;

    (void)fold_pass;
    // wave-level counters of the DEFER loop (scalar registers): every live lane casts once per trip
    unsigned w_casts = 0, w_bounces = 0, w_draws = 0;
#ifdef RTM_EXP_TRIPS
    unsigned w_trips64 = 0;
#endif
    if constexpr (DEFER && REUSE) {
        // ---- primary-hit reuse (see REUSE above) ----
        // the primary hit of the lane's current sub-pixel: recomputed when the sub-pixel changes (once per S samples)
        double pdis;
        int pid = nearest_hit<M, UNROLL>(sc, P.cam_org, pdir, pdis);
        int id = pid;       // the hit the lane is about to classify ...
        double dis = pdis;  // ... for its ray (org, dir) at `depth`; every lane starts at its first sample's primary hit
        bool have_fresh_rays = false;
        for (;;) {
            const unsigned long long m_live0 = __builtin_amdgcn_ballot_w64(n < n_end);
            if (m_live0 == 0ull) break;
            if (have_fresh_rays) id = nearest_hit<M, UNROLL>(sc, org, dir, dis);  // wave-uniform: one search per lane and trip
            // ---- classify until every live lane holds a hit whose path continues (src/Renderer.cpp:74-78, :112, :116) ----
            bool settled = !(n < n_end);  // a lane past its range: nothing to classify; it shades a dummy below
            for (;;) {
                const bool pending = !settled;
                const unsigned long long m_pend = __builtin_amdgcn_ballot_w64(pending);
                if (m_pend == 0ull) break;
                bool ends = false, drew = false, fifo_full = false;
                if (pending) {
                    const bool capped = P.max_bounces >= 0 && depth >= P.max_bounces;
                    drew = id >= 0 && !capped;
                    bool rr_pass = false;
                    if (drew) rr_pass = rng_next_m(rng) <= sc.kd24(id);  // :78
                    ends = !rr_pass;
                    settled = rr_pass;
                    if (ends) {
                        // queue this path end: ring position = tail + rank among the lanes ending now
                        const unsigned long long ending = __builtin_amdgcn_ballot_w64(true);
                        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(ending >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((unsigned)ending, 0u));
                        const unsigned pos = (fq_tail + rank) & (kFoldRing - 1);
                        const unsigned term_id = (unsigned)(id < 0 ? scene_n : id);
                        fq_in[pos] = uint4{(unsigned)recq, (unsigned)(recq >> 32), term_id, 0u};
                        const unsigned pend = fq_pend[lane] + 1u;
                        fq_fifo[lane] = (fq_fifo[lane] << 8) | pos;
                        fq_pend[lane] = pend;
                        fifo_full = pend >= 8u;
                        // next sample of this pixel: its primary ray and — new — its primary hit
                        ++n;
                        if (--left_in_sub == 0) {
                            left_in_sub = P.S;
                            if (n < n_end) {
                                const int sub = (int)(n / (unsigned)P.S);
                                int px, py;
                                pixel_xy(px, py);
                                pdir = primary_dir_lds(P, cam, px, py, sub / P.SS + 1, sub % P.SS + 1);
                                park[3 * 64 + lane] = pdir.x;
                                park[4 * 64 + lane] = pdir.y;
                                park[5 * 64 + lane] = pdir.z;
                                pid = nearest_hit<M, UNROLL>(sc, P.cam_org, pdir, pdis);  // once per S samples
                            }
                        }
                        org = P.cam_org;
                        dir = d3(park[3 * 64 + lane], park[4 * 64 + lane], park[5 * 64 + lane]);
                        depth = 0;
                        recq = packed8_empty(scene_n);
                        rng = rng_open(pkey, n);
                        id = pid;
                        dis = pdis;
                        settled = !(n < n_end);  // out of samples: done (shades a dummy below)
                    }
                }
                // counters: every classified hit is one PathTracing invocation of the reference
                w_casts += (unsigned)__builtin_popcountll(m_pend);
                w_draws += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(drew));
                const unsigned added = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ends));
                if (added != 0u) {
                    fq_tail = (fq_tail + added) & (kFoldRing - 1);
                    fq_count += added;
                    const bool force = __builtin_amdgcn_ballot_w64(fifo_full) != 0;
                    while (fq_count >= 64u || (force && fq_count > 0u)) fold_pass(ModeWhole{});
                }
            }
            // ---- bounce: every lane holds a hit that passed the roulette (lanes out of samples: a dummy) ----
            const bool live = n < n_end;
            const unsigned n_live = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(live));
            if (n_live == 0u) break;
            const int sid = id < 0 ? 0 : id;  // dummies may hold a miss; the scene has at least one sphere here
            ShadeOut o;
            {
                MathSpec m;
                m.set_lds(shade_lds);
                path_bounce_core(m, sc, sid, dis, mode, org, dir, rng, o);
                if (__builtin_amdgcn_ballot_w64(m.bad) != 0) {
                    MathRefI r;
                    r.set_lds(shade_lds);
                    path_bounce_core(r, sc, sid, dis, mode, org, dir, rng, o);
                }
            }
            w_draws += 2u * n_live;
            w_bounces += n_live;
            push(depth, sid);
            depth++;
            org = o.org;
            dir = o.dir;
            rng.ctr = o.ctr;
            have_fresh_rays = true;
        }
        while (fq_count > 0u) fold_pass(ModeWhole{});
    }
    if constexpr (DEFER && !REUSE) {
      bool to_tail = false;  // STEAL: the main loop was left because a lane ran out of samples (wave-uniform)
      auto trace = [&](auto mode_tag) {
        constexpr bool SMALL = SPLIT && decltype(mode_tag)::value == 1;
        // Wave-uniform loop: fold_pass needs all 64 lanes whatever their own state, so a lane that has finished
        // its samples cannot leave.  It does not idle either: it keeps tracing (samples beyond its range, results
        // discarded — `live` gates the queue, the counters and nothing else), which costs nothing — the lanes are
        // there anyway — and keeps the whole body free of a per-lane "am I still in range" region, i.e. of the
        // exec-mask bookkeeping and register copies hipcc generates around one (-14 VALU per trip).
        for (;;) {
            // lane masks as scalars (SGPR pairs straight from the compares): every count below is a popcount
            const unsigned long long m_live = __builtin_amdgcn_ballot_w64(n < n_end);
            if (m_live == 0ull) break;
            if constexpr (STEAL && !SMALL) {
                // a whole tile's first lane has run out of samples: the rest of the tile is the tail loop's
                if (whole && m_live != ~0ull) {
                    to_tail = true;
                    return;
                }
            }
            const bool live = n < n_end;
            bool fifo_full = false;
            D3 term;
            int hit_id;
            const int depth_before = depth;
            PathCounters unused = {0, 0, 0};
            bool cont;
#if RTM_TOL
            if constexpr (kPrimFix) {
                double dis;
                hit_id = nearest_hit<M, UNROLL>(sc, org, dir, dis);
                const bool flagged = prim_fix && depth == 0;  // rare: a primary ray whose hit last-bit differences could change
                cont = path_shade_spec_fix(sc, hit_id, dis, mode, P.max_bounces, org, dir, depth, rng, term, unused, push, shade_lds,
                                           flagged, [&](int& id_fix, double& dis_fix) {
                                               double dis_ref;
                                               const int id_ref = nearest_hit_exactfp(sc, org, dir, dis_ref);
                                               id_fix = flagged ? id_ref : id_fix;
                                               dis_fix = flagged ? dis_ref : dis_fix;
                                           });
            } else
#endif
            cont = path_step<M, UNROLL>(sc, mode, P.max_bounces, org, dir, depth, rng, term, unused, push,
                                        shade_lds, &hit_id);
            if (PACKL && cont && stack.overflow) cont = false;  // records exhausted: the call fails loudly
            // counters (src/Renderer.cpp has none; rtm_stats): one cast per live lane, one draw for the RR test of a
            // hit below the depth cap, two more and a bounce when the path continues
            const unsigned long long m_cont = __builtin_amdgcn_ballot_w64(cont) & m_live;
            unsigned long long m_drew = __builtin_amdgcn_ballot_w64(hit_id >= 0) & m_live;
            if (P.max_bounces >= 0) m_drew &= ~__builtin_amdgcn_ballot_w64(depth_before >= P.max_bounces);
            const unsigned n_cont = (unsigned)__builtin_popcountll(m_cont);
            w_casts += (unsigned)__builtin_popcountll(m_live);
#ifdef RTM_EXP_TRIPS
            w_trips64 += 64u;
#endif
            w_draws += (unsigned)__builtin_popcountll(m_drew) + 2u * n_cont;
            w_bounces += n_cont;
            if (!cont) {
                if (live) {
                    // queue this path end: ring position = tail + rank among the lanes ending now
                    const unsigned long long ending = __builtin_amdgcn_ballot_w64(true);
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(ending >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((unsigned)ending, 0u));
                    const unsigned pos = (fq_tail + rank) & (kFoldRing - 1);
                    const unsigned term_id = (unsigned)(hit_id < 0 ? scene_n : hit_id);
                    if constexpr (PACKL && kScatter && !SMALL) {
                        // fq_pend: bits 0..7 the lane's path ends so far, 8..15 that count after its last DEEP path, 16: there was one
                        const unsigned word = fq_pend[lane];
                        const unsigned count = word & 0xFFu, next = (count + 1u) & 0xFFu;
                        fq_in[pos] = uint4{(unsigned)recq, (unsigned)(recq >> 32),
                                           term_id | ((unsigned)depth << 8) | ((unsigned)lane << 18) | (count << 24),
                                           (unsigned)(stack.slot < 0 ? 0 : stack.slot)};
                        fq_in1[pos] = rec_w1[lane];
                        if (depth > 16) {
                            // Levels from 16 up sit in the lane's pooled stack and are read when the entry is folded; the lane has
                            // TWO pooled stacks and changes over whenever it queues such a path.  Only if its previous deep entry
                            // may still be waiting — fq_done has not reached the count the lane had after it (mod 256: a count
                            // that ran far ahead reads as "waiting", which only costs a pass) — is a pass forced.
                            const unsigned behind = (((word >> 8) & 0xFFu) - fq_done[lane]) & 0xFFu;
                            fifo_full = (word & 0x10000u) != 0u && behind != 0u && behind <= 128u;
                            stack.slot ^= 1;
                            fq_pend[lane] = next | (next << 8) | 0x10000u;
                        } else {
                            fq_pend[lane] = next | (word & 0x1FF00u);
                        }
                    } else if constexpr (PACKL) {
                        fq_in[pos] = uint4{(unsigned)recq, (unsigned)(recq >> 32), term_id | ((unsigned)depth << 8),
                                           (unsigned)(stack.slot < 0 ? 0 : stack.slot)};
                        fq_in1[pos] = rec_w1[lane];
                    } else if constexpr (kScatter && !SMALL) {
                        // (fold_pass: the folding lane adds the term to this lane's pixel when the count says it is its turn)
                        const unsigned ends_so_far = fq_pend[lane];
                        fq_in[pos] = uint4{(unsigned)recq, (unsigned)(recq >> 32), term_id, (unsigned)lane | ((ends_so_far & 0xFFu) << 8)};
                        fq_pend[lane] = ends_so_far + 1u;
                    } else {
                        fq_in[pos] = uint4{(unsigned)recq, (unsigned)(recq >> 32), term_id, 0u};
                    }
                    if constexpr (SPLIT) {
                        if constexpr (SMALL) fq_tag[pos] = (unsigned short)((unsigned)lane | ((n - n_first) << 6));
                    }
                    if constexpr (!(kScatter && !SMALL)) {
                    // whole waves: count of the lane's waiting entries, and (PACKL) bit 8: a deep entry of the lane may be waiting.
                    // A small wave never picks its terms up again, so it keeps no count — only (PACKL) the NUMBER of the
                    // lane's deep entries still waiting, in bits 8 up, taken down by whoever folds one (fold_pass)
                    const unsigned pend_word = fq_pend[lane];
                    const unsigned pend = SMALL ? 0u : (pend_word & 0xFFu) + 1u;
                    if constexpr (!SMALL) fq_fifo[lane] = (fq_fifo[lane] << 8) | pos;
                    fifo_full = !SMALL && pend >= 8u;
                    const bool deep_now = PACKL && depth > 16;
                    if constexpr (SMALL) {
                        if constexpr (PACKL) fq_pend[lane] = pend_word + (deep_now ? 0x100u : 0u);
                    } else {
                        fq_pend[lane] = pend | (pend_word & 0x100u) | (deep_now ? 0x100u : 0u);
                    }
                    if constexpr (PACKL) {
                        // Levels from 16 up sit in the lane's pooled stack and are read when the entry is folded, so the
                        // lane's next path must not write there before.  The lane has TWO pooled stacks and changes over
                        // whenever it queues such a path; only if the other one may still be waiting in the queue (a second
                        // deep path before the lane's entries have all been folded: rare squared) is a pass forced.  Forcing
                        // one for every deep path (P(depth > 16) = 1.5 % of the paths: every fifth trip of a wave) ran the
                        // fold half empty that often: +4 % on the 512x512 unlimited-depth frame.
                        if (deep_now) {
                            fifo_full = fifo_full || (pend_word >> 8) != 0u;
                            stack.slot ^= 1;
                        }
                    }
                    }  // (the FIFO form, or a small wave)
                }
                if constexpr (PACKL) {
                    if (depth > 8) rec_w1[lane] = packed8_empty(scene_n);
                }
                // next sample of this pixel (src/Renderer.cpp:236-239); a lane past its range stays at n_end, so that
                // "n - pending" remains the sample index of its queued path ends (fold_pass, SPLIT)
                n += live ? 1u : 0u;
                if (--left_in_sub == 0) {
                    left_in_sub = P.S;
                    const int sub = (int)(n / (unsigned)P.S);
                    int px, py;
                    pixel_xy(px, py);
                    if constexpr (kPrimDirs) pdir = primary_of((unsigned)lane, px, py, (unsigned)sub);
                    else pdir = primary_dir_lds(P, cam, px, py, sub / P.SS + 1, sub % P.SS + 1);
                    park[3 * 64 + lane] = pdir.x;
                    park[4 * 64 + lane] = pdir.y;
                    park[5 * 64 + lane] = pdir.z;
                    if constexpr (kPrimFix) prim_fix = n < n_end && prim_flag_of(own_prim_mask(), (unsigned)sub);
                }
                if constexpr (kPrimFix) prim_fix = prim_fix && n < n_end;
                org = P.cam_org;
                dir = d3(park[3 * 64 + lane], park[4 * 64 + lane], park[5 * 64 + lane]);
                depth = 0;
                recq = packed8_empty(scene_n);
                rng = rng_open(pkey, n);
            }
            const unsigned added = (unsigned)__builtin_popcountll(m_live & ~m_cont);
            if (added != 0u) {
                fq_tail = (fq_tail + added) & (kFoldRing - 1);
                fq_count += added;
                const bool force = __builtin_amdgcn_ballot_w64(fifo_full) != 0;  // a lane's FIFO holds 8 positions
                while (fq_count >= 64u || (force && fq_count > 0u)) fold_pass(mode_tag);
            }
        }
        while (fq_count > 0u) fold_pass(mode_tag);
      };
      if constexpr (SPLIT) {
          if (n_first != 0u) trace(ModeSmall{});  // (a small wave's first sample is never 0: wave 0 keeps at least one share)
          else trace(ModeWhole{});
      } else {
          trace(ModeWhole{});
      }
      if constexpr (STEAL) {
       if (whole) {
        unsigned end_own = valid ? P.total_samples : 0u;  // one past the last sample this lane's pixel accumulated itself
        if (to_tail) {
          // ---- the tail loop of a whole tile (see STEAL above) ----
          unsigned* st_next = reinterpret_cast<unsigned*>(fq_pend + 64);  // per pixel: the next own sample its lane will START
          unsigned* st_end = st_next + 64;                               // per pixel: one past its last OWN sample
          const unsigned total = P.total_samples;
          bool busy = n < n_end;                                         // this lane has a path in flight ...
          if constexpr (kPrimFix) prim_fix = prim_fix && busy;
          unsigned cur = ((unsigned)lane << 16) | (busy ? n : 0u);       // ... of (pixel lane, sample)
          unsigned own_next = busy ? n + 1u : total;
          st_next[lane] = own_next;
          st_end[lane] = valid ? total : 0u;
          unsigned claimed = 0u;                                         // samples taken from other pixels so far (wave-uniform)
          const unsigned cap = P.steal_rows * 64u;
          const unsigned floor_s = total > P.steal_depth ? total - P.steal_depth : 0u;
          const unsigned tile_x8 = (blockIdx.x % (unsigned)P.tiles_x) * 8u, tile_y = blockIdx.x / (unsigned)P.tiles_x;
          for (;;) {
            const unsigned long long m_busy = __builtin_amdgcn_ballot_w64(busy);
            if (m_busy == 0ull) break;
            bool fifo_full = false;
            D3 term;
            int hit_id;
            const int depth_before = depth;
            PathCounters unused = {0, 0, 0};
            bool cont;
#if RTM_TOL
            if constexpr (kPrimFix) {
                double dis;
                hit_id = nearest_hit<M, UNROLL>(sc, org, dir, dis);
                const bool flagged = prim_fix && depth == 0;  // rare: a primary ray whose hit last-bit differences could change
                cont = path_shade_spec_fix(sc, hit_id, dis, mode, P.max_bounces, org, dir, depth, rng, term, unused, push, shade_lds,
                                           flagged, [&](int& id_fix, double& dis_fix) {
                                               double dis_ref;
                                               const int id_ref = nearest_hit_exactfp(sc, org, dir, dis_ref);
                                               id_fix = flagged ? id_ref : id_fix;
                                               dis_fix = flagged ? dis_ref : dis_fix;
                                           });
            } else
#endif
            cont = path_step<M, UNROLL>(sc, mode, P.max_bounces, org, dir, depth, rng, term, unused, push, shade_lds, &hit_id);
            const unsigned long long m_cont = __builtin_amdgcn_ballot_w64(cont) & m_busy;
            unsigned long long m_drew = __builtin_amdgcn_ballot_w64(hit_id >= 0) & m_busy;
            if (P.max_bounces >= 0) m_drew &= ~__builtin_amdgcn_ballot_w64(depth_before >= P.max_bounces);
            const unsigned n_cont = (unsigned)__builtin_popcountll(m_cont);
            w_casts += (unsigned)__builtin_popcountll(m_busy);
            w_draws += (unsigned)__builtin_popcountll(m_drew) + 2u * n_cont;
            w_bounces += n_cont;
#ifdef RTM_EXP_TRIPS
            w_trips64 += 64u;
#endif
            const bool ended = busy && !cont;
            const bool need = !busy || !cont;  // this lane wants a new sample
            if (ended) {
                const unsigned long long ending = __builtin_amdgcn_ballot_w64(true);
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(ending >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ending, 0u));
                const unsigned pos = (fq_tail + rank) & (kFoldRing - 1);
                const unsigned term_id = (unsigned)(hit_id < 0 ? scene_n : hit_id);
                const bool own = (cur >> 16) == (unsigned)lane;
                if constexpr (kScatter) {
                    const unsigned ends_so_far = fq_pend[lane];  // (of this lane's OWN pixel)
                    fq_in[pos] = uint4{(unsigned)recq, (unsigned)(recq >> 32), term_id,
                                       own ? ((unsigned)lane | ((ends_so_far & 0xFFu) << 8)) : (0x80000000u | cur)};
                    if (own) fq_pend[lane] = ends_so_far + 1u;
                } else {
                fq_in[pos] = uint4{(unsigned)recq, (unsigned)(recq >> 32), term_id, own ? 0u : (0x80000000u | cur)};
                if (own) {  // the owner picks its term up again, in the order its paths ended
                    const unsigned pend = (fq_pend[lane] & 0xFFu) + 1u;
                    fq_fifo[lane] = (fq_fifo[lane] << 8) | pos;
                    fq_pend[lane] = pend;
                    fifo_full = pend >= 8u;
                }
                }
            }
            const unsigned added = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ended));
            // (1) the lane's own next sample, if its pixel still has one that nobody took
            bool got = false;
            if (need && own_next < st_end[lane]) {
                cur = ((unsigned)lane << 16) | own_next;
                ++own_next;
                st_next[lane] = own_next;
                got = true;
                if (--left_in_sub == 0) {
                    left_in_sub = P.S;
                    const int sub = (int)((cur & 0xFFFFu) / (unsigned)P.S);
                    int px, py;
                    pixel_xy(px, py);
                    if constexpr (kPrimDirs) pdir = primary_of((unsigned)lane, px, py, (unsigned)sub);
                    else pdir = primary_dir_lds(P, cam, px, py, sub / P.SS + 1, sub % P.SS + 1);
                    park[3 * 64 + lane] = pdir.x;
                    park[4 * 64 + lane] = pdir.y;
                    park[5 * 64 + lane] = pdir.z;
                }
                dir = d3(park[3 * 64 + lane], park[4 * 64 + lane], park[5 * 64 + lane]);
                rng = rng_open(pkey, cur & 0xFFFFu);
                if constexpr (kPrimFix) {
                    const unsigned s_own = cur & 0xFFFFu;
                    prim_fix = prim_flag_of(prim_mask[lane], P.S == 1 ? s_own : __umulhi(s_own, P.magic_S));
                }
            }
            // (2) lanes left without one take the LAST unstarted samples of the pixel that has most to give: what it has
            // not started, but one, and not below the depth the finalize kernel's index covers
            const unsigned long long m_want = __builtin_amdgcn_ballot_w64(need && !got);
            if (m_want != 0ull && claimed < cap) {
                const unsigned e_l = st_end[lane], n_l = st_next[lane];
                unsigned give = e_l > n_l + 1u ? e_l - n_l - 1u : 0u;
                const unsigned above = e_l > floor_s ? e_l - floor_s : 0u;
                give = give < above ? give : above;
                give = give < 255u ? give : 255u;
                unsigned long long cand = ~0ull;  // bitwise search for the largest `give`, lowest lane on ties
#pragma unroll
                for (int b = 7; b >= 0; --b) {
                    const unsigned long long mb = __builtin_amdgcn_ballot_w64(((give >> b) & 1u) != 0u) & cand;
                    cand = mb != 0ull ? mb : cand;
                }
                const int v = (int)__builtin_ctzll(cand);
                const unsigned give_v = (unsigned)__builtin_amdgcn_readlane((int)give, v);
                unsigned take = (unsigned)__builtin_popcountll(m_want);
                take = take < give_v ? take : give_v;
                take = take < cap - claimed ? take : cap - claimed;
                if (take != 0u) {
                    const unsigned e_v = (unsigned)__builtin_amdgcn_readlane((int)e_l, v);
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m_want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_want, 0u));
                    if (need && !got && rank < take) {
                        const unsigned s_st = e_v - 1u - rank;
                        cur = ((unsigned)v << 16) | s_st;
                        got = true;
                        // pixel v of this tile, sample s_st: its primary ray and RNG stream from scratch (wave-uniform pixel)
                        const int px = (int)tile_x8 + (v & 7), py = band_row(P, (int)tile_y, v >> 3);
                        const RngPixelKey vkey = rng_pixel_key(P.seed_mult, (uint32_t)py * (uint32_t)P.W + (uint32_t)px);
                        // s_st / S and sub / SS by multiplication (exact for operands below 2^16; a divisor of 1 has no magic)
                        const unsigned sub = P.S == 1 ? s_st : __umulhi(s_st, P.magic_S);
                        const unsigned sxm1 = P.SS == 1 ? sub : __umulhi(sub, P.magic_SS);
                        if constexpr (kPrimDirs) dir = primary_of((unsigned)v, px, py, sub);
                        else dir = primary_dir_lds(P, cam, px, py, (int)sxm1 + 1, (int)(sub - sxm1 * (unsigned)P.SS) + 1);
                        rng = rng_open(vkey, s_st);
                        if constexpr (kPrimFix) prim_fix = prim_flag_of(prim_mask[v], sub);
                    }
                    if (lane == v) st_end[lane] = e_v - take;
                    claimed += take;
                }
            }
            if (need) {
                busy = got;
                org = P.cam_org;
                depth = 0;
                recq = packed8_empty(scene_n);
                if constexpr (kPrimFix) prim_fix = prim_fix && got;
            }
            if (added != 0u) {
                fq_tail = (fq_tail + added) & (kFoldRing - 1);
                fq_count += added;
                const bool force = __builtin_amdgcn_ballot_w64(fifo_full) != 0;
                while (fq_count >= 64u || (force && fq_count > 0u)) fold_pass(ModeTail{});
            }
          }
          while (fq_count > 0u) fold_pass(ModeTail{});
          end_own = st_end[lane];
        }
        // the tile's block: accumulators, every pixel's own-sample end, the number of stolen terms
        unsigned char* blk = P.steal_ws + (size_t)blockIdx.x * steal_tile_bytes(P.steal_rows);
        double* part = reinterpret_cast<double*>(blk + kStealHdrBytes) + lane;
        part[0] = park[0 * 64 + lane];
        part[64] = park[1 * 64 + lane];
        part[128] = park[2 * 64 + lane];
        reinterpret_cast<unsigned short*>(blk + 128)[lane] = (unsigned short)end_own;
        if (lane == 0) reinterpret_cast<unsigned*>(blk)[0] = stolen_out;
       }
      }
    }
    unsigned long long st_near = 0, st_shade = 0, st_end = 0, st_iters = 0;
    while (!DEFER && n < n_end) {
        D3 term;
        bool cont;
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
        if constexpr (STAMP) {
            ts0 = stamp_now();
            double dis;
            const int id = nearest_hit<M, UNROLL>(sc, org, dir, dis);
            ts1 = stamp_now();
            cont = path_shade_spec(sc, id, dis, mode, P.max_bounces, org, dir, depth, rng, term, pc, push, shade_lds);
            ts2 = stamp_now();
        } else {
            cont = path_step<M, UNROLL>(sc, mode, P.max_bounces, org, dir, depth, rng, term, pc, push, shade_lds);
        }
        if (cont && stack.overflow) {  // records exhausted: stop the path; the call fails loudly
            cont = false;
            term = d3(0, 0, 0);
            depth = 0;
        }
        if (!cont) {
            // all ending lanes within the LDS levels (always, when max_bounces < 16): blocked fold
            const bool deep = depth > LDS_D;
            D3 L;
            if constexpr (PACK8)
                L = path_fold_packed8(sc, term, depth, recq);
            else
                L = (__builtin_amdgcn_ballot_w64(deep) == 0)
                        ? path_fold_blocked(sc, term, depth, [&](int d) { return (int)rec[d * 64 + lane]; })
                        : path_fold(sc, term, depth, pop);
            // :240 cal / SS / SS / S.  x/2^k and x*2^-k are the same correctly rounded value, so
            // power-of-two divisors are applied as multiplications.
            const D3 cal = pow2 ? ((L * P.inv_ss) * P.inv_ss) * P.inv_s : ((L / P.dSS) / P.dSS) / P.dS;
            const D3 add = clamp01_d3(cal);
            if constexpr (PARK) {  // :241-242 on the LDS-resident accumulator
                park[0 * 64 + lane] += add.x;
                park[1 * 64 + lane] += add.y;
                park[2 * 64 + lane] += add.z;
            } else {
                acc = acc + add;
            }
            ++n;
            if (--left_in_sub == 0) {
                left_in_sub = P.S;
                if (n < n_end) {
                    const int sub = (int)(n / (unsigned)P.S);
                    int px, py;
                    pixel_xy(px, py);
                    pdir = primary_dir_lds(P, cam, px, py, sub / P.SS + 1, sub % P.SS + 1);
                    if constexpr (PARK) {
                        park[3 * 64 + lane] = pdir.x;
                        park[4 * 64 + lane] = pdir.y;
                        park[5 * 64 + lane] = pdir.z;
                    }
                }
            }
            org = P.cam_org;
            if constexpr (PARK)
                dir = d3(park[3 * 64 + lane], park[4 * 64 + lane], park[5 * 64 + lane]);
            else
                dir = pdir;
            depth = 0;
            if constexpr (PACK8) recq = packed8_empty(scene_n);
            rng = rng_open(pkey, n);
        }
        if constexpr (STAMP) {
            const unsigned long long ts3 = stamp_now();
            st_near += ts1 - ts0;
            st_shade += ts2 - ts1;
            st_end += ts3 - ts2;
            st_iters += 1;
        }
    }
    if constexpr (STAMP) {
        if (lane == 0 && P.stamps) {
            P.stamps[blockIdx.x * 4 + 0] = st_near;
            P.stamps[blockIdx.x * 4 + 1] = st_shade;
            P.stamps[blockIdx.x * 4 + 2] = st_end;
            P.stamps[blockIdx.x * 4 + 3] = st_iters;
        }
    }

    if constexpr (PARK) acc = d3(park[0 * 64 + lane], park[1 * 64 + lane], park[2 * 64 + lane]);
    if (STEAL && whole) {
        // steal_finalize_kernel stores the pixel (the tail loop has left the tile's block in P.steal_ws)
    } else if (whole) {
        int px, py;
        pixel_xy(px, py);
        store_pixel(P, (px < P.W) && (py < P.row_end), px, py, acc);
    } else if (n_first == 0u) {  // the head wave of a split tile
        unsigned hf, split_tile;
        split_wave(hf, split_tile);
        double* dst = P.partial + (size_t)split_tile * 192 + lane;
        dst[0] = acc.x;
        dst[64] = acc.y;
        dst[128] = acc.z;
    }
    if (P.counters) {
        if constexpr (DEFER) {
            if (lane == 0) {
                atomicAdd(P.counters + 0, (unsigned long long)w_casts);
                atomicAdd(P.counters + 1, (unsigned long long)w_bounces);
#ifdef RTM_EXP_TRIPS
                atomicAdd(P.counters + 2, (unsigned long long)w_trips64);  // EXPERIMENT: "draws" = 64 x trips of the wave
#else
                atomicAdd(P.counters + 2, (unsigned long long)w_draws);
#endif
            }
        } else {
            wave_add_counter(P.counters + 0, pc.casts);
            wave_add_counter(P.counters + 1, pc.bounces);
            wave_add_counter(P.counters + 2, pc.draws);
        }
        if (stack.overflow) atomicOr(P.counters + 3, 1ull);
    }
}

// Second half of a SPLIT render: image[pixel] = (((partial + term[split_head]) + term[split_head+1]) + ...), the
// accumulation order of src/Renderer.cpp:241-242.  One workgroup of four waves per split tile.  The small waves left
// their terms in FOLD order (render_tiles_kernel: rows of 64 terms + tags); for one small wave at a time the block
// scatters its rows into LDS by (sample, owner lane) — the tag — and waves 0..2 then add, each for ONE colour channel
// (the three sums are independent), their pixel's terms in sample order.  LDS: split_len x 64 x 3 doubles (96 KB at
// the 64 samples a small wave traces at most).
__global__ __launch_bounds__(256) void split_finalize_kernel(const RenderParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* slot = reinterpret_cast<double*>(lds_raw);  // [channel][sample][lane]
    const int tid = threadIdx.x, lane = tid & 63, chan = tid >> 6;
    const unsigned tile = P.split_first + blockIdx.x;  // blockIdx.x: index among the split tiles
    const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const int px = tx * 8 + (lane & 7), py = band_row(P, ty, lane >> 3);
    const bool valid = (px < P.W) && (py < P.row_end);
    // terms a small wave of this tile has stored: one per valid pixel and sample of its range
    const unsigned n_valid = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(valid));
    const unsigned n_terms = n_valid * P.split_len;
    const unsigned plane_sz = P.split_len * 64u;  // doubles per channel
    double acc = chan < 3 ? P.partial[(size_t)blockIdx.x * 192 + chan * 64 + lane] : 0.0;
    for (unsigned f = 1; f < P.split; ++f) {
        const unsigned char* rows = P.contrib + ((size_t)blockIdx.x * (P.split - 1u) + (f - 1u)) * P.split_len * kTermRowBytes;
        for (unsigned e = (unsigned)tid; e < n_terms; e += 256u) {
            const unsigned char* row = rows + (size_t)(e >> 6) * kTermRowBytes;
            const double* v = reinterpret_cast<const double*>(row) + (e & 63u);
            const unsigned tag = reinterpret_cast<const unsigned short*>(row + 3 * 64 * sizeof(double))[e & 63u];
            const unsigned at = (tag >> 6) * 64u + (tag & 63u);  // [sample][owner lane]
            slot[at] = v[0];
            slot[plane_sz + at] = v[64];
            slot[2 * plane_sz + at] = v[128];
        }
        __syncthreads();
        if (chan < 3) {
            const double* mine = slot + (size_t)chan * plane_sz + lane;
            for (unsigned m = 0; m < P.split_len; ++m) acc = acc + mine[m * 64u];  // (an invalid pixel adds stale LDS: never stored)
        }
        __syncthreads();
    }
    if (chan < 3) slot[chan * 64 + lane] = acc;
    __syncthreads();
    if (chan == 0) store_pixel(P, valid, px, py, d3(slot[lane], slot[64 + lane], slot[128 + lane]));
}

// Second half of a STEAL render (whole tiles): pixel = ((partial + term[end]) + term[end + 1]) + ... — the accumulation
// order of src/Renderer.cpp:241-242 — where `partial` holds the samples [0, end) the pixel's own lane traced and the terms
// of [end, total) were traced by other lanes of the tile and lie, in fold order and tagged (pixel lane, sample), in the
// tile's rows.  One wave per tile: an index of row positions by (sample, pixel) in LDS, then every lane walks its pixel's
// stolen samples in order.  LDS: steal_depth x 64 two-byte entries.
__global__ __launch_bounds__(64) void steal_finalize_kernel(const RenderParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned short* idx = reinterpret_cast<unsigned short*>(lds_raw);  // [sample - floor][pixel lane] -> position in the rows
    const int lane = threadIdx.x;
    const unsigned tile = blockIdx.x;
    const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const int px = tx * 8 + (lane & 7), py = band_row(P, ty, lane >> 3);
    const bool valid = (px < P.W) && (py < P.row_end);
    const unsigned char* blk = P.steal_ws + (size_t)tile * steal_tile_bytes(P.steal_rows);
    unsigned n_terms = reinterpret_cast<const unsigned*>(blk)[0];
    unsigned end_own = reinterpret_cast<const unsigned short*>(blk + 128)[lane];
    // (never out of range when the render kernel wrote the block; a stale or foreign block must not turn into a fault)
    n_terms = n_terms <= P.steal_rows * 64u ? n_terms : 0u;
    const unsigned floor_chk = P.total_samples > P.steal_depth ? P.total_samples - P.steal_depth : 0u;
    end_own = (n_terms != 0u && end_own >= floor_chk && end_own <= P.total_samples) ? end_own : P.total_samples;
    const double* part = reinterpret_cast<const double*>(blk + kStealHdrBytes) + lane;
    D3 acc = d3(part[0], part[64], part[128]);
    if (n_terms != 0u) {  // wave-uniform
        const unsigned total = P.total_samples;
        const unsigned floor_s = total > P.steal_depth ? total - P.steal_depth : 0u;
        const unsigned char* rows = blk + kStealHdrBytes + 3 * 64 * sizeof(double);
        for (unsigned e = (unsigned)lane; e < n_terms; e += 64u) {
            const unsigned tag = reinterpret_cast<const unsigned*>(rows + (size_t)(e >> 6) * kStealRowBytes + 3 * 64 * sizeof(double))[e & 63u];
            const unsigned ts = tag & 0xFFFFu, tv = (tag >> 16) & 63u;
            if (ts >= floor_s && ts < total) idx[(ts - floor_s) * 64u + tv] = (unsigned short)e;
        }
        __syncthreads();
        if (valid) {
            for (unsigned s = end_own; s < total; ++s) {
                unsigned e = idx[(s - floor_s) * 64u + (unsigned)lane];
                e = e < n_terms ? e : 0u;
                const double* v = reinterpret_cast<const double*>(rows + (size_t)(e >> 6) * kStealRowBytes) + (e & 63u);
                acc = acc + d3(v[0], v[64], v[128]);
            }
        }
    }
    store_pixel(P, valid, px, py, acc);
}

// The pre-pass of the deferred-fold kernels, one wave per tile of the launch, lane = pixel: every sub-pixel's primary direction
// (src/Renderer.cpp:227-232: shared by the sub-pixel's S samples) goes to RenderParams::prim_dirs, computed HERE by all 64
// lanes at once with the very function the render loop would call — so that a lane of the render kernel that moves on to its
// next sub-pixel loads three doubles instead of running five correctly rounded divisions alone in its wave (about every 4.4
// trips some lane of a wave does: 4 % of the frame; same bits by construction).  In the tolerance translation unit it also
// leaves, per pixel, a bit per sub-pixel (the first 64 of them): "this primary ray is one whose nearest hit last-bit
// differences could change" (rtm_path.h: primary_tie_risk; the shipped Cornell box has such rays along the image's
// diagonals, where the seam of two wall spheres projects exactly) — the render kernel settles those rays' primary hits with
// the reference's own arithmetic (nearest_hit_exactfp).
__global__ __launch_bounds__(64) void prim_prepass_kernel(const RenderParams P, unsigned long long* __restrict__ masks,
                                                          double* __restrict__ dirs) {
    const int lane = threadIdx.x;
    const unsigned tile = blockIdx.x;
    auto store_dir = [&](const unsigned sub, const D3 d) {  // RenderParams::prim_dirs: [tile][pixel][sub-pixel][component]
        const unsigned all_s = (unsigned)(P.SS * P.SS);
#if RTM_DIR_PIXEL_MAJOR
        double* q = dirs + (((size_t)tile * 64 + (unsigned)lane) * all_s + sub) * 3;
        constexpr int kStep = 1;
#else
        double* q = dirs + ((size_t)tile * all_s + sub) * 192 + (unsigned)lane;
        constexpr int kStep = 64;
#endif
        __builtin_nontemporal_store(d.x, q);
        __builtin_nontemporal_store(d.y, q + kStep);
        __builtin_nontemporal_store(d.z, q + 2 * kStep);
    };
    const int x = (int)(tile % (unsigned)P.tiles_x) * 8 + (lane & 7);
    const int y = band_row(P, (int)(tile / (unsigned)P.tiles_x), lane >> 3);
    const double cam[9] = {P.ax.x, P.ax.y, P.ax.z, P.by.x, P.by.y, P.by.z, P.cz.x, P.cz.y, P.cz.z};
    [[maybe_unused]] unsigned long long mask = 0ull;
    const unsigned all = (unsigned)(P.SS * P.SS);
    for (unsigned sub = 0; sub < all; ++sub) {
        if (dirs == nullptr && (masks == nullptr || sub >= 64u)) break;
        const D3 d = primary_dir_lds(P, cam, x, y, (int)(sub / (unsigned)P.SS) + 1, (int)(sub % (unsigned)P.SS) + 1);
#if RTM_TOL
        if (masks != nullptr && sub < 64u) {
            SceneGlobal sc;
            sc.v = P.scene;
            mask |= primary_tie_risk(sc, P.cam_org, d) ? (1ull << sub) : 0ull;
        }
#endif
        if (dirs != nullptr) store_dir(sub, d);
    }
#if RTM_TOL
    if (masks != nullptr) masks[(size_t)tile * 64 + (unsigned)lane] = mask;
#endif
}

}  // namespace RTM_NS
