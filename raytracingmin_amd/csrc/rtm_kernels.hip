// rtm_kernels.hip — the device half of the C ABI in include/rtm.h: ONE translation unit that pulls in the hand-written
// HIP kernels for gfx950 / CDNA4 and holds their host side.
//
//   rtm_path.h           Math and Scene policies, SphereObject::Intersect / PlaneObject::Intersect, the nearest-hit search,
//                        one PathTracing invocation, the back-to-front fold (src/Renderer.cpp:57-117, src/SettingData.cpp:197-249)
//   rtm_render_kernel.h  render_tiles_kernel — the hot kernel (Renderer::Render's pixel/sample loop, src/Renderer.cpp:215-250)
//                        — and split_finalize_kernel
//   rtm_wavefront.h      the large-scene pipeline (BASELINE configs[4])
//   rtm_fp32.h           the separately labelled single-precision row
//   rtm_seam_kernels.h   per-ray / per-call seams, probes, self-checks, the fp64 peak kernel
//   this file            scene lifetime (flattening, upload, deferred release), per-(device, stream) contexts and scratch,
//                        variant selection, the sample-split plan, launches, the blocking conveniences, the test hooks
// Shape of the render kernel: DESIGN.md §4.  No MFMA: there is no dense contraction on this path.  fp64 throughout,
// float islands kept.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <thread>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <tuple>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/rtm.h"
#include "rtm_device.h"
#include "rtm_internal.h"
#include "rtm_path.h"

#include "rtm_render_kernel.h"
#include "rtm_grid_kernel.h"
#include "rtm_wavefront.h"
#include "rtm_fp32.h"
#include "rtm_seam_kernels.h"
#include "rtm_surface.h"

namespace rtm {

// ================================================================================================
// Host side of the device path
// ================================================================================================
constexpr int kModeFlags = RTM_MODE_HOST_TRIG | RTM_MODE_COUNT_TESTS | RTM_MODE_SURFACE_SAMPLE;  // flags OR-ed into rtm_options.mode
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
const char* last_error() { return g_last_error.c_str(); }

#define RTM_HIP_CHECK(expr)                                                               \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            (void)hipGetLastError(); /* do not leave a sticky error for the host app */  \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(e__));           \
            return RTM_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

// src/Ray.h / src/Renderer.cpp:202-208 on the host, same operation order as the device code
namespace host {
struct H3 {
    double x, y, z;
};
static H3 sub(H3 a, H3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static H3 cross(H3 a, H3 b) {
    return {a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x};
}
static H3 normalize(H3 a) {
    const float len2 = (float)(a.x * a.x + a.y * a.y + a.z * a.z);
    const double m = (double)sqrtf(len2);
    return {a.x / m, a.y / m, a.z / m};
}
}  // namespace host

// Flatten rtm_sphere[] to the kernel layout.  kd and colorKD are hoisted: the same
// IEEE operations the reference redoes per bounce (src/SettingData.h:11-16).
static void flatten_scene(const rtm_sphere* sp, size_t n, std::vector<double>& geom,
                          std::vector<double>& mat, std::vector<double>* surf = nullptr) {
    if (surf) {  // png::SurfaeSample's extras: the raw colour and the radius as the float it is (rtm_surface.h)
        surf->resize((n ? n : 1) * 4);
        for (size_t i = 0; i < n; ++i) {
            for (int k = 0; k < 3; ++k) (*surf)[i * 4 + k] = sp[i].color[k];
            (*surf)[i * 4 + 3] = (double)sp[i].radius;
        }
    }
    geom.resize(n * 4);
    mat.assign((n + 1) * 8, 0.0);
    mat[n * 8 + 0] = mat[n * 8 + 1] = mat[n * 8 + 2] = 1.0;  // identity row for path_fold_blocked
    for (size_t i = 0; i < n; ++i) {
        const float r2 = sp[i].radius * sp[i].radius;  // float product, src/SettingData.cpp:200
        geom[i * 4 + 0] = sp[i].center[0];
        geom[i * 4 + 1] = sp[i].center[1];
        geom[i * 4 + 2] = sp[i].center[2];
        geom[i * 4 + 3] = (double)r2;
        double m = sp[i].color[0] < sp[i].color[1] ? sp[i].color[1] : sp[i].color[0];
        m = m < sp[i].color[2] ? sp[i].color[2] : m;
        const double kd = (double)(float)m;  // kd() returns float
        mat[i * 8 + 0] = sp[i].color[0] / kd;
        mat[i * 8 + 1] = sp[i].color[1] / kd;
        mat[i * 8 + 2] = sp[i].color[2] / kd;
        mat[i * 8 + 3] = sp[i].emission[0];
        mat[i * 8 + 4] = sp[i].emission[1];
        mat[i * 8 + 5] = sp[i].emission[2];
        mat[i * 8 + 6] = kd;
        mat[i * 8 + 7] = kd * 16777216.0;  // kd * 2^24 (exact): the RR test compares it with the draw's integer
    }
}

// SceneView::axis_pat from flattened geometry rows (cx, cy, cz, r*r): 1 / 2 / 3 where x / y / z is the centre's only
// coordinate that is not +-0 (all finite), else 0; a plane's row (negative "r*r") is 0.
static uint64_t axis_pattern(const double* geom, size_t n) {
    uint64_t pat = 0;
    for (size_t i = 0; i < n && i < 32; ++i) {
        const double* g = geom + i * 4;
        if (!(std::isfinite(g[0]) && std::isfinite(g[1]) && std::isfinite(g[2])) || !(g[3] >= 0.0)) continue;
        const bool zx = g[0] == 0.0, zy = g[1] == 0.0, zz = g[2] == 0.0;
        const uint64_t a = (!zx && zy && zz) ? 1u : (zx && !zy && zz) ? 2u : (zx && zy && !zz) ? 3u : 0u;
        pat |= a << (2 * i);
    }
    return pat;
}

// SceneView::fold_flags from flattened material rows (colorKD[3], emission[3], kd, kd * 2^24): see kFoldNoLevelEmission
// kSceneCompact from flattened geometry rows (cx, cy, cz, r*r): every sphere within kCompactExtent of the origin
static unsigned compact_flag_of(const double* geom, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const double* g = geom + i * 4;
        const double reach = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]) + std::sqrt(std::fabs(g[3]));
        if (!(reach <= kCompactExtent)) return 0u;  // (NaN fails too)
    }
    return kSceneCompact;
}
static unsigned fold_flags_of(const double* mat, size_t n) {
    auto plain = [](double v) { return !std::signbit(v) && !std::isnan(v); };  // +0 or positive (or +inf)
    for (size_t i = 0; i < n; ++i) {
        const double* m = mat + i * 8;
        if (!(plain(m[3]) && plain(m[4]) && plain(m[5]))) return 0u;  // a path may END on any object: L starts as its emission
        if (m[6] > 0.0) {  // a path can bounce off this one (the roulette passes only for a draw <= kd, and draws are > 0)
            if (!(m[3] == 0.0 && m[4] == 0.0 && m[5] == 0.0)) return 0u;
            if (!(plain(m[0]) && plain(m[1]) && plain(m[2]))) return 0u;
        }
    }
    return kFoldNoLevelEmission;
}

__global__ void flatten_scene_kernel(const rtm_sphere* __restrict__ sp, size_t n,
                                     double* __restrict__ geom, double* __restrict__ mat, double* __restrict__ surf = nullptr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == n) {  // identity row for path_fold_blocked
        for (int k = 0; k < 8; ++k) mat[n * 8 + k] = (k < 3) ? 1.0 : 0.0;
        return;
    }
    if (i >= n) return;
    if (surf) {
        for (int k = 0; k < 3; ++k) surf[i * 4 + k] = sp[i].color[k];
        surf[i * 4 + 3] = (double)sp[i].radius;
    }
    const float r2 = sp[i].radius * sp[i].radius;
    geom[i * 4 + 0] = sp[i].center[0];
    geom[i * 4 + 1] = sp[i].center[1];
    geom[i * 4 + 2] = sp[i].center[2];
    geom[i * 4 + 3] = (double)r2;
    double m = sp[i].color[0] < sp[i].color[1] ? sp[i].color[1] : sp[i].color[0];
    m = m < sp[i].color[2] ? sp[i].color[2] : m;
    const double kd = (double)(float)m;
    mat[i * 8 + 0] = sp[i].color[0] / kd;
    mat[i * 8 + 1] = sp[i].color[1] / kd;
    mat[i * 8 + 2] = sp[i].color[2] / kd;
    mat[i * 8 + 3] = sp[i].emission[0];
    mat[i * 8 + 4] = sp[i].emission[1];
    mat[i * 8 + 5] = sp[i].emission[2];
    mat[i * 8 + 6] = kd;
    mat[i * 8 + 7] = kd * 16777216.0;
}

// ---- RAII for everything the host side owns: an early return (RTM_HIP_CHECK) releases it -------
// A library-owned stream per device for allocations that must come and go WITHOUT a device-wide wait: hipFree (like
// hipMalloc) synchronises the whole device — measured: destroying an idle scene took the 0.5 s another stream was busy
// for — while a stream-ordered free on this stream only returns the block to the device's memory pool.  Heap objects
// that are never destroyed (no HIP calls from static destructors at exit).
inline hipStream_t pool_stream(int device) {
    static std::mutex* mu = new std::mutex;
    static std::map<int, hipStream_t>* streams = new std::map<int, hipStream_t>;
    std::lock_guard<std::mutex> lock(*mu);
    auto it = streams->find(device);
    if (it != streams->end()) return it->second;
    hipStream_t st = nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
        uint64_t keep = ~(uint64_t)0;  // freed blocks stay in the pool instead of going back to the driver at every sync
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
    if (prev >= 0) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    (*streams)[device] = st;
    return st;
}

struct DevMem {
    void* p = nullptr;
    int pooled_device = -1;  // >= 0: from the device's memory pool through pool_stream (alloc_pooled)
    DevMem() = default;
    DevMem(const DevMem&) = delete;
    DevMem& operator=(const DevMem&) = delete;
    ~DevMem() { reset(); }
    void reset() {
        if (p) {
            if (pooled_device >= 0) (void)hipFreeAsync(p, pool_stream(pooled_device));
            else (void)hipFree(p);
        }
        p = nullptr;
        pooled_device = -1;
    }
    int alloc(size_t bytes) {
        reset();
        RTM_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
        return RTM_OK;
    }
    // Stream-ordered allocation on the library's own stream, waited for here (nothing else is ever queued on that
    // stream, so the wait is for the allocation alone): the block is usable on every stream on return, and its
    // release (reset) does not wait for the device.
    int alloc_pooled(size_t bytes, int device) {
        reset();
        hipStream_t st = pool_stream(device);
        if (!st) return alloc(bytes);
        RTM_HIP_CHECK(hipMallocAsync(&p, bytes ? bytes : 1, st));
        pooled_device = device;
        RTM_HIP_CHECK(hipStreamSynchronize(st));
        return RTM_OK;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};
// stream-ordered temporary (hipMallocAsync / hipFreeAsync on the same stream): no synchronisation
struct AsyncMem {
    void* p = nullptr;
    hipStream_t stream = nullptr;
    AsyncMem() = default;
    AsyncMem(const AsyncMem&) = delete;
    AsyncMem& operator=(const AsyncMem&) = delete;
    ~AsyncMem() {
        if (p) (void)hipFreeAsync(p, stream);
    }
    int alloc(size_t bytes, hipStream_t st) {
        stream = st;
        RTM_HIP_CHECK(hipMallocAsync(&p, bytes ? bytes : 1, st));
        return RTM_OK;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};
struct PinnedMem {
    void* p = nullptr;
    ~PinnedMem() {
        if (p) (void)hipHostFree(p);
    }
    int alloc(size_t bytes) {
        RTM_HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        return RTM_OK;
    }
};
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    int create() {
        RTM_HIP_CHECK(hipEventCreate(&a));
        RTM_HIP_CHECK(hipEventCreate(&b));
        return RTM_OK;
    }
};

}  // namespace rtm

// The scene as the kernels read it, resident on one device (include/rtm.h: rtm_scene).  geom / mat as in
// DESIGN.md §2; `aux` holds the large-scene rejection-test data (rtm_wavefront.h: bounds[4], w'[n_pad],
// float list), computed once here instead of per render.
struct rtm_scene {
    int device = 0;
    size_t n = 0;
    rtm::DevMem geom, mat, aux, plane;  // plane: 16 doubles per object, only for scenes that hold planes
    rtm::DevMem surf;                   // raw colour + radius per object: png::SurfaeSample's extras (rtm_surface.h)
    rtm::DevMem grid;                   // large all-sphere scenes: GridHeader + cell offsets + cell lists + big list
    size_t grid_cells = 0, grid_refs = 0, grid_big = 0;
    rtm::GridHeader grid_hdr;           // host copy (grid_for: is the camera within the pads' reach?)
    bool grid_far_bounces = false;      // a diffuse sphere encloses the gridded ones from beyond the pads' reach (build_scene_grid)
    bool has_planes = false;
    unsigned fold_flags = 0;            // SceneView::fold_flags (fold_flags_of)
    uint64_t axis_pat = 0;              // SceneView::axis_pat: which of the first 32 spheres sit on a coordinate axis (axis_pattern)
    uint64_t content_hash = 0;            // cache entries only ...
    std::vector<unsigned char> content;   // ... and the bytes the hash was taken of (compared on a hash hit)
    // Streams that have rendered from this scene, one event each, re-recorded behind every render that names the scene:
    // rtm_scene_destroy looks at them instead of waiting for the device (scene_destroy below).
    mutable std::mutex use_mu;
    mutable std::vector<std::pair<hipStream_t, hipEvent_t>> uses;
    ~rtm_scene() {
        for (auto& u : uses)
            if (u.second) (void)hipEventDestroy(u.second);
    }
};

namespace rtm {

static size_t scene_aux_doubles(size_t n) {
    const size_t n_pad = (n + 7) & ~(size_t)7;
    return n_pad * 5 + 4;  // bounds[4], w'[n_pad], float list in sphere pairs (n_pad x 16 B), float list per sphere (same)
}
// bounds / w' / float list of the large-scene rejection tests, on `stream`
static int launch_scene_aux(const double* geom, size_t n, double* aux, hipStream_t stream) {
    const int n_pad = (int)((n + 7) & ~(size_t)7);
    RTM_HIP_CHECK(hipMemsetAsync(aux, 0, 4 * sizeof(double), stream));
    float4* g32 = reinterpret_cast<float4*>(aux + 4 + n_pad);
    if (n_pad > 0) {
        wf_scene_scale_kernel<<<(n_pad + 255) / 256, 256, 0, stream>>>((const double4*)geom, (int)n,
                                                                      reinterpret_cast<long long*>(aux + 2));
        wf_scene_aux_kernel<<<(n_pad + 255) / 256, 256, 0, stream>>>((const double4*)geom, (int)n, n_pad, aux + 4,
                                                                    reinterpret_cast<unsigned long long*>(aux), g32, g32 + n_pad);
    }
    RTM_HIP_CHECK(hipGetLastError());
    return RTM_OK;
}
static SceneView scene_view(const double* geom, const double* mat, const double* aux, size_t n,
                            const double* plane = nullptr, const void* grid = nullptr, const double* surf = nullptr,
                            uint64_t axis_pat = 0, unsigned fold_flags = 0) {
    SceneView v{(const double4*)geom, mat, (int)n};
    v.axis_pat = axis_pat;
    v.fold_flags = fold_flags;
    v.plane = plane;
    v.surf = surf;
    v.grid = static_cast<const GridHeader*>(grid);
    if (aux) {
        const size_t n_pad = (n + 7) & ~(size_t)7;
        v.bounds = aux;
        v.wprime = aux + 4;
        v.geom32 = reinterpret_cast<const float4*>(aux + 4 + n_pad);
        v.geom32s = v.geom32 + n_pad;
    }
    return v;
}

// ---- uniform grid of a large scene (rtm_path.h: GridHeader, GridWalk — where the pads are derived) ------------------
// Built on the host from the flattened geometry rows (cx, cy, cz, float r*r) when a scene OBJECT is created: 25 ms for
// 100 000 spheres on one core, once per scene.  ~2 cells per sphere (RTM_DEBUG_GRID_CELLS); a sphere whose padded box
// covers more than kGridBigCells cells goes to the list every ray tests.  No grid (the other kernels serve the scene)
// for fewer than kGridMinSpheres gridded spheres (below that the packed-record kernels are as fast or faster:
// profiles/r3/grid_crossover.txt), non-finite geometry, more than kGridMaxBig big spheres or planes.
#ifndef RTM_GRID_MIN
#define RTM_GRID_MIN 64
#endif
constexpr size_t kGridMinSpheres = RTM_GRID_MIN;  // gridded spheres a scene needs to get a grid (profiles/r3/grid_crossover.txt)
constexpr int kGridBigCells = 125, kGridMaxBig = 1024, kGridMaxDim = 1024;
constexpr double kGridDdTol = 4e-7;  // |dir.dir - 1| the pads cover: twice what the reference's float-sqrt Normalize leaves (1.8e-7)
static double grid_cells_per_sphere() {
    static const double v = [] {
        const char* e = std::getenv("RTM_DEBUG_GRID_CELLS");  // tuning knob: cells per sphere; 0 = build no grid
        return e ? std::strtod(e, nullptr) : 1.5;  // (round 4, with the next-cell fill: flat between 1.35 and 1.65, 2.0 is 2.3 % slower — profiles/r4/grid_cells.txt)
    }();
    return v;
}
struct GridBuild {
    GridHeader hdr;
    std::vector<unsigned> cell_start, items;
    std::vector<int> big;
    size_t refs = 0;  // entries of `items` that are real (items holds one dummy when there is none)
};
static bool n_refs_valid(const GridBuild& B) { return B.refs != 0; }
static bool make_grid(const double* g, size_t n, GridBuild& out, double* pads_out = nullptr) {
    const double lambda = grid_cells_per_sphere();
    if (n < kGridMinSpheres || n >= (1u << 29) || !(lambda > 0.0)) return false;
    std::vector<double> R(n), pad(n);
    std::vector<char> is_big(n, 0);
    std::vector<char> is_plane(n, 0);  // a png::PlaneObject's geometry row is (position, negative "r*r"): tested by every ray
    for (size_t i = 0; i < n; ++i) {
        const double* r = g + i * 4;
        if (!(std::isfinite(r[0]) && std::isfinite(r[1]) && std::isfinite(r[2]) && std::isfinite(r[3]))) return false;
        if (r[3] < 0.0) {
            is_plane[i] = is_big[i] = 1;
            R[i] = 0.0;
            continue;
        }
        unsigned long long wbits;
        std::memcpy(&wbits, &r[3], sizeof wbits);
        if (wbits & 0x1FFFFFFFull) return false;  // (never: r*r is a float widened to double — the records keep the index there)
        R[i] = std::sqrt(r[3]);
    }
    double lo[3], hi[3], h = 0.0, t_ok = 0.0;
    size_t n_small = 0;
    for (size_t i = 0; i < n; ++i) n_small += !is_plane[i];
    if (n_small < kGridMinSpheres) return false;
    for (int round = 0; round < 4; ++round) {
        // box of the small spheres (unpadded), cell edge from its volume, pads from its diagonal
        for (int k = 0; k < 3; ++k) {
            lo[k] = HUGE_VAL;
            hi[k] = -HUGE_VAL;
        }
        for (size_t i = 0; i < n; ++i) {
            if (is_big[i]) continue;
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(lo[k], g[i * 4 + k] - R[i]);
                hi[k] = std::max(hi[k], g[i * 4 + k] + R[i]);
            }
        }
        double ext[3], diag2 = 0.0, vol = 1.0, longest = 0.0;
        for (int k = 0; k < 3; ++k) {
            ext[k] = hi[k] - lo[k];
            diag2 += ext[k] * ext[k];
            longest = std::max(longest, ext[k]);
        }
        if (!(longest > 0.0) || !std::isfinite(diag2)) return false;
        for (int k = 0; k < 3; ++k) vol *= std::max(ext[k], longest * 1e-3);  // (a flat scene still gets cells of a sane size)
        h = std::cbrt(vol / (lambda * (double)n_small));
        h = std::max(h, longest / (double)(kGridMaxDim - 2));
        t_ok = 2.5 * std::sqrt(diag2);
        size_t changed = 0;
        n_small = 0;
        for (size_t i = 0; i < n; ++i) {
            // (rtm_path.h: a hit at parameter t lies sqrt(r^2 + t^2 (d.d - 1)) from the centre)
            pad[i] = 0.05 * h + (std::sqrt(R[i] * R[i] + kGridDdTol * t_ok * t_ok) - R[i]) + 1e-6 * t_ok;
            const double side = 2.0 * (R[i] + pad[i]) / h + 1.0;
            const char b = is_plane[i] || !(side * side * side <= (double)kGridBigCells);
            changed += b != is_big[i];
            is_big[i] = b;
            n_small += !b;
        }
        if (n_small < kGridMinSpheres) return false;
        if (!changed && round > 0) break;
    }
    out.big.clear();
    for (size_t i = 0; i < n; ++i)
        if (is_big[i]) out.big.push_back((int)i);
    if (out.big.size() > (size_t)kGridMaxBig) return false;
    // the grid's box: the small spheres' padded boxes
    for (int k = 0; k < 3; ++k) {
        lo[k] = HUGE_VAL;
        hi[k] = -HUGE_VAL;
    }
    for (size_t i = 0; i < n; ++i) {
        if (is_big[i]) continue;
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], g[i * 4 + k] - R[i] - pad[i]);
            hi[k] = std::max(hi[k], g[i * 4 + k] + R[i] + pad[i]);
        }
    }
    if (pads_out) std::memcpy(pads_out, pad.data(), n * sizeof(double));
    GridHeader& H = out.hdr;
    std::memset(&H, 0, sizeof H);
    size_t cells = 1;
    for (int k = 0; k < 3; ++k) {
        lo[k] -= 0.01 * h;
        int d = (int)std::ceil((hi[k] + 0.01 * h - lo[k]) / h);
        d = d < 1 ? 1 : d;
        if (d > kGridMaxDim) return false;
        H.lo[k] = lo[k];
        H.hi[k] = lo[k] + (double)d * h;
        H.dim[k] = d;
        cells *= (size_t)d;
    }
    if (cells > 64 * n + 4096) return false;  // (cannot happen with cells of the volume's size; a guard for the host's memory)
    H.h = h;
    H.inv_h = 1.0 / h;
    H.t_ok = t_ok;
    double half2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        H.cb[k] = 0.5 * (H.lo[k] + H.hi[k]);
        half2 += 0.25 * (H.hi[k] - H.lo[k]) * (H.hi[k] - H.lo[k]);
    }
    const double reach = t_ok / 1.001 - std::sqrt(half2);  // (t_ok = 2.5 diagonals of the unpadded box: about two of them)
    if (!(reach > 0.0)) return false;
    H.reach2 = reach * reach;
    H.dd_tol = kGridDdTol;
    H.n_big = (int)out.big.size();
    // counting sort of (cell, sphere) pairs; spheres are visited in index order, so every cell's list ascends
    auto cell_range = [&](size_t i, int k, int& a, int& b) {
        const double c = g[i * 4 + k], e = R[i] + pad[i];
        a = (int)std::floor((c - e - H.lo[k]) / h);
        b = (int)std::floor((c + e - H.lo[k]) / h);
        a = a < 0 ? 0 : a;
        b = b >= H.dim[k] ? H.dim[k] - 1 : b;
    };
    out.cell_start.assign(cells + 1, 0u);
    size_t refs = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (size_t i = 0; i < n; ++i) {
            if (is_big[i]) continue;
            int a[3], b[3];
            for (int k = 0; k < 3; ++k) cell_range(i, k, a[k], b[k]);
            // of the box's cells, those within R + pad of the centre (a hit point is, and lies in its cell)
            const double reach2 = (R[i] + pad[i]) * (R[i] + pad[i]) * (1.0 + 1e-9);
            auto gap = [&](int k, int cell) {  // distance from the centre to the cell's slab along axis k
                const double c0 = H.lo[k] + (double)cell * h, ck = g[i * 4 + k];
                return ck < c0 ? c0 - ck : (ck > c0 + h ? ck - (c0 + h) : 0.0);
            };
            for (int z = a[2]; z <= b[2]; ++z)
                for (int y = a[1]; y <= b[1]; ++y)
                    for (int x = a[0]; x <= b[0]; ++x) {
                        const double gx = gap(0, x), gy = gap(1, y), gz = gap(2, z);
                        if (gx * gx + gy * gy + gz * gz > reach2) continue;
                        const size_t c = ((size_t)z * H.dim[1] + y) * H.dim[0] + x;
                        if (pass == 0) {
                            ++out.cell_start[c + 1];
                            ++refs;
                        } else {
                            out.items[out.cell_start[c]++] = (unsigned)i;
                        }
                    }
        }
        if (pass == 0) {
            if (refs > 0xFFFFFF00ull) return false;
            for (size_t c = 0; c < cells; ++c) out.cell_start[c + 1] += out.cell_start[c];
            out.items.assign(refs ? refs : 1, 0u);
            out.refs = refs;
        } else {  // the fill advanced every start to its end: shift back
            for (size_t c = cells; c > 0; --c) out.cell_start[c] = out.cell_start[c - 1];
            out.cell_start[0] = 0u;
        }
    }
    return true;
}
// rtm_debug_grid_build: the builder alone, on the HOST (no device is touched) — what tests/test_host_io.py checks the
// lists' invariants on.  info[12] = {cells, records, big, dim x, y, z} then, as doubles' bit patterns, {lo x, y, z, h, reach,
// t_ok}; pads: n doubles (a big sphere's is its pad too); with buffers, ranges: 2 x cells unsigned, items: `records` sphere
// indices, big: `big` sphere indices.  RTM_ERR_UNSUPPORTED when the scene gets no grid, RTM_ERR_CAPACITY when a buffer is short.
int grid_build_host(const rtm_sphere* sp, size_t n, uint64_t* info, double* pads, uint32_t* ranges, size_t ranges_cap,
                    uint32_t* items, size_t items_cap, int32_t* big, size_t big_cap) {
    if (!sp || !n || !info) return RTM_ERR_INVALID_ARGUMENT;
    std::vector<double> hg, hm;
    flatten_scene(sp, n, hg, hm);
    GridBuild B;
    if (!make_grid(hg.data(), n, B, pads)) {
        set_last_error("this scene gets no grid");
        return RTM_ERR_UNSUPPORTED;
    }
    const size_t cells = B.cell_start.size() - 1;
    info[0] = cells;
    info[1] = B.refs;
    info[2] = B.big.size();
    for (int k = 0; k < 3; ++k) info[3 + k] = (uint64_t)B.hdr.dim[k];
    const double d[6] = {B.hdr.lo[0], B.hdr.lo[1], B.hdr.lo[2], B.hdr.h, std::sqrt(B.hdr.reach2), B.hdr.t_ok};
    std::memcpy(info + 6, d, sizeof d);
    if ((ranges && ranges_cap < cells * 2) || (items && items_cap < B.refs) || (big && big_cap < B.big.size())) return RTM_ERR_CAPACITY;
    if (ranges)
        for (size_t c = 0; c < cells; ++c) {
            ranges[c * 2] = B.cell_start[c];
            ranges[c * 2 + 1] = B.cell_start[c + 1];
        }
    if (items) std::memcpy(items, B.items.data(), B.refs * sizeof(unsigned));
    if (big && !B.big.empty()) std::memcpy(big, B.big.data(), B.big.size() * sizeof(int));
    return RTM_OK;
}

// rtm_debug_scene_facts: axis_pattern and fold_flags_of on the HOST, from the caller's spheres (tests/test_host_io.py)
int scene_facts_host(const rtm_sphere* sp, size_t n, uint64_t* facts) {
    if ((!sp && n) || !facts) return RTM_ERR_INVALID_ARGUMENT;
    std::vector<double> hg, hm;
    flatten_scene(sp, n, hg, hm);
    facts[0] = axis_pattern(hg.data(), n);
    facts[1] = fold_flags_of(hm.data(), n) | compact_flag_of(hg.data(), n);
    return RTM_OK;
}

// Build + upload; a scene that gets no grid keeps sc.grid empty (not an error).  `hg`: the host copy of the geometry rows.
// `hm`: the material rows (kd in column 6), or null.
static int build_scene_grid(rtm_scene& sc, const double* hg, const double* hm, size_t n, int device,
                            const double* plane_rows = nullptr) {
    GridBuild B;
    if (!make_grid(hg, n, B)) return RTM_OK;
    // A diffuse sphere that ENCLOSES the gridded ones from farther away than the pads reach (an environment sphere) sends
    // its bounces back from origins the walk cannot serve: each of them would take the exhaustive loop inside the grid
    // kernel, one lane at a time.  Such a scene keeps its grid for variant 17 by name; variant 0 leaves it alone (grid_for).
    sc.grid_far_bounces = false;
    const double reach = std::sqrt(B.hdr.reach2);
    for (int i : B.big) {
        const bool diffuse = !hm || hm[(size_t)i * 8 + 6] > 0.0;
        if (hg[(size_t)i * 4 + 3] < 0.0) {  // a plane: any of its square's corners beyond the reach?
            if (!plane_rows) return RTM_OK;  // (no rows to judge it by: no grid)
            const double* pl = plane_rows + (size_t)i * 16;
            for (int c = 0; c < 4 && diffuse; ++c) {
                double d2 = 0.0;
                for (int k = 0; k < 3; ++k) {
                    const double p = pl[k] + ((c & 1) ? pl[6 + k] : -pl[6 + k]) + ((c & 2) ? pl[9 + k] : -pl[9 + k]);
                    d2 += (p - B.hdr.cb[k]) * (p - B.hdr.cb[k]);
                }
                if (!(std::sqrt(d2) <= reach)) sc.grid_far_bounces = true;
            }
            continue;
        }
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) d2 += (hg[(size_t)i * 4 + k] - B.hdr.cb[k]) * (hg[(size_t)i * 4 + k] - B.hdr.cb[k]);
        if (diffuse && std::sqrt(hg[(size_t)i * 4 + 3]) - std::sqrt(d2) > reach) sc.grid_far_bounces = true;
    }
    const size_t off_cs = (sizeof(GridHeader) + 255) & ~(size_t)255;
    const size_t n_cells = B.cell_start.size() - 1;
    std::vector<unsigned> ranges(n_cells * 2);  // (first, one past last) per cell: one 8-byte load per cell step
    for (size_t c = 0; c < n_cells; ++c) {
        ranges[c * 2] = B.cell_start[c];
        ranges[c * 2 + 1] = B.cell_start[c + 1];
    }
    const size_t off_items = off_cs + ((ranges.size() * sizeof(unsigned) + 255) & ~(size_t)255);
    const size_t off_big = off_items + ((B.items.size() * 4 * sizeof(double) + 255) & ~(size_t)255);
    // the cell lists as self-contained records: the geometry row with the sphere's index in r*r's 29 zero mantissa bits
    std::vector<double> recs(B.items.size() * 4);
    for (size_t k = 0; k < B.items.size(); ++k) {
        const unsigned i = n_refs_valid(B) ? B.items[k] : 0u;
        std::memcpy(&recs[k * 4], hg + (size_t)i * 4, 4 * sizeof(double));
        unsigned long long wbits;
        std::memcpy(&wbits, &recs[k * 4 + 3], sizeof wbits);
        wbits |= (unsigned long long)i;
        std::memcpy(&recs[k * 4 + 3], &wbits, sizeof wbits);
    }
    const size_t bytes = off_big + (B.big.size() + 1) * sizeof(int);
    const int rc = sc.grid.alloc_pooled(bytes, device);
    if (rc != RTM_OK) return rc;
    unsigned char* base = sc.grid.as<unsigned char>();
    B.hdr.cell_range = reinterpret_cast<const uint2*>(base + off_cs);
    B.hdr.recs = reinterpret_cast<const double4*>(base + off_items);
    B.hdr.n_recs = (unsigned)B.items.size();
    B.hdr.big = reinterpret_cast<const int*>(base + off_big);
    RTM_HIP_CHECK(hipMemcpy(base, &B.hdr, sizeof B.hdr, hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(base + off_cs, ranges.data(), ranges.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(base + off_items, recs.data(), recs.size() * sizeof(double), hipMemcpyHostToDevice));
    if (!B.big.empty())
        RTM_HIP_CHECK(hipMemcpy(base + off_big, B.big.data(), B.big.size() * sizeof(int), hipMemcpyHostToDevice));
    sc.grid_hdr = B.hdr;
    sc.grid_cells = B.cell_start.size() - 1;
    sc.grid_refs = B.items.size();
    sc.grid_big = B.big.size();
    return RTM_OK;
}

// Flatten + upload a HOST sphere array; returns when the tables are resident (the caller's array and
// the staging vectors may go away).  Runs on the device's null stream.
static int scene_build_host(rtm_scene& sc, const rtm_sphere* sp, size_t n, int device, bool want_grid = true) {
    RTM_HIP_CHECK(hipSetDevice(device));
    sc.device = device;
    sc.n = n;
    std::vector<double> hg, hm, hs;
    flatten_scene(sp, n, hg, hm, &hs);
    sc.axis_pat = axis_pattern(hg.data(), n);
    sc.fold_flags = fold_flags_of(hm.data(), n) | compact_flag_of(hg.data(), n);
    int rc = sc.geom.alloc_pooled((n ? n : 1) * 4 * sizeof(double), device);
    if (rc == RTM_OK) rc = sc.mat.alloc_pooled((n + 1) * 8 * sizeof(double), device);
    if (rc == RTM_OK) rc = sc.aux.alloc_pooled(scene_aux_doubles(n) * sizeof(double), device);
    if (rc == RTM_OK) rc = sc.surf.alloc_pooled(hs.size() * sizeof(double), device);
    if (rc != RTM_OK) return rc;
    RTM_HIP_CHECK(hipMemcpy(sc.surf.p, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice));
    if (n) RTM_HIP_CHECK(hipMemcpy(sc.geom.p, hg.data(), hg.size() * sizeof(double), hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(sc.mat.p, hm.data(), hm.size() * sizeof(double), hipMemcpyHostToDevice));
    rc = launch_scene_aux(sc.geom.as<double>(), n, sc.aux.as<double>(), nullptr);
    if (rc != RTM_OK) return rc;
    RTM_HIP_CHECK(hipStreamSynchronize(nullptr));
    return want_grid ? build_scene_grid(sc, hg.data(), hm.data(), n, device) : RTM_OK;
}
// The same from a DEVICE sphere array.
static int scene_build_device(rtm_scene& sc, const rtm_sphere* sp_dev, size_t n, int device) {
    RTM_HIP_CHECK(hipSetDevice(device));
    sc.device = device;
    sc.n = n;
    int rc = sc.geom.alloc_pooled((n ? n : 1) * 4 * sizeof(double), device);
    if (rc == RTM_OK) rc = sc.mat.alloc_pooled((n + 1) * 8 * sizeof(double), device);
    if (rc == RTM_OK) rc = sc.aux.alloc_pooled(scene_aux_doubles(n) * sizeof(double), device);
    if (rc == RTM_OK) rc = sc.surf.alloc_pooled((n ? n : 1) * 4 * sizeof(double), device);
    if (rc != RTM_OK) return rc;
    flatten_scene_kernel<<<(unsigned)((n + 1 + 255) / 256), 256>>>(sp_dev, n, sc.geom.as<double>(), sc.mat.as<double>(), sc.surf.as<double>());
    RTM_HIP_CHECK(hipGetLastError());
    rc = launch_scene_aux(sc.geom.as<double>(), n, sc.aux.as<double>(), nullptr);
    if (rc != RTM_OK) return rc;
    RTM_HIP_CHECK(hipStreamSynchronize(nullptr));
    if (n > 0 && n <= 32) {  // a small scene's rows come back for SceneView::axis_pat (1 KB)
        std::vector<double> rows(n * 4);
        RTM_HIP_CHECK(hipMemcpy(rows.data(), sc.geom.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
        sc.axis_pat = axis_pattern(rows.data(), n);
        std::vector<double> mrows((n + 1) * 8);
        RTM_HIP_CHECK(hipMemcpy(mrows.data(), sc.mat.p, mrows.size() * sizeof(double), hipMemcpyDeviceToHost));
        sc.fold_flags = fold_flags_of(mrows.data(), n) | compact_flag_of(rows.data(), n);
    }
    if (n < kGridMinSpheres) return RTM_OK;
    std::vector<double> hg(n * 4);  // the grid is built on the host: the geometry rows come back once
    RTM_HIP_CHECK(hipMemcpy(hg.data(), sc.geom.p, hg.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> hm((n + 1) * 8);
    RTM_HIP_CHECK(hipMemcpy(hm.data(), sc.mat.p, hm.size() * sizeof(double), hipMemcpyDeviceToHost));
    return build_scene_grid(sc, hg.data(), hm.data(), n, device);
}

int scene_create(const rtm_sphere* sp, size_t n, int on_device, int device, rtm_scene** out) {
    if (!out || (!sp && n) || n > 0x7FFFFFFFull) {
        set_last_error("null argument or scene too large");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    *out = nullptr;
    std::unique_ptr<rtm_scene> sc(new rtm_scene);
    const int rc = on_device ? scene_build_device(*sc, sp, n, device) : scene_build_host(*sc, sp, n, device);
    if (rc != RTM_OK) return rc;
    *out = sc.release();
    return RTM_OK;
}
// png::PlaneObject's constructor (src/SettingData.cpp:235-242) on the host, in the reference's operation order;
// `upv` and the two squared half-extents are this build's completion (include/rtm.h).
static void plane_row(const rtm_object& o, double row[16]) {
    using namespace host;
    const H3 pos = {o.position[0], o.position[1], o.position[2]};
    const H3 nrm = normalize(sub(H3{o.target[0], o.target[1], o.target[2]}, pos));  // :240
    const H3 rn = normalize(cross(nrm, H3{o.up[0], o.up[1], o.up[2]}));             // :241 Normalize(Cross(..))
    const H3 right = {rn.x * 0.5 * o.width, rn.y * 0.5 * o.width, rn.z * 0.5 * o.width};  // (v * 0.5) * width
    const H3 upv = cross(right, nrm);
    const double v[16] = {pos.x, pos.y, pos.z, nrm.x, nrm.y, nrm.z, right.x, right.y, right.z, upv.x, upv.y, upv.z,
                          right.x * right.x + right.y * right.y + right.z * right.z,
                          upv.x * upv.x + upv.y * upv.y + upv.z * upv.z, 0.0, 0.0};
    std::memcpy(row, v, sizeof v);
}

int scene_create_objects(const rtm_object* objs, size_t n, int device, rtm_scene** out) {
    if (!out || (!objs && n) || n > 0x7FFFFFFFull) {
        set_last_error("null argument or scene too large");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    *out = nullptr;
    std::vector<rtm_sphere> view(n ? n : 1);
    std::vector<double> rows(n * 16, 0.0);
    bool any_plane = false;
    for (size_t i = 0; i < n; ++i) {
        if (objs[i].type != RTM_OBJECT_SPHERE && objs[i].type != RTM_OBJECT_PLANE) {
            set_last_error("unknown object type (1 = sphere, 2 = plane)");
            return RTM_ERR_INVALID_SCENE;
        }
        std::memset(&view[i], 0, sizeof view[i]);
        for (int k = 0; k < 3; ++k) {
            view[i].center[k] = objs[i].position[k];
            view[i].color[k] = objs[i].color[k];
            view[i].emission[k] = objs[i].emission[k];
        }
        view[i].radius = objs[i].size;
        if (objs[i].type == RTM_OBJECT_PLANE) {
            any_plane = true;
            plane_row(objs[i], &rows[i * 16]);
        }
    }
    std::unique_ptr<rtm_scene> sc(new rtm_scene);
    int rc = scene_build_host(*sc, view.data(), n, device, !any_plane);
    if (rc != RTM_OK) return rc;
    if (any_plane) {
        // a plane's geometry row is (position, -1): the negative "r*r" marks it for the per-object loop
        std::vector<double> hg(n * 4);
        RTM_HIP_CHECK(hipMemcpy(hg.data(), sc->geom.p, hg.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i)
            if (objs[i].type == RTM_OBJECT_PLANE) hg[i * 4 + 3] = -1.0;
        RTM_HIP_CHECK(hipMemcpy(sc->geom.p, hg.data(), hg.size() * sizeof(double), hipMemcpyHostToDevice));
        if ((rc = sc->plane.alloc_pooled(rows.size() * sizeof(double), device)) != RTM_OK) return rc;
        RTM_HIP_CHECK(hipMemcpy(sc->plane.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
        sc->has_planes = true;
        sc->axis_pat = 0;  // (the object kernels do not read it; a plane's position row is not a sphere's centre)
        // the grid of a scene with planes: over its spheres, the planes among the objects every ray tests (build_scene_grid)
        std::vector<double> hm((n + 1) * 8);
        RTM_HIP_CHECK(hipMemcpy(hm.data(), sc->mat.p, hm.size() * sizeof(double), hipMemcpyDeviceToHost));
        if ((rc = build_scene_grid(*sc, hg.data(), hm.data(), n, device, rows.data())) != RTM_OK) return rc;
    }
    *out = sc.release();
    return RTM_OK;
}

int intersect_objects_batch(const rtm_object* objs, const double* org, const double* dir, size_t n, int mode,
                            int32_t* out_hit, double* out_t, double* out_normal) {
    if (!objs || !org || !dir || !out_hit || !out_t || !out_normal) {
        set_last_error("null argument");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return RTM_OK;
    int device = 0;
    RTM_HIP_CHECK(hipGetDevice(&device));
    rtm_scene* raw = nullptr;
    int rc = scene_create_objects(objs, n, device, &raw);
    if (rc != RTM_OK) return rc;
    std::unique_ptr<rtm_scene> ds(raw);
    DevMem d_org, d_dir, d_t, d_n, d_hit;
    const size_t vb = n * 3 * sizeof(double);
    if ((rc = d_org.alloc(vb)) != RTM_OK || (rc = d_dir.alloc(vb)) != RTM_OK || (rc = d_n.alloc(vb)) != RTM_OK ||
        (rc = d_t.alloc(n * sizeof(double))) != RTM_OK || (rc = d_hit.alloc(n * sizeof(int32_t))) != RTM_OK)
        return rc;
    RTM_HIP_CHECK(hipMemcpy(d_org.p, org, vb, hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(d_dir.p, dir, vb, hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(d_n.p, out_normal, vb, hipMemcpyHostToDevice));  // untouched where no hit
    RTM_HIP_CHECK(hipMemcpy(d_t.p, out_t, n * sizeof(double), hipMemcpyHostToDevice));
    intersect_pairs_kernel<<<(unsigned)((n + 63) / 64), 64>>>((const double4*)ds->geom.p, d_org.as<double>(), d_dir.as<double>(),
                                                              n, mode, d_hit.as<int32_t>(), d_t.as<double>(), d_n.as<double>(),
                                                              ds->has_planes ? ds->plane.as<double>() : nullptr);
    RTM_HIP_CHECK(hipGetLastError());
    RTM_HIP_CHECK(hipMemcpy(out_hit, d_hit.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    RTM_HIP_CHECK(hipMemcpy(out_t, d_t.p, n * sizeof(double), hipMemcpyDeviceToHost));
    RTM_HIP_CHECK(hipMemcpy(out_normal, d_n.p, vb, hipMemcpyDeviceToHost));
    return RTM_OK;
}

// ---- scene release ------------------------------------------------------------------------------------------------
// include/rtm.h: a scene may be destroyed as soon as the last render that uses it has been ENQUEUED, and the call
// does not wait for that work.  Every render records an event behind its launches on its stream (note_scene_use);
// destroy frees the tables at once when all of them have completed, and otherwise parks the scene in a process-wide
// list that later library calls (and rtm_release_scratch, which does wait) reap with hipEventQuery.  The list is a
// heap object that is never destroyed: nothing here calls HIP from a static destructor at process exit.
namespace {
struct Graveyard {
    std::mutex mu;
    std::vector<rtm_scene*> parked;
};
Graveyard& graveyard() {
    static Graveyard* g = new Graveyard;
    return *g;
}
struct DeviceGuard {  // the caller's current device is restored on every path
    int prev = -1;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
        (void)hipGetLastError();
    }
};
bool scene_idle(const rtm_scene* sc, bool wait) {
    std::lock_guard<std::mutex> lock(sc->use_mu);
    for (auto& u : sc->uses) {
        if (!u.second) continue;
        if (wait) {
            (void)hipEventSynchronize(u.second);
        } else if (hipEventQuery(u.second) == hipErrorNotReady) {
            (void)hipGetLastError();
            return false;
        }
    }
    return true;
}
}  // namespace

static void note_scene_use(const rtm_scene* sc, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(sc->use_mu);
    for (auto& u : sc->uses)
        if (u.first == stream) {
            (void)hipEventRecord(u.second, stream);
            return;
        }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    (void)hipEventRecord(e, stream);
    sc->uses.emplace_back(stream, e);
}

// Free every parked scene whose renders have finished (wait: all of them, after waiting).  Never blocks otherwise.
static void reap_scenes(bool wait, int device = -1) {
    std::vector<rtm_scene*> done;
    {
        Graveyard& g = graveyard();
        std::lock_guard<std::mutex> lock(g.mu);
        if (g.parked.empty()) return;
        for (auto it = g.parked.begin(); it != g.parked.end();) {
            if ((device < 0 || (*it)->device == device) && scene_idle(*it, wait)) {
                done.push_back(*it);
                it = g.parked.erase(it);
            } else {
                ++it;
            }
        }
    }
    if (done.empty()) return;
    DeviceGuard guard;
    for (rtm_scene* sc : done) {
        (void)hipSetDevice(sc->device);
        delete sc;
    }
}

int scene_destroy(rtm_scene* sc) {
    if (!sc) return RTM_OK;
    reap_scenes(false);
    if (scene_idle(sc, false)) {
        DeviceGuard guard;
        (void)hipSetDevice(sc->device);
        delete sc;
    } else {  // renders from it are still queued or running: its memory goes when they have finished
        Graveyard& g = graveyard();
        std::lock_guard<std::mutex> lock(g.mu);
        g.parked.push_back(sc);
    }
    return RTM_OK;
}
size_t scene_size(const rtm_scene* sc) { return sc ? sc->n : 0; }

// RTM_MODE_HOST_TRIG: the table that turns the device's sin/cos of r1 into the host libm's.
// r1 = 6.283185307179586 * u with u = (2k+1) 2^-24, k < 2^23 (src/Renderer.cpp:88, rng_bits_to_u01), so
// the whole domain is evaluated once per device: the device with the shading block's own sincos, the
// host with glibc's sincos() (several threads; g++ -O2 turns the reference's adjacent cos(r1), sin(r1)
// into that one call, and it is not bit-identical to sin()/cos()), and the bit-pattern differences, which must be
// -1, 0 or +1, are packed as two signed 2-bit fields per k (sin low, cos high), 8 k per word, 4 MB.
__global__ __launch_bounds__(256) void trig_domain_kernel(double* __restrict__ sn, double* __restrict__ cs) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    const double r1 = 6.283185307179586 * rng_bits_to_u01(k << 9);
    double s, c;
    sincos_small(r1, s, c);
    sn[k] = s;
    cs[k] = c;
}

namespace {
std::mutex g_trig_mu;
std::map<int, uint32_t*> g_trig_ready;  // per device; dropped by release_scratch
}  // namespace

static int ensure_trig_fix(int device, const uint32_t** out) {
    std::lock_guard<std::mutex> lock(g_trig_mu);
    auto& ready = g_trig_ready;
    auto it = ready.find(device);
    if (it != ready.end()) {
        *out = it->second;
        return RTM_OK;
    }
    constexpr uint32_t N = 1u << 23;
    std::vector<double> hs(N), hc(N);
    {
        DevMem d_s, d_c;
        int rc;
        if ((rc = d_s.alloc((size_t)N * 8)) != RTM_OK || (rc = d_c.alloc((size_t)N * 8)) != RTM_OK) return rc;
        trig_domain_kernel<<<N / 256, 256>>>(d_s.as<double>(), d_c.as<double>());
        RTM_HIP_CHECK(hipGetLastError());
        RTM_HIP_CHECK(hipMemcpy(hs.data(), d_s.p, (size_t)N * 8, hipMemcpyDeviceToHost));
        RTM_HIP_CHECK(hipMemcpy(hc.data(), d_c.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    }
    std::vector<uint32_t> table(N / 8, 0u);
    std::atomic<unsigned long long> out_of_range{0};
    unsigned threads = std::thread::hardware_concurrency();
    threads = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
    const uint32_t words = N / 8;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            for (uint32_t w = (uint32_t)((unsigned long long)words * t / threads);
                 w < (uint32_t)((unsigned long long)words * (t + 1) / threads); ++w) {
                uint32_t packed = 0;
                for (uint32_t j = 0; j < 8; ++j) {
                    const uint32_t k = w * 8 + j;
                    const double r1 = 6.283185307179586 * rng_bits_to_u01(k << 9);
                    double want[2];  // glibc sincos(): what g++ -O2 emits for the reference's cos(r1), sin(r1)
                    ::sincos(r1, &want[0], &want[1]);
                    const double have[2] = {hs[k], hc[k]};
                    for (int f = 0; f < 2; ++f) {
                        long long a, b;
                        std::memcpy(&a, &want[f], 8);
                        std::memcpy(&b, &have[f], 8);
                        const long long d = a - b;
                        if (d < -1 || d > 1) out_of_range++;
                        packed |= ((uint32_t)(d & 3)) << (j * 4 + f * 2);
                    }
                }
                table[w] = packed;
            }
        });
    for (auto& th : pool) th.join();
    if (out_of_range.load() != 0) {
        set_last_error("host and device sin/cos differ by more than one ulp somewhere: RTM_MODE_HOST_TRIG unavailable");
        return RTM_ERR_UNSUPPORTED;
    }
    DevMem d_table;
    const int rc = d_table.alloc((size_t)words * 4);
    if (rc != RTM_OK) return rc;
    RTM_HIP_CHECK(hipMemcpy(d_table.p, table.data(), (size_t)words * 4, hipMemcpyHostToDevice));
    ready[device] = d_table.as<uint32_t>();
    *out = d_table.as<uint32_t>();
    d_table.p = nullptr;  // owned by g_trig_ready until release_scratch
    return RTM_OK;
}

// Rows a call renders and stores: the whole strip, or its bands band_index, band_index + band_count, ...
// (8 rows each; the last band of the strip may be shorter).
int output_rows(const rtm_options* opt) {
    const int span = opt->row_end - opt->row_begin;
    if (span <= 0) return 0;
    if (opt->band_count <= 1) return span;
    const int bands = (span + 7) / 8;
    if (opt->band_index < 0 || opt->band_index >= bands) return 0;
    const int mine = (bands - opt->band_index + opt->band_count - 1) / opt->band_count;
    int rows = mine * 8;
    const int last = opt->band_index + (mine - 1) * opt->band_count;
    if (last == bands - 1 && span % 8) rows -= 8 - span % 8;
    return rows;
}

static int validate(const rtm_settings* st, const rtm_sphere* sp, size_t n, const rtm_options* opt) {
    if (!st || !opt || (!sp && n)) {
        set_last_error("null argument");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if (st->width <= 0 || st->height <= 0 || st->samples <= 0 || st->super_samples <= 0) {
        set_last_error("width, height, samples and superSamples must be positive");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if (opt->row_begin < 0 || opt->row_end > st->height || opt->row_begin > opt->row_end) {
        set_last_error("row range outside the image");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if (opt->band_count < 0 || (opt->band_count > 1 && (opt->band_index < 0 || opt->band_index >= opt->band_count))) {
        set_last_error("band_index outside [0, band_count)");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if ((opt->mode & ~kModeFlags) != RTM_MODE_LITERAL && (opt->mode & ~kModeFlags) != RTM_MODE_REPAIRED) {
        set_last_error("unknown mode");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if ((uint64_t)st->super_samples * st->super_samples * st->samples > 0xFFFFFFFFull ||
        (uint64_t)st->width * st->height > 0xFFFFFFFFull || n > 0x7FFFFFFFull) {
        set_last_error("image, sample or scene count exceeds 32-bit indexing");
        return RTM_ERR_UNSUPPORTED;
    }
    if (opt->max_bounces > 16 + kPoolLevels) {
        set_last_error("max_bounces exceeds the hit-record capacity (976); use -1 for unlimited");
        return RTM_ERR_UNSUPPORTED;
    }
    if (opt->variant < 0 || opt->variant >= num_variants()) {
        set_last_error("unknown kernel variant");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    return RTM_OK;
}

// variant 0 = auto (the fastest parity-validated kernel for the scene size)
// Retired numbers keep their place (profiles of rounds 1-2 refer to them) and are refused with RTM_ERR_UNSUPPORTED; the
// kernels behind them live in profiles/r3/retired_variants.patch, with the A/B results that retired them.
#define RTM_RETIRED(what) "retired: " what " (profiles/r3/retired_variants.patch)"
static const char* kVariantNames[] = {"auto", "ref-math-global-scene", "fast-math-lds-tables-chunk8-park-pack8",
                                      "fast-math-global-scene-chunk8", RTM_RETIRED("monolithic LDS scene tiles"),
                                      RTM_RETIRED("variant 2 without the LDS-resident accumulator"),
                                      RTM_RETIRED("variant 2 with LDS record stacks"),
                                      "diagnostic-stamped (segment cycle shares, not for timing)",
                                      RTM_RETIRED("wavefront pipeline, LDS scene tiles"),
                                      "fast-math-lds-tables-chunk8-park-pack8-sample-split",
                                      RTM_RETIRED("wavefront pipeline, scalar scene stream without a rejection test"),
                                      RTM_RETIRED("wavefront pipeline, fp64 rejection test"),
                                      "wavefront-scalar-scene-reject-f32",
                                      RTM_RETIRED("variant 2 with the immediate fold"),
                                      "fast-math-global-scene-chunk8-park-pack8",
                                      ("LABELLED-primary-hit-reuse (one nearest-hit search per sub-pixel for its S primary rays; "
                                       "not the reference's work per sample)"),
                                      ("LABELLED-fp32-fast (single precision, hardware sqrt/rsq/sin/cos, fused multiply-adds, "
                                       "forward throughput: NOT a parity path, reported with its out-of-tolerance pixel fraction)"),
                                      ("fast-math-uniform-grid (large scenes held by an rtm_scene: the reference loop's nearest hit "
                                       "through a uniform grid over the spheres, same image bit for bit)"),
                                      ("fp64-tolerance (the default kernel's source compiled with FMA contraction and one-ulp division / "
                                       "square root; float islands, RNG, thresholds, fold and addition order kept, primary-ray ties settled "
                                       "in the reference's arithmetic: within north_star's 1e-4 per pixel by test, not bit-exact by "
                                       "construction; reported with its differing-pixel count)"),
                                      ("surface-sample-integrator (png::SurfaeSample, the reference's second integrator — never selected by "
                                       "its Render —, through a general per-object kernel with the compiler's math; chosen by "
                                       "RTM_MODE_SURFACE_SAMPLE, not by number)")};
#undef RTM_RETIRED
constexpr int kVariantAuto = 0, kVariantRef = 1, kVariantFastLds = 2, kVariantFastGlobal = 3, kVariantStamped = 7,
              kVariantSplit = 9, kVariantWavefrontRejectF32 = 12, kVariantGlobalDefer = 14, kVariantPrimaryReuse = 15,
              kVariantFp32 = 16, kVariantGrid = 17, kVariantTol = 18, kVariantSurface = 19;
static bool variant_retired(int v) { return v == 4 || v == 5 || v == 6 || v == 8 || v == 10 || v == 11 || v == 13; }
constexpr int kLdsTableMaxSpheres = 256;  // 96 B per sphere of LDS: 24 KiB at the cap
constexpr int kAutoLdsTableSpheres = 24, kAutoWavefrontSpheres = 512;
int num_variants() { return (int)(sizeof(kVariantNames) / sizeof(kVariantNames[0])); }
const char* variant_name(int v) { return (v >= 0 && v < num_variants()) ? kVariantNames[v] : nullptr; }

static void fill_render_params(RenderParams& P, const rtm_settings* st, const rtm_options* opt) {
    using namespace host;
    const H3 origin = {st->camera.origin[0], st->camera.origin[1], st->camera.origin[2]};
    const H3 target = {st->camera.target[0], st->camera.target[1], st->camera.target[2]};
    const H3 up = {st->camera.up[0], st->camera.up[1], st->camera.up[2]};
    const H3 direction = normalize(sub(target, origin));  // src/Renderer.cpp:202
    const H3 c0 = normalize(cross(direction, up));        // :203
    const H3 cam_x = {-c0.x, -c0.y, -c0.z};
    const H3 cam_y = cross(cam_x, direction);  // :204
    const double fovx = (double)st->camera.fov;
    const double fovy = fovx * st->height / st->width;  // :208
    P.W = st->width;
    P.H = st->height;
    P.S = st->samples;
    P.SS = st->super_samples;
    P.row_begin = opt->row_begin;
    P.row_end = opt->row_end;
    P.band_count = opt->band_count > 1 ? opt->band_count : 1;
    P.band_index = opt->band_count > 1 ? opt->band_index : 0;
    P.tiles_x = (st->width + 7) / 8;
    P.mode = opt->mode & ~kModeFlags;
    P.max_bounces = opt->max_bounces;
    P.total_samples = (unsigned)st->super_samples * st->super_samples * st->samples;
    P.rate = (float)(1.0 / (1 + st->super_samples));
    P.dSS = (double)st->super_samples;
    P.dS = (double)st->samples;
    auto is_pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (opt->variant != kVariantRef && is_pow2(st->samples) && is_pow2(st->super_samples)) {
        P.inv_ss = 1.0 / P.dSS;  // exact
        P.inv_s = 1.0 / P.dS;
    }
    P.cam_org = D3{origin.x, origin.y, origin.z};
    P.ax = D3{cam_x.x * fovx, cam_x.y * fovx, cam_x.z * fovx};  // l_camX * fovx, :229
    P.by = D3{cam_y.x * fovy, cam_y.y * fovy, cam_y.z * fovy};  // l_camY * fovy, :230
    P.cz = D3{direction.x, direction.y, direction.z};
    P.seed_mult = seed_multiplier(opt->seed);
}

// LDS record levels: 16 when the cap guarantees depth < 16; otherwise 64 (u8) / 32 (u32) levels in
// LDS plus the global pool for the rare deeper path.
template <typename RecT>
constexpr int deep_lds_levels() { return sizeof(RecT) == 1 ? 64 : 32; }
static bool needs_pool(const RenderParams& P) { return !(P.max_bounces >= 0 && P.max_bounces < 16); }

// diagnostic knob (profiles/occupancy_sweep.sh): extra dynamic LDS per workgroup caps the waves/CU
static size_t debug_lds_pad() {
    static const size_t pad = [] {
        const char* e = std::getenv("RTM_DEBUG_LDS_PAD");
        return e ? (size_t)std::strtoul(e, nullptr, 10) : (size_t)0;
    }();
    return pad;
}

// Scenes that hold planes (render_view: up to kLdsTableMaxSpheres objects): the deferred-fold kernels with the object
// chunk, packed records for a depth cap of at most 8 (PACK8), by position for any depth (PACKL)
constexpr size_t kUnitTableBytes = (size_t)(kShadeConstCount - kTrigConstCount) * sizeof(double);  // rtm_device.h: the near-unit Normalize table
// The launch's LDS with the near-unit Normalize table where it does not cost a wave per CU (P.unit_tab tells the kernel)
static size_t with_unit_table(RenderParams& P, size_t lds) {
    P.unit_tab = unit_table_fits(lds) ? 1u : 0u;
    return lds + (P.unit_tab ? kUnitTableBytes : 0);
}
template <bool SPLIT>
static void launch_render_planes(const RenderParams& P_in, unsigned grid, hipStream_t stream) {
    RenderParams P = P_in;
    const size_t tab = lds_table_bytes(P.scene.n) + (10 + kTrigConstCount) * sizeof(double) + 6 * 64 * sizeof(double) + debug_lds_pad();
    constexpr size_t tag = SPLIT ? kFoldTagBytes : 0;
    if (P.max_bounces >= 0 && P.max_bounces <= 8) {
        const size_t lds = with_unit_table(P, tab + kFoldQueueBytesS + tag);
        render_tiles_kernel<MathFast, true, 8, uint8_t, 16, 4, true, false, true, SPLIT, true, false, false, true>
            <<<grid, 64, lds, stream>>>(P);
    } else {
        const size_t lds = with_unit_table(P, tab + kFoldQueueBytesLS + tag);
        render_tiles_kernel<MathFast, true, 8, uint8_t, 0, 4, true, false, false, SPLIT, true, true, false, true>
            <<<grid, 64, lds, stream>>>(P);
    }
}

constexpr size_t kStealLdsBytes = 2 * 64 * sizeof(unsigned);  // STEAL: every pixel's next own sample and own-sample end
template <class M, bool LDS_TAB, int UNROLL, typename RecT, int WPE = 1, bool PARK = false, bool STAMP = false,
          bool TRY_PACK8 = false, bool SPLIT = false, bool DEFER = false, bool STEAL = false>
static void launch_render_depth(const RenderParams& P_in, unsigned grid, hipStream_t stream) {
    RenderParams P = P_in;  // (unit_tab: the near-unit Normalize table where the launch's LDS has the room, with_unit_table)
    const size_t tab = (LDS_TAB ? lds_table_bytes(P.scene.n) : 0) + (10 + kTrigConstCount) * sizeof(double) +
                       (PARK ? 6 * 64 * sizeof(double) : 0) + debug_lds_pad() + (SPLIT ? kFoldTagBytes : 0);
    constexpr int DEEP = deep_lds_levels<RecT>();
    if constexpr (TRY_PACK8 && sizeof(RecT) == 1) {
        if (P.max_bounces >= 0 && P.max_bounces <= 8 && P.scene.n < 256) {  // ids and the identity index in a byte
            if constexpr (DEFER && UNROLL == -8) {
                // scenes under 8 spheres (every shipped scene): the instantiation for exactly n spheres
                const size_t lds = with_unit_table(P, tab + kFoldQueueBytesS + (STEAL ? kStealLdsBytes : 0));
#if RTM_OPT_AXIS
#define RTM_AXIS_CASE(k, sig)                                                                                                  \
    if (P.scene.n == k && P.scene.axis_pat == sig && P.mode == RTM_MODE_REPAIRED && P.unit_tab == 1u) { /* rtm_path.h: sphere_disc */                        \
        render_tiles_kernel<M, LDS_TAB, axis_unroll(k, sig), RecT, 16, WPE, PARK, STAMP, true, SPLIT, true, false, false, false, STEAL> \
            <<<grid, 64, lds, stream>>>(P);                                                                                    \
        return;                                                                                                                \
    }
                RTM_AXIS_SIGNATURES(RTM_AXIS_CASE)
#undef RTM_AXIS_CASE
#endif
                switch (P.scene.n) {
#define RTM_EXACT_N(k)                                                                                     \
    case k:                                                                                                \
        render_tiles_kernel<M, LDS_TAB, -100 - k, RecT, 16, WPE, PARK, STAMP, true, SPLIT, true, false, false, false, STEAL> \
            <<<grid, 64, lds, stream>>>(P);                                                                \
        return;
                    RTM_EXACT_N(1) RTM_EXACT_N(2) RTM_EXACT_N(3) RTM_EXACT_N(4) RTM_EXACT_N(5) RTM_EXACT_N(6) RTM_EXACT_N(7)
#undef RTM_EXACT_N
                    default: break;
                }
            }
            if constexpr (DEFER) {
                const size_t lds = with_unit_table(P, tab + kFoldQueueBytesS + (STEAL ? kStealLdsBytes : 0));
                render_tiles_kernel<M, LDS_TAB, UNROLL, RecT, 16, WPE, PARK, STAMP, true, SPLIT, true, false, false, false, STEAL>
                    <<<grid, 64, lds, stream>>>(P);
            } else {
                const size_t lds = with_unit_table(P, tab);
                render_tiles_kernel<M, LDS_TAB, UNROLL, RecT, 16, WPE, PARK, STAMP, true, false><<<grid, 64, lds, stream>>>(P);  // (never split: the split rides on the fold queue)
            }
            return;
        }
    }
    if constexpr (DEFER && sizeof(RecT) == 1) {
        if (P.scene.n < 256) {  // any depth: packed records + pooled stack, deferred fold
            const size_t lds = with_unit_table(P, tab + kFoldQueueBytesLS);  // (8 280 bytes for a 7-sphere scene; 10 072 with the FIFO form: no room for the table then)
#if RTM_OPT_AXIS
            if constexpr (UNROLL == -8) {
#define RTM_AXIS_CASE(k, sig)                                                                                              \
    if (P.scene.n == k && P.scene.axis_pat == sig && P.mode == RTM_MODE_REPAIRED && P.unit_tab == 1u) { /* rtm_path.h: sphere_disc */                    \
        render_tiles_kernel<M, LDS_TAB, axis_unroll(k, sig), RecT, 0, WPE, PARK, STAMP, false, SPLIT, true, true>           \
            <<<grid, 64, lds, stream>>>(P);                                                                                \
        return;                                                                                                            \
    }
                RTM_AXIS_SIGNATURES(RTM_AXIS_CASE)
#undef RTM_AXIS_CASE
            }
#endif
            render_tiles_kernel<M, LDS_TAB, UNROLL, RecT, 0, WPE, PARK, STAMP, false, SPLIT, true, true>
                <<<grid, 64, lds, stream>>>(P);
            return;
        }
    }
    if (!needs_pool(P)) {
        const size_t lds = with_unit_table(P, tab + 16 * 64 * sizeof(RecT));
        render_tiles_kernel<M, LDS_TAB, UNROLL, RecT, 16, WPE, PARK, STAMP, false, false><<<grid, 64, lds, stream>>>(P);
    } else {
        const size_t lds = with_unit_table(P, tab + DEEP * 64 * sizeof(RecT));
        render_tiles_kernel<M, LDS_TAB, UNROLL, RecT, DEEP, WPE, PARK, STAMP, false, false><<<grid, 64, lds, stream>>>(P);
    }
}

// `variant` is resolved (render_view): one of ref, fast-lds, fast-global, stamped, global-defer, primary-reuse, fp32.
static void launch_render(int variant, const RenderParams& P, unsigned grid, hipStream_t stream) {
    const int n = P.scene.n;
    const unsigned split_grid = P.split_first + P.n_tiles * P.split;
    if (variant == kVariantFastLds && n > kLdsTableMaxSpheres) variant = kVariantFastGlobal;
    switch (variant) {
        case kVariantRef:
            if (n <= 256) launch_render_depth<MathRef, false, 1, uint8_t>(P, grid, stream);
            else launch_render_depth<MathRef, false, 1, uint32_t>(P, grid, stream);
            return;
        case kVariantFastLds:  // (n < 8: no full chunk of 8, the instantiation without the chunk loop)
            if (P.scene.plane != nullptr) {  // render_view: n <= kLdsTableMaxSpheres (ids and the identity index in a byte: n < 256)
                if (P.split > 1) {
                    launch_render_planes<true>(P, split_grid, stream);
                    split_finalize_kernel<<<P.n_tiles, 256, (size_t)P.split_len * 64 * 3 * sizeof(double), stream>>>(P);
                } else {
                    launch_render_planes<false>(P, grid, stream);
                }
                return;
            }
            if (P.steal_ws != nullptr) {  // render_view: depth cap <= 8; the whole tiles balance their lanes by sample stealing
                if (P.split > 1) {
                    if (n < 8) launch_render_depth<MathFast, true, -8, uint8_t, 4, true, false, true, true, true, true>(P, split_grid, stream);
                    else launch_render_depth<MathFast, true, 8, uint8_t, 4, true, false, true, true, true, true>(P, split_grid, stream);
                    split_finalize_kernel<<<P.n_tiles, 256, (size_t)P.split_len * 64 * 3 * sizeof(double), stream>>>(P);
                } else if (n < 8) {
                    launch_render_depth<MathFast, true, -8, uint8_t, 4, true, false, true, false, true, true>(P, grid, stream);
                } else {
                    launch_render_depth<MathFast, true, 8, uint8_t, 4, true, false, true, false, true, true>(P, grid, stream);
                }
                const unsigned n_whole = P.split > 1 ? P.split_first : grid;
                if (n_whole) steal_finalize_kernel<<<n_whole, 64, (size_t)P.steal_depth * 64 * sizeof(unsigned short), stream>>>(P);
                return;
            }
            if (P.split > 1) {
                if (n < 8) launch_render_depth<MathFast, true, -8, uint8_t, 4, true, false, true, true, true>(P, split_grid, stream);
                else launch_render_depth<MathFast, true, 8, uint8_t, 4, true, false, true, true, true>(P, split_grid, stream);
                split_finalize_kernel<<<P.n_tiles, 256, (size_t)P.split_len * 64 * 3 * sizeof(double), stream>>>(P);
            } else if (n < 8) {
                launch_render_depth<MathFast, true, -8, uint8_t, 4, true, false, true, false, true>(P, grid, stream);
            } else {
                launch_render_depth<MathFast, true, 8, uint8_t, 4, true, false, true, false, true>(P, grid, stream);
            }
            return;
        case kVariantGlobalDefer:  // render_view: n < 256
            if (P.split > 1) {
                launch_render_depth<MathFast, false, 8, uint8_t, 4, true, false, true, true, true>(P, split_grid, stream);
                split_finalize_kernel<<<P.n_tiles, 256, (size_t)P.split_len * 64 * 3 * sizeof(double), stream>>>(P);
            } else {
                launch_render_depth<MathFast, false, 8, uint8_t, 4, true, false, true, false, true>(P, grid, stream);
            }
            return;
        case kVariantFp32: {  // validated by render_view: repaired mode, 1 <= n <= 256
            const size_t lds = (((size_t)n * kFp32Row * sizeof(float) + 15) & ~(size_t)15) + 10 * sizeof(double);
            render_fp32_kernel<<<grid, 64, lds, stream>>>(P);
            return;
        }
        case kVariantPrimaryReuse: {  // validated by render_view: 1 <= n <= 24, 0 <= max_bounces <= 8
            RenderParams Q = P;
            const size_t lds = with_unit_table(Q, lds_table_bytes(n) + (10 + kTrigConstCount) * sizeof(double) + 6 * 64 * sizeof(double) +
                                                      kFoldQueueBytes + debug_lds_pad());
            if (n < 8)
                render_tiles_kernel<MathFast, true, -8, uint8_t, 16, 4, true, false, true, false, true, false, true>
                    <<<grid, 64, lds, stream>>>(Q);
            else
                render_tiles_kernel<MathFast, true, 8, uint8_t, 16, 4, true, false, true, false, true, false, true>
                    <<<grid, 64, lds, stream>>>(Q);
            return;
        }
        case kVariantStamped:  // render_view: n <= kLdsTableMaxSpheres
            launch_render_depth<MathFast, true, 8, uint8_t, 4, true, true, true>(P, grid, stream);
            return;
        default:  // kVariantFastGlobal: the chunked kernel with global-memory tables, any n
            if (n <= 256) launch_render_depth<MathFast, false, 8, uint8_t, 4>(P, grid, stream);
            else launch_render_depth<MathFast, false, 8, uint32_t, 4>(P, grid, stream);
    }
}

// ---- per-(device, stream) context -----------------------------------------------------------------
// Large work buffers (the sample split's term buffer, the pooled record stacks, the wavefront state) are kept
// per (device, stream, role) and grown on demand: calls on one stream are ordered by the stream, so the
// buffer of the previous call can be reused without any wait, and different streams get different buffers.
// Allocating them stream-ordered per call instead let the pool hand out FRESH multi-GB blocks whenever the
// host ran ahead of the GPU (a 2.4 GB first touch per step: 10 -> 40 ms for a 512x512 unlimited-depth frame,
// intermittently).  release_scratch() frees them.
//
// The context also owns the stream's STICKY status word: renders without rtm_stats count into `sticky`
// (casts, bounces, draws — ignored — and the overflow flag, which only ever gets OR-ed), every render ends
// with a stream-ordered copy of the flag into a pinned host word, and the next call / rtm_stream_status
// report it.  Threading (include/rtm.h): `mu` is held while a call enqueues its work, so two host threads
// naming the same (device, stream) take turns and never see each other's half-grown buffers; `g_gate` is
// held shared by every render and exclusively by release_scratch, which therefore never frees anything
// under a call that is between acquiring a buffer and launching on it.
enum { kScratchTerms = 0, kScratchPool = 1, kScratchWavefront = 2, kScratchSteal = 3, kScratchPrim = 4, kScratchRoles = 5 };
namespace {
struct ScratchBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
};
struct StreamCtx {
    std::mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    ScratchBuf scratch[kScratchRoles];
    unsigned long long* sticky = nullptr;    // device: casts, bounces, draws (ignored), overflow flag
    unsigned long long* counters = nullptr;  // device: the same four words for a call WITH stats
    unsigned long long* flag_host = nullptr; // pinned: last copy of sticky[3]
    unsigned* wf_count_host = nullptr;       // pinned: [2] active-pixel counts of the large-scene pipeline
    bool ready = false;
    int init() {
        if (ready) return RTM_OK;
        RTM_HIP_CHECK(hipMalloc((void**)&sticky, 16 * sizeof(unsigned long long)));
        counters = sticky + 8;  // casts, bounces, draws, overflow flag, object tests (RTM_MODE_COUNT_TESTS)
        RTM_HIP_CHECK(hipMemset(sticky, 0, 16 * sizeof(unsigned long long)));
        RTM_HIP_CHECK(hipHostMalloc((void**)&flag_host, 64, hipHostMallocDefault));
        std::memset(flag_host, 0, 64);
        wf_count_host = reinterpret_cast<unsigned*>(flag_host + 2);
        ready = true;
        return RTM_OK;
    }
    void free_all() {  // caller: device idle, nobody else in the context
        for (auto& b : scratch) {
            if (b.ptr) (void)hipFree(b.ptr);
            b = ScratchBuf{};
        }
        if (sticky) (void)hipFree(sticky);
        if (flag_host) (void)hipHostFree(flag_host);
        sticky = counters = flag_host = nullptr;
        wf_count_host = nullptr;
        ready = false;
    }
};
std::shared_mutex g_gate;  // shared: a render is being enqueued; exclusive: release_scratch
std::mutex g_ctx_mu;
std::map<std::pair<int, hipStream_t>, std::unique_ptr<StreamCtx>> g_ctx;

StreamCtx* get_ctx(int device, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    auto& slot = g_ctx[{device, stream}];
    if (!slot) {
        slot.reset(new StreamCtx);
        slot->device = device;
        slot->stream = stream;
    }
    return slot.get();
}

// Scene cache of rtm_render_device(host array): content-addressed, a few entries per process.
// Entries are shared: a render keeps its scene alive while it enqueues, and whoever drops the last reference
// (an eviction, release_scratch, or that render) waits for the device before the tables are freed.
struct SceneCache {
    std::mutex mu;
    std::vector<std::shared_ptr<rtm_scene>> entries;  // most recently used last
};
// a heap object that is never destroyed: its entries' deleter calls HIP, which must not happen from a static destructor
// at process exit (the runtime may be gone by then)
SceneCache& scene_cache() {
    static SceneCache* c = new SceneCache;
    return *c;
}
#define g_scene_cache scene_cache()
constexpr size_t kSceneCacheEntries = 8;

uint64_t hash_bytes(const void* data, size_t bytes) {  // FNV-1a over 8-byte words (+ tail bytes)
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint64_t h = 1469598103934665603ull;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 1099511628211ull;
    }
    for (; i < bytes; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}
}  // namespace

// Caller holds ctx.mu.  Growing waits for the stream (queued work may still use the smaller buffer).
static int scratch_acquire(StreamCtx& ctx, int role, size_t bytes, void** out) {
    ScratchBuf& b = ctx.scratch[role];
    if (b.bytes < bytes) {
        if (b.ptr) {
            RTM_HIP_CHECK(hipStreamSynchronize(ctx.stream));
            (void)hipFree(b.ptr);
            b.ptr = nullptr;
            b.bytes = 0;
        }
        RTM_HIP_CHECK(hipMalloc(&b.ptr, bytes));
        b.bytes = bytes;
    }
    *out = b.ptr;
    return RTM_OK;
}

// The cached scene for a host array (uploading it on a miss).
static int cached_scene(const rtm_sphere* sp, size_t n, int device, std::shared_ptr<rtm_scene>* out) {
    const uint64_t h = hash_bytes(sp, n * sizeof(rtm_sphere));
    std::lock_guard<std::mutex> lock(g_scene_cache.mu);
    auto& e = g_scene_cache.entries;
    for (size_t i = 0; i < e.size(); ++i)
        if (e[i]->device == device && e[i]->n == n && e[i]->content_hash == h &&
            e[i]->content.size() == n * sizeof(rtm_sphere) && std::memcmp(e[i]->content.data(), sp, n * sizeof(rtm_sphere)) == 0) {
            if (i + 1 != e.size()) std::rotate(e.begin() + (long)i, e.begin() + (long)i + 1, e.end());
            *out = e.back();
            return RTM_OK;
        }
    std::shared_ptr<rtm_scene> sc(new rtm_scene, [](rtm_scene* p) { (void)scene_destroy(p); });
    const int rc = scene_build_host(*sc, sp, n, device);
    if (rc != RTM_OK) return rc;
    sc->content_hash = h;
    sc->content.assign(reinterpret_cast<const unsigned char*>(sp), reinterpret_cast<const unsigned char*>(sp) + n * sizeof(rtm_sphere));
    if (e.size() >= kSceneCacheEntries) e.erase(e.begin());  // (the deleter parks it if a render from it is still queued)
    e.push_back(sc);
    *out = sc;
    return RTM_OK;
}

// Content hash of a device array, order-independent over (index, word) pairs: sum of smfin64(word + (index + 1) * odd).
__global__ __launch_bounds__(256) void hash_words_kernel(const unsigned long long* __restrict__ w, size_t n_words,
                                                         unsigned long long* __restrict__ out) {
    unsigned long long acc = 0ull;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256)
        acc += smfin64(w[i] + (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
namespace {
struct DeviceArrayCache {
    struct Entry {
        int device;
        size_t n;
        unsigned long long hash;
        std::shared_ptr<rtm_scene> scene;
    };
    std::mutex mu;
    std::vector<Entry> entries;  // most recently used last
};
DeviceArrayCache& device_array_cache() {  // (never destroyed: its scenes' deleter calls HIP)
    static DeviceArrayCache* c = new DeviceArrayCache;
    return *c;
}
}  // namespace
// The cached scene object for a DEVICE sphere array (render_device).  Waits for `stream` once (the hash).
static int cached_device_scene(const rtm_sphere* sp_dev, size_t n, int device, hipStream_t stream, std::shared_ptr<rtm_scene>* out) {
    unsigned long long h = 0ull;
    {
        StreamCtx& ctx = *get_ctx(device, stream);
        std::lock_guard<std::mutex> lock(ctx.mu);
        int rc = ctx.init();
        if (rc != RTM_OK) return rc;
        unsigned long long* slot = ctx.sticky + 7;                     // device word of the context nobody else uses
        volatile unsigned long long* host = ctx.flag_host + 4;         // pinned
        static_assert(sizeof(rtm_sphere) % 8 == 0, "the array is hashed in 8-byte words");
        const size_t words = n * sizeof(rtm_sphere) / 8;
        RTM_HIP_CHECK(hipMemsetAsync(slot, 0, 8, stream));
        const unsigned blocks = (unsigned)std::min<size_t>(1024, (words + 255) / 256);
        hash_words_kernel<<<blocks, 256, 0, stream>>>(reinterpret_cast<const unsigned long long*>(sp_dev), words, slot);
        RTM_HIP_CHECK(hipGetLastError());
        RTM_HIP_CHECK(hipMemcpyAsync((void*)host, slot, 8, hipMemcpyDeviceToHost, stream));
        RTM_HIP_CHECK(hipStreamSynchronize(stream));
        h = *host;
    }
    DeviceArrayCache& cache = device_array_cache();
    std::lock_guard<std::mutex> lock(cache.mu);
    auto& e = cache.entries;
    for (size_t i = 0; i < e.size(); ++i)
        if (e[i].device == device && e[i].n == n && e[i].hash == h) {
            if (i + 1 != e.size()) std::rotate(e.begin() + (long)i, e.begin() + (long)i + 1, e.end());
            *out = e.back().scene;
            return RTM_OK;
        }
    std::vector<rtm_sphere> host_copy(n);
    RTM_HIP_CHECK(hipMemcpy(host_copy.data(), sp_dev, n * sizeof(rtm_sphere), hipMemcpyDeviceToHost));
    std::shared_ptr<rtm_scene> sc(new rtm_scene, [](rtm_scene* p) { (void)scene_destroy(p); });
    const int rc = scene_build_host(*sc, host_copy.data(), n, device);
    if (rc != RTM_OK) return rc;
    if (e.size() >= kSceneCacheEntries) e.erase(e.begin());  // (the deleter parks it if a render from it is still queued)
    e.push_back(DeviceArrayCache::Entry{device, n, h, sc});
    *out = sc;
    return RTM_OK;
}

int release_scratch(int device) {
    std::unique_lock<std::shared_mutex> gate(g_gate);  // no render is mid-call from here on
    {
        std::lock_guard<std::mutex> lock(g_trig_mu);
        for (auto it = g_trig_ready.begin(); it != g_trig_ready.end();) {
            if (device < 0 || it->first == device) {
                (void)hipSetDevice(it->first);
                (void)hipDeviceSynchronize();
                (void)hipFree(it->second);
                it = g_trig_ready.erase(it);
            } else {
                ++it;
            }
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_scene_cache.mu);
        auto& e = g_scene_cache.entries;
        for (auto it = e.begin(); it != e.end();) {
            if (device < 0 || (*it)->device == device) {
                it = e.erase(it);  // parked if a render from it is still queued; reaped below, after the wait
            } else {
                ++it;
            }
        }
    }
    {
        DeviceArrayCache& cache = device_array_cache();
        std::lock_guard<std::mutex> lock(cache.mu);
        for (auto it = cache.entries.begin(); it != cache.entries.end();) {
            if (device < 0 || it->device == device) it = cache.entries.erase(it);
            else ++it;
        }
    }
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    for (auto it = g_ctx.begin(); it != g_ctx.end();) {
        if (device < 0 || it->first.first == device) {
            (void)hipSetDevice(it->first.first);
            (void)hipDeviceSynchronize();
            it->second->free_all();
            it = g_ctx.erase(it);
        } else {
            ++it;
        }
    }
    reap_scenes(true, device);  // scenes destroyed while renders from them were queued: the device is idle now
    (void)hipGetLastError();
    return RTM_OK;
}

// Sample split of the default kernel.  A launch ends with SIMDs falling idle one after another while the last
// tiles finish (a tile is ~22 ms of one wave on a full chip, and max/mean per-tile cost on the Cornell box is
// ~1.27, profiles/r1/strip_balance.txt): 4-5 % of the 7.9-round headline launch, a quarter of a one-round launch
// (one GPU's share of eight).  The LAST tiles of every launch are therefore cut finer — wave 0 of such a tile
// takes half of its samples, small waves 1/g of them each (g = 16: profiles/r1/band_split_sweep_final.json) —
// and, coming last in the grid, fill that ramp.  Only the last ones: a small wave pays the kernel's prologue and
// its own ragged end (lanes waiting for the slowest of 64) once per 32 samples instead of once per 1024, so
// splitting EVERY tile of a launch of 2-4 rounds, round 1's rule, cost 8 % against splitting 1 536 of them
// (profiles/r2/tail_split.txt).  Launches of up to 1 536 tiles are split whole, as before.
struct SplitPlan {
    unsigned g;      // granularity: small waves trace 1/g of a pixel's samples (1: no split)
    unsigned tiles;  // how many of the launch's LAST tiles are split
    unsigned head;   // samples wave 0 of a split tile keeps (a multiple of total / g)
};
constexpr unsigned kSplitMaxLen = 64;  // samples per small wave: split_finalize_kernel sorts one small wave's terms in LDS (1.5 KB per sample)
static SplitPlan choose_split(unsigned n_tiles, unsigned total_samples, int device, bool forced) {
    static const long env = [] {
        const char* e = std::getenv("RTM_DEBUG_SPLIT");  // tuning knob: 1 = never split, g = every tile with that granularity
        return e ? std::strtol(e, nullptr, 10) : 0L;
    }();
    static const long env_tail = [] {
        const char* e = std::getenv("RTM_DEBUG_TAIL");  // tuning knob: number of split tiles (0 = none)
        return e ? std::strtol(e, nullptr, 10) : -1L;
    }();
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const unsigned slots = (unsigned)cus * 16u;
    const SplitPlan none{1u, 0u, total_samples};
    // wave 0 keeps about half of the samples (its terms never leave the chip, and it starts in the launch's first rounds,
    // so the small waves still set the tail: profiles/r2/tail_split.txt); an odd g: the larger half
    static const long env_head = [] {
        const char* e = std::getenv("RTM_DEBUG_HEAD");  // tuning knob: wave 0 keeps this many of g shares (default: half)
        return e ? std::strtol(e, nullptr, 10) : 0L;
    }();
    auto head_of = [&](unsigned g) {
        if (env_head > 0 && (unsigned)env_head < g) return (total_samples / g) * (unsigned)env_head;
        // 9 of 16 shares: as fast as 8 (headline 170.8 / 170.6 ms, 512x512x256spp 6.68 / 6.72, one GPU's share of eight equal),
        // an eighth fewer terms to store; 10 of 16 costs the headline frame 0.5 % (profiles/r3/head_ab.txt)
        if (g % 16u == 0u) return (total_samples / g) * (9u * (g / 16u));
        return g >= 4 ? (total_samples / g) * ((g + 1u) / 2u) : total_samples / g;
    };
    auto valid_g = [&](unsigned g) {
        return g >= 2 && total_samples % g == 0 && total_samples / g >= 8 && total_samples / g <= kSplitMaxLen;
    };
    // a small wave traces 1/16 of a pixel's samples (profiles/r1/band_split_sweep_final.json), at least 8 and at most
    // kSplitMaxLen of them: the divisor of the sample count nearest to that from below, else from above
    auto pick_g = [&]() -> unsigned {
        unsigned want = total_samples / 16u;
        want = want < 8u ? 8u : (want > kSplitMaxLen ? kSplitMaxLen : want);
        for (unsigned len = want; len >= 8u; --len)
            if (total_samples % len == 0 && valid_g(total_samples / len)) return total_samples / len;
        for (unsigned len = want + 1u; len <= kSplitMaxLen; ++len)
            if (total_samples % len == 0 && valid_g(total_samples / len)) return total_samples / len;
        return 1u;
    };
    // the terms of the split tiles: a 1 664-byte row per (tile, deferred sample), at most 24 GiB per stream
    auto fits = [&](unsigned g, unsigned tiles) {
        return (double)tiles * (double)(total_samples - head_of(g)) * (double)kTermRowBytes <= 24.0 * 1024 * 1024 * 1024;
    };
    if (env > 0) {
        const unsigned g = (unsigned)env;
        return (env > 1 && valid_g(g) && fits(g, n_tiles)) ? SplitPlan{g, n_tiles, head_of(g)} : none;
    }
    const unsigned g = pick_g();
    if (g <= 1u) return none;
    // the last 3/8 of a round of tiles (1 536 on this chip; profiles/r2/tail_split.txt).  1 024 would keep the launch's term
    // traffic under 2 GB (profiles/r3/term_store_modes.txt) but costs the headline frame 0.5 % and a one-round launch like
    // 512 x 512 x 256 spp 17 % (8.0 against 6.9 ms: there the slowest WHOLE tile sets the time; profiles/r3/ab_r2_vs_r3.txt)
    // (Re-checked with in-wave sample stealing in place, profiles/r3/tail_after_steal.txt: still the best single value — a
    // longer tail helps some contiguous strips, the frame's top rows by 12 %, and costs the interleaved-band parts of a
    // multi-GPU frame 4 %.)
    unsigned tail = forced ? n_tiles : env_tail >= 0 ? (unsigned)env_tail : 3u * slots / 8u;
    if (tail > n_tiles) tail = n_tiles;
    while (tail > 64u && !forced && !fits(g, tail)) tail /= 2u;  // very many samples per pixel: a shorter tail whose terms fit
    return (tail > 0u && fits(g, tail)) ? SplitPlan{g, tail, head_of(g)} : none;
}

// Wavefront pipeline for large scenes (rtm_wavefront.h): nearest / shade launches until the compacted active
// list is empty.  The active count stays on the device, and so does everything that depends on it (the number of
// parts the sphere list is cut into, which blocks of a launch have work); two ways to drive the trips:
//  * FIXED BUDGET (a depth cap, spp x (max_bounces + 1) <= kWfAsyncTrips): a pixel needs at most max_bounces + 1 casts
//    per sample and advances one cast per trip, so that many trips finish every frame.  All of them are enqueued at
//    once, with grids sized for the worst case; a launch whose list is empty falls through in a few microseconds (the
//    headline stress frame needs ~1 100 of its 2 304 trips: ~10 ms of empty launches behind 24 s of work).  The call
//    only enqueues — rtm_render_scene's contract for every other scene size.
//  * FOLLOWING THE COUNT (a call with rtm_stats, which waits anyway; unlimited depth; a budget too long to enqueue
//    blindly): launches are queued a batch of
//    trips at a time, a batch ends with a stream-ordered copy of the count into a pinned host word, and the host reads
//    the count of batch b-1 only after batch b has been queued behind it — the GPU never waits for the host, but the
//    call returns only when the count has been seen at zero.
constexpr unsigned long long kWfAsyncTrips = 16384;
struct RenderPlan;
static int run_wavefront_impl(const RenderParams& P, int rows, StreamCtx& ctx, bool may_block, int levels, size_t bytes) {
    hipStream_t stream = ctx.stream;
    WfState S;
    std::memset(&S, 0, sizeof S);
    S.npix = (unsigned)rows * (unsigned)P.W;
    S.levels = levels;  // (plan_render: the cap when there is one, else as many as 2 GiB of HBM buy)
    const size_t N = S.npix;
    unsigned char* ws = nullptr;
    int rc = scratch_acquire(ctx, kScratchWavefront, bytes, (void**)&ws);
    if (rc != RTM_OK) return rc;
    unsigned char* q = ws;
    auto take = [&](size_t b) {
        unsigned char* r = q;
        q += (b + 15) & ~(size_t)15;
        return r;
    };
    S.org = (double*)take(N * 24);
    S.dir = (double*)take(N * 24);
    S.pdir = (double*)take(N * 24);
    S.acc = (double*)take(N * 24);
    S.hit_t = (double*)take(N * 8);
    S.hit_id = (int*)take(N * 4);
    S.rng_ctr = (unsigned*)take(N * 4);
    S.rng_k1 = (unsigned*)take(N * 4);
    S.n = (unsigned*)take(N * 4);
    S.left = (int*)take(N * 4);
    S.depth = (int*)take(N * 4);
    S.rec = (unsigned*)take(N * 4 * (size_t)S.levels);
    S.active[0] = (unsigned*)take(N * 4);
    S.active[1] = (unsigned*)take(N * 4);
    S.n_active = (unsigned*)take(16);
    S.part_t = (double*)take((size_t)(kWfMaxParts - 1) * kWfPartSlots * 8);
    S.part_id = (int*)take((size_t)(kWfMaxParts - 1) * kWfPartSlots * 4);
    S.part_slots = kWfPartSlots;
    if (!P.scene.geom32) {
        set_last_error("scene without rejection-test data");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    const unsigned grid = (unsigned)((N + 255) / 256);
    wf_init_kernel<<<grid, 256, 0, stream>>>(P, S);  // also sets n_active = {npix, 0}
    RTM_HIP_CHECK(hipGetLastError());

    const unsigned shade_grid = grid;
    // (a call WITH rtm_stats waits for the stream anyway: it follows the count and skips the empty launches)
    if (!may_block && P.max_bounces >= 0 &&
        (unsigned long long)P.total_samples * (unsigned long long)(P.max_bounces + 1) <= kWfAsyncTrips) {
        const unsigned trips = P.total_samples * (unsigned)(P.max_bounces + 1);
        const unsigned near_grid = wf_nearest_grid(S.npix, P.scene.n);
        int cur = 0;
        for (unsigned t = 0; t < trips; ++t, cur ^= 1) {
            wf_nearest_f32_kernel<MathFast, 256, 8><<<near_grid, 256, 256 * kWfCandCap * sizeof(unsigned), stream>>>(P, S, cur);
            wf_shade_kernel<<<shade_grid, 256, 0, stream>>>(P, S, cur);
        }
        RTM_HIP_CHECK(hipGetLastError());
        return RTM_OK;
    }

    struct Events {
        hipEvent_t e[2] = {nullptr, nullptr};
        ~Events() {
            for (auto v : e)
                if (v) (void)hipEventDestroy(v);
        }
    } ev;
    for (auto& v : ev.e) RTM_HIP_CHECK(hipEventCreateWithFlags(&v, hipEventDisableTiming));

    unsigned na = S.npix;  // last count the host has seen (an upper bound of the device's)
    int cur = 0;
    // every cast of every pixel is one trip; a pixel needs at most total_samples * (depth cap + 1)
    const unsigned long long max_trips = (unsigned long long)P.total_samples * (unsigned long long)(S.levels + 1) + 8;
    unsigned long long trip = 0;
    for (unsigned long long b = 0;; ++b) {
        if (trip > max_trips + 64) {
            set_last_error("wavefront loop did not terminate");
            return RTM_ERR_HIP;
        }
        // short trips (few spheres or few rays): eight per batch, so that the host's lag of one batch hides the
        // read-back; long trips: one
        const int batch = ((unsigned long long)na * (unsigned long long)P.scene.n < 2000000000ull) ? 8 : 1;
        const unsigned near_grid = wf_nearest_grid(na, P.scene.n);  // covers every count up to the last one seen
        const unsigned g = (na + 255) / 256;
        for (int k = 0; k < batch; ++k, ++trip) {
            wf_nearest_f32_kernel<MathFast, 256, 8><<<near_grid, 256, 256 * kWfCandCap * sizeof(unsigned), stream>>>(P, S, cur);
            wf_shade_kernel<<<g, 256, 0, stream>>>(P, S, cur);
            cur ^= 1;
        }
        RTM_HIP_CHECK(hipGetLastError());
        RTM_HIP_CHECK(hipMemcpyAsync(&ctx.wf_count_host[b & 1], S.n_active + cur, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        RTM_HIP_CHECK(hipEventRecord(ev.e[b & 1], stream));
        if (b >= 1) {  // batch b is queued: now look at what batch b-1 left
            RTM_HIP_CHECK(hipEventSynchronize(ev.e[(b - 1) & 1]));
            na = ctx.wf_count_host[(b - 1) & 1];
            static const bool trace = std::getenv("RTM_DEBUG_WF_TRACE") != nullptr;  // diagnostic: the active count, batch by batch
            if (trace) std::fprintf(stderr, "[rtm wf] trips %llu active %u\n", trip, na);
            if (na == 0) break;
        }
    }
    return RTM_OK;
}

// Small work buffers are hipMallocAsync/hipFreeAsync pairs; without a release threshold the default pool
// hands multi-GB buffers back to the driver at every synchronisation and re-maps them per call.
static void keep_stream_ordered_memory(int device) {
    static std::mutex mu;
    static std::vector<int> done;
    std::lock_guard<std::mutex> lock(mu);
    for (int d : done)
        if (d == device) return;
    done.push_back(device);
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
        uint64_t keep = ~(uint64_t)0;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    (void)hipGetLastError();
}

static const char* kOverflowText =
    "a path ran deeper than the hit-record capacity (16-64 on-chip levels + 960 pooled levels)";

// Report-and-clear of the stream's sticky overflow flag.  Caller holds ctx.mu; `wait` synchronises first.
static int take_stream_status(StreamCtx& ctx, bool wait) {
    if (!ctx.ready) return RTM_OK;
    if (wait) RTM_HIP_CHECK(hipStreamSynchronize(ctx.stream));
    if (*(volatile unsigned long long*)ctx.flag_host == 0ull) return RTM_OK;
    // error path: make sure no copy of the old flag is still in flight, then clear both sides
    RTM_HIP_CHECK(hipStreamSynchronize(ctx.stream));
    RTM_HIP_CHECK(hipMemsetAsync(ctx.sticky + 3, 0, sizeof(unsigned long long), ctx.stream));
    RTM_HIP_CHECK(hipStreamSynchronize(ctx.stream));
    *ctx.flag_host = 0ull;
    set_last_error(std::string("an earlier render on this stream was truncated: ") + kOverflowText);
    return RTM_ERR_UNSUPPORTED;
}

// rtm_stream_release: the (device, stream) context goes — after the stream's queued work, which may still use its buffers.
// The gate is taken EXCLUSIVELY, like release_scratch: a render on the same pair that has looked its context up
// (render_view: get_ctx) and is about to lock it must not find the mutex and the scratch pointers freed under it.
int stream_release(int device, void* stream_v) {
    std::unique_lock<std::shared_mutex> gate(g_gate);
    DeviceGuard guard;
    RTM_HIP_CHECK(hipSetDevice(device));
    std::unique_ptr<StreamCtx> ctx;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        auto it = g_ctx.find({device, (hipStream_t)stream_v});
        if (it == g_ctx.end()) return RTM_OK;
        ctx = std::move(it->second);
        g_ctx.erase(it);
    }
    int rc;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);  // a call that is enqueueing on the pair finishes first
        rc = take_stream_status(*ctx, true);        // waits for the stream; reports an overflow nobody has seen
        ctx->free_all();
    }
    reap_scenes(false);
    return rc;
}

int stream_status(int device, void* stream_v) {
    std::shared_lock<std::shared_mutex> gate(g_gate);
    RTM_HIP_CHECK(hipSetDevice(device));
    StreamCtx* ctx = get_ctx(device, (hipStream_t)stream_v);
    std::lock_guard<std::mutex> lock(ctx->mu);
    return take_stream_status(*ctx, true);
}

// The render proper: `view` is resident on opt->device and stays valid until the queued work has run.
// The caller holds g_gate (shared).
// Variant 17 (rtm_grid_kernel.h): the render kernel leaves every sample's term in the stream's term buffer, the finalize
// kernel adds them per pixel in sample order.  The buffer holds tiles x spp x 2 048 bytes (17.0 GB for the 32 400 tiles of
// a 1080p frame at 256 spp): beyond kGridTermBudget — or what the device will give — the frame is rendered in several
// launches of as many tiles as fit, one behind the other on the stream, each followed by its finalize.
constexpr size_t kGridTermBudget = (size_t)16 << 30;
static void launch_grid_kernel(const RenderParams& P, unsigned base, unsigned cnt, size_t lds, bool deep, bool planes, bool count,
                               hipStream_t stream) {
    (void)planes;  // (a plane is found at run time, wave-uniformly, among the objects every ray tests: no instantiation of its own)
    if (count) {
        if (deep) render_grid_kernel<uint32_t, 32, true><<<cnt, 64, lds, stream>>>(P, base);
        else render_grid_kernel<uint32_t, 16, true><<<cnt, 64, lds, stream>>>(P, base);
    } else {
        if (deep) render_grid_kernel<uint32_t, 32, false><<<cnt, 64, lds, stream>>>(P, base);
        else render_grid_kernel<uint32_t, 16, false><<<cnt, 64, lds, stream>>>(P, base);
    }
}

// ---- planning a render: everything that is decided before anything is allocated or launched ---------------------------
// The resolved kernel variant, the sample split and stealing plans, and the bytes of every per-(device, stream) work buffer
// the call will ask for (0 = none).  render_view acquires exactly these (and falls back where the plan says a buffer is
// optional); rtm_scratch_bytes reports them without touching the device's memory.
static SurfaceConsts surface_consts();
// The deferred-fold kernels' pre-pass (rtm_render_kernel.h: prim_prepass_kernel) leaves, per tile of the launch, every
// sub-pixel's primary direction — up to 4 GiB per launch, else the kernels compute them where they need them — and, for the
// tolerance row, 64 mask words.
static size_t prim_dir_bytes(unsigned tiles, int ss) {
    static const bool off = [] {
        const char* e = std::getenv("RTM_DEBUG_PRIM_DIRS");  // A/B knob: 0 = no direction table
        return e && e[0] == '0';
    }();
    const size_t b = (size_t)tiles * (size_t)ss * (size_t)ss * 3 * 64 * sizeof(double);
    return (off || b > ((size_t)4 << 30)) ? 0 : b;
}
static size_t prim_mask_bytes(unsigned tiles) { return (size_t)tiles * 64 * sizeof(unsigned long long); }
struct RenderPlan {
    int variant = 0;          // resolved
    bool tol = false;         // the labelled tolerance row (launched from rtm_kernels_tol.hip, planned like variant 2)
    bool force_split = false;
    bool packl = false;
    bool count_tests = false; // RTM_MODE_COUNT_TESTS
    unsigned grid = 0;        // tiles of the launch
    int rows = 0;
    size_t bytes[kScratchRoles] = {0, 0, 0, 0, 0};
    bool optional[kScratchRoles] = {false, false, false, false, false};  // without room the launch runs without the feature
    bool prepass = false;     // the deferred-fold kernels' pre-pass table (primary directions; the tolerance row: + masks)
    // steal plan
    unsigned steal_rows = 0, steal_depth = 0;
    bool steal = false;
    // wavefront plan
    int wf_levels = 0;
    // grid plan
    size_t grid_chunk_tiles = 0;
};

static size_t grid_term_budget() {
    static const size_t budget = [] {
        const char* e = std::getenv("RTM_DEBUG_GRID_BUDGET_MB");  // test knob: the term buffer's budget in MiB (several launches per frame)
        return e ? (size_t)std::strtoull(e, nullptr, 10) << 20 : kGridTermBudget;
    }();
    return budget;
}
static size_t wavefront_bytes(size_t npix, int levels) {
    const size_t part_bytes = (size_t)(kWfMaxParts - 1) * kWfPartSlots * 12 + 64;
    return npix * (3 * 8 * 4 + 8 + 4 + 4 * 5 + (size_t)levels * 4 + 2 * 4) + 256 + part_bytes;
}
static int wavefront_levels(const RenderParams& P, size_t npix) {
    // record levels per pixel: the cap when there is one, else as many as 2 GiB of HBM buy (64..1024); a deeper path
    // raises the overflow flag like in the other variants
    if (P.max_bounces >= 0 && P.max_bounces <= 1024) return P.max_bounces > 0 ? P.max_bounces : 1;
    const size_t budget = ((size_t)2 << 30) / (4 * (npix ? npix : 1));
    return (int)(budget < 64 ? 64 : (budget > 1024 ? 1024 : budget));
}

// Fills P (camera, sizes, split fields) and the plan.  `view`: what the scene offers (planes, grid).
static int plan_render(const rtm_settings* st, const SceneView& view, size_t n, const rtm_options* opt, RenderParams& P,
                       RenderPlan& plan) {
    std::memset(&P, 0, sizeof P);
    fill_render_params(P, st, opt);
    P.scene = view;
    plan.rows = output_rows(opt);
    plan.count_tests = (opt->mode & RTM_MODE_COUNT_TESTS) != 0;
    const unsigned tiles_y = (unsigned)((plan.rows + 7) / 8);
    const unsigned grid = (unsigned)P.tiles_x * tiles_y;
    plan.grid = grid;
    int variant = opt->variant;
    P.n_tiles = grid;
    P.split = 1;
    P.split_len = P.split_head = P.total_samples;
    // auto (profiles/r1/variant_thresholds.txt): LDS tables up to 24 spheres (8.5 KB of LDS per wave keeps
    // 16 waves per CU); global-memory tables up to 511 (the tables no longer cost occupancy); from 512
    // spheres the wavefront pipeline with its rejection test wins over the monolithic kernel
    plan.force_split = variant == kVariantSplit;
    if (variant < 0 || variant >= num_variants() || variant_retired(variant)) {
        set_last_error(variant >= 0 && variant < num_variants() ? std::string("variant ") + std::to_string(variant) + ": " + kVariantNames[variant]
                                                                : std::string("no such variant"));
        return RTM_ERR_UNSUPPORTED;
    }
    if (opt->mode & RTM_MODE_SURFACE_SAMPLE) {
        // the integrator switch: png::SurfaeSample through its own general kernel (rtm_surface.h), whatever the scene's size
        if (variant != kVariantAuto && variant != kVariantSurface) {
            set_last_error("RTM_MODE_SURFACE_SAMPLE is served by its own kernel: variant 0 (or 19)");
            return RTM_ERR_UNSUPPORTED;
        }
        if (view.surf == nullptr || n == 0) {
            set_last_error("RTM_MODE_SURFACE_SAMPLE needs a scene with at least one object");
            return RTM_ERR_UNSUPPORTED;
        }
        if (needs_pool(P)) {
            P.pool_slots = 65536u;
            plan.bytes[kScratchPool] = (size_t)P.pool_slots * kPoolLevels * sizeof(uint2) + 64;
        }
        plan.variant = kVariantSurface;
        return RTM_OK;
    }
    if (variant == kVariantSurface) {
        set_last_error("variant 19 is chosen by RTM_MODE_SURFACE_SAMPLE in rtm_options.mode");
        return RTM_ERR_UNSUPPORTED;
    }
    if (variant == kVariantAuto)
        variant = n <= (size_t)kAutoLdsTableSpheres ? kVariantFastLds :
                  view.grid != nullptr ? kVariantGrid :  // (grid_for: the scene has one and the camera is within its reach)
                  n < 256 ? kVariantGlobalDefer :
                  n < (size_t)kAutoWavefrontSpheres ? kVariantFastGlobal : kVariantWavefrontRejectF32;
    if (variant == kVariantGrid && view.grid == nullptr) {
        set_last_error("variant 17 (uniform grid) serves scenes of 64 gridded spheres or more held by an rtm_scene "
                       "(rtm_scene_create*, rtm_render_rows*) or made per call from a device array; this scene has no grid");
        return RTM_ERR_UNSUPPORTED;
    }
    else if (variant == kVariantSplit)
        variant = kVariantFastLds;
    // variant 18, the labelled tolerance row (rtm_kernels_tol.hip): planned exactly like the default kernel of small scenes —
    // sample split of the last tiles, in-wave sample stealing — and launched from the other translation unit
    const bool tol = variant == kVariantTol;
    plan.tol = tol;
    if (tol) {
        if (!(n >= 1 && n <= (size_t)kAutoLdsTableSpheres && view.plane == nullptr && P.total_samples < 65536u)) {
            set_last_error("variant 18 (fp64 tolerance row) serves all-sphere scenes of 1..24 spheres with fewer than 65 536 samples per pixel");
            return RTM_ERR_UNSUPPORTED;
        }
        variant = kVariantFastLds;
    }
    if (view.plane != nullptr && variant != kVariantGrid) {
        // png::PlaneObject in the scene: the chunked LDS-table kernels serve it up to 255 objects (variants 0, 2, 9), the
        // grid kernel scenes that have a grid (planes are tested by every ray, next to the spheres that span the scene), the
        // per-object loop with the compiler's math (variant 1) any size; the other kernels know spheres only
        if (opt->variant != kVariantAuto && opt->variant != kVariantRef && opt->variant != kVariantFastLds &&
            opt->variant != kVariantSplit) {
            set_last_error("scenes that hold planes are rendered by variants 0 (auto), 1 (per-object loop), 2, 9 and 17");
            return RTM_ERR_UNSUPPORTED;
        }
        if (opt->variant == kVariantRef || n >= 256)
            variant = kVariantRef;
        else
            variant = kVariantFastLds;
    }
    // beyond what their records / tables hold, the packed-record and the stamped kernels hand over to the global-table one
    if ((variant == kVariantStamped && n > (size_t)kLdsTableMaxSpheres) || (variant == kVariantGlobalDefer && n >= 256))
        variant = kVariantFastGlobal;
    if (variant == kVariantFp32 && !(n >= 1 && n <= 256 && P.mode == RTM_MODE_REPAIRED)) {
        set_last_error("variant 16 (fp32 fast row) serves repaired-mode scenes of 1..256 spheres");
        return RTM_ERR_UNSUPPORTED;
    }
    if (variant == kVariantPrimaryReuse &&
        !(n >= 1 && n <= (size_t)kAutoLdsTableSpheres && P.max_bounces >= 0 && P.max_bounces <= 8)) {
        set_last_error("variant 15 (primary-hit reuse) serves scenes of 1..24 spheres with 0 <= max_bounces <= 8");
        return RTM_ERR_UNSUPPORTED;
    }
    // the sample split rides on the packed-record kernels (explicit variant 2 never splits)
    if (n < 256 && (opt->variant == kVariantAuto || plan.force_split || tol) &&
        (variant == kVariantFastLds || variant == kVariantGlobalDefer)) {
        const SplitPlan sp = choose_split(grid, P.total_samples, opt->device, plan.force_split);
        P.split = sp.g;
        if (P.split > 1) {
            P.n_tiles = sp.tiles;
            P.split_first = grid - sp.tiles;
            P.split_len = P.total_samples / P.split;
            P.split_head = sp.head;
            P.split = 1u + (P.total_samples - P.split_head) / P.split_len;
            const size_t part = (size_t)P.n_tiles * 192 * sizeof(double);
            const size_t terms = (size_t)P.n_tiles * (P.total_samples - P.split_head) * kTermRowBytes;  // a row per (tile, sample)
            plan.bytes[kScratchTerms] = part + terms;
            plan.optional[kScratchTerms] = true;  // no room for the terms: the launch runs unsplit (same image, a longer tail)
        }
    }
    // In-wave sample stealing for the whole tiles of the packed-record LDS-table kernel (rtm_render_kernel.h, STEAL): a
    // block per whole tile for the accumulators and the stolen samples' terms.  Rows per tile: the stolen share of a tile
    // shrinks like 1 / sqrt(samples) (2.2 % of 64 x 1024 samples, 4.5 % of 64 x 256), i.e. ~0.7 sqrt(spp) rows; capacity
    // is three times that.  Without room for the blocks the launch runs without stealing (same image).
    static const bool steal_off = [] {
        const char* e = std::getenv("RTM_DEBUG_STEAL");  // tuning knob: 0 = no sample stealing
        return e && e[0] == '0';
    }();
    const unsigned n_whole = P.split > 1 ? P.split_first : grid;
    if (!steal_off && variant == kVariantFastLds && view.plane == nullptr && n >= 1 && n < 256 /* packed records: PACK8 */ &&
        P.max_bounces >= 0 && P.max_bounces <= 8 && P.total_samples >= 16 && P.total_samples < 65536u && st->samples < 65536 &&
        opt->variant != kVariantFastLds /* explicit variant 2 stays the plain kernel: the A/B twin */ &&
        !(tol && std::getenv("RTM_DEBUG_TOL_NOSTEAL") != nullptr) /* test knob: the tolerance row's stealing-free path */ &&
        n_whole != 0) {
        unsigned rows = 2u * (unsigned)std::ceil(std::sqrt((double)P.total_samples)) + 4u;
        rows = rows < 8u ? 8u : (rows > 68u ? 68u : rows);
        unsigned depth = P.total_samples / 2u;
        depth = depth > 256u ? 256u : depth;
        plan.steal = true;
        plan.steal_rows = rows;
        plan.steal_depth = depth;
        plan.bytes[kScratchSteal] = (size_t)n_whole * steal_tile_bytes(rows);
        plan.optional[kScratchSteal] = !tol;
    } else if (tol && P.max_bounces >= 0 && P.max_bounces <= 8) {
        // the depth-capped tolerance row has the STEAL instantiation only: frames the stealing declines (fewer than 16 samples
        // per pixel) run it with no row to steal into — every lane traces its own samples, the tile's block carries the
        // accumulators to steal_finalize_kernel.  (Any other depth: the any-depth kernel, which has no stealing.)
        plan.bytes[kScratchSteal] = (size_t)(n_whole ? n_whole : 1) * steal_tile_bytes(0);
    }
    // the pre-pass table of the deferred-fold kernels: primary directions (optional), the tolerance row's masks (needed)
    if (variant == kVariantFastLds || variant == kVariantGlobalDefer) {
        const bool defer = n < 256;  // (launch_render: from 256 objects these variants hand over to the kernels without a fold queue)
        plan.bytes[kScratchPrim] = (defer ? prim_dir_bytes(grid, P.SS) : 0) + (tol ? prim_mask_bytes(grid) : 0);
        plan.prepass = plan.bytes[kScratchPrim] != 0;
        plan.optional[kScratchPrim] = !tol;
    }
    // deep-path record pool.  Kernels with an LDS record stack take a slot only for the
    // rare path beyond 64/32 levels (65536 slots x 960 records: 60 MiB u8 / 240 MiB u32); the packed-record
    // kernels (PACKL) keep levels >= 16 there, which nearly every pixel needs once: one slot per lane.
    plan.packl = n < 256 && !(P.max_bounces >= 0 && P.max_bounces <= 8) &&
                 (variant == kVariantFastLds || variant == kVariantGlobalDefer);
    if (needs_pool(P) && variant != kVariantWavefrontRejectF32 && variant != kVariantFp32) {
        // the record type of the kernel that will run: the grid kernel is instantiated for 4-byte records whatever n
        const size_t rec_bytes = (variant == kVariantGrid || n > 256) ? 4 : 1;
        P.pool_slots = plan.packl ? (P.split_first + P.n_tiles * P.split) * 128u : 65536u;  // two per lane
        plan.bytes[kScratchPool] = (size_t)P.pool_slots * kPoolLevels * rec_bytes + 64;
    }
    if (variant == kVariantWavefrontRejectF32) {
        const size_t npix = (size_t)plan.rows * (size_t)P.W;
        plan.wf_levels = wavefront_levels(P, npix);
        plan.bytes[kScratchWavefront] = wavefront_bytes(npix, plan.wf_levels);
    }
    if (variant == kVariantGrid) {
        const size_t per_tile = grid_tile_term_bytes(P.total_samples) + grid_tile_bit_bytes(P.total_samples);
        plan.grid_chunk_tiles = std::min<size_t>(grid, std::max<size_t>(1, grid_term_budget() / per_tile));
        plan.bytes[kScratchTerms] = plan.grid_chunk_tiles * per_tile;
        plan.optional[kScratchTerms] = true;  // what the device will not give is made up for by more launches of fewer tiles
    }
    plan.variant = variant;
    return RTM_OK;
}

static int run_wavefront(const RenderParams& P, int rows, StreamCtx& ctx, bool may_block, const RenderPlan& plan) {
    return run_wavefront_impl(P, rows, ctx, may_block, plan.wf_levels, plan.bytes[kScratchWavefront]);
}

static int run_grid(RenderParams& P, unsigned tiles, StreamCtx& ctx, const RenderPlan& plan) {
    static const bool xcd_off = [] {
        const char* e = std::getenv("RTM_DEBUG_GRID_XCD");  // tuning knob: 0 = blocks render tiles in launch order
        return e && e[0] == '0';
    }();
    const size_t per_tile = grid_tile_term_bytes(P.total_samples) + grid_tile_bit_bytes(P.total_samples);
    size_t chunk = plan.grid_chunk_tiles;
    void* ws = nullptr;
    // What the device can give, asked once (a failed multi-GB hipMalloc after the old buffer has been freed costs a stream
    // synchronisation per attempt): a buffer that has to grow is sized to what is free, never below 64 tiles.
    if (ctx.scratch[kScratchTerms].bytes < chunk * per_tile) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t usable = free_b + ctx.scratch[kScratchTerms].bytes;  // the old buffer goes before the new one comes
            const size_t keep = (size_t)512 << 20;
            if (usable > keep && (usable - keep) / per_tile < chunk) chunk = std::max<size_t>(std::min<size_t>(64, chunk), (usable - keep) / per_tile);
        }
        (void)hipGetLastError();
    }
    for (;;) {
        if (scratch_acquire(ctx, kScratchTerms, chunk * per_tile, &ws) == RTM_OK) break;
        (void)hipGetLastError();
        if (chunk <= 64) {
            set_last_error("no device memory for the grid kernel's term buffer");
            return RTM_ERR_HIP;
        }
        chunk = (chunk + 1) / 2;
    }
    P.contrib = static_cast<unsigned char*>(ws);
    P.nz_bits = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(ws) + chunk * grid_tile_term_bytes(P.total_samples));
    const bool deep = needs_pool(P);
    const bool planes = P.scene.plane != nullptr;
    size_t lds = (10 + kTrigConstCount) * sizeof(double) + (size_t)(deep ? 32 : 16) * 64 * sizeof(uint32_t) + 16 +
                 GridWalk<MathFast, SceneGlobal>::queue_bytes(64) + debug_lds_pad();
    P.unit_tab = unit_table_fits(lds) ? 1u : 0u;
    if (P.unit_tab) lds += kUnitTableBytes;
    for (unsigned base = 0; base < tiles; base += (unsigned)chunk) {
        const unsigned cnt = (unsigned)std::min<size_t>(chunk, tiles - base);
        P.xcd_on = xcd_off ? 0u : 1u;
        P.xcd_q = cnt / 8u;
        P.xcd_rem = cnt % 8u;
        RTM_HIP_CHECK(hipMemsetAsync(P.nz_bits, 0, (size_t)cnt * grid_tile_bit_bytes(P.total_samples), ctx.stream));
        launch_grid_kernel(P, base, cnt, lds, deep, planes, plan.count_tests, ctx.stream);
        grid_finalize_kernel<<<cnt, 64, 0, ctx.stream>>>(P, base);
    }
    RTM_HIP_CHECK(hipGetLastError());
    return RTM_OK;
}

static int render_view(const rtm_settings* st, const SceneView& view, size_t n, const rtm_options* opt,
                       double* out64, float* out32, uint8_t* out8, hipStream_t stream, rtm_stats* stats) {
    RTM_HIP_CHECK(hipSetDevice(opt->device));
    if (stats) std::memset(stats, 0, sizeof *stats);
    StreamCtx& ctx = *get_ctx(opt->device, stream);
    std::lock_guard<std::mutex> lock(ctx.mu);
    int rc = ctx.init();
    if (rc != RTM_OK) return rc;
    rc = take_stream_status(ctx, false);  // an overflow of an earlier render that has reached the host
    if (rc != RTM_OK) return rc;
    if (output_rows(opt) == 0) return RTM_OK;

    RenderParams P;
    RenderPlan plan;
    rc = plan_render(st, view, n, opt, P, plan);
    if (rc != RTM_OK) return rc;
    const int rows = plan.rows;
    const unsigned grid = plan.grid;
    const int variant = plan.variant;
    const bool tol = plan.tol;
    if (opt->mode & RTM_MODE_HOST_TRIG) {
        const uint32_t* fix = nullptr;
        rc = ensure_trig_fix(opt->device, &fix);
        if (rc != RTM_OK) return rc;
        P.scene.trig_fix = fix;
    }
    P.out64 = out64;
    P.out32 = out32;
    P.out8 = out8;
    P.counters = stats ? ctx.counters : ctx.sticky;
    if (stats) RTM_HIP_CHECK(hipMemsetAsync(ctx.counters, 0, 5 * sizeof(unsigned long long), stream));
    // the sample split's terms (the default kernels only; the grid kernel sizes its own buffer in run_grid)
    if (P.split > 1) {
        double* split_ws = nullptr;
        rc = scratch_acquire(ctx, kScratchTerms, plan.bytes[kScratchTerms], (void**)&split_ws);
        if (rc == RTM_OK) {
            P.partial = split_ws;
            P.contrib = reinterpret_cast<unsigned char*>(split_ws) + (size_t)P.n_tiles * 192 * sizeof(double);
        } else {  // no room for the terms: the launch runs unsplit (same image, a longer tail) instead of failing
            (void)hipGetLastError();
            P.split = 1;
            P.n_tiles = grid;
            P.split_first = 0;
            P.split_len = P.split_head = P.total_samples;
        }
    }
    if (plan.bytes[kScratchSteal] != 0) {
        const unsigned n_whole = P.split > 1 ? P.split_first : grid;  // (an unsplit fallback has more whole tiles than planned)
        unsigned srows = plan.steal ? plan.steal_rows : 0u;
        void* ws = nullptr;
        size_t blocks = (size_t)(n_whole ? n_whole : 1) * steal_tile_bytes(srows);
        rc = (n_whole || tol) ? scratch_acquire(ctx, kScratchSteal, blocks, &ws) : RTM_ERR_HIP;
        if (rc != RTM_OK && tol) {  // the tolerance row cannot run without its blocks: the smallest form, or fail
            (void)hipGetLastError();
            srows = 0;
            blocks = (size_t)(n_whole ? n_whole : 1) * steal_tile_bytes(0);
            rc = scratch_acquire(ctx, kScratchSteal, blocks, &ws);
            if (rc != RTM_OK) return rc;
        }
        if (rc == RTM_OK) {
            P.steal_ws = static_cast<unsigned char*>(ws);
            P.steal_rows = srows;
            P.steal_depth = srows ? plan.steal_depth : 0u;
            if (srows || tol) {  // (the tolerance row divides a sample index by S in its tail loop whether it steals or not)
                P.magic_S = (unsigned)(0x100000000ull / (unsigned long long)st->samples) + 1u;
                P.magic_SS = (unsigned)(0x100000000ull / (unsigned long long)st->super_samples) + 1u;
            }
        } else {
            (void)hipGetLastError();
        }
    }
    if (plan.prepass) {
        // (an unsplit fallback keeps the grid's tile count: the table is indexed by tile of the launch)
        void* ws = nullptr;
        size_t dir_bytes = plan.bytes[kScratchPrim] - (tol ? prim_mask_bytes(grid) : 0);
        rc = scratch_acquire(ctx, kScratchPrim, plan.bytes[kScratchPrim], &ws);
        if (rc != RTM_OK && tol) {  // no room for the directions: the masks alone, or fail
            (void)hipGetLastError();
            dir_bytes = 0;
            rc = scratch_acquire(ctx, kScratchPrim, prim_mask_bytes(grid), &ws);
            if (rc != RTM_OK) return rc;
        }
        if (rc == RTM_OK) {
            unsigned char* base = static_cast<unsigned char*>(ws);
            if (tol) P.prim_masks = reinterpret_cast<const unsigned long long*>(base);
            if (dir_bytes != 0) P.prim_dirs = reinterpret_cast<const double*>(base + (tol ? prim_mask_bytes(grid) : 0));
        } else {
            (void)hipGetLastError();
        }
    }
    if (plan.bytes[kScratchPool] != 0) {
        unsigned char* pool = nullptr;
        if (plan.packl) P.pool_slots = (P.split_first + P.n_tiles * P.split) * 128u;  // (after a possible unsplit fallback)
        const size_t rec_bytes = variant == kVariantSurface ? sizeof(uint2) : (variant == kVariantGrid || n > 256) ? 4 : 1;
        const size_t pool_bytes = (size_t)P.pool_slots * kPoolLevels * rec_bytes;
        rc = scratch_acquire(ctx, kScratchPool, pool_bytes + 64, (void**)&pool);
        if (rc != RTM_OK) return rc;
        P.pool = pool;
        P.pool_next = reinterpret_cast<unsigned*>(pool + pool_bytes);
        RTM_HIP_CHECK(hipMemsetAsync(P.pool_next, 0, sizeof(unsigned), stream));
    }

    DevMem stamps;
    if (variant == kVariantStamped) {
        rc = stamps.alloc((size_t)grid * 4 * sizeof(unsigned long long));
        if (rc != RTM_OK) return rc;
        RTM_HIP_CHECK(hipMemset(stamps.p, 0, (size_t)grid * 4 * sizeof(unsigned long long)));
        P.stamps = stamps.as<unsigned long long>();
    }

    EventPair ev;
    if (stats) {
        rc = ev.create();
        if (rc != RTM_OK) return rc;
        RTM_HIP_CHECK(hipEventRecord(ev.a, stream));
    }
    if (variant == kVariantWavefrontRejectF32) {
        rc = run_wavefront(P, rows, ctx, stats != nullptr, plan);
        if (rc != RTM_OK) return rc;
    } else if (variant == kVariantGrid) {
        rc = run_grid(P, grid, ctx, plan);
        if (rc != RTM_OK) return rc;
    } else if (tol) {
        rc = launch_tol(&P, sizeof P, grid, debug_lds_pad(), stream);
        if (rc != RTM_OK) return rc;
    } else if (variant == kVariantSurface) {
        render_surface_kernel<<<grid, 64, (size_t)kSurfLdsLevels * 64 * sizeof(uint2), stream>>>(P, surface_consts());
    } else {
        if (P.prim_dirs != nullptr)  // the deferred-fold kernels' pre-pass: every sub-pixel's primary direction, once
            prim_prepass_kernel<<<P.split > 1 ? P.split_first + P.n_tiles : grid, 64, 0, stream>>>(P, nullptr, const_cast<double*>(P.prim_dirs));
        launch_render(variant, P, grid, stream);
    }
    RTM_HIP_CHECK(hipGetLastError());
    if (stamps.p) {  // diagnostic variant: print the per-wave segment shares
        RTM_HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)grid * 4);
        RTM_HIP_CHECK(hipMemcpy(h.data(), stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double a[4] = {0, 0, 0, 0};
        for (size_t b = 0; b < grid; ++b)
            for (int k = 0; k < 4; ++k) a[k] += (double)h[b * 4 + k];
        std::fprintf(stderr,
                     "[rtm stamps] per wave-iteration cycles: nearest %.0f, shade %.0f, end+loop %.0f (iterations/wave %.0f)\n",
                     a[0] / a[3], a[1] / a[3], a[2] / a[3], a[3] / grid);
    }
    if (!stats) {
        // the stream's sticky flag travels to the host behind the render, without anybody waiting for it
        RTM_HIP_CHECK(hipMemcpyAsync(ctx.flag_host, ctx.sticky + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        return RTM_OK;
    }
    RTM_HIP_CHECK(hipEventRecord(ev.b, stream));
    unsigned long long c[5];
    RTM_HIP_CHECK(hipMemcpyAsync(c, ctx.counters, sizeof c, hipMemcpyDeviceToHost, stream));
    RTM_HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0.f;
    RTM_HIP_CHECK(hipEventElapsedTime(&ms, ev.a, ev.b));
    stats->samples = (uint64_t)rows * st->width * P.total_samples;
    stats->casts = c[0];
    stats->bounces = c[1];
    stats->draws = c[2];
    stats->kernel_ms = ms;
    stats->variant = tol ? kVariantTol : variant;
    stats->split = (int32_t)P.split;
    // Intersect calls (src/Renderer.cpp:66) the render made: every object for every cast in the exhaustive kernels; counted
    // by the grid kernel when RTM_MODE_COUNT_TESTS asks for its counting instantiation (0: not counted)
    stats->object_tests = variant == kVariantGrid ? (plan.count_tests ? c[4] : 0ull) : c[0] * (uint64_t)n;
    if (c[3]) {
        set_last_error(kOverflowText);
        return RTM_ERR_UNSUPPORTED;
    }
    return RTM_OK;
}

// The scene's grid as this render may use it: always when variant 17 is asked for by name; for the automatic choice only
// when the camera is within the reach the grid's pads were sized for (GridHeader::reach2 — a camera farther than about
// two scene diagonals away would send every primary ray through the exhaustive loop instead).
static const void* grid_for(const rtm_scene* sc, const rtm_settings* st, const rtm_options* opt) {
    if (!sc->grid.p) return nullptr;
    if (opt->variant == kVariantGrid) return sc->grid.p;
    if (sc->grid_far_bounces) return nullptr;
    double d2 = 0.0;
    for (int k = 0; k < 3; ++k) d2 += (st->camera.origin[k] - sc->grid_hdr.cb[k]) * (st->camera.origin[k] - sc->grid_hdr.cb[k]);
    return (d2 <= sc->grid_hdr.reach2) ? sc->grid.p : nullptr;  // (NaN: no)
}

// rtm_scratch_bytes: what a render with these arguments asks of the per-(device, stream) work buffers (plan_render), without
// touching the device's memory.  out[0] total, out[1] per-sample terms (sample split / grid kernel), out[2] pooled hit
// records, out[3] the exhaustive pipeline's path state, out[4] stolen samples' rows, out[5] the pre-pass's table.
int scratch_bytes(const rtm_settings* st, const rtm_scene* scene, const rtm_options* opt, uint64_t out[6]) {
    if (!scene || !out) {
        set_last_error("null argument");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    int rc = validate(st, nullptr, 0, opt);
    if (rc != RTM_OK) return rc;
    RenderParams P;
    RenderPlan plan;
    const SceneView view = scene_view(scene->geom.as<double>(), scene->mat.as<double>(), scene->aux.as<double>(), scene->n,
                                      scene->has_planes ? scene->plane.as<double>() : nullptr, grid_for(scene, st, opt), scene->surf.as<double>(), scene->axis_pat, scene->fold_flags);
    for (int k = 0; k < 6; ++k) out[k] = 0;
    if (output_rows(opt) == 0) return RTM_OK;
    rc = plan_render(st, view, scene->n, opt, P, plan);
    if (rc != RTM_OK) return rc;
    out[1] = plan.bytes[kScratchTerms];
    out[2] = plan.bytes[kScratchPool];
    out[3] = plan.bytes[kScratchWavefront];
    out[4] = plan.bytes[kScratchSteal];
    out[5] = plan.bytes[kScratchPrim];
    out[0] = out[1] + out[2] + out[3] + out[4] + out[5];
    return RTM_OK;
}

int render_scene(const rtm_settings* st, const rtm_scene* scene, const rtm_options* opt, double* out64, float* out32,
                 uint8_t* out8, void* stream_v, rtm_stats* stats) {
    if (!scene) {
        set_last_error("null scene");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    int rc = validate(st, nullptr, 0, opt);
    if (rc == RTM_OK && scene->n > 0x7FFFFFFFull) rc = RTM_ERR_UNSUPPORTED;
    if (rc != RTM_OK) return rc;
    if (scene->device != opt->device) {
        set_last_error("the scene lives on another device than rtm_options.device");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    std::shared_lock<std::shared_mutex> gate(g_gate);
    reap_scenes(false);
    rc = render_view(st, scene_view(scene->geom.as<double>(), scene->mat.as<double>(), scene->aux.as<double>(), scene->n,
                                    scene->has_planes ? scene->plane.as<double>() : nullptr, grid_for(scene, st, opt), scene->surf.as<double>(), scene->axis_pat, scene->fold_flags),
                     scene->n, opt, out64, out32, out8, (hipStream_t)stream_v, stats);
    note_scene_use(scene, (hipStream_t)stream_v);  // also after a failure: part of the work may have been queued
    return rc;
}

int render_device(const rtm_settings* st, const rtm_sphere* sp, size_t n, int on_device,
                  const rtm_options* opt, double* out64, float* out32, uint8_t* out8,
                  void* stream_v, rtm_stats* stats) {
    int rc = validate(st, sp, n, opt);
    if (rc != RTM_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_v;
    std::shared_lock<std::shared_mutex> gate(g_gate);
    RTM_HIP_CHECK(hipSetDevice(opt->device));
    if (!on_device) {
        std::shared_ptr<rtm_scene> sc;
        rc = cached_scene(sp, n, opt->device, &sc);
        if (rc != RTM_OK) return rc;
        rc = render_view(st, scene_view(sc->geom.as<double>(), sc->mat.as<double>(), sc->aux.as<double>(), n, nullptr, grid_for(sc.get(), st, opt), sc->surf.as<double>(), sc->axis_pat, sc->fold_flags),
                         n, opt, out64, out32, out8, stream, stats);
        note_scene_use(sc.get(), stream);  // an eviction while this render is queued parks the tables instead of waiting
        return rc;
    }
    // A device-resident array of a size that gets a grid (rtm_scene_create), for the variants that use one: a scene object
    // made from it — flattened tables + the host-built grid — is kept in a small cache keyed by a hash of the array's
    // CONTENT, taken on the device behind the stream's queued work (the caller may have written the array there); the call
    // waits for those 8 bytes.  On a miss the array comes back once for the grid build.  Before round 4 such calls had
    // no scene object and therefore no grid: the exhaustive pipeline, 118 x the time for BASELINE configs[4].
    if (n >= kGridMinSpheres && (opt->variant == kVariantAuto || opt->variant == kVariantGrid)) {
        std::shared_ptr<rtm_scene> sc;
        rc = cached_device_scene(sp, n, opt->device, stream, &sc);
        if (rc != RTM_OK) return rc;
        rc = render_view(st, scene_view(sc->geom.as<double>(), sc->mat.as<double>(), sc->aux.as<double>(), n, nullptr, grid_for(sc.get(), st, opt), sc->surf.as<double>(), sc->axis_pat, sc->fold_flags),
                         n, opt, out64, out32, out8, stream, stats);
        note_scene_use(sc.get(), stream);
        return rc;
    }
    // smaller device-resident arrays: flattened on the stream into stream-ordered temporaries, which are
    // released (hipFreeAsync) behind the render's launches
    keep_stream_ordered_memory(opt->device);
    AsyncMem geom, mat, aux, surf;
    rc = geom.alloc((n ? n : 1) * 4 * sizeof(double), stream);
    if (rc == RTM_OK) rc = mat.alloc((n + 1) * 8 * sizeof(double), stream);
    if (rc == RTM_OK) rc = aux.alloc(scene_aux_doubles(n) * sizeof(double), stream);
    if (rc == RTM_OK) rc = surf.alloc((n ? n : 1) * 4 * sizeof(double), stream);
    if (rc != RTM_OK) return rc;
    flatten_scene_kernel<<<(unsigned)((n + 1 + 255) / 256), 256, 0, stream>>>(sp, n, geom.as<double>(), mat.as<double>(), surf.as<double>());
    RTM_HIP_CHECK(hipGetLastError());
    rc = launch_scene_aux(geom.as<double>(), n, aux.as<double>(), stream);
    if (rc != RTM_OK) return rc;
    return render_view(st, scene_view(geom.as<double>(), mat.as<double>(), aux.as<double>(), n, nullptr, nullptr, surf.as<double>()), n, opt,
                       out64, out32, out8, stream, stats);
}

int render_host(const rtm_settings* st, const rtm_sphere* sp, size_t n, const rtm_options* opt,
                double* out64, float* out32, uint8_t* out8, rtm_stats* stats) {
    int rc = validate(st, sp, n, opt);
    if (rc != RTM_OK) return rc;
    RTM_HIP_CHECK(hipSetDevice(opt->device));
    const size_t vals = (size_t)output_rows(opt) * st->width * 3;
    DevMem d64, d32, d8;
    rtm_stats local;
    if (vals) {
        if (out64 && (rc = d64.alloc(vals * sizeof(double))) != RTM_OK) return rc;
        if (out32 && (rc = d32.alloc(vals * sizeof(float))) != RTM_OK) return rc;
        if (out8 && (rc = d8.alloc(vals)) != RTM_OK) return rc;
    }
    rc = render_device(st, sp, n, 0, opt, d64.as<double>(), d32.as<float>(), d8.as<uint8_t>(), nullptr, &local);
    if (rc == RTM_OK && vals) {
        if (out64) RTM_HIP_CHECK(hipMemcpy(out64, d64.p, vals * sizeof(double), hipMemcpyDeviceToHost));
        if (out32) RTM_HIP_CHECK(hipMemcpy(out32, d32.p, vals * sizeof(float), hipMemcpyDeviceToHost));
        if (out8) RTM_HIP_CHECK(hipMemcpy(out8, d8.p, vals, hipMemcpyDeviceToHost));
    }
    if (stats) *stats = local;
    return rc;
}

int render_host_objects(const rtm_settings* st, const rtm_object* objs, size_t n, const rtm_options* opt,
                        double* out64, float* out32, uint8_t* out8, rtm_stats* stats) {
    int rc = validate(st, nullptr, 0, opt);
    if (rc != RTM_OK) return rc;
    rtm_scene* raw = nullptr;
    rc = scene_create_objects(objs, n, opt->device, &raw);
    if (rc != RTM_OK) return rc;
    struct Destroy {
        void operator()(rtm_scene* p) const { (void)scene_destroy(p); }
    };
    std::unique_ptr<rtm_scene, Destroy> scene(raw);
    const size_t vals = (size_t)output_rows(opt) * st->width * 3;
    DevMem d64, d32, d8;
    rtm_stats local;
    if (vals) {
        if (out64 && (rc = d64.alloc(vals * sizeof(double))) != RTM_OK) return rc;
        if (out32 && (rc = d32.alloc(vals * sizeof(float))) != RTM_OK) return rc;
        if (out8 && (rc = d8.alloc(vals)) != RTM_OK) return rc;
    }
    rc = render_scene(st, scene.get(), opt, d64.as<double>(), d32.as<float>(), d8.as<uint8_t>(), nullptr, &local);
    if (rc == RTM_OK && vals) {
        if (out64) RTM_HIP_CHECK(hipMemcpy(out64, d64.p, vals * sizeof(double), hipMemcpyDeviceToHost));
        if (out32) RTM_HIP_CHECK(hipMemcpy(out32, d32.p, vals * sizeof(float), hipMemcpyDeviceToHost));
        if (out8) RTM_HIP_CHECK(hipMemcpy(out8, d8.p, vals, hipMemcpyDeviceToHost));
    }
    if (stats) *stats = local;
    return rc;
}

// SphereObject::ComputeSurfacePoint's localPoint (src/SettingData.cpp:229-231) as the HOST's libm evaluates it at run time
// (the oracle does the same): sin(2 pi) sin(pi / 2), sin(2 pi) cos(pi / 2), cos(2 pi).
static SurfaceConsts surface_consts() {
    static const SurfaceConsts k = [] {
        volatile double pi = 3.14159265358979323846;  // (volatile: evaluated by libm, not folded by the compiler)
        const double theta = 2.0 * pi, phi = 0.5 * pi;
        return SurfaceConsts{std::sin(theta) * std::sin(phi), std::sin(theta) * std::cos(phi), std::cos(theta)};
    }();
    return k;
}

// rtm_surface_sample_batch: png::SurfaeSample (depth 0 entry) for a batch of rays; ray i draws from stream (seed, i, 0)
int surface_sample_batch(const rtm_sphere* sp, size_t n, const rtm_options* opt, const double* org, const double* dir,
                         size_t n_rays, double* out, uint32_t* out_draws, uint32_t* out_casts) {
    if (!opt || !sp || !n || !org || !dir || !out) {
        set_last_error("null argument or empty scene");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if ((opt->mode & ~kModeFlags) != RTM_MODE_LITERAL && (opt->mode & ~kModeFlags) != RTM_MODE_REPAIRED) {
        set_last_error("unknown mode");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if (n_rays == 0) return RTM_OK;
    rtm_scene ds;
    int rc = scene_build_host(ds, sp, n, opt->device, false);
    if (rc != RTM_OK) return rc;
    DevMem d_org, d_dir, d_out, d_draws, d_casts, d_scratch, d_counters;
    const size_t vb = n_rays * 3 * sizeof(double);
    if ((rc = d_org.alloc(vb)) != RTM_OK || (rc = d_dir.alloc(vb)) != RTM_OK || (rc = d_out.alloc(vb)) != RTM_OK ||
        (rc = d_draws.alloc(n_rays * 4)) != RTM_OK || (rc = d_casts.alloc(n_rays * 4)) != RTM_OK ||
        (rc = d_scratch.alloc((size_t)SURF_MAX_DEPTH * n_rays * sizeof(uint2))) != RTM_OK || (rc = d_counters.alloc(32)) != RTM_OK)
        return rc;
    RTM_HIP_CHECK(hipMemset(d_counters.p, 0, 32));
    RTM_HIP_CHECK(hipMemcpy(d_org.p, org, vb, hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(d_dir.p, dir, vb, hipMemcpyHostToDevice));
    SurfBatchParams P;
    std::memset(&P, 0, sizeof P);
    P.scene = scene_view(ds.geom.as<double>(), ds.mat.as<double>(), nullptr, n, nullptr, nullptr, ds.surf.as<double>());
    P.K = surface_consts();
    P.mode = opt->mode & ~kModeFlags;
    P.max_bounces = opt->max_bounces;
    P.seed_mult = seed_multiplier(opt->seed);
    P.org = d_org.as<double>();
    P.dir = d_dir.as<double>();
    P.n_rays = n_rays;
    P.out = d_out.as<double>();
    P.out_draws = d_draws.as<uint32_t>();
    P.out_casts = d_casts.as<uint32_t>();
    P.scratch = d_scratch.as<uint2>();
    P.counters = d_counters.as<unsigned long long>();
    surface_sample_rays_kernel<<<(unsigned)((n_rays + 63) / 64), 64>>>(P);
    RTM_HIP_CHECK(hipGetLastError());
    RTM_HIP_CHECK(hipDeviceSynchronize());
    unsigned long long c[4];
    RTM_HIP_CHECK(hipMemcpy(c, d_counters.p, sizeof c, hipMemcpyDeviceToHost));
    RTM_HIP_CHECK(hipMemcpy(out, d_out.p, vb, hipMemcpyDeviceToHost));
    if (out_draws) RTM_HIP_CHECK(hipMemcpy(out_draws, d_draws.p, n_rays * 4, hipMemcpyDeviceToHost));
    if (out_casts) RTM_HIP_CHECK(hipMemcpy(out_casts, d_casts.p, n_rays * 4, hipMemcpyDeviceToHost));
    if (c[3]) {
        set_last_error("a SurfaeSample recursion ran deeper than SURF_MAX_DEPTH");
        return RTM_ERR_UNSUPPORTED;
    }
    return RTM_OK;
}

int path_trace_batch(const rtm_sphere* sp, size_t n, const rtm_options* opt, const double* org,
                     const double* dir, size_t n_rays, double* out, uint32_t* out_draws,
                     uint32_t* out_casts) {
    if (!opt || (!sp && n) || !org || !dir || !out) {
        set_last_error("null argument");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if (opt->band_count < 0 || (opt->band_count > 1 && (opt->band_index < 0 || opt->band_index >= opt->band_count))) {
        set_last_error("band_index outside [0, band_count)");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if ((opt->mode & ~kModeFlags) != RTM_MODE_LITERAL && (opt->mode & ~kModeFlags) != RTM_MODE_REPAIRED) {
        set_last_error("unknown mode");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if (n_rays == 0) return RTM_OK;
    rtm_scene ds;
    int rc = scene_build_host(ds, sp, n, opt->device);
    if (rc != RTM_OK) return rc;
    DevMem d_org, d_dir, d_out, d_draws, d_casts, d_scratch, d_counters, d_trace;
    const size_t vb = n_rays * 3 * sizeof(double);
    if ((rc = d_org.alloc(vb)) != RTM_OK || (rc = d_dir.alloc(vb)) != RTM_OK || (rc = d_out.alloc(vb)) != RTM_OK ||
        (rc = d_draws.alloc(n_rays * 4)) != RTM_OK || (rc = d_casts.alloc(n_rays * 4)) != RTM_OK ||
        (rc = d_scratch.alloc((size_t)RAY_MAX_DEPTH * n_rays * 4)) != RTM_OK || (rc = d_counters.alloc(32)) != RTM_OK)
        return rc;
    RTM_HIP_CHECK(hipMemset(d_counters.p, 0, 32));
    RTM_HIP_CHECK(hipMemcpy(d_org.p, org, vb, hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(d_dir.p, dir, vb, hipMemcpyHostToDevice));
    RayBatchParams P;
    std::memset(&P, 0, sizeof P);
    P.scene = scene_view(ds.geom.as<double>(), ds.mat.as<double>(), nullptr, n);
    if (opt->mode & RTM_MODE_HOST_TRIG) {
        const uint32_t* fix = nullptr;
        rc = ensure_trig_fix(opt->device, &fix);
        if (rc != RTM_OK) return rc;
        P.scene.trig_fix = fix;
    }
    P.mode = opt->mode & ~kModeFlags;
    P.max_bounces = opt->max_bounces;
    P.seed_mult = seed_multiplier(opt->seed);
    if (const char* key = std::getenv("RTM_DEBUG_SEAM_KEY")) {
        unsigned px = 0, sm = 0;
        if (std::sscanf(key, "%u,%u", &px, &sm) == 2) {
            P.key_override = 1;
            P.key_pixel = px;
            P.key_sample = sm;
        }
    }
    P.org = d_org.as<double>();
    P.dir = d_dir.as<double>();
    P.n_rays = n_rays;
    P.out = d_out.as<double>();
    P.out_draws = d_draws.as<uint32_t>();
    P.out_casts = d_casts.as<uint32_t>();
    P.scratch = d_scratch.as<uint32_t>();
    P.counters = d_counters.as<unsigned long long>();
    const char* trace_file = std::getenv("RTM_DEBUG_SEAM_TRACE");  // file that receives ray 0's states
    if (trace_file) {
        P.trace_cap = 256;
        if ((rc = d_trace.alloc((size_t)P.trace_cap * 6 * sizeof(double))) != RTM_OK) return rc;
        RTM_HIP_CHECK(hipMemset(d_trace.p, 0, (size_t)P.trace_cap * 6 * sizeof(double)));
        P.trace = d_trace.as<double>();
    }
    path_trace_rays_kernel<<<(unsigned)((n_rays + 63) / 64), 64>>>(P);
    RTM_HIP_CHECK(hipGetLastError());
    RTM_HIP_CHECK(hipDeviceSynchronize());
    if (d_trace.p) {
        std::vector<double> h((size_t)P.trace_cap * 6);
        RTM_HIP_CHECK(hipMemcpy(h.data(), d_trace.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(trace_file, "wb")) {
            std::fwrite(h.data(), sizeof(double), h.size(), f);
            std::fclose(f);
        }
    }
    unsigned long long c[4];
    RTM_HIP_CHECK(hipMemcpy(c, d_counters.p, sizeof c, hipMemcpyDeviceToHost));
    RTM_HIP_CHECK(hipMemcpy(out, d_out.p, vb, hipMemcpyDeviceToHost));
    if (out_draws) RTM_HIP_CHECK(hipMemcpy(out_draws, d_draws.p, n_rays * 4, hipMemcpyDeviceToHost));
    if (out_casts) RTM_HIP_CHECK(hipMemcpy(out_casts, d_casts.p, n_rays * 4, hipMemcpyDeviceToHost));
    if (c[3]) {
        set_last_error("a path ran deeper than RAY_MAX_DEPTH");
        return RTM_ERR_UNSUPPORTED;
    }
    return RTM_OK;
}

int intersect_batch(const rtm_sphere* sp, const double* org, const double* dir, size_t n, int mode,
                    int32_t* out_hit, double* out_t, double* out_normal) {
    if (!sp || !org || !dir || !out_hit || !out_t || !out_normal) {
        set_last_error("null argument");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return RTM_OK;
    int device = 0;
    RTM_HIP_CHECK(hipGetDevice(&device));
    rtm_scene ds;
    int rc = scene_build_host(ds, sp, n, device);
    if (rc != RTM_OK) return rc;
    DevMem d_org, d_dir, d_t, d_n, d_hit;
    const size_t vb = n * 3 * sizeof(double);
    if ((rc = d_org.alloc(vb)) != RTM_OK || (rc = d_dir.alloc(vb)) != RTM_OK || (rc = d_n.alloc(vb)) != RTM_OK ||
        (rc = d_t.alloc(n * sizeof(double))) != RTM_OK || (rc = d_hit.alloc(n * sizeof(int32_t))) != RTM_OK)
        return rc;
    RTM_HIP_CHECK(hipMemcpy(d_org.p, org, vb, hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(d_dir.p, dir, vb, hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(d_n.p, out_normal, vb, hipMemcpyHostToDevice));  // untouched where no hit
    RTM_HIP_CHECK(hipMemcpy(d_t.p, out_t, n * sizeof(double), hipMemcpyHostToDevice));
    intersect_pairs_kernel<<<(unsigned)((n + 63) / 64), 64>>>((const double4*)ds.geom.p, d_org.as<double>(), d_dir.as<double>(),
                                                              n, mode, d_hit.as<int32_t>(), d_t.as<double>(), d_n.as<double>());
    RTM_HIP_CHECK(hipGetLastError());
    RTM_HIP_CHECK(hipMemcpy(out_hit, d_hit.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    RTM_HIP_CHECK(hipMemcpy(out_t, d_t.p, n * sizeof(double), hipMemcpyDeviceToHost));
    RTM_HIP_CHECK(hipMemcpy(out_normal, d_n.p, vb, hipMemcpyDeviceToHost));
    return RTM_OK;
}

int rng_batch(uint64_t seed, uint32_t pixel0, uint32_t n_pixels, uint32_t sample, uint32_t n_draws,
              double* out) {
    if (!out) return RTM_ERR_INVALID_ARGUMENT;
    const size_t total = (size_t)n_pixels * n_draws;
    if (!total) return RTM_OK;
    DevMem d;
    const int rc = d.alloc(total * sizeof(double));
    if (rc != RTM_OK) return rc;
    rng_batch_kernel<<<(n_pixels + 255) / 256, 256>>>(seed_multiplier(seed), pixel0, n_pixels, sample,
                                                      n_draws, d.as<double>());
    RTM_HIP_CHECK(hipGetLastError());
    RTM_HIP_CHECK(hipMemcpy(out, d.p, total * sizeof(double), hipMemcpyDeviceToHost));
    return RTM_OK;
}

double rng_u01_host(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t index) {
    return rng_u01_at(seed_multiplier(seed), pixel, sample, index);
}

int math_probe(int op, const double* a, const double* b, size_t n, double* out) {
    if (!a || !out) return RTM_ERR_INVALID_ARGUMENT;
    if (!n) return RTM_OK;
    DevMem da, db, dout;
    int rc;
    if ((rc = da.alloc(n * 8)) != RTM_OK || (rc = dout.alloc(n * 8)) != RTM_OK) return rc;
    RTM_HIP_CHECK(hipMemcpy(da.p, a, n * 8, hipMemcpyHostToDevice));
    if (b) {
        if ((rc = db.alloc(n * 8)) != RTM_OK) return rc;
        RTM_HIP_CHECK(hipMemcpy(db.p, b, n * 8, hipMemcpyHostToDevice));
    }
    const uint32_t* fix = nullptr;
    if (op >= 16 && op <= 18) {
        rc = ensure_trig_fix(0, &fix);
        if (rc != RTM_OK) return rc;
    }
    if (op >= 32) {  // the tolerance row's arithmetic (rtm_kernels_tol.hip)
        if ((rc = tol_math_probe(op, da.as<double>(), b ? db.as<double>() : nullptr, n, dout.as<double>())) != RTM_OK) return rc;
    } else {
        math_probe_kernel<<<(unsigned)((n + 255) / 256), 256>>>(op, da.as<double>(), b ? db.as<double>() : nullptr, n,
                                                                dout.as<double>(), fix);
    }
    RTM_HIP_CHECK(hipGetLastError());
    RTM_HIP_CHECK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
    return RTM_OK;
}

// which: 0 nearest ref/1, 1 nearest fast/1 (sphere_test loop), 2 nearest fast/4 (batched), 3 nearest
// fast/2, 10 shade ref, 11 shade fast.  Returns average cycles per repetition per wave.
int component_bench(int which, const rtm_sphere* sp, size_t n, int reps, int blocks, int lds_pad,
                    double* cycles_per_rep) {
    int device = 0;
    RTM_HIP_CHECK(hipGetDevice(&device));
    rtm_scene ds;
    int rc = scene_build_host(ds, sp, n, device);
    if (rc != RTM_OK) return rc;
    DevMem d_out, d_cyc;
    if ((rc = d_out.alloc((size_t)blocks * 64 * 8)) != RTM_OK || (rc = d_cyc.alloc((size_t)blocks * 8)) != RTM_OK) return rc;
    double* out = d_out.as<double>();
    unsigned long long* cyc = d_cyc.as<unsigned long long>();
    const SceneView sv = scene_view(ds.geom.as<double>(), ds.mat.as<double>(), nullptr, n);
    const D3 org0 = D3{0.0, 0.0, -10.0};
    for (int pass = 0; pass < 2; ++pass) {
        const int r = pass ? reps : 4;
        switch (which) {
            case 0: nearest_bench_kernel<MathRef, 1><<<blocks, 64, lds_pad>>>(sv, org0, r, out, cyc); break;
            case 1: nearest_bench_kernel<MathFast, 1><<<blocks, 64, lds_pad>>>(sv, org0, r, out, cyc); break;
            case 2: nearest_bench_kernel<MathFast, 8><<<blocks, 64, lds_pad>>>(sv, org0, r, out, cyc); break;
            case 10: shade_bench_kernel<MathRef><<<blocks, 64, lds_pad>>>(sv, org0, r, out, cyc); break;
            case 11: shade_bench_kernel<MathFast><<<blocks, 64, lds_pad>>>(sv, org0, r, out, cyc); break;
            default: return RTM_ERR_INVALID_ARGUMENT;
        }
        RTM_HIP_CHECK(hipGetLastError());
        RTM_HIP_CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(blocks);
    RTM_HIP_CHECK(hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto v : h) sum += (double)v;
    *cycles_per_rep = sum / blocks / reps;
    return RTM_OK;
}

// Test hook: nearest hit for caller-given rays.  kind 1: the reference's loop, nothing in front of it (the literal
// per-sphere loop with the compiler's math, one lane per ray); kind 3: the large-scene kernel with its packed-fp32
// rejection test and candidate lists.  Host buffers; out_id/out_t per ray.
int wf_nearest_probe(int kind, const rtm_sphere* sp, size_t n, const double* org, const double* dir, size_t n_rays,
                     int32_t* out_id, double* out_t) {
    if (!sp || !n || !org || !dir || !n_rays || !out_id || !out_t || (kind != 1 && kind != 3) || n_rays > 0x7FFFFFFFull)
        return RTM_ERR_INVALID_ARGUMENT;
    int device = 0;
    RTM_HIP_CHECK(hipGetDevice(&device));
    rtm_scene ds;
    int rc = scene_build_host(ds, sp, n, device);
    if (rc != RTM_OK) return rc;
    RenderParams P;
    std::memset(&P, 0, sizeof P);
    P.scene = scene_view(ds.geom.as<double>(), ds.mat.as<double>(), nullptr, n);
    WfState S;
    std::memset(&S, 0, sizeof S);
    const size_t N = n_rays;
    S.npix = (unsigned)N;
    const int n_pad = ((int)n + 7) & ~7;
    const size_t bytes = N * (24 + 24 + 8 + 4 + 4) + 64 + ((size_t)n_pad * 5 + 4) * 8 + 256;
    DevMem ws_mem;
    if ((rc = ws_mem.alloc(bytes)) != RTM_OK) return rc;
    unsigned char* ws = ws_mem.as<unsigned char>();
    unsigned char* q = ws;
    auto take = [&](size_t b) {
        unsigned char* r = q;
        q += (b + 15) & ~(size_t)15;
        return r;
    };
    S.org = (double*)take(N * 24);
    S.dir = (double*)take(N * 24);
    S.hit_t = (double*)take(N * 8);
    S.hit_id = (int*)take(N * 4);
    S.active[0] = (unsigned*)take(N * 4);
    S.n_active = (unsigned*)take(16);
    double* aux = (double*)take(((size_t)n_pad * 5 + 4) * 8);
    std::vector<double> soa(N * 3);
    std::vector<unsigned> ident(N);
    for (size_t i = 0; i < N; ++i) {
        ident[i] = (unsigned)i;
        for (int k = 0; k < 3; ++k) soa[(size_t)k * N + i] = org[i * 3 + k];
    }
    RTM_HIP_CHECK(hipMemcpy(S.org, soa.data(), N * 24, hipMemcpyHostToDevice));
    for (size_t i = 0; i < N; ++i)
        for (int k = 0; k < 3; ++k) soa[(size_t)k * N + i] = dir[i * 3 + k];
    RTM_HIP_CHECK(hipMemcpy(S.dir, soa.data(), N * 24, hipMemcpyHostToDevice));
    RTM_HIP_CHECK(hipMemcpy(S.active[0], ident.data(), N * 4, hipMemcpyHostToDevice));
    const unsigned counts[2] = {(unsigned)N, 0u};
    RTM_HIP_CHECK(hipMemcpy(S.n_active, counts, sizeof counts, hipMemcpyHostToDevice));
    const unsigned g = (unsigned)((N + 255) / 256);
    if (kind == 3) {
        RTM_HIP_CHECK(hipMemset(aux, 0, 32));
        float4* g32 = reinterpret_cast<float4*>(aux + 4 + n_pad);
        wf_scene_scale_kernel<<<(n_pad + 255) / 256, 256>>>(P.scene.geom, P.scene.n, reinterpret_cast<long long*>(aux + 2));
        wf_scene_aux_kernel<<<(n_pad + 255) / 256, 256>>>(P.scene.geom, P.scene.n, n_pad, aux + 4,
                                                          reinterpret_cast<unsigned long long*>(aux), g32, g32 + n_pad);
        P.scene.bounds = aux;
        P.scene.wprime = aux + 4;
        P.scene.geom32 = g32;
        P.scene.geom32s = g32 + n_pad;
        wf_nearest_f32_kernel<MathFast, 256, 8><<<g, 256, 256 * kWfCandCap * sizeof(unsigned)>>>(P, S, 0);
    } else {
        nearest_probe_kernel<<<g, 256>>>(P.scene, S.org, S.dir, (unsigned)N, S.hit_id, S.hit_t);
    }
    RTM_HIP_CHECK(hipGetLastError());
    RTM_HIP_CHECK(hipDeviceSynchronize());
    RTM_HIP_CHECK(hipMemcpy(out_id, S.hit_id, N * 4, hipMemcpyDeviceToHost));
    RTM_HIP_CHECK(hipMemcpy(out_t, S.hit_t, N * 8, hipMemcpyDeviceToHost));
    return RTM_OK;
}

// rtm_debug_grid_nearest: the grid search of variant 17 on the caller's rays (AoS org/dir, n_rays x 3), with the per-ray
// counts of sphere tests and cell steps and the grid's size — tests/ compare it with kind 1 of rtm_debug_wf_nearest.
int grid_nearest_probe(const rtm_sphere* sp, size_t n, const double* org, const double* dir, size_t n_rays, int32_t* out_id,
                       double* out_t, uint32_t* out_tests, uint32_t* out_steps, uint64_t* info) {
    if (!sp || !n || !org || !dir || !n_rays || !out_id || !out_t || n_rays > 0x7FFFFFFFull) return RTM_ERR_INVALID_ARGUMENT;
    int device = 0;
    RTM_HIP_CHECK(hipGetDevice(&device));
    rtm_scene ds;
    int rc = scene_build_host(ds, sp, n, device);
    if (rc != RTM_OK) return rc;
    if (!ds.grid.p) {
        set_last_error("this scene gets no grid (fewer than 64 gridded spheres, non-finite geometry, or too many spheres "
                       "that span the scene)");
        return RTM_ERR_UNSUPPORTED;
    }
    if (info) {
        GridHeader H;
        RTM_HIP_CHECK(hipMemcpy(&H, ds.grid.p, sizeof H, hipMemcpyDeviceToHost));
        info[0] = ds.grid_cells;
        info[1] = ds.grid_refs;
        info[2] = ds.grid_big;
        info[3] = (uint64_t)H.dim[0];
        info[4] = (uint64_t)H.dim[1];
        info[5] = (uint64_t)H.dim[2];
    }
    const size_t N = n_rays;
    DevMem d_org, d_dir, d_t, d_id, d_tests, d_steps;
    if ((rc = d_org.alloc(N * 24)) != RTM_OK || (rc = d_dir.alloc(N * 24)) != RTM_OK || (rc = d_t.alloc(N * 8)) != RTM_OK ||
        (rc = d_id.alloc(N * 4)) != RTM_OK || (rc = d_tests.alloc(N * 4)) != RTM_OK || (rc = d_steps.alloc(N * 4)) != RTM_OK)
        return rc;
    std::vector<double> soa(N * 3);
    for (size_t i = 0; i < N; ++i)
        for (int k = 0; k < 3; ++k) soa[(size_t)k * N + i] = org[i * 3 + k];
    RTM_HIP_CHECK(hipMemcpy(d_org.p, soa.data(), N * 24, hipMemcpyHostToDevice));
    for (size_t i = 0; i < N; ++i)
        for (int k = 0; k < 3; ++k) soa[(size_t)k * N + i] = dir[i * 3 + k];
    RTM_HIP_CHECK(hipMemcpy(d_dir.p, soa.data(), N * 24, hipMemcpyHostToDevice));
    const SceneView sv = scene_view(ds.geom.as<double>(), ds.mat.as<double>(), nullptr, n, nullptr, ds.grid.p);
    grid_probe_kernel<<<(unsigned)((N + 255) / 256), 256>>>(sv, d_org.as<double>(), d_dir.as<double>(), (unsigned)N,
                                                            d_id.as<int>(), d_t.as<double>(), d_tests.as<unsigned>(),
                                                            d_steps.as<unsigned>());
    RTM_HIP_CHECK(hipGetLastError());
    RTM_HIP_CHECK(hipDeviceSynchronize());
    RTM_HIP_CHECK(hipMemcpy(out_id, d_id.p, N * 4, hipMemcpyDeviceToHost));
    RTM_HIP_CHECK(hipMemcpy(out_t, d_t.p, N * 8, hipMemcpyDeviceToHost));
    if (out_tests) RTM_HIP_CHECK(hipMemcpy(out_tests, d_tests.p, N * 4, hipMemcpyDeviceToHost));
    if (out_steps) RTM_HIP_CHECK(hipMemcpy(out_steps, d_steps.p, N * 4, hipMemcpyDeviceToHost));
    return RTM_OK;
}

int fp64_peak(int waves_per_simd, double min_ms, double* tflops, double* kernel_ms) {
    if (!tflops || waves_per_simd < 1 || waves_per_simd > 8) return RTM_ERR_INVALID_ARGUMENT;
    int device = 0, cus = 0;
    RTM_HIP_CHECK(hipGetDevice(&device));
    RTM_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    const int blocks = cus * waves_per_simd;  // 256 threads = one wave per SIMD of a CU
    DevMem out;
    int rc = out.alloc((size_t)blocks * 256 * sizeof(double));
    if (rc != RTM_OK) return rc;
    EventPair ev;
    if ((rc = ev.create()) != RTM_OK) return rc;
    int iters = 4000;
    float ms = 0.f;
    for (int attempt = 0; attempt < 6; ++attempt) {  // grow until the timed launch is long enough for the clock to settle
        fp64_peak_kernel<<<blocks, 256>>>(out.as<double>(), iters / 10 + 1);
        RTM_HIP_CHECK(hipEventRecord(ev.a, nullptr));
        fp64_peak_kernel<<<blocks, 256>>>(out.as<double>(), iters);
        RTM_HIP_CHECK(hipEventRecord(ev.b, nullptr));
        RTM_HIP_CHECK(hipEventSynchronize(ev.b));
        RTM_HIP_CHECK(hipEventElapsedTime(&ms, ev.a, ev.b));
        if (ms >= min_ms) break;
        iters = (int)(iters * (1.2 * min_ms / (ms > 0.05f ? ms : 0.05f))) + 1;
    }
    RTM_HIP_CHECK(hipGetLastError());
    // FMA = 2 flops (the vendor's convention): blocks x 256 lanes x iters x 64 instructions
    *tflops = 2.0 * (double)blocks * 256.0 * (double)iters * 64.0 / ((double)ms * 1e-3) / 1e12;
    if (kernel_ms) *kernel_ms = ms;
    return RTM_OK;
}

int selfcheck(int kind, unsigned long long* mismatches) {
    if (!mismatches) return RTM_ERR_INVALID_ARGUMENT;
    DevMem d;
    const int rc = d.alloc(8);
    if (rc != RTM_OK) return rc;
    RTM_HIP_CHECK(hipMemset(d.p, 0, 8));
    selfcheck_kernel<<<4096, 256>>>(kind, d.as<unsigned long long>());
    RTM_HIP_CHECK(hipGetLastError());
    RTM_HIP_CHECK(hipMemcpy(mismatches, d.p, 8, hipMemcpyDeviceToHost));
    return RTM_OK;
}

int device_count(int* count) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        if (count) *count = 0;
        set_last_error(std::string("no HIP device: ") + hipGetErrorString(e));
        return RTM_ERR_NO_DEVICE;
    }
    if (count) *count = c;
    return RTM_OK;
}

}  // namespace rtm
