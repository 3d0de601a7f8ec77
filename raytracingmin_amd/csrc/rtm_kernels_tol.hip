// rtm_kernels_tol.hip — the fp64 TOLERANCE row (rtm_options.variant 18), a separately labelled variant of the hot kernel.
//
// north_star's bar for the image is a per-pixel delta of 1e-4 against the CPU renderer; the default kernels meet it with 0
// (bit equality) by keeping every IEEE operation of the reference: no FMA contraction (the reference's x86-64 build has
// none) and correctly rounded division / square-root sequences.  This translation unit compiles THE SAME kernel source
// (rtm_device.h, rtm_path.h, rtm_render_kernel.h: same loop nest, same counter RNG and Russian-roulette thresholds, the
// float islands of src/Ray.h:67-72, src/SettingData.h:14-16 and src/SettingData.cpp:200,208, the same order of the
// additions into Renderer::image) a second time, into namespace rtm_tol, with
//   * FMA contraction allowed (-ffp-contract=fast-honor-pragmas), and
//   * division and square root to about one ulp (RTM_TOL: rtm_path.h, seq_rcp / seq_quot / seq_sqrt), sin / cos as the
//     device evaluates them (within one ulp of the host libm's; RTM_MODE_HOST_TRIG has no effect on this row);
//   * the fold L = colorKD * L + emission (src/Renderer.cpp:109) kept unfused (rtm_device.h: fold_step), so that a
//     sample's value stays a function of its path's hit ids alone: the image differs from the exact kernel's only where
//     a last-bit difference in a distance or a direction changes WHICH sphere a ray hits.
// What is instantiated: the default kernels of scenes up to 24 spheres — with a depth cap of at most 8 the LDS tables, chunked
// search, packed records, deferred fold, in-wave sample stealing and the sample split of a launch's last tiles; for any other
// depth (the reference's own unlimited recursion) the same with records packed by position and the pooled stack, no
// stealing — the class of every BASELINE Cornell configuration.  Never the default; bench.py reports it as a labelled row with its own
// roofline fraction and the count of pixels that differ from the exact frame (tests/test_tolerance_gpu.py).
#define RTM_NS rtm_tol
#define RTM_TOL 1
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "rtm_internal.h"
#include "rtm_render_kernel.h"

namespace rtm_tol {

// The axis-signature instantiations of this unit take the search's roots without their residual step (rtm_path.h:
// seq_sqrt_batch, LIGHT): for compact scenes seen by a camera that is within the same extent — anything else runs the plain
// exact-n kernels with the full roots.
static bool compact_launch(const RenderParams& P) {
#if RTM_TOL_LIGHT_ROOTS
    const double cam = std::sqrt(P.cam_org.x * P.cam_org.x + P.cam_org.y * P.cam_org.y + P.cam_org.z * P.cam_org.z);
    return (P.scene.fold_flags & kSceneCompact) != 0u && cam <= kCompactExtent;
#else
    (void)P;
    return true;
#endif
}

#ifndef RTM_TOL_WPE
#define RTM_TOL_WPE 4  // waves per SIMD of the depth-capped kernel's launch bound (A/B knob: profiles/r4/tol_wpe_ab.txt)
#endif
template <int UNROLL, bool SPLIT>
static void launch_one(const RenderParams& P_in, unsigned grid, size_t lds_pad, hipStream_t stream) {
    RenderParams P = P_in;
    size_t lds = lds_table_bytes(P.scene.n) + (10 + kTrigConstCount) * sizeof(double) + 6 * 64 * sizeof(double) +
                 kFoldQueueBytesS + 2 * 64 * sizeof(unsigned) + 64 * sizeof(unsigned long long) /* prim_mask */ + lds_pad;
    // the near-unit Normalize table (rtm_device.h) where its 256 bytes do not cost a wave per CU
    P.unit_tab = unit_table_fits(lds) ? 1u : 0u;
    if (P.unit_tab) lds += (size_t)(kShadeConstCount - kTrigConstCount) * sizeof(double);
    // <M, LDS_TAB, UNROLL, RecT, LDS_D, WPE, PARK, STAMP, PACK8, SPLIT, DEFER, PACKL, REUSE, PLANES, STEAL>
    render_tiles_kernel<MathFast, true, UNROLL, uint8_t, 16, RTM_TOL_WPE, true, false, true, SPLIT, true, false, false, false, true>
        <<<grid, 64, lds, stream>>>(P);
}
// Any depth (max_bounces < 0 — the reference's own semantics — or > 8): the deferred fold with records packed by position
// and the pooled stack from level 16 (PACKL, rtm_render_kernel.h), no in-wave stealing; a whole tile stores its own pixels.
template <int UNROLL, bool SPLIT>
static void launch_one_any(const RenderParams& P_in, unsigned grid, size_t lds_pad, hipStream_t stream) {
    RenderParams P = P_in;
    size_t lds = lds_table_bytes(P.scene.n) + (10 + kTrigConstCount) * sizeof(double) + 6 * 64 * sizeof(double) + kFoldQueueBytesLS + lds_pad;
    P.unit_tab = unit_table_fits(lds) ? 1u : 0u;
    if (P.unit_tab) lds += (size_t)(kShadeConstCount - kTrigConstCount) * sizeof(double);
    // <M, LDS_TAB, UNROLL, RecT, LDS_D, WPE, PARK, STAMP, PACK8, SPLIT, DEFER, PACKL>
    render_tiles_kernel<MathFast, true, UNROLL, uint8_t, 0, 4, true, false, false, SPLIT, true, true><<<grid, 64, lds, stream>>>(P);
}
template <bool SPLIT>
static void launch_n_any(const RenderParams& P, unsigned grid, size_t lds_pad, hipStream_t stream) {
#if RTM_OPT_AXIS
    const bool table_fits = unit_table_fits(lds_table_bytes(P.scene.n) + (10 + kTrigConstCount) * sizeof(double) + 6 * 64 * sizeof(double) +
                                            kFoldQueueBytesLS + lds_pad);  // (launch_one_any's rule)
    if (P.scene.n == 7 && P.scene.axis_pat == kAxisSigCornell7 && P.mode == RTM_MODE_REPAIRED && table_fits && compact_launch(P)) {  // (the shipped Cornell box: rtm_path.h, sphere_disc)
        launch_one_any<axis_unroll(7, kAxisSigCornell7), SPLIT>(P, grid, lds_pad, stream);
        return;
    }
#endif
    if (P.scene.n < 8) launch_one_any<-8, SPLIT>(P, grid, lds_pad, stream);
    else launch_one_any<8, SPLIT>(P, grid, lds_pad, stream);
}

template <bool SPLIT>
static void launch_n(const RenderParams& P, unsigned grid, size_t lds_pad, hipStream_t stream) {
#if RTM_OPT_AXIS
    // (the axis-signature instantiations take the near-unit Normalize table's presence as a compile-time fact: launch_one's rule)
    const bool table_fits = unit_table_fits(lds_table_bytes(P.scene.n) + (10 + kTrigConstCount) * sizeof(double) + 6 * 64 * sizeof(double) +
                                            kFoldQueueBytesS + 2 * 64 * sizeof(unsigned) + 64 * sizeof(unsigned long long) + lds_pad);
#define RTM_AXIS_CASE(k, sig)                                                                   \
    if (P.scene.n == k && P.scene.axis_pat == sig && P.mode == RTM_MODE_REPAIRED && table_fits && compact_launch(P)) { /* rtm_path.h: sphere_disc */ \
        launch_one<axis_unroll(k, sig), SPLIT>(P, grid, lds_pad, stream);                       \
        return;                                                                                 \
    }
    RTM_AXIS_SIGNATURES(RTM_AXIS_CASE)
#undef RTM_AXIS_CASE
#endif
    switch (P.scene.n) {  // the shipped scenes' sizes run the instantiation for exactly their sphere count
        case 3: launch_one<-103, SPLIT>(P, grid, lds_pad, stream); return;
        case 5: launch_one<-105, SPLIT>(P, grid, lds_pad, stream); return;
        case 7: launch_one<-107, SPLIT>(P, grid, lds_pad, stream); return;
        default: break;
    }
    if (P.scene.n < 8) launch_one<-8, SPLIT>(P, grid, lds_pad, stream);
    else launch_one<8, SPLIT>(P, grid, lds_pad, stream);
}

// rtm_debug_math_probe ops 32..: this translation unit's arithmetic on caller data (tests/test_tolerance_gpu.py measures the
// distance to the correctly rounded results in ulps).  32 the unscaled square root, 33 x / y by reciprocal, 34 the
// reciprocal alone, 35 x * y + 1.0 (contracted here: the other translation unit's op 7 must not be), 36 / 37 sin / cos of the
// branch-free sincos as compiled here, 38 one level of the fold (x * y + 0.25: must NOT be contracted), 39 / 40 sin / cos of
// 2 pi (x 2^-24) by the quadrant-exact sequence the shading block uses (x: a draw's 24-bit integer), 41 the search's light root
__global__ void tol_math_probe_kernel(int op, const double* __restrict__ a, const double* __restrict__ b, size_t n,
                                      double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b ? b[i] : 0.0;
    double r = 0.0, s, c;
    switch (op) {
        case 32: r = seq_sqrt(x); break;
        case 33: r = seq_quot(x, y, seq_rcp(y)); break;
        case 34: r = seq_rcp(x); break;
        case 35: r = x * y + 1.0; break;
        case 36: sincos_small(x, s, c); r = s; break;
        case 37: sincos_small(x, s, c); r = c; break;
        case 38: r = fold_step(d3(x, x, x), d3(y, y, y), d3(0.25, 0.25, 0.25)).y; break;
        case 39: sincos_turn24_k(TrigFromRegs{}, x, s, c); r = s; break;
        case 40: sincos_turn24_k(TrigFromRegs{}, x, s, c); r = c; break;
        case 41: {  // the search's light root (seq_sqrt_batch<K, true>)
            const double in[1] = {x};
            double out1[1];
            seq_sqrt_batch<1, true>(in, out1);
            r = out1[0];
            break;
        }
        default: break;
    }
    out[i] = r;
}

}  // namespace rtm_tol

namespace rtm {

int tol_math_probe(int op, const double* a_dev, const double* b_dev, size_t n, double* out_dev) {
    rtm_tol::tol_math_probe_kernel<<<(unsigned)((n + 255) / 256), 256>>>(op, a_dev, b_dev, n, out_dev);
    return hipGetLastError() == hipSuccess ? RTM_OK : RTM_ERR_HIP;
}

// `params`: the caller's rtm::RenderParams (the same struct, compiled from the same header into the other namespace),
// planned exactly as for the default kernel: split_first / n_tiles / split* for the sample split, steal_ws / steal_rows /
// steal_depth for the whole tiles (steal_ws must be there: the whole tiles' pixels are stored by steal_finalize_kernel).
int launch_tol(const void* params, size_t params_bytes, unsigned grid, size_t lds_pad, void* stream_v) {
    rtm_tol::RenderParams P;
    if (params_bytes != sizeof P) {
        set_last_error("tolerance row: RenderParams layout mismatch between the two translation units");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    std::memcpy(&P, params, sizeof P);
    // RTM_MODE_HOST_TRIG turns the device's sin / cos into the host libm's, one ulp apart on 3 % of the arguments: a
    // distinction this unit's own arithmetic does not keep anywhere else.  The row takes the device's (one table gather and
    // ten instructions per bounce less); the flag is accepted and has no effect here.
    P.scene.trig_fix = nullptr;
    hipStream_t stream = (hipStream_t)stream_v;
    const bool any_depth = P.max_bounces < 0 || P.max_bounces > 8;
    if ((!any_depth && P.steal_ws == nullptr) || P.scene.n < 1 || P.scene.n > 24 || P.scene.plane != nullptr ||
        P.total_samples >= 65536u) {
        set_last_error("variant 18 (fp64 tolerance row) serves all-sphere scenes of 1..24 spheres with fewer than 65 536 samples per pixel");
        return RTM_ERR_UNSUPPORTED;
    }
    if (P.prim_masks == nullptr) {
        set_last_error("tolerance row: no primary-ray mask buffer");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    // tiles of the launch: the whole ones and, behind them, the split ones (each once)
    const unsigned n_tiles_all = P.split > 1 ? P.split_first + P.n_tiles : grid;
    static const bool no_masks = [] {
        const char* e = std::getenv("RTM_DEBUG_TOL_PRIMFIX");  // A/B knob: 0 = no primary ray is flagged (NOT within tolerance on the Cornell diagonals)
        return e && e[0] == '0';
    }();
    if (no_masks) P.prim_dirs = nullptr;
    if (no_masks)
        (void)hipMemsetAsync(const_cast<unsigned long long*>(P.prim_masks), 0, (size_t)n_tiles_all * 64 * sizeof(unsigned long long), stream);
    else
        rtm_tol::prim_prepass_kernel<<<n_tiles_all, 64, 0, stream>>>(P, const_cast<unsigned long long*>(P.prim_masks),
                                                                 const_cast<double*>(P.prim_dirs));
    if (any_depth) {  // (no stealing: every whole tile stores its pixels itself)
        if (P.split > 1) {
            rtm_tol::launch_n_any<true>(P, P.split_first + P.n_tiles * P.split, lds_pad, stream);
            rtm_tol::split_finalize_kernel<<<P.n_tiles, 256, (size_t)P.split_len * 64 * 3 * sizeof(double), stream>>>(P);
        } else {
            rtm_tol::launch_n_any<false>(P, grid, lds_pad, stream);
        }
    } else {
        if (P.split > 1) {
            rtm_tol::launch_n<true>(P, P.split_first + P.n_tiles * P.split, lds_pad, stream);
            rtm_tol::split_finalize_kernel<<<P.n_tiles, 256, (size_t)P.split_len * 64 * 3 * sizeof(double), stream>>>(P);
        } else {
            rtm_tol::launch_n<false>(P, grid, lds_pad, stream);
        }
        const unsigned n_whole = P.split > 1 ? P.split_first : grid;
        if (n_whole) rtm_tol::steal_finalize_kernel<<<n_whole, 64, (size_t)P.steal_depth * 64 * sizeof(unsigned short), stream>>>(P);
    }
    if (hipGetLastError() != hipSuccess) {
        set_last_error("tolerance row: launch failed");
        return RTM_ERR_HIP;
    }
    return RTM_OK;
}

}  // namespace rtm
