/*
 * rtm_debug.h — test and diagnostic hooks exported by librtm_hip.so next to the C ABI of rtm.h.
 *
 * NOT part of the drop-in boundary: nothing here replaces a seam of the reference; they expose
 * building blocks of the device path (device math, the exhaustive self-checks, the large-scene
 * nearest-hit kernels, isolated loop timings) to tests/ and profiles/.  Host buffers throughout;
 * same status codes as rtm.h.  May change without an ABI version bump.
 */
#ifndef RTM_DEBUG_H
#define RTM_DEBUG_H

#include "rtm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* device sqrt / sqrtf / division / sin / cos and the exact-fast forms on caller data; `op` as in
 * math_probe_kernel (csrc/rtm_seam_kernels.h); ops 32..41: the tolerance row's arithmetic — one-ulp square root, division and
 * reciprocal, a contracted multiply-add, its sin / cos, the unfused fold step (csrc/rtm_kernels_tol.hip) */
int rtm_debug_math_probe(int op, const double* a, const double* b, size_t n, double* out);
/* exhaustive device self-checks; *mismatches = number of failing inputs (kind 0: fast sqrtf) */
int rtm_debug_selfcheck(int kind, unsigned long long* mismatches);
/* nearest hit for caller-given rays: kind 1 the reference's loop as written (src/Renderer.cpp:58-73, the compiler's
 * math, nothing in front of it), kind 3 the large-scene kernel (packed-fp32 rejection test + per-lane candidate lists
 * in front of the same arithmetic); any other kind is RTM_ERR_INVALID_ARGUMENT */
int rtm_debug_wf_nearest(int kind, const rtm_sphere* spheres, size_t n, const double* org, const double* dir,
                         size_t n_rays, int32_t* out_id, double* out_t);
/* the fp64 vector peak of the current device by wall clock: a chip-filling v_fma_f64 kernel (`waves_per_simd` waves on
 * every SIMD, 8 independent accumulators per lane) timed with HIP events over a launch of at least `min_ms`;
 * *tflops counts an FMA as 2 flops.  bench.py puts it beside the datasheet figure ("verify by
 * microbenchmark"); profiles/ubench/fp64_peak.hip is the stand-alone form with the per-instruction price list. */
int rtm_debug_fp64_peak(int waves_per_simd, double min_ms, double* tflops, double* kernel_ms);
/* The uniform-grid nearest-hit search of variant 17 (csrc/rtm_path.h: nearest_hit_grid) on caller-supplied rays
 * (org, dir: n_rays x 3 doubles, HOST): hit object (-1: none) and distance per ray as the reference loop
 * (src/Renderer.cpp:58-73) finds them — kind 1 of rtm_debug_wf_nearest is that loop — plus, where the pointers are not
 * NULL, the sphere tests and cell steps each ray took and info[6] = {cells, cell-list entries, spheres tested by every
 * ray, dim x, dim y, dim z}.  RTM_ERR_UNSUPPORTED when the scene gets no grid. */
int rtm_debug_grid_nearest(const rtm_sphere* spheres, size_t n, const double* org, const double* dir, size_t n_rays,
                           int32_t* out_id, double* out_t, uint32_t* out_tests, uint32_t* out_steps, uint64_t* info);
/* The grid's builder alone, on the HOST (no device is touched; runs in the CPU test suite): info[12] = {cells, list
 * entries, spheres tested by every ray, dim x, y, z} followed by the bit patterns of six doubles {box lo x, y, z, cell edge,
 * reach of origins from the box's centre, largest hit parameter the pads cover}; pads[n]: every sphere's pad; and, where the
 * pointers are not NULL, ranges[2 x cells] (first, one past last entry of each cell's list; x fastest), items[entries]
 * (sphere indices, ascending within a cell) and big[...].  RTM_ERR_UNSUPPORTED: no grid; RTM_ERR_CAPACITY: a buffer is short
 * (info is filled: call once without buffers for the sizes). */
int rtm_debug_grid_build(const rtm_sphere* spheres, size_t n, uint64_t* info, double* pads, uint32_t* ranges, size_t ranges_cap,
                         uint32_t* items, size_t items_cap, int32_t* big, size_t big_cap);
/* What the host finds in a sphere array when it flattens it into the kernels' tables, on the HOST (no device is touched; runs in
 * the CPU test suite): facts[0] = two bits per sphere for the first 32 — 1 / 2 / 3: the centre's only coordinate that is not
 * +-0 is x / y / z (the axis-signature instantiations of the exact-n kernels, csrc/rtm_path.h: sphere_disc) —, facts[1] bit 0:
 * every object a path can bounce off (kd > 0) emits (+0, +0, +0) and no colorKD or emission carries a sign bit (the packed
 * folds then leave a bounce level's "+ emission" out: SceneView::fold_flags), bit 1: every |centre| + radius is at most 1e7 (a
 * compact scene: the tolerance row's search may take its square roots without the residual step). */
int rtm_debug_scene_facts(const rtm_sphere* spheres, size_t n, uint64_t facts[2]);
/* isolated nearest-hit / shading loops timed with s_memtime (profiles/component_bench.py) */
int rtm_debug_component_bench(int which, const rtm_sphere* spheres, size_t n, int reps, int blocks, int lds_pad,
                              double* cycles_per_rep);

#ifdef __cplusplus
}
#endif
#endif /* RTM_DEBUG_H */
