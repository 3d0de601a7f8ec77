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
 * math_probe_kernel (csrc/rtm_kernels.hip) */
int rtm_debug_math_probe(int op, const double* a, const double* b, size_t n, double* out);
/* exhaustive device self-checks; *mismatches = number of failing inputs (kind 0: fast sqrtf) */
int rtm_debug_selfcheck(int kind, unsigned long long* mismatches);
/* the large-scene nearest-hit kernels on caller-given rays: kind 0 LDS tiles, 1 scalar stream,
 * 2 + fp64 rejection test, 3 + packed-fp32 rejection test (two spheres per instruction: the default),
 * 4 the same with two rays per instruction (A/B twin) */
int rtm_debug_wf_nearest(int kind, const rtm_sphere* spheres, size_t n, const double* org, const double* dir,
                         size_t n_rays, int32_t* out_id, double* out_t);
/* isolated nearest-hit / shading loops timed with s_memtime (profiles/component_bench.py) */
int rtm_debug_component_bench(int which, const rtm_sphere* spheres, size_t n, int reps, int blocks, int lds_pad,
                              double* cycles_per_rep);

#ifdef __cplusplus
}
#endif
#endif /* RTM_DEBUG_H */
