/*
 * oracle/cpu_ref.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64) of the reference's hot path, used only as the parity checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * raytracingmin_amd/ or include/ may link, load or call it.
 *
 * Parity status: PINNED by the known answers of SURVEY.md §8(c) (L0 whole-image FNV hashes,
 * L0/L1 per-ray radiance to 17 significant digits), which were produced from the reference's own
 * compiled code in the survey stage.  The reference itself is unbuildable in this image without
 * writing stand-ins (stb_image_write.h is absent; libstdc++ 11 has no std::sqrtf), so there is no
 * oracle/_ref; see DESIGN.md §Oracle.
 *
 * All citations are file:line in the reference checkout.
 */
#ifndef RTM_ORACLE_CPU_REF_H
#define RTM_ORACLE_CPU_REF_H

#include "../include/rtm.h" /* POD layouts only (rtm_settings, rtm_sphere, rtm_options) */

#ifdef __cplusplus
extern "C" {
#endif

typedef double (*rtmo_rng_fn)(void* ctx);

typedef struct rtmo_counters {
    uint64_t samples, casts, bounces, draws;
    uint64_t sphere_tests;      /* Intersect calls                         */
    uint64_t sphere_tests_d4;   /* ... that got past D4 < 0 (took a sqrt)  */
    uint64_t libc_rand_calls;   /* L0 only: calls of the ::rand generator (D3); expected 0 */
    uint64_t max_depth;         /* deepest recursion reached               */
} rtmo_counters;

/* src/Ray.h:61-72 */
double rtmo_dot(const double a[3], const double b[3]);
void rtmo_cross(const double a[3], const double b[3], double out[3]);
double rtmo_magnitude(const double a[3]);
void rtmo_normalize(const double a[3], double out[3]);
void rtmo_sin_cos_array(const double* x, size_t n, double* out_sin, double* out_cos);
void rtmo_set_trace(double* buf, int capacity); /* debugging aid, see cpu_ref.c */

/* src/SettingData.h:11-16 */
float rtmo_kd(const rtm_sphere* s);
void rtmo_color_kd(const rtm_sphere* s, double out[3]);

/* src/SettingData.cpp:197-226.  mode = RTM_MODE_LITERAL: out_normal is never written (D2). */
int rtmo_intersect(const rtm_sphere* s, const double org[3], const double dir[3], int mode,
                   double* out_t, double out_normal[3]);

/* src/Renderer.cpp:57-117 (recursive, like the reference). */
void rtmo_path_trace(const rtm_sphere* spheres, size_t n, int mode, int max_bounces,
                     const double org[3], const double dir[3], rtmo_rng_fn rng, void* rng_ctx,
                     double out_radiance[3], rtmo_counters* counters);

/* png::SurfaeSample, src/Renderer.cpp:119-198 with SphereObject::ComputeSurfacePoint src/SettingData.cpp:227-233 — the
 * reference's second integrator (never selected: src/Renderer.cpp:234 asks for a U[0,1) draw >= 1.0), restated
 * literally; depth 0 entry.  max_bounces >= 0: an invocation at depth > max_bounces returns 0 without drawing. */
void rtmo_surface_sample(const rtm_sphere* spheres, size_t n, int mode, int max_bounces, const double org[3],
                         const double dir[3], rtmo_rng_fn rng, void* rng_ctx, double out_radiance[3],
                         rtmo_counters* counters);
void rtmo_surface_sample_stream(const rtm_sphere* spheres, size_t n, int mode, int max_bounces, const double org[3],
                                const double dir[3], uint64_t seed, uint32_t pixel, uint32_t sample,
                                double out_radiance[3], rtmo_counters* counters);

/* Same, drawing from the build RNG stream (seed, pixel, sample) — what rtm_path_trace_batch uses */
void rtmo_path_trace_stream(const rtm_sphere* spheres, size_t n, int mode, int max_bounces,
                            const double org[3], const double dir[3], uint64_t seed,
                            uint32_t pixel, uint32_t sample, double out_radiance[3],
                            rtmo_counters* counters);

/* src/Renderer.cpp:202-208 */
void rtmo_camera_basis(const rtm_settings* st, double cam_x[3], double cam_y[3], double cam_z[3],
                       double* fovx, double* fovy);
/* src/Renderer.cpp:227-232; sx, sy in 1..SS */
void rtmo_primary_dir(const rtm_settings* st, int x, int y, int sx, int sy, double out_dir[3]);

/* Build-defined RNG (spec in DESIGN.md §RNG). */
double rtmo_rng_u01(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t index);

/* src/Renderer.cpp:215-250.  out: (row_end-row_begin)*width*3 doubles.
 * structure 0: per-pixel parallel (OpenMP over all pixels of the tile, dynamic);
 * structure 1: reference structure (serial rows, "omp parallel for" over x inside each row).
 * threads <= 0: OpenMP default. */
int rtmo_render(const rtm_settings* st, const rtm_sphere* spheres, size_t n,
                const rtm_options* opt, double* out, rtmo_counters* counters, int threads,
                int structure);

/* Objects of any type (include/rtm.h: rtm_object; planes are a build-defined completion of png::PlaneObject):
 * the Intersect seam and the render, per-pixel parallel. */
int rtmo_intersect_object(const rtm_object* o, const double org[3], const double dir[3], int mode, double* out_t,
                          double out_normal[3]);
int rtmo_render_objects(const rtm_settings* st, const rtm_object* objects, size_t n, const rtm_options* opt,
                        double* out, rtmo_counters* counters, int threads);

/* The per-pixel loop of rtmo_render for a list of pixels (xy: n_pixels pairs x, y); out: n_pixels*3. */
int rtmo_render_pixels(const rtm_settings* st, const rtm_sphere* spheres, size_t n, const rtm_options* opt,
                       const int32_t* xy, size_t n_pixels, double* out, rtmo_counters* counters, int threads);

/* Radiance of one primary sample (pixel x,y, sub-pixel sx,sy in 1..SS, sample s) before the
 * /SS/SS/S normalisation — the value the GPU's per-sample path must reproduce. */
void rtmo_sample_radiance(const rtm_settings* st, const rtm_sphere* spheres, size_t n,
                          const rtm_options* opt, int x, int y, int sx, int sy, int s,
                          double out_radiance[3], rtmo_counters* counters);

/* src/Renderer.cpp:251-254 */
void rtmo_quantise(const double* image, size_t n_values, uint8_t* out);

/* FNV-1a-64 over raw little-endian doubles, one multiply per 8-byte word */
uint64_t rtmo_fnv1a64_f64(const double* v, size_t n);

/* BASELINE config 5 stress scene. */
void rtmo_make_stress_scene(uint64_t seed, size_t n, rtm_settings* st, rtm_sphere* spheres);

int rtmo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
