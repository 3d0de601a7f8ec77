// rtm_internal.h — C++ entry points behind the C ABI (include/rtm.h).  Not installed.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

#include "../../include/rtm.h"

namespace rtm {

void set_last_error(const std::string& s);
const char* last_error();

// device path (rtm_kernels.hip)
int device_count(int* count);
int num_variants();
int output_rows(const rtm_options* opt);
int release_scratch(int device);
int stream_release(int device, void* stream);
int wf_nearest_probe(int kind, const rtm_sphere* sp, size_t n, const double* org, const double* dir, size_t n_rays,
                     int32_t* out_id, double* out_t);
const char* variant_name(int v);
int scene_create(const rtm_sphere* sp, size_t n, int on_device, int device, rtm_scene** out);
int scene_create_objects(const rtm_object* objs, size_t n, int device, rtm_scene** out);
int intersect_objects_batch(const rtm_object* objs, const double* org, const double* dir, size_t n, int mode,
                            int32_t* out_hit, double* out_t, double* out_normal);
int scene_destroy(rtm_scene* sc);
size_t scene_size(const rtm_scene* sc);
int stream_status(int device, void* stream);
int scratch_bytes(const rtm_settings* st, const rtm_scene* scene, const rtm_options* opt, uint64_t out[6]);
int render_scene(const rtm_settings* st, const rtm_scene* scene, const rtm_options* opt, double* out64, float* out32,
                 uint8_t* out8, void* stream, rtm_stats* stats);
int render_device(const rtm_settings* st, const rtm_sphere* sp, size_t n, int spheres_on_device,
                  const rtm_options* opt, double* out64, float* out32, uint8_t* out8, void* stream,
                  rtm_stats* stats);
int render_host(const rtm_settings* st, const rtm_sphere* sp, size_t n, const rtm_options* opt,
                double* out64, float* out32, uint8_t* out8, rtm_stats* stats);
int render_host_objects(const rtm_settings* st, const rtm_object* objs, size_t n, const rtm_options* opt,
                        double* out64, float* out32, uint8_t* out8, rtm_stats* stats);
int path_trace_batch(const rtm_sphere* sp, size_t n, const rtm_options* opt, const double* org,
                     const double* dir, size_t n_rays, double* out, uint32_t* out_draws,
                     uint32_t* out_casts);
int surface_sample_batch(const rtm_sphere* sp, size_t n, const rtm_options* opt, const double* org, const double* dir,
                         size_t n_rays, double* out, uint32_t* out_draws, uint32_t* out_casts);
int intersect_batch(const rtm_sphere* sp, const double* org, const double* dir, size_t n, int mode,
                    int32_t* out_hit, double* out_t, double* out_normal);
int rng_batch(uint64_t seed, uint32_t pixel0, uint32_t n_pixels, uint32_t sample, uint32_t n_draws,
              double* out);
double rng_u01_host(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t index);
int component_bench(int which, const rtm_sphere* sp, size_t n, int reps, int blocks, int lds_pad,
                    double* cycles_per_rep);
int selfcheck(int kind, unsigned long long* mismatches);
int grid_nearest_probe(const rtm_sphere* sp, size_t n, const double* org, const double* dir, size_t n_rays, int32_t* out_id,
                       double* out_t, uint32_t* out_tests, uint32_t* out_steps, uint64_t* info);
int scene_facts_host(const rtm_sphere* sp, size_t n, uint64_t* facts);
int grid_build_host(const rtm_sphere* sp, size_t n, uint64_t* info, double* pads, uint32_t* ranges, size_t ranges_cap,
                    uint32_t* items, size_t items_cap, int32_t* big, size_t big_cap);
int fp64_peak(int waves_per_simd, double min_ms, double* tflops, double* kernel_ms);
int math_probe(int op, const double* a, const double* b, size_t n, double* out);
// the labelled tolerance row (rtm_kernels_tol.hip): launches for a planned rtm::RenderParams passed by bytes
int launch_tol(const void* render_params, size_t params_bytes, unsigned grid, size_t lds_pad, void* stream);
int tol_math_probe(int op, const double* a_dev, const double* b_dev, size_t n, double* out_dev);  // ops 32.. of math_probe

// host side (rtm_scene.cpp, rtm_image.cpp)
int scene_parse_json(const char* text, size_t len, int literal_loader, rtm_settings* st,
                     rtm_sphere* spheres, size_t capacity, size_t* n_spheres);
int scene_parse_json_objects(const char* text, size_t len, int literal_loader, rtm_settings* st,
                             rtm_object* objects, size_t capacity, size_t* n_objects);
int scene_load_json_objects(const char* path, int literal_loader, rtm_settings* st, rtm_object* objects,
                            size_t capacity, size_t* n_objects);
int scene_load_json(const char* path, int literal_loader, rtm_settings* st, rtm_sphere* spheres,
                    size_t capacity, size_t* n_spheres);
int scene_save_sample_json(const char* path);
int scene_make_stress(uint64_t seed, size_t n, rtm_settings* st, rtm_sphere* spheres);
int quantise(const double* image, size_t n_values, uint8_t* out);
int write_bmp(const char* filename, int w, int h, int comp, const void* data);
int write_jpg(const char* filename, int w, int h, int comp, const void* data, int quality);

}  // namespace rtm
