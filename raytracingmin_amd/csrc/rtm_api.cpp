// rtm_api.cpp — the extern "C" surface of include/rtm.h.  Thin: argument checks live with the
// implementations; no exception crosses the ABI.
#include <exception>

#include "../../include/rtm_debug.h"
#include "rtm_internal.h"

#define RTM_GUARD(call)                                  \
    try {                                                \
        return (call);                                   \
    } catch (const std::exception& e) {                  \
        rtm::set_last_error(e.what());                   \
        return RTM_ERR_INVALID_ARGUMENT;                 \
    } catch (...) {                                      \
        rtm::set_last_error("unknown exception");        \
        return RTM_ERR_INVALID_ARGUMENT;                 \
    }

extern "C" {

int rtm_abi_version(void) { return RTM_ABI_VERSION; }

const char* rtm_strerror(int status) {
    switch (status) {
        case RTM_OK: return "ok";
        case RTM_ERR_INVALID_ARGUMENT: return "invalid argument";
        case RTM_ERR_INVALID_SCENE: return "invalid scene";
        case RTM_ERR_IO: return "i/o error";
        case RTM_ERR_PARSE: return "json parse error";
        case RTM_ERR_NO_DEVICE: return "no HIP device";
        case RTM_ERR_HIP: return "HIP runtime error";
        case RTM_ERR_CAPACITY: return "buffer too small";
        case RTM_ERR_UNSUPPORTED: return "unsupported";
    }
    return "unknown status";
}
const char* rtm_last_error_detail(void) { return rtm::last_error(); }
int rtm_device_count(int* count) { RTM_GUARD(rtm::device_count(count)) }
int rtm_num_variants(void) { return rtm::num_variants(); }
int rtm_output_rows(const rtm_options* options) { return options ? rtm::output_rows(options) : 0; }
int rtm_release_scratch(int device) { RTM_GUARD(rtm::release_scratch(device)) }
int rtm_stream_release(int device, void* stream) { RTM_GUARD(rtm::stream_release(device, stream)) }
int rtm_scratch_bytes(const rtm_settings* settings, const rtm_scene* scene, const rtm_options* options, uint64_t out_bytes[6]) {
    RTM_GUARD(rtm::scratch_bytes(settings, scene, options, out_bytes))
}
const char* rtm_variant_name(int variant) { return rtm::variant_name(variant); }

int rtm_scene_create(const rtm_sphere* spheres, size_t n_spheres, int spheres_on_device, int device,
                     rtm_scene** out_scene) {
    RTM_GUARD(rtm::scene_create(spheres, n_spheres, spheres_on_device, device, out_scene))
}
int rtm_scene_create_objects(const rtm_object* objects, size_t n_objects, int device, rtm_scene** out_scene) {
    RTM_GUARD(rtm::scene_create_objects(objects, n_objects, device, out_scene))
}
int rtm_intersect_objects_batch(const rtm_object* objects, const double* org, const double* dir, size_t n, int mode,
                                int32_t* out_hit, double* out_t, double* out_normal) {
    RTM_GUARD(rtm::intersect_objects_batch(objects, org, dir, n, mode, out_hit, out_t, out_normal))
}
int rtm_scene_load_json_objects(const char* path, int literal_loader, rtm_settings* settings,
                                rtm_object* objects, size_t capacity, size_t* n_objects) {
    RTM_GUARD(rtm::scene_load_json_objects(path, literal_loader, settings, objects, capacity, n_objects))
}
int rtm_scene_parse_json_objects(const char* text, size_t len, int literal_loader, rtm_settings* settings,
                                 rtm_object* objects, size_t capacity, size_t* n_objects) {
    RTM_GUARD(rtm::scene_parse_json_objects(text, len, literal_loader, settings, objects, capacity, n_objects))
}
int rtm_scene_destroy(rtm_scene* scene) { RTM_GUARD(rtm::scene_destroy(scene)) }
size_t rtm_scene_size(const rtm_scene* scene) { return rtm::scene_size(scene); }
int rtm_stream_status(int device, void* stream) { RTM_GUARD(rtm::stream_status(device, stream)) }
int rtm_render_scene(const rtm_settings* settings, const rtm_scene* scene, const rtm_options* options,
                     double* out_f64_dev, float* out_f32_dev, uint8_t* out_u8_dev, void* stream,
                     rtm_stats* stats) {
    RTM_GUARD(rtm::render_scene(settings, scene, options, out_f64_dev, out_f32_dev, out_u8_dev, stream, stats))
}
int rtm_render_device(const rtm_settings* settings, const rtm_sphere* spheres, size_t n_spheres,
                      int spheres_on_device, const rtm_options* options, double* out_f64_dev,
                      float* out_f32_dev, uint8_t* out_u8_dev, void* stream, rtm_stats* stats) {
    RTM_GUARD(rtm::render_device(settings, spheres, n_spheres, spheres_on_device, options,
                                 out_f64_dev, out_f32_dev, out_u8_dev, stream, stats))
}
int rtm_render(const rtm_settings* settings, const rtm_sphere* spheres, size_t n_spheres,
               const rtm_options* options, double* out_f64, float* out_f32, uint8_t* out_u8,
               rtm_stats* stats) {
    RTM_GUARD(rtm::render_host(settings, spheres, n_spheres, options, out_f64, out_f32, out_u8, stats))
}
int rtm_render_objects(const rtm_settings* settings, const rtm_object* objects, size_t n_objects,
                       const rtm_options* options, double* out_f64, float* out_f32, uint8_t* out_u8,
                       rtm_stats* stats) {
    RTM_GUARD(rtm::render_host_objects(settings, objects, n_objects, options, out_f64, out_f32, out_u8, stats))
}
int rtm_path_trace_batch(const rtm_sphere* spheres, size_t n_spheres, const rtm_options* options,
                         const double* org, const double* dir, size_t n_rays, double* out_radiance,
                         uint32_t* out_draws, uint32_t* out_casts) {
    RTM_GUARD(rtm::path_trace_batch(spheres, n_spheres, options, org, dir, n_rays, out_radiance,
                                    out_draws, out_casts))
}
int rtm_surface_sample_batch(const rtm_sphere* spheres, size_t n_spheres, const rtm_options* options, const double* org,
                             const double* dir, size_t n_rays, double* out_radiance, uint32_t* out_draws, uint32_t* out_casts) {
    RTM_GUARD(rtm::surface_sample_batch(spheres, n_spheres, options, org, dir, n_rays, out_radiance, out_draws, out_casts))
}
int rtm_intersect_batch(const rtm_sphere* spheres, const double* org, const double* dir, size_t n,
                        int mode, int32_t* out_hit, double* out_t, double* out_normal) {
    RTM_GUARD(rtm::intersect_batch(spheres, org, dir, n, mode, out_hit, out_t, out_normal))
}
double rtm_rng_u01(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t index) {
    return rtm::rng_u01_host(seed, pixel, sample, index);
}
int rtm_rng_batch(uint64_t seed, uint32_t pixel0, uint32_t n_pixels, uint32_t sample,
                  uint32_t n_draws, double* out) {
    RTM_GUARD(rtm::rng_batch(seed, pixel0, n_pixels, sample, n_draws, out))
}
/* ---- test and diagnostic hooks: declared in include/rtm_debug.h, not part of the drop-in boundary ---- */
int rtm_debug_math_probe(int op, const double* a, const double* b, size_t n, double* out) {
    RTM_GUARD(rtm::math_probe(op, a, b, n, out))
}

int rtm_debug_component_bench(int which, const rtm_sphere* sp, size_t n, int reps, int blocks, int lds_pad,
                              double* cycles_per_rep) {
    RTM_GUARD(rtm::component_bench(which, sp, n, reps, blocks, lds_pad, cycles_per_rep))
}

int rtm_debug_wf_nearest(int kind, const rtm_sphere* sp, size_t n, const double* org, const double* dir, size_t n_rays,
                         int32_t* out_id, double* out_t) {
    RTM_GUARD(rtm::wf_nearest_probe(kind, sp, n, org, dir, n_rays, out_id, out_t))
}
int rtm_debug_selfcheck(int kind, unsigned long long* mismatches) { RTM_GUARD(rtm::selfcheck(kind, mismatches)) }
int rtm_debug_grid_nearest(const rtm_sphere* sp, size_t n, const double* org, const double* dir, size_t n_rays,
                           int32_t* out_id, double* out_t, uint32_t* out_tests, uint32_t* out_steps, uint64_t* info) {
    RTM_GUARD(rtm::grid_nearest_probe(sp, n, org, dir, n_rays, out_id, out_t, out_tests, out_steps, info))
}
int rtm_debug_grid_build(const rtm_sphere* sp, size_t n, uint64_t* info, double* pads, uint32_t* ranges, size_t ranges_cap,
                         uint32_t* items, size_t items_cap, int32_t* big, size_t big_cap) {
    RTM_GUARD(rtm::grid_build_host(sp, n, info, pads, ranges, ranges_cap, items, items_cap, big, big_cap))
}
int rtm_debug_scene_facts(const rtm_sphere* sp, size_t n, uint64_t facts[2]) { RTM_GUARD(rtm::scene_facts_host(sp, n, facts)) }
int rtm_debug_fp64_peak(int waves_per_simd, double min_ms, double* tflops, double* kernel_ms) {
    RTM_GUARD(rtm::fp64_peak(waves_per_simd, min_ms, tflops, kernel_ms))
}

int rtm_scene_load_json(const char* path, int literal_loader, rtm_settings* settings,
                        rtm_sphere* spheres, size_t capacity, size_t* n_spheres) {
    RTM_GUARD(rtm::scene_load_json(path, literal_loader, settings, spheres, capacity, n_spheres))
}
int rtm_scene_parse_json(const char* text, size_t len, int literal_loader, rtm_settings* settings,
                         rtm_sphere* spheres, size_t capacity, size_t* n_spheres) {
    RTM_GUARD(rtm::scene_parse_json(text, len, literal_loader, settings, spheres, capacity, n_spheres))
}
int rtm_scene_save_sample_json(const char* path) { RTM_GUARD(rtm::scene_save_sample_json(path)) }
int rtm_scene_make_stress(uint64_t seed, size_t n, rtm_settings* settings, rtm_sphere* spheres) {
    RTM_GUARD(rtm::scene_make_stress(seed, n, settings, spheres))
}
int rtm_quantise(const double* image, size_t n_values, uint8_t* out) {
    RTM_GUARD(rtm::quantise(image, n_values, out))
}
int rtm_write_bmp(const char* filename, int w, int h, int comp, const void* data) {
    try {
        return rtm::write_bmp(filename, w, h, comp, data);
    } catch (...) {
        return 0;
    }
}
int rtm_write_jpg(const char* filename, int w, int h, int comp, const void* data, int quality) {
    try {
        return rtm::write_jpg(filename, w, h, comp, data, quality);
    } catch (...) {
        return 0;
    }
}

}  // extern "C"
