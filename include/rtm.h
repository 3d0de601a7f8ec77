/*
 * rtm.h — C ABI of the MI355X-native path-tracing hot path for RaytracingMin scenes.
 *
 * This is the drop-in boundary.  The reference has no FFI; the seams this ABI
 * replaces are (paths relative to the reference checkout):
 *
 *   png::Renderer::Renderer(SettingData&)              src/Renderer.h:11, src/Renderer.cpp:20-23
 *   void png::Renderer::Render(std::string fileName)   src/Renderer.h:13, src/Renderer.cpp:200-258
 *   png::vec3 png::PathTracing(Ray, SettingData&, fn)  src/Renderer.cpp:57-117
 *   bool png::SphereObject::Intersect(...)             src/SettingData.cpp:197-226
 *   png::LoadData::LoadData(std::string)               src/SettingData.cpp:6-12,129-186
 *   stbi_write_bmp / stbi_write_jpg call sites         src/Renderer.cpp:251-257
 *
 * Conventions: plain C types only; the caller owns every buffer; nothing allocated inside is
 * handed out; every entry point returns RTM_OK (0) or a negative rtm_status, never throws.
 * "Device" pointers are HIP device pointers on the GPU selected in rtm_options.device; streams are
 * hipStream_t passed as void*.
 *
 * Threading: every entry point may be called from any host thread.  Renders on DIFFERENT
 * (device, stream) pairs run concurrently and share nothing; calls that name the SAME (device, stream)
 * are serialised by the library for the time it takes to enqueue them (the work itself is ordered by
 * the stream), so the work buffers the library keeps per (device, stream) are never handed to two
 * calls at once.  rtm_release_scratch waits for the renders it affects.  rtm_last_error_detail is
 * per thread.  Diagnostic hooks (rtm_debug_*) are declared in rtm_debug.h, not here.
 */
#ifndef RTM_H
#define RTM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTM_ABI_VERSION 5

typedef enum rtm_status {
    RTM_OK = 0,
    RTM_ERR_INVALID_ARGUMENT = -1, /* null pointer, non-positive size, bad enum value          */
    RTM_ERR_INVALID_SCENE = -2,    /* unsupported objectType, malformed numbers                 */
    RTM_ERR_IO = -3,               /* file cannot be opened / written                           */
    RTM_ERR_PARSE = -4,            /* JSON syntax or type error (reference: nlohmann exception) */
    RTM_ERR_NO_DEVICE = -5,        /* no HIP device / HIP runtime failure at init               */
    RTM_ERR_HIP = -6,              /* a HIP call failed; see rtm_last_error_detail()            */
    RTM_ERR_CAPACITY = -7,         /* caller buffer too small                                   */
    RTM_ERR_UNSUPPORTED = -8       /* valid request this build cannot serve                     */
} rtm_status;

/* png::Camera — src/SettingData.h:43-46.  fov is a float and is a tangent scale, not degrees
 * (src/Renderer.cpp:207-208). */
typedef struct rtm_camera {
    double origin[3];
    double target[3];
    double up[3];
    float fov;
    float _pad;
} rtm_camera;

/* png::SphereObject + png::Material flattened — src/SettingData.h:8-17,25-32.
 * radius is a float exactly like SphereObject::m_size. */
typedef struct rtm_sphere {
    double center[3];
    double color[3];
    double emission[3];
    float radius;
    float _pad;
} rtm_sphere;

/* Any object of png::SettingData::object — src/SettingData.h:18-42 — for scenes that are not all spheres.
 * type 1: png::SphereObject (position = centre, size = radius as a float, like m_size).
 * type 2: png::PlaneObject(position, up, target, width, mat) — src/SettingData.cpp:235-242.  The reference
 *   constructs m_normal = Normalize(target - position) and m_right = Normalize(Cross(m_normal, up)) * 0.5 * width
 *   but leaves Intersect unfinished (it falls off its end, :243-246) and never builds one (the loader only
 *   knows type 1).  This build completes it as the finite square the constructor describes — centre
 *   `position`, side `width`, spanned by m_right and Cross(m_right, m_normal):
 *       dn = Dot(m_normal, dir);           if |dn| < FLT_EPSILON: no hit            (the reference's own line, :244)
 *       t  = Dot(m_normal, position - org) / dn;      if !(t > 0.001): no hit       (the sphere's near threshold)
 *       d  = org + dir * t - position;     if |Dot(d, m_right)| > Dot(m_right, m_right)
 *                                          or |Dot(d, m_upv)|   > Dot(m_upv, m_upv): no hit
 *       out_dis = t; out_normal = m_normal  (the caller orients it against the ray, src/Renderer.cpp:82-83)
 *   A BUILD-DEFINED semantics: the reference has none to match; oracle and device agree bit for bit. */
enum { RTM_OBJECT_SPHERE = 1, RTM_OBJECT_PLANE = 2 };
typedef struct rtm_object {
    int32_t type;        /* RTM_OBJECT_*                                                       */
    float size;          /* sphere radius (float, SphereObject::m_size); unused for planes     */
    double position[3];  /* sphere centre / plane centre                                        */
    double color[3];
    double emission[3];
    double up[3];        /* plane only                                                          */
    double target[3];    /* plane only: the normal points from position towards target          */
    double width;        /* plane only: side of the square (a double, like the constructor's)   */
} rtm_object;

/* png::SettingData minus the object vector — src/SettingData.h:47-51. */
typedef struct rtm_settings {
    int32_t width, height, samples, super_samples;
    rtm_camera camera;
} rtm_settings;

enum { RTM_MODE_LITERAL = 0,  /* L0: HEAD as shipped (normal lost, recursion gets ::rand)       */
       RTM_MODE_REPAIRED = 1, /* L1: the three one-line defects fixed    */
       /* flag, OR-ed into mode: sin/cos of src/Renderer.cpp:93-94 return exactly what the HOST's
        * libm returns.  The argument takes 2^23 values, so on first use the library evaluates them
        * all on both sides (a fraction of a second, once per device) and keeps the one-ulp
        * differences in a 4 MB table.  Costs ~2 % of a frame; makes GPU and CPU-libm renders agree
        * bit for bit even where one-ulp differences are amplified over many bounces.            */
       RTM_MODE_HOST_TRIG = 0x100,
       /* flag, OR-ed into mode (diagnostic): a render WITH rtm_stats also counts the Intersect evaluations it makes
        * (rtm_stats.object_tests).  The exhaustive kernels make n_objects per cast by construction; the uniform-grid kernel
        * (variant 17) runs a counting instantiation that is a few per cent slower, so time a frame without the flag.     */
       RTM_MODE_COUNT_TESTS = 0x200,
       /* flag, OR-ed into mode: the INTEGRATOR is png::SurfaeSample (src/Renderer.cpp:119-198 with
        * SphereObject::ComputeSurfacePoint, src/SettingData.cpp:227-233) instead of png::PathTracing — the branch
        * src/Renderer.cpp:234-236 takes when a U[0,1) draw is >= 1.0, i.e. never in the reference; the function is defined,
        * has external linkage and takes the same injectable generator, so it is restated (oracle first) and served by a
        * general per-object kernel, not by the tuned ones.  max_bounces >= 0: an invocation at depth > max_bounces returns
        * 0 without drawing (the reference has no bound); records for up to 16 + 960 levels, beyond which the call fails
        * with RTM_ERR_UNSUPPORTED like a PathTracing render does. */
       RTM_MODE_SURFACE_SAMPLE = 0x400 };

typedef struct rtm_options {
    int32_t mode;         /* RTM_MODE_*                                                        */
    int32_t max_bounces;  /* <0: unlimited (reference recursion); k>=0: cast k+1 returns
                             emission on hit without drawing (build extension)     */
    uint64_t seed;        /* RNG seed (build-defined counter RNG, rtm_rng_u01 below)           */
    int32_t row_begin;    /* first image row of this tile                                      */
    int32_t row_end;      /* one past the last row; full image = [0, height)                   */
    int32_t device;       /* HIP device ordinal                                                */
    int32_t variant;      /* kernel variant, 0 = default (see rtm_variant_name)                */
    int32_t band_count;   /* interleaved bands for multi-GPU load balance: when > 1 the call
                             renders only the 8-row bands b = band_index, band_index + band_count,
                             ... of [row_begin,row_end) (band b = rows row_begin + 8b ..) and
                             stores them back to back; 0 or 1 = every band                     */
    int32_t band_index;   /* 0 <= band_index < band_count                                       */
} rtm_options;

/* Per-render counters (sum over the rendered tile); filled when the pointer is non-null. */
typedef struct rtm_stats {
    uint64_t samples;     /* primary samples traced = output rows*width*SS*SS*S                */
    uint64_t casts;       /* PathTracing invocations (ray casts)                               */
    uint64_t bounces;     /* casts that continued (RR passed)                                  */
    uint64_t draws;       /* RNG draws consumed                                                */
    double kernel_ms;     /* device time of the render kernel(s), HIP events on the stream     */
    int32_t variant;      /* the kernel variant that ran (rtm_options.variant resolved; see
                             rtm_variant_name), and                                              */
    int32_t split;        /* waves per 8x8 tile of the sample split (1 = not split)            */
    uint64_t object_tests; /* Object::Intersect evaluations (src/Renderer.cpp:66): casts x n_objects for the
                             exhaustive kernels; for the uniform-grid kernel the count of RTM_MODE_COUNT_TESTS, else 0 */
} rtm_stats;

/* A scene flattened to the kernels' layout and resident on one device (opaque).  Created once,
 * used by any number of renders on any stream of that device, destroyed by the caller after the
 * last render that uses it has been ENQUEUED: rtm_scene_destroy does not wait for that work.  When the
 * renders that named the scene have all finished, its device memory is freed in the call; otherwise the
 * scene is parked and freed by a later library call (any render, create or destroy) that finds them
 * finished, or by rtm_release_scratch, which waits.  Other streams of the device are never waited for. */
typedef struct rtm_scene rtm_scene;

/* ---- library ---- */
int rtm_abi_version(void);
const char* rtm_strerror(int status);
const char* rtm_last_error_detail(void);   /* thread-local text of the last failure            */
int rtm_device_count(int* count);          /* RTM_ERR_NO_DEVICE if the HIP runtime has none    */
int rtm_num_variants(void);
const char* rtm_variant_name(int variant);

/* Rows a call with these options renders and stores (row_end - row_begin unless banded). */
int rtm_output_rows(const rtm_options* options);
/* The library keeps its large work buffers (per device and stream, grown on demand), the scene cache
 * of rtm_render_device and the RTM_MODE_HOST_TRIG tables between calls; this frees them for one
 * device, or for all with device < 0.  Waits for the device to go idle.  Scenes made by
 * rtm_scene_create are the caller's and stay.
 * What "large" means: the sample split of a launch's last tiles keeps 2 KiB per (split tile, deferred sample) —
 * 1.6 GB for the headline frame, capped at 24 GiB per stream (a launch whose terms would not fit splits fewer
 * tiles, or none), and in-wave sample stealing 1.8 KiB x (2 sqrt(spp) + 5) per whole tile (3.8 GB for the headline
 * frame; without room the launch runs without it); unlimited-depth renders keep two pooled record stacks per lane
 * (5.5 GB for a 1080p frame); the large-scene grid kernel 32 B per sample of a launch (17.0 GB for a 1080p frame at
 * 256 spp: above the 16 GiB budget, so that frame is rendered in two launches — beyond the budget, or what the device
 * gives, a frame is cut into several launches of as many tiles as fit); the exhaustive
 * large-scene pipeline ~300 B per pixel plus 4 B per pixel and record level. */
int rtm_release_scratch(int device);
/* What a render with these arguments will ask of those work buffers, in bytes, before anything is allocated:
 * out_bytes[0] the total, [1] per-sample terms (sample split of a launch's last tiles / the grid kernel's term buffer:
 * one launch's worth, at most the 16 GiB budget), [2] pooled hit records (unlimited depth or a cap of 16 and more),
 * [3] the exhaustive large-scene pipeline's path state, [4] the rows of in-wave sample stealing, [5] the table of primary
 * directions a pre-pass leaves for the default kernels (1.5 KiB per tile and sub-pixel, at most 4 GiB).  The buffers are per
 * (device, stream), grown on demand and kept until rtm_release_scratch / rtm_stream_release; a stream that has rendered
 * larger frames already holds more.  Where the device cannot give an OPTIONAL buffer (terms, stolen rows) the render
 * runs without the feature or in more launches — same image. */
int rtm_scratch_bytes(const rtm_settings* settings, const rtm_scene* scene, const rtm_options* options,
                      uint64_t out_bytes[6]);
/* The same for ONE stream: waits for that stream's queued work (and, like rtm_release_scratch, for every render call
 * that is being enqueued at that moment), frees the buffers and the sticky status word the
 * library keeps for (device, stream) and forgets the pair.  Call it before destroying a stream that has rendered
 * (a later stream may be given the same handle and would inherit the context otherwise).  An overflow that was
 * never reported is returned here (RTM_ERR_UNSUPPORTED), like rtm_stream_status would. */
int rtm_stream_release(int device, void* stream);

/* ---- scene lifetime: png::SettingData::object (src/SettingData.h:47-51) on the device ----
 * rtm_scene_create flattens `spheres` (HOST pointer, or a DEVICE pointer on `device` when
 * spheres_on_device != 0) and uploads the tables; it returns when they are resident, so the caller's
 * array may be freed at once.  A scene belongs to one device.
 * For an all-sphere scene of 64 spheres or more (not counting those that span the scene, like the Cornell walls) the
 * call also builds, on the host (~25 ms per 100 000 spheres plus the upload), a uniform grid over the spheres (~32 B x 6 per sphere +
 * 8 B per cell, ~2 cells per sphere): renders then find the reference loop's nearest hit (src/Renderer.cpp:58-73:
 * same object, same distance, same image) through the grid instead of testing every sphere for every cast
 * (rtm_options.variant 17; variant 0 picks it when the camera is within about two scene diagonals of the scene and no
 * diffuse sphere encloses the scene from farther away than that — its bounces would start where the grid cannot serve them;
 * variants 3 / 14 and, from 512 spheres, 12 are the exhaustive kernels). */
int rtm_scene_create(const rtm_sphere* spheres, size_t n_spheres, int spheres_on_device, int device,
                     rtm_scene** out_scene);
/* The same for a list of objects of any type (spheres and planes in the reference's vector order, which
 * decides ties: the lowest index wins).  Scenes that contain a plane are rendered by the default chunked kernel
 * up to 255 objects (rtm_options.variant 0, 2 or 9) and by the general per-object kernel beyond (variant 1, which
 * may also be asked for); the other variants know spheres only and refuse them.  HOST pointer. */
int rtm_scene_create_objects(const rtm_object* objects, size_t n_objects, int device, rtm_scene** out_scene);
int rtm_scene_destroy(rtm_scene* scene);
size_t rtm_scene_size(const rtm_scene* scene);

/* ---- the hot path: Renderer::Render's pixel/sample loop (src/Renderer.cpp:215-250) ----
 * Renders rows [row_begin,row_end) of the image into caller-owned DEVICE buffers; any of the
 * three outputs may be null.  Layout is the reference's: row-major, row 0 first, RGB interleaved
 * (src/Renderer.cpp:246-248).  out_f64 holds Renderer::image bit-for-bit semantics (double);
 * out_f32 is the same value rounded to float (the float3 accumulation buffer); out_u8 is the
 * reference's 8-bit quantisation (src/Renderer.cpp:251-254).
 *
 * rtm_render_scene with stats == NULL only ENQUEUES work on `stream` and returns: no copy from pageable
 * memory and no wait — except that the FIRST call on a (device, stream) pair sets up its context, and a call
 * that needs a larger work buffer than the pair has (see rtm_release_scratch) grows it, which allocates and
 * waits for that stream's queued work once; steady-state calls do neither.  Scenes that have a grid
 * (rtm_scene_create) are two launches per frame like any other; scenes of 512 spheres or more without one — rtm_render_device on a device
 * array, scenes the grid declines (planes, non-finite geometry), a far-away camera, variant 12 by name — they run a
 * pipeline of two launches per ray cast of the slowest pixel: with a depth cap such that samples x superSamples^2 x
 * (max_bounces + 1) <= 16 384 every trip that could be needed is enqueued at once (those after the last active pixel
 * has finished fall through) and the call returns like any other; with unlimited depth or a longer budget the host
 * follows the device-resident active-pixel count one batch of trips behind and the call returns when it has seen the
 * count at zero, i.e. it blocks for about the duration of the render.
 * With stats != NULL the call synchronises the stream to read the counters and the timing.
 *
 * Hit records of paths deeper than the on-chip levels spill to a pooled buffer (capacity: 976
 * bounces per path); exceeding it truncates that path and raises the stream's sticky overflow flag:
 *   - with stats != NULL the call itself returns RTM_ERR_UNSUPPORTED;
 *   - with stats == NULL it is reported by rtm_stream_status(device, stream) — call it after
 *     synchronising the stream — and by the NEXT render enqueued on that stream once the flag has
 *     reached the host (RTM_ERR_UNSUPPORTED, nothing enqueued).  The flag is cleared by the call
 *     that reports it. */
int rtm_render_scene(const rtm_settings* settings, const rtm_scene* scene, const rtm_options* options,
                     double* out_f64_dev, float* out_f32_dev, uint8_t* out_u8_dev, void* stream,
                     rtm_stats* stats);
/* RTM_OK, or RTM_ERR_UNSUPPORTED when a render enqueued on (device, stream) since the last report
 * overflowed its hit records.  Waits for the stream's queued work (hipStreamSynchronize). */
int rtm_stream_status(int device, void* stream);

/* The same render from a sphere ARRAY.  spheres is a HOST pointer (tiny for shipped scenes) unless
 * spheres_on_device != 0.  Host arrays are looked up in a small per-device cache keyed by their
 * content, so repeated calls with an unchanged scene behave like rtm_render_scene (the first call
 * with new content uploads it and waits for the upload); device arrays are flattened on the stream
 * per call. */
int rtm_render_device(const rtm_settings* settings, const rtm_sphere* spheres, size_t n_spheres,
                      int spheres_on_device, const rtm_options* options, double* out_f64_dev,
                      float* out_f32_dev, uint8_t* out_u8_dev, void* stream, rtm_stats* stats);

/* Blocking convenience: same render, HOST output buffers (any may be null). */
int rtm_render(const rtm_settings* settings, const rtm_sphere* spheres, size_t n_spheres,
               const rtm_options* options, double* out_f64, float* out_f32, uint8_t* out_u8,
               rtm_stats* stats);

/* The blocking render for a list of objects of any type (rtm_object: spheres and planes), HOST buffers. */
int rtm_render_objects(const rtm_settings* settings, const rtm_object* objects, size_t n_objects,
                       const rtm_options* options, double* out_f64, float* out_f32, uint8_t* out_u8,
                       rtm_stats* stats);

/* ---- per-ray seam: png::PathTracing (src/Renderer.cpp:57-117) for a batch of rays on device.
 * Ray i uses the RNG stream keyed (seed, pixel = i, sample = 0).  Host buffers.
 * org/dir: n*3 doubles; out_radiance: n*3; out_draws/out_casts: n (nullable). */
int rtm_path_trace_batch(const rtm_sphere* spheres, size_t n_spheres, const rtm_options* options,
                         const double* org, const double* dir, size_t n_rays,
                         double* out_radiance, uint32_t* out_draws, uint32_t* out_casts);

/* ---- per-ray seam of the reference's second integrator: png::SurfaeSample (src/Renderer.cpp:119-198, entered at depth 0)
 * for a batch of rays on device, like rtm_path_trace_batch: ray i uses the RNG stream keyed (seed, pixel = i, sample = 0);
 * host buffers; out_draws = generator calls, out_casts = nearest-hit loops.  options->max_bounces >= 0 bounds the recursion
 * (RTM_MODE_SURFACE_SAMPLE above); unbounded, a recursion deeper than 1 024 levels fails with RTM_ERR_UNSUPPORTED. */
int rtm_surface_sample_batch(const rtm_sphere* spheres, size_t n_spheres, const rtm_options* options,
                             const double* org, const double* dir, size_t n_rays,
                             double* out_radiance, uint32_t* out_draws, uint32_t* out_casts);

/* ---- per-call seam: SphereObject::Intersect (src/SettingData.cpp:197-226) on device.
 * Pair i tests ray i against sphere i.  Host buffers.  out_t/out_normal are left untouched
 * (caller-initialised) where out_hit[i]==0; in RTM_MODE_LITERAL out_normal is never written. */
int rtm_intersect_batch(const rtm_sphere* spheres, const double* org, const double* dir,
                        size_t n, int mode, int32_t* out_hit, double* out_t, double* out_normal);

/* The same seam for objects of any type (rtm_object): pair i tests ray i against object i. */
int rtm_intersect_objects_batch(const rtm_object* objects, const double* org, const double* dir, size_t n, int mode,
                                int32_t* out_hit, double* out_t, double* out_normal);

/* ---- the build-defined RNG (reference seeds std::mt19937 from random_device:
 * src/Renderer.cpp:210-213, so there is no reference stream to match).  Host-side evaluation of
 * draw `index` of stream (seed, pixel, sample); device side must agree (rtm_rng_batch). */
double rtm_rng_u01(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t index);
int rtm_rng_batch(uint64_t seed, uint32_t pixel0, uint32_t n_pixels, uint32_t sample,
                  uint32_t n_draws, double* out /* n_pixels*n_draws, host, computed on device */);

/* ---- host side: scene loading (png::LoadData, src/SettingData.cpp:6-12,129-186) ----
 * Reads the reference's JSON schema.  literal_loader != 0 reproduces HEAD's position bug
 * (src/SettingData.cpp:165-167: centre = (json_z, 0, 0)).  Shipped files load unchanged:
 * missing "00 objectType" => sphere, objects without "00 position" skipped, "00 sample" accepted
 * for "00 samples", missing samples/superSamples => 10 / 1.
 * spheres may be NULL with capacity 0 to query *n_spheres. */
int rtm_scene_load_json(const char* path, int literal_loader, rtm_settings* settings,
                        rtm_sphere* spheres, size_t capacity, size_t* n_spheres);
int rtm_scene_parse_json(const char* text, size_t len, int literal_loader, rtm_settings* settings,
                         rtm_sphere* spheres, size_t capacity, size_t* n_spheres);
/* The same loaders for files that may hold planes ("00 objectType": 2 with "00 position", "01 size" = width,
 * "03 up", "04 target", "02 material" — a build-defined extension of the schema; the sphere-only loaders
 * above reject such a file with RTM_ERR_INVALID_SCENE). */
int rtm_scene_load_json_objects(const char* path, int literal_loader, rtm_settings* settings,
                                rtm_object* objects, size_t capacity, size_t* n_objects);
int rtm_scene_parse_json_objects(const char* text, size_t len, int literal_loader, rtm_settings* settings,
                                 rtm_object* objects, size_t capacity, size_t* n_objects);
/* LoadData::SaveSampleJson (src/SettingData.cpp:14-24,100-127): 960x540, samples 10, SS 4. */
int rtm_scene_save_sample_json(const char* path);
/* Stress scene of BASELINE config 5: SplitMix64(seed), n spheres. */
int rtm_scene_make_stress(uint64_t seed, size_t n, rtm_settings* settings, rtm_sphere* spheres);

/* ---- host side: image output (src/Renderer.cpp:251-257) ----
 * rtm_quantise: u8 = (unsigned char)(255 * min(v, 1.0)) per component. */
int rtm_quantise(const double* image, size_t n_values, uint8_t* out);
/* stb_image_write-compatible signatures (w, h, comp = 3, RGB rows top-down). Return 1 on success
 * like stb (0 on failure) so real stb can replace them. */
int rtm_write_bmp(const char* filename, int w, int h, int comp, const void* data);
int rtm_write_jpg(const char* filename, int w, int h, int comp, const void* data, int quality);

#ifdef __cplusplus
}
#endif
#endif /* RTM_H */
