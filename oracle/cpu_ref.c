#define _GNU_SOURCE /* sincos() */
/*
 * oracle/cpu_ref.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see cpu_ref.h).
 *
 * CPU restatement of the reference's hot path in plain C, fp64, with every float "island" of the
 * reference kept.  Each function cites the reference lines it follows.
 * Build with -ffp-contract=off: x86-64 -O2 emits no FMA for the reference either.
 */
#include "cpu_ref.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct v3 {
    double x, y, z;
} v3;

static inline v3 v3_make(double x, double y, double z) {
    v3 r = {x, y, z};
    return r;
}
static inline v3 v3_from(const double a[3]) { return v3_make(a[0], a[1], a[2]); }
static inline void v3_to(v3 a, double out[3]) {
    out[0] = a.x;
    out[1] = a.y;
    out[2] = a.z;
}
/* src/Ray.h:15-35 */
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, double s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_div(v3 a, double s) { return v3_make(a.x / s, a.y / s, a.z / s); }

/* src/Ray.h:61-63 */
static inline double dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* src/Ray.h:64-66 — note the middle component is (-a.x)*b.z + a.z*b.x */
static inline v3 cross3(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x);
}
/* src/Ray.h:67-69 — std::sqrtf on a double argument: round to float, float sqrt, widen (Q9) */
static inline double magnitude3(v3 a) {
    float len2 = (float)(a.x * a.x + a.y * a.y + a.z * a.z);
    return (double)sqrtf(len2);
}
/* src/Ray.h:70-72 — three true divisions */
static inline v3 normalize3(v3 a) { return v3_div(a, magnitude3(a)); }

double rtmo_dot(const double a[3], const double b[3]) { return dot3(v3_from(a), v3_from(b)); }
void rtmo_cross(const double a[3], const double b[3], double out[3]) {
    v3_to(cross3(v3_from(a), v3_from(b)), out);
}
double rtmo_magnitude(const double a[3]) { return magnitude3(v3_from(a)); }
void rtmo_normalize(const double a[3], double out[3]) { v3_to(normalize3(v3_from(a)), out); }

/* The host libm's sin / cos exactly as the path calls them (src/Renderer.cpp:93-94 restated below),
 * for checking the device's sincos over every argument the RNG can produce.  "As the path calls them":
 * g++ -O2 compiles the reference's adjacent cos(r1) and sin(r1) into ONE glibc sincos() call, and glibc
 * 2.35's sincos() differs from its sin()/cos() by one ulp on 0.07 % of these arguments (5 713 sines and
 * 5 683 cosines of 2^23) — so sincos() is called explicitly here and in path_trace, instead of leaving
 * the choice to the optimiser. */
void rtmo_sin_cos_array(const double* x, size_t n, double* out_sin, double* out_cos) {
    for (size_t i = 0; i < n; ++i) sincos(x[i], &out_sin[i], &out_cos[i]);
}

/* src/SettingData.h:14-16 — float return: the max is rounded to float (Q10) */
float rtmo_kd(const rtm_sphere* s) {
    double m = s->color[0] < s->color[1] ? s->color[1] : s->color[0]; /* std::max(x, y) */
    m = m < s->color[2] ? s->color[2] : m;                            /* std::max(.., z) */
    return (float)m;
}
/* src/SettingData.h:11-13 — color / kd(), kd promoted back to double */
void rtmo_color_kd(const rtm_sphere* s, double out[3]) {
    double kd = (double)rtmo_kd(s);
    out[0] = s->color[0] / kd;
    out[1] = s->color[1] / kd;
    out[2] = s->color[2] / kd;
}

/* src/SettingData.cpp:197-226 */
static int intersect(const rtm_sphere* s, v3 org, v3 dir, int mode, double* out_t,
                     v3* out_normal, rtmo_counters* c) {
    const v3 p_o = v3_sub(v3_from(s->center), org); /* :198 */
    const double b = dot3(p_o, dir);                /* :199 */
    const float r2f = s->radius * s->radius;        /* :200 float product (Q11) */
    const double D4 = b * b - dot3(p_o, p_o) + (double)r2f;
    if (c) c->sphere_tests++;
    if (D4 < 0.0) return 0; /* :202 */
    if (c) c->sphere_tests_d4++;
    const double sqrt_D4 = sqrt(D4); /* :205 */
    const double t1 = b - sqrt_D4, t2 = b + sqrt_D4;
    const float min_value = 1e-5f; /* :208 */
    if (t1 < (double)min_value && t2 < (double)min_value) return 0;
    const double t = (t1 > 0.001) ? t1 : t2; /* :212-223, both arms identical but for t */
    *out_t = t;
    if (mode != RTM_MODE_LITERAL) { /* D2: by-value out_normal never reaches the caller */
        const v3 hit = v3_add(org, v3_scale(dir, t));
        *out_normal = normalize3(v3_sub(hit, v3_from(s->center)));
    }
    return 1;
}

int rtmo_intersect(const rtm_sphere* s, const double org[3], const double dir[3], int mode,
                   double* out_t, double out_normal[3]) {
    v3 n = v3_from(out_normal);
    int hit = intersect(s, v3_from(org), v3_from(dir), mode, out_t, &n, NULL);
    if (hit && mode != RTM_MODE_LITERAL) v3_to(n, out_normal);
    return hit;
}

/* D3: in HEAD the recursion is handed ::rand (int rand()) as its double() generator. */
static double libc_rand_generator(void* ctx) {
    rtmo_counters* c = (rtmo_counters*)ctx;
    if (c) c->libc_rand_calls++;
    return (double)rand();
}

typedef struct pt_env {
    const rtm_sphere* spheres; /* geometry of spheres; materials of every object */
    size_t n;
    int mode;
    int max_bounces;
    rtmo_counters* c;
    const rtm_object* objects; /* NULL: all spheres.  Else parallel to `spheres`: type and plane data */
    int integrator;            /* 0: png::PathTracing, 1: png::SurfaeSample (RTM_MODE_SURFACE_SAMPLE) */
} pt_env;

/* png::PlaneObject — constructor src/SettingData.cpp:235-242 restated; Intersect COMPLETED by this build
 * (the reference's falls off its end after its first line, :243-246): include/rtm.h, rtm_object. */
typedef struct plane_data {
    v3 position, normal, right, upv;
    double rr, uu;
} plane_data;
static plane_data plane_make(const rtm_object* o) {
    plane_data p;
    p.position = v3_from(o->position);
    p.normal = normalize3(v3_sub(v3_from(o->target), p.position));                             /* :240 */
    p.right = v3_scale(v3_scale(normalize3(cross3(p.normal, v3_from(o->up))), 0.5), o->width); /* :241 */
    p.upv = cross3(p.right, p.normal); /* build-defined: the in-plane direction across m_right */
    p.rr = dot3(p.right, p.right);
    p.uu = dot3(p.upv, p.upv);
    return p;
}
static int intersect_plane(const rtm_object* o, v3 org, v3 dir, int mode, double* out_t, v3* out_normal) {
    const plane_data p = plane_make(o);
    const double dn = dot3(p.normal, dir);
    if (fabs(dn) < (double)FLT_EPSILON) return 0; /* :244, the reference's only line */
    const double t = dot3(p.normal, v3_sub(p.position, org)) / dn;
    if (!(t > 0.001)) return 0; /* the near threshold of SphereObject::Intersect's t1 (:212) */
    const v3 d = v3_sub(v3_add(org, v3_scale(dir, t)), p.position);
    if (fabs(dot3(d, p.right)) > p.rr || fabs(dot3(d, p.upv)) > p.uu) return 0;
    *out_t = t;
    if (mode != RTM_MODE_LITERAL) *out_normal = p.normal; /* D2 applies to every Object::Intersect */
    return 1;
}
int rtmo_intersect_object(const rtm_object* o, const double org[3], const double dir[3], int mode, double* out_t,
                          double out_normal[3]) {
    if (o->type == RTM_OBJECT_PLANE) {
        v3 n = v3_from(out_normal);
        const int hit = intersect_plane(o, v3_from(org), v3_from(dir), mode, out_t, &n);
        if (hit && mode != RTM_MODE_LITERAL) v3_to(n, out_normal);
        return hit;
    }
    rtm_sphere s;
    memset(&s, 0, sizeof s);
    memcpy(s.center, o->position, sizeof s.center);
    s.radius = o->size;
    return rtmo_intersect(&s, org, dir, mode, out_t, out_normal);
}

static inline double draw(rtmo_rng_fn rng, void* ctx, rtmo_counters* c, int is_libc) {
    if (c && !is_libc) c->draws++;
    return rng(ctx);
}

/* debugging aid: (org, dir) of every cast of the path being traced, by depth (not thread-safe) */
static double* g_trace = NULL;
static int g_trace_cap = 0;
void rtmo_set_trace(double* buf, int capacity) {
    g_trace = buf;
    g_trace_cap = capacity;
}

/* src/Renderer.cpp:57-117 */
static v3 path_trace(const pt_env* env, v3 org, v3 dir, rtmo_rng_fn rng, void* rng_ctx,
                     int rng_is_libc, int depth) {
    if (g_trace && depth < g_trace_cap) {
        double* t = g_trace + (size_t)depth * 6;
        t[0] = org.x; t[1] = org.y; t[2] = org.z; t[3] = dir.x; t[4] = dir.y; t[5] = dir.z;
    }
    int hit_object = -1;
    v3 normal = v3_make(0, 0, 0);
    double dis = DBL_MAX;
    rtmo_counters* c = env->c;
    if (c) {
        c->casts++;
        if ((uint64_t)depth > c->max_depth) c->max_depth = (uint64_t)depth;
    }
    for (size_t i = 0; i < env->n; ++i) { /* :62-72 */
        double tmp_dis = 0.0;
        v3 tmp_normal = v3_make(0, 0, 0);
        int tmp_hit;
        if (env->objects && env->objects[i].type == RTM_OBJECT_PLANE)
            tmp_hit = intersect_plane(&env->objects[i], org, dir, env->mode, &tmp_dis, &tmp_normal);
        else
            tmp_hit = intersect(&env->spheres[i], org, dir, env->mode, &tmp_dis, &tmp_normal, c);
        if (tmp_hit && tmp_dis < dis && tmp_dis > 0) { /* :67 strict <, lowest index wins ties */
            dis = tmp_dis;
            normal = tmp_normal;
            hit_object = (int)i;
        }
    }
    if (hit_object != -1) {
        const rtm_sphere* obj = &env->spheres[hit_object];
        const v3 emission = v3_from(obj->emission);
        if (dis > 0) {
            /* build extension: cast number max_bounces+1 returns emission, no draw */
            if (env->max_bounces >= 0 && depth >= env->max_bounces) return emission;
            if (draw(rng, rng_ctx, c, rng_is_libc) <= (double)rtmo_kd(obj)) { /* :78 */
                const v3 hit_point = v3_add(v3_scale(dir, dis), org);     /* :79 */
                const v3 orienting =
                    dot3(normal, dir) < 0.0 ? normal : v3_scale(normal, -1.0); /* :82-83 */
                const v3 w = orienting;
                const double r1 = 2 * 3.14159265358979323846 * draw(rng, rng_ctx, c, rng_is_libc);
                const double r2 = draw(rng, rng_ctx, c, rng_is_libc); /* :89 */
                const double r2s = sqrt(r2);
                v3 u;
                if (fabs(w.x) > (double)FLT_MIN) /* :96 */
                    u = normalize3(cross3(v3_make(0, 1, 0), w));
                else
                    u = normalize3(cross3(v3_make(1, 0, 0), w));
                const v3 v = cross3(w, u); /* :102 */
                double sin_r1, cos_r1;
                sincos(r1, &sin_r1, &cos_r1); /* what g++ -O2 makes of :103-104, see rtmo_sin_cos_array */
                const v3 a = v3_scale(v3_scale(u, cos_r1), r2s);
                const v3 bq = v3_scale(v3_scale(v, sin_r1), r2s);
                const v3 cq = v3_scale(w, sqrt(1.0 - r2));
                const v3 next_dir = normalize3(v3_add(v3_add(a, bq), cq)); /* :103-107 */
                if (c) c->bounces++;
                v3 next;
                if (env->mode == RTM_MODE_LITERAL) /* :108 D3 */
                    next = path_trace(env, hit_point, next_dir, libc_rand_generator, c, 1,
                                      depth + 1);
                else
                    next = path_trace(env, hit_point, next_dir, rng, rng_ctx, rng_is_libc,
                                      depth + 1);
                double ckd[3];
                rtmo_color_kd(obj, ckd);
                return v3_add(v3_mul(v3_from(ckd), next), emission); /* :109 */
            } else {
                return emission; /* :112 */
            }
        }
    }
    return v3_make(0, 0, 0); /* :116 */
}

/* SphereObject::ComputeSurfacePoint src/SettingData.cpp:227-233 (a CONSTANT point: theta and phi are literals and the
 * generator is not used) and PlaneObject::ComputeSurfacePoint :247-249 (the origin). */
static v3 compute_surface_point(const pt_env* env, size_t i) {
    if (env->objects && env->objects[i].type == RTM_OBJECT_PLANE) return v3_make(0, 0, 0); /* :248 */
    const double pi = 3.14159265358979323846; /* std::numbers::pi */
    const double theta = 2.0 * pi, phi = 0.5 * pi;                                        /* :229-230 */
    const v3 local = v3_make(sin(theta) * sin(phi), sin(theta) * cos(phi), cos(theta));   /* :231 */
    return v3_add(v3_scale(local, (double)env->spheres[i].radius), v3_from(env->spheres[i].center)); /* :232 */
}

/* the nearest-hit loop the reference writes out three times (src/Renderer.cpp:62-72, :126-137, :163-174) */
static int nearest_object(const pt_env* env, v3 org, v3 dir, double* out_dis, v3* out_normal) {
    int hit_object = -1;
    double dis = DBL_MAX;
    v3 normal = v3_make(0, 0, 0);
    if (env->c) env->c->casts++;
    for (size_t i = 0; i < env->n; ++i) {
        double tmp_dis = 0.0;
        v3 tmp_normal = v3_make(0, 0, 0);
        int tmp_hit;
        if (env->objects && env->objects[i].type == RTM_OBJECT_PLANE)
            tmp_hit = intersect_plane(&env->objects[i], org, dir, env->mode, &tmp_dis, &tmp_normal);
        else
            tmp_hit = intersect(&env->spheres[i], org, dir, env->mode, &tmp_dis, &tmp_normal, env->c);
        if (tmp_hit && tmp_dis < dis && tmp_dis > 0) {
            dis = tmp_dis;
            normal = tmp_normal;
            hit_object = (int)i;
        }
    }
    *out_dis = dis;
    *out_normal = normal;
    return hit_object;
}

/* png::SurfaeSample, src/Renderer.cpp:119-198 — the reference's second, experimental integrator (selected at :234 only
 * if a U[0,1) draw is >= 1.0, i.e. never; it has external linkage and takes the same injectable generator).  Literal
 * restatement, recursive like the reference.  max_bounces >= 0 is the same build extension as PathTracing's: an
 * invocation at depth > max_bounces returns (0,0,0) without drawing (the reference has no bound; the recursion ends
 * by its visibility, orientation and roulette tests).  Counters: casts = nearest-hit loops, bounces = invocations at
 * depth > 0 that recursed, draws = generator calls. */
static v3 surface_sample(const pt_env* env, v3 org, v3 dir, int depth, rtmo_rng_fn rng, void* rng_ctx) {
    rtmo_counters* c = env->c;
    if (c && (uint64_t)depth > c->max_depth) c->max_depth = (uint64_t)depth;
    if (depth <= 0) { /* :120-146 */
        double dis;
        v3 normal;
        const int hit = nearest_object(env, org, dir, &dis, &normal);
        if (hit == -1) return v3_make(0, 0, 0); /* :138-140 */
        const v3 hitpoint = v3_add(org, v3_scale(dir, dis)); /* :133 */
        const rtm_sphere* obj = &env->spheres[hit];
        const v3 cal = surface_sample(env, hitpoint, normal, depth + 1, rng, rng_ctx); /* :146 */
        return v3_add(v3_mul(cal, v3_from(obj->color)), v3_from(obj->emission));         /* :147 */
    }
    if (env->max_bounces >= 0 && depth > env->max_bounces) return v3_make(0, 0, 0); /* build extension */
    const int object_index = (int)(draw(rng, rng_ctx, c, 0) * (double)env->n);  /* :150 */
    const rtm_sphere* object = &env->spheres[object_index];
    const v3 surface_point = compute_surface_point(env, (size_t)object_index);      /* :152 */
    const v3 check_dir = normalize3(v3_sub(surface_point, org));                    /* :160 */
    double dis;
    v3 normal;
    const int hit = nearest_object(env, org, check_dir, &dis, &normal);             /* :163-174 */
    if (hit == -1 || hit != object_index) return v3_make(0, 0, 0);                  /* :175-177 */
    const v3 hitpoint = v3_add(org, v3_scale(check_dir, dis));                      /* :171 */
    const v3 d = normalize3(v3_sub(surface_point, org));                            /* :179 */
    const double dot1 = dot3(dir, d);                                               /* :180 */
    const double dot2 = dot3(v3_neg(d), normal);                                    /* :181 */
    if (dot1 <= 0 || dot2 <= 0) return v3_make(0, 0, 0);                            /* :182-184 */
    const double distance = magnitude3(v3_sub(org, hitpoint));                      /* :185 */
    const double probability = (distance < 1.0) ? 1.0 : distance;                   /* :186 std::max(a, b) = (a < b) ? b : a */
    const double div = (1.0 < distance) ? 1.0 : distance;                           /* :187 std::min(a, b) = (b < a) ? b : a */
    const v3 emission = v3_from(object->emission);
    if (dot1 * dot2 * (double)rtmo_kd(object) * probability < draw(rng, rng_ctx, c, 0)) return emission; /* :188-190 */
    if (c) c->bounces++;
    const v3 next = surface_sample(env, hitpoint, normal, depth + 1, rng, rng_ctx); /* :191 */
    double ckd[3];
    rtmo_color_kd(object, ckd);
    return v3_add(v3_scale(v3_mul(next, v3_from(ckd)), div), emission);             /* :192 */
}

void rtmo_surface_sample(const rtm_sphere* spheres, size_t n, int mode, int max_bounces, const double org[3],
                         const double dir[3], rtmo_rng_fn rng, void* rng_ctx, double out_radiance[3],
                         rtmo_counters* counters) {
    pt_env env = {spheres, n, mode, max_bounces, counters, NULL, 0};
    v3_to(surface_sample(&env, v3_from(org), v3_from(dir), 0, rng, rng_ctx), out_radiance);
}

void rtmo_path_trace(const rtm_sphere* spheres, size_t n, int mode, int max_bounces,
                     const double org[3], const double dir[3], rtmo_rng_fn rng, void* rng_ctx,
                     double out_radiance[3], rtmo_counters* counters) {
    pt_env env = {spheres, n, mode, max_bounces, counters, NULL, 0};
    v3_to(path_trace(&env, v3_from(org), v3_from(dir), rng, rng_ctx, 0, 0), out_radiance);
}

/* src/Renderer.cpp:202-208 */
static void camera_basis(const rtm_settings* st, v3* cx, v3* cy, v3* cz, double* fovx,
                         double* fovy) {
    const v3 direction =
        normalize3(v3_sub(v3_from(st->camera.target), v3_from(st->camera.origin)));
    *cx = v3_neg(normalize3(cross3(direction, v3_from(st->camera.up))));
    *cy = cross3(*cx, direction);
    *cz = direction;
    *fovx = (double)st->camera.fov;
    *fovy = *fovx * st->height / st->width;
}

void rtmo_camera_basis(const rtm_settings* st, double cam_x[3], double cam_y[3], double cam_z[3],
                       double* fovx, double* fovy) {
    v3 cx, cy, cz;
    camera_basis(st, &cx, &cy, &cz, fovx, fovy);
    v3_to(cx, cam_x);
    v3_to(cy, cam_y);
    v3_to(cz, cam_z);
}

/* src/Renderer.cpp:227-232 */
static v3 primary_dir(const rtm_settings* st, v3 cx, v3 cy, v3 cz, double fovx, double fovy,
                      int x, int y, int sx, int sy) {
    const float rate = (float)(1.0 / (1 + st->super_samples)); /* :227 */
    const double px = 2.0f * ((double)x + (double)(rate * (float)sx)) / st->width - 1.0f;
    const double py = 2.0f * ((double)y + (double)(rate * (float)sy)) / st->height - 1.0f;
    const v3 a = v3_scale(v3_scale(cx, fovx), px);
    const v3 b = v3_scale(v3_scale(cy, fovy), py);
    return normalize3(v3_add(v3_add(a, b), cz));
}

void rtmo_primary_dir(const rtm_settings* st, int x, int y, int sx, int sy, double out_dir[3]) {
    v3 cx, cy, cz;
    double fovx, fovy;
    camera_basis(st, &cx, &cy, &cz, &fovx, &fovy);
    v3_to(primary_dir(st, cx, cy, cz, fovx, fovy, x, y, sx, sy), out_dir);
}

/* ---- build-defined RNG ---- */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
static inline uint64_t smfin64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* pixel key: one 64-bit mix per pixel; sample key: two 32-bit mixes; draw: one 32-bit mix */
typedef struct rng_stream {
    uint32_t k0, k1, index;
    rtmo_counters* c;
} rng_stream;

static inline uint64_t seed_multiplier(uint64_t seed) {
    return smfin64(seed + 0x9E3779B97F4A7C15ull) | 1ull;
}
static inline void stream_init(rng_stream* s, uint64_t seed_mult, uint32_t pixel, uint32_t sample) {
    const uint64_t z = smfin64(((uint64_t)pixel + 1ull) * seed_mult);
    const uint32_t p0 = (uint32_t)z, p1 = (uint32_t)(z >> 32);
    s->k0 = mix32(p0 + sample * 0x9E3779B9u);
    s->k1 = mix32(p1 ^ (sample * 0x85EBCA6Bu));
    s->index = 0;
}
static inline double stream_u01(uint32_t k0, uint32_t k1, uint32_t index) {
    const uint32_t x = mix32((k0 + index * 0x9E3779B9u) ^ k1);
    /* 23 random bits, never 0 or 1, exact in fp32 and fp64 */
    return (double)(2u * (x >> 9) + 1u) * (1.0 / 16777216.0);
}
static double stream_next(void* ctx) {
    rng_stream* s = (rng_stream*)ctx;
    return stream_u01(s->k0, s->k1, s->index++);
}
double rtmo_rng_u01(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t index) {
    rng_stream s;
    stream_init(&s, seed_multiplier(seed), pixel, sample);
    return stream_u01(s.k0, s.k1, index);
}

/* src/Renderer.cpp:43-49 — std::max<double>(v, 0) then std::min<double>(.., (double)1.0f) */
static inline double clamp01(double v) {
    const double lo = (v < 0.0) ? 0.0 : v;
    return (1.0 < lo) ? 1.0 : lo;
}

static v3 sample_radiance(const rtm_settings* st, const pt_env* env, v3 cx, v3 cy, v3 cz,
                          double fovx, double fovy, uint64_t seed_mult, int x, int y, int sx,
                          int sy, int s) {
    const int SS = st->super_samples, S = st->samples;
    const v3 dir = primary_dir(st, cx, cy, cz, fovx, fovy, x, y, sx, sy);
    rng_stream rs;
    const uint32_t pixel = (uint32_t)y * (uint32_t)st->width + (uint32_t)x;
    const uint32_t sample = (uint32_t)(((sx - 1) * SS + (sy - 1)) * S + s);
    stream_init(&rs, seed_mult, pixel, sample);
    /* src/Renderer.cpp:234's selector draw (always false for U[0,1)) is omitted: Appendix D */
    if (env->c) env->c->samples++;
    if (env->integrator) /* RTM_MODE_SURFACE_SAMPLE: the branch src/Renderer.cpp:234-236 would take */
        return surface_sample(env, v3_from(st->camera.origin), dir, 0, stream_next, &rs);
    return path_trace(env, v3_from(st->camera.origin), dir, stream_next, &rs, 0, 0);
}

void rtmo_sample_radiance(const rtm_settings* st, const rtm_sphere* spheres, size_t n,
                          const rtm_options* opt, int x, int y, int sx, int sy, int s,
                          double out_radiance[3], rtmo_counters* counters) {
    v3 cx, cy, cz;
    double fovx, fovy;
    camera_basis(st, &cx, &cy, &cz, &fovx, &fovy);
    pt_env env = {spheres, n, opt->mode & 0xFF, opt->max_bounces, counters, NULL, (opt->mode & RTM_MODE_SURFACE_SAMPLE) != 0};
    v3_to(sample_radiance(st, &env, cx, cy, cz, fovx, fovy, seed_multiplier(opt->seed), x, y, sx,
                          sy, s),
          out_radiance);
}

/* png::SurfaeSample on an arbitrary ray, drawing from the build RNG stream (seed, pixel, sample) */
void rtmo_surface_sample_stream(const rtm_sphere* spheres, size_t n, int mode, int max_bounces, const double org[3],
                                const double dir[3], uint64_t seed, uint32_t pixel, uint32_t sample,
                                double out_radiance[3], rtmo_counters* counters) {
    pt_env env = {spheres, n, mode, max_bounces, counters, NULL, 1};
    rng_stream rs;
    stream_init(&rs, seed_multiplier(seed), pixel, sample);
    v3_to(surface_sample(&env, v3_from(org), v3_from(dir), 0, stream_next, &rs), out_radiance);
}

/* png::PathTracing on an arbitrary ray, drawing from the build RNG stream (seed, pixel, sample) */
void rtmo_path_trace_stream(const rtm_sphere* spheres, size_t n, int mode, int max_bounces,
                            const double org[3], const double dir[3], uint64_t seed,
                            uint32_t pixel, uint32_t sample, double out_radiance[3],
                            rtmo_counters* counters) {
    pt_env env = {spheres, n, mode, max_bounces, counters, NULL, 0};
    rng_stream rs;
    stream_init(&rs, seed_multiplier(seed), pixel, sample);
    v3_to(path_trace(&env, v3_from(org), v3_from(dir), stream_next, &rs, 0, 0), out_radiance);
}

/* src/Renderer.cpp:222-248 for one pixel */
static void render_pixel(const rtm_settings* st, const pt_env* env, v3 cx, v3 cy, v3 cz,
                         double fovx, double fovy, uint64_t seed_mult, int x, int y,
                         double* out_px) {
    const int SS = st->super_samples, S = st->samples;
    v3 acc = v3_make(0, 0, 0);
    for (int sx = 1; sx <= SS; ++sx)
        for (int sy = 1; sy <= SS; ++sy)
            for (int s = 0; s < S; ++s) {
                v3 cal =
                    sample_radiance(st, env, cx, cy, cz, fovx, fovy, seed_mult, x, y, sx, sy, s);
                cal = v3_div(v3_div(v3_div(cal, (double)SS), (double)SS), (double)S); /* :240 */
                cal = v3_make(clamp01(cal.x), clamp01(cal.y), clamp01(cal.z));      /* :241 */
                acc = v3_add(acc, cal);                                            /* :242 */
            }
    out_px[0] = 0.0 + acc.x; /* :246-248, image[] starts at 0 */
    out_px[1] = 0.0 + acc.y;
    out_px[2] = 0.0 + acc.z;
}

static void counters_add(rtmo_counters* dst, const rtmo_counters* src) {
    dst->samples += src->samples;
    dst->casts += src->casts;
    dst->bounces += src->bounces;
    dst->draws += src->draws;
    dst->sphere_tests += src->sphere_tests;
    dst->sphere_tests_d4 += src->sphere_tests_d4;
    dst->libc_rand_calls += src->libc_rand_calls;
    if (src->max_depth > dst->max_depth) dst->max_depth = src->max_depth;
}

int rtmo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int render_core(const rtm_settings* st, const rtm_sphere* spheres, const rtm_object* objects, size_t n,
                       const rtm_options* opt, double* out, rtmo_counters* counters, int threads,
                       int structure);
int rtmo_render(const rtm_settings* st, const rtm_sphere* spheres, size_t n,
                const rtm_options* opt, double* out, rtmo_counters* counters, int threads,
                int structure) {
    return render_core(st, spheres, NULL, n, opt, out, counters, threads, structure);
}
/* The same for objects of any type: materials and sphere geometry through a parallel rtm_sphere view. */
int rtmo_render_objects(const rtm_settings* st, const rtm_object* objects, size_t n, const rtm_options* opt,
                        double* out, rtmo_counters* counters, int threads) {
    if (!objects && n) return RTM_ERR_INVALID_ARGUMENT;
    rtm_sphere* view = (rtm_sphere*)calloc(n ? n : 1, sizeof(rtm_sphere));
    if (!view) return RTM_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; ++i) {
        memcpy(view[i].center, objects[i].position, sizeof view[i].center);
        memcpy(view[i].color, objects[i].color, sizeof view[i].color);
        memcpy(view[i].emission, objects[i].emission, sizeof view[i].emission);
        view[i].radius = objects[i].size;
    }
    const int rc = render_core(st, view, objects, n, opt, out, counters, threads, 0);
    free(view);
    return rc;
}
static int render_core(const rtm_settings* st, const rtm_sphere* spheres, const rtm_object* objects, size_t n,
                       const rtm_options* opt, double* out, rtmo_counters* counters, int threads,
                       int structure) {
    if (!st || !opt || !out || (!spheres && n)) return RTM_ERR_INVALID_ARGUMENT;
    if (st->width <= 0 || st->height <= 0 || st->samples <= 0 || st->super_samples <= 0)
        return RTM_ERR_INVALID_ARGUMENT;
    if (opt->row_begin < 0 || opt->row_end > st->height || opt->row_begin > opt->row_end)
        return RTM_ERR_INVALID_ARGUMENT;
    v3 cx, cy, cz;
    double fovx, fovy;
    camera_basis(st, &cx, &cy, &cz, &fovx, &fovy);
    const uint64_t seed_mult = seed_multiplier(opt->seed);
    const int W = st->width, r0 = opt->row_begin, rows = opt->row_end - opt->row_begin;
    rtmo_counters total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    if (structure == 1) {
        /* reference structure: serial rows, a new parallel region over x per row (:215-222) */
        for (int yy = 0; yy < rows; ++yy) {
#pragma omp parallel num_threads(threads)
            {
                rtmo_counters local;
                memset(&local, 0, sizeof local);
                pt_env env = {spheres, n, opt->mode & 0xFF, opt->max_bounces, counters ? &local : NULL, objects, (opt->mode & RTM_MODE_SURFACE_SAMPLE) != 0};
#pragma omp for
                for (int x = 0; x < W; ++x)
                    render_pixel(st, &env, cx, cy, cz, fovx, fovy, seed_mult, x, r0 + yy,
                                 out + ((size_t)yy * W + x) * 3);
#pragma omp critical
                counters_add(&total, &local);
            }
        }
    } else {
#pragma omp parallel num_threads(threads)
        {
            rtmo_counters local;
            memset(&local, 0, sizeof local);
            pt_env env = {spheres, n, opt->mode & 0xFF, opt->max_bounces, counters ? &local : NULL, objects, (opt->mode & RTM_MODE_SURFACE_SAMPLE) != 0};
#pragma omp for schedule(dynamic, 16)
            for (long p = 0; p < (long)rows * W; ++p) {
                const int yy = (int)(p / W), x = (int)(p % W);
                render_pixel(st, &env, cx, cy, cz, fovx, fovy, seed_mult, x, r0 + yy,
                             out + (size_t)p * 3);
            }
#pragma omp critical
            counters_add(&total, &local);
        }
    }
    if (counters) *counters = total;
    return RTM_OK;
}

/* The same per-pixel loop (src/Renderer.cpp:222-248) for a LIST of pixels (x, y pairs): spot checks of
 * frames too costly to render whole on the CPU.  out: n_pixels * 3 doubles. */
int rtmo_render_pixels(const rtm_settings* st, const rtm_sphere* spheres, size_t n, const rtm_options* opt,
                       const int32_t* xy, size_t n_pixels, double* out, rtmo_counters* counters, int threads) {
    if (!st || !opt || !out || !xy || (!spheres && n)) return RTM_ERR_INVALID_ARGUMENT;
    for (size_t p = 0; p < n_pixels; ++p)
        if (xy[2 * p] < 0 || xy[2 * p] >= st->width || xy[2 * p + 1] < 0 || xy[2 * p + 1] >= st->height)
            return RTM_ERR_INVALID_ARGUMENT;
    v3 cx, cy, cz;
    double fovx, fovy;
    camera_basis(st, &cx, &cy, &cz, &fovx, &fovy);
    const uint64_t seed_mult = seed_multiplier(opt->seed);
    rtmo_counters total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads)
    {
        rtmo_counters local;
        memset(&local, 0, sizeof local);
        pt_env env = {spheres, n, opt->mode & 0xFF, opt->max_bounces, counters ? &local : NULL, NULL, (opt->mode & RTM_MODE_SURFACE_SAMPLE) != 0};
#pragma omp for schedule(dynamic, 1)
        for (long p = 0; p < (long)n_pixels; ++p)
            render_pixel(st, &env, cx, cy, cz, fovx, fovy, seed_mult, xy[2 * p], xy[2 * p + 1], out + (size_t)p * 3);
#pragma omp critical
        counters_add(&total, &local);
    }
    if (counters) *counters = total;
    return RTM_OK;
}

/* src/Renderer.cpp:251-254: (unsigned char)255 * std::min(image[i], 1.0), stored to u8 */
void rtmo_quantise(const double* image, size_t n_values, uint8_t* out) {
    for (size_t i = 0; i < n_values; ++i) {
        const double m = (1.0 < image[i]) ? 1.0 : image[i]; /* std::min(a,b) = (b<a)?b:a */
        const double v = 255 * m;
        out[i] = (v >= 0.0 && v < 256.0) ? (uint8_t)v : 0; /* out of range is UB upstream */
    }
}

uint64_t rtmo_fnv1a64_f64(const double* v, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) {
        uint64_t b;
        memcpy(&b, &v[i], 8);
        h = (h ^ b) * 1099511628211ull;
    }
    return h;
}

/* SURVEY.md Appendix D — SplitMix64 stress scene */
static uint64_t splitmix_next(uint64_t* state) {
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void rtmo_make_stress_scene(uint64_t seed, size_t n, rtm_settings* st, rtm_sphere* spheres) {
    uint64_t state = seed;
#define U() ((double)(splitmix_next(&state) >> 11) * (1.0 / 9007199254740992.0))
    for (size_t i = 0; i < n; ++i) {
        rtm_sphere* s = &spheres[i];
        memset(s, 0, sizeof *s);
        s->center[0] = -50.0 + 100.0 * U();
        s->center[1] = -50.0 + 100.0 * U();
        s->center[2] = -50.0 + 100.0 * U();
        s->radius = (float)(0.2 + 0.8 * U());
        s->color[0] = 0.1 + 0.8 * U();
        s->color[1] = 0.1 + 0.8 * U();
        s->color[2] = 0.1 + 0.8 * U();
        const double e = (i % 50 == 0) ? 5.0 : 0.0;
        s->emission[0] = s->emission[1] = s->emission[2] = e;
    }
#undef U
    memset(st, 0, sizeof *st);
    st->width = 1920;
    st->height = 1080;
    st->samples = 256;
    st->super_samples = 1;
    st->camera.origin[2] = -60.0;
    st->camera.up[1] = 1.0;
    st->camera.fov = 1.0f;
}
