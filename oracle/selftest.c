/* oracle/selftest.c — TEST INFRASTRUCTURE.  Sanitizer target for the CPU restatement: built with
 * -fsanitize=address,undefined (oracle/Makefile: selftest_asan) and run by tests/test_oracle_asan.py.
 * Renders small frames in both modes (all loop structures, several threads), the stress-scene
 * generator and the per-ray entry points; any out-of-bounds access, leak or UB aborts the run. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cpu_ref.h"

static double frac_rng(void* ctx) {
    long* k = (long*)ctx;
    double v = (double)(++*k) * 0.6180339887498949;
    return v - floor(v);
}

int main(void) {
    /* the Cornell box of the shipped scene file (tests/golden/scenes/cornellBoxSetting.json) */
    const double c[7][3] = {{0, 10, 0}, {10010, 0, 0}, {-10010, 0, 0}, {0, 10010, 0}, {0, -10010, 0}, {0, 0, 10010}, {0, 0, -10010}};
    const double col[7][3] = {{0, 0, 0}, {.9, .25, .25}, {.25, .9, .25}, {.25, .25, .9}, {.7, .7, .7}, {.9, .25, .9}, {.25, .9, .9}};
    rtm_sphere sp[7];
    memset(sp, 0, sizeof sp);
    for (int i = 0; i < 7; ++i) {
        memcpy(sp[i].center, c[i], sizeof c[i]);
        memcpy(sp[i].color, col[i], sizeof col[i]);
        sp[i].radius = i ? 10000.f : 5.f;
        if (!i) sp[i].emission[0] = sp[i].emission[1] = sp[i].emission[2] = 5;
    }
    rtm_settings st;
    memset(&st, 0, sizeof st);
    st.width = 37;
    st.height = 21;
    st.samples = 3;
    st.super_samples = 2;
    st.camera.origin[2] = -10;
    st.camera.up[1] = 1;
    st.camera.fov = 2.f;
    double* img = malloc(sizeof(double) * 37 * 21 * 3);
    double* img2 = malloc(sizeof(double) * 37 * 21 * 3);
    for (int mode = 0; mode < 2; ++mode)
        for (int mb = -1; mb <= 8; mb += 9) {
            rtm_options o;
            memset(&o, 0, sizeof o);
            o.mode = mode;
            o.max_bounces = mb;
            o.seed = 99;
            o.row_end = st.height;
            rtmo_counters cnt;
            if (rtmo_render(&st, sp, 7, &o, img, &cnt, 3, 0) != 0) return 2;
            if (rtmo_render(&st, sp, 7, &o, img2, NULL, 2, 1) != 0) return 3;
            if (memcmp(img, img2, sizeof(double) * 37 * 21 * 3) != 0) return 4;
            o.row_begin = 5;
            o.row_end = 9;
            if (rtmo_render(&st, sp, 7, &o, img2, &cnt, 0, 0) != 0) return 5;
            if (memcmp(img + 5 * 37 * 3, img2, sizeof(double) * 4 * 37 * 3) != 0) return 6;
        }
    /* L1 known answer: dir N(-0.5, 0.1, 1) -> 1.3888889624748728 */
    {
        double d[3] = {-0.5, 0.1, 1}, dn[3], org[3] = {0, 0, -10}, L[3];
        long k = 0;
        rtmo_counters cnt;
        memset(&cnt, 0, sizeof cnt);
        rtmo_normalize(d, dn);
        rtmo_path_trace(sp, 7, RTM_MODE_REPAIRED, -1, org, dn, frac_rng, &k, L, &cnt);
        if (L[0] != 1.3888889624748728 || k != 7) return 7;
    }
    uint8_t* q = malloc(37 * 21 * 3);
    rtmo_quantise(img, 37 * 21 * 3, q);
    (void)rtmo_fnv1a64_f64(img, 37 * 21 * 3);
    rtm_sphere* big = malloc(sizeof(rtm_sphere) * 1000);
    rtmo_make_stress_scene(12345, 1000, &st, big);
    st.width = 16;
    st.height = 8;
    st.samples = 2;
    rtm_options o;
    memset(&o, 0, sizeof o);
    o.mode = 1;
    o.max_bounces = 4;
    o.row_end = 8;
    if (rtmo_render(&st, big, 1000, &o, img, NULL, 2, 0) != 0) return 8;
    free(big);
    free(q);
    free(img);
    free(img2);
    puts("oracle selftest ok");
    return 0;
}
