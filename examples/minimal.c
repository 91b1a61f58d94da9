/* minimal.c - the C ABI (include/tinyrt.h) from plain C: one sphere over a ground sphere, 64x48, 16 spp -> minimal.ppm
 *   gcc -std=c11 -Iinclude examples/minimal.c -Ltiny-raytracer_amd -ltinyrt -Wl,-rpath,$PWD/tiny-raytracer_amd -o build/minimal
 * Mirrors the reference's renderer test scene (renderer/renderer.rs:87-123) in miniature; exits 1 with the library's
 * message when no MI355X is visible (the product has no CPU path). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tinyrt.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != TRT_OK) { fprintf(stderr, "%s: %s\n", #call, trt_last_error()); return 1; } } while (0)

int main(void) {
    trt_world *world = NULL;
    trt_scene *scene = NULL;
    CHECK(trt_world_create(&world));
    const trt_material ground = {TRT_LAMBERTIAN, {0.0f, 1.0f, 0.0f}, 0.0f}, centre = {TRT_LAMBERTIAN, {1.0f, 0.0f, 0.0f}, 0.0f};
    CHECK(trt_world_add_material(world, "ground", &ground));
    CHECK(trt_world_add_material(world, "center", &centre));
    uint32_t mg = 0, mc = 0;
    CHECK(trt_world_get_material(world, "ground", &mg));
    CHECK(trt_world_get_material(world, "center", &mc));
    const trt_vec3 cg = {0.0f, -100.5f, -1.0f}, cc = {0.0f, 0.0f, -1.2f};
    CHECK(trt_world_add_sphere(world, cg, 100.0f, mg));
    CHECK(trt_world_add_sphere(world, cc, 0.5f, mc));
    CHECK(trt_scene_create(world, &scene));                                  /* World::get_bvh */

    trt_camera camera;
    const trt_vec3 pos = {-2.0f, 2.0f, 1.0f}, look_at = {0.0f, 0.0f, -1.0f}, up = {0.0f, 1.0f, 0.0f};
    const uint32_t w = 64, h = 48;
    CHECK(trt_camera_init(&camera, 3.4f, 10.0f, pos, look_at, up, 20.0f, w, h));     /* Camera::new */

    trt_render_params p = {0};                                                /* Renderer::new(16, _, 10, _, Some(bg)) */
    p.samples_per_pixel = 16;
    p.max_bounces = 10;
    p.background.x = 0.7f; p.background.y = 0.8f; p.background.z = 1.0f;
    p.seed = 1;
    p.backend = TRT_BACKEND_AUTO;
    float *accum = calloc((size_t)w * h * 3, sizeof(float));
    unsigned char *rgb = malloc((size_t)w * h * 3);
    trt_stats st;
    if (!accum || !rgb) return 1;
    CHECK(trt_render(scene, &camera, &p, accum, &st));                        /* Renderer::render */
    /* Scheduling is data (ABI 3), like the reference's constructor arguments - never the environment.  Any value renders the SAME frame: */
    trt_tuning tuning;
    trt_tuning_default(&tuning);                                              /* the library's defaults ... */
    tuning.stream_waves_per_simd = 5;                                         /* ... with another wave budget and leaf-stack depth */
    tuning.leaf_slots = 3;
    p.tuning = &tuning;
    float *again = calloc((size_t)w * h * 3, sizeof(float));
    if (!again) return 1;
    CHECK(trt_render(scene, &camera, &p, again, NULL));
    if (memcmp(accum, again, (size_t)w * h * 3 * sizeof(float)) != 0) { fprintf(stderr, "tuning changed the frame\n"); return 1; }
    free(again);
    CHECK(trt_tonemap_u8(accum, w * h, 2.2f, rgb));                           /* Imager finalisation */
    FILE *f = fopen("minimal.ppm", "wb");
    if (!f) return 1;
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    fwrite(rgb, 1, (size_t)w * h * 3, f);
    fclose(f);
    printf("%llu rays in %.2f ms -> minimal.ppm\n", (unsigned long long)st.rays, st.kernel_ms);
    free(accum); free(rgb);
    trt_scene_destroy(scene);
    trt_world_destroy(world);
    return 0;
}
