/* multi_gpu.c — one call, every visible GPU: the Cornell box rendered by trt_render_multi_device over ALL devices of the node
 * into a frame in HBM on device 0 (each shard's bands go there with one strided 2-D device-to-device copy over xGMI), checked
 * against the same frame rendered by device 0 alone.  Plain C11 against include/tinyrt.h; the only HIP calls are the
 * allocation of the destination frame and the copy back.
 *
 *   hipcc -x c -std=c11 -I include examples/multi_gpu.c -L tiny-raytracer_amd -ltinyrt -Wl,-rpath,$PWD/tiny-raytracer_amd -o multi_gpu
 *   ./multi_gpu [--verify-each-device] [width height spp]
 *
 * --verify-each-device: before anything is gathered, the whole frame is rendered on EVERY visible ordinal by itself and compared with
 * device 0's - scene replication (the blob uploaded per device), the per-device contexts and every device's kernels are validated one
 * by one, so that on a new multi-GPU box a wrong frame is pinned to a device before the gather (peer access, strided device-to-device
 * copies) comes into play.  The report also says how the gather went (trt_stats.gather_per_band: shards that fell back to band-by-band
 * copies).
 *
 * On a one-GPU box it renders with shards {0, 0} (two shards sharing the device), which exercises the same band layout; the
 * peer path (hipDeviceEnablePeerAccess + device-to-device gather between two ordinals) needs a box with at least two GPUs. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "tinyrt.h"

#define OK(call) do { int rc_ = (call); if (rc_ != TRT_OK) { fprintf(stderr, "%s: %d %s\n", #call, rc_, trt_last_error()); return 1; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

static int quad(trt_world *w, float cx, float cy, float cz, float ux, float uy, float uz, float vx, float vy, float vz, uint32_t m) {
    trt_vec3 c = {cx, cy, cz}, u = {ux, uy, uz}, v = {vx, vy, vz};
    return trt_world_add_quad(w, c, u, v, m);
}

int main(int argc, char **argv) {
    int verify_each = 0;
    if (argc > 1 && strcmp(argv[1], "--verify-each-device") == 0) { verify_each = 1; argc--; argv++; }
    const uint32_t W = argc > 2 ? (uint32_t)atoi(argv[1]) : 512, H = argc > 2 ? (uint32_t)atoi(argv[2]) : 500;   /* 500: ragged last band */
    const uint32_t spp = argc > 3 ? (uint32_t)atoi(argv[3]) : 16;
    int ndev = trt_device_count();
    if (ndev < 1) { fprintf(stderr, "no GPU visible\n"); return 2; }

    /* the room of src/main.rs:29-87 without the two boxes: 6 quads */
    trt_world *world;
    OK(trt_world_create(&world));
    trt_material red = {TRT_LAMBERTIAN, {0.65f, 0.05f, 0.05f}, 0}, white = {TRT_LAMBERTIAN, {0.73f, 0.73f, 0.73f}, 0};
    trt_material green = {TRT_LAMBERTIAN, {0.12f, 0.45f, 0.15f}, 0}, light = {TRT_LIGHT, {15, 15, 15}, 0};
    OK(trt_world_add_material(world, "red", &red)); OK(trt_world_add_material(world, "white", &white));
    OK(trt_world_add_material(world, "green", &green)); OK(trt_world_add_material(world, "light", &light));
    OK(quad(world, 100, 0, 0, 0, 100, 0, 0, 0, 100, 2));
    OK(quad(world, 0, 0, 0, 0, 100, 0, 0, 0, 100, 0));
    OK(quad(world, 35, 99.9f, 40, 30, 0, 0, 0, 0, 20, 3));
    OK(quad(world, 0, 0, 0, 100, 0, 0, 0, 0, 100, 1));
    OK(quad(world, 100, 100, 100, -100, 0, 0, 0, 0, -100, 1));
    OK(quad(world, 0, 0, 100, 100, 0, 0, 0, 100, 0, 1));
    trt_scene *scene;
    OK(trt_scene_create(world, &scene));
    trt_world_destroy(world);
    trt_camera cam;
    trt_vec3 pos = {50, 50, -140}, at = {50, 50, 0}, up = {0, 1, 0};
    OK(trt_camera_init(&cam, 140.0f, 0.6f, pos, at, up, 40.0f, W, H));
    trt_render_params p;
    memset(&p, 0, sizeof p);
    p.samples_per_pixel = spp; p.max_bounces = 20; p.background.x = p.background.y = p.background.z = 0.001f; p.seed = 1;
    p.backend = TRT_BACKEND_AUTO;

    const size_t n = (size_t)W * H * 3;
    float *one = malloc(n * sizeof(float)), *all = malloc(n * sizeof(float));
    trt_stats st1, stn;
    OK(trt_set_device(0));
    OK(trt_render(scene, &cam, &p, one, &st1));                                  /* device 0 alone */
    if (verify_each) {                                                          /* ... and every other device alone, before any gather */
        int bad = 0;
        for (int d = 0; d < ndev; d++) {
            trt_stats sd;
            OK(trt_set_device(d));
            OK(trt_render(scene, &cam, &p, all, &sd));
            const int ok = memcmp(one, all, n * sizeof(float)) == 0 && sd.rays == st1.rays;
            printf("device %d alone: %.1f ms, %llu rays, frame %s\n", d, sd.kernel_ms, (unsigned long long)sd.rays, ok ? "identical to device 0's" : "DIFFERS");
            bad += !ok;
        }
        OK(trt_set_device(0));
        if (bad) { fprintf(stderr, "%d device(s) render another frame than device 0: not gathering\n", bad); return 1; }
    }

    int devices[64];
    uint32_t shards = ndev > 1 ? (uint32_t)ndev : 2u;
    for (uint32_t r = 0; r < shards; r++) devices[r] = ndev > 1 ? (int)r : 0;
    float *d_frame = NULL;
    HIP(hipSetDevice(devices[0]));
    HIP(hipMalloc((void **)&d_frame, n * sizeof(float)));
    OK(trt_render_multi_device(scene, &cam, &p, devices, shards, d_frame, &stn)); /* every device, frame gathered on devices[0] */
    HIP(hipMemcpy(all, d_frame, n * sizeof(float), hipMemcpyDeviceToHost));
    /* a second pass continues the running sums through the same gather path (reads the frame back shard by shard) */
    p.accumulate = 1;
    OK(trt_render_multi_device(scene, &cam, &p, devices, shards, d_frame, NULL));
    HIP(hipFree(d_frame));

    int same = memcmp(one, all, n * sizeof(float)) == 0 && st1.rays == stn.rays;
    printf("%u x %u, %u spp: %d device(s) visible, %u shards, %.1f ms on the slowest shard (one device: %.1f ms), %llu rays, gather: %s, frames %s\n", W, H, spp,
           ndev, shards, stn.kernel_ms, st1.kernel_ms, (unsigned long long)stn.rays,
           stn.gather_per_band ? "band by band on some shards (no peer access, or the strided peer copy was refused)" : "one strided 2-D copy per shard",
           same ? "IDENTICAL" : "DIFFER");
    OK(trt_scene_trim(scene));
    trt_scene_destroy(scene);
    free(one); free(all);
    return same ? 0 : 1;
}
