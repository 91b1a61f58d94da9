//! Raw declarations of `include/tinyrt.h` (ABI version 3).  Field order, scalar types and function parameter lists are
//! checked against the header by tests/test_rust_bindings.py; the crate itself has never been compiled (no Rust
//! toolchain in the build image).
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const TRT_ABI_VERSION: u32 = 4;

// enum trt_status
pub const TRT_OK: c_int = 0;
pub const TRT_ERR_INVALID_ARG: c_int = -1;
pub const TRT_ERR_DUPLICATE: c_int = -2;
pub const TRT_ERR_NOT_FOUND: c_int = -3;
pub const TRT_ERR_HIP: c_int = -4;
pub const TRT_ERR_NO_DEVICE: c_int = -5;
pub const TRT_ERR_OOM: c_int = -6;

// enum trt_material_kind
pub const TRT_LAMBERTIAN: u32 = 0;
pub const TRT_METAL: u32 = 1;
pub const TRT_DIELECTRIC: u32 = 2;
pub const TRT_LIGHT: u32 = 3;

// enum trt_backend
pub const TRT_BACKEND_MEGAKERNEL: u32 = 0;
pub const TRT_BACKEND_WAVEFRONT: u32 = 1;
pub const TRT_BACKEND_AUTO: u32 = 2;
pub const TRT_BACKEND_STREAMED: u32 = 3;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_vec3 {
    pub x: f32,
    pub y: f32,
    pub z: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_ray {
    pub origin: trt_vec3,
    pub direction: trt_vec3,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_sample_point {
    pub x: u32,
    pub y: u32,
    pub ray: trt_ray,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_sampled_color {
    pub x: u32,
    pub y: u32,
    pub color: trt_vec3,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_material {
    pub kind: u32,
    pub albedo: trt_vec3,
    pub param: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_scene_info {
    pub num_nodes: u32,
    pub num_spheres: u32,
    pub num_quads: u32,
    pub num_materials: u32,
    pub max_depth: u32,
    pub device_bytes: u32,
    pub lds_bytes: u32,
    pub num_cull_nodes: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_camera {
    pub position: trt_vec3,
    pub viewport_upper_left: trt_vec3,
    pub forward: trt_vec3,
    pub horizontal: trt_vec3,
    pub vertical: trt_vec3,
    pub defocus_disk_u: trt_vec3,
    pub defocus_disk_v: trt_vec3,
    pub width: u32,
    pub height: u32,
}

/// Placement of a compiled scene (every value renders the same frames); fill with `trt_scene_options_default`.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_scene_options {
    pub cull_prune: f32,
    pub flat_walk: i32,
    pub compact_nodes: i32,
    pub top_nodes: u32,
    pub scratch_cap_bytes: u64,
    pub reserved: [u32; 6],
}

/// Scheduling of a render (every value renders the same frame); fill with `trt_tuning_default`.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_tuning {
    pub stream_waves_per_simd: u32,
    pub stream_big_threads: u32,
    pub stream_batch_spp: u32,
    pub radiance_gb: u32,
    pub leaf_slots: u32,
    pub lds_leaf_stack: u32,
    pub ray_pool: u32,
    pub stragglers: u32,
    pub lds_stragglers: u32,
    pub dual_walk: u32,
    pub runtime_walk: u32,
    pub xcd_remap: u32,
    pub mega_waves_per_simd: u32,
    pub mega_threads: u32,
    pub mega_global_waves8: u32,
    pub wf_waves_per_simd: u32,
    pub wf_serve_min: u32,
    pub reserved: [u32; 7],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct trt_render_params {
    pub samples_per_pixel: u32,
    pub max_bounces: u32,
    pub background: trt_vec3,
    pub seed: u32,
    pub backend: u32,
    pub sample_begin: u32,
    pub sample_end: u32,
    pub accumulate: u32,
    pub band_rows: u32,
    pub band_stride: u32,
    pub band_offset: u32,
    pub rows_local: u32,
    pub collect_stats: u32,
    pub tuning: *const trt_tuning,
}
impl Default for trt_render_params {
    fn default() -> Self {
        // all-zero: whole image, all samples, the library's default tuning (null)
        unsafe { std::mem::zeroed() }
    }
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_band_copy {
    pub rows_local: u32,
    pub full_bands: u32,
    pub tail_rows: u32,
    pub reserved: u32,
    pub band_bytes: u64,
    pub local_pitch: u64,
    pub frame_pitch: u64,
    pub frame_offset: u64,
    pub tail_bytes: u64,
    pub tail_local_offset: u64,
    pub tail_frame_offset: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_launch_plan {
    pub scene_mode: u32,
    pub threads_per_workgroup: u32,
    pub waves_per_simd: u32,
    pub workgroups_per_cu: u32,
    pub lds_bytes: u32,
    pub scene_lds_bytes: u32,
    pub leaf_slots: u32,
    pub lds_leaf_stack: u32,
    pub ray_pool: u32,
    pub walk: u32,
    pub specialised: u32,
    pub has_kernel: u32,
    pub kernel_waves_per_simd: u32,
    pub kernel_threads: u32,
    pub kernel_walk: u32,
    pub kernel_ray_pool: u32,
    pub kernel_counting: u32,
    pub chunk_spp: u32,
    pub dual_walk: u32,
    pub workspace_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct trt_stats {
    pub samples: u64,
    pub rays: u64,
    pub node_tests: u64,
    pub sphere_tests: u64,
    pub quad_plane_tests: u64,
    pub quad_inside_tests: u64,
    pub shades: u64,
    pub kernel_ms: f64,
    pub wave_trips: [u64; 4],
    pub gather_per_band: u64,
}

/// Opaque handles (owned by the library).
#[repr(C)]
pub struct trt_world {
    _private: [u8; 0],
}
#[repr(C)]
pub struct trt_scene {
    _private: [u8; 0],
}

extern "C" {
    pub fn trt_world_create(out: *mut *mut trt_world) -> c_int;
    pub fn trt_world_destroy(w: *mut trt_world);
    pub fn trt_world_add_material(w: *mut trt_world, name: *const c_char, m: *const trt_material) -> c_int;
    pub fn trt_world_get_material(w: *const trt_world, name: *const c_char, index: *mut u32) -> c_int;
    pub fn trt_world_add_sphere(w: *mut trt_world, center: trt_vec3, radius: f32, material: u32) -> c_int;
    pub fn trt_world_add_spheres(w: *mut trt_world, n: u32, center_radius: *const f32, material: *const u32) -> c_int;
    pub fn trt_world_add_quad(w: *mut trt_world, corner: trt_vec3, u: trt_vec3, v: trt_vec3, material: u32) -> c_int;
    pub fn trt_world_num_geometries(w: *const trt_world) -> c_int;
    pub fn trt_world_num_materials(w: *const trt_world) -> c_int;

    pub fn trt_scene_create(w: *const trt_world, out: *mut *mut trt_scene) -> c_int;
    pub fn trt_scene_options_default(out: *mut trt_scene_options);
    pub fn trt_scene_create_ex(w: *const trt_world, options: *const trt_scene_options, out: *mut *mut trt_scene) -> c_int;
    pub fn trt_tuning_default(out: *mut trt_tuning);
    pub fn trt_scene_destroy(s: *mut trt_scene);
    pub fn trt_scene_trim(s: *mut trt_scene) -> c_int;
    pub fn trt_scene_get_info(s: *const trt_scene, out: *mut trt_scene_info) -> c_int;
    pub fn trt_scene_get_nodes(s: *const trt_scene, bbox6: *mut f32, prim: *mut i32, skip: *mut i32, cap: u32) -> c_int;
    pub fn trt_scene_get_cull_nodes(s: *const trt_scene, bbox6: *mut f32, prim: *mut i32, skip: *mut i32, cap: u32) -> c_int;
    pub fn trt_scene_get_compact_nodes(s: *const trt_scene, words4: *mut u32, cap: u32) -> c_int;

    pub fn trt_camera_init(out: *mut trt_camera, focus_distance: f32, defocus_angle_deg: f32, position: trt_vec3,
                           look_at: trt_vec3, up: trt_vec3, vertical_fov_deg: f32, width: u32, height: u32) -> c_int;

    pub fn trt_render(s: *mut trt_scene, cam: *const trt_camera, p: *const trt_render_params, accum: *mut f32,
                      stats: *mut trt_stats) -> c_int;
    pub fn trt_render_multi(s: *mut trt_scene, cam: *const trt_camera, p: *const trt_render_params, devices: *const c_int,
                            ndev: u32, accum: *mut f32, stats: *mut trt_stats) -> c_int;
    pub fn trt_render_multi_device(s: *mut trt_scene, cam: *const trt_camera, p: *const trt_render_params,
                                   devices: *const c_int, ndev: u32, d_accum: *mut f32, stats: *mut trt_stats) -> c_int;
    pub fn trt_band_rows_local(height: u32, ndev: u32, rank: u32, rows_local: *mut u32) -> c_int;
    pub fn trt_band_copy_plan(width: u32, height: u32, ndev: u32, rank: u32, out: *mut trt_band_copy) -> c_int;
    pub fn trt_streamed_launch_plan(s: *const trt_scene, cam: *const trt_camera, p: *const trt_render_params,
                                    out: *mut trt_launch_plan) -> c_int;
    pub fn trt_render_device(s: *mut trt_scene, cam: *const trt_camera, p: *const trt_render_params, d_accum: *mut f32,
                             d_counters: *mut u64, stream: *mut c_void) -> c_int;
    pub fn trt_sample_batch(s: *mut trt_scene, input: *const trt_sample_point, n: u32, out: *mut trt_sampled_color,
                            max_bounces: u32, background: trt_vec3, seed: u32, stats: *mut trt_stats) -> c_int;
    pub fn trt_tonemap_u8(accum: *const f32, npixels: u32, gamma: f32, rgb: *mut u8) -> c_int;
    pub fn trt_tonemap_u8_device(d_accum: *const f32, npixels: u32, gamma: f32, d_rgb: *mut u8, stream: *mut c_void) -> c_int;
    pub fn trt_streamed_chunk_spp(width: u32, rows: u32) -> u32;
    pub fn trt_kernel_timing_begin() -> c_int;
    pub fn trt_kernel_timing_end(total_ms: *mut f64, launches: *mut u32) -> c_int;

    pub fn trt_dominant_kernel(s: *const trt_scene, cam: *const trt_camera, p: *const trt_render_params) -> *const c_char;

    pub fn trt_last_error() -> *const c_char;
    pub fn trt_device_count() -> c_int;
    pub fn trt_set_device(ordinal: c_int) -> c_int;
    pub fn trt_abi_version() -> u32;
}
