//! The reference crate's World / Camera / Renderer::render surface over libtinyrt.so (MI355X).
//!
//! Names, argument order and meaning follow the reference (paths relative to raytracer/src):
//!   World::{new, add_material, get_material, add_geometry}     hittable/world.rs:16-41
//!   Sphere::new(center, radius, material), Quad::new(corner, u, v, material)   sphere.rs:16, quad.rs:20
//!   Lambertian / Metal / Dielectric / Light ::new              material/*.rs
//!   Camera::new(focus_distance, defocus_angle, position, look_at, up, vertical_fov, width, height)   camera.rs:17-26
//!   Renderer::new(samples_per_pixel, num_sampler_threads, max_bounces, progressbar, background_color)   renderer.rs:21-35
//!   Renderer::render(&camera, &world)                          renderer.rs:37-79
//! Where the reference panics (duplicate material name, world.rs:29-31) this returns `Err(Error)`.
//! Never compiled (no Rust toolchain in the build image); see ../README.md.

use std::ffi::{CStr, CString};
use std::io::Write;
use std::ptr;

use tinyrt_sys as sys;

pub type Float = f32; // lib.rs:4

#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "tinyrt error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

fn check(rc: i32) -> Result<(), Error> {
    if rc == sys::TRT_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::trt_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct Vec3 {
    pub x: Float,
    pub y: Float,
    pub z: Float,
}

impl Vec3 {
    pub fn new(x: Float, y: Float, z: Float) -> Self {
        Vec3 { x, y, z }
    }
    pub fn new_diagonal(v: Float) -> Self {
        Vec3 { x: v, y: v, z: v } // math/vec3.rs:23-25
    }
    pub fn zero() -> Self {
        Vec3::default() // math/vec3.rs:27-29
    }
    fn raw(self) -> sys::trt_vec3 {
        sys::trt_vec3 { x: self.x, y: self.y, z: self.z }
    }
}

/// Stands in for `Arc<Box<dyn Material>>`: an index into the world's material table.
pub type MaterialHandle = u32;

pub trait Material {
    fn pod(&self) -> sys::trt_material;
}

pub struct Lambertian(Vec3);
pub struct Metal(Vec3, Float);
pub struct Dielectric(Vec3, Float);
pub struct Light(Vec3);

impl Lambertian {
    pub fn new(albedo: Vec3) -> Self {
        Lambertian(albedo)
    }
}
impl Metal {
    pub fn new(albedo: Vec3, fuzz: Float) -> Self {
        Metal(albedo, fuzz)
    }
}
impl Dielectric {
    pub fn new(albedo: Vec3, refraction_index: Float) -> Self {
        Dielectric(albedo, refraction_index)
    }
}
impl Light {
    pub fn new(color: Vec3) -> Self {
        Light(color)
    }
}
impl Material for Lambertian {
    fn pod(&self) -> sys::trt_material {
        sys::trt_material { kind: sys::TRT_LAMBERTIAN, albedo: self.0.raw(), param: 0.0 }
    }
}
impl Material for Metal {
    fn pod(&self) -> sys::trt_material {
        sys::trt_material { kind: sys::TRT_METAL, albedo: self.0.raw(), param: self.1 }
    }
}
impl Material for Dielectric {
    fn pod(&self) -> sys::trt_material {
        sys::trt_material { kind: sys::TRT_DIELECTRIC, albedo: self.0.raw(), param: self.1 }
    }
}
impl Material for Light {
    fn pod(&self) -> sys::trt_material {
        sys::trt_material { kind: sys::TRT_LIGHT, albedo: self.0.raw(), param: 0.0 }
    }
}

pub enum Geometry {
    Sphere { center: Vec3, radius: Float, material: MaterialHandle },
    Quad { corner: Vec3, u: Vec3, v: Vec3, material: MaterialHandle },
}

pub struct Sphere;
impl Sphere {
    pub fn new(center: Vec3, radius: Float, material: MaterialHandle) -> Geometry {
        Geometry::Sphere { center, radius, material }
    }
}
pub struct Quad;
impl Quad {
    pub fn new(corner: Vec3, u: Vec3, v: Vec3, material: MaterialHandle) -> Geometry {
        Geometry::Quad { corner, u, v, material }
    }
}

pub struct World {
    handle: *mut sys::trt_world,
    scene: *mut sys::trt_scene, // World::get_bvh(), cached until the world changes
}

// The library's handles are plain host data behind a mutex-free API; one thread at a time per handle.
unsafe impl Send for World {}

impl World {
    pub fn new() -> Result<Self, Error> {
        let mut handle = ptr::null_mut();
        check(unsafe { sys::trt_world_create(&mut handle) })?;
        Ok(World { handle, scene: ptr::null_mut() })
    }

    pub fn add_material(&mut self, name: &str, material: &dyn Material) -> Result<(), Error> {
        let cname = CString::new(name).map_err(|_| Error { code: sys::TRT_ERR_INVALID_ARG, message: "name contains NUL".into() })?;
        let pod = material.pod();
        check(unsafe { sys::trt_world_add_material(self.handle, cname.as_ptr(), &pod) })
    }

    pub fn get_material(&self, name: &str) -> Option<MaterialHandle> {
        let cname = CString::new(name).ok()?;
        let mut index = 0u32;
        let rc = unsafe { sys::trt_world_get_material(self.handle, cname.as_ptr(), &mut index) };
        if rc == sys::TRT_OK {
            Some(index)
        } else {
            None // world.rs:35-41 returns Option
        }
    }

    pub fn add_geometry(&mut self, geometry: Geometry) -> Result<(), Error> {
        self.invalidate();
        match geometry {
            Geometry::Sphere { center, radius, material } => {
                check(unsafe { sys::trt_world_add_sphere(self.handle, center.raw(), radius, material) })
            }
            Geometry::Quad { corner, u, v, material } => {
                check(unsafe { sys::trt_world_add_quad(self.handle, corner.raw(), u.raw(), v.raw(), material) })
            }
        }
    }

    /// World::get_bvh (world.rs:43-45): the reference-order BVH, packed for the GPU.
    fn get_bvh(&mut self) -> Result<*mut sys::trt_scene, Error> {
        if self.scene.is_null() {
            check(unsafe { sys::trt_scene_create(self.handle, &mut self.scene) })?;
        }
        Ok(self.scene)
    }

    /// Frees the device scratch (render workspaces, frame buffers) the compiled scene caches between renders.
    pub fn trim(&mut self) -> Result<(), Error> {
        if self.scene.is_null() {
            return Ok(());
        }
        check(unsafe { sys::trt_scene_trim(self.scene) })
    }

    fn invalidate(&mut self) {
        if !self.scene.is_null() {
            unsafe { sys::trt_scene_destroy(self.scene) };
            self.scene = ptr::null_mut();
        }
    }
}

impl Drop for World {
    fn drop(&mut self) {
        self.invalidate();
        unsafe { sys::trt_world_destroy(self.handle) };
    }
}

pub struct Camera {
    pod: sys::trt_camera,
}

impl Camera {
    #[allow(clippy::too_many_arguments)]
    pub fn new(focus_distance: Float, defocus_angle: Float, position: Vec3, look_at: Vec3, up: Vec3, vertical_fov: Float,
               width: usize, height: usize) -> Result<Self, Error> {
        let mut pod = sys::trt_camera::default();
        check(unsafe {
            sys::trt_camera_init(&mut pod, focus_distance, defocus_angle, position.raw(), look_at.raw(), up.raw(), vertical_fov,
                                 width as u32, height as u32)
        })?;
        Ok(Camera { pod })
    }

    pub fn get_image_size(&self) -> (usize, usize) {
        (self.pod.width as usize, self.pod.height as usize) // camera.rs:68-70
    }
}

/// utils/image.rs Image with gamma 2.2 as the Imager builds it (imager.rs:37-41); holds the linear sums.
pub struct Image {
    width: usize,
    height: usize,
    gamma: Float,
    data: Vec<Float>,
}

impl Image {
    pub fn size(&self) -> (usize, usize) {
        (self.width, self.height)
    }
    /// Linear (not gamma-corrected) pixel, image.rs:46-48.
    pub fn get_pixel(&self, x: usize, y: usize) -> Vec3 {
        let i = (y * self.width + x) * 3;
        Vec3::new(self.data[i], self.data[i + 1], self.data[i + 2])
    }
    pub fn linear(&self) -> &[Float] {
        &self.data
    }
    /// Color::gamma_correction + From<Color> for Rgb<u8> (image.rs:92-111).
    pub fn to_rgb8(&self) -> Result<Vec<u8>, Error> {
        let mut rgb = vec![0u8; self.data.len()];
        check(unsafe { sys::trt_tonemap_u8(self.data.as_ptr(), (self.width * self.height) as u32, self.gamma, rgb.as_mut_ptr()) })?;
        Ok(rgb)
    }
    /// Binary PPM; the reference writes PNG through the `image` crate (image.rs:66-69): feed `to_rgb8` to it for that.
    pub fn save(&self, filename: &str) -> Result<(), Box<dyn std::error::Error>> {
        let rgb = self.to_rgb8()?;
        let mut f = std::fs::File::create(filename)?;
        write!(f, "P6\n{} {}\n255\n", self.width, self.height)?;
        f.write_all(&rgb)?;
        Ok(())
    }
}

#[derive(Clone, Copy)]
pub struct Renderer {
    samples_per_pixel: usize,
    #[allow(dead_code)]
    num_sampler_threads: usize, // the GPU needs no sampler-task count; kept for signature parity
    max_bounces: usize,
    #[allow(dead_code)]
    progressbar: bool,
    background_color: Vec3,
    pub seed: u32,    // trt-rng v1 seed (the reference has no seed API)
    pub backend: u32, // one of the TRT_BACKEND constants of tinyrt-sys
    /// Scheduling knobs (`sys::trt_tuning`, from `Renderer::default_tuning()`); `None` = the library defaults.  Scheduling only:
    /// whatever the values, the frame is the same.
    pub tuning: Option<sys::trt_tuning>,
}

impl Renderer {
    pub fn new(samples_per_pixel: usize, num_sampler_threads: usize, max_bounces: usize, progressbar: bool,
               background_color: Option<Vec3>) -> Self {
        Renderer {
            samples_per_pixel,
            num_sampler_threads,
            max_bounces,
            progressbar,
            background_color: background_color.unwrap_or_else(Vec3::zero), // renderer.rs:33
            seed: 1,
            backend: sys::TRT_BACKEND_AUTO,
            tuning: None,
        }
    }

    /// The library's default tuning (built-in values; TRT_* environment variables override them once, when the library is loaded).
    pub fn default_tuning() -> sys::trt_tuning {
        let mut t = sys::trt_tuning::default();
        unsafe { sys::trt_tuning_default(&mut t) };
        t
    }

    /// Renderer::render (renderer.rs:37-79), synchronous; wrap in `tokio::task::spawn_blocking` for a JoinHandle<Image>.
    pub fn render(&self, camera: &Camera, world: &mut World) -> Result<Image, Error> {
        let (width, height) = camera.get_image_size();
        let scene = world.get_bvh()?;
        let params = sys::trt_render_params {
            samples_per_pixel: self.samples_per_pixel as u32,
            max_bounces: self.max_bounces as u32,
            background: self.background_color.raw(),
            seed: self.seed,
            backend: self.backend,
            tuning: self.tuning.as_ref().map_or(ptr::null(), |t| t as *const sys::trt_tuning),
            ..Default::default()
        };
        let mut data = vec![0.0 as Float; width * height * 3];
        check(unsafe { sys::trt_render(scene, &camera.pod, &params, data.as_mut_ptr(), ptr::null_mut()) })?;
        Ok(Image { width, height, gamma: 2.2, data })
    }

    /// The same call over `ndev` GPUs of the node (0 = every visible device): the image's 16-row bands are dealt round-robin
    /// over the devices and every finished band is copied to its place in the frame (trt_render_multi).
    pub fn render_multi(&self, camera: &Camera, world: &mut World, ndev: u32) -> Result<Image, Error> {
        let (width, height) = camera.get_image_size();
        let scene = world.get_bvh()?;
        let params = sys::trt_render_params {
            samples_per_pixel: self.samples_per_pixel as u32,
            max_bounces: self.max_bounces as u32,
            background: self.background_color.raw(),
            seed: self.seed,
            backend: self.backend,
            tuning: self.tuning.as_ref().map_or(ptr::null(), |t| t as *const sys::trt_tuning),
            ..Default::default()
        };
        let mut data = vec![0.0 as Float; width * height * 3];
        check(unsafe { sys::trt_render_multi(scene, &camera.pod, &params, ptr::null(), ndev, data.as_mut_ptr(), ptr::null_mut()) })?;
        Ok(Image { width, height, gamma: 2.2, data })
    }
}
