"""Host-side mirror of the reference crate's World / Camera / Renderer surface over the C ABI.

Names, argument order and meaning follow the reference (paths relative to raytracer/src):
  World.add_material / add_geometry / get_material / get_bvh      hittable/world.rs:16-45
  Sphere(center, radius, material), Quad(corner, u, v, material)  hittable/sphere.rs:16, quad.rs:20
  Lambertian / Metal / Dielectric / Light                          material/*.rs
  Camera(focus_distance, defocus_angle, position, look_at, up, vertical_fov, width, height)   camera.rs:17-26
  Renderer(samples_per_pixel, num_sampler_threads, max_bounces, progressbar, background_color)   renderer/renderer.rs:21-35
  Renderer.render(camera, world) -> Image                          renderer/renderer.rs:37-79
Where the reference panics (duplicate material name, world.rs:29-31) this raises TinyRTError.
Everything numeric happens inside libtinyrt.so; this file only moves arguments.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BACKEND_MEGAKERNEL, BACKEND_WAVEFRONT, DIELECTRIC, LAMBERTIAN, LIGHT, METAL, CameraPOD, Material,
                   RenderParams, SampledColor, SamplePoint, SceneInfo, SceneOptions, Stats, TinyRTError, Tuning, Vec3, check, lib)


def _v(v):
    return v if isinstance(v, Vec3) else Vec3(*v)


# ---- materials (material/{lambertian,metal,dielectric,light}.rs) ----
class Lambertian:
    def __init__(self, albedo):
        self.pod = Material(LAMBERTIAN, _v(albedo), 0.0)


class Metal:
    def __init__(self, albedo, fuzz):
        self.pod = Material(METAL, _v(albedo), float(fuzz))


class Dielectric:
    def __init__(self, albedo, refraction_index):
        self.pod = Material(DIELECTRIC, _v(albedo), float(refraction_index))


class Light:
    def __init__(self, color):
        self.pod = Material(LIGHT, _v(color), 0.0)


# ---- geometry (hittable/sphere.rs, quad.rs) ----
class Sphere:
    def __init__(self, center, radius, material):
        self.center, self.radius, self.material = _v(center), float(radius), int(material)


class Quad:
    def __init__(self, corner, u, v, material):
        self.corner, self.u, self.v, self.material = _v(corner), _v(u), _v(v), int(material)


def scene_options(**over):
    """The library's default trt_scene_options with `over` applied."""
    opt = SceneOptions()
    lib.trt_scene_options_default(C.byref(opt))
    for k, v in over.items():
        if k not in dict(SceneOptions._fields_) or k == "reserved":
            raise TypeError(f"unknown scene option {k!r}")
        setattr(opt, k, v)
    return opt


def tuning(**over):
    """The library's default trt_tuning (built-in values, overridden once at load by TRT_* environment variables) with `over` applied."""
    t = Tuning()
    lib.trt_tuning_default(C.byref(t))
    for k, v in over.items():
        if k not in Tuning.FIELDS:
            raise TypeError(f"unknown tuning field {k!r}")
        setattr(t, k, int(v))
    return t


class Scene:
    """World::get_bvh(): the reference-order BVH, packed for the GPU (uploaded on first render)."""

    def __init__(self, world, **options):
        """options: fields of tinyrt.h trt_scene_options (cull_prune, flat_walk, compact_nodes, top_nodes, scratch_cap_bytes) - placement
        only: whatever they are, the scene renders the same frames."""
        self._h = C.c_void_p()
        opt = scene_options(**options)
        check(lib.trt_scene_create_ex(world._h, C.byref(opt), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.trt_scene_destroy(self._h)
            self._h = None

    def info(self):
        out = SceneInfo()
        check(lib.trt_scene_get_info(self._h, C.byref(out)))
        return {n: getattr(out, n) for n, _ in out._fields_}

    def _dump(self, fn, n):
        bbox = np.zeros((n, 6), np.float32)
        prim = np.zeros(n, np.int32)
        skip = np.zeros(n, np.int32)
        check(fn(self._h, bbox.ctypes.data, prim.ctypes.data, skip.ctypes.data, n))
        return bbox, prim, skip

    def nodes(self):
        """The reference tree (bvh.rs:42-84 node for node), pre-order: (bbox[n,6], prim[n], skip[n])."""
        return self._dump(lib.trt_scene_get_nodes, self.info()["num_nodes"])

    def cull_nodes(self):
        """The culling tree the kernels walk: same leaves in the same order, re-clustered inner nodes."""
        return self._dump(lib.trt_scene_get_cull_nodes, self.info()["num_cull_nodes"])


    def compact_nodes(self):
        """The culling tree as 16-byte nodes (f16 boxes rounded outward) if the scene is walked from global memory:
        (lo[n,3] float16, hi[n,3] float16, link[n] uint32), else None."""
        n = self.info()["num_cull_nodes"]
        words = np.zeros((n, 4), np.uint32)
        rc = lib.trt_scene_get_compact_nodes(self._h, words.ctypes.data, n)
        if rc == _lib.ERR_NOT_FOUND:
            return None
        check(rc)
        h = words[:, :3].copy().view(np.float16).reshape(n, 6)
        return h[:, :3], h[:, 3:], words[:, 3].copy()


class World:
    def __init__(self):
        self._h = C.c_void_p()
        check(lib.trt_world_create(C.byref(self._h)))
        self._scene = None
        self._scene_options = {}

    @property
    def scene_options(self):
        """trt_scene_options fields get_bvh() compiles the scene with (placement only: the frames never change)."""
        return dict(self._scene_options)

    @scene_options.setter
    def scene_options(self, options):
        self._scene_options = dict(options)
        self._scene = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.trt_world_destroy(self._h)
            self._h = None

    def add_material(self, name, material):
        check(lib.trt_world_add_material(self._h, name.encode(), C.byref(material.pod)))

    def get_material(self, name):
        idx = C.c_uint32()
        rc = lib.trt_world_get_material(self._h, name.encode(), C.byref(idx))
        if rc == _lib.ERR_NOT_FOUND:
            return None                      # world.rs:35-41 returns Option
        check(rc)
        return idx.value

    def add_geometry(self, geometry):
        self._scene = None
        if isinstance(geometry, Sphere):
            check(lib.trt_world_add_sphere(self._h, geometry.center, geometry.radius, geometry.material))
        elif isinstance(geometry, Quad):
            check(lib.trt_world_add_quad(self._h, geometry.corner, geometry.u, geometry.v, geometry.material))
        else:
            raise TypeError("geometry must be a Sphere or a Quad")

    def add_spheres(self, center_radius, material):
        """n x add_geometry(Sphere(...)) in array order, one call: center_radius float32 [n, 4] (x, y, z, radius), material uint32 [n]."""
        self._scene = None
        cr = np.ascontiguousarray(center_radius, np.float32).reshape(-1, 4)
        m = np.ascontiguousarray(material, np.uint32).reshape(-1)
        if len(m) != len(cr):
            raise ValueError("one material index per sphere")
        check(lib.trt_world_add_spheres(self._h, len(cr), cr.ctypes.data, m.ctypes.data))

    def num_geometries(self):
        return lib.trt_world_num_geometries(self._h)

    def get_bvh(self, **options):
        """World::get_bvh (world.rs:43-45).  `options`: see Scene; a scene compiled with options is not cached."""
        if options:
            return Scene(self, **options)
        if self._scene is None:
            self._scene = Scene(self, **self._scene_options)
        return self._scene


class Camera:
    def __init__(self, focus_distance, defocus_angle, position, look_at, up, vertical_fov, width, height):
        self.pod = CameraPOD()
        check(lib.trt_camera_init(C.byref(self.pod), focus_distance, defocus_angle, _v(position), _v(look_at), _v(up),
                                  vertical_fov, width, height))

    def get_image_size(self):
        return (self.pod.width, self.pod.height)


class Image:
    """utils/image.rs Image with gamma: holds the Imager's linear f32 sums; quantises on demand."""

    def __init__(self, accum, gamma=2.2):
        self.data = accum                    # (H, W, 3) float32, linear
        self.gamma = gamma

    @property
    def height(self):
        return self.data.shape[0]

    @property
    def width(self):
        return self.data.shape[1]

    def size(self):
        return (self.width, self.height)

    def to_u8(self):
        rgb = np.zeros(self.data.shape, np.uint8)
        src = np.ascontiguousarray(self.data, np.float32)
        check(lib.trt_tonemap_u8(src.ctypes.data, self.width * self.height, self.gamma, rgb.ctypes.data))
        return rgb

    def save(self, filename):
        from PIL import Image as PILImage
        PILImage.fromarray(self.to_u8(), "RGB").save(filename)


class Renderer:
    def __init__(self, samples_per_pixel, num_sampler_threads=1, max_bounces=50, progressbar=False,
                 background_color=None, seed=1, backend=_lib.BACKEND_AUTO):
        self.samples_per_pixel = int(samples_per_pixel)
        self.num_sampler_threads = int(num_sampler_threads)      # kept for signature parity; the GPU ignores it
        self.max_bounces = int(max_bounces)
        self.progressbar = bool(progressbar)
        self.background_color = _v(background_color) if background_color is not None else Vec3(0.0, 0.0, 0.0)
        self.seed = int(seed)
        self.backend = int(backend)
        self.last_stats = None
        self.tuning = {}                 # trt_tuning fields this renderer overrides (scheduling only: frames never change)

    def params(self, tuning=None, **over):
        """trt_render_params for this renderer; `tuning`: dict of trt_tuning fields for this call (on top of self.tuning)."""
        p = RenderParams()
        knobs = dict(self.tuning)
        knobs.update(tuning or {})
        if knobs:
            p._tuning_keepalive = globals()["tuning"](**knobs)       # the POD holds a pointer: keep the struct alive with it
            p.tuning = C.pointer(p._tuning_keepalive)
        p.samples_per_pixel = self.samples_per_pixel
        p.max_bounces = self.max_bounces
        p.background = self.background_color
        p.seed = self.seed
        p.backend = self.backend
        for k, v in over.items():
            setattr(p, k, v)
        return p

    def render(self, camera, world, collect_stats=False, accum=None, **over):
        """Renderer::render: returns the finished Image (the reference returns a JoinHandle<Image>).
        `accum` continues earlier passes when over["accumulate"] is set."""
        scene = world.get_bvh() if isinstance(world, World) else world
        p = self.params(collect_stats=int(collect_stats), **over)      # 0 | 1 (reference tree) | 2 (culling tree)
        w, h = camera.get_image_size()
        rows = p.rows_local if p.band_rows else h
        if accum is None:
            accum = np.zeros((rows, w, 3), np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.shape == (rows, w, 3)
        st = Stats()
        check(lib.trt_render(scene._h, C.byref(camera.pod), C.byref(p), accum.ctypes.data, C.byref(st)))
        self.last_stats = st.as_dict()
        return Image(accum)

    def render_multi(self, camera, world, devices=None, accum=None, d_accum_ptr=None, collect_stats=False, **over):
        """Renderer::render over several GPUs of one node (trt_render_multi): one call, the whole Image back.  `devices`:
        list of ordinals (None = all visible).  With d_accum_ptr the frame is gathered into that buffer on devices[0]
        instead (trt_render_multi_device) and nothing is returned."""
        scene = world.get_bvh() if isinstance(world, World) else world
        p = self.params(collect_stats=int(collect_stats), **over)
        w, h = camera.get_image_size()
        ndev = len(devices) if devices is not None else 0
        dv = (C.c_int * ndev)(*devices) if ndev else None
        st = Stats()
        if d_accum_ptr is not None:
            check(lib.trt_render_multi_device(scene._h, C.byref(camera.pod), C.byref(p), dv, ndev, C.c_void_p(d_accum_ptr), C.byref(st)))
            self.last_stats = st.as_dict()
            return None
        if accum is None:
            accum = np.zeros((h, w, 3), np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.shape == (h, w, 3)
        check(lib.trt_render_multi(scene._h, C.byref(camera.pod), C.byref(p), dv, ndev, accum.ctypes.data, C.byref(st)))
        self.last_stats = st.as_dict()
        return Image(accum)

    def launch_plan(self, camera, scene, **over):
        """trt_streamed_launch_plan: how the streamed backend launches this render (host arithmetic only), as a dict."""
        plan = _lib.LaunchPlan()
        p = self.params(**over)
        check(lib.trt_streamed_launch_plan(scene._h, C.byref(camera.pod), C.byref(p), C.byref(plan)))
        return {n: getattr(plan, n) for n, _ in plan._fields_}

    def render_device(self, camera, scene, d_accum_ptr, stream_ptr=0, d_counters_ptr=0, **over):
        """Enqueue one pass on buffers already in HBM (device pointers as integers); asynchronous."""
        p = self.params(**over)
        check(lib.trt_render_device(scene._h, C.byref(camera.pod), C.byref(p), C.c_void_p(d_accum_ptr),
                                    C.c_void_p(d_counters_ptr), C.c_void_p(stream_ptr)))


def tonemap_u8_device(d_accum_ptr, npixels, d_rgb_ptr, gamma=2.2, stream_ptr=0):
    """Imager finalisation on buffers in HBM (device pointers as integers): linear f32 sums -> gamma-corrected RGB8."""
    check(lib.trt_tonemap_u8_device(C.c_void_p(d_accum_ptr), int(npixels), float(gamma), C.c_void_p(d_rgb_ptr),
                                    C.c_void_p(stream_ptr)))


def sample_batch(scene, points, max_bounces, background, seed=1, collect_stats=True):
    """trait Sampler in batch form: ctypes array of SamplePoint -> SampledColor.  collect_stats=False runs the production walk
    (no traversal counters) instead of the counting kernel on the reference tree."""
    n = len(points)
    out = (SampledColor * max(n, 1))()
    st = Stats()
    check(lib.trt_sample_batch(scene._h, C.byref(points) if n else None, n, C.byref(out), max_bounces, _v(background),
                               seed, C.byref(st) if collect_stats else None))
    return out, st.as_dict()
