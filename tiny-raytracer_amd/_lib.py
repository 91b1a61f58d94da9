"""ctypes binding of libtinyrt.so — exactly the entry points include/tinyrt.h declares.

The library is the product: hand-written HIP for gfx950 behind a C ABI.  There is no Python or
CPU fallback; if the shared object is missing or fails to load, importing this module raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRT_LIB_PATH") or os.path.join(_HERE, "libtinyrt.so")     # override: A/B of two builds
CSRC = os.path.join(_HERE, "csrc")

TRT_OK = 0
ABI_VERSION = 4          # include/tinyrt.h TRT_ABI_VERSION
ERR_INVALID_ARG, ERR_DUPLICATE, ERR_NOT_FOUND, ERR_HIP, ERR_NO_DEVICE, ERR_OOM = -1, -2, -3, -4, -5, -6
LAMBERTIAN, METAL, DIELECTRIC, LIGHT = 0, 1, 2, 3
BACKEND_MEGAKERNEL, BACKEND_WAVEFRONT, BACKEND_AUTO, BACKEND_STREAMED = 0, 1, 2, 3


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(float(x), float(y), float(z))

    def tolist(self):
        return [self.x, self.y, self.z]


class Ray(C.Structure):
    _fields_ = [("origin", Vec3), ("direction", Vec3)]


class SamplePoint(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("ray", Ray)]


class SampledColor(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("color", Vec3)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", Vec3), ("param", C.c_float)]


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("num_nodes", "num_spheres", "num_quads", "num_materials", "max_depth", "device_bytes", "lds_bytes",
                 "num_cull_nodes")]


class CameraPOD(C.Structure):
    _fields_ = [("position", Vec3), ("viewport_upper_left", Vec3), ("forward", Vec3), ("horizontal", Vec3),
                ("vertical", Vec3), ("defocus_disk_u", Vec3), ("defocus_disk_v", Vec3),
                ("width", C.c_uint32), ("height", C.c_uint32)]


class BandCopy(C.Structure):
    _fields_ = [("rows_local", C.c_uint32), ("full_bands", C.c_uint32), ("tail_rows", C.c_uint32), ("reserved", C.c_uint32),
                ("band_bytes", C.c_uint64), ("local_pitch", C.c_uint64), ("frame_pitch", C.c_uint64), ("frame_offset", C.c_uint64),
                ("tail_bytes", C.c_uint64), ("tail_local_offset", C.c_uint64), ("tail_frame_offset", C.c_uint64)]


class LaunchPlan(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("scene_mode", "threads_per_workgroup", "waves_per_simd", "workgroups_per_cu", "lds_bytes", "scene_lds_bytes",
                 "leaf_slots", "lds_leaf_stack", "ray_pool", "walk", "specialised", "has_kernel", "kernel_waves_per_simd",
                 "kernel_threads", "kernel_walk", "kernel_ray_pool", "kernel_counting", "chunk_spp", "dual_walk")] + [("workspace_bytes", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Tuning(C.Structure):
    """tinyrt.h trt_tuning: scheduling / placement knobs of a render; every value renders the same frame."""
    FIELDS = ("stream_waves_per_simd", "stream_big_threads", "stream_batch_spp", "radiance_gb", "leaf_slots", "lds_leaf_stack", "ray_pool",
              "stragglers", "lds_stragglers", "dual_walk", "runtime_walk", "xcd_remap", "mega_waves_per_simd", "mega_threads",
              "mega_global_waves8", "wf_waves_per_simd", "wf_serve_min")
    _fields_ = [(n, C.c_uint32) for n in FIELDS] + [("reserved", C.c_uint32 * 7)]

    def as_dict(self):
        return {n: getattr(self, n) for n in self.FIELDS}


class SceneOptions(C.Structure):
    """tinyrt.h trt_scene_options: how a scene is compiled (placement only) and how much idle device scratch its handle keeps."""
    _fields_ = [("cull_prune", C.c_float), ("flat_walk", C.c_int32), ("compact_nodes", C.c_int32), ("top_nodes", C.c_uint32),
                ("scratch_cap_bytes", C.c_uint64), ("reserved", C.c_uint32 * 6)]


class RenderParams(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_uint32), ("max_bounces", C.c_uint32), ("background", Vec3),
                ("seed", C.c_uint32), ("backend", C.c_uint32),
                ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32), ("accumulate", C.c_uint32),
                ("band_rows", C.c_uint32), ("band_stride", C.c_uint32), ("band_offset", C.c_uint32),
                ("rows_local", C.c_uint32), ("collect_stats", C.c_uint32), ("tuning", C.POINTER(Tuning))]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("node_tests", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("quad_plane_tests", C.c_uint64), ("quad_inside_tests", C.c_uint64),
                ("shades", C.c_uint64), ("kernel_ms", C.c_double), ("wave_trips", C.c_uint64 * 4), ("gather_per_band", C.c_uint64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "wave_trips"}
        d["wave_trips"] = list(self.wave_trips)
        return d


# name -> (restype, argtypes); the test-suite checks this table against include/tinyrt.h
SIGNATURES = {
    "trt_world_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "trt_world_destroy": (None, [C.c_void_p]),
    "trt_world_add_material": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(Material)]),
    "trt_world_get_material": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32)]),
    "trt_world_add_sphere": (C.c_int, [C.c_void_p, Vec3, C.c_float, C.c_uint32]),
    "trt_world_add_spheres": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "trt_world_add_quad": (C.c_int, [C.c_void_p, Vec3, Vec3, Vec3, C.c_uint32]),
    "trt_world_num_geometries": (C.c_int, [C.c_void_p]),
    "trt_world_num_materials": (C.c_int, [C.c_void_p]),
    "trt_scene_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "trt_scene_options_default": (None, [C.POINTER(SceneOptions)]),
    "trt_scene_create_ex": (C.c_int, [C.c_void_p, C.POINTER(SceneOptions), C.POINTER(C.c_void_p)]),
    "trt_tuning_default": (None, [C.POINTER(Tuning)]),
    "trt_scene_destroy": (None, [C.c_void_p]),
    "trt_scene_trim": (C.c_int, [C.c_void_p]),
    "trt_scene_get_info": (C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
    "trt_scene_get_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "trt_scene_get_cull_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "trt_scene_get_compact_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "trt_camera_init": (C.c_int, [C.POINTER(CameraPOD), C.c_float, C.c_float, Vec3, Vec3, Vec3, C.c_float,
                                  C.c_uint32, C.c_uint32]),
    "trt_render": (C.c_int, [C.c_void_p, C.POINTER(CameraPOD), C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    "trt_render_multi": (C.c_int, [C.c_void_p, C.POINTER(CameraPOD), C.POINTER(RenderParams), C.POINTER(C.c_int), C.c_uint32,
                                   C.c_void_p, C.POINTER(Stats)]),
    "trt_render_multi_device": (C.c_int, [C.c_void_p, C.POINTER(CameraPOD), C.POINTER(RenderParams), C.POINTER(C.c_int),
                                          C.c_uint32, C.c_void_p, C.POINTER(Stats)]),
    "trt_band_rows_local": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "trt_band_copy_plan": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(BandCopy)]),
    "trt_streamed_launch_plan": (C.c_int, [C.c_void_p, C.POINTER(CameraPOD), C.POINTER(RenderParams), C.POINTER(LaunchPlan)]),
    "trt_kernel_timing_begin": (C.c_int, []),
    "trt_kernel_timing_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "trt_render_device": (C.c_int, [C.c_void_p, C.POINTER(CameraPOD), C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "trt_sample_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, Vec3, C.c_uint32,
                                   C.POINTER(Stats)]),
    "trt_streamed_chunk_spp": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "trt_tonemap_u8": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]),
    "trt_tonemap_u8_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]),
    "trt_dominant_kernel": (C.c_char_p, [C.c_void_p, C.POINTER(CameraPOD), C.POINTER(RenderParams)]),
    "trt_last_error": (C.c_char_p, []),
    "trt_device_count": (C.c_int, []),
    "trt_set_device": (C.c_int, [C.c_int]),
    "trt_abi_version": (C.c_uint32, []),
}


def build(force=False):
    """Compile libtinyrt.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-C", CSRC, "--no-print-directory"], check=True)
    return LIB_PATH


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same SONAMEs as /opt/rocm's).
    Two HSA runtimes in one process cannot both own the GPU, so when torch is installed its copy is mapped first
    and libtinyrt.so (DT_NEEDED libamdhip64.so.7) binds to that one: device pointers and streams handed over
    from torch then belong to the same runtime.  Without torch the system ROCm runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        return cand
    return None


def load():
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C tiny-raytracer_amd/csrc). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError here = ABI symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.trt_abi_version() != ABI_VERSION:
        raise ImportError("libtinyrt.so ABI version mismatch")
    return lib


lib = load()


class TinyRTError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"tinyrt error {code}: {message}")
        self.code = code


def check(rc):
    if rc != TRT_OK:
        raise TinyRTError(rc, lib.trt_last_error().decode("utf-8", "replace"))
