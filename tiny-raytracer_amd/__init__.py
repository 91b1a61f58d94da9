"""tiny-raytracer_amd — MI355X-native path-tracing sampler behind the reference crate's
World / Camera / Renderer surface.  All compute is in libtinyrt.so (HIP, gfx950); importing this
package fails if that library is not built.  The directory name has a hyphen, so import it with
importlib.import_module("tiny-raytracer_amd") or through the top-level alias module `tinyrt_amd`.
"""
from . import scenes  # noqa: F401
from ._lib import (BACKEND_AUTO, BACKEND_MEGAKERNEL, BACKEND_STREAMED, BACKEND_WAVEFRONT, DIELECTRIC, LAMBERTIAN, LIGHT, METAL, CameraPOD,  # noqa: F401
                   Material, Ray, RenderParams, SampledColor, SamplePoint, SceneOptions, Stats, TinyRTError, Tuning, Vec3, lib)
from .api import (Camera, Dielectric, Image, Lambertian, Light, Metal, Quad, Renderer, Scene, Sphere, World,  # noqa: F401
                  sample_batch, scene_options, tonemap_u8_device, tuning)

_MATERIAL_CTORS = {LAMBERTIAN: lambda a, p: Lambertian(a), METAL: Metal, DIELECTRIC: Dielectric,
                   LIGHT: lambda a, p: Light(a)}


def world_from_description(desc, **scene_options):
    """Build a product World (and its Camera) from a scenes.* description; `scene_options`: trt_scene_options fields its get_bvh() uses."""
    w = scenes.build_world(desc, World(), lambda k, a, p: _MATERIAL_CTORS[k](a, p), Sphere, Quad)
    if scene_options:
        w.scene_options = scene_options
    cam = Camera(**desc["camera"])
    return w, cam
