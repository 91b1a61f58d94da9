"""CPU-side checks of the product: the C ABI library loads and exports every symbol include/tinyrt.h declares,
POD layouts match the reference's #[repr(C)] structs, the host scene compiler reproduces the reference's BVH
(compared with the oracle's pointer tree), Camera::new matches, errors come back as codes.  No compute calls:
those need a GPU and live in test_gpu_parity.py."""
import ctypes as C
import os
import re

import numpy as np
import pytest
from conftest import sym

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(trt):
    header = open(os.path.join(ROOT, "include", "tinyrt.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(trt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    from importlib import import_module
    _lib = import_module("tiny-raytracer_amd._lib")
    raw = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in tinyrt.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert trt.lib.trt_abi_version() == trt._lib.ABI_VERSION == 4


def test_pod_layouts_match_reference(trt):
    # Vec3 12 B, Ray 24 B, SamplePoint 32 B (pointgen.rs:7-13), SampledColor 20 B (imager.rs:9-15)
    assert C.sizeof(trt.Vec3) == 12 and C.sizeof(trt.Ray) == 24
    assert C.sizeof(trt.SamplePoint) == 32 and C.sizeof(trt.SampledColor) == 20
    assert trt.SamplePoint.ray.offset == 8 and trt.SampledColor.color.offset == 8


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "tiny-raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "rt_oracle" not in text and "liboracle" not in text and "oracle." not in text.replace("oracle wrapper", ""), fn


@pytest.mark.parametrize("scene", ["cornell", "dummy", "quad_test", "random_spheres", "grid3000"])
def test_scene_compiler_reproduces_reference_bvh(trt, orc, scene):
    desc = {"cornell": lambda: trt.scenes.cornell(), "dummy": lambda: trt.scenes.dummy_spheres("renderer"),
            "quad_test": lambda: trt.scenes.quad_test(), "random_spheres": lambda: trt.scenes.random_spheres(),
            "grid3000": lambda: trt.scenes.sphere_grid(3000, 64, 36)}[scene]()
    pw, pcam = trt.world_from_description(desc)
    ow, ocam = orc.world_from_description(desc)
    bbox, prim, skip = pw.get_bvh().nodes()
    obox, oprim, osub = ow.bvh_dump()
    n = len(desc["geometries"])
    assert len(prim) == 2 * n - 1                                  # one primitive per leaf (bvh.rs:49-57)
    assert np.array_equal(bbox.view(np.uint32), obox.view(np.uint32))
    assert np.array_equal(prim, oprim)
    assert np.array_equal(skip, np.arange(len(skip)) + osub)       # skip link = pre-order index after the subtree
    assert sorted(prim[prim >= 0].tolist()) == list(range(n))
    assert bytes(pcam.pod) == bytes(ocam)                          # Camera::new, bit for bit
    info = pw.get_bvh().info()
    assert info["num_nodes"] == 2 * n - 1 and info["num_spheres"] + info["num_quads"] == n


def test_camera_new_kat_through_c_abi(trt, golden):
    g = golden["camera_new"]
    cam = trt.Camera(g["focus_distance"], g["defocus_angle"], sym(g["position"]), sym(g["look_at"]), sym(g["up"]),
                     g["vertical_fov"], g["width"], g["height"])
    for field, expected in g["expected"].items():
        got = np.array(getattr(cam.pod, field).tolist())
        assert np.abs(got - np.array(sym(expected))).max() < 1e-6, field       # reference asserts with tolerant ==
    assert cam.get_image_size() == (16, 9)


def test_world_errors_are_codes_not_panics(trt):
    w = trt.World()
    w.add_material("red", trt.Lambertian((1, 0, 0)))
    with pytest.raises(trt.TinyRTError) as e:                      # world.rs:29-31 panics here
        w.add_material("red", trt.Lambertian((0, 1, 0)))
    assert e.value.code == -2 and "already in the material table" in str(e.value)
    assert w.get_material("nope") is None                          # world.rs:35-41 -> None
    assert w.get_material("red") == 0
    with pytest.raises(trt.TinyRTError):
        w.add_geometry(trt.Sphere((0, 0, 0), 1.0, 7))              # dangling material handle
    with pytest.raises(trt.TinyRTError) as e:
        w.get_bvh()                                                # empty world: BVH::new would index objects[0]
    assert e.value.code == -1
    with pytest.raises(trt.TinyRTError):
        trt.Camera(1.0, 0.0, (0, 0, 0), (0, 0, 1), (0, 1, 0), 90.0, 0, 9)


def test_metal_fuzz_is_clamped_like_metal_new(trt, orc):
    # Metal::new clamps fuzz to [0,1] (metal.rs:12-14); checked through the packed scene
    desc = dict(materials=[("m", 1, (0.5, 0.5, 0.5), 7.0)], geometries=[("sphere", (0, 0, 0), 1.0, "m")],
                camera=dict(focus_distance=1.0, defocus_angle=0.0, position=(0, 0, 3), look_at=(0, 0, 0), up=(0, 1, 0),
                            vertical_fov=40.0, width=4, height=4))
    pw, _ = trt.world_from_description(desc)
    assert pw.get_bvh().info()["num_materials"] == 1


def test_compute_calls_fail_loudly_without_a_gpu(trt):
    if trt.lib.trt_device_count() > 0:
        pytest.skip("a GPU is present")
    w, cam = trt.world_from_description(trt.scenes.cornell(16, 16))
    with pytest.raises(trt.TinyRTError) as e:
        trt.Renderer(1, 1, 2, False, (0, 0, 0)).render(cam, w)
    assert e.value.code == -5                                      # TRT_ERR_NO_DEVICE: no CPU fallback exists


def test_tonemap_matches_oracle_and_reference_rules(trt, orc):
    rng = np.random.default_rng(0)
    acc = rng.uniform(-0.2, 1.4, (17, 9, 3)).astype(np.float32)
    acc[0, 0] = [np.nan, 0.0, 1.0]
    acc[0, 1] = [np.inf, -np.inf, 0.999 ** 2.2]
    mine = trt.Image(acc).to_u8()
    ref = orc.tonemap_u8(acc)
    assert np.array_equal(mine, ref)
    assert mine[0, 0].tolist() == [0, 0, 254]                      # NaN -> 0 (image.rs:106-108), clamp at 0.999
    assert mine.max() == 254
    # any bit pattern, several gammas: host form (trt_pow.h) == oracle (rt_oracle.c m_powf), byte for byte
    vals = np.concatenate([rng.integers(0, 2**32, 299_997, dtype=np.uint32).view(np.float32),
                           np.exp(rng.uniform(-100, 90, 300_000)).astype(np.float32), rng.random(300_000).astype(np.float32)])
    with np.errstate(all="ignore"):
        for gamma in (2.2, 1.0, 1.8, 0.45, 3.0):
            assert np.array_equal(trt.Image(vals.reshape(1, -1, 3), gamma).to_u8(), orc.tonemap_u8(vals.reshape(1, -1, 3), gamma)), gamma


def test_band_layout_covers_image_once(trt):
    from importlib import import_module
    tiles = import_module("tiny-raytracer_amd.tiles")
    for height in (1, 15, 16, 17, 100, 2048):
        for ws in (1, 2, 3, 8):
            rows = []
            for r in range(ws):
                lay = tiles.band_layout(height, ws, r)
                assert lay["rows_local"] == len(lay["rows"])
                for local, y in enumerate(lay["rows"]):          # the mapping documented in tinyrt.h
                    assert y == ((local // 16) * ws + r) * 16 + local % 16
                rows += lay["rows"]
            assert sorted(rows) == list(range(height))


def _build_cpp_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "cornell")
    libdir = os.path.join(ROOT, "tiny-raytracer_amd")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "cornell.cpp"), "-L" + libdir, "-ltinyrt", "-Wl,-rpath," + libdir, "-o", exe],
                   check=True)
    return exe


def test_cpp_mirror_compiles_and_fails_loudly_without_gpu(trt, tmp_path):
    """include/tinyrt.hpp (World/Camera/Renderer in C++, as the reference is compiled code) builds against the C ABI;
    without a GPU the render throws the library's NO_DEVICE error instead of falling back to anything."""
    import subprocess
    exe = _build_cpp_example(tmp_path)
    if trt.lib.trt_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "16", "16", "1"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "no HIP device visible" in r.stderr


@pytest.mark.parametrize("scene", ["cornell", "random_spheres", "grid3000", "mixed"])
def test_culling_tree_is_a_hierarchy_over_the_reference_leaf_sequence(trt, scene):
    """The kernels walk a re-clustered tree.  It gives bit-identical hits because (scene.h): same leaves, same order,
    every inner box the exact union of the leaf boxes below it.  Those three facts are checked here; the images are
    checked against the oracle in the GPU tests."""
    if scene == "mixed":
        desc = trt.scenes.cornell()
        desc["materials"].append(("glass", 2, (1.0, 1.0, 1.0), 1.5))
        desc["geometries"] += [("sphere", (30.0, 70.0, 30.0), 9.0, "glass"), ("sphere", (70.0, 45.0, 60.0), 14.0, "white")]
    else:
        desc = {"cornell": lambda: trt.scenes.cornell(), "random_spheres": lambda: trt.scenes.random_spheres(),
                "grid3000": lambda: trt.scenes.sphere_grid(3000, 64, 36)}[scene]()
    pw, _ = trt.world_from_description(desc)
    sc = pw.get_bvh()
    rb, rp, rs = sc.nodes()
    cb, cp, cs = sc.cull_nodes()
    ref_leaves = np.flatnonzero(rp >= 0)
    cull_leaves = np.flatnonzero(cp >= 0)
    assert np.array_equal(rp[ref_leaves], cp[cull_leaves])                       # same primitives, same order
    assert np.array_equal(rb[ref_leaves].view(np.uint32), cb[cull_leaves].view(np.uint32))   # same leaf boxes, bit for bit
    n = len(cp)
    assert len(cp) <= len(rp) and (cs > np.arange(n)).all() and cs.max() == n and cs[0] <= n
    leaf_rank = np.cumsum(cp >= 0) - (cp >= 0)                                   # leaves before node i
    for i in range(n):
        if cp[i] >= 0:
            assert cs[i] == i + 1
            continue
        inside = cull_leaves[(cull_leaves > i) & (cull_leaves < cs[i])]
        assert len(inside) >= 2
        lo = cb[inside, :3].min(axis=0)
        hi = cb[inside, 3:].max(axis=0)
        assert np.array_equal(cb[i, :3], lo) and np.array_equal(cb[i, 3:], hi)   # exact union, no slack and no shortfall
        # children are complete subtrees: every node strictly inside (i, skip[i]) ends inside
        assert (cs[i + 1:cs[i]] <= cs[i]).all()
    assert leaf_rank[-1] + (cp[-1] >= 0) == len(ref_leaves)


@pytest.mark.parametrize("size", [(37, 23), (300, 260)])              # the second one needs more than one 65535-byte stored block
def test_cpp_image_save_writes_a_valid_png(trt, tmp_path, size):
    """tinyrt::Image::save (utils/image.rs:66-69 writes PNG): the header-only PNG writer decodes (PIL) to exactly the
    library's quantised frame, and the PPM form carries the same bytes."""
    import subprocess
    from PIL import Image as PILImage
    w, h = size
    exe = str(tmp_path / "image_save_check")
    libdir = os.path.join(ROOT, "tiny-raytracer_amd")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "native", "image_save_check.cpp"), "-L" + libdir, "-ltinyrt", "-Wl,-rpath," + libdir,
                    "-o", exe], check=True)
    png, ppm = str(tmp_path / "o.png"), str(tmp_path / "o.ppm")
    subprocess.run([exe, str(w), str(h), png, ppm], check=True)
    x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    lin = np.stack([x / np.float32(w), y / np.float32(h) * np.float32(1.5),
                    np.where((x + y) % 7 == 0, np.float32(-0.25), np.float32(0.18))], axis=-1).astype(np.float32)
    with np.errstate(invalid="ignore"):
        want = trt.Image(lin).to_u8()
    got = np.asarray(PILImage.open(png).convert("RGB"))
    assert got.shape == (h, w, 3) and np.array_equal(got, want)
    assert np.array_equal(np.asarray(PILImage.open(ppm).convert("RGB")), want)
    assert want[..., 1].max() == 254 and want[0, 0, 2] == 0


def test_compact_nodes_are_the_culling_tree_rounded_outward(trt):
    """Scenes walked from global memory carry the culling tree as 16-byte nodes with f16 boxes.  The walk stays exact
    only if every f16 box CONTAINS the f32 box (then it passes whenever the exact box passes); it stays cheap if each
    bound is the nearest such f16.  Links: an inner node's skip, a leaf's sequence number."""
    desc = trt.scenes.sphere_grid(3000, 64, 36)                              # hot part > 64 KB
    pw, _ = trt.world_from_description(desc)
    sc = pw.get_bvh()
    assert sc.info()["lds_bytes"] == 0
    lo16, hi16, link = sc.compact_nodes()
    box, prim, skip = sc.cull_nodes()
    lo32, hi32 = box[:, :3], box[:, 3:]
    lo, hi = lo16.astype(np.float32), hi16.astype(np.float32)
    assert (lo <= lo32).all() and (hi >= hi32).all()
    # round 5: the boxes are first grown by eps = 2^-19 B per axis (B = the tree's largest |coordinate| on the axis) - the slack the fused slab
    # arithmetic of the hand-written walk needs (rt_path.h box_loop_compact) - and THEN rounded outward to the nearest f16
    B = np.maximum(np.abs(lo32[0]), np.abs(hi32[0])).astype(np.float32)
    eps = (B * np.float32(2.0 ** -19)).astype(np.float32)
    lo_grown, hi_grown = (lo32 - eps).astype(np.float32), (hi32 + eps).astype(np.float32)
    assert (lo <= lo_grown).all() and (hi >= hi_grown).all()
    # tightest: one f16 step inwards would cut into the grown box
    assert (np.nextafter(lo16, np.float16(np.inf)).astype(np.float32) > lo_grown).all()
    assert (np.nextafter(hi16, np.float16(-np.inf)).astype(np.float32) < hi_grown).all()
    leaf = prim >= 0
    assert ((link & 0x80000000) != 0).tolist() == leaf.tolist()
    assert np.array_equal(link[~leaf], skip[~leaf].astype(np.uint32))
    assert np.array_equal(link[leaf] & 0x7FFFFFFF, np.arange(leaf.sum(), dtype=np.uint32))
    # a scene that fits LDS has no such array
    small, _ = trt.world_from_description(trt.scenes.cornell())
    assert small.get_bvh().compact_nodes() is None


def test_fused_slab_arithmetic_on_the_grown_f16_boxes_is_conservative(trt):
    """rt_path.h box_loop_compact (round 5) evaluates a plane's distance as fma(x, 1/d, -fl(o / d)) on the 16-byte nodes instead of the reference's
    fl(fl(x - o) * 1/d).  Replayed on the host - the fused form in extended precision rounded once, the reference's form in float32 step by step - for
    random rays from inside the domain (|o| <= 4 B per axis, its very edge included) against EVERY node: whenever the reference's slab test passes on a
    node's exact f32 box (aabb.rs:36-61 with t in [0.001, t_best)), the fused test passes on its grown f16 box, and its interval starts no later."""
    desc = trt.scenes.sphere_grid(3000, 64, 36)
    sc = trt.world_from_description(desc)[0].get_bvh()
    lo16, hi16, _ = sc.compact_nodes()
    box, _, _ = sc.cull_nodes()
    lo32, hi32 = box[:, :3].astype(np.float32), box[:, 3:].astype(np.float32)
    lo_c, hi_c = lo16.astype(np.float32), hi16.astype(np.float32)
    B = np.maximum(np.abs(lo32[0]), np.abs(hi32[0])).astype(np.float32)
    rng = np.random.default_rng(5)
    f32, ld = np.float32, np.longdouble
    checked = passed = 0
    for r in range(300):
        if r % 4 == 0:                                                            # on the domain's edge, in a corner of it
            o = (rng.choice([-4.0, 4.0], 3) * B).astype(f32)
        elif r % 4 == 1:                                                          # anywhere in the domain
            o = (rng.uniform(-4.0, 4.0, 3) * B).astype(f32)
        else:                                                                     # where the rays of a render start: among the spheres
            o = np.array([rng.uniform(-30, 30), rng.uniform(0.0, 0.5), rng.uniform(-30, 30)], f32)
        d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(f32)
        d[np.abs(d) < 1e-6] = f32(1e-6)
        inv = (f32(1.0) / d).astype(f32)
        t_best = f32(np.inf) if r % 3 == 0 else f32(rng.uniform(0.5, 3000.0))
        with np.errstate(over="ignore", invalid="ignore"):
            # the reference on the exact box, float32 step by step
            a, b = ((lo32 - o).astype(f32) * inv).astype(f32), ((hi32 - o).astype(f32) * inv).astype(f32)
            start = np.maximum(np.minimum(a, b).max(axis=1), f32(0.001))
            end = np.minimum(np.maximum(a, b).min(axis=1), t_best)
            ref_pass = ~(end <= start)
            # the fused form on the grown f16 box: m = fl(o * inv); g = fl(x * inv - m), one rounding
            m = (o * inv).astype(f32)
            ga = (lo_c.astype(ld) * inv.astype(ld) - m.astype(ld)).astype(f32)
            gb = (hi_c.astype(ld) * inv.astype(ld) - m.astype(ld)).astype(f32)
            cstart = np.maximum(np.minimum(ga, gb).max(axis=1), f32(0.001))
            cend = np.minimum(np.maximum(ga, gb).min(axis=1), t_best)
            coarse_pass = ~(cend <= cstart)
        assert not (ref_pass & ~coarse_pass).any(), (r, o, d)
        assert (cstart[ref_pass] <= start[ref_pass]).all(), (r, o, d)
        checked += len(ref_pass)
        passed += int(ref_pass.sum())
    assert checked > 1e6 and passed > 1000


def test_f16_outward_rounding_handles_the_edges(trt):
    """Coordinates beyond the f16 range, zeros, subnormals and exact f16 values (trt_scene_options.compact_nodes = 1 forces the array
    for a small scene)."""
    mats = [("m", 0, (0.5, 0.5, 0.5), 0.0)]
    geos = [("sphere", (1.0e5, 0.0, 0.0), 1.0, "m"), ("sphere", (-7.0e4, 3.0e-6, -3.0e-6), 0.5, "m"),
            ("sphere", (0.5, 2.0, -1.0), 0.25, "m"), ("sphere", (1.0e-7, -1.0e-7, 0.0), 1.0e-7, "m"),
            ("quad", (0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), "m"), ("sphere", (65504.0, -65504.0, 2049.0), 1.0, "m")]
    cam = dict(focus_distance=1.0, defocus_angle=0.0, position=(0.0, 0.0, 5.0), look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
               vertical_fov=40.0, width=8, height=8)
    pw, _ = trt.world_from_description(dict(name="edges", materials=mats, geometries=geos, camera=cam, background=(0, 0, 0)), compact_nodes=1)
    sc = pw.get_bvh()
    lo16, hi16, _ = sc.compact_nodes()
    box, _, _ = sc.cull_nodes()
    lo, hi = lo16.astype(np.float32), hi16.astype(np.float32)
    assert (lo <= box[:, :3]).all() and (hi >= box[:, 3:]).all()
    assert np.isinf(hi).any() and np.isinf(lo).any()                             # beyond 65504: the conservative infinity
    fin = np.isfinite(lo) & np.isfinite(hi)
    B = np.maximum(np.abs(box[0, :3]), np.abs(box[0, 3:])).astype(np.float32)
    eps = (B * np.float32(2.0 ** -19)).astype(np.float32)                        # the slack of the fused slab arithmetic (0.19 here: B = 1e5)
    with np.errstate(over="ignore"):
        assert (np.nextafter(lo16, np.float16(np.inf)).astype(np.float32)[fin] > (box[:, :3] - eps).astype(np.float32)[fin]).all()
        assert (np.nextafter(hi16, np.float16(-np.inf)).astype(np.float32)[fin] < (box[:, 3:] + eps).astype(np.float32)[fin]).all()


def test_header_is_valid_c_and_the_c_example_fails_loudly_without_gpu(trt, tmp_path):
    """include/tinyrt.h is a C header: examples/minimal.c builds with gcc -std=c11 -pedantic against it; without a GPU
    the example exits 1 with the library's NO_DEVICE message."""
    import subprocess
    exe = str(tmp_path / "minimal")
    libdir = os.path.join(ROOT, "tiny-raytracer_amd")
    subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "minimal.c"), "-L" + libdir, "-ltinyrt", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, cwd=str(tmp_path))
    if trt.lib.trt_device_count() > 0:
        assert r.returncode == 0 and "rays in" in r.stdout and os.path.exists(tmp_path / "minimal.ppm")
    else:
        assert r.returncode == 1 and "no HIP device visible" in r.stderr


def _build_multi_gpu_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "multi_gpu")
    libdir = os.path.join(ROOT, "tiny-raytracer_amd")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "multi_gpu.c"), "-L" + libdir, "-ltinyrt", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_multi_gpu_example_builds_as_c_and_fails_loudly_without_gpu(trt, tmp_path):
    """examples/multi_gpu.c: trt_render_multi_device over every visible device into a frame in HBM.  Built here as C11; without a
    GPU it must say so and exit 2 (on the GPU box tests/test_gpu_multi.py runs it)."""
    import subprocess
    exe = _build_multi_gpu_example(tmp_path)
    if trt.lib.trt_device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no GPU visible" in r.stderr


def test_c_band_layout_equals_the_python_one(trt):
    """trt_render_multi (C, one host thread per device) and tiles.py (one process per GPU over torch.distributed) must cut
    the image the same way: 16-row bands dealt round-robin."""
    from importlib import import_module
    tiles = import_module("tiny-raytracer_amd.tiles")
    for height in (1, 15, 16, 17, 150, 1080, 2048, 2160):
        for ndev in (1, 2, 3, 4, 7, 8, 13):
            total = 0
            for rank in range(ndev):
                rows = C.c_uint32(0)
                assert trt.lib.trt_band_rows_local(height, ndev, rank, C.byref(rows)) == 0
                assert rows.value == tiles.band_layout(height, ndev, rank)["rows_local"], (height, ndev, rank)
                total += rows.value
            assert total == height
    assert trt.lib.trt_band_rows_local(100, 0, 0, C.byref(C.c_uint32())) == -1
    assert trt.lib.trt_band_rows_local(100, 2, 2, C.byref(C.c_uint32())) == -1


# ---- round 3: the streamed backend's launch plan and the multi-GPU gather, checked without a device ----
def _plan(trt, scene, cam, tuning=None, **over):
    import ctypes as C
    p = trt.Renderer(64, 1, 50, False, (0.1, 0.1, 0.1)).params(tuning=tuning, **over)
    out = trt._lib.LaunchPlan()
    trt._lib.check(trt.lib.trt_streamed_launch_plan(scene._h, C.byref(cam.pod), C.byref(p), C.byref(out)))
    return out.as_dict()


def _check_plan(pl, stats, tag):
    """What the kernels assume about their launch (streamed.hip): LDS = scene copy | leaf stack (threads x slots x 8 B) | ray pool
    (36 B per lane); the lock-step walk pushes two leaves per trip; every plan has an instantiation of the planned shape."""
    al = (pl["scene_lds_bytes"] + 15) & ~15
    assert pl["has_kernel"] == 1, tag
    assert pl["kernel_threads"] == pl["threads_per_workgroup"] and pl["threads_per_workgroup"] in (256, 512, 768), tag
    assert 1 <= pl["leaf_slots"] <= 16, tag
    if pl["lds_leaf_stack"]:
        stacks = 2 if pl["dual_walk"] else 1                                      # two paths per lane: two leaf stacks per lane
        assert pl["lds_bytes"] == al + stacks * pl["threads_per_workgroup"] * pl["leaf_slots"] * 8 + (pl["threads_per_workgroup"] * 36 if pl["ray_pool"] else 0), tag
        if pl["dual_walk"]:
            assert pl["walk"] == 3 and pl["ray_pool"] and pl["specialised"] and pl["leaf_slots"] >= 2 and not stats, tag      # a parked walk needs two slots
    else:
        assert pl["lds_bytes"] == pl["scene_lds_bytes"] and not pl["ray_pool"] and pl["walk"] == 5, tag
    assert pl["lds_bytes"] <= 160 * 1024 and pl["lds_bytes"] * pl["workgroups_per_cu"] <= 160 * 1024, tag
    assert 1 <= pl["workgroups_per_cu"] and pl["workgroups_per_cu"] * pl["threads_per_workgroup"] <= 8 * 256, tag      # <= 8 waves per SIMD
    assert pl["kernel_ray_pool"] == pl["ray_pool"], tag
    if pl["walk"] == 2:
        assert pl["leaf_slots"] >= 2 and pl["lds_leaf_stack"], tag                # walk_flat: limit = stack + 64 * (slots - 2)
    if pl["walk"] == 3:
        assert pl["scene_mode"] == 0 and pl["lds_leaf_stack"], tag
    assert pl["kernel_counting"] == (1 if stats else 0), tag
    assert pl["kernel_walk"] == (pl["walk"] if pl["specialised"] else 0), tag
    if pl["kernel_waves_per_simd"] >= 5:
        # the grid is sized for no more waves per SIMD than the kernel's launch bound provides registers for
        assert pl["workgroups_per_cu"] * pl["threads_per_workgroup"] <= pl["kernel_waves_per_simd"] * 256 or pl["threads_per_workgroup"] == 768, tag
    assert pl["workspace_bytes"] > 0 and pl["chunk_spp"] in (1, 2, 4, 8, 16, 32, 64, 128, 256), tag


def test_streamed_launch_plan_invariants_for_every_scene_size_and_knob(trt):
    """Round 2's GPU suite dumped core ONCE under TRT_STREAM_MINW=8 and nothing kept the reason.  Part of the
    audit: every (scene size, trt_tuning) combination a caller can ask for yields a plan whose LDS parts add up, whose stack is
    deep enough for its walk, and for which a kernel instantiation of that exact shape exists - checked here for all of them,
    without a GPU."""
    import itertools
    scenes = []
    scenes.append(("cornell", trt.scenes.cornell(64, 64)))
    scenes.append(("random_spheres", trt.scenes.random_spheres(64, 48)))
    for n in (1, 2, 3, 31, 32, 33, 60, 120, 200, 330, 520, 800, 1500):           # LDS-resident sizes on both sides of every threshold
        scenes.append((f"grid{n}", trt.scenes.sphere_grid(n, 64, 48)))
    scenes.append(("grid4000", trt.scenes.sphere_grid(4000, 64, 48)))            # read from global memory
    knobs = {
        "stream_waves_per_simd": (None, 4, 5, 6, 7, 8, 9),
        "leaf_slots": (None, 1, 2, 3, 4, 8, 16, 64),
        "lds_leaf_stack": (None, 0, 2),
        "ray_pool": (None, 0),
        "stream_big_threads": (None, 512, 768),
        "runtime_walk": (None, 1),
        "dual_walk": (None, 1),
    }
    n_checked = 0
    modes = set()
    for name, desc in scenes:
        for flat in ((None, 0) if name in ("cornell", "grid31") else (None,)):
            w, cam = trt.world_from_description(desc, **({} if flat is None else {"flat_walk": flat}))
            scene = w.get_bvh()
            for combo in itertools.product(*knobs.values()):
                if combo[-1] == 1 and name != "grid4000":
                    continue                                                      # two paths per lane exist for scenes in global memory only
                tn = {k: v for k, v in zip(knobs, combo) if v is not None}
                for stats in (0, 1, 2):
                    pl = _plan(trt, scene, cam, tuning=tn, collect_stats=stats)
                    _check_plan(pl, stats, (name, flat, combo, stats, pl))
                    modes.add((pl["scene_mode"], pl["walk"], pl["threads_per_workgroup"], pl["ray_pool"], pl["dual_walk"]))
                    n_checked += 1
    assert n_checked > 20000 and len(modes) >= 8, (n_checked, sorted(modes))


def test_default_plans_of_the_three_baseline_scenes(trt):
    w, cam = trt.world_from_description(trt.scenes.cornell(64, 64))
    pl = _plan(trt, w.get_bvh(), cam)
    assert (pl["walk"], pl["threads_per_workgroup"], pl["waves_per_simd"], pl["leaf_slots"], pl["ray_pool"], pl["specialised"]) == (2, 256, 6, 7, 1, 1)
    w, cam = trt.world_from_description(trt.scenes.random_spheres(64, 48))
    pl = _plan(trt, w.get_bvh(), cam)
    assert (pl["walk"], pl["threads_per_workgroup"], pl["workgroups_per_cu"], pl["ray_pool"], pl["specialised"]) == (1, 768, 2, 0, 1)
    w, cam = trt.world_from_description(trt.scenes.sphere_grid(4000, 64, 48))
    pl = _plan(trt, w.get_bvh(), cam)
    assert (pl["scene_mode"], pl["walk"], pl["waves_per_simd"], pl["ray_pool"], pl["specialised"]) == (0, 3, 7, 1, 1)      # fits L2: 7 waves, no spills
    pl = _plan(trt, w.get_bvh(), cam, tuning={"stream_waves_per_simd": 8})
    assert (pl["walk"], pl["waves_per_simd"], pl["ray_pool"], pl["specialised"]) == (3, 8, 1, 1)


def test_bulk_sphere_add_is_the_loop_of_single_adds(trt, orc):
    """trt_world_add_spheres (ABI 4) = `for i in 0..n { world.add_geometry(Sphere::new(..)) }` (world.rs:23-25) in array order: the compiled scene is
    the same byte for byte as after n single adds, in the product and in the oracle; an index out of range adds nothing (all or nothing)."""
    desc = trt.scenes.sphere_field(3000, 32, 24, palette=16)
    cr, mat_pos = desc["bulk_spheres"]
    names = [m[0] for m in desc["materials"]]
    listed = dict(desc, bulk_spheres=None,
                  geometries=list(desc["geometries"]) + [("sphere", tuple(float(v) for v in cr[i, :3]), float(cr[i, 3]), names[mat_pos[i]]) for i in range(len(cr))])
    wb, _ = trt.world_from_description(desc)
    wl, _ = trt.world_from_description(listed)
    assert wb.num_geometries() == wl.num_geometries() == 3001
    for a, b in zip(wb.get_bvh().nodes() + wb.get_bvh().cull_nodes() + wb.get_bvh().compact_nodes(), wl.get_bvh().nodes() + wl.get_bvh().cull_nodes() + wl.get_bvh().compact_nodes()):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
    ob, _ = orc.world_from_description(desc)
    ol, _ = orc.world_from_description(listed)
    for a, b in zip(ob.bvh_dump(), ol.bvh_dump()):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    # product tree == oracle tree on the bulk scene too
    assert np.array_equal(wb.get_bvh().nodes()[0].view(np.uint32), ob.bvh_dump()[0].view(np.uint32))
    before = wb.num_geometries()
    bad = np.array([0, 1, 10 ** 6], np.uint32)                               # the third index does not exist
    with pytest.raises(trt.TinyRTError):
        wb.add_spheres(cr[:3], bad)
    assert wb.num_geometries() == before
    wb.add_spheres(np.zeros((0, 4), np.float32), np.zeros(0, np.uint32))       # empty: fine
    assert wb.num_geometries() == before
    # the generator is a pure function of (n, seed)
    again = trt.scenes.sphere_field(3000, 32, 24, palette=16)
    assert np.array_equal(again["bulk_spheres"][0], cr) and np.array_equal(again["bulk_spheres"][1], mat_pos)
    assert not np.array_equal(trt.scenes.sphere_field(3000, 32, 24, seed=45, palette=16)["bulk_spheres"][0], cr)


@pytest.mark.parametrize("height", [1, 15, 16, 17, 31, 32, 33, 250, 500, 1080, 2048, 2160])
@pytest.mark.parametrize("ndev", [1, 2, 3, 8, 13])
def test_band_copy_plan_places_every_row_exactly_once(trt, height, ndev):
    """trt_render_multi's gather: one strided 2-D copy per shard (+ one 1-D copy for the ragged last band).  The pitch arithmetic
    is replayed here on host arrays for ragged heights and more shards than bands, against tiles.band_layout."""
    import ctypes as C
    import importlib
    tiles = importlib.import_module("tiny-raytracer_amd.tiles")
    width = 5
    row_bytes = width * 12
    frame = np.full(height * row_bytes, 0xFF, np.uint8)
    written = np.zeros(height, np.int32)
    for rank in range(ndev):
        pl = trt._lib.BandCopy()
        trt._lib.check(trt.lib.trt_band_copy_plan(width, height, ndev, rank, C.byref(pl)))
        lay = tiles.band_layout(height, ndev, rank)
        rows_local = C.c_uint32()
        trt._lib.check(trt.lib.trt_band_rows_local(height, ndev, rank, C.byref(rows_local)))
        assert pl.rows_local == lay["rows_local"] == rows_local.value
        assert pl.full_bands * 16 + pl.tail_rows == pl.rows_local and pl.tail_rows < 16
        assert pl.band_bytes == 16 * row_bytes == pl.local_pitch and pl.frame_pitch == ndev * pl.band_bytes
        local = np.zeros(pl.rows_local * row_bytes, np.uint8)
        for r, y in enumerate(lay["rows"]):                       # local row r holds image row y: tag every byte with (y mod 251)
            local[r * row_bytes:(r + 1) * row_bytes] = y % 251
        for k in range(pl.full_bands):                            # the 2-D copy, row k of it
            src = k * pl.local_pitch
            dst = pl.frame_offset + k * pl.frame_pitch
            assert dst + pl.band_bytes <= frame.size
            frame[dst:dst + pl.band_bytes] = local[src:src + pl.band_bytes]
            written[dst // row_bytes:(dst + pl.band_bytes) // row_bytes] += 1
        if pl.tail_rows:                                          # the 1-D copy
            assert pl.tail_bytes == pl.tail_rows * row_bytes and pl.tail_frame_offset + pl.tail_bytes == frame.size
            frame[pl.tail_frame_offset:pl.tail_frame_offset + pl.tail_bytes] = local[pl.tail_local_offset:pl.tail_local_offset + pl.tail_bytes]
            written[pl.tail_frame_offset // row_bytes:] += 1
    assert np.all(written == 1)
    assert np.array_equal(frame.reshape(height, row_bytes)[:, 0], np.arange(height) % 251)


# ---- round 4 (ABI v3): configuration as data ----
def test_abi3_pod_layouts_match_the_c_compiler(trt, tmp_path):
    """trt_tuning, trt_scene_options, trt_render_params (with its pointer member), trt_stats and trt_launch_plan as ctypes lays them out against
    what gcc makes of include/tinyrt.h: sizes and the offsets that an alignment rule could move."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tinyrt.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(trt_tuning), sizeof(trt_scene_options), offsetof(trt_scene_options, scratch_cap_bytes), sizeof(trt_render_params), '
                   'offsetof(trt_render_params, tuning), sizeof(trt_stats), offsetof(trt_stats, gather_per_band), sizeof(trt_launch_plan), '
                   'offsetof(trt_launch_plan, workspace_bytes)); return 0; }\n')
    exe = str(tmp_path / "sz")
    subprocess.run(["gcc", "-std=c11", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
    got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    L = trt._lib
    want = [C.sizeof(L.Tuning), C.sizeof(L.SceneOptions), L.SceneOptions.scratch_cap_bytes.offset, C.sizeof(L.RenderParams), L.RenderParams.tuning.offset,
            C.sizeof(L.Stats), L.Stats.gather_per_band.offset, C.sizeof(L.LaunchPlan), L.LaunchPlan.workspace_bytes.offset]
    assert got == want, (got, want)
    assert C.sizeof(L.Tuning) == 96 and C.sizeof(L.SceneOptions) == 48


def test_tuning_and_scene_option_defaults(trt):
    """trt_tuning_default / trt_scene_options_default: the built-in values (kernels.h tuning_builtin; no TRT_* variable is set in the test environment),
    unknown fields are refused by the Python mirror, and a render parameter block without a tuning is the NULL pointer."""
    if any(k.startswith("TRT_") and k not in ("TRT_LIB_PATH", "TRT_BENCH_REHEARSAL") for k in os.environ):
        pytest.skip("TRT_* variables are set: the library's defaults were overridden at load")
    t = trt.tuning().as_dict()
    assert (t["stream_batch_spp"], t["radiance_gb"], t["lds_leaf_stack"], t["ray_pool"], t["stragglers"], t["lds_stragglers"]) == (8, 16, 1, 1, 8, 8)
    assert all(t[k] == 0 for k in ("stream_waves_per_simd", "stream_big_threads", "leaf_slots", "dual_walk", "runtime_walk", "xcd_remap",
                                   "mega_waves_per_simd", "mega_threads", "mega_global_waves8", "wf_waves_per_simd", "wf_serve_min"))
    o = trt.scene_options()
    assert abs(o.cull_prune - 0.5) < 1e-7 and (o.flat_walk, o.compact_nodes, o.top_nodes, o.scratch_cap_bytes) == (-1, -1, 0, 32 << 30)
    with pytest.raises(TypeError):
        trt.tuning(no_such_knob=1)
    with pytest.raises(TypeError):
        trt.scene_options(no_such_option=1)
    r = trt.Renderer(4, 1, 8, False, (0, 0, 0))
    assert not r.params().tuning                                             # NULL: the library's defaults
    p = r.params(tuning={"stragglers": 3})
    assert p.tuning and p.tuning.contents.stragglers == 3 and p.tuning.contents.radiance_gb == 16
    w, _ = trt.world_from_description(trt.scenes.cornell(32, 32))
    with pytest.raises(trt.TinyRTError):
        w.get_bvh(cull_prune=0.0)                                            # must be in (0, 1]


@pytest.mark.parametrize("options", [dict(cull_prune=0.9), dict(cull_prune=0.2), dict(flat_walk=0), dict(compact_nodes=1), dict(top_nodes=63),
                                     dict(cull_prune=0.3, compact_nodes=1, flat_walk=1, top_nodes=31)])
def test_scene_options_are_placement_only(trt, options):
    """Whatever trt_scene_options say, the reference tree is the reference's node for node and the culling tree keeps the reference's leaf
    sequence with the leaves' exact boxes - the two facts the bit-identical hits rest on; only inner nodes and layout move."""
    for desc in (trt.scenes.cornell(32, 32), trt.scenes.random_spheres(32, 24), trt.scenes.sphere_grid(700, 32, 24)):
        base, _ = trt.world_from_description(desc)
        other, _ = trt.world_from_description(desc, **options)
        b0, p0, s0 = base.get_bvh().nodes()
        b1, p1, s1 = other.get_bvh().nodes()
        assert np.array_equal(b0.view(np.uint32), b1.view(np.uint32)) and np.array_equal(p0, p1) and np.array_equal(s0, s1)
        cb0, cp0, _ = base.get_bvh().cull_nodes()
        cb1, cp1, _ = other.get_bvh().cull_nodes()
        assert np.array_equal(cp0[cp0 >= 0], cp1[cp1 >= 0])                                       # same leaves, same order
        assert np.array_equal(cb0[cp0 >= 0].view(np.uint32), cb1[cp1 >= 0].view(np.uint32))      # same leaf boxes
        if options.get("compact_nodes") == 1:
            assert other.get_bvh().compact_nodes() is not None                                    # forced on, whatever the scene's size
