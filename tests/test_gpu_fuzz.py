"""Randomised scenes (seeded): every backend against the CPU oracle, bit for bit.  Covers what the hand-made scenes do
not: overlapping and nested boxes, mixed spheres and quads in one tree, all four materials on both primitive kinds,
fuzz outside [0,1] (clamped like Metal::new), refraction indices below 1, degenerate primitives, and a scene with a
non-finite coordinate (every ray then takes the reference tree with the reference's compare-and-assign slab test)."""
import numpy as np
import pytest
from test_gpu_parity import STAT_KEYS, assert_bit_equal

pytestmark = pytest.mark.gpu


def random_scene(seed, n_prims, width=48, height=32, degenerate=False, nonfinite=False, p_sphere=0.5):
    rng = np.random.default_rng(seed)
    f = lambda a: tuple(float(np.float32(v)) for v in a)
    mats = []
    for i in range(6):
        kind = int(rng.integers(0, 4))
        albedo = f(rng.uniform(0.1, 1.0, 3)) if kind != 3 else f(rng.uniform(1.0, 8.0, 3))
        param = float(np.float32(rng.uniform(-0.5, 1.5))) if kind == 1 else float(np.float32(rng.choice([1.5, 1.0 / 1.5, 2.4, 0.9])))
        mats.append(("m%d" % i, kind, albedo, param))
    geos = []
    for i in range(n_prims):
        m = "m%d" % int(rng.integers(0, 6))
        c = rng.uniform(-4, 4, 3)
        if rng.random() < p_sphere:
            geos.append(("sphere", f(c), float(np.float32(rng.uniform(0.2, 1.5))), m))
        else:
            geos.append(("quad", f(c), f(rng.uniform(-2, 2, 3)), f(rng.uniform(-2, 2, 3)), m))
    if degenerate:
        geos += [("sphere", (0.5, 0.5, 0.5), 0.0, "m0"), ("sphere", (1.0, -1.0, 0.0), -0.7, "m1"),
                 ("quad", (0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (2.0, 0.0, 0.0), "m2"),        # u parallel to v: n = 0
                 ("quad", (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), "m3"),        # zero-length edge
                 ("sphere", (0.25, 0.25, 0.25), 0.2, "m4"), ("sphere", (0.25, 0.25, 0.25), 0.2, "m5")]   # coincident twins
    if nonfinite:
        geos.append(("sphere", (float("inf"), 0.0, 0.0), 1.0, "m0"))
    cam = dict(focus_distance=9.0, defocus_angle=float(rng.choice([0.0, 2.0])), position=(0.0, 1.0, 9.0), look_at=(0.0, 0.0, 0.0),
               up=(0.0, 1.0, 0.0), vertical_fov=50.0, width=width, height=height)
    return dict(name="fuzz%d" % seed, materials=mats, geometries=geos, camera=cam, background=f(rng.uniform(0.0, 1.0, 3)))


def check(trt, orc, desc, spp=4, depth=12):
    ow, ocam = orc.world_from_description(desc)
    cpu, cst = orc.render(ow, ocam, spp, depth, desc["background"], seed=3, nthreads=8)
    for backend in (0, 1, 3):
        pw, pcam = trt.world_from_description(desc)
        r = trt.Renderer(spp, 1, depth, False, desc["background"], seed=3, backend=backend)
        gpu = r.render(pcam, pw).data
        assert_bit_equal(gpu, cpu, f"{desc['name']} backend {backend}")
        counted = r.render(pcam, pw, collect_stats=1).data
        assert_bit_equal(counted, cpu, f"{desc['name']} backend {backend} counting")
        for k in STAT_KEYS:
            assert r.last_stats[k] == cst[k], (k, backend)


@pytest.mark.parametrize("seed", range(8))
def test_random_mixed_scenes(trt, orc, seed):
    check(trt, orc, random_scene(100 + seed, n_prims=int(3 + 9 * seed)))


def test_single_primitive_scenes(trt, orc):
    for seed in (1, 2):
        check(trt, orc, random_scene(seed, n_prims=1))
    check(trt, orc, random_scene(3, n_prims=2))


def test_degenerate_primitives(trt, orc):
    check(trt, orc, random_scene(7, n_prims=12, degenerate=True))


def test_non_finite_scene_takes_the_exact_path(trt, orc):
    desc = random_scene(9, n_prims=10, nonfinite=True)
    check(trt, orc, desc, spp=2, depth=6)


def test_large_random_scene_from_global_memory(trt, orc):
    desc = random_scene(11, n_prims=2600, width=64, height=40)       # hot part > 64 KB: read through L1/L2
    pw, _ = trt.world_from_description(desc)
    assert pw.get_bvh().info()["lds_bytes"] == 0
    check(trt, orc, desc, spp=2, depth=8)


@pytest.mark.parametrize("seed", range(6))
def test_walk_schedules_agree_on_many_rays(trt, seed):
    """GPU against GPU, ~10^8 rays per scene (the oracle checks above see ~10^5): the postponed-leaf walk with its slots
    in LDS or in registers, the plain one-slot walk and the megakernel run the same primitive tests and give the same
    frame on mixed random scenes (overlapping spheres and quads, all materials): rounding-level hazards of the
    speculative walk would show here as a differing ray count."""
    desc = random_scene(300 + seed, n_prims=int(20 + 60 * seed), width=1280, height=800)
    pw, pcam = trt.world_from_description(desc)
    ref_img = ref_stats = None
    for backend, slots, lds in ((3, 1, 0), (3, 4, 2), (3, 4, 0), (3, 2, 0), (0, 4, 0), (3, 8, 2)):
        r = trt.Renderer(16, 1, 16, False, desc["background"], seed=11, backend=backend)
        r.tuning = {"leaf_slots": slots, "lds_leaf_stack": lds}
        img = r.render(pcam, pw, collect_stats=2)
        st = dict(r.last_stats)
        if ref_img is None:
            ref_img, ref_stats = img.data.copy(), st
            continue
        assert_bit_equal(img.data, ref_img, f"{desc['name']} backend {backend} slots {slots} lds {lds}")
        for k in ("samples", "rays", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades"):
            assert st[k] == ref_stats[k], (backend, slots, lds, k)


@pytest.mark.parametrize("spheres_only", [False, True])
def test_global_memory_walks_agree_on_many_rays(trt, spheres_only):
    """Scenes walked from global memory (hot part > 64 KB), ~10^8 rays: the 16-byte-node walk (f16 boxes, exact leaf boxes
    re-tested), the 32-byte-node walk, the plain one-slot walk and the megakernel - frames, ray counts and primitive-test
    counters."""
    desc = random_scene(500 + int(spheres_only), n_prims=3000, width=1280, height=800, p_sphere=1.0 if spheres_only else 0.5)
    ref_img = ref_stats = None
    for backend, compact, slots, dual in ((3, 0, 1, 0), (3, 1, None, 0), (3, 0, None, 0), (3, 1, 2, 0), (0, 0, None, 0), (3, 1, None, 1), (3, 1, 3, 1)):
        pw, pcam = trt.world_from_description(desc, compact_nodes=compact)         # trt_scene_options: read when the scene is compiled
        assert pw.get_bvh().info()["lds_bytes"] == 0
        assert (pw.get_bvh().compact_nodes() is not None) == (compact == 1)
        r = trt.Renderer(16, 1, 16, False, desc["background"], seed=13, backend=backend)
        r.tuning = dict({"dual_walk": dual}, **({} if slots is None else {"leaf_slots": slots}))
        img = r.render(pcam, pw, collect_stats=2)
        st = dict(r.last_stats)
        plain = r.render(pcam, pw)                                                   # production kernel
        if ref_img is None:
            ref_img, ref_stats = img.data.copy(), st
        tag = f"{desc['name']} backend {backend} compact {compact} slots {slots}"
        assert_bit_equal(img.data, ref_img, tag + " (counting)")
        assert_bit_equal(plain.data, ref_img, tag)
        for k in ("samples", "rays", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades"):
            assert st[k] == ref_stats[k], (tag, k)


def test_shared_reciprocal_division_is_ieee_division(tmp_path):
    """rt_device.h div_shared (Vec3::normalized's three divisions sharing one refined reciprocal) must give the bits of `/`:
    tools/micro/div_exact.hip compares the sequence with IEEE division on ~1e10 random operand pairs (exponents in [-48, 48],
    edge mantissas included) on the device and exits non-zero on any mismatch."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "div_exact")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-DRANGE=48", "-o", exe,
                    os.path.join(root, "tools", "micro", "div_exact.hip")], check=True)
    r = subprocess.run([exe, "512"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr


def test_short_sqrt_is_sqrtf_on_every_float_in_range(tmp_path):
    """rt_device.h sqrt_in_range (hipcc's sqrtf expansion minus the scaling and the class test) against sqrtf on EVERY float in
    [2^-80, 2^81): tools/micro/sqrt_exact.hip, exhaustive, exits non-zero on any mismatch."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sqrt_exact")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-o", exe,
                    os.path.join(root, "tools", "micro", "sqrt_exact.hip")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr


def test_fma_mix_is_convert_then_fma_for_every_f16(tmp_path):
    """rt_path.h box_loop_compact (round 5) lets v_fma_mix_f32 read a node's f16 coordinate straight out of a half of its word: one instruction
    per plane instead of a conversion and an FMA.  tools/micro/fma_mix_exact.hip compares the two forms for EVERY f16 bit pattern (subnormals
    included - flushing them would move the planes of a millimetre-sized scene by more than their boxes' slack) in both halves of a word against
    4096 (1/d, -m) pairs each: 5.4e8 fused operations, exits non-zero on any mismatch."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fma_mix_exact")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-o", exe,
                    os.path.join(root, "tools", "micro", "fma_mix_exact.hip")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr


def test_device_math_equals_the_oracles_on_every_input_the_path_can_produce(tmp_path, orc):
    """trt-math v2 on the device (rt_device.h) against the CPU checker's statement of it, EXHAUSTIVELY over the path's input domain:
    random::<f32>() has 2^23 values, and vec3extend.rs:15-30 turns one each into theta = 2 pi u, phi = acos(1 - 2u), r = cbrt(u) -
    tests/native/math_exhaustive.hip evaluates sin / cos(theta), acos, sin / cos(phi) and cbrt for all of them on the GPU, and the
    composed random_in_unit_sphere / random_unit_vector on 2^22 generator states, and compares every float with liboracle's, bit for bit."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "math_exhaustive")
    odir = os.path.join(root, "oracle")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I", os.path.join(root, "tiny-raytracer_amd", "csrc"),
                    "-I", odir, "-o", exe, os.path.join(root, "tests", "native", "math_exhaustive.hip"), "-L", odir, "-loracle",
                    "-Wl,-rpath," + odir], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("n_prims", [400, 560])
def test_medium_mixed_scenes_in_lds(trt, orc, n_prims):
    """Mixed spheres and quads with an LDS-resident hot part above 20 KB: the regime of the 768-lane workgroups with an LDS leaf
    stack (400 primitives: the copy and the stack fit twice per CU) and of their 512-lane fallback with register slots (560)."""
    desc = random_scene(700 + n_prims, n_prims=n_prims, width=96, height=64)
    pw, _ = trt.world_from_description(desc)
    lds = pw.get_bvh().info()["lds_bytes"]
    assert 20 * 1024 < lds <= 64 * 1024, lds
    check(trt, orc, desc, spp=3, depth=10)


def test_albedo_above_one_takes_the_general_kernels(trt, orc):
    """SceneLayout::lazy_color is off as soon as a scattering material's albedo exceeds 1 (the attenuation could overflow, and
    `color += inf * 0` would be NaN in the reference): such scenes run the kernels that carry the colour; same bits as the oracle,
    and a huge albedo really does produce the reference's NaN / inf pixels."""
    desc = random_scene(41, n_prims=14)
    desc["materials"] = [(n, k, (a[0] * 1.6, a[1], a[2]) if k != 3 else a, p) for (n, k, a, p) in desc["materials"]]
    check(trt, orc, desc, spp=4, depth=12)
    desc["materials"] = [(n, k, (3.0e30, 2.0e30, 1.0) if k == 0 else a, p) for (n, k, a, p) in desc["materials"]]
    ow, ocam = orc.world_from_description(desc)
    cpu, _ = orc.render(ow, ocam, 2, 40, desc["background"], seed=3, nthreads=8)
    pw, pcam = trt.world_from_description(desc)
    gpu = trt.Renderer(2, 1, 40, False, desc["background"], seed=3).render(pcam, pw).data
    assert_bit_equal(gpu, cpu, "overflowing attenuation")
