"""Checks of the oracle beyond the reference's own KATs: the parts the reference leaves untested
(BVH build/traversal, slab test, scatter functions, distributions) and the two pieces
this project defines (trt-rng v1, trt-math v2)."""
import ctypes as C
import json
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rng_state(orc, seed, pixel, sample):
    st = (C.c_uint32 * 2)()
    orc.lib.orc_rng_seed(seed, pixel, sample, C.byref(st))
    return st


def _ulp_err(fn, ref, xs):
    out = np.array([fn(float(x)) for x in xs], dtype=np.float32).astype(np.float64)
    r = ref(xs.astype(np.float64))
    ulp = np.spacing(np.abs(r).astype(np.float32)).astype(np.float64)
    return (np.abs(out - r) / ulp).max()


def test_trt_math_close_to_libm(orc):
    """trt-math v2 stands in for the platform libm the reference calls (vec3extend.rs:21-27): it has to be an
    accurate sin/cos/acos/cbrt (a couple of ulp) on the ranges the path feeds it, and the cube root - the one function
    with no division left in it - exact on perfect cubes and below one ulp from subnormals up to 2^100."""
    rng = np.random.default_rng(0)
    theta = rng.uniform(0, 2 * math.pi, 40000).astype(np.float32)
    assert _ulp_err(orc.lib.orc_sinf, np.sin, theta) < 2.0
    assert _ulp_err(orc.lib.orc_cosf, np.cos, theta) < 2.0
    assert _ulp_err(orc.lib.orc_acosf, np.arccos, rng.uniform(-1, 1, 40000).astype(np.float32)) < 2.0
    assert _ulp_err(orc.lib.orc_cbrtf, np.cbrt, rng.uniform(0, 1, 40000).astype(np.float32)) < 1.0
    assert _ulp_err(orc.lib.orc_cbrtf, np.cbrt, (np.arange(1, 20000) / 2.0 ** 23).astype(np.float32)) < 1.0
    assert orc.lib.orc_cbrtf(0.0) == 0.0 and orc.lib.orc_cbrtf(1.0) == 1.0 and orc.lib.orc_cbrtf(-27.0) == -3.0
    for k in range(1, 200):                                            # perfect cubes (and their 2^-3n scalings) come out exact
        assert orc.lib.orc_cbrtf(float(k ** 3)) == float(k) and orc.lib.orc_cbrtf(-float(k ** 3) / 512.0) == -k / 8.0, k
    wide = np.exp2(rng.uniform(-149, 100, 40000)).astype(np.float32)   # subnormals .. 2^100
    assert _ulp_err(orc.lib.orc_cbrtf, np.cbrt, wide[wide > 0]) < 1.0
    assert _ulp_err(orc.lib.orc_cbrtf, np.cbrt, -wide[wide > 0]) < 1.0
    assert math.isinf(orc.lib.orc_cbrtf(float("inf"))) and math.isnan(orc.lib.orc_cbrtf(float("nan")))
    far = rng.uniform(-8000, 8000, 40000).astype(np.float32)           # whole reduction range: absolute error (zeros of sin lie inside)
    assert np.abs(np.array([orc.lib.orc_sinf(float(x)) for x in far]) - np.sin(far.astype(np.float64))).max() < 2.5e-7
    assert np.abs(np.array([orc.lib.orc_cosf(float(x)) for x in far]) - np.cos(far.astype(np.float64))).max() < 2.5e-7
    assert orc.lib.orc_acosf(1.0) == 0.0 and abs(orc.lib.orc_acosf(-1.0) - math.pi) < 1e-6
    assert math.isnan(orc.lib.orc_acosf(1.5))


def test_trt_rng_uniform_and_streams_distinct(orc):
    n = 20000
    st = _rng_state(orc, 1, 12345, 7)
    u = np.array([orc.lib.orc_rng_random(C.byref(st)) for _ in range(n)])
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.03
    # neighbouring pixels / samples / seeds start unrelated streams
    firsts = set()
    for seed in (1, 2):
        for pixel in range(64):
            for sample in range(16):
                st = _rng_state(orc, seed, pixel, sample)
                firsts.add((st[0], st[1]))
    assert len(firsts) == 2 * 64 * 16
    a = np.array([orc.lib.orc_rng_random(C.byref(_rng_state(orc, 1, p, 0))) for p in range(4000)])
    b = np.array([orc.lib.orc_rng_random(C.byref(_rng_state(orc, 1, p, 1))) for p in range(4000)])
    assert abs(a.mean() - 0.5) < 0.03 and abs(np.corrcoef(a, b)[0, 1]) < 0.06
    r = np.array([orc.lib.orc_rng_random_range(C.byref(st), -1.0, 1.0) for _ in range(4000)])
    assert r.min() >= -1.0 and r.max() < 1.0 and abs(r.mean()) < 0.05


def test_random_distributions(orc):
    """vec3extend.rs:15-53: points uniform in the unit ball / on the sphere / in the unit disk."""
    st = _rng_state(orc, 3, 1, 1)
    ball = np.array([orc.lib.orc_random_in_unit_sphere(C.byref(st)).tolist() for _ in range(6000)])
    r = np.linalg.norm(ball, axis=1)
    assert r.max() <= 1.0 + 1e-6
    assert abs((r ** 3).mean() - 0.5) < 0.02                      # r^3 uniform
    assert np.abs(ball.mean(axis=0)).max() < 0.03
    unit = np.array([orc.lib.orc_random_unit_vector(C.byref(st)).tolist() for _ in range(6000)])
    assert np.abs(np.linalg.norm(unit, axis=1) - 1.0).max() < 1e-6
    assert abs(unit[:, 2].var() - 1 / 3) < 0.02                   # z uniform in [-1,1]
    disk = np.array([orc.lib.orc_random_in_unit_disk(C.byref(st)).tolist() for _ in range(6000)])
    assert (np.linalg.norm(disk[:, :2], axis=1) < 1.0).all() and (disk[:, 2] == 0).all()
    assert abs((disk[:, 0] ** 2 + disk[:, 1] ** 2).mean() - 0.5) < 0.02


def test_reflect_refract(orc):
    v = orc.Vec3(1 / math.sqrt(2), -1 / math.sqrt(2), 0.0)
    n = orc.Vec3(0.0, 1.0, 0.0)
    r = orc.lib.orc_vec3_reflect(v, n)
    assert np.allclose(r.tolist(), [1 / math.sqrt(2), 1 / math.sqrt(2), 0.0], atol=1e-6)
    # Snell: sin(t) = eta * sin(i)
    eta = 1.0 / 1.5
    t = np.array(orc.lib.orc_vec3_refract(v, n, eta).tolist())
    assert abs(np.linalg.norm(t) - 1.0) < 1e-6
    assert abs(t[0] - eta * (1 / math.sqrt(2))) < 1e-6 and t[1] < 0


def test_bvh_equals_bruteforce(orc, trt):
    """The BVH's closest hit is the brute-force closest hit (same primitive, same t) on random rays."""
    rng = np.random.default_rng(5)
    for desc in (trt.scenes.cornell(), trt.scenes.dummy_spheres("renderer"), trt.scenes.random_spheres(64, 36)):
        w, cam = orc.world_from_description(desc)
        n_hit = n_tie = 0
        for _ in range(1500):
            o = orc.Vec3(*(np.array(desc["camera"]["position"]) + rng.normal(0, 1.0, 3)))
            tgt = np.array(desc["camera"]["look_at"]) + rng.normal(0, 20.0 if desc["name"] == "cornell" else 3.0, 3)
            ray = orc.lib.orc_ray_new(o, orc.Vec3(*(tgt - np.array(o.tolist()))))
            a, _ = w.hit(ray)
            b = w.hit_bruteforce(ray)
            assert (a is None) == (b is None)
            if a is not None:
                n_hit += 1
                assert a.t == b.t and a.point.tolist() == b.point.tolist()
                # equal t on two primitives (Cornell's light lies in the ceiling's plane, src/main.rs:42-59): the BVH
                # keeps the first in left-first order (bvh.rs:96-101), brute force the first in insertion order
                n_tie += a.material != b.material
        assert n_hit > 300 and n_tie <= (0.05 * n_hit if desc["name"] == "cornell" else 0)


def test_slab_test_nan_semantics(orc):
    """aabb.rs:36-61 leaves NaN (0 * inf) untouched: a ray lying in a face plane with a zero direction
    component still passes on that axis; `end <= start` rejects an empty interval."""
    box = orc.Aabb(orc.Vec3(0, 0, 0), orc.Vec3(1, 1, 1))
    ray = orc.Ray(orc.Vec3(0.0, 0.5, -1.0), orc.Vec3(0.0, 0.0, 1.0))       # x == box.min.x, d.x == 0 -> NaN on x
    assert orc.lib.orc_aabb_intersect(C.byref(box), C.byref(ray), 0.001, math.inf) == 1
    ray = orc.Ray(orc.Vec3(2.0, 0.5, -1.0), orc.Vec3(0.0, 0.0, 1.0))       # outside on x: -inf..-inf interval
    assert orc.lib.orc_aabb_intersect(C.byref(box), C.byref(ray), 0.001, math.inf) == 0
    ray = orc.Ray(orc.Vec3(0.5, 0.5, -1.0), orc.Vec3(0.0, 0.0, 1.0))
    assert orc.lib.orc_aabb_intersect(C.byref(box), C.byref(ray), 0.001, 1.0) == 0     # t range ends at the near face
    assert orc.lib.orc_aabb_intersect(C.byref(box), C.byref(ray), 0.001, 1.0001) == 1
    nan = float("nan")
    ray = orc.Ray(orc.Vec3(5, 5, 5), orc.Vec3(nan, nan, nan))                 # NaN direction passes every box
    assert orc.lib.orc_aabb_intersect(C.byref(box), C.byref(ray), 0.001, math.inf) == 1


def test_bounce_loop_accounting(orc, trt):
    """cpu.rs:47-62: emission is added before scatter; a Light ends the path; a miss adds the background;
    an exhausted budget adds nothing."""
    desc = dict(materials=[("light", 3, (2.0, 3.0, 4.0), 0.0), ("white", 0, (0.5, 0.5, 0.5), 0.0)],
                geometries=[("quad", (-1.0, -1.0, -2.0), (2.0, 0.0, 0.0), (0.0, 2.0, 0.0), "light"),
                            ("sphere", (0.0, 0.0, 5.0), 1.0, "white")],
                camera=dict(focus_distance=1.0, defocus_angle=0.0, position=(0, 0, 0), look_at=(0, 0, -1), up=(0, 1, 0),
                            vertical_fov=40.0, width=4, height=4))
    w, _ = orc.world_from_description(desc)
    pts = (orc.SamplePoint * 3)()
    pts[0].ray = orc.lib.orc_ray_new(orc.Vec3(0, 0, 0), orc.Vec3(0, 0, -1))      # hits the light
    pts[1].ray = orc.lib.orc_ray_new(orc.Vec3(0, 0, 0), orc.Vec3(0, 1, 0))       # misses everything
    pts[2].ray = orc.lib.orc_ray_new(orc.Vec3(0, 0, 0), orc.Vec3(0, 0, 1))       # hits the diffuse sphere
    out, st = orc.sample_batch(w, pts, 1, (0.25, 0.5, 0.75))
    assert out[0].color.tolist() == [2.0, 3.0, 4.0]
    assert out[1].color.tolist() == [0.25, 0.5, 0.75]
    assert out[2].color.tolist() == [0.0, 0.0, 0.0]                              # budget of 1 spent on the scatter
    assert st["rays"] == 3 and st["shades"] == 2


def test_progressive_and_row_ranges_compose(orc, trt):
    desc = trt.scenes.cornell(40, 30)
    w, cam = orc.world_from_description(desc)
    full, st = orc.render(w, cam, 6, 8, desc["background"], nthreads=4)
    part, _ = orc.render(w, cam, 6, 8, desc["background"], sample_end=2)
    part, _ = orc.render(w, cam, 6, 8, desc["background"], sample_begin=2, accum=part, nthreads=3)
    assert np.array_equal(full, part)
    rows = np.zeros_like(full)
    orc.render(w, cam, 6, 8, desc["background"], row_begin=0, row_end=11, accum=None)
    a, _ = orc.render(w, cam, 6, 8, desc["background"], row_begin=0, row_end=11)
    b, _ = orc.render(w, cam, 6, 8, desc["background"], row_begin=11, row_end=30)
    rows[:11], rows[11:] = a[:11], b[11:]
    assert np.array_equal(full, rows)


def test_oracle_matches_reference_renders_statistically(orc, trt):
    """The reference's RNG is unseeded, so its frames pin the path's expectation, not its samples.  Block means of
    the oracle's frame (same scene, spp/depth/background from the reference source) agree with block means of the
    PNGs the reference ships (tests/golden/reference_png_blocks.json)."""
    with open(os.path.join(ROOT, "tests", "golden", "reference_png_blocks.json")) as f:
        fx = json.load(f)

    def lin_blocks(u8, block):
        lin = ((u8.astype(np.float64) + 0.5) / 255.0) ** 2.2
        h, w, _ = u8.shape
        bh, bw = h // block, w // block
        return lin[: bh * block, : bw * block].reshape(bh, block, bw, block, 3).mean(axis=(1, 3))

    cases = [("cornell", trt.scenes.cornell(300, 300), 48, 20, 0.97), ("quad_test", trt.scenes.quad_test(), 24, 10, 0.999),
             ("render_test", trt.scenes.dummy_spheres("renderer"), 24, 10, 0.999)]
    for key, desc, spp, depth, min_corr in cases:
        f = fx[key]
        ref = np.array(f["mean"])
        ok = ~np.array(f["saturated"]).astype(bool)
        w, cam = orc.world_from_description(desc)
        acc, _ = orc.render(w, cam, spp, depth, desc["background"], nthreads=8, seed=11)
        mine = lin_blocks(orc.tonemap_u8(acc), f["block"])
        assert abs(mine[ok].mean() / ref[ok].mean() - 1.0) < 0.03, key
        assert np.corrcoef(mine[ok].ravel(), ref[ok].ravel())[0, 1] > min_corr, key


def test_own_math_vs_libm_mode_agree_statistically(orc, trt):
    """Swapping trt-math v2 for the platform libm (what the Rust original calls) leaves the frame's statistics alone."""
    desc = trt.scenes.cornell(60, 60)
    w, cam = orc.world_from_description(desc)
    a, _ = orc.render(w, cam, 64, 12, desc["background"], nthreads=8)
    orc.lib.orc_set_use_libm(1)
    try:
        b, _ = orc.render(w, cam, 64, 12, desc["background"], nthreads=8)
    finally:
        orc.lib.orc_set_use_libm(0)
    assert abs(a.mean() / b.mean() - 1.0) < 0.02


def test_powf_v1_is_the_correctly_rounded_power_almost_everywhere(orc):
    """trt-math v2 powf (rt_oracle.c m_powf, f64 inside, one rounding to f32) against numpy's f64 power rounded to f32:
    identical on this sample (the f64 result carries ~1e-14 relative error, so about one input in 10^6 may round the other
    way), and C99's special cases.  The reference's libm powf (image.rs:94-96) is pinned by none of its tests."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(20000).astype(np.float32) * 2, np.exp(rng.uniform(-80, 80, 20000)).astype(np.float32),
                        np.array([1e-45, 1e-40, 1.1754944e-38, 3.4028235e38], np.float32)])
    y = np.concatenate([np.full(20000, np.float32(1.0) / np.float32(2.2), np.float32), rng.uniform(-3, 3, 20000).astype(np.float32),
                        np.array([0.5, 0.45454547, 2.0, 0.25], np.float32)])
    got = np.array([orc.lib.orc_powf(float(a), float(b)) for a, b in zip(x, y)], np.float32)
    with np.errstate(all="ignore"):
        want = np.power(x.astype(np.float64), y.astype(np.float64)).astype(np.float32)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp != 0).mean() < 1e-4
    inf, nan = float("inf"), float("nan")
    f = orc.lib.orc_powf
    assert f(nan, 0.0) == 1.0 and f(1.0, nan) == 1.0 and f(5.0, 0.0) == 1.0 and f(-0.0, -0.0) == 1.0
    assert math.isnan(f(nan, 2.0)) and math.isnan(f(2.0, nan)) and math.isnan(f(-2.0, 0.5))
    assert f(0.0, 0.5) == 0.0 and f(-0.0, 0.5) == 0.0 and f(0.0, -1.0) == inf and f(-0.0, -1.0) == -inf and f(-0.0, -2.0) == inf
    assert f(inf, 0.5) == inf and f(inf, -0.5) == 0.0 and f(-inf, 3.0) == -inf and f(-inf, 2.0) == inf and f(-inf, 0.5) == inf
    assert f(-2.0, 3.0) == -8.0 and f(-2.0, 2.0) == 4.0 and f(2.0, 10.0) == 1024.0 and f(-1.0, inf) == 1.0
    assert f(0.5, inf) == 0.0 and f(0.5, -inf) == inf and f(2.0, inf) == inf and f(2.0, -inf) == 0.0
    assert f(3.0, 1.0) == 3.0 and f(2.0, 200.0) == inf and f(2.0, -200.0) == 0.0 and f(4.0, 0.5) == 2.0
    # libm mode for comparison: at most 1 ulp apart from glibc's powf
    orc.lib.orc_set_use_libm(1)
    try:
        libm = np.array([orc.lib.orc_powf(float(a), float(b)) for a, b in zip(x[:5000], y[:5000])], np.float32)
    finally:
        orc.lib.orc_set_use_libm(0)
    assert np.abs(libm.view(np.int32).astype(np.int64) - got[:5000].view(np.int32).astype(np.int64)).max() <= 1


def test_u8_frame_of_trt_powf_against_platform_libm_powf(orc, trt):
    """host tonemap, device tonemap and oracle share ONE powf (trt-math v2), so their byte-exact agreement compares the
    algorithm with itself; the reference calls the platform's libm powf (utils/image.rs:94-96).  Here the quantised frame of the
    product's tonemap is compared with the oracle's in LIBM mode (glibc powf) on a rendered frame and on 2 M synthetic channel
    values: a channel may differ by one least-significant bit (where the two powf round differently right at a quantisation step)
    and does so for fewer than one value in 10^4 - a regression of either implementation shows up as a larger step or rate."""
    desc = trt.scenes.cornell(96, 96)
    ow, ocam = orc.world_from_description(desc)
    frame, _ = orc.render(ow, ocam, 16, 12, desc["background"], seed=3, nthreads=4)
    rng = np.random.default_rng(11)
    synth = np.concatenate([rng.random(1_000_000), np.exp(rng.uniform(-12, 3, 1_000_000))]).astype(np.float32)
    synth = np.resize(synth, (len(synth) // 3) * 3)
    for values in (frame.reshape(-1), synth):
        img = trt.Image(values.reshape(-1, 1, 3).copy())
        mine = img.to_u8().reshape(-1).astype(np.int32)                 # libtinyrt's host tonemap (trt-math v2 powf), no GPU needed
        orc.lib.orc_set_use_libm(1)
        try:
            libm = orc.tonemap_u8(values.reshape(-1, 3)).reshape(-1).astype(np.int32)
        finally:
            orc.lib.orc_set_use_libm(0)
        own = orc.tonemap_u8(values.reshape(-1, 3)).reshape(-1).astype(np.int32)
        assert np.array_equal(mine, own)                                # same function on both sides: equal
        diff = np.abs(mine - libm)
        assert diff.max() <= 1, int(diff.max())
        assert (diff != 0).mean() < 1e-4, float((diff != 0).mean())
