"""Scene descriptions used by the tests and the bench: plain data, no numerics.

  cornell()         the reference binary's scene, src/main.rs:23-125 (6 quads + two boxes of 6 quads;
                    the `HittableList` boxes flatten to 18 quads in push order) and camera src/main.rs:8-16
  dummy_spheres()   the 5-sphere worlds of the reference's own tests (renderer/sampler/cpu.rs:96-128,
                    renderer/renderer.rs:87-123)
  quad_test()       hittable/quad.rs:98-150
  random_spheres()  BASELINE config 3: RTIOW-style field of small spheres (build-defined, seed 42)
  sphere_grid()     BASELINE config 5: N spheres scattered on a square + ground sphere (build-defined, seed 43)

A description is {materials: [(name, kind, albedo, param)], geometries: [("sphere", c, r, mat) |
("quad", corner, u, v, mat)], camera: {...}, background: (r,g,b)}.  `build_world` feeds it to any
World-like API (the product's api.World, or the oracle wrapper) so both sides see identical inputs.
"""
import math
import struct

LAMBERTIAN, METAL, DIELECTRIC, LIGHT = 0, 1, 2, 3


def _f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


class _Lcg:
    """Tiny deterministic generator for scene layout only (never used for rendering)."""

    def __init__(self, seed):
        self.s = (seed * 2654435761 + 1013904223) & 0xFFFFFFFF

    def next(self):
        self.s = (self.s * 1664525 + 1013904223) & 0xFFFFFFFF
        x = self.s
        x ^= x >> 16
        x = (x * 0x7FEB352D) & 0xFFFFFFFF
        x ^= x >> 15
        return _f32((x >> 8) / 16777216.0)

    def range(self, lo, hi):
        return _f32(lo + (hi - lo) * self.next())


def _box(a, b, mat):
    """new_box, src/main.rs:89-125: six quads, in the order the reference pushes them."""
    mn = [min(a[i], b[i]) for i in range(3)]
    mx = [max(a[i], b[i]) for i in range(3)]
    dx = (mx[0] - mn[0], 0.0, 0.0)
    dy = (0.0, mx[1] - mn[1], 0.0)
    dz = (0.0, 0.0, mx[2] - mn[2])
    neg = lambda v: (-v[0], -v[1], -v[2])
    return [
        ("quad", (mn[0], mn[1], mx[2]), dx, dy, mat),
        ("quad", (mx[0], mn[1], mx[2]), neg(dz), dy, mat),
        ("quad", (mx[0], mn[1], mn[2]), neg(dx), dy, mat),
        ("quad", (mn[0], mn[1], mn[2]), dz, dy, mat),
        ("quad", (mn[0], mx[1], mx[2]), dx, neg(dz), mat),
        ("quad", (mn[0], mn[1], mn[2]), dx, dz, mat),
    ]


def cornell(width=300, height=300):
    mats = [("red", LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), ("white", LAMBERTIAN, (0.73, 0.73, 0.73), 0.0),
            ("green", LAMBERTIAN, (0.12, 0.45, 0.15), 0.0), ("light", LIGHT, (15.0, 15.0, 15.0), 0.0)]
    geos = [
        ("quad", (100.0, 0.0, 0.0), (0.0, 100.0, 0.0), (0.0, 0.0, 100.0), "green"),
        ("quad", (0.0, 0.0, 0.0), (0.0, 100.0, 0.0), (0.0, 0.0, 100.0), "red"),
        ("quad", (65.0, 100.0, 60.0), (-30.0, 0.0, 0.0), (0.0, 0.0, -20.0), "light"),
        ("quad", (0.0, 0.0, 0.0), (100.0, 0.0, 0.0), (0.0, 0.0, 100.0), "white"),
        ("quad", (100.0, 100.0, 100.0), (-100.0, 0.0, 0.0), (0.0, 0.0, -100.0), "white"),
        ("quad", (0.0, 0.0, 100.0), (100.0, 0.0, 0.0), (0.0, 100.0, 0.0), "white"),
    ]
    geos += _box((25.0, 0.0, 50.0), (55.0, 60.0, 80.0), "white")
    geos += _box((45.0, 0.0, 10.0), (75.0, 30.0, 40.0), "white")
    cam = dict(focus_distance=140.0, defocus_angle=0.6, position=(50.0, 50.0, -140.0), look_at=(50.0, 50.0, 0.0),
               up=(0.0, 1.0, 0.0), vertical_fov=40.0, width=width, height=height)
    return dict(name="cornell", materials=mats, geometries=geos, camera=cam, background=(0.001, 0.001, 0.001))


def dummy_spheres(kind="renderer", width=400, height=300):
    """kind='sampler': cpu.rs:96-128 (all one Lambertian); kind='renderer': renderer.rs:87-123."""
    centers = [((0.0, -100.5, -1.0), 100.0), ((0.0, 0.0, -1.2), 0.5), ((1.0, 0.0, -1.0), 0.5), ((1.0, 0.0, -1.0), 0.4),
               ((-1.0, 0.0, -1.0), 0.5)]
    if kind == "sampler":
        mats = [("dummy", LAMBERTIAN, (1.0, 1.0, 1.0), 0.0)]
        names = ["dummy"] * 5
        bg = (0.0, 0.0, 0.0)
    else:
        mats = [("ground", LAMBERTIAN, (0.0, 1.0, 0.0), 0.0), ("center", LAMBERTIAN, (1.0, 0.0, 0.0), 0.0),
                ("left_outer", DIELECTRIC, (1.0, 1.0, 1.0), 1.5), ("left_inner", DIELECTRIC, (1.0, 1.0, 1.0), _f32(1.0 / 1.5)),
                ("right", METAL, (0.4, 0.4, 1.0), 0.3)]
        names = ["ground", "center", "left_outer", "left_inner", "right"]
        bg = (0.7, 0.8, 1.0)
    geos = [("sphere", c, r, n) for (c, r), n in zip(centers, names)]
    cam = dict(focus_distance=3.4, defocus_angle=10.0, position=(-2.0, 2.0, 1.0), look_at=(0.0, 0.0, -1.0),
               up=(0.0, 1.0, 0.0), vertical_fov=20.0, width=width, height=height)
    return dict(name="dummy_spheres_" + kind, materials=mats, geometries=geos, camera=cam, background=bg)


def quad_test(width=400, height=300):
    mats = [("red", LAMBERTIAN, (1.0, 0.2, 0.2), 0.0), ("green", LAMBERTIAN, (0.2, 1.0, 0.2), 0.0),
            ("blue", LAMBERTIAN, (0.2, 0.2, 1.0), 0.0), ("orange", LAMBERTIAN, (1.0, 0.5, 0.0), 0.0),
            ("teal", LAMBERTIAN, (0.2, 0.8, 0.8), 0.0)]
    geos = [
        ("quad", (-3.0, -2.0, 5.0), (0.0, 0.0, -4.0), (0.0, 4.0, 0.0), "red"),
        ("quad", (-2.0, -2.0, 0.0), (4.0, 0.0, 0.0), (0.0, 4.0, 0.0), "green"),
        ("quad", (3.0, -2.0, 1.0), (0.0, 0.0, 4.0), (0.0, 4.0, 0.0), "blue"),
        ("quad", (-2.0, 3.0, 1.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), "orange"),
        ("quad", (-2.0, -3.0, 5.0), (4.0, 0.0, 0.0), (0.0, 0.0, -4.0), "teal"),
    ]
    cam = dict(focus_distance=1.0, defocus_angle=0.0, position=(0.0, 0.0, 9.0), look_at=(0.0, 0.0, 0.0),
               up=(0.0, 1.0, 0.0), vertical_fov=80.0, width=width, height=height)
    return dict(name="quad_test", materials=mats, geometries=geos, camera=cam, background=(0.7, 0.8, 1.0))


def _sphere_material(rng, i, mats):
    choose = rng.next()
    name = "m%d" % i
    if choose < 0.8:
        alb = tuple(_f32(rng.next() * rng.next()) for _ in range(3))
        mats.append((name, LAMBERTIAN, alb, 0.0))
    elif choose < 0.95:
        alb = tuple(rng.range(0.5, 1.0) for _ in range(3))
        mats.append((name, METAL, alb, rng.range(0.0, 0.5)))
    else:
        mats.append((name, DIELECTRIC, (1.0, 1.0, 1.0), 1.5))
    return name


def random_spheres(width=1920, height=1080, seed=42):
    rng = _Lcg(seed)
    mats = [("ground", LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)]
    geos = [("sphere", (0.0, -1000.0, 0.0), 1000.0, "ground")]
    i = 0
    for a in range(-11, 11):
        for b in range(-11, 11):
            c = (_f32(a + 0.9 * rng.next()), 0.2, _f32(b + 0.9 * rng.next()))
            name_rng_state = rng.s
            if math.sqrt((c[0] - 4.0) ** 2 + (c[1] - 0.2) ** 2 + c[2] ** 2) > 0.9:
                geos.append(("sphere", c, 0.2, _sphere_material(rng, i, mats)))
                i += 1
            else:
                rng.s = name_rng_state
    mats += [("glass", DIELECTRIC, (1.0, 1.0, 1.0), 1.5), ("matte", LAMBERTIAN, (0.4, 0.2, 0.1), 0.0),
             ("mirror", METAL, (0.7, 0.6, 0.5), 0.0)]
    geos += [("sphere", (0.0, 1.0, 0.0), 1.0, "glass"), ("sphere", (-4.0, 1.0, 0.0), 1.0, "matte"),
             ("sphere", (4.0, 1.0, 0.0), 1.0, "mirror")]
    cam = dict(focus_distance=10.0, defocus_angle=0.6, position=(13.0, 2.0, 3.0), look_at=(0.0, 0.0, 0.0),
               up=(0.0, 1.0, 0.0), vertical_fov=20.0, width=width, height=height)
    return dict(name="random_spheres", materials=mats, geometries=geos, camera=cam, background=(0.7, 0.8, 1.0))


def sphere_grid(n=100000, width=3840, height=2160, seed=43):
    """n spheres of radius 0.2 uniformly scattered over a sqrt(n) x sqrt(n) square at y = 0.2."""
    rng = _Lcg(seed)
    side = math.sqrt(n)
    half = side / 2.0
    mats = [("ground", LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)]
    geos = [("sphere", (0.0, -1000.0, 0.0), 1000.0, "ground")]
    for i in range(n):
        c = (rng.range(-half, half), 0.2, rng.range(-half, half))
        geos.append(("sphere", c, 0.2, _sphere_material(rng, i, mats)))
    d = _f32(0.12 * side)
    cam = dict(focus_distance=_f32(math.sqrt(3.0) * d), defocus_angle=0.0, position=(d, _f32(0.5 * d), d),
               look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vertical_fov=35.0, width=width, height=height)
    return dict(name="sphere_grid_%d" % n, materials=mats, geometries=geos, camera=cam, background=(0.7, 0.8, 1.0))


def _mix32(x):
    """lowbias32 on a numpy uint32 array (scene layout only)."""
    import numpy as np
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def sphere_field(n=4_000_000, width=3840, height=2160, seed=44, palette=4096):
    """The deep-BVH scene (sphere_grid, BASELINE config 5) at sizes BEYOND the 256 MiB Infinity Cache: n spheres of radius 0.2 uniformly
    scattered over a sqrt(n) x sqrt(n) square at y = 0.2 + the ground sphere, same camera recipe and material mix (80 % Lambertian,
    15 % Metal, 5 % Dielectric) drawn from a palette of `palette` materials.  Build-defined like sphere_grid; generated with numpy from a
    counter-based hash (sphere i's values depend on (seed, i) only), and handed over as ARRAYS: `bulk_spheres` = (float32 [n, 4] centre +
    radius, material NAME index into `materials`) which build_world adds through World.add_spheres - one call instead of n."""
    import numpy as np
    side = math.sqrt(n)
    half = side / 2.0
    rng = _Lcg(seed)
    mats = [("ground", LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)]
    for i in range(palette):
        _sphere_material(rng, i, mats)
    i = np.arange(n, dtype=np.uint32)
    key = np.uint32((seed * 2654435761) & 0xFFFFFFFF)

    def u01(salt):
        return (_mix32(i * np.uint32(3) + np.uint32(salt) + key) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)

    cr = np.empty((n, 4), np.float32)
    cr[:, 0] = np.float32(-half) + np.float32(2.0 * half) * u01(0)
    cr[:, 1] = np.float32(0.2)
    cr[:, 2] = np.float32(-half) + np.float32(2.0 * half) * u01(1)
    cr[:, 3] = np.float32(0.2)
    mat = (np.uint32(1) + _mix32(i * np.uint32(3) + np.uint32(2) + key) % np.uint32(palette)).astype(np.uint32)      # index into mats (0 = ground)
    d = _f32(0.12 * side)
    cam = dict(focus_distance=_f32(math.sqrt(3.0) * d), defocus_angle=0.0, position=(d, _f32(0.5 * d), d),
               look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vertical_fov=35.0, width=width, height=height)
    return dict(name="sphere_field_%d" % n, materials=mats, geometries=[("sphere", (0.0, -1000.0, 0.0), 1000.0, "ground")],
                bulk_spheres=(cr, mat), camera=cam, background=(0.7, 0.8, 1.0))


def build_world(desc, world, material_ctor, sphere_ctor, quad_ctor):
    """Feed a description to a World-like object.  material_ctor(kind, albedo, param) -> material value;
    sphere_ctor/quad_ctor build geometry values from (…, material handle).  `bulk_spheres` (sphere_field) follow the listed geometries,
    in array order, through world.add_spheres."""
    _build_world_listed(desc, world, material_ctor, sphere_ctor, quad_ctor)
    if desc.get("bulk_spheres") is not None:
        import numpy as np
        cr, mat_pos = desc["bulk_spheres"]
        handle_of = np.array([world.get_material(m[0]) for m in desc["materials"]], dtype=np.uint32)      # position in `materials` -> world handle
        world.add_spheres(cr, handle_of[mat_pos])
    return world


def _build_world_listed(desc, world, material_ctor, sphere_ctor, quad_ctor):
    for name, kind, albedo, param in desc["materials"]:
        world.add_material(name, material_ctor(kind, albedo, param))
    handles = {}
    for g in desc["geometries"]:
        mname = g[-1]
        if mname not in handles:
            handles[mname] = world.get_material(mname)
        if g[0] == "sphere":
            world.add_geometry(sphere_ctor(g[1], g[2], handles[mname]))
        else:
            world.add_geometry(quad_ctor(g[1], g[2], g[3], handles[mname]))
    return world
