// event_costs.hip — one kernel per EVENT of the path (cpu.rs:39-65 and callees), each doing that event once per lane on operands it loads
// from memory, plus a baseline kernel with the same loads and stores and no event.  Nothing here runs: tools/isa_event_costs.py compiles
// this file for gfx950 with the library's flags, counts the VALU instructions (v_*) of every kernel in the emitted ISA and subtracts the
// baseline: the table profiles/isa_event_costs.json = VALU lane-instructions ONE lane needs for ONE such event, which bench.py multiplies
// by the event counts of its counting pass to get `roofline.useful_frac`.  Straight-line static counts: a branch's both sides
// are counted where the compiler kept a branch (the rejection loop of the unit disk is counted once per iteration, 4/pi expected).
#define TRT_EVENT_COSTS 1      // rt_device.h: the never-taken plain-IEEE fallbacks stay out of line (not counted)
#include "../../tiny-raytracer_amd/csrc/rt_path.h"

using namespace trt;

struct EvIn { float f[20]; uint32_t u[4]; };
struct EvOut { float f[12]; uint32_t u[4]; };

TRT_DEV Ray load_ray(const EvIn& v) { return Ray{v3(v.f[0], v.f[1], v.f[2]), v3(v.f[3], v.f[4], v.f[5])}; }

#define EV_PROLOGUE                                             \
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x; \
    const EvIn v = in[tid];                                     \
    EvOut o;                                                    \
    for (int k = 0; k < 12; k++) o.f[k] = v.f[k];               \
    for (int k = 0; k < 4; k++) o.u[k] = v.u[k];
#define EV_EPILOGUE out[tid] = o;

extern "C" __global__ void ev_baseline(const EvIn* in, EvOut* out, SceneDev scd) { EV_PROLOGUE EV_EPILOGUE }

// per ray: 1/d per axis and the finiteness test (aabb.rs:42 hoisted; rt_path.h trav_begin)
extern "C" __global__ void ev_ray_setup(const EvIn* in, EvOut* out, SceneDev scd) {
    EV_PROLOGUE
    const SceneAcc<MODE_GLOBAL> sc{scd.blob, scd.L};
    const Ray ray = load_ray(v);
    const Trav tr = trav_begin(sc, ray, false);
    o.f[0] = tr.inv.x; o.f[1] = tr.inv.y; o.f[2] = tr.inv.z; o.u[0] = tr.fast ? 1u : 0u; o.u[1] = tr.n;
    EV_EPILOGUE
}

// one slab test on a 32-byte node already in registers (aabb.rs:36-61 as slab_fast_entry) + the descend / skip select
extern "C" __global__ void ev_box_test(const EvIn* in, EvOut* out, SceneDev scd) {
    EV_PROLOGUE
    const float4 na = make_float4(v.f[6], v.f[7], v.f[8], v.f[9]), nb = make_float4(v.f[10], v.f[11], v.f[12], v.f[13]);
    float start;
    const bool pass = slab_fast_entry(na, nb, v3(v.f[0], v.f[1], v.f[2]), v3(v.f[3], v.f[4], v.f[5]), kTMin, v.f[14], start);
    const uint32_t link = __float_as_uint(nb.w);
    const bool inner = (link & NODE_INNER_BIT) != 0u;
    o.u[0] = (pass && inner) ? (link & ~NODE_INNER_BIT) : __float_as_uint(nb.z);
    o.u[1] = (pass && !inner) ? link : PRIM_NONE;
    o.f[0] = start;
    EV_EPILOGUE
}

// Quad::hit (quad.rs:33-54) as the branch-free trav_leaf evaluates it
extern "C" __global__ void ev_quad_test(const EvIn* in, EvOut* out, SceneDev scd) {
    EV_PROLOGUE
    const SceneAcc<MODE_GLOBAL> sc{scd.blob, scd.L};
    const Ray ray = load_ray(v);
    Trav tr;
    tr.t_best = v.f[14]; tr.prim_best = v.u[1];
    Counters<false> ctr;
    trav_leaf<MODE_GLOBAL, false>(sc, ray, tr, PRIM_QUAD_BIT | (v.u[0] & 0xFFFFu), ctr);
    o.f[0] = tr.t_best; o.u[0] = tr.prim_best;
    EV_EPILOGUE
}

// Sphere::hit (sphere.rs:29-54)
extern "C" __global__ void ev_sphere_test(const EvIn* in, EvOut* out, SceneDev scd) {
    EV_PROLOGUE
    const SceneAcc<MODE_GLOBAL> sc{scd.blob, scd.L};
    const Ray ray = load_ray(v);
    Trav tr;
    tr.t_best = v.f[14]; tr.prim_best = v.u[1];
    Counters<false> ctr;
    trav_leaf<MODE_GLOBAL, false>(sc, ray, tr, v.u[0] & 0xFFFFu, ctr);
    o.f[0] = tr.t_best; o.u[0] = tr.prim_best;
    EV_EPILOGUE
}

// cpu.rs:48-62 after the hit query for ONE material kind: HitRecord::new for the winning primitive, emitted, scatter, attenuation, Ray::new
template <uint32_t KIND, bool QUAD>
TRT_DEV void shade_event(const EvIn& v, EvOut& o, const SceneDev& scd) {
    const SceneAcc<MODE_GLOBAL> sc{scd.blob, scd.L};
    Path p;
    p.ray = load_ray(v);
    p.color = v3(0.0f, 0.0f, 0.0f);
    p.atten = v3(v.f[6], v.f[7], v.f[8]);
    p.remain = v.u[2];
    p.rng = Rng{v.u[0], v.u[1]};
    const uint32_t prim = (QUAD ? PRIM_QUAD_BIT : 0u) | (v.u[3] & 0xFFFFu);
    if (sc.material_kind(prim_material(sc, prim)) != KIND) return;              // the compiler then knows the kind inside shade_hit
    Counters<false> ctr;
    const bool ended = shade_hit<MODE_GLOBAL, false, true>(sc, p, prim, v.f[14], v3(v.f[9], v.f[10], v.f[11]), ctr);
    o.f[0] = p.ray.o.x; o.f[1] = p.ray.o.y; o.f[2] = p.ray.o.z; o.f[3] = p.ray.d.x; o.f[4] = p.ray.d.y; o.f[5] = p.ray.d.z;
    o.f[6] = p.atten.x; o.f[7] = p.atten.y; o.f[8] = p.atten.z; o.f[9] = p.color.x; o.f[10] = p.color.y; o.f[11] = p.color.z;
    o.u[0] = p.rng.s0; o.u[1] = p.rng.s1; o.u[2] = p.remain; o.u[3] = ended ? 1u : 0u;
}
#define EV_SHADE(NAME, KIND, QUAD) \
    extern "C" __global__ void NAME(const EvIn* in, EvOut* out, SceneDev scd) { EV_PROLOGUE shade_event<KIND, QUAD>(v, o, scd); EV_EPILOGUE }
EV_SHADE(ev_shade_lambertian_quad, TRT_LAMBERTIAN, true)
EV_SHADE(ev_shade_light_quad, TRT_LIGHT, true)
EV_SHADE(ev_shade_lambertian_sphere, TRT_LAMBERTIAN, false)
EV_SHADE(ev_shade_metal_sphere, TRT_METAL, false)
EV_SHADE(ev_shade_dielectric_sphere, TRT_DIELECTRIC, false)

// a miss: color += attenuation * background (cpu.rs:58-61)
extern "C" __global__ void ev_shade_miss(const EvIn* in, EvOut* out, SceneDev scd) {
    EV_PROLOGUE
    const SceneAcc<MODE_GLOBAL> sc{scd.blob, scd.L};
    Path p;
    p.ray = load_ray(v);
    p.atten = v3(v.f[6], v.f[7], v.f[8]);
    Counters<false> ctr;
    shade_hit<MODE_GLOBAL, false, true>(sc, p, PRIM_NONE, v.f[14], v3(v.f[9], v.f[10], v.f[11]), ctr);
    o.f[9] = p.color.x; o.f[10] = p.color.y; o.f[11] = p.color.z;
    EV_EPILOGUE
}

// one sample's start: RNG stream + primary ray (pointgen.rs:41-43, camera.rs:58-66)
extern "C" __global__ void ev_primary_ray(const EvIn* in, EvOut* out, SceneDev scd, CameraDev cam, RenderArgs ra) {
    EV_PROLOGUE
    Path p;
    path_begin(p, cam, ra, v.u[0] & 0xFFFu, v.u[1] & 0xFFFu, v.u[2]);
    o.f[0] = p.ray.o.x; o.f[1] = p.ray.o.y; o.f[2] = p.ray.o.z; o.f[3] = p.ray.d.x; o.f[4] = p.ray.d.y; o.f[5] = p.ray.d.z;
    o.u[0] = p.rng.s0; o.u[1] = p.rng.s1;
    EV_EPILOGUE
}
