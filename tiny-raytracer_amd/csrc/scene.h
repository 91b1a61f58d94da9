// scene.h — host-side scene model and the packed device layout.
//
// World mirrors hittable/world.rs:10-45 (insertion-ordered geometry list + name->material map).
// SceneHost is World::get_bvh() (world.rs:43-45 -> bvh.rs:12-22,42-84) compiled for the GPU:
// the reference's median-split tree, one primitive per leaf, laid out in PRE-ORDER with a skip
// link per node.  The reference always descends left first (bvh.rs:96-106), so "next node" is
// either i+1 (box hit) or skip[i] (box missed / subtree done): a fixed-order traversal needs no
// stack at all and visits exactly the reference's node sequence.
#pragma once

#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tinyrt.h"

namespace trt {

struct F4 { float x, y, z, w; };            // 16-byte plane element (float4 on the device)

enum : uint32_t { PRIM_NONE = 0xFFFFFFFFu, PRIM_QUAD_BIT = 0x40000000u, PRIM_INDEX_MASK = 0x3FFFFFFFu };

struct Geometry {
    uint32_t kind;                           // 0 sphere, 1 quad
    uint32_t material;
    trt_vec3 a, b, c;                        // sphere: a=center, b.x=radius; quad: a=corner, b=u, c=v
};

struct World {
    std::vector<Geometry> geometries;                       // world.rs:11 (insertion order matters: bvh.rs:62-63)
    std::vector<trt_material> materials;
    std::unordered_map<std::string, uint32_t> material_index;   // world.rs:12
};

// Packed scene.  One contiguous blob of 16-byte elements followed by two u32 arrays; the same
// offsets address it in HBM and, for small scenes, in its LDS copy.
//   [node: 2n]  node i = elements 2i, 2i+1 (32 contiguous bytes: a traversal step touches one 32-byte sector)
//               2i:   (min.x, min.y, min.z, max.x)
//               2i+1: (max.y, max.z, bits(skip), bits(prim))    prim: PRIM_NONE | kind bit | index
//   [sphere: ns] (center.xyz, radius)
//   [quad plane 0: nq] (n.xyz, d)            Quad::hit stage 1   (quad.rs:34-37)
//   [quad plane 1: nq] (corner.xyz, bits(material))
//   [quad plane 2: nq] (v.xyz, w.x)          stage 2             (quad.rs:38-41)
//   [quad plane 3: nq] (w.y, w.z, u.x, u.y)
//   [quad plane 4: nq] (u.z, n_unit.xyz)     HitRecord normal    (hittable/mod.rs:35-40)
//   [material: nm] (albedo.rgb, param)
//   [sphere_material: ns] u32   [material_kind: nm] u32
struct SceneLayout {
    uint32_t n_nodes, n_spheres, n_quads, n_materials;
    uint32_t off_sphere, off_quad, off_material;                  // in 16-byte elements (nodes start at 0)
    uint32_t off_sphere_mat, off_material_kind;                   // in 4-byte elements from blob start
    uint32_t blob_bytes;
    uint32_t all_finite;        // 1: every coordinate is finite and small enough for the fast slab test
};

struct SceneHost {
    SceneLayout layout;
    std::vector<uint8_t> blob;
    uint32_t max_depth = 0;
    // test/inspection copies (pre-order)
    std::vector<float> bbox6;
    std::vector<int32_t> prim_geo;           // geometry insertion index or -1
    std::vector<int32_t> skip;
};

// Builds the reference's BVH over `w` and packs it.  Returns false (with msg) on an empty world.
bool compile_scene(const World& w, SceneHost& out, std::string& msg);

// Camera::new (camera.rs:17-56)
void camera_init(trt_camera& out, float focus_distance, float defocus_angle_deg, trt_vec3 position, trt_vec3 look_at,
                 trt_vec3 up, float vertical_fov_deg, uint32_t width, uint32_t height);

// Image finalisation (imager.rs:52-53; utils/image.rs:92-111)
void tonemap_u8(const float* accum, uint32_t npixels, float gamma, uint8_t* rgb);

}  // namespace trt
