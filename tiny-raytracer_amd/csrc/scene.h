// scene.h — host-side scene model and the packed device layout.
//
// World mirrors hittable/world.rs:10-45 (insertion-ordered geometry list + name->material map).
// SceneHost is World::get_bvh() (world.rs:43-45 -> bvh.rs:12-22,42-84) compiled for the GPU.
//
// Two node arrays are packed, both in PRE-ORDER with a skip link per node.  The reference always
// descends left first (bvh.rs:96-106), so "next node" is either i+1 (box hit) or skip[i] (box
// missed / subtree done): a fixed-order traversal needs no stack.
//
//  * the REFERENCE tree: the reference's median-split BVH, node for node.  Walking it performs
//    exactly the reference's box tests; it is what the counting kernels walk (so their counters
//    equal the CPU oracle's) and what rays take whose slab arithmetic can produce NaN.
//  * the CULLING tree: another hierarchy over the SAME LEAF SEQUENCE (same leaf boxes, same
//    order), re-clustered by surface area with near-redundant inner nodes dropped.  It yields
//    bit-identical hits: inner boxes are exact unions of their leaves' boxes, and for a finite
//    ray the f32 slab interval of a box contains that of any box inside it (rounding is monotonic),
//    while t_best only shrinks between an ancestor's test and a descendant's.  So "leaf box
//    passes" implies "every ancestor passed" in ANY such hierarchy: the set and order of
//    primitive tests, hence every accepted hit, depend on the leaf sequence alone.  Inner nodes
//    only decide how much work is skipped.
#pragma once

#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tinyrt.h"

namespace trt {

struct F4 { float x, y, z, w; };            // 16-byte plane element (float4 on the device)

enum : uint32_t { PRIM_NONE = 0xFFFFFFFFu, PRIM_QUAD_BIT = 0x40000000u, PRIM_INDEX_MASK = 0x3FFFFFFFu,
                  NODE_INNER_BIT = 0x80000000u };         // node link: inner -> NODE_INNER_BIT | first child; leaf -> primitive ref

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

// Packed scene: one blob of 16-byte elements.  The first `hot_bytes` are everything a render
// touches per ray and are what small scenes copy into LDS; the reference tree follows.
//   [culling nodes: 2m]  node i = elements 2i, 2i+1 (32 contiguous bytes = one sector per visit)
//               2i:   (min.x, min.y, min.z, max.x)
//               2i+1: (max.y, max.z, bits(skip), bits(link))    link: NODE_INNER_BIT | first child, or kind bit | primitive index
//               pre-order
//   [sphere: ns] (center.xyz, radius)
//   [quad: 5 nq] one record of five consecutive elements per quad (every test and every shade reads all of them: one address,
//               the elements at immediate offsets, 80 contiguous bytes when the scene is read from global memory)
//               +0 (n.xyz, d)                 Quad::hit stage 1   (quad.rs:34-37)
//               +1 (corner.xyz, bits(material))
//               +2 (v.xyz, w.x)               stage 2             (quad.rs:38-41)
//               +3 (w.y, w.z, u.x, u.y)
//               +4 (u.z, n_unit.xyz)          HitRecord normal    (hittable/mod.rs:35-40)
//   [material: nm] (albedo.rgb, param)
//   [sphere_material: ns] u32   [material_kind: nm] u32          (padded to 16 bytes)   <- hot_bytes end here
//   [reference nodes: 2n] same node format
//   [leaf list: 2 per leaf (+ kLeafListPad copies of the last)] the leaves of the trees in walk order, same node format (skip = successor)
//   [compact culling tree: 1 per node] scenes too large for LDS only: (f16 lo.xy | lo.z,hi.x | hi.yz | skip or LEAF|k),
//       boxes rounded OUTWARD to f16, pre-order (an inner node's first child is the next node)
struct SceneLayout {
    uint32_t n_nodes;           // reference tree (2N-1)
    uint32_t n_cull_nodes;      // culling tree
    uint32_t reserved0;         // (rounds 1-4: n_top_nodes)
    uint32_t n_spheres, n_quads, n_materials;
    uint32_t off_sphere, off_quad, off_material;                  // in 16-byte elements (culling nodes start at 0)
    uint32_t off_sphere_mat, off_material_kind;                   // in 4-byte elements from blob start
    uint32_t off_ref_nodes;                                       // in 16-byte elements
    uint32_t hot_bytes, blob_bytes;
    uint32_t all_finite;        // 1: every coordinate is finite and small enough for the fast slab test
    uint32_t n_leaves;          // leaves of either tree (= primitives), in walk order
    uint32_t off_leaf_list;     // in 16-byte elements: the leaves alone, node format, skip = successor (cold part of the blob)
    uint32_t flat_walk;         // 1: few enough leaves that the streamed kernel steps the leaf list in lock-step (rt_path.h walk_flat)
    uint32_t off_compact;       // in 16-byte elements: the culling tree as 16-byte nodes (f16 boxes rounded outward), pre-order; 0 = absent
    float compact_origin_limit[3];   // 16-byte nodes: |ray origin| per axis up to which walk_compact's fused slab arithmetic is conservative (0: never)
    uint32_t lazy_color;        // 1: every scattering material's albedo has |component| <= 1 (so a path's attenuation stays finite and
                                //    `color += attenuation * 0` leaves colour at +0 until the path ends): kernels need not carry the colour
};
// Largest hot blob (SceneLayout::hot_bytes) copied whole into LDS, once per workgroup; larger scenes are read from global memory.
constexpr uint32_t kLdsSceneMaxBytes = 64u * 1024u;
constexpr uint32_t kLeafListPad = 4;          // copies of the last leaf behind the leaf list (walk_flat reads ahead without clamping)
constexpr uint32_t kFlatWalkMaxLeaves = 32;   // at most this many primitives: lock-step leaf list instead of the culling tree

struct NodeDump {                            // pre-order inspection copy of one tree
    std::vector<float> bbox6;
    std::vector<int32_t> prim_geo;           // geometry insertion index or -1
    std::vector<int32_t> skip;
};

struct SceneHost {
    SceneLayout layout;
    std::vector<uint8_t> blob;
    uint32_t max_depth = 0;
    NodeDump reference, culling;
};

// Builds the reference's BVH over `w`, derives the culling tree and packs both.  Returns false (with msg) on an empty world.
// `opt`: trt_scene_options (tinyrt.h) - placement only, every value packs a scene that renders the same frames.
bool compile_scene(const World& w, const trt_scene_options& opt, SceneHost& out, std::string& msg);
// The built-in defaults (cull_prune 0.5, flat_walk / compact_nodes automatic, no top-level cache, 32 GiB idle scratch).
trt_scene_options scene_options_builtin();

// Camera::new (camera.rs:17-56)
void camera_init(trt_camera& out, float focus_distance, float defocus_angle_deg, trt_vec3 position, trt_vec3 look_at,
                 trt_vec3 up, float vertical_fov_deg, uint32_t width, uint32_t height);

// Image finalisation (imager.rs:52-53; utils/image.rs:92-111)
void tonemap_u8(const float* accum, uint32_t npixels, float gamma, uint8_t* rgb);

}  // namespace trt
