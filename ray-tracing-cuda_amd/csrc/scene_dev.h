// Device-resident scene layout consumed by the trace kernel.
//
// Access pattern decides the layout (DESIGN.md "Data layout in HBM"):
//  * world runs, spheres, parallelograms, triangles are read at a wave-uniform
//    index by every lane at once -> packed, 32/64/128-byte aligned records that
//    one s_load_dwordx8/x16 brings into SGPRs (scalar cache), ray-independent
//    terms (edges, unit normal, r^2) precomputed on the host with the same
//    binary32 operations the reference performs per ray;
//  * BVH nodes / faces are gathered per lane -> 16-byte aligned records read
//    with dwordx4 loads;
//  * per-pixel data (RNG state, radiance) is struct-of-arrays / tile-major so a
//    wave's loads and stores are contiguous;
//  * the material table is small and indexed per lane -> staged in LDS.
#pragma once
#include <stdint.h>

#include "vec.h"
#include "margins.h"

namespace rtmi {

enum RunKind : int32_t { RUN_SKY = 0, RUN_SPHERE = 1, RUN_TRIS = 2, RUN_BVH = 4 };

// A maximal stretch of consecutive world-list entries of one kind, in list
// order (order carries the tie rule of HitableList::Hit, hitable_list.cu:18).
struct Run {
  int32_t kind, first, count, pad;
};

struct SphereRec {  // 32 B
  float cx, cy, cz;
  int32_t mat;
  double radius;
  double r2;  // radius * radius (pow(radius, 2), sphere.cu:16)
};

// Sphere runs of 32 or more spheres are searched through spatial GROUPS (closest_hit.h, RUN_SPHERE): the run's
// spheres sorted along a Morton curve and cut into groups of up to 16.  Both records are read at a wave-uniform
// index (one s_load_dwordx8 each).  What HitableList::Hit returns for a stretch of spheres is the smallest accepted
// t, the first in list order among equal ones, so the order in which they are LOOKED AT is free (see the kernel);
// `orig` keeps the list position for the tie rule.
struct SphGroup {  // 32 B: bounds of the members' own boxes (centre -+ radius), padded; first member, count
  float mn[3], mx[3];
  int32_t first, count;
};
struct SphMember {  // 32 B: all the pre-test and the binary64 test need
  float cx, cy, cz;
  float r2f;       // (float)(radius * radius): operand of the binary32 pre-test
  int32_t orig;    // index of the sphere in SceneDev::spheres (list order)
  int32_t pad;
  double r2;       // radius * radius (pow(radius, 2), sphere.cu:16)
};
constexpr int kSphGroupSize = 16;
constexpr int kSphGroupMin = 32;   // shorter runs are scanned sphere by sphere
constexpr int kSphCand = 16;       // candidate slots per lane in LDS (uint16 each); the scan flushes before a group could overflow them

struct TriRec {  // 48 B: Moller-Trumbore operands that do not depend on the ray
  float p0[3];
  float e1[3];  // p1 - p0
  float e2[3];  // p2 - p0
  float n[3];   // normalize(cross(e1, e2))   (utils.cu:79)
};

// One triangle of the world list, 64 B = one s_load_dwordx16.  A Parallelogram is two
// consecutive records, (p0,p1,p2) then (p1,p2,p3) with TRI_SECOND set: the second is
// tested only by lanes whose first test missed (parallelogram.cu:25,33).
// TRI_SAME_E2 (second records only): e2 has the bit pattern of the first record's e2.
enum : int32_t { TRI_SECOND = 1, TRI_PGRAM = 2, TRI_SAME_E2 = 4 };
struct HotTri {
  float p0[3];
  float e1[3];
  float e2[3];
  float n[3];
  int32_t mat;
  int32_t flags;
  int32_t pad[2];
};

// The same world-list triangles once more, per PAIR (a Parallelogram, or a lone Triangle), for the
// culled form of the list scan (closest_hit.h: RUN_TRIS): `PairBox` is read at a
// wave-uniform index (one s_load_dwordx8): the pair's bounds, padded like the mesh search boxes;
// `PairPts` is staged in LDS and gathered per lane: the four corners, from which the kernel forms the
// edges with the same binary32 subtractions the host used for HotTri.
enum : int32_t { PAIR_SECOND = 1, PAIR_SAME_E2 = 2 };
struct PairBox {  // 32 B
  float mn[3], mx[3];
  int32_t pad[2];
};
struct alignas(16) PairPts {  // 64 B
  float p0[3], p1[3], p2[3], p3[3];  // p3 = p1 + p2 - p0 (parallelogram.cu:13); unused for a lone Triangle
  int32_t flags;
  int32_t pad[3];
};
// What shading needs of the winning world-list triangle, staged in LDS next to the pairs (one record per
// HotTri, same index): the unit normal and the material (low 24 bits) | HotTri::flags << 24.  A per-lane read
// of 16 bytes from LDS instead of a 64-byte gather from global memory on every hit.
struct alignas(16) TriNrm {
  float n[3];
  int32_t mat_flags;
};
constexpr int kCullMinPairs = 4;  // shorter lists are scanned plainly: the cull and the hand-over through LDS cost more than they save
// (ray, pair) / (ray, group) / (ray, sphere) tasks of the shared candidate tests that one wave holds in LDS at a time.
// The capacity decides how many workgroups a CU's LDS holds, so it is per variant: the plain list kernel (six waves per
// SIMD, 25 KiB per workgroup at depth 50) keeps 192 (96: C2 +2.5 % time, more flush rounds); with image textures a task
// carries six result words and 192 of them held the variant at four workgroups per CU -- with 96 a fifth fits (C5 shard
// 3.16 -> 2.93 s together with the 96-VGPR build); the grouped sphere scan: 128 (spheres 1024^2: 22.2 -> 20.9 ms).
constexpr int kListTasks(uint32_t features) { return (features & 16u) ? 96 : (features & 4u) ? 128 : 192; }  // (16 = F_TEX, 4 = F_SGROUP)
// words of one wave's region for shared candidate tests: 64 ray records of 8 words, the tasks, and per task
// the results -- t of a pair's two triangles (+ their u, v with image textures), or a sphere's v (binary64) and index
constexpr int kListWaveWords(uint32_t features) {
  return 64 * 8 + kListTasks(features) * (1 + ((features & 16u) ? 6 : (features & 1u) ? 3 : 2));  // (16 = F_TEX, 1 = F_SPHERE)
}
constexpr int kLdsPairs = 128;  // at most this many PairPts records are staged in LDS (8 KiB); longer lists use the plain scan

enum MatKind : int32_t { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_LIGHT = 3, MAT_SKY = 4 };

struct alignas(16) MatRec {  // 32 B; colour + kind are one 16-byte LDS read
  float r, g, b;   // constant albedo / attenuation / emission
  int32_t kind;
  float param;     // Metal: fuzz (already min(fuzz,1)); Dielectric: (float)refractive_index
  int32_t tex;     // >= 0: image texture supplies the colour instead of r,g,b
  int32_t pad[2];
};

struct TexRec {  // 32 B
  const uint8_t *rgba;  // pitched RGBA8 in device memory
  int32_t height, width;
  int64_t pitch;
  int64_t pad;
};

// The reference's tree (bvh.cuh:113-121): exact bounds, median split of the
// positions_[0].x-sorted faces, leaves of up to kMin faces; inner: left/right = child node
// indices; leaf: left = -1, right = -(face count).  The kernel never descends it to find
// faces; it only replays the box tests on the root-to-leaf paths of the leaves that hold a hit.
struct BvhNode {  // 32 B
  float mn[3];
  float mx[3];
  int32_t left;
  int32_t right;  // negative marks a leaf
};

// The search structure: one 4-wide tree per mesh over ALL its faces (binned-SAH binary tree
// collapsed to four children per node).  A node is 64 bytes = one 64-byte line per lane and step
// (measured: a per-lane gather of 64-byte records runs 2.9x the rate of 128-byte ones, see
// profiles/r02a_gather_microbench.txt): child boxes are 8-bit grid coordinates relative to the
// node's own corner, grid step 2^exp per axis, rounded OUTWARD from bounds that are themselves
// padded (scene.hip: padded_bounds), so the decoded box always contains the padded one.
//   word 0: origin.xyz, exp.xyz (int8) | word 1: qlo x,y,z (one byte per child), qhi x |
//   word 2: qhi y, qhi z, spare       | word 3: child[4]
// child[i] >= 0: node index; child[i] < 0: faces, encoded -(first*8 + count) - 1 with count in
// 1..4; unused slots are -1 (count 0).
struct alignas(16) QNode4 {  // 64 B
  float origin[3];
  int8_t exp[3];
  int8_t slack_exp;  // the children's boxes are widened by 2^slack_exp times the distance slack (scene.hip: face_slack_exponent)
  uint8_t qlo[3][4];
  uint8_t qhi[3][4];
  uint32_t pad1[2];
  int32_t child[4];
};
constexpr int kSlackExpMax = 24;  // 2^24 x 2^-16 of the ray's distance: every box is touched
constexpr int kSubDepthMax = 80;   // deepest search tree (levels of QNode4) the wave-wide stack can serve
constexpr int kMeshFaceSlack = 68;  // face-block entries that can wait on the stack while nodes are popped one at a time
constexpr int kHitSlots = 4;    // per-ray candidate list: one (code, face, t) entry per leaf holding a hit
constexpr int kHitWords = 3;    // words per entry: code, face, t (binary32: TriangleHit's t, utils.cu:53)
// Per-wave LDS of the mesh search (render_body, mesh variants): 64 ray records + one two-ended stack.
// Ray record (words): [0..3] origin, t_to | [4..7] direction, (high word of a binary64 t_to) |
// [8..11] safe 1/direction, padded far bound | [12..15] hit count, cut, lo_code, claim |
// [16..27] kHitSlots x (code, face, t).
constexpr int kMeshRayWords = 16 + kHitSlots * kHitWords;
constexpr int kMeshStackWords = 512;  // node entries grow up from 0, face-block entries down from the top
                                      // (256 words: C3 +27 % time.  A 768-word build faulted in round 4: the replay's
                                      // 64-word `leaves` array was sized by this constant, closest_hit.h -- now capped
                                      // there; a larger stack also costs the third workgroup per CU its LDS)
constexpr int kMeshWaveWords = 64 * kMeshRayWords + kMeshStackWords;
constexpr uint32_t kCodeNone = 0xffffffffu;  // "no cut": no leaf code has bit 0 set (kRefDepthMax = 31)
constexpr int kSparseStride = 16;  // outlier tiles of mesh frames: one pixel per this many lanes (power of two)
constexpr int kTopEntries = 64;   // entries of a mesh's top table: one per lane
constexpr int kTopRays = 16;      // a search that starts with at most this many rays starts from the top table
constexpr int kLeafPathMax = 1 << 26;  // words of SceneDev::leaf_paths (leaves x deepest path, summed over the meshes)
constexpr int kLdsPaths = 1024;        // ... of which this many are staged in LDS (4 KiB)
constexpr int kRefDepthMax = 31;  // decisions below the root that a leaf's path code can hold
constexpr int kMeshMaxFaces = 1 << 23;   // a stack entry is 6 bits of ray + 26 bits of (first face * 8 + count) ...
constexpr int kMeshMaxNodes = 1 << 26;   // ... or of node index

struct BvhRec {  // one per BVH hitable
  int32_t root;      // reference-tree node index
  int32_t mat;       // -1: keep "material_ptr_ == nullptr"
  int32_t has_uv;
  int32_t face_base;  // first face of this mesh: face_uv row = face_base + FaceRec::orig
  int32_t sub_root;   // root of the mesh's 4-wide search tree
  float mag;          // largest |coordinate| of the mesh's bounds (scales the search's distance slack)
  int32_t ref_depth;  // decisions on the longest root-to-leaf path of the reference tree (0: the root is a leaf)
  int32_t path_base;  // first word of this mesh's rows in SceneDev::leaf_paths
  float root_mn[3], root_mx[3];  // the reference root's bounds (nodes[root]): the record is one s_load_dwordx16, and the
  int32_t n_faces;               // mesh-bounds pre-test of a query waits for one scalar load instead of two in a row
  int32_t slack_exp;             // largest slack exponent of the mesh's faces (the pre-test's share of it)
};
static_assert(sizeof(BvhRec) == 64, "BvhRec is read with one s_load_dwordx16");

struct alignas(16) FaceRec {  // 48 B; the unit normal is recomputed for the winner only
  float p0[3];
  float e1[3];
  float e2[3];
  int32_t orig;   // index in the reference's (sorted) face order: decides ties, addresses face_uv
  uint32_t code;  // the reference leaf holding the face: its left(0)/right(1) decisions below the
                  // root, first decision in bit 31, zero-filled.  Leaves are never prefixes of each
                  // other, so codes identify leaves and increase in the reference's visiting order.
  int32_t leaf;   // ordinal of that leaf in visiting order: its row in the mesh's leaf-path table
};

struct CameraDev {
  V3 position, llc, horizontal, vertical, u, v;
  float lens_radius;
  int32_t defocus;
};

// The wave-priority table (render_body.h: wave_priority_update): one row of 16 words per SIMD, row = XCC_ID[3:0] << 10 |
// HW_ID[15:8] (SE, SH, CU) << 2 | HW_ID[5:4] (SIMD).
constexpr int kPrioRows = 1 << 14;
constexpr size_t kPrioTabBytes = (size_t)kPrioRows * 16 * sizeof(uint32_t);

constexpr int kLdsMats = 512;   // at most this many material records are staged in LDS
constexpr int kLdsNodes = 512;  // ... and this many reference-tree nodes (16 KiB)

struct SceneDev {
  const Run *runs;
  const SphereRec *spheres;
  const HotTri *tris;  // one inert record of padding follows the last (prefetch target)
  const PairBox *pair_boxes;  // per pair of `tris` records (index = tri index / 2), one inert record of padding
  const SphGroup *sph_groups;  // per grouped sphere run: Run::pad = its first group, (count + 15) / 16 groups
  const SphMember *sph_members;
  float sph_mag;               // largest |coordinate| of the grouped spheres' bounds (scales the distance slack)
  int32_t n_sph_groups;
  const PairPts *pair_pts;
  const TriNrm *tri_nrm;      // per `tris` record (2 * n_pairs), staged in LDS with pair_pts
  const BvhRec *bvhs;
  const BvhNode *nodes;
  const QNode4 *qnodes;
  const BvhNode *tops;        // per BVH record: kTopEntries sub-trees of its search tree (mn, mx, left = child reference)
  const int32_t *leaf_paths;  // per mesh (BvhRec::path_base): one row of ref_depth node indices per reference leaf,
                              // the nodes below the root on the way to the leaf, -1 past it
  int32_t n_leaf_paths;       // words in leaf_paths
  const FaceRec *faces;
  const float *face_uv;  // 6 floats per face or nullptr
#ifdef RTMI_CHECK_MARGINS
  const int32_t *face_of_orig;  // (diagnostic build) per mesh, face_base + reference index -> index into `faces`
#endif
  const MatRec *mats;
  const TexRec *texs;
  int32_t n_runs, n_mats, n_nodes;
  int32_t n_pairs;    // pairs in pair_boxes / pair_pts (0: the culled list scan is off)
  float list_mag;     // largest |coordinate| of the world-list triangles (scales the cull's distance slack)
  int32_t sub_reserve;  // 3 * (deepest search tree) + 3 + kMeshFaceSlack: stack words the wave-wide search keeps free
                        // after a wide step (0 without meshes); see mesh_search
  int32_t det_safe;   // 1: |e1| * |e2| <= 2^120 for every world-list triangle, so that the Moller-Trumbore determinant
                      // e1 . (d x e2) of a unit (or NaN) direction stays below rcp_rn's limit 2^126 and the list scan
                      // needs no run-time check in front of its reciprocal (tri_test_flat)
  int32_t unsigned_colours;  // 1: no material colour has its sign bit set (not even -0): then every layer
                             // product is +0, positive or NaN and `emitted(0) + product` is the product itself
  CameraDev cam;
};

// Feature bits selecting a kernel specialisation.
enum : uint32_t {
  F_SPHERE = 1u,   // double-precision t
  F_TRIS = 2u,    // parallelograms / boxes / triangles in the world list
  F_SGROUP = 4u,  // some sphere run is long enough for the grouped scan (with F_SPHERE)
  F_BVH = 8u,
  F_TEX = 16u,     // some material reads an image texture (u,v needed)
  F_DEFOCUS = 32u,
  F_ALL = 63u
};

struct FrameDev {
  int32_t height, width, spp, max_depth, post;
  // The samples THIS launch renders: [k_begin, k_end) of a pixel's spp (0 and spp for a frame rendered in one launch).
  // A launch that starts behind sample 0 resumes every pixel from what the earlier launch left in the caller's buffers
  // -- RNG state, raw radiance sum, ray count -- and only the launch that reaches spp post-processes.  A pixel's
  // arithmetic is the same sequence of operations either way (render_body.h: take_item).
  int32_t k_begin, k_end;
  int32_t rank, world;
  int32_t tiles_x, tiles_y, n_tiles;   // global tile grid
  int32_t local_tiles;                 // tiles owned by this rank
  int64_t items;                       // local_tiles * 64
};

}  // namespace rtmi
