// Host-side scene recorder: the reference's constructors (Sphere, Parallelogram,
// Parallelepiped, BVH, Lambertian, ... and Camera) as plain records, plus the
// flatten pass that turns the recorded world list into the device layout of
// scene_dev.h.  Constructor arithmetic that the reference performs at scene build
// time (parallelogram 4th corner, parallelepiped corners, camera frame, BVH
// bounds/sort/split) is done here, on the host, with the same binary32 operations.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "scene_dev.h"
#include "vec.h"

namespace rtmi {

struct HostTex {
  bool image = false;
  V3 rgb{0, 0, 0};
  std::vector<uint8_t> rgba;  // tightly packed
  int h = 0, w = 0;
};

struct HostMat {
  int32_t kind = MAT_LAMBERTIAN;
  V3 rgb{0, 0, 0};
  float param = 0.f;
  int tex = -1;  // texture handle (Lambertian(Texture*) / DiffuseLight(Texture*))
};

enum ObjKind { OBJ_SPHERE, OBJ_TRI, OBJ_PGRAM, OBJ_BOX, OBJ_SKY, OBJ_BVH };

struct HostObj {
  ObjKind kind;
  int mat = -1;
  V3 p[18];           // sphere: p[0]=centre; tri/pgram: p[0..2]; box: six faces of three points
  double radius = 0;  // sphere
  int bvh = -1;       // index into Scene::bvhs
};

struct HostBvh {
  std::vector<float> faces;  // n*9
  std::vector<float> uvs;    // n*6 or empty
  int n = 0, mat = -1, leaf_max = 2048;
};

struct Scene {
  std::vector<HostTex> texs;
  std::vector<HostMat> mats;
  std::vector<HostObj> world;  // HitableList order, nested lists already inlined at their position (scene.hip: flatten)
  std::vector<int> list_counts{0};  // entries of the world list [0] and of every nested HitableList still open
  std::vector<HostBvh> bvhs;
  CameraDev cam{};
  V3 cam_w{0, 0, 0};
  bool has_camera = false;

  // ---- flattened (valid after commit)
  bool committed = false;
  uint32_t features = 0;
  std::vector<Run> runs;
  std::vector<SphereRec> spheres;
  std::vector<HotTri> tris;
  std::vector<PairBox> pair_boxes;
  std::vector<PairPts> pair_pts;
  std::vector<TriNrm> tri_nrm;
  std::vector<SphGroup> sph_groups;
  std::vector<SphMember> sph_members;
  float sph_mag = 0.f;
  int64_t sliver_faces = 0;  // mesh faces thinner than 1.8 degrees: their nodes carry a slack exponent (rtmi_scene_sliver_faces)
  float list_mag = 0.f;
  int n_pgrams = 0, n_triangles = 0, n_spheres = 0;
  std::vector<BvhRec> bvh_recs;
  std::vector<BvhNode> nodes;
  std::vector<QNode4> qnodes;
  std::vector<BvhNode> tops;        // per BVH record: kTopEntries sub-trees of its search tree (box, child reference)
  std::vector<int32_t> leaf_paths;  // per mesh: rows of ref_depth reference-node indices, one row per reference leaf
  int sub_depth = 0;  // deepest search tree (levels of QNode4)
  std::vector<FaceRec> faces;
  std::vector<float> face_uv;
  std::vector<int32_t> face_of_orig;  // (used by -DRTMI_CHECK_MARGINS builds only) reference face order -> physical
  std::vector<MatRec> mat_recs;
  std::vector<TexRec> tex_recs;
  std::vector<void *> dev_allocs;
  SceneDev dev{};
  unsigned long long *d_counters = nullptr;  // [0] work queue head, [1] total rays, [2] abandoned mesh searches
  int device = -1;
  int64_t bytes_per_ray = 0;
  // scratch of the longest-first scheduler (probe states, probe ray counts, costs, order)
  void *d_sched = nullptr;
  size_t sched_bytes = 0;

  // host-only flatten (no HIP calls); returns "" or an error message
  std::string flatten();
  std::string check_margins() const;  // the error budget of the culls' structures, record by record (margins.h)
};

TriRec make_tri(V3 p0, V3 p1, V3 p2);
void camera_pinhole(Scene &s, V3 pos, V3 look_at, V3 up, double fov, double aspect);
void camera_defocus(Scene &s, V3 pos, V3 look_at, V3 up, double fov, double aspect, double aperture, double focus);
void camera_raw(Scene &s, V3 pos, V3 llc, V3 horiz, V3 vert);
void box_from_points(const V3 p[4], V3 out[8]);
void box_faces(const V3 corners[8], V3 faces[18]);

}  // namespace rtmi
