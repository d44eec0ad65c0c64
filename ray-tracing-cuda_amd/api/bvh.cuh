#pragma once
#include <glm/glm.hpp>
#include "hitable.cuh"

// Bounding-volume tag type of the reference (bvh.cuh:10-19); only used as a template argument.
struct AABB : public Hitable {
  glm::vec3 min, max;
  RT_API AABB() : Hitable(rtapi::H_AABB) {}
};

// Face<HasTexCoord>: a POD triangle, with texture coordinates when HasTexCoord (bvh.cuh:21-99).
// sizeof(Face<false>) == 36 and sizeof(Face<true>) == 60, as the scene sources assume when they
// size their cudaMalloc/cudaMemcpy (scenes/bunny.cu:67-71).
template <bool HasTexCoord>
class Face {
 public:
  glm::vec3 positions_[3];
  glm::vec2 tex_coords_[3];
  constexpr static bool kHasTexCoord = true;
  RT_API glm::vec3 &position(int i) { return positions_[i]; }
  RT_API glm::vec2 &tex_coord(int i) { return tex_coords_[i]; }
};
template <>
class Face<false> {
 public:
  glm::vec3 positions_[3];
  constexpr static bool kHasTexCoord = false;
  RT_API glm::vec3 &position(int i) { return positions_[i]; }
};

// BVH<T, BV>(faces, n, material) (bvh.cuh:161-183): records the device face array; the tree
// (stable sort on positions_[0].x, median split, leaves <= 2048 faces) is built by librtmi.so.
struct RtBvhBase : public Hitable {
  const void *objs_;
  int n_;
  int has_tex_coord_;
  Material *material_ptr_;
  RT_API RtBvhBase(const void *objs, int n, int has_uv, Material *m)
      : Hitable(rtapi::H_BVH), objs_(objs), n_(n), has_tex_coord_(has_uv), material_ptr_(m) {}
};

template <typename T, typename BV>
class BVH : public RtBvhBase {
 public:
  RT_API explicit BVH(T *objs, int n, Material *material_ptr)
      : RtBvhBase(objs, n, T::kHasTexCoord ? 1 : 0, material_ptr) {}
};
