#pragma once
#include <glm/glm.hpp>
#include <nvfunctional>
#include "hitable_list.cuh"
#include "parallelogram.cuh"

// Parallelepiped: six parallelograms, three around corner p[0] and three around the opposite
// corner q[0] (parallelepiped.cu:8-55).  Both constructors record the 18 face points in the
// order AddCorner appends them.
class Parallelepiped : public Hitable {
 public:
  glm::vec3 faces_[6][3];
  Material *material_ptr_;

  // four points: a corner and its three neighbours (parallelepiped.cu:8-18)
  RT_API Parallelepiped(glm::vec3 p[4], Material *material_ptr)
      : Hitable(rtapi::H_PARALLELEPIPED), material_ptr_(material_ptr) {
    glm::vec3 q[4];
    q[3] = p[2] + p[1] - p[0];
    q[2] = p[3] + p[1] - p[0];
    q[1] = p[3] + p[2] - p[0];
    q[0] = q[3] + q[2] - p[1];
    corner(0, p);
    corner(3, q);
  }
  // edge lengths + a transform applied to the eight axis-aligned corners (parallelepiped.cu:34-55)
  RT_API Parallelepiped(glm::vec3 lengths, Material *material_ptr, nvstd::function<glm::vec3(glm::vec3)> transform)
      : Hitable(rtapi::H_PARALLELEPIPED), material_ptr_(material_ptr) {
    glm::vec3 p[4], q[4];
    p[0] = glm::vec3(0);
    q[0] = lengths;
    for (int i = 1; i <= 3; i++) {
      p[i] = glm::vec3(0);
      p[i][i - 1] = lengths[i - 1];
      q[i] = lengths;
      q[i][i - 1] = 0;
    }
    for (int i = 0; i < 4; i++) {
      p[i] = transform(p[i]);
      q[i] = transform(q[i]);
    }
    corner(0, p);
    corner(3, q);
  }

 private:
  // AddCorner (parallelepiped.cu:25-32): faces (c0,c1,c2), (c0,c2,c3), (c0,c3,c1)
  RT_API void corner(int first, const glm::vec3 c[4]) {
    for (int i = 1; i <= 3; i++) {
      int x = i, y = (i + 1 == 4) ? 1 : x + 1;
      faces_[first + i - 1][0] = c[0];
      faces_[first + i - 1][1] = c[x];
      faces_[first + i - 1][2] = c[y];
    }
  }
};
