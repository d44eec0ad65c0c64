#pragma once
#include <curand_kernel.h>
#include <glm/glm.hpp>
#include "cuda_copyable.cuh"
#include "ray.cuh"

// Camera: the three constructors of camera.cu:6-47 computing the same frame
// (w = normalize(pos - look_at), u = normalize(up x w), v = normalize(w x u), image plane at
// distance 1 — or at the focus distance for the defocus camera).  Ray generation runs in
// librtmi.so.  half_height_/aspect_ are kept so Main can re-derive the frame when
// RT_WIDTH/RT_HEIGHT override the scene's compile-time resolution.
class Camera : public CudaCopyable {
 public:
  glm::vec3 position_, lower_left_corner_, horizontal_, vertical_, u_, v_, w_;
  bool is_defocus_camera_ = false;
  double lens_radius_ = -1;
  double half_height_ = 0, aspect_ = 0;  // 0: raw-frame camera, nothing to re-derive

  Camera() = delete;
  RT_API Camera(glm::vec3 position, glm::vec3 look_at, glm::vec3 up, double field_of_view, double width_height_aspect) {
    frame(position, look_at, up);
    half_height_ = tan(field_of_view / 2);
    aspect_ = width_height_aspect;
    plane();
  }
  RT_API Camera(glm::vec3 position, glm::vec3 look_at, glm::vec3 up, double field_of_view, double width_height_aspect,
                double aperture, double focus_distance) {
    is_defocus_camera_ = true;
    frame(position, look_at, up);
    half_height_ = focus_distance * tan(field_of_view / 2);
    aspect_ = width_height_aspect;
    plane();
    lens_radius_ = aperture / 2;
  }
  RT_API Camera(glm::vec3 position, glm::vec3 lower_left_corner, glm::vec3 horizontal, glm::vec3 vertical)
      : position_(position), lower_left_corner_(lower_left_corner), horizontal_(horizontal), vertical_(vertical) {}

  RT_API glm::vec3 position() const { return position_; }
  RT_API glm::vec3 lower_left_corner() const { return lower_left_corner_; }
  RT_API glm::vec3 horizontal() const { return horizontal_; }
  RT_API glm::vec3 vertical() const { return vertical_; }
  RT_API bool is_defocus_camera() const { return is_defocus_camera_; }

  // horizontal/vertical/lower-left from u,v,w and the half extents (camera.cu:15-20, 32-37)
  RT_API void plane() {
    double half_width = aspect_ * half_height_;
    horizontal_ = u_ * static_cast<float>(2 * half_width);
    vertical_ = v_ * static_cast<float>(2 * half_height_);
    lower_left_corner_ = position_ - w_ - u_ * static_cast<float>(half_width) - v_ * static_cast<float>(half_height_);
  }

 private:
  RT_API void frame(glm::vec3 position, glm::vec3 look_at, glm::vec3 up) {
    position_ = position;
    w_ = glm::normalize(position - look_at);
    u_ = glm::normalize(glm::cross(up, w_));
    v_ = glm::normalize(glm::cross(w_, u_));
  }
};
