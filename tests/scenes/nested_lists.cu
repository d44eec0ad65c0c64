// A scene program of this repository's own (not one of the reference's four), written against the
// same scene-facing API: a world that appends HitableLists to HitableLists
// (/root/reference/ray-tracing-cuda/hitable_list.cuh:8 makes a list a Hitable), with coincident
// surfaces at different nesting levels.  tests/test_gpu_scene_programs.py renders it through Main()
// and compares the frame with the same world recorded through the Python binding
// (tests/test_gpu_random_scenes.py: _nested_world with no random extras).
#include <glm/glm.hpp>
#include <glm/gtc/constants.hpp>

#include "camera.cuh"
#include "dielectric.cuh"
#include "diffuse_light.cuh"
#include "hitable_list.cuh"
#include "lambertian.cuh"
#include "metal.cuh"
#include "parallelepiped.cuh"
#include "parallelogram.cuh"
#include "ray_tracing.cuh"
#include "sky.cuh"
#include "sphere.cuh"
#include "textures/constant_texture.cuh"
#include "triangle.cuh"
#include "utils.cuh"

const int WIDTH = 52, HEIGHT = 40;

curandState *d_states;
Camera *d_camera;
HitableList *d_world;
glm::vec3 *d_image;

using glm::vec3;

__device__ Parallelogram *Wall(float z, Material *m, float dx) {
  vec3 p[3] = {vec3(-1.5f + dx, -0.2f, z), vec3(1.5f + dx, -0.2f, z), vec3(-1.5f + dx, 1.8f, z)};
  return new Parallelogram(p, m);
}

__global__ void BuildNestedWorld(HitableList *world, Camera *camera) {
  new (world) HitableList();
  new (camera) Camera(vec3(0, 0.8, 2.2), vec3(0, 0.6, -1), vec3(0, 1, 0), glm::pi<double>() / 3, double(WIDTH) / HEIGHT);
  Material *m0 = new Lambertian(vec3(0.2, 0.6, 0.8));
  Material *m1 = new Lambertian(vec3(0.8, 0.3, 0.2));
  Material *m2 = new Metal(vec3(0.7, 0.7, 0.6), 0.25f);
  Material *m3 = new Dielectric(vec3(1, 1, 1), 1.5);
  Material *m4 = new DiffuseLight(new ConstantTexture(vec3(3, 3, 3)));

  world->Append(new Sky());
  HitableList *a = new HitableList();
  a->Append(Wall(-2.0f, m0, 0.0f));
  HitableList *b = new HitableList();
  b->Append(Wall(-2.0f, m1, 0.7f));
  b->Append(new Sphere(vec3(-0.6, 0.5, -1.0), 0.45, m2));
  HitableList *c = new HitableList();
  vec3 box[4] = {vec3(0.3, 0.0, -1.4), vec3(0.9, 0.0, -1.4), vec3(0.3, 0.7, -1.4), vec3(0.3, 0.0, -0.8)};
  c->Append(new Parallelepiped(box, m3));
  vec3 lid[3] = {vec3(0.3, 0.0, -0.8), vec3(0.9, 0.0, -0.8), vec3(0.3, 0.7, -0.8)};
  c->Append(new Parallelogram(lid, m1));
  b->Append(c);
  a->Append(b);
  vec3 tri[3] = {vec3(-1.4, 1.2, -1.9), vec3(-0.4, 1.2, -1.9), vec3(-0.9, 1.9, -1.9)};
  a->Append(new Triangle(tri, m4));
  world->Append(a);
  world->Append(new HitableList());  // an empty list
  HitableList *f = new HitableList();
  f->Append(new Sphere(vec3(0, -100.2, -1), 100.0, m0));
  world->Append(f);
  world->Append(Wall(-2.0f, m2, -0.9f));
}

int main() {
  Main(
      &d_states, &d_camera, &d_world, &d_image,
      [](HitableList *world, Camera *camera) { BuildNestedWorld<<<1, 1>>>(world, camera); }, HEIGHT, WIDTH, 4);
  return 0;
}
