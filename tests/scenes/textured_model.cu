#define STB_IMAGE_IMPLEMENTATION  // as scenes/bunny.cu:1 does: model.h calls stbi_load
// A scene program of this repository's own, written against the reference's scene-facing API: a mesh loaded by
// Model<true> (model.h:49-113) whose OBJ names two materials -- one with a map_Kd diffuse texture, one without --
// rendered as two BVH<Face<true>, AABB> hitables (bvh.cuh:161-183), the first under Lambertian(ImageTexture)
// (lambertian.cu:9-12, textures/image_texture.cu:9-38), the second under a Metal.  It exercises the loader's
// usemtl / mtllib / map_Kd path and the per-face texture coordinates that no shipped scene uses
// (scenes/bunny.cu loads Model<false>; scenes/birthday.cu textures a sphere).
// tests/test_gpu_round3.py writes resources/models/quilt.obj + .mtl and resources/textures/quilt.ppm, runs this
// program and compares its frame with the oracle's render of the same world.
#include <glm/glm.hpp>
#include <glm/gtc/constants.hpp>

#include "bvh.cuh"
#include "camera.cuh"
#include "hitable_list.cuh"
#include "lambertian.cuh"
#include "metal.cuh"
#include "model.h"
#include "parallelogram.cuh"
#include "ray_tracing.cuh"
#include "sky.cuh"
#include "textures/constant_texture.cuh"
#include "textures/image_texture.cuh"
#include "utils.cuh"

const int WIDTH = 56, HEIGHT = 40;

curandState *d_states;
Camera *d_camera;
HitableList *d_world;
glm::vec3 *d_image;

using glm::vec3;

__global__ void BuildStage(HitableList *world, Camera *camera) {
  new (world) HitableList();
  new (camera) Camera(vec3(0.2, 1.1, 2.6), vec3(0, 0.5, 0), vec3(0, 1, 0), glm::pi<double>() / 3, double(WIDTH) / HEIGHT);
  vec3 floor_pts[3] = {vec3(-4, 0, -4), vec3(4, 0, -4), vec3(-4, 0, 4)};
  world->Append(new Parallelogram(floor_pts, new Lambertian(vec3(0.55, 0.55, 0.5))));
  world->Append(new Sky());
}

__global__ void AddTexturedMesh(HitableList *world, Face<true> *faces, int n, cudaTextureObject_t tex) {
  world->Append(new BVH<Face<true>, AABB>(faces, n, new Lambertian(new ImageTexture(tex))));
}

__global__ void AddPlainMesh(HitableList *world, Face<true> *faces, int n) {
  world->Append(new BVH<Face<true>, AABB>(faces, n, new Metal(vec3(0.8, 0.75, 0.6), 0.1f)));
}

static Face<true> *Upload(const std::vector<Face<true>> &faces) {
  Face<true> *d = nullptr;
  auto err = cudaMalloc(&d, sizeof(Face<true>) * faces.size());
  CHECK(err == cudaSuccess) << cudaGetErrorString(err);
  err = cudaMemcpy(d, faces.data(), sizeof(Face<true>) * faces.size(), cudaMemcpyHostToDevice);
  CHECK(err == cudaSuccess) << cudaGetErrorString(err);
  return d;
}

int main() {
  Main(
      &d_states, &d_camera, &d_world, &d_image,
      [](HitableList *world, Camera *camera) {
        BuildStage<<<1, 1>>>(world, camera);
        Model<true> model("resources/models/quilt.obj", glm::mat4(1));
        CHECK(model.meshes.size() == 2) << "expected one mesh per material";
        CHECK(model.meshes[0].texture_id == 0) << "the first material names a diffuse texture";
        CHECK(model.meshes[1].texture_id < 0) << "the second material has none";
        const Image &im = model.textures[0];
        uint8_t *d_tex = nullptr;
        uint64_t pitch = 0;
        cudaMallocPitch(&d_tex, &pitch, 4 * im.width, im.height);
        cudaMemcpy2D(d_tex, pitch, im.data.data(), 4 * im.width, 4 * im.width, im.height, cudaMemcpyHostToDevice);
        auto tex = ImageTexture::CreateCudaTextureObj(d_tex, im.height, im.width, pitch);
        AddTexturedMesh<<<1, 1>>>(world, Upload(model.meshes[0].faces), (int)model.meshes[0].faces.size(), tex);
        AddPlainMesh<<<1, 1>>>(world, Upload(model.meshes[1].faces), (int)model.meshes[1].faces.size());
        cudaDeviceSynchronize();
        auto err = cudaGetLastError();
        CHECK(err == cudaSuccess) << cudaGetErrorString(err);
      },
      HEIGHT, WIDTH, 4);
  return 0;
}
