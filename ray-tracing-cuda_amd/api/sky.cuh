#pragma once
#include "hitable.cuh"
#include "material.cuh"

// Sky: an always-hit object at t = 1e9 whose material emits the blue-white gradient (sky.cu).
class SkyMaterial : public Material {
 public:
  RT_API SkyMaterial() : Material(rtapi::M_SKY) {}
};

class Sky : public Hitable {
 public:
  RT_API Sky() : Hitable(rtapi::H_SKY) {}
  RT_API Material *material_ptr() { return &material_; }

 private:
  SkyMaterial material_;
};
