// Type tags carried by every recorded scene object.  The classes in this directory keep the
// reference's names and constructor signatures (so scenes/*.cu compile unchanged, including
// device-side `new` inside the user's InitWorld<<<1,1>>> kernel) but hold no rendering code:
// they record their constructor arguments, and after init_world a flatten kernel
// (src/rt_api.hip) walks world->list_ and hands the scene to librtmi.so through the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#define RT_API __host__ __device__

namespace rtapi {
enum HitableKind : int { H_LIST = 1, H_SPHERE, H_TRIANGLE, H_PARALLELOGRAM, H_PARALLELEPIPED, H_SKY, H_BVH, H_AABB };
enum MaterialKind : int { M_LAMBERTIAN = 1, M_METAL, M_DIELECTRIC, M_DIFFUSE_LIGHT, M_SKY };
enum TextureKind : int { T_CONSTANT = 1, T_IMAGE };
}  // namespace rtapi
