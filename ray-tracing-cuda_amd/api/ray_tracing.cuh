#pragma once
#include <glm/glm.hpp>
#include "camera.cuh"
#include "hitable_list.cuh"
#include "ray.cuh"

// The reference declares its trace kernel here (ray_tracing.cuh:17-21).  In this build the
// kernel lives in librtmi.so behind include/rtmi.h (rtmi_render); scene programs reach it
// through Main / DistributedMain (utils.cuh).
