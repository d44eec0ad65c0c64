// CudaCopyable of the reference (cuda_copyable.cuh) is a base class whose two helpers are
// never called; it is kept as an empty base so Camera/Material/Texture derive from the same
// name.  Like the reference's header it pulls in the CUDA-runtime names and LOG/CHECK, which
// the scene sources use without including them themselves.
#pragma once
#include <cuda_runtime.h>
#include <glog/logging.h>
class CudaCopyable {};
