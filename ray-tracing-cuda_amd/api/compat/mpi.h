// Process-topology stub for the scene sources' MPI calls (scenes/spheres.cu:83-98,108).
// The reference distributes over MPI ranks; this build runs one process per GPU and takes
// rank / world size from the launcher's environment (RANK / WORLD_SIZE as set by
// torch.distributed.run, or OMPI_COMM_WORLD_* / PMI_* when started by an MPI launcher).
// No message passing happens here: the frame exchange is an RCCL gather inside DistributedMain.
#pragma once
#include <cstdlib>

typedef int MPI_Comm;
#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0

namespace rt_mpi {
inline int env_int(const char *const *names, int dflt) {
  for (int i = 0; names[i]; i++) {
    const char *v = std::getenv(names[i]);
    if (v && *v) return std::atoi(v);
  }
  return dflt;
}
inline int rank() {
  static const char *n[] = {"RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", nullptr};
  return env_int(n, 0);
}
inline int size() {
  static const char *n[] = {"WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", nullptr};
  return env_int(n, 1);
}
}  // namespace rt_mpi

inline int MPI_Init(int *, char ***) { return MPI_SUCCESS; }
inline int MPI_Finalize() { return MPI_SUCCESS; }
inline int MPI_Comm_size(MPI_Comm, int *out) { *out = rt_mpi::size(); return MPI_SUCCESS; }
inline int MPI_Comm_rank(MPI_Comm, int *out) { *out = rt_mpi::rank(); return MPI_SUCCESS; }
