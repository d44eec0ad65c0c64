// Which bits of HW_ID / XCC_ID identify a compute unit on this device?  One workgroup per slot of the
// grid records its registers while all of them are resident; prints how many distinct values each
// field takes and how the workgroups spread over (XCC, SE, SH, CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(unsigned *o, int spin) {
  unsigned hw = __builtin_amdgcn_s_getreg(0xF804), xcc = __builtin_amdgcn_s_getreg(0xF814);
  volatile int x = 0;
  for (int i = 0; i < spin; i++) x += i;  // stay resident while the rest of the grid arrives
  if (threadIdx.x == 0) o[blockIdx.x * 2] = hw, o[blockIdx.x * 2 + 1] = xcc;
}
int main() {
  int blocks = 256 * 3;
  unsigned *d;
  hipMalloc(&d, blocks * 8);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 48 * 1024, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * 2);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu, f_cu, f_sh, f_se, f_xcc;
  for (int b = 0; b < blocks; b++) {
    unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 15u;
    per_cu[(xcc << 8) | ((hw >> 8) & 255u)]++;
    f_cu[(hw >> 8) & 15u]++, f_sh[(hw >> 12) & 1u]++, f_se[(hw >> 13) & 7u]++, f_xcc[xcc]++;
  }
  printf("distinct (xcc, hw[15:8]) = %zu for %d workgroups\n", per_cu.size(), blocks);
  std::map<int, int> hist;
  for (auto &kv : per_cu) hist[kv.second]++;
  for (auto &kv : hist) printf("  %d slots hold %d workgroups\n", kv.second, kv.first);
  printf("cu_id values:"); for (auto &kv : f_cu) printf(" %u:%d", kv.first, kv.second); printf("\n");
  printf("sh_id values:"); for (auto &kv : f_sh) printf(" %u:%d", kv.first, kv.second); printf("\n");
  printf("se_id values:"); for (auto &kv : f_se) printf(" %u:%d", kv.first, kv.second); printf("\n");
  printf("xcc values:"); for (auto &kv : f_xcc) printf(" %u:%d", kv.first, kv.second); printf("\n");
  printf("first raw: "); for (int b = 0; b < 6; b++) printf("%08x/%x ", h[b * 2], h[b * 2 + 1]); printf("\n");
  return 0;
}
