// Does gfx950 execute scalar memory atomics (s_atomic_add ... glc returns the pre-op value)?  hipcc --offload-arch=gfx950 -O3 -o sat scalar_atomic_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(int* ctr, int* out) {
  int v = 1;
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = v;
}
int main() {
  int *ctr, *out; const int nb = 2048, wpb = 4;
  hipMalloc(&ctr, 4); hipMalloc(&out, 4 * nb * wpb); hipMemset(ctr, 0, 4);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64 * wpb), 0, 0, ctr, out);
  int c; std::vector<int> h(nb * wpb);
  hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), out, 4 * nb * wpb, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool uniq = true; for (int i = 0; i < nb * wpb; ++i) uniq &= h[i] == i;
  printf("counter %d (expected %d), tickets unique and dense: %s\n", c, nb * wpb, uniq ? "yes" : "NO");
  return 0;
}
