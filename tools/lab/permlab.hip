// what v_permlane32_swap / v_permlane16_swap (gfx950) do, lane by lane
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/permlab tools/lab/permlab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = threadIdx.x + 100;
  const v2u r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const v2u q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[4] = {"swap32[0]", "swap32[1]", "swap16[0]", "swap16[1]"};
  for (int v = 0; v < 4; ++v) { printf("%s:", nm[v]); for (int l = 0; l < 64; l += 8) printf(" %u", h[v * 64 + l]); printf("\n"); }
  return 0;
}
