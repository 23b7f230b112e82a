// masklab.hip -- where does a CU-masked stream put its workgroups?  (tools only; the side queue of DESIGN.md section 12)
//   hipcc -O2 --offload-arch=gfx950 -o tools/lab/masklab tools/lab/masklab.hip
// For masks of the first M bits (and, for comparison, every (256/M)-th bit) of hipExtStreamCreateWithCUMask: G workgroups of
// 256 threads spin ~30 us each and note XCC_ID / HW_ID; the host counts the distinct (XCD, SE, SH, CU) they ran on and the
// workgroups that shared a CU at the same time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>
__global__ void k_where(unsigned* out, long long ticks) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
  }
}
static void report(const char* what, hipStream_t st, int G, unsigned* dev) {
  std::vector<unsigned> h((size_t)2 * G);
  hipMemsetAsync(dev, 0xff, sizeof(unsigned) * 2 * G, st);
  hipLaunchKernelGGL(k_where, dim3(G), dim3(256), 0, st, dev, 75000LL);    // ~30 us at 2.5 ticks per ns
  hipStreamSynchronize(st);
  hipMemcpy(h.data(), dev, sizeof(unsigned) * 2 * G, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu; std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int g = 0; g < G; ++g) {
    const unsigned xcc = h[2 * g] & 0xf, hw = h[2 * g + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[key] += 1; per_xcc[xcc].insert(key & 0xfff);
  }
  printf("%-28s %4d workgroups on %3zu distinct CUs:", what, G, per_cu.size());
  for (auto& x : per_xcc) printf("  xcd%u:%zu", x.first, x.second.size());
  int mx = 0; for (auto& c : per_cu) if (c.second > mx) mx = c.second;
  printf("  (most workgroups on one CU over the launch: %d)\n", mx);
}
int main() {
  unsigned* dev; hipMalloc((void**)&dev, sizeof(unsigned) * 2 * 4096);
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  printf("device reports %d CUs\n", ncu);
  hipStream_t plain; hipStreamCreate(&plain);
  report("no mask, 256 workgroups", plain, 256, dev);
  report("no mask, 1024 workgroups", plain, 1024, dev);
  for (int M : {32, 64, 96, 128}) {
    uint32_t mask[8] = {0};
    for (int c = 0; c < M; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t st; if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("mask stream failed\n"); return 1; }
    char name[64]; snprintf(name, sizeof name, "first %d bits, G=%d", M, M); report(name, st, M, dev);
    snprintf(name, sizeof name, "first %d bits, G=%d", M, 4 * M); report(name, st, 4 * M, dev);
    hipStreamDestroy(st);
    uint32_t m2[8] = {0};
    for (int c = 0; c < M; ++c) { const int bit = c * (256 / M); m2[bit >> 5] |= 1u << (bit & 31); }
    if (hipExtStreamCreateWithCUMask(&st, 8, m2) != hipSuccess) { printf("mask stream failed\n"); return 1; }
    snprintf(name, sizeof name, "every %dth bit (%d), G=%d", 256 / M, M, 4 * M); report(name, st, 4 * M, dev);
    hipStreamDestroy(st);
  }
  return 0;
}
