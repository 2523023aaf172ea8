// Diagnostic (not product): where do the blocks of a launch land?  Each block records (XCC id, HW_ID) and its start / end
// time; the host prints how many blocks each CU received for grids of 256 .. 1024 blocks of 256 threads at several LDS sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k_census(unsigned* out, int spin) {
  extern __shared__ float lds[];
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  lds[threadIdx.x] = a;
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[4 * blockIdx.x] = hw; out[4 * blockIdx.x + 1] = xcc; out[4 * blockIdx.x + 2] = (unsigned)t0; out[4 * blockIdx.x + 3] = (unsigned)t1 + (lds[5] > 1e30f); }
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 16);
  std::vector<unsigned> h(4096 * 4);
  for (int lds : {4096, 36000, 49152, 70000}) for (int blocks : {256, 352, 504, 1024}) {
    hipLaunchKernelGGL(k_census, dim3(blocks), dim3(256), lds, 0, d, 20000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu; std::map<unsigned, int> per_xcc;
    unsigned tmin = ~0u, tmax = 0;
    for (int b = 0; b < blocks; ++b) {
      const unsigned hw = h[4 * b], xcc = h[4 * b + 1] & 0xf;
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++; per_xcc[xcc]++;
      tmin = std::min(tmin, h[4 * b + 2]); tmax = std::max(tmax, h[4 * b + 3]);
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("lds %6d blocks %5d: distinct CUs %3zu | blocks-per-CU histogram:", lds, blocks, per_cu.size());
    for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf(" | per XCC:");
    for (auto& kv : per_xcc) printf(" %d", kv.second);
    printf(" | span %.2f us\n", (tmax - tmin) * 0.01);
  }
  return 0;
}
