// Does it matter WHICH block of the previous kernel produced the data a block reads first?  A linear hipGraph of 28 nodes, each
// `k_hop`: block b reads the 1 KB (x `lines`) region that block (b + shift) % nb of the PREVIOUS node wrote, and writes its own.
// Blocks are dealt round-robin over the 8 XCDs, so shift 0 = same block slot (same XCD, probably same CU), shift 8 = same XCD,
// shift 1 = the neighbouring XCD.   hipcc --offload-arch=gfx950 -O3 tools/xcd_affinity.hip -o /tmp/xcd_affinity
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_hop(const float4* src, float4* dst, int shift, int lines) {
  const int nb = gridDim.x, b = blockIdx.x, sb = (b + shift) % nb;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < lines; ++l) {                      // (independent loads: one round trip)
    const float4 v = src[((size_t)sb * lines + l) * 256 + threadIdx.x];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  for (int l = 0; l < lines; ++l) dst[((size_t)b * lines + l) * 256 + threadIdx.x] = make_float4(acc.x + 1.f, acc.y, acc.z, acc.w);
}
static double run(hipGraphExec_t ge, hipStream_t s, int reps) {
  for (int i = 0; i < 200; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 28;
  for (int nb : {64, 256, 512}) for (int lines : {1, 4, 16}) {
    float4 *a, *b;
    const size_t bytes = (size_t)nb * lines * 256 * sizeof(float4);
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    printf("blocks %3d x %2d KB:", nb, lines * 4);
    for (int shift : {0, 8, 1, 3, nb / 2 + 1}) {
      hipGraph_t g; CK(hipGraphCreate(&g, 0));
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_hop, dim3(nb), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, shift, lines);
      CK(hipStreamEndCapture(s, &g));
      hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      double best = 1e9;
      for (int r = 0; r < 3; ++r) { const double us = run(ge, s, 2000); if (us < best) best = us; }
      printf("  shift %3d: %5.2f us/node", shift, best / N);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    printf("\n");
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
