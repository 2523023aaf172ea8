// Launch-floor probe for hipGraph chains on one MI355X: (a) a linear chain of N tiny kernels, (b) the same with a 2-node side
// branch that forks after node 0 and joins at node J, (c) empty kernels.  Prints us per replay and per node.
//   hipcc --offload-arch=gfx950 -O3 tools/graph_branch.hip -o /tmp/graph_branch && /tmp/graph_branch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty() {}
__global__ void k_touch(float* p, int n) {           // 64 blocks x 256 threads, one dependent read-modify-write each
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
struct Args { float* p; int n; };
static hipGraphNode_t add(hipGraph_t g, const std::vector<hipGraphNode_t>& deps, bool empty, float* p, int n, int blocks, Args* store) {
  hipKernelNodeParams kp{};
  store->p = p; store->n = n;
  static void* argv_store[4096][2]; static int na = 0;
  void** argv = argv_store[na++];
  argv[0] = &store->p; argv[1] = &store->n;
  kp.func = empty ? (void*)k_empty : (void*)k_touch;
  kp.gridDim = dim3(empty ? 1 : blocks); kp.blockDim = dim3(256); kp.kernelParams = empty ? nullptr : argv;
  hipGraphNode_t node;
  if (hipGraphAddKernelNode(&node, g, deps.data(), deps.size(), &kp) != hipSuccess) { printf("add failed\n"); exit(1); }
  return node;
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
  float* buf; const int n = 64 * 256; CK(hipMalloc(&buf, 64 * n * sizeof(float))); CK(hipMemset(buf, 0, 64 * n * sizeof(float)));
  static Args store[4096]; int ns = 0;
  for (int variant = 0; variant < 6; ++variant) {
    // 0: 14 linear touch; 1: 12 linear touch; 2: 12 + side branch of 2 (fork after 0, join at 7); 3: 14 linear empty; 4: 28 linear touch; 5: 12 + side of 2, side touches 256 blocks
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    const bool empty = variant == 3;
    const int N = variant == 0 || variant == 3 ? 14 : variant == 4 ? 28 : 12;
    const bool side = variant == 2 || variant == 5;
    std::vector<hipGraphNode_t> prev;
    hipGraphNode_t first{}, side_end{};
    for (int i = 0; i < N; ++i) {
      std::vector<hipGraphNode_t> deps = prev;
      if (side && i == 7) deps.push_back(side_end);
      hipGraphNode_t nd = add(g, deps, empty, buf, n, 64, &store[ns++]);
      if (i == 0) {
        first = nd;
        if (side) {
          hipGraphNode_t s0 = add(g, {first}, false, buf + (size_t)8 * n, n, 64, &store[ns++]);
          side_end = add(g, {s0}, false, buf + (size_t)8 * n, n, 64, &store[ns++]);
        }
      }
      prev = {nd};
    }
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double best = 1e9;
    for (int r = 0; r < 3; ++r) { const double us = run(ge, s, 3000); if (us < best) best = us; }
    const char* names[] = {"14 linear", "12 linear", "12 linear + 2-node side branch (fork@0, join@7)", "14 linear EMPTY", "28 linear", "12 + side (again)"};
    printf("%-52s %8.2f us/replay  %6.2f us/node(main chain)\n", names[variant], best, best / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
