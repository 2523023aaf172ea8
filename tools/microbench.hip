// Launch-floor microbenchmarks for design decisions (not part of the product):
//   per-node cost of trivial kernels in a captured linear graph, effect of grid size / kernarg size /
//   dependent global loads, and whether fork-join branches of a graph overlap.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { float* p; int pad[60]; };
__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] += (float)b.pad[59]; }
__global__ void k_chain(float* p, int hops) {  // dependent loads: p[i] holds the next index
  if (threadIdx.x == 0 && blockIdx.x == 0) { int i = 0; for (int h = 0; h < hops; ++h) i = (int)p[i]; p[1] = (float)i; }
}
__global__ void k_spin(float* p, int iters) { float x = p[threadIdx.x & 3]; for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f; if (x == 123.f) p[0] = x; }

template <class F>
static double time_graph(hipStream_t s, F&& build, int reps, int* nodes) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  build();
  hipStreamEndCapture(s, &g);
  size_t n = 0; hipGraphGetNodes(g, nullptr, &n); *nodes = (int)n;
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us;
}

int main() {
  hipStream_t s, s2, s3, s4; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s4, hipStreamNonBlocking));
  float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  Big big{}; big.p = d;
  const int N = 48, reps = 300; int nodes;
  double t;
  t = time_graph(s, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, reps, &nodes);
  printf("linear %d x empty<1,64>          : %7.2f us/replay  %6.2f us/node\n", nodes, t, t / nodes);
  t = time_graph(s, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }, reps, &nodes);
  printf("linear %d x empty<256,256>       : %7.2f us/replay  %6.2f us/node\n", nodes, t, t / nodes);
  t = time_graph(s, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(32), dim3(1024), 0, s); }, reps, &nodes);
  printf("linear %d x empty<32,1024>       : %7.2f us/replay  %6.2f us/node\n", nodes, t, t / nodes);
  t = time_graph(s, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, d); }, reps, &nodes);
  printf("linear %d x touch<64,256>        : %7.2f us/replay  %6.2f us/node\n", nodes, t, t / nodes);
  t = time_graph(s, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s, big); }, reps, &nodes);
  printf("linear %d x 256B-kernarg<64,256> : %7.2f us/replay  %6.2f us/node\n", nodes, t, t / nodes);
  for (int hops : {1, 4, 16}) {
    t = time_graph(s, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s, d, hops); }, reps, &nodes);
    printf("linear %d x %2d dependent loads   : %7.2f us/replay  %6.2f us/node\n", nodes, hops, t, t / nodes);
  }
  for (int iters : {1000, 4000}) {
    t = time_graph(s, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_spin, dim3(16), dim3(256), 0, s, d, iters); }, reps, &nodes);
    double serial = t;
    printf("linear %d x spin(%d)<16,256>    : %7.2f us/replay  %6.2f us/node\n", nodes, iters, t, t / nodes);
    // fork-join: 4 branches x N/4 kernels
    hipEvent_t ef, e2, e3, e4; hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    hipEventCreateWithFlags(&e3, hipEventDisableTiming); hipEventCreateWithFlags(&e4, hipEventDisableTiming);
    t = time_graph(s, [&] {
      hipLaunchKernelGGL(k_spin, dim3(16), dim3(256), 0, s, d, iters);
      hipEventRecord(ef, s);
      hipStreamWaitEvent(s2, ef, 0); hipStreamWaitEvent(s3, ef, 0); hipStreamWaitEvent(s4, ef, 0);
      for (int i = 0; i < N / 4; ++i) {
        hipLaunchKernelGGL(k_spin, dim3(16), dim3(256), 0, s, d, iters);
        hipLaunchKernelGGL(k_spin, dim3(16), dim3(256), 0, s2, d + 1024, iters);
        hipLaunchKernelGGL(k_spin, dim3(16), dim3(256), 0, s3, d + 2048, iters);
        hipLaunchKernelGGL(k_spin, dim3(16), dim3(256), 0, s4, d + 4096, iters);
      }
      hipEventRecord(e2, s2); hipEventRecord(e3, s3); hipEventRecord(e4, s4);
      hipStreamWaitEvent(s, e2, 0); hipStreamWaitEvent(s, e3, 0); hipStreamWaitEvent(s, e4, 0);
      hipLaunchKernelGGL(k_spin, dim3(16), dim3(256), 0, s, d, iters);
    }, reps, &nodes);
    printf("fork-join 4 x %d spin(%d)        : %7.2f us/replay  (serial %d nodes: %.2f) nodes=%d\n", N / 4, iters, t, N, serial, nodes);
  }
  // eager stream launches for comparison
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, d);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < 100; ++r) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, d);
  hipStreamSynchronize(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100;
  printf("eager  %d x touch<64,256>        : %7.2f us/batch   %6.2f us/launch\n", N, us, us / N);
  return 0;
}
