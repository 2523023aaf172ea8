// Design aid: what does a fork/join across two streams cost with plain (non-graph) launches?  A, B, C, D are ~4 us kernels
// on a fraction of the chip; V0 runs A B C D in one stream, V1 runs B on a second stream beside C (event fork + join).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_work(float* p, int iters) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
int main() {
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  float *a, *b; CK(hipMalloc(&a, 64 * 256 * 4)); CK(hipMalloc(&b, 64 * 256 * 4)); CK(hipMemset(a, 0, 64 * 256 * 4)); CK(hipMemset(b, 0, 64 * 256 * 4));
  for (int iters : {200, 1500}) {
    auto run = [&](int variant, int reps) {
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s1, a, iters);                       // A
        if (variant == 0) {
          hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s1, b, iters);                     // B
          hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s1, a, iters);                     // C
        } else {
          hipEventRecord(e1, s1); hipStreamWaitEvent(s2, e1, 0);
          hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s2, b, iters);                     // B on the side stream
          hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s1, a, iters);                     // C
          hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0);
        }
        hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s1, a, iters);                       // D
      }
    };
    for (int variant : {0, 1, 0, 1}) {
      run(variant, 50); hipStreamSynchronize(s1); hipStreamSynchronize(s2);
      auto t0 = std::chrono::steady_clock::now();
      run(variant, 2000); hipStreamSynchronize(s1); hipStreamSynchronize(s2);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
      printf("iters=%d variant=%d: %.2f us per A..D sequence\n", iters, variant, us);
    }
  }
  return 0;
}
