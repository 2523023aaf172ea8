// cycle cost of the transcendental sequence of the tanh-Gaussian head (design aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../sac-td3-cudagraphs-pytorch_amd/csrc/philox.h"
__global__ void k(long long* out, const float* in, float* sink) {
  const int l = threadIdx.x;
  float u0 = in[l], u1 = in[64 + l], e = in[128 + l], sc = 1.0f, bi = 0.f;
  __builtin_amdgcn_s_waitcnt(0);
  long long c0 = clock64();
  const float tt = tanhf(u1);
  long long c1 = clock64();
  const float sd = expf(-5.0f + 3.5f * (tt + 1.0f));
  long long c2 = clock64();
  const float x = u0 + e * sd;
  const float yt = tanhf(x);
  long long c3 = clock64();
  const float dx = x - u0;
  float lp = -(dx * dx) / (2.0f * sd * sd) - logf(sd) - 0.9189385332046727f;
  long long c4 = clock64();
  lp -= logf(sc * (1.0f - yt * yt) + 1e-6f);
  long long c5 = clock64();
  const Philox4 r = philox4x32_10(7u, 0u, 16u, (uint32_t)l, 123u, 456u);
  long long c6 = clock64();
  const float n1 = sqrtf(-2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(philox_u01(r.v[0]))) * __builtin_amdgcn_sinf(philox_u01(r.v[1]));
  long long c7 = clock64();
  const float n2 = sqrtf(-2.0f * logf(philox_u01(r.v[2]))) * sinf(6.283185307179586f * philox_u01(r.v[3]));
  long long c8 = clock64();
  sink[l] = lp + yt * sc + bi + n1 + n2;
  if (l == 0) { long long c[9] = {c0, c1, c2, c3, c4, c5, c6, c7, c8}; for (int i = 0; i < 8; ++i) out[i] = c[i + 1] - c[i]; }
}
int main() {
  long long* d; float *in, *sink; hipMalloc(&d, 128); hipMalloc(&in, 1024); hipMalloc(&sink, 1024);
  float h[192]; for (int i = 0; i < 192; ++i) h[i] = 0.37f * (i % 7) - 1.1f; hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, in, sink); hipDeviceSynchronize(); }
  long long o[8]; hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
  printf("tanhf %lld | expf %lld | tanhf %lld | div+logf %lld | logf %lld | philox %lld | fast BM %lld | accurate BM %lld cycles\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
  return 0;
}
