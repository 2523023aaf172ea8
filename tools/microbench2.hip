// In-kernel cycle probes (design aid, not product): shader clock under a graph of tiny kernels,
// cost of wave reductions (ds_bpermute shuffles vs DPP), latency of loading data another kernel just wrote.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp<0x141>(v);   // row_half_mirror
  v += dpp<0x140>(v);   // row_mirror  -> every lane of a 16-lane row holds the row sum
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
         (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

__device__ __forceinline__ float wave_sum_dpp2(float v) {   // row reduce, then row_bcast15 / row_bcast31, read lane 63
  v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float row16_sum(float v) {  // every lane gets the sum over its 16-lane row
  v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
  return v;
}
__global__ void k_check(float* out) {
  const int lane = threadIdx.x;
  const float x = (float)(lane * lane % 17) + 0.25f * lane;
  out[lane] = wave_sum_shfl(x); out[64 + lane] = wave_sum_dpp(x); out[128 + lane] = wave_sum_dpp2(x); out[192 + lane] = row16_sum(x);
}
__global__ void k_probe(long long* out, const float* in, float* sink) {
  const int lane = threadIdx.x;
  long long c0 = clock64(); long long w0 = wall_clock64();
  float x = in[lane & 3];
  for (int i = 0; i < 2000; ++i) x = x * 1.0001f + 0.5f;
  long long c1 = clock64(); long long w1 = wall_clock64();
  float a = x;
  for (int i = 0; i < 64; ++i) a = wave_sum_shfl(a) * 0.5f;
  long long c2 = clock64();
  float b = x;
  for (int i = 0; i < 64; ++i) b = wave_sum_dpp(b) * 0.5f;
  long long c3 = clock64();
  float c = x;
  for (int i = 0; i < 64; ++i) c = wave_sum_dpp2(c) * 0.5f;
  long long c4 = clock64();
  if (lane == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = c2 - c1; out[3] = c3 - c2; out[7] = c4 - c3; }
  sink[lane] = a + b + c;
}
__global__ void k_produce(float* buf, int n, float v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) buf[i] = v + i; }
__global__ void k_consume(long long* out, const float* buf, float* sink, int stride) {
  // one wave: timed single dependent load of freshly produced data, then 8 independent loads, then 8 dependent
  const int lane = threadIdx.x;
  long long c0 = clock64();
  float v = buf[lane * 4];
  sink[lane] = v;   // force completion
  __builtin_amdgcn_s_waitcnt(0);
  long long c1 = clock64();
  float s = 0.f;
  for (int i = 1; i <= 8; ++i) s += buf[i * stride + lane * 4];
  sink[64 + lane] = s;
  __builtin_amdgcn_s_waitcnt(0);
  long long c2 = clock64();
  int idx = lane;
  for (int i = 0; i < 8; ++i) idx = ((int)buf[(9 + i) * stride + (idx & 63)]) & 1023;
  sink[128 + lane] = (float)idx;
  __builtin_amdgcn_s_waitcnt(0);
  long long c3 = clock64();
  if (lane == 0 && blockIdx.x == 0) { out[4] = c1 - c0; out[5] = c2 - c1; out[6] = c3 - c2; }
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  long long* d_out; float *d_in, *d_sink, *d_buf;
  const int NB = 1 << 22;
  CK(hipMalloc(&d_out, 64 * 8)); CK(hipMalloc(&d_in, 1024)); CK(hipMalloc(&d_sink, 4096)); CK(hipMalloc(&d_buf, NB * 4));
  CK(hipMemset(d_in, 0, 1024)); CK(hipMemset(d_out, 0, 512));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_produce, dim3(NB / 256 / 16), dim3(256), 0, s, d_buf, NB / 16, (float)i);
  hipLaunchKernelGGL(k_produce, dim3(NB / 256), dim3(256), 0, s, d_buf, NB, 1.0f);
  hipLaunchKernelGGL(k_consume, dim3(1), dim3(64), 0, s, d_out, d_buf, d_sink, 65536);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, s, d_out, d_in, d_sink);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  {
    float* d_chk; CK(hipMalloc(&d_chk, 1024)); float hc[256];
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, s, d_chk); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(hc, d_chk, 1024, hipMemcpyDeviceToHost));
    double want = 0; for (int l = 0; l < 64; ++l) want += (float)(l * l % 17) + 0.25f * l;
    int bad = 0; for (int l = 0; l < 64; ++l) { if (hc[l] != hc[64 + l] && fabs(hc[l]-hc[64+l]) > 1e-3) bad |= 1; if (fabs(hc[l] - hc[128 + l]) > 1e-3) bad |= 2; }
    double r0 = 0; for (int l = 0; l < 16; ++l) r0 += (float)(l * l % 17) + 0.25f * l;
    printf("reduction check: want %.3f shfl %.3f dpp %.3f dpp2 %.3f bad=%d ; row16 want %.3f got lane0 %.3f lane15 %.3f\n", want, hc[0], hc[64], hc[128], bad, r0, hc[192], hc[192 + 15]);
  }
  long long h[8];
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 200; ++i) hipGraphLaunch(ge, s);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    float chk = 0.f;
    double mhz = (double)h[0] / ((double)h[1] / 100.0);   // wall_clock64 ticks at 100 MHz
    printf("spin 2000 fma: %lld shader cycles, %lld wall ticks -> %.0f MHz, %.1f cyc/iter\n", h[0], h[1], mhz, h[0] / 2000.0);
    printf("wave_sum shfl: %.1f cyc each   dpp+4readlane: %.1f   dpp+bcast: %.1f\n", h[2] / 64.0, h[3] / 64.0, h[7] / 64.0);
    printf("load of data written by the previous kernel: first %lld cyc; 8 independent %lld cyc; 8 dependent %lld cyc\n", h[4], h[5], h[6]);
  }
  return 0;
}
