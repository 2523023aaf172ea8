// Fetch-rate probe (design aid): 256-thread blocks each pull `nld` x 16 B per thread (1 KB per wave-instruction)
// with different sharing / access shapes, right after another kernel rewrote the source (cold L2).
//   shape 0: coalesced (lane l -> 16 B at l*16 within a 1 KB run)
//   shape 1: MFMA-fragment style (lane (r = l&15, kq = l>>4) -> row r (stride 1 KB), 16 B at kq*16 + chunk*64)
//   share 0: every block reads its own region; 1: blocks b and b+8k share (same XCD under round-robin);
//   share 2: ALL blocks read the same region
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ long long g_cyc[4];
__global__ void k_touch(float* p, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 0.5f + 1.f; }
template <int SHAPE>
__global__ __launch_bounds__(256) void k_fetch(const float* src, float* sink, int nld, int share, int regions) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int region = share == 0 ? blockIdx.x : (share == 1 ? (blockIdx.x % 8) + 8 * ((blockIdx.x / 8) % (regions / 8)) : 0);
  const float* base = src + (long)region * (nld * 4096 / 4) * 1;   // block region = nld KB x 4 waves
  long long c0 = clock64();
  float4 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (u < nld) {
      long off;
      if (SHAPE == 0) off = ((long)(wave * nld + u) * 1024 + lane * 16) / 4;
      else off = ((long)(wave * nld * 1024) + (lane & 15) * (nld * 64) + u * 64 + (lane >> 4) * 16) / 4;   // 16 rows of nld*64 B
      v[u] = *reinterpret_cast<const float4*>(base + off);
    } else v[u] = make_float4(0, 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int u = 0; u < 16; ++u) s += v[u].x + v[u].w;
  __builtin_amdgcn_s_waitcnt(0);
  long long c1 = clock64();
  if (s == 12345.f) sink[t] = s;
  if (t == 0 && blockIdx.x == gridDim.x - 1) { g_cyc[0] = c1 - c0; }
  if (t == 0 && blockIdx.x == 0) { g_cyc[1] = c1 - c0; }
}
template <int SHAPE> static int run(hipStream_t s, float* src, float* sink, long nfl, int blocks, int nld, int share, const char* tag) {
  hipGraph_t g; hipGraphExec_t ge;
  const int regions = share == 0 ? blocks : (share == 1 ? 64 : 1);
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 10; ++i) {
    hipLaunchKernelGGL(k_touch, dim3((unsigned)((nfl + 255) / 256)), dim3(256), 0, s, src, nfl);
    hipLaunchKernelGGL(k_fetch<SHAPE>, dim3(blocks), dim3(256), 0, s, src, sink, nld, share, regions);
  }
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  // time the pair and the touch alone to subtract
  auto timeit = [&](hipGraphExec_t e) { for (int i = 0; i < 3; ++i) hipGraphLaunch(e, s); hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now(); for (int i = 0; i < 20; ++i) hipGraphLaunch(e, s); hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200.0; };
  double both = timeit(ge);
  hipGraph_t g2; hipGraphExec_t ge2;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_touch, dim3((unsigned)((nfl + 255) / 256)), dim3(256), 0, s, src, nfl);
  hipStreamEndCapture(s, &g2); hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
  double touch = timeit(ge2);
  long long h[4]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cyc), sizeof(h));
  printf("%-10s blocks=%4d KB/block=%3d share=%d : kernel %6.2f us  (last block %6lld cyc, block0 %6lld cyc)  total %.1f MB\n", tag, blocks, nld * 4, share,
         both - touch, h[0], h[1], blocks * nld * 4096.0 / 1e6);
  hipGraphExecDestroy(ge); hipGraphDestroy(g); hipGraphExecDestroy(ge2); hipGraphDestroy(g2);
  return 0;
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const long nfl = 1024L * 16 * 4096 / 4;   // 64 MB source (1024 blocks x 64 KB)
  float *src, *sink; CK(hipMalloc(&src, nfl * 4)); CK(hipMalloc(&sink, 4096)); CK(hipMemset(src, 0, nfl * 4));
  for (int share : {0, 1, 2})
    for (int blocks : {256, 512, 1024})
      for (int nld : {8, 16}) {
        run<0>(s, src, sink, share == 0 ? (long)blocks * nld * 1024 : 64L * nld * 1024, blocks, nld, share, "coalesced");
        run<1>(s, src, sink, share == 0 ? (long)blocks * nld * 1024 : 64L * nld * 1024, blocks, nld, share, "fragment");
      }
  return 0;
}
