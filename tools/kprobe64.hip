// Diagnostic harness (not product): k_nt64 at the Humanoid B=1024 shapes, graph-timed, with phase stamps.
#define SACTD3_STAMPS 1
#include "../sac-td3-cudagraphs-pytorch_amd/csrc/kernels.h"
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <class F> static double graph_us(hipStream_t s, F&& launch, int n_in_graph, int reps) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n_in_graph; ++i) launch();
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps / n_in_graph;
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us;
}
static void show(const char* name, double us, int n) {
  long long h[16]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h));
  printf("%-34s %6.2f us/launch | last block phases (cycles):", name, us);
  for (int i = 1; i < n; ++i) printf(" %lld", h[i] - h[i - 1]);
  printf("  total %lld\n", h[n - 1] - h[0]);
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int M = 1024, nets = 4, ld = 400;
  float *X, *P, *Y;
  const long pn = 256L * 400 + 1024;
  CK(hipMalloc(&X, (long)nets * M * ld * 4)); CK(hipMalloc(&P, nets * pn * 4)); CK(hipMalloc(&Y, (long)nets * M * 256 * 4));
  std::vector<float> hp(nets * M * ld); for (size_t i = 0; i < hp.size(); ++i) hp[i] = 0.01f * (float)((i * 2654435761u) % 200) - 1.0f;
  CK(hipMemcpy(X, hp.data(), hp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(P, hp.data(), nets * pn * 4, hipMemcpyHostToDevice));
  for (int K : {393, 256, 64}) {
    NtArgs g{};
    g.npg = nets; g.oW = 0; g.ldw = K == 393 ? 400 : 256; g.oBias = 256 * 400; g.p_ns = pn; g.ld_in = g.ldw; g.in_ns = (long)M * g.ldw;
    g.ldy = 256; g.y_ns = (long)M * 256; g.M = M; g.N = 256; g.K = K; g.g[0].in = X; g.g[0].P = P; g.g[0].Y = Y;
    const dim3 grid(16 * 4 * nets);
    double us = graph_us(s, [&] { hipLaunchKernelGGL((k_nt64<4, 2, 2>), grid, dim3(512), 0, s, g); }, 20, 50);
    char nm[64]; snprintf(nm, 64, "k_nt64 K=%d", K); show(nm, us, 3);
  }
  return 0;
}
