// Diagnostic harness (not product): the large-batch weight-gradient GEMM (k_tn64) in several tile shapes at the Humanoid
// B = 1024 shapes (critics: 2 nets, dW2 256x256 + dW1 256x393; actor: head 34x256 + dW2 + dW1 256x376), with in-kernel phase
// stamps, and k_adam_red.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kprobe_tn64.hip -o tools/kprobe_tn64
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
  printf("%-44s %6.2f us/launch | last block phases (cycles):", name, us);
  for (int i = 1; i < n; ++i) printf(" %lld", h[i] - h[i - 1]);
  long long ph[8]; hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph));
  printf(" | loop: issue+mfma %lld drain %lld barrier %lld\n", ph[0], ph[1], ph[2]);
}

struct Shapes { int nets; int nprob; int N[3], K[3], ldw[3]; };

static int g_light = 0;   // "pmc" mode: few plain launches per variant (for rocprofv3 --pmc)
static int g_ld256 = 256;   // row stride (floats) of the 256-wide operands: 256 as in the engine, or padded (channel-conflict experiment)
template <int WN, int WK, int AN, int AK, int TM>
static int run(hipStream_t s, const char* label, const Shapes& sh, int M, float* dY, float* X, float* Gp, int S_override) {
  constexpr int TN = 16 * AN * WN, TK = 16 * AK * WK;
  Tn64Args a{};
  a.nprob = sh.nprob; a.M = M; a.nets = sh.nets; a.Gp = Gp; a.g_ns = 180000;
  int tiles = 0, woff = 0;
  for (int i = 0; i < sh.nprob; ++i) {
    Tn64Prob& t = a.pr[i];
    t.dY = dY + (size_t)i * 2 * M * 400; t.ldy = sh.N[i] >= 256 ? g_ld256 : 36; t.dy_ns = (long)M * g_ld256; t.N = sh.N[i];
    t.X = X + (size_t)i * 2 * M * 400; t.ldx = sh.K[i] == 256 ? g_ld256 : sh.ldw[i]; t.x_ns = sh.K[i] == 256 ? (long)M * g_ld256 : 0; t.K = sh.K[i];
    t.w_off = woff; t.ldw = sh.ldw[i]; woff += sh.N[i] * sh.ldw[i]; t.b_off = woff; woff += 256;
    t.tiles_n = (t.N + TN - 1) / TN; t.tiles_k = (t.ldw + TK - 1) / TK; t.tile0 = tiles; tiles += t.tiles_n * t.tiles_k;
  }
  a.tiles_per_net = tiles;
  const int nch = (M + TM - 1) / TM;
  int S = S_override > 0 ? S_override : (2 * 256 + tiles * sh.nets / 2) / (tiles * sh.nets);
  S = std::max(1, std::min(S, std::min(8, nch)));
  a.S = S;
  const dim3 grid(tiles * sh.nets * S);
  if (g_light) {
    for (int i = 0; i < 6; ++i) hipLaunchKernelGGL((k_tn64<WN, WK, AN, AK, TM>), grid, dim3(256), 0, s, a);
    hipStreamSynchronize(s);
    printf("%s tile %dx%d TM=%d S=%d blocks=%d\n", label, TN, TK, TM, S, grid.x);
    return 0;
  }
  double us = graph_us(s, [&] { hipLaunchKernelGGL((k_tn64<WN, WK, AN, AK, TM>), grid, dim3(256), 0, s, a); }, 20, 30);
  char nm[128]; snprintf(nm, 128, "%s tile %dx%d TM=%d S=%d blocks=%d", label, TN, TK, TM, S, grid.x);
  show(nm, us, 4);
  return 0;
}

int main(int argc, char** argv) {
  g_light = argc > 1;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int M = 1024;
  float *dY, *X, *Gp;
  CK(hipMalloc(&dY, (size_t)3 * 2 * M * 400 * 4)); CK(hipMalloc(&X, (size_t)3 * 2 * M * 400 * 4)); CK(hipMalloc(&Gp, (size_t)8 * 2 * 180000 * 4));
  std::vector<float> h((size_t)3 * 2 * M * 400); for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 200) - 1.0f;
  CK(hipMemcpy(dY, h.data(), (size_t)3 * 2 * M * 400 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(Gp, 0, (size_t)8 * 2 * 180000 * 4));
  const Shapes critics{2, 2, {256, 256, 0}, {256, 393, 0}, {256, 396, 0}};
  const Shapes actor{1, 3, {34, 256, 256}, {256, 256, 376}, {256, 256, 376}};
  for (int pass = 0; pass < 4; ++pass) {
    const Shapes& sh = (pass & 1) == 0 ? critics : actor;
    g_ld256 = pass < 2 ? 256 : 272;
    const char* lb = pass == 0 ? "critics ld256" : (pass == 1 ? "actor ld256" : (pass == 2 ? "critics ld272" : "actor ld272"));
    if (g_light && pass != 0) break;
    for (int S : {0, 2}) {
      if (g_light) {
        if (S == 0) run<2, 2, 2, 1, 64>(s, lb, sh, M, dY, X, Gp, 0);
        else { run<2, 2, 2, 2, 64>(s, lb, sh, M, dY, X, Gp, 2); run<2, 2, 4, 2, 32>(s, lb, sh, M, dY, X, Gp, 2); }
        continue;
      }
      run<4, 1, 1, 2, 32>(s, lb, sh, M, dY, X, Gp, S);
      run<4, 1, 1, 2, 64>(s, lb, sh, M, dY, X, Gp, S);
      run<2, 2, 2, 1, 32>(s, lb, sh, M, dY, X, Gp, S);
      run<2, 2, 2, 1, 64>(s, lb, sh, M, dY, X, Gp, S);
      run<2, 2, 2, 2, 32>(s, lb, sh, M, dY, X, Gp, S);
      run<2, 2, 2, 2, 64>(s, lb, sh, M, dY, X, Gp, S);
      run<2, 2, 4, 2, 32>(s, lb, sh, M, dY, X, Gp, S);
    }
  }
  return 0;
}
