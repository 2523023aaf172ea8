// Diagnostic harness (not product): runs single kernels of csrc/kernels.h at the bench shapes with in-kernel
// phase stamps (SACTD3_STAMPS) and host-side per-launch times inside a captured graph.
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
  printf("%-22s %6.2f us/launch | last block phases (cycles):", name, us);
  for (int i = 1; i < n; ++i) printf(" %lld", h[i] - h[i - 1]);
  printf("  total %lld\n", h[n - 1] - h[0]);
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int B = 256, o = 11, a = 3, ldc = 16;
  float *X, *P, *Z2, *Y, *DX, *XH, *H, *RS;
  CK(hipMalloc(&X, B * ldc * 4)); CK(hipMalloc(&P, 2 * 80000 * 4)); CK(hipMalloc(&Z2, 2 * B * 256 * 4)); CK(hipMalloc(&Y, 2 * B * 256 * 4));
  CK(hipMalloc(&DX, 2 * B * 256 * 4)); CK(hipMalloc(&XH, 2 * B * 256 * 4)); CK(hipMalloc(&H, 2 * B * 256 * 4)); CK(hipMalloc(&RS, 2 * B * 4));
  std::vector<float> hp(2 * 80000); for (size_t i = 0; i < hp.size(); ++i) hp[i] = 0.01f * (float)((i * 2654435761u) % 200) - 1.0f;
  CK(hipMemcpy(P, hp.data(), hp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(X, hp.data(), B * ldc * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Z2, hp.data(), 2 * B * 256 * 4, hipMemcpyHostToDevice));
  const int W1 = 0, b1 = 256 * 16, g1 = b1 + 256, be1 = g1 + 256, W2 = be1 + 256, b2 = W2 + 65536;
  for (int ks : {4, 2, 1}) {
    const int nets = ks == 4 ? 1 : (ks == 2 ? 2 : 4);
    NtArgs h{};
    h.npg = nets; h.g[0].in = X; h.g[0].P = P; h.g[0].Y = Z2; h.g[0].xh_out = XH; h.g[0].h_out = H; h.g[0].rstd_out = RS;
    h.ld_in = ldc; h.in_ns = 0; h.oW = W2; h.ldw = 256; h.oBias = b2; h.oG = g1; h.oBe = be1; h.p_ns = 0;
    h.ldy = 256; h.y_ns = 0; h.M = B; h.N = 256; h.K = 256; h.K1 = o + a; h.oW1 = W1; h.ldw1 = 16; h.oB1 = b1; h.act_ns = 0;
    const int rb = 64 / ks;
    const dim3 grid((B / rb) * 16, 1, nets);
    double us;
    if (ks == 4) us = graph_us(s, [&] { hipLaunchKernelGGL((k_nt<1, true, 4, 1>), grid, dim3(256), 0, s, h); }, 20, 50);
    else if (ks == 2) us = graph_us(s, [&] { hipLaunchKernelGGL((k_nt<1, true, 2, 1>), grid, dim3(256), 0, s, h); }, 20, 50);
    else us = graph_us(s, [&] { hipLaunchKernelGGL((k_nt<1, true, 1, 1>), grid, dim3(256), 0, s, h); }, 20, 50);
    char nm[64]; snprintf(nm, 64, "k_nt<1,fused,KS=%d> nets=%d", ks, nets); show(nm, us, 6);
  }
  {  // critic weight-gradient launch: dW2 (256 x 256) + dW1 (256 x 16), 2 nets, Adam on
    float *G, *Mo, *Vo, *T, *adam, *part, *ps;
    CK(hipMalloc(&G, 2 * 80000 * 4)); CK(hipMalloc(&Mo, 2 * 80000 * 4)); CK(hipMalloc(&Vo, 2 * 80000 * 4)); CK(hipMalloc(&T, 2 * 80000 * 4));
    CK(hipMalloc(&adam, 16)); CK(hipMalloc(&part, 2 * 16 * NSLOT * 256 * 4)); CK(hipMalloc(&ps, 2 * 16 * 2 * 4));
    CK(hipMemset(G, 0, 2 * 80000 * 4)); CK(hipMemset(Mo, 0, 2 * 80000 * 4)); CK(hipMemset(Vo, 0, 2 * 80000 * 4)); CK(hipMemset(T, 0, 2 * 80000 * 4));
    float ha[2] = {1e-3f, 0.03f}; CK(hipMemcpy(adam, ha, 8, hipMemcpyHostToDevice)); CK(hipMemset(part, 0, 2 * 16 * NSLOT * 256 * 4)); CK(hipMemset(ps, 0, 2 * 16 * 2 * 4));
    TnArgs g{}; g.nprob = 2; g.M = B; g.G = G; g.g_ns = 80000;
    TnProb q0{}; q0.dY = Z2; q0.ldy = 256; q0.dy_ns = B * 256; q0.N = 256; q0.X = H; q0.ldx = 256; q0.x_ns = B * 256; q0.K = 256; q0.w_off = W2; q0.ldw = 256; q0.b_off = b2;
    q0.nfin = 3; q0.fin_slot[0] = 0; q0.fin_off[0] = b2 + 256; q0.fin_slot[1] = 1; q0.fin_off[1] = b2 + 512; q0.fin_slot[2] = 2; q0.fin_off[2] = b2 + 768; q0.fin_s_off = b2 + 768 + 256; q0.fin_s_nblk = 16; q0.fin_nblk[0] = q0.fin_nblk[1] = q0.fin_nblk[2] = 16; q0.tile0 = 0;
    TnProb q1{}; q1.dY = Y; q1.ldy = 256; q1.dy_ns = B * 256; q1.N = 256; q1.X = X; q1.ldx = ldc; q1.x_ns = 0; q1.K = o + a; q1.w_off = W1; q1.ldw = 16; q1.b_off = b1;
    q1.nfin = 2; q1.fin_slot[0] = 3; q1.fin_off[0] = g1; q1.fin_slot[1] = 4; q1.fin_off[1] = be1; q1.fin_s_off = -1; q1.fin_nblk[0] = q1.fin_nblk[1] = 16; q1.tile0 = 256;
    g.pr[0] = q0; g.pr[1] = q1; g.tiles = 272; g.apply = 1; g.P = P; g.Mo = Mo; g.Vo = Vo; g.T = T; g.tau = 0.005f; g.adam = adam; g.b1 = 0.9f; g.b2 = 0.999f; g.eps = 1e-8f;
    g.part = part; g.pstride = 16; g.part_s = ps; g.loss_part = ps; g.loss_n = 32; g.loss_stride = 2; g.loss_off = 1; g.loss_scale = 1.f / B; g.loss_dst = adam + 2;
    double us = graph_us(s, [&] { hipLaunchKernelGGL(k_tn<1>, dim3(272, 1, 2), dim3(256), 0, s, g); }, 20, 50);
    show("k_tn critics (+Adam)", us, 5);
    g.apply = 0;
    us = graph_us(s, [&] { hipLaunchKernelGGL(k_tn<1>, dim3(272, 1, 2), dim3(256), 0, s, g); }, 20, 50);
    show("k_tn critics (grads only)", us, 5);
  }
  {
    NetLayout L{}; L.K = o; L.ld1 = 12; L.nh = 2 * a; L.W1 = 0; L.b1 = b1; L.g1 = g1; L.be1 = be1; L.W2 = W2; L.b2 = b2; L.g2 = b2 + 256; L.be2 = b2 + 512; L.Wh = b2 + 768; L.bh = L.Wh + 6 * 256;
    DevCtl hc{}; DevCtl* ctl; CK(hipMalloc(&ctl, sizeof(DevCtl))); CK(hipMemcpy(ctl, &hc, sizeof(hc), hipMemcpyHostToDevice));
    float *eps, *sc, *logp, *tg; CK(hipMalloc(&eps, B * 4 * 4)); CK(hipMalloc(&sc, 64)); CK(hipMalloc(&logp, B * 4)); CK(hipMalloc(&tg, B * 16 * 4));
    float hs[16] = {1, 1, 1, 1, 0, 0, 0, 0, -1, -1, -1, -1, 1, 1, 1, 1}; CK(hipMemcpy(sc, hs, 64, hipMemcpyHostToDevice));
    ActorTail t{}; t.z2 = Z2; t.P = P; t.L = L; t.B = B; t.o = o; t.a = a; t.ln = 1; t.sac = 1; t.mode = 0; t.train = 1;
    t.ctl = ctl; t.ctr = &ctl->noise_ctr; t.site_buf = 1; t.site_code = 16; t.eps = eps; t.scale = sc; t.bias = sc + 4; t.min_ac = sc + 8; t.max_ac = sc + 12;
    t.dst = X; t.ldd = ldc; t.dst_off = o; t.logp = logp; t.h2 = H; t.xh2 = XH; t.rstd2 = RS; t.tg = tg; t.a4 = 4;
    double us = graph_us(s, [&] { hipLaunchKernelGGL(k_actor_tail, dim3(B / 16), dim3(256), 0, s, t); }, 20, 50);
    show("k_actor_tail(train)", us, 6);
    float *q3, *dz, *part, *ps; CK(hipMalloc(&q3, 8 * B * 4)); CK(hipMalloc(&dz, 2 * B * 256 * 4)); CK(hipMalloc(&part, 2 * 64 * NSLOT * 256 * 4)); CK(hipMalloc(&ps, 2 * 64 * 2 * 4));
    NetLayout Lc = L; Lc.nh = 1;
    CriticTail c{}; c.z2t = Z2; c.z2 = Y; c.PT = P; c.P = P; c.p_ns = 80000; c.L = Lc; c.rew = logp; c.done = logp; c.logp_next = logp; c.log_alpha = sc + 4;
    c.B = B; c.ln = 1; c.sac = 1; c.bcq = 0; c.gamma = 0.99f; c.qt = q3; c.y = q3 + 2 * B; c.q = q3 + 4 * B; c.dz2 = dz; c.part = part; c.part_s = ps; c.pstride = 16;
    us = graph_us(s, [&] { hipLaunchKernelGGL(k_critic_tail<16>, dim3(16, 2), dim3(256), 0, s, c); }, 20, 50);
    show("k_critic_tail", us, 4);
  }
  return 0;
}
