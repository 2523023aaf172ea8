// Host side of libsactd3_hip.so: memory plan, kernel sequences, hipGraph capture, C ABI (include/sactd3.h).
//
// The kernel sequences restate agents/agent.py:183-331 of the reference in the decomposition that
// oracle/manual_grads.py documents (and checks against autograd).  One engine owns:
//   - parameter / gradient / Adam arenas (flat, padded so that every row is float4 aligned),
//   - the HBM replay ring (one 64-byte-aligned record per transition),
//   - one batch slot and all activations of one iteration,
//   - a HIP stream and the captured graphs of update_qnets / update_actor / whole iterations.
// No torch, no BLAS: the .so depends on libamdhip64 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/sactd3.h"
#include "kernels.h"

static thread_local std::string g_create_error;

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// ceil(2^32 / d) if it divides every n < n_max exactly via mulhi (n_max * d < 2^32), else 0 (= use a real division)
static inline unsigned magic_div(unsigned d, unsigned long long n_max) {
  return (d > 1 && n_max * d < (1ull << 32)) ? (unsigned)(((1ull << 32) + d - 1) / d) : 0u;
}

static NetLayout make_layout(int K, int nh) {
  NetLayout L{};
  L.K = K; L.ld1 = round_up(K, 4); L.nh = nh;
  int off = 0;
  L.W1 = off; off += HID * L.ld1;
  L.b1 = off; off += HID; L.g1 = off; off += HID; L.be1 = off; off += HID;
  L.W2 = off; off += HID * HID;
  L.b2 = off; off += HID; L.g2 = off; off += HID; L.be2 = off; off += HID;
  L.Wh = off; off += nh * HID;
  L.bh = off; off += round_up(nh, 4);
  L.size = off;
  return L;
}

// reference (state_dict order, unpadded) <-> arena (padded) for one net
static int64_t ref_count(const NetLayout& L, int ln) {
  return (int64_t)HID * L.K + HID + (ln ? 2 * HID : 0) + (int64_t)HID * HID + HID + (ln ? 2 * HID : 0) + (int64_t)L.nh * HID + L.nh;
}
static void pack_net(const NetLayout& L, int ln, const float* src, float* dst, bool is_param) {
  std::fill(dst, dst + L.size, 0.f);
  if (is_param && !ln) {  // unused LN affine: identity
    std::fill(dst + L.g1, dst + L.g1 + HID, 1.f);
    std::fill(dst + L.g2, dst + L.g2 + HID, 1.f);
  }
  for (int r = 0; r < HID; ++r) { memcpy(dst + L.W1 + (size_t)r * L.ld1, src, sizeof(float) * L.K); src += L.K; }
  memcpy(dst + L.b1, src, sizeof(float) * HID); src += HID;
  if (ln) { memcpy(dst + L.g1, src, sizeof(float) * HID); src += HID; memcpy(dst + L.be1, src, sizeof(float) * HID); src += HID; }
  memcpy(dst + L.W2, src, sizeof(float) * HID * HID); src += HID * HID;
  memcpy(dst + L.b2, src, sizeof(float) * HID); src += HID;
  if (ln) { memcpy(dst + L.g2, src, sizeof(float) * HID); src += HID; memcpy(dst + L.be2, src, sizeof(float) * HID); src += HID; }
  memcpy(dst + L.Wh, src, sizeof(float) * L.nh * HID); src += (size_t)L.nh * HID;
  memcpy(dst + L.bh, src, sizeof(float) * L.nh);
}
static void unpack_net(const NetLayout& L, int ln, const float* src, float* dst) {
  for (int r = 0; r < HID; ++r) { memcpy(dst, src + L.W1 + (size_t)r * L.ld1, sizeof(float) * L.K); dst += L.K; }
  memcpy(dst, src + L.b1, sizeof(float) * HID); dst += HID;
  if (ln) { memcpy(dst, src + L.g1, sizeof(float) * HID); dst += HID; memcpy(dst, src + L.be1, sizeof(float) * HID); dst += HID; }
  memcpy(dst, src + L.W2, sizeof(float) * HID * HID); dst += HID * HID;
  memcpy(dst, src + L.b2, sizeof(float) * HID); dst += HID;
  if (ln) { memcpy(dst, src + L.g2, sizeof(float) * HID); dst += HID; memcpy(dst, src + L.be2, sizeof(float) * HID); dst += HID; }
  memcpy(dst, src + L.Wh, sizeof(float) * L.nh * HID); dst += (size_t)L.nh * HID;
  memcpy(dst, src + L.bh, sizeof(float) * L.nh);
}

static const int BIG_BATCH = 1024;   // from here on the hidden layers run as 64 x 64-tiled GEMMs + a LayerNorm row kernel
enum { G_Q = 0, G_A = 1, G_STEP00 = 2, G_STEP01 = 3, G_STEP10 = 4, G_STEP11 = 5, G_PERIOD = 6, G_PERIOD_B = 7, G_OPENING = 8,
       G_PREFIX = 9 /* + 2 (m - 1) + variant, m = 1, 2: the first m iterations of a period (sactd3_step_prefix) */, G_COUNT = 13 };
static const int NSTAGE = 32;

struct NodeInfo { std::string name; double flops; double bytes; long threads; };

struct sactd3_engine {
  sactd3_config cfg{};
  std::string err;
  hipStream_t stream = nullptr;
  std::vector<void*> dev_allocs, host_allocs;
  std::vector<hipEvent_t> events;

  int o = 0, a = 0, B = 0, ldc = 0, ldo = 0, a4 = 0, nh = 0, ldu = 0, rec_f = 0, rec4 = 0, cx = 0, cn = 0;
  int nq_actor = 2;            // critics evaluated in the actor update (SAC 2, TD3 1)
  int nblk = 0, nblk4 = 0;     // row-kernel blocks for B rows: 16 rows each (MFMA-using tails) / 4 rows each (plain row kernels)
  AlphaArgs pending_alpha{}; bool alpha_pending = false;   // a temperature step deferred into the next update's trunk launch (enqueue time only)
  bool alpha_tick_owed = false;   // ... deferred across an ITERATION boundary (period graphs): its counter tick is made up by the next critic update
  int maxn = 0;                // rows accepted by predict
  int num_cus = 256;
  int stage_rows = 0;
  NetLayout La{}, Lc{};

  DevCtl* ctl = nullptr;
  float *min_ac = nullptr, *max_ac = nullptr, *scale = nullptr, *bias = nullptr;
  float *Pa = nullptr, *Ta = nullptr, *Ga = nullptr, *Ma = nullptr, *Va = nullptr;
  float *Ta2 = nullptr, *Ta3 = nullptr;     // TD3 period graphs: the actor target one / two Polyak steps ahead (TnArgs::T2, T3)
  float *Pc = nullptr, *Tc = nullptr, *Gc = nullptr, *Mc = nullptr, *Vc = nullptr;
  float *la = nullptr, *gscale = nullptr;
  float* ring = nullptr; float* stage_dev = nullptr;
  float *X = nullptr, *Xn = nullptr, *Xp = nullptr, *rew = nullptr, *done = nullptr;
  int* idx = nullptr;
  float *logp_n = nullptr, *logp_pi = nullptr, *logp_al = nullptr, *act_scratch = nullptr;
  float* eps[SACTD3_NUM_SITES] = {};
  float *a_z1 = nullptr, *a_xh1 = nullptr, *a_h1 = nullptr, *a_rs1 = nullptr, *a_z2 = nullptr, *a_xh2 = nullptr, *a_h2 = nullptr, *a_rs2 = nullptr, *a_tg = nullptr;
  float *a_du = nullptr, *a_dz2 = nullptr, *a_dh1 = nullptr, *a_dz1 = nullptr;
  float *qa_ps = nullptr, *qa_S = nullptr;   // dQ/da partials of the fused actor update (QaFold): [nq][B][16][16], [nq][16][8]
  float *a_ps = nullptr, *c_ps = nullptr;     // row-sum partials of the folded layer-1 LayerNorm backward (TnProb::fold): [nets][B][PS_W]
  float* a_z2n = nullptr;        // layer-2 output of the s' pass when the pi(s) pass shares its launch
  float* ah_z1[4] = {}; float* ah_z2[4] = {};   // layer outputs of the passes that run ahead (pipelined period, see BatchSlot)
  // Batch slots.  Slot 0 is THE batch slot (X, Xn, rew, done, idx, logp_n, eps of the critic site above).  A pipelined period graph
  // (sactd3_step_period, SAC) samples and runs the next-action pass of its critic-only iterations AHEAD, inside the last actor
  // update's launches -- the actor does not change in between -- into slots 1 and 2, which those iterations then train on; and it
  // does the same for the opening pair of the NEXT period's first iteration (next-action pass on s' and the first actor update's
  // policy pass on s), whose slot alternates between 0 and 3 from one period to the next (the running period still reads its own).
  struct BatchSlot { float *X, *Xn, *rew, *done, *logp_n, *eps_c; int* idx; } bs[4] = {};
  int cur_slot = 0;              // the slot the most recent iteration trained on (what read_batch / read_noise / debug_read report)
  // chain_ready = v (0 / 1): the previous sactd3_step_period left the opening pair of the next period precomputed in slot (v ? 3 : 0)
  // and nothing has touched the state it depends on since (parameters, ring length, counters, noise injection, the slots);
  // -1: not so -- the next period starts with the opening graph.  Every state-changing ABI call resets it (CHAIN_BREAK).
  int chain_ready = -1;
  float *c_z1 = nullptr, *c_xh1 = nullptr, *c_h1 = nullptr, *c_rs1 = nullptr, *c_z2 = nullptr, *c_dz2 = nullptr, *c_dh1 = nullptr, *c_dz1 = nullptr;
  float *t_z1 = nullptr, *t_z2 = nullptr, *q = nullptr, *qt = nullptr, *y = nullptr, *q_pi = nullptr, *dA = nullptr;
  float* s_h1 = nullptr;         // large-batch path: layer-1 activations of nets whose caller keeps no copy ([4][B][256])
  float* Gp = nullptr; int gp_slabs = 0;      // large-batch path: split-M partial slabs of the weight gradients ([slab][nets][arena])
  float *part = nullptr, *part_s = nullptr, *part_sa = nullptr;   // column partials; scalar partials of the critic / actor updates
  float *p_x = nullptr, *p_z1 = nullptr, *p_z2 = nullptr, *p_act = nullptr;
  float *h_obs = nullptr, *h_act = nullptr;      // pinned predict staging
  int* h_done = nullptr; int predict_calls = 0;  // pinned completion word of the acting tail / calls issued (see ActorTail::done_flag)
  float* h_stage[NSTAGE] = {}; int stage_next = 0, stage_used = 0;   // pinned staging slots of rb_extend
  float* h_batch = nullptr;                      // pinned [B][rec_f]

  int64_t rb_len = 0, rb_cursor = 0, qnet_updates = 0;
  hipGraphExec_t graphs[G_COUNT] = {}; int graph_nodes[G_COUNT] = {};
  std::vector<hipGraphExec_t> predict_graphs;   // [explore][n]: the two launches of sactd3_predict, captured per row count
  // kernel-selection knobs: fixed defaults in the shipped library; tuning builds (-DSACTD3_TUNING) read them from the environment at create
  int tune_ks = 0, tune_nt = 0, tune_tn_kt = 0, tune_pad64 = 0, tune_tn64_min = 0, tune_rows4 = 0, tune_nn16 = 0, tune_xr = -1;
  // node registry of the enqueue_* sequences (sactd3_time_nodes): every kernel launch of the path goes through
  // node_on(), which numbers it; with node_only >= 0 only that launch is issued (the others are skipped), with
  // node_log set the launch's name and algorithmic FLOPs / bytes are recorded.
  int node_seq = 0, node_only = -1;
  const char* node_role = "";                 // which part of the iteration is being enqueued (names the nodes)
  std::vector<NodeInfo>* node_log = nullptr;

  int fail(int code, const char* what, hipError_t he = hipSuccess) {
    err = what;
    if (he != hipSuccess) { err += ": "; err += hipGetErrorString(he); }
    return code;
  }
};

#define HIPCHK(call)                                                   \
  do {                                                                 \
    hipError_t _he = (call);                                           \
    if (_he != hipSuccess) return e->fail(SACTD3_EHIP, #call, _he);    \
  } while (0)
// every entry point runs against the engine's own device, whatever the calling thread's current device is
#define USE_DEVICE(e)                                                                        \
  do {                                                                                       \
    hipError_t _he = hipSetDevice((e)->cfg.device_id);                                       \
    if (_he != hipSuccess) return (e)->fail(SACTD3_EHIP, "hipSetDevice", _he);               \
    (void)hipGetLastError();   /* a stale error of an unrelated earlier call must not be blamed on this one */ \
  } while (0)
// any call that changes what a precomputed opening pair depends on (see sactd3_engine::chain_ready)
#define CHAIN_BREAK(e) do { (e)->chain_ready = -1; } while (0)
#define RCCHK(call)                    \
  do {                                 \
    int _rc = (call);                  \
    if (_rc != 0) return _rc;          \
  } while (0)

template <class T>
static int dalloc(sactd3_engine* e, T** p, size_t count, bool zero = true) {
  void* v = nullptr;
  HIPCHK(hipMalloc(&v, std::max<size_t>(count, 4) * sizeof(T)));
  e->dev_allocs.push_back(v);
  if (zero) HIPCHK(hipMemset(v, 0, std::max<size_t>(count, 4) * sizeof(T)));
  *p = (T*)v;
  return 0;
}
template <class T>
static int halloc(sactd3_engine* e, T** p, size_t count) {
  void* v = nullptr;
  HIPCHK(hipHostMalloc(&v, std::max<size_t>(count, 4) * sizeof(T), hipHostMallocDefault));
  e->host_allocs.push_back(v);
  memset(v, 0, std::max<size_t>(count, 4) * sizeof(T));
  *p = (T*)v;
  return 0;
}

// ------------------------------------------------------------------------------------------------ launches
// Every kernel launch of the update path is numbered here (see sactd3_engine::node_seq).  flops = 2 x MACs of the GEMMs
// the launch contains (SURVEY.md 8d counts GEMM FLOPs only); bytes = the operands it has to read and the results it has
// to write, each counted once (what a perfect cache hierarchy would move).
// `name` = "<kernel instance as rocprofv3 prints it, without blanks>[.detail]"; logged as "instance:role[/detail]" with
// the launch's total thread count (what rocprofv3 calls Grid_Size), so that a profile row can be matched to a node.
static inline bool node_on(sactd3_engine* e, const char* name, double flops, double bytes, dim3 grid, dim3 block) {
  if (e->node_only < 0 && !e->node_log) return true;      // the normal case: not being timed
  const int k = e->node_seq++;
  if (e->node_log) {
    std::string n(name), detail;
    const size_t dot = n.find('.');
    if (dot != std::string::npos) { detail = "/" + n.substr(dot + 1); n.resize(dot); }
    e->node_log->push_back(NodeInfo{n + ":" + e->node_role + detail, flops, bytes,
                                    (long)grid.x * grid.y * grid.z * block.x * block.y * block.z});
  }
  return e->node_only < 0 || e->node_only == k;
}
#define LAUNCH_DYN(name, flops, bytes, kernel, grid, block, dyn_lds, ...)    \
  do {                                                                      \
    if (node_on(e, name, flops, bytes, grid, block)) {                      \
      hipLaunchKernelGGL(kernel, grid, block, dyn_lds, s, __VA_ARGS__);     \
      HIPCHK(hipGetLastError());                                            \
    }                                                                       \
  } while (0)
#define LAUNCH(name, flops, bytes, kernel, grid, block, ...) LAUNCH_DYN(name, flops, bytes, kernel, grid, block, 0, __VA_ARGS__)

static inline dim3 tile_grid(int tiles, int nets) { return dim3((unsigned)((tiles + 3) / 4), 1, (unsigned)nets); }

// one float4 chunk per thread while that still fills the chip; GATHER_CPT chunks (loads in flight) per thread beyond
static unsigned gather_blocks(long chunks) {
  const long one = (chunks + 255) / 256;
  return (unsigned)(one <= 4096 ? one : (chunks + 256L * GATHER_CPT - 1) / (256L * GATHER_CPT));
}
static GatherArgs gather_args(sactd3_engine* e, const float* ring, int identity_len, int slot = 0, int ctr_add = 0) {
  GatherArgs g{};
  const sactd3_engine::BatchSlot& S = e->bs[slot];
  g.ring = (const float4*)ring; g.rec4 = e->rec4; g.cx = e->cx; g.cn = e->cn; g.ctl = e->ctl; g.idx = S.idx;
  g.X = (float4*)S.X; g.Xn = (float4*)S.Xn; g.rew = S.rew; g.done = S.done;
  g.B = e->B; g.len_override = identity_len; g.ctr_add = ctr_add;
  g.rec4_magic = magic_div((unsigned)e->rec4, (unsigned long long)e->B * e->rec4 + 1);
  const long chunks = (long)e->B * e->rec4;
  g.cpb = (int)((chunks + 256L * gather_blocks(chunks) - 1) / (256L * gather_blocks(chunks)));
  return g;
}
template <int PRO, bool F1, int C1>
static void launch_nt_ks(hipStream_t s, int ks, dim3 grid, const NtArgs& g) {
  if (ks == 4) hipLaunchKernelGGL((k_nt<PRO, F1, 4, C1>), grid, dim3(256), 0, s, g);
  else if (ks == 2) hipLaunchKernelGGL((k_nt<PRO, F1, 2, C1>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((k_nt<PRO, F1, 1, C1>), grid, dim3(256), 0, s, g);
}
template <int PRO>
static void launch_nt_f1(hipStream_t s, int ks, int nt, dim3 grid, const NtArgs& g) {
  const int c1 = (g.K1 + 15) / 16;
  if (nt == 2) {   // (KS == 2 only) two column tiles per block
    if (c1 <= 1) hipLaunchKernelGGL((k_nt<PRO, true, 2, 1, 2>), grid, dim3(256), 0, s, g);
    else if (c1 == 2) hipLaunchKernelGGL((k_nt<PRO, true, 2, 2, 2>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_nt<PRO, true, 2, 4, 2>), grid, dim3(256), 0, s, g);
    return;
  }
  if (c1 <= 1) launch_nt_ks<PRO, true, 1>(s, ks, grid, g);
  else if (c1 == 2) launch_nt_ks<PRO, true, 2>(s, ks, grid, g);
  else launch_nt_ks<PRO, true, 4>(s, ks, grid, g);
}
static void launch_nt_c4(hipStream_t s, int nt, dim3 grid, const NtArgs& g) {      // (LayerNorm form, fused first layer, KS = 2 with KS = 4's sums)
  const int c1 = (g.K1 + 15) / 16;
  if (nt == 2) {
    if (c1 <= 1) hipLaunchKernelGGL((k_nt<1, true, 2, 1, 2, true>), grid, dim3(256), 0, s, g);
    else if (c1 == 2) hipLaunchKernelGGL((k_nt<1, true, 2, 2, 2, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_nt<1, true, 2, 4, 2, true>), grid, dim3(256), 0, s, g);
  } else {
    if (c1 <= 1) hipLaunchKernelGGL((k_nt<1, true, 2, 1, 1, true>), grid, dim3(256), 0, s, g);
    else if (c1 == 2) hipLaunchKernelGGL((k_nt<1, true, 2, 2, 1, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_nt<1, true, 2, 4, 1, true>), grid, dim3(256), 0, s, g);
  }
}
// XCD row groups for xcd_tile (kernels.h) of an R x C tile grid whose row operand is A bytes and column operand W bytes: the
// split xr x xc = 8 with the least total fetch xc A + xr W among those that divide the grid; 0 = keep row-major numbering.
static int pick_xr(const sactd3_engine* e, int R, int C, double A, double W) {
  if (e->tune_xr >= 0) return (e->tune_xr == 0 || (R % e->tune_xr == 0 && C % (8 / e->tune_xr) == 0)) ? e->tune_xr : 0;
  int best = 0; double cost = 1e300;
  for (int xr = 1; xr <= 8; xr *= 2) {
    const int xc = 8 / xr;
    if (R % xr || C % xc) continue;
    const double c = xc * A + xr * W;
    if (c < cost) { cost = c; best = xr; }
  }
  return best;
}
// pro == 0: the generic-K form (unfused first layer).  Otherwise K == 256 and the block shape (16 / 32 / 64 rows x 16
// columns) is chosen so that the launch has about one block per CU.
static int launch_nt(sactd3_engine* e, hipStream_t s, const char* name, int pro, bool fuse1, const NtArgs& g, int nets, int force_ks = 0) {
  const int tiles_m = (g.M + 15) / 16, tiles_n = (g.N + 15) / 16;
  // algorithmic work: the (fused) first layer + this layer; operands: input rows, the weight blocks, the output (+ stored activations)
  const double fl = 2.0 * nets * (double)g.M * g.N * (g.K + (fuse1 ? g.K1 : 0));
  double by = 4.0 * ((double)nets * g.N * (g.K + 3) + (double)nets * g.M * g.N);
  by += fuse1 ? 4.0 * ((double)nets * HID * (g.K1 + 1) + (double)(nets / g.npg) * g.M * g.K1) : 4.0 * (double)nets * g.M * g.K;
  for (int i = 0; i < nets / g.npg; ++i) by += 4.0 * g.npg * (double)g.M * HID * ((g.g[i].xh_out ? 1 : 0) + (g.g[i].h_out ? 1 : 0));
  if (g.gblocks) by += (g.gblocks / std::max(g.gb_each, 1)) * 8.0 * (double)g.ga[0].B * 4 * (g.ga[0].cx + g.ga[0].cn + 1);
  if (pro == 0) {
    const dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)nets);
    if (!node_on(e, "k_nt_wide.layer1", fl, by, grid, dim3(256))) return 0;
    hipLaunchKernelGGL(k_nt_wide, grid, dim3(256), 0, s, g);
  } else {
    const int tiles = tiles_m * tiles_n * nets;
    int ks = tiles >= 2 * e->num_cus ? 2 : 4;   // measured on 256 .. 4096-tile launches (KS = 1 never won)
    if (e->tune_ks) ks = e->tune_ks;
    if (force_ks) ks = force_ks;
    // a launch that must reproduce the KS = 4 sums (force_ks: the run-ahead passes of a period graph) but holds several groups: 32-row
    // blocks with the KS = 4 summation tree (k_nt's C4 form) -- half as many blocks fetch W1 and their W2 tile
    const bool c4 = force_ks == 4 && fuse1 && pro == 1 && tiles >= 2 * e->num_cus && !(e->tune_rows4 & 32768);
    if (c4) ks = 2;
    const int rb = 64 / ks;
    // two column tiles per block when the launch would otherwise put two rounds of blocks on every CU: the fused first
    // layer is then recomputed (or the A rows fetched and normalised) by half as many blocks
    int nt = (ks == 2 && ((g.M + rb - 1) / rb) * tiles_n * nets >= 2 * e->num_cus && tiles_n % 2 == 0) ? 2 : 1;
    if (e->tune_nt == 1 || (force_ks && !c4)) nt = 1;
    // (C4: two column tiles per block only when that leaves at most one block per CU: 4 groups -> 256 blocks; 5 groups would be 320,
    //  a quarter of the CUs with two -- they take 640 single-tile blocks instead)
    if (c4 && nt == 2 && ((g.M + rb - 1) / rb) * (tiles_n / 2) * nets > e->num_cus) nt = 1;
    NtArgs gg = g;
    gg.nt_blocks = ((g.M + rb - 1) / rb) * (tiles_n / nt);
    // unfused launches read whole input rows: place the tiles so that an XCD pulls few rows and few weight columns (the fused
    // form's input rows are a few dozen bytes: it keeps one weight column tile per XCD, the row-major numbering)
    gg.xr = fuse1 ? 0 : pick_xr(e, (g.M + rb - 1) / rb, tiles_n / nt, 4.0 * g.M * g.K, 4.0 * g.N * g.K);
    int nzb = 0;
    for (int i = 0; i < gg.nz_n && i < 5; ++i) nzb += gg.nz[i].blocks;
    const int riders = gg.gblocks + gg.alpha_block + nzb;
    // (see NtArgs::flat; with many riders -- the run-ahead launches' gathers and noise blocks -- the 3-D grid's order, net 0's tiles,
    //  the riders, then the other nets' tiles, measured 0.3 us per TD3 iteration better; TD3's 4-net critic trunk, one block per CU by
    //  its LDS, with its 1-3 noise riders: 0.5 us better in the 3-D grid too)
    gg.flat = (nets > 1 && gg.alpha_block && riders <= 8 && !(e->tune_rows4 & 65536)) ? nets : 0;      // (the critic trunk that carries a deferred temperature step)
    gg.flat_r = gg.flat ? (riders + 7) & ~7 : 0; gg.flat_n = riders;
    const dim3 grid = gg.flat ? dim3((unsigned)(gg.nt_blocks * nets + gg.flat_r)) : dim3((unsigned)(gg.nt_blocks + riders), 1, (unsigned)nets);
    char inst[64] = "k_nt";
    if (e->node_log) {
      const int c1 = (g.K1 + 15) / 16;
      snprintf(inst, sizeof(inst), c4 ? "k_nt<%d,%s,%d,%d,%d,true>.%s" : "k_nt<%d,%s,%d,%d,%d>.%s", pro, fuse1 ? "true" : "false", ks, fuse1 ? (c1 <= 1 ? 1 : (c1 == 2 ? 2 : 4)) : 0, nt, name);
    }
    if (!node_on(e, inst, fl, by, grid, dim3(256))) return 0;
    if (c4) launch_nt_c4(s, nt, grid, gg);
    else if (fuse1) { if (pro == 1) launch_nt_f1<1>(s, ks, nt, grid, gg); else launch_nt_f1<2>(s, ks, nt, grid, gg); }
    else if (nt == 2) {   // (KS == 2) the A rows are fetched and normalised by half as many blocks
      if (pro == 1) hipLaunchKernelGGL((k_nt<1, false, 2, 0, 2>), grid, dim3(256), 0, s, gg);
      else hipLaunchKernelGGL((k_nt<2, false, 2, 0, 2>), grid, dim3(256), 0, s, gg);
    } else { if (pro == 1) launch_nt_ks<1, false, 0>(s, ks, grid, gg); else launch_nt_ks<2, false, 0>(s, ks, grid, gg); }
  }
  HIPCHK(hipGetLastError());
  return 0;
}
static int launch_nn(sactd3_engine* e, hipStream_t s, const char* name, const NnArgs& g, int nets) {
  if (g.M >= BIG_BATCH && g.Kout == HID && g.k_off == 0 && !e->tune_nn16) {   // large batch: LDS-tiled form, ~1 block per CU
    const double fl = 2.0 * nets * (double)g.M * HID * HID, by = 4.0 * nets * (2.0 * g.M * HID + (double)HID * HID);
    if (nets >= 2) {
      const dim3 grid((unsigned)(((g.M + 31) / 32) * (HID / 64) * nets));
      LAUNCH("k_nn64<2,2,2>.dh1", fl, by, (k_nn64<2, 2, 2>), grid, dim3(256), g);
    } else {
      const dim3 grid((unsigned)(((g.M + 31) / 32) * (HID / 32) * nets));
      LAUNCH("k_nn64<2,2,1>.dh1", fl, by, (k_nn64<2, 2, 1>), grid, dim3(256), g);
    }
    (void)name;
    return 0;
  }
  const dim3 grid((unsigned)(((g.M + 15) / 16) * ((g.Kout + 15) / 16)), 1, (unsigned)nets);
  NnArgs gg = g;
  gg.xr = pick_xr(e, (g.M + 15) / 16, (g.Kout + 15) / 16, 4.0 * g.M * HID, 4.0 * HID * g.Kout);
  LAUNCH(name, 2.0 * nets * (double)g.M * HID * g.Kout, 4.0 * nets * ((double)g.M * HID + (double)HID * g.Kout + (double)g.M * g.Kout),
         k_nn, grid, dim3(256), gg);
  return 0;
}
// block tile / chunk rows of the large-batch weight-gradient GEMM (must agree with the k_tn64 instance launched)
#ifndef TN64_CFG
#define TN64_CFG 2, 2, 2, 1, 64      /* 64 (n) x 32 (k) tile, 64-row chunks: the fastest of tools/kprobe_tn64.hip's shapes */
#define TN64_N 64
#define TN64_K 32
#define TN64_M 64
#endif
#define TN64_KERNEL (k_tn64<TN64_CFG>)
// Large batches: split-M GEMM into partial slabs (k_tn64) + slab sum / vector gradients / Adam / Polyak (k_adam_red).
static int launch_tn64(sactd3_engine* e, hipStream_t s, const char* name, const TnArgs& g, int nets, int tick_extra = 0) {
  Tn64Args a{};
  a.nprob = g.nprob; a.M = g.M; a.nets = nets; a.Gp = e->Gp; a.g_ns = nets > 1 ? g.g_ns : (long)e->La.size;
  AdamRedArgs r{};
  r.s_off = -1;
  int tiles = 0;
  double fl = 0.0, by = 0.0;
  for (int i = 0; i < g.nprob; ++i) {
    const TnProb& q = g.pr[i];
    Tn64Prob& t = a.pr[i];
    t.dY = q.dY; t.ldy = q.ldy; t.dy_ns = q.dy_ns; t.N = q.N; t.X = q.X; t.ldx = q.ldx; t.x_ns = q.x_ns; t.K = q.K;
    t.w_off = q.w_off; t.ldw = q.ldw; t.b_off = q.b_off;
    t.tiles_n = (q.N + TN64_N - 1) / TN64_N; t.tiles_k = (q.ldw + TN64_K - 1) / TN64_K; t.tile0 = tiles;
    tiles += t.tiles_n * t.tiles_k;
    for (int f = 0; f < q.nfin; ++f) { r.vec[r.nvec].off = q.fin_off[f]; r.vec[r.nvec].slot = q.fin_slot[f]; r.vec[r.nvec].nblk = q.fin_nblk[f]; ++r.nvec; }
    if (q.fin_s_off >= 0) { r.s_off = q.fin_s_off; r.s_nblk = q.fin_s_nblk; }
    fl += 2.0 * nets * (double)g.M * q.N * q.K;
    by += 4.0 * nets * (double)g.M * (q.N + q.K);
  }
  a.tiles_per_net = tiles;
  const int nch = (g.M + TN64_M - 1) / TN64_M;
  // slices: about two blocks per CU, all of (nearly) the same length
  int S = (2 * e->num_cus + tiles * nets / 2) / (tiles * nets);
  S = std::max(1, std::min(S, std::min(e->gp_slabs, nch)));
  a.S = S;
  const long size = a.g_ns;
  {
    const dim3 grid((unsigned)(tiles * nets * S));
    char inst[96];
    snprintf(inst, sizeof(inst), "k_tn64<2,2,2,1,64>.dW(split-M x%d)", S);   // (instance = TN64_CFG: the name rocprofv3 reports)
    LAUNCH(inst, fl, by + 4.0 * S * nets * (double)size, TN64_KERNEL, grid, dim3(256), a);
  }
  r.Gp = e->Gp; r.S = S; r.nets = nets; r.g_ns = size; r.G = g.G;
  r.apply = g.apply; r.P = g.P; r.Mo = g.Mo; r.Vo = g.Vo; r.T = g.T; r.tau = g.tau; r.adam = g.adam; r.b1 = g.b1; r.b2 = g.b2; r.eps = g.eps;
  r.part = g.part; r.pstride = g.pstride; r.part_s = g.part_s;
  r.loss_part = g.loss_part; r.loss_n = g.loss_n; r.loss_stride = g.loss_stride; r.loss_off = g.loss_off; r.loss_scale = g.loss_scale;
  r.loss_dst = g.loss_dst; r.tick = g.tick; r.tick_extra = tick_extra;
  const dim3 grid((unsigned)((size / 4 + 255) / 256 + 4 * r.nvec + 1), (unsigned)nets);
  LAUNCH(g.apply ? (g.T ? "k_adam_red.sum+adam+polyak" : "k_adam_red.sum+adam") : "k_adam_red.sum", 0.0,
         nets * (double)size * (4.0 * S + 4.0 + (g.apply ? 24.0 : 0.0) + (g.apply && g.T ? 8.0 : 0.0)), k_adam_red, grid, dim3(256), r);
  (void)name;
  return 0;
}

static inline int tn_width(const TnProb& q) { return q.kw > 0 ? q.kw : q.ldw; }
static int launch_tn(sactd3_engine* e, hipStream_t s, const char* name, TnArgs& g, int nets, int tick_extra = 0) {
  if (g.M >= BIG_BATCH && e->Gp && !e->tune_tn_kt) {
    // the split-M form pays an extra node (k_adam_red): taken when the launch has enough 64 x 32 tiles to fill the chip with
    // long slices (the critics' 168 at Humanoid: 25.6 -> 18.6 us); the actor's 88 tiles gain nothing (14.3 vs 14.4 us)
    int tiles = 0;
    for (int i = 0; i < g.nprob; ++i) tiles += ((g.pr[i].N + TN64_N - 1) / TN64_N) * ((g.pr[i].ldw + TN64_K - 1) / TN64_K);
    if (tiles * nets >= e->tune_tn64_min) return launch_tn64(e, s, name, g, nets, tick_extra);
  }
  auto count = [&](int kt) {
    int tiles = 0;
    for (int i = 0; i < g.nprob; ++i) {
      g.pr[i].tile0 = tiles;
      tiles += ((g.pr[i].N + 15) / 16) * (((tn_width(g.pr[i]) + 15) / 16 + kt - 1) / kt);
    }
    return tiles;
  };
  // two k tiles per block (one dY slice fetched and transposed for both) once single tiles would be more than two blocks per CU
  // (the critics' launch at B = 256: 544 blocks -> 288, -0.6 us per iteration)
  int kt = (count(1) * nets > 2 * e->num_cus && g.M < BIG_BATCH) ? 2 : 1;   // (measured: no gain with a thousand rows per tile)
  if (e->tune_tn_kt) kt = e->tune_tn_kt;
  const int tiles = count(kt);
  for (int i = 0; i < g.nprob; ++i)      // n tiles need dY columns, k tiles X columns (both M rows long)
    g.pr[i].xr = g.pr[i].xr_force ? g.pr[i].xr_force
                                  : pick_xr(e, (g.pr[i].N + 15) / 16, ((tn_width(g.pr[i]) + 15) / 16 + kt - 1) / kt, 4.0 * g.M * g.pr[i].N, 4.0 * g.M * tn_width(g.pr[i]));
  // dW = dY^T X of every problem; operands dY, X once each; the weight block's gradient written, and with the fused
  // optimiser step p, m, v read and written (+ the Polyak target): 4 (g) + 24 (Adam) + 8 (Polyak) bytes per parameter
  double fl = 0.0, by = 0.0;
  for (int i = 0; i < g.nprob; ++i) {
    fl += 2.0 * nets * (double)g.M * g.pr[i].N * g.pr[i].K;
    by += 4.0 * nets * (double)g.M * ((g.pr[i].x_dup == 2 ? 0 : g.pr[i].N) + (g.pr[i].x_dup == 1 ? 0 : g.pr[i].K)) + (4.0 + (g.apply ? 24.0 : 0.0) + (g.apply && g.T ? 8.0 : 0.0)) * nets * (double)g.pr[i].N * (g.pr[i].K + 1);
  }
  bool fold = false;
  for (int i = 0; i < g.nprob; ++i) fold = fold || g.pr[i].fold;
  char inst[96];
  snprintf(inst, sizeof(inst), fold ? "k_tn<%d,true>.%s" : "k_tn<%d>.%s", kt, name);
  g.tiles = tiles;
  if (g.pk_blocks) by += 12.0 * (g.pk.n0 + g.pk.n1);
  {   // what is not a GEMM tile goes to riding blocks (see TnArgs::fin)
    AdamRedArgs& r = g.fin;
    r = AdamRedArgs{};
    r.s_off = -1;
    for (int i = 0; i < g.nprob; ++i) {
      const TnProb& q = g.pr[i];
      for (int f = 0; f < q.nfin; ++f) { r.vec[r.nvec].off = q.fin_off[f]; r.vec[r.nvec].slot = q.fin_slot[f]; r.vec[r.nvec].nblk = q.fin_nblk[f]; ++r.nvec; }
      if (q.fin_s_off >= 0) { r.s_off = q.fin_s_off; r.s_nblk = q.fin_s_nblk; }
    }
    r.nets = nets; r.g_ns = g.g_ns; r.G = g.G;
    r.apply = g.apply; r.P = g.P; r.Mo = g.Mo; r.Vo = g.Vo; r.T = g.T; r.tau = g.tau; r.adam = g.adam; r.b1 = g.b1; r.b2 = g.b2; r.eps = g.eps;
    r.T2 = g.T2; r.T3 = g.T3;
    r.part = g.part; r.pstride = g.pstride; r.part_s = g.part_s;
    r.loss_part = g.loss_part; r.loss_n = g.loss_n; r.loss_stride = g.loss_stride; r.loss_off = g.loss_off; r.loss_scale = g.loss_scale;
    r.loss_dst = g.loss_dst; r.tick = g.tick; r.tick_extra = tick_extra;
    g.fin_blocks = 4 * r.nvec + 1;
  }
  const dim3 grid((unsigned)((tiles + g.pk_blocks + g.fin_blocks + (nets > 1 ? 7 : 0)) & (nets > 1 ? ~7 : ~0)), 1, (unsigned)nets);
  if (!node_on(e, inst, fl, by, grid, dim3(256))) return 0;
  if (fold) {
    if (kt == 2) hipLaunchKernelGGL((k_tn<2, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_tn<1, true>), grid, dim3(256), 0, s, g);
  } else if (kt == 2) hipLaunchKernelGGL(k_tn<2>, grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL(k_tn<1>, grid, dim3(256), 0, s, g);
  HIPCHK(hipGetLastError());
  return 0;
}
static TnProb tn_prob(const float* dY, int ldy, long dy_ns, int N, const float* X, int ldx, long x_ns, int K,
                      int w_off, int ldw, int b_off) {
  TnProb q{};
  q.dY = dY; q.ldy = ldy; q.dy_ns = dy_ns; q.N = N; q.X = X; q.ldx = ldx; q.x_ns = x_ns; q.K = K;
  q.w_off = w_off; q.ldw = ldw; q.b_off = b_off; q.nfin = 0; q.fin_s_off = -1;
  return q;
}
// Rows [n_lo, n_lo + n) of a weight-gradient problem as a problem of its own (the same GEMM, cut along dW's rows).  A launch with a
// folded layer-1 problem puts it BETWEEN two such pieces of the W2 problem: a launch of 300 blocks gives the first ~50 CUs a second
// block (ids 256 ..), and the folded blocks -- 80 KB of operands, the launch's long pole -- must not be among the first ~50 ids (the
// critics' net-0 ones were: 6.8 us against net 1's 6.1, tools/blocks_probe.py).  Vector finalisations stay with the first piece.
static TnProb tn_rows(const TnProb& q, int n_lo, int n) {
  TnProb r = q;
  r.dY = q.dY + n_lo; r.N = n; r.w_off = q.w_off + n_lo * q.ldw; r.b_off = q.b_off >= 0 ? q.b_off + n_lo : -1;
  if (n_lo) { r.nfin = 0; r.fin_s_off = -1; r.x_dup = 1; }
  return r;
}
// Columns [k_lo, k_lo + k) of a weight-gradient problem as a problem of its own (cut along dW's COLUMNS: every piece has all of the
// rows).  With xr_force = 8 the 8 XCDs each own two 16-row tiles of a piece -- the rows of W2 (and of its Polyak target) that the
// next trunk launch reads on that very XCD (its column tile t = rows 32 t .. 32 t + 31 sits on XCD t mod 8): the optimiser epilogue
// leaves them in the L2 that will want them.
static TnProb tn_cols(const TnProb& q, int k_lo, int k) {
  TnProb r = q;
  r.X = q.X + k_lo; r.K = k; r.kw = k; r.w_off = q.w_off + k_lo;
  if (k_lo) { r.b_off = -1; r.nfin = 0; r.fin_s_off = -1; r.x_dup = 2; }
  return r;
}
static void tn_fin(TnProb& q, int slot, int off, int nblk) { q.fin_slot[q.nfin] = slot; q.fin_off[q.nfin] = off; q.fin_nblk[q.nfin++] = nblk; }

// The two hidden layers of MLP trunks: z2 = relu(LN(x W1^T + b1)) W2^T + b2 for up to two groups of nets
// (a group = nets sharing an input and a parameter arena) in ONE launch.  One kernel when the input is narrow
// (first layer recomputed per output tile), two otherwise.  Optional stores of layer 1's xhat / h / rstd.
// ring: the group's rows are read from the replay ring (field offset ring_off) -- the sample drawn with sample_ctr + sctr_add
struct TrunkGrp { const float* x; const float* P; float* z1; float* z2; float* xh; float* h; float* rstd; int ring_off = 0;
                  bool ring = false; int sctr_add = 0; const int* ring_idx = nullptr; };
struct TrunkTicks { int* tick0; int* tick1; float* adam_out; double* adam_pw; float lr;
                    int ngather = 0; GatherArgs gather[3] = {};   // replay gathers into batch slots riding as extra blocks (launches with ring groups)
                    const AlphaArgs* alpha = nullptr;      // alpha: a pending temperature step to carry as one extra block
                    int nnoise = 0; NoiseJob noise[5] = {}; bool* noise_taken = nullptr;      // the following tails' draws (see NoiseJob)
                    int force_ks = 0;                      // keep the single-net launch's K split (bit-equal results across launch shapes)
                    bool no_tiled64 = false;               // keep the 32 x 32-tile launches (the ones that can read ring rows / carry gathers)
                    int* tick0b = nullptr; float* adam_out_b = nullptr; double* adam_pw_b = nullptr; float lr_b = 0.f; };   // a second step counter
static int enqueue_trunk(sactd3_engine* e, hipStream_t s, int ldx, int K, int M, const NetLayout& L, long p_ns,
                         int ngrp, int npg, const TrunkGrp* grp, TrunkTicks tk) {
  const int pro = e->cfg.layer_norm ? 1 : 2;
  NtArgs h{};
  h.npg = npg; h.oW = L.W2; h.ldw = HID; h.oBias = L.b2; h.oG = L.g1; h.oBe = L.be1; h.p_ns = p_ns;
  h.ldy = HID; h.y_ns = (long)M * HID; h.M = M; h.N = HID; h.K = HID; h.act_ns = (long)M * HID;
  for (int i = 0; i < ngrp; ++i) {
    h.g[i].P = grp[i].P; h.g[i].Y = grp[i].z2; h.g[i].xh_out = grp[i].xh; h.g[i].h_out = grp[i].h; h.g[i].rstd_out = grp[i].rstd;
  }
  const int nets = ngrp * npg;
  const bool big_path = M >= BIG_BATCH && M == e->B && ((M + 63) / 64) * (HID / 64) * nets >= (3 * e->num_cus) / 4 && !tk.no_tiled64;
  auto set_ring = [&](NtArgs& a) {   // which groups read their rows from the replay ring, and the gathers that ride along
    for (int i = 0; i < ngrp; ++i) { a.g[i].ring = grp[i].ring ? 1 : 0; a.g[i].ring_off = grp[i].ring_off; a.g[i].sctr_add = grp[i].sctr_add; a.g[i].ring_idx = grp[i].ring_idx; }
    a.ga[0] = gather_args(e, e->ring, -1);                                     // (ring groups take the ring / control block from ga[0])
    for (int i = 0; i < tk.ngather && i < 3; ++i) a.ga[i] = tk.gather[i];
    a.gb_each = tk.ngather > 0 ? (int)gather_blocks((long)e->B * e->rec4) : 0;
    a.gblocks = tk.ngather * a.gb_each;
  };
  auto set_noise = [&](NtArgs& a) {  // the draws of the tail(s) that follow ride in this launch as a few extra blocks
    if (tk.nnoise <= 0) return;
    a.nz_n = tk.nnoise; a.nz_ctl = e->ctl;
    for (int i = 0; i < tk.nnoise; ++i) { a.nz[i] = tk.noise[i]; a.nz[i].blocks = (tk.noise[i].n + 1023) / 1024; }
    if (tk.noise_taken) *tk.noise_taken = true;
  };
  // MFMA-bound sizes with enough 64 x 64 tiles to fill the chip: tiled GEMM -> LayerNorm row kernel -> tiled GEMM
  if (big_path) {
    NtArgs g{};
    g.npg = npg; g.oW = L.W1; g.ldw = L.ld1; g.oBias = L.b1; g.p_ns = p_ns; g.ld_in = ldx; g.in_ns = 0;
    g.ldy = HID; g.y_ns = (long)M * HID; g.M = M; g.N = HID; g.K = K;
    for (int i = 0; i < ngrp; ++i) { g.g[i].in = grp[i].x; g.g[i].P = grp[i].P; g.g[i].Y = grp[i].z1; }
    g.tick0 = tk.tick0; g.tick1 = tk.tick1; g.adam_out = tk.adam_out; g.adam_pw = tk.adam_pw; g.lr = tk.lr; g.b1 = e->cfg.adam_beta1; g.b2 = e->cfg.adam_beta2;
    g.tick0b = tk.tick0b; g.adam_out_b = tk.adam_out_b; g.adam_pw_b = tk.adam_pw_b; g.lr_b = tk.lr_b;
    const dim3 grid((unsigned)(((M + 63) / 64) * (HID / 64) * nets));
    const double by_w = 4.0 * nets * (double)HID, by_rows = 4.0 * nets * (double)M * HID;
    {   // a pending temperature step rides as one extra block of the first-layer launch (nothing in the trunk reads log_alpha)
      dim3 grid1 = grid;
      if (tk.alpha) { g.alpha_block = 1; g.al = *tk.alpha; g.nt_blocks = (int)grid.x; grid1.x += 1; }
      LAUNCH_DYN("k_nt64<4,2,2>.layer1", 2.0 * nets * (double)M * HID * K, by_w * (K + 1) + 4.0 * ngrp * (double)M * K + by_rows,
                 (k_nt64<4, 2, 2>), grid1, dim3(512), e->tune_pad64, g);
    }
    if (e->tune_rows4 & 128) {   // (A/B: the separate LayerNorm row kernel between the two tiled GEMMs)
      LnFwd l{};
      l.npg = npg; l.oG = L.g1; l.oBe = L.be1; l.p_ns = p_ns; l.B = M; l.ln = e->cfg.layer_norm;
      for (int i = 0; i < ngrp; ++i) {
        l.P[i] = grp[i].P; l.zin[i] = grp[i].z1;
        l.h[i] = grp[i].h ? grp[i].h : e->s_h1 + (size_t)i * npg * M * HID;
        l.xh[i] = grp[i].xh; l.rstd[i] = grp[i].rstd;
      }
      {
        double st = 1.0;   // rows written: h always, xhat where the caller keeps it
        for (int i = 0; i < ngrp; ++i) st += grp[i].xh ? 1.0 / ngrp : 0.0;
        LAUNCH("k_ln_fwd", 0.0, by_rows * (1.0 + st), k_ln_fwd, dim3((unsigned)((M + 15) / 16), (unsigned)nets), dim3(256), l);
      }
      NtArgs h2{};
      h2.npg = npg; h2.oW = L.W2; h2.ldw = HID; h2.oBias = L.b2; h2.p_ns = p_ns; h2.ld_in = HID; h2.in_ns = (long)M * HID;
      h2.ldy = HID; h2.y_ns = (long)M * HID; h2.M = M; h2.N = HID; h2.K = HID;
      for (int i = 0; i < ngrp; ++i) { h2.g[i].in = l.h[i]; h2.g[i].P = grp[i].P; h2.g[i].Y = grp[i].z2; }
      LAUNCH_DYN("k_nt64<4,2,2>.layer2", 2.0 * nets * (double)M * HID * HID, by_w * (HID + 1) + 2.0 * by_rows, (k_nt64<4, 2, 2>), grid, dim3(512), e->tune_pad64, h2);
      return 0;
    }
    // second layer with the LayerNorm + ReLU of the first as its prologue (k_nt64_ln): no row kernel in between
    NtArgs h2{};
    h2.npg = npg; h2.oW = L.W2; h2.ldw = HID; h2.oBias = L.b2; h2.oG = L.g1; h2.oBe = L.be1; h2.p_ns = p_ns; h2.ld_in = HID; h2.in_ns = (long)M * HID;
    h2.ldy = HID; h2.y_ns = (long)M * HID; h2.M = M; h2.N = HID; h2.K = HID; h2.act_ns = (long)M * HID; h2.ln_pro = e->cfg.layer_norm ? 1 : 0;
    for (int i = 0; i < ngrp; ++i) {
      h2.g[i].in = grp[i].z1; h2.g[i].P = grp[i].P; h2.g[i].Y = grp[i].z2;
      h2.g[i].h_out = grp[i].h; h2.g[i].xh_out = grp[i].xh; h2.g[i].rstd_out = grp[i].rstd;
    }
    {
      double st = 0.0;   // rows written besides the output: h and xhat where the caller keeps them
      for (int i = 0; i < ngrp; ++i) st += ((grp[i].xh ? 1.0 : 0.0) + (grp[i].h ? 1.0 : 0.0)) / ngrp;
      LAUNCH_DYN("k_nt64_ln<4,2,2>.layer2", 2.0 * nets * (double)M * HID * HID, by_w * (HID + 3) + by_rows * (2.0 + st), (k_nt64_ln<4, 2, 2>), grid, dim3(512), e->tune_pad64, h2);
    }
    return 0;
  }
  if (K <= 64) {
    for (int i = 0; i < ngrp; ++i) h.g[i].in = grp[i].x;
    h.ld_in = ldx; h.in_ns = 0; h.K1 = K; h.oW1 = L.W1; h.ldw1 = L.ld1; h.oB1 = L.b1;
    h.tick0 = tk.tick0; h.tick1 = tk.tick1; h.adam_out = tk.adam_out; h.adam_pw = tk.adam_pw; h.lr = tk.lr; h.b1 = e->cfg.adam_beta1; h.b2 = e->cfg.adam_beta2;
    h.w1_magic = magic_div((unsigned)L.ld1, 4u * HID * (unsigned)L.ld1);
    if (tk.alpha) { h.alpha_block = 1; h.al = *tk.alpha; }
    set_noise(h);
    set_ring(h);            // x = a field of the sampled records; extra blocks fill the batch slot(s) (see NtArgs)
    h.tick0b = tk.tick0b; h.adam_out_b = tk.adam_out_b; h.adam_pw_b = tk.adam_pw_b; h.lr_b = tk.lr_b;
    return launch_nt(e, s, "layers1+2", pro, true, h, nets, tk.force_ks);
  }
  NtArgs g{};
  g.npg = npg; g.oW = L.W1; g.ldw = L.ld1; g.oBias = L.b1; g.p_ns = p_ns; g.ld_in = ldx; g.in_ns = 0;
  g.ldy = HID; g.y_ns = (long)M * HID; g.M = M; g.N = HID; g.K = K;
  for (int i = 0; i < ngrp; ++i) { g.g[i].in = grp[i].x; g.g[i].P = grp[i].P; g.g[i].Y = grp[i].z1; }
  g.tick0 = tk.tick0; g.tick1 = tk.tick1; g.adam_out = tk.adam_out; g.adam_pw = tk.adam_pw; g.lr = tk.lr; g.b1 = e->cfg.adam_beta1; g.b2 = e->cfg.adam_beta2;
  g.tick0b = tk.tick0b; g.adam_out_b = tk.adam_out_b; g.adam_pw_b = tk.adam_pw_b; g.lr_b = tk.lr_b;
  if (M >= BIG_BATCH && M == e->B) {   // large batch, too few nets for 64 x 64 tiles to fill the chip: 32 x 32 LDS-tiled form
    unsigned nblk = (unsigned)(((M + 31) / 32) * (HID / 32) * nets);
    set_ring(g);            // x = the field of the sampled records itself; extra blocks fill the batch slot(s) (as in the fused k_nt form)
    g.nt_blocks = (int)nblk;
    nblk += (unsigned)g.gblocks;
    const dim3 grid(nblk);
    LAUNCH("k_nt64<2,2,1>.layer1", 2.0 * nets * (double)M * HID * K, 4.0 * (nets * (double)HID * (K + 1) + ngrp * (double)M * K + nets * (double)M * HID),
           (k_nt64<2, 2, 1>), grid, dim3(256), g);
  } else RCCHK(launch_nt(e, s, "layer1", 0, false, g, nets));
  for (int i = 0; i < ngrp; ++i) h.g[i].in = grp[i].z1;
  h.ld_in = HID; h.in_ns = (long)M * HID;
  if (tk.alpha) { h.alpha_block = 1; h.al = *tk.alpha; }   // (never together with noise blocks: those read the counter it ticks)
  set_noise(h);            // (as in the fused form) the following tail's draws as extra blocks of the layer-2 launch
  if (M >= BIG_BATCH && M == e->B && !(e->tune_rows4 & 256)) {   // large batch: the 32 x 32 LDS-tiled form with the LayerNorm prologue
    h.ln_pro = e->cfg.layer_norm ? 1 : 0;
    h.nt_blocks = ((M + 31) / 32) * (HID / 32) * nets;
    int nzb = 0;
    for (int i = 0; i < h.nz_n && i < 5; ++i) nzb += h.nz[i].blocks;
    double st = 0.0;
    for (int i = 0; i < ngrp; ++i) st += ((grp[i].xh ? 1.0 : 0.0) + (grp[i].h ? 1.0 : 0.0)) / ngrp;
    LAUNCH("k_nt64_ln<2,2,1>.layer2", 2.0 * nets * (double)M * HID * HID, 4.0 * nets * ((double)HID * (HID + 3) + (double)M * HID * (2.0 + st)),
           (k_nt64_ln<2, 2, 1>), dim3((unsigned)(h.nt_blocks + h.alpha_block + nzb)), dim3(256), h);
    return 0;
  }
  return launch_nt(e, s, "layer2", pro, false, h, nets);
}

static ActorTail tail_args(sactd3_engine* e, const float* z2, const float* P, int M, int mode, int train,
                           int site_buf, unsigned site_code, float* dst, int ldd, int dst_off, float* logp) {
  ActorTail t{};
  t.z2 = z2; t.P = P; t.L = e->La; t.B = M; t.o = e->o; t.a = e->a; t.ln = e->cfg.layer_norm;
  t.sac = !e->cfg.prefer_td3_over_sac; t.mode = mode; t.train = train;
  t.ctl = e->ctl; t.ctr = site_buf == SACTD3_SITE_PREDICT ? &e->ctl->predict_ctr : &e->ctl->noise_ctr;
  t.site_buf = site_buf; t.site_code = site_code; t.eps = e->eps[site_buf];
  t.scale = e->scale; t.bias = e->bias; t.min_ac = e->min_ac; t.max_ac = e->max_ac;
  t.dst = dst; t.ldd = ldd; t.dst_off = dst_off; t.logp = logp;
  t.h2 = e->a_h2; t.xh2 = e->a_xh2; t.rstd2 = e->a_rs2; t.tg = e->a_tg; t.a4 = e->a4;
  t.td3_std = e->cfg.td3_std; t.td3_c = e->cfg.td3_c; t.noise_std = e->cfg.actor_noise_std;
  return t;
}
static NoiseJob noise_job(sactd3_engine* e, int site_buf, unsigned site_code, int ctr_add, int rows) {
  NoiseJob z{};
  z.eps = e->eps[site_buf]; z.ctr = site_buf == SACTD3_SITE_PREDICT ? &e->ctl->predict_ctr : &e->ctl->noise_ctr;
  z.ctr_add = ctr_add; z.site_code = site_code; z.site_buf = site_buf; z.n = rows * e->a;
  return z;
}
// rows of the batch one block of the actor-tail kernel handles (narrow heads: one wave per 4 rows, see k_actor_tail_s)
static int tail_rows_per_block(const ActorTail& t) { return (t.L.nh <= 8 && t.a <= 8) ? 4 : 16; }
static int launch_tail(sactd3_engine* e, hipStream_t s, const ActorTail& t) {
  const int nh = t.L.nh;
  if (tail_rows_per_block(t) == 4) {
    LAUNCH("k_actor_tail_s<4>", 2.0 * t.B * (double)HID * nh,
           4.0 * ((double)t.B * HID * (t.train ? 3 : 1) + (double)nh * (HID + 1) + 2.0 * HID + (double)t.B * (3 * t.a + 2) + (t.obs_src ? 2.0 * t.B * t.o : 0.0)),
           k_actor_tail_s<4>, dim3((t.B + 3) / 4), dim3(64), t);
    return 0;
  }
  LAUNCH("k_actor_tail", 2.0 * t.B * (double)HID * nh,
         4.0 * ((double)t.B * HID * (t.train ? 3 : 1) + (double)nh * (HID + 1) + 2.0 * HID + (double)t.B * (3 * t.a + 2) + (t.obs_src ? 2.0 * t.B * t.o : 0.0)),
         k_actor_tail, dim3((t.B + 15) / 16), dim3(256), t);
  return 0;
}

static int enqueue_gather(sactd3_engine* e, hipStream_t s, const float* ring, int identity_len) {
  const GatherArgs g = gather_args(e, ring, identity_len);
  // 2 B T + 4 B, T = 4 (2o + a + 1) + 1
  LAUNCH("k_gather", 0.0, 2.0 * e->B * (4.0 * (2 * e->o + e->a + 1) + 1.0) + 4.0 * e->B, k_gather, dim3(gather_blocks((long)e->B * e->rec4)), dim3(256), g);
  return 0;
}

static AdamArgs adam_args(sactd3_engine* e, float* p, const float* g, float* m, float* v, long n, const float* adam) {
  AdamArgs a{};
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.adam = adam;
  a.b1 = e->cfg.adam_beta1; a.b2 = e->cfg.adam_beta2; a.eps = e->cfg.adam_eps;
  return a;
}
static int launch_adam(sactd3_engine* e, hipStream_t s, const AdamArgs& a) {
  const int blocks = (int)std::min<long>(512, (a.n / 4 + 255) / 256);
  LAUNCH("k_adam", 0.0, 28.0 * a.n, k_adam, dim3(std::max(blocks, 1)), dim3(256), a);
  return 0;
}

// Can the actor trunk that opens an iteration (1 net, or 2 groups when the policy pass is merged in) carry a temperature step as an
// extra block?  The k_nt launches can (fused first layer, or the layer-2 launch of the layer-by-layer form); the tiled k_nt64 pair
// of launches with >= 3/4 of the chip in 64 x 64 tiles cannot (see enqueue_trunk).
static bool opening_trunk_carries_alpha(const sactd3_engine* e);
// Does the actor trunk that opens a fused iteration read its rows from the ring itself, with the gather into the batch slot riding in
// the same launch?  Narrow observations below the large-batch threshold (fused k_nt), and wide ones at large batch (k_nt64<2,2,1>).
static bool opening_trunk_gathers(const sactd3_engine* e) {
  if (e->tune_rows4 & 32) return e->o <= 64 && e->B < BIG_BATCH;
  return (e->o <= 64 && e->B < BIG_BATCH) || (e->o > 64 && e->B >= BIG_BATCH && opening_trunk_carries_alpha(e));
}
static bool opening_trunk_carries_alpha(const sactd3_engine* e) {
  const int B = e->B;
  const bool tiled = B >= BIG_BATCH && ((B + 63) / 64) * (HID / 64) * 2 >= (3 * e->num_cus) / 4;
  return !tiled;
}

// agents/agent.py:183-242
// fused_sample: this update opens a fused iteration and owns the replay sampling (orchestrator.py:338); with a narrow
// observation the gather rides in the first trunk kernel, otherwise enqueue_step has launched k_gather just before.
// with_policy: also run the first actor update's policy pass pi(s) in the opening launches (fused iterations with actor updates);
// *policy_done tells the caller whether that happened.
// actor_targ_rides: (TD3, fused iteration without actor updates) the actor target's Polyak update as extra blocks of the
// weight-gradient launch; *actor_targ_done reports whether the launch could carry it.
// slot / pre_sampled: (pipelined period, SAC) the batch slot this update trains on; pre_sampled = its sample, gather and next-action
// pass (a', log pi(a'|s')) were already produced ahead by the last actor update of the period's first iteration (enqueue_update_actor,
// `ahead`): the update starts at the twin-critic trunk, which then carries the step-counter tick and a deferred temperature step.
static int enqueue_update_qnets(sactd3_engine* e, hipStream_t s, bool fused_sample, float* fused_polyak_targ,
                                bool with_policy = false, bool* policy_done = nullptr, bool actor_targ_rides = false, bool* actor_targ_done = nullptr,
                                int slot = 0, bool pre_sampled = false, int open_mode = 0) {
  // open_mode (chained periods): 1 = ONLY the opening pair (trunk + tails incl. the first actor update's policy pass), without the
  // step-counter / sample-counter ticks, then return -- the "opening graph"; 2 = (with pre_sampled) this is the period's first
  // iteration whose opening pair was precomputed: the twin-critic trunk also makes those ticks (critics', actor's, sample counter)
  const sactd3_config& c = e->cfg;
  const int B = e->B, ln = c.layer_norm, td3 = c.prefer_td3_over_sac;
  const long BH = (long)B * HID;
  const sactd3_engine::BatchSlot& S = e->bs[slot];
  int ctr_owed = 0;
  // target action: SAC a' ~ pi(s') with the ONLINE actor (agent.py:205); TD3 pi_targ(s') + clipped noise (agent.py:194-200)
  const float* Pact = td3 ? e->Ta : e->Pa;
  if (!pre_sampled) {
    const bool in_kernel_gather = fused_sample && opening_trunk_gathers(e);
    // the FIRST actor update's pi(s) pass rides along (see enqueue_step): in the fused-first-layer launches of narrow observations,
    // and as a second group of the layer-by-layer launches of wide ones at large batch (Humanoid: two nodes fewer per actor iteration)
    const bool wide_merge = with_policy && fused_sample && e->o > 64 && B >= BIG_BATCH && !(e->tune_rows4 & 16);
    const bool merge_policy = with_policy && ((in_kernel_gather && e->o <= 64) || wide_merge);
    e->node_role = fused_sample ? (merge_policy ? "critic/next-action+sample & actor0/policy" : "critic/next-action+sample") : "critic/next-action";
    // (layer-by-layer launches materialise z1: the target-action group borrows the target critics' z1 slab, idle until the next launch)
    TrunkGrp g[2] = {{S.Xn, Pact, wide_merge ? e->t_z1 : e->a_z1, merge_policy ? e->a_z2n : e->a_z2, nullptr, nullptr, nullptr, e->ldc},
                     {S.X, e->Pa, e->a_z1, e->a_z2, e->a_xh1, e->a_h1, e->a_rs1, 0}};
    TrunkTicks tk{&e->ctl->t_q, (fused_sample && !in_kernel_gather) ? &e->ctl->sample_ctr : nullptr, e->ctl->adam_q, e->ctl->pw_q, c.qnets_lr};
    if (open_mode == 1) { tk.tick0 = nullptr; tk.tick1 = nullptr; tk.adam_out = nullptr; tk.adam_pw = nullptr; }
    if (in_kernel_gather) {   // the trunk reads its rows from the ring itself; the gather into the batch slot rides along
      tk.ngather = 1; tk.gather[0] = gather_args(e, e->ring, -1, slot);
      for (auto& gg : g) { gg.ring = true; gg.ring_idx = S.idx; }
    }
    const int mode = td3 ? (c.targ_actor_smoothing ? 1 : 0) : 0;
    bool eps_ready = false;
    if (!td3 || mode == 1) { tk.noise[tk.nnoise++] = noise_job(e, SACTD3_SITE_CRITIC, 0u, 0, B); tk.noise[tk.nnoise - 1].eps = S.eps_c; tk.noise_taken = &eps_ready; }
    // a temperature step deferred from the previous iteration of the same graph (sactd3_step_period) rides in this launch; its
    // tick of the noise counter is owed: this iteration's draws count one ahead and the critics' last kernel ticks by two
    int owed = 0;
    if (e->alpha_pending && e->alpha_tick_owed && fused_sample && opening_trunk_carries_alpha(e)) {
      tk.alpha = &e->pending_alpha; e->alpha_pending = false; e->alpha_tick_owed = false;
      owed = 1;
      for (int i = 0; i < tk.nnoise; ++i) tk.noise[i].ctr_add += 1;
    }
    ctr_owed = owed;
    if (merge_policy) {
      // the policy sample of the first actor update: same actor parameters (the critic update does not touch them), the stream
      // counter one ahead (the critic update's last kernel bumps it before the actor update would have read it)
      if (!td3) { tk.noise[tk.nnoise++] = noise_job(e, SACTD3_SITE_ACTOR0, 16u, 1 + owed, B); tk.noise_taken = &eps_ready; }
      if (!wide_merge) tk.force_ks = 4;
      if (open_mode != 1) { tk.tick0b = &e->ctl->t_a; tk.adam_out_b = e->ctl->adam_a; tk.adam_pw_b = e->ctl->pw_a; tk.lr_b = c.actor_lr; }
    }
    RCCHK(enqueue_trunk(e, s, e->ldc, e->o, B, e->La, 0, merge_policy ? 2 : 1, 1, g, tk));
    ActorTail t = tail_args(e, merge_policy ? e->a_z2n : e->a_z2, Pact, B, mode, 0, SACTD3_SITE_CRITIC, 0u, S.Xn, e->ldc, e->o, S.logp_n);
    t.eps = S.eps_c;
    t.eps_ready = eps_ready; t.ctr_add = owed;
    if (in_kernel_gather && open_mode != 1) t.tick = &e->ctl->sample_ctr;   // every reader of the index stream (the trunk kernel) is done
    if (merge_policy && tail_rows_per_block(t) == 4) {
      ActorTail t1 = tail_args(e, e->a_z2, e->Pa, B, 0, 1, SACTD3_SITE_ACTOR0, 16u, e->Xp, e->ldc, e->o, e->logp_pi);
      t1.obs_src = S.X; t1.lds = e->ldc; t1.ctr_add = 1 + owed; t1.eps_ready = eps_ready;     // Xp = [s | pi(s)]
      const int nb = (B + 3) / 4;
      LAUNCH("k_actor_tail_s2<4>", 2.0 * 2 * B * (double)HID * t.L.nh, 4.0 * ((double)B * HID * 4 + 2.0 * t.L.nh * (HID + 1) + 4.0 * HID + (double)B * (6 * e->a + 4 + 2 * e->o)),
             k_actor_tail_s2<4>, dim3(2 * nb), dim3(64), t, t1, nb);
      if (policy_done) *policy_done = true;
    } else if (merge_policy) {                  // (wide heads: the two tails as one launch of the general kernel)
      ActorTail t1 = tail_args(e, e->a_z2, e->Pa, B, 0, 1, SACTD3_SITE_ACTOR0, 16u, e->Xp, e->ldc, e->o, e->logp_pi);
      t1.obs_src = S.X; t1.lds = e->ldc; t1.ctr_add = 1 + owed; t1.eps_ready = eps_ready;
      const int nb = (B + 15) / 16;
      LAUNCH("k_actor_tail2", 2.0 * 2 * B * (double)HID * t.L.nh, 4.0 * ((double)B * HID * 4 + 2.0 * t.L.nh * (HID + 1) + 4.0 * HID + (double)B * (6 * e->a + 4 + 2 * e->o)),
             k_actor_tail2, dim3(2 * nb), dim3(256), t, t1, nb);
      if (policy_done) *policy_done = true;
    } else RCCHK(launch_tail(e, s, t));
    if (open_mode == 1) return (merge_policy && policy_done && *policy_done) ? 0 : e->fail(SACTD3_ESTATE, "opening graph: the policy pass did not merge");
  }
  {  // twin target critics on (s', a') and twin online critics on (s, a) in one launch (agent.py:208-210, 230-232).
     // (Measured: running the online pair on a fork/join side branch of the graph instead costs +30 us per replay on
     //  ROCm 7.2 -- cross-stream edges are far dearer than the 1.7 us of a linear edge -- so graphs stay linear.)
    const TrunkGrp g[2] = {{S.Xn, e->Tc, e->t_z1, e->t_z2, nullptr, nullptr, nullptr},
                           {S.X, e->Pc, e->c_z1, e->c_z2, e->c_xh1, e->c_h1, e->c_rs1}};
    e->node_role = "critic/twin-q(2 target + 2 online)";
    TrunkTicks tk{nullptr, nullptr, nullptr, nullptr, 0.f};
    if (pre_sampled) {   // this launch opens the update: the critics' step counter + Adam scalars, and a temperature step deferred from the
                         // period's first iteration (its tick of the noise counter is made up by this update's last kernel)
      tk.tick0 = &e->ctl->t_q; tk.adam_out = e->ctl->adam_q; tk.adam_pw = e->ctl->pw_q; tk.lr = c.qnets_lr;
      if (open_mode == 2) {   // ... the ticks the precomputed opening pair left undone: the sample counter and the actor's step counter
        tk.tick1 = &e->ctl->sample_ctr;
        tk.tick0b = &e->ctl->t_a; tk.adam_out_b = e->ctl->adam_a; tk.adam_pw_b = e->ctl->pw_a; tk.lr_b = c.actor_lr;
      }
      if (e->alpha_pending && e->alpha_tick_owed) { tk.alpha = &e->pending_alpha; e->alpha_pending = false; e->alpha_tick_owed = false; ctr_owed = 1; }
    }
    RCCHK(enqueue_trunk(e, s, e->ldc, e->o + e->a, B, e->Lc, e->Lc.size, 2, 2, g, tk));
  }
  e->node_role = "critic/loss+backward";
  bool fused_tail_nn = false, fold_ln1 = false;
  {
    CriticTail t{};
    t.z2t = e->t_z2; t.z2 = e->c_z2; t.PT = e->Tc; t.P = e->Pc; t.p_ns = e->Lc.size; t.L = e->Lc;
    t.rew = S.rew; t.done = S.done; t.logp_next = S.logp_n; t.log_alpha = e->la;
    t.B = B; t.ln = ln; t.sac = !td3; t.bcq = c.bcq_style_targ_mix; t.gamma = c.gamma;
    t.qt = e->qt; t.y = e->y; t.q = e->q; t.dz2 = e->c_dz2; t.part = e->part; t.part_s = e->part_s; t.pstride = e->nblk4;
    fused_tail_nn = B < BIG_BATCH && !(e->tune_rows4 & 2048);
    if (fused_tail_nn) {   // the tail AND dh1 = dz2 W2 in one launch (k_ctail_nn): 16-row blocks x 32-column tiles
      CtailNn f{};
      f.c = t; f.c.pstride = e->nblk4; f.Wt = e->Pc + e->Lc.W2; f.ldw = HID; f.dX = e->c_dh1;
      fold_ln1 = !(e->tune_rows4 & 4096);   // layer 1's LayerNorm backward inside the weight-gradient launch (TnProb::fold)
      f.f.fold = fold_ln1; f.f.ln = ln; f.f.h1 = e->c_h1; f.f.xh1 = e->c_xh1; f.f.g1_off = e->Lc.g1; f.f.ps = e->c_ps; f.f.gsnap = e->c_ps + 2L * B * PS_W;
      f.xr = pick_xr(e, e->nblk, HID / 32, 4.0 * 3 * B * HID, 4.0 * HID * HID);
      LAUNCH("k_ctail_nn<2>", 2.0 * 4 * B * (double)HID + 2.0 * 2 * (double)B * HID * HID, 4.0 * (6.0 * BH + 4.0 * 4 * HID + 8.0 * B) + 4.0 * 2 * ((double)HID * HID + (double)B * HID),
             k_ctail_nn<2>, dim3((unsigned)(e->nblk * (HID / 32)), 1, 2), dim3(256), f);
    } else if (e->tune_rows4 & 1) LAUNCH("k_critic_tail<4>", 2.0 * 4 * B * (double)HID, 4.0 * (6.0 * BH + 4.0 * 4 * HID + 8.0 * B), k_critic_tail<4>, dim3(e->nblk4, 2), dim3(64), t);
    else LAUNCH("k_critic_tail<16>", 2.0 * 4 * B * (double)HID, 4.0 * (6.0 * BH + 4.0 * 4 * HID + 8.0 * B), k_critic_tail<16>, dim3(e->nblk, 2), dim3(256), t);
  }
  if (!fused_tail_nn) {  // dh1 = dz2 W2
    NnArgs g{};
    g.dY = e->c_dz2; g.dy_ns = BH; g.Wt = e->Pc + e->Lc.W2; g.ldw = HID; g.p_ns = e->Lc.size; g.k_off = 0;
    g.dX = e->c_dh1; g.ldx = HID; g.dx_ns = BH; g.M = B; g.Kout = HID;
    RCCHK(launch_nn(e, s, "k_nn.dh1", g, 2));
  }
  if (!fold_ln1) {
    LnBwd l{};
    l.dh = e->c_dh1; l.xh = e->c_xh1; l.h = e->c_h1; l.rstd = e->c_rs1;
    l.gamma = e->Pc + e->Lc.g1; l.p_ns = e->Lc.size; l.B = B; l.ln = ln; l.want_part = ln;
    l.dz = e->c_dz1; l.part = e->part; l.pstride = e->nblk4;
    if (e->tune_rows4 & 2) LAUNCH("k_ln_bwd<4>", 0.0, 4.0 * 2 * (4.0 * BH + B + HID), k_ln_bwd<4>, dim3(e->nblk4, 2), dim3(64), l);
    else LAUNCH("k_ln_bwd<16>", 0.0, 4.0 * 2 * (4.0 * BH + B + HID), k_ln_bwd<16>, dim3(e->nblk, 2), dim3(256), l);
  }
  {  // every critic gradient + the Adam step (+ Polyak) in one launch:
     //   dW2 = dz2^T h1, db2, dgamma2, dbeta2, dWhead, dbhead ; dW1 = dz1^T [s|a], db1, dgamma1, dbeta1
    TnArgs g{};
    g.nprob = 2; g.M = B; g.G = e->Gc; g.g_ns = e->Lc.size;
    const int i2 = 0, i1 = fold_ln1 ? 2 : 1;           // folded: [W2 rows 0 .. 127][layer 1][W2 rows 128 .. 255] (tn_rows)
    g.pr[i2] = tn_prob(e->c_dz2, HID, BH, HID, e->c_h1, HID, BH, HID, e->Lc.W2, HID, e->Lc.b2);
    const int nb_tail = (!fused_tail_nn && (e->tune_rows4 & 1)) ? e->nblk4 : e->nblk, nb_ln = (e->tune_rows4 & 2) ? e->nblk4 : e->nblk;   // row blocks that wrote the partials
    if (ln) { tn_fin(g.pr[i2], 0, e->Lc.g2, nb_tail); tn_fin(g.pr[i2], 1, e->Lc.be2, nb_tail); }
    tn_fin(g.pr[i2], 2, e->Lc.Wh, nb_tail); g.pr[i2].fin_s_off = e->Lc.bh; g.pr[i2].fin_s_nblk = nb_tail;
    g.pr[i1] = tn_prob(fold_ln1 ? e->c_dh1 : e->c_dz1, HID, BH, HID, S.X, e->ldc, 0, e->o + e->a, e->Lc.W1, e->Lc.ld1, e->Lc.b1);
    if (fold_ln1) {
      TnProb& q = g.pr[i1];
      q.fold = 1; q.f_ln = ln; q.f_g_off = e->Lc.g1; q.f_be_off = e->Lc.be1; q.f_xh = e->c_xh1; q.f_rstd = e->c_rs1; q.f_ps = e->c_ps; q.f_dz = e->c_dz1; q.f_g = e->c_ps + 2L * B * PS_W;
    } else if (ln) { tn_fin(g.pr[i1], 3, e->Lc.g1, nb_ln); tn_fin(g.pr[i1], 4, e->Lc.be1, nb_ln); }
    if (fold_ln1) {
      g.nprob = 3;
      if (e->tune_rows4 & 262144) { g.pr[1] = tn_rows(g.pr[0], HID / 2, HID / 2); g.pr[0] = tn_rows(g.pr[0], 0, HID / 2); }
      else { g.pr[1] = tn_cols(g.pr[0], HID / 2, HID / 2); g.pr[0] = tn_cols(g.pr[0], 0, HID / 2); g.pr[0].xr_force = g.pr[1].xr_force = 8; }
      std::swap(g.pr[1], g.pr[2]);
    }
    g.part = e->part; g.pstride = e->nblk4; g.part_s = e->part_s;
    g.apply = 1; g.P = e->Pc; g.Mo = e->Mc; g.Vo = e->Vc; g.T = fused_polyak_targ; g.tau = c.polyak; g.adam = e->ctl->adam_q;
    g.b1 = c.adam_beta1; g.b2 = c.adam_beta2; g.eps = c.adam_eps;
    g.loss_part = e->part_s; g.loss_n = 2 * e->nblk4; g.loss_stride = 2; g.loss_off = 1; g.loss_scale = 1.0f / (float)B;   // unused tail entries stay 0
    g.loss_dst = &e->ctl->metrics[SACTD3_M_QF_LOSS]; g.tick = &e->ctl->noise_ctr;
    const bool rides = actor_targ_rides && !(B >= BIG_BATCH && e->Gp);       // (the split-M route has no riding blocks)
    if (rides) {
      g.pk.t0 = e->Ta; g.pk.p0 = e->Pa; g.pk.n0 = e->La.size; g.pk.tau = c.polyak;
      // (few of them: the launch already has more blocks than the chip has CUs, and every extra one lands beside a tile block --
      //  64 riding blocks instead of 8: +0.8 us per TD3 iteration)
      g.pk_blocks = (int)std::min<long>(8, (e->La.size / 4 + 255) / 256);
      if (actor_targ_done) *actor_targ_done = true;
    }
    RCCHK(launch_tn(e, s, fused_polyak_targ ? (rides ? "dW+adam+polyak & actor-target polyak" : "dW+adam+polyak") : "dW+adam", g, 2, ctr_owed));
  }
  return 0;
}

// agents/agent.py:244-318.  j = index of this actor update inside the iteration (selects the noise buffers).
// head_done: this update's policy sample was already produced by the previous update's dual tail (fused iteration);
// merge_next: produce the NEXT update's policy sample together with this update's temperature draw.
// polyak_targ: (TD3, last actor update of a fused iteration) lerp the actor target towards the freshly stepped actor in the same
// kernel that applies the step (agents/agent.py:331 after :286)
// defer_alpha: (last actor update of an iteration that is followed by another one in the same graph) leave the temperature step
// to the next iteration's opening launch
// ahead: (pipelined period, SAC with autotune, last actor update of the period's first iteration) the sampling, gather and
// next-action pass a' ~ pi(s') of the `ahead` critic-only iterations that follow, into batch slots 1 .. ahead: the actor does not
// change any more before they run, so their passes ride in this update's last trunk / tail launches (the temperature draw's) as
// extra groups -- those iterations then start at their twin-critic trunk (enqueue_update_qnets, pre_sampled).
// chain_slot >= 0: ... and the opening pair of the NEXT period's first iteration into that batch slot (see BatchSlot, chain_ready).
// slot: the batch slot this iteration trains on.
static int enqueue_update_actor(sactd3_engine* e, hipStream_t s, int j, bool head_done = false, bool merge_next = false, float* polyak_targ = nullptr,
                                bool defer_alpha = false, int ahead = 0, int chain_slot = -1, int slot = 0) {
  const sactd3_config& c = e->cfg;
  const int B = e->B, ln = c.layer_norm, td3 = c.prefer_td3_over_sac, nq = e->nq_actor;
  const long BH = (long)B * HID;
  const int sb_a = SACTD3_SITE_ACTOR0 + (j & 1), sb_l = SACTD3_SITE_ALPHA0 + (j & 1);
  const bool clip = c.clip_norm > 0.f;
  const float* SX = e->bs[slot].X;          // the observations of the batch this iteration trains on
  const bool small_head = e->nh <= 8 && e->a <= 8 && !(e->tune_rows4 & 4);      // single-wave 4-row head backward (k_actor_head_bwd_s)
  e->node_role = (j & 1) ? "actor1/policy" : "actor0/policy";
  if (!head_done) {  // a_pi, logp = pi(s) with stores for the backward pass
    const TrunkGrp g{SX, e->Pa, e->a_z1, e->a_z2, e->a_xh1, e->a_h1, e->a_rs1};
    TrunkTicks tk{&e->ctl->t_a, nullptr, e->ctl->adam_a, e->ctl->pw_a, c.actor_lr};
    bool eps_ready = false;
    if (!td3) { tk.nnoise = 1; tk.noise[0] = noise_job(e, sb_a, 16u, 0, B); tk.noise_taken = &eps_ready; }
    RCCHK(enqueue_trunk(e, s, e->ldc, e->o, B, e->La, 0, 1, 1, &g, tk));
    ActorTail t = tail_args(e, e->a_z2, e->Pa, B, 0, 1, sb_a, 16u, e->Xp, e->ldc, e->o, e->logp_pi);
    t.obs_src = SX; t.lds = e->ldc;   // Xp = [s | pi(s)]
    t.eps_ready = eps_ready;
    RCCHK(launch_tail(e, s, t));
  }
  e->node_role = (j & 1) ? "actor1/q(s,pi)" : "actor0/q(s,pi)";
  {  // Q_i(s, a_pi) through the online critics as constants (agent.py:272-278)
    const TrunkGrp g{e->Xp, e->Pc, e->c_z1, e->c_z2, e->c_xh1, e->c_h1, e->c_rs1};
    TrunkTicks tk{nullptr, nullptr, nullptr, nullptr, 0.f};
    if (e->alpha_pending) { tk.alpha = &e->pending_alpha; e->alpha_pending = false; }   // the previous update's temperature step
    RCCHK(enqueue_trunk(e, s, e->ldc, e->o + e->a, B, e->Lc, e->Lc.size, 1, nq, &g, tk));
  }
  e->node_role = (j & 1) ? "actor1/loss+backward" : "actor0/loss+backward";
  const bool fused_qtail_nn = B < BIG_BATCH && !(e->tune_rows4 & 2048);
  // dQ/da finished inside the two fused launches around it (QaFold): narrow heads, ac_dim <= 7
  const bool qa_fold = fused_qtail_nn && small_head && e->a <= 7 && !(e->tune_rows4 & 16384);    // (small_head: k_headbwd_nn is the consumer)
  const int qa_ntile = HID / (nq == 2 ? 32 : 16), qa_pqw = e->a <= 3 ? 8 : 16;
  {
    ActorQTail t{};
    t.z2c = e->c_z2; t.P = e->Pc; t.p_ns = e->Lc.size; t.L = e->Lc; t.logp = e->logp_pi; t.log_alpha = e->la;
    t.B = B; t.ln = ln; t.sac = !td3; t.q = e->q_pi; t.dz2 = e->c_dz2; t.part_s = e->part_sa;
    if (fused_qtail_nn) {   // the tail AND dh1_i = dz2_i W2_i in one launch (k_qtail_nn): loss partials per 16-row block
      QtailNn f{};
      f.c = t; f.Wt = e->Pc + e->Lc.W2; f.ldw = HID; f.dX = e->c_dh1;
      if (qa_fold) {
        f.qa.on = 1; f.qa.ln = ln; f.qa.a = e->a; f.qa.pqw = qa_pqw; f.qa.ntile = qa_ntile; f.qa.h1 = e->c_h1; f.qa.xh1 = e->c_xh1;
        f.qa.g1_off = e->Lc.g1; f.qa.w1_off = e->Lc.W1; f.qa.ld1 = e->Lc.ld1; f.qa.k_off = e->o; f.qa.ps = e->qa_ps; f.qa.S = e->qa_S;
      }
      const double fl = 2.0 * nq * B * (double)HID + 2.0 * nq * (double)B * HID * HID, by = 4.0 * nq * (2.0 * BH + 4.0 * HID + 2.0 * B) + 4.0 * nq * ((double)HID * HID + (double)B * HID);
      if (nq == 2) {
        f.xr = pick_xr(e, e->nblk, HID / 32, 4.0 * 2 * B * HID, 4.0 * HID * HID);
        LAUNCH("k_qtail_nn<2>", fl, by, k_qtail_nn<2>, dim3((unsigned)(e->nblk * (HID / 32)), 1, 2), dim3(256), f);
      } else {
        f.xr = pick_xr(e, e->nblk, HID / 16, 4.0 * B * HID, 4.0 * HID * HID);
        LAUNCH("k_qtail_nn<1>", fl, by, k_qtail_nn<1>, dim3((unsigned)(e->nblk * (HID / 16)), 1, 1), dim3(256), f);
      }
    } else LAUNCH("k_actorq_tail<4>", 2.0 * nq * B * (double)HID, 4.0 * nq * (2.0 * BH + 4.0 * HID + 2.0 * B), k_actorq_tail<4>, dim3(e->nblk4), dim3(64), t);
  }
  if (!fused_qtail_nn) {
    NnArgs g{};
    g.dY = e->c_dz2; g.dy_ns = BH; g.Wt = e->Pc + e->Lc.W2; g.ldw = HID; g.p_ns = e->Lc.size; g.k_off = 0;
    g.dX = e->c_dh1; g.ldx = HID; g.dx_ns = BH; g.M = B; g.Kout = HID;
    RCCHK(launch_nn(e, s, "k_nn.dh1", g, nq));
  }
  if (!qa_fold) {
    LnBwd l{};
    l.dh = e->c_dh1; l.xh = e->c_xh1; l.h = e->c_h1; l.rstd = e->c_rs1;
    l.gamma = e->Pc + e->Lc.g1; l.p_ns = e->Lc.size; l.B = B; l.ln = ln; l.want_part = 0;
    l.dz = e->c_dz1; l.part = e->part; l.pstride = e->nblk4;
    // fused: dA_i = dz1_i W1_i[:, o:o+a]  (gradient of Q_i with respect to the action)
    l.W1 = e->Pc + e->Lc.W1; l.ldw1 = e->Lc.ld1; l.k_off = e->o; l.na = e->a; l.dA = e->dA; l.ldA = e->a4;
    LAUNCH("k_ln_bwd<16>.dQ/da", 2.0 * nq * B * (double)HID * e->a, 4.0 * nq * (4.0 * BH + B + HID + (double)HID * e->a + (double)B * e->a),
           k_ln_bwd<16>, dim3(e->nblk, nq), dim3(256), l);
  }
  bool fused_head_nn = false, fold_ln1 = false;
  {
    ActorHeadBwd h{};
    h.dA = e->dA; h.dA_ns = (long)B * e->a4; h.ldA = e->a4; h.nq = nq; h.tg = e->a_tg; h.a4 = e->a4; h.eps = e->eps[sb_a];
    h.log_alpha = e->la; h.scale = e->scale; h.P = e->Pa; h.L = e->La; h.xh2 = e->a_xh2; h.rstd2 = e->a_rs2; h.h2 = e->a_h2;
    h.B = B; h.a = e->a; h.ln = ln; h.sac = !td3; h.du = e->a_du; h.ldu = e->ldu; h.dz2 = e->a_dz2;
    h.part = e->part;
    fused_head_nn = small_head && B < BIG_BATCH && !(e->tune_rows4 & 2048);
    if (fused_head_nn) {   // the head backward AND dh1 = dz2 W2 in one launch (k_headbwd_nn): column partials per 16-row block
      HeadBwdNn f{};
      f.c = h; f.Wt = e->Pa + e->La.W2; f.ldw = HID; f.dX = e->a_dh1;
      if (qa_fold) {
        f.qa.on = 1; f.qa.ln = ln; f.qa.a = e->a; f.qa.pqw = qa_pqw; f.qa.ntile = qa_ntile; f.qa.ps = e->qa_ps; f.qa.S = e->qa_S;
        f.qa.rstd = e->c_rs1; f.qa.dA = e->dA;
      }
      fold_ln1 = !(e->tune_rows4 & 4096);   // as the critics' (enqueue_update_qnets)
      f.f.fold = fold_ln1; f.f.ln = ln; f.f.h1 = e->a_h1; f.f.xh1 = e->a_xh1; f.f.g1_off = e->La.g1; f.f.ps = e->a_ps; f.f.gsnap = e->a_ps + (long)B * PS_W;
      f.xr = pick_xr(e, e->nblk, HID / 16, 4.0 * 2 * B * HID, 4.0 * HID * HID);
      LAUNCH("k_headbwd_nn", 2.0 * B * (double)HID * e->nh + 2.0 * (double)B * HID * HID,
             4.0 * (3.0 * BH + (double)e->nh * HID + (double)B * (nq * e->a + 4 * e->a + e->nh)) + 4.0 * ((double)HID * HID + (double)B * HID),
             k_headbwd_nn, dim3((unsigned)(e->nblk * (HID / 16))), dim3(256), f);
    } else if (small_head) LAUNCH("k_actor_head_bwd_s<4>", 2.0 * B * (double)HID * e->nh, 4.0 * (3.0 * BH + (double)e->nh * HID + (double)B * (nq * e->a + 4 * e->a + e->nh)),
                           k_actor_head_bwd_s<4>, dim3(e->nblk4), dim3(64), h);
    else LAUNCH("k_actor_head_bwd", 2.0 * B * (double)HID * e->nh, 4.0 * (3.0 * BH + (double)e->nh * HID + (double)B * (nq * e->a + 4 * e->a + e->nh)),
                k_actor_head_bwd, dim3(e->nblk), dim3(256), h);
  }
  if (!fused_head_nn) {
    NnArgs g{};
    g.dY = e->a_dz2; g.Wt = e->Pa + e->La.W2; g.ldw = HID; g.k_off = 0; g.dX = e->a_dh1; g.ldx = HID; g.M = B; g.Kout = HID;
    RCCHK(launch_nn(e, s, "k_nn.dh1", g, 1));
  }
  if (!fold_ln1) {
    LnBwd l{};
    l.dh = e->a_dh1; l.xh = e->a_xh1; l.h = e->a_h1; l.rstd = e->a_rs1; l.gamma = e->Pa + e->La.g1;
    l.B = B; l.ln = ln; l.want_part = ln; l.dz = e->a_dz1; l.part = e->part; l.pstride = e->nblk4;
    LAUNCH("k_ln_bwd<16>", 0.0, 4.0 * (4.0 * BH + B + HID), k_ln_bwd<16>, dim3(e->nblk, 1), dim3(256), l);
  }
  {  // every actor gradient (+ Adam unless clip_grad_norm_ needs the global norm first) in one launch:
     //   dWhead = du^T h2, dbhead ; dW2 = dz2^T h1, db2, dgamma2, dbeta2 ; dW1 = dz1^T s, db1, dgamma1, dbeta1
    TnArgs g{};
    g.nprob = 3; g.M = B; g.G = e->Ga; g.g_ns = 0;
    const int ih = 0, i2 = 1, i1 = fold_ln1 ? 3 : 2;      // folded: [head][W2 rows 0 .. 47][layer 1][W2 rows 48 .. 255] (tn_rows)
    g.pr[ih] = tn_prob(e->a_du, e->ldu, 0, e->nh, e->a_h2, HID, 0, HID, e->La.Wh, HID, e->La.bh);
    g.pr[i2] = tn_prob(e->a_dz2, HID, 0, HID, e->a_h1, HID, 0, HID, e->La.W2, HID, e->La.b2);
    const int nb_head = (small_head && !fused_head_nn) ? e->nblk4 : e->nblk;      // row blocks that wrote the head backward's column partials
    if (ln) { tn_fin(g.pr[i2], 0, e->La.g2, nb_head); tn_fin(g.pr[i2], 1, e->La.be2, nb_head); }
    g.pr[i1] = tn_prob(fold_ln1 ? e->a_dh1 : e->a_dz1, HID, 0, HID, SX, e->ldc, 0, e->o, e->La.W1, e->La.ld1, e->La.b1);
    if (fold_ln1) {
      TnProb& q = g.pr[i1];
      q.fold = 1; q.f_ln = ln; q.f_g_off = e->La.g1; q.f_be_off = e->La.be1; q.f_xh = e->a_xh1; q.f_rstd = e->a_rs1; q.f_ps = e->a_ps; q.f_dz = e->a_dz1; q.f_g = e->a_ps + (long)B * PS_W;
    } else if (ln) { tn_fin(g.pr[i1], 3, e->La.g1, e->nblk); tn_fin(g.pr[i1], 4, e->La.be1, e->nblk); }
    if (fold_ln1) { g.nprob = 4; g.pr[2] = tn_rows(g.pr[1], 48, HID - 48); g.pr[1] = tn_rows(g.pr[1], 0, 48); std::swap(g.pr[2], g.pr[3]); }
    g.part = e->part; g.pstride = e->nblk4; g.part_s = e->part_s;
    g.apply = clip ? 0 : 1; g.P = e->Pa; g.Mo = e->Ma; g.Vo = e->Va; g.T = clip ? nullptr : polyak_targ; g.tau = c.polyak; g.adam = e->ctl->adam_a;
    if (td3 && g.T && (ahead > 0 || chain_slot >= 0)) { g.T2 = e->Ta2; g.T3 = e->Ta3; }   // the actor target of the next two Polyak updates, now
    g.b1 = c.adam_beta1; g.b2 = c.adam_beta2; g.eps = c.adam_eps;
    g.loss_part = e->part_sa; g.loss_n = e->nblk4; g.loss_stride = 2; g.loss_off = 1; g.loss_scale = 1.0f / (float)B;
    g.loss_dst = &e->ctl->metrics[SACTD3_M_ACTOR_LOSS]; g.tick = (td3 && !clip) ? &e->ctl->noise_ctr : nullptr;
    RCCHK(launch_tn(e, s, clip ? "dW" : (polyak_targ ? "dW+adam+polyak" : "dW+adam"), g, 1));
  }
  if (clip) {
    NormArgs n{e->Ga, (long)e->La.size, c.clip_norm, e->gscale};
    LAUNCH("k_gradnorm", 0.0, 4.0 * e->La.size, k_gradnorm, dim3(1), dim3(1024), n);
    AdamArgs a = adam_args(e, e->Pa, e->Ga, e->Ma, e->Va, e->La.size, e->ctl->adam_a);
    a.gscale = e->gscale;
    a.targ = polyak_targ; a.tau = c.polyak;
    a.tick = td3 ? &e->ctl->noise_ctr : nullptr;
    RCCHK(launch_adam(e, s, a));
  }
  if (td3 && (ahead > 0 || chain_slot >= 0)) {
    // TD3, pipelined period (see BatchSlot): the online actor is final until the next period's actor updates, and the TARGET actor
    // of the following iterations is a fixed sequence of lerps towards it -- Ta (after this update's Polyak step), Ta2, Ta3 (written by
    // the Adam epilogue above).  So the sampling, gather and next-action pass of the period's critic-only iterations, and the opening
    // pair of the next period's first iteration (its next-action pass through Ta3 / Ta2, its first policy pass through Pa), run here
    // as ONE trunk + tail pair.  Streams: iteration k samples with sample_ctr + (k - 1) and draws its smoothing noise with noise
    // counter + (k - 1) (the counter already counts this update's tick; one tick per critic update in between); the next period's
    // first iteration with + ahead -- the values their own opening launches would use.
    e->node_role = chain_slot >= 0 ? "next-action passes ahead & next period's opening" : "next-action passes ahead";
    const float* Tk[3] = {e->Ta, e->Ta2, e->Ta3};
    const int mode = c.targ_actor_smoothing ? 1 : 0;
    TrunkGrp g[5] = {};
    TrunkTicks tk{nullptr, nullptr, nullptr, nullptr, 0.f};
    ActorTail5 T{};
    bool eps_ready = false;
    int ng = 0;
    auto add_next = [&](int slot_k, int order) {
      const sactd3_engine::BatchSlot& S = e->bs[slot_k];
      g[ng] = TrunkGrp{S.Xn, Tk[order], e->ah_z1[ng], e->ah_z2[ng], nullptr, nullptr, nullptr, e->ldc};
      g[ng].ring = true; g[ng].sctr_add = order; g[ng].ring_idx = S.idx;
      tk.gather[tk.ngather++] = gather_args(e, e->ring, -1, slot_k, order);
      if (mode == 1) { tk.noise[tk.nnoise] = noise_job(e, SACTD3_SITE_CRITIC, 0u, order, B); tk.noise[tk.nnoise++].eps = S.eps_c; tk.noise_taken = &eps_ready; }
      T.t[ng] = tail_args(e, e->ah_z2[ng], Tk[order], B, mode, 0, SACTD3_SITE_CRITIC, 0u, S.Xn, e->ldc, e->o, S.logp_n);
      T.t[ng].eps = S.eps_c; T.t[ng].ctr_add = order;
      ++ng;
    };
    for (int k = 1; k <= ahead; ++k) add_next(k, k - 1);
    if (chain_slot >= 0) {
      add_next(chain_slot, ahead);
      const sactd3_engine::BatchSlot& S = e->bs[chain_slot];
      g[ng] = TrunkGrp{S.X, e->Pa, e->ah_z1[ng], e->ah_z2[ng], e->a_xh1, e->a_h1, e->a_rs1, 0};
      g[ng].ring = true; g[ng].sctr_add = ahead; g[ng].ring_idx = S.idx;
      T.t[ng] = tail_args(e, e->ah_z2[ng], e->Pa, B, 0, 1, SACTD3_SITE_ACTOR0, 16u, e->Xp, e->ldc, e->o, e->logp_pi);
      T.t[ng].obs_src = S.X; T.t[ng].lds = e->ldc;                      // Xp = [s | pi(s)] (the gather above has filled S.X)
      ++ng;
    }
    tk.force_ks = 4; tk.no_tiled64 = true;      // (the launch shapes of the single-net passes: ring rows, riding gathers, bit-equal results)
    RCCHK(enqueue_trunk(e, s, e->ldc, e->o, B, e->La, 0, ng, 1, g, tk));
    if (ahead) { T.t[0].tick = &e->ctl->sample_ctr; T.t[0].tick_add = ahead - 1; }      // every reader of the index streams is done
    for (int i = 0; i < ng; ++i) T.t[i].eps_ready = eps_ready;
    for (int i = ng; i < 5; ++i) T.t[i] = T.t[0];
    const int rpb = tail_rows_per_block(T.t[0]);
    T.nb = (B + rpb - 1) / rpb; T.n = ng;
    const double fl = 2.0 * ng * B * (double)HID * e->nh;
    const double by = 4.0 * ng * ((double)B * HID + (double)e->nh * (HID + 1) + 2.0 * HID + (double)B * (3 * e->a + 2)) + (chain_slot >= 0 ? 8.0 * (double)B * (HID + e->o) : 0.0);
    if (rpb == 4) LAUNCH("k_actor_tail_s5<4>", fl, by, k_actor_tail_s5<4>, dim3(ng * T.nb), dim3(64), T);
    else LAUNCH("k_actor_tail5", fl, by, k_actor_tail5, dim3(ng * T.nb), dim3(256), T);
  }
  e->node_role = (j & 1) ? "actor1/alpha" : "actor0/alpha";
  if (!td3) {
    if (c.autotune && merge_next) {
      // The temperature draw (agent.py:297-299) and the next actor update's policy sample (agent.py:254) both go through
      // the SAME freshly updated actor on the SAME observations: one trunk launch (with the next update's backward
      // stores and Adam tick) and one tail launch with two draws.  Streams: temperature (ctr, 32), policy (ctr + 1, 16),
      // exactly what the two separate launches would consume.
      const TrunkGrp g{SX, e->Pa, e->a_z1, e->a_z2, e->a_xh1, e->a_h1, e->a_rs1};
      const int sb_a_next = SACTD3_SITE_ACTOR0 + ((j + 1) & 1);
      TrunkTicks tk{&e->ctl->t_a, nullptr, e->ctl->adam_a, e->ctl->pw_a, c.actor_lr};
      bool eps_ready = false;
      tk.nnoise = 2; tk.noise[0] = noise_job(e, sb_a_next, 16u, 1, B); tk.noise[1] = noise_job(e, sb_l, 32u, 0, B); tk.noise_taken = &eps_ready;
      RCCHK(enqueue_trunk(e, s, e->ldc, e->o, B, e->La, 0, 1, 1, &g, tk));
      ActorTail t = tail_args(e, e->a_z2, e->Pa, B, 0, 1, sb_a_next, 16u, e->Xp, e->ldc, e->o, e->logp_pi);
      t.obs_src = SX; t.lds = e->ldc; t.ctr_add = 1;
      t.dual = 1; t.site_buf2 = sb_l; t.site_code2 = 32u; t.eps2 = e->eps[sb_l]; t.logp2 = e->logp_al;
      t.eps_ready = eps_ready;
      RCCHK(launch_tail(e, s, t));
    } else if (c.autotune) {  // fresh draw through the already-updated actor (agent.py:297-299)
      TrunkGrp g[5] = {{e->bs[slot].X, e->Pa, e->a_z1, e->a_z2, nullptr, nullptr, nullptr}, {}, {}, {}, {}};
      TrunkTicks tk{nullptr, nullptr, nullptr, nullptr, 0.f};
      bool eps_ready = false;
      tk.nnoise = 1; tk.noise[0] = noise_job(e, sb_l, 32u, 0, B); tk.noise_taken = &eps_ready;
      ActorTail5 T{};
      int ng = 1;
      // The noise counter stands at N + (this update's index + 1) here (one tick by the critic update, one per finished temperature
      // step), the sample counter already counts this iteration.  Iteration k of the period (k = 1 .. ahead): its sample is the
      // one drawn with sample_ctr + (k - 1), its critic-site draws those of noise counter + k (the tick owed by this update's
      // deferred temperature step, one per critic update in between) -- exactly what its own opening launches would have used.
      for (int k = 1; k <= ahead; ++k) {
        const sactd3_engine::BatchSlot& S = e->bs[k];
        g[ng] = TrunkGrp{S.Xn, e->Pa, e->ah_z1[k - 1], e->ah_z2[k - 1], nullptr, nullptr, nullptr, e->ldc};
        g[ng].ring = true; g[ng].sctr_add = k - 1; g[ng].ring_idx = S.idx;
        tk.gather[tk.ngather++] = gather_args(e, e->ring, -1, k, k - 1);
        tk.noise[tk.nnoise] = noise_job(e, SACTD3_SITE_CRITIC, 0u, k, B); tk.noise[tk.nnoise++].eps = S.eps_c;
        T.t[ng] = tail_args(e, e->ah_z2[k - 1], e->Pa, B, 0, 0, SACTD3_SITE_CRITIC, 0u, S.Xn, e->ldc, e->o, S.logp_n);
        T.t[ng].eps = S.eps_c; T.t[ng].ctr_add = k;
        ++ng;
      }
      // ... and the opening pair of the NEXT period's first iteration, into batch slot `chain_slot`: sample sample_ctr + ahead,
      // critic-site draws of noise counter + ahead + 1 (where that period starts), policy draws (site 16) one further -- the values
      // its own opening launches use (enqueue_update_qnets: merge_policy); the policy pass keeps its stores for the backward pass
      if (chain_slot >= 0) {
        const sactd3_engine::BatchSlot& S = e->bs[chain_slot];
        g[ng] = TrunkGrp{S.Xn, e->Pa, e->ah_z1[2], e->ah_z2[2], nullptr, nullptr, nullptr, e->ldc};
        g[ng].ring = true; g[ng].sctr_add = ahead; g[ng].ring_idx = S.idx;
        g[ng + 1] = TrunkGrp{S.X, e->Pa, e->ah_z1[3], e->ah_z2[3], e->a_xh1, e->a_h1, e->a_rs1, 0};
        g[ng + 1].ring = true; g[ng + 1].sctr_add = ahead; g[ng + 1].ring_idx = S.idx;
        tk.gather[tk.ngather++] = gather_args(e, e->ring, -1, chain_slot, ahead);
        tk.noise[tk.nnoise] = noise_job(e, SACTD3_SITE_CRITIC, 0u, ahead + 1, B); tk.noise[tk.nnoise++].eps = S.eps_c;
        tk.noise[tk.nnoise++] = noise_job(e, SACTD3_SITE_ACTOR0, 16u, ahead + 2, B);
        T.t[ng] = tail_args(e, e->ah_z2[2], e->Pa, B, 0, 0, SACTD3_SITE_CRITIC, 0u, S.Xn, e->ldc, e->o, S.logp_n);
        T.t[ng].eps = S.eps_c; T.t[ng].ctr_add = ahead + 1;
        T.t[ng + 1] = tail_args(e, e->ah_z2[3], e->Pa, B, 0, 1, SACTD3_SITE_ACTOR0, 16u, e->Xp, e->ldc, e->o, e->logp_pi);
        T.t[ng + 1].obs_src = S.X; T.t[ng + 1].lds = e->ldc; T.t[ng + 1].ctr_add = ahead + 2;      // Xp = [s | pi(s)] (the gather above has filled S.X)
        ng += 2;
      }
      if (ng > 1) {
        tk.force_ks = 4; tk.no_tiled64 = true;
        e->node_role = chain_slot >= 0 ? "alpha & next-action passes ahead & next period's opening" : "alpha & next-action passes ahead";
      }
      RCCHK(enqueue_trunk(e, s, e->ldc, e->o, B, e->La, 0, ng, 1, g, tk));
      ActorTail t = tail_args(e, e->a_z2, e->Pa, B, 0, 0, sb_l, 32u, e->act_scratch, e->a4, 0, e->logp_al);
      t.eps_ready = eps_ready;
      if (ng == 1) RCCHK(launch_tail(e, s, t));
      else {
        if (ahead) { t.tick = &e->ctl->sample_ctr; t.tick_add = ahead - 1; }      // every reader of the index streams (the trunk launch above) is done
        T.t[0] = t;
        for (int i = 1; i < ng; ++i) T.t[i].eps_ready = eps_ready;
        for (int i = ng; i < 5; ++i) T.t[i] = T.t[0];                               // (never run: blocks exist for n tails only)
        const int rpb = tail_rows_per_block(t);
        T.nb = (B + rpb - 1) / rpb; T.n = ng;
        const double fl = 2.0 * ng * B * (double)HID * t.L.nh;
        const double by = 4.0 * ng * ((double)B * HID + (double)t.L.nh * (HID + 1) + 2.0 * HID + (double)B * (3 * e->a + 2)) + (chain_slot >= 0 ? 8.0 * (double)B * (HID + e->o) : 0.0);
        if (rpb == 4) LAUNCH("k_actor_tail_s5<4>", fl, by, k_actor_tail_s5<4>, dim3(ng * T.nb), dim3(64), T);
        else LAUNCH("k_actor_tail5", fl, by, k_actor_tail5, dim3(ng * T.nb), dim3(256), T);
      }
    }
    AlphaArgs al{};
    al.logp = e->logp_al; al.B = B; al.targ_ent = -(float)e->a; al.autotune = c.autotune; al.la = e->la; al.ctl = e->ctl;
    al.lr = c.log_alpha_lr; al.b1 = c.adam_beta1; al.b2 = c.adam_beta2; al.eps = c.adam_eps; al.tick = &e->ctl->noise_ctr;
    if (merge_next) { e->pending_alpha = al; e->alpha_pending = true; }   // rides in the next update's critic-trunk launch
    else if (defer_alpha) { al.tick = nullptr; e->pending_alpha = al; e->alpha_pending = true; e->alpha_tick_owed = true; }
    else {
      LAUNCH("k_alpha_step", 0.0, 4.0 * B, k_alpha_step, dim3(1), dim3(256), al);
    }
  }
  return 0;
}

// agents/agent.py:328-331
static int enqueue_polyak(sactd3_engine* e, hipStream_t s, bool critics, bool actor) {
  e->node_role = "targets";
  PolyakArgs p{};
  p.tau = e->cfg.polyak;
  if (critics) { p.t0 = e->Tc; p.p0 = e->Pc; p.n0 = 2L * e->Lc.size; }
  if (actor) { p.t1 = e->Ta; p.p1 = e->Pa; p.n1 = e->La.size; }
  const long n = p.n0 + p.n1;
  if (n == 0) return 0;
  LAUNCH("k_polyak", 0.0, 12.0 * n, k_polyak, dim3((unsigned)std::min<long>(512, (n / 4 + 255) / 256)), dim3(256), p);
  return 0;
}

// orchestrator.py:337-352 as one sequence
// slot / pre_sampled / ahead: the pipelined form of a period graph (period_is_pipelined): iteration i of the period trains on batch
// slot i; the first one (with the actor updates) also runs the sampling + next-action passes of the `ahead` iterations behind it,
// which are then `pre_sampled`.
// chain_slot / policy_pre: ... and, chained periods, the opening pair of the next period's first iteration into batch slot chain_slot;
// policy_pre = this iteration's own opening pair (sample, next-action pass, the first actor update's policy pass) is already there.
static int enqueue_step(sactd3_engine* e, hipStream_t s, bool do_actor, bool do_polyak, bool next_in_same_graph = false,
                        int slot = 0, bool pre_sampled = false, int ahead = 0, int chain_slot = -1, bool policy_pre = false) {
  const bool td3 = e->cfg.prefer_td3_over_sac;
  e->node_role = "sample";
  if (!pre_sampled && !opening_trunk_gathers(e)) RCCHK(enqueue_gather(e, s, e->ring, -1));   // otherwise the gather is inside the first trunk kernel
  // SAC: critic targets are lerped towards the freshly stepped critics inside the Adam kernel (same element,
  // same order as agent.py:328 after :236); TD3 also needs the actor target, done after the actor updates.
  // Target updates (agents/agent.py:320-331) are folded into the kernels that apply the optimiser steps: the critic targets are
  // lerped towards the freshly stepped critics inside the critics' Adam epilogue (same element, same order as :328 after :236;
  // nothing between there and the end of the iteration reads the targets).  TD3 also moves the actor target every iteration:
  // in the last actor update's Adam epilogue when the iteration has actor updates, otherwise -- the actor did not change -- as a
  // few extra blocks of the critics' weight-gradient launch.
  bool policy_done = false, actor_targ_done = false;
  const bool actor_targ = do_polyak && td3;
  RCCHK(enqueue_update_qnets(e, s, true, do_polyak ? e->Tc : nullptr, do_actor, &policy_done, actor_targ && !do_actor, &actor_targ_done, slot, pre_sampled,
                             policy_pre ? 2 : 0));
  if (policy_pre) policy_done = true;
  if (do_actor) {
    const int n = e->cfg.actor_update_delay;
    const bool can_merge = !td3 && e->cfg.autotune;
    for (int j = 0; j < n; ++j) {
      const bool last = j + 1 == n;
      float* pt = nullptr;
      if (actor_targ && last) { pt = e->Ta; actor_targ_done = true; }
      // the last temperature step can wait for the next iteration's opening trunk launch when that launch can carry it
      const bool defer = last && next_in_same_graph && !td3 && (ahead > 0 || opening_trunk_carries_alpha(e));
      RCCHK(enqueue_update_actor(e, s, j, j == 0 ? policy_done : can_merge, can_merge && !last, pt, defer, last ? ahead : 0, last ? chain_slot : -1, slot));
    }
  }
  if (actor_targ && !actor_targ_done) RCCHK(enqueue_polyak(e, s, false, true));
  return 0;
}

template <class F>
static int run_graph_slot(sactd3_engine* e, hipGraphExec_t* slot, int* nodes, F&& enqueue, bool launch = true);
template <class F>
static int run_graph(sactd3_engine* e, int which, F&& enqueue, bool launch = true) {
  return run_graph_slot(e, &e->graphs[which], &e->graph_nodes[which], enqueue, launch);
}
// launch == false: capture + instantiate only (sactd3_instantiate_graphs)
template <class F>
static int run_graph_slot(sactd3_engine* e, hipGraphExec_t* slot, int* nodes, F&& enqueue, bool launch) {
  if (!e->cfg.use_graphs) return launch ? enqueue(e->stream) : 0;
  if (!*slot) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue(e->stream);
    hipError_t he = hipStreamEndCapture(e->stream, &g);
    if (rc != 0) { if (g) hipGraphDestroy(g); return rc; }
    if (he != hipSuccess) return e->fail(SACTD3_EHIP, "hipStreamEndCapture", he);
    size_t n = 0;
    HIPCHK(hipGraphGetNodes(g, nullptr, &n));
    if (nodes) *nodes = (int)n;
    he = hipGraphInstantiate(slot, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (he != hipSuccess) return e->fail(SACTD3_EHIP, "hipGraphInstantiate", he);
    // move the executable graph's launch resources to the device now, not inside its first launch (a loop that instantiates ahead --
    // sactd3_instantiate_graphs -- then pays nothing extra the first time each graph runs); not every runtime implements it: best effort
    if (hipGraphUpload(*slot, e->stream) != hipSuccess) (void)hipGetLastError();
  }
  if (launch) HIPCHK(hipGraphLaunch(*slot, e->stream));
  return 0;
}

__global__ void k_set_int1(int* dst, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v; }
static int set_flag(sactd3_engine* e, int* dst, int v) {
  hipLaunchKernelGGL(k_set_int1, dim3(1), dim3(1), 0, e->stream, dst, v);
  HIPCHK(hipGetLastError());
  return 0;
}
// ctl->rb_len and ctl->rb_cursor are adjacent ints
static int publish_rb_state(sactd3_engine* e) {
  hipLaunchKernelGGL(k_set_int2, dim3(1), dim3(1), 0, e->stream, &e->ctl->rb_len, (int)e->rb_len, (int)e->rb_cursor);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ C ABI
#pragma GCC visibility push(default)
extern "C" {

int sactd3_abi_version(void) { return SACTD3_ABI_VERSION; }

void sactd3_default_config(sactd3_config* c, int td3) {
  memset(c, 0, sizeof(*c));
  c->abi_version = SACTD3_ABI_VERSION;
  c->batch_size = 256; c->rb_capacity = 1000000; c->max_envs = 4;
  c->prefer_td3_over_sac = td3 ? 1 : 0; c->layer_norm = 1; c->autotune = 1;
  c->bcq_style_targ_mix = td3 ? 1 : 0; c->targ_actor_smoothing = 1;
  c->actor_update_delay = 2; c->crit_targ_update_freq = 1; c->use_graphs = 1;
  c->actor_lr = 3e-4f; c->qnets_lr = td3 ? 3e-4f : 1e-3f; c->log_alpha_lr = 1e-3f;
  c->gamma = 0.99f; c->polyak = 0.005f; c->alpha_init = 0.2f; c->clip_norm = 0.f;
  c->td3_std = 0.2f; c->td3_c = 0.5f; c->actor_noise_std = 0.1f;
  c->adam_beta1 = 0.9f; c->adam_beta2 = 0.999f; c->adam_eps = 1e-8f;
}

const char* sactd3_last_error(const sactd3_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

void sactd3_destroy(sactd3_engine* e) {
  if (!e) return;
  if (e->stream) (void)hipSetDevice(e->cfg.device_id);   // (a failed create may carry a device_id that was never valid)
  if (e->stream) hipStreamSynchronize(e->stream);
  for (auto& g : e->graphs) if (g) hipGraphExecDestroy(g);
  for (auto& g : e->predict_graphs) if (g) hipGraphExecDestroy(g);
  for (auto ev : e->events) hipEventDestroy(ev);
  for (void* p : e->dev_allocs) hipFree(p);
  for (void* p : e->host_allocs) hipHostFree(p);
  if (e->stream) hipStreamDestroy(e->stream);
  delete e;
}

static int create_impl(sactd3_engine* e, const float* min_ac, const float* max_ac) {
  const sactd3_config& c = e->cfg;
  if (c.abi_version != SACTD3_ABI_VERSION) return e->fail(SACTD3_EINVAL, "abi_version mismatch");
  if (c.ob_dim < 1 || c.ac_dim < 1 || c.ac_dim > 32) return e->fail(SACTD3_EINVAL, "ob_dim >= 1 and 1 <= ac_dim <= 32 required");
  if (c.batch_size < 1 || c.rb_capacity < 1 || c.max_envs < 1) return e->fail(SACTD3_EINVAL, "batch_size, rb_capacity, max_envs must be positive");
  if (c.actor_update_delay < 0 || c.crit_targ_update_freq < 1) return e->fail(SACTD3_EINVAL, "actor_update_delay >= 0 and crit_targ_update_freq >= 1 required");
  if (!min_ac || !max_ac) return e->fail(SACTD3_EINVAL, "min_ac / max_ac required");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return e->fail(SACTD3_ENODEV, "no HIP device visible");
  if (c.device_id < 0 || c.device_id >= ndev) return e->fail(SACTD3_EINVAL, "device_id out of range");
  HIPCHK(hipSetDevice(c.device_id));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, c.device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return e->fail(SACTD3_ENODEV, "device is not gfx950 (this library carries gfx950 code objects only)");
  e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // kernel-selection defaults (what the shipped library always runs): 4-row single-wave k_critic_tail / k_ln_bwd below B = 1024
  // (-1.0 us per Hopper iteration, +-0 at Humanoid); the split-M weight-gradient route from half a chip's worth of 64 x 32 tiles
  e->tune_rows4 = c.batch_size < BIG_BATCH ? 3 : 0;
  e->tune_tn64_min = e->num_cus / 2;
#ifdef SACTD3_TUNING
  // A/B switches of the tuning builds ONLY (`make tune` -> libsactd3_hip_tune.so, used by tools/ab_*.py / tools/ab_iter.sh through
  // SACTD3_LIBRARY): the shipped library reads no environment variable at all.  Read once here, never on a launch path.
  //   SACTD3_KS 1|2|4 / SACTD3_NT 1 / SACTD3_TN_KT 1|2: block shapes of k_nt / k_tn;  SACTD3_PAD64: dynamic-LDS pad of k_nt64;
  //   SACTD3_NN16 1: k_nn instead of k_nn64;  SACTD3_XR -1|0|1|2|4|8: XCD tile placement (-1 = least-fetch split per launch);
  //   SACTD3_TN64_MIN: tiles x nets from which the split-M route is taken;
  //   SACTD3_ROWS4: bit mask.  1 / 2: the 4-row single-wave k_critic_tail / k_ln_bwd.  The other bits turn a default OFF:
  //   4 k_actor_head_bwd_s, 16 the policy-pass merge for wide observations, 32 the in-kernel replay gather of the wide opening trunk,
  //   128 k_nt64_ln in the 4-net trunk (-> k_ln_fwd + k_nt64), 256 k_nt64_ln<2,2,1> for the 1- / 2-net second layers (-> k_nt).
  if (const char* f = getenv("SACTD3_KS")) { const int v = atoi(f); if (v == 1 || v == 2 || v == 4) e->tune_ks = v; }
  if (const char* f = getenv("SACTD3_NT")) { if (atoi(f) == 1) e->tune_nt = 1; }
  if (const char* f = getenv("SACTD3_PAD64")) e->tune_pad64 = atoi(f);
  if (const char* f = getenv("SACTD3_NN16")) e->tune_nn16 = atoi(f);
  if (const char* f = getenv("SACTD3_XR")) e->tune_xr = atoi(f);
  if (const char* f = getenv("SACTD3_ROWS4")) e->tune_rows4 = atoi(f);
  if (const char* f = getenv("SACTD3_TN64_MIN")) e->tune_tn64_min = atoi(f);
  if (const char* f = getenv("SACTD3_TN_KT")) { const int v = atoi(f); if (v == 1 || v == 2) e->tune_tn_kt = v; }
#endif
  HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));

  e->o = c.ob_dim; e->a = c.ac_dim; e->B = c.batch_size;
  const int td3 = c.prefer_td3_over_sac;
  e->nh = td3 ? e->a : 2 * e->a; e->ldu = round_up(e->nh, 4); e->a4 = round_up(e->a, 4);
  e->ldc = round_up(e->o + e->a, 4); e->ldo = round_up(e->o, 4);
  e->cx = e->ldc / 4; e->cn = e->ldo / 4;
  e->rec_f = round_up(e->ldc + e->ldo + 2, 16); e->rec4 = e->rec_f / 4;
  if ((long long)e->B * e->rec4 >= (1ll << 31)) return e->fail(SACTD3_EINVAL, "batch_size x record size too large");
  e->La = make_layout(e->o, e->nh); e->Lc = make_layout(e->o + e->a, 1);
  e->nq_actor = td3 ? 1 : 2;
  e->nblk = (e->B + 15) / 16; e->nblk4 = (e->B + 3) / 4;
  e->maxn = c.max_envs;
  e->stage_rows = std::max(c.max_envs, 256);
  const size_t B = e->B, BH = B * HID;

  RCCHK(dalloc(e, &e->ctl, 1));
  RCCHK(dalloc(e, &e->min_ac, e->a4)); RCCHK(dalloc(e, &e->max_ac, e->a4));
  RCCHK(dalloc(e, &e->scale, e->a4)); RCCHK(dalloc(e, &e->bias, e->a4));
  RCCHK(dalloc(e, &e->Pa, e->La.size)); RCCHK(dalloc(e, &e->Ta, e->La.size)); RCCHK(dalloc(e, &e->Ga, e->La.size));
  RCCHK(dalloc(e, &e->Ma, e->La.size)); RCCHK(dalloc(e, &e->Va, e->La.size));
  RCCHK(dalloc(e, &e->Ta2, e->La.size)); RCCHK(dalloc(e, &e->Ta3, e->La.size));
  RCCHK(dalloc(e, &e->Pc, 2 * e->Lc.size)); RCCHK(dalloc(e, &e->Tc, 2 * e->Lc.size)); RCCHK(dalloc(e, &e->Gc, 2 * e->Lc.size));
  RCCHK(dalloc(e, &e->Mc, 2 * e->Lc.size)); RCCHK(dalloc(e, &e->Vc, 2 * e->Lc.size));
  RCCHK(dalloc(e, &e->la, 4)); RCCHK(dalloc(e, &e->gscale, 4));
  RCCHK(dalloc(e, &e->ring, (size_t)c.rb_capacity * e->rec_f, false));
  RCCHK(dalloc(e, &e->stage_dev, B * e->rec_f));
  RCCHK(dalloc(e, &e->X, B * e->ldc)); RCCHK(dalloc(e, &e->Xn, B * e->ldc)); RCCHK(dalloc(e, &e->Xp, B * e->ldc));
  RCCHK(dalloc(e, &e->rew, B)); RCCHK(dalloc(e, &e->done, B)); RCCHK(dalloc(e, &e->idx, B));
  RCCHK(dalloc(e, &e->logp_n, B)); RCCHK(dalloc(e, &e->logp_pi, B)); RCCHK(dalloc(e, &e->logp_al, B));
  RCCHK(dalloc(e, &e->act_scratch, B * e->a4));
  for (int s = 0; s < SACTD3_NUM_SITES; ++s)
    RCCHK(dalloc(e, &e->eps[s], std::max<size_t>(B, e->maxn) * e->a));
  RCCHK(dalloc(e, &e->a_z1, BH)); RCCHK(dalloc(e, &e->a_xh1, BH)); RCCHK(dalloc(e, &e->a_h1, BH)); RCCHK(dalloc(e, &e->a_rs1, B));
  RCCHK(dalloc(e, &e->a_z2, BH)); RCCHK(dalloc(e, &e->a_xh2, BH)); RCCHK(dalloc(e, &e->a_h2, BH)); RCCHK(dalloc(e, &e->a_rs2, B));
  RCCHK(dalloc(e, &e->a_tg, B * 4 * e->a4)); RCCHK(dalloc(e, &e->a_du, B * e->ldu)); RCCHK(dalloc(e, &e->a_z2n, BH));
  e->bs[0] = {e->X, e->Xn, e->rew, e->done, e->logp_n, e->eps[SACTD3_SITE_CRITIC], e->idx};
  for (int k = 0; k < 4; ++k) { RCCHK(dalloc(e, &e->ah_z1[k], BH)); RCCHK(dalloc(e, &e->ah_z2[k], BH)); }
  for (int k = 1; k < 4; ++k) {
    sactd3_engine::BatchSlot& S = e->bs[k];
    RCCHK(dalloc(e, &S.X, B * e->ldc)); RCCHK(dalloc(e, &S.Xn, B * e->ldc)); RCCHK(dalloc(e, &S.rew, B)); RCCHK(dalloc(e, &S.done, B));
    RCCHK(dalloc(e, &S.logp_n, B)); RCCHK(dalloc(e, &S.eps_c, std::max<size_t>(B, e->maxn) * e->a)); RCCHK(dalloc(e, &S.idx, B));
  }
  RCCHK(dalloc(e, &e->a_dz2, BH)); RCCHK(dalloc(e, &e->a_dh1, BH)); RCCHK(dalloc(e, &e->a_dz1, BH));
  RCCHK(dalloc(e, &e->qa_ps, 2L * B * 256)); RCCHK(dalloc(e, &e->qa_S, 2L * 16 * 8));
  RCCHK(dalloc(e, &e->a_ps, (long)B * PS_W + HID)); RCCHK(dalloc(e, &e->c_ps, 2L * B * PS_W + 2 * HID));   // (+ the gamma1 snapshot behind them)
  RCCHK(dalloc(e, &e->c_z1, 2 * BH)); RCCHK(dalloc(e, &e->c_xh1, 2 * BH)); RCCHK(dalloc(e, &e->c_h1, 2 * BH)); RCCHK(dalloc(e, &e->c_rs1, 2 * B));
  RCCHK(dalloc(e, &e->c_z2, 2 * BH)); RCCHK(dalloc(e, &e->c_dz2, 2 * BH)); RCCHK(dalloc(e, &e->c_dh1, 2 * BH)); RCCHK(dalloc(e, &e->c_dz1, 2 * BH));
  RCCHK(dalloc(e, &e->t_z1, 2 * BH)); RCCHK(dalloc(e, &e->t_z2, 2 * BH));
  RCCHK(dalloc(e, &e->q, 2 * B)); RCCHK(dalloc(e, &e->qt, 2 * B)); RCCHK(dalloc(e, &e->y, B)); RCCHK(dalloc(e, &e->q_pi, 2 * B));
  RCCHK(dalloc(e, &e->dA, 2 * B * e->a4));
  if (e->B >= BIG_BATCH) RCCHK(dalloc(e, &e->s_h1, 4 * BH));
  if (e->B >= BIG_BATCH) {
    e->gp_slabs = 8;
    RCCHK(dalloc(e, &e->Gp, (size_t)e->gp_slabs * std::max<size_t>(2 * (size_t)e->Lc.size, e->La.size)));
  }
  RCCHK(dalloc(e, &e->part, 2 * (size_t)e->nblk4 * NSLOT * HID)); RCCHK(dalloc(e, &e->part_s, 2 * (size_t)e->nblk4 * 2)); RCCHK(dalloc(e, &e->part_sa, (size_t)e->nblk4 * 2));
  RCCHK(dalloc(e, &e->p_x, (size_t)e->maxn * e->ldo)); RCCHK(dalloc(e, &e->p_z1, (size_t)e->maxn * HID));
  RCCHK(dalloc(e, &e->p_z2, (size_t)e->maxn * HID)); RCCHK(dalloc(e, &e->p_act, (size_t)e->maxn * e->a4));
  RCCHK(halloc(e, &e->h_obs, (size_t)e->maxn * e->ldo)); RCCHK(halloc(e, &e->h_act, (size_t)e->maxn * e->a4));
  RCCHK(halloc(e, &e->h_done, 4));
  RCCHK(halloc(e, &e->h_batch, B * e->rec_f));
  for (int i = 0; i < NSTAGE; ++i) {
    RCCHK(halloc(e, &e->h_stage[i], (size_t)e->stage_rows * e->rec_f));
  }

  std::vector<float> hb(4 * e->a4, 0.f);
  for (int j = 0; j < e->a; ++j) {
    hb[j] = min_ac[j]; hb[e->a4 + j] = max_ac[j];
    hb[2 * e->a4 + j] = (max_ac[j] - min_ac[j]) / 2.0f;   // agents/nets.py:133-136
    hb[3 * e->a4 + j] = (max_ac[j] + min_ac[j]) / 2.0f;
  }
  HIPCHK(hipMemcpy(e->min_ac, hb.data(), sizeof(float) * e->a4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->max_ac, hb.data() + e->a4, sizeof(float) * e->a4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->scale, hb.data() + 2 * e->a4, sizeof(float) * e->a4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->bias, hb.data() + 3 * e->a4, sizeof(float) * e->a4, hipMemcpyHostToDevice));
  DevCtl hc{};
  hc.seed = c.seed;
  hc.pw_q[0] = hc.pw_q[1] = hc.pw_a[0] = hc.pw_a[1] = hc.pw_l[0] = hc.pw_l[1] = 1.0;
  HIPCHK(hipMemcpy(e->ctl, &hc, sizeof(hc), hipMemcpyHostToDevice));
  const float la0[4] = {logf(c.alpha_init), 0.f, 0.f, 0.f};   // agents/agent.py:128
  HIPCHK(hipMemcpy(e->la, la0, sizeof(la0), hipMemcpyHostToDevice));
  // LN gamma = 1 (agents/nets.py:44-47); weights stay 0 until sactd3_set_params
  {
    std::vector<float> ha(e->La.size, 0.f), hq(2 * (size_t)e->Lc.size, 0.f);
    std::fill(ha.begin() + e->La.g1, ha.begin() + e->La.g1 + HID, 1.f);
    std::fill(ha.begin() + e->La.g2, ha.begin() + e->La.g2 + HID, 1.f);
    for (int i = 0; i < 2; ++i) {
      std::fill(hq.begin() + i * e->Lc.size + e->Lc.g1, hq.begin() + i * e->Lc.size + e->Lc.g1 + HID, 1.f);
      std::fill(hq.begin() + i * e->Lc.size + e->Lc.g2, hq.begin() + i * e->Lc.size + e->Lc.g2 + HID, 1.f);
    }
    HIPCHK(hipMemcpy(e->Pa, ha.data(), sizeof(float) * ha.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->Ta, ha.data(), sizeof(float) * ha.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->Pc, hq.data(), sizeof(float) * hq.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->Tc, hq.data(), sizeof(float) * hq.size(), hipMemcpyHostToDevice));
  }
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

int sactd3_create(const sactd3_config* cfg, const float* min_ac, const float* max_ac, sactd3_engine** out) {
  if (!cfg || !out) { g_create_error = "null argument"; return SACTD3_EINVAL; }
  *out = nullptr;
  sactd3_engine* e = new sactd3_engine();
  e->cfg = *cfg;
  const int rc = create_impl(e, min_ac, max_ac);
  if (rc != 0) { g_create_error = e->err; sactd3_destroy(e); return rc; }
  *out = e;
  return 0;
}

// ---- parameters
static bool which_arena(sactd3_engine* e, int which, float** P, float** M, float** V, const NetLayout** L, int* nets, int** t) {
  switch (which) {
    case SACTD3_ACTOR: *P = e->Pa; *M = e->Ma; *V = e->Va; *L = &e->La; *nets = 1; *t = &e->ctl->t_a; return true;
    case SACTD3_CRITICS: *P = e->Pc; *M = e->Mc; *V = e->Vc; *L = &e->Lc; *nets = 2; *t = &e->ctl->t_q; return true;
    case SACTD3_ACTOR_TARGET: *P = e->Ta; *M = *V = nullptr; *L = &e->La; *nets = 1; *t = nullptr; return true;
    case SACTD3_CRITICS_TARGET: *P = e->Tc; *M = *V = nullptr; *L = &e->Lc; *nets = 2; *t = nullptr; return true;
    default: return false;
  }
}

int64_t sactd3_param_count(const sactd3_engine* e, int which) {
  if (!e) return SACTD3_EINVAL;
  switch (which) {
    case SACTD3_ACTOR: case SACTD3_ACTOR_TARGET: return ref_count(e->La, e->cfg.layer_norm);
    case SACTD3_CRITICS: case SACTD3_CRITICS_TARGET: return 2 * ref_count(e->Lc, e->cfg.layer_norm);
    case SACTD3_LOG_ALPHA: return 1;
    default: return SACTD3_EINVAL;
  }
}

static int read_arena(sactd3_engine* e, const float* dev, const NetLayout& L, int nets, float* dst) {
  std::vector<float> h((size_t)nets * L.size);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(h.data(), dev, sizeof(float) * h.size(), hipMemcpyDeviceToHost));
  const int64_t rc = ref_count(L, e->cfg.layer_norm);
  for (int i = 0; i < nets; ++i) unpack_net(L, e->cfg.layer_norm, h.data() + (size_t)i * L.size, dst + i * rc);
  return 0;
}
static int write_arena(sactd3_engine* e, float* dev, const NetLayout& L, int nets, const float* src, bool is_param) {
  std::vector<float> h((size_t)nets * L.size);
  const int64_t rc = ref_count(L, e->cfg.layer_norm);
  for (int i = 0; i < nets; ++i) pack_net(L, e->cfg.layer_norm, src + i * rc, h.data() + (size_t)i * L.size, is_param);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(dev, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  return 0;
}

int sactd3_get_params(sactd3_engine* e, int which, float* dst) {
  if (!e || !dst) return SACTD3_EINVAL;
  USE_DEVICE(e);
  if (which == SACTD3_LOG_ALPHA) {
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(dst, e->la, sizeof(float), hipMemcpyDeviceToHost));
    return 0;
  }
  float *P, *M, *V; const NetLayout* L; int nets; int* t;
  if (!which_arena(e, which, &P, &M, &V, &L, &nets, &t)) return e->fail(SACTD3_EINVAL, "bad `which`");
  return read_arena(e, P, *L, nets, dst);
}

int sactd3_set_params(sactd3_engine* e, int which, const float* src) {
  if (!e || !src) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  if (which == SACTD3_LOG_ALPHA) {
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->la, src, sizeof(float), hipMemcpyHostToDevice));
    return 0;
  }
  float *P, *M, *V; const NetLayout* L; int nets; int* t;
  if (!which_arena(e, which, &P, &M, &V, &L, &nets, &t)) return e->fail(SACTD3_EINVAL, "bad `which`");
  return write_arena(e, P, *L, nets, src, true);
}

int sactd3_get_adam_state(sactd3_engine* e, int which, float* m, float* v, int64_t* step) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  DevCtl hc;
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(&hc, e->ctl, sizeof(hc), hipMemcpyDeviceToHost));
  if (which == SACTD3_LOG_ALPHA) {
    float h[4];
    HIPCHK(hipMemcpy(h, e->la, sizeof(h), hipMemcpyDeviceToHost));
    if (m) *m = h[1];
    if (v) *v = h[2];
    if (step) *step = hc.t_l;
    return 0;
  }
  if (which != SACTD3_ACTOR && which != SACTD3_CRITICS) return e->fail(SACTD3_EINVAL, "no optimiser owns this parameter set");
  float *P, *M, *V; const NetLayout* L; int nets; int* t;
  which_arena(e, which, &P, &M, &V, &L, &nets, &t);
  if (m) RCCHK(read_arena(e, M, *L, nets, m));
  if (v) RCCHK(read_arena(e, V, *L, nets, v));
  if (step) *step = which == SACTD3_ACTOR ? hc.t_a : hc.t_q;
  return 0;
}

int sactd3_set_adam_state(sactd3_engine* e, int which, const float* m, const float* v, int64_t step) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  const int st = (int)step;
  if (which == SACTD3_LOG_ALPHA) {
    if (m) HIPCHK(hipMemcpy(e->la + 1, m, sizeof(float), hipMemcpyHostToDevice));
    if (v) HIPCHK(hipMemcpy(e->la + 2, v, sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(&e->ctl->t_l, &st, sizeof(int), hipMemcpyHostToDevice));
    const double pw[2] = {pow((double)e->cfg.adam_beta1, (double)st), pow((double)e->cfg.adam_beta2, (double)st)};
    HIPCHK(hipMemcpy(e->ctl->pw_l, pw, sizeof(pw), hipMemcpyHostToDevice));
    return 0;
  }
  if (which != SACTD3_ACTOR && which != SACTD3_CRITICS) return e->fail(SACTD3_EINVAL, "no optimiser owns this parameter set");
  float *P, *M, *V; const NetLayout* L; int nets; int* t;
  which_arena(e, which, &P, &M, &V, &L, &nets, &t);
  if (m) RCCHK(write_arena(e, M, *L, nets, m, false));
  if (v) RCCHK(write_arena(e, V, *L, nets, v, false));
  HIPCHK(hipMemcpy(t, &st, sizeof(int), hipMemcpyHostToDevice));
  const double pw[2] = {pow((double)e->cfg.adam_beta1, (double)st), pow((double)e->cfg.adam_beta2, (double)st)};
  HIPCHK(hipMemcpy(which == SACTD3_ACTOR ? e->ctl->pw_a : e->ctl->pw_q, pw, sizeof(pw), hipMemcpyHostToDevice));
  return 0;
}

// ---- replay buffer
static void pack_record(const sactd3_engine* e, float* rec, const float* ob, const float* ac, float rw, const float* nob, uint8_t dn) {
  memset(rec, 0, sizeof(float) * e->rec_f);
  memcpy(rec, ob, sizeof(float) * e->o);
  memcpy(rec + e->o, ac, sizeof(float) * e->a);
  memcpy(rec + e->ldc, nob, sizeof(float) * e->o);
  rec[e->ldc + e->ldo] = rw;
  rec[e->ldc + e->ldo + 1] = dn ? 1.f : 0.f;
}

int sactd3_rb_extend(sactd3_engine* e, const float* obs, const float* act, const float* rew, const float* nobs, const uint8_t* dones, int n) {
  if (!e || !obs || !act || !rew || !nobs || !dones || n < 0) return e ? e->fail(SACTD3_EINVAL, "rb_extend: bad argument") : SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  const int64_t cap = e->cfg.rb_capacity;
  int done_rows = 0;
  while (done_rows < n) {
    const int chunk = (int)std::min<int64_t>(std::min(n - done_rows, e->stage_rows), cap);
    // staging slots are reused round-robin; one stream sync per lap of the ring of slots guarantees that the copy
    // that last read a slot has completed (events / device-written flags cost more per call than this amortised sync)
    if (++e->stage_used == NSTAGE) { HIPCHK(hipStreamSynchronize(e->stream)); e->stage_used = 1; }
    const int slot = e->stage_next; e->stage_next = (e->stage_next + 1) % NSTAGE;
    float* st = e->h_stage[slot];
    for (int i = 0; i < chunk; ++i) {
      const int r = done_rows + i;
      pack_record(e, st + (size_t)i * e->rec_f, obs + (size_t)r * e->o, act + (size_t)r * e->a, rew[r], nobs + (size_t)r * e->o, dones[r]);
    }
    // one launch: the kernel reads the staged rows from pinned host memory, writes them round-robin into the ring
    // (wrapping at capacity) and publishes the new length / cursor
    IngestArgs g{};
    g.src = (const float4*)st; g.ring = (float4*)e->ring; g.rec4 = e->rec4; g.n = chunk; g.cursor = (int)e->rb_cursor; g.cap = (int)cap;
    e->rb_cursor = (e->rb_cursor + chunk) % cap;
    e->rb_len = std::min<int64_t>(cap, e->rb_len + chunk);
    g.len_cursor = &e->ctl->rb_len; g.new_len = (int)e->rb_len; g.new_cursor = (int)e->rb_cursor;
    const int blocks = (int)std::min<long>(256, ((long)chunk * e->rec4 + 255) / 256);
    hipLaunchKernelGGL(k_rb_ingest, dim3(std::max(blocks, 1)), dim3(256), 0, e->stream, g);
    HIPCHK(hipGetLastError());
    done_rows += chunk;
  }
  return 0;
}

int sactd3_rb_layout(const sactd3_engine* e, int32_t out[4]) {
  if (!e || !out) return SACTD3_EINVAL;
  out[0] = e->rec_f; out[1] = e->ldc; out[2] = e->ldo; out[3] = e->cfg.rb_capacity;
  return 0;
}

// rb.extend with packed records that are already in device memory (shared-replay variant: every rank's rows, all-gathered
// over RCCL into one device slab): the same k_rb_ingest launch as the host path, reading the slab instead of a pinned slot.
int sactd3_rb_extend_device(sactd3_engine* e, const float* records, int n) {
  if (!e || !records || n < 0) return e ? e->fail(SACTD3_EINVAL, "rb_extend_device: bad argument") : SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, records) != hipSuccess || at.type != hipMemoryTypeDevice) {
    (void)hipGetLastError();
    return e->fail(SACTD3_EINVAL, "rb_extend_device: `records` is not a device pointer");
  }
  const int64_t cap = e->cfg.rb_capacity;
  int done_rows = 0;
  while (done_rows < n) {
    const int chunk = (int)std::min<int64_t>(n - done_rows, cap);
    IngestArgs g{};
    g.src = (const float4*)(records + (size_t)done_rows * e->rec_f); g.ring = (float4*)e->ring; g.rec4 = e->rec4; g.n = chunk;
    g.cursor = (int)e->rb_cursor; g.cap = (int)cap;
    e->rb_cursor = (e->rb_cursor + chunk) % cap;
    e->rb_len = std::min<int64_t>(cap, e->rb_len + chunk);
    g.len_cursor = &e->ctl->rb_len; g.new_len = (int)e->rb_len; g.new_cursor = (int)e->rb_cursor;
    const int blocks = (int)std::min<long>(1024, ((long)chunk * e->rec4 + 255) / 256);
    hipLaunchKernelGGL(k_rb_ingest, dim3(std::max(blocks, 1)), dim3(256), 0, e->stream, g);
    HIPCHK(hipGetLastError());
    done_rows += chunk;
  }
  return 0;
}

int64_t sactd3_rb_len(const sactd3_engine* e) { return e ? e->rb_len : SACTD3_EINVAL; }

int sactd3_rb_sample(sactd3_engine* e) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  if (e->rb_len <= 0) return e->fail(SACTD3_ESTATE, "rb_sample: buffer is empty");
  e->cur_slot = 0;
  RCCHK(enqueue_gather(e, e->stream, e->ring, -1));
  hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, e->stream, &e->ctl->sample_ctr, (int*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

int sactd3_rb_sample_with_indices(sactd3_engine* e, const int64_t* idx, int n) {
  if (!e || !idx) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  if (n != e->B) return e->fail(SACTD3_EINVAL, "rb_sample_with_indices: n must equal batch_size");
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) {
    if (idx[i] < 0 || idx[i] >= e->rb_len) return e->fail(SACTD3_EINVAL, "rb_sample_with_indices: index out of range");
    h[i] = (int)idx[i];
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(e->idx, h.data(), sizeof(int) * n, hipMemcpyHostToDevice));
  e->cur_slot = 0;
  RCCHK(set_flag(e, &e->ctl->inject_idx, 1));
  RCCHK(enqueue_gather(e, e->stream, e->ring, -1));
  return set_flag(e, &e->ctl->inject_idx, 0);
}

int sactd3_load_batch(sactd3_engine* e, const float* obs, const float* act, const float* rew, const float* nobs, const uint8_t* dones, int n) {
  if (!e || !obs || !act || !rew || !nobs || !dones) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  if (n != e->B) return e->fail(SACTD3_EINVAL, "load_batch: n must equal batch_size");
  HIPCHK(hipStreamSynchronize(e->stream));
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) {
    pack_record(e, e->h_batch + (size_t)i * e->rec_f, obs + (size_t)i * e->o, act + (size_t)i * e->a, rew[i], nobs + (size_t)i * e->o, dones[i]);
    h[i] = i;
  }
  e->cur_slot = 0;
  HIPCHK(hipMemcpy(e->stage_dev, e->h_batch, sizeof(float) * (size_t)n * e->rec_f, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->idx, h.data(), sizeof(int) * n, hipMemcpyHostToDevice));
  RCCHK(set_flag(e, &e->ctl->inject_idx, 1));
  RCCHK(enqueue_gather(e, e->stream, e->stage_dev, n));
  return set_flag(e, &e->ctl->inject_idx, 0);
}

int sactd3_read_batch(sactd3_engine* e, float* obs, float* act, float* rew, float* nobs, uint8_t* dones, int64_t* idx) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  const int B = e->B;
  HIPCHK(hipStreamSynchronize(e->stream));
  std::vector<float> hx((size_t)B * e->ldc), hn((size_t)B * e->ldc), hr(B), hd(B);
  std::vector<int> hi(B);
  const sactd3_engine::BatchSlot& S = e->bs[e->cur_slot];      // the slot of the most recent iteration
  HIPCHK(hipMemcpy(hx.data(), S.X, sizeof(float) * hx.size(), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hn.data(), S.Xn, sizeof(float) * hn.size(), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hr.data(), S.rew, sizeof(float) * B, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hd.data(), S.done, sizeof(float) * B, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hi.data(), S.idx, sizeof(int) * B, hipMemcpyDeviceToHost));
  for (int b = 0; b < B; ++b) {
    if (obs) memcpy(obs + (size_t)b * e->o, hx.data() + (size_t)b * e->ldc, sizeof(float) * e->o);
    if (act) memcpy(act + (size_t)b * e->a, hx.data() + (size_t)b * e->ldc + e->o, sizeof(float) * e->a);
    if (nobs) memcpy(nobs + (size_t)b * e->o, hn.data() + (size_t)b * e->ldc, sizeof(float) * e->o);
    if (rew) rew[b] = hr[b];
    if (dones) dones[b] = hd[b] != 0.f;
    if (idx) idx[b] = hi[b];
  }
  return 0;
}

int sactd3_rb_fill_synthetic(sactd3_engine* e, int64_t n, uint64_t seed) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  if (n < 1 || n > e->cfg.rb_capacity) return e->fail(SACTD3_EINVAL, "rb_fill_synthetic: 1 <= n <= rb_capacity");
  FillArgs f{(float4*)e->ring, e->rec4, e->cx, e->cn, e->o, e->a, (long)n, seed, e->min_ac, e->max_ac};
  const long threads = (long)n * e->rec4;
  hipLaunchKernelGGL(k_rb_fill, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream, f);
  HIPCHK(hipGetLastError());
  e->rb_len = std::max<int64_t>(e->rb_len, n);
  e->rb_cursor = n % e->cfg.rb_capacity;
  return publish_rb_state(e);
}

// ---- noise
int sactd3_set_noise(sactd3_engine* e, int site, const float* eps, int n) {
  if (!e || !eps || site < 0 || site >= SACTD3_NUM_SITES) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  if (n < 1 || n > std::max(e->B, e->maxn)) return e->fail(SACTD3_EINVAL, "set_noise: too many rows");
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(e->eps[site], eps, sizeof(float) * (size_t)n * e->a, hipMemcpyHostToDevice));
  if (site == SACTD3_SITE_CRITIC)      // (every batch slot: injected draws are "sticky until cleared", whichever slot an iteration trains on)
    for (int k = 1; k < 4; ++k) HIPCHK(hipMemcpy(e->bs[k].eps_c, eps, sizeof(float) * (size_t)n * e->a, hipMemcpyHostToDevice));
  const int one = 1;
  HIPCHK(hipMemcpy(&e->ctl->inject_eps[site], &one, sizeof(int), hipMemcpyHostToDevice));
  return 0;
}
int sactd3_clear_noise(sactd3_engine* e, int site) {
  if (!e || site >= SACTD3_NUM_SITES) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  const int zeros[8] = {0};
  if (site < 0) HIPCHK(hipMemcpy(&e->ctl->inject_eps[0], zeros, sizeof(int) * 8, hipMemcpyHostToDevice));
  else HIPCHK(hipMemcpy(&e->ctl->inject_eps[site], zeros, sizeof(int), hipMemcpyHostToDevice));
  return 0;
}
int sactd3_read_noise(sactd3_engine* e, int site, float* eps, int n) {
  if (!e || !eps || site < 0 || site >= SACTD3_NUM_SITES || n < 1 || n > std::max(e->B, e->maxn)) return SACTD3_EINVAL;
  USE_DEVICE(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  const float* src = site == SACTD3_SITE_CRITIC ? e->bs[e->cur_slot].eps_c : e->eps[site];
  HIPCHK(hipMemcpy(eps, src, sizeof(float) * (size_t)n * e->a, hipMemcpyDeviceToHost));
  return 0;
}

// ---- updates
int sactd3_update_qnets(sactd3_engine* e) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  e->cur_slot = 0;
  return run_graph(e, G_Q, [&](hipStream_t s) { return enqueue_update_qnets(e, s, false, nullptr); });
}
int sactd3_update_actor(sactd3_engine* e) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  return run_graph(e, G_A, [&](hipStream_t s) { return enqueue_update_actor(e, s, 0); });
}
int sactd3_update_targ_nets(sactd3_engine* e, int64_t qnet_updates_so_far) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  const bool td3 = e->cfg.prefer_td3_over_sac;
  if (td3 || qnet_updates_so_far % e->cfg.crit_targ_update_freq == 0) return enqueue_polyak(e, e->stream, true, td3);
  return 0;
}
int sactd3_step(sactd3_engine* e, int do_actor) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  if (e->rb_len <= 0) return e->fail(SACTD3_ESTATE, "step: buffer is empty");
  const int64_t updates = e->qnet_updates + 1;     // (the engine's own counter advances only once the iteration has been launched)
  const bool polyak = e->cfg.prefer_td3_over_sac || (updates % e->cfg.crit_targ_update_freq == 0);
  const bool act = do_actor != 0 && e->cfg.actor_update_delay > 0;
  const int which = G_STEP00 + (act ? 2 : 0) + (polyak ? 1 : 0);
  RCCHK(run_graph(e, which, [&](hipStream_t s) { return enqueue_step(e, s, act, polyak); }));
  e->qnet_updates = updates;
  e->cur_slot = 0;
  return 0;
}

// Is the period graph built in its pipelined form (see BatchSlot)?  One or two critic-only iterations behind the one with the actor
// updates, and an opening trunk that reads ring rows itself (narrow observations below the large-batch threshold, wide ones at large batch).
// does the actor's weight-gradient launch take the split-M route (k_tn64 + k_adam_red: no T2 / T3 support)?  (launch_tn's rule)
static bool actor_dw_is_tiled64(const sactd3_engine* e) {
  if (e->B < BIG_BATCH || !e->Gp || e->tune_tn_kt) return false;
  const int tiles = ((e->nh + TN64_N - 1) / TN64_N) * ((HID + TN64_K - 1) / TN64_K) + ((HID + TN64_N - 1) / TN64_N) * ((HID + TN64_K - 1) / TN64_K)
                    + ((HID + TN64_N - 1) / TN64_N) * ((e->La.ld1 + TN64_K - 1) / TN64_K);
  return tiles >= e->tune_tn64_min;
}
static bool period_is_pipelined(const sactd3_engine* e) {
  const sactd3_config& c = e->cfg;
  if (c.actor_update_delay < 1 || c.actor_update_delay > 2 || !opening_trunk_gathers(e)) return false;
  // SAC: needs the temperature draw's trunk + tail pair at the end of the last actor update (autotune);
  // TD3: the run-ahead through the target actors of the next Polyak updates (T2 / T3 in the k_tn epilogue: no gradient clipping,
  // no split-M route for the actor's weight gradients)
  if (c.prefer_td3_over_sac) return c.clip_norm <= 0.f && !actor_dw_is_tiled64(e);
  return c.autotune != 0;
}
// variant (pipelined form only): which batch slot the period's first iteration trains on (0: slot 0, 1: slot 3) -- the other one
// receives the opening pair of the NEXT period, so consecutive periods alternate (chain_ready).  The pipelined form assumes its own
// opening pair is already in place: sactd3_step_period runs the opening graph first when it is not.
static int enqueue_period(sactd3_engine* e, hipStream_t s, int variant = 0) {
  const int n = e->cfg.actor_update_delay + 1;
  if (!period_is_pipelined(e)) {
    for (int i = 0; i < n; ++i) RCCHK(enqueue_step(e, s, i == 0 && e->cfg.actor_update_delay > 0, true, i + 1 < n));
  } else {
    const int P = variant ? 3 : 0, Pnext = variant ? 0 : 3;
    RCCHK(enqueue_step(e, s, true, true, true, P, true, n - 1, Pnext, true));
    for (int i = 1; i < n; ++i) RCCHK(enqueue_step(e, s, false, true, i + 1 < n, i, true));
  }
  if (e->alpha_pending) return e->fail(SACTD3_ESTATE, "step_period: a deferred temperature step was left over");
  return 0;
}
// The first m <= delay iterations of a period (pipelined form): the period graph cut short -- the iteration with the actor updates on
// the precomputed opening pair of slot P, the sampling and next-action passes of the m - 1 critic-only iterations behind it run ahead,
// nothing is left behind for a next period.
static int enqueue_prefix(sactd3_engine* e, hipStream_t s, int variant, int m) {
  const int P = variant ? 3 : 0;
  RCCHK(enqueue_step(e, s, true, true, m > 1, P, true, m - 1, -1, true));
  for (int i = 1; i < m; ++i) RCCHK(enqueue_step(e, s, false, true, i + 1 < m, i, true));
  if (e->alpha_pending) return e->fail(SACTD3_ESTATE, "step_prefix: a deferred temperature step was left over");
  return 0;
}
// the opening pair of a period's first iteration into batch slot `slot`, without its counter ticks (enqueue_update_qnets, open_mode 1)
static int enqueue_opening(sactd3_engine* e, hipStream_t s, int slot) {
  bool policy_done = false;
  return enqueue_update_qnets(e, s, true, nullptr, true, &policy_done, false, nullptr, slot, false, 1);
}

// One period of the actor schedule (orchestrator.py:345-349: iteration i with i % (delay + 1) == 0 runs the actor updates,
// the next `delay` iterations are critic-only) as ONE graph launch: delay + 1 iterations back to back, no host call and no
// inter-replay gap in between.  Only when every iteration takes the same target-update branch (TD3, or crit_targ_update_freq
// == 1); otherwise the caller gets SACTD3_ESTATE and issues the iterations one by one.
int sactd3_step_period(sactd3_engine* e) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  if (e->rb_len <= 0) return e->fail(SACTD3_ESTATE, "step_period: buffer is empty");
  const bool td3 = e->cfg.prefer_td3_over_sac;
  if (!td3 && e->cfg.crit_targ_update_freq != 1) return e->fail(SACTD3_ESTATE, "step_period: needs crit_targ_update_freq == 1");
  const int n = e->cfg.actor_update_delay + 1;
  if (!period_is_pipelined(e)) {
    e->chain_ready = -1;
    RCCHK(run_graph(e, G_PERIOD, [&](hipStream_t s) { return enqueue_period(e, s); }));
    e->cur_slot = 0;
  } else {
    // chained periods: this period's opening pair is either left over from the previous one (chain_ready names the variant) or is
    // produced now by the 2- / 3-node opening graph; the period graph then leaves the NEXT period's behind
    const int v = e->chain_ready >= 0 ? e->chain_ready : 0;
    const bool have = e->chain_ready >= 0;
    e->chain_ready = -1;
    if (!have) RCCHK(run_graph(e, G_OPENING, [&](hipStream_t s) { return enqueue_opening(e, s, 0); }));
    RCCHK(run_graph(e, v ? G_PERIOD_B : G_PERIOD, [&](hipStream_t s) { return enqueue_period(e, s, v); }));
    e->chain_ready = 1 - v;
    e->cur_slot = n - 1;
  }
  e->qnet_updates += n;
  return 0;
}

// The first m iterations of a period (1 <= m <= actor_update_delay) as ONE graph launch: what is left of a run of iterations behind
// its last whole period (orchestrator.py:337-352 for a number of iterations that is not a multiple of the period).  Equal to
// sactd3_step(1) followed by m - 1 sactd3_step(0), bit for bit; in the pipelined form it uses the opening pair the previous period
// left behind (or runs the opening graph) exactly as sactd3_step_period does.  Same preconditions as sactd3_step_period.
int sactd3_step_prefix(sactd3_engine* e, int m) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  if (m < 1 || m > e->cfg.actor_update_delay) return e->fail(SACTD3_EINVAL, "step_prefix: 1 <= m <= actor_update_delay");
  if (e->rb_len <= 0) return e->fail(SACTD3_ESTATE, "step_prefix: buffer is empty");
  if (!e->cfg.prefer_td3_over_sac && e->cfg.crit_targ_update_freq != 1) return e->fail(SACTD3_ESTATE, "step_prefix: needs crit_targ_update_freq == 1");
  if (!period_is_pipelined(e) || m > 2) {      // no cut-short form: the iterations one by one
    for (int i = 0; i < m; ++i) RCCHK(sactd3_step(e, i == 0));
    return 0;
  }
  const int v = e->chain_ready >= 0 ? e->chain_ready : 0;
  const bool have = e->chain_ready >= 0;
  e->chain_ready = -1;
  if (!have) RCCHK(run_graph(e, G_OPENING, [&](hipStream_t s) { return enqueue_opening(e, s, 0); }));
  RCCHK(run_graph(e, G_PREFIX + 2 * (m - 1) + v, [&](hipStream_t s) { return enqueue_prefix(e, s, v, m); }));
  e->cur_slot = m - 1;
  e->qnet_updates += m;
  return 0;
}

// Capture + instantiate the graphs of sactd3_step (both schedules, with the target update) and sactd3_step_period now instead of
// at their first use, without launching anything: a caller that times its first iterations (or must not stall in the loop) calls
// this once after sactd3_create.  No-op with use_graphs == 0.
int sactd3_instantiate_graphs(sactd3_engine* e) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  const bool same_branch = e->cfg.prefer_td3_over_sac || e->cfg.crit_targ_update_freq == 1;
  for (int act = 0; act < 2; ++act)
    for (int pol = same_branch ? 1 : 0; pol < 2; ++pol) {      // (the variants without the target update exist only when it is gated)
      const bool a = act != 0 && e->cfg.actor_update_delay > 0, py = pol != 0;
      RCCHK(run_graph(e, G_STEP00 + (a ? 2 : 0) + (py ? 1 : 0), [&](hipStream_t s) { return enqueue_step(e, s, a, py); }, false));
    }
  if (same_branch && e->cfg.actor_update_delay > 0) {
    RCCHK(run_graph(e, G_PERIOD, [&](hipStream_t s) { return enqueue_period(e, s, 0); }, false));
    if (period_is_pipelined(e)) {
      RCCHK(run_graph(e, G_PERIOD_B, [&](hipStream_t s) { return enqueue_period(e, s, 1); }, false));
      RCCHK(run_graph(e, G_OPENING, [&](hipStream_t s) { return enqueue_opening(e, s, 0); }, false));
      for (int m = 1; m <= e->cfg.actor_update_delay && m <= 2; ++m)
        for (int v = 0; v < 2; ++v)
          RCCHK(run_graph(e, G_PREFIX + 2 * (m - 1) + v, [&](hipStream_t s) { return enqueue_prefix(e, s, v, m); }, false));
    }
  }
  return 0;
}

int sactd3_predict(sactd3_engine* e, const float* obs, int n, int explore, float* actions) {
  if (!e || !obs || !actions) return SACTD3_EINVAL;
  USE_DEVICE(e);
  if (n < 1 || n > e->maxn) return e->fail(SACTD3_EINVAL, "predict: 1 <= n <= max_envs");
  const bool td3 = e->cfg.prefer_td3_over_sac;
  // (the pinned staging is free: every call returns only after its own kernels have finished, see below)
  for (int i = 0; i < n; ++i) {
    memset(e->h_obs + (size_t)i * e->ldo, 0, sizeof(float) * e->ldo);
    memcpy(e->h_obs + (size_t)i * e->ldo, obs + (size_t)i * e->o, sizeof(float) * e->o);
  }
  // The kernels read the observations from, and write the actions to, the pinned host buffers themselves (a few hundred
  // bytes over the host link): two kernels and one synchronisation per call, no copy commands, no separate counter kernel.
  // Nothing in the launches depends on the call but (n, explore): they are captured once per pair and replayed.
  if (e->predict_graphs.empty()) e->predict_graphs.assign(2 * (size_t)(e->maxn + 1), nullptr);
  RCCHK(run_graph_slot(e, &e->predict_graphs[(size_t)(explore ? 1 : 0) * (e->maxn + 1) + n], nullptr, [&](hipStream_t) {
    bool eps_ready = false;
    {
      const TrunkGrp g{e->h_obs, e->Pa, e->p_z1, e->p_z2, nullptr, nullptr, nullptr};
      TrunkTicks tk{nullptr, nullptr, nullptr, nullptr, 0.f};
      if (explore) { tk.nnoise = 1; tk.noise[0] = noise_job(e, SACTD3_SITE_PREDICT, 48u, 0, n); tk.noise_taken = &eps_ready; }
      RCCHK(enqueue_trunk(e, e->stream, e->ldo, e->o, n, e->La, 0, 1, 1, &g, tk));
    }
    const int mode = td3 ? (explore ? 2 : 0) : (explore ? 0 : 1);
    ActorTail t = tail_args(e, e->p_z2, e->Pa, n, mode, 0, SACTD3_SITE_PREDICT, 48u, e->h_act, e->a4, 0, nullptr);
    t.eps_ready = eps_ready;
    // the tail reads predict_ctr (its noise stream) and may only advance it itself when it is a single block: with more
    // rows than one block holds, a late block could read the counter after block 0 has bumped it
    const bool one_block = n <= tail_rows_per_block(t);
    if (explore && one_block) t.tick = &e->ctl->predict_ctr;
    if (one_block) { t.seq = &e->ctl->predict_seq; t.done_flag = e->h_done; }
    RCCHK(launch_tail(e, e->stream, t));
    if (explore && !one_block) {
      hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, e->stream, &e->ctl->predict_ctr, (int*)nullptr);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }));
  // completion: a single-block tail publishes the call's sequence number to a pinned host word after its last store; spin on
  // it (a stream synchronisation costs a marker packet and a signal wake-up on top of the kernels).  Anything unexpected, or a
  // multi-block tail: synchronise the stream.
  bool done = false;
  if (n <= tail_rows_per_block(tail_args(e, e->p_z2, e->Pa, n, 0, 0, SACTD3_SITE_PREDICT, 48u, e->h_act, e->a4, 0, nullptr))) {
    const int want = ++e->predict_calls;
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0; !done; ++spins) {
      done = __atomic_load_n(e->h_done, __ATOMIC_ACQUIRE) == want;
      if (!done && (spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (!done) {
      HIPCHK(hipStreamSynchronize(e->stream));
      if (__atomic_load_n(e->h_done, __ATOMIC_ACQUIRE) != want) return e->fail(SACTD3_ESTATE, "predict: the acting kernels did not report completion");
      done = true;
    }
  }
  if (!done) HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < n; ++i) memcpy(actions + (size_t)i * e->a, e->h_act + (size_t)i * e->a4, sizeof(float) * e->a);
  return 0;
}

int sactd3_read_metrics(sactd3_engine* e, float out[SACTD3_NUM_METRICS]) {
  if (!e || !out) return SACTD3_EINVAL;
  USE_DEVICE(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(out, e->ctl->metrics, sizeof(float) * SACTD3_NUM_METRICS, hipMemcpyDeviceToHost));
  return 0;
}

int sactd3_device_handles(sactd3_engine* e, void** stream, float** metrics) {
  if (!e) return SACTD3_EINVAL;
  if (stream) *stream = (void*)e->stream;
  if (metrics) *metrics = e->ctl->metrics;
  return 0;
}

// Wait for the engine's stream: poll it for up to 2 ms (a blocking hipStreamSynchronize sleeps on an interrupt and wakes up tens of
// microseconds after the last kernel has finished -- as long as an iteration takes), then block.
static int stream_wait(sactd3_engine* e) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int spins = 0;; ++spins) {
    const hipError_t q = hipStreamQuery(e->stream);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) return e->fail(SACTD3_EHIP, "hipStreamQuery", q);
    if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

int sactd3_sync(sactd3_engine* e) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  return stream_wait(e);
}

// ---- introspection
struct DbgEntry { const char* name; const float* ptr; int64_t n; };
static std::vector<DbgEntry> dbg_table(sactd3_engine* e) {
  const int64_t B = e->B, BH = B * HID;
  const sactd3_engine::BatchSlot& S = e->bs[e->cur_slot];
  return {
      {"X", S.X, B * e->ldc}, {"Xn", S.Xn, B * e->ldc}, {"Xp", e->Xp, B * e->ldc}, {"rew", S.rew, B}, {"done", S.done, B},
      {"logp_next", S.logp_n, B}, {"logp_pi", e->logp_pi, B}, {"logp_alpha", e->logp_al, B},
      {"a_xh1", e->a_xh1, BH}, {"a_h1", e->a_h1, BH}, {"a_z2", e->a_z2, BH}, {"a_h2", e->a_h2, BH}, {"a_du", e->a_du, B * e->ldu},
      {"a_dz2", e->a_dz2, BH}, {"a_dh1", e->a_dh1, BH}, {"a_dz1", e->a_dz1, BH},
      {"c_xh1", e->c_xh1, 2 * BH}, {"c_h1", e->c_h1, 2 * BH}, {"c_z2", e->c_z2, 2 * BH}, {"c_dz2", e->c_dz2, 2 * BH},
      {"c_dh1", e->c_dh1, 2 * BH}, {"c_dz1", e->c_dz1, 2 * BH}, {"t_z2", e->t_z2, 2 * BH},
      {"q", e->q, 2 * B}, {"q_target", e->qt, 2 * B}, {"targ_q", e->y, B}, {"q_pi", e->q_pi, 2 * B}, {"dA", e->dA, 2 * B * e->a4},
  };
}
const char* sactd3_debug_names(void) {
  return "X Xn Xp rew done logp_next logp_pi logp_alpha a_xh1 a_h1 a_z2 a_h2 a_du a_dz2 a_dh1 a_dz1 c_xh1 c_h1 c_z2 c_dz2 c_dh1 c_dz1 "
         "t_z2 q q_target targ_q q_pi dA grad_actor grad_critics";
}
int64_t sactd3_debug_read(sactd3_engine* e, const char* name, float* dst, int64_t max_floats) {
  if (!e || !name) return SACTD3_EINVAL;
  USE_DEVICE(e);
  if (!strcmp(name, "grad_actor") || !strcmp(name, "grad_critics")) {   // reference (unpadded) layout
    const bool act = !strcmp(name, "grad_actor");
    const int64_t n = sactd3_param_count(e, act ? SACTD3_ACTOR : SACTD3_CRITICS);
    if (!dst) return n;
    if (max_floats < n) return e->fail(SACTD3_EINVAL, "debug_read: buffer too small");
    RCCHK(read_arena(e, act ? e->Ga : e->Gc, act ? e->La : e->Lc, act ? 1 : 2, dst));
    return n;
  }
  for (const DbgEntry& d : dbg_table(e)) {
    if (strcmp(d.name, name)) continue;
    if (!dst) return d.n;
    if (max_floats < d.n) return e->fail(SACTD3_EINVAL, "debug_read: buffer too small");
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(dst, d.ptr, sizeof(float) * d.n, hipMemcpyDeviceToHost));
    return d.n;
  }
  return e->fail(SACTD3_EINVAL, "debug_read: unknown buffer name");
}

int sactd3_graph_kernel_count(sactd3_engine* e, int which_graph) {
  if (!e) return SACTD3_EINVAL;
  static const int map[8] = {G_Q, G_A, G_STEP01, G_STEP11, G_PERIOD, G_OPENING, G_PREFIX, G_PREFIX + 2};
  if (which_graph < 0 || which_graph > 7) return SACTD3_EINVAL;
  int w = map[which_graph];
  if (!e->graphs[w] && (which_graph == 2 || which_graph == 3)) w -= 1;   // the no-Polyak variant, if that is the one in use
  return e->graph_nodes[w];
}

int sactd3_time_kernel(sactd3_engine* e, const char* kernel, int iters, float* usec) {
  if (!e || !kernel || !usec || iters < 1) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  hipEvent_t t0, t1;
  HIPCHK(hipEventCreate(&t0)); HIPCHK(hipEventCreate(&t1));
  int rc = 0;
  auto body = [&]() -> int {
    if (!strcmp(kernel, "gather")) {   // a fresh index draw per launch (k_tick bumps the sample counter): rows come from HBM, not from the caches
      RCCHK(enqueue_gather(e, e->stream, e->ring, -1));
      hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, e->stream, &e->ctl->sample_ctr, (int*)nullptr);
      HIPCHK(hipGetLastError());
      return 0;
    }
    if (!strcmp(kernel, "polyak")) return enqueue_polyak(e, e->stream, true, e->cfg.prefer_td3_over_sac);
    if (!strcmp(kernel, "trunk_critics")) {   // the 4-net hidden-layer launch of update_qnets (no state is modified)
      const TrunkGrp g[2] = {{e->Xn, e->Tc, e->t_z1, e->t_z2, nullptr, nullptr, nullptr},
                             {e->X, e->Pc, e->c_z1, e->c_z2, e->c_xh1, e->c_h1, e->c_rs1}};
      return enqueue_trunk(e, e->stream, e->ldc, e->o + e->a, e->B, e->Lc, e->Lc.size, 2, 2, g, TrunkTicks{nullptr, nullptr, nullptr, nullptr, 0.f});
    }
    return e->fail(SACTD3_EINVAL, "time_kernel: unknown kernel (gather | polyak | trunk_critics)");
  };
  for (int i = 0; i < 3 && rc == 0; ++i) rc = body();   // warm-up
  if (rc == 0) {
    hipEventRecord(t0, e->stream);
    for (int i = 0; i < iters && rc == 0; ++i) rc = body();
    hipEventRecord(t1, e->stream);
    hipEventSynchronize(t1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, t0, t1);
    *usec = ms * 1000.f / (float)iters;
  }
  hipEventDestroy(t0); hipEventDestroy(t1);
  return rc;
}

// Per-node device time of one fused iteration.  The enqueue sequence of sactd3_step(do_actor) is walked once to list
// its kernel launches (node registry), then each launch alone is issued `iters` times back to back between two HIP
// events on the engine's stream (the other launches of the sequence are skipped).  The learner's state is consumed by
// this (optimiser steps repeat on stale gradients): use a scratch engine.
int sactd3_time_nodes(sactd3_engine* e, int do_actor, int iters, int max_nodes, char* names, int names_cap,
                      float* usec, double* flops, double* bytes, int64_t* threads) {
  if (!e || iters < 1 || max_nodes < 1 || !usec) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  if (e->rb_len <= 0) return e->fail(SACTD3_ESTATE, "time_nodes: buffer is empty");
  const bool act = do_actor != 0 && e->cfg.actor_update_delay > 0;
  const bool period = do_actor == 2 && e->cfg.actor_update_delay > 0 && (e->cfg.prefer_td3_over_sac || e->cfg.crit_targ_update_freq == 1);
  std::vector<NodeInfo> log;
  auto seq = [&]() -> int { e->node_seq = 0; return period ? enqueue_period(e, e->stream) : enqueue_step(e, e->stream, act, true); };
  auto done = [&](int rc) { e->node_only = -1; e->node_log = nullptr; e->node_seq = 0; e->node_role = ""; return rc; };
  e->node_log = &log; e->node_only = 1 << 30;               // list only, launch nothing
  int rc = seq();
  e->node_log = nullptr;
  if (rc != 0) return done(rc);
  const int n = (int)log.size();
  if (n > max_nodes) return done(e->fail(SACTD3_EINVAL, "time_nodes: max_nodes too small"));
  std::string joined;
  for (int k = 0; k < n; ++k) { joined += log[k].name; joined += '\n'; }
  if (names) {
    if ((int)joined.size() + 1 > names_cap) return done(e->fail(SACTD3_EINVAL, "time_nodes: names buffer too small"));
    memcpy(names, joined.c_str(), joined.size() + 1);
  }
  hipEvent_t t0, t1;
  HIPCHK(hipEventCreate(&t0)); HIPCHK(hipEventCreate(&t1));
  // clocks up before anything is timed (an engine is usually created just before this call: the GPU has been idle), then
  // every node in 3 batches of `iters` launches, the fastest batch counting (a batch hit by a clock ramp or by another
  // process's work on the card would otherwise show up as a 100x outlier)
  e->node_only = -1;
  for (int i = 0; i < 60 && rc == 0; ++i) rc = seq();
  for (int k = 0; k < n && rc == 0; ++k) {
    e->node_only = k;
    for (int i = 0; i < 3 && rc == 0; ++i) rc = seq();
    float best = 0.f;
    for (int rep = 0; rep < 3 && rc == 0; ++rep) {
      hipEventRecord(t0, e->stream);
      for (int i = 0; i < iters && rc == 0; ++i) rc = seq();
      hipEventRecord(t1, e->stream);
      if (hipEventSynchronize(t1) != hipSuccess) rc = e->fail(SACTD3_EHIP, "time_nodes: hipEventSynchronize");
      float ms = 0.f;
      hipEventElapsedTime(&ms, t0, t1);
      if (rep == 0 || ms < best) best = ms;
    }
    usec[k] = best * 1000.f / (float)iters;
    if (flops) flops[k] = log[k].flops;
    if (bytes) bytes[k] = log[k].bytes;
    if (threads) threads[k] = log[k].threads;
  }
  hipEventDestroy(t0); hipEventDestroy(t1);
  return done(rc == 0 ? n : rc);
}

int sactd3_time_gather_sweep(sactd3_engine* e, int batch, int iters, float* usec, double* algo_bytes) {
  if (!e || !usec || batch < 1 || iters < 1) return SACTD3_EINVAL;
  USE_DEVICE(e);
  CHAIN_BREAK(e);
  if (e->rb_len <= 0) return e->fail(SACTD3_ESTATE, "gather sweep: buffer is empty");
  if ((long long)batch * e->rec4 >= (1ll << 31)) return e->fail(SACTD3_EINVAL, "gather sweep: batch too large");
  float *X = nullptr, *Xn = nullptr, *rw = nullptr, *dn = nullptr; int* ix = nullptr;
  const size_t rows = batch;
  hipError_t he = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (he == hipSuccess) he = hipMalloc(p, bytes); };
  A((void**)&X, rows * e->ldc * 4); A((void**)&Xn, rows * e->ldc * 4);
  A((void**)&rw, rows * 4); A((void**)&dn, rows * 4); A((void**)&ix, rows * 4);
  int rc = 0;
  if (he != hipSuccess) rc = e->fail(SACTD3_EHIP, "gather sweep: hipMalloc", he);
  if (rc == 0) {
    GatherArgs g{};
    g.ring = (const float4*)e->ring; g.rec4 = e->rec4; g.cx = e->cx; g.cn = e->cn; g.ctl = e->ctl; g.idx = ix;
    g.X = (float4*)X; g.Xn = (float4*)Xn; g.rew = rw; g.done = dn; g.B = batch; g.len_override = -1;
    g.rec4_magic = magic_div((unsigned)e->rec4, (unsigned long long)batch * e->rec4 + 1);
    const dim3 grid(gather_blocks((long)batch * e->rec4));
    g.cpb = (int)(((long)batch * e->rec4 + 256L * grid.x - 1) / (256L * grid.x));
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_gather, grid, dim3(256), 0, e->stream, g);
    hipEventRecord(t0, e->stream);
    for (int i = 0; i < iters; ++i) {
      hipLaunchKernelGGL(k_gather, grid, dim3(256), 0, e->stream, g);
      hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, e->stream, &e->ctl->sample_ctr, (int*)nullptr);
    }
    hipEventRecord(t1, e->stream);
    he = hipEventSynchronize(t1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, t0, t1);
    *usec = ms * 1000.f / (float)iters;
    hipEventDestroy(t0); hipEventDestroy(t1);
    if (he != hipSuccess) rc = e->fail(SACTD3_EHIP, "gather sweep", he);
    // 2*B*T + 4*B with T = 4*(2o+a+1)+1 bytes
    if (algo_bytes) *algo_bytes = 2.0 * batch * (4.0 * (2 * e->o + e->a + 1) + 1.0) + 4.0 * batch;
  }
  hipFree(X); hipFree(Xn); hipFree(rw); hipFree(dn); hipFree(ix);
  return rc;
}

}  // extern "C"
#pragma GCC visibility pop

#ifdef SACTD3_STAMPS
// diagnostic builds only (make stamps; tools/blocks_probe.py): the per-block begin / end stamps of the last stamped launch
extern "C" __attribute__((visibility("default"))) int sactd3_debug_blocks(sactd3_engine* e, long long* out, int n) {
  if (!e || !out || n < 1 || n > 4096) return SACTD3_EINVAL;
  USE_DEVICE(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk), sizeof(long long) * 2 * n));
  return 0;
}
extern "C" __attribute__((visibility("default"))) int sactd3_debug_blocks_select(sactd3_engine* e, int kernel) {
  if (!e) return SACTD3_EINVAL;
  USE_DEVICE(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_blk_kernel), &kernel, sizeof(int)));
  std::vector<long long> z(4096 * 8, 0);
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_blk), z.data(), sizeof(long long) * 4096 * 2));
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_ph), z.data(), sizeof(long long) * 4096 * 8));
  return 0;
}
extern "C" __attribute__((visibility("default"))) int sactd3_debug_phases(sactd3_engine* e, long long* out, int n) {
  if (!e || !out || n < 1 || n > 4096) return SACTD3_EINVAL;
  USE_DEVICE(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ph), sizeof(long long) * 8 * n));
  return 0;
}
#endif
