// Device code of the SAC/TD3 update engine for gfx950 (MI355X, CDNA4).  wave = 64 lanes throughout.
//
// The whole path is latency-bound (B <= 4096 rows, 256-wide layers: everything lives in the Infinity Cache), so every
// kernel is built the same way: issue ALL of its global loads first (one memory round trip; unconditional, address-
// clamped and masked with selects -- see ld4_cols / PIN for what makes the compiler serialise them otherwise), compute
// out of registers / LDS, reduce with DPP (no ds_bpermute), store last, never index a local array dynamically (that is
// scratch memory).
//
//   k_gather          replay ring -> batch slot (float4 record chunks; Philox index drawn once per record into LDS)
//   k_nt / k_nt_wide  Y = pro(A) W^T + b on v_mfma_f32_16x16x4_f32; pro = LayerNorm+ReLU of the producing layer applied to
//                     the A fragments in registers; optionally the FIRST layer (x W1^T + b1) is computed in the same kernel
//                     straight into those fragments (narrow inputs); W tiles parked in LDS, block shape picked per launch
//   k_nt64 / k_ln_fwd large-batch form of the hidden layers: 64 x 64 LDS-tiled GEMM -> LayerNorm row kernel -> tiled GEMM
//   k_nn / k_nn64     dX = dY W                     (16 x 16 tile per block, reduction split over the 4 waves; B >= 1024: LDS-tiled)
//   k_tn              dW = dY^T X for every weight of an update in one launch (+ bias gradient) with the Adam step and the Polyak
//                     update in the tile's epilogue; extra blocks of the launch finalise the LayerNorm-affine / head gradients
//                     from the row kernels' partials, the loss and the noise-counter tick, or lerp TD3's actor target
//   k_tn64 / k_adam_red   large-batch form: split-M partial tiles into slabs -> fixed-order sum + Adam + Polyak
//   k_actor_tail(_s, _s2)  LN+ReLU -> head -> tanh-Gaussian sample + log-prob | TD3 policy; _s: narrow heads, a row per 16
//                     lanes of one wave, no LDS / MFMA; _s2: two such tails in one launch; acting launches publish a completion word
//   k_critic_tail     twin target Q -> min/mix -> entropy -> Bellman target -> twin MSE -> dQ -> head bwd -> LN bwd
//   k_actorq_tail     twin Q(s, pi(s)) -> min -> actor loss -> dQ routing -> head bwd -> LN bwd
//   k_actor_head_bwd(_s)  d(action), d(logp) -> tanh-Gaussian bwd -> head bwd (MFMA; _s: DPP broadcast + FMAs) -> LN bwd
//   k_ln_bwd          LN+ReLU backward of hidden layer 1 (+ the dQ/da slice product of the actor update)
//   k_ctail_nn / k_qtail_nn / k_headbwd_nn   (B < 1024) a row kernel AND the dh1 = dz2 W2 GEMM behind it in one launch: every column-tile
//                     block redoes the tail of its 16 rows; the epilogues leave what the next launch needs of layer 1's LayerNorm
//                     backward (NnFold -> k_tn<.., true>) and of dQ/da (QaFold: per-tile partial sums by MFMA, finished in k_headbwd_nn)
//   k_nt<.., C4>      32-row blocks with the summation tree of the 16-row form (bit-identical): the run-ahead launches of a period graph
//   k_adam / k_polyak / k_alpha_step / k_gradnorm   flat optimiser kernels (stand-alone forms)
// Riding blocks: launches carry independent work as extra blocks (replay gather, N(0,1) draws for the next tail, the previous
// temperature step, target lerps, gradient finalisation) instead of paying a 2 us graph node for it.  Tile -> XCD placement: xcd_tile.
// Row kernels: 16 RPB threads = RPB rows x 16 threads (RPB = 16, or 4 = one wave); thread (row, sub) owns columns {4*sub + 64*q + e}.
// Math follows oracle/manual_grads.py (which is checked against autograd) line by line.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "philox.h"

#define HID 256
#define LN_EPS 1e-5f
#define NSLOT 5            // column-partial slots per row block: 0 dgamma2, 1 dbeta2, 2 dWhead (critic), 3 dgamma1, 4 dbeta1
#define AS 260             // LDS row stride (floats) of 256-wide row tiles: 16-byte aligned, odd in float4 units
#define WS 68              // LDS row stride of 64-wide strips read with ds_read_b32 (WS mod 8 == 4: conflict-free)
#define YS 20              // LDS row stride of 16-wide tiles read with ds_read_b32

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic builds only (tools/kprobe.hip): shader-clock stamps of the LAST block's thread 0 at phase boundaries.
#ifdef SACTD3_STAMPS
__device__ long long g_stamps[16];
__device__ long long g_phase[8];
#define PHASE_DECL long long ph_t = 0, ph_acc[6] = {0, 0, 0, 0, 0, 0}
#define PHASE_START do { __builtin_amdgcn_s_waitcnt(0xc07f); ph_t = clock64(); } while (0)
#define PHASE(i) do { const long long n_ = clock64(); ph_acc[i] += n_ - ph_t; ph_t = n_; } while (0)
#define PHASE_DRAIN(i) do { __builtin_amdgcn_s_waitcnt(0); const long long n_ = clock64(); ph_acc[i] += n_ - ph_t; ph_t = n_; } while (0)
#define PHASE_OUT do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) for (int i_ = 0; i_ < 6; ++i_) g_phase[i_] = ph_acc[i_]; } while (0)
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && blockIdx.z == 0) { __builtin_amdgcn_s_waitcnt(0); g_stamps[i] = clock64(); } } while (0)
// per-block timeline of a launch (tools/blocks_probe.py): begin / end of every block on the 100 MHz device-wide clock
__device__ long long g_blk[4096 * 2];
__device__ long long g_ph[4096 * 8];
__device__ int g_blk_kernel;         // which kernel family records: 0 k_tn, 1 k_nt (sactd3_debug_blocks_select)
#define BLK_MARK_K(k, e) do { if (g_blk_kernel == (k)) { if (e) __syncthreads(); if (threadIdx.x == 0) { const int b_ = blockIdx.x + gridDim.x * blockIdx.z; if (b_ < 4096) { if (e) __builtin_amdgcn_s_waitcnt(0); g_blk[2 * b_ + (e)] = wall_clock64(); g_ph[8 * b_ + 6 + (e)] = clock64(); } } } } while (0)   /* + the shader clock: ph[7] - ph[6] cycles over end - begin */
#define BLK_PH_K(k, i) do { if (g_blk_kernel == (k) && threadIdx.x == 0) { const int b_ = blockIdx.x + gridDim.x * blockIdx.z; if (b_ < 4096) { if ((k) == 0) __builtin_amdgcn_s_waitcnt(0); g_ph[8 * b_ + (i)] = wall_clock64(); } } } while (0)   /* k_nt: issue times only (no drain: its loads are meant to stay in flight) */
#define BLK_MARK(e) BLK_MARK_K(0, e)
#define BLK_PH(i) BLK_PH_K(0, i)
#else
#define BLK_PH(i)
#define BLK_MARK(e)
#define BLK_PH_K(k, i)
#define BLK_MARK_K(k, e)
#define STAMP(i)
#define PHASE_DECL
#define PHASE_START
#define PHASE(i)
#define PHASE_DRAIN(i)
#define PHASE_OUT
#endif

struct DevCtl {
  unsigned long long seed;
  int sample_ctr;     // bumped after every index draw
  int noise_ctr;      // bumped at the end of every update graph
  int predict_ctr;
  int t_q, t_a, t_l;  // Adam step counts (critics, actor, log_alpha)
  int rb_len, rb_cursor;
  int inject_idx;
  int inject_eps[8];
  int predict_seq;    // completed sactd3_predict calls (the acting tail publishes it to a pinned host word, see ActorTail::done_flag)
  float metrics[8];
  float adam_q[2], adam_a[2];   // (lr / (1 - b1^t), sqrt(1 - b2^t)) of the critics' / actor's current step
  double pw_q[2], pw_a[2], pw_l[2];   // running b1^t, b2^t of the three optimisers (no pow() on the critical path)
};

struct NetLayout {   // float offsets inside one net's parameter block (all multiples of 4)
  int K, ld1, nh;
  int W1, b1, g1, be1, W2, b2, g2, be2, Wh, bh;
  int size;
};

// ------------------------------------------------------------------------------------------------ helpers
// exact n / d for n * d < 2^32 with magic = ceil(2^32 / d) (host-computed); magic == 0: plain division
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, unsigned magic) { return magic ? __umulhi(n, magic) : n / d; }
// one thread: advance an optimiser's step count and publish the step's Adam scalars (torch.optim.Adam bias corrections)
// Block -> tile placement that minimises what each XCD has to pull from memory (tools/xcd_affinity.hip: data another XCD wrote in
// the previous kernel comes back at ~2.5 TB/s for the whole chip, ~8 GB/s per CU, where the local L2 delivers it 2-3x faster; the
// rocprofv3 FETCH_SIZE of these launches was 3-7x their operand bytes because every XCD pulled whole operands).  Workgroups are
// dealt round-robin over the 8 XCDs by linear id, so the blocks b = x (mod 8) share an L2: they get a compact (R / xr) x (C / xc)
// sub-grid of the R x C tile grid (xr xc = 8).  A GEMM whose row operand is A bytes and column operand W bytes then fetches
// xc A + xr W in total instead of 8 A + W.  Needs R % xr == 0 and C % xc == 0 (the host checks; xr = 0: row-major numbering).
// n / d for 0 <= n < 2^22, d >= 1: a float reciprocal and one correction step -- the generic 32-bit division is ~40 dependent
// instructions, and the tile decode sits in front of every GEMM-shaped kernel's first load
__device__ __forceinline__ int small_div(int n, int d) {
  int q = (int)((float)n * __frcp_rn((float)d));
  const int r = n - q * d;
  if (r < 0) --q; else if (r >= d) ++q;
  return q;
}
__device__ __forceinline__ void xcd_tile(int b, int R, int C, int xr, int& tr, int& tc) {
  if (xr <= 0) { tr = small_div(b, C); tc = b - tr * C; return; }
  const int lr = __builtin_ctz((unsigned)xr), lc = 3 - lr;      // xr in {1, 2, 4, 8}, xc = 8 / xr: shifts, the host guarantees divisibility
  const int x = b & 7, j = b >> 3;
  const int cl = C >> lc, rl = R >> lr;
  const int gr = x >> lc, gc = x - (gr << lc);
  const int jr = small_div(j, cl);
  tr = gr * rl + jr; tc = gc * cl + (j - jr * cl);
}

__device__ __forceinline__ void adam_tick(int* t, double* pw, float* out, float lr, float b1, float b2) {
  const int tv = *t;
  double q0 = 0.0, q1 = 0.0;
  if (out) { q0 = pw[0]; q1 = pw[1]; }
  *t = tv + 1;
  if (out) {
    const double p1 = q0 * (double)b1, p2 = q1 * (double)b2;
    pw[0] = p1; pw[1] = p2;
    out[0] = (float)((double)lr / (1.0 - p1));
    out[1] = (float)sqrt(1.0 - p2);
  }
}
// The counter work of a launch (done by one thread of block 0): step counter + this step's Adam scalars, and a second
// counter.  Absent targets are pointed at a sink so that all loads go out together, before any of the stores -- three
// dependent load -> store round trips in front of block 0's work are a microsecond of critical path.
__device__ int g_sink_i[2];
__device__ double g_sink_d[2];
__device__ float g_sink_f[2];
__device__ __forceinline__ void tick_all(int* t0, int* t1, double* pw, float* out, float lr, float b1, float b2) {
  const bool pub = t0 != nullptr && out != nullptr;
  t0 = t0 ? t0 : g_sink_i; t1 = t1 ? t1 : g_sink_i + 1;
  pw = pub ? pw : g_sink_d; out = pub ? out : g_sink_f;
  const int a = *t0, b = *t1;
  const double q0 = pw[0], q1 = pw[1];
  const double p1 = q0 * (double)b1, p2 = q1 * (double)b2;
  *t0 = a + 1; *t1 = b + 1;
  pw[0] = p1; pw[1] = p2;
  out[0] = (float)((double)lr / (1.0 - p1));
  out[1] = (float)sqrt(1.0 - p2);
}
// Make a loaded value materialise HERE: without it the compiler sinks an early load into the (divergent) epilogue
// branch that uses it, and every conditional store there then drains the memory queue (s_waitcnt vmcnt(0)) in turn.
#define PIN(x) asm volatile("" : "+v"(x))
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row; every lane of the row gets it
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float lane_bcast(float v, int l) {   // v_readlane takes/returns raw 32-bit words
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float b) { return make_float4(a.x * b, a.y * b, a.z * b, a.w * b); }
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float4 gate4(float4 v, float4 y) {  // v where y > 0 else 0
  return make_float4(y.x > 0.f ? v.x : 0.f, y.y > 0.f ? v.y : 0.f, y.z > 0.f ? v.z : 0.f, y.w > 0.f ? v.w : 0.f);
}
// Columns k .. k+3 of a row that holds Kr = round4(K) floats, zero beyond column K (and everywhere when !row_ok).  The load
// is always issued (address clamped into the row) and masked with selects: a branch on `k < K` around the load makes
// the compiler wait for each load before it issues the next one, which serialises a whole operand fetch.
__device__ __forceinline__ float4 ld4_cols(const float* row, int k, int K, int Kr, bool row_ok = true) {
  const float4 v = ld4(row + min(k, Kr - 4));
  float4 o;
  o.x = (row_ok && k < K) ? v.x : 0.f;
  o.y = (row_ok && k + 1 < K) ? v.y : 0.f;
  o.z = (row_ok && k + 2 < K) ? v.z : 0.f;
  o.w = (row_ok && k + 3 < K) ? v.w : 0.f;
  return o;
}

// The two halves of ld4_cols for software-pipelined loops: the load is issued a whole stage before its value is masked.
// (ld4_cols' selects sit right behind the load, and the compiler then waits for the load THERE -- in k_nt64 / k_tn that put
// every stage's global round trip in front of the stage's MFMAs instead of under them.)
__device__ __forceinline__ float4 ld4_raw(const float* row, int k, int Kr) { return ld4(row + min(k, Kr - 4)); }
__device__ __forceinline__ float4 mask4_cols(float4 v, int k, int K, bool row_ok = true) {
  float4 o;
  o.x = (row_ok && k < K) ? v.x : 0.f;
  o.y = (row_ok && k + 1 < K) ? v.y : 0.f;
  o.z = (row_ok && k + 2 < K) ? v.z : 0.f;
  o.w = (row_ok && k + 3 < K) ? v.w : 0.f;
  return o;
}

// standard normal for element e of (ctr, site): 4 normals per Philox block via two Box-Muller pairs
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned ctr, unsigned site, unsigned e) {
  const Philox4 r = philox4x32_10(ctr, 0u, SACTD3_STREAM_NOISE + site, e >> 2, (uint32_t)seed, (uint32_t)(seed >> 32));
  const unsigned k = e & 3u;
  // (selects, not r.v[k]: a dynamically indexed array would live in scratch memory)
  const float u1 = philox_u01((k & 2u) ? r.v[2] : r.v[0]), u2 = philox_u01((k & 2u) ? r.v[3] : r.v[1]);
  // v_log_f32 / v_sin_f32 / v_cos_f32: the trig units take their argument in revolutions, so sin(2 pi u2) is one instruction
  const float rad = sqrtf(-2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(u1));
  return (k & 1u) ? rad * __builtin_amdgcn_sinf(u2) : rad * __builtin_amdgcn_cosf(u2);
}

// all four normals of Philox block q (elements 4q .. 4q + 3): identical to philox_normal(seed, ctr, site, 4q + k), k = 0..3
__device__ __forceinline__ float4 philox_normal4(unsigned long long seed, unsigned ctr, unsigned site, unsigned q) {
  const Philox4 r = philox4x32_10(ctr, 0u, SACTD3_STREAM_NOISE + site, q, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float ua = philox_u01(r.v[0]), ub = philox_u01(r.v[1]), uc = philox_u01(r.v[2]), ud = philox_u01(r.v[3]);
  const float r0 = sqrtf(-2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(ua));
  const float r1 = sqrtf(-2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(uc));
  return make_float4(r0 * __builtin_amdgcn_cosf(ub), r0 * __builtin_amdgcn_sinf(ub), r1 * __builtin_amdgcn_cosf(ud), r1 * __builtin_amdgcn_sinf(ud));
}
// The N(0,1) draws of one noise site, produced AHEAD of the kernel that consumes them: the actor tail sits on the critical
// path and Philox + Box-Muller is ~1500 cycles behind its first memory round trip (the stream counter), while the trunk
// launch in front of it has idle CUs -- a few extra blocks of that launch fill the site's eps buffer (unless the caller
// injected values there), and the tail just loads them like injected ones.
struct NoiseJob { float* eps; const int* ctr; int ctr_add; unsigned site_code; int site_buf; int n; int blocks; };
__device__ __forceinline__ void noise_body(const NoiseJob& z, const DevCtl* ctl, unsigned local_block) {
  const int inject = ctl->inject_eps[z.site_buf], ctr = *z.ctr;
  const unsigned long long seed = ctl->seed;
  const unsigned q = local_block * 256u + threadIdx.x;
  if (inject || 4 * q >= (unsigned)z.n) return;
  const float4 v = philox_normal4(seed, (unsigned)(ctr + z.ctr_add), z.site_code, q);
  float* d = z.eps + 4 * (long)q;
  d[0] = v.x;
  if (4 * q + 1 < (unsigned)z.n) d[1] = v.y;
  if (4 * q + 2 < (unsigned)z.n) d[2] = v.z;
  if (4 * q + 3 < (unsigned)z.n) d[3] = v.w;
}

// ---- row-owner layout: thread (row = t >> 4, sub = t & 15) holds v[q] = columns 4*sub + 64*q .. +3, q = 0..3
struct Row16 { float4 v[4]; };
__device__ __forceinline__ Row16 row_ld(const float* row, int sub) {
  Row16 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) r.v[q] = ld4(row + 4 * sub + 64 * q);
  return r;
}
__device__ __forceinline__ void row_pin(Row16& r) {      // see PIN
#pragma unroll
  for (int q = 0; q < 4; ++q) { PIN(r.v[q].x); PIN(r.v[q].y); PIN(r.v[q].z); PIN(r.v[q].w); }
}
__device__ __forceinline__ void row_st(float* row, int sub, const Row16& r) {
#pragma unroll
  for (int q = 0; q < 4; ++q) st4(row + 4 * sub + 64 * q, r.v[q]);
}
__device__ __forceinline__ float row_dot(const Row16& a, const Row16& b) {
  return (dot4(a.v[0], b.v[0]) + dot4(a.v[1], b.v[1])) + (dot4(a.v[2], b.v[2]) + dot4(a.v[3], b.v[3]));
}
__device__ __forceinline__ float row_total(const Row16& a) { return (sum4(a.v[0]) + sum4(a.v[1])) + (sum4(a.v[2]) + sum4(a.v[3])); }

// LayerNorm of one row (ln == 0: identity): xhat, pre-ReLU output y, rstd
__device__ __forceinline__ void ln_fwd(const Row16& z, const Row16& g, const Row16& be, int ln, Row16& xh, Row16& y, float& rstd) {
  if (ln) {
    const float mean = row16_sum(row_total(z)) * (1.0f / HID);
    Row16 d;
#pragma unroll
    for (int q = 0; q < 4; ++q) d.v[q] = z.v[q] - f4(mean);
    const float var = row16_sum(row_dot(d, d)) * (1.0f / HID);
    rstd = 1.0f / sqrtf(var + LN_EPS);
#pragma unroll
    for (int q = 0; q < 4; ++q) { xh.v[q] = d.v[q] * rstd; y.v[q] = xh.v[q] * g.v[q] + be.v[q]; }
  } else {
    rstd = 1.f; xh = z; y = z;
  }
}
// LayerNorm backward of one row: dy = gradient at the LN output with the ReLU gate applied
__device__ __forceinline__ Row16 ln_bwd(const Row16& dy, const Row16& xh, float rstd, const Row16& g, int ln) {
  if (!ln) return dy;
  Row16 dxh, dz;
#pragma unroll
  for (int q = 0; q < 4; ++q) dxh.v[q] = dy.v[q] * g.v[q];
  const float m1 = row16_sum(row_total(dxh)) * (1.0f / HID);
  const float m2 = row16_sum(row_dot(dxh, xh)) * (1.0f / HID);
#pragma unroll
  for (int q = 0; q < 4; ++q) dz.v[q] = (dxh.v[q] - f4(m1) - xh.v[q] * m2) * rstd;
  return dz;
}
// Column sums over the RPB rows of a block (blockDim = 16 RPB) for `nslots` quantities; cs = __shared__ float[nslots][RPB][HID].
// Column c of every slot is written to dst[slot][c].  Contains a barrier: every thread of the block calls it.
template <int RPB>
__device__ __forceinline__ void block_colsum(float* cs, const Row16* vals, int nslots, int row, int sub, float* dst) {
  for (int s = 0; s < nslots; ++s) row_st(cs + (s * RPB + row) * HID, sub, vals[s]);
  __syncthreads();
  for (int c = threadIdx.x; c < HID; c += 16 * RPB)
    for (int s = 0; s < nslots; ++s) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < RPB; ++r) a += cs[(s * RPB + r) * HID + c];
      dst[s * HID + c] = a;
    }
}

// ------------------------------------------------------------------------------------------------ replay
struct GatherArgs {
  const float4* ring; int rec4;           // record = rec4 float4 chunks: [s|a : cx][s' : cn][r,d,0,0][pad]
  int cx, cn;                             // chunks of the [s|a] field and of the s' field
  const DevCtl* ctl; int* idx;
  float4* X; float4* Xn;                  // row stride = cx chunks
  float* rew; float* done;
  int B; int len_override;                // len_override >= 0: use it instead of ctl->rb_len (staged batches)
  unsigned rec4_magic;                    // ceil(2^32 / rec4) when B * rec4 * rec4 < 2^32, else 0
  int cpb;                                // chunks per thread = ceil(B * rec4 / (256 * blocks of the launch)), <= GATHER_CPT
  int ctr_add;                            // the draw uses stream counter sample_ctr + ctr_add (a LATER iteration's sample, gathered ahead)
};

#ifndef GATHER_CPT
#define GATHER_CPT 4    // float4 chunks per thread: loads in flight per lane
#endif
// A block moves a CONTIGUOUS span of cpb * 256 chunks (cpb <= GATHER_CPT, chosen by the host with the grid) = a handful of
// records.  Their ring indices are drawn once per record into LDS (Philox is ~150 instructions: drawn per chunk, as a
// one-pass kernel would, the 196 chunks of a Humanoid record made the kernel ALU-bound at half of the copy rate).
__device__ __forceinline__ void gather_body(const GatherArgs& p, unsigned block) {
  __shared__ int ids_s[GATHER_CPT * 256 + 2];
  // one batch of requests for the sampling state (a field read behind a branch on another field is a second round trip)
  const int inject = p.ctl->inject_idx, rb_len = p.ctl->rb_len, ctr = p.ctl->sample_ctr + p.ctr_add;
  const unsigned long long seed = p.ctl->seed;
  const int len = p.len_override >= 0 ? p.len_override : rb_len;
  const unsigned total = (unsigned)p.B * (unsigned)p.rec4;             // host guarantees B * rec4 < 2^31
  const unsigned c0 = block * (unsigned)p.cpb * 256u;
  if (c0 >= total) return;                                             // (block-uniform)
  const unsigned c1 = min(c0 + (unsigned)p.cpb * 256u, total);
  const unsigned r0 = fast_div(c0, (unsigned)p.rec4, p.rec4_magic), r1 = fast_div(c1 - 1u, (unsigned)p.rec4, p.rec4_magic);
  for (unsigned i = threadIdx.x; i <= r1 - r0; i += 256u)
    ids_s[i] = inject ? p.idx[r0 + i] : (int)philox_index(seed, (unsigned)ctr, r0 + i, (unsigned)len);
  __syncthreads();
  int bb[GATHER_CPT], cc[GATHER_CPT], ids[GATHER_CPT]; float4 v[GATHER_CPT]; bool on[GATHER_CPT], first[GATHER_CPT];
  // All loads first, all stores afterwards: a store that may alias a later load makes the compiler drain the memory
  // queue (s_waitcnt vmcnt(0)) in between, which would serialise the record fetches.
#pragma unroll
  for (int u = 0; u < GATHER_CPT; ++u) {            // consecutive threads -> consecutive chunks of a record
    const unsigned g = c0 + (unsigned)u * 256u + threadIdx.x;
    on[u] = u < p.cpb && g < c1;
    const unsigned q = on[u] ? fast_div(g, (unsigned)p.rec4, p.rec4_magic) : r0;
    bb[u] = (int)q; cc[u] = on[u] ? (int)(g - q * (unsigned)p.rec4) : 0;
    ids[u] = ids_s[q - r0];
    first[u] = !inject && on[u] && cc[u] == 0;
    on[u] = on[u] && cc[u] <= p.cx + p.cn;           // trailing pad chunk(s) are not moved
  }
#pragma unroll
  for (int u = 0; u < GATHER_CPT; ++u) v[u] = p.ring[(long)ids[u] * p.rec4 + (on[u] ? cc[u] : 0)];
#pragma unroll
  for (int u = 0; u < GATHER_CPT; ++u) {
    if (first[u]) p.idx[bb[u]] = ids[u];
    if (!on[u]) continue;
    const int b = bb[u], c = cc[u];
    if (c < p.cx) p.X[(long)b * p.cx + c] = v[u];
    else if (c < p.cx + p.cn) p.Xn[(long)b * p.cx + (c - p.cx)] = v[u];
    else { p.rew[b] = v[u].x; p.done[b] = v[u].y; }
  }
}
__global__ __launch_bounds__(256) void k_gather(GatherArgs p) { gather_body(p, blockIdx.x); }

__global__ void k_tick(int* a, int* b) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { if (a) *a += 1; if (b) *b += 1; }
}
__global__ void k_set_int2(int* dst, int v0, int v1) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { dst[0] = v0; dst[1] = v1; }
}

// rb.extend (orchestrator.py:100-113): rows packed by the host into a pinned staging slot are copied into the ring by the
// kernel itself (round-robin from `cursor`, wrapping at `cap`) and the new length / cursor are published with them --
// one launch per call instead of a copy command plus a kernel.
struct IngestArgs { const float4* src; float4* ring; int rec4, n, cursor, cap; int* len_cursor; int new_len, new_cursor; };
__global__ __launch_bounds__(256) void k_rb_ingest(IngestArgs p) {
  const int total = p.n * p.rec4;
  for (int g = blockIdx.x * 256 + threadIdx.x; g < total; g += gridDim.x * 256) {
    const int row = g / p.rec4, c = g - row * p.rec4;
    int dst = p.cursor + row;
    if (dst >= p.cap) dst -= p.cap;
    p.ring[(long)dst * p.rec4 + c] = p.src[g];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { p.len_cursor[0] = p.new_len; p.len_cursor[1] = p.new_cursor; }
}

struct FillArgs { float4* ring; int rec4, cx, cn, o, a; long n; unsigned long long seed; const float* min_ac; const float* max_ac; };
__global__ __launch_bounds__(256) void k_rb_fill(FillArgs p) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = g / p.rec4; const int c = (int)(g % p.rec4);
  if (row >= p.n) return;
  const Philox4 r = philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), SACTD3_STREAM_FILL, (uint32_t)c,
                                  (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
  float nrm[4], uni[4];
  {
    const float u0 = philox_u01(r.v[0]), u1 = philox_u01(r.v[1]), u2 = philox_u01(r.v[2]), u3 = philox_u01(r.v[3]);
    const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
    nrm[0] = r0 * cosf(6.283185307179586f * u1); nrm[1] = r0 * sinf(6.283185307179586f * u1);
    nrm[2] = r1 * cosf(6.283185307179586f * u3); nrm[3] = r1 * sinf(6.283185307179586f * u3);
    uni[0] = u0; uni[1] = u1; uni[2] = u2; uni[3] = u3;
  }
  float out[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < p.cx) {
    for (int i = 0; i < 4; ++i) {
      const int e = 4 * c + i;
      if (e < p.o) out[i] = nrm[i];
      else if (e < p.o + p.a) { const int j = e - p.o; out[i] = p.min_ac[j] + (p.max_ac[j] - p.min_ac[j]) * uni[i]; }
    }
  } else if (c < p.cx + p.cn) {
    for (int i = 0; i < 4; ++i) if (4 * (c - p.cx) + i < p.o) out[i] = nrm[i];
  } else if (c == p.cx + p.cn) {
    out[0] = nrm[0];
    out[1] = uni[1] < 0.01f ? 1.f : 0.f;
  }
  p.ring[row * p.rec4 + c] = make_float4(out[0], out[1], out[2], out[3]);
}

// ------------------------------------------------------------------------------------------------ GEMMs
// Operand maps of v_mfma_f32_16x16x4_f32 (lane l): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
// D: col j = l&15, row i = 4*(l>>4) + reg.  The k slot of a lane may name ANY k as long as A and B agree, so
// lane-group kq takes 4 CONSECUTIVE k's (one 16-byte read) and feeds them to 4 successive MFMAs.
__device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }   // (i is a constant after unrolling)
#define MFMA4(acc, a, b)                                                   \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).x, (b).x, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).y, (b).y, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).z, (b).z, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).w, (b).w, acc, 0, 0, 0);

struct AlphaArgs {
  const float* logp; int B; float targ_ent; int autotune;
  float* la;                             // [0] log_alpha, [1] exp_avg, [2] exp_avg_sq
  DevCtl* ctl; float lr, b1, b2, eps;
  int* tick;
};
// (a device function: also runs as one extra block of the next update's critic-trunk launch, see NtArgs::al; in a block of more
//  than 256 threads only the first 256 take part, so the sum is partitioned -- and rounds -- exactly as in a 256-thread block)
__device__ __forceinline__ void alpha_body(const AlphaArgs& a) {
  __shared__ float red[8];
  // the lead thread's state is requested together with the log-probs: fetched one after the other behind the
  // reduction it would be four dependent round trips in a kernel that does nothing else
  const bool lead = threadIdx.x == 0;
  float la = 0.f, m = 0.f, v = 0.f; int tl = 0, tk = 0; double q0 = 0.0, q1 = 0.0;
  if (lead) {
    la = a.la[0];
    if (a.autotune) { m = a.la[1]; v = a.la[2]; tl = a.ctl->t_l; q0 = a.ctl->pw_l[0]; q1 = a.ctl->pw_l[1]; }
    if (a.tick) tk = *a.tick;
  }
  float s = 0.f;
  if (a.autotune && threadIdx.x < 256)
    for (int i = threadIdx.x; i < a.B; i += 256) s += -a.logp[i] - a.targ_ent;
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) & 7] = s;
  __syncthreads();
  if (lead) {
    if (a.autotune) {
      const float mean_term = ((red[0] + red[1]) + (red[2] + red[3])) / (float)a.B;
      const float g = expf(la) * mean_term;          // d/dlog_alpha of alpha * mean_term; also the loss value
      a.ctl->metrics[2] = g;
      const double p1 = q0 * (double)a.b1, p2 = q1 * (double)a.b2;      // adam_tick on (t_l, pw_l)
      a.ctl->t_l = tl + 1; a.ctl->pw_l[0] = p1; a.ctl->pw_l[1] = p2;
      const float sc0 = (float)((double)a.lr / (1.0 - p1)), sc1 = (float)sqrt(1.0 - p2);
      m = m + (g - m) * (1.0f - a.b1);
      v = v * a.b2 + g * g * (1.0f - a.b2);
      la -= sc0 * (m / (sqrtf(v) / sc1 + a.eps));
      a.la[0] = la; a.la[1] = m; a.la[2] = v;
    }
    a.ctl->metrics[3] = expf(la);
    if (a.tick) *a.tick = tk + 1;
  }
}
__global__ __launch_bounds__(256) void k_alpha_step(AlphaArgs a) { alpha_body(a); }

struct NtGrp {               // one group of nets sharing an input and a parameter arena
  const float* in;                          // input rows: A (plain) or x (FUSE1); net stride in_ns
  const float* P;                           // parameter arena of the group's first net; net stride p_ns
  float* Y;                                 // output, net stride y_ns
  float* xh_out; float* h_out; float* rstd_out;   // optional stores of the prologue rows by the column-tile-0 blocks
  // ring != 0: the input rows of this group are read straight from the replay ring -- row m = the record drawn for batch row m with
  // stream counter sample_ctr + sctr_add (0: this iteration's sample; > 0: a later iteration's, whose next-action pass runs ahead),
  // field at float offset ring_off; ring_idx = where the (injected) indices of that sample live
  int ring; int ring_off; int sctr_add; const int* ring_idx;
};
struct NtArgs {              // Y[M,N] = pro(A)[M,K] * W[N,K]^T + bias ; block = one 16 x 16 output tile, K split over the 4 waves
  NtGrp g[5]; int npg;                      // blockIdx.z = grp * npg + net-in-group
  int ld_in; long in_ns;
  int oW, ldw, oBias, oG, oBe; long p_ns;   // offsets inside a net's parameter block: W [N][ldw], bias (-1: none), LN affine of the A rows
  int ldy; long y_ns;
  int M, N, K;                              // PRO != 0 or FUSE1: K == 256
  int K1, oW1, ldw1, oB1;                   // FUSE1: A rows = x W1^T + b1 computed here (x: [M][ld_in], K1 <= 64)
  long act_ns;                              // net stride of xh_out / h_out ([M][256]); rstd_out is [nets][M]
  int* tick0; int* tick1;                   // optional counters bumped by (block 0, thread 0, net 0)
  float* adam_out; double* adam_pw; float lr, b1, b2;   // with tick0: publish this step's Adam scalars
  unsigned w1_magic;                        // ceil(2^32 / ldw1) for the W1 parking index arithmetic
  // Fused replay sampling: groups with NtGrp::ring read their rows straight from the replay ring, and `gblocks` extra blocks at the
  // end of the grid do the k_gather copy into the batch slot for the later kernels -- the gather leaves the critical path.  Up to three
  // gathers ride (gb_each blocks each: the samples of the two critic-only iterations of a period and of the NEXT period's first
  // iteration, into their own batch slots).
  int nt_blocks; int gblocks; int gb_each;
  int flat, flat_r, flat_n;              // (below: NtArgs::flat) nets, rider slots (padded), riders -- beside nt_blocks: one scalar-cache line
  GatherArgs ga[3];
  // a second step counter + Adam scalars (a launch that opens the critic AND the actor update): done by thread 64 of block 0
  int* tick0b; float* adam_out_b; double* adam_pw_b; float lr_b;
  int xr;                                // XCD row-block groups of the tile placement (xcd_tile; unfused launches), 0 = row-major numbering
  int ln_pro;                            // k_nt64_ln: 1 = LayerNorm + ReLU of the input rows, 0 = ReLU only (layer_norm off)
  // One more block (after the gather blocks, net 0 only) can carry the temperature step of the PREVIOUS actor update
  // (k_alpha_step's body): nothing in this launch reads log_alpha or the noise counter, the next kernel does.
  int alpha_block; AlphaArgs al;
  // ... and the noise draws of the actor tail that follows this launch (see NoiseJob): nz_n jobs, nz[i].blocks blocks each
  int nz_n; NoiseJob nz[5]; const DevCtl* nz_ctl;
  // flat > 0 (k_nt, launches of several nets that carry riders): a 1-D grid -- flat_r rider slots FIRST (a multiple of 8: the tiles keep
  // their XCDs; riders are the long latency chains of a launch, gathers out of a cold ring, and should start at once), then flat x
  // nt_blocks tile blocks, net-major.  In the 3-D grid every net gets the riders' x slots (empty blocks for net > 0), and a 4-net
  // launch of exactly one block per CU then has 260 blocks: four CUs run two TILE blocks side by side and finish 2 us after the rest;
  // here the blocks that share a CU share it with a rider that is gone after a microsecond or two (fields: beside nt_blocks)
};

// Sum the 4 waves' accumulators of a block (split-K); the total is returned in wave 0.  Contains a barrier.
__device__ __forceinline__ f32x4 splitk_reduce(float* red /*[4][64][4]*/, f32x4 acc, int wave, int lane) {
  st4(red + (wave * 64 + lane) * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
  __syncthreads();
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (wave == 0) {
    const float4 a = ld4(red + lane * 4), b = ld4(red + (64 + lane) * 4), c = ld4(red + (128 + lane) * 4), d = ld4(red + (192 + lane) * 4);
    o[0] = (a.x + b.x) + (c.x + d.x); o[1] = (a.y + b.y) + (c.y + d.y);
    o[2] = (a.z + b.z) + (c.z + d.z); o[3] = (a.w + b.w) + (c.w + d.w);
  }
  return o;
}

// riding block x (behind the gather blocks) of a trunk launch: the pending temperature step, then the noise jobs
__device__ __forceinline__ void riding_body(const NtArgs& p, int x) {
  if (x < p.alpha_block) { alpha_body(p.al); return; }
  x -= p.alpha_block;
  if (x < p.nz[0].blocks) { noise_body(p.nz[0], p.nz_ctl, x); return; }
  x -= p.nz[0].blocks;
  if (p.nz_n > 1 && x < p.nz[1].blocks) { noise_body(p.nz[1], p.nz_ctl, x); return; }
  x -= p.nz[1].blocks;
  if (p.nz_n > 2 && x < p.nz[2].blocks) { noise_body(p.nz[2], p.nz_ctl, x); return; }
  x -= p.nz[2].blocks;
  if (p.nz_n > 3 && x < p.nz[3].blocks) { noise_body(p.nz[3], p.nz_ctl, x); return; }
  x -= p.nz[3].blocks;
  if (p.nz_n > 4) noise_body(p.nz[4], p.nz_ctl, x);
}
// riding gather block x of a trunk launch (up to three gathers of gb_each blocks)
__device__ __forceinline__ void riding_gather(const NtArgs& p, int x) {
  if (x < p.gb_each) gather_body(p.ga[0], x);
  else if (x < 2 * p.gb_each) gather_body(p.ga[1], x - p.gb_each);
  else gather_body(p.ga[2], x - 2 * p.gb_each);
}

// PRO: 0 none, 1 LayerNorm+ReLU, 2 ReLU.  FUSE1: the A rows are produced by a fused first layer.
// KS: how many waves split K for one 16 x 16 tile.  A block always has 4 waves and one 16-column tile of W:
//   KS = 4 -> 16 rows per block, KS = 2 -> 32 rows, KS = 1 -> 64 rows (each wave a full-K tile of its own rows).
// The host picks KS so that a launch has about one block per CU: what a CU can fetch per clock (~13 B coalesced,
// ~7 B in 64-byte pieces) bounds these kernels, so W tiles (and the whole of W1) are fetched ONCE per block with
// fully coalesced loads, parked in LDS and shared by the block's waves.
// NT = 16-column tiles per block: 2 halves the number of blocks that recompute the fused first layer (4-net launches, where
// NT = 1 means two rounds of blocks per CU)
// C4 (KS == 2 only): 32 rows per block with the SUMMATION TREE of the KS = 4 form -- every wave treats its 8 k chunks as the two
// groups of 4 that two waves of a KS = 4 block would hold (row statistics: two (mean, M2) partials per lane; second layer: two
// partial accumulators per wave, summed across the block in the KS = 4 order), so its results are bit-identical to a KS = 4
// launch of the same rows.  The run-ahead launches of a period graph (up to 5 groups: 1 280 blocks of 16 rows, each fetching the
// whole of W1 and a W2 tile) use it: they must reproduce, bit for bit, what the single-net KS = 4 launches of those passes compute.
template <int PRO, bool FUSE1, int KS, int C1, int NT = 1, bool C4 = false>      // C1 = 16-wide k chunks of the fused first layer (1, 2 or 4)
__global__ __launch_bounds__(256) void k_nt(NtArgs p) {
  static_assert(!C4 || (KS == 2 && PRO == 1), "C4: the KS = 2 LayerNorm form only");
  constexpr int RB = 64 / KS, CW = 16 / KS, NP = C4 ? 16 : 4 * KS, SS = 2 * NP + 4;   // rows/block, k chunks/wave, stat partials/row
  constexpr int W1S = 16 * (C1 > 0 ? C1 : 1) + 4;                           // LDS row stride of W1 (K1 <= 16 C1)
  __shared__ __attribute__((aligned(16))) float W2s[NT * 16 * AS];
  __shared__ __attribute__((aligned(16))) float W1s[FUSE1 ? HID * W1S : 4];
  __shared__ __attribute__((aligned(16))) float vec[3 * HID];               // b1 | gamma | beta
  __shared__ __attribute__((aligned(16))) float stat[RB * SS];
  // the cross-wave sums of the second layer go where W1 was (fused form: W1 is dead behind the first layer, a barrier ago): the
  // 5-group run-ahead launch needs 3 blocks per CU to be one round, and 52 KB + the riders' arrays is a few hundred bytes too many
  constexpr int RED = KS > 1 ? (C4 ? 2 : 1) * NT * 4 * 64 * 4 : 4;
  static_assert(!FUSE1 || RED <= HID * W1S, "red fits W1's slab");
  __shared__ __attribute__((aligned(16))) float red_own[FUSE1 ? 4 : RED];
  float* red = FUSE1 ? W1s : red_own;
  BLK_MARK_K(1, 0);
  int bx = blockIdx.x, net = blockIdx.z;                      // tile block within the net / net (3-D grid)
  bool rider_net0 = blockIdx.z == 0;
  if (p.flat) {                                               // 1-D grid: flat_r rider slots, then flat x nt_blocks tile blocks
    rider_net0 = true;
    if (bx < p.flat_r) { if (bx >= p.flat_n) return; bx += p.nt_blocks; net = 0; }      // (padding slots exit)
    else { bx -= p.flat_r; net = small_div(bx, p.nt_blocks); bx -= net * p.nt_blocks; }
  }
  if ((p.gblocks || p.alpha_block || p.nz_n) && bx >= p.nt_blocks) {   // block-uniform: replay gather / temperature step / noise
    const int x = bx - p.nt_blocks;
    if (x < p.gblocks) { if (rider_net0) riding_gather(p, x); }
    else if (!rider_net0) { }
    else riding_body(p, x - p.gblocks);
    BLK_MARK_K(1, 1);
    return;
  }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int mt = wave / KS, ks = wave % KS;
  if (bx == 0 && net == 0) {
    if (t == 0 && (p.tick0 || p.tick1)) tick_all(p.tick0, p.tick1, p.adam_pw, p.adam_out, p.lr, p.b1, p.b2);
    if (t == 64 && p.tick0b) tick_all(p.tick0b, nullptr, p.adam_pw_b, p.adam_out_b, p.lr_b, p.b1, p.b2);
  }
  const int grp = net / p.npg, ni = net - grp * p.npg;
  const NtGrp G = p.g[grp];
  const float* Pn = G.P + ni * p.p_ns;
  const int tiles_n = (p.N + 16 * NT - 1) / (16 * NT);
  int tmb, tn;
  xcd_tile(bx, (p.M + RB - 1) / RB, tiles_n, FUSE1 ? 0 : p.xr, tmb, tn);
  const int m0 = tmb * RB + 16 * mt, n0 = tn * 16 * NT;          // this wave's rows / the block's columns
  const int mrow = min(m0 + r, p.M - 1);
  const int kb = (ks * CW) * 16 + 4 * kq;                    // first k of this lane's fragments
  // requested here, used by the epilogue: a load issued among the epilogue's stores makes each store wait for the last
  float bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias[nt] = p.oBias >= 0 ? Pn[p.oBias + min(n0 + 16 * nt + r, p.N - 1)] : 0.f;
  STAMP(0); BLK_PH_K(1, 0);
  // ---- 1. every global load, coalesced where the data is shared by the block
  float4 w2r[4 * NT], w1r[C1 > 0 ? 4 * C1 : 1], vr = f4(0.f), xv[C1 > 0 ? C1 : 1], av[CW];
  const int w1n = FUSE1 ? (HID * p.ldw1) >> 2 : 0;           // float4s of W1 (rows are 16-byte multiples, contiguous)
  if (FUSE1) {
#pragma unroll
    for (int u = 0; u < 4 * C1; ++u) {                        // ldw1 <= 16 C1 floats -> at most 4 C1 float4 per thread
      const int f = min(t + 256 * u, w1n - 1);               // clamped, not predicated: parked only where t + 256 u < w1n
      w1r[u] = ld4(Pn + p.oW1 + 4 * (long)f);
    }
    const float* xrow = G.in + ni * p.in_ns + (long)mrow * p.ld_in;
    if (G.ring) {                                            // (block-uniform) this sample's record in the replay ring
      const int inject = p.ga[0].ctl->inject_idx, rb_len = p.ga[0].ctl->rb_len, sctr = p.ga[0].ctl->sample_ctr + G.sctr_add;   // one batch of requests
      const unsigned long long seed = p.ga[0].ctl->seed;
      int id = (int)philox_index(seed, (unsigned)sctr, (unsigned)mrow, (unsigned)max(rb_len, 1));
      if (inject) id = G.ring_idx[mrow];
      xrow = reinterpret_cast<const float*>(p.ga[0].ring) + (long)id * (4 * p.ga[0].rec4) + G.ring_off;
    }
#pragma unroll
    for (int c1 = 0; c1 < C1; ++c1) {
      const int k = 16 * c1 + 4 * kq;
      xv[c1] = ld4_cols(xrow, k, p.K1, (p.K1 + 3) & ~3);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CW; ++c) av[c] = ld4(G.in + ni * p.in_ns + (long)mrow * p.ld_in + kb + 16 * c);
  }
  if (t < 192) {                                             // b1 | gamma | beta as one 3 x 256 vector
    const int which = t >> 6, c4 = t & 63;
    const int off = which == 0 ? p.oB1 : (which == 1 ? p.oG : p.oBe);
    if ((which == 0 && FUSE1) || (which != 0 && PRO == 1)) vr = ld4(Pn + off + 4 * c4);
  }
  // the second layer's W tile is requested LAST (loads return in order): the first layer and the row statistics run
  // while it is still in flight
#pragma unroll
  for (int u = 0; u < 4 * NT; ++u) {                         // 16 NT rows x 256 floats, one row per wave-instruction
    const int i = t + 256 * u, row = i >> 6, c4 = i & 63, n = min(n0 + row, p.N - 1);
    w2r[u] = ld4(Pn + p.oW + (long)n * p.ldw + 4 * c4);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 2. park the shared operands in LDS (W1 first: the first layer only needs W1, x, b1)
  if (FUSE1) {
#pragma unroll
    for (int u = 0; u < 4 * C1; ++u) {
      const int f = t + 256 * u;
      if (f < w1n) { const int e = 4 * f, row = (int)fast_div((unsigned)e, (unsigned)p.ldw1, p.w1_magic), col = e - row * p.ldw1; st4(W1s + row * W1S + col, w1r[u]); }
    }
  }
  if (t < 192) st4(vec + 4 * t, vr);
  __syncthreads();
  STAMP(1); BLK_PH_K(1, 1);
  if (FUSE1) {
    // first layer as a transposed product, D[i = n1][j = m] = sum_k W1[n1][k] x[m][k]: lane (m = r, kq) receives
    // z1[m][16 (ks CW + c) + 4 kq .. +3] -- exactly its A fragment of chunk c for the second layer
    // four tiles at a time: their W1 fragments are requested together and the MFMAs of the four (independent) accumulators alternate,
    // so neither an LDS read nor the previous MFMA of the same tile is waited for in front of every instruction
    constexpr int GT = CW < 4 ? CW : 4;
#pragma unroll
    for (int c0 = 0; c0 < CW; c0 += GT) {
      float4 wf[GT][C1 > 0 ? C1 : 1], b1[GT];
      f32x4 z[GT];
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        const int tile = ks * CW + c0 + j;
#pragma unroll
        for (int c1 = 0; c1 < C1; ++c1) {
          const int k = 16 * c1 + 4 * kq;
          wf[j][c1] = k < p.ldw1 ? ld4(W1s + (tile * 16 + r) * W1S + k) : f4(0.f);
        }
        b1[j] = ld4(vec + tile * 16 + 4 * kq);
        z[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int c1 = 0; c1 < C1; ++c1)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < GT; ++j) z[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(wf[j][c1], q), f4c(xv[c1], q), z[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < GT; ++j) av[c0 + j] = make_float4(z[j][0] + b1[j].x, z[j][1] + b1[j].y, z[j][2] + b1[j].z, z[j][3] + b1[j].w);
    }
  }
#pragma unroll
  for (int u = 0; u < 4 * NT; ++u) { const int i = t + 256 * u; st4(W2s + (i >> 6) * AS + 4 * (i & 63), w2r[u]); }
  // ---- 3. prologue on the A rows held in registers
  float4 xh[CW];
  float rstd = 1.f;
  if (PRO == 1) {
    // this lane holds 4 CW of the row's 256 values; combine the NP (mean, M2) partials of the row (Chan et al.)
    constexpr int CG = C4 ? 4 : CW;                          // chunks per (mean, M2) partial
    constexpr float nl = 4.0f * CG;
    float* srow = stat + (16 * mt + r) * SS;
#pragma unroll
    for (int h = 0; h < CW / CG; ++h) {
      float ml = 0.f;
#pragma unroll
      for (int c = 0; c < CG; ++c) ml += sum4(av[CG * h + c]);
      ml *= (1.0f / nl);
      float m2 = 0.f;
#pragma unroll
      for (int c = 0; c < CG; ++c) { const float4 d = av[CG * h + c] - f4(ml); m2 += dot4(d, d); }
      const int slot = (C4 ? 2 * ks + h : ks) * 4 + kq;      // (C4: the slot of wave 2 ks + h of a KS = 4 block)
      srow[2 * slot] = ml;
      srow[2 * slot + 1] = m2;
    }
    __syncthreads();
    float4 sp[NP / 2];
#pragma unroll
    for (int i = 0; i < NP / 2; ++i) sp[i] = ld4(srow + 4 * i);    // (mean, M2) x 2 per float4
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < NP / 2; ++i) mean += sp[i].x + sp[i].z;
    mean *= (1.0f / NP);
    float M2 = 0.f;
#pragma unroll
    for (int i = 0; i < NP / 2; ++i) {
      const float d0 = sp[i].x - mean, d1 = sp[i].z - mean;
      M2 += (sp[i].y + sp[i].w) + nl * (d0 * d0 + d1 * d1);
    }
    rstd = 1.0f / sqrtf(M2 * (1.0f / HID) + LN_EPS);
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const float4 gv = ld4(vec + HID + kb + 16 * c), bv = ld4(vec + 2 * HID + kb + 16 * c);
      xh[c] = (av[c] - f4(mean)) * rstd;
      av[c] = relu4(xh[c] * gv + bv);
    }
  } else {
    __syncthreads();
    if (PRO == 2) {
#pragma unroll
      for (int c = 0; c < CW; ++c) { xh[c] = av[c]; av[c] = relu4(av[c]); }
    }
  }
  STAMP(2); BLK_PH_K(1, 2);
  // The normalised rows are kept for the backward pass.  Every column-tile block of a row block holds the same rows, so
  // each stores only ITS share of the 16 column chunks (16 / tiles_n of them) instead of the tn == 0 blocks storing whole
  // rows: 64 KB per row block spread over all of its blocks -- those few blocks were the launch's long pole.
  if (PRO != 0 && m0 + r < p.M) {
    const long ro = ni * p.act_ns + (long)(m0 + r) * HID + kb;
    const int gsz = 16 / tiles_n, g0 = tn * gsz;               // this block's chunks: [g0, g0 + gsz) (tiles_n is 8 or 16)
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int gc = ks * CW + c;                              // (wave-uniform)
      if (gc >= g0 && gc < g0 + gsz) {
        if (G.xh_out) st4(G.xh_out + ro + 16 * c, xh[c]);
        if (G.h_out) st4(G.h_out + ro + 16 * c, av[c]);
      }
    }
    if (G.rstd_out && tn == 0 && ks == 0 && kq == 0) G.rstd_out[(long)ni * p.M + m0 + r] = rstd;
  }
  // ---- 4. second layer: this wave's rows x the block's 16 NT columns over its K range (W tiles from LDS)
  // Even chunks accumulate in a0, odd ones in a1 (per column tile): 2 NT independent MFMA chains, issued alternately, with the W
  // fragments of four chunks requested together -- no instruction waits for the LDS read or the MFMA right in front of it.  (Each
  // accumulator still sees its chunks in the same order: the sums are bit-identical to the chunk-by-chunk loop this replaces.)
  // C4: one (a0, a1) pair per group of four chunks -- the partial sums two waves of a KS = 4 block would hold.
  constexpr int NH = C4 ? 2 : 1;                              // partial sums per wave and column tile
  f32x4 acc[NT], ph[NT][NH];
  {
    constexpr int GC = CW < 4 ? CW : 4;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      f32x4 a0[NT], a1[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { a0[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int c0 = h * (CW / NH); c0 < (h + 1) * (CW / NH); c0 += GC) {
        float4 wv[NT][GC];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < GC; ++j) wv[nt][j] = ld4(W2s + (16 * nt + r) * AS + kb + 16 * (c0 + j));
#pragma unroll
        for (int j = 0; j < GC; j += 2)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              a0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(av[c0 + j], q), f4c(wv[nt][j], q), a0[nt], 0, 0, 0);
              if (j + 1 < GC) a1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(av[c0 + j + 1], q), f4c(wv[nt][j + 1], q), a1[nt], 0, 0, 0);
            }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { ph[nt][h][0] = a0[nt][0] + a1[nt][0]; ph[nt][h][1] = a0[nt][1] + a1[nt][1]; ph[nt][h][2] = a0[nt][2] + a1[nt][2]; ph[nt][h][3] = a0[nt][3] + a1[nt][3]; }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = ph[nt][0];
  }
  STAMP(3); BLK_PH_K(1, 3);
  if (KS > 1) {                               // sum the K-slices of each 16 x 16 tile in slice order; the ks == 0 wave keeps the total
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int h = 0; h < NH; ++h)
        st4(red + (((nt * 4 + wave) * NH + h) * 64 + lane) * 4, make_float4(ph[nt][h][0], ph[nt][h][1], ph[nt][h][2], ph[nt][h][3]));
    __syncthreads();
    if (ks == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            if (j == 0 && h == 0) continue;                   // (the sum starts from this wave's first partial)
            const float4 o = ld4(red + (((nt * 4 + wave + j) * NH + h) * 64 + lane) * 4);
            acc[nt][0] += o.x; acc[nt][1] += o.y; acc[nt][2] += o.z; acc[nt][3] += o.w;
          }
    }
  }
  STAMP(4); BLK_PH_K(1, 4);
  if (ks == 0) {
    float* y = G.Y + ni * p.y_ns;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + 16 * nt + (lane & 15);
      if (col >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + 4 * (lane >> 4) + i;
        if (row < p.M) y[(long)row * p.ldy + col] = acc[nt][i] + bias[nt];
      }
    }
  }
  STAMP(5); BLK_PH_K(1, 5);
  BLK_MARK_K(1, 1);
}

// Generic-K form (an unfused first layer wider than 64 inputs): 16 x 16 tile per block, wave w takes k chunks w, w+4, ...
__global__ __launch_bounds__(256) void k_nt_wide(NtArgs p) {
  __shared__ __attribute__((aligned(16))) float red[4 * 64 * 4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, net = blockIdx.z;
  const int r = lane & 15, kq = lane >> 4;
  if (blockIdx.x == 0 && t == 0 && net == 0) {
    if (p.tick0 || p.tick1) tick_all(p.tick0, p.tick1, p.adam_pw, p.adam_out, p.lr, p.b1, p.b2);
  }
  if (blockIdx.x == 0 && t == 64 && net == 0 && p.tick0b) tick_all(p.tick0b, nullptr, p.adam_pw_b, p.adam_out_b, p.lr_b, p.b1, p.b2);
  const int grp = net / p.npg, ni = net - grp * p.npg;
  const NtGrp G = p.g[grp];
  const float* Pn = G.P + ni * p.p_ns;
  const int tiles_n = (p.N + 15) >> 4;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * 16, n0 = tn * 16;
  const float* Arow = G.in + ni * p.in_ns + (long)min(m0 + r, p.M - 1) * p.ld_in;
  const float* Wrow = Pn + p.oW + (long)min(n0 + r, p.N - 1) * p.ldw;
  const int chunks = (p.K + 15) >> 4, Kr = (p.K + 3) & ~3;
  float bias = p.oBias >= 0 ? Pn[p.oBias + min(n0 + r, p.N - 1)] : 0.f;   // for the epilogue, requested up front
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int cb = 0; cb < chunks; cb += 32) {
    float4 a[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = cb + wave + 4 * u, k = 16 * c + 4 * kq;
      a[u] = ld4_cols(Arow, k, p.K, Kr);
      w[u] = ld4_cols(Wrow, k, p.K, Kr);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) { MFMA4(acc, a[u], w[u]); }
  }
  PIN(bias);
  acc = splitk_reduce(red, acc, wave, lane);
  const int col = n0 + (lane & 15);
  if (wave == 0 && col < p.N) {
    float* y = G.Y + ni * p.y_ns;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 4 * (lane >> 4) + i;
      if (row < p.M) y[(long)row * p.ldy + col] = acc[i] + bias;
    }
  }
}

// ---- large-batch form (M >= 1024): the layers are big enough to be MFMA-bound, so they run as a classic LDS-tiled GEMM
// (64 x 64 outputs per block, K streamed in 64-wide chunks through a double-buffered LDS stage, every operand byte
// fetched once per block with full-line loads) with the LayerNorm+ReLU between them as a row kernel of its own.
// 8 waves per block, two per SIMD (wave = 16 rows x 32 columns): a SIMD issues its waves in order, so with a single
// wave the LDS fragment reads, the parking of the next chunk and the barrier all sit in front of the MFMAs; a second
// wave fills those gaps (measured at B = 1024, K = 256: 4000 -> 3500 cycles per 64-wide chunk; MFMA issue alone is 2048,
// the rest is the 32 KB each CU has to pull per chunk).
#define KC64 64
#define LS64 (KC64 + 4)
// WM x WN waves, each 16 rows x (16 TT) columns: <4, 2, 2> is the 64 x 64 tile of the multi-net launches (512 threads),
// <2, 2, 1> a 32 x 32 tile (256 threads) for a single net's wide first layer, where 64 x 64 tiles would leave 3/4 of the
// CUs without a block.  Each thread stages 2 float4 of A and 2 of W per 64-wide chunk in both shapes.
template <int WM, int WN, int TT>
__global__ __launch_bounds__(64 * WM * WN) void k_nt64(NtArgs p) {       // Y[M,N] = A[M,K] W[N,K]^T + bias
  constexpr int TM = 16 * WM, TN = 16 * TT * WN, NTH = 64 * WM * WN;
  constexpr int RA = 16 * TM / NTH, RW = 16 * TN / NTH, SR = NTH / 16;    // float4 per thread per chunk, rows per staging pass
  static_assert(RA >= 1 && RW >= 1 && 16 * TM % NTH == 0 && 16 * TN % NTH == 0, "staging map");
  __shared__ __attribute__((aligned(16))) float As[2][TM * LS64];
  __shared__ __attribute__((aligned(16))) float Ws[2][TN * LS64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave % WM, wn = wave / WM;
  const int r = lane & 15, kq = lane >> 4;
  // (256-thread instances) the replay gather into the batch slot as extra blocks behind the tile blocks, as in k_nt: the tile
  // blocks read their rows straight from the ring (ring_rows below), so nothing in this launch waits for the batch slot
  const int tile_blocks = (p.gblocks || p.alpha_block) ? p.nt_blocks : (int)gridDim.x;
  if ((int)blockIdx.x >= tile_blocks) {           // (block-uniform) riding blocks: replay gathers (256-thread instances), a pending temperature step
    const int x = (int)blockIdx.x - tile_blocks;
    if (NTH == 256 && x < p.gblocks) riding_gather(p, x);
    else if (x >= p.gblocks && p.alpha_block) alpha_body(p.al);
    return;
  }
  if (blockIdx.x == 0 && t == 0) {
    if (p.tick0 || p.tick1) tick_all(p.tick0, p.tick1, p.adam_pw, p.adam_out, p.lr, p.b1, p.b2);
  }
  if (blockIdx.x == 0 && t == 64 && p.tick0b) tick_all(p.tick0b, nullptr, p.adam_pw_b, p.adam_out_b, p.lr_b, p.b1, p.b2);
  // Workgroups are dealt to the 8 XCDs round-robin; give each XCD a contiguous run of tiles (half a net's 16 x 4 tile
  // grid at B = 1024) so that the A-row and W-column tiles its 32 CUs share are fetched into that XCD's L2 once.
  const int tiles_n = (p.N + TN - 1) / TN, tiles = tiles_n * ((p.M + TM - 1) / TM);
  int L = blockIdx.x;
  { const int per = tile_blocks >> 3; if (L < per * 8) L = (L & 7) * per + (L >> 3); }
  const int net = L / tiles, idx = L - net * tiles;
  const int grp = net / p.npg, ni = net - grp * p.npg;
  const NtGrp G = p.g[grp];
  const float* Pn = G.P + ni * p.p_ns;
  const float* A = G.in + ni * p.in_ns;
  const int bm = idx / tiles_n, bn = idx - bm * tiles_n;
  const int m0 = bm * TM, n0 = bn * TN;
  // staging map: thread -> rows (t >> 4) + SR u, 16-byte column (t & 15): 64 floats of every row per chunk
  const int sr0 = t >> 4, sc = (t & 15) * 4;
  const float* ap[RA]; const float* wp[RW];
#pragma unroll
  for (int u = 0; u < RA; ++u) ap[u] = A + (long)min(m0 + sr0 + SR * u, p.M - 1) * p.ld_in;
  if (G.ring) {                                   // (block-uniform) row m = the sampled record's field: same draw as gather_body's
    const int inject = p.ga[0].ctl->inject_idx, rb_len = p.ga[0].ctl->rb_len, sctr = p.ga[0].ctl->sample_ctr + G.sctr_add;   // one batch of requests
    const unsigned long long seed = p.ga[0].ctl->seed;
    int inj_id[RA];                                 // injected indices: requested with the control words, used only in injected mode
#pragma unroll                                      // (a load behind `if (inject)` would be a second dependent round trip)
    for (int u = 0; u < RA; ++u) inj_id[u] = G.ring_idx[min(m0 + sr0 + SR * u, p.M - 1)];
    __builtin_amdgcn_sched_barrier(0);              // all of the above are in flight before the first of them is waited for
#pragma unroll
    for (int u = 0; u < RA; ++u) {
      const int m = min(m0 + sr0 + SR * u, p.M - 1);
      const int drawn = (int)philox_index(seed, (unsigned)sctr, (unsigned)m, (unsigned)max(rb_len, 1));
      const int id = inject ? min(max(inj_id[u], 0), max(rb_len, 1) - 1) : drawn;
      ap[u] = reinterpret_cast<const float*>(p.ga[0].ring) + (long)id * (4 * p.ga[0].rec4) + G.ring_off;
    }
  }
#pragma unroll
  for (int u = 0; u < RW; ++u) wp[u] = Pn + p.oW + (long)min(n0 + sr0 + SR * u, p.N - 1) * p.ldw;
  const int Kr = (p.K + 3) & ~3;
  f32x4 acc[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nc = (p.K + KC64 - 1) / KC64;
  float bias[TT];                                 // loaded up front: a load issued among the epilogue's stores would make
#pragma unroll                                    // every store wait for the one before it
  for (int tt = 0; tt < TT; ++tt) bias[tt] = p.oBias >= 0 ? Pn[p.oBias + min(n0 + 16 * TT * wn + 16 * tt + r, p.N - 1)] : 0.f;
  float4 ra[RA], rw[RW];
  auto fetch = [&](int c) {                       // raw loads only: the values are first touched by park(), a stage later
    const int k = c * KC64 + sc;
#pragma unroll
    for (int u = 0; u < RA; ++u) ra[u] = ld4_raw(ap[u], k, Kr);
#pragma unroll
    for (int u = 0; u < RW; ++u) rw[u] = ld4_raw(wp[u], k, Kr);
  };
  // (Measured inside the Humanoid iteration, full-iteration A/B on one box: a uniform "full chunk: skip the masks" branch here
  //  costs 11 us per iteration -- the compiler then waits for the loads in front of the branch -- and parking the next chunk
  //  BEFORE the MFMAs, with fragments requested a k step ahead, 4.5 us, although each looks cheaper on paper.)
  auto park = [&](int buf, int c) {
    const int k = c * KC64 + sc;
#pragma unroll
    for (int u = 0; u < RA; ++u) st4(As[buf] + (sr0 + SR * u) * LS64 + sc, mask4_cols(ra[u], k, p.K));
#pragma unroll
    for (int u = 0; u < RW; ++u) st4(Ws[buf] + (sr0 + SR * u) * LS64 + sc, mask4_cols(rw[u], k, p.K));
  };
  STAMP(0);
  fetch(0);
  park(0, 0);
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    const int buf = c & 1;
    if (c + 1 < nc) fetch(c + 1);                 // next chunk's loads fly under this chunk's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    const float* ab = As[buf] + (16 * wm + r) * LS64 + 4 * kq;
    const float* wb = Ws[buf] + (16 * TT * wn + r) * LS64 + 4 * kq;
#pragma unroll
    for (int s2 = 0; s2 < KC64 / 16; ++s2) {
      const float4 a = ld4(ab + 16 * s2);
      float4 b[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) b[tt] = ld4(wb + 16 * tt * LS64 + 16 * s2);
      // the column tiles' accumulators are independent: interleaved, no MFMA waits on its predecessor
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[tt].x, acc[tt], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[tt].y, acc[tt], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[tt].z, acc[tt], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[tt].w, acc[tt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < nc) park(buf ^ 1, c + 1);
    __syncthreads();
  }
  STAMP(1);
  float* y = G.Y + ni * p.y_ns;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int col = n0 + 16 * TT * wn + 16 * tt + r;
    if (col >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 16 * wm + 4 * kq + i;
      if (row < p.M) y[(long)row * p.ldy + col] = acc[tt][i] + bias[tt];
    }
  }
  STAMP(2);
}

// Second hidden layer of the large-batch multi-net trunks WITH the first layer's LayerNorm + ReLU as its prologue (the k_ln_fwd row
// kernel between the two tiled GEMMs, one graph node and 10 MB of traffic, folded in).  K = 256 = one row per 16 threads in exactly
// k_nt64's staging map (thread (row = t >> 4 + SR u, sub = t & 15) owns columns 4 sub + 64 q -- the Row16 layout), so a block loads
// its TM rows ONCE, up front, normalises them in registers with the same two DPP row reductions as k_ln_fwd (bit-identical
// statistics), and parks chunk c of the activated rows from registers while the loop streams only W.  The column-tile-0 block of
// every row block stores h / xhat / rstd where the backward pass wants them.
template <int WM, int WN, int TT>
__global__ __launch_bounds__(64 * WM * WN) void k_nt64_ln(NtArgs p) {       // Y[M,N] = relu(LN(Z))[M,256] W[N,256]^T + bias
  constexpr int TM = 16 * WM, TN = 16 * TT * WN, NTH = 64 * WM * WN, NC = HID / KC64;
  constexpr int RA = 16 * TM / NTH, RW = 16 * TN / NTH, SR = NTH / 16;
  static_assert(RA >= 1 && RW >= 1 && 16 * TM % NTH == 0 && 16 * TN % NTH == 0 && NC == 4, "staging map");
  __shared__ __attribute__((aligned(16))) float As[2][TM * LS64];
  __shared__ __attribute__((aligned(16))) float Ws[2][TN * LS64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave % WM, wn = wave / WM;
  const int r = lane & 15, kq = lane >> 4;
  // (256-thread instance) riding blocks behind the tile blocks, as in k_nt: a pending temperature step, the next tail's N(0,1) draws
  const int tile_blocks = (p.alpha_block || p.nz_n) ? p.nt_blocks : (int)gridDim.x;
  if (NTH == 256 && (int)blockIdx.x >= tile_blocks) {
    riding_body(p, (int)blockIdx.x - tile_blocks);
    return;
  }
  const int tiles_n = (p.N + TN - 1) / TN, tiles = tiles_n * ((p.M + TM - 1) / TM);
  int L = blockIdx.x;
  { const int per = tile_blocks >> 3; if (L < per * 8) L = (L & 7) * per + (L >> 3); }      // XCD-contiguous tile runs (see k_nt64)
  const int net = L / tiles, idx = L - net * tiles;
  const int grp = net / p.npg, ni = net - grp * p.npg;
  const NtGrp G = p.g[grp];
  const float* Pn = G.P + ni * p.p_ns;
  const float* Z = G.in + ni * p.in_ns;
  const int bm = idx / tiles_n, bn = idx - bm * tiles_n;
  const int m0 = bm * TM, n0 = bn * TN;
  const int sr0 = t >> 4, sub = t & 15, sc = sub * 4;
  // ---- every load of the prologue + the first W chunk
  Row16 z[RA];
#pragma unroll
  for (int u = 0; u < RA; ++u) z[u] = row_ld(Z + (long)min(m0 + sr0 + SR * u, p.M - 1) * p.ld_in, sub);
  const Row16 g = row_ld(Pn + (p.ln_pro ? p.oG : 0), sub), be = row_ld(Pn + (p.ln_pro ? p.oBe : 0), sub);   // unconditional: one batch
  const float* wp[RW];
#pragma unroll
  for (int u = 0; u < RW; ++u) wp[u] = Pn + p.oW + (long)min(n0 + sr0 + SR * u, p.N - 1) * p.ldw;
  float bias[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) bias[tt] = p.oBias >= 0 ? Pn[p.oBias + min(n0 + 16 * TT * wn + 16 * tt + r, p.N - 1)] : 0.f;
  float4 rw[RW];
  auto fetch = [&](int c) {
#pragma unroll
    for (int u = 0; u < RW; ++u) rw[u] = ld4(wp[u] + c * KC64 + sc);
  };
  fetch(0);
  // ---- LayerNorm + ReLU of this thread's rows (k_ln_fwd's arithmetic), stores by the column-tile-0 block
  Row16 h[RA];
#pragma unroll
  for (int u = 0; u < RA; ++u) {
    Row16 xh, y; float rstd;
    ln_fwd(z[u], g, be, p.ln_pro, xh, y, rstd);
#pragma unroll
    for (int q = 0; q < 4; ++q) h[u].v[q] = relu4(y.v[q]);
    const int row = m0 + sr0 + SR * u;
    if (bn == 0 && row < p.M) {                    // (block-uniform first half)
      const long ro = ni * p.act_ns + (long)row * HID;
      if (G.h_out) row_st(G.h_out + ro, sub, h[u]);
      if (G.xh_out) row_st(G.xh_out + ro, sub, xh);
      if (G.rstd_out && sub == 0) G.rstd_out[(long)ni * p.M + row] = rstd;
    }
  }
  auto park = [&](int buf, auto ctag) {
    constexpr int c = decltype(ctag)::value;
#pragma unroll
    for (int u = 0; u < RA; ++u) st4(As[buf] + (sr0 + SR * u) * LS64 + sc, h[u].v[c]);
#pragma unroll
    for (int u = 0; u < RW; ++u) st4(Ws[buf] + (sr0 + SR * u) * LS64 + sc, rw[u]);
  };
  f32x4 acc[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](int buf) {
    const float* ab = As[buf] + (16 * wm + r) * LS64 + 4 * kq;
    const float* wb = Ws[buf] + (16 * TT * wn + r) * LS64 + 4 * kq;
#pragma unroll
    for (int s2 = 0; s2 < KC64 / 16; ++s2) {
      const float4 a = ld4(ab + 16 * s2);
      float4 b[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) b[tt] = ld4(wb + 16 * tt * LS64 + 16 * s2);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[tt].x, acc[tt], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[tt].y, acc[tt], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[tt].z, acc[tt], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[tt].w, acc[tt], 0, 0, 0);
    }
  };
  // 4 chunks, unrolled by hand (the parked register set is selected at compile time: a run-time index would be scratch memory)
  park(0, std::integral_constant<int, 0>{});
  __syncthreads();
  fetch(1); __builtin_amdgcn_sched_barrier(0); mma(0); __builtin_amdgcn_sched_barrier(0); park(1, std::integral_constant<int, 1>{}); __syncthreads();
  fetch(2); __builtin_amdgcn_sched_barrier(0); mma(1); __builtin_amdgcn_sched_barrier(0); park(0, std::integral_constant<int, 2>{}); __syncthreads();
  fetch(3); __builtin_amdgcn_sched_barrier(0); mma(0); __builtin_amdgcn_sched_barrier(0); park(1, std::integral_constant<int, 3>{}); __syncthreads();
  mma(1);
  float* y = G.Y + ni * p.y_ns;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int col = n0 + 16 * TT * wn + 16 * tt + r;
    if (col >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 16 * wm + 4 * kq + i;
      if (row < p.M) y[(long)row * p.ldy + col] = acc[tt][i] + bias[tt];
    }
  }
}

struct LnFwd {               // h = relu(LN(z) * gamma + beta) per row; stores h, xhat, rstd   (nets on blockIdx.y)
  const float* P[2]; int npg; int oG, oBe; long p_ns;
  float* h[2]; float* xh[2]; float* rstd[2];     // per group; xh / rstd may be null
  const float* zin[2];                           // per group input rows [npg][B][HID]
  int B, ln;
};
__global__ __launch_bounds__(256) void k_ln_fwd(LnFwd p) {
  const int t = threadIdx.x, row = t >> 4, sub = t & 15, net = blockIdx.y;
  const int grp = net / p.npg, ni = net - grp * p.npg;
  const int b = blockIdx.x * 16 + row;
  if (b >= p.B) return;
  const long ro = ((long)ni * p.B + b) * HID;
  const float* Pn = p.P[grp] + ni * p.p_ns;
  const Row16 z = row_ld(p.zin[grp] + ro, sub);
  Row16 g = row_ld(Pn + (p.ln ? p.oG : 0), sub), be = row_ld(Pn + (p.ln ? p.oBe : 0), sub);   // unconditional: one batch of requests
  row_pin(g); row_pin(be);
  Row16 xh, y, hh;
  float rstd;
  ln_fwd(z, g, be, p.ln, xh, y, rstd);
#pragma unroll
  for (int q = 0; q < 4; ++q) hh.v[q] = relu4(y.v[q]);
  row_st(p.h[grp] + ro, sub, hh);
  if (p.xh[grp]) row_st(p.xh[grp] + ro, sub, xh);
  if (p.rstd[grp] && sub == 0) p.rstd[grp][(long)ni * p.B + b] = rstd;
}

struct NnArgs {              // dX[M,Kout] = dY[M,256] * W[256, k_off : k_off+Kout] ; block = one 16 x 16 tile, n split over waves
  const float* dY; long dy_ns;              // row stride HID
  const float* Wt; int ldw; long p_ns; int k_off;
  float* dX; int ldx; long dx_ns;
  int M, Kout;
  int xr;                                   // k_nn: XCD row groups of the tile placement (xcd_tile), 0 = row-major numbering
};

__global__ __launch_bounds__(256) void k_nn(NnArgs p) {
  __shared__ __attribute__((aligned(16))) float red[4 * 64 * 4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, net = blockIdx.z;
  const int r = lane & 15, kq = lane >> 4;
  const int tiles_k = (p.Kout + 15) >> 4;
  int tm, tk;
  xcd_tile(blockIdx.x, (p.M + 15) >> 4, tiles_k, p.xr, tm, tk);
  const int m0 = tm * 16, k0 = tk * 16;
  const int mrow = min(m0 + r, p.M - 1);
  const int nb = 64 * wave + 4 * kq;
  const float* Drow = p.dY + net * p.dy_ns + (long)mrow * HID + nb;
  const float* Wc = p.Wt + net * p.p_ns + (long)nb * p.ldw + p.k_off + min(k0 + r, p.Kout - 1);
  STAMP(0);
  float4 av[4], bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    av[c] = ld4(Drow + 16 * c);
    const float* w = Wc + (long)(16 * c) * p.ldw;
    bv[c] = make_float4(w[0], w[p.ldw], w[2 * (long)p.ldw], w[3 * (long)p.ldw]);
  }
  __builtin_amdgcn_sched_barrier(0);
  STAMP(1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) { MFMA4(acc, av[c], bv[c]); }
  STAMP(2);
  acc = splitk_reduce(red, acc, wave, lane);
  STAMP(3);
  const int col = k0 + (lane & 15);
  if (wave == 0 && col < p.Kout) {
    float* x = p.dX + net * p.dx_ns;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 4 * (lane >> 4) + i;
      if (row < p.M) x[(long)row * p.ldx + col] = acc[i];
    }
  }
  STAMP(4);
}

// Large-batch form of k_nn (M >= 1024, Kout = 256: the dh1 = dz2 W2 products): LDS-tiled like k_nt64 -- (16 WM) x (16 TT WN)
// outputs per block, the 256-long reduction streamed in 64-wide double-buffered chunks, every operand byte fetched once per
// block with full-line loads.  W is used as stored ([n][c], rows = the reduction index): its chunk is parked in that layout and
// the B fragments are read as columns (4 ds_read_b32, conflict-free: row stride = 4 mod 8), so no transposed copy of the
// weights exists anywhere.  k_nn's 16 x 16 tiles fetch 18 MB per launch for 4 MB of operands at B = 1024 (rocprofv3 FETCH_SIZE).
template <int WM, int WN, int TT>
__global__ __launch_bounds__(64 * WM * WN) void k_nn64(NnArgs p) {        // dX[M,256] = dY[M,256] W[256,256]
  constexpr int TM = 16 * WM, TN = 16 * TT * WN, NTH = 64 * WM * WN, LSB = TN + 4;
  constexpr int RA = 16 * TM / NTH, RB = (KC64 * TN / 4) / NTH;          // float4 per thread per chunk
  static_assert(RA >= 1 && RB >= 1 && 16 * TM % NTH == 0 && (KC64 * TN / 4) % NTH == 0, "staging map");
  __shared__ __attribute__((aligned(16))) float As[2][TM * LS64];
  __shared__ __attribute__((aligned(16))) float Bs[2][KC64 * LSB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave % WM, wn = wave / WM;
  const int r = lane & 15, kq = lane >> 4;
  const int tiles_n = HID / TN, tiles = tiles_n * ((p.M + TM - 1) / TM);
  int L = blockIdx.x;
  { const int per = (int)gridDim.x >> 3; if (L < per * 8) L = (L & 7) * per + (L >> 3); }     // XCD-contiguous tile runs (see k_nt64)
  const int net = L / tiles, idx = L - net * tiles;
  const int bm = idx / tiles_n, bn = idx - bm * tiles_n;
  const int m0 = bm * TM, n0 = bn * TN;
  const float* A = p.dY + net * p.dy_ns;
  const float* W = p.Wt + net * p.p_ns + p.k_off + n0;
  // staging maps: A chunk = TM rows x 16 float4 -> (row = t >> 4 [+ NTH/16 u], c4 = t & 15); B chunk = 64 rows x TN/4 float4
  const int sr0 = t >> 4, sc = (t & 15) * 4;
  constexpr int BC4 = TN / 4;
  const float* ap[RA];
#pragma unroll
  for (int u = 0; u < RA; ++u) ap[u] = A + (long)min(m0 + sr0 + (NTH / 16) * u, p.M - 1) * HID + sc;
  float4 ra[RA], rb[RB];
  auto fetch = [&](int c) {
#pragma unroll
    for (int u = 0; u < RA; ++u) ra[u] = ld4(ap[u] + c * KC64);
#pragma unroll
    for (int u = 0; u < RB; ++u) { const int i = t + NTH * u, row = i / BC4, c4 = i % BC4; rb[u] = ld4(W + (long)(c * KC64 + row) * p.ldw + 4 * c4); }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int u = 0; u < RA; ++u) st4(As[buf] + (sr0 + (NTH / 16) * u) * LS64 + sc, ra[u]);
#pragma unroll
    for (int u = 0; u < RB; ++u) { const int i = t + NTH * u, row = i / BC4, c4 = i % BC4; st4(Bs[buf] + row * LSB + 4 * c4, rb[u]); }
  };
  f32x4 acc[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch(0);
  park(0);
  __syncthreads();
  constexpr int nc = HID / KC64;
  for (int c = 0; c < nc; ++c) {
    const int buf = c & 1;
    if (c + 1 < nc) fetch(c + 1);                 // next chunk's loads fly under this chunk's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    const float* ab = As[buf] + (16 * wm + r) * LS64 + 4 * kq;
    const float* bb = Bs[buf] + (4 * kq) * LSB + 16 * TT * wn + r;
#pragma unroll
    for (int s2 = 0; s2 < KC64 / 16; ++s2) {
      const float4 a = ld4(ab + 16 * s2);
      float4 b[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float* b0 = bb + 16 * s2 * LSB + 16 * tt;
        b[tt] = make_float4(b0[0], b0[LSB], b0[2 * LSB], b0[3 * LSB]);
      }
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[tt].x, acc[tt], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[tt].y, acc[tt], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[tt].z, acc[tt], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[tt].w, acc[tt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < nc) park(buf ^ 1);
    __syncthreads();
  }
  float* x = p.dX + net * p.dx_ns;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int col = n0 + 16 * TT * wn + 16 * tt + r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 16 * wm + 4 * kq + i;
      if (row < p.M) x[(long)row * p.ldx + col] = acc[tt][i];
    }
  }
}

// Sum of the split-M partial slabs + the vector gradients the row kernels left as per-row-block partials, then the
// optimiser step of torch.optim.Adam (agents/agent.py:236,286) and the Polyak update of the same element (agents/agent.py:328).
// Sources by offset range: `vec` ranges come from part[net][blk][slot][256] summed over the row blocks (16 threads per
// float4, DPP-reduced), one scalar from part_s[net][blk][0], everything else from the S <= 8 slabs.  Sums run in a fixed
// order, so replays are bit-reproducible.
struct AdamRedVec { int off; int slot; int nblk; };
struct AdamRedArgs {
  const float* Gp; int S; int nets; long g_ns;           // slabs [S][nets][g_ns]
  float* G;                                              // gradient arena (always written)
  int apply; float* P; float* Mo; float* Vo; float* T; float tau;
  float* T2; float* T3;                                  // optional: the target one / two further Polyak steps ahead (see TnArgs)
  const float* adam; float b1, b2, eps;
  AdamRedVec vec[5]; int nvec;                           // 256-wide vectors finalised from row-block partials (nblk blocks each)
  const float* part; int pstride;
  int s_off; int s_nblk; const float* part_s;            // one scalar element (critic head bias), s_off < 0: none
  const float* loss_part; int loss_n, loss_stride, loss_off; float loss_scale; float* loss_dst; int* tick;
  int tick_extra;                                        // the counter advances by 1 + tick_extra (a deferred temperature step's tick)
};
__device__ __forceinline__ void adam_red_commit(const AdamRedArgs& a, long off, float4 g, float4 w, float4 m, float4 v, float4 tt, float step, float sq2) {
  st4(a.G + off, g);
  if (a.apply) {
    const float omb1 = 1.0f - a.b1, omb2 = 1.0f - a.b2;
    m = m + (g - m) * omb1;
    v = v * a.b2 + g * g * omb2;
    w.x -= step * (m.x / (sqrtf(v.x) / sq2 + a.eps)); w.y -= step * (m.y / (sqrtf(v.y) / sq2 + a.eps));
    w.z -= step * (m.z / (sqrtf(v.z) / sq2 + a.eps)); w.w -= step * (m.w / (sqrtf(v.w) / sq2 + a.eps));
    st4(a.Mo + off, m); st4(a.Vo + off, v); st4(a.P + off, w);
    if (a.T) {
      const float4 t1 = tt + (w - tt) * a.tau;
      st4(a.T + off, t1);
      if (a.T2) {
        const float4 t2 = t1 + (w - t1) * a.tau;
        st4(a.T2 + off, t2);
        if (a.T3) st4(a.T3 + off, t2 + (w - t2) * a.tau);
      }
    }
  }
}
// The "tail" blocks of a gradient finalisation: block `rel` < 4 nvec = one quarter of a 256-wide vector gradient (float4 column
// (t >> 4) + 16 quarter; 16 threads share its row-block partials and DPP-reduce them); block 4 nvec = the scalar element
// (critic head bias) + the loss finalisation + the counter tick.  Used by k_adam_red and, as riding blocks, by k_tn -- whose
// tile blocks then do nothing but their GEMM tile and its optimiser step.
__device__ __forceinline__ void adam_red_tail_body(const AdamRedArgs& a, int rel, int net) {
  const int t = threadIdx.x;
  const float step = a.apply ? a.adam[0] : 0.f, sq2 = a.apply ? a.adam[1] : 1.f;
  if (rel < 4 * a.nvec) {
    const int e = rel >> 2, quarter = rel & 3;
    int voff = a.vec[0].off, vslot = a.vec[0].slot, nb = a.vec[0].nblk;
    if (e == 1) { voff = a.vec[1].off; vslot = a.vec[1].slot; nb = a.vec[1].nblk; }
    if (e == 2) { voff = a.vec[2].off; vslot = a.vec[2].slot; nb = a.vec[2].nblk; }
    if (e == 3) { voff = a.vec[3].off; vslot = a.vec[3].slot; nb = a.vec[3].nblk; }
    if (e == 4) { voff = a.vec[4].off; vslot = a.vec[4].slot; nb = a.vec[4].nblk; }
    const int c4 = (t >> 4) + 16 * quarter, sub = t & 15;
    const long off = net * a.g_ns + voff + 4 * c4;
    float4 w = f4(0.f), m = f4(0.f), v = f4(0.f), tt = f4(0.f);
    if (a.apply && sub == 0) { w = ld4(a.P + off); m = ld4(a.Mo + off); v = ld4(a.Vo + off); if (a.T) tt = ld4(a.T + off); }
    const float* pp = a.part + ((long)net * a.pstride * NSLOT + vslot) * HID + 4 * c4;
    float4 g = f4(0.f);
    for (int blk = sub; blk < nb; blk += 64) {             // 4 independent requests per trip
      const float4 v0 = ld4(pp + (long)blk * NSLOT * HID), v1 = ld4(pp + (long)min(blk + 16, nb - 1) * NSLOT * HID);
      const float4 v2 = ld4(pp + (long)min(blk + 32, nb - 1) * NSLOT * HID), v3 = ld4(pp + (long)min(blk + 48, nb - 1) * NSLOT * HID);
      g = g + v0;
      if (blk + 16 < nb) g = g + v1;
      if (blk + 32 < nb) g = g + v2;
      if (blk + 48 < nb) g = g + v3;
    }
    g.x = row16_sum(g.x); g.y = row16_sum(g.y); g.z = row16_sum(g.z); g.w = row16_sum(g.w);
    if (sub == 0) adam_red_commit(a, off, g, w, m, v, tt, step, sq2);
    return;
  }
  if (t >= 64) return;
  // scalar element (+ the 3 padding floats behind it), then block-level extras by net 0
  const bool extras = net == 0;
  float loss_acc = 0.f; int tick_v = 0;
  if (extras) {
    if (a.loss_dst) for (int k = t; k < a.loss_n; k += 64) loss_acc += a.loss_part[(long)k * a.loss_stride + a.loss_off];
    if (t == 0 && a.tick) tick_v = *a.tick;
  }
  if (a.s_off >= 0) {
    const long off = net * a.g_ns + a.s_off;
    float4 w = f4(0.f), m = f4(0.f), v = f4(0.f), tt = f4(0.f);
    if (a.apply && t == 0) { w = ld4(a.P + off); m = ld4(a.Mo + off); v = ld4(a.Vo + off); if (a.T) tt = ld4(a.T + off); }
    float sg = 0.f;
    for (int blk = t; blk < a.s_nblk; blk += 64) sg += a.part_s[((long)net * a.pstride + blk) * 2];
    sg = wave_sum(sg);
    if (t == 0) adam_red_commit(a, off, make_float4(sg, 0.f, 0.f, 0.f), w, m, v, tt, step, sq2);
  }
  if (extras) {
    if (a.loss_dst) {
      const float sl = wave_sum(loss_acc);
      if (t == 0) *a.loss_dst = sl * a.loss_scale;
    }
    if (t == 0 && a.tick) *a.tick = tick_v + 1 + a.tick_extra;
  }
}

// agents/agent.py:328-331: t <- t + tau (p - t) over up to two arenas; `block` of `nblocks` blocks of 256 threads
struct PolyakArgs { float* t0; const float* p0; long n0; float* t1; const float* p1; long n1; float tau; };
__device__ __forceinline__ void polyak_body(const PolyakArgs& a, unsigned block, unsigned nblocks) {
  const long n = a.n0 + a.n1;
  for (long i = ((long)block * 256 + threadIdx.x) * 4; i < n; i += (long)nblocks * 1024) {
    float* t = i < a.n0 ? a.t0 + i : a.t1 + (i - a.n0);
    const float* p = i < a.n0 ? a.p0 + i : a.p1 + (i - a.n0);
    const float4 tt = ld4(t);
    st4(t, tt + (ld4(p) - tt) * a.tau);
  }
}

struct TnProb {              // one weight-gradient GEMM: dW[N, ldw] = dY[M,N]^T * X[M,K] (columns K..ldw-1 are 0)
  const float* dY; int ldy; long dy_ns; int N;
  const float* X; int ldx; long x_ns; int K;
  int w_off, ldw;                        // where the weight block sits inside a net's parameter / gradient block
  int b_off;                             // its bias (gradient = column sums of dY), -1: none
  int nfin; int fin_slot[3]; int fin_off[3]; int fin_nblk[3];   // vectors from row-kernel partials: part[net][blk][slot][n] summed over blk < fin_nblk
  int fin_s_off; int fin_s_nblk;         // scalar from part_s[net][blk][0] summed over blk (critic head bias), -1: none
  int tile0;                             // first blockIdx.x of this problem
  int xr;                                // XCD row (n-tile) groups of the tile placement (xcd_tile), 0 = row-major numbering
  // fold != 0: dY holds dy = relu'(.) dh1 of hidden layer 1 (written by k_ctail_nn / k_headbwd_nn, N = HID) and this problem's
  // blocks apply the LayerNorm backward to their [M][16] slice on the way into LDS -- the row launch (k_ln_bwd) between the
  // dh1 GEMM and this one is gone.  What a row needs from its other 240 columns are the two sums over the row; the producer
  // left them as one partial per 16-column tile (f_ps[net][row][0:16] = sum dy g, [16:32] = sum dy g xhat), summed here in
  // a fixed order.  The k-tile-0 blocks also own dgamma1 / dbeta1 of their 16 columns (column sums of dy xhat and dy over the
  // batch) and keep dz1 readable (f_dz).  f_ln == 0: no LayerNorm, dz1 = dy.
  int x_dup;                             // host-side bookkeeping: a later piece of a cut problem: an operand that is already counted (1: X, 2: dY)
  int xr_force;                          // host-side: placement chosen by the caller instead of pick_xr
  int kw;                                // > 0: the problem is a COLUMN piece of a wider weight block (tn_cols): its width (ldw stays the row stride)
  int fold, f_ln, f_g_off, f_be_off;
  const float* f_xh; const float* f_rstd; const float* f_ps; float* f_dz;    // [nets][M][HID], [nets][M], [nets][M][32], [nets][M][HID]
  const float* f_g;                      // [nets][HID]: gamma1 as the producer saw it (the k-tile-0 blocks of THIS launch step the live one)
};
constexpr int PS_W = 32;                 // floats per row of the row-sum partials: 16 tiles x 2 sums
struct TnArgs {              // up to 4 problems per launch; block = one 16 x 16 tile of one problem, M split over the 4 waves
  TnProb pr[4]; int nprob; int M;       // (a GEMM may be cut in two problems to steer which blocks share a CU: engine.hip, tn_split)
  float* G; long g_ns;                   // gradient arena (always written; net stride g_ns)
  // optimiser step fused into the epilogue (torch.optim.Adam, agents/agent.py:236,286), optionally with the Polyak
  // update of the same element (agents/agent.py:328).  apply == 0: gradients only (clip_grad_norm_ path).
  // T2 / T3 (TD3 period graphs): the parameters do not change again before the next two Polyak updates of this target, so what
  // those will produce -- t2 = t1 + (w - t1) tau, t3 = t2 + (w - t2) tau, the very expressions they evaluate -- is written to two
  // more arenas now: the next-action passes of the following iterations run ahead through them (engine.hip: enqueue_update_actor)
  int apply; float* P; float* Mo; float* Vo; float* T; float tau; float* T2; float* T3;
  const float* adam;                     // [0] lr / (1 - b1^t), [1] sqrt(1 - b2^t), published by the first kernel of the update
  float b1, b2, eps;
  const float* part; int pstride; const float* part_s;   // pstride = partial blocks allocated per net
  // extras done once by block 0 / net 0: loss finalisation and a counter tick
  const float* loss_part; int loss_n, loss_stride, loss_off; float loss_scale; float* loss_dst; int* tick;
  // `pk_blocks` more blocks (net 0, after the `tiles` GEMM blocks) run a Polyak update of an arena this launch does not
  // otherwise touch (TD3: the actor target in the iterations that have no actor update, agents/agent.py:329-331)
  int tiles; int pk_blocks; PolyakArgs pk;
  // ... and `fin_blocks` more (after those, every net) finalise what is not a GEMM tile: the vector gradients from the row
  // kernels' partials, the scalar head-bias gradient, the loss and the counter tick (adam_red_tail_body).  Tile blocks that also
  // did this (as the k-tile-0 blocks first did) were the launch's long pole: 2.5 memory round trips before their first MFMA.
  int fin_blocks; AdamRedArgs fin;
};

struct AdamState { float w, m, v, t; };
__device__ __forceinline__ AdamState adam_fetch(const TnArgs& p, long off) {
  AdamState s{0.f, 0.f, 0.f, 0.f};
  if (p.apply) { s.w = p.P[off]; s.m = p.Mo[off]; s.v = p.Vo[off]; if (p.T) s.t = p.T[off]; }
  return s;
}
__device__ __forceinline__ void adam_commit(const TnArgs& p, long off, float g, AdamState s, float step, float sq2) {
  p.G[off] = g;
  if (p.apply) {
    const float m = s.m + (g - s.m) * (1.0f - p.b1);
    const float v = s.v * p.b2 + g * g * (1.0f - p.b2);
    const float w = s.w - step * (m / (sqrtf(v) / sq2 + p.eps));
    p.Mo[off] = m; p.Vo[off] = v; p.P[off] = w;
    if (p.T) {
      const float t1 = s.t + (w - s.t) * p.tau;
      p.T[off] = t1;
      if (p.T2) {
        const float t2 = t1 + (w - t1) * p.tau;
        p.T2[off] = t2;
        if (p.T3) p.T3[off] = t2 + (w - t2) * p.tau;
      }
    }
  }
}

// KT = 16-column k tiles per block (1 or 2): with 2 the dY slice is fetched and transposed once for two weight tiles, wave 0
// and wave 1 commit one each -- used when a launch would otherwise put more than two blocks on every CU.
// FOLD: the instance whose layer-1 problem applies the LayerNorm backward itself (TnProb::fold); launches without such a problem
// take the plain instance (the fold's operands cost registers: the actor's B = 1024 launch was 5 us slower through one kernel).
template <int KT, bool FOLD = false>
__global__ __launch_bounds__(256) void k_tn(TnArgs p) {
  __shared__ __attribute__((aligned(16))) float red[KT * 4 * 64 * 4];
  __shared__ __attribute__((aligned(16))) float Ys[256 * YS];
  __shared__ __attribute__((aligned(16))) float Xs[KT * 256 * YS];
  __shared__ float cred[16 * 17];                        // bias gradient: [16 partial groups][16 columns]
  __shared__ __attribute__((aligned(16))) float fsum[FOLD ? 16 * 4 * 8 : 4];   // folded LayerNorm backward: [wave x DPP row][column quad][dgamma 4 | dbeta 4]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, net = blockIdx.z;
  const int r = lane & 15, kq = lane >> 4;
  BLK_MARK(0);
  if ((int)blockIdx.x >= p.tiles) {                         // (block-uniform) riding blocks
    const int x = (int)blockIdx.x - p.tiles;
    if (x < p.pk_blocks) { if (net == 0) polyak_body(p.pk, x, p.pk_blocks); }
    else if (x - p.pk_blocks < p.fin_blocks) adam_red_tail_body(p.fin, x - p.pk_blocks, net);      // (behind them: padding up to a multiple of 8 blocks
    BLK_MARK(1);                                                                                  //  per net, so that every net's tile ids keep their XCDs)
    return;
  }
  int pi = 0;
  if (p.nprob > 1 && (int)blockIdx.x >= p.pr[1].tile0) pi = 1;
  if (p.nprob > 2 && (int)blockIdx.x >= p.pr[2].tile0) pi = 2;
  if (p.nprob > 3 && (int)blockIdx.x >= p.pr[3].tile0) pi = 3;
  const TnProb q = p.pr[pi];     // ONE batch of scalar loads for the whole problem (field-by-field they came in 3-4 dependent rounds)
  const int local = blockIdx.x - q.tile0;
  const int kw = q.kw > 0 ? q.kw : q.ldw;                 // columns of this problem's piece of dW
  const int tiles_k = (((kw + 15) >> 4) + KT - 1) / KT;
  int tn, tk;
  xcd_tile(local, (q.N + 15) >> 4, tiles_k, q.xr, tn, tk);
  const int n0 = tn * 16, k0 = tk * 16 * KT;
  const long nbase = net * p.g_ns;
  // the epilogue's elements: wave kt < KT, lane (j = lane & 15, rq = lane >> 4) owns rows n0 + 4 rq + i, column k0 + 16 kt + j
  const int ecol = k0 + 16 * min(wave, KT - 1) + (lane & 15);
  AdamState st[4], sv = {0.f, 0.f, 0.f, 0.f};
  STAMP(0); BLK_PH(0);
  // operand tiles are column slices ([M rows][16 floats]): fetched as float4 (64-byte pieces), transposed through LDS
  const float* dYn = q.dY + net * q.dy_ns;
  const float* Xn = q.X + net * q.x_ns;
  const int Nr = (q.N + 3) & ~3, Kr = (q.K + 3) & ~3;     // rows hold at least round4(.) floats
  f32x4 acc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;                                       // thread (col = t & 15, part = t >> 4): partial column sums of dY
  float4 vy[4], vx[KT][4];
  const bool fold = FOLD && q.fold != 0, fold_ln = fold && q.f_ln;   // (block-uniform)
  float4 vxh[4], vp1[4], vp2[4], gq = f4(1.f), cg = f4(0.f), cb = f4(0.f);
  float vrs[4];
  auto fetch = [&](int mb) {                      // raw loads; masked when the slab is parked in LDS, a stage later (see ld4_raw)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = t + 256 * u, row = i >> 2, c4 = i & 3, m = mb + row, n = n0 + 4 * c4, k = k0 + 4 * c4;
      const long mc = min(m, p.M - 1);
      vy[u] = ld4_raw(dYn + mc * q.ldy, n, Nr);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) vx[kt][u] = ld4_raw(Xn + mc * q.ldx, k + 16 * kt, Kr);
      if (fold_ln) {
        const long rn = (long)net * p.M + mc;
        vxh[u] = ld4(q.f_xh + rn * HID + n);
        vp1[u] = ld4(q.f_ps + rn * PS_W + 4 * c4);
        vp2[u] = ld4(q.f_ps + rn * PS_W + 16 + 4 * c4);
        vrs[u] = q.f_rstd[rn];
      }
    }
  };
  fetch(0);
  if (fold_ln) gq = ld4(q.f_g + net * HID + n0 + 4 * (t & 3));
  const float step = p.apply ? p.adam[0] : 0.f, sq2 = p.apply ? p.adam[1] : 1.f;
  if (wave < KT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // clamped, not predicated (the commit is predicated)
      const int row = min(n0 + 4 * (lane >> 4) + i, q.N - 1);
      st[i] = adam_fetch(p, nbase + q.w_off + (long)row * q.ldw + min(ecol, kw - 1));
    }
  }
  // k-tile-0 blocks also produce the bias gradient of their 16 columns (column sums of dY, collected from the LDS tile in the
  // main loop): wave 3, lanes 0 .. 15 commit it; its optimiser state is requested now
  const int fcol = t & 15, fpart = t >> 4, fn = n0 + fcol;         // (column, partial-group) of this thread
  const bool want_bias = tk == 0 && q.b_off >= 0;
  AdamState fstate = sv;
  long foff = -1;
  if (want_bias && wave == 3 && lane < 16 && fn < q.N) { foff = nbase + q.b_off + fn; fstate = adam_fetch(p, foff); }
  const bool fold_vec = fold_ln && tk == 0;                // dgamma1 / dbeta1 of the 16 columns: wave 3, lanes 16 .. 31 / 32 .. 47
  if (fold_vec && wave == 3 && lane >= 16 && lane < 48) { foff = nbase + (lane < 32 ? q.f_g_off : q.f_be_off) + fn; fstate = adam_fetch(p, foff); }
  for (int mb = 0; mb < p.M; mb += 256) {
    if (mb) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1); BLK_PH(1);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = t + 256 * u, c4 = i & 3;
      const bool row_ok = mb + (i >> 2) < p.M;
      if (fold) {                                          // N = HID: no column mask
        float4 dz = row_ok ? vy[u] : f4(0.f);
        if (fold_ln) {
          const float a1 = sum4(vp1[u]), a2 = sum4(vp2[u]);                       // 4 of the 16 tile partials each; the quad holds the row
          const float s1 = (dpp_mov<0x00>(a1) + dpp_mov<0x55>(a1)) + (dpp_mov<0xAA>(a1) + dpp_mov<0xFF>(a1));   // quad_perm broadcasts
          const float s2 = (dpp_mov<0x00>(a2) + dpp_mov<0x55>(a2)) + (dpp_mov<0xAA>(a2) + dpp_mov<0xFF>(a2));
          const float m1 = s1 * (1.0f / HID), m2 = s2 * (1.0f / HID);
          const float4 dy = dz;
          cg = cg + dy * vxh[u]; cb = cb + dy;
          dz = (dy * gq - f4(m1) - vxh[u] * m2) * vrs[u];                         // ln_bwd's expression
          if (!row_ok) dz = f4(0.f);
        }
        st4(Ys + (i >> 2) * YS + 4 * c4, dz);
        if (tk == 0 && row_ok && q.f_dz) st4(q.f_dz + ((long)net * p.M + mb + (i >> 2)) * HID + n0 + 4 * c4, dz);
      } else
      st4(Ys + (i >> 2) * YS + 4 * c4, mask4_cols(vy[u], n0 + 4 * c4, q.N, row_ok));
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) st4(Xs + kt * 256 * YS + (i >> 2) * YS + 4 * c4, mask4_cols(vx[kt][u], k0 + 4 * c4 + 16 * kt, q.K, row_ok));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (mb + 256 < p.M) fetch(mb + 256);                   // the next slab's rows fly under this slab's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {                          // wave w: 16-row chunks w, w+4, w+8, w+12 of this slab
      const float* y0 = Ys + (16 * (wave + 4 * u) + 4 * kq) * YS + r;
      const float4 a = make_float4(y0[0], y0[YS], y0[2 * YS], y0[3 * YS]);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const float* x0 = Xs + kt * 256 * YS + (16 * (wave + 4 * u) + 4 * kq) * YS + r;
        const float4 b = make_float4(x0[0], x0[YS], x0[2 * YS], x0[3 * YS]);
        MFMA4(acc[kt], a, b);
      }
    }
    if (want_bias) {
      const int col = t & 15, part = t >> 4;
#pragma unroll
      for (int i = 0; i < 16; ++i) asum += Ys[(part * 16 + i) * YS + col];
    }
  }
  STAMP(2); BLK_PH(2);
  // sum the 4 waves' accumulators of every tile (split-M); wave kt gets the total of tile kt
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) st4(red + ((kt * 4 + wave) * 64 + lane) * 4, make_float4(acc[kt][0], acc[kt][1], acc[kt][2], acc[kt][3]));
  if (want_bias) cred[fpart * 17 + fcol] = asum;
  if (fold_vec) {                                          // lanes with equal (lane & 3) hold the same 4 columns: sum the 4 of a DPP row,
    float v[8] = {cg.x, cg.y, cg.z, cg.w, cb.x, cb.y, cb.z, cb.w};     // then one partial per (wave, row) into LDS
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] += dpp_mov<0x124>(v[j]); v[j] += dpp_mov<0x128>(v[j]); }   // row_ror:4, row_ror:8
    if ((lane & 15) < 4) {
      float* d = fsum + ((wave * 4 + (lane >> 4)) * 4 + (lane & 3)) * 8;
      st4(d, make_float4(v[0], v[1], v[2], v[3])); st4(d + 4, make_float4(v[4], v[5], v[6], v[7]));
    }
  }
  __syncthreads();
  STAMP(3); BLK_PH(3);
  if (wave < KT && ecol < kw) {
    const float* rr = red + (wave * 4 * 64 + lane) * 4;
    const float4 a = ld4(rr), b = ld4(rr + 256), c = ld4(rr + 512), d = ld4(rr + 768);
    const float o[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = n0 + 4 * (lane >> 4) + i;
      if (row < q.N) adam_commit(p, nbase + q.w_off + (long)row * q.ldw + ecol, ecol < q.K ? o[i] : 0.f, st[i], step, sq2);
    }
  }
  if (foff >= 0) {
    float v = 0.f;
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v += cred[i * 17 + fcol];
    } else {
      const float* f = fsum + (fcol >> 2) * 8 + (lane < 32 ? 0 : 4) + (fcol & 3);
#pragma unroll
      for (int i = 0; i < 16; ++i) v += f[i * 32];
    }
    adam_commit(p, foff, v, fstate, step, sq2);
  }
  STAMP(4); BLK_PH(4);
  BLK_MARK(1);
}

// ---- large-batch form of the weight gradients (M >= 1024).  k_tn's 16 x 16 tiles make every block re-read a [M][16]
// column slice of both operands: at B = 1024 that is 168 MB of L2 -> CU traffic for 21 MB of operands (rocprofv3
// FETCH_SIZE 81 MB: each XCD's L2 fills its own copy), and a block's 16 MFMAs per 256-row slab cannot cover the slab's
// fetch + LDS transpose.  Here a block owns a (16 AN WN) x (16 AK WK) tile of dW over a SLICE of the batch rows (split-M):
//   - operands stream as [TM rows][tile columns] chunks with whole-line loads, two chunks ahead of the MFMAs (register
//     sets A / B), parked in a double-buffered LDS stage in their memory layout; the MFMA fragments are read as columns
//     (4 ds_read_b32 per fragment, conflict-free with row strides = 4 mod 8): no transposed copies anywhere;
//   - tiles x slices is sized to ~2 blocks per CU, all of the same length (the dispatcher spreads them evenly: tools/census.hip);
//   - each block writes its partial tile into slab s of a [S][nets][arena] scratch; k_adam_red sums the slabs in a fixed
//     order (bit-reproducible), adds the vector gradients from the row kernels' partials and applies Adam (+ Polyak).
struct Tn64Prob {
  const float* dY; int ldy; long dy_ns; int N;
  const float* X; int ldx; long x_ns; int K;
  int w_off, ldw, b_off;                 // weight block / bias inside a net's arena (b_off < 0: none)
  int tiles_n, tiles_k, tile0;           // tile grid of this problem and its first logical tile (within a net)
};
struct Tn64Args {
  Tn64Prob pr[3]; int nprob; int M; int nets; int S;
  int tiles_per_net;                     // sum over problems of tiles_n * tiles_k
  float* Gp; long g_ns;                  // partial slabs [S][nets][g_ns]
};

// WN x WK waves (WN * WK == 4), each AN x AK MFMA tiles of 16 x 16: block tile = (16 AN WN) x (16 AK WK); TM rows per chunk.
// VALU work in the loop competes with the f32 MFMAs for the SIMD, so the loop of an INTERIOR block (tile and rows inside
// the operands: block-uniform) has no masks, scalar chunk bases + per-thread offsets computed once, and LDS addresses that
// are per-thread constants + immediates; EDGE blocks take the masked, clamped form.
template <int WN, int WK, int AN, int AK, int TM>
__global__ __launch_bounds__(256) void k_tn64(Tn64Args p) {
  constexpr int TN = 16 * AN * WN, TK = 16 * AK * WK, YS_ = TN + 4, XS_ = TK + 4;
  constexpr int YF = TM * TN / 4 / 256, XF = TM * TK / 4 / 256;      // float4 per thread and chunk
  constexpr int G = TM / 16;
  static_assert(WN * WK == 4 && YF >= 1 && XF >= 1 && TM % 16 == 0 && (TM * TN) % 1024 == 0 && (TM * TK) % 1024 == 0, "shape");
  __shared__ __attribute__((aligned(16))) float Ys[2 * TM * YS_];
  __shared__ __attribute__((aligned(16))) float Xs[2 * TM * XS_];
  __shared__ float csum[256];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wn = wave % WN, wk = wave / WN;
  const int r = lane & 15, kq = lane >> 4;
  // logical tile: blocks are dealt to the 8 XCDs round-robin; give each XCD a contiguous run of tiles (order: net, problem,
  // n tile, k tile, slice) so that the operand columns its CUs share are filled into that XCD's L2 once
  int L = blockIdx.x;
  { const int per = (int)gridDim.x >> 3; if (L < per * 8) L = (L & 7) * per + (L >> 3); }
  const int sl = L % p.S; L /= p.S;
  const int net = L / p.tiles_per_net; L -= net * p.tiles_per_net;
  int pi = 0;
  if (p.nprob > 1 && L >= p.pr[1].tile0) pi = 1;
  if (p.nprob > 2 && L >= p.pr[2].tile0) pi = 2;
  const Tn64Prob q = p.pr[pi];   // by value: one batch of scalar loads (see k_tn)
  L -= q.tile0;
  const int tn = L / q.tiles_k, tk = L - tn * q.tiles_k;
  const int n0 = tn * TN, k0 = tk * TK;
  // this slice's chunks of TM rows: the first (nch % S) slices take one chunk more
  const int nch = (p.M + TM - 1) / TM, base = nch / p.S, extra = nch % p.S;
  const int cb = sl * base + min(sl, extra), nc = base + (sl < extra ? 1 : 0);
  const float* dYn = q.dY + net * q.dy_ns;
  const float* Xn = q.X + net * q.x_ns;
  const int Nr = (q.N + 3) & ~3, Kr = (q.K + 3) & ~3;
  const bool edge = n0 + TN > q.N || k0 + TK > q.K || (cb + nc) * TM > p.M;     // block-uniform
  STAMP(0);
  // per-thread constants: element offsets of its float4s inside a chunk, LDS addresses of the parked copies / of its fragments
  int yo[YF], xo[XF], yl[YF], xl[XF];
#pragma unroll
  for (int u = 0; u < YF; ++u) { const int i = t + 256 * u, row = i / (TN / 4), c4 = i % (TN / 4); yo[u] = row * q.ldy + n0 + 4 * c4; yl[u] = row * YS_ + 4 * c4; }
#pragma unroll
  for (int u = 0; u < XF; ++u) { const int i = t + 256 * u, row = i / (TK / 4), c4 = i % (TK / 4); xo[u] = row * q.ldx + k0 + 4 * c4; xl[u] = row * XS_ + 4 * c4; }
  const float* yfrag = Ys + (4 * kq) * YS_ + 16 * AN * wn + r;
  const float* xfrag = Xs + (4 * kq) * XS_ + 16 * AK * wk + r;
  float4 ya[YF], xa[XF], yb[YF], xb[XF];
  f32x4 acc[AN][AK];
#pragma unroll
  for (int i = 0; i < AN; ++i)
#pragma unroll
    for (int j = 0; j < AK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;                                         // thread (n = t % TN, part = t / TN): column sums of dY (k tile 0 only)
  const bool want_bias = tk == 0 && q.b_off >= 0;

  auto fetch = [&](auto edge_tag, int c, float4* y, float4* x) {     // raw loads: first touched when parked, two chunks later
    constexpr bool EDGE = decltype(edge_tag)::value;
    const int m = (cb + c) * TM;
    if (!EDGE) {
      const float* yb_ = dYn + (long)m * q.ldy;                      // scalar bases: the per-thread part is constant
      const float* xb_ = Xn + (long)m * q.ldx;
#pragma unroll
      for (int u = 0; u < YF; ++u) y[u] = ld4(yb_ + yo[u]);
#pragma unroll
      for (int u = 0; u < XF; ++u) x[u] = ld4(xb_ + xo[u]);
    } else {
#pragma unroll
      for (int u = 0; u < YF; ++u) { const int i = t + 256 * u, row = i / (TN / 4), c4 = i % (TN / 4); y[u] = ld4_raw(dYn + (long)min(m + row, p.M - 1) * q.ldy, n0 + 4 * c4, Nr); }
#pragma unroll
      for (int u = 0; u < XF; ++u) { const int i = t + 256 * u, row = i / (TK / 4), c4 = i % (TK / 4); x[u] = ld4_raw(Xn + (long)min(m + row, p.M - 1) * q.ldx, k0 + 4 * c4, Kr); }
    }
  };
  auto park = [&](auto edge_tag, int c, int buf, const float4* y, const float4* x) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    const int m = (cb + c) * TM;
#pragma unroll
    for (int u = 0; u < YF; ++u) {
      const int i = t + 256 * u, row = i / (TN / 4), c4 = i % (TN / 4);
      st4(Ys + buf * TM * YS_ + yl[u], EDGE ? mask4_cols(y[u], n0 + 4 * c4, q.N, m + row < p.M) : y[u]);
    }
#pragma unroll
    for (int u = 0; u < XF; ++u) {
      const int i = t + 256 * u, row = i / (TK / 4), c4 = i % (TK / 4);
      st4(Xs + buf * TM * XS_ + xl[u], EDGE ? mask4_cols(x[u], k0 + 4 * c4, q.K, m + row < p.M) : x[u]);
    }
  };
  // fragments of 16-row group g: columns of the parked chunk (4 ds_read_b32 each; conflict-free, row strides = 4 mod 8)
  auto ldfrag = [&](int buf, int g, float4* a, float4* b) {
    const float* y0 = yfrag + buf * TM * YS_ + 16 * g * YS_;
    const float* x0 = xfrag + buf * TM * XS_ + 16 * g * XS_;
#pragma unroll
    for (int i = 0; i < AN; ++i) a[i] = make_float4(y0[16 * i], y0[YS_ + 16 * i], y0[2 * YS_ + 16 * i], y0[3 * YS_ + 16 * i]);
#pragma unroll
    for (int j = 0; j < AK; ++j) b[j] = make_float4(x0[16 * j], x0[XS_ + 16 * j], x0[2 * XS_ + 16 * j], x0[3 * XS_ + 16 * j]);
  };
  auto mfma16 = [&](const float4* a, const float4* b) {     // independent accumulators interleaved: no MFMA waits on its predecessor
#pragma unroll
    for (int i = 0; i < AN; ++i)
#pragma unroll
      for (int j = 0; j < AK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < AN; ++i)
#pragma unroll
      for (int j = 0; j < AK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < AN; ++i)
#pragma unroll
      for (int j = 0; j < AK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < AN; ++i)
#pragma unroll
      for (int j = 0; j < AK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
  };
  // group g + 1's fragments are requested before group g's MFMAs (one wave per SIMD: nothing else covers an LDS read)
  auto compute = [&](int buf) {
    float4 a0[AN], b0[AK], a1[AN], b1[AK];
    ldfrag(buf, 0, a0, b0);
#pragma unroll
    for (int g = 0; g < G; g += 2) {       // (sched_barrier: the compiler otherwise re-serialises read -> wait -> 4 MFMAs)
      if (g + 1 < G) ldfrag(buf, g + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 2 < G) ldfrag(buf, g + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < G) mfma16(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (want_bias) {                                        // (block-uniform)
      constexpr int PARTS = 256 / TN, RP = TM / PARTS;
      const float* yc0 = Ys + buf * TM * YS_ + ((t / TN) * RP) * YS_ + (t % TN);
#pragma unroll
      for (int i = 0; i < RP; ++i) asum += yc0[i * YS_];
    }
  };
  auto pipeline = [&](auto edge_tag) {
    // prologue: chunks 0 and 1 requested, chunk 0 parked
    fetch(edge_tag, 0, ya, xa);
    if (nc > 1) fetch(edge_tag, 1, yb, xb);
    __builtin_amdgcn_sched_barrier(0);
    park(edge_tag, 0, 0, ya, xa);
    __syncthreads();
    STAMP(1);
    for (int c = 0; c < nc; c += 2) {
      // even chunk c (LDS buffer 0): set A is free (chunk c was parked), chunk c + 1 waits in set B
      if (c + 2 < nc) fetch(edge_tag, c + 2, ya, xa);
      __builtin_amdgcn_sched_barrier(0);
      compute(0);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < nc) park(edge_tag, c + 1, 1, yb, xb);
      __syncthreads();
      if (c + 1 >= nc) break;
      // odd chunk c + 1 (LDS buffer 1): set B is free, chunk c + 2 waits in set A
      if (c + 3 < nc) fetch(edge_tag, c + 3, yb, xb);
      __builtin_amdgcn_sched_barrier(0);
      compute(1);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < nc) park(edge_tag, c + 2, 0, ya, xa);
      __syncthreads();
    }
  };
  if (edge) pipeline(std::true_type{}); else pipeline(std::false_type{});
  STAMP(2);
  // partial tile -> slab `sl` (columns K .. ldw-1 of a weight row are padding: the masked X columns make them exact zeros)
  float* Gs = p.Gp + ((long)sl * p.nets + net) * p.g_ns;
#pragma unroll
  for (int j = 0; j < AK; ++j) {
    const int col = k0 + 16 * (AK * wk + j) + r;
    if (col >= q.ldw) continue;
#pragma unroll
    for (int i2 = 0; i2 < AN; ++i2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = n0 + 16 * (AN * wn + i2) + 4 * kq + i;
        if (row < q.N) Gs[q.w_off + (long)row * q.ldw + col] = acc[i2][j][i];
      }
  }
  if (want_bias) {
    constexpr int PARTS = 256 / TN;
    csum[t] = asum;
    __syncthreads();
    if (t < TN && n0 + t < q.N) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < PARTS; ++i) v += csum[i * TN + t];
      Gs[q.b_off + n0 + t] = v;
    }
  }
  STAMP(3);
}

// grid = (main blocks + 4 nvec vector blocks + 1 scalar block, nets); main blocks = ceil(g_ns / 1024)
__global__ __launch_bounds__(256) void k_adam_red(AdamRedArgs a) {
  const int net = blockIdx.y, t = threadIdx.x;
  const int main_blocks = (int)((a.g_ns / 4 + 255) / 256);
  const int bx = blockIdx.x;
  if (bx >= main_blocks) { adam_red_tail_body(a, bx - main_blocks, net); return; }
  const float step = a.apply ? a.adam[0] : 0.f, sq2 = a.apply ? a.adam[1] : 1.f;
  // slab-sourced elements: one float4 per thread (the vector ranges and the scalar's float4 belong to the tail blocks)
  const long i = ((long)bx * 256 + t) * 4;
  bool mine = i < a.g_ns && !(a.s_off >= 0 && i == a.s_off);
#pragma unroll
  for (int e = 0; e < 5; ++e)
    if (e < a.nvec && i >= a.vec[e].off && i < a.vec[e].off + HID) mine = false;
  if (mine) {
    const long off = net * a.g_ns + i;
    float4 w = f4(0.f), m = f4(0.f), v = f4(0.f), tt = f4(0.f);
    if (a.apply) { w = ld4(a.P + off); m = ld4(a.Mo + off); v = ld4(a.Vo + off); if (a.T) tt = ld4(a.T + off); }
    float4 gs[8];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) gs[sl] = ld4(a.Gp + ((long)min(sl, a.S - 1) * a.nets + net) * a.g_ns + i);   // all requests first
    float4 g = gs[0];
#pragma unroll
    for (int sl = 1; sl < 8; ++sl) if (sl < a.S) g = g + gs[sl];
    adam_red_commit(a, off, g, w, m, v, tt, step, sq2);
  }
}

// ------------------------------------------------------------------------------------------------ row kernels
struct ActorTail {
  const float* z2;                       // [B][HID] pre-LN output of hidden layer 2
  const float* P; NetLayout L;           // actor parameter block (online or target)
  int B, o, a, ln, sac, mode, train;
  // mode: SAC 0 = sample, 1 = mode (tanh(mean));  TD3 0 = policy, 1 = target smoothing, 2 = explore
  const DevCtl* ctl; const int* ctr; int site_buf; unsigned site_code;
  float* eps;                            // [B][a] draws used (written in native mode, read when injected)
  const float* scale; const float* bias; const float* min_ac; const float* max_ac;
  float* dst; int ldd; int dst_off;      // action -> dst[b * ldd + dst_off + j]
  const float* obs_src; int lds;         // optional: dst[b][0:o] = obs_src[b][0:o] first (builds [s | pi(s)] rows)
  float* logp;                           // [B] (SAC)
  float* h2; float* xh2; float* rstd2; float* tg;   // train stores: h2, xhat2 [B][HID], rstd2 [B], tg [B][4][a4] (t, std, y)
  int a4;
  float td3_std, td3_c, noise_std;
  int ctr_add;                           // the primary draw uses stream counter *ctr + ctr_add
  int* tick; int tick_add;               // optional counter advanced by 1 + tick_add by (block 0, thread 0)
  // SAC, dual mode: a SECOND, gradient-free draw through the same head outputs (the temperature step's fresh sample,
  // agents/agent.py:297-299) sharing this kernel with the next actor update's sample: only its log-prob is kept
  int dual; int site_buf2; unsigned site_code2; float* eps2; float* logp2;
  int eps_ready;                         // the eps buffers were filled by the preceding trunk launch's noise blocks: load, draw nothing
  // acting path, single-block launches only: after its last store the block bumps *seq and publishes the new value to a pinned
  // host word -- the host spins on that word instead of paying a stream synchronisation (marker packet + signal wake-up)
  int* seq; int* done_flag;
};
__device__ __forceinline__ void tail_publish(const ActorTail& p, int seq_v) {      // thread 0, after a system-scope fence by all
  *p.seq = seq_v + 1;
  __hip_atomic_store(p.done_flag, seq_v + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Memory discipline of this kernel (and of every kernel here): ALL global loads of the common case go out first, behind
// uniform branches only and with clamped addresses + selects instead of per-lane conditions; they are made to land
// (PIN) before the first global store.  A load that follows a store, or one that is first used inside a divergent
// branch, costs a full drain of the memory queue (s_waitcnt vmcnt(0)) -- a round trip each, and there were ~20 here.
__device__ __forceinline__ void actor_tail_body(const ActorTail& p, int block) {
  __shared__ __attribute__((aligned(16))) float Hs[16 * AS];
  __shared__ float Up[4 * 16 * 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row = t >> 4, sub = t & 15, r = lane & 15, kq = lane >> 4;
  const int b = block * 16 + row, bc = min(b, p.B - 1);
  const bool valid = b < p.B;
  const int nh = p.L.nh, T = (nh + 15) >> 4;     // head column tiles (<= 4)
  const float* Wh = p.P + p.L.Wh;
  STAMP(0);
  // ---- loads: counter, stream state, my row, LN affine, my head-weight fragments (wave w: k chunks 4w .. 4w+3),
  // the operands of this thread's first output element (j = sub), its injected draws, the observation slice to copy
  const bool ticker = p.tick && block == 0 && t == 0;
  int tick_v = 0, seq_v = 0;
  if (ticker) tick_v = *p.tick;
  if (p.done_flag && t == 0) seq_v = *p.seq;     // (per-lane like the ticker's: a uniform read would be waited for on the spot)
  const bool need_eps = p.sac ? (p.mode == 0) : (p.mode != 0);
  // (eps_ready: the buffers already hold this launch's draws -- treated like injected ones)
  // (the injection flags are fetched unconditionally and combined only where they are used: inside a `need_eps ? load | x : 0`
  //  the compiler waits for the load on the spot, one round trip ahead of everything below)
  const int inj1_raw = p.ctl->inject_eps[p.site_buf], inj2_raw = p.ctl->inject_eps[p.dual ? p.site_buf2 : p.site_buf];
  const unsigned long long seed = p.ctl->seed;
  const int ctr = *p.ctr;
  const Row16 z = row_ld(p.z2 + (long)bc * HID, sub);
  const Row16 g = row_ld(p.P + (p.ln ? p.L.g2 : 0), sub), be = row_ld(p.P + (p.ln ? p.L.be2 : 0), sub);
  float4 wf[4][4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      // unconditional, row clamped (tiles beyond the head re-read its last row: L1 hits); rows >= nh are zeroed
      // after the loads are in -- not here, where a select on the value would wait for it
      wf[tt][ci] = ld4(Wh + (long)min(tt * 16 + r, nh - 1) * HID + (4 * wave + ci) * 16 + 4 * kq);
    }
  const int j0 = min(sub, p.a - 1);
  const long ej = (long)bc * p.a + j0;
  float e_bh0 = p.P[p.L.bh + j0], e_bh1 = p.sac ? p.P[p.L.bh + p.a + j0] : 0.f;
  float e_sc = p.scale[j0], e_bi = p.bias[j0];
  const bool smooth = !p.sac && p.mode == 1;     // TD3 target smoothing clamps to the action bounds
  float e_lo = 0.f, e_hi = 0.f;
  if (smooth) { e_lo = p.min_ac[j0]; e_hi = p.max_ac[j0]; }
  float e_in1 = 0.f, e_in2 = 0.f;
  if (need_eps) e_in1 = p.eps[ej];               // fetched whether or not it is an injected draw: branching on the
  if (p.dual) e_in2 = p.eps2[ej];                // (loaded) injection flag here would hold back every later request
  // a second output element per thread when ac_dim > 16 (Humanoid: 17): its operands are requested here with everything else
  // (fetched one by one behind the first element, as this kernel first did, they were 3-4 dependent round trips)
  const bool two = p.a > 16;                     // (uniform)
  const int j1 = min(sub + 16, p.a - 1);
  const long ej1 = (long)bc * p.a + j1;
  float f_bh0 = 0.f, f_bh1 = 0.f, f_sc = 0.f, f_bi = 0.f, f_lo = 0.f, f_hi = 0.f, f_in1 = 0.f, f_in2 = 0.f;
  if (two) {
    f_bh0 = p.P[p.L.bh + j1]; f_bh1 = p.sac ? p.P[p.L.bh + p.a + j1] : 0.f;
    f_sc = p.scale[j1]; f_bi = p.bias[j1];
    if (smooth) { f_lo = p.min_ac[j1]; f_hi = p.max_ac[j1]; }
    if (need_eps) f_in1 = p.eps[ej1];
    if (p.dual) f_in2 = p.eps2[ej1];
  }
  // observation slice to copy (builds [s | pi(s)] rows): 16-byte chunks sub, sub + 16, ... of the row (up to 8 per thread:
  // 512 floats) and the < 4 trailing floats (the chunk that straddles the action columns is NOT moved as a whole: another
  // thread writes the action there); wider observations finish in a loop behind the stores
  const int o4 = p.o >> 2, orem = p.o & 3;
  constexpr int OBC = 8;
  float4 ob4[OBC];
  float obt = 0.f;
  const int nob = p.obs_src ? min((o4 + 15) >> 4, OBC) : 0;     // (uniform) chunks per thread actually needed
#pragma unroll
  for (int i = 0; i < OBC; ++i) ob4[i] = f4(0.f);
  if (p.obs_src) {
    const float* srow = p.obs_src + (long)bc * p.lds;
#pragma unroll
    for (int i = 0; i < OBC; ++i)
      if (i < nob) ob4[i] = ld4(srow + 4 * min(sub + 16 * i, max(o4 - 1, 0)));
    obt = srow[min(4 * o4 + sub, p.o - 1)];
  }
  // native draws (no memory involved)
  const bool gen = !p.eps_ready;                 // (uniform)
  const float e_nat1 = (gen && need_eps) ? philox_normal(seed, (unsigned)(ctr + p.ctr_add), p.site_code, (unsigned)ej) : 0.f;
  const float e_nat2 = (gen && p.dual) ? philox_normal(seed, (unsigned)ctr, p.site_code2, (unsigned)ej) : 0.f;
  const float f_nat1 = (gen && two && need_eps) ? philox_normal(seed, (unsigned)(ctr + p.ctr_add), p.site_code, (unsigned)ej1) : 0.f;
  const float f_nat2 = (gen && two && p.dual) ? philox_normal(seed, (unsigned)ctr, p.site_code2, (unsigned)ej1) : 0.f;
  STAMP(1);
  Row16 xh, y; float rstd;
  ln_fwd(z, g, be, p.ln, xh, y, rstd);
  Row16 h;
#pragma unroll
  for (int q = 0; q < 4; ++q) h.v[q] = relu4(y.v[q]);
  row_st(Hs + row * AS, sub, h);
  // everything requested above has to be here before the first store goes out
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      PIN(wf[tt][ci].x); PIN(wf[tt][ci].y); PIN(wf[tt][ci].z); PIN(wf[tt][ci].w);
      if (tt * 16 + r >= nh) wf[tt][ci] = f4(0.f);
    }
  PIN(e_bh0); PIN(e_bh1); PIN(e_sc); PIN(e_bi); PIN(e_lo); PIN(e_hi); PIN(e_in1); PIN(e_in2); PIN(tick_v);
  PIN(f_bh0); PIN(f_bh1); PIN(f_sc); PIN(f_bi); PIN(f_lo); PIN(f_hi); PIN(f_in1); PIN(f_in2);
#pragma unroll
  for (int i = 0; i < OBC; ++i) { PIN(ob4[i].x); PIN(ob4[i].y); PIN(ob4[i].z); PIN(ob4[i].w); }
  PIN(obt);
  const int inj1 = (inj1_raw | p.eps_ready) & -(int)need_eps, inj2 = (inj2_raw | p.eps_ready) & -(int)(p.dual != 0);   // (branch-free: see above)
  const float e_eps = (need_eps && sub < p.a) ? (inj1 ? e_in1 : e_nat1) : 0.f;
  const float e_eps2 = (p.dual && sub < p.a) ? (inj2 ? e_in2 : e_nat2) : 0.f;
  const bool has2 = two && sub + 16 < p.a;
  const float f_eps = (need_eps && has2) ? (inj1 ? f_in1 : f_nat1) : 0.f;
  const float f_eps2 = (p.dual && has2) ? (inj2 ? f_in2 : f_nat2) : 0.f;
  if (valid) {
    if (p.train) {
      row_st(p.h2 + (long)b * HID, sub, h);
      row_st(p.xh2 + (long)b * HID, sub, xh);
      if (sub == 0) p.rstd2[b] = rstd;
    }
    if (sub < p.a) {
      if (need_eps && !inj1) p.eps[ej] = e_nat1;
      if (p.dual && !inj2) p.eps2[ej] = e_nat2;
    }
    if (has2) {
      if (need_eps && !inj1) p.eps[ej1] = f_nat1;
      if (p.dual && !inj2) p.eps2[ej1] = f_nat2;
    }
    if (p.obs_src) {
      const float* __restrict__ src = p.obs_src + (long)b * p.lds;
      float* __restrict__ dst = p.dst + (long)b * p.ldd;
#pragma unroll
      for (int i = 0; i < OBC; ++i)
        if (sub + 16 * i < o4) st4(dst + 4 * (sub + 16 * i), ob4[i]);
      if (sub < orem) dst[4 * o4 + sub] = obt;
      for (int c = sub + 16 * OBC; c < o4; c += 64) {      // observations wider than 512 floats: four chunks in flight per trip
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ld4(src + 4 * min(c + 16 * i, o4 - 1));
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (c + 16 * i < o4) st4(dst + 4 * (c + 16 * i), v[i]);
      }
    }
  }
  __syncthreads();
  STAMP(2);
  // head: u[16][nh] = h[16][256] Wh^T, K split over the 4 waves
  {
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const float4 av = ld4(Hs + r * AS + (4 * wave + ci) * 16 + 4 * kq);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        if (tt < T) { MFMA4(acc[tt], av, wf[tt][ci]); }
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
      if (tt < T)
#pragma unroll
        for (int i = 0; i < 4; ++i) Up[(wave * 16 + 4 * kq + i) * 64 + tt * 16 + r] = acc[tt][i];
  }
  __syncthreads();
  STAMP(3);
  float lp = 0.f, lp2 = 0.f;
  auto element = [&](int j, float bh0, float bh1, float sc, float bi, float lo, float hi, float e, float e2) {
    const float* u = Up + row * 64;
    const float u0 = ((u[j] + u[1024 + j]) + (u[2048 + j] + u[3072 + j])) + bh0;
    float act;
    if (p.sac) {
      const int j1 = p.a + j;
      const float u1 = ((u[j1] + u[1024 + j1]) + (u[2048 + j1] + u[3072 + j1])) + bh1;
      const float tt = tanhf(u1);
      const float log_std = -5.0f + 3.5f * (tt + 1.0f);
      const float sd = expf(log_std);
      const float x = u0 + e * sd;
      const float yt = tanhf(x);
      act = yt * sc + bi;
      const float dx = x - u0;
      float l = -(dx * dx) / (2.0f * sd * sd) - logf(sd) - 0.9189385332046727f;
      l -= logf(sc * (1.0f - yt * yt) + 1e-6f);
      lp += l;
      if (p.dual) {                                      // second draw: log-prob only
        const float x2 = u0 + e2 * sd, y2 = tanhf(x2), d2 = x2 - u0;
        lp2 += -(d2 * d2) / (2.0f * sd * sd) - logf(sd) - 0.9189385332046727f - logf(sc * (1.0f - y2 * y2) + 1e-6f);
      }
      if (p.mode == 1) act = tanhf(u0) * sc + bi;
      if (p.train && valid) {
        float* tg = p.tg + (long)b * 4 * p.a4;
        tg[j] = tt; tg[p.a4 + j] = sd; tg[2 * p.a4 + j] = yt;
      }
    } else {
      const float th = tanhf(u0);
      act = th * sc + bi;
      if (p.mode == 1) {
        const float nz = fminf(fmaxf(e * p.td3_std, -p.td3_c), p.td3_c);
        act = fminf(fmaxf(act + nz, lo), hi);
      } else if (p.mode == 2) {
        act = act + e * (sc * p.noise_std);
      }
      if (p.train && valid) p.tg[(long)b * 4 * p.a4 + j] = th;
    }
    if (valid) p.dst[(long)b * p.ldd + p.dst_off + j] = act;
  };
  if (sub < p.a) element(sub, e_bh0, e_bh1, e_sc, e_bi, e_lo, e_hi, e_eps, e_eps2);       // operands already in registers
  if (has2) element(sub + 16, f_bh0, f_bh1, f_sc, f_bi, f_lo, f_hi, f_eps, f_eps2);       // ac_dim in 17 .. 32
  STAMP(4);
  if (p.sac && p.logp) {
    lp = row16_sum(lp);
    if (sub == 0 && valid) p.logp[b] = lp;
  }
  if (p.dual) {
    lp2 = row16_sum(lp2);
    if (sub == 0 && valid) p.logp2[b] = lp2;
  }
  if (ticker) *p.tick = tick_v + 1 + p.tick_add;
  if (p.done_flag) {                             // (uniform)
    __threadfence_system();
    __syncthreads();
    if (t == 0) tail_publish(p, seq_v);
  }
  STAMP(5);
}
__global__ __launch_bounds__(256) void k_actor_tail(ActorTail p) { actor_tail_body(p, blockIdx.x); }
// two tails in one launch (wide heads; see k_actor_tail_s2)
__global__ __launch_bounds__(256) void k_actor_tail2(ActorTail a, ActorTail b, int nb_a) {
  if ((int)blockIdx.x < nb_a) actor_tail_body(a, blockIdx.x);
  else actor_tail_body(b, blockIdx.x - nb_a);
}
// ... up to five (a pipelined period: the last actor update's temperature draw + the next-action passes of the two critic-only
// iterations [+ the next period's opening pair: next-action pass and policy pass]); nb blocks per tail, n tails
struct ActorTail5 { ActorTail t[5]; int nb; int n; };
__global__ __launch_bounds__(256) void k_actor_tail5(ActorTail5 p) {
  const int which = (int)blockIdx.x / p.nb, blk = (int)blockIdx.x - which * p.nb;       // (selects, not a run-time index: kernarg loads stay scalar)
  if (which == 0) actor_tail_body(p.t[0], blk);
  else if (which == 1) actor_tail_body(p.t[1], blk);
  else if (which == 2) actor_tail_body(p.t[2], blk);
  else if (which == 3) actor_tail_body(p.t[3], blk);
  else actor_tail_body(p.t[4], blk);
}

// Narrow heads (nh <= 8: Hopper's SAC head 2 x 3, HalfCheetah's TD3 head 6 ...).  The general kernel above spends two block
// barriers, an LDS round trip each way and an MFMA tile on a [16 x 256] x [256 x 6] product; here a row lives in ONE wave
// (RPB rows x 16 threads = 64 threads per block, so 4x as many blocks), the head is nh dot products reduced with DPP, and there
// is no LDS, no barrier and no MFMA at all.  Same arithmetic per element as k_actor_tail (the dot products' summation order
// differs, inside the 1e-5 budget of north_star); dual draws, training stores, the [s | pi(s)] row build and the counter
// tick behave identically.
template <int RPB>
__device__ __forceinline__ void actor_tail_s_body(const ActorTail& p, int block) {
  const int t = threadIdx.x, row = t >> 4, sub = t & 15;
  const int b = block * RPB + row, bc = min(b, p.B - 1);
  const bool valid = b < p.B;
  const int nh = p.L.nh;                         // <= 8, and a <= 8
  const float* Wh = p.P + p.L.Wh;
  const bool ticker = p.tick && block == 0 && t == 0;
  int tick_v = 0, seq_v = 0;
  if (ticker) tick_v = *p.tick;
  if (p.done_flag && t == 0) seq_v = *p.seq;     // (per-lane like the ticker's: a uniform read would be waited for on the spot)
  const bool need_eps = p.sac ? (p.mode == 0) : (p.mode != 0);
  // (the injection flags are fetched unconditionally and combined only where they are used: inside a `need_eps ? load | x : 0`
  //  the compiler waits for the load on the spot, one round trip ahead of everything below)
  const int inj1_raw = p.ctl->inject_eps[p.site_buf], inj2_raw = p.ctl->inject_eps[p.dual ? p.site_buf2 : p.site_buf];
  const unsigned long long seed = p.ctl->seed;
  const int ctr = *p.ctr;
  const Row16 z = row_ld(p.z2 + (long)bc * HID, sub);
  const Row16 g = row_ld(p.P + (p.ln ? p.L.g2 : 0), sub), be = row_ld(p.P + (p.ln ? p.L.be2 : 0), sub);
  Row16 w[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) w[n] = row_ld(Wh + (long)min(n, nh - 1) * HID, sub);   // unconditional, row clamped
  const int j0 = min(sub, p.a - 1);
  const long ej = (long)bc * p.a + j0;
  float e_bh0 = p.P[p.L.bh + j0], e_bh1 = p.sac ? p.P[p.L.bh + p.a + j0] : 0.f;
  float e_sc = p.scale[j0], e_bi = p.bias[j0];
  const bool smooth = !p.sac && p.mode == 1;
  float e_lo = 0.f, e_hi = 0.f;
  if (smooth) { e_lo = p.min_ac[j0]; e_hi = p.max_ac[j0]; }
  float e_in1 = 0.f, e_in2 = 0.f;
  if (need_eps) e_in1 = p.eps[ej];
  if (p.dual) e_in2 = p.eps2[ej];
  const int o4 = p.o >> 2, orem = p.o & 3;
  float4 ob4[2] = {f4(0.f), f4(0.f)};
  float obt = 0.f;
  if (p.obs_src) {
    const float* srow = p.obs_src + (long)bc * p.lds;
#pragma unroll
    for (int i = 0; i < 2; ++i) ob4[i] = ld4(srow + 4 * min(sub + 16 * i, max(o4 - 1, 0)));
    obt = srow[min(4 * o4 + sub, p.o - 1)];
  }
  const bool gen = !p.eps_ready;
  const float e_nat1 = (gen && need_eps) ? philox_normal(seed, (unsigned)(ctr + p.ctr_add), p.site_code, (unsigned)ej) : 0.f;
  const float e_nat2 = (gen && p.dual) ? philox_normal(seed, (unsigned)ctr, p.site_code2, (unsigned)ej) : 0.f;
  Row16 xh, y; float rstd;
  ln_fwd(z, g, be, p.ln, xh, y, rstd);
  Row16 h;
#pragma unroll
  for (int q = 0; q < 4; ++q) h.v[q] = relu4(y.v[q]);
  // head: every lane of the row gets all nh outputs; lane `sub` keeps output sub (and a + sub for the SAC log-std half)
  float u0 = 0.f, u1 = 0.f;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    row_pin(w[n]);
    const float tot = row16_sum(row_dot(h, w[n]));
    if (n < nh) { u0 = (n == sub) ? tot : u0; u1 = (n == p.a + sub) ? tot : u1; }
  }
  PIN(e_bh0); PIN(e_bh1); PIN(e_sc); PIN(e_bi); PIN(e_lo); PIN(e_hi); PIN(e_in1); PIN(e_in2); PIN(tick_v);
#pragma unroll
  for (int i = 0; i < 2; ++i) { PIN(ob4[i].x); PIN(ob4[i].y); PIN(ob4[i].z); PIN(ob4[i].w); }
  PIN(obt);
  const int inj1 = (inj1_raw | p.eps_ready) & -(int)need_eps, inj2 = (inj2_raw | p.eps_ready) & -(int)(p.dual != 0);   // (branch-free: see above)
  const float e = (need_eps && sub < p.a) ? (inj1 ? e_in1 : e_nat1) : 0.f;
  const float e2 = (p.dual && sub < p.a) ? (inj2 ? e_in2 : e_nat2) : 0.f;
  if (valid) {
    if (p.train) {
      row_st(p.h2 + (long)b * HID, sub, h);
      row_st(p.xh2 + (long)b * HID, sub, xh);
      if (sub == 0) p.rstd2[b] = rstd;
    }
    if (sub < p.a) {
      if (need_eps && !inj1) p.eps[ej] = e_nat1;
      if (p.dual && !inj2) p.eps2[ej] = e_nat2;
    }
    if (p.obs_src) {
      const float* __restrict__ src = p.obs_src + (long)b * p.lds;
      float* __restrict__ dst = p.dst + (long)b * p.ldd;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (sub + 16 * i < o4) st4(dst + 4 * (sub + 16 * i), ob4[i]);
      if (sub < orem) dst[4 * o4 + sub] = obt;
      for (int c = sub + 32; c < o4; c += 16) st4(dst + 4 * c, ld4(src + 4 * c));      // (observations wider than 128 floats)
    }
  }
  float lp = 0.f, lp2 = 0.f;
  if (sub < p.a) {
    const int j = sub;
    const float v0 = u0 + e_bh0;
    float act;
    if (p.sac) {
      const float v1 = u1 + e_bh1;
      const float tt = tanhf(v1);
      const float log_std = -5.0f + 3.5f * (tt + 1.0f);
      const float sd = expf(log_std);
      const float x = v0 + e * sd;
      const float yt = tanhf(x);
      act = yt * e_sc + e_bi;
      const float dx = x - v0;
      float l = -(dx * dx) / (2.0f * sd * sd) - logf(sd) - 0.9189385332046727f;
      l -= logf(e_sc * (1.0f - yt * yt) + 1e-6f);
      lp = l;
      if (p.dual) {
        const float x2 = v0 + e2 * sd, y2 = tanhf(x2), d2 = x2 - v0;
        lp2 = -(d2 * d2) / (2.0f * sd * sd) - logf(sd) - 0.9189385332046727f - logf(e_sc * (1.0f - y2 * y2) + 1e-6f);
      }
      if (p.mode == 1) act = tanhf(v0) * e_sc + e_bi;
      if (p.train && valid) {
        float* tg = p.tg + (long)b * 4 * p.a4;
        tg[j] = tt; tg[p.a4 + j] = sd; tg[2 * p.a4 + j] = yt;
      }
    } else {
      const float th = tanhf(v0);
      act = th * e_sc + e_bi;
      if (p.mode == 1) {
        const float nz = fminf(fmaxf(e * p.td3_std, -p.td3_c), p.td3_c);
        act = fminf(fmaxf(act + nz, e_lo), e_hi);
      } else if (p.mode == 2) {
        act = act + e * (e_sc * p.noise_std);
      }
      if (p.train && valid) p.tg[(long)b * 4 * p.a4 + j] = th;
    }
    if (valid) p.dst[(long)b * p.ldd + p.dst_off + j] = act;
  }
  if (p.sac && p.logp) {
    lp = row16_sum(lp);
    if (sub == 0 && valid) p.logp[b] = lp;
  }
  if (p.dual) {
    lp2 = row16_sum(lp2);
    if (sub == 0 && valid) p.logp2[b] = lp2;
  }
  if (ticker) *p.tick = tick_v + 1 + p.tick_add;
  if (p.done_flag) {                             // (uniform; RPB == 4: the block is one wave)
    __threadfence_system();
    if (RPB > 4) __syncthreads();
    if (t == 0) tail_publish(p, seq_v);
  }
}
template <int RPB>
__global__ __launch_bounds__(16 * RPB) void k_actor_tail_s(ActorTail p) { actor_tail_s_body<RPB>(p, blockIdx.x); }
// two tails in one launch (blocks [0, nb_a) run `a`, the rest `b`): the critic update's target-action tail on s' and the first
// actor update's policy tail on s -- same actor parameters, nothing between them depends on the other
template <int RPB>
__global__ __launch_bounds__(16 * RPB) void k_actor_tail_s2(ActorTail a, ActorTail b, int nb_a) {
  if ((int)blockIdx.x < nb_a) actor_tail_s_body<RPB>(a, blockIdx.x);
  else actor_tail_s_body<RPB>(b, blockIdx.x - nb_a);
}
template <int RPB>
__global__ __launch_bounds__(16 * RPB) void k_actor_tail_s5(ActorTail5 p) {
  const int which = (int)blockIdx.x / p.nb, blk = (int)blockIdx.x - which * p.nb;
  if (which == 0) actor_tail_s_body<RPB>(p.t[0], blk);
  else if (which == 1) actor_tail_s_body<RPB>(p.t[1], blk);
  else if (which == 2) actor_tail_s_body<RPB>(p.t[2], blk);
  else if (which == 3) actor_tail_s_body<RPB>(p.t[3], blk);
  else actor_tail_s_body<RPB>(p.t[4], blk);
}

struct CriticTail {
  const float* z2t; const float* z2;     // [2][B][HID] target / online pre-LN layer-2 outputs
  const float* PT; const float* P; long p_ns; NetLayout L;
  const float* rew; const float* done; const float* logp_next; const float* log_alpha;
  int B, ln, sac, bcq; float gamma;
  float* qt; float* y; float* q;         // [2][B], [B], [2][B]
  float* dz2;                            // [2][B][HID]
  float* part; float* part_s; int pstride;  // [2][pstride][NSLOT][HID], [2][pstride][2] (sum dq, sum sq-err)
};

// RPB rows per block (blockDim = 16 RPB): fewer rows per block = fewer bytes fetched per CU for these fetch-bound kernels
template <int RPB>
__global__ __launch_bounds__(16 * RPB) void k_critic_tail(CriticTail p) {
  __shared__ __attribute__((aligned(16))) float cs[3 * RPB * HID];
  __shared__ float sc[RPB][2];
  const int t = threadIdx.x, row = t >> 4, sub = t & 15, net = blockIdx.y;
  const int b = blockIdx.x * RPB + row, bc = min(b, p.B - 1);
  const bool valid = b < p.B;
  const float* Pn = p.P + net * p.p_ns;
  STAMP(0);
  // loads first
  const Row16 zt0 = row_ld(p.z2t + (long)bc * HID, sub), zt1 = row_ld(p.z2t + ((long)p.B + bc) * HID, sub);
  const Row16 zo = row_ld(p.z2 + ((long)net * p.B + bc) * HID, sub);
  const Row16 wt0 = row_ld(p.PT + p.L.Wh, sub), wt1 = row_ld(p.PT + p.p_ns + p.L.Wh, sub), wo = row_ld(Pn + p.L.Wh, sub);
  Row16 gt0, bt0, gt1, bt1, go, bo;
  if (p.ln) {
    gt0 = row_ld(p.PT + p.L.g2, sub); bt0 = row_ld(p.PT + p.L.be2, sub);
    gt1 = row_ld(p.PT + p.p_ns + p.L.g2, sub); bt1 = row_ld(p.PT + p.p_ns + p.L.be2, sub);
    go = row_ld(Pn + p.L.g2, sub); bo = row_ld(Pn + p.L.be2, sub);
  }
  const float bht0 = p.PT[p.L.bh], bht1 = p.PT[p.p_ns + p.L.bh], bho = Pn[p.L.bh];
  const float rw = p.rew[bc], dn = p.done[bc];
  const float alpha = p.sac ? expf(*p.log_alpha) : 0.f;
  const float lpn = p.sac ? p.logp_next[bc] : 0.f;
  STAMP(1);
  Row16 xh, y, h; float rs;
  ln_fwd(zt0, gt0, bt0, p.ln, xh, y, rs);
#pragma unroll
  for (int q = 0; q < 4; ++q) h.v[q] = relu4(y.v[q]);
  const float qt0 = row16_sum(row_dot(h, wt0)) + bht0;
  ln_fwd(zt1, gt1, bt1, p.ln, xh, y, rs);
#pragma unroll
  for (int q = 0; q < 4; ++q) h.v[q] = relu4(y.v[q]);
  const float qt1 = row16_sum(row_dot(h, wt1)) + bht1;
  const float qmin = fminf(qt0, qt1);
  float qp = p.bcq ? 0.75f * qmin + 0.25f * fmaxf(qt0, qt1) : qmin;
  if (p.sac) qp -= alpha * lpn;
  const float yv = rw + (1.0f - dn) * p.gamma * qp;
  float rstd;
  ln_fwd(zo, go, bo, p.ln, xh, y, rstd);
#pragma unroll
  for (int q = 0; q < 4; ++q) h.v[q] = relu4(y.v[q]);
  const float qv = row16_sum(row_dot(h, wo)) + bho;
  const float err = valid ? qv - yv : 0.f;
  const float dq = 2.0f * err / (float)p.B;
  Row16 dy, vals[3];
#pragma unroll
  for (int q = 0; q < 4; ++q) dy.v[q] = gate4(wo.v[q] * dq, y.v[q]);
  const Row16 dz = ln_bwd(dy, xh, rstd, go, p.ln);
#pragma unroll
  for (int q = 0; q < 4; ++q) { vals[0].v[q] = dy.v[q] * xh.v[q]; vals[1].v[q] = dy.v[q]; vals[2].v[q] = h.v[q] * dq; }
  if (valid) {
    row_st(p.dz2 + ((long)net * p.B + b) * HID, sub, dz);
    if (sub == 0) {
      p.q[(long)net * p.B + b] = qv;
      if (net == 0) { p.qt[b] = qt0; p.qt[p.B + b] = qt1; p.y[b] = yv; }
    }
  }
  if (sub == 0) { sc[row][0] = dq; sc[row][1] = err * err; }
  STAMP(2);
  const long blk = (long)net * p.pstride + blockIdx.x;
  block_colsum<RPB>(cs, vals, 3, row, sub, p.part + blk * NSLOT * HID);   // (has the barrier that publishes sc)
  if (t < 2) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RPB; ++i) s += sc[i][t];
    p.part_s[blk * 2 + t] = s;
  }
  STAMP(3);
}

// k_critic_tail + k_nn in ONE launch (B < 1024): a block = 16 batch rows x 16 NT columns of dh1 = dz2 W2.  Every block redoes the
// tail arithmetic of its 16 rows (256 threads = 16 rows x 16 threads, exactly k_critic_tail<16>'s layout: three LayerNorms, three
// head dot products, the Bellman target, the LayerNorm backward -- a few hundred cycles), parks the dz2 rows in LDS as the A operand
// and multiplies them with its W2 column tiles (requested with the first batch of loads, as k_nn does).  What the tail kernel
// stored once per row block is spread over the row block's column-tile blocks: each stores ITS 16 NT columns of dz2 and of the
// three column partials; the scalars (q, target q, y, loss partials) come from the column-tile-0 block.  One graph node and one
// cold round trip fewer per critic update; the price is 57 KB of tail operands per block instead of 16 KB of dz2 rows.
// fold != 0: dX gets dy1 = relu'(.) dh1 instead of dh1, and ps[net][row][tile] / [16 + tile] the row's two LayerNorm-backward
// sums over this 16-column tile (TnProb::fold): h1 / xh1 = layer 1's activations and xhat, g1 = its gamma.
struct NnFold { int fold, ln; const float* h1; const float* xh1; int g1_off; float* ps; float* gsnap; };   // gsnap[net][HID] = gamma1 (TnProb::f_g)
struct CtailNn { CriticTail c; const float* Wt; int ldw; float* dX; int xr; NnFold f; };
// (wave-uniform caller; o[i] = dh1 of row 4 (lane >> 4) + i of the 16-row block, column `col`)
__device__ __forceinline__ void nn_fold_store(const NnFold& f, const float* o, const float* hv, const float* xv, float gcol,
                                              float* x, float* ps, float* gsnap, int row0, int B, int col, int lane) {
  if (f.fold && f.ln && row0 == 0 && lane < 16) gsnap[col] = gcol;     // (here, not at the load: a store there waits for every load)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int orow = row0 + 4 * (lane >> 4) + i;
    const float dy = f.fold ? (hv[i] > 0.f ? o[i] : 0.f) : o[i];
    if (orow < B) x[(long)orow * HID + col] = dy;
    if (f.fold && f.ln) {
      const float dxh = dy * gcol;
      const float s1 = row16_sum(dxh), s2 = row16_sum(dxh * xv[i]);
      if ((lane & 15) == 0 && orow < B) { ps[(long)orow * PS_W + (col >> 4)] = s1; ps[(long)orow * PS_W + 16 + (col >> 4)] = s2; }
    }
  }
}
template <int NT>
__global__ __launch_bounds__(256) void k_ctail_nn(CtailNn a) {
  const CriticTail& p = a.c;
  constexpr int CB = 16 * NT;                               // columns per block
  __shared__ __attribute__((aligned(16))) float Dz[16 * AS];
  __shared__ __attribute__((aligned(16))) float cs[3 * 16 * CB];
  __shared__ __attribute__((aligned(16))) float red[NT * 4 * 64 * 4];
  __shared__ float sc[16][2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, row = t >> 4, sub = t & 15, net = blockIdx.z;
  const int r = lane & 15, kq = lane >> 4;
  int tm, tk;
  xcd_tile(blockIdx.x, (p.B + 15) >> 4, HID / CB, a.xr, tm, tk);
  const int b = tm * 16 + row, bc = min(b, p.B - 1), c_lo = tk * CB;
  const bool valid = b < p.B;
  const float* Pn = p.P + net * p.p_ns;
  // ---- every load: the tail's rows and parameters, then this block's W2 fragments (k_nn's B operand)
  const Row16 zt0 = row_ld(p.z2t + (long)bc * HID, sub), zt1 = row_ld(p.z2t + ((long)p.B + bc) * HID, sub);
  const Row16 zo = row_ld(p.z2 + ((long)net * p.B + bc) * HID, sub);
  const Row16 wt0 = row_ld(p.PT + p.L.Wh, sub), wt1 = row_ld(p.PT + p.p_ns + p.L.Wh, sub), wo = row_ld(Pn + p.L.Wh, sub);
  Row16 gt0, bt0, gt1, bt1, go, bo;
  if (p.ln) {
    gt0 = row_ld(p.PT + p.L.g2, sub); bt0 = row_ld(p.PT + p.L.be2, sub);
    gt1 = row_ld(p.PT + p.p_ns + p.L.g2, sub); bt1 = row_ld(p.PT + p.p_ns + p.L.be2, sub);
    go = row_ld(Pn + p.L.g2, sub); bo = row_ld(Pn + p.L.be2, sub);
  }
  const float bht0 = p.PT[p.L.bh], bht1 = p.PT[p.p_ns + p.L.bh], bho = Pn[p.L.bh];
  const float rw = p.rew[bc], dn = p.done[bc];
  const float alpha = p.sac ? expf(*p.log_alpha) : 0.f;
  const float lpn = p.sac ? p.logp_next[bc] : 0.f;
  const int nb = 64 * wave + 4 * kq;
  float4 bv[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float* Wc = a.Wt + net * p.p_ns + (long)nb * a.ldw + c_lo + 16 * nt + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* w = Wc + (long)(16 * c) * a.ldw;
      bv[nt][c] = make_float4(w[0], w[a.ldw], w[2 * (long)a.ldw], w[3 * (long)a.ldw]);
    }
  }
  float fh[4] = {1.f, 1.f, 1.f, 1.f}, fx[4] = {0.f, 0.f, 0.f, 0.f}, fg = 1.f;      // the epilogue's layer-1 operands (waves < NT)
  if (a.f.fold && wave < NT) {
    const int col = c_lo + 16 * wave + (lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long o = ((long)net * p.B + min(tm * 16 + 4 * (lane >> 4) + i, p.B - 1)) * HID + col;
      fh[i] = a.f.h1[o]; if (a.f.ln) fx[i] = a.f.xh1[o];
    }
    if (a.f.ln) fg = Pn[a.f.g1_off + col];
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- the tail (k_critic_tail's arithmetic, agents/agent.py:208-233)
  Row16 xh, y, h; float rs;
  ln_fwd(zt0, gt0, bt0, p.ln, xh, y, rs);
#pragma unroll
  for (int q = 0; q < 4; ++q) h.v[q] = relu4(y.v[q]);
  const float qt0 = row16_sum(row_dot(h, wt0)) + bht0;
  ln_fwd(zt1, gt1, bt1, p.ln, xh, y, rs);
#pragma unroll
  for (int q = 0; q < 4; ++q) h.v[q] = relu4(y.v[q]);
  const float qt1 = row16_sum(row_dot(h, wt1)) + bht1;
  const float qmin = fminf(qt0, qt1);
  float qp = p.bcq ? 0.75f * qmin + 0.25f * fmaxf(qt0, qt1) : qmin;
  if (p.sac) qp -= alpha * lpn;
  const float yv = rw + (1.0f - dn) * p.gamma * qp;
  float rstd;
  ln_fwd(zo, go, bo, p.ln, xh, y, rstd);
#pragma unroll
  for (int q = 0; q < 4; ++q) h.v[q] = relu4(y.v[q]);
  const float qv = row16_sum(row_dot(h, wo)) + bho;
  const float err = valid ? qv - yv : 0.f;
  const float dq = 2.0f * err / (float)p.B;
  Row16 dy, vals[3];
#pragma unroll
  for (int q = 0; q < 4; ++q) dy.v[q] = gate4(wo.v[q] * dq, y.v[q]);
  const Row16 dz = ln_bwd(dy, xh, rstd, go, p.ln);          // (rows beyond the batch: err = 0 -> all zeros)
#pragma unroll
  for (int q = 0; q < 4; ++q) { vals[0].v[q] = dy.v[q] * xh.v[q]; vals[1].v[q] = dy.v[q]; vals[2].v[q] = h.v[q] * dq; }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int c = 0; c < 4; ++c) { PIN(bv[nt][c].x); PIN(bv[nt][c].y); PIN(bv[nt][c].z); PIN(bv[nt][c].w); }
  // ---- stores of this block's share: its columns of dz2 and of the three column partials; scalars by the column-tile-0 block
  row_st(Dz + row * AS, sub, dz);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = 4 * sub + 64 * q;                      // (the thread's float4 of chunk q lies inside one 16-column tile)
    if (col >= c_lo && col < c_lo + CB) {
      if (valid) st4(p.dz2 + ((long)net * p.B + b) * HID + col, dz.v[q]);
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) st4(cs + (sl * 16 + row) * CB + (col - c_lo), vals[sl].v[q]);
    }
  }
  if (tk == 0 && sub == 0) {
    sc[row][0] = dq; sc[row][1] = err * err;
    if (valid) {
      p.q[(long)net * p.B + b] = qv;
      if (net == 0) { p.qt[b] = qt0; p.qt[p.B + b] = qt1; p.y[b] = yv; }
    }
  }
  __syncthreads();
  const long blk = (long)net * p.pstride + tm;
  if (t < 3 * CB) {
    const int sl = t / CB, c = t - sl * CB;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += cs[(sl * 16 + i) * CB + c];
    p.part[(blk * NSLOT + sl) * HID + c_lo + c] = sum;
  }
  if (tk == 0 && t >= 128 && t < 130) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += sc[i][t - 128];
    p.part_s[blk * 2 + (t - 128)] = sum;
  }
  // ---- dh1 tile(s): rows from LDS, W2 fragments from registers, the 256-long reduction split over the 4 waves (k_nn)
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 av = ld4(Dz + r * AS + nb + 16 * c);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { MFMA4(acc[nt], av, bv[nt][c]); }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) st4(red + ((nt * 4 + wave) * 64 + lane) * 4, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
  __syncthreads();
  if (wave < NT) {
    const float* rr = red + (wave * 4 * 64 + lane) * 4;
    const float4 x0 = ld4(rr), x1 = ld4(rr + 256), x2 = ld4(rr + 512), x3 = ld4(rr + 768);
    const float o[4] = {(x0.x + x1.x) + (x2.x + x3.x), (x0.y + x1.y) + (x2.y + x3.y), (x0.z + x1.z) + (x2.z + x3.z), (x0.w + x1.w) + (x2.w + x3.w)};
    const int col = c_lo + 16 * wave + (lane & 15);
    nn_fold_store(a.f, o, fh, fx, fg, a.dX + (long)net * p.B * HID, a.f.ps + (long)net * p.B * PS_W, a.f.gsnap + net * HID, tm * 16, p.B, col, lane);
  }
}

struct ActorQTail {
  const float* z2c;                      // [nq][B][HID]
  const float* P; long p_ns; NetLayout L;// online critics
  const float* logp; const float* log_alpha;
  int B, ln, sac;
  float* q; float* dz2;                  // [nq][B], [nq][B][HID]
  float* part_s;                         // [blocks][2]: loss partial in [.][1]
};

template <int RPB>
__global__ __launch_bounds__(16 * RPB) void k_actorq_tail(ActorQTail p) {
  __shared__ float sc[RPB];
  const int t = threadIdx.x, row = t >> 4, sub = t & 15;
  const int b = blockIdx.x * RPB + row, bc = min(b, p.B - 1);
  const bool valid = b < p.B;
  const int nq = p.sac ? 2 : 1;
  Row16 z[2], w[2], g[2], be[2], xh[2], y[2];
  float rstd[2], qv[2] = {0.f, 0.f}, bh[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (i < nq) {
      const float* Pn = p.P + i * p.p_ns;
      z[i] = row_ld(p.z2c + ((long)i * p.B + bc) * HID, sub);
      w[i] = row_ld(Pn + p.L.Wh, sub);
      if (p.ln) { g[i] = row_ld(Pn + p.L.g2, sub); be[i] = row_ld(Pn + p.L.be2, sub); }
      bh[i] = Pn[p.L.bh];
    }
  const float alpha = p.sac ? expf(*p.log_alpha) : 0.f;
  const float lpv = p.sac ? p.logp[bc] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (i < nq) {
      ln_fwd(z[i], g[i], be[i], p.ln, xh[i], y[i], rstd[i]);
      Row16 h;
#pragma unroll
      for (int q = 0; q < 4; ++q) h.v[q] = relu4(y[i].v[q]);
      qv[i] = row16_sum(row_dot(h, w[i])) + bh[i];
    }
  const bool first = p.sac ? (qv[0] <= qv[1]) : true;
  const float loss = p.sac ? (alpha * lpv - (first ? qv[0] : qv[1])) : -qv[0];
  const float invB = 1.0f / (float)p.B;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (i < nq) {
      const float dq = ((i == 0) == first) ? -invB : 0.f;
      Row16 dy;
#pragma unroll
      for (int q = 0; q < 4; ++q) dy.v[q] = gate4(w[i].v[q] * dq, y[i].v[q]);
      const Row16 dz = ln_bwd(dy, xh[i], rstd[i], g[i], p.ln);
      if (valid) {
        row_st(p.dz2 + ((long)i * p.B + b) * HID, sub, dz);
        if (sub == 0) p.q[(long)i * p.B + b] = qv[i];
      }
    }
  if (sub == 0) sc[row] = valid ? loss : 0.f;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RPB; ++i) s += sc[i];
    p.part_s[blockIdx.x * 2 + 1] = s;
  }
}

// k_actorq_tail + k_nn in ONE launch (B < 1024), as k_ctail_nn: a block = 16 batch rows x 16 NT columns of dh1_i = dz2_i W2_i of
// critic i = blockIdx.z.  Every block evaluates BOTH critics' heads on its rows (the min decides which critic a row's gradient
// flows through), keeps its own critic's dz2 rows in LDS as the A operand; nothing downstream reads dz2 itself in the actor
// update, so it is not stored.  Q(s, pi(s)) and the loss partial come from the column-tile-0 blocks.
// dQ/da without a launch of its own (B < 1024, narrow heads, ac_dim <= 7).  What k_ln_bwd<16>'s dQ/da form computes per batch row --
//   dA[j] = sum_c dz1[c] W1[c][o + j],  dz1 = rstd (dxh - m1 - xhat m2),  dxh = relu'(.) dh1 gamma1,  m1 = mean_c dxh,  m2 = mean_c dxh xhat
// -- is linear in sums over the 256 columns:  dA[j] = rstd (P[j] - m1 S[j] - m2 X[j])  with  P[j] = sum_c dxh[c] W1a[c][j],
// X[j] = sum_c xhat[c] W1a[c][j],  S[j] = sum_c W1a[c][j]  (W1a = the action columns of W1).  Every column-tile block of k_qtail_nn
// holds a [16 rows] x [16 NT columns] piece of dh1 when its GEMM is done: wave 0 gates and scales it, turns it round through LDS and
// gets the block's share of P (and of m1: a column of ones behind W1a) and of X with 4 NT MFMAs each, m2's share with DPP row sums, and
// stores them as ONE partial per (row, block): ps[net][row][tile][pqw] = P[0 .. a-1], sum dxh, X[0 .. a-1], sum dxh xhat.  The
// row-block-0 blocks also store their share of S.  k_headbwd_nn sums the tiles in a fixed order and finishes dA.  dh1 itself is not
// stored any more (nothing else reads it in the actor update).
struct QaFold {
  int on, ln, a, pqw, ntile;             // pqw = floats per partial (8: a <= 3, else 16); ntile = column-tile blocks per row block
  const float* h1; const float* xh1;     // [nq][B][HID]: the critics' layer-1 activations / xhat at (s, pi(s))
  int g1_off, w1_off, ld1, k_off;        // gamma1 and W1 inside a critic's parameter block, W1's row stride, first action column (= ob_dim)
  float* ps;                             // [nq][B][ntile][pqw]
  float* S;                              // [nq][ntile][8]
};
struct QtailNn { ActorQTail c; const float* Wt; int ldw; float* dX; int xr; QaFold qa; };
template <int NT>
__global__ __launch_bounds__(256) void k_qtail_nn(QtailNn a) {
  const ActorQTail& p = a.c;
  constexpr int CB = 16 * NT;
  __shared__ __attribute__((aligned(16))) float Dz[16 * AS];
  __shared__ __attribute__((aligned(16))) float red[NT * 4 * 64 * 4];
  __shared__ float sc[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, row = t >> 4, sub = t & 15, net = blockIdx.z;
  const int r = lane & 15, kq = lane >> 4;
  int tm, tk;
  xcd_tile(blockIdx.x, (p.B + 15) >> 4, HID / CB, a.xr, tm, tk);
  const int b = tm * 16 + row, bc = min(b, p.B - 1), c_lo = tk * CB;
  const bool valid = b < p.B;
  const int nq = p.sac ? 2 : 1;
  Row16 z[2], w[2], g[2], be[2], xh[2], y[2];
  float rstd[2], qv[2] = {0.f, 0.f}, bh[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (i < nq) {
      const float* Pi = p.P + i * p.p_ns;
      z[i] = row_ld(p.z2c + ((long)i * p.B + bc) * HID, sub);
      w[i] = row_ld(Pi + p.L.Wh, sub);
      if (p.ln) { g[i] = row_ld(Pi + p.L.g2, sub); be[i] = row_ld(Pi + p.L.be2, sub); }
      bh[i] = Pi[p.L.bh];
    }
  const float alpha = p.sac ? expf(*p.log_alpha) : 0.f;
  const float lpv = p.sac ? p.logp[bc] : 0.f;
  const int nb = 64 * wave + 4 * kq;
  float4 bv[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float* Wc = a.Wt + net * p.p_ns + (long)nb * a.ldw + c_lo + 16 * nt + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* wp = Wc + (long)(16 * c) * a.ldw;
      bv[nt][c] = make_float4(wp[0], wp[a.ldw], wp[2 * (long)a.ldw], wp[3 * (long)a.ldw]);
    }
  }
  // the dQ/da epilogue's operands (wave 0: every 16-column tile of the block)
  float qh[NT][4], qx[NT][4], qg[NT];
  float4 qw[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    qg[nt] = 1.f; qw[nt] = f4(0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) { qh[nt][i] = 1.f; qx[nt][i] = 0.f; }
  }
  if (a.qa.on && wave == 0) {
    const float* Pn = p.P + net * p.p_ns;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = c_lo + 16 * nt + (lane & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long o = ((long)net * p.B + min(tm * 16 + 4 * (lane >> 4) + i, p.B - 1)) * HID + col;
        qh[nt][i] = a.qa.h1[o]; if (a.qa.ln) qx[nt][i] = a.qa.xh1[o];
      }
      if (a.qa.ln) qg[nt] = Pn[a.qa.g1_off + col];
      // B fragments: lane (j = r, kq) holds W1[c_lo + 16 nt + 4 kq + m][k_off + j], m = 0 .. 3 (column clamped; masked once the loads are in)
      const float* wr = Pn + a.qa.w1_off + (long)(c_lo + 16 * nt + 4 * kq) * a.qa.ld1 + a.qa.k_off + min(r, a.qa.a - 1);
      qw[nt] = make_float4(wr[0], wr[a.qa.ld1], wr[2 * (long)a.qa.ld1], wr[3 * (long)a.qa.ld1]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (i < nq) {
      ln_fwd(z[i], g[i], be[i], p.ln, xh[i], y[i], rstd[i]);
      Row16 h;
#pragma unroll
      for (int q = 0; q < 4; ++q) h.v[q] = relu4(y[i].v[q]);
      qv[i] = row16_sum(row_dot(h, w[i])) + bh[i];
    }
  const bool first = p.sac ? (qv[0] <= qv[1]) : true;
  const float loss = p.sac ? (alpha * lpv - (first ? qv[0] : qv[1])) : -qv[0];
  const float invB = 1.0f / (float)p.B;
  // this block's critic: dq = -1/B where it is the minimum, else 0 (agents/agent.py:272-281); rows beyond the batch: 0
  const bool mine = valid && ((net == 0) == first);
  const float dq = mine ? -invB : 0.f;
  Row16 dy;
  const Row16& wn = net == 0 ? w[0] : w[1];
  const Row16& yn = net == 0 ? y[0] : y[1];
  const Row16& xn = net == 0 ? xh[0] : xh[1];
  const Row16& gn = net == 0 ? g[0] : g[1];
#pragma unroll
  for (int q = 0; q < 4; ++q) dy.v[q] = gate4(wn.v[q] * dq, yn.v[q]);
  const Row16 dz = ln_bwd(dy, xn, net == 0 ? rstd[0] : rstd[1], gn, p.ln);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int c = 0; c < 4; ++c) { PIN(bv[nt][c].x); PIN(bv[nt][c].y); PIN(bv[nt][c].z); PIN(bv[nt][c].w); }
  row_st(Dz + row * AS, sub, dz);
  if (tk == 0 && sub == 0) {
    if (valid) p.q[(long)net * p.B + b] = net == 0 ? qv[0] : qv[1];
    if (net == 0) sc[row] = valid ? loss : 0.f;
  }
  __syncthreads();
  if (tk == 0 && net == 0 && t == 0) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += sc[i];
    p.part_s[tm * 2 + 1] = sum;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 av = ld4(Dz + r * AS + nb + 16 * c);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { MFMA4(acc[nt], av, bv[nt][c]); }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) st4(red + ((nt * 4 + wave) * 64 + lane) * 4, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
  __syncthreads();
  if (a.qa.on) {                                            // (uniform) the dQ/da partials instead of dh1: see QaFold
    if (wave != 0) return;
    const int na = a.qa.a;
    f32x4 aP = {0.f, 0.f, 0.f, 0.f}, aX = {0.f, 0.f, 0.f, 0.f}, aS = {0.f, 0.f, 0.f, 0.f};
    float s2r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float* T = red + nt * 4 * 64 * 4;                     // this tile's slab of `red`: read first, then scratch for the two transposes
      const float4 x0 = ld4(T + lane * 4), x1 = ld4(T + 256 + lane * 4), x2 = ld4(T + 512 + lane * 4), x3 = ld4(T + 768 + lane * 4);
      const float o[4] = {(x0.x + x1.x) + (x2.x + x3.x), (x0.y + x1.y) + (x2.y + x3.y), (x0.z + x1.z) + (x2.z + x3.z), (x0.w + x1.w) + (x2.w + x3.w)};
      float4 bw = qw[nt];                                   // lanes j < a: W1a; j == a: ones (the row sum of dxh); else nothing
      if (r >= na) bw = f4(r == na ? 1.f : 0.f);
      __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the slab has been read before it is overwritten
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dy = qh[nt][i] > 0.f ? o[i] : 0.f;
        const float dxh = a.qa.ln ? dy * qg[nt] : dy;
        if (a.qa.ln) s2r[i] += row16_sum(dxh * qx[nt][i]);
        T[(4 * (lane >> 4) + i) * 20 + (lane & 15)] = dxh;
        T[320 + (4 * (lane >> 4) + i) * 20 + (lane & 15)] = qx[nt][i];
      }
      const float4 fa = ld4(T + r * 20 + 4 * kq), fx4 = ld4(T + 320 + r * 20 + 4 * kq);
      MFMA4(aP, fa, bw);
      if (a.qa.ln) { MFMA4(aX, fx4, bw); }
      if (a.qa.ln && tm == 0) { const float4 one = f4(1.f); MFMA4(aS, one, bw); }
    }
    // lane (j = lane & 15, row group lane >> 4): aP[i] = P[j] (j < a) or sum dxh (j == a) of row 4 (lane >> 4) + i, aX[i] = X[j]
    float* ps = a.qa.ps + (((long)net * p.B) * a.qa.ntile + tk) * a.qa.pqw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int orow = tm * 16 + 4 * (lane >> 4) + i;
      if (orow >= p.B) continue;
      float* d = ps + (long)orow * a.qa.ntile * a.qa.pqw;
      if (r <= na) d[r] = aP[i];
      if (a.qa.ln && r < na) d[na + 1 + r] = aX[i];
      if (a.qa.ln && r == 0) d[2 * na + 1] = s2r[i];
    }
    if (a.qa.ln && tm == 0 && lane < 16 && r < na) a.qa.S[((long)net * a.qa.ntile + tk) * 8 + r] = aS[0];
    return;
  }
  if (wave < NT) {
    const float* rr = red + (wave * 4 * 64 + lane) * 4;
    const float4 x0 = ld4(rr), x1 = ld4(rr + 256), x2 = ld4(rr + 512), x3 = ld4(rr + 768);
    const float o[4] = {(x0.x + x1.x) + (x2.x + x3.x), (x0.y + x1.y) + (x2.y + x3.y), (x0.z + x1.z) + (x2.z + x3.z), (x0.w + x1.w) + (x2.w + x3.w)};
    const int col = c_lo + 16 * wave + (lane & 15);
    float* x = a.dX + (long)net * p.B * HID;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int orow = tm * 16 + 4 * (lane >> 4) + i;
      if (orow < p.B) x[(long)orow * HID + col] = o[i];
    }
  }
}

struct LnBwd {               // dz = LNbwd(relu'(.) * dh) for hidden layer 1; optional (dgamma, dbeta) partials
  const float* dh; const float* xh; const float* h; const float* rstd;   // [nets][B][HID] x3, [nets][B]
  const float* gamma; long p_ns;
  int B, ln, want_part;
  float* dz;
  float* part; int pstride;
  // optional fused input gradient of a slice of layer 1 (RPB == 16 only): dA[net][b][0:na] = dz[b][:] . W1[:, k_off : k_off + na]
  // (the dQ/da path of the actor update, agents/agent.py:272-283; na <= 32, not combined with want_part)
  const float* W1; int ldw1, k_off, na; float* dA; int ldA;
};

template <int RPB>
__global__ __launch_bounds__(16 * RPB) void k_ln_bwd(LnBwd p) {
  __shared__ __attribute__((aligned(16))) float cs[2 * RPB * HID];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, row = t >> 4, sub = t & 15, net = blockIdx.y;
  const int r = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x * RPB + row, bc = min(b, p.B - 1);
  const bool valid = b < p.B;
  const long ro = ((long)net * p.B + bc) * HID;
  const Row16 dh = row_ld(p.dh + ro, sub), hh = row_ld(p.h + ro, sub), xh = row_ld(p.xh + ro, sub);
  Row16 g;
  if (p.ln) g = row_ld(p.gamma + net * p.p_ns, sub);
  const float rstd = p.ln ? p.rstd[(long)net * p.B + bc] : 1.f;
  // B operand of the fused slice product, requested up front: W1[n = 16 (4 wave + ci) + 4 kq + jj][k_off + 16 tt + r]
  const int T = (RPB == 16 && p.dA) ? (p.na + 15) >> 4 : 0;
  float4 wf[2][4];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      wf[tt][ci] = f4(0.f);
      if (T > 0) {                         // uniform; column clamped, lanes beyond na are zeroed once the loads are in
        const float* w = p.W1 + net * p.p_ns + (long)(16 * (4 * wave + ci) + 4 * kq) * p.ldw1 + p.k_off + min(16 * tt + r, p.na - 1);
        wf[tt][ci] = make_float4(w[0], w[p.ldw1], w[2 * (long)p.ldw1], w[3 * (long)p.ldw1]);
      }
    }
  Row16 dy, vals[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) { dy.v[q] = gate4(dh.v[q], hh.v[q]); if (!valid) dy.v[q] = f4(0.f); }
  const Row16 dz = ln_bwd(dy, xh, rstd, g, p.ln);
  if (RPB == 16) {                       // the slice-product operands land before the first store goes out (see PIN)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        PIN(wf[tt][ci].x); PIN(wf[tt][ci].y); PIN(wf[tt][ci].z); PIN(wf[tt][ci].w);
        if (16 * tt + r >= p.na) wf[tt][ci] = f4(0.f);
      }
  }
  if (valid) row_st(p.dz + ((long)net * p.B + b) * HID, sub, dz);
  if (p.want_part) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { vals[0].v[q] = dy.v[q] * xh.v[q]; vals[1].v[q] = dy.v[q]; }
    block_colsum<RPB>(cs, vals, 2, row, sub, p.part + (((long)net * p.pstride + blockIdx.x) * NSLOT + 3) * HID);   // slots 3, 4
  } else if (RPB == 16 && p.dA) {
    float* Dz = cs;                      // [16][AS]
    float* Up = cs + 16 * AS;            // [4 waves][16 rows][32]
    row_st(Dz + row * AS, sub, dz);
    __syncthreads();
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const float4 av = ld4(Dz + r * AS + (4 * wave + ci) * 16 + 4 * kq);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        if (tt < T) { MFMA4(acc[tt], av, wf[tt][ci]); }
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
      if (tt < T)
#pragma unroll
        for (int i = 0; i < 4; ++i) Up[(wave * 16 + 4 * kq + i) * 32 + tt * 16 + r] = acc[tt][i];
    __syncthreads();
    for (int j = sub; j < p.na; j += 16) {
      const float* u = Up + row * 32 + j;
      if (valid) p.dA[((long)net * p.B + b) * p.ldA + j] = (u[0] + u[512]) + (u[1024] + u[1536]);
    }
  }
}

struct ActorHeadBwd {
  const float* dA; long dA_ns; int ldA; int nq;   // [nq][B][ldA] grads wrt the action from each critic
  const float* tg; int a4; const float* eps; const float* log_alpha; const float* scale;
  const float* P; NetLayout L;
  const float* xh2; const float* rstd2; const float* h2;
  int B, a, ln, sac;
  float* du; int ldu;                    // [B][ldu] grad wrt head outputs
  float* dz2;                            // [B][HID]
  float* part;                           // [blocks][NSLOT][HID]
};

__global__ __launch_bounds__(256) void k_actor_head_bwd(ActorHeadBwd p) {
  __shared__ __attribute__((aligned(16))) float cs[2 * 16 * HID];   // also holds dh2 [16][AS] before the column sums
  __shared__ __attribute__((aligned(16))) float Du[16 * 68];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row = t >> 4, sub = t & 15, r = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x * 16 + row, bc = min(b, p.B - 1);
  const bool valid = b < p.B;
  const int nh = p.L.nh, C = (nh + 15) >> 4;      // k chunks of the head-backward product (<= 4)
  const float* Wh = p.P + p.L.Wh;
  // loads first
  const long ro = (long)bc * HID;
  const Row16 hh = row_ld(p.h2 + ro, sub), xh = row_ld(p.xh2 + ro, sub);
  Row16 g;
  if (p.ln) g = row_ld(p.P + p.L.g2, sub);
  const float rstd = p.ln ? p.rstd2[bc] : 1.f;
  float4 wf[4][4];                       // B operand of dh2 = du Wh: Wh[k = 16c + 4kq + jj][n = (4*tt + wave)*16 + r]
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = (4 * tt + wave) * 16 + r, k = 16 * c + 4 * kq;
      wf[tt][c] = f4(0.f);
      if (c < C) {
        if (k < nh) wf[tt][c].x = Wh[(long)k * HID + n];
        if (k + 1 < nh) wf[tt][c].y = Wh[(long)(k + 1) * HID + n];
        if (k + 2 < nh) wf[tt][c].z = Wh[(long)(k + 2) * HID + n];
        if (k + 3 < nh) wf[tt][c].w = Wh[(long)(k + 3) * HID + n];
      }
    }
  // operands of this thread's first head element (j = sub), requested with the rest (see k_actor_tail)
  const int j0 = min(sub, p.a - 1);
  const float* tgr = p.tg + (long)bc * 4 * p.a4;
  float la = p.sac ? *p.log_alpha : 0.f;
  float o_dA = p.dA[(long)bc * p.ldA + j0], o_dA1 = p.nq == 2 ? p.dA[p.dA_ns + (long)bc * p.ldA + j0] : 0.f;
  float o_sc = p.scale[j0], o_t0 = tgr[j0];
  float o_t1 = 0.f, o_t2 = 0.f, o_e = 0.f;
  if (p.sac) { o_t1 = tgr[p.a4 + j0]; o_t2 = tgr[2 * p.a4 + j0]; o_e = p.eps[(long)bc * p.a + j0]; }
  for (int j = sub; j < 64; j += 16) Du[row * 68 + j] = 0.f;
  PIN(la); PIN(o_dA); PIN(o_dA1); PIN(o_sc); PIN(o_t0); PIN(o_t1); PIN(o_t2); PIN(o_e);
  __syncthreads();
  const float dlogp = p.sac ? expf(la) / (float)p.B : 0.f;
  auto element = [&](int j, float dAj, float sc, float t0, float t1, float t2, float e) {
    float g_mean, g_raw = 0.f;
    if (p.sac) {
      const float tt = t0, sd = t1, yt = t2;
      const float omy2 = 1.0f - yt * yt;
      const float g0 = dAj * sc * omy2 + dlogp * (2.0f * sc * yt * omy2) / (sc * omy2 + 1e-6f);
      g_mean = g0;
      g_raw = (g0 * e * sd - dlogp) * 3.5f * (1.0f - tt * tt);
    } else {
      const float th = t0;
      g_mean = dAj * sc * (1.0f - th * th);
    }
    if (!valid) { g_mean = 0.f; g_raw = 0.f; }
    Du[row * 68 + j] = g_mean;
    if (p.sac) Du[row * 68 + p.a + j] = g_raw;
    if (valid) {
      p.du[(long)b * p.ldu + j] = g_mean;
      if (p.sac) p.du[(long)b * p.ldu + p.a + j] = g_raw;
    }
  };
  if (sub < p.a) element(sub, o_dA + o_dA1, o_sc, o_t0, o_t1, o_t2, o_e);
  for (int j = sub + 16; j < p.a; j += 16) {            // ac_dim > 16 only
    float dAj = p.dA[(long)bc * p.ldA + j];
    if (p.nq == 2) dAj += p.dA[p.dA_ns + (long)bc * p.ldA + j];
    element(j, dAj, p.scale[j], tgr[j], p.sac ? tgr[p.a4 + j] : 0.f, p.sac ? tgr[2 * p.a4 + j] : 0.f, p.sac ? p.eps[(long)bc * p.a + j] : 0.f);
  }
  __syncthreads();
  // dh2[16][256] = du[16][nh] Wh[nh][256]; wave w owns column tiles w, w+4, w+8, w+12
  float* DH = cs;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) { const float4 av = ld4(Du + r * 68 + 16 * c + 4 * kq); MFMA4(acc, av, wf[tt][c]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) DH[(4 * kq + i) * AS + (4 * tt + wave) * 16 + r] = acc[i];
  }
  __syncthreads();
  const Row16 dh = row_ld(DH + row * AS, sub);
  __syncthreads();                       // DH is reused by the column sums below
  Row16 dy, vals[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) dy.v[q] = gate4(dh.v[q], hh.v[q]);
  const Row16 dz = ln_bwd(dy, xh, rstd, g, p.ln);
  if (valid) row_st(p.dz2 + (long)b * HID, sub, dz);
#pragma unroll
  for (int q = 0; q < 4; ++q) { vals[0].v[q] = dy.v[q] * xh.v[q]; vals[1].v[q] = dy.v[q]; }
  block_colsum<16>(cs, vals, 2, row, sub, p.part + (long)blockIdx.x * NSLOT * HID);
}

// Narrow heads (nh <= 8), as k_actor_tail_s: a row lives in ONE wave (RPB rows x 16 threads = 64 threads per block).  Lane n of
// a row computes head-output gradient du[n] itself (lanes a .. 2a-1 redo their element's arithmetic for the log-std half instead
// of fetching it across lanes), du[n] reaches the row's other lanes by DPP row broadcast, dh2 = du Wh is nh FMAs per column on
// the vector ALU, and the dgamma / dbeta partials are summed over the block's rows through wave-local LDS: no MFMA, no block barrier.
template <int RPB>
__global__ __launch_bounds__(16 * RPB) void k_actor_head_bwd_s(ActorHeadBwd p) {
  __shared__ __attribute__((aligned(16))) float cs[2 * RPB * HID];
  const int t = threadIdx.x, row = t >> 4, sub = t & 15;
  const int b = blockIdx.x * RPB + row, bc = min(b, p.B - 1);
  const bool valid = b < p.B;
  const int nh = p.L.nh;                         // <= 8, and a <= 8
  const float* Wh = p.P + p.L.Wh;
  // loads first
  const long ro = (long)bc * HID;
  const Row16 hh = row_ld(p.h2 + ro, sub), xh = row_ld(p.xh2 + ro, sub);
  Row16 g;
  if (p.ln) g = row_ld(p.P + p.L.g2, sub);
  const float rstd = p.ln ? p.rstd2[bc] : 1.f;
  Row16 w[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) w[n] = row_ld(Wh + (long)min(n, nh - 1) * HID, sub);   // unconditional, row clamped
  // lane `sub` < nh owns head output n = sub: action element j = n (mean half) or n - a (log-std half, SAC)
  const bool second = sub >= p.a;
  const int j0 = min(second ? sub - p.a : sub, p.a - 1);
  const float* tgr = p.tg + (long)bc * 4 * p.a4;
  float la = p.sac ? *p.log_alpha : 0.f;
  float o_dA = p.dA[(long)bc * p.ldA + j0], o_dA1 = p.nq == 2 ? p.dA[p.dA_ns + (long)bc * p.ldA + j0] : 0.f;
  float o_sc = p.scale[j0], o_t0 = tgr[j0];
  float o_t1 = 0.f, o_t2 = 0.f, o_e = 0.f;
  if (p.sac) { o_t1 = tgr[p.a4 + j0]; o_t2 = tgr[2 * p.a4 + j0]; o_e = p.eps[(long)bc * p.a + j0]; }
  PIN(la); PIN(o_dA); PIN(o_dA1); PIN(o_sc); PIN(o_t0); PIN(o_t1); PIN(o_t2); PIN(o_e);
  const float dlogp = p.sac ? expf(la) / (float)p.B : 0.f;
  float d;                                        // du[sub]
  {
    const float dAj = o_dA + o_dA1, sc = o_sc;
    float g_mean, g_raw = 0.f;
    if (p.sac) {
      const float tt = o_t0, sd = o_t1, yt = o_t2;
      const float omy2 = 1.0f - yt * yt;
      const float g0 = dAj * sc * omy2 + dlogp * (2.0f * sc * yt * omy2) / (sc * omy2 + 1e-6f);
      g_mean = g0;
      g_raw = (g0 * o_e * sd - dlogp) * 3.5f * (1.0f - tt * tt);
    } else {
      g_mean = dAj * sc * (1.0f - o_t0 * o_t0);
    }
    d = second ? g_raw : g_mean;
    if (!valid || sub >= nh) d = 0.f;
  }
  // dh2[c] = sum_n du[n] Wh[n][c]
  Row16 dh;
#pragma unroll
  for (int q = 0; q < 4; ++q) dh.v[q] = f4(0.f);
  auto add_n = [&](float dn, const Row16& wn) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dh.v[q] = dh.v[q] + wn.v[q] * dn;
  };
  row_pin(w[0]); add_n(dpp_mov<0x150>(d), w[0]);           // row_newbcast:n = lane n of every 16-lane row to all of its lanes
  row_pin(w[1]); add_n(dpp_mov<0x151>(d), w[1]);
  row_pin(w[2]); add_n(dpp_mov<0x152>(d), w[2]);
  row_pin(w[3]); add_n(dpp_mov<0x153>(d), w[3]);
  row_pin(w[4]); add_n(dpp_mov<0x154>(d), w[4]);
  row_pin(w[5]); add_n(dpp_mov<0x155>(d), w[5]);
  row_pin(w[6]); add_n(dpp_mov<0x156>(d), w[6]);
  row_pin(w[7]); add_n(dpp_mov<0x157>(d), w[7]);           // (lanes >= nh hold d = 0: their clamped weight rows add nothing)
  if (valid && sub < nh) p.du[(long)b * p.ldu + sub] = d;
  Row16 dy, vals[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) dy.v[q] = gate4(dh.v[q], hh.v[q]);
  const Row16 dz = ln_bwd(dy, xh, rstd, g, p.ln);
  if (valid) row_st(p.dz2 + (long)b * HID, sub, dz);
#pragma unroll
  for (int q = 0; q < 4; ++q) { vals[0].v[q] = dy.v[q] * xh.v[q]; vals[1].v[q] = dy.v[q]; }
  block_colsum<RPB>(cs, vals, 2, row, sub, p.part + (long)blockIdx.x * NSLOT * HID);
}

// k_actor_head_bwd_s + k_nn in ONE launch (narrow heads, B < 1024), as k_ctail_nn: a block = 16 batch rows x 16 columns of the
// actor's dh1 = dz2 W2.  Every block redoes the head backward of its 16 rows (k_actor_head_bwd_s's arithmetic: lane n of a row
// computes du[n], DPP row broadcast, nh FMAs per column, LayerNorm backward), keeps the dz2 rows in LDS as the A operand, and
// stores ITS 16 columns of dz2 and of the two column partials; du comes from the column-tile-0 block.
// qa.on: dA is not read but finished here from k_qtail_nn's partials (QaFold): rstd = the critics' layer-1 rstd [nq][B]
struct QaIn { int on, ln, a, pqw, ntile; const float* ps; const float* S; const float* rstd; float* dA; };
struct HeadBwdNn { ActorHeadBwd c; const float* Wt; int ldw; float* dX; int xr; NnFold f; QaIn qa; };
__global__ __launch_bounds__(256) void k_headbwd_nn(HeadBwdNn a) {
  const ActorHeadBwd& p = a.c;
  constexpr int CB = 16;
  __shared__ __attribute__((aligned(16))) float Dz[16 * AS];
  __shared__ __attribute__((aligned(16))) float cs[2 * 16 * CB];
  __shared__ __attribute__((aligned(16))) float red[4 * 64 * 4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, row = t >> 4, sub = t & 15;
  const int r = lane & 15, kq = lane >> 4;
  int tm, tk;
  xcd_tile(blockIdx.x, (p.B + 15) >> 4, HID / CB, a.xr, tm, tk);
  const int b = tm * 16 + row, bc = min(b, p.B - 1), c_lo = tk * CB;
  const bool valid = b < p.B;
  const int nh = p.L.nh;                         // <= 8, and a <= 8
  const float* Wh = p.P + p.L.Wh;
  // ---- loads first (k_actor_head_bwd_s), then this block's W2 fragments (k_nn's B operand)
  const long ro = (long)bc * HID;
  const Row16 hh = row_ld(p.h2 + ro, sub), xh = row_ld(p.xh2 + ro, sub);
  Row16 g;
  if (p.ln) g = row_ld(p.P + p.L.g2, sub);
  const float rstd = p.ln ? p.rstd2[bc] : 1.f;
  Row16 w[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) w[n] = row_ld(Wh + (long)min(n, nh - 1) * HID, sub);   // unconditional, row clamped
  const bool second = sub >= p.a;
  const int j0 = min(second ? sub - p.a : sub, p.a - 1);
  const float* tgr = p.tg + (long)bc * 4 * p.a4;
  float la = p.sac ? *p.log_alpha : 0.f;
  float o_dA = 0.f, o_dA1 = 0.f;
  // (qa.on) the row's partials: thread `sub` takes float4 #sub, #sub + 16, ... of the row's ntile x pqw floats of each critic
  float4 qv[2] = {f4(0.f), f4(0.f)}, qs[2] = {f4(0.f), f4(0.f)};
  float qrs[2] = {1.f, 1.f};
  if (a.qa.on) {
    const int n4 = a.qa.ntile * a.qa.pqw / 4, s4 = a.qa.ntile * 2;     // float4s per row / of a critic's S partials
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < p.nq) {
        const float* pr = a.qa.ps + ((long)q * p.B + bc) * (long)n4 * 4;
        float4 v0 = ld4(pr + 4 * sub), v1 = f4(0.f), v2 = f4(0.f), v3 = f4(0.f);
        if (n4 > 16) v1 = ld4(pr + 4 * (sub + 16));
        if (n4 > 32) { v2 = ld4(pr + 4 * (sub + 32)); v3 = ld4(pr + 4 * (sub + 48)); }
        qv[q] = (v0 + v1) + (v2 + v3);
        const float* sr = a.qa.S + (long)q * s4 * 4;
        float4 t0 = ld4(sr + 4 * sub), t1 = f4(0.f);
        if (s4 > 16) t1 = ld4(sr + 4 * (sub + 16));
        qs[q] = t0 + t1;
        if (a.qa.ln) qrs[q] = a.qa.rstd[(long)q * p.B + bc];
      }
  } else {
    o_dA = p.dA[(long)bc * p.ldA + j0]; o_dA1 = p.nq == 2 ? p.dA[p.dA_ns + (long)bc * p.ldA + j0] : 0.f;
  }
  float o_sc = p.scale[j0], o_t0 = tgr[j0];
  float o_t1 = 0.f, o_t2 = 0.f, o_e = 0.f;
  if (p.sac) { o_t1 = tgr[p.a4 + j0]; o_t2 = tgr[2 * p.a4 + j0]; o_e = p.eps[(long)bc * p.a + j0]; }
  const int nb = 64 * wave + 4 * kq;
  float4 bv[4];
  {
    const float* Wc = a.Wt + (long)nb * a.ldw + c_lo + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* wp = Wc + (long)(16 * c) * a.ldw;
      bv[c] = make_float4(wp[0], wp[a.ldw], wp[2 * (long)a.ldw], wp[3 * (long)a.ldw]);
    }
  }
  float fh[4] = {1.f, 1.f, 1.f, 1.f}, fx[4] = {0.f, 0.f, 0.f, 0.f}, fg = 1.f;      // the epilogue's layer-1 operands (wave 0)
  if (a.f.fold && wave == 0) {
    const int col = c_lo + (lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long o = (long)min(tm * 16 + 4 * (lane >> 4) + i, p.B - 1) * HID + col;
      fh[i] = a.f.h1[o]; if (a.f.ln) fx[i] = a.f.xh1[o];
    }
    if (a.f.ln) fg = p.P[a.f.g1_off + col];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (a.qa.on) {
    // sum the tiles: the float4s a thread took all hold the same part of a partial (16 is a multiple of pqw / 4), so lanes with equal
    // sub mod (pqw / 4) add up (DPP row rotations); lane g < pqw / 4 then parks part g of the row's sums in LDS for the whole row to read
    float* Vs = red + (row * 2) * 24;                       // [16 rows][2 critics][16 sums | 8 S]
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < p.nq) {
        float v[8] = {qv[q].x, qv[q].y, qv[q].z, qv[q].w, qs[q].x, qs[q].y, qs[q].z, qs[q].w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (k >= 4 || a.qa.pqw == 8) v[k] += dpp_mov<0x122>(v[k]);     // row_ror:2 (S partials are 8 floats: parts 0, 1)
          v[k] += dpp_mov<0x124>(v[k]); v[k] += dpp_mov<0x128>(v[k]);       // row_ror:4, row_ror:8
        }
        if (sub < a.qa.pqw / 4) st4(Vs + q * 24 + 4 * sub, make_float4(v[0], v[1], v[2], v[3]));
        if (sub < 2) st4(Vs + q * 24 + 16 + 4 * sub, make_float4(v[4], v[5], v[6], v[7]));
      }
    const int na = a.qa.a;
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < p.nq) {
        const float* V = Vs + q * 24;
        const float P = V[j0];
        float dq = P;
        if (a.qa.ln) dq = (P - V[na] * (1.0f / HID) * V[16 + j0] - V[2 * na + 1] * (1.0f / HID) * V[na + 1 + j0]) * qrs[q];
        if (a.qa.dA && tk == 0 && valid && sub < na) a.qa.dA[((long)q * p.B + b) * p.ldA + sub] = dq;
        tot += dq;
      }
    o_dA = tot; o_dA1 = 0.f;
  }
  PIN(la); PIN(o_dA); PIN(o_dA1); PIN(o_sc); PIN(o_t0); PIN(o_t1); PIN(o_t2); PIN(o_e);
  const float dlogp = p.sac ? expf(la) / (float)p.B : 0.f;
  float d;                                        // du[sub]
  {
    const float dAj = o_dA + o_dA1, sc = o_sc;
    float g_mean, g_raw = 0.f;
    if (p.sac) {
      const float tt = o_t0, sd = o_t1, yt = o_t2;
      const float omy2 = 1.0f - yt * yt;
      const float g0 = dAj * sc * omy2 + dlogp * (2.0f * sc * yt * omy2) / (sc * omy2 + 1e-6f);
      g_mean = g0;
      g_raw = (g0 * o_e * sd - dlogp) * 3.5f * (1.0f - tt * tt);
    } else {
      g_mean = dAj * sc * (1.0f - o_t0 * o_t0);
    }
    d = second ? g_raw : g_mean;
    if (!valid || sub >= nh) d = 0.f;
  }
  Row16 dh;
#pragma unroll
  for (int q = 0; q < 4; ++q) dh.v[q] = f4(0.f);
  auto add_n = [&](float dn, const Row16& wn) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dh.v[q] = dh.v[q] + wn.v[q] * dn;
  };
  row_pin(w[0]); add_n(dpp_mov<0x150>(d), w[0]);           // row_newbcast:n = lane n of every 16-lane row to all of its lanes
  row_pin(w[1]); add_n(dpp_mov<0x151>(d), w[1]);
  row_pin(w[2]); add_n(dpp_mov<0x152>(d), w[2]);
  row_pin(w[3]); add_n(dpp_mov<0x153>(d), w[3]);
  row_pin(w[4]); add_n(dpp_mov<0x154>(d), w[4]);
  row_pin(w[5]); add_n(dpp_mov<0x155>(d), w[5]);
  row_pin(w[6]); add_n(dpp_mov<0x156>(d), w[6]);
  row_pin(w[7]); add_n(dpp_mov<0x157>(d), w[7]);
#pragma unroll
  for (int c = 0; c < 4; ++c) { PIN(bv[c].x); PIN(bv[c].y); PIN(bv[c].z); PIN(bv[c].w); }
  if (tk == 0 && valid && sub < nh) p.du[(long)b * p.ldu + sub] = d;
  Row16 dy, vals[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) dy.v[q] = gate4(dh.v[q], hh.v[q]);
  const Row16 dz = ln_bwd(dy, xh, rstd, g, p.ln);           // (rows beyond the batch: d = 0 -> all zeros)
#pragma unroll
  for (int q = 0; q < 4; ++q) { vals[0].v[q] = dy.v[q] * xh.v[q]; vals[1].v[q] = dy.v[q]; }
  row_st(Dz + row * AS, sub, dz);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = 4 * sub + 64 * q;
    if (col >= c_lo && col < c_lo + CB) {
      if (valid) st4(p.dz2 + (long)b * HID + col, dz.v[q]);
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) st4(cs + (sl * 16 + row) * CB + (col - c_lo), vals[sl].v[q]);
    }
  }
  __syncthreads();
  if (t < 2 * CB) {
    const int sl = t / CB, c = t - sl * CB;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += cs[(sl * 16 + i) * CB + c];
    p.part[((long)tm * NSLOT + sl) * HID + c_lo + c] = sum;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) { const float4 av = ld4(Dz + r * AS + nb + 16 * c); MFMA4(acc, av, bv[c]); }
  acc = splitk_reduce(red, acc, wave, lane);
  if (wave == 0) {
    const float o[4] = {acc[0], acc[1], acc[2], acc[3]};
    nn_fold_store(a.f, o, fh, fx, fg, a.dX, a.f.ps, a.f.gsnap, tm * 16, p.B, c_lo + (lane & 15), lane);
  }
}

// ------------------------------------------------------------------------------------------------ optimiser
struct AdamArgs {
  float* p; const float* g; float* m; float* v; long n;   // n multiple of 4
  const float* adam; float b1, b2, eps;  // adam: (lr / (1 - b1^t), sqrt(1 - b2^t))
  const float* gscale;                   // optional gradient scale (clip_grad_norm_)
  float* targ; float tau;                // optional fused Polyak of the same arena
  const float* loss_part; int loss_n; int loss_stride; int loss_off; float loss_scale; float* loss_dst;
  int* tick;
};

__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
  const float step = a.adam[0], sq2 = a.adam[1];   // published by the first kernel of the update (adam_tick)
  const float gs = a.gscale ? *a.gscale : 1.0f;
  const float omb1 = 1.0f - a.b1, omb2 = 1.0f - a.b2;
  // block 0's extras: operands requested before the element loop, written after it (loads first, stores last)
  const bool extras = blockIdx.x == 0 && threadIdx.x < 64;
  float loss_acc = 0.f; int tick_v = 0;
  if (extras) {
    if (a.loss_dst) for (int i = threadIdx.x; i < a.loss_n; i += 64) loss_acc += a.loss_part[(long)i * a.loss_stride + a.loss_off];
    if (threadIdx.x == 0 && a.tick) tick_v = *a.tick;
  }
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += (long)gridDim.x * 1024) {
    float4 g = ld4(a.g + i) * gs, m = ld4(a.m + i), v = ld4(a.v + i), p = ld4(a.p + i);
    float4 tt = f4(0.f);
    if (a.targ) tt = ld4(a.targ + i);
    m = m + (g - m) * omb1;
    v = v * a.b2 + g * g * omb2;
    p.x -= step * (m.x / (sqrtf(v.x) / sq2 + a.eps)); p.y -= step * (m.y / (sqrtf(v.y) / sq2 + a.eps));
    p.z -= step * (m.z / (sqrtf(v.z) / sq2 + a.eps)); p.w -= step * (m.w / (sqrtf(v.w) / sq2 + a.eps));
    st4(a.m + i, m); st4(a.v + i, v); st4(a.p + i, p);
    if (a.targ) st4(a.targ + i, tt + (p - tt) * a.tau);
  }
  if (extras) {
    if (a.loss_dst) {
      const float s = wave_sum(loss_acc);
      if (threadIdx.x == 0) *a.loss_dst = s * a.loss_scale;
    }
    if (threadIdx.x == 0 && a.tick) *a.tick = tick_v + 1;
  }
}

__global__ __launch_bounds__(256) void k_polyak(PolyakArgs a) { polyak_body(a, blockIdx.x, gridDim.x); }

struct NormArgs { const float* g; long n; float clip; float* gscale; };
__global__ __launch_bounds__(1024) void k_gradnorm(NormArgs a) {   // single block; clip_grad_norm_ (agents/agent.py:284-285)
  __shared__ float red[16];
  float s = 0.f;
  for (long i = threadIdx.x * 4; i < a.n; i += 4096) { const float4 g = ld4(a.g + i); s += dot4(g, g); }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    *a.gscale = fminf(1.0f, a.clip / (sqrtf(t) + 1e-6f));
  }
}
