// Device code of the SAC/TD3 update engine for gfx950 (MI355X, CDNA4).  wave = 64 lanes throughout.
//
// Kernel families
//   k_gather        replay ring -> batch slot (float4 record chunks, Philox index draw fused)
//   k_gemm_nt/nn/tn fp32 MFMA (v_mfma_f32_16x16x4_f32) GEMMs, one 16x16 output tile per wave, operands
//                   streamed straight from L2 (everything on this path is L2-resident), with the
//                   LayerNorm+ReLU of the producing layer fused into the A-operand load (nt) and the
//                   bias / LN-affine gradient column sums fused into the weight-gradient GEMM (tn)
//   k_actor_tail    LN+ReLU -> head GEMV -> tanh-Gaussian sample + log-prob (SAC) / tanh policy (TD3)
//   k_critic_tail   twin target Q -> min/mix -> entropy -> Bellman target -> twin MSE -> dQ -> head bwd -> LN bwd
//   k_actorq_tail   twin Q(s, pi(s)) -> min -> actor loss -> dQ routing -> head bwd -> LN bwd (dX only)
//   k_actor_head_bwd  d(action), d(logp) -> tanh-Gaussian bwd -> head bwd -> LN bwd
//   k_ln_bwd        LN+ReLU backward of a hidden layer (row per wave)
//   k_adam / k_polyak / k_alpha_step / k_gradnorm   flat optimiser kernels
// Math follows oracle/manual_grads.py (which is checked against autograd) line by line.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "philox.h"

#define HID 256
#define LN_EPS 1e-5f
#define NSLOT 4            // column-partial slots per block: 0 dgamma, 1 dbeta, 2 dWhead (critic), 3 spare

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DevCtl {
  unsigned long long seed;
  int sample_ctr;     // bumped after every index draw
  int noise_ctr;      // bumped at the end of every update graph
  int predict_ctr;
  int t_q, t_a, t_l;  // Adam step counts (critics, actor, log_alpha)
  int rb_len, rb_cursor;
  int inject_idx;
  int inject_eps[8];
  int pad_;
  float metrics[8];
};

struct NetLayout {   // float offsets inside one net's parameter block (all multiples of 4)
  int K, ld1, nh;
  int W1, b1, g1, be1, W2, b2, g2, be2, Wh, bh;
  int size;
};

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
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

// standard normal for element e of (ctr, site): 4 normals per Philox block via two Box-Muller pairs
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned ctr, unsigned site, unsigned e) {
  const Philox4 r = philox4x32_10(ctr, 0u, SACTD3_STREAM_NOISE + site, e >> 2, (uint32_t)seed, (uint32_t)(seed >> 32));
  const unsigned k = e & 3u;
  const float u1 = philox_u01(r.v[k & 2u]), u2 = philox_u01(r.v[(k & 2u) + 1u]);
  const float rad = sqrtf(-2.0f * logf(u1));
  const float th = 6.283185307179586f * u2;
  return (k & 1u) ? rad * sinf(th) : rad * cosf(th);
}

// One row of 256 held as 4 consecutive columns per lane: LayerNorm (+affine) then the pre-ReLU value.
// ln == 0: y = z, xhat/rstd unused.
__device__ __forceinline__ void ln_row(float4 z, const float* gamma, const float* beta, int lane, int ln,
                                       float4& xhat, float4& y, float& mean, float& rstd) {
  if (ln) {
    mean = wave_sum(sum4(z)) * (1.0f / HID);
    const float4 d = z - f4(mean);
    const float var = wave_sum(dot4(d, d)) * (1.0f / HID);
    rstd = 1.0f / sqrtf(var + LN_EPS);
    xhat = d * rstd;
    y = xhat * ld4(gamma + 4 * lane) + ld4(beta + 4 * lane);
  } else {
    mean = 0.f; rstd = 1.f; xhat = z; y = z;
  }
}

// LN backward for one row given dy (grad at the LN output, ReLU gate already applied)
__device__ __forceinline__ float4 ln_row_bwd(float4 dy, float4 xhat, float rstd, const float* gamma, int lane, int ln) {
  if (!ln) return dy;
  const float4 dxh = dy * ld4(gamma + 4 * lane);
  const float m1 = wave_sum(sum4(dxh)) * (1.0f / HID);
  const float m2 = wave_sum(dot4(dxh, xhat)) * (1.0f / HID);
  return (dxh - f4(m1) - xhat * m2) * rstd;
}

// Combine the 4 waves' per-lane float4 column accumulators of a block and store one partial row per slot.
// red: __shared__ float[4 waves][nslots][HID].  Every thread must call this (it contains barriers).
__device__ __forceinline__ void block_store_partials(float* red, const float4* acc, int nslots, float* dst /*[NSLOT][HID] of this block*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = 0; s < nslots; ++s) st4(red + (wave * NSLOT + s) * HID + 4 * lane, acc[s]);
  __syncthreads();
  const int t = threadIdx.x;  // 256 threads = 256 columns
  for (int s = 0; s < nslots; ++s) {
    const float v = ((red[(0 * NSLOT + s) * HID + t] + red[(1 * NSLOT + s) * HID + t]) +
                     (red[(2 * NSLOT + s) * HID + t] + red[(3 * NSLOT + s) * HID + t]));
    dst[s * HID + t] = v;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ replay
struct GatherArgs {
  const float4* ring; int rec4;           // record = rec4 float4 chunks: [s|a : cx][s' : cn][r,d,0,0][pad]
  int cx, cn;                             // chunks of the [s|a] field and of the s' field
  const DevCtl* ctl; int* idx;
  float4* X; float4* Xp; float4* Xn;      // row stride = cx chunks
  float* rew; float* done;
  int B; int len_override;                // len_override >= 0: use it instead of ctl->rb_len (staged batches)
};

__global__ __launch_bounds__(256) void k_gather(GatherArgs p) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = (int)(g / p.rec4), c = (int)(g % p.rec4);
  if (b >= p.B) return;
  int id;
  if (p.ctl->inject_idx) {
    id = p.idx[b];
  } else {
    const int len = p.len_override >= 0 ? p.len_override : p.ctl->rb_len;
    id = (int)philox_index(p.ctl->seed, (unsigned)p.ctl->sample_ctr, (unsigned)b, (unsigned)len);
    if (c == 0) p.idx[b] = id;
  }
  if (c > p.cx + p.cn) return;            // trailing pad chunk(s)
  const float4 v = p.ring[(long)id * p.rec4 + c];
  if (c < p.cx) {
    p.X[(long)b * p.cx + c] = v;
    p.Xp[(long)b * p.cx + c] = v;
  } else if (c < p.cx + p.cn) {
    p.Xn[(long)b * p.cx + (c - p.cx)] = v;
  } else {
    p.rew[b] = v.x;
    p.done[b] = v.y;
  }
}

__global__ void k_tick(int* a, int* b) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { if (a) *a += 1; if (b) *b += 1; }
}
__global__ void k_set_int2(int* dst, int v0, int v1) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { dst[0] = v0; dst[1] = v1; }
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
// D: col j = l&15, row i = 4*(l>>4) + reg.  The k slot of a lane is free to name ANY k as long as A and B
// agree, so lane-group kq takes 4 CONSECUTIVE k's (one float4 load) and feeds them to 4 successive MFMAs.
#define MFMA4(acc, a, b)                                                   \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).x, (b).x, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).y, (b).y, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).z, (b).z, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).w, (b).w, acc, 0, 0, 0);

struct GemmNT {              // Y[M,N] = pro(A)[M,K] * W[N,K]^T + bias
  const float* A; int lda; long a_ns;
  const float* Wt; int ldw;             // W = Wt + net * p_ns
  const float* bias;                    // may be null
  const float* gamma; const float* beta;// LN affine applied to A (PRO == 1)
  long p_ns;                            // net stride of every parameter pointer
  float* Y; int ldy; long y_ns;
  float* Hout; long h_ns;               // PRO != 0, optional: store pro(A) rows (tile_n == 0 waves)
  float* stats; long st_ns;             // PRO == 1, optional: (mean, rstd) per row
  int M, N, K;
  int* tick0; int* tick1;               // optional counters bumped by (block 0, thread 0, net 0)
};

// PRO: 0 none (any K), 1 LayerNorm+ReLU (K == 256), 2 ReLU (K == 256)
template <int PRO>
__global__ __launch_bounds__(256) void k_gemm_nt(GemmNT p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, net = blockIdx.z;
  if (blockIdx.x == 0 && threadIdx.x == 0 && net == 0) {
    if (p.tick0) *p.tick0 += 1;
    if (p.tick1) *p.tick1 += 1;
  }
  const int tiles_n = (p.N + 15) >> 4, tiles_m = (p.M + 15) >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= tiles_m * tiles_n) return;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int r = lane & 15, kq = lane >> 4;
  const int m = min(tm * 16 + r, p.M - 1), n = min(tn * 16 + r, p.N - 1);
  const float* Arow = p.A + net * p.a_ns + (long)m * p.lda;
  const float* Wrow = p.Wt + net * p.p_ns + (long)n * p.ldw;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (PRO == 0) {
#pragma unroll 4
    for (int k0 = 0; k0 < p.K; k0 += 16) {
      const int k = k0 + 4 * kq;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), w = a;
      if (k < p.K) {   // lda, ldw >= round4(K): the float4 is in-bounds; zero what lies beyond K
        a = ld4(Arow + k); w = ld4(Wrow + k);
        if (k + 3 >= p.K) {
          if (k + 1 >= p.K) { a.y = 0.f; w.y = 0.f; }
          if (k + 2 >= p.K) { a.z = 0.f; w.z = 0.f; }
          a.w = 0.f; w.w = 0.f;
        }
      }
      MFMA4(acc, a, w);
    }
  } else {
    float4 av[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) av[c] = ld4(Arow + c * 16 + 4 * kq);
    float mean = 0.f, rstd = 1.f;
    if (PRO == 1) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) s += sum4(av[c]);
      s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
      mean = s * (1.0f / HID);
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) { const float4 d = av[c] - f4(mean); q += dot4(d, d); }
      q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
      rstd = 1.0f / sqrtf(q * (1.0f / HID) + LN_EPS);
      const float* g = p.gamma + net * p.p_ns; const float* be = p.beta + net * p.p_ns;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        av[c] = relu4((av[c] - f4(mean)) * rstd * ld4(g + c * 16 + 4 * kq) + ld4(be + c * 16 + 4 * kq));
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) av[c] = relu4(av[c]);
    }
    if (tn == 0 && tm * 16 + r < p.M) {
      if (p.Hout) {
        float* h = p.Hout + net * p.h_ns + (long)m * HID;
#pragma unroll
        for (int c = 0; c < 16; ++c) st4(h + c * 16 + 4 * kq, av[c]);
      }
      if (PRO == 1 && p.stats && kq == 0) {
        float* st = p.stats + net * p.st_ns + 2 * (long)m;
        st[0] = mean; st[1] = rstd;
      }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4 w = ld4(Wrow + c * 16 + 4 * kq);
      MFMA4(acc, av[c], w);
    }
  }
  const int col = tn * 16 + (lane & 15);
  if (col < p.N) {
    const float bv = p.bias ? p.bias[net * p.p_ns + col] : 0.f;
    float* y = p.Y + net * p.y_ns;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = tm * 16 + 4 * (lane >> 4) + i;
      if (row < p.M) y[(long)row * p.ldy + col] = acc[i] + bv;
    }
  }
}

struct GemmNN {              // dX[M,Kout] = dY[M,256] * W[256, k_off : k_off+Kout]
  const float* dY; long dy_ns;          // row stride HID
  const float* Wt; int ldw; long p_ns; int k_off;
  float* dX; int ldx; long dx_ns;
  int M, Kout;
};

__global__ __launch_bounds__(256) void k_gemm_nn(GemmNN p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, net = blockIdx.z;
  const int tiles_k = (p.Kout + 15) >> 4, tiles_m = (p.M + 15) >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= tiles_m * tiles_k) return;
  const int tm = tile / tiles_k, tk = tile % tiles_k;
  const int r = lane & 15, kq = lane >> 4;
  const int m = min(tm * 16 + r, p.M - 1);
  const int kc = p.k_off + min(tk * 16 + r, p.Kout - 1);
  const float* Drow = p.dY + net * p.dy_ns + (long)m * HID;
  const float* Wc = p.Wt + net * p.p_ns + kc;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int c = 0; c < HID / 16; ++c) {
    const int n = c * 16 + 4 * kq;
    const float4 a = ld4(Drow + n);
    float4 b;
    b.x = Wc[(long)(n + 0) * p.ldw]; b.y = Wc[(long)(n + 1) * p.ldw];
    b.z = Wc[(long)(n + 2) * p.ldw]; b.w = Wc[(long)(n + 3) * p.ldw];
    MFMA4(acc, a, b);
  }
  const int col = tk * 16 + (lane & 15);
  if (col < p.Kout) {
    float* x = p.dX + net * p.dx_ns;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = tm * 16 + 4 * (lane >> 4) + i;
      if (row < p.M) x[(long)row * p.ldx + col] = acc[i];
    }
  }
}

struct GemmTN {              // dW[N, ldw] = dY[M,N]^T * X[M,K]   (columns K..ldw-1 written as 0)
  const float* dY; int ldy; long dy_ns; int N;
  const float* X; int ldx; long x_ns; int K;
  float* dW; int ldw;
  float* dbias;                          // optional: dbias[n] = sum_m dY[m][n]
  long g_ns;                             // net stride of every gradient pointer
  int M;
  // column partials produced by a row kernel: part[net][blk][NSLOT][HID] -> fin_dst[e][n] = sum_blk part[..][fin_slot[e]][n]
  const float* part; int nblk; int nfin; int fin_slot[3]; float* fin_dst[3];
  const float* part_s; float* fin_s;     // optional scalar: fin_s[0] = sum_blk part_s[net][blk][0]
};

__global__ __launch_bounds__(256) void k_gemm_tn(GemmTN p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, net = blockIdx.z;
  const int tiles_k = (p.ldw + 15) >> 4, tiles_n = (p.N + 15) >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= tiles_n * tiles_k) return;
  const int tn = tile / tiles_k, tk = tile % tiles_k;
  const int r = lane & 15, kq = lane >> 4;
  const int nA = min(tn * 16 + r, p.N - 1);
  const int kB = tk * 16 + r;
  const bool kval = kB < p.K;
  const float* Dc = p.dY + net * p.dy_ns + nA;
  const float* Xc = p.X + net * p.x_ns + min(kB, p.K - 1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;
  const int chunks = (p.M + 15) >> 4;
#pragma unroll 2
  for (int c = 0; c < chunks; ++c) {
    float4 a, b;
    float* ap = &a.x; float* bp = &b.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int mm = c * 16 + 4 * kq + j;
      const bool v = mm < p.M;
      const int mc = v ? mm : p.M - 1;
      const float av = Dc[(long)mc * p.ldy], bv = Xc[(long)mc * p.ldx];
      ap[j] = v ? av : 0.f;
      bp[j] = (v && kval) ? bv : 0.f;
    }
    asum += (a.x + a.y) + (a.z + a.w);
    MFMA4(acc, a, b);
  }
  if (tk == 0) {
    const int n = tn * 16 + r;
    if (p.dbias) {
      float s = asum;
      s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
      if (kq == 0 && n < p.N) p.dbias[net * p.g_ns + n] = s;
    }
    for (int e = 0; e < p.nfin; ++e) {
      float s = 0.f;
      if (n < p.N)
        for (int blk = kq; blk < p.nblk; blk += 4)
          s += p.part[(((long)net * p.nblk + blk) * NSLOT + p.fin_slot[e]) * HID + n];
      s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
      if (kq == 0 && n < p.N) p.fin_dst[e][net * p.g_ns + n] = s;
    }
    if (p.fin_s && tn == 0) {
      float s = 0.f;
      for (int blk = lane; blk < p.nblk; blk += 64) s += p.part_s[((long)net * p.nblk + blk) * 2];
      s = wave_sum(s);
      if (lane == 0) p.fin_s[net * p.g_ns] = s;
    }
  }
  const int col = tk * 16 + (lane & 15);
  if (col < p.ldw) {
    float* w = p.dW + net * p.g_ns;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = tn * 16 + 4 * (lane >> 4) + i;
      if (row < p.N) w[(long)row * p.ldw + col] = (col < p.K) ? acc[i] : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------ row kernels
// Layout of every row kernel: 256 threads = 4 waves, one row of 256 per wave at a time (lane holds columns
// 4*lane .. 4*lane+3), `rpw` rows per wave.  Row index = (block * 4 + wave) * rpw + it.

struct ActorTail {
  const float* z2;                       // [B][HID] pre-LN output of hidden layer 2
  const float* P; NetLayout L;           // actor parameter block (online or target)
  int B, o, a, ln, sac, mode, train, rpw;
  // mode: SAC 0 = sample, 1 = mode (tanh(mean));  TD3 0 = policy, 1 = target smoothing, 2 = explore
  const DevCtl* ctl; const int* ctr; int site_buf; unsigned site_code;
  float* eps;                            // [B][a] draws used (written in native mode, read when injected)
  const float* scale; const float* bias; const float* min_ac; const float* max_ac;
  float* dst; int ldd; int dst_off;      // action -> dst[b * ldd + dst_off + j]
  float* logp;                           // [B] (SAC)
  float* h2; float* st2; float* tg;      // train stores: h2 [B][HID], st2 [B][2], tg [B][4][a4] (t, std, y, -)
  int a4;
  float td3_std, td3_c, noise_std;
};

__global__ __launch_bounds__(256) void k_actor_tail(ActorTail p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* g2 = p.P + p.L.g2; const float* be2 = p.P + p.L.be2;
  const float* Wh = p.P + p.L.Wh; const float* bh = p.P + p.L.bh;
  const int nh = p.L.nh;
  for (int it = 0; it < p.rpw; ++it) {
    const int b = (blockIdx.x * 4 + wave) * p.rpw + it;
    if (b >= p.B) break;
    float4 xhat, y; float mean, rstd;
    ln_row(ld4(p.z2 + (long)b * HID + 4 * lane), g2, be2, lane, p.ln, xhat, y, mean, rstd);
    const float4 h = relu4(y);
    if (p.train) {
      st4(p.h2 + (long)b * HID + 4 * lane, h);
      if (lane == 0) { p.st2[2 * b] = mean; p.st2[2 * b + 1] = rstd; }
    }
    // head GEMV: output j lands in lane (j mod a) as u0 (j < a) or u1 (j >= a)
    float u0 = 0.f, u1 = 0.f;
    for (int j = 0; j < nh; ++j) {
      const float s = wave_sum(dot4(h, ld4(Wh + (long)j * HID + 4 * lane))) + bh[j];
      if (j < p.a) { if (lane == j) u0 = s; } else { if (lane == j - p.a) u1 = s; }
    }
    float lp = 0.f;
    if (lane < p.a) {
      const int j = lane;
      float e = 0.f;
      const bool need_eps = p.sac ? (p.mode == 0) : (p.mode != 0);
      if (need_eps) {
        if (p.ctl->inject_eps[p.site_buf]) e = p.eps[(long)b * p.a + j];
        else { e = philox_normal(p.ctl->seed, (unsigned)*p.ctr, p.site_code, (unsigned)(b * p.a + j)); p.eps[(long)b * p.a + j] = e; }
      }
      const float sc = p.scale[j], bi = p.bias[j];
      float act;
      if (p.sac) {
        const float t = tanhf(u1);
        const float log_std = -5.0f + 3.5f * (t + 1.0f);
        const float sd = expf(log_std);
        const float x = u0 + e * sd;
        const float yt = tanhf(x);
        act = yt * sc + bi;
        const float dx = x - u0;
        lp = -(dx * dx) / (2.0f * sd * sd) - logf(sd) - 0.9189385332046727f;
        lp -= logf(sc * (1.0f - yt * yt) + 1e-6f);
        if (p.mode == 1) act = tanhf(u0) * sc + bi;
        if (p.train) {
          float* tg = p.tg + (long)b * 4 * p.a4;
          tg[j] = t; tg[p.a4 + j] = sd; tg[2 * p.a4 + j] = yt;
        }
      } else {
        const float th = tanhf(u0);
        act = th * sc + bi;
        if (p.mode == 1) {
          const float nz = fminf(fmaxf(e * p.td3_std, -p.td3_c), p.td3_c);
          act = fminf(fmaxf(act + nz, p.min_ac[j]), p.max_ac[j]);
        } else if (p.mode == 2) {
          act = act + e * (sc * p.noise_std);
        }
        if (p.train) p.tg[(long)b * 4 * p.a4 + j] = th;
      }
      p.dst[(long)b * p.ldd + p.dst_off + j] = act;
    }
    if (p.sac && p.logp) {
      const float s = wave_sum(lp);
      if (lane == 0) p.logp[b] = s;
    }
  }
}

struct CriticTail {
  const float* z2t; const float* z2;     // [2][B][HID] target / online pre-LN layer-2 outputs
  const float* PT; const float* P; long p_ns; NetLayout L;
  const float* rew; const float* done; const float* logp_next; const float* log_alpha;
  int B, ln, sac, bcq, rpw; float gamma;
  float* qt; float* y; float* q;         // [2][B], [B], [2][B]
  float* dz2;                            // [2][B][HID]
  float* part; float* part_s; int nblk;  // [2][nblk][NSLOT][HID], [2][nblk][2] (sum dq, sum sq-err)
};

__global__ __launch_bounds__(256) void k_critic_tail(CriticTail p) {
  __shared__ float red[4 * NSLOT * HID];
  __shared__ float red_s[4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, net = blockIdx.y;
  const float* Pn = p.P + net * p.p_ns;
  const float4 wh = ld4(Pn + p.L.Wh + 4 * lane);
  const float bh = Pn[p.L.bh];
  float4 acc[3] = {f4(0.f), f4(0.f), f4(0.f)};
  float s_dq = 0.f, s_loss = 0.f;
  const float alpha = p.sac ? expf(*p.log_alpha) : 0.f;
  for (int it = 0; it < p.rpw; ++it) {
    const int b = (blockIdx.x * 4 + wave) * p.rpw + it;
    if (b >= p.B) break;
    float qtv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* Pt = p.PT + i * p.p_ns;
      float4 xh, yy; float mu, rs;
      ln_row(ld4(p.z2t + ((long)i * p.B + b) * HID + 4 * lane), Pt + p.L.g2, Pt + p.L.be2, lane, p.ln, xh, yy, mu, rs);
      qtv[i] = wave_sum(dot4(relu4(yy), ld4(Pt + p.L.Wh + 4 * lane))) + Pt[p.L.bh];
    }
    const float qmin = fminf(qtv[0], qtv[1]);
    float qp = p.bcq ? 0.75f * qmin + 0.25f * fmaxf(qtv[0], qtv[1]) : qmin;
    if (p.sac) qp -= alpha * p.logp_next[b];
    const float yv = p.rew[b] + (1.0f - p.done[b]) * p.gamma * qp;
    float4 xhat, yln; float mean, rstd;
    ln_row(ld4(p.z2 + ((long)net * p.B + b) * HID + 4 * lane), Pn + p.L.g2, Pn + p.L.be2, lane, p.ln, xhat, yln, mean, rstd);
    const float4 h = relu4(yln);
    const float qv = wave_sum(dot4(h, wh)) + bh;
    const float err = qv - yv;
    const float dq = 2.0f * err / (float)p.B;
    const float4 dy = gate4(wh * dq, yln);
    const float4 dz = ln_row_bwd(dy, xhat, rstd, Pn + p.L.g2, lane, p.ln);
    st4(p.dz2 + ((long)net * p.B + b) * HID + 4 * lane, dz);
    acc[0] = acc[0] + dy * xhat; acc[1] = acc[1] + dy; acc[2] = acc[2] + h * dq;
    s_dq += dq; s_loss += err * err;
    if (lane == 0) {
      p.q[(long)net * p.B + b] = qv;
      if (net == 0) { p.qt[b] = qtv[0]; p.qt[p.B + b] = qtv[1]; p.y[b] = yv; }
    }
  }
  const long blk = (long)net * p.nblk + blockIdx.x;
  block_store_partials(red, acc, 3, p.part + blk * NSLOT * HID);
  if (lane == 0) { red_s[wave][0] = s_dq; red_s[wave][1] = s_loss; }
  __syncthreads();
  if (threadIdx.x < 2)
    p.part_s[blk * 2 + threadIdx.x] = (red_s[0][threadIdx.x] + red_s[1][threadIdx.x]) + (red_s[2][threadIdx.x] + red_s[3][threadIdx.x]);
}

struct ActorQTail {
  const float* z2c;                      // [nq][B][HID]
  const float* P; long p_ns; NetLayout L;// online critics
  const float* logp; const float* log_alpha;
  int B, ln, sac, rpw;
  float* q; float* dz2;                  // [nq][B], [nq][B][HID]
  float* part_s; int nblk;               // [nblk][2]: loss partial in [.][1]
};

__global__ __launch_bounds__(256) void k_actorq_tail(ActorQTail p) {
  __shared__ float red_s[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = p.sac ? 2 : 1;
  const float alpha = p.sac ? expf(*p.log_alpha) : 0.f;
  float s_loss = 0.f;
  for (int it = 0; it < p.rpw; ++it) {
    const int b = (blockIdx.x * 4 + wave) * p.rpw + it;
    if (b >= p.B) break;
    float4 xhat[2], yln[2]; float rstd[2], qv[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i < nq) {
        const float* Pn = p.P + i * p.p_ns; float mu;
        ln_row(ld4(p.z2c + ((long)i * p.B + b) * HID + 4 * lane), Pn + p.L.g2, Pn + p.L.be2, lane, p.ln, xhat[i], yln[i], mu, rstd[i]);
        qv[i] = wave_sum(dot4(relu4(yln[i]), ld4(Pn + p.L.Wh + 4 * lane))) + Pn[p.L.bh];
      }
    }
    const bool first = p.sac ? (qv[0] <= qv[1]) : true;
    s_loss += p.sac ? (alpha * p.logp[b] - (first ? qv[0] : qv[1])) : -qv[0];
    const float invB = 1.0f / (float)p.B;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i < nq) {
        const float* Pn = p.P + i * p.p_ns;
        const float dq = ((i == 0) == first) ? -invB : 0.f;
        const float4 dy = gate4(ld4(Pn + p.L.Wh + 4 * lane) * dq, yln[i]);
        const float4 dz = ln_row_bwd(dy, xhat[i], rstd[i], Pn + p.L.g2, lane, p.ln);
        st4(p.dz2 + ((long)i * p.B + b) * HID + 4 * lane, dz);
        if (lane == 0) p.q[(long)i * p.B + b] = qv[i];
      }
    }
  }
  if (lane == 0) red_s[wave] = s_loss;
  __syncthreads();
  if (threadIdx.x == 0) p.part_s[blockIdx.x * 2 + 1] = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
}

struct LnBwd {               // dz = LNbwd(relu'(.) * dh) for a hidden layer; optional (dgamma, dbeta) partials
  const float* dh; long dh_ns; const float* z; long z_ns; const float* st; long st_ns; const float* h; long h_ns;
  const float* gamma; long p_ns;
  int B, ln, rpw, want_part;
  float* dz; long dz_ns;
  float* part; int nblk;
};

__global__ __launch_bounds__(256) void k_ln_bwd(LnBwd p) {
  __shared__ float red[4 * NSLOT * HID];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, net = blockIdx.y;
  const float* g = p.gamma + net * p.p_ns;
  float4 acc[2] = {f4(0.f), f4(0.f)};
  for (int it = 0; it < p.rpw; ++it) {
    const int b = (blockIdx.x * 4 + wave) * p.rpw + it;
    if (b >= p.B) break;
    const long ro = (long)b * HID + 4 * lane;
    const float4 dy = gate4(ld4(p.dh + net * p.dh_ns + ro), ld4(p.h + net * p.h_ns + ro));
    float4 xhat = f4(0.f); float rstd = 1.f;
    if (p.ln) {
      const float* st = p.st + net * p.st_ns + 2 * (long)b;
      rstd = st[1];
      xhat = (ld4(p.z + net * p.z_ns + ro) - f4(st[0])) * rstd;
    }
    st4(p.dz + net * p.dz_ns + ro, ln_row_bwd(dy, xhat, rstd, g, lane, p.ln));
    acc[0] = acc[0] + dy * xhat; acc[1] = acc[1] + dy;
  }
  if (p.want_part)
    block_store_partials(red, acc, 2, p.part + ((long)net * p.nblk + blockIdx.x) * NSLOT * HID);
}

struct ActorHeadBwd {
  const float* dA; long dA_ns; int ldA; int nq;   // [nq][B][ldA] grads wrt the action from each critic
  const float* tg; int a4; const float* eps; const float* log_alpha; const float* scale;
  const float* P; NetLayout L;
  const float* z2; const float* st2; const float* h2;
  int B, a, ln, sac, rpw;
  float* du; int ldu;                    // [B][ldu] grad wrt head outputs
  float* dz2;                            // [B][HID]
  float* part; int nblk;                 // [nblk][NSLOT][HID]
};

__global__ __launch_bounds__(256) void k_actor_head_bwd(ActorHeadBwd p) {
  __shared__ float red[4 * NSLOT * HID];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* Wh = p.P + p.L.Wh;
  float4 acc[2] = {f4(0.f), f4(0.f)};
  const float dlogp = p.sac ? expf(*p.log_alpha) / (float)p.B : 0.f;
  for (int it = 0; it < p.rpw; ++it) {
    const int b = (blockIdx.x * 4 + wave) * p.rpw + it;
    if (b >= p.B) break;
    float g_mean = 0.f, g_raw = 0.f;
    if (lane < p.a) {
      const int j = lane;
      float dAj = p.dA[(long)b * p.ldA + j];
      if (p.nq == 2) dAj += p.dA[p.dA_ns + (long)b * p.ldA + j];
      const float sc = p.scale[j];
      const float* tg = p.tg + (long)b * 4 * p.a4;
      if (p.sac) {
        const float t = tg[j], sd = tg[p.a4 + j], yt = tg[2 * p.a4 + j], e = p.eps[(long)b * p.a + j];
        const float omy2 = 1.0f - yt * yt;
        const float g0 = dAj * sc * omy2 + dlogp * (2.0f * sc * yt * omy2) / (sc * omy2 + 1e-6f);
        g_mean = g0;
        g_raw = (g0 * e * sd - dlogp) * 3.5f * (1.0f - t * t);
        p.du[(long)b * p.ldu + j] = g_mean;
        p.du[(long)b * p.ldu + p.a + j] = g_raw;
      } else {
        const float th = tg[j];
        g_mean = dAj * sc * (1.0f - th * th);
        p.du[(long)b * p.ldu + j] = g_mean;
      }
    }
    float4 dh = f4(0.f);
    for (int j = 0; j < p.a; ++j) {
      dh = dh + ld4(Wh + (long)j * HID + 4 * lane) * __shfl(g_mean, j);
      if (p.sac) dh = dh + ld4(Wh + (long)(p.a + j) * HID + 4 * lane) * __shfl(g_raw, j);
    }
    const long ro = (long)b * HID + 4 * lane;
    const float4 dy = gate4(dh, ld4(p.h2 + ro));
    float4 xhat = f4(0.f); float rstd = 1.f;
    if (p.ln) { rstd = p.st2[2 * b + 1]; xhat = (ld4(p.z2 + ro) - f4(p.st2[2 * b])) * rstd; }
    st4(p.dz2 + ro, ln_row_bwd(dy, xhat, rstd, p.P + p.L.g2, lane, p.ln));
    acc[0] = acc[0] + dy * xhat; acc[1] = acc[1] + dy;
  }
  block_store_partials(red, acc, 2, p.part + (long)blockIdx.x * NSLOT * HID);
}

// ------------------------------------------------------------------------------------------------ optimiser
struct AdamArgs {
  float* p; const float* g; float* m; float* v; long n;   // n multiple of 4
  const int* t; float lr, b1, b2, eps;
  const float* gscale;                   // optional gradient scale (clip_grad_norm_)
  float* targ; float tau;                // optional fused Polyak of the same arena
  const float* loss_part; int loss_n; int loss_stride; int loss_off; float loss_scale; float* loss_dst;
  int* tick;
};

__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
  const int t = *a.t;
  const double bc1 = 1.0 - pow((double)a.b1, (double)t), bc2 = 1.0 - pow((double)a.b2, (double)t);
  const float step = (float)((double)a.lr / bc1), sq2 = (float)sqrt(bc2);
  const float gs = a.gscale ? *a.gscale : 1.0f;
  const float omb1 = 1.0f - a.b1, omb2 = 1.0f - a.b2;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += (long)gridDim.x * 1024) {
    float4 g = ld4(a.g + i) * gs, m = ld4(a.m + i), v = ld4(a.v + i), p = ld4(a.p + i);
    m = m + (g - m) * omb1;
    v = v * a.b2 + g * g * omb2;
    p.x -= step * (m.x / (sqrtf(v.x) / sq2 + a.eps)); p.y -= step * (m.y / (sqrtf(v.y) / sq2 + a.eps));
    p.z -= step * (m.z / (sqrtf(v.z) / sq2 + a.eps)); p.w -= step * (m.w / (sqrtf(v.w) / sq2 + a.eps));
    st4(a.m + i, m); st4(a.v + i, v); st4(a.p + i, p);
    if (a.targ) { const float4 tt = ld4(a.targ + i); st4(a.targ + i, tt + (p - tt) * a.tau); }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    if (a.loss_dst) {
      float s = 0.f;
      for (int i = threadIdx.x; i < a.loss_n; i += 64) s += a.loss_part[(long)i * a.loss_stride + a.loss_off];
      s = wave_sum(s);
      if (threadIdx.x == 0) *a.loss_dst = s * a.loss_scale;
    }
    if (threadIdx.x == 0 && a.tick) *a.tick += 1;
  }
}

struct PolyakArgs { float* t0; const float* p0; long n0; float* t1; const float* p1; long n1; float tau; };
__global__ __launch_bounds__(256) void k_polyak(PolyakArgs a) {
  const long n = a.n0 + a.n1;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    float* t = i < a.n0 ? a.t0 + i : a.t1 + (i - a.n0);
    const float* p = i < a.n0 ? a.p0 + i : a.p1 + (i - a.n0);
    const float4 tt = ld4(t);
    st4(t, tt + (ld4(p) - tt) * a.tau);
  }
}

struct AlphaArgs {
  const float* logp; int B; float targ_ent; int autotune;
  float* la;                             // [0] log_alpha, [1] exp_avg, [2] exp_avg_sq
  DevCtl* ctl; float lr, b1, b2, eps;
  int* tick;
};
__global__ __launch_bounds__(256) void k_alpha_step(AlphaArgs a) {
  __shared__ float red[4];
  float s = 0.f;
  if (a.autotune)
    for (int i = threadIdx.x; i < a.B; i += 256) s += -a.logp[i] - a.targ_ent;
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float la = a.la[0];
    if (a.autotune) {
      const float mean_term = ((red[0] + red[1]) + (red[2] + red[3])) / (float)a.B;
      const float g = expf(la) * mean_term;          // d/dlog_alpha of alpha * mean_term; also the loss value
      a.ctl->metrics[2] = g;
      const int t = a.ctl->t_l + 1; a.ctl->t_l = t;
      float m = a.la[1], v = a.la[2];
      m = m + (g - m) * (1.0f - a.b1);
      v = v * a.b2 + g * g * (1.0f - a.b2);
      const double bc1 = 1.0 - pow((double)a.b1, (double)t), bc2 = 1.0 - pow((double)a.b2, (double)t);
      la -= (float)((double)a.lr / bc1) * (m / (sqrtf(v) / (float)sqrt(bc2) + a.eps));
      a.la[0] = la; a.la[1] = m; a.la[2] = v;
    }
    a.ctl->metrics[3] = expf(la);
    if (a.tick) *a.tick += 1;
  }
}

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
