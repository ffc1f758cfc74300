// Per-frame conditioning codes of a motion network as ONE workgroup per pass:
//   enc_a = AudioAttNet(AudioNet(a))            scene/motion_net.py:29-64, :67-99 (used at :283-289 / :672-677)
//   enc_e = cat(exp_encode_net(e[:-1]), e[-1:]) scene/motion_net.py:152-173, :297-299 / :684-686
// The reference runs this as ~11 cuDNN conv1d / GEMM launches plus their activations (and three times that in
// backward) on an 8-window batch: launch-latency bound.  Here every activation lives in LDS, weights are staged
// through LDS per layer group, and the backward pass produces every parameter gradient without atomics (deterministic).
//
// Layer stack (B = 8 windows, W = 16 samples, D = dim_in, M = mid, A = dim_aud, LeakyReLU slope 0.02):
//   conv k3 s2 p1: D->M (16->8), M->M (8->4), M->64 (4->2), 64->64 (2->1); fc 64->64 (leaky), fc 64->A
//   attention (B=1, channels = A features, length = 8 windows): conv k3 s1 p1 A->16->8->4->2->1 (leaky each),
//   linear 8->8, softmax, enc_a = sum_t y[t] * feat[t]
//   expression: relu(W1[16x5] e[:5]), W2[5x16], enc_e = (.., e[5])
#include "common.hpp"

namespace instag {
namespace {

constexpr int FT = 1024;          // threads of the single workgroup
constexpr int NB = 8;             // audio windows per frame (= attention sequence length)
constexpr int WIN = 16;
constexpr float SLOPE = 0.02f;
constexpr int NPARAM = 26;

struct FrameDims { int D, M, A, has_exp; };

// float offsets of every activation inside one LDS / saved block
struct FrameLayout {
  int x0, a1, a2, a3, a4, f1, f2, xt, c1, c2, c3, c4, c5, z, y, eh, end;
};

__host__ __device__ inline FrameLayout frame_layout(int D, int M, int A) {
  FrameLayout L;
  int o = 0;
  L.x0 = o; o += NB * D * WIN;
  L.a1 = o; o += NB * M * 8;
  L.a2 = o; o += NB * M * 4;
  L.a3 = o; o += NB * 64 * 2;
  L.a4 = o; o += NB * 64;
  L.f1 = o; o += NB * 64;
  L.f2 = o; o += NB * A;
  L.xt = o; o += A * NB;
  L.c1 = o; o += 16 * NB;
  L.c2 = o; o += 8 * NB;
  L.c3 = o; o += 4 * NB;
  L.c4 = o; o += 2 * NB;
  L.c5 = o; o += NB;
  L.z = o; o += NB;
  L.y = o; o += NB;
  L.eh = o; o += 16;
  L.end = o;
  return L;
}

struct ParamPtrs { const float* p[NPARAM]; };
struct GradPtrs { float* p[NPARAM]; };

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : SLOPE * v; }
__device__ __forceinline__ float dleaky(float post) { return post > 0.f ? 1.f : SLOPE; }

// Weights are staged into LDS one layer group at a time (coalesced 16-byte loads, all issued before the first
// wait): a MAC loop that reads its weight from global memory pays one L2 round trip per iteration.
// Groups: 0 conv1, 1 conv2, 2 conv3, 3 conv4, 4 fc1+fc2, 5 attention convs + linear + expression MLP.
__host__ __device__ inline int att_floats(int A) { return 48 * A + 784; }   // segments padded to 4 floats
__host__ __device__ inline int weight_stage_floats(int D, int M, int A) {
  int m = M * D * 3 + M;
  m = max(m, M * M * 3 + M);
  m = max(m, 64 * M * 3 + 64);
  m = max(m, 64 * 64 * 3 + 64);
  m = max(m, 64 * 64 + 64 + A * 64 + A);
  m = max(m, att_floats(A));
  return (m + 3) & ~3;
}

__device__ __forceinline__ void stage(float* dst, const float* __restrict__ src, int n) {
  if (src == nullptr) return;
  if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll 4
    for (int i = threadIdx.x; i < n / 4; i += FT) d4[i] = s4[i];
  } else {
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += FT) dst[i] = src[i];
  }
}

// offsets of the group-5 segments inside the staging buffer
struct AttW { int w[6], b[6], e1, e2; };
__device__ __forceinline__ AttW att_layout(int A) {
  AttW o;
  const int cin[6] = {A, 16, 8, 4, 2, 8}, cout[6] = {16, 8, 4, 2, 1, 8}, K[6] = {3, 3, 3, 3, 3, 1};
  int p = 0;
  for (int i = 0; i < 6; ++i) {
    o.w[i] = p; p += (cout[i] * cin[i] * K[i] + 3) & ~3;
    o.b[i] = p; p += (cout[i] + 3) & ~3;
  }
  o.e1 = p; p += 80;
  o.e2 = p;
  return o;
}

// y[b][co][l] = act(bias[co] + sum_{ci,k} w[co][ci][k] * x[b][ci][STRIDE*l + k - pad]); K = 1 is a linear layer.
// x, w, bias live in LDS; the ci loop is unrolled so that several LDS reads are in flight per thread (one
// workgroup on one CU: the passes are latency-, not throughput-bound).
template <int K, int STRIDE>
__device__ __forceinline__ void conv_forward(const float* x, const float* w, const float* bias, float* y, int B, int cin,
                                             int cout, int lin, int lout, bool act) {
  constexpr int PAD = K == 3 ? 1 : 0;
  const int total = B * cout * lout;
  for (int o = threadIdx.x; o < total; o += FT) {
    const int l = o % lout, co = (o / lout) % cout, b = o / (lout * cout);
    const float* wr = w + co * cin * K;
    const float* xb = x + b * cin * lin;
    const int p0 = STRIDE * l - PAD;
    bool ok[K];
#pragma unroll
    for (int k = 0; k < K; ++k) ok[k] = p0 + k >= 0 && p0 + k < lin;
    float acc0 = bias ? bias[co] : 0.f, acc1 = 0.f;
#pragma unroll 4
    for (int ci = 0; ci < cin; ++ci) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float xv = ok[k] ? xb[ci * lin + p0 + k] : 0.f;
        if (ci & 1) acc1 += wr[ci * K + k] * xv; else acc0 += wr[ci * K + k] * xv;
      }
    }
    const float acc = acc0 + acc1;
    y[o] = act ? leaky(acc) : acc;
  }
}

// g_out holds d(pre-activation) of the layer's output.  Writes dW, dbias to global and, when g_in != nullptr,
// d(pre-activation) of the layer's input (x is a leaky output when in_act) or the plain input gradient.
template <int K, int STRIDE>
__device__ __forceinline__ void conv_backward(const float* x, const float* g_out, const float* w, float* __restrict__ dW,
                                              float* __restrict__ dbias, float* g_in, int B, int cin, int cout, int lin,
                                              int lout, bool in_act) {
  constexpr int PAD = K == 3 ? 1 : 0;
  const int nW = cout * cin * K;
  const int nB = dbias ? cout : 0;
  const int nI = g_in ? B * cin * lin : 0;
  for (int idx = threadIdx.x; idx < nW + nB + nI; idx += FT) {
    if (idx < nW) {
      const int k = idx % K, ci = (idx / K) % cin, co = idx / (K * cin);
      float acc = 0.f;
      for (int b = 0; b < B; ++b) {
        const float* gb = g_out + (b * cout + co) * lout;
        const float* xb = x + (b * cin + ci) * lin;
#pragma unroll 4
        for (int l = 0; l < lout; ++l) {
          const int p = STRIDE * l + k - PAD;
          const float xv = (p >= 0 && p < lin) ? xb[p] : 0.f;
          acc += gb[l] * xv;
        }
      }
      dW[idx] = acc;
    } else if (idx < nW + nB) {
      const int co = idx - nW;
      float acc = 0.f;
      for (int b = 0; b < B; ++b)
#pragma unroll 4
        for (int l = 0; l < lout; ++l) acc += g_out[(b * cout + co) * lout + l];
      dbias[co] = acc;
    } else {
      const int i = idx - nW - nB;
      const int p = i % lin, ci = (i / lin) % cin, b = i / (lin * cin);
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int q = p + PAD - k;               // = STRIDE * l
        const int l = q / STRIDE;
        const bool ok = q >= 0 && (q % STRIDE) == 0 && l < lout;
        if (ok) {
          const float* gb = g_out + b * cout * lout + l;
          const float* wk = w + ci * K + k;
#pragma unroll 4
          for (int co = 0; co < cout; ++co) {
            const float t = wk[co * cin * K] * gb[co * lout];
            if (co & 1) acc1 += t; else acc0 += t;
          }
        }
      }
      float acc = acc0 + acc1;
      if (in_act) acc *= dleaky(x[i]);
      g_in[i] = acc;
    }
  }
}

__global__ void __launch_bounds__(FT)
frame_code_forward_kernel(FrameDims d, ParamPtrs P, const float* __restrict__ a, const float* __restrict__ e,
                          float* __restrict__ enc_a, float* __restrict__ enc_e, float* __restrict__ saved) {
  extern __shared__ __align__(16) float s[];
  const FrameLayout L = frame_layout(d.D, d.M, d.A);
  float* sw = s + ((L.end + 3) & ~3);            // weight staging buffer
  const int D = d.D, M = d.M, A = d.A;
  for (int i = threadIdx.x; i < NB * D * WIN; i += FT) s[L.x0 + i] = a[i];
  stage(sw, P.p[0], M * D * 3); stage(sw + M * D * 3, P.p[1], M);
  __syncthreads();
  conv_forward<3, 2>(s + L.x0, sw, sw + M * D * 3, s + L.a1, NB, D, M, 16, 8, true);
  __syncthreads();
  stage(sw, P.p[2], M * M * 3); stage(sw + M * M * 3, P.p[3], M);
  __syncthreads();
  conv_forward<3, 2>(s + L.a1, sw, sw + M * M * 3, s + L.a2, NB, M, M, 8, 4, true);
  __syncthreads();
  stage(sw, P.p[4], 64 * M * 3); stage(sw + 64 * M * 3, P.p[5], 64);
  __syncthreads();
  conv_forward<3, 2>(s + L.a2, sw, sw + 64 * M * 3, s + L.a3, NB, M, 64, 4, 2, true);
  __syncthreads();
  stage(sw, P.p[6], 64 * 64 * 3); stage(sw + 64 * 64 * 3, P.p[7], 64);
  __syncthreads();
  conv_forward<3, 2>(s + L.a3, sw, sw + 64 * 64 * 3, s + L.a4, NB, 64, 64, 2, 1, true);
  __syncthreads();
  stage(sw, P.p[8], 4096); stage(sw + 4096, P.p[9], 64); stage(sw + 4160, P.p[10], A * 64);
  stage(sw + 4160 + A * 64, P.p[11], A);
  __syncthreads();
  conv_forward<1, 1>(s + L.a4, sw, sw + 4096, s + L.f1, NB, 64, 64, 1, 1, true);
  __syncthreads();
  conv_forward<1, 1>(s + L.f1, sw + 4160, sw + 4160 + A * 64, s + L.f2, NB, 64, A, 1, 1, false);
  __syncthreads();
  const AttW aw = att_layout(A);
  {
    const int cin[6] = {A, 16, 8, 4, 2, 8}, cout[6] = {16, 8, 4, 2, 1, 8}, K[6] = {3, 3, 3, 3, 3, 1};
    for (int i = 0; i < 6; ++i) {
      stage(sw + aw.w[i], P.p[12 + 2 * i], cout[i] * cin[i] * K[i]);
      stage(sw + aw.b[i], P.p[13 + 2 * i], cout[i]);
    }
    if (d.has_exp) { stage(sw + aw.e1, P.p[24], 80); stage(sw + aw.e2, P.p[25], 80); }
  }
  for (int i = threadIdx.x; i < A * NB; i += FT) {            // xt[j][t] = feat[t][j]
    const int t = i % NB, j = i / NB;
    s[L.xt + i] = s[L.f2 + t * A + j];
  }
  __syncthreads();
  if (d.has_exp && threadIdx.x >= FT - 16) {                     // expression hidden layer on an idle part of the block
    const int h = threadIdx.x - (FT - 16);
    float acc = 0.f;
    for (int i = 0; i < 5; ++i) acc += sw[aw.e1 + h * 5 + i] * e[i];
    s[L.eh + h] = fmaxf(acc, 0.f);
  }
  conv_forward<3, 1>(s + L.xt, sw + aw.w[0], sw + aw.b[0], s + L.c1, 1, A, 16, NB, NB, true);  __syncthreads();
  conv_forward<3, 1>(s + L.c1, sw + aw.w[1], sw + aw.b[1], s + L.c2, 1, 16, 8, NB, NB, true);  __syncthreads();
  conv_forward<3, 1>(s + L.c2, sw + aw.w[2], sw + aw.b[2], s + L.c3, 1, 8, 4, NB, NB, true);   __syncthreads();
  conv_forward<3, 1>(s + L.c3, sw + aw.w[3], sw + aw.b[3], s + L.c4, 1, 4, 2, NB, NB, true);   __syncthreads();
  conv_forward<3, 1>(s + L.c4, sw + aw.w[4], sw + aw.b[4], s + L.c5, 1, 2, 1, NB, NB, true);   __syncthreads();
  conv_forward<1, 1>(s + L.c5, sw + aw.w[5], sw + aw.b[5], s + L.z, 1, NB, NB, 1, 1, false);   __syncthreads();
  if (threadIdx.x < NB) {
    float m = s[L.z];
    for (int t = 1; t < NB; ++t) m = fmaxf(m, s[L.z + t]);
    float sum = 0.f;
    for (int t = 0; t < NB; ++t) sum += expf(s[L.z + t] - m);
    s[L.y + threadIdx.x] = expf(s[L.z + threadIdx.x] - m) / sum;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < A; j += FT) {
    float acc = 0.f;
    for (int t = 0; t < NB; ++t) acc += s[L.y + t] * s[L.f2 + t * A + j];
    enc_a[j] = acc;
  }
  if (d.has_exp && threadIdx.x >= FT - 6) {
    const int q = threadIdx.x - (FT - 6);
    float acc;
    if (q < 5) {
      acc = 0.f;
      for (int h = 0; h < 16; ++h) acc += sw[aw.e2 + q * 16 + h] * s[L.eh + h];
    } else {
      acc = e[5];
    }
    enc_e[q] = acc;
  }
  for (int i = threadIdx.x; i < L.end - L.a1; i += FT) saved[i] = s[L.a1 + i];
}

__device__ __forceinline__ int co_of(int o, int lout) { return o / lout; }

// ---- forward over NB workgroups --------------------------------------------------------------------------------
// AudioNet's eight windows are independent (scene/motion_net.py:67-99): workgroup b runs window b through the four
// convolutions and two linear layers with EVERY weight already in LDS (all staging loads are issued at kernel start:
// one memory round trip instead of one per layer), the last workgroup to finish runs AudioAttNet over the eight feature
// vectors (+ the expression MLP).  The single-workgroup form took 97 us, ~5 us per layer, almost all of it waiting
// for the layer's weights and for barriers.
constexpr int FW = 256;           // threads per workgroup of the forward pass

// one output per group of `tpo` lanes (a power of two <= 64), each lane sums a slice of the input channels
template <int K, int STRIDE>
__device__ __forceinline__ void conv_forward_split(const float* x, const float* w, const float* bias, float* y, int cin,
                                                   int cout, int lin, int lout, bool act) {
  constexpr int PAD = K == 3 ? 1 : 0;
  const int total = cout * lout;
  int tpo = 1;
  while (tpo * 2 * total <= FW && tpo * 2 <= cin && tpo < 64) tpo *= 2;
  for (int base = 0; base < total; base += FW / tpo) {
    const int o = base + (int)threadIdx.x / tpo, part = (int)threadIdx.x % tpo;
    float acc = 0.f;
    if (o < total) {
      const int l = o % lout, co = o / lout;
      const float* wr = w + co * cin * K;
      const int p0 = STRIDE * l - PAD;
#pragma unroll 4
      for (int ci = part; ci < cin; ci += tpo) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int p = p0 + k;
          const float xv = (p >= 0 && p < lin) ? x[ci * lin + p] : 0.f;
          acc += wr[ci * K + k] * xv;
        }
      }
    }
    for (int m = tpo >> 1; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (o < total && part == 0) {
      acc += bias ? bias[co_of(o, lout)] : 0.f;
      y[o] = act ? leaky(acc) : acc;
    }
  }
}

__global__ void __launch_bounds__(FW)
frame_code_forward_split_kernel(FrameDims d, ParamPtrs P, const float* __restrict__ a, const float* __restrict__ e,
                                float* __restrict__ enc_a, float* __restrict__ enc_e, float* __restrict__ saved,
                                uint32_t* __restrict__ arrivals) {
  extern __shared__ __align__(16) float s[];      // (no static LDS: the dynamic block may be the whole 160 KB)
  const int D = d.D, M = d.M, A = d.A, b = blockIdx.x, tid = threadIdx.x;
  const FrameLayout L = frame_layout(D, M, A);
  // LDS: [weights of the six AudioNet layers][attention + expression weights][this window's activations][attention acts]
  const int w1 = 0, b1 = w1 + M * D * 3, w2 = b1 + ((M + 3) & ~3), b2 = w2 + M * M * 3, w3 = b2 + ((M + 3) & ~3),
            b3 = w3 + 64 * M * 3, w4 = b3 + 64, b4 = w4 + 64 * 64 * 3, w5 = b4 + 64, b5 = w5 + 4096, w6 = b5 + 64,
            b6 = w6 + A * 64, watt = b6 + ((A + 3) & ~3), acts = watt + att_floats(A);
  float* x0 = s + acts;                     // [D][16]
  float* a1 = x0 + D * WIN;                 // [M][8]
  float* a2 = a1 + M * 8;                   // [M][4]
  float* a3 = a2 + M * 4;                   // [64][2]
  float* a4 = a3 + 128;                     // [64]
  float* f1 = a4 + 64;                      // [64]
  float* f2 = f1 + 64;                      // [A]
  float* att = f2 + ((A + 3) & ~3);         // last workgroup: f2 of all windows [8][A], then xt .. eh in FrameLayout order
  volatile uint32_t* s_ticket = reinterpret_cast<volatile uint32_t*>(att + NB * A + (L.end - L.xt));
  // every load of the kernel's weights is issued here
  for (int i = tid; i < D * WIN; i += FW) x0[i] = a[b * D * WIN + i];
  auto stage = [&](int off, const float* src, int n) {
    if (src == nullptr) return;
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)(off * 4)) & 15) == 0) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
      float4* d4 = reinterpret_cast<float4*>(s + off);
#pragma unroll 4
      for (int i = tid; i < n / 4; i += FW) d4[i] = s4[i];
    } else {
#pragma unroll 4
      for (int i = tid; i < n; i += FW) s[off + i] = src[i];
    }
  };
  stage(w1, P.p[0], M * D * 3); stage(b1, P.p[1], M);
  stage(w2, P.p[2], M * M * 3); stage(b2, P.p[3], M);
  stage(w3, P.p[4], 64 * M * 3); stage(b3, P.p[5], 64);
  stage(w4, P.p[6], 64 * 64 * 3); stage(b4, P.p[7], 64);
  stage(w5, P.p[8], 4096); stage(b5, P.p[9], 64);
  stage(w6, P.p[10], A * 64); stage(b6, P.p[11], A);
  const AttW aw = att_layout(A);
  {
    const int cin[6] = {A, 16, 8, 4, 2, 8}, cout[6] = {16, 8, 4, 2, 1, 8}, K[6] = {3, 3, 3, 3, 3, 1};
    for (int i = 0; i < 6; ++i) {
      stage(watt + aw.w[i], P.p[12 + 2 * i], cout[i] * cin[i] * K[i]);
      stage(watt + aw.b[i], P.p[13 + 2 * i], cout[i]);
    }
    if (d.has_exp) { stage(watt + aw.e1, P.p[24], 80); stage(watt + aw.e2, P.p[25], 80); }
  }
  __syncthreads();
  conv_forward_split<3, 2>(x0, s + w1, s + b1, a1, D, M, 16, 8, true);    __syncthreads();
  conv_forward_split<3, 2>(a1, s + w2, s + b2, a2, M, M, 8, 4, true);     __syncthreads();
  conv_forward_split<3, 2>(a2, s + w3, s + b3, a3, M, 64, 4, 2, true);    __syncthreads();
  conv_forward_split<3, 2>(a3, s + w4, s + b4, a4, 64, 64, 2, 1, true);   __syncthreads();
  conv_forward_split<1, 1>(a4, s + w5, s + b5, f1, 64, 64, 1, 1, true);   __syncthreads();
  conv_forward_split<1, 1>(f1, s + w6, s + b6, f2, 64, A, 1, 1, false);   __syncthreads();
  // this window's slices of the saved activations (batch-major arrays of FrameLayout, offsets relative to a1)
  {
    float* sv = saved - L.a1;
    for (int i = tid; i < M * 8; i += FW) sv[L.a1 + b * M * 8 + i] = a1[i];
    for (int i = tid; i < M * 4; i += FW) sv[L.a2 + b * M * 4 + i] = a2[i];
    for (int i = tid; i < 128; i += FW) sv[L.a3 + b * 128 + i] = a3[i];
    for (int i = tid; i < 64; i += FW) { sv[L.a4 + b * 64 + i] = a4[i]; sv[L.f1 + b * 64 + i] = f1[i]; }
    for (int i = tid; i < A; i += FW) sv[L.f2 + b * A + i] = f2[i];
  }
  // hand-off to the last workgroup: stores drained by every wave, workgroup barrier, agent-scope release, then the
  // arrival counter; the workgroup that draws the last ticket acquires and reads all eight feature vectors
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *s_ticket = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (*s_ticket != NB - 1) return;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *arrivals = 0u;                          // ready for the next launch (stream-ordered behind this one)
  }
  __syncthreads();
  // ---- AudioAttNet over the eight windows + the expression MLP (activations in FrameLayout order from xt on) ----
  float* F2 = att;                           // [8][A]
  float* aa = F2 + NB * A - L.xt;            // aa[L.<act>] for act in xt .. eh
  const float* svr = saved - L.a1;
  for (int i = tid; i < NB * A; i += FW) F2[i] = svr[L.f2 + i];
  __syncthreads();
  for (int i = tid; i < A * NB; i += FW) {            // xt[j][t] = feat[t][j]
    const int t = i % NB, j = i / NB;
    aa[L.xt + i] = F2[t * A + j];
  }
  if (d.has_exp && tid >= FW - 16) {                   // expression hidden layer on an idle part of the block
    const int h = tid - (FW - 16);
    float acc = 0.f;
    for (int i = 0; i < 5; ++i) acc += s[watt + aw.e1 + h * 5 + i] * e[i];
    aa[L.eh + h] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  const float* wa = s + watt;
  conv_forward_split<3, 1>(aa + L.xt, wa + aw.w[0], wa + aw.b[0], aa + L.c1, A, 16, NB, NB, true);  __syncthreads();
  conv_forward_split<3, 1>(aa + L.c1, wa + aw.w[1], wa + aw.b[1], aa + L.c2, 16, 8, NB, NB, true);  __syncthreads();
  conv_forward_split<3, 1>(aa + L.c2, wa + aw.w[2], wa + aw.b[2], aa + L.c3, 8, 4, NB, NB, true);   __syncthreads();
  conv_forward_split<3, 1>(aa + L.c3, wa + aw.w[3], wa + aw.b[3], aa + L.c4, 4, 2, NB, NB, true);   __syncthreads();
  conv_forward_split<3, 1>(aa + L.c4, wa + aw.w[4], wa + aw.b[4], aa + L.c5, 2, 1, NB, NB, true);   __syncthreads();
  conv_forward_split<1, 1>(aa + L.c5, wa + aw.w[5], wa + aw.b[5], aa + L.z, NB, NB, 1, 1, false);   __syncthreads();
  if (tid < NB) {
    float m = aa[L.z];
    for (int t = 1; t < NB; ++t) m = fmaxf(m, aa[L.z + t]);
    float sum = 0.f;
    for (int t = 0; t < NB; ++t) sum += expf(aa[L.z + t] - m);
    aa[L.y + tid] = expf(aa[L.z + tid] - m) / sum;
  }
  __syncthreads();
  for (int j = tid; j < A; j += FW) {
    float acc = 0.f;
    for (int t = 0; t < NB; ++t) acc += aa[L.y + t] * F2[t * A + j];
    enc_a[j] = acc;
  }
  if (d.has_exp && tid >= FW - 6) {
    const int q = tid - (FW - 6);
    float acc;
    if (q < 5) {
      acc = 0.f;
      for (int h = 0; h < 16; ++h) acc += s[watt + aw.e2 + q * 16 + h] * aa[L.eh + h];
    } else {
      acc = e[5];
    }
    enc_e[q] = acc;
  }
  {
    float* sv = saved - L.a1;
    for (int i = tid; i < L.end - L.xt; i += FW) sv[L.xt + i] = aa[L.xt + i];
  }
}

// floats of parameter i (C ABI order: 4 conv + 2 fc layers of AudioNet, 5 conv + 1 linear of AudioAttNet, the two
// expression layers; weight, bias each)
__host__ __device__ inline int param_floats(int i, int D, int M, int A) {
  const int n[NPARAM] = {M * D * 3, M, M * M * 3, M, 64 * M * 3, 64, 64 * 64 * 3, 64, 4096, 64, A * 64, A,
                         16 * A * 3, 16, 8 * 16 * 3, 8, 4 * 8 * 3, 4, 2 * 4 * 3, 2, 1 * 2 * 3, 1, NB * NB, NB, 80, 80};
  return n[i];
}
__host__ __device__ inline int param_total(int D, int M, int A) {
  int t = 0;
  for (int i = 0; i < NPARAM; ++i) t += param_floats(i, D, M, A);
  return t;
}

// SPLIT: one workgroup per audio window.  Every workgroup repeats the small attention / expression
// stage (it needs all eight windows' features) and then runs AudioNet's backward for ITS window only; the parameter
// gradients go to the workgroup's own row of `partial` [NB][param_total] and frame_code_grad_reduce_kernel adds the rows up
// in order (AudioNet's parameters) or takes row 0 (the attention stage's, identical in every row).
template <bool SPLIT>
__global__ void __launch_bounds__(FT)
frame_code_backward_kernel(FrameDims d, ParamPtrs P, GradPtrs G, const float* __restrict__ a,
                           const float* __restrict__ e, const float* __restrict__ saved,
                           const float* __restrict__ d_enc_a, const float* __restrict__ d_enc_e,
                           float* __restrict__ partial) {
  extern __shared__ __align__(16) float s[];
  const FrameLayout L = frame_layout(d.D, d.M, d.A);
  const int D = d.D, M = d.M, A = d.A;
  const int wb = SPLIT ? (int)blockIdx.x : 0;       // first window and number of windows of this workgroup
  const int WB = SPLIT ? 1 : NB;
  if (SPLIT) {
    float* row = partial + (size_t)blockIdx.x * param_total(D, M, A);
    int off = 0;
    for (int i = 0; i < NPARAM; ++i) { G.p[i] = row + off; off += param_floats(i, D, M, A); }
  }
  float* g = s + L.end - L.a1;                  // g[L.<act>] = gradient buffer of activation <act> (offsets >= a1)
  float* sw = s + ((2 * L.end - L.a1 + 3) & ~3);
  const AttW aw = att_layout(A);
  for (int i = threadIdx.x; i < NB * D * WIN; i += FT) s[L.x0 + i] = a[i];
  for (int i = threadIdx.x; i < L.end - L.a1; i += FT) s[L.a1 + i] = saved[i];
  {
    const int cin[6] = {A, 16, 8, 4, 2, 8}, cout[6] = {16, 8, 4, 2, 1, 8}, K[6] = {3, 3, 3, 3, 3, 1};
    for (int i = 0; i < 6; ++i) stage(sw + aw.w[i], P.p[12 + 2 * i], cout[i] * cin[i] * K[i]);
    if (d.has_exp) stage(sw + aw.e2, P.p[25], 80);
  }
  __syncthreads();

  // enc_a = sum_t y[t] feat[t]: d_y, d_feat; then softmax
  if (threadIdx.x < NB) {
    float acc = 0.f;
    for (int j = 0; j < A; ++j) acc += d_enc_a[j] * s[L.f2 + threadIdx.x * A + j];
    g[L.y + threadIdx.x] = acc;
  }
  for (int i = threadIdx.x; i < NB * A; i += FT) g[L.f2 + i] = s[L.y + i / A] * d_enc_a[i % A];
  if (d.has_exp) {
    // enc_e[q<5] = sum_h W2[q][h] eh[h], eh = relu(W1 e[:5]); 80 + 80 threads own one weight each
    const int t = (int)threadIdx.x - 64;
    if (t >= 0 && t < 80) {
      const int q = t / 16, h = t % 16;
      G.p[25][t] = (d_enc_e ? d_enc_e[q] : 0.f) * s[L.eh + h];
    } else if (t >= 80 && t < 160) {
      const int i = (t - 80) % 5, h = (t - 80) / 5;
      float dh = 0.f;
      if (d_enc_e && s[L.eh + h] > 0.f)
        for (int q = 0; q < 5; ++q) dh += sw[aw.e2 + q * 16 + h] * d_enc_e[q];
      G.p[24][h * 5 + i] = dh * e[i];
    }
  }
  __syncthreads();
  if (threadIdx.x < NB) {
    float dot = 0.f;
    for (int t = 0; t < NB; ++t) dot += s[L.y + t] * g[L.y + t];
    g[L.z + threadIdx.x] = s[L.y + threadIdx.x] * (g[L.y + threadIdx.x] - dot);
  }
  __syncthreads();
  conv_backward<1, 1>(s + L.c5, g + L.z, sw + aw.w[5], G.p[22], G.p[23], g + L.c5, 1, NB, NB, 1, 1, true);   __syncthreads();
  conv_backward<3, 1>(s + L.c4, g + L.c5, sw + aw.w[4], G.p[20], G.p[21], g + L.c4, 1, 2, 1, NB, NB, true);  __syncthreads();
  conv_backward<3, 1>(s + L.c3, g + L.c4, sw + aw.w[3], G.p[18], G.p[19], g + L.c3, 1, 4, 2, NB, NB, true);  __syncthreads();
  conv_backward<3, 1>(s + L.c2, g + L.c3, sw + aw.w[2], G.p[16], G.p[17], g + L.c2, 1, 8, 4, NB, NB, true);  __syncthreads();
  conv_backward<3, 1>(s + L.c1, g + L.c2, sw + aw.w[1], G.p[14], G.p[15], g + L.c1, 1, 16, 8, NB, NB, true); __syncthreads();
  conv_backward<3, 1>(s + L.xt, g + L.c1, sw + aw.w[0], G.p[12], G.p[13], g + L.xt, 1, A, 16, NB, NB, false);
  __syncthreads();
  for (int i = threadIdx.x; i < NB * A; i += FT) {            // feat[t][j] also feeds xt[j][t]
    const int j = i % A, t = i / A;
    g[L.f2 + i] += g[L.xt + j * NB + t];
  }
  stage(sw, P.p[8], 4096); stage(sw + 4096, P.p[10], A * 64);
  __syncthreads();
  // AudioNet, windows wb .. wb + WB - 1 (activations are [window][channel][position])
#define WIN_AT(base, c, l) ((base) + wb * (c) * (l))
  conv_backward<1, 1>(WIN_AT(s + L.f1, 64, 1), WIN_AT(g + L.f2, A, 1), sw + 4096, G.p[10], G.p[11], WIN_AT(g + L.f1, 64, 1), WB, 64, A, 1, 1, true); __syncthreads();
  conv_backward<1, 1>(WIN_AT(s + L.a4, 64, 1), WIN_AT(g + L.f1, 64, 1), sw, G.p[8], G.p[9], WIN_AT(g + L.a4, 64, 1), WB, 64, 64, 1, 1, true);         __syncthreads();
  stage(sw, P.p[6], 64 * 64 * 3);
  __syncthreads();
  conv_backward<3, 2>(WIN_AT(s + L.a3, 64, 2), WIN_AT(g + L.a4, 64, 1), sw, G.p[6], G.p[7], WIN_AT(g + L.a3, 64, 2), WB, 64, 64, 2, 1, true);         __syncthreads();
  stage(sw, P.p[4], 64 * M * 3);
  __syncthreads();
  conv_backward<3, 2>(WIN_AT(s + L.a2, M, 4), WIN_AT(g + L.a3, 64, 2), sw, G.p[4], G.p[5], WIN_AT(g + L.a2, M, 4), WB, M, 64, 4, 2, true);          __syncthreads();
  stage(sw, P.p[2], M * M * 3);
  __syncthreads();
  conv_backward<3, 2>(WIN_AT(s + L.a1, M, 8), WIN_AT(g + L.a2, M, 4), sw, G.p[2], G.p[3], WIN_AT(g + L.a1, M, 8), WB, M, M, 8, 4, true);           __syncthreads();
  conv_backward<3, 2>(WIN_AT(s + L.x0, D, 16), WIN_AT(g + L.a1, M, 8), nullptr, G.p[0], G.p[1], nullptr, WB, D, M, 16, 8, false);
#undef WIN_AT
}

__global__ void __launch_bounds__(256)
frame_code_grad_reduce_kernel(FrameDims d, GradPtrs G, const float* __restrict__ partial) {
  const int PT = param_total(d.D, d.M, d.A);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= PT) return;
  int i = 0, off = 0;
  for (; i < NPARAM; ++i) {
    const int n = param_floats(i, d.D, d.M, d.A);
    if (t < off + n) break;
    off += n;
  }
  if (G.p[i] == nullptr) return;
  float v = partial[t];
  if (i < 12) {                         // AudioNet: one contribution per window, fixed order
    for (int b = 1; b < NB; ++b) v += partial[(size_t)b * PT + t];
  }
  G.p[i][t - off] = v;
}

inline int set_lds_limit() {
  if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(frame_code_forward_kernel), 160 * 1024)) return rc;
  if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(frame_code_backward_kernel<false>), 160 * 1024)) return rc;
  if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(frame_code_backward_kernel<true>), 160 * 1024)) return rc;
  return set_max_dynamic_lds(reinterpret_cast<const void*>(frame_code_forward_split_kernel), 160 * 1024);
}

// LDS floats of the split forward kernel: all weights + one window's activations + the attention stage
inline size_t split_forward_floats(int D, int M, int A) {
  const FrameLayout L = frame_layout(D, M, A);
  size_t w = (size_t)M * D * 3 + ((M + 3) & ~3) + (size_t)M * M * 3 + ((M + 3) & ~3) + (size_t)64 * M * 3 + 64 +
             64 * 64 * 3 + 64 + 4096 + 64 + (size_t)A * 64 + ((A + 3) & ~3) + att_floats(A);
  size_t acts = (size_t)D * WIN + M * 8 + M * 4 + 128 + 64 + 64 + ((A + 3) & ~3) + NB * A + (L.end - L.xt);
  return w + acts + 8;
}

inline bool dims_ok(int D, int M, int A) {
  if (D < 1 || M < 1 || A < 1) return false;
  const FrameLayout L = frame_layout(D, M, A);
  return (size_t)(2 * L.end - L.a1 + 4 + weight_stage_floats(D, M, A)) * sizeof(float) <= 160u * 1024u;
}

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

int64_t instag_frame_code_saved_floats(int32_t dim_in, int32_t mid, int32_t dim_aud) {
  if (!dims_ok(dim_in, mid, dim_aud)) return -1;
  const FrameLayout L = frame_layout(dim_in, mid, dim_aud);
  return L.end - L.a1;
}

int instag_frame_code_forward(const float* a, const float* e, const float* const* params, float* enc_a,
                              float* enc_e, float* saved, int32_t dim_in, int32_t mid, int32_t dim_aud,
                              uint32_t* arrivals, instag_stream_t stream) {
  INSTAG_REQUIRE(a && params && enc_a && saved, "frame_code_forward: NULL tensor");
  INSTAG_REQUIRE(dims_ok(dim_in, mid, dim_aud), "frame_code: activations do not fit the 160 KB LDS");
  INSTAG_REQUIRE((e == nullptr) == (enc_e == nullptr), "frame_code_forward: e and enc_e go together");
  const int has_exp = e != nullptr;
  ParamPtrs P;
  for (int i = 0; i < NPARAM; ++i) {
    P.p[i] = params[i];
    INSTAG_REQUIRE(P.p[i] || (i >= 24 && !has_exp), "frame_code_forward: NULL parameter");
  }
  if (int rc = set_lds_limit()) return rc;
  const FrameLayout L = frame_layout(dim_in, mid, dim_aud);
  const FrameDims d{dim_in, mid, dim_aud, has_exp};
  const size_t split_bytes = split_forward_floats(dim_in, mid, dim_aud) * sizeof(float);
  if (arrivals != nullptr && split_bytes <= 160u * 1024u) {
    // one workgroup per audio window, the last one to arrive runs the attention stage (see the kernel)
    frame_code_forward_split_kernel<<<NB, FW, split_bytes, (hipStream_t)stream>>>(d, P, a, e, enc_a, enc_e, saved,
                                                                                   arrivals);
    INSTAG_CHECK_LAUNCH();
    return INSTAG_OK;
  }
  frame_code_forward_kernel<<<1, FT, (size_t)(L.end + 4 + weight_stage_floats(dim_in, mid, dim_aud)) * sizeof(float),
                              (hipStream_t)stream>>>(d, P, a, e, enc_a, enc_e,
                                                                                        saved);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

size_t instag_frame_code_backward_workspace_bytes(int32_t dim_in, int32_t mid, int32_t dim_aud) {
  if (!dims_ok(dim_in, mid, dim_aud)) return 0;
  return (size_t)NB * param_total(dim_in, mid, dim_aud) * sizeof(float);
}

int instag_frame_code_backward(const float* a, const float* e, const float* const* params, const float* saved,
                               const float* d_enc_a, const float* d_enc_e, float* const* grads, int32_t dim_in,
                               int32_t mid, int32_t dim_aud, void* workspace, size_t workspace_bytes,
                               instag_stream_t stream) {
  INSTAG_REQUIRE(a && params && saved && d_enc_a && grads, "frame_code_backward: NULL tensor");
  INSTAG_REQUIRE(dims_ok(dim_in, mid, dim_aud), "frame_code: activations do not fit the 160 KB LDS");
  const int has_exp = e != nullptr;
  ParamPtrs P;
  GradPtrs G;
  for (int i = 0; i < NPARAM; ++i) {
    P.p[i] = params[i];
    G.p[i] = grads[i];
    INSTAG_REQUIRE((P.p[i] && G.p[i]) || (i >= 24 && !has_exp), "frame_code_backward: NULL parameter / gradient");
  }
  if (int rc = set_lds_limit()) return rc;
  const FrameLayout L = frame_layout(dim_in, mid, dim_aud);
  const FrameDims d{dim_in, mid, dim_aud, has_exp};
  const size_t lds = (size_t)(2 * L.end - L.a1 + 4 + weight_stage_floats(dim_in, mid, dim_aud)) * sizeof(float);
  if (workspace != nullptr && workspace_bytes >= instag_frame_code_backward_workspace_bytes(dim_in, mid, dim_aud)) {
    // one workgroup per audio window + a fixed-order sum of the eight rows of parameter gradients
    frame_code_backward_kernel<true><<<NB, FT, lds, (hipStream_t)stream>>>(d, P, G, a, e, saved, d_enc_a, d_enc_e,
                                                                          (float*)workspace);
    INSTAG_CHECK_LAUNCH();
    const int PT = param_total(dim_in, mid, dim_aud);
    frame_code_grad_reduce_kernel<<<(PT + 255) / 256, 256, 0, (hipStream_t)stream>>>(d, G, (const float*)workspace);
    INSTAG_CHECK_LAUNCH();
    return INSTAG_OK;
  }
  frame_code_backward_kernel<false><<<1, FT, lds, (hipStream_t)stream>>>(d, P, G, a, e, saved, d_enc_a, d_enc_e, nullptr);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // extern "C"
