// Bias-free ReLU MLP (2 or 3 layers) over N rows on the f32 matrix cores of gfx950.
//
// Replaces the per-Gaussian `MLP` chains of the reference's motion networks
// (scene/motion_net.py:152-173, used at :234-238, :600-604: sigma_net 74->64->64->11 / 74->32->32->11,
// aud_ch_att_net 36->32->32, eye_att_net 36->16->6, align_net 36->32->6), which the reference runs as
// separate eager Linear/ReLU kernels with B = N Gaussians.
//
// MI355X design.  v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) computes the TRANSPOSED product
// Z^T[o][row] = sum_k W[o][k] * X^T[k][row]: the 32 rows of a wave's tile sit on the lanes
// (row = lane&31) and every lane keeps 16 features per 32-feature block in registers,
//     block b, register r, half h = lane>>5   <->   feature 32b + (r&3) + 8(r>>2) + 4h.
// The accumulator layout of one layer IS the B-operand layout of the next (register r of block b
// feeds MFMA k-step (b, r) directly), so the whole chain -- forward and the backward-data chain
// dX^T = W^T dZ^T -- runs in registers with no shuffles and no LDS round trip for activations.
// Weights live in LDS row-major with an odd row stride: the forward A-operand read (lanes vary the
// output row) and the backward A-operand read (lanes vary the input column) are both conflict-free.
// Weight gradients dW[o][k] = sum_rows dZ[row][o] * In[row][k] reduce over rows = the MFMA k index,
// with both operands read straight from global memory (one coalesced 128-B row segment per half
// wave); per-wave partial tiles are combined in a fixed order (LDS, then a second pass), so the
// result is bitwise reproducible.
#include <cstdlib>

#include "common.hpp"

namespace instag {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int MLP_BLOCK = 256;

__device__ __forceinline__ constexpr int feat(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- layout-L tile load / store (rows on lanes) -------------------------------------------------
// Branch-free: every load is issued unconditionally from a clamped address and the value selected afterwards.  (With
// the loads inside `if (in range)` regions writing registers that were zero-initialised first, the compiler put an
// s_waitcnt vmcnt(1) behind every 8-byte load of the K = 74 path -- twenty serialised round trips per tile.)
// KC > 0: the row length is known at compile time (every alignment and range test folds).
template <int KC = 0>
__device__ __forceinline__ void load_block(const float* __restrict__ X, size_t row, bool valid, int Krt, int b, int h,
                                           f32x16& v) {
  const int K = KC > 0 ? KC : Krt;
  const float* __restrict__ base = X + (valid ? row * (size_t)K : 0);
  if ((K & 3) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f0 = 32 * b + 8 * q + 4 * h;
      const bool ok = valid && f0 < K;
      const float4 t = *reinterpret_cast<const float4*>(base + (ok ? f0 : 0));
      v[4 * q + 0] = ok ? t.x : 0.f; v[4 * q + 1] = ok ? t.y : 0.f;
      v[4 * q + 2] = ok ? t.z : 0.f; v[4 * q + 3] = ok ? t.w : 0.f;
    }
  } else if ((K & 1) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f0 = 32 * b + 8 * q + 4 * h;
      const bool ok0 = valid && f0 < K, ok1 = valid && f0 + 2 < K;
      const float2 a = *reinterpret_cast<const float2*>(base + (ok0 ? f0 : 0));
      const float2 c = *reinterpret_cast<const float2*>(base + (ok1 ? f0 + 2 : 0));
      v[4 * q + 0] = ok0 ? a.x : 0.f; v[4 * q + 1] = ok0 ? a.y : 0.f;
      v[4 * q + 2] = ok1 ? c.x : 0.f; v[4 * q + 3] = ok1 ? c.y : 0.f;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 32 * b + 8 * q + 4 * h + i;
        const bool ok = valid && f < K;
        const float t = base[ok ? f : 0];
        v[4 * q + i] = ok ? t : 0.f;
      }
  }
}

// The alignment case is chosen once per block (not per group of four features): per-slot scalar branches cost more
// than the stores they guard.
template <int KC = 0>
__device__ __forceinline__ void store_block(float* __restrict__ Y, size_t row, bool valid, int Krt, int b, int h,
                                            const f32x16& v) {
  const int K = KC > 0 ? KC : Krt;
  float* __restrict__ base = Y + row * (size_t)K;
  if ((K & 3) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f0 = 32 * b + 8 * q + 4 * h;
      if (valid && f0 < K)
        *reinterpret_cast<float4*>(base + f0) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
  } else if ((K & 1) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f0 = 32 * b + 8 * q + 4 * h;
      if (valid && f0 < K) *reinterpret_cast<float2*>(base + f0) = make_float2(v[4 * q], v[4 * q + 1]);
      if (valid && f0 + 2 < K) *reinterpret_cast<float2*>(base + f0 + 2) = make_float2(v[4 * q + 2], v[4 * q + 3]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 32 * b + 8 * q + 4 * h + i;
        if (valid && f < K) base[f] = v[4 * q + i];
      }
  }
}

// NB blocks of a tile whose row length is most likely KC (the hidden widths 16 / 32 / 64 = 8 * HQ): one uniform test,
// then straight-line code
template <int NB, int KC>
__device__ __forceinline__ void load_blocks(const float* __restrict__ X, size_t row, bool valid, int K, int h,
                                            f32x16 (&v)[NB]) {
  if (K == KC) {
#pragma unroll
    for (int b = 0; b < NB; ++b) load_block<KC>(X, row, valid, K, b, h, v[b]);
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) load_block<0>(X, row, valid, K, b, h, v[b]);
  }
}

template <int NB, int KC>
__device__ __forceinline__ void store_blocks(float* __restrict__ Y, size_t row, bool valid, int K, int h,
                                             const f32x16 (&v)[NB]) {
  if (K == KC) {
#pragma unroll
    for (int b = 0; b < NB; ++b) store_block<KC>(Y, row, valid, K, b, h, v[b]);
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) store_block<0>(Y, row, valid, K, b, h, v[b]);
  }
}

// ---- weights: global [O][K] row-major -> LDS [OP][32*KB+1], zero padded ------------------------------
// 8 rows x 32 consecutive columns per pass: coalesced, division-free, fully unrolled (every load of a thread is
// issued before the first wait).  The pad column (index 32*KB) is never read.
template <int OP, int KB, int THREADS = 256>
__device__ __forceinline__ void stage_weights(float* __restrict__ lds, const float* __restrict__ W, int O, int K) {
  constexpr int KS = KB * 32 + 1, RO = THREADS / 32;      // RO rows per pass
  static_assert(OP % RO == 0, "stage_weights: rows per pass must divide the padded row count");
  const int ro = threadIdx.x >> 5, kk = threadIdx.x & 31;
  float v[OP / RO][KB];
#pragma unroll
  for (int i = 0; i < OP / RO; ++i)
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      const int o = ro + RO * i, k = 32 * b + kk;
      v[i][b] = (o < O && k < K) ? W[o * K + k] : 0.f;
    }
#pragma unroll
  for (int i = 0; i < OP / RO; ++i)
#pragma unroll
    for (int b = 0; b < KB; ++b) lds[(ro + RO * i) * KS + 32 * b + kk] = v[i][b];
}

// Z^T = W X^T for one layer: in[KB] (layout L) -> acc[OB] (layout L).  KQ = number of 8-feature groups of the
// input that hold data (compile time: a run-time test per k-step would split the chain into basic blocks and
// make the compiler shuttle the accumulators between AGPRs and VGPRs around every MFMA).
template <int KQ, int KB, int OB>
__device__ __forceinline__ void layer_forward(const float* __restrict__ Wl, const f32x16 (&in)[KB], f32x16 (&acc)[OB],
                                              int l31, int h) {
  constexpr int KS = KB * 32 + 1;
#pragma unroll
  for (int t = 0; t < OB; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
  for (int b = 0; b < KB; ++b) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (4 * b + (r >> 2) < KQ) {       // folds at compile time
        const int k = 32 * b + feat(r, h);
#pragma unroll
        for (int t = 0; t < OB; ++t) {
          const float a = Wl[(32 * t + l31) * KS + k];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in[b][r], acc[t], 0, 0, 0);
        }
      }
    }
  }
}

// dIn^T = W^T dZ^T for one layer: dz[OB] -> din[KB]; OQ = number of 8-feature groups of dz that hold data
template <int OQ, int OB, int KB>
__device__ __forceinline__ void layer_backward(const float* __restrict__ Wl, const f32x16 (&dz)[OB], f32x16 (&din)[KB],
                                               int l31, int h) {
  constexpr int KS = KB * 32 + 1;
#pragma unroll
  for (int b = 0; b < KB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) din[b][i] = 0.f;
#pragma unroll
  for (int t = 0; t < OB; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (4 * t + (r >> 2) < OQ) {
        const int o = 32 * t + feat(r, h);
#pragma unroll
        for (int b = 0; b < KB; ++b) {
          const float a = Wl[o * KS + 32 * b + l31];
          din[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dz[t][r], din[b], 0, 0, 0);
        }
      }
    }
  }
}

template <int NB>
__device__ __forceinline__ void relu_blocks(f32x16 (&v)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) v[b][i] = fmaxf(v[b][i], 0.f);
}

template <int NB>
__device__ __forceinline__ void mask_blocks(f32x16 (&g)[NB], const f32x16 (&act)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) g[b][i] = act[b][i] > 0.f ? g[b][i] : 0.f;
}

struct MlpDims { int N, K0, H, O; };

constexpr int GLUE_KX = 36, GLUE_KA = 32, GLUE_KE = 6;      // the universal field's widths (groups of four features)

// GLUE (forward): sigma_net's input cat(enc_x, enc_a * aud, enc_e * relu(eye_pre)) is formed in the registers that feed
// the first layer instead of by motion_glue_forward_kernel (17 us at 100k rows); the assembled rows are still written
// out once (h_in: the first layer's weight gradient reads them), and so are the two row norms (amb).
struct GlueFwd {
  const float* enc_x; const float* aud; const float* eye_pre; const float* enc_a; const float* enc_e;
  float* h_in; float* amb;
};

// KQ0 / HQ / OQ: input, hidden and output widths in groups of 8 features (rounded up)
template <int KQ0, int HQ, int OQ, int NL, bool GLUE = false>
__global__ void __launch_bounds__(MLP_BLOCK)
mlp_forward_kernel(MlpDims d, const float* __restrict__ X, const float* __restrict__ W1,
                   const float* __restrict__ W2, const float* __restrict__ W3, float* __restrict__ Y,
                   float* __restrict__ A1, float* __restrict__ A2, GlueFwd gf = GlueFwd{}) {
  extern __shared__ __align__(16) float s_w[];
  constexpr int KB0 = (KQ0 + 3) / 4, HB = (HQ + 3) / 4;
  constexpr int KP0 = KB0 * 32, HP = HB * 32;
  float* w1 = s_w;                                    // [HP][KP0+1]
  float* w2 = w1 + HP * (KP0 + 1);                    // [NL==3 ? HP : 32][HP+1]
  float* w3 = w2 + (NL == 3 ? HP : 32) * (HP + 1);    // [32][HP+1]  (NL==3 only)
  stage_weights<HP, KB0>(w1, W1, d.H, d.K0);
  stage_weights<(NL == 3 ? HP : 32), HB>(w2, W2, NL == 3 ? d.H : d.O, d.H);
  if (NL == 3) stage_weights<32, HB>(w3, W3, d.O, d.H);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int ntiles = (d.N + 31) / 32;
  const int tstride = gridDim.x * 4;
  // The next tile's input is fetched BEFORE this tile's products and stores are issued: memory operations retire in
  // order, so a fetch issued after the stores could only be consumed once those had drained (measured at 100k rows:
  // every round of tiles paid the fetch latency plus the store drain on top of its MFMA time).
  f32x16 raw[KB0];
  // GLUE: per-frame multipliers of this lane's feature slots (1 for the position code), loop invariant
  f32x16 mul[GLUE ? KB0 : 1];
  if (GLUE) {
#pragma unroll
    for (int b = 0; b < KB0; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = 32 * b + 8 * q + 4 * h + i;
          float m = 1.f;
          if (f >= GLUE_KX && f < GLUE_KX + GLUE_KA) m = gf.enc_a[f - GLUE_KX];
          else if (f >= GLUE_KX + GLUE_KA && f < GLUE_KX + GLUE_KA + GLUE_KE) m = gf.enc_e[f - GLUE_KX - GLUE_KA];
          mul[b][4 * q + i] = m;
        }
  }
  auto fetch = [&](int tile) {
    const size_t row = (size_t)tile * 32 + l31;
    const bool valid = row < (size_t)d.N;
    if (GLUE) {
#pragma unroll
      for (int b = 0; b < KB0; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f0 = 32 * b + 8 * q + 4 * h;
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (valid) {
            if (f0 + 4 <= GLUE_KX) {
              t = *reinterpret_cast<const float4*>(gf.enc_x + row * GLUE_KX + f0);
            } else if (f0 >= GLUE_KX && f0 + 4 <= GLUE_KX + GLUE_KA) {
              t = *reinterpret_cast<const float4*>(gf.aud + row * GLUE_KA + (f0 - GLUE_KX));
            } else if (f0 >= GLUE_KX + GLUE_KA && f0 < GLUE_KX + GLUE_KA + GLUE_KE) {
              const int k = f0 - GLUE_KX - GLUE_KA;
              const float* e = gf.eye_pre + row * GLUE_KE + k;
              t.x = e[0];
              if (k + 1 < GLUE_KE) t.y = e[1];
              if (k + 2 < GLUE_KE) t.z = e[2];
              if (k + 3 < GLUE_KE) t.w = e[3];
            }
          }
          raw[b][4 * q] = t.x; raw[b][4 * q + 1] = t.y; raw[b][4 * q + 2] = t.z; raw[b][4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
      for (int b = 0; b < KB0; ++b) load_block(X, row, valid, d.K0, b, h, raw[b]);
    }
  };
  int tile = blockIdx.x * 4 + wave;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += tstride) {
    const size_t row = (size_t)tile * 32 + l31;
    const bool valid = row < (size_t)d.N;
    f32x16 in0[KB0];
#pragma unroll
    for (int b = 0; b < KB0; ++b) in0[b] = raw[b];
    if (tile + tstride < ntiles) fetch(tile + tstride);
    if (GLUE) {
      constexpr int K0 = GLUE_KX + GLUE_KA + GLUE_KE;
      float sa = 0.f, se = 0.f;
#pragma unroll
      for (int b = 0; b < KB0; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f0 = 32 * b + 8 * q + 4 * h;
          float v[4] = {in0[b][4 * q], in0[b][4 * q + 1], in0[b][4 * q + 2], in0[b][4 * q + 3]};
          if (f0 >= GLUE_KX && f0 + 4 <= GLUE_KX + GLUE_KA) {
            sa += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= mul[b][4 * q + i];
          } else if (f0 >= GLUE_KX + GLUE_KA && f0 < K0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (f0 + i < K0) {
                const float act = fmaxf(v[i], 0.f);
                se += act * act;
                v[i] = mul[b][4 * q + i] * act;
              }
          }
          if (valid && f0 < K0 && gf.h_in != nullptr) {            // (K0 = 74 is even: 8-byte stores; null: forward only)
            float* hp = gf.h_in + row * K0 + f0;
            *reinterpret_cast<float2*>(hp) = make_float2(v[0], v[1]);
            if (f0 + 2 < K0) *reinterpret_cast<float2*>(hp + 2) = make_float2(v[2], v[3]);
          }
          in0[b][4 * q] = v[0]; in0[b][4 * q + 1] = v[1]; in0[b][4 * q + 2] = v[2]; in0[b][4 * q + 3] = v[3];
        }
      sa += __shfl_xor(sa, 32);            // the other half wave holds the row's other feature groups
      se += __shfl_xor(se, 32);
      if (valid && h == 0) { gf.amb[3 * row] = sqrtf(sa); gf.amb[3 * row + 1] = sqrtf(se); gf.amb[3 * row + 2] = 0.f; }
    }
    f32x16 h1[HB];
    layer_forward<KQ0, KB0, HB>(w1, in0, h1, l31, h);
    relu_blocks<HB>(h1);
    if (A1) {
      store_blocks<HB, 8 * HQ>(A1, row, valid, d.H, h, h1);
    }
    f32x16 out[1];
    if (NL == 3) {
      f32x16 h2[HB];
      layer_forward<HQ, HB, HB>(w2, h1, h2, l31, h);
      relu_blocks<HB>(h2);
      if (A2) {
        store_blocks<HB, 8 * HQ>(A2, row, valid, d.H, h, h2);
      }
      layer_forward<HQ, HB, 1>(w3, h2, out, l31, h);
    } else {
      layer_forward<HQ, HB, 1>(w2, h1, out, l31, h);
    }
    store_block(Y, row, valid, d.O, 0, h, out[0]);
  }
}

// GLUE: sigma_net's input is the glue operator's output cat(enc_x [KX], enc_a * aud [KA], enc_e * relu(eye_pre) [KE])
// (scene/motion_net.py:291-306).  Instead of storing dX [N, K0] for a second kernel to read back, split and multiply
// (motion_glue_backward_kernel: 35 us + a column-sum launch at 100k rows), the epilogue below writes d_enc_x, d_aud and
// d_eye_pre straight from the accumulator registers and keeps the per-frame vectors' column sums in registers across the
// wave's tiles (one row of per-workgroup partial sums at the end, added up in a fixed order by the caller).
struct GlueBwd {
  const float* aud; const float* eye_pre; const float* enc_a; const float* enc_e; const float* amb; const float* d_amb;
  float* d_enc_x; float* d_aud; float* d_eye; float* col_partials;
};

template <int KQ0, int HQ, int OQ, int NL, bool GLUE = false>
__global__ void __launch_bounds__(MLP_BLOCK) __attribute__((amdgpu_waves_per_eu(2)))
mlp_backward_kernel(MlpDims d, const float* __restrict__ dY, const float* __restrict__ A1,
                    const float* __restrict__ A2, const float* __restrict__ W1, const float* __restrict__ W2,
                    const float* __restrict__ W3, float* __restrict__ dZ1, float* __restrict__ dZ2,
                    float* dX, const float* dXadd /* [N,K0] added to the input gradient, may alias dX, or null */,
                    GlueBwd gl = GlueBwd{}) {
  extern __shared__ __align__(16) float s_w[];
  __shared__ float s_col[GLUE ? 4 : 1][GLUE ? 40 : 1];
  constexpr int KBG = (KQ0 + 3) / 4;
  float part[GLUE ? KBG : 1][4][4];          // column sums of gw * a over this lane's rows (glue groups only)
  if (GLUE) {
#pragma unroll
    for (int b = 0; b < KBG; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[b][q][i] = 0.f;
  }
  constexpr int KB0 = (KQ0 + 3) / 4, HB = (HQ + 3) / 4;
  constexpr int KP0 = KB0 * 32, HP = HB * 32;
  float* w1 = s_w;
  float* w2 = w1 + HP * (KP0 + 1);
  float* w3 = w2 + (NL == 3 ? HP : 32) * (HP + 1);
  if (dX || GLUE) stage_weights<HP, KB0>(w1, W1, d.H, d.K0);
  stage_weights<(NL == 3 ? HP : 32), HB>(w2, W2, NL == 3 ? d.H : d.O, d.H);
  if (NL == 3) stage_weights<32, HB>(w3, W3, d.O, d.H);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int ntiles = (d.N + 31) / 32;
  const int tstride = gridDim.x * 4;
  // next tile's dY and activations fetched before this tile's products and stores are issued (see mlp_forward_kernel)
  f32x16 ndy, nact2[NL == 3 ? HB : 1], nact1[HB];
  auto fetch = [&](int tile) {
    const size_t row = (size_t)tile * 32 + l31;
    const bool valid = row < (size_t)d.N;
    load_block(dY, row, valid, d.O, 0, h, ndy);
    if constexpr (NL == 3) {
      load_blocks<HB, 8 * HQ>(A2, row, valid, d.H, h, nact2);
    }
    load_blocks<HB, 8 * HQ>(A1, row, valid, d.H, h, nact1);
  };
  // (the GLUE variant's epilogue needs the registers: with the prefetch it spills)
  constexpr bool PREFETCH = !GLUE;
  int tile = blockIdx.x * 4 + wave;
  if (PREFETCH && tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += tstride) {
    const size_t row = (size_t)tile * 32 + l31;
    const bool valid = row < (size_t)d.N;
    if (!PREFETCH) fetch(tile);
    f32x16 dy[1], act2[NL == 3 ? HB : 1], act1[HB];
    dy[0] = ndy;
#pragma unroll
    for (int b = 0; b < HB; ++b) {
      act1[b] = nact1[b];
      if constexpr (NL == 3) act2[b] = nact2[b];
    }
    if (PREFETCH && tile + tstride < ntiles) fetch(tile + tstride);
    f32x16 g1[HB];
    if constexpr (NL == 3) {
      f32x16 g2[HB];
      layer_backward<OQ, 1, HB>(w3, dy, g2, l31, h);
      mask_blocks<HB>(g2, act2);
      store_blocks<HB, 8 * HQ>(dZ2, row, valid, d.H, h, g2);
      layer_backward<HQ, HB, HB>(w2, g2, g1, l31, h);
    } else {
      layer_backward<OQ, 1, HB>(w2, dy, g1, l31, h);
    }
    mask_blocks<HB>(g1, act1);
    store_blocks<HB, 8 * HQ>(dZ1, row, valid, d.H, h, g1);
    if (GLUE) {
      f32x16 gx[KB0];
      layer_backward<HQ, HB, KB0>(w1, g1, gx, l31, h);
      if (valid) {
        const float na = gl.amb[3 * row], ne = gl.amb[3 * row + 1];
        const float ga = (gl.d_amb && na > 0.f) ? gl.d_amb[3 * row] / na : 0.f;
        const float ge = (gl.d_amb && ne > 0.f) ? gl.d_amb[3 * row + 1] / ne : 0.f;
#pragma unroll
        for (int b = 0; b < KB0; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * b + 8 * q + 4 * h;              // first of this lane's four features of the group
            const float gw[4] = {gx[b][4 * q], gx[b][4 * q + 1], gx[b][4 * q + 2], gx[b][4 * q + 3]};
            if (f0 + 4 <= GLUE_KX) {
              *reinterpret_cast<float4*>(gl.d_enc_x + row * GLUE_KX + f0) = make_float4(gw[0], gw[1], gw[2], gw[3]);
            } else if (f0 >= GLUE_KX && f0 + 4 <= GLUE_KX + GLUE_KA) {
              const int k = f0 - GLUE_KX;
              const float4 a = *reinterpret_cast<const float4*>(gl.aud + row * GLUE_KA + k);
              const float4 ea = *reinterpret_cast<const float4*>(gl.enc_a + k);
              *reinterpret_cast<float4*>(gl.d_aud + row * GLUE_KA + k) =
                  make_float4(ea.x * gw[0] + ga * a.x, ea.y * gw[1] + ga * a.y, ea.z * gw[2] + ga * a.z,
                              ea.w * gw[3] + ga * a.w);
              part[b][q][0] += gw[0] * a.x; part[b][q][1] += gw[1] * a.y;
              part[b][q][2] += gw[2] * a.z; part[b][q][3] += gw[3] * a.w;
            } else if (f0 >= GLUE_KX + GLUE_KA && f0 < GLUE_KX + GLUE_KA + GLUE_KE) {
              const int k = f0 - GLUE_KX - GLUE_KA;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (k + i < GLUE_KE) {
                  const float pre = gl.eye_pre[row * GLUE_KE + k + i];
                  const float act = fmaxf(pre, 0.f);
                  gl.d_eye[row * GLUE_KE + k + i] = pre > 0.f ? (gl.enc_e[k + i] * gw[i] + ge * act) : 0.f;
                  part[b][q][i] += gw[i] * act;
                }
              }
            }
          }
      }
    } else if (dX) {
      f32x16 gx[KB0];
      layer_backward<HQ, HB, KB0>(w1, g1, gx, l31, h);
      if (dXadd) {
        // the input has other consumers: their gradient is summed here instead of by a separate elementwise launch
        // (every element is read and written by the same lane, so dXadd may be dX itself)
#pragma unroll
        for (int b = 0; b < KB0; ++b) {
          f32x16 prev;
          load_block(dXadd, row, valid, d.K0, b, h, prev);
          gx[b] += prev;
        }
      }
#pragma unroll
      for (int b = 0; b < KB0; ++b) store_block(dX, row, valid, d.K0, b, h, gx[b]);
    }
  }
  if (GLUE) {
    // column sums: over the 32 rows of the half wave (xor steps below 32 stay inside it), then over the four waves in
    // order, one row of partial sums per workgroup
    for (int i = threadIdx.x; i < 4 * 40; i += MLP_BLOCK) (&s_col[0][0])[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < KBG; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * b + 8 * q + 4 * h;
        if (f0 >= GLUE_KX && f0 < GLUE_KX + GLUE_KA + GLUE_KE) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = part[b][q][i];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (l31 == 0 && f0 + i < GLUE_KX + GLUE_KA + GLUE_KE) s_col[wave][f0 + i - GLUE_KX] = v;
          }
        }
      }
    __syncthreads();
    if ((int)threadIdx.x < GLUE_KA + GLUE_KE)
      gl.col_partials[(size_t)blockIdx.x * (GLUE_KA + GLUE_KE) + threadIdx.x] =
          ((s_col[0][threadIdx.x] + s_col[1][threadIdx.x]) + s_col[2][threadIdx.x]) + s_col[3][threadIdx.x];
  }
}


// ---- two 2-layer MLPs over the SAME input in one launch ------------------------------------------------------------
// aud_ch_att_net and eye_att_net of the universal field both read the tri-plane features (scene/motion_net.py:281-290).
// As two launches each pays its own weight staging, tile-load latency and launch tail (19 + 12 us forward, 28 + 17 us
// backward at 100k rows, for 0.6 GFLOP); here a wave loads its 32-row tile once and runs both heads on it, and the
// backward adds the two input gradients (and the gradient of the input's other consumers) in registers.
struct Mlp2Dims { int N, K0, HA, OA, HB, OB; };

template <int KQ0, int HQA, int HQB>
size_t mlp2_lds_bytes() {
  constexpr int KP0 = (KQ0 + 3) / 4 * 32, HPA = (HQA + 3) / 4 * 32, HPB = (HQB + 3) / 4 * 32;
  return sizeof(float) * (HPA * (KP0 + 1) + 32 * (HPA + 1) + HPB * (KP0 + 1) + 32 * (HPB + 1));
}

template <int KQ0, int HQA, int OQA, int HQB, int OQB>
__global__ void __launch_bounds__(MLP_BLOCK)
mlp2_forward_kernel(Mlp2Dims d, const float* __restrict__ X, const float* __restrict__ WA1,
                    const float* __restrict__ WA2, const float* __restrict__ WB1, const float* __restrict__ WB2,
                    float* __restrict__ YA, float* __restrict__ YB, float* __restrict__ A1A, float* __restrict__ A1B) {
  extern __shared__ __align__(16) float s_w[];
  constexpr int KB0 = (KQ0 + 3) / 4, HBA = (HQA + 3) / 4, HBB = (HQB + 3) / 4;
  constexpr int KP0 = KB0 * 32, HPA = HBA * 32, HPB = HBB * 32;
  float* wa1 = s_w;
  float* wa2 = wa1 + HPA * (KP0 + 1);
  float* wb1 = wa2 + 32 * (HPA + 1);
  float* wb2 = wb1 + HPB * (KP0 + 1);
  stage_weights<HPA, KB0>(wa1, WA1, d.HA, d.K0);
  stage_weights<32, HBA>(wa2, WA2, d.OA, d.HA);
  stage_weights<HPB, KB0>(wb1, WB1, d.HB, d.K0);
  stage_weights<32, HBB>(wb2, WB2, d.OB, d.HB);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int ntiles = (d.N + 31) / 32;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const size_t row = (size_t)tile * 32 + l31;
    const bool valid = row < (size_t)d.N;
    f32x16 in0[KB0];
#pragma unroll
    for (int b = 0; b < KB0; ++b) load_block(X, row, valid, d.K0, b, h, in0[b]);
    {
      f32x16 h1[HBA], out[1];
      layer_forward<KQ0, KB0, HBA>(wa1, in0, h1, l31, h);
      relu_blocks<HBA>(h1);
      if (A1A) store_blocks<HBA, 8 * HQA>(A1A, row, valid, d.HA, h, h1);      // (null: forward only, nothing kept)
      layer_forward<HQA, HBA, 1>(wa2, h1, out, l31, h);
      store_block(YA, row, valid, d.OA, 0, h, out[0]);
    }
    {
      f32x16 h1[HBB], out[1];
      layer_forward<KQ0, KB0, HBB>(wb1, in0, h1, l31, h);
      relu_blocks<HBB>(h1);
      if (A1B) store_blocks<HBB, 8 * HQB>(A1B, row, valid, d.HB, h, h1);
      layer_forward<HQB, HBB, 1>(wb2, h1, out, l31, h);
      store_block(YB, row, valid, d.OB, 0, h, out[0]);
    }
  }
}

template <int KQ0, int HQA, int OQA, int HQB, int OQB>
__global__ void __launch_bounds__(MLP_BLOCK)
mlp2_backward_kernel(Mlp2Dims d, const float* __restrict__ dYA, const float* __restrict__ dYB,
                     const float* __restrict__ A1A, const float* __restrict__ A1B, const float* __restrict__ WA1,
                     const float* __restrict__ WA2, const float* __restrict__ WB1, const float* __restrict__ WB2,
                     float* __restrict__ dZ1A, float* __restrict__ dZ1B, float* dX,
                     const float* dXadd /* [N,K0] added to the input gradient, may alias dX, or null */) {
  extern __shared__ __align__(16) float s_w[];
  constexpr int KB0 = (KQ0 + 3) / 4, HBA = (HQA + 3) / 4, HBB = (HQB + 3) / 4;
  constexpr int KP0 = KB0 * 32, HPA = HBA * 32, HPB = HBB * 32;
  float* wa1 = s_w;
  float* wa2 = wa1 + HPA * (KP0 + 1);
  float* wb1 = wa2 + 32 * (HPA + 1);
  float* wb2 = wb1 + HPB * (KP0 + 1);
  if (dX) { stage_weights<HPA, KB0>(wa1, WA1, d.HA, d.K0); stage_weights<HPB, KB0>(wb1, WB1, d.HB, d.K0); }
  stage_weights<32, HBA>(wa2, WA2, d.OA, d.HA);
  stage_weights<32, HBB>(wb2, WB2, d.OB, d.HB);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int ntiles = (d.N + 31) / 32;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const size_t row = (size_t)tile * 32 + l31;
    const bool valid = row < (size_t)d.N;
    f32x16 gx[KB0];
#pragma unroll
    for (int b = 0; b < KB0; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) gx[b][i] = 0.f;
    {
      f32x16 dy[1], g1[HBA], act[HBA];
      load_block(dYA, row, valid, d.OA, 0, h, dy[0]);
      layer_backward<OQA, 1, HBA>(wa2, dy, g1, l31, h);
      load_blocks<HBA, 8 * HQA>(A1A, row, valid, d.HA, h, act);
      mask_blocks<HBA>(g1, act);
      store_blocks<HBA, 8 * HQA>(dZ1A, row, valid, d.HA, h, g1);
      if (dX) layer_backward<HQA, HBA, KB0>(wa1, g1, gx, l31, h);
    }
    {
      f32x16 dy[1], g1[HBB], act[HBB], gb[KB0];
      load_block(dYB, row, valid, d.OB, 0, h, dy[0]);
      layer_backward<OQB, 1, HBB>(wb2, dy, g1, l31, h);
      load_blocks<HBB, 8 * HQB>(A1B, row, valid, d.HB, h, act);
      mask_blocks<HBB>(g1, act);
      store_blocks<HBB, 8 * HQB>(dZ1B, row, valid, d.HB, h, g1);
      if (dX) {
        layer_backward<HQB, HBB, KB0>(wb1, g1, gb, l31, h);
#pragma unroll
        for (int b = 0; b < KB0; ++b) gx[b] += gb[b];
      }
    }
    if (dX) {
      if (dXadd) {
#pragma unroll
        for (int b = 0; b < KB0; ++b) {
          f32x16 prev;
          load_block(dXadd, row, valid, d.K0, b, h, prev);
          gx[b] += prev;
        }
      }
#pragma unroll
      for (int b = 0; b < KB0; ++b) store_block(dX, row, valid, d.K0, b, h, gx[b]);
    }
  }
}

// ---- dW[o][k] = sum_rows dZ[row][o] * In[row][k] ---------------------------------------------------
// A workgroup owns a contiguous range of rows and produces one partial [O][K] matrix.  Round 3: the OB x KB output
// tiles (32 x 32 each) are DEALT OUT to the four waves instead of every wave holding all of them over a quarter of the
// rows: a wave keeps at most two accumulator tiles (32 AGPRs instead of up to 96) and fetches only the operand columns
// of its tiles, so the kernel needs 120 registers instead of 231 (4 waves per SIMD instead of 2) and 12 KB of LDS instead
// of 24, and jobs of four or more tiles need no cross-wave combine at all.  (It did not make the launch faster -- see the
// A/B at weight_grad_batched_kernel -- but it takes half the register file and LDS from the kernels it runs beside.)  The waves of a workgroup read the same rows at the same time, so the
// operand columns two of them share come from L1.  Jobs of fewer than four tiles split the workgroup's rows between
// wave groups as before and combine through LDS in a fixed order (bitwise reproducible either way).
// VIRT: the input rows are not stored anywhere -- row r of `In` is cat(xa[r] (Ka), xb[r] * mb (Kb), relu(xc[r]) * mc (Kc)),
// sigma_net's input as the glue forms it (scene/motion_net.py:291-306); the forward then does not write those 30 MB.
struct WgVirt {
  int job;                                   // index of the job whose input is virtual, -1: none
  const float *xb, *xc, *mb, *mc;            // xa = the job's `in` pointer
  int Ka, Kb, Kc, pad;
};

template <int OB, int KB, bool VIRT = false>
__device__ __forceinline__ void weight_grad_body(const float* __restrict__ dZ, const float* __restrict__ In, int N, int O,
                                                 int K, float* __restrict__ partial, float* s_acc, int block,
                                                 int nblocks, const WgVirt* virt = nullptr) {
  constexpr int TILES = OB * KB;
  constexpr int TW = TILES >= 4 ? 4 : TILES;           // waves holding distinct tile sets
  constexpr int RS = 4 / TW;                           // row splits inside the workgroup (3 tiles: one wave idles)
  constexpr int MYT = (TILES + TW - 1) / TW;           // tiles per wave at most: 6 -> 2, else 1
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int tw = wave % TW, rs = wave / TW;
  const bool busy = wave < TW * RS;
  // rows of this workgroup, then of this wave's row split (even counts: a k-step is a row pair)
  const int per_block = (((N + nblocks - 1) / nblocks) + 1) & ~1;
  const long b0 = (long)block * per_block, b1 = min((long)N, b0 + per_block);
  const int per_split = ((((int)max(0l, b1 - b0) + RS - 1) / RS) + 1) & ~1;
  const long w0 = b0 + (long)rs * per_split;
  const long w1 = busy ? min(b1, w0 + per_split) : w0;
  int tt[MYT], tb[MYT];
  bool live[MYT], oin[MYT], kin[MYT];
#pragma unroll
  for (int j = 0; j < MYT; ++j) {
    const int idx = tw + j * TW;
    live[j] = idx < TILES;
    tt[j] = live[j] ? idx / KB : 0;
    tb[j] = live[j] ? idx % KB : 0;
    oin[j] = live[j] && 32 * tt[j] + l31 < O;
    kin[j] = live[j] && 32 * tb[j] + l31 < K;
  }
  // this lane's input column of every tile it holds: where it lives (VIRT: in which of the three pieces)
  const float* bp[MYT];
  int bs[MYT];
  float bm[MYT];
  bool brelu[MYT];
#pragma unroll
  for (int j = 0; j < MYT; ++j) {
    const int col = 32 * tb[j] + l31;
    bp[j] = In + col; bs[j] = K; bm[j] = 1.f; brelu[j] = false;
    if (VIRT && kin[j]) {
      if (col < virt->Ka) { bp[j] = In + col; bs[j] = virt->Ka; }
      else if (col < virt->Ka + virt->Kb) {
        bp[j] = virt->xb + (col - virt->Ka); bs[j] = virt->Kb; bm[j] = virt->mb[col - virt->Ka];
      } else {
        const int c2 = col - virt->Ka - virt->Kb;
        bp[j] = virt->xc + c2; bs[j] = virt->Kc; bm[j] = virt->mc[c2]; brelu[j] = true;
      }
    }
  }
  f32x16 acc[MYT];
#pragma unroll
  for (int j = 0; j < MYT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
#ifndef INSTAG_WG_UNROLL
#define INSTAG_WG_UNROLL 6
#endif
  constexpr int UNROLL = INSTAG_WG_UNROLL;             // 2 * UNROLL rows of both operands in flight per wave
  for (long r0 = w0; r0 < w1; r0 += 2 * UNROLL) {
    float a[UNROLL][MYT], bb[UNROLL][MYT];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long row = r0 + 2 * u + h;
      const bool ok = row < w1;
#pragma unroll
      for (int j = 0; j < MYT; ++j) {
        a[u][j] = (ok && oin[j]) ? dZ[row * O + 32 * tt[j] + l31] : 0.f;
        float v = (ok && kin[j]) ? bp[j][row * bs[j]] : 0.f;
        if (VIRT) {
          v = brelu[j] ? fmaxf(v, 0.f) : v;
          v *= bm[j];
        }
        bb[u][j] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int j = 0; j < MYT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], bb[u][j], acc[j], 0, 0, 0);
  }
  float* dst = partial + (size_t)block * O * K;
  if (RS == 1) {
    // every tile belongs to exactly one wave: straight from the accumulators (128-byte row segments)
#pragma unroll
    for (int j = 0; j < MYT; ++j) {
      if (!live[j] || !busy) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = 32 * tt[j] + feat(i, h), k = 32 * tb[j] + l31;
        if (o < O && k < K) dst[o * K + k] = acc[j][i];
      }
    }
    return;
  }
  // TILES < 4 (MYT == 1): combine the RS row splits of every tile in a fixed order, then store
  for (int r = 0; r < RS; ++r) {
    if (busy && rs == r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* p = &s_acc[(tw * 16 + i) * 64 + lane];
        *p = (r == 0) ? acc[0][i] : (*p + acc[0][i]);
      }
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < TILES * 1024; idx += MLP_BLOCK) {
    const int ln = idx & 63, i = (idx >> 6) & 15, tile = idx >> 10;
    const int t = tile / KB, b = tile - t * KB;
    const int o = 32 * t + feat(i, ln >> 5), k = 32 * b + (ln & 31);
    if (o < O && k < K) dst[o * K + k] = s_acc[idx];
  }
}

template <int OB, int KB>
__global__ void __launch_bounds__(MLP_BLOCK)
weight_grad_kernel(const float* __restrict__ dZ, const float* __restrict__ In, int N, int O, int K,
                   float* __restrict__ partial) {
  __shared__ float s_acc[(OB * KB < 4 ? OB * KB : 1) * 1024];
  weight_grad_body<OB, KB>(dZ, In, N, O, K, partial, s_acc, blockIdx.x, gridDim.x);
}

// Several weight-gradient GEMMs in ONE launch (blockIdx.y = job): a train step has nine of them (three MLPs of the
// universal field, one of the personalised field), each too small to fill the chip on its own and none needed before
// the optimizer runs.
constexpr int WG_MAX_JOBS = 16;
struct WgJob { const float* dz; const float* in; float* partial; float* dw; int N, O, K, pad; };
struct WgBatch { WgJob j[WG_MAX_JOBS]; WgVirt virt; };

// (same-box A/B of the C3 step, round 3: 3 waves per SIMD x 16 rows in flight 0.9000 ms, 4 x 12 0.8989, 5 x 8 0.9014 --
// and 0.9005 for the round-2 kernel that held every tile in every wave at 2 waves per SIMD: the launch is not bound by
// its own occupancy; kept for the smaller footprint)
#ifndef INSTAG_WG_WAVES
#define INSTAG_WG_WAVES 4
#endif
__global__ void __launch_bounds__(MLP_BLOCK) __attribute__((amdgpu_waves_per_eu(INSTAG_WG_WAVES)))
weight_grad_batched_kernel(WgBatch b) {
  __shared__ float s_acc[3 * 1024];          // only jobs of fewer than four tiles combine through LDS
  const WgJob job = b.j[blockIdx.y];
  const int ob = (job.O + 31) / 32, kb = (job.K + 31) / 32;
  if ((int)blockIdx.y == b.virt.job) {           // (host side: only shapes with kb == 3 are accepted)
    if (ob == 1) weight_grad_body<1, 3, true>(job.dz, job.in, job.N, job.O, job.K, job.partial, s_acc, blockIdx.x, gridDim.x, &b.virt);
    else weight_grad_body<2, 3, true>(job.dz, job.in, job.N, job.O, job.K, job.partial, s_acc, blockIdx.x, gridDim.x, &b.virt);
    return;
  }
  if (ob == 1 && kb == 1) weight_grad_body<1, 1>(job.dz, job.in, job.N, job.O, job.K, job.partial, s_acc, blockIdx.x, gridDim.x);
  else if (ob == 1 && kb == 2) weight_grad_body<1, 2>(job.dz, job.in, job.N, job.O, job.K, job.partial, s_acc, blockIdx.x, gridDim.x);
  else if (ob == 1 && kb == 3) weight_grad_body<1, 3>(job.dz, job.in, job.N, job.O, job.K, job.partial, s_acc, blockIdx.x, gridDim.x);
  else if (ob == 2 && kb == 1) weight_grad_body<2, 1>(job.dz, job.in, job.N, job.O, job.K, job.partial, s_acc, blockIdx.x, gridDim.x);
  else if (ob == 2 && kb == 2) weight_grad_body<2, 2>(job.dz, job.in, job.N, job.O, job.K, job.partial, s_acc, blockIdx.x, gridDim.x);
  else weight_grad_body<2, 3>(job.dz, job.in, job.N, job.O, job.K, job.partial, s_acc, blockIdx.x, gridDim.x);
}

__global__ void __launch_bounds__(256)
weight_grad_reduce_batched_kernel(WgBatch b, int nparts) {
  __shared__ float s_part[4][64];
  const WgJob job = b.j[blockIdx.y];
  const int count = job.O * job.K;
  if ((int)blockIdx.x * 64 >= count) return;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < count) {
    const int per = (nparts + 3) / 4;
    const int p0 = g * per, p1 = min(nparts, p0 + per);
#pragma unroll 8
    for (int p = p0; p < p1; ++p) s += job.partial[(size_t)p * count + i];
  }
  s_part[g][e] = s;
  __syncthreads();
  if (g == 0 && i < count) job.dw[i] = ((s_part[0][e] + s_part[1][e]) + s_part[2][e]) + s_part[3][e];
}

// dW[i] = sum_p partial[p][i]: 64 elements x 4 partial-groups per workgroup, fixed summation order
__global__ void __launch_bounds__(256)
weight_grad_reduce_kernel(const float* __restrict__ partial, int nparts, int count, float* __restrict__ dW) {
  __shared__ float s_part[4][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < count) {
    const int per = (nparts + 3) / 4;
    const int p0 = g * per, p1 = min(nparts, p0 + per);
#pragma unroll 8
    for (int p = p0; p < p1; ++p) s += partial[(size_t)p * count + i];
  }
  s_part[g][e] = s;
  __syncthreads();
  if (g == 0 && i < count) dW[i] = ((s_part[0][e] + s_part[1][e]) + s_part[2][e]) + s_part[3][e];
}

// (256 workgroups per GEMM is the measured optimum, round 2: 128 -> +33 % per GEMM, 512 -> +5 %; round 3 in the step:
// 128 / 64 workgroups -- to leave the personalised field's backward, which runs beside the GEMMs, more of the chip --
// 0.907 / 0.925 ms per step against 0.908-0.915 with 256)
inline int wg_blocks(int N) { return std::max(1, std::min(256, (N + 255) / 256)); }

// Two workgroups per CU, each staging the weights once and walking a strided list of 32-row tiles.  (768 and 1024
// workgroups -- one tile per wave at 100k rows -- measured the same or slower: sigma_net forward 44.9 / 44.6 / 47.8 us.)
// Where the sigma-net forward's 39 us at 100k rows go (scripts/mlp_probe.py, scripts/probes/mfma_rate_probe.hip):
// one v_mfma_f32_32x32x2_f32 per 64 cycles and SIMD at ~2.1 GHz = 5.35 us per tile; 3125 tiles on 1024 SIMDs are 3.05
// per SIMD but some SIMD runs 4 (100,000 rows cost what 131,072 do); with the global loads and stores compiled out
// the kernel takes 28.7 us (6 us fixed + 4 x 5.5), with them 38-39: the 85 MB it moves would take ~15 us at the
// achievable HBM rate, as long as the MFMA work itself, and the two overlap only partly.  One 512-thread workgroup per
// CU drawing tiles from an LDS counter, with the second wave of each SIMD at lower issue priority (s_setprio), measured
// the same (38.4 us) and was slower for 32k rows (22 vs 16 us): not kept.
inline int mlp_blocks(int ntiles) {
  return std::max(1, std::min(512, (ntiles + 3) / 4));
}

template <int KB0, int HB, int NL>
size_t mlp_lds_bytes() {
  constexpr int KP0 = KB0 * 32, HP = HB * 32;
  return sizeof(float) * (HP * (KP0 + 1) + (NL == 3 ? HP : 32) * (HP + 1) + (NL == 3 ? 32 * (HP + 1) : 0));
}

template <int KQ0, int HQ, int OQ, int NL>
int run_fwd(const MlpDims& d, const float* x, const float* w1, const float* w2, const float* w3, float* y, float* a1,
            float* a2, hipStream_t s) {
  const int ntiles = (d.N + 31) / 32;
  const int blocks = mlp_blocks(ntiles);
  ProfScope p(K_MLP_FWD, s);
  mlp_forward_kernel<KQ0, HQ, OQ, NL><<<blocks, MLP_BLOCK, mlp_lds_bytes<(KQ0 + 3) / 4, (HQ + 3) / 4, NL>(), s>>>(
      d, x, w1, w2, w3, y, a1, a2);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}
template <int KQ0, int HQ, int OQ, int NL>
int run_bwd(const MlpDims& d, const float* dy, const float* a1, const float* a2, const float* w1, const float* w2,
            const float* w3, float* dz1, float* dz2, float* dx, const float* dx_add, hipStream_t s) {
  const int ntiles = (d.N + 31) / 32;
  const int blocks = mlp_blocks(ntiles);
  ProfScope p(K_MLP_BWD, s);
  mlp_backward_kernel<KQ0, HQ, OQ, NL><<<blocks, MLP_BLOCK, mlp_lds_bytes<(KQ0 + 3) / 4, (HQ + 3) / 4, NL>(), s>>>(
      d, dy, a1, a2, w1, w2, w3, dz1, dz2, dx, dx_add);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}
template <int OB, int KB>
int run_wg(const float* dz, const float* in, int N, int O, int K, float* partial, float* dw, hipStream_t s) {
  const int blocks = wg_blocks(N);
  ProfScope p(K_MLP_WGRAD, s);
  weight_grad_kernel<OB, KB><<<blocks, MLP_BLOCK, 0, s>>>(dz, in, N, O, K, partial);
  INSTAG_CHECK_LAUNCH();
  weight_grad_reduce_kernel<<<(O * K + 63) / 64, 256, 0, s>>>(partial, blocks, O * K, dw);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

// Exact 8-feature-group counts for the widths the motion networks use (74 / 36 inputs, 64 / 32 / 16 hidden,
// 6 / 7 / 11 / 32 outputs); any other shape runs on whole 32-feature blocks (zero-padded weights).
#define MLP_DISPATCH(FN, ...)                                                                      \
  do {                                                                                             \
    const int kq = (d.K0 + 7) / 8, hq = (d.H + 7) / 8, oq = (d.O + 7) / 8;                          \
    if (NL == 3) {                                                                                 \
      if (kq == 10 && hq == 8 && oq == 2) return FN<10, 8, 2, 3>(__VA_ARGS__);                      \
      if (kq == 10 && hq == 4 && oq == 2) return FN<10, 4, 2, 3>(__VA_ARGS__);                      \
      if (kq == 10 && hq == 2 && oq == 1) return FN<10, 2, 1, 3>(__VA_ARGS__);                      \
    } else {                                                                                       \
      if (kq == 5 && hq == 4 && oq == 4) return FN<5, 4, 4, 2>(__VA_ARGS__);                        \
      if (kq == 5 && hq == 4 && oq == 1) return FN<5, 4, 1, 2>(__VA_ARGS__);                        \
      if (kq == 5 && hq == 2 && oq == 1) return FN<5, 2, 1, 2>(__VA_ARGS__);                        \
    }                                                                                              \
    const int kb = (d.K0 + 31) / 32, hb = (d.H + 31) / 32;                                           \
    if (NL == 2) {                                                                                 \
      if (kb == 1 && hb == 1) return FN<4, 4, 4, 2>(__VA_ARGS__);                                   \
      if (kb == 2 && hb == 1) return FN<8, 4, 4, 2>(__VA_ARGS__);                                   \
      if (kb == 3 && hb == 1) return FN<12, 4, 4, 2>(__VA_ARGS__);                                  \
      if (kb == 1 && hb == 2) return FN<4, 8, 4, 2>(__VA_ARGS__);                                   \
      if (kb == 2 && hb == 2) return FN<8, 8, 4, 2>(__VA_ARGS__);                                   \
      if (kb == 3 && hb == 2) return FN<12, 8, 4, 2>(__VA_ARGS__);                                  \
    } else {                                                                                       \
      if (kb == 1 && hb == 1) return FN<4, 4, 4, 3>(__VA_ARGS__);                                   \
      if (kb == 2 && hb == 1) return FN<8, 4, 4, 3>(__VA_ARGS__);                                   \
      if (kb == 3 && hb == 1) return FN<12, 4, 4, 3>(__VA_ARGS__);                                  \
      if (kb == 1 && hb == 2) return FN<4, 8, 4, 3>(__VA_ARGS__);                                   \
      if (kb == 2 && hb == 2) return FN<8, 8, 4, 3>(__VA_ARGS__);                                   \
      if (kb == 3 && hb == 2) return FN<12, 8, 4, 3>(__VA_ARGS__);                                  \
    }                                                                                              \
  } while (0)

int check_dims(int N, int K0, int H, int O, int NL) {
  INSTAG_REQUIRE(N >= 0, "mlp: N must be >= 0");
  INSTAG_REQUIRE(NL == 2 || NL == 3, "mlp: only 2- or 3-layer MLPs are supported");
  INSTAG_REQUIRE(K0 >= 1 && K0 <= 96, "mlp: input width must be in [1,96]");
  INSTAG_REQUIRE(H >= 1 && H <= 64, "mlp: hidden width must be in [1,64]");
  INSTAG_REQUIRE(O >= 1 && O <= 32, "mlp: output width must be in [1,32]");
  return INSTAG_OK;
}

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

int instag_mlp_forward(const float* x, const float* w1, const float* w2, const float* w3, float* y, float* a1,
                       float* a2, int32_t N, int32_t K0, int32_t H, int32_t O, int32_t NL, instag_stream_t stream) {
  if (int e = check_dims(N, K0, H, O, NL)) return e;
  INSTAG_REQUIRE(x && w1 && w2 && y && (NL == 2 || w3), "mlp_forward: NULL tensor");
  if (N == 0) return INSTAG_OK;
  const MlpDims d{N, K0, H, O};
  hipStream_t s = (hipStream_t)stream;
  MLP_DISPATCH(run_fwd, d, x, w1, w2, w3, y, a1, a2, s);
  set_error("mlp_forward: unsupported shape");
  return INSTAG_E_ARG;
}

int instag_mlp_backward_add(const float* dy, const float* a1, const float* a2, const float* w1, const float* w2,
                            const float* w3, float* dz1, float* dz2, float* dx, const float* dx_add, int32_t N,
                            int32_t K0, int32_t H, int32_t O, int32_t NL, instag_stream_t stream) {
  if (int e = check_dims(N, K0, H, O, NL)) return e;
  INSTAG_REQUIRE(dy && a1 && w1 && w2 && dz1 && (NL == 2 || (w3 && a2 && dz2)), "mlp_backward: NULL tensor");
  INSTAG_REQUIRE(dx_add == nullptr || dx != nullptr, "mlp_backward: dx_add needs dx");
  if (N == 0) return INSTAG_OK;
  const MlpDims d{N, K0, H, O};
  hipStream_t s = (hipStream_t)stream;
  MLP_DISPATCH(run_bwd, d, dy, a1, a2, w1, w2, w3, dz1, dz2, dx, dx_add, s);
  set_error("mlp_backward: unsupported shape");
  return INSTAG_E_ARG;
}

int instag_mlp_backward(const float* dy, const float* a1, const float* a2, const float* w1, const float* w2,
                        const float* w3, float* dz1, float* dz2, float* dx, int32_t N, int32_t K0, int32_t H,
                        int32_t O, int32_t NL, instag_stream_t stream) {
  return instag_mlp_backward_add(dy, a1, a2, w1, w2, w3, dz1, dz2, dx, nullptr, N, K0, H, O, NL, stream);
}

/* sigma_net's backward with the glue operator's backward as its epilogue (universal field: K0 = 36 + 32 + 6, H = 64 or 32,
 * O <= 16): writes dz1, dz2 (for the weight gradients), d_enc_x [N,36], d_aud [N,32], d_eye_pre [N,6] and one row of
 * column partial sums [KA + KE] per workgroup (instag_mlp_backward_glue_num_partials rows, to be added up in order) --
 * the [N,74] input gradient is never stored.  d_amb [N,3] may be NULL. */
int instag_mlp_backward_glue_supported(int32_t K0, int32_t H, int32_t O, int32_t KX, int32_t KA, int32_t KE) {
  const int kq = (K0 + 7) / 8, hq = (H + 7) / 8, oq = (O + 7) / 8;
  return K0 == KX + KA + KE && KX == GLUE_KX && KA == GLUE_KA && KE == GLUE_KE && kq == 10 && oq == 2 &&
         (hq == 8 || hq == 4);
}

int instag_mlp_backward_glue_num_partials(int32_t N) { return mlp_blocks((N + 31) / 32); }

int instag_mlp_backward_glue(const float* dy, const float* a1, const float* a2, const float* w1, const float* w2,
                             const float* w3, float* dz1, float* dz2, const float* aud, const float* eye_pre,
                             const float* enc_a, const float* enc_e, const float* amb, const float* d_amb,
                             float* d_enc_x, float* d_aud, float* d_eye_pre, float* col_partials, int32_t N,
                             int32_t H, int32_t O, instag_stream_t stream) {
  const int K0 = GLUE_KX + GLUE_KA + GLUE_KE;
  INSTAG_REQUIRE(instag_mlp_backward_glue_supported(K0, H, O, GLUE_KX, GLUE_KA, GLUE_KE), "mlp_backward_glue: unsupported shape");
  INSTAG_REQUIRE(dy && a1 && a2 && w1 && w2 && w3 && dz1 && dz2 && aud && eye_pre && enc_a && enc_e && amb && d_enc_x &&
                     d_aud && d_eye_pre && col_partials, "mlp_backward_glue: NULL tensor");
  if (N <= 0) return INSTAG_OK;
  const MlpDims d{N, K0, H, O};
  const GlueBwd gl{aud, eye_pre, enc_a, enc_e, amb, d_amb, d_enc_x, d_aud, d_eye_pre, col_partials};
  hipStream_t s = (hipStream_t)stream;
  const int blocks = mlp_blocks((N + 31) / 32);
  ProfScope p(K_MLP_BWD, s);
  if ((H + 7) / 8 == 8)
    mlp_backward_kernel<10, 8, 2, 3, true><<<blocks, MLP_BLOCK, mlp_lds_bytes<3, 2, 3>(), s>>>(
        d, dy, a1, a2, w1, w2, w3, dz1, dz2, nullptr, nullptr, gl);
  else
    mlp_backward_kernel<10, 4, 2, 3, true><<<blocks, MLP_BLOCK, mlp_lds_bytes<3, 1, 3>(), s>>>(
        d, dy, a1, a2, w1, w2, w3, dz1, dz2, nullptr, nullptr, gl);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

/* sigma_net's forward with the glue operator's forward in front of its first layer (same shapes as
 * instag_mlp_backward_glue): reads enc_x [N,36], aud [N,32], eye_pre [N,6], enc_a [32], enc_e [6]; writes y [N,O], the
 * saved activations a1, a2 [N,H], the assembled input h_in [N,74] and amb [N,3]. */
int instag_mlp_forward_glue(const float* enc_x, const float* aud, const float* eye_pre, const float* enc_a,
                            const float* enc_e, const float* w1, const float* w2, const float* w3, float* y, float* a1,
                            float* a2, float* h_in, float* amb, int32_t N, int32_t H, int32_t O,
                            instag_stream_t stream) {
  const int K0 = GLUE_KX + GLUE_KA + GLUE_KE;
  INSTAG_REQUIRE(instag_mlp_backward_glue_supported(K0, H, O, GLUE_KX, GLUE_KA, GLUE_KE), "mlp_forward_glue: unsupported shape");
  INSTAG_REQUIRE(enc_x && aud && eye_pre && enc_a && enc_e && w1 && w2 && w3 && y && amb, "mlp_forward_glue: NULL tensor");
  INSTAG_REQUIRE((a1 == nullptr) == (a2 == nullptr) && (h_in == nullptr || a1 != nullptr),
                 "mlp_forward_glue: a1 and a2 go together (NULL: forward only); h_in (may be NULL: the caller's weight "
                 "gradient assembles the rows itself, instag_linear_weight_grad_batched_glue) needs them");
  if (N <= 0) return INSTAG_OK;
  const MlpDims d{N, K0, H, O};
  const GlueFwd gf{enc_x, aud, eye_pre, enc_a, enc_e, h_in, amb};
  hipStream_t s = (hipStream_t)stream;
  const int blocks = mlp_blocks((N + 31) / 32);
  ProfScope p(K_MLP_FWD, s);
  if ((H + 7) / 8 == 8)
    mlp_forward_kernel<10, 8, 2, 3, true><<<blocks, MLP_BLOCK, mlp_lds_bytes<3, 2, 3>(), s>>>(
        d, nullptr, w1, w2, w3, y, a1, a2, gf);
  else
    mlp_forward_kernel<10, 4, 2, 3, true><<<blocks, MLP_BLOCK, mlp_lds_bytes<3, 1, 3>(), s>>>(
        d, nullptr, w1, w2, w3, y, a1, a2, gf);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

/* Two 2-layer MLPs over the same input in one launch (see mlp2_forward_kernel).  Shapes: the universal field's
 * attention heads (36 -> 32 -> 32 and 36 -> 16 -> 6 in groups of eight features); instag_mlp2_supported says whether a
 * shape pair has a kernel -- otherwise the caller launches the heads one by one (instag_mlp_forward / _backward_add). */
int instag_mlp2_supported(int32_t K0, int32_t HA, int32_t OA, int32_t HB, int32_t OB) {
  const int kq = (K0 + 7) / 8, ha = (HA + 7) / 8, oa = (OA + 7) / 8, hb = (HB + 7) / 8, ob = (OB + 7) / 8;
  return kq == 5 && ha == 4 && oa == 4 && hb == 2 && ob == 1;
}

int instag_mlp2_forward(const float* x, const float* wa1, const float* wa2, const float* wb1, const float* wb2,
                        float* ya, float* yb, float* a1a, float* a1b, int32_t N, int32_t K0, int32_t HA, int32_t OA,
                        int32_t HB, int32_t OB, instag_stream_t stream) {
  INSTAG_REQUIRE(instag_mlp2_supported(K0, HA, OA, HB, OB), "mlp2_forward: unsupported shape pair");
  INSTAG_REQUIRE(x && wa1 && wa2 && wb1 && wb2 && ya && yb, "mlp2_forward: NULL tensor");
  INSTAG_REQUIRE((a1a == nullptr) == (a1b == nullptr), "mlp2_forward: a1a and a1b go together (both NULL: forward only)");
  if (N <= 0) return INSTAG_OK;
  const Mlp2Dims d{N, K0, HA, OA, HB, OB};
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_MLP_FWD, s);
  mlp2_forward_kernel<5, 4, 4, 2, 1><<<mlp_blocks((N + 31) / 32), MLP_BLOCK, mlp2_lds_bytes<5, 4, 2>(), s>>>(
      d, x, wa1, wa2, wb1, wb2, ya, yb, a1a, a1b);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_mlp2_backward(const float* dya, const float* dyb, const float* a1a, const float* a1b, const float* wa1,
                         const float* wa2, const float* wb1, const float* wb2, float* dz1a, float* dz1b, float* dx,
                         const float* dx_add, int32_t N, int32_t K0, int32_t HA, int32_t OA, int32_t HB, int32_t OB,
                         instag_stream_t stream) {
  INSTAG_REQUIRE(instag_mlp2_supported(K0, HA, OA, HB, OB), "mlp2_backward: unsupported shape pair");
  INSTAG_REQUIRE(dya && dyb && a1a && a1b && wa1 && wa2 && wb1 && wb2 && dz1a && dz1b, "mlp2_backward: NULL tensor");
  INSTAG_REQUIRE(dx_add == nullptr || dx != nullptr, "mlp2_backward: dx_add needs dx");
  if (N <= 0) return INSTAG_OK;
  const Mlp2Dims d{N, K0, HA, OA, HB, OB};
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_MLP_BWD, s);
  mlp2_backward_kernel<5, 4, 4, 2, 1><<<mlp_blocks((N + 31) / 32), MLP_BLOCK, mlp2_lds_bytes<5, 4, 2>(), s>>>(
      d, dya, dyb, a1a, a1b, wa1, wa2, wb1, wb2, dz1a, dz1b, dx, dx_add);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

size_t instag_linear_weight_grad_workspace_bytes(int32_t N, int32_t O, int32_t K) {
  return (size_t)wg_blocks(N) * (size_t)O * (size_t)K * sizeof(float);
}

int instag_linear_weight_grad(const float* dz, const float* in, float* dw, void* workspace, size_t workspace_bytes,
                              int32_t N, int32_t O, int32_t K, instag_stream_t stream) {
  INSTAG_REQUIRE(dz && in && dw, "linear_weight_grad: NULL tensor");
  INSTAG_REQUIRE(O >= 1 && O <= 64 && K >= 1 && K <= 96, "linear_weight_grad: need O <= 64 and K <= 96");
  INSTAG_REQUIRE(N >= 1, "linear_weight_grad: N must be >= 1");
  if (workspace == nullptr || workspace_bytes < instag_linear_weight_grad_workspace_bytes(N, O, K)) {
    set_error("linear_weight_grad: workspace too small");
    return INSTAG_E_SPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  const int ob = (O + 31) / 32, kb = (K + 31) / 32;
  if (ob == 1 && kb == 1) return run_wg<1, 1>(dz, in, N, O, K, part, dw, s);
  if (ob == 1 && kb == 2) return run_wg<1, 2>(dz, in, N, O, K, part, dw, s);
  if (ob == 1 && kb == 3) return run_wg<1, 3>(dz, in, N, O, K, part, dw, s);
  if (ob == 2 && kb == 1) return run_wg<2, 1>(dz, in, N, O, K, part, dw, s);
  if (ob == 2 && kb == 2) return run_wg<2, 2>(dz, in, N, O, K, part, dw, s);
  return run_wg<2, 3>(dz, in, N, O, K, part, dw, s);
}

static int weight_grad_batched_impl(const instag_wgrad_job* jobs, int32_t n_jobs, const WgVirt& virt, void* workspace,
                                   size_t workspace_bytes, instag_stream_t stream) {
  INSTAG_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= WG_MAX_JOBS, "linear_weight_grad_batched: 1..16 jobs");
  WgBatch b;
  b.virt = virt;
  size_t off = 0;
  int N0 = jobs[0].N, max_count = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const instag_wgrad_job& q = jobs[i];
    INSTAG_REQUIRE(q.dz && q.in && q.dw, "linear_weight_grad_batched: NULL tensor");
    INSTAG_REQUIRE(q.O >= 1 && q.O <= 64 && q.K >= 1 && q.K <= 96, "linear_weight_grad: need O <= 64 and K <= 96");
    INSTAG_REQUIRE(q.N == N0 && q.N >= 1, "linear_weight_grad_batched: all jobs must have the same N >= 1");
    b.j[i] = WgJob{q.dz, q.in, (float*)((char*)workspace + off), q.dw, q.N, q.O, q.K, 0};
    off += (instag_linear_weight_grad_workspace_bytes(q.N, q.O, q.K) + 255) / 256 * 256;
    max_count = std::max(max_count, q.O * q.K);
  }
  if (workspace == nullptr || workspace_bytes < off) {
    set_error("linear_weight_grad_batched: workspace too small");
    return INSTAG_E_SPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int blocks = wg_blocks(N0);
  ProfScope p(K_MLP_WGRAD, s);
  weight_grad_batched_kernel<<<dim3(blocks, n_jobs), MLP_BLOCK, 0, s>>>(b);
  INSTAG_CHECK_LAUNCH();
  weight_grad_reduce_batched_kernel<<<dim3((max_count + 63) / 64, n_jobs), 256, 0, s>>>(b, blocks);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_linear_weight_grad_batched(const instag_wgrad_job* jobs, int32_t n_jobs, void* workspace,
                                      size_t workspace_bytes, instag_stream_t stream) {
  return weight_grad_batched_impl(jobs, n_jobs, WgVirt{-1, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0}, workspace,
                                  workspace_bytes, stream);
}

int instag_linear_weight_grad_batched_glue(const instag_wgrad_job* jobs, int32_t n_jobs, int32_t glue_job,
                                           const float* aud, const float* eye_pre, const float* enc_a,
                                           const float* enc_e, int32_t KA, int32_t KE, void* workspace,
                                           size_t workspace_bytes, instag_stream_t stream) {
  INSTAG_REQUIRE(jobs && glue_job >= 0 && glue_job < n_jobs, "linear_weight_grad_batched_glue: glue_job out of range");
  INSTAG_REQUIRE(aud && eye_pre && enc_a && enc_e && KA >= 1 && KE >= 1, "linear_weight_grad_batched_glue: NULL tensor");
  const int K = jobs[glue_job].K, KX = K - KA - KE;
  INSTAG_REQUIRE(KX >= 1 && (K + 31) / 32 == 3, "linear_weight_grad_batched_glue: the glue job needs 65 <= K <= 96 and KX >= 1");
  return weight_grad_batched_impl(jobs, n_jobs, WgVirt{glue_job, aud, eye_pre, enc_a, enc_e, KX, KA, KE, 0}, workspace,
                                  workspace_bytes, stream);
}

}  // extern "C"
