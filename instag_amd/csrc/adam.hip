// Multi-tensor Adam / AdamW: every parameter tensor of every group of an optimizer in ONE launch.
//
// Replaces the optimizer steps of the reference's train loop (train_face.py:781-788:
// motion_optimizer.step() = AdamW over the UMF groups, gaussians.optimizer.step() = Adam(eps 1e-15) over
// the 7 per-Gaussian groups + the PMF groups, scene/gaussian_model.py:369-403), which eager PyTorch
// runs as one fused-multi-tensor launch per group plus a step-counter update per group (~35 launches).
// Arithmetic = torch.optim.Adam / AdamW (amsgrad=False, maximize=False):
//   AdamW: p *= 1 - lr*wd;  Adam: g += wd*p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// Learning rates and the step counter live in device memory so that a captured hipGraph can be replayed.
#include <cstring>

#include "common.hpp"

namespace instag {
namespace {

struct AdamTensor {     // one parameter tensor (device pointers), 48 bytes
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  int32_t group;
  int32_t pad;
};
struct AdamGroup {      // hyper-parameters of one group, 24 bytes
  float beta1, beta2, eps, weight_decay;
  int32_t decoupled;    // 1 = AdamW
  int32_t pad;
};
constexpr int ADAM_BLOCK = 256;
#ifndef ADAM_CHUNK_N
#define ADAM_CHUNK_N 4096
#endif
constexpr int ADAM_CHUNK = ADAM_CHUNK_N;   // elements per workgroup (16 per thread; 2,048 and 8,192 measured: no better)

// per-tensor step counters (torch keeps state['step'] per parameter; a parameter without gradient is not stepped)
__global__ void adam_tick_kernel(const AdamTensor* __restrict__ tensors, int n, float* __restrict__ steps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && tensors[i].g != nullptr) steps[i] += 1.0f;
}

__global__ void __launch_bounds__(ADAM_BLOCK)
adam_step_kernel(const AdamTensor* __restrict__ tensors, const AdamGroup* __restrict__ groups,
                 const float* __restrict__ lrs, const int2* __restrict__ chunks, const float* __restrict__ step) {
  const int2 ch = chunks[blockIdx.x];            // (tensor index, chunk index)
  const AdamTensor t = tensors[ch.x];
  if (t.g == nullptr) return;
  const AdamGroup gr = groups[t.group];
  const float lr = lrs[t.group];
  const float tstep = step[ch.x];
  const float bc1 = 1.0f - powf(gr.beta1, tstep);
  const float bc2_sqrt = sqrtf(1.0f - powf(gr.beta2, tstep));
  const float step_size = lr / bc1;
  const int64_t base = (int64_t)ch.y * ADAM_CHUNK;
  const int64_t end = min(t.n, base + ADAM_CHUNK);
  for (int64_t i = base + threadIdx.x; i < end; i += ADAM_BLOCK) {
    float p = t.p[i], g = t.g[i], m = t.m[i], v = t.v[i];
    if (gr.decoupled) p *= 1.0f - lr * gr.weight_decay;
    else if (gr.weight_decay != 0.f) g += gr.weight_decay * p;
    m = gr.beta1 * m + (1.0f - gr.beta1) * g;
    v = gr.beta2 * v + (1.0f - gr.beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + gr.eps;
    p -= step_size * (m / denom);
    t.p[i] = p; t.m[i] = m; t.v[i] = v;
  }
}

// The same two kernels with the GRADIENT pointers passed by value in the kernel arguments (<= 384 tensors = 3 KB):
// parameter / moment pointers change only when the parameter set is rebuilt and stay in the device table, the
// gradient pointers change every step -- this way no host-to-device copy precedes the launch, and in a captured step
// the pointers are part of the kernel node instead of a memcpy node that has to land before the optimizer can start.
constexpr int ADAM_GRADS_MAX = 384;
struct AdamGrads { const float* g[ADAM_GRADS_MAX]; };

__global__ void adam_tick_grads_kernel(AdamGrads gr, int n, float* __restrict__ steps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && gr.g[i] != nullptr) steps[i] += 1.0f;
}

// TICKET: no separate counter launch.  Every workgroup reads the tensor's counter of COMPLETED steps, uses counter + 1,
// and draws a ticket when it is done; the workgroup that draws the tensor's last ticket stores the new count and
// clears the tickets.  All reads of the counter precede the store (each workgroup reads before it draws), and the
// next launch is ordered behind this one by the stream.
template <bool TICKET>
__global__ void __launch_bounds__(ADAM_BLOCK)
adam_step_grads_kernel(const AdamTensor* __restrict__ tensors, AdamGrads grads, const AdamGroup* __restrict__ groups,
                       const float* __restrict__ lrs, const int2* __restrict__ chunks, float* step,
                       int32_t* __restrict__ tickets) {
  const int2 ch = chunks[blockIdx.x];            // (tensor index, chunk index)
  const float* __restrict__ tg = grads.g[ch.x];
  if (tg == nullptr) return;
  const AdamTensor t = tensors[ch.x];
  const AdamGroup gr = groups[t.group];
  const float lr = lrs[t.group];
  const float tstep = TICKET ? step[ch.x] + 1.0f : step[ch.x];
  const float bc1 = 1.0f - powf(gr.beta1, tstep);
  const float bc2_sqrt = sqrtf(1.0f - powf(gr.beta2, tstep));
  const float step_size = lr / bc1;
  const int64_t base = (int64_t)ch.y * ADAM_CHUNK;
  const int64_t end = min(t.n, base + ADAM_CHUNK);
  auto update = [&](float& p, float g, float& m, float& v) {
    if (gr.decoupled) p *= 1.0f - lr * gr.weight_decay;
    else if (gr.weight_decay != 0.f) g += gr.weight_decay * p;
    m = gr.beta1 * m + (1.0f - gr.beta1) * g;
    v = gr.beta2 * v + (1.0f - gr.beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + gr.eps;
    p -= step_size * (m / denom);
  };
  // 16-byte loads and stores where the four arrays allow it (a chunk starts at a multiple of 4,096 elements, so the
  // tensors' own alignment decides; gradients handed back as views of a flat bucket may sit at any multiple of 4 bytes)
  const bool wide = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(tg) | reinterpret_cast<uintptr_t>(t.m) |
                      reinterpret_cast<uintptr_t>(t.v)) & 15u) == 0u;
  const int64_t end4 = wide ? base + ((end - base) & ~(int64_t)3) : base;
  for (int64_t i = base + 4 * threadIdx.x; i < end4; i += 4 * ADAM_BLOCK) {
    float4 p = *reinterpret_cast<const float4*>(t.p + i), m = *reinterpret_cast<const float4*>(t.m + i);
    float4 v = *reinterpret_cast<const float4*>(t.v + i);
    const float4 g = *reinterpret_cast<const float4*>(tg + i);
    update(p.x, g.x, m.x, v.x); update(p.y, g.y, m.y, v.y); update(p.z, g.z, m.z, v.z); update(p.w, g.w, m.w, v.w);
    *reinterpret_cast<float4*>(t.p + i) = p; *reinterpret_cast<float4*>(t.m + i) = m; *reinterpret_cast<float4*>(t.v + i) = v;
  }
  for (int64_t i = end4 + threadIdx.x; i < end; i += ADAM_BLOCK) {
    float p = t.p[i], m = t.m[i], v = t.v[i];
    update(p, tg[i], m, v);
    t.p[i] = p; t.m[i] = m; t.v[i] = v;
  }
  if (TICKET) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int nchunks = (int)((t.n + ADAM_CHUNK - 1) / ADAM_CHUNK);
      if (atomicAdd(&tickets[ch.x], 1) == nchunks - 1) {
        tickets[ch.x] = 0;
        step[ch.x] = tstep;
      }
    }
  }
}

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

int instag_adam_chunk_elems(void) { return ADAM_CHUNK; }

/* tensors: device array of 48-byte records {p, g, m, v, int64 n, int32 group, int32 pad}; groups: device array of
 * 24-byte records {beta1, beta2, eps, weight_decay, int32 decoupled, int32 pad}; lrs: device float[n_groups];
 * chunks: device int32[n_chunks][2] = (tensor index, chunk index within the tensor, instag_adam_chunk_elems()
 * elements each); step: device float[n_tensors], the per-tensor step counters, incremented by this call (for tensors
 * with a gradient) before they are used. */
int instag_adam_step(const void* tensors, int32_t n_tensors, const void* groups, const float* lrs,
                     const int32_t* chunks, int32_t n_chunks, float* step, instag_stream_t stream) {
  INSTAG_REQUIRE(tensors && groups && lrs && chunks && step, "adam_step: NULL argument");
  if (n_chunks <= 0 || n_tensors <= 0) return INSTAG_OK;
  hipStream_t s = (hipStream_t)stream;
  adam_tick_kernel<<<(n_tensors + 63) / 64, 64, 0, s>>>((const AdamTensor*)tensors, n_tensors, step);
  INSTAG_CHECK_LAUNCH();
  adam_step_kernel<<<n_chunks, ADAM_BLOCK, 0, s>>>((const AdamTensor*)tensors, (const AdamGroup*)groups, lrs,
                                                   (const int2*)chunks, step);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_adam_grads_max(void) { return ADAM_GRADS_MAX; }

/* As instag_adam_step, but the gradient pointers come as a HOST array `host_grads` (uint64[n_tensors], 0 = the tensor
 * has no gradient this step; n_tensors <= instag_adam_grads_max()) and travel in the kernel arguments; the `g` field of
 * the device records is ignored, so the device table only has to be uploaded when the parameter set changes. */
int instag_adam_step_grads(const void* tensors, const void* host_grads, int32_t n_tensors, const void* groups,
                           const float* lrs, const int32_t* chunks, int32_t n_chunks, float* step,
                           instag_stream_t stream) {
  INSTAG_REQUIRE(tensors && host_grads && groups && lrs && chunks && step, "adam_step: NULL argument");
  INSTAG_REQUIRE(n_tensors <= ADAM_GRADS_MAX, "adam_step_grads: more tensors than instag_adam_grads_max()");
  if (n_chunks <= 0 || n_tensors <= 0) return INSTAG_OK;
  hipStream_t s = (hipStream_t)stream;
  AdamGrads gr;
  memset(&gr, 0, sizeof(gr));
  memcpy(gr.g, host_grads, (size_t)n_tensors * sizeof(const float*));
  adam_tick_grads_kernel<<<(n_tensors + 63) / 64, 64, 0, s>>>(gr, n_tensors, step);
  INSTAG_CHECK_LAUNCH();
  adam_step_grads_kernel<false><<<n_chunks, ADAM_BLOCK, 0, s>>>((const AdamTensor*)tensors, gr, (const AdamGroup*)groups,
                                                                lrs, (const int2*)chunks, step, nullptr);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

/* As instag_adam_step_grads in ONE launch: `tickets` = int32[n_tensors] in device memory, zero before the first call
 * and left zero by every call; the step counters are incremented by the kernel itself (by the workgroup of each tensor
 * that finishes last).  `chunks` must list every chunk of every tensor it names exactly once (a step may be made
 * of several calls over disjoint sets of tensors). */
int instag_adam_step_grads_ticketed(const void* tensors, const void* host_grads, int32_t n_tensors, const void* groups,
                                    const float* lrs, const int32_t* chunks, int32_t n_chunks, float* step,
                                    int32_t* tickets, instag_stream_t stream) {
  INSTAG_REQUIRE(tensors && host_grads && groups && lrs && chunks && step && tickets, "adam_step: NULL argument");
  INSTAG_REQUIRE(n_tensors <= ADAM_GRADS_MAX, "adam_step_grads: more tensors than instag_adam_grads_max()");
  if (n_chunks <= 0 || n_tensors <= 0) return INSTAG_OK;
  hipStream_t s = (hipStream_t)stream;
  AdamGrads gr;
  memset(&gr, 0, sizeof(gr));
  memcpy(gr.g, host_grads, (size_t)n_tensors * sizeof(const float*));
  adam_step_grads_kernel<true><<<n_chunks, ADAM_BLOCK, 0, s>>>((const AdamTensor*)tensors, gr, (const AdamGroup*)groups,
                                                               lrs, (const int2*)chunks, step, tickets);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // extern "C"
