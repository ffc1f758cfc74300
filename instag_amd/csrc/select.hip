// The k largest and the k smallest values of a vector, each sorted, k <= 64.
//
// Replaces the two torch.topk selections of the mouth branch's jaw-movement feature
// (gaussian_renderer/__init__.py:341-349: `motion_preds_face['d_xyz'][..., 1].topk(k)` largest / smallest over the
// ~100k face Gaussians).  torch's 1-D topk takes a full-sort path on ROCm for >= 10,000 elements (not capturable, and
// 190 us for both selections in its row-wise form); here every 4,096-value chunk is sorted once in LDS (bitonic, 1,024
// threads) and contributes its k extremes at both ends, and the two candidate lists are reduced the same way (one launch
// per level for both) until one chunk of each is left.  Values only (the caller never needs the indices).
#include "common.hpp"
#include <math.h>

namespace instag {
namespace {

constexpr int SEL_THREADS = 1024;
constexpr int SEL_CHUNK = 4096;
constexpr int SEL_KMAX = 64;

// Compare-exchange distances of 64 and below stay inside the 128 values a wave owns (pairs 64w .. 64w+63 of either
// half touch values [128w, 128w+128) only), and a wave's LDS operations execute in order: those stages need no
// workgroup barrier -- 20 barriers per sort instead of 78.
__device__ __forceinline__ void bitonic_sort_chunk(float* s) {        // ascending, SEL_CHUNK values, SEL_THREADS threads
  for (int k = 2; k <= SEL_CHUNK; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int t = 0; t < SEL_CHUNK / 2 / SEL_THREADS; ++t) {
        const int p = threadIdx.x + t * SEL_THREADS;                  // pair index
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));          // lower element of the pair
        const int l = i | j;
        const bool up = (i & k) == 0;
        const float a = s[i], b = s[l];
        if ((a > b) == up) { s[i] = b; s[l] = a; }
      }
      if (j > 64 || (j == 1 && k >= 128)) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// mode 0: both ends of `in` (first level); mode 1: the first `chunks` workgroups reduce the candidate list `in` to its
// smallest, the other `chunks` workgroups reduce `in2` to its largest.  Chunk c writes its k smallest (ascending, +inf
// padded) to out_small[c*k ..] and / or its k largest (descending, -inf padded) to out_large[c*k ..].
__global__ void __launch_bounds__(SEL_THREADS)
chunk_extremes_kernel(const float* __restrict__ in, const float* __restrict__ in2, int n, int k, int mode, int chunks,
                      float* __restrict__ out_small, float* __restrict__ out_large) {
  __shared__ float s[SEL_CHUNK];
  const bool large_half = mode == 1 && (int)blockIdx.x >= chunks;
  const int chunk = large_half ? blockIdx.x - chunks : blockIdx.x;
  const float* __restrict__ src = large_half ? in2 : in;
  const int base = chunk * SEL_CHUNK;
  const int valid = min(SEL_CHUNK, n - base);
  const float pad = large_half ? -INFINITY : INFINITY;                // mode 0: +inf pads, the largest are taken below them
  for (int i = threadIdx.x; i < SEL_CHUNK; i += SEL_THREADS) s[i] = i < valid ? src[base + i] : pad;
  __syncthreads();
  bitonic_sort_chunk(s);
  if ((int)threadIdx.x < k) {
    const int t = threadIdx.x;
    if (!large_half) out_small[chunk * k + t] = t < valid ? s[t] : INFINITY;
    if (mode == 0) out_large[chunk * k + t] = t < valid ? s[valid - 1 - t] : -INFINITY;
    if (large_half) out_large[chunk * k + t] = t < valid ? s[SEL_CHUNK - 1 - t] : -INFINITY;   // (-inf pads sort first)
  }
}

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

size_t instag_extreme_values_workspace_bytes(int32_t N, int32_t k) {
  size_t total = 0;
  for (long n = N; n > SEL_CHUNK;) {
    const long chunks = (n + SEL_CHUNK - 1) / SEL_CHUNK;
    total += (size_t)2 * chunks * k * sizeof(float);
    n = chunks * k;
  }
  return total + 256;
}

/* largest [k] (descending) and smallest [k] (ascending) values of v [N]; N >= 1, 1 <= k <= 64; entries beyond N values
 * are -inf / +inf.  workspace: instag_extreme_values_workspace_bytes(N, k). */
int instag_extreme_values(const float* v, int32_t N, int32_t k, float* largest, float* smallest, void* workspace,
                          size_t workspace_bytes, instag_stream_t stream) {
  INSTAG_REQUIRE(v && largest && smallest, "extreme_values: NULL tensor");
  INSTAG_REQUIRE(N >= 1 && k >= 1 && k <= SEL_KMAX, "extreme_values: need N >= 1 and 1 <= k <= 64");
  INSTAG_REQUIRE(workspace_bytes >= instag_extreme_values_workspace_bytes(N, k) && (workspace || N <= SEL_CHUNK),
                 "extreme_values: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (N <= SEL_CHUNK) {
    chunk_extremes_kernel<<<1, SEL_THREADS, 0, s>>>(v, nullptr, N, k, 0, 1, smallest, largest);
    INSTAG_CHECK_LAUNCH();
    return INSTAG_OK;
  }
  float* w = (float*)workspace;
  int chunks = (N + SEL_CHUNK - 1) / SEL_CHUNK;
  float *small = w, *large = w + (size_t)chunks * k;
  w += (size_t)2 * chunks * k;
  chunk_extremes_kernel<<<chunks, SEL_THREADS, 0, s>>>(v, nullptr, N, k, 0, chunks, small, large);
  INSTAG_CHECK_LAUNCH();
  int n = chunks * k;
  while (n > SEL_CHUNK) {
    chunks = (n + SEL_CHUNK - 1) / SEL_CHUNK;
    float *small2 = w, *large2 = w + (size_t)chunks * k;
    w += (size_t)2 * chunks * k;
    chunk_extremes_kernel<<<2 * chunks, SEL_THREADS, 0, s>>>(small, large, n, k, 1, chunks, small2, large2);
    INSTAG_CHECK_LAUNCH();
    small = small2; large = large2; n = chunks * k;
  }
  chunk_extremes_kernel<<<2, SEL_THREADS, 0, s>>>(small, large, n, k, 1, 1, smallest, largest);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // extern "C"
