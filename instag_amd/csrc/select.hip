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
                      float* __restrict__ out_small, float* __restrict__ out_large, int stride = 1, int offset = 0) {
  __shared__ float s[SEL_CHUNK];
  const bool large_half = mode == 1 && (int)blockIdx.x >= chunks;
  const int chunk = large_half ? blockIdx.x - chunks : blockIdx.x;
  const float* __restrict__ src = large_half ? in2 : in;
  const int base = chunk * SEL_CHUNK;
  const int valid = min(SEL_CHUNK, n - base);
  const float pad = large_half ? -INFINITY : INFINITY;                // mode 0: +inf pads, the largest are taken below them
  // (mode 0 reads element i of the caller's vector at in[i * stride + offset]: a column of a row-major matrix)
  const int st = mode == 0 ? stride : 1, of = mode == 0 ? offset : 0;
  for (int i = threadIdx.x; i < SEL_CHUNK; i += SEL_THREADS) s[i] = i < valid ? src[(size_t)(base + i) * st + of] : pad;
  __syncthreads();
  bitonic_sort_chunk(s);
  if ((int)threadIdx.x < k) {
    const int t = threadIdx.x;
    if (!large_half) out_small[chunk * k + t] = t < valid ? s[t] : INFINITY;
    if (mode == 0) out_large[chunk * k + t] = t < valid ? s[valid - 1 - t] : -INFINITY;
    if (large_half) out_large[chunk * k + t] = t < valid ? s[SEL_CHUNK - 1 - t] : -INFINITY;   // (-inf pads sort first)
  }
}

// the mouth field's jaw-movement feature from the sorted extremes (gaussian_renderer/__init__.py:341-349):
// [max, min, max - min] * 1e2 with max / min = the k-th largest / smallest value * scale; k from the device or the host
__global__ void jaw_feature_kernel(const float* __restrict__ largest, const float* __restrict__ smallest, int kmax,
                                   const long long* __restrict__ k_dev, int k_host, float scale,
                                   float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long k = k_dev ? k_dev[0] : (long long)k_host;
  const int idx = (int)min((long long)(kmax - 1), max(0ll, k - 1));
  const float mx = largest[idx] * scale, mn = smallest[idx] * scale;
  out[0] = mx * 1e2f;
  out[1] = mn * 1e2f;
  out[2] = (mx - mn) * 1e2f;
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

static int extreme_values_strided(const float* v, int32_t N, int32_t stride, int32_t offset, int32_t k, float* largest,
                                  float* smallest, void* workspace, hipStream_t s) {
  if (N <= SEL_CHUNK) {
    chunk_extremes_kernel<<<1, SEL_THREADS, 0, s>>>(v, nullptr, N, k, 0, 1, smallest, largest, stride, offset);
    INSTAG_CHECK_LAUNCH();
    return INSTAG_OK;
  }
  float* w = (float*)workspace;
  int chunks = (N + SEL_CHUNK - 1) / SEL_CHUNK;
  float *small = w, *large = w + (size_t)chunks * k;
  w += (size_t)2 * chunks * k;
  chunk_extremes_kernel<<<chunks, SEL_THREADS, 0, s>>>(v, nullptr, N, k, 0, chunks, small, large, stride, offset);
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

/* largest [k] (descending) and smallest [k] (ascending) values of v [N]; N >= 1, 1 <= k <= 64; entries beyond N values
 * are -inf / +inf.  workspace: instag_extreme_values_workspace_bytes(N, k). */
int instag_extreme_values(const float* v, int32_t N, int32_t k, float* largest, float* smallest, void* workspace,
                          size_t workspace_bytes, instag_stream_t stream) {
  INSTAG_REQUIRE(v && largest && smallest, "extreme_values: NULL tensor");
  INSTAG_REQUIRE(N >= 1 && k >= 1 && k <= SEL_KMAX, "extreme_values: need N >= 1 and 1 <= k <= 64");
  INSTAG_REQUIRE(workspace_bytes >= instag_extreme_values_workspace_bytes(N, k) && (workspace || N <= SEL_CHUNK),
                 "extreme_values: workspace too small");
  return extreme_values_strided(v, N, 1, 0, k, largest, smallest, workspace, (hipStream_t)stream);
}

size_t instag_jaw_feature_workspace_bytes(int32_t N, int32_t kmax) {
  return instag_extreme_values_workspace_bytes(N, kmax) + (size_t)2 * SEL_KMAX * sizeof(float);
}

/* out[3] = [max, min, max - min] * 1e2 with max / min = scale * the k-th largest / smallest of the N values
 * v[i * stride + offset] (gaussian_renderer/__init__.py:341-349: the y displacement the face field predicts, one column
 * of its head output).  1 <= kmax <= 64 candidates are kept on each side; k (1-based, clamped to [1, kmax]) comes from
 * k_dev (int64 on the device: a captured step draws it per replay) or, when that is NULL, from k_host. */
int instag_jaw_feature(const float* v, int32_t N, int32_t stride, int32_t offset, float scale, int32_t kmax,
                       const int64_t* k_dev, int32_t k_host, float* out, void* workspace, size_t workspace_bytes,
                       instag_stream_t stream) {
  INSTAG_REQUIRE(v && out && workspace, "jaw_feature: NULL tensor");
  INSTAG_REQUIRE(N >= 1 && kmax >= 1 && kmax <= SEL_KMAX && kmax <= N, "jaw_feature: need 1 <= kmax <= min(64, N)");
  INSTAG_REQUIRE(stride >= 1 && offset >= 0 && offset < stride, "jaw_feature: bad stride / offset");
  INSTAG_REQUIRE(workspace_bytes >= instag_jaw_feature_workspace_bytes(N, kmax), "jaw_feature: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* largest = (float*)workspace;
  float* smallest = largest + SEL_KMAX;
  if (int rc = extreme_values_strided(v, N, stride, offset, kmax, largest, smallest, smallest + SEL_KMAX, s)) return rc;
  jaw_feature_kernel<<<1, 64, 0, s>>>(largest, smallest, kmax, (const long long*)k_dev, k_host, scale, out);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // extern "C"
