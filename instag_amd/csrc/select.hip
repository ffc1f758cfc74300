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

// ---- two-launch selection for long vectors (round 3) ---------------------------------------------------------------
// Launch 1, one workgroup per 512 values (196 for the 100k face Gaussians, where the 4,096-value chunks above gave the
// chip 25 workgroups of 78 bitonic stages each): sort, keep the k extremes of both ends.  Launch 2, ONE workgroup: the
// k-th largest of the chunk maxima is a lower bound T of the global k-th largest, so only candidates >= T can be among
// the k largest -- about a hundred of the chunks' 64 x 196 in all but adversarial inputs (ties at T) -- which are
// compacted, sorted and cut to k; the same for the smallest; then the jaw feature itself.  56 us in three launches
// (26 + 25 + 5) -> two short ones.
constexpr int SEL2_CHUNK = 512;
constexpr int SEL2_THREADS = 256;
constexpr int SEL2_CAP = 16384;               // candidates the final workgroup can hold (64 KB of LDS)

template <int CHUNK, int THREADS>
__device__ __forceinline__ void bitonic_sort_lds(float* s, int n /* power of two <= CHUNK */) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = threadIdx.x; p < n / 2; p += THREADS) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int l = i | j;
        const bool up = (i & k) == 0;
        const float a = s[i], b = s[l];
        if ((a > b) == up) { s[i] = b; s[l] = a; }
      }
      // (pairs 64w .. 64w+63 of a wave touch values [128w, 128w+128) only while j <= 64 and there is one pass per stage)
      if (j > 64 || n / 2 > THREADS || (j == 1 && k >= 128)) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

__global__ void __launch_bounds__(SEL2_THREADS)
chunk512_extremes_kernel(const float* __restrict__ in, int n, int k, int stride, int offset,
                         float* __restrict__ out_small, float* __restrict__ out_large) {
  __shared__ float s[SEL2_CHUNK];
  const int chunk = blockIdx.x;
  const int base = chunk * SEL2_CHUNK;
  const int valid = min(SEL2_CHUNK, n - base);
  for (int i = threadIdx.x; i < SEL2_CHUNK; i += SEL2_THREADS)
    s[i] = i < valid ? in[(size_t)(base + i) * stride + offset] : INFINITY;
  __syncthreads();
  bitonic_sort_lds<SEL2_CHUNK, SEL2_THREADS>(s, SEL2_CHUNK);
  if ((int)threadIdx.x < k) {
    const int t = threadIdx.x;
    out_small[chunk * k + t] = t < valid ? s[t] : INFINITY;
    out_large[chunk * k + t] = t < valid ? s[valid - 1 - t] : -INFINITY;
  }
}

// Launch 2: both sides at once -- threads 0..511 work on the largest, 512..1023 on the smallest, every barrier shared
// (the two halves run the same sequence of steps on their own LDS arrays; sizes that steer the control flow are the
// maximum over both sides).  Values are handled as sign * value, so both halves look for "the largest".
//   step 1  T = a lower bound of the global k-th largest: the k-th largest of the chunks' m best values each,
//           m = ceil(k / chunks) (with chunks >= k: of the chunk maxima).  k of the candidates are >= it by construction,
//           and in all but adversarial inputs few more (the best chunk's own k-th value, a bound too, lets ~5 % of a
//           Gaussian vector through: 5,500 candidates instead of 100).
//   step 2  the candidates >= T: every chunk's list is sorted best first, so one thread per chunk walks its list, eight
//           values per round trip, until it falls below T.
//   step 3  sort them, cut to k.
// Sorting few values (<= 512) = RANK COUNTING: every thread counts how many values beat its own (n broadcast reads, ~1 us
// for a few hundred values; a bitonic network costs 0.1-0.15 us per stage, 28-45 stages); more = the bitonic network.
constexpr int SEL2_HALF = 512;

// dst[rank] = value, descending, ties by index; n <= SEL2_HALF, called by both halves (t = thread within the half)
__device__ __forceinline__ void rank_sort_desc(const float* src, float* dst, int n, int t) {
  const float mine = t < n ? src[t] : 0.f;
  int rank = 0;
  for (int j = 0; j < n; ++j) {
    const float o = src[j];                         // same address for the whole wave: a broadcast read
    rank += (o > mine || (o == mine && j < t)) ? 1 : 0;
  }
  if (t < n) dst[rank] = mine;
}

// ascending bitonic sort of P values per half (P the same for both halves, a power of two), SEL2_HALF threads each
__device__ __forceinline__ void bitonic_sort_halves(float* s, int P, int t) {
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = t; p < P / 2; p += SEL2_HALF) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int l = i | j;
        const bool up = (i & k) == 0;
        const float a = s[i], b = s[l];
        if ((a > b) == up) { s[i] = b; s[l] = a; }
      }
      __syncthreads();
    }
}

// src (n values, capacity SEL2_CAP) -> dst[0 .. min(n, SEL2_HALF)) descending; n_max = max of n over both halves
__device__ __forceinline__ void sort_desc(float* src, float* dst, int n, int n_max, int t) {
  if (n_max <= SEL2_HALF) {
    rank_sort_desc(src, dst, n, t);
    __syncthreads();
    return;
  }
  int P = 1024;
  while (P < n_max) P <<= 1;                          // <= SEL2_CAP
  for (int i = n + t; i < P; i += SEL2_HALF) src[i] = -INFINITY;
  __syncthreads();
  bitonic_sort_halves(src, P, t);
  for (int i = t; i < min(n, SEL2_HALF); i += SEL2_HALF) dst[i] = src[P - 1 - i];
  __syncthreads();
}

// candidates [chunks][k] per side (each chunk's list sorted, best first) -> largest [k] descending, smallest [k]
// ascending; then (out != null) the jaw feature
__global__ void __launch_bounds__(1024)
final_extremes_kernel(const float* __restrict__ cand_small, const float* __restrict__ cand_large, int chunks, int k,
                      float* __restrict__ smallest, float* __restrict__ largest, const long long* __restrict__ k_dev,
                      int k_host, float scale, float* __restrict__ out) {
  extern __shared__ float s_all[];               // per side: SEL2_CAP gathered values + SEL2_HALF sorted ones
  __shared__ int s_cnt[2];
  const int side = threadIdx.x / SEL2_HALF, t = threadIdx.x % SEL2_HALF;     // 0: largest, 1: smallest
  float* s_c = s_all + (size_t)side * (SEL2_CAP + SEL2_HALF);
  float* s_o = s_c + SEL2_CAP;
  const float* __restrict__ cand = side == 0 ? cand_large : cand_small;
  const float sign = side == 0 ? 1.f : -1.f;
  // step 1
  const int m = (k + chunks - 1) / chunks;
  const int ns = chunks * m;                     // <= chunks * k <= SEL2_CAP, the same on both sides
  for (int i = t; i < ns; i += SEL2_HALF) s_c[i] = sign * cand[(i / m) * k + (i % m)];
  if (t == 0) s_cnt[side] = 0;
  __syncthreads();
  sort_desc(s_c, s_o, ns, ns, t);
  const float T = s_o[k - 1];
  __syncthreads();
  // step 2 (the order of the compacted values is arbitrary: they are sorted next)
  for (int c = t; c < chunks; c += SEL2_HALF) {
    const float* __restrict__ row = cand + (size_t)c * k;
    bool more = true;
    for (int j0 = 0; j0 < k && more; j0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (j0 + j < k) ? sign * row[j0 + j] : -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        more = more && (j0 + j < k) && v[j] >= T;
        if (more) s_c[atomicAdd(&s_cnt[side], 1)] = v[j];       // (<= chunks * k <= SEL2_CAP slots)
      }
    }
  }
  __syncthreads();
  const int cnt = s_cnt[side];                   // >= k
  const int cnt_max = max(s_cnt[0], s_cnt[1]);
  // step 3
  sort_desc(s_c, s_o, cnt, cnt_max, t);
  float* __restrict__ dst = side == 0 ? largest : smallest;
  if (t < k) dst[t] = sign * s_o[t];
  __syncthreads();
  if (out != nullptr && threadIdx.x == 0) {
    const long long kk = k_dev ? k_dev[0] : (long long)k_host;
    const int idx = (int)min((long long)(k - 1), max(0ll, kk - 1));
    const float mx = largest[idx] * scale, mn = smallest[idx] * scale;
    out[0] = mx * 1e2f;
    out[1] = mn * 1e2f;
    out[2] = (mx - mn) * 1e2f;
  }
}

inline bool use_two_launch(int N, int k) {
  const long chunks = ((long)N + SEL2_CHUNK - 1) / SEL2_CHUNK;
  return N > SEL_CHUNK && chunks * k <= SEL2_CAP;
}

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

size_t instag_extreme_values_workspace_bytes(int32_t N, int32_t k) {
  if (use_two_launch(N, k)) return (size_t)2 * ((N + SEL2_CHUNK - 1) / SEL2_CHUNK) * k * sizeof(float) + 256;
  size_t total = 0;
  for (long n = N; n > SEL_CHUNK;) {
    const long chunks = (n + SEL_CHUNK - 1) / SEL_CHUNK;
    total += (size_t)2 * chunks * k * sizeof(float);
    n = chunks * k;
  }
  return total + 256;
}

// jaw_out (with k_dev / k_host / scale): the two-launch path writes the jaw feature in its final launch (returns 1 in
// *jaw_done); the chunked path leaves it to the caller
static int extreme_values_strided(const float* v, int32_t N, int32_t stride, int32_t offset, int32_t k, float* largest,
                                  float* smallest, void* workspace, hipStream_t s, float* jaw_out = nullptr,
                                  const int64_t* k_dev = nullptr, int32_t k_host = 0, float scale = 1.f,
                                  int* jaw_done = nullptr) {
  if (jaw_done) *jaw_done = 0;
  if (use_two_launch(N, k)) {
    const int chunks = (N + SEL2_CHUNK - 1) / SEL2_CHUNK;
    float* small = (float*)workspace;
    float* large = small + (size_t)chunks * k;
    chunk512_extremes_kernel<<<chunks, SEL2_THREADS, 0, s>>>(v, N, k, stride, offset, small, large);
    INSTAG_CHECK_LAUNCH();
    if (int e = set_max_dynamic_lds((const void*)final_extremes_kernel, 2 * (SEL2_CAP + SEL2_HALF) * (int)sizeof(float))) return e;
    final_extremes_kernel<<<1, 1024, 2 * (SEL2_CAP + SEL2_HALF) * sizeof(float), s>>>(small, large, chunks, k, smallest, largest,
                                                                  (const long long*)k_dev, k_host, scale, jaw_out);
    INSTAG_CHECK_LAUNCH();
    if (jaw_done) *jaw_done = jaw_out != nullptr;
    return INSTAG_OK;
  }
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
  int jaw_done = 0;
  if (int rc = extreme_values_strided(v, N, stride, offset, kmax, largest, smallest, smallest + SEL_KMAX, s, out, k_dev,
                                      k_host, scale, &jaw_done)) return rc;
  if (jaw_done) return INSTAG_OK;               // (the two-launch selection wrote it in its final launch)
  jaw_feature_kernel<<<1, 64, 0, s>>>(largest, smallest, kmax, (const long long*)k_dev, k_host, scale, out);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // extern "C"
