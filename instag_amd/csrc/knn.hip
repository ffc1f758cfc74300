// Mean squared distance to the three nearest neighbours of every point: the provider behind
// `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20, used once at :246 to initialise the Gaussian
// scales from the point cloud).  The reference's `simple_knn` is an absent third-party submodule; this restates its
// published contract (per point: (d1^2 + d2^2 + d3^2) / 3 over the three closest OTHER points) with an exact
// brute-force search: one query per thread, the cloud streamed through LDS in 256-point tiles.  N^2 pair evaluations
// (1e10 for 100k points, a few milliseconds) are affordable for an operator that runs once per training run.
#include "common.hpp"

namespace instag {
namespace {

constexpr int KB_ = 256;

__global__ void __launch_bounds__(KB_)
knn3_mean_dist2_kernel(const float* __restrict__ pts, float* __restrict__ out, int N) {
  __shared__ float4 s_p[KB_];
  const int i = blockIdx.x * KB_ + threadIdx.x;
  const bool valid = i < N;
  const float qx = valid ? pts[3 * i] : 0.f, qy = valid ? pts[3 * i + 1] : 0.f, qz = valid ? pts[3 * i + 2] : 0.f;
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;      // three smallest squared distances, ascending
  for (int t0 = 0; t0 < N; t0 += KB_) {
    const int j = t0 + threadIdx.x;
    s_p[threadIdx.x] = j < N ? make_float4(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], 0.f)
                             : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
    __syncthreads();
    const int cnt = min(KB_, N - t0);
#pragma unroll 8
    for (int k = 0; k < cnt; ++k) {
      const float4 p = s_p[k];
      const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
      float d = dx * dx + dy * dy + dz * dz;
      d = (t0 + k == i) ? INFINITY : d;                     // the point itself is not its own neighbour
      // branch-free insertion into the sorted triple
      const float m0 = fminf(b0, d), x0 = fmaxf(b0, d);
      const float m1 = fminf(b1, x0), x1 = fmaxf(b1, x0);
      b0 = m0; b1 = m1; b2 = fminf(b2, x1);
    }
    __syncthreads();
  }
  if (valid) {
    // fewer than three other points: average over the neighbours that exist (0 for a single point)
    const int nn = min(3, N - 1);
    float s = 0.f;
    if (nn >= 1) s += b0;
    if (nn >= 2) s += b1;
    if (nn >= 3) s += b2;
    out[i] = nn > 0 ? s / (float)nn : 0.f;
  }
}

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

int instag_knn3_mean_dist2(const float* points, float* out, int32_t N, instag_stream_t stream) {
  INSTAG_REQUIRE(N >= 0, "knn: N must be >= 0");
  if (N == 0) return INSTAG_OK;
  INSTAG_REQUIRE(points && out, "knn: NULL tensor");
  knn3_mean_dist2_kernel<<<(N + KB_ - 1) / KB_, KB_, 0, (hipStream_t)stream>>>(points, out, N);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // extern "C"
