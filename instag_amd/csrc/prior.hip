// Monocular geometry priors of the face branch after warm_step + 2000, forward and backward:
//
//   loss  = w_n * mean_{head ^ mouth} sum_c (1 - n_gt[c] * n[c])                              train_face.py:466
//         + w_d * mean_{face ^ mouth} | normalize(d) - normalize(d_gt) |                      train_face.py:478-504
//   normalize(x)[r][j] = (x[r][j] - mean_r) / (std_r + 0.01 * std_all)                         utils/loss_utils.py:17-20
//
// (row statistics over dim 1, torch.std = unbiased).  The reference runs this as ~40 eager launches forward and ~60
// backward on [H,W] maps; here: row statistics, loss partials, finalize | row sums of the incoming gradient, apply.
// One workgroup per image row; the statistics of all rows are re-reduced by every workgroup (H values) instead of by
// one more launch.  Sums run in a fixed order: bitwise reproducible.
#include "common.hpp"

namespace instag {
namespace {

constexpr int PB = 256;

__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

struct RowStat { float mean_d, std_d, ss_d, mean_g, std_g, ss_g, pad0, pad1; };   // ss = sum (x - mean_row)^2

// row r: mean, unbiased std and centred sum of squares of the rendered and of the monocular depth
__global__ void __launch_bounds__(PB)
prior_stats_kernel(const float* __restrict__ depth, const float* __restrict__ gt_depth, int H, int W,
                   RowStat* __restrict__ stat) {
  __shared__ float s_red[4];
  const int r = blockIdx.x;
  const float* d = depth + (size_t)r * W;
  const float* g = gt_depth + (size_t)r * W;
  float sd = 0.f, sg = 0.f;
  for (int j = threadIdx.x; j < W; j += PB) { sd += d[j]; sg += g[j]; }
  const float md = block_sum(sd, s_red) / (float)W;
  const float mg = block_sum(sg, s_red) / (float)W;
  float qd = 0.f, qg = 0.f;
  for (int j = threadIdx.x; j < W; j += PB) {
    const float a = d[j] - md, b = g[j] - mg;
    qd += a * a; qg += b * b;
  }
  qd = block_sum(qd, s_red);
  qg = block_sum(qg, s_red);
  if (threadIdx.x == 0) {
    const float den = (float)max(W - 1, 1);
    stat[r] = RowStat{md, sqrtf(qd / den), qd, mg, sqrtf(qg / den), qg, 0.f, 0.f};
  }
}

struct Global { float mean_d, std_d, mean_g, std_g; };

// statistics of the whole map from the row statistics: sum (x - M)^2 = sum_r [ss_r + W (mean_r - M)^2]
__device__ __forceinline__ Global global_stats(const RowStat* __restrict__ stat, int H, int W, float* s_red) {
  float a = 0.f, b = 0.f;
  for (int r = threadIdx.x; r < H; r += PB) { a += stat[r].mean_d; b += stat[r].mean_g; }
  const float Md = block_sum(a, s_red) / (float)H, Mg = block_sum(b, s_red) / (float)H;
  a = b = 0.f;
  for (int r = threadIdx.x; r < H; r += PB) {
    const float ed = stat[r].mean_d - Md, eg = stat[r].mean_g - Mg;
    a += stat[r].ss_d + (float)W * ed * ed;
    b += stat[r].ss_g + (float)W * eg * eg;
  }
  const float den = (float)max((long)H * W - 1, 1L);
  return Global{Md, sqrtf(block_sum(a, s_red) / den), Mg, sqrtf(block_sum(b, s_red) / den)};
}

// per row: [sum |nd - ngt| sel, sum sel, sum m * sum_c (1 - n_gt n), sum m]
__global__ void __launch_bounds__(PB)
prior_loss_kernel(const float* __restrict__ normal, const float* __restrict__ depth,
                  const float* __restrict__ gt_normal, const float* __restrict__ gt_depth,
                  const uint8_t* __restrict__ face, const uint8_t* __restrict__ hair, const uint8_t* __restrict__ mouth,
                  int H, int W, int use_depth, const RowStat* __restrict__ stat, float* __restrict__ parts) {
  __shared__ float s_red[4];
  const int r = blockIdx.x;
  const size_t P = (size_t)H * W, base = (size_t)r * W;
  float sd = 1.f, sg = 1.f, md = 0.f, mg = 0.f;
  if (use_depth) {
    const Global G = global_stats(stat, H, W, s_red);
    const RowStat st = stat[r];
    md = st.mean_d; mg = st.mean_g;
    sd = st.std_d + 0.01f * G.std_d;
    sg = st.std_g + 0.01f * G.std_g;
  }
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = threadIdx.x; j < W; j += PB) {
    const size_t p = base + j;
    const bool f = face[p] != 0, h = hair[p] != 0, mo = mouth[p] != 0;
    if (use_depth && (f != mo)) {
      v[0] += fabsf((depth[p] - md) / sd - (gt_depth[p] - mg) / sg);
      v[1] += 1.f;
    }
    if ((f || h) != mo) {
      v[2] += (1.f - gt_normal[p] * normal[p]) + (1.f - gt_normal[P + p] * normal[P + p]) +
              (1.f - gt_normal[2 * P + p] * normal[2 * P + p]);
      v[3] += 1.f;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float s = block_sum(v[k], s_red);
    if (threadIdx.x == 0) parts[4 * r + k] = s;
  }
}

// out = [loss, S_depth, N_sel, S_normal, N_m]
__global__ void __launch_bounds__(PB)
prior_finalize_kernel(const float* __restrict__ parts, int H, int use_depth, float w_normal, float w_depth,
                      float* __restrict__ out) {
  __shared__ float s_red[4];
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = threadIdx.x; r < H; r += PB)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += parts[4 * r + k];
  float s[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] = block_sum(v[k], s_red);
  if (threadIdx.x == 0) {
    float loss = w_normal * s[2] / s[3];
    if (use_depth) loss += w_depth * s[0] / s[1];
    out[0] = loss; out[1] = s[0]; out[2] = s[1]; out[3] = s[2]; out[4] = s[3];
  }
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// row sums of the gradient that reaches the normalised depth: A_r = sum g, B_r = sum g (d - mean_r); also writes the
// gradient of the normal image (it has no cross-pixel coupling)
__global__ void __launch_bounds__(PB)
prior_backward_rows_kernel(const float* __restrict__ go, const float* __restrict__ depth,
                           const float* __restrict__ gt_normal, const float* __restrict__ gt_depth,
                           const uint8_t* __restrict__ face, const uint8_t* __restrict__ hair,
                           const uint8_t* __restrict__ mouth, int H, int W, int use_depth, float w_normal, float w_depth,
                           const RowStat* __restrict__ stat, const float* __restrict__ out, float* __restrict__ rowb,
                           float* __restrict__ d_normal) {
  __shared__ float s_red[4];
  const int r = blockIdx.x;
  const size_t P = (size_t)H * W, base = (size_t)r * W;
  const float g0 = go[0];
  const float cn = -g0 * w_normal / out[4];
  float sd = 1.f, sg = 1.f, md = 0.f, mg = 0.f, cd = 0.f;
  if (use_depth) {
    const Global G = global_stats(stat, H, W, s_red);
    const RowStat st = stat[r];
    md = st.mean_d; mg = st.mean_g;
    sd = st.std_d + 0.01f * G.std_d;
    sg = st.std_g + 0.01f * G.std_g;
    cd = g0 * w_depth / out[2];
  }
  float A = 0.f, B = 0.f;
  for (int j = threadIdx.x; j < W; j += PB) {
    const size_t p = base + j;
    const bool f = face[p] != 0, h = hair[p] != 0, mo = mouth[p] != 0;
    const float m = ((f || h) != mo) ? cn : 0.f;
    d_normal[p] = m * gt_normal[p];
    d_normal[P + p] = m * gt_normal[P + p];
    d_normal[2 * P + p] = m * gt_normal[2 * P + p];
    if (use_depth && (f != mo)) {
      const float e = depth[p] - md;
      const float g = cd * sgn(e / sd - (gt_depth[p] - mg) / sg);
      A += g; B += g * e;
    }
  }
  if (use_depth) {
    A = block_sum(A, s_red);
    B = block_sum(B, s_red);
    if (threadIdx.x == 0) { rowb[2 * r] = A; rowb[2 * r + 1] = B; }
  }
}

// d loss / d depth[i] = g_i / s_r - A_r / (W s_r) - (d_i - mean_r) / ((W-1) std_r) * B_r / s_r^2
//                       - 0.01 (d_i - M) / ((P-1) std_all) * sum_r' B_r' / s_r'^2
__global__ void __launch_bounds__(PB)
prior_backward_apply_kernel(const float* __restrict__ go, const float* __restrict__ depth,
                            const float* __restrict__ gt_depth, const uint8_t* __restrict__ face,
                            const uint8_t* __restrict__ mouth, int H, int W, float w_depth,
                            const RowStat* __restrict__ stat, const float* __restrict__ out,
                            const float* __restrict__ rowb, float* __restrict__ d_depth) {
  __shared__ float s_red[4];
  const int r = blockIdx.x;
  const size_t base = (size_t)r * W;
  const Global G = global_stats(stat, H, W, s_red);
  float c = 0.f;
  for (int q = threadIdx.x; q < H; q += PB) {
    const float s = stat[q].std_d + 0.01f * G.std_d;
    c += rowb[2 * q + 1] / (s * s);
  }
  const float C = block_sum(c, s_red);
  const RowStat st = stat[r];
  const float sd = st.std_d + 0.01f * G.std_d, sg = st.std_g + 0.01f * G.std_g;
  const float cd = go[0] * w_depth / out[2];
  const float A = rowb[2 * r], B = rowb[2 * r + 1];
  const float k_row = st.std_d > 0.f ? B / (sd * sd) / ((float)max(W - 1, 1) * st.std_d) : 0.f;
  const float k_all = G.std_d > 0.f ? 0.01f * C / ((float)max((long)H * W - 1, 1L) * G.std_d) : 0.f;
  for (int j = threadIdx.x; j < W; j += PB) {
    const size_t p = base + j;
    const float e = depth[p] - st.mean_d;
    float g = 0.f;
    if ((face[p] != 0) != (mouth[p] != 0)) g = cd * sgn(e / sd - (gt_depth[p] - st.mean_g) / sg);
    d_depth[p] = g / sd - A / ((float)W * sd) - e * k_row - (depth[p] - G.mean_d) * k_all;
  }
}

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

/* floats of the `stat` / `parts` / `rowb` workspaces for an H-row image: 8H, 4H, 2H; `out` holds 5 floats
 * [loss, S_depth, N_sel, S_normal, N_m]. */
int instag_geometry_prior_forward(const float* normal, const float* depth, const float* gt_normal,
                                  const float* gt_depth, const uint8_t* face_mask, const uint8_t* hair_mask,
                                  const uint8_t* mouth_mask, int32_t H, int32_t W, int32_t use_depth, float w_normal,
                                  float w_depth, float* stat, float* parts, float* out, instag_stream_t stream) {
  INSTAG_REQUIRE(normal && gt_normal && face_mask && hair_mask && mouth_mask && parts && out, "geometry_prior: NULL tensor");
  INSTAG_REQUIRE(!use_depth || (depth && gt_depth && stat), "geometry_prior: the depth term needs depth, gt_depth, stat");
  INSTAG_REQUIRE(H >= 1 && W >= 1, "geometry_prior: empty image");
  hipStream_t s = (hipStream_t)stream;
  if (use_depth) {
    prior_stats_kernel<<<H, PB, 0, s>>>(depth, gt_depth, H, W, (RowStat*)stat);
    INSTAG_CHECK_LAUNCH();
  }
  prior_loss_kernel<<<H, PB, 0, s>>>(normal, depth, gt_normal, gt_depth, face_mask, hair_mask, mouth_mask, H, W,
                                     use_depth, (const RowStat*)stat, parts);
  INSTAG_CHECK_LAUNCH();
  prior_finalize_kernel<<<1, PB, 0, s>>>(parts, H, use_depth, w_normal, w_depth, out);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

/* g_loss: device scalar (upstream gradient of the loss); d_normal [3,H,W] is always written, d_depth [H,W] when
 * use_depth. */
int instag_geometry_prior_backward(const float* g_loss, const float* depth, const float* gt_normal,
                                   const float* gt_depth, const uint8_t* face_mask, const uint8_t* hair_mask,
                                   const uint8_t* mouth_mask, int32_t H, int32_t W, int32_t use_depth, float w_normal,
                                   float w_depth, const float* stat, const float* out, float* rowb, float* d_normal,
                                   float* d_depth, instag_stream_t stream) {
  INSTAG_REQUIRE(g_loss && gt_normal && face_mask && hair_mask && mouth_mask && out && d_normal, "geometry_prior: NULL tensor");
  INSTAG_REQUIRE(!use_depth || (depth && gt_depth && stat && rowb && d_depth), "geometry_prior: the depth term needs its buffers");
  hipStream_t s = (hipStream_t)stream;
  prior_backward_rows_kernel<<<H, PB, 0, s>>>(g_loss, depth, gt_normal, gt_depth, face_mask, hair_mask, mouth_mask, H, W,
                                              use_depth, w_normal, w_depth, (const RowStat*)stat, out, rowb, d_normal);
  INSTAG_CHECK_LAUNCH();
  if (use_depth) {
    prior_backward_apply_kernel<<<H, PB, 0, s>>>(g_loss, depth, gt_depth, face_mask, mouth_mask, H, W, w_depth,
                                                 (const RowStat*)stat, out, rowb, d_depth);
    INSTAG_CHECK_LAUNCH();
  }
  return INSTAG_OK;
}

}  // extern "C"
