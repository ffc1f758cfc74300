// Per-tile front-to-back alpha blending (forward) and its back-to-front gradient pass.
//
// Replaces renderCUDA forward/backward of the reference's absent `diff_gauss` extension
// (call sites gaussian_renderer/__init__.py:111-121); semantics restated in
// oracle/rasterize_ref.py::blend.
//
// Layout: one 256-thread workgroup (4 wave64) per 16x16 tile, one pixel per lane.  Gaussians of
// the tile's depth-sorted list are staged through LDS as 64-byte records (one coalesced 64-B line
// per Gaussian from rec2d) and broadcast to all lanes with conflict-free ds_read_b128.
//
// Backward is deterministic and atomic-free: every (tile, Gaussian) instance owns one 64-byte
// gradient row in `inst_grad` (slot = Gaussian's exclusive instance offset + position of the tile
// inside its rectangle).  Per Gaussian the 14 partial gradients are reduced across the 64 lanes of
// each wave with DPP row shifts / row broadcasts, the four wave partials are combined through LDS
// in a fixed order and the row is written once with plain 16-byte stores.  The per-Gaussian sum
// over its rows happens in raster_backward.hip.
#include "raster_internal.hpp"

namespace instag {
namespace {

constexpr int BLOCK = 256;
constexpr int NCH = 8;  // r g b depth nx ny nz extra
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_MIN = 0.0001f;

__global__ void __launch_bounds__(BLOCK)
blend_forward_kernel(Camera c, const int32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                     const float* __restrict__ rec2d, uint32_t* __restrict__ n_contrib,
                     float* __restrict__ final_T, float* __restrict__ out_color,
                     float* __restrict__ out_depth, float* __restrict__ out_normal,
                     float* __restrict__ out_alpha, float* __restrict__ out_extra) {
  __shared__ float4 s_rec[BLOCK][4];
  const int tile = blockIdx.x;
  const int tx = tile % c.grid_x, ty = tile / c.grid_x;
  const int tid = threadIdx.x;
  const int pxi = tx * TILE_X + (tid & 15), pyi = ty * TILE_Y + (tid >> 4);
  const bool inside = pxi < c.W && pyi < c.H;
  const float pxf = (float)pxi, pyf = (float)pyi;
  const int start = ranges[2 * tile], end = ranges[2 * tile + 1];
  const int rounds = (end - start + BLOCK - 1) / BLOCK;
  int toDo = end - start;

  bool done = !inside;
  float T = 1.0f;
  uint32_t contributor = 0, last_contributor = 0;
  float acc[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) acc[k] = 0.f;

  for (int i = 0; i < rounds; ++i, toDo -= BLOCK) {
    if (__syncthreads_count(done) == BLOCK) break;
    const int progress = i * BLOCK + tid;
    if (start + progress < end) {
      const uint32_t gid = point_list[start + progress];
      const float4* r = reinterpret_cast<const float4*>(rec2d + (size_t)gid * REC_FLOATS);
      s_rec[tid][0] = r[0]; s_rec[tid][1] = r[1]; s_rec[tid][2] = r[2]; s_rec[tid][3] = r[3];
    }
    __syncthreads();
    const int cnt = min(BLOCK, toDo);
    for (int j = 0; !done && j < cnt; ++j) {
      ++contributor;
      const float4 a = s_rec[j][0];  // x y conA conB
      const float4 b = s_rec[j][1];  // conC op r g
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
      if (power > 0.0f) continue;
      const float alpha = fminf(0.99f, b.y * __expf(power));
      if (alpha < ALPHA_MIN) continue;
      const float test_T = T * (1.0f - alpha);
      if (test_T < T_MIN) { done = true; continue; }
      const float w = alpha * T;
      const float4 cc = s_rec[j][2];  // b depth nx ny
      const float4 dd = s_rec[j][3];  // nz extra . .
      acc[0] += b.z * w; acc[1] += b.w * w; acc[2] += cc.x * w; acc[3] += cc.y * w;
      acc[4] += cc.z * w; acc[5] += cc.w * w; acc[6] += dd.x * w; acc[7] += dd.y * w;
      T = test_T;
      last_contributor = contributor;
    }
  }
  if (inside) {
    const size_t P = (size_t)c.H * c.W;
    const size_t pix = (size_t)pyi * c.W + pxi;
    final_T[pix] = T;
    n_contrib[pix] = last_contributor;
    out_color[pix] = acc[0] + T * c.bg[0];
    out_color[P + pix] = acc[1] + T * c.bg[1];
    out_color[2 * P + pix] = acc[2] + T * c.bg[2];
    out_depth[pix] = acc[3];
    out_normal[pix] = acc[4];
    out_normal[P + pix] = acc[5];
    out_normal[2 * P + pix] = acc[6];
    out_alpha[pix] = 1.0f - T;
    if (out_extra) out_extra[pix] = acc[7];
  }
}

// ---- wave64 sum, result valid in lane 63 ------------------------------------------------------
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, BOUND);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0x111, 0xf, true>(v);   // row_shr:1
  v = dpp_add<0x112, 0xf, true>(v);   // row_shr:2
  v = dpp_add<0x114, 0xf, true>(v);   // row_shr:4
  v = dpp_add<0x118, 0xf, true>(v);   // row_shr:8  -> lane 15 of each row = row total
  v = dpp_add<0x142, 0xa, false>(v);  // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xc, false>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = wave total
  return v;
}

constexpr int BB = 128;   // Gaussians per backward batch
constexpr int NG = 14;    // gradient components per instance: x y conA conB conC op r g b depth nx ny nz extra

__global__ void __launch_bounds__(BLOCK)
blend_backward_kernel(Camera c, const int32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const float* __restrict__ rec2d, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ final_T, const float* __restrict__ dL_dcolor,
                      const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dnormal,
                      const float* __restrict__ dL_dalpha_img, const float* __restrict__ dL_dextra,
                      float* __restrict__ inst_grad) {
  __shared__ float4 s_rec[BB][4];
  __shared__ float s_part[4][BB][16];
  __shared__ int s_max[4];
  const int tile = blockIdx.x;
  const int tx = tile % c.grid_x, ty = tile / c.grid_x;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int pxi = tx * TILE_X + (tid & 15), pyi = ty * TILE_Y + (tid >> 4);
  const bool inside = pxi < c.W && pyi < c.H;
  const float pxf = (float)pxi, pyf = (float)pyi;
  const int start = ranges[2 * tile], end = ranges[2 * tile + 1];
  const size_t P = (size_t)c.H * c.W;
  const size_t pix = (size_t)pyi * c.W + pxi;

  const int last_contributor = inside ? (int)n_contrib[pix] : 0;
  // only the first max(last_contributor) Gaussians of the list reached any pixel of this tile
  int m = last_contributor;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if (lane == 0) s_max[wave] = m;
  __syncthreads();
  const int n = min(end - start, max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
  if (n <= 0) return;

  const float T_final = inside ? final_T[pix] : 0.f;
  float T = T_final;
  float dpix[NCH];
  dpix[0] = (inside && dL_dcolor) ? dL_dcolor[pix] : 0.f;
  dpix[1] = (inside && dL_dcolor) ? dL_dcolor[P + pix] : 0.f;
  dpix[2] = (inside && dL_dcolor) ? dL_dcolor[2 * P + pix] : 0.f;
  dpix[3] = (inside && dL_ddepth) ? dL_ddepth[pix] : 0.f;
  dpix[4] = (inside && dL_dnormal) ? dL_dnormal[pix] : 0.f;
  dpix[5] = (inside && dL_dnormal) ? dL_dnormal[P + pix] : 0.f;
  dpix[6] = (inside && dL_dnormal) ? dL_dnormal[2 * P + pix] : 0.f;
  dpix[7] = (inside && dL_dextra) ? dL_dextra[pix] : 0.f;
  const float dalpha_img = (inside && dL_dalpha_img) ? dL_dalpha_img[pix] : 0.f;
  // d(T_final)/d(alpha_i) = -T_final/(1-alpha_i); T_final enters image (+bg) and alpha image (-1)
  const float tail = (c.bg[0] * dpix[0] + c.bg[1] * dpix[1] + c.bg[2] * dpix[2]) - dalpha_img;

  float accum[NCH], last_col[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) { accum[k] = 0.f; last_col[k] = 0.f; }
  float last_alpha = 0.f;

  const uint32_t tile_xy = (uint32_t)tx | ((uint32_t)ty << 16);
  const int rounds = (n + BB - 1) / BB;
  for (int i = 0; i < rounds; ++i) {
    __syncthreads();
    const int base = n - 1 - i * BB;          // list index of batch element 0 (walks backwards)
    const int cnt = min(BB, n - i * BB);
    if (tid < cnt) {
      const uint32_t gid = point_list[start + base - tid];
      const float4* r = reinterpret_cast<const float4*>(rec2d + (size_t)gid * REC_FLOATS);
      s_rec[tid][0] = r[0]; s_rec[tid][1] = r[1]; s_rec[tid][2] = r[2]; s_rec[tid][3] = r[3];
    }
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const int idx = base - j;
      const float4 a = s_rec[j][0];
      const float4 b = s_rec[j][1];
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
      const float G = __expf(power);
      const float alpha = fminf(0.99f, b.y * G);
      const bool valid = (idx < last_contributor) && !(power > 0.0f) && !(alpha < ALPHA_MIN);
      if (__ballot(valid) == 0ull) {
        if (lane < 16) s_part[wave][j][lane] = 0.f;
        continue;
      }
      float g[NG];
#pragma unroll
      for (int k = 0; k < NG; ++k) g[k] = 0.f;
      if (valid) {
        const float4 cc = s_rec[j][2];
        const float4 dd = s_rec[j][3];
        const float col[NCH] = {b.z, b.w, cc.x, cc.y, cc.z, cc.w, dd.x, dd.y};
        T = T / (1.0f - alpha);
        const float w = alpha * T;
        float dL_dalpha = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          accum[k] = last_alpha * last_col[k] + (1.0f - last_alpha) * accum[k];
          last_col[k] = col[k];
          dL_dalpha += (col[k] - accum[k]) * dpix[k];
          g[6 + k] = w * dpix[k];
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.0f - alpha)) * tail;
        const float dL_dG = b.y * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        g[0] = dL_dG * (-gdx * a.z - gdy * a.w);
        g[1] = dL_dG * (-gdy * b.x - gdx * a.w);
        g[2] = -0.5f * gdx * dx * dL_dG;
        g[3] = -gdx * dy * dL_dG;
        g[4] = -0.5f * gdy * dy * dL_dG;
        g[5] = G * dL_dalpha;
      }
#pragma unroll
      for (int k = 0; k < NG; ++k) g[k] = wave_sum_to_lane63(g[k]);
      if (lane == 63) {
        float4* dst = reinterpret_cast<float4*>(&s_part[wave][j][0]);
        dst[0] = make_float4(g[0], g[1], g[2], g[3]);
        dst[1] = make_float4(g[4], g[5], g[6], g[7]);
        dst[2] = make_float4(g[8], g[9], g[10], g[11]);
        dst[3] = make_float4(g[12], g[13], 0.f, 0.f);
      }
    }
    __syncthreads();
    if (tid < cnt) {
      float4 r4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 p0 = reinterpret_cast<const float4*>(&s_part[0][tid][0])[q];
        const float4 p1 = reinterpret_cast<const float4*>(&s_part[1][tid][0])[q];
        const float4 p2 = reinterpret_cast<const float4*>(&s_part[2][tid][0])[q];
        const float4 p3 = reinterpret_cast<const float4*>(&s_part[3][tid][0])[q];
        r4[q] = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                            ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
      }
      const uint32_t off = __float_as_uint(s_rec[tid][3].z);
      const uint32_t rect = __float_as_uint(s_rec[tid][3].w);
      const uint32_t rminx = rect & 1023u, rminy = (rect >> 10) & 1023u, rw = rect >> 20;
      const uint32_t slot = off + ((tile_xy >> 16) - rminy) * rw + ((tile_xy & 0xffffu) - rminx);
      float4* dst = reinterpret_cast<float4*>(inst_grad + (size_t)slot * REC_FLOATS);
      dst[0] = r4[0]; dst[1] = r4[1]; dst[2] = r4[2]; dst[3] = r4[3];
    }
  }
}

}  // namespace

int launch_blend_forward(const Camera& c, const int32_t* ranges, const uint32_t* point_list,
                         const float* rec2d, uint32_t* n_contrib, float* final_T, float* out_color,
                         float* out_depth, float* out_normal, float* out_alpha, float* out_extra,
                         hipStream_t s) {
  const int tiles = c.grid_x * c.grid_y;
  if (tiles == 0) return INSTAG_OK;
  ProfScope p(K_BLEND_FWD, s);
  blend_forward_kernel<<<tiles, BLOCK, 0, s>>>(c, ranges, point_list, rec2d, n_contrib, final_T, out_color,
                                               out_depth, out_normal, out_alpha, out_extra);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int launch_blend_backward(const Camera& c, const int32_t* ranges, const uint32_t* point_list,
                          const float* rec2d, const uint32_t* n_contrib, const float* final_T,
                          const float* dL_dcolor, const float* dL_ddepth, const float* dL_dnormal,
                          const float* dL_dalpha, const float* dL_dextra, float* inst_grad,
                          hipStream_t s) {
  const int tiles = c.grid_x * c.grid_y;
  if (tiles == 0) return INSTAG_OK;
  ProfScope p(K_BLEND_BWD, s);
  blend_backward_kernel<<<tiles, BLOCK, 0, s>>>(c, ranges, point_list, rec2d, n_contrib, final_T, dL_dcolor,
                                                dL_ddepth, dL_dnormal, dL_dalpha, dL_dextra, inst_grad);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // namespace instag
