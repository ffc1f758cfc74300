// Per-Gaussian stage of the rasterizer: projection, EWA covariance, tile rectangle, SH->RGB,
// plus instance duplication and tile-range identification.
//
// This translation unit is compiled with -ffp-contract=off and every fp32 expression is written
// in the exact order of oracle/rasterize_ref.py::preprocess so that radii, tile rectangles,
// depths and therefore sort keys / bin counts are BIT-EXACT against the CPU oracle.
//
// Replaces the per-Gaussian stage of the reference's absent `diff_gauss` extension (call sites
// gaussian_renderer/__init__.py:58-73,111-121); semantics = published 3DGS preprocess.
#include "raster_internal.hpp"

namespace instag {

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

// Exact tile culling.  A Gaussian can only contribute to a pixel where alpha = o*exp(-q/2) >= 1/255, i.e. where
// q(d) = A dx^2 + 2 B dx dy + C dy^2 <= 2 ln(255 o).  A tile of the bounding rectangle is kept iff the minimum of q
// over the rectangle spanned by the tile's pixel centres is <= thr = 2 ln(255 o) * 1.001 + 0.001 (the margin keeps
// the test conservative under fp32 rounding: no contributing (tile, Gaussian) pair is ever dropped, so images and
// gradients are unchanged, only the instance lists get shorter).  Same fp32 operation order as
// oracle/rasterize_ref.py::tile_keep_mask.
__device__ __forceinline__ bool tile_kept(float px, float py, float A, float B, float C, float thr, int tx, int ty) {
  const float x0 = (float)(tx * TILE_X), y0 = (float)(ty * TILE_Y);
  const float dxl = x0 - px, dxr = (x0 + (float)(TILE_X - 1)) - px;
  const float dyl = y0 - py, dyr = (y0 + (float)(TILE_Y - 1)) - py;
  if (dxl <= 0.f && dxr >= 0.f && dyl <= 0.f && dyr >= 0.f) return true;   // centre inside the tile
  const float B2 = 2.0f * B;
  const float ya = fminf(fmaxf(-(B * dxl) / C, dyl), dyr);
  const float yb = fminf(fmaxf(-(B * dxr) / C, dyl), dyr);
  const float xa = fminf(fmaxf(-(B * dyl) / A, dxl), dxr);
  const float xb = fminf(fmaxf(-(B * dyr) / A, dxl), dxr);
  const float e1 = ((A * dxl) * dxl + (B2 * dxl) * ya) + (C * ya) * ya;
  const float e2 = ((A * dxr) * dxr + (B2 * dxr) * yb) + (C * yb) * yb;
  const float e3 = ((A * xa) * xa + (B2 * xa) * dyl) + (C * dyl) * dyl;
  const float e4 = ((A * xb) * xb + (B2 * xb) * dyr) + (C * dyr) * dyr;
  return fminf(fminf(e1, e2), fminf(e3, e4)) <= thr;
}

// ln(x), x > 0, from IEEE + - * / only (this TU is compiled with -ffp-contract=off): the CPU oracle evaluates the same
// expression tree (oracle/rasterize_ref.py::_det_ln), so the culling threshold -- and with it the kept set -- is
// bit-identical on any host, whatever its libm's log() rounds to.  x = m 2^e with m in [sqrt(1/2), sqrt(2)),
// ln m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716: fourteen terms of the odd series reach 1e-17.
__device__ __forceinline__ double det_ln(double x) {
  int e;
  double m = frexp(x, &e);
  if (m < 0.7071067811865476) { m = m * 2.0; e -= 1; }
  const double s = (m - 1.0) / (m + 1.0);
  const double t = s * s;
  double p = 1.0 / 27.0;
  p = p * t + 1.0 / 25.0; p = p * t + 1.0 / 23.0; p = p * t + 1.0 / 21.0; p = p * t + 1.0 / 19.0;
  p = p * t + 1.0 / 17.0; p = p * t + 1.0 / 15.0; p = p * t + 1.0 / 13.0; p = p * t + 1.0 / 11.0;
  p = p * t + 1.0 / 9.0;  p = p * t + 1.0 / 7.0;  p = p * t + 1.0 / 5.0;  p = p * t + 1.0 / 3.0;
  p = p * t + 1.0;
  return (double)e * 0.6931471805599453 + (2.0 * s) * p;
}

struct PreIn {
  const float *means3D, *shs, *colors, *opac, *scales, *rots, *cov3Dp, *extra, *shs_rest;
};

__global__ void __launch_bounds__(256)
preprocess_kernel(Camera c, PreIn in, float* __restrict__ rec2d, float* __restrict__ cov3d,
                  uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ flags_out,
                  float* __restrict__ cull_thr, int32_t* __restrict__ radii,
                  uint32_t* __restrict__ zero_words, uint32_t n_zero_words) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  // housekeeping for a later launch: the look-back state of the instance-offset scan starts out zero
  for (uint32_t k = (uint32_t)g; k < n_zero_words; k += gridDim.x * 256u) zero_words[k] = 0u;
  if (g >= c.N) return;
  radii[g] = 0;
  tiles_touched[g] = 0;
  const float* __restrict__ V = c.view;
  const float* __restrict__ P = c.proj;
  const float px = in.means3D[3 * g + 0], py = in.means3D[3 * g + 1], pz = in.means3D[3 * g + 2];

  const float tx = ((V[0] * px + V[4] * py) + V[8] * pz) + V[12];
  const float ty = ((V[1] * px + V[5] * py) + V[9] * pz) + V[13];
  const float tz = ((V[2] * px + V[6] * py) + V[10] * pz) + V[14];
  if (!(tz > 0.2f)) return;  // near-plane cull

  const float hx = ((P[0] * px + P[4] * py) + P[8] * pz) + P[12];
  const float hy = ((P[1] * px + P[5] * py) + P[9] * pz) + P[13];
  const float hw = ((P[3] * px + P[7] * py) + P[11] * pz) + P[15];
  const float p_w = 1.0f / (hw + 1e-7f);
  const float ndc_x = hx * p_w;
  const float ndc_y = hy * p_w;
  const float pix_x = ((ndc_x + 1.0f) * (float)c.W - 1.0f) * 0.5f;
  const float pix_y = ((ndc_y + 1.0f) * (float)c.H - 1.0f) * 0.5f;

  uint32_t flags = 0;
  float S00, S01, S02, S11, S12, S22;
  float nvx = 0.f, nvy = 0.f, nvz = 0.f;
  if (in.cov3Dp) {
    S00 = in.cov3Dp[6 * g + 0]; S01 = in.cov3Dp[6 * g + 1]; S02 = in.cov3Dp[6 * g + 2];
    S11 = in.cov3Dp[6 * g + 3]; S12 = in.cov3Dp[6 * g + 4]; S22 = in.cov3Dp[6 * g + 5];
  } else {
    const float sx = c.scale_modifier * in.scales[3 * g + 0];
    const float sy = c.scale_modifier * in.scales[3 * g + 1];
    const float sz = c.scale_modifier * in.scales[3 * g + 2];
    const float r = in.rots[4 * g + 0], x = in.rots[4 * g + 1], y = in.rots[4 * g + 2], z = in.rots[4 * g + 3];
    const float R00 = 1.0f - 2.0f * (y * y + z * z);
    const float R01 = 2.0f * (x * y - r * z);
    const float R02 = 2.0f * (x * z + r * y);
    const float R10 = 2.0f * (x * y + r * z);
    const float R11 = 1.0f - 2.0f * (x * x + z * z);
    const float R12 = 2.0f * (y * z - r * x);
    const float R20 = 2.0f * (x * z - r * y);
    const float R21 = 2.0f * (y * z + r * x);
    const float R22 = 1.0f - 2.0f * (x * x + y * y);
    const float M00 = R00 * sx, M01 = R01 * sy, M02 = R02 * sz;
    const float M10 = R10 * sx, M11 = R11 * sy, M12 = R12 * sz;
    const float M20 = R20 * sx, M21 = R21 * sy, M22 = R22 * sz;
    S00 = (M00 * M00 + M01 * M01) + M02 * M02;
    S01 = (M00 * M10 + M01 * M11) + M02 * M12;
    S02 = (M00 * M20 + M01 * M21) + M02 * M22;
    S11 = (M10 * M10 + M11 * M11) + M12 * M12;
    S12 = (M10 * M20 + M11 * M21) + M12 * M22;
    S22 = (M20 * M20 + M21 * M21) + M22 * M22;
    // shortest axis (first minimum) = column k of R, rotated into view space, facing the camera
    const bool k0 = (sx <= sy) && (sx <= sz);
    const bool k1 = (!k0) && (sy <= sz);
    const int k = k0 ? 0 : (k1 ? 1 : 2);
    const float nwx = k0 ? R00 : (k1 ? R01 : R02);
    const float nwy = k0 ? R10 : (k1 ? R11 : R12);
    const float nwz = k0 ? R20 : (k1 ? R21 : R22);
    nvx = (nwx * V[0] + nwy * V[4]) + nwz * V[8];
    nvy = (nwx * V[1] + nwy * V[5]) + nwz * V[9];
    nvz = (nwx * V[2] + nwy * V[6]) + nwz * V[10];
    const bool away = ((nvx * tx + nvy * ty) + nvz * tz) > 0.f;
    const float sgn = away ? -1.0f : 1.0f;
    nvx *= sgn; nvy *= sgn; nvz *= sgn;
    flags |= (uint32_t)k << 3;
    if (away) flags |= 1u << 5;
  }

  // EWA splat covariance
  const float limx = 1.3f * c.tanfovx;
  const float limy = 1.3f * c.tanfovy;
  const float txtz = tx / tz;
  const float tytz = ty / tz;
  if (!(txtz >= -limx && txtz <= limx)) flags |= 1u << 6;
  if (!(tytz >= -limy && tytz <= limy)) flags |= 1u << 7;
  const float txc = fminf(limx, fmaxf(-limx, txtz)) * tz;
  const float tyc = fminf(limy, fmaxf(-limy, tytz)) * tz;
  const float J00 = c.focal_x / tz;
  const float J02 = -(c.focal_x * txc) / (tz * tz);
  const float J11 = c.focal_y / tz;
  const float J12 = -(c.focal_y * tyc) / (tz * tz);
  const float T00 = J00 * V[0] + J02 * V[2];
  const float T01 = J00 * V[4] + J02 * V[6];
  const float T02 = J00 * V[8] + J02 * V[10];
  const float T10 = J11 * V[1] + J12 * V[2];
  const float T11 = J11 * V[5] + J12 * V[6];
  const float T12 = J11 * V[9] + J12 * V[10];
  const float U00 = (S00 * T00 + S01 * T01) + S02 * T02;
  const float U10 = (S01 * T00 + S11 * T01) + S12 * T02;
  const float U20 = (S02 * T00 + S12 * T01) + S22 * T02;
  const float U01 = (S00 * T10 + S01 * T11) + S02 * T12;
  const float U11 = (S01 * T10 + S11 * T11) + S12 * T12;
  const float U21 = (S02 * T10 + S12 * T11) + S22 * T12;
  const float c00 = ((T00 * U00 + T01 * U10) + T02 * U20) + 0.3f;
  const float c01 = (T00 * U01 + T01 * U11) + T02 * U21;
  const float c11 = ((T10 * U01 + T11 * U11) + T12 * U21) + 0.3f;

  const float det = c00 * c11 - c01 * c01;
  if (det == 0.0f) return;
  const float det_inv = 1.0f / det;
  const float conA = c11 * det_inv;
  const float conB = -c01 * det_inv;
  const float conC = c00 * det_inv;
  const float mid = 0.5f * (c00 + c11);
  const float sq = sqrtf(fmaxf(mid * mid - det, 0.1f));
  const float lam = fmaxf(mid + sq, mid - sq);
  const float radius_f = ceilf(3.0f * sqrtf(lam));
  if (!(radius_f > 0.0f) || !isfinite(radius_f) || !isfinite(pix_x) || !isfinite(pix_y)) return;
  const int radius = (int)radius_f;
  const float rf = (float)radius;
  int rminx = min(c.grid_x, max(0, (int)((pix_x - rf) / (float)TILE_X)));
  int rminy = min(c.grid_y, max(0, (int)((pix_y - rf) / (float)TILE_Y)));
  int rmaxx = min(c.grid_x, max(0, (int)((((pix_x + rf) + (float)TILE_X) - 1.0f) / (float)TILE_X)));
  int rmaxy = min(c.grid_y, max(0, (int)((((pix_y + rf) + (float)TILE_Y) - 1.0f) / (float)TILE_Y)));
  const int tiles = (rmaxx - rminx) * (rmaxy - rminy);
  if (tiles <= 0) return;

  float cr, cg, cb;
  if (in.shs) {
    // the active coefficients in registers, from the concatenated [N,M,3] tensor or from the (dc | rest) pair
    float sh[48];
    {
      const int nsh = (c.sh_degree + 1) * (c.sh_degree + 1);
      const float* __restrict__ s0 = in.shs_rest ? in.shs + (size_t)g * 3 : in.shs + (size_t)g * c.M * 3;
      const float* __restrict__ s1 = in.shs_rest ? in.shs_rest + (size_t)g * (c.M - 1) * 3 - 3 : s0;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (m < nsh) {
          const float* src = m == 0 ? s0 : s1 + 3 * m;
          sh[3 * m] = src[0]; sh[3 * m + 1] = src[1]; sh[3 * m + 2] = src[2];
        }
      }
    }
    const float dx = px - c.campos[0], dy = py - c.campos[1], dz = pz - c.campos[2];
    const float ln = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float x = dx / ln, y = dy / ln, z = dz / ln;
    float res[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float v = SH_C0 * sh[ch];
      if (c.sh_degree > 0) {
        v = v - SH_C1 * y * sh[3 + ch] + SH_C1 * z * sh[6 + ch] - SH_C1 * x * sh[9 + ch];
        if (c.sh_degree > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          v = v + SH_C2[0] * xy * sh[12 + ch] + SH_C2[1] * yz * sh[15 + ch] +
              SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + SH_C2[3] * xz * sh[21 + ch] +
              SH_C2[4] * (xx - yy) * sh[24 + ch];
          if (c.sh_degree > 2) {
            v = v + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] + SH_C3[1] * xy * z * sh[30 + ch] +
                SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] + SH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
          }
        }
      }
      v = v + 0.5f;
      if (v < 0.0f) flags |= 1u << ch;
      res[ch] = fmaxf(v, 0.0f);
    }
    cr = res[0]; cg = res[1]; cb = res[2];
  } else {
    cr = in.colors[3 * g + 0]; cg = in.colors[3 * g + 1]; cb = in.colors[3 * g + 2];
  }

  // instances = tiles of the rectangle that the alpha >= 1/255 ellipse can reach
  const float op = in.opac[g];
  const float thr = (float)((2.0 * det_ln(255.0 * (double)op)) * 1.001 + 0.001);
  int kept = 0;
  if (thr >= 0.0f) {
    for (int ty_ = rminy; ty_ < rmaxy; ++ty_)
      for (int tx_ = rminx; tx_ < rmaxx; ++tx_) kept += tile_kept(pix_x, pix_y, conA, conB, conC, thr, tx_, ty_) ? 1 : 0;
  }
  cull_thr[g] = thr;
  radii[g] = radius;
  tiles_touched[g] = (uint32_t)kept;
  flags_out[g] = flags | ((uint32_t)(rmaxy - rminy) << 16);   // bits 16..31: rectangle height in tiles
  float* c3 = cov3d + (size_t)g * 6;
  c3[0] = S00; c3[1] = S01; c3[2] = S02; c3[3] = S11; c3[4] = S12; c3[5] = S22;

  float4* rec = reinterpret_cast<float4*>(rec2d + (size_t)g * REC_FLOATS);
  const uint32_t rect = (uint32_t)rminx | ((uint32_t)rminy << 10) | ((uint32_t)(rmaxx - rminx) << 20);
  rec[0] = make_float4(pix_x, pix_y, conA, conB);
  rec[1] = make_float4(conC, op, cr, cg);
  rec[2] = make_float4(cb, tz, nvx, nvy);
  rec[3] = make_float4(nvz, (c.E > 0 && in.extra) ? in.extra[g] : 0.0f, 0.0f, __uint_as_float(rect));
}

// Depth keys of ALL Gaussians (the float bits of view-space z, the same expression as in preprocess_kernel: this TU is
// compiled without FMA contraction) + the per-block digit histograms of the four 8-bit passes of the depth sort + the
// clearing of that sort's tickets and look-back words.  It depends on means3D and the view matrix only, so the whole
// depth sort runs BESIDE preprocess_kernel on a second stream.  Culled Gaussians (z <= 0.2, no kept tile) take part in
// the sort, wherever they land: they emit no instance, so the instance order -- Gaussians with instances by
// (depth bits, index) -- is that of the published (tile << 32 | depth) key sort.
constexpr int DK_IPT = 4, DK_THREADS = 1024;   // 4,096 keys per block: 25 partial histograms for 100k Gaussians
__global__ void __launch_bounds__(DK_THREADS)
depth_key_kernel(int N, const float* __restrict__ view, const float* __restrict__ means3D,
                 uint32_t* __restrict__ depth_key, uint32_t* __restrict__ partials,
                 uint32_t* __restrict__ zero_words, uint32_t n_zero_words) {
  __shared__ uint32_t s_h[4][256];
  const int tid = threadIdx.x;
  for (int k = tid; k < 4 * 256; k += DK_THREADS) (&s_h[0][0])[k] = 0u;
  for (uint32_t k = blockIdx.x * DK_THREADS + tid; k < n_zero_words; k += gridDim.x * DK_THREADS) zero_words[k] = 0u;
  __syncthreads();
  const float v2 = view[2], v6 = view[6], v10 = view[10], v14 = view[14];
  float pos[DK_IPT][3];
#pragma unroll
  for (int k = 0; k < DK_IPT; ++k) {                      // unconditional loads (index clamped), back to back
    const int g = min((blockIdx.x * DK_IPT + k) * DK_THREADS + tid, N - 1);
    pos[k][0] = means3D[3 * g + 0]; pos[k][1] = means3D[3 * g + 1]; pos[k][2] = means3D[3 * g + 2];
  }
#pragma unroll
  for (int k = 0; k < DK_IPT; ++k) {
    const int g = (blockIdx.x * DK_IPT + k) * DK_THREADS + tid;
    if (g < N) {
      const float tz = ((v2 * pos[k][0] + v6 * pos[k][1]) + v10 * pos[k][2]) + v14;
      const uint32_t key = __float_as_uint(tz);
      depth_key[g] = key;
      atomicAdd(&s_h[0][key & 255u], 1u);
      atomicAdd(&s_h[1][(key >> 8) & 255u], 1u);
      atomicAdd(&s_h[2][(key >> 16) & 255u], 1u);
      atomicAdd(&s_h[3][key >> 24], 1u);
    }
  }
  __syncthreads();
  uint32_t* out = partials + (size_t)blockIdx.x * 4 * 256;
  for (int k = tid; k < 4 * 256; k += DK_THREADS) out[k] = (&s_h[0][0])[k];
}

// Sixteen lanes per Gaussian, Gaussians in DEPTH ORDER: emit the tile id of every KEPT tile of its rectangle (the
// depth half of the published (tile<<32 | depth) key is implied by the emission order + a stable sort).  Lane q tests
// tile q, q+16, ... of the rectangle (row-major, the emission order), the kept ones are compacted with a ballot.
// The sorted value is the instance's own unsorted slot u (also the row of its gradient in blend-backward);
// gid_unsorted[u] maps the slot back to the Gaussian.  The exclusive instance offset is recorded in the blend record.
//
// The kernel also prepares the instance sort that follows it: per-block histograms of the tile-id digits (LDS integer
// atomics, written out with plain stores: the sort needs no zero-initialised global counters), the clearing of the
// sort's tickets / look-back words and of the tile ranges, the number of instances to sort (a device word: unused
// capacity is neither padded nor sorted) and, in capacity mode, the status words.
constexpr int DUP_LANES = 16;
constexpr int DUP_PER_BLOCK = 256 / DUP_LANES;

__global__ void __launch_bounds__(256)
duplicate_kernel(int N, int ngroups, int grid_x, float* __restrict__ rec2d, const uint32_t* __restrict__ order,
                 const uint32_t* __restrict__ point_offsets,
                 const uint32_t* __restrict__ flags, const float* __restrict__ cull_thr,
                 uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ gid_unsorted,
                 uint32_t capacity, int32_t* __restrict__ ranges, uint32_t nranges, int packed,
                 int32_t* __restrict__ status, uint32_t* __restrict__ sort_count, TilePasses tp,
                 uint32_t* __restrict__ partials, uint32_t* __restrict__ zero_words, uint32_t n_zero_words) {
  __shared__ uint32_t s_dh[3][256];
  const uint32_t gtid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t total = gridDim.x * 256u;
  for (int k = threadIdx.x; k < 3 * 256; k += 256) (&s_dh[0][0])[k] = 0u;
  if (gtid == 0) {
    // instances that fit: all of them, or (capacity mode, overflow) those of the Gaussians in front of the first one
    // whose slots would cross the capacity -- the offsets are non-decreasing, so a binary search finds it
    const uint32_t R = point_offsets[N - 1];
    uint32_t used = R;
    if (R > capacity) {
      int lo = 0, hi = N;                         // first rank whose inclusive offset exceeds the capacity
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (point_offsets[mid] > capacity) hi = mid; else lo = mid + 1;
      }
      used = lo > 0 ? point_offsets[lo - 1] : 0u;
    }
    *sort_count = used;
    if (status != nullptr) {
      // [0] instances needed by this call, [1] STICKY overflow flag (the host clears it), [2] largest need seen since
      // the host last cleared it, [3] instances actually binned by this call
      status[0] = (int32_t)R;
      status[1] = status[1] | (R > capacity ? 1 : 0);
      status[2] = max(status[2], (int32_t)R);
      status[3] = (int32_t)used;
    }
  }
  // housekeeping that used to be memset nodes: tile ranges start out empty, the sort's tickets / look-back words zero
  for (uint32_t k = gtid; k < nranges; k += total) ranges[k] = 0;
  for (uint32_t k = gtid; k < n_zero_words; k += total) zero_words[k] = 0u;
  __syncthreads();

  const int q = threadIdx.x % DUP_LANES;
  const int grp_shift = ((threadIdx.x & 63) / DUP_LANES) * DUP_LANES;        // position of the group's bits in a ballot
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int i = grp * DUP_PER_BLOCK + (threadIdx.x / DUP_LANES);          // rank in depth order
    uint32_t tt = 0, off = 0, g = 0;
    bool work = i < N, fits = false;
    if (work) {
      const uint32_t po = point_offsets[i];
      tt = po - (i > 0 ? point_offsets[i - 1] : 0u);                         // kept tiles of this Gaussian
      off = po - tt;
      fits = po <= capacity;                  // capacity mode: instances beyond the buffer are dropped (flagged)
      work = tt != 0;
    }
    int rminx = 0, rminy = 0, rw = 1, ntiles = 0;
    float px = 0.f, py = 0.f, A = 0.f, B = 0.f, C = 0.f, thr = -1.f;
    if (work) {
      g = order[i];
      float* rec = rec2d + (size_t)g * REC_FLOATS;
      // (also for a dropped Gaussian: the backward recognises it by offset + count > capacity)
      if (q == 0) rec[R_OFFSET] = __uint_as_float(off);
      if (fits) {
        const uint32_t rect = __float_as_uint(rec[R_RECT]);
        rminx = (int)(rect & 1023u); rminy = (int)((rect >> 10) & 1023u); rw = (int)(rect >> 20);
        ntiles = rw * (int)(flags[g] >> 16);
        px = rec[R_X]; py = rec[R_Y]; A = rec[R_CA]; B = rec[R_CB]; C = rec[R_CC]; thr = cull_thr[g];
      }
    }
    // (all 16 lanes of a group share `ntiles`; groups of one wave may differ: a finished group's ballot bits are 0)
    for (int t0 = 0; t0 < ntiles; t0 += DUP_LANES) {
      const int t = t0 + q;
      const int y = rminy + t / rw, x = rminx + t % rw;
      const bool kept = t < ntiles && tile_kept(px, py, A, B, C, thr, x, y);
      const uint32_t bits = (uint32_t)(__builtin_amdgcn_ballot_w64(kept) >> grp_shift) & ((1u << DUP_LANES) - 1u);
      if (kept) {
        const uint32_t o = off + (uint32_t)__builtin_popcount(bits & ((1u << q) - 1u));
        const uint32_t tile = (uint32_t)(y * grid_x + x);
        if (packed) {
          keys[o] = (tile << PACK_SHIFT) | o;              // key-only sort: the slot rides in the low bits
        } else {
          keys[o] = tile;
          vals[o] = o;
        }
        gid_unsorted[o] = g;
        for (int p = 0; p < tp.npass; ++p)
          atomicAdd(&s_dh[p][(tile >> (p * tp.bits_per)) & ((1u << tp.nbits[p]) - 1u)], 1u);
      }
      off += (uint32_t)__builtin_popcount(bits);
    }
  }
  __syncthreads();
  uint32_t* out = partials + (size_t)blockIdx.x * tp.npass * 256;
  for (int k = threadIdx.x; k < tp.npass * 256; k += 256) out[k] = (&s_dh[0][0])[k];
}

// debug / parity: the 64-bit (tile<<32 | depth bits) key of every sorted instance
__global__ void __launch_bounds__(256)
export_keys_kernel(int64_t R, const uint32_t* __restrict__ tile_keys, const uint32_t* __restrict__ point_list,
                   const float* __restrict__ rec2d, uint64_t* __restrict__ keys64, int packed) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const uint32_t depth_bits = __float_as_uint(rec2d[(size_t)point_list[i] * REC_FLOATS + R_DEPTH]);
  const uint32_t tile = packed ? tile_keys[i] >> PACK_SHIFT : tile_keys[i];
  keys64[i] = ((uint64_t)tile << 32) | depth_bits;
}

// Four consecutive instances per thread: one 16-byte load of the sorted keys, four independent gathers of the Gaussian
// ids in flight, 16-byte stores (one element per thread, each with two dependent loads in a row, took 11.6 us for 1.06 M
// instances -- latency, not bytes).
__global__ void __launch_bounds__(256)
ranges_kernel(const uint32_t* __restrict__ count_ptr, const uint32_t* __restrict__ keys,
              uint32_t* __restrict__ slots_sorted,
              const uint32_t* __restrict__ gid_unsorted, uint32_t* __restrict__ point_list,
              int32_t* __restrict__ ranges, uint32_t ntiles, int packed) {
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t R = (int64_t)*count_ptr;         // instances actually binned (device word written by duplicate_kernel)
  if (i0 >= R) return;
  auto tile_of = [&](uint32_t k) { return packed ? k >> PACK_SHIFT : k; };
  const int n = (int)min((int64_t)4, R - i0);
  uint32_t key[4], slot[4], gid[4];
  if (n == 4) {
    const uint4 k4 = *reinterpret_cast<const uint4*>(keys + i0);
    key[0] = k4.x; key[1] = k4.y; key[2] = k4.z; key[3] = k4.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) key[j] = j < n ? keys[i0 + j] : 0xFFFFFFFFu;
  }
  const uint32_t prev_key = i0 > 0 ? keys[i0 - 1] : 0u;
  if (packed) {
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[j] = key[j] & ((1u << PACK_SHIFT) - 1u);
  } else if (n == 4) {
    const uint4 s4 = *reinterpret_cast<const uint4*>(slots_sorted + i0);
    slot[0] = s4.x; slot[1] = s4.y; slot[2] = s4.z; slot[3] = s4.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[j] = j < n ? slots_sorted[i0 + j] : 0u;
  }
  bool in_range[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    in_range[j] = j < n && tile_of(key[j]) < ntiles;
    gid[j] = in_range[j] ? gid_unsorted[slot[j]] : 0u;          // four independent gathers
  }
  if (n == 4 && in_range[0] && in_range[1] && in_range[2] && in_range[3]) {
    *reinterpret_cast<uint4*>(point_list + i0) = make_uint4(gid[0], gid[1], gid[2], gid[3]);
    if (packed) *reinterpret_cast<uint4*>(slots_sorted + i0) = make_uint4(slot[0], slot[1], slot[2], slot[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (in_range[j]) {
        point_list[i0 + j] = gid[j];
        if (packed) slots_sorted[i0 + j] = slot[j];
      }
  }
  uint32_t prev = tile_of(prev_key);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j >= n) break;
    const int64_t i = i0 + j;
    const uint32_t tile = tile_of(key[j]);
    if (i == 0) {
      if (tile < ntiles) ranges[2 * tile] = 0;
    } else if (prev != tile) {
      if (prev < ntiles) ranges[2 * prev + 1] = (int32_t)i;
      if (tile < ntiles) ranges[2 * tile] = (int32_t)i;
    }
    if (i == R - 1 && tile < ntiles) ranges[2 * tile + 1] = (int32_t)R;
    prev = tile;
  }
}


}  // namespace

Camera make_camera(const instag_raster_args* a) {
  Camera c;
  c.N = a->N; c.M = a->M; c.sh_degree = a->sh_degree; c.E = a->E;
  c.H = a->image_height; c.W = a->image_width;
  c.tanfovx = a->tanfovx; c.tanfovy = a->tanfovy;
  c.focal_x = (float)c.W / (2.0f * a->tanfovx);
  c.focal_y = (float)c.H / (2.0f * a->tanfovy);
  c.scale_modifier = a->scale_modifier;
  c.grid_x = (c.W + TILE_X - 1) / TILE_X;
  c.grid_y = (c.H + TILE_Y - 1) / TILE_Y;
  c.bg = a->bg; c.view = a->viewmatrix; c.proj = a->projmatrix; c.campos = a->campos;
  return c;
}

int launch_preprocess(const Camera& c, const instag_raster_args* a, float* rec2d, float* cov3d,
                      uint32_t* tiles_touched, uint32_t* flags, float* cull_thr, int32_t* radii,
                      uint32_t* zero_words, uint32_t n_zero_words, hipStream_t s) {
  if (c.N == 0) return INSTAG_OK;
  PreIn in{a->means3D, a->shs, a->colors_precomp, a->opacities, a->scales, a->rotations,
           a->cov3Ds_precomp, a->extra_attrs, a->shs_rest};
  ProfScope p(K_PREPROCESS, s);
  preprocess_kernel<<<div_up(c.N, 256), 256, 0, s>>>(c, in, rec2d, cov3d, tiles_touched, flags, cull_thr, radii,
                                                     zero_words, n_zero_words);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

uint32_t depth_key_blocks(int32_t N) { return (uint32_t)div_up(std::max(N, 1), DK_THREADS * DK_IPT); }

int launch_depth_keys(const Camera& c, const float* means3D, uint32_t* depth_key, uint32_t* partials,
                      uint32_t* zero_words, uint32_t n_zero_words, hipStream_t s) {
  if (c.N == 0) return INSTAG_OK;
  depth_key_kernel<<<depth_key_blocks(c.N), DK_THREADS, 0, s>>>(c.N, c.view, means3D, depth_key, partials, zero_words,
                                                             n_zero_words);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

uint32_t duplicate_blocks(int32_t N) { return (uint32_t)std::min(div_up(std::max(N, 1), DUP_PER_BLOCK), 1024); }

int launch_duplicate(const Camera& c, float* rec2d, const uint32_t* order,
                     const uint32_t* point_offsets, const uint32_t* flags, const float* cull_thr, uint32_t* keys,
                     uint32_t* vals, uint32_t* gid_unsorted, uint32_t capacity, int32_t* ranges,
                     bool packed, int32_t* status, uint32_t* sort_count, const TilePasses& tp, uint32_t* partials,
                     uint32_t* zero_words, uint32_t n_zero_words, hipStream_t s) {
  if (c.N == 0) return INSTAG_OK;
  ProfScope p(K_DUPLICATE, s);
  const int ngroups = div_up(c.N, DUP_PER_BLOCK);
  duplicate_kernel<<<duplicate_blocks(c.N), 256, 0, s>>>(c.N, ngroups, c.grid_x, rec2d, order, point_offsets, flags,
                                                         cull_thr, keys, vals, gid_unsorted, capacity, ranges,
                                                         (uint32_t)(2 * c.grid_x * c.grid_y), packed ? 1 : 0, status,
                                                         sort_count, tp, partials, zero_words, n_zero_words);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int launch_export_keys(int64_t R, const uint32_t* tile_keys, const uint32_t* point_list, const float* rec2d,
                       uint64_t* keys64, bool packed, hipStream_t s) {
  if (R == 0) return INSTAG_OK;
  export_keys_kernel<<<(unsigned)div_up<int64_t>(R, 256), 256, 0, s>>>(R, tile_keys, point_list, rec2d, keys64,
                                                                       packed ? 1 : 0);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int launch_ranges(int64_t R, const uint32_t* count_ptr, const uint32_t* keys_sorted, uint32_t* slots_sorted,
                  const uint32_t* gid_unsorted, uint32_t* point_list, int32_t* ranges, uint32_t ntiles, bool packed,
                  hipStream_t s) {
  if (R == 0) return INSTAG_OK;
  ProfScope p(K_RANGES, s);
  ranges_kernel<<<(unsigned)div_up<int64_t>(R, 1024), 256, 0, s>>>(count_ptr, keys_sorted, slots_sorted, gid_unsorted,
                                                                  point_list, ranges, ntiles, packed ? 1 : 0);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // namespace instag
