// Fused L1 + SSIM image loss (forward and backward) for gfx950.
//
// Replaces the eager chain of the reference's loss block: utils/loss_utils.py l1_loss :26-27 and
// ssim :42-72 (five 11x11 depthwise Gaussian convolutions, sigma 1.5, zero padding, C1=0.01^2,
// C2=0.03^2, mean over all pixels), used at train_face.py:450-456.  One workgroup filters one 16x16
// tile of one channel: the 26x26 halo of both images is staged in LDS once, the separable filter runs
// as a horizontal pass into LDS and a vertical pass in registers, the SSIM map is reduced per
// workgroup (fixed order -> deterministic partial sums, final sum by the caller).  Forward also stores
// the three derivative maps dS/dmu1, dS/dE[x^2], dS/dE[xy]; backward filters them with the same
// (symmetric) window and adds the L1 sign term, so the whole loss gradient is ONE kernel.
#include "common.hpp"

namespace instag {
namespace {

constexpr int TS = 16;            // tile side
constexpr int RAD = 5;            // 11 taps
constexpr int HS = TS + 2 * RAD;  // 26
constexpr float C1 = 0.01f * 0.01f;
constexpr float C2 = 0.03f * 0.03f;

// normalised 1-D Gaussian window, size 11, sigma 1.5 (loss_utils.py:33-35)
__device__ constexpr float GW[11] = {0.0010283801f, 0.0075987581f, 0.0360007721f, 0.1093606895f, 0.2130055377f,
                                     0.2660117249f, 0.2130055377f, 0.1093606895f, 0.0360007721f, 0.0075987581f,
                                     0.0010283801f};

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  return ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

__global__ void __launch_bounds__(256)
l1_ssim_forward_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                       float* __restrict__ maps, float* __restrict__ part_ssim, float* __restrict__ part_l1) {
  __shared__ float s_x[HS][HS + 1], s_y[HS][HS + 1];
  __shared__ float s_h[5][HS][TS + 1];
  __shared__ float s_red[4];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const size_t plane = (size_t)H * W;
  const float* p1 = img1 + c * plane;
  const float* p2 = img2 + c * plane;
  for (int i = threadIdx.x; i < HS * HS; i += 256) {
    const int ly = i / HS, lx = i - ly * HS;
    const int gy = y0 + ly - RAD, gx = x0 + lx - RAD;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    s_x[ly][lx] = in ? p1[(size_t)gy * W + gx] : 0.f;
    s_y[ly][lx] = in ? p2[(size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  // horizontal pass: HS rows x TS columns x 5 quantities
  for (int i = threadIdx.x; i < HS * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float xv = s_x[ly][lx + k], yv = s_y[ly][lx + k], w = GW[k];
      a += w * xv; b += w * yv; aa += w * xv * xv; bb += w * yv * yv; ab += w * xv * yv;
    }
    s_h[0][ly][lx] = a; s_h[1][ly][lx] = b; s_h[2][ly][lx] = aa; s_h[3][ly][lx] = bb; s_h[4][ly][lx] = ab;
  }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int gx = x0 + tx, gy = y0 + ty;
  const bool inside = gx < W && gy < H;
  float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = GW[k];
    mu1 += w * s_h[0][ty + k][tx]; mu2 += w * s_h[1][ty + k][tx];
    e11 += w * s_h[2][ty + k][tx]; e22 += w * s_h[3][ty + k][tx]; e12 += w * s_h[4][ty + k][tx];
  }
  float ssim_v = 0.f, l1_v = 0.f;
  if (inside) {
    const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
    const float s11 = e11 - mu1s, s22 = e22 - mu2s, s12 = e12 - mu12;
    const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1s + mu2s + C1, B2 = s11 + s22 + C2;
    const float inv = 1.f / (B1 * B2);
    ssim_v = A1 * A2 * inv;
    // derivatives w.r.t. (mu1 | E[x^2] | E[xy]) with s11 = E11 - mu1^2, s12 = E12 - mu1 mu2
    const float dS_ds11 = -ssim_v / B2;
    const float dS_ds12 = 2.f * A1 * inv;
    const float dS_dmu1 = 2.f * mu2 * A2 * inv - 2.f * mu1 * ssim_v / B1 + dS_ds11 * (-2.f * mu1) + dS_ds12 * (-mu2);
    const size_t pix = (size_t)gy * W + gx;
    const size_t C3 = (size_t)gridDim.z * plane;
    maps[c * plane + pix] = dS_dmu1;
    maps[C3 + c * plane + pix] = dS_ds11;
    maps[2 * C3 + c * plane + pix] = dS_ds12;
    l1_v = fabsf(s_x[ty + RAD][tx + RAD] - s_y[ty + RAD][tx + RAD]);
  }
  const float bs = block_sum_256(ssim_v, s_red);
  __syncthreads();
  const float bl = block_sum_256(l1_v, s_red);
  if (threadIdx.x == 0) {
    const int bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    part_ssim[bid] = bs;
    part_l1[bid] = bl;
  }
}

__global__ void __launch_bounds__(256)
l1_ssim_backward_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                        const float* __restrict__ maps, const float* __restrict__ g_ssim,
                        const float* __restrict__ g_l1, int C, int H, int W, float* __restrict__ dimg1) {
  __shared__ float s_m[3][HS][HS + 1];
  __shared__ float s_h[3][HS][TS + 1];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const size_t plane = (size_t)H * W;
  const size_t C3 = (size_t)C * plane;
  for (int i = threadIdx.x; i < HS * HS; i += 256) {
    const int ly = i / HS, lx = i - ly * HS;
    const int gy = y0 + ly - RAD, gx = x0 + lx - RAD;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t o = c * plane + (size_t)gy * W + gx;
    s_m[0][ly][lx] = in ? maps[o] : 0.f;
    s_m[1][ly][lx] = in ? maps[C3 + o] : 0.f;
    s_m[2][ly][lx] = in ? maps[2 * C3 + o] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HS * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float w = GW[k];
      a += w * s_m[0][ly][lx + k]; b += w * s_m[1][ly][lx + k]; d += w * s_m[2][ly][lx + k];
    }
    s_h[0][ly][lx] = a; s_h[1][ly][lx] = b; s_h[2][ly][lx] = d;
  }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int gx = x0 + tx, gy = y0 + ty;
  if (gx >= W || gy >= H) return;
  float fm = 0.f, f11 = 0.f, f12 = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = GW[k];
    fm += w * s_h[0][ty + k][tx]; f11 += w * s_h[1][ty + k][tx]; f12 += w * s_h[2][ty + k][tx];
  }
  const size_t pix = c * plane + (size_t)gy * W + gx;
  const float x = img1[pix], y = img2[pix];
  const float inv_n = 1.f / (float)C3;
  const float gs = g_ssim ? g_ssim[0] : 0.f, gl = g_l1 ? g_l1[0] : 0.f;
  const float diff = x - y;
  const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
  dimg1[pix] = inv_n * (gs * (fm + 2.f * x * f11 + y * f12) + gl * sgn);
}


// ---- fused face-branch loss block (train_face.py:415-416, 426-456, 508-575) ----------------------------------------
//   gt_white = head & ~mouth ? gt : bg          (head = face | hair; hair -> bg too under hair_mask_iter)
//   loss = L1 + w_dssim (1 - SSIM) + w_alpha (mean((1-alpha) head) + mean(alpha ~head))
//        + w_hair (mean(attn[1][hair]) + mean(attn[0][hair])) + w_lips mean(attn[1, r0:r1, c0:c1]) + w_extra * extra
// One SSIM-tile kernel composes gt_white while staging and, on the channel-0 workgroups, accumulates the alpha
// and attention sums of its tile; a one-workgroup kernel folds the partial sums (fixed order) into the scalars.
struct FaceCfg {
  int H, W, flags;
  float w_dssim, w_alpha, w_hair, w_lips, w_extra;
};
constexpr int F_HAIR_BG = 1, F_ALPHA = 2, F_HAIR_ATTN = 4, F_LIPS = 8;
// F_MOUTH: the mouth branch's compositing (train_mouth.py:186-221) instead of the face branch's --
//   image_green = (lips ^ mouth) ? bg : image,  gt_green = mouth ? gt : bg,  alpha terms over the lips rectangle
// (lips = rows [lips[0], lips[1]) x columns [lips[2], lips[3]) of the image); face / hair masks are not read.
constexpr int F_MOUTH = 16;
// F_PLAIN: whole-frame L1 + DSSIM of image against gt (the fuse stage, train_fuse_con.py:176-181): no mask, background
// or rectangle is read
constexpr int F_PLAIN = 32;

struct FaceIn {
  const float* image; const float* gt; const uint8_t* face; const uint8_t* hair; const uint8_t* mouth;
  const float* bg; const float* alpha; const float* attn; const int32_t* lips; const float* extra;
};

__device__ __forceinline__ bool in_lips(const FaceIn& in, int gy, int gx) {
  return gy >= in.lips[0] && gy < in.lips[1] && gx >= in.lips[2] && gx < in.lips[3];
}

__device__ __forceinline__ void face_pixel(const FaceCfg& cfg, const FaceIn& in, int c, size_t plane, size_t pix,
                                           float& x, float& y) {
  if (cfg.flags & F_PLAIN) {
    x = in.image[c * plane + pix];
    y = in.gt[c * plane + pix];
    return;
  }
  const float bgc = in.bg[c];
  if (cfg.flags & F_MOUTH) {
    const int gy = (int)(pix / (size_t)cfg.W), gx = (int)(pix - (size_t)gy * cfg.W);
    const bool mouth = in.mouth[pix] != 0, lips = in_lips(in, gy, gx);
    x = (lips != mouth) ? bgc : in.image[c * plane + pix];
    y = mouth ? in.gt[c * plane + pix] : bgc;
    return;
  }
  const bool hair = in.hair[pix] != 0, head = hair || in.face[pix] != 0, mouth = in.mouth[pix] != 0;
  const bool hair_bg = (cfg.flags & F_HAIR_BG) && hair;
  x = hair_bg ? bgc : in.image[c * plane + pix];
  y = (head && !mouth && !hair_bg) ? in.gt[c * plane + pix] : bgc;
}

__global__ void __launch_bounds__(256)
face_loss_forward_kernel(FaceCfg cfg, FaceIn in, float* __restrict__ maps, float* __restrict__ part) {
  __shared__ float s_x[HS][HS + 1], s_y[HS][HS + 1];
  __shared__ float s_h[5][HS][TS + 1];
  __shared__ float s_red[4];
  const int H = cfg.H, W = cfg.W;
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const size_t plane = (size_t)H * W;
  for (int i = threadIdx.x; i < HS * HS; i += 256) {
    const int ly = i / HS, lx = i - ly * HS;
    const int gy = y0 + ly - RAD, gx = x0 + lx - RAD;
    float xv = 0.f, yv = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) face_pixel(cfg, in, c, plane, (size_t)gy * W + gx, xv, yv);
    s_x[ly][lx] = xv;
    s_y[ly][lx] = yv;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HS * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float xv = s_x[ly][lx + k], yv = s_y[ly][lx + k], w = GW[k];
      a += w * xv; b += w * yv; aa += w * xv * xv; bb += w * yv * yv; ab += w * xv * yv;
    }
    s_h[0][ly][lx] = a; s_h[1][ly][lx] = b; s_h[2][ly][lx] = aa; s_h[3][ly][lx] = bb; s_h[4][ly][lx] = ab;
  }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int gx = x0 + tx, gy = y0 + ty;
  const bool inside = gx < W && gy < H;
  float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = GW[k];
    mu1 += w * s_h[0][ty + k][tx]; mu2 += w * s_h[1][ty + k][tx];
    e11 += w * s_h[2][ty + k][tx]; e22 += w * s_h[3][ty + k][tx]; e12 += w * s_h[4][ty + k][tx];
  }
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // ssim, l1, a_in, a_out, hair1, hair0, hair_cnt, lips
  if (inside) {
    const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
    const float s11 = e11 - mu1s, s22 = e22 - mu2s, s12 = e12 - mu12;
    const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1s + mu2s + C1, B2 = s11 + s22 + C2;
    const float inv = 1.f / (B1 * B2);
    const float ssim_v = A1 * A2 * inv;
    const float dS_ds11 = -ssim_v / B2;
    const float dS_ds12 = 2.f * A1 * inv;
    const float dS_dmu1 = 2.f * mu2 * A2 * inv - 2.f * mu1 * ssim_v / B1 + dS_ds11 * (-2.f * mu1) + dS_ds12 * (-mu2);
    const size_t pix = (size_t)gy * W + gx;
    const size_t C3 = 3 * plane;
    maps[c * plane + pix] = dS_dmu1;
    maps[C3 + c * plane + pix] = dS_ds11;
    maps[2 * C3 + c * plane + pix] = dS_ds12;
    v[0] = ssim_v;
    v[1] = fabsf(s_x[ty + RAD][tx + RAD] - s_y[ty + RAD][tx + RAD]);
    if (c == 0) {
      const bool mouth_mode = (cfg.flags & F_MOUTH) != 0, plain = (cfg.flags & F_PLAIN) != 0;
      const bool hair = !mouth_mode && !plain && in.hair[pix] != 0;
      const bool head = plain ? false : (mouth_mode ? in_lips(in, gy, gx) : (hair || in.face[pix] != 0));
      if (cfg.flags & F_ALPHA) {
        const float a = in.alpha[pix];
        v[2] = head ? 1.f - a : 0.f;
        v[3] = head ? 0.f : a;
      }
      if ((cfg.flags & F_HAIR_ATTN) && hair) {
        v[4] = in.attn[plane + pix];
        v[5] = in.attn[pix];
        v[6] = 1.f;
      }
      if (cfg.flags & F_LIPS) {
        if (gy >= in.lips[0] && gy < in.lips[1] && gx >= in.lips[2] && gx < in.lips[3]) v[7] = in.attn[plane + pix];
      }
    }
  }
  const int tiles = gridDim.x * gridDim.y;
  const int tile = blockIdx.y * gridDim.x + blockIdx.x;
  const int nk = c == 0 ? 8 : 2;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k >= nk) break;                 // uniform per workgroup
    const float bsum = block_sum_256(v[k], s_red);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (k < 2) part[(size_t)k * 3 * tiles + (size_t)c * tiles + tile] = bsum;
      else part[(size_t)6 * tiles + (size_t)(k - 2) * tiles + tile] = bsum;
    }
  }
}

// out[0] loss, out[1] L1, out[2] SSIM, out[3] 1 / max(#hair pixels, 1), out[4] 1 / lips-rect area (0 when empty)
// one wave per partial-sum array (fixed summation order inside a wave: lane-strided, then a butterfly); called by
// `nwaves` waves of one workgroup (8: the finalize kernel; 4: the backward kernel's first workgroup, which takes the
// arrays in two rounds) -- the same additions in the same order either way, so both produce the same bits
__device__ __forceinline__ void face_loss_partial_sums(const float* __restrict__ part, int tiles,
                                                       const float* __restrict__ extra, int n_extra, float* s_sum,
                                                       int nwaves) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = wave; k < 8; k += nwaves) {
    const float* p = k < 2 ? part + (size_t)k * 3 * tiles : part + (size_t)6 * tiles + (size_t)(k - 2) * tiles;
    const int n = k < 2 ? 3 * tiles : tiles;
    float acc = 0.f;
#pragma unroll 8
    for (int i = lane; i < n; i += 64) acc += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) s_sum[k] = acc;
    if (k == 7) {                       // the shortest array's wave also adds up the `extra` terms
      float e = 0.f;
      for (int i = lane; i < n_extra; i += 64) e += extra[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
      if (lane == 0) s_sum[8] = e;
    }
  }
}

__device__ __forceinline__ float lips_inv_area(const FaceCfg& cfg, const int32_t* __restrict__ lips) {
  const int r0 = max(lips[0], 0), r1 = min(lips[1], cfg.H), c0 = max(lips[2], 0), c1 = min(lips[3], cfg.W);
  const float area = (float)max(r1 - r0, 0) * (float)max(c1 - c0, 0);
  return area > 0.f ? 1.f / area : 0.f;
}

__device__ __forceinline__ void face_loss_scalars(const FaceCfg& cfg, const float* s_sum, const int32_t* __restrict__ lips,
                                                  bool has_extra, float* __restrict__ out) {
  const float npix = (float)cfg.H * (float)cfg.W;
  const float l1 = s_sum[1] / (3.f * npix), ssim = s_sum[0] / (3.f * npix);
  float loss = l1 + cfg.w_dssim * (1.f - ssim);
  if (cfg.flags & F_ALPHA) loss += cfg.w_alpha * (s_sum[2] / npix + s_sum[3] / npix);
  const float inv_cnt = 1.f / fmaxf(s_sum[6], 1.f);
  if (cfg.flags & F_HAIR_ATTN) loss += cfg.w_hair * (s_sum[4] * inv_cnt + s_sum[5] * inv_cnt);
  float inv_area = 0.f;
  if (cfg.flags & F_LIPS) {
    inv_area = lips_inv_area(cfg, lips);
    loss += cfg.w_lips * s_sum[7] * inv_area;
  }
  if (has_extra) loss += cfg.w_extra * s_sum[8];
  out[0] = loss; out[1] = l1; out[2] = ssim; out[3] = inv_cnt; out[4] = inv_area;
}

__global__ void __launch_bounds__(512)
face_loss_finalize_kernel(FaceCfg cfg, const float* __restrict__ part, int tiles, const int32_t* __restrict__ lips,
                          const float* __restrict__ extra, int n_extra, float* __restrict__ out) {
  __shared__ float s_sum[9];
  face_loss_partial_sums(part, tiles, extra, n_extra, s_sum, 8);
  __syncthreads();
  if (threadIdx.x == 0) face_loss_scalars(cfg, s_sum, lips, extra != nullptr, out);
}

// The forward's scalar stage folded into the backward launch (instag_face_loss_*_deferred): the gradients never needed
// the loss VALUE -- only 1 / #hair pixels and 1 / lips area, which every workgroup that uses them derives itself -- so
// the one-workgroup kernel between the two tile kernels was 7 us of the step's critical chain for nothing.
struct FaceFin {
  const float* part; int tiles; const float* extra; int n_extra; float* out;
};

__global__ void __launch_bounds__(256)
face_loss_backward_kernel(FaceCfg cfg, FaceIn in, const float* __restrict__ maps, const float* __restrict__ out,
                          const float* __restrict__ g_loss, const float* __restrict__ g_l1,
                          float* __restrict__ d_image, float* __restrict__ d_alpha, float* __restrict__ d_attn,
                          FaceFin fin) {
  __shared__ float s_m[3][HS][HS + 1];
  __shared__ float s_h[3][HS][TS + 1];
  __shared__ float s_sum[9];
  const int H = cfg.H, W = cfg.W;
  const int c = blockIdx.z;
  // deferred scalar stage (fin.part != null): 1 / #hair pixels for the channel-0 workgroups (the per-tile counts are
  // small integers: any summation order gives the same float), everything for the first workgroup, which also writes
  // the loss scalars the forward left out
  float inv_cnt = 0.f;
  if (fin.part != nullptr) {
    const bool first = blockIdx.x == 0 && blockIdx.y == 0 && c == 0;
    if (first) {
      face_loss_partial_sums(fin.part, fin.tiles, fin.extra, fin.n_extra, s_sum, 4);
      __syncthreads();
      if (threadIdx.x == 0) face_loss_scalars(cfg, s_sum, in.lips, fin.extra != nullptr, fin.out);
      inv_cnt = 1.f / fmaxf(s_sum[6], 1.f);
    } else if (c == 0 && (cfg.flags & F_HAIR_ATTN)) {
      const float* hp = fin.part + (size_t)6 * fin.tiles + (size_t)4 * fin.tiles;
      float acc = 0.f;
      for (int i = threadIdx.x; i < fin.tiles; i += 256) acc += hp[i];
      acc = block_sum_256(acc, s_sum);
      inv_cnt = 1.f / fmaxf(acc, 1.f);
    }
    __syncthreads();
  }
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const size_t plane = (size_t)H * W;
  const size_t C3 = 3 * plane;
  for (int i = threadIdx.x; i < HS * HS; i += 256) {
    const int ly = i / HS, lx = i - ly * HS;
    const int gy = y0 + ly - RAD, gx = x0 + lx - RAD;
    const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t o = c * plane + (size_t)gy * W + gx;
    s_m[0][ly][lx] = inb ? maps[o] : 0.f;
    s_m[1][ly][lx] = inb ? maps[C3 + o] : 0.f;
    s_m[2][ly][lx] = inb ? maps[2 * C3 + o] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HS * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float w = GW[k];
      a += w * s_m[0][ly][lx + k]; b += w * s_m[1][ly][lx + k]; d += w * s_m[2][ly][lx + k];
    }
    s_h[0][ly][lx] = a; s_h[1][ly][lx] = b; s_h[2][ly][lx] = d;
  }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int gx = x0 + tx, gy = y0 + ty;
  if (gx >= W || gy >= H) return;
  float fm = 0.f, f11 = 0.f, f12 = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = GW[k];
    fm += w * s_h[0][ty + k][tx]; f11 += w * s_h[1][ty + k][tx]; f12 += w * s_h[2][ty + k][tx];
  }
  const size_t pix = (size_t)gy * W + gx;
  float x, y;
  face_pixel(cfg, in, c, plane, pix, x, y);
  const float g = g_loss ? g_loss[0] : 0.f;
  const float gl = g + (g_l1 ? g_l1[0] : 0.f);       // the L1 value is also returned on its own
  const float gs = -cfg.w_dssim * g;
  const float diff = x - y;
  const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
  const bool mouth_mode = (cfg.flags & F_MOUTH) != 0, plain = (cfg.flags & F_PLAIN) != 0;
  const bool hair = !mouth_mode && !plain && in.hair[pix] != 0;
  // pixel overwritten by the background: no gradient
  const bool frozen = plain ? false
                            : (mouth_mode ? (in_lips(in, gy, gx) != (in.mouth[pix] != 0)) : ((cfg.flags & F_HAIR_BG) && hair));
  d_image[c * plane + pix] = frozen ? 0.f : (1.f / (float)C3) * (gs * (fm + 2.f * x * f11 + y * f12) + gl * sgn);
  if (c == 0) {
    const bool head = plain ? false : (mouth_mode ? in_lips(in, gy, gx) : (hair || in.face[pix] != 0));
    if (d_alpha) d_alpha[pix] = (cfg.flags & F_ALPHA) ? g * cfg.w_alpha / (float)plane * (head ? -1.f : 1.f) : 0.f;
    if (d_attn) {
      const float gh = ((cfg.flags & F_HAIR_ATTN) && hair) ? g * cfg.w_hair * (fin.part ? inv_cnt : out[3]) : 0.f;
      float gp = 0.f;
      if (cfg.flags & F_LIPS)
        if (gy >= in.lips[0] && gy < in.lips[1] && gx >= in.lips[2] && gx < in.lips[3])
          gp = g * cfg.w_lips * (fin.part ? lips_inv_area(cfg, in.lips) : out[4]);
      d_attn[pix] = gh;
      d_attn[plane + pix] = gh + gp;
      d_attn[2 * plane + pix] = 0.f;
    }
  }
}

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

int instag_l1_ssim_num_partials(int32_t C, int32_t H, int32_t W) {
  return C * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
}

int instag_l1_ssim_forward(const float* img1, const float* img2, int32_t C, int32_t H, int32_t W, float* maps,
                           float* partial_ssim, float* partial_l1, instag_stream_t stream) {
  INSTAG_REQUIRE(img1 && img2 && maps && partial_ssim && partial_l1, "l1_ssim_forward: NULL tensor");
  INSTAG_REQUIRE(C >= 1 && C <= 65535 && H >= 1 && W >= 1, "l1_ssim_forward: bad shape");
  dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, C);
  ProfScope p(K_LOSS_FWD, (hipStream_t)stream);
  l1_ssim_forward_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(img1, img2, H, W, maps, partial_ssim, partial_l1);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_l1_ssim_backward(const float* img1, const float* img2, const float* maps, const float* g_ssim,
                            const float* g_l1, int32_t C, int32_t H, int32_t W, float* dimg1,
                            instag_stream_t stream) {
  INSTAG_REQUIRE(img1 && img2 && maps && dimg1, "l1_ssim_backward: NULL tensor");
  INSTAG_REQUIRE(C >= 1 && C <= 65535 && H >= 1 && W >= 1, "l1_ssim_backward: bad shape");
  dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, C);
  ProfScope p(K_LOSS_BWD, (hipStream_t)stream);
  l1_ssim_backward_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(img1, img2, maps, g_ssim, g_l1, C, H, W, dimg1);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

static int face_cfg(const instag_face_loss_cfg* c, FaceCfg* out) {
  INSTAG_REQUIRE(c, "face_loss: NULL config");
  INSTAG_REQUIRE(c->H >= 1 && c->W >= 1, "face_loss: bad image size");
  *out = FaceCfg{c->H, c->W, c->flags, c->w_dssim, c->w_alpha, c->w_attn_hair, c->w_attn_lips, c->w_extra};
  return INSTAG_OK;
}

int64_t instag_face_loss_num_partials(int32_t H, int32_t W) {
  return (int64_t)12 * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
}

static int face_loss_forward_impl(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                                  const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                                  const float* bg, const float* alpha, const float* attn, const int32_t* lips_rect,
                                  const float* extra, int32_t n_extra, float* maps, float* partials, float* out,
                                  bool finalize, instag_stream_t stream) {
  FaceCfg c;
  if (int rc = face_cfg(cfg, &c)) return rc;
  const bool mouth_mode = (c.flags & F_MOUTH) != 0;
  const bool plain = (c.flags & F_PLAIN) != 0;
  INSTAG_REQUIRE(image && gt && maps && partials && out, "face_loss_forward: NULL tensor");
  INSTAG_REQUIRE(plain || (mouth_mask && bg && (mouth_mode || (face_mask && hair_mask))), "face_loss_forward: NULL mask");
  INSTAG_REQUIRE(!plain || !(c.flags & ~F_PLAIN), "face_loss_forward: the plain mode takes no other term");
  INSTAG_REQUIRE(!mouth_mode || (lips_rect && !(c.flags & (F_HAIR_BG | F_HAIR_ATTN | F_LIPS))),
                 "face_loss_forward: the mouth mode needs lips_rect and none of the face-branch terms");
  INSTAG_REQUIRE(!(c.flags & F_ALPHA) || alpha, "face_loss_forward: alpha term without alpha");
  INSTAG_REQUIRE(!(c.flags & (F_HAIR_ATTN | F_LIPS)) || attn, "face_loss_forward: attention term without attn");
  INSTAG_REQUIRE(!(c.flags & F_LIPS) || lips_rect, "face_loss_forward: lips term without lips_rect");
  INSTAG_REQUIRE(extra == nullptr || n_extra >= 1, "face_loss_forward: n_extra must be >= 1");
  const FaceIn in{image, gt, face_mask, hair_mask, mouth_mask, bg, alpha, attn, lips_rect, extra};
  dim3 grid((c.W + TS - 1) / TS, (c.H + TS - 1) / TS, 3);
  {
    ProfScope p(K_LOSS_FWD, (hipStream_t)stream);
    face_loss_forward_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(c, in, maps, partials);
    INSTAG_CHECK_LAUNCH();
  }
  if (!finalize) return INSTAG_OK;
  face_loss_finalize_kernel<<<1, 512, 0, (hipStream_t)stream>>>(c, partials, (int)(grid.x * grid.y), lips_rect, extra,
                                                              extra ? n_extra : 0, out);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_face_loss_forward(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                             const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                             const float* bg, const float* alpha, const float* attn, const int32_t* lips_rect,
                             const float* extra, int32_t n_extra, float* maps, float* partials, float* out,
                             instag_stream_t stream) {
  return face_loss_forward_impl(cfg, image, gt, face_mask, hair_mask, mouth_mask, bg, alpha, attn, lips_rect, extra,
                                n_extra, maps, partials, out, /*finalize=*/true, stream);
}

int instag_face_loss_forward_deferred(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                                      const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                                      const float* bg, const float* alpha, const float* attn, const int32_t* lips_rect,
                                      const float* extra, int32_t n_extra, float* maps, float* partials,
                                      instag_stream_t stream) {
  INSTAG_REQUIRE(partials != nullptr, "face_loss_forward_deferred: NULL partials");
  float dummy = 0.f;       // (`out` is written by instag_face_loss_backward_deferred; the forward only checks it for NULL)
  return face_loss_forward_impl(cfg, image, gt, face_mask, hair_mask, mouth_mask, bg, alpha, attn, lips_rect, extra,
                                n_extra, maps, partials, &dummy, /*finalize=*/false, stream);
}

static int face_loss_backward_impl(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                                   const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                                   const float* bg, const int32_t* lips_rect, const float* maps, const float* out,
                                   const float* g_loss, const float* g_l1, float* d_image, float* d_alpha, float* d_attn,
                                   const FaceFin& fin, instag_stream_t stream) {
  FaceCfg c;
  if (int rc = face_cfg(cfg, &c)) return rc;
  const bool mouth_mode = (c.flags & F_MOUTH) != 0;
  const bool plain = (c.flags & F_PLAIN) != 0;
  INSTAG_REQUIRE(image && gt && maps && out && d_image, "face_loss_backward: NULL tensor");
  INSTAG_REQUIRE(plain || (mouth_mask && bg && (mouth_mode || (face_mask && hair_mask))), "face_loss_backward: NULL mask");
  INSTAG_REQUIRE(!mouth_mode || lips_rect, "face_loss_backward: the mouth mode needs lips_rect");
  INSTAG_REQUIRE(!(c.flags & F_LIPS) || lips_rect, "face_loss_backward: lips term without lips_rect");
  const FaceIn in{image, gt, face_mask, hair_mask, mouth_mask, bg, nullptr, nullptr, lips_rect, nullptr};
  dim3 grid((c.W + TS - 1) / TS, (c.H + TS - 1) / TS, 3);
  FaceFin f = fin;
  f.tiles = (int)(grid.x * grid.y);
  ProfScope p(K_LOSS_BWD, (hipStream_t)stream);
  face_loss_backward_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(c, in, maps, out, g_loss, g_l1, d_image, d_alpha,
                                                                   d_attn, f);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_face_loss_backward(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                              const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                              const float* bg, const int32_t* lips_rect, const float* maps, const float* out,
                              const float* g_loss, const float* g_l1, float* d_image, float* d_alpha, float* d_attn,
                              instag_stream_t stream) {
  return face_loss_backward_impl(cfg, image, gt, face_mask, hair_mask, mouth_mask, bg, lips_rect, maps, out, g_loss,
                                 g_l1, d_image, d_alpha, d_attn, FaceFin{nullptr, 0, nullptr, 0, nullptr}, stream);
}

int instag_face_loss_backward_deferred(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                                       const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                                       const float* bg, const int32_t* lips_rect, const float* maps,
                                       const float* partials, const float* extra, int32_t n_extra, float* out,
                                       const float* g_loss, const float* g_l1, float* d_image, float* d_alpha,
                                       float* d_attn, instag_stream_t stream) {
  INSTAG_REQUIRE(partials && out, "face_loss_backward_deferred: NULL partials / out");
  INSTAG_REQUIRE(extra == nullptr || n_extra >= 1, "face_loss_backward_deferred: n_extra must be >= 1");
  return face_loss_backward_impl(cfg, image, gt, face_mask, hair_mask, mouth_mask, bg, lips_rect, maps, out, g_loss,
                                 g_l1, d_image, d_alpha, d_attn,
                                 FaceFin{partials, 0, extra, extra ? n_extra : 0, out}, stream);
}

}  // extern "C"
