// Per-Gaussian backward: deterministic sum of the per-instance gradient rows written by
// blend-backward, then the chain rule through conic / EWA covariance / projection / SH / cov3D /
// normal back to the operator's inputs.
//
// Replaces the preprocess backward of the reference's absent `diff_gauss` extension; the
// derivatives are those of oracle/rasterize_ref.py::preprocess (torch autograd), including its two
// documented conventions (a clamped t.x/t.z is a constant; means2D gradient = pixel gradient *
// 0.5*(W,H)).
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

struct BwdIO {
  const float *means3D, *shs, *scales, *rots, *cov3Dp;
  float *dmeans3D, *dmeans2D, *dshs, *dcolors, *dopac, *dscales, *drots, *dcov3D, *dextra;
  const float* shs_rest;     // split SH storage: shs = [N,1,3], shs_rest = [N,M-1,3]; gradients likewise
  float* dshs_rest;
  // rows of the fused auxiliary pass (rgb-only main pass): slots 9..11 = dL/d(aux colour), 12..13 = the aux image's
  // share of dL/d(screen-space mean), which reaches means2D only (the reference renders the aux image from detached
  // geometry, gaussian_renderer/__init__.py:256-268)
  float* daux;
};

// gradient slot of SH coefficient m of Gaussian g (concatenated or split storage)
__device__ __forceinline__ float* dsh_slot(const BwdIO& io, int g, int m, int M) {
  if (io.shs_rest) return m == 0 ? io.dshs + (size_t)g * 3 : io.dshs_rest + ((size_t)g * (M - 1) + (m - 1)) * 3;
  return io.dshs + ((size_t)g * M + m) * 3;
}

__global__ void __launch_bounds__(256)
preprocess_backward_kernel(Camera c, BwdIO io, const float* __restrict__ rec2d,
                           const float* __restrict__ cov3d, const uint32_t* __restrict__ tiles_touched,
                           const uint32_t* __restrict__ flags_in, const int32_t* __restrict__ radii,
                           const float* __restrict__ inst_grad, uint8_t* __restrict__ row_flag, uint32_t capacity) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= c.N) return;
  bool visible = radii[g] > 0;
  if (visible) {   // capacity mode: a Gaussian whose instances were dropped has no gradient rows
    const uint32_t off0 = __float_as_uint(rec2d[(size_t)g * REC_FLOATS + R_OFFSET]);
    if ((uint64_t)off0 + tiles_touched[g] > (uint64_t)capacity) visible = false;
  }

  float gs[14];
#pragma unroll
  for (int k = 0; k < 14; ++k) gs[k] = 0.f;
  if (visible) {
    const uint32_t tt = tiles_touched[g];
    const uint32_t off = __float_as_uint(rec2d[(size_t)g * REC_FLOATS + R_OFFSET]);
    const float4* rows = reinterpret_cast<const float4*>(inst_grad + (size_t)off * REC_FLOATS);
    // only the rows the blend pass wrote exist (a tile's list is walked up to its last contributor: C3 walks 13 % of
    // the entries); their flags are cleared here, so that the set is clean for the next backward pass over this state
    // flags of four rows first, then the rows that exist: their loads are issued together
    for (uint32_t t0 = 0; t0 < tt; t0 += 4) {
      uint8_t f[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) f[k] = (t0 + k < tt) ? row_flag[off + t0 + k] : (uint8_t)0;
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        if (f[k] == 0) continue;
        const uint32_t t = t0 + k;
        row_flag[off + t] = 0;
        const float4 q0 = rows[4 * t + 0], q1 = rows[4 * t + 1], q2 = rows[4 * t + 2], q3 = rows[4 * t + 3];
        gs[0] += q0.x; gs[1] += q0.y; gs[2] += q0.z; gs[3] += q0.w;
        gs[4] += q1.x; gs[5] += q1.y; gs[6] += q1.z; gs[7] += q1.w;
        gs[8] += q2.x; gs[9] += q2.y; gs[10] += q2.z; gs[11] += q2.w;
        gs[12] += q3.x; gs[13] += q3.y;
      }
    }
  }
  const float dL_dx = gs[0], dL_dy = gs[1], gA = gs[2], gB = gs[3], gC = gs[4];
  const float dL_dop = gs[5];
  const float dcol[3] = {gs[6], gs[7], gs[8]};
  const bool aux_rows = io.daux != nullptr;
  const float dL_ddepth = aux_rows ? 0.f : gs[9];
  const float dn[3] = {aux_rows ? 0.f : gs[10], aux_rows ? 0.f : gs[11], aux_rows ? 0.f : gs[12]};
  const float dL_dex = aux_rows ? 0.f : gs[13];
  if (aux_rows) { io.daux[3 * g + 0] = gs[9]; io.daux[3 * g + 1] = gs[10]; io.daux[3 * g + 2] = gs[11]; }
  const float aux_dx = aux_rows ? gs[12] : 0.f, aux_dy = aux_rows ? gs[13] : 0.f;

  float dmean[3] = {0.f, 0.f, 0.f};
  float dscale[3] = {0.f, 0.f, 0.f};
  float drot[4] = {0.f, 0.f, 0.f, 0.f};
  float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nsh = (c.sh_degree + 1) * (c.sh_degree + 1);

  if (visible) {
    const uint32_t flags = flags_in[g];
    const float* __restrict__ V = c.view;
    const float* __restrict__ P = c.proj;
    const float px = io.means3D[3 * g + 0], py = io.means3D[3 * g + 1], pz = io.means3D[3 * g + 2];

    // ---- screen position (NDC) and depth ----------------------------------------------------
    const float hx = P[0] * px + P[4] * py + P[8] * pz + P[12];
    const float hy = P[1] * px + P[5] * py + P[9] * pz + P[13];
    const float hw = P[3] * px + P[7] * py + P[11] * pz + P[15];
    const float m_w = 1.0f / (hw + 1e-7f);
    const float dndcx = 0.5f * (float)c.W * dL_dx;
    const float dndcy = 0.5f * (float)c.H * dL_dy;
    const float mul1 = hx * m_w * m_w, mul2 = hy * m_w * m_w;
    dmean[0] = (P[0] * m_w - P[3] * mul1) * dndcx + (P[1] * m_w - P[3] * mul2) * dndcy + V[2] * dL_ddepth;
    dmean[1] = (P[4] * m_w - P[7] * mul1) * dndcx + (P[5] * m_w - P[7] * mul2) * dndcy + V[6] * dL_ddepth;
    dmean[2] = (P[8] * m_w - P[11] * mul1) * dndcx + (P[9] * m_w - P[11] * mul2) * dndcy + V[10] * dL_ddepth;
    if (io.dmeans2D) {
      io.dmeans2D[3 * g + 0] = dndcx + 0.5f * (float)c.W * aux_dx;
      io.dmeans2D[3 * g + 1] = dndcy + 0.5f * (float)c.H * aux_dy;
      io.dmeans2D[3 * g + 2] = 0.f;
    }

    // ---- conic -> 2D covariance ---------------------------------------------------------------
    const float tx = V[0] * px + V[4] * py + V[8] * pz + V[12];
    const float ty = V[1] * px + V[5] * py + V[9] * pz + V[13];
    const float tz = V[2] * px + V[6] * py + V[10] * pz + V[14];
    const float limx = 1.3f * c.tanfovx, limy = 1.3f * c.tanfovy;
    const float txc = fminf(limx, fmaxf(-limx, tx / tz)) * tz;
    const float tyc = fminf(limy, fmaxf(-limy, ty / tz)) * tz;
    const float okx = (flags & (1u << 6)) ? 0.f : 1.f;
    const float oky = (flags & (1u << 7)) ? 0.f : 1.f;
    const float* S = cov3d + (size_t)g * 6;
    const float S00 = S[0], S01 = S[1], S02 = S[2], S11 = S[3], S12 = S[4], S22 = S[5];
    const float itz = 1.0f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
    const float J00 = c.focal_x * itz, J02 = -(c.focal_x * txc) * itz2;
    const float J11 = c.focal_y * itz, J12 = -(c.focal_y * tyc) * itz2;
    const float T0[3] = {J00 * V[0] + J02 * V[2], J00 * V[4] + J02 * V[6], J00 * V[8] + J02 * V[10]};
    const float T1[3] = {J11 * V[1] + J12 * V[2], J11 * V[5] + J12 * V[6], J11 * V[9] + J12 * V[10]};
    // S * T0^T and S * T1^T
    const float ST0[3] = {S00 * T0[0] + S01 * T0[1] + S02 * T0[2], S01 * T0[0] + S11 * T0[1] + S12 * T0[2],
                          S02 * T0[0] + S12 * T0[1] + S22 * T0[2]};
    const float ST1[3] = {S00 * T1[0] + S01 * T1[1] + S02 * T1[2], S01 * T1[0] + S11 * T1[1] + S12 * T1[2],
                          S02 * T1[0] + S12 * T1[1] + S22 * T1[2]};
    const float a = (T0[0] * ST0[0] + T0[1] * ST0[1] + T0[2] * ST0[2]) + 0.3f;
    const float b = T0[0] * ST1[0] + T0[1] * ST1[1] + T0[2] * ST1[2];
    const float cc = (T1[0] * ST1[0] + T1[1] * ST1[1] + T1[2] * ST1[2]) + 0.3f;
    const float det = a * cc - b * b;
    const float id2 = 1.0f / (det * det);
    const float dL_da = (-cc * cc * gA + b * cc * gB - b * b * gC) * id2;
    const float dL_db = (2.f * b * cc * gA - (a * cc + b * b) * gB + 2.f * a * b * gC) * id2;
    const float dL_dc = (-b * b * gA + a * b * gB - a * a * gC) * id2;

    // ---- 2D covariance -> 3D covariance (6 unique entries) and T -----------------------------------
    dcov[0] = dL_da * T0[0] * T0[0] + dL_db * T0[0] * T1[0] + dL_dc * T1[0] * T1[0];
    dcov[3] = dL_da * T0[1] * T0[1] + dL_db * T0[1] * T1[1] + dL_dc * T1[1] * T1[1];
    dcov[5] = dL_da * T0[2] * T0[2] + dL_db * T0[2] * T1[2] + dL_dc * T1[2] * T1[2];
    dcov[1] = 2.f * dL_da * T0[0] * T0[1] + dL_db * (T0[0] * T1[1] + T0[1] * T1[0]) + 2.f * dL_dc * T1[0] * T1[1];
    dcov[2] = 2.f * dL_da * T0[0] * T0[2] + dL_db * (T0[0] * T1[2] + T0[2] * T1[0]) + 2.f * dL_dc * T1[0] * T1[2];
    dcov[4] = 2.f * dL_da * T0[1] * T0[2] + dL_db * (T0[1] * T1[2] + T0[2] * T1[1]) + 2.f * dL_dc * T1[1] * T1[2];
    float dT0[3], dT1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dT0[k] = 2.f * dL_da * ST0[k] + dL_db * ST1[k];
      dT1[k] = 2.f * dL_dc * ST1[k] + dL_db * ST0[k];
    }
    const float dJ00 = V[0] * dT0[0] + V[4] * dT0[1] + V[8] * dT0[2];
    const float dJ02 = V[2] * dT0[0] + V[6] * dT0[1] + V[10] * dT0[2];
    const float dJ11 = V[1] * dT1[0] + V[5] * dT1[1] + V[9] * dT1[2];
    const float dJ12 = V[2] * dT1[0] + V[6] * dT1[1] + V[10] * dT1[2];
    const float dtx = okx * (-c.focal_x * itz2 * dJ02);
    const float dty = oky * (-c.focal_y * itz2 * dJ12);
    const float dtz = -c.focal_x * itz2 * dJ00 - c.focal_y * itz2 * dJ11 +
                      2.f * c.focal_x * txc * itz3 * dJ02 + 2.f * c.focal_y * tyc * itz3 * dJ12;
    dmean[0] += V[0] * dtx + V[1] * dty + V[2] * dtz;
    dmean[1] += V[4] * dtx + V[5] * dty + V[6] * dtz;
    dmean[2] += V[8] * dtx + V[9] * dty + V[10] * dtz;

    // ---- colour: SH coefficients and view direction ---------------------------------------------
    if (io.shs) {
      float sh[48];
      {
        const float* __restrict__ s0 = io.shs_rest ? io.shs + (size_t)g * 3 : io.shs + (size_t)g * c.M * 3;
        const float* __restrict__ s1 = io.shs_rest ? io.shs_rest + (size_t)g * (c.M - 1) * 3 - 3 : s0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          if (m < nsh) {
            const float* src = m == 0 ? s0 : s1 + 3 * m;
            sh[3 * m] = src[0]; sh[3 * m + 1] = src[1]; sh[3 * m + 2] = src[2];
          }
        }
      }
      const bool want_dsh = io.dshs != nullptr;
      const float ddx = px - c.campos[0], ddy = py - c.campos[1], ddz = pz - c.campos[2];
      const float inv_len = 1.0f / sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
      const float x = ddx * inv_len, y = ddy * inv_len, z = ddz * inv_len;
      float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
      float basis[16];
      basis[0] = SH_C0;
      if (c.sh_degree > 0) {
        basis[1] = -SH_C1 * y; basis[2] = SH_C1 * z; basis[3] = -SH_C1 * x;
        if (c.sh_degree > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          basis[4] = SH_C2[0] * xy; basis[5] = SH_C2[1] * yz; basis[6] = SH_C2[2] * (2.f * zz - xx - yy);
          basis[7] = SH_C2[3] * xz; basis[8] = SH_C2[4] * (xx - yy);
          if (c.sh_degree > 2) {
            basis[9] = SH_C3[0] * y * (3.f * xx - yy); basis[10] = SH_C3[1] * xy * z;
            basis[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
            basis[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
            basis[13] = SH_C3[4] * x * (4.f * zz - xx - yy); basis[14] = SH_C3[5] * z * (xx - yy);
            basis[15] = SH_C3[6] * x * (xx - 3.f * yy);
          }
        }
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float dc = (flags & (1u << ch)) ? 0.f : dcol[ch];
        if (want_dsh) {
          for (int m = 0; m < c.M; ++m) dsh_slot(io, g, m, c.M)[ch] = (m < nsh) ? basis[m] * dc : 0.f;
        }
        if (c.sh_degree > 0) {
          const float s1 = sh[3 + ch], s2 = sh[6 + ch], s3 = sh[9 + ch];
          float gx = -SH_C1 * s3, gy = -SH_C1 * s1, gz = SH_C1 * s2;
          if (c.sh_degree > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float s4 = sh[12 + ch], s5 = sh[15 + ch], s6 = sh[18 + ch], s7 = sh[21 + ch], s8 = sh[24 + ch];
            gx += SH_C2[0] * y * s4 + SH_C2[2] * 2.f * -x * s6 + SH_C2[3] * z * s7 + SH_C2[4] * 2.f * x * s8;
            gy += SH_C2[0] * x * s4 + SH_C2[1] * z * s5 + SH_C2[2] * 2.f * -y * s6 + SH_C2[4] * 2.f * -y * s8;
            gz += SH_C2[1] * y * s5 + SH_C2[2] * 2.f * 2.f * z * s6 + SH_C2[3] * x * s7;
            if (c.sh_degree > 2) {
              const float s9 = sh[27 + ch], s10 = sh[30 + ch], s11 = sh[33 + ch], s12 = sh[36 + ch];
              const float s13 = sh[39 + ch], s14 = sh[42 + ch], s15 = sh[45 + ch];
              gx += SH_C3[0] * s9 * 3.f * 2.f * xy + SH_C3[1] * s10 * yz + SH_C3[2] * s11 * -2.f * xy +
                    SH_C3[3] * s12 * -3.f * 2.f * xz + SH_C3[4] * s13 * (-3.f * xx + 4.f * zz - yy) +
                    SH_C3[5] * s14 * 2.f * xz + SH_C3[6] * s15 * 3.f * (xx - yy);
              gy += SH_C3[0] * s9 * 3.f * (xx - yy) + SH_C3[1] * s10 * xz +
                    SH_C3[2] * s11 * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * s12 * -3.f * 2.f * yz +
                    SH_C3[4] * s13 * -2.f * xy + SH_C3[5] * s14 * -2.f * yz + SH_C3[6] * s15 * -3.f * 2.f * xy;
              gz += SH_C3[1] * s10 * xy + SH_C3[2] * s11 * 4.f * 2.f * yz +
                    SH_C3[3] * s12 * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * s13 * 4.f * 2.f * xz +
                    SH_C3[5] * s14 * (xx - yy);
            }
          }
          dRGBdx[ch] = gx; dRGBdy[ch] = gy; dRGBdz[ch] = gz;
          // accumulate dL/d(dir)
          dRGBdx[ch] *= dc; dRGBdy[ch] *= dc; dRGBdz[ch] *= dc;
        }
      }
      if (c.sh_degree > 0) {
        const float ddirx = dRGBdx[0] + dRGBdx[1] + dRGBdx[2];
        const float ddiry = dRGBdy[0] + dRGBdy[1] + dRGBdy[2];
        const float ddirz = dRGBdz[0] + dRGBdz[1] + dRGBdz[2];
        // through dir = d/|d|
        const float dot = x * ddirx + y * ddiry + z * ddirz;
        dmean[0] += (ddirx - x * dot) * inv_len;
        dmean[1] += (ddiry - y * dot) * inv_len;
        dmean[2] += (ddirz - z * dot) * inv_len;
      }
    } else if (io.dcolors) {
      io.dcolors[3 * g + 0] = dcol[0]; io.dcolors[3 * g + 1] = dcol[1]; io.dcolors[3 * g + 2] = dcol[2];
    }

    // ---- 3D covariance -> scale, rotation; normal -> rotation ---------------------------------------
    if (!io.cov3Dp) {
      const float mod = c.scale_modifier;
      const float sx = mod * io.scales[3 * g + 0], sy = mod * io.scales[3 * g + 1], sz = mod * io.scales[3 * g + 2];
      const float r = io.rots[4 * g + 0], x = io.rots[4 * g + 1], y = io.rots[4 * g + 2], z = io.rots[4 * g + 3];
      float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                       {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                       {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
      const float sv[3] = {sx, sy, sz};
      // dL/dSigma as a full symmetric matrix (off-diagonals carry half of the unique-entry gradient)
      const float Gm[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                              {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                              {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
      float dR[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          // dL/dM = 2 * G * M,  M = R diag(s)
          const float dM = 2.f * (Gm[i][0] * R[0][j] + Gm[i][1] * R[1][j] + Gm[i][2] * R[2][j]) * sv[j];
          dR[i][j] = dM * sv[j];
          dscale[j] += R[i][j] * dM;
        }
      }
      dscale[0] *= mod; dscale[1] *= mod; dscale[2] *= mod;
      // normal = sign * (R[:,k]^T * Vrot)
      const int k = (flags >> 3) & 3;
      const float sgn = (flags & (1u << 5)) ? -1.f : 1.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float v = sgn * (V[4 * i + 0] * dn[0] + V[4 * i + 1] * dn[1] + V[4 * i + 2] * dn[2]);
        if (k == 0) dR[i][0] += v; else if (k == 1) dR[i][1] += v; else dR[i][2] += v;
      }
      drot[0] = -2.f * z * dR[0][1] + 2.f * y * dR[0][2] + 2.f * z * dR[1][0] - 2.f * x * dR[1][2] -
                2.f * y * dR[2][0] + 2.f * x * dR[2][1];
      drot[1] = 2.f * y * dR[0][1] + 2.f * z * dR[0][2] + 2.f * y * dR[1][0] - 4.f * x * dR[1][1] -
                2.f * r * dR[1][2] + 2.f * z * dR[2][0] + 2.f * r * dR[2][1] - 4.f * x * dR[2][2];
      drot[2] = -4.f * y * dR[0][0] + 2.f * x * dR[0][1] + 2.f * r * dR[0][2] + 2.f * x * dR[1][0] +
                2.f * z * dR[1][2] - 2.f * r * dR[2][0] + 2.f * z * dR[2][1] - 4.f * y * dR[2][2];
      drot[3] = -4.f * z * dR[0][0] - 2.f * r * dR[0][1] + 2.f * x * dR[0][2] + 2.f * r * dR[1][0] -
                4.f * z * dR[1][1] + 2.f * y * dR[1][2] + 2.f * x * dR[2][0] + 2.f * y * dR[2][1];
    }
  } else {
    if (io.dmeans2D) { io.dmeans2D[3 * g + 0] = 0.f; io.dmeans2D[3 * g + 1] = 0.f; io.dmeans2D[3 * g + 2] = 0.f; }
    if (io.shs && io.dshs) {
      for (int m = 0; m < c.M; ++m) {
        float* d = dsh_slot(io, g, m, c.M);
        d[0] = 0.f; d[1] = 0.f; d[2] = 0.f;
      }
    }
    if (!io.shs && io.dcolors) { io.dcolors[3 * g + 0] = 0.f; io.dcolors[3 * g + 1] = 0.f; io.dcolors[3 * g + 2] = 0.f; }
  }

  if (io.dmeans3D) { io.dmeans3D[3 * g + 0] = dmean[0]; io.dmeans3D[3 * g + 1] = dmean[1]; io.dmeans3D[3 * g + 2] = dmean[2]; }
  if (io.dopac) io.dopac[g] = dL_dop;
  if (io.dextra) io.dextra[g] = dL_dex;
  if (io.cov3Dp) {
    if (io.dcov3D) {
#pragma unroll
      for (int k = 0; k < 6; ++k) io.dcov3D[6 * g + k] = dcov[k];
    }
  } else {
    if (io.dscales) { io.dscales[3 * g + 0] = dscale[0]; io.dscales[3 * g + 1] = dscale[1]; io.dscales[3 * g + 2] = dscale[2]; }
    if (io.drots) { io.drots[4 * g + 0] = drot[0]; io.drots[4 * g + 1] = drot[1]; io.drots[4 * g + 2] = drot[2]; io.drots[4 * g + 3] = drot[3]; }
  }
}

// Auxiliary colour set (blended over detached geometry): only the colours and the screen-space means receive a
// gradient.  Sums the Gaussian's instance rows written by blend-backward run with the colour override.
__global__ void __launch_bounds__(256)
aux_backward_reduce_kernel(Camera c, const float* __restrict__ rec2d, const uint32_t* __restrict__ tiles_touched,
                           const int32_t* __restrict__ radii, const float* __restrict__ inst_grad,
                           uint8_t* __restrict__ row_flag, uint32_t capacity,
                           float* __restrict__ dL_daux, float* __restrict__ dL_dmeans2D, int accumulate) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= c.N) return;
  float gx = 0.f, gy = 0.f, dc[3] = {0.f, 0.f, 0.f};
  if (radii[g] > 0) {
    const uint32_t tt = tiles_touched[g];
    const uint32_t off = __float_as_uint(rec2d[(size_t)g * REC_FLOATS + R_OFFSET]);
    if ((uint64_t)off + tt <= (uint64_t)capacity) {
      const float4* rows = reinterpret_cast<const float4*>(inst_grad + (size_t)off * REC_FLOATS);
#pragma unroll 4
      for (uint32_t t = 0; t < tt; ++t) {
        if (row_flag[off + t] == 0) continue;
        row_flag[off + t] = 0;
        const float4 q0 = rows[4 * t + 0], q1 = rows[4 * t + 1], q2 = rows[4 * t + 2];
        gx += q0.x; gy += q0.y;
        dc[0] += q1.z; dc[1] += q1.w; dc[2] += q2.x;
      }
    }
  }
  if (dL_daux) { dL_daux[3 * g] = dc[0]; dL_daux[3 * g + 1] = dc[1]; dL_daux[3 * g + 2] = dc[2]; }
  if (dL_dmeans2D) {
    const float ax = 0.5f * (float)c.W * gx, ay = 0.5f * (float)c.H * gy;
    if (accumulate) {
      dL_dmeans2D[3 * g] += ax; dL_dmeans2D[3 * g + 1] += ay;
    } else {
      dL_dmeans2D[3 * g] = ax; dL_dmeans2D[3 * g + 1] = ay; dL_dmeans2D[3 * g + 2] = 0.f;
    }
  }
}

}  // namespace

int launch_preprocess_backward(const Camera& c, const instag_raster_args* a, const float* rec2d,
                               const float* cov3d, const uint32_t* tiles_touched, const uint32_t* flags,
                               const int32_t* radii, const float* inst_grad, uint8_t* row_flag, uint32_t capacity,
                               float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                               float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                               float* dL_dcov3D, float* dL_dextra, float* dL_dshs_rest, float* dL_daux_colors,
                               hipStream_t s) {
  if (c.N == 0) return INSTAG_OK;
  BwdIO io{a->means3D, a->shs, a->scales, a->rotations, a->cov3Ds_precomp,
           dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations,
           dL_dcov3D, dL_dextra, a->shs_rest, dL_dshs_rest, dL_daux_colors};
  ProfScope p(K_PREPROCESS_BWD, s);
  preprocess_backward_kernel<<<div_up(c.N, 256), 256, 0, s>>>(c, io, rec2d, cov3d, tiles_touched, flags,
                                                               radii, inst_grad, row_flag, capacity);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int launch_aux_backward_reduce(const Camera& c, const float* rec2d, const uint32_t* tiles_touched, const int32_t* radii,
                               const float* inst_grad, uint8_t* row_flag, uint32_t capacity, float* dL_daux_colors,
                               float* dL_dmeans2D, bool accumulate_means2D, hipStream_t s) {
  if (c.N == 0) return INSTAG_OK;
  ProfScope p(K_PREPROCESS_BWD, s);
  aux_backward_reduce_kernel<<<div_up(c.N, 256), 256, 0, s>>>(c, rec2d, tiles_touched, radii, inst_grad, row_flag, capacity,
                                                               dL_daux_colors, dL_dmeans2D, accumulate_means2D ? 1 : 0);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // namespace instag
