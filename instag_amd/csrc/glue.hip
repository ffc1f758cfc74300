// Fused per-Gaussian glue of the InsTaG render/train step: the elementwise chains that sit between the
// encoders, the MLPs and the rasterizer, each as ONE forward and ONE backward kernel instead of dozens of
// eager elementwise launches.
//
//  motion_glue      scene/motion_net.py:291-306 (UMF) / :679-692 (PMF):
//                     h_in = cat(enc_x, enc_a * aud_ch_att, enc_e * relu(eye_pre)),
//                     ambient_aud = ||aud_ch_att||, ambient_eye = ||relu(eye_pre)||; amb rows are (aud, eye, 0): the
//                     attention colours of gaussian_renderer/__init__.py:243-246 without a cat
//  deform_activate  gaussian_renderer/__init__.py:200-235 for render_motion(personalized=False, align=True):
//                     means3D = xyz + (h[:, :3]*1e-2) * (tanh(p[:,3:]/5)*0.25+1), scales = softplus(scaling + h[:,8:11]),
//                     rotations = normalize(rotation + h[:,3:7]), opacity = sigmoid(opacity_raw)
//  motion_l1_reg    train_face.py:510-514: sum of the five mean-|.| regularisers of the motion outputs
#include "common.hpp"

namespace instag {
namespace {

constexpr int GB = 256;

// ---- motion glue ----------------------------------------------------------------------------------------------
struct GlueDims { int N, KX, KA, KE; };   // enc_x width, audio width, eye width; h_in width = KX+KA+KE

// Walks the elements i0, i0 + stride, i0 + 2 stride, ... of a row-major [rows x K] matrix as (row, column) pairs
// with one division at the start instead of one (64-bit) division per element.
struct Walk {
  int r, c, qr, qc, K;
  __device__ Walk(size_t i0, size_t stride, int K_) : K(K_) {
    r = (int)(i0 / (size_t)K_); c = (int)(i0 - (size_t)r * K_);
    qr = (int)(stride / (size_t)K_); qc = (int)(stride - (size_t)qr * K_);
  }
  __device__ __forceinline__ void next() {
    r += qr; c += qc;
    if (c >= K) { c -= K; ++r; }
  }
};

__global__ void __launch_bounds__(GB)
motion_glue_forward_kernel(GlueDims d, const float* __restrict__ enc_x, const float* __restrict__ aud,
                           const float* __restrict__ eye_pre, const float* __restrict__ enc_a,
                           const float* __restrict__ enc_e, float* __restrict__ h_in, float* __restrict__ amb) {
  const int K = d.KX + d.KA + d.KE;
  const size_t stride = (size_t)gridDim.x * GB, i0 = (size_t)blockIdx.x * GB + threadIdx.x;
  // element-parallel part: coalesced write of h_in, U rows in flight per thread.  The launcher makes the grid stride
  // a multiple of K: the thread's column (and therefore its source) is fixed, rows advance by w.qr.
  {
    Walk w(i0, stride, K);
    const int c = w.c, dr = w.qr;
    constexpr int U = 4;
    const float* src; int ks; float mul = 1.f; bool relu = false;
    if (c < d.KX) { src = enc_x + c; ks = d.KX; }
    else if (c < d.KX + d.KA) { src = aud + (c - d.KX); ks = d.KA; mul = enc_a[c - d.KX]; }
    else { src = eye_pre + (c - d.KX - d.KA); ks = d.KE; mul = enc_e[c - d.KX - d.KA]; relu = true; }
    for (int r = w.r; r < d.N; r += U * dr) {
      float v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) v[j] = src[(size_t)min(r + j * dr, d.N - 1) * ks];
#pragma unroll
      for (int j = 0; j < U; ++j)
        if (r + j * dr < d.N) h_in[(size_t)(r + j * dr) * K + c] = mul * (relu ? fmaxf(v[j], 0.f) : v[j]);
    }
  }
  // row-parallel part: the two norms
  for (int r = blockIdx.x * GB + threadIdx.x; r < d.N; r += gridDim.x * GB) {
    float sa = 0.f, se = 0.f;
#pragma unroll 8
    for (int k = 0; k < d.KA; ++k) { const float v = aud[(size_t)r * d.KA + k]; sa += v * v; }
    for (int k = 0; k < d.KE; ++k) { const float v = fmaxf(eye_pre[(size_t)r * d.KE + k], 0.f); se += v * v; }
    amb[3 * r] = sqrtf(sa);
    amb[3 * r + 1] = sqrtf(se);
    amb[3 * r + 2] = 0.f;
  }
}

// One pass over the rows of d_h_in (each 128-B line is fetched once): the thread's column decides whether it copies
// (enc_x block), scales by enc_a and adds the norm term (audio block) or does the same through the ReLU (eye block).
// The launcher picks a grid whose stride is a multiple of K = KX+KA+KE, so a thread stays on ONE column: the column
// sums d_enc_a / d_enc_e are register partials, combined per workgroup in a fixed order (LDS float atomics are slow
// on gfx950) and written as one row of per-workgroup partial sums (no global atomics: bitwise reproducible).
__global__ void __launch_bounds__(GB)
motion_glue_backward_kernel(GlueDims d, const float* __restrict__ d_h_in, const float* __restrict__ d_amb,
                            const float* __restrict__ aud, const float* __restrict__ eye_pre,
                            const float* __restrict__ enc_a, const float* __restrict__ enc_e,
                            const float* __restrict__ amb, float* __restrict__ d_enc_x, float* __restrict__ d_aud,
                            float* __restrict__ d_eye_pre, float* __restrict__ col_partials) {
  __shared__ float s_part[GB];
  const int K = d.KX + d.KA + d.KE;
  const size_t stride = (size_t)gridDim.x * GB, i0 = (size_t)blockIdx.x * GB + threadIdx.x;
  Walk w(i0, stride, K);
  const int c = w.c;                       // constant along the walk (stride % K == 0, so rows advance by w.qr)
  const int dr = w.qr;
  constexpr int U = 4;                     // rows in flight per thread (one load per row would be latency-bound)
  float part = 0.f;
  if (c < d.KX) {
    for (int r = w.r; r < d.N; r += U * dr) {
      float v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) v[j] = (r + j * dr < d.N) ? d_h_in[(size_t)(r + j * dr) * K + c] : 0.f;
#pragma unroll
      for (int j = 0; j < U; ++j)
        if (r + j * dr < d.N) d_enc_x[(size_t)(r + j * dr) * d.KX + c] = v[j];
    }
  } else if (c < d.KX + d.KA) {
    const int k = c - d.KX;
    const float ea = enc_a[k];
    for (int r = w.r; r < d.N; r += U * dr) {
      float a[U], gw[U], na[U], da[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int rr = min(r + j * dr, d.N - 1);
        a[j] = aud[(size_t)rr * d.KA + k]; gw[j] = d_h_in[(size_t)rr * K + c];
        na[j] = amb[3 * rr]; da[j] = d_amb ? d_amb[3 * rr] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (r + j * dr < d.N) {
          const float ga = na[j] > 0.f ? da[j] / na[j] : 0.f;
          d_aud[(size_t)(r + j * dr) * d.KA + k] = ea * gw[j] + ga * a[j];
          part += gw[j] * a[j];
        }
      }
    }
  } else {
    const int k = c - d.KX - d.KA;
    const float ee = enc_e[k];
    for (int r = w.r; r < d.N; r += U * dr) {
      float pre[U], gw[U], ne[U], de[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int rr = min(r + j * dr, d.N - 1);
        pre[j] = eye_pre[(size_t)rr * d.KE + k]; gw[j] = d_h_in[(size_t)rr * K + c];
        ne[j] = amb[3 * rr + 1]; de[j] = d_amb ? d_amb[3 * rr + 1] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (r + j * dr < d.N) {
          const float act = fmaxf(pre[j], 0.f);
          const float ge = ne[j] > 0.f ? de[j] / ne[j] : 0.f;
          d_eye_pre[(size_t)(r + j * dr) * d.KE + k] = pre[j] > 0.f ? (ee * gw[j] + ge * act) : 0.f;
          part += gw[j] * act;
        }
      }
    }
  }
  // threads t, t + K, t + 2K, ... of the workgroup share a column
  s_part[threadIdx.x] = part;
  __syncthreads();
  if ((int)threadIdx.x < K && c >= d.KX) {
    float sum = 0.f;
    for (int t = threadIdx.x; t < GB; t += K) sum += s_part[t];
    // per-workgroup partial sums [gridDim.x][KA + KE]; the caller adds them up in a fixed order (deterministic)
    col_partials[(size_t)blockIdx.x * (d.KA + d.KE) + (c - d.KX)] = sum;
  }
}

// ---- deform + activations ---------------------------------------------------------------------------------------
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void __launch_bounds__(GB)
deform_activate_forward_kernel(int N, const float* __restrict__ xyz, const float* __restrict__ scaling,
                               const float* __restrict__ rotation, const float* __restrict__ opacity,
                               const float* __restrict__ h /*[N,11]*/, const float* __restrict__ p /*[N,6]*/,
                               float* __restrict__ means3D, float* __restrict__ scales, float* __restrict__ rots,
                               float* __restrict__ opac, float* __restrict__ reg_partials, float reg_weight) {
  __shared__ float s_red[GB / 64];
  const int r = blockIdx.x * GB + threadIdx.x;
  float reg = 0.f;
  if (r < N) {
    const float* hr = h + (size_t)r * 11;
    const float* pr = p + (size_t)r * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float ps = tanhf(pr[3 + k] * 0.2f) * 0.25f + 1.0f;
      means3D[3 * r + k] = xyz[3 * r + k] + (hr[k] * 1e-2f) * ps;
      scales[3 * r + k] = softplus_f(scaling[3 * r + k] + hr[8 + k]);
    }
    float q[4], n2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { q[k] = rotation[4 * r + k] + hr[3 + k]; n2 += q[k] * q[k]; }
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
    for (int k = 0; k < 4; ++k) rots[4 * r + k] = q[k] * inv;
    opac[r] = sigmoid_f(opacity[r]);
    if (reg_partials) {
      // the motion regulariser of train_face.py:510-514 rides along.  Its d_xyz term sees the displacement AFTER the
      // reference's in-place d_xyz *= p_scale (gaussian_renderer/__init__.py:217 mutates the returned dictionary's
      // entry): |h * 1e-2 * p_scale|, with p_scale = tanh(.) * 0.25 + 1 > 0
      const float w_xyz = 1e-2f / (3.f * N), w_rot = 1.f / (4.f * N), w_opa = 1.f / (float)N, w_sc = 1.f / (3.f * N);
      float ps3[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) ps3[k] = tanhf(pr[3 + k] * 0.2f) * 0.25f + 1.0f;
      reg = w_xyz * (fabsf(hr[0]) * ps3[0] + fabsf(hr[1]) * ps3[1] + fabsf(hr[2]) * ps3[2]) + w_rot * (fabsf(hr[3]) + fabsf(hr[4]) + fabsf(hr[5]) + fabsf(hr[6]))
          + w_opa * fabsf(hr[7]) + w_sc * (fabsf(hr[8]) + fabsf(hr[9]) + fabsf(hr[10]))
          + w_xyz * (fabsf(pr[0]) + fabsf(pr[1]) + fabsf(pr[2]));
    }
  }
  if (reg_partials) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) reg += __shfl_xor(reg, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = reg;
    __syncthreads();
    if (threadIdx.x == 0) reg_partials[blockIdx.x] = reg_weight * (((s_red[0] + s_red[1]) + s_red[2]) + s_red[3]);
  }
}

// ---- mouth branch: displacement gate + activations (gaussian_renderer/__init__.py:404-420, scene/motion_net.py:446-452)
// means3D = xyz + ((h[:, :3] * (sx, sy, sz)) * sigmoid(hs)) * 2, scales = softplus(scaling), rotations = normalize(rotation),
// opacity = sigmoid(opacity): eleven elementwise launches forward and twenty-eight backward otherwise.
__global__ void __launch_bounds__(GB)
mouth_activate_forward_kernel(int N, const float* __restrict__ xyz, const float* __restrict__ scaling,
                              const float* __restrict__ rotation, const float* __restrict__ opacity,
                              const float* __restrict__ h /*[N,7]*/, const float* __restrict__ hs /*[N,1]*/, float sx,
                              float sy, float sz, float* __restrict__ means3D, float* __restrict__ scales,
                              float* __restrict__ rots, float* __restrict__ opac) {
  const int r = blockIdx.x * GB + threadIdx.x;
  if (r >= N) return;
  const float sg = sigmoid_f(hs[r]);
  const float xs[3] = {sx, sy, sz};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    means3D[3 * r + k] = xyz[3 * r + k] + ((h[(size_t)r * 7 + k] * xs[k]) * sg) * 2.0f;
    scales[3 * r + k] = softplus_f(scaling[3 * r + k]);
  }
  float q[4], n2 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { q[k] = rotation[4 * r + k]; n2 += q[k] * q[k]; }
  const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
  for (int k = 0; k < 4; ++k) rots[4 * r + k] = q[k] * inv;
  opac[r] = sigmoid_f(opacity[r]);
}

__global__ void __launch_bounds__(GB)
mouth_activate_backward_kernel(int N, const float* __restrict__ scaling, const float* __restrict__ rotation,
                               const float* __restrict__ opacity, const float* __restrict__ h,
                               const float* __restrict__ hs, float sx, float sy, float sz,
                               const float* __restrict__ g_means, const float* __restrict__ g_scales,
                               const float* __restrict__ g_rots, const float* __restrict__ g_opac,
                               float* __restrict__ d_xyz, float* __restrict__ d_scaling, float* __restrict__ d_rotation,
                               float* __restrict__ d_opacity, float* __restrict__ d_h /*[N,7]*/,
                               float* __restrict__ d_hs /*[N,1]*/) {
  const int r = blockIdx.x * GB + threadIdx.x;
  if (r >= N) return;
  const float sg = sigmoid_f(hs[r]);
  const float xs[3] = {sx, sy, sz};
  float dgate = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float gm = g_means ? g_means[3 * r + k] : 0.f;
    d_xyz[3 * r + k] = gm;
    d_h[(size_t)r * 7 + k] = gm * 2.0f * sg * xs[k];
    dgate += gm * 2.0f * (h[(size_t)r * 7 + k] * xs[k]);
    const float gs = g_scales ? g_scales[3 * r + k] : 0.f;
    d_scaling[3 * r + k] = gs * sigmoid_f(scaling[3 * r + k]);           // d softplus = sigmoid
  }
#pragma unroll
  for (int k = 3; k < 7; ++k) d_h[(size_t)r * 7 + k] = 0.f;            // d_rot is not applied by the mouth render
  d_hs[r] = dgate * sg * (1.f - sg);
  float q[4], n2 = 0.f, dot = 0.f, gr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    q[k] = rotation[4 * r + k];
    n2 += q[k] * q[k];
    gr[k] = g_rots ? g_rots[4 * r + k] : 0.f;
  }
  const float nrm = sqrtf(n2);
  const float inv = 1.0f / fmaxf(nrm, 1e-12f);
#pragma unroll
  for (int k = 0; k < 4; ++k) dot += gr[k] * q[k];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    d_rotation[4 * r + k] = nrm > 1e-12f ? (gr[k] - q[k] * inv * dot * inv) * inv : gr[k] * inv;
  const float so = sigmoid_f(opacity[r]);
  d_opacity[r] = (g_opac ? g_opac[r] : 0.f) * so * (1.f - so);
}

// ---- mean |x[:, :ncols] * scale| as per-workgroup partial sums (train_mouth.py:203 `p_xyz.abs().mean()` on the raw
// alignment head output p [N,6], p_xyz = p[:, :3] * 1e-2); backward writes the whole [N, stride] gradient ------------
constexpr int AM_MAX_PARTIALS = 64;

__global__ void __launch_bounds__(GB)
abs_mean_forward_kernel(int N, int stride, int ncols, float scale, const float* __restrict__ x,
                        float* __restrict__ partials) {
  __shared__ float s_red[GB / 64];
  const float w = 1.0f / ((float)N * (float)ncols);
  float acc = 0.f;
  for (int r = blockIdx.x * GB + threadIdx.x; r < N; r += gridDim.x * GB)
    for (int c = 0; c < ncols; ++c) acc += fabsf(x[(size_t)r * stride + c] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = w * (((s_red[0] + s_red[1]) + s_red[2]) + s_red[3]);
}

__global__ void __launch_bounds__(GB)
abs_mean_backward_kernel(int N, int stride, int ncols, float scale, const float* __restrict__ x,
                         const float* __restrict__ g /*device scalar*/, float* __restrict__ dx) {
  const int i = blockIdx.x * GB + threadIdx.x;
  if (i >= N * stride) return;
  const int c = i % stride;
  const float w = g[0] * scale / ((float)N * (float)ncols);
  dx[i] = c < ncols ? w * sgn(x[i] * scale) : 0.f;
}

// ---- mouth field input assembly (scene/motion_net.py:437-444): in_sigma = [enc_x | enc_a | move], in_scaler =
// [enc_x | move] with the per-frame vectors enc_a [KA], move [KM] repeated over the rows; backward: d_enc_x = the two
// gradients' enc_x columns added, per-workgroup column sums of the enc_a columns (move carries no gradient) ------------
__global__ void __launch_bounds__(GB)
mouth_glue_forward_kernel(int N, int KX, int KA, int KM, const float* __restrict__ enc_x,
                          const float* __restrict__ enc_a, const float* __restrict__ move,
                          float* __restrict__ in_sigma, float* __restrict__ in_scaler) {
  const int K1 = KX + KA + KM, K2 = KX + KM;
  const size_t total = (size_t)N * (K1 + K2);
  for (size_t i = (size_t)blockIdx.x * GB + threadIdx.x; i < total; i += (size_t)gridDim.x * GB) {
    if (i < (size_t)N * K1) {
      const int r = (int)(i / K1), c = (int)(i - (size_t)r * K1);
      in_sigma[i] = c < KX ? enc_x[(size_t)r * KX + c] : (c < KX + KA ? enc_a[c - KX] : move[c - KX - KA]);
    } else {
      const size_t j = i - (size_t)N * K1;
      const int r = (int)(j / K2), c = (int)(j - (size_t)r * K2);
      in_scaler[j] = c < KX ? enc_x[(size_t)r * KX + c] : move[c - KX];
    }
  }
}

constexpr int MG_MAX_PARTIALS = 64;

__global__ void __launch_bounds__(GB)
mouth_glue_backward_kernel(int N, int KX, int KA, int KM, const float* __restrict__ d_sigma /*[N,KX+KA+KM] or null*/,
                           const float* __restrict__ d_scaler /*[N,KX+KM] or null*/, float* __restrict__ d_enc_x,
                           float* __restrict__ col_partials /*[gridDim.x][KA]*/) {
  __shared__ float s_col[GB / 64][32];
  const int K1 = KX + KA + KM, K2 = KX + KM;
  const int rows_per = (N + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per, r1 = min(N, r0 + rows_per);
  // d_enc_x: thread per element of this workgroup's rows
  for (int i = threadIdx.x; i < (r1 - r0) * KX; i += GB) {
    const int r = r0 + i / KX, c = i % KX;
    d_enc_x[(size_t)r * KX + c] = (d_sigma ? d_sigma[(size_t)r * K1 + c] : 0.f) + (d_scaler ? d_scaler[(size_t)r * K2 + c] : 0.f);
  }
  // column sums of the enc_a block: lane = column (KA <= 32), the 8 half-waves of the workgroup take rows in turn
  const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
  float acc = 0.f;
  if (d_sigma && col < KA)
    for (int r = r0 + part; r < r1; r += GB / 32) acc += d_sigma[(size_t)r * K1 + KX + col];
  acc += __shfl_xor(acc, 32);
  if ((threadIdx.x & 63) < 32) s_col[threadIdx.x >> 6][col] = acc;
  __syncthreads();
  if (threadIdx.x < KA)
    col_partials[(size_t)blockIdx.x * KA + threadIdx.x] =
        ((s_col[0][threadIdx.x] + s_col[1][threadIdx.x]) + s_col[2][threadIdx.x]) + s_col[3][threadIdx.x];
}

// ---- fuse stage composition (train_fuse_con.py:102-121): both passes were rendered over bg; bg is taken out again and
// the mouth shows through the face, the scene background through both --------------------------------------------------
//   mouth_image = mouth - bg (1 - a_m) + scene (1 - a_m);   image = face - bg (1 - a_f) + mouth_image (1 - a_f)
__global__ void __launch_bounds__(GB)
fuse_compose_forward_kernel(int HW, const float* __restrict__ face, const float* __restrict__ a_face,
                            const float* __restrict__ mouth, const float* __restrict__ a_mouth,
                            const float* __restrict__ bg /*[3]*/, const float* __restrict__ scene /*[3,HW] or null*/,
                            float* __restrict__ image, float* __restrict__ mouth_image) {
  const int i = blockIdx.x * GB + threadIdx.x;
  if (i >= HW) return;
  const float tf = 1.0f - a_face[i], tm = 1.0f - a_mouth[i];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float sc = scene ? scene[c * HW + i] : 0.f;
    const float mi = (mouth[c * HW + i] - bg[c] * tm) + sc * tm;
    mouth_image[c * HW + i] = mi;
    image[c * HW + i] = (face[c * HW + i] - bg[c] * tf) + mi * tf;
  }
}

__global__ void __launch_bounds__(GB)
fuse_compose_backward_kernel(int HW, const float* __restrict__ g_image /*[3,HW] or null*/,
                             const float* __restrict__ g_mouth_image /*[3,HW] or null*/,
                             const float* __restrict__ a_face, const float* __restrict__ mouth_image,
                             const float* __restrict__ bg, const float* __restrict__ scene,
                             float* __restrict__ d_face, float* __restrict__ d_a_face, float* __restrict__ d_mouth,
                             float* __restrict__ d_a_mouth) {
  const int i = blockIdx.x * GB + threadIdx.x;
  if (i >= HW) return;
  const float tf = 1.0f - a_face[i];
  float da_f = 0.f, da_m = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float g = g_image ? g_image[c * HW + i] : 0.f;
    const float sc = scene ? scene[c * HW + i] : 0.f;
    d_face[c * HW + i] = g;
    da_f += g * (bg[c] - mouth_image[c * HW + i]);           // d/d a_f of -bg (1 - a_f) + mouth_image (1 - a_f)
    const float dmi = g * tf + (g_mouth_image ? g_mouth_image[c * HW + i] : 0.f);
    d_mouth[c * HW + i] = dmi;
    da_m += dmi * (bg[c] - sc);
  }
  d_a_face[i] = da_f;
  d_a_mouth[i] = da_m;
}

__global__ void __launch_bounds__(GB)
deform_activate_backward_kernel(int N, const float* __restrict__ scaling, const float* __restrict__ rotation,
                                const float* __restrict__ opacity, const float* __restrict__ h,
                                const float* __restrict__ p, const float* __restrict__ g_means,
                                const float* __restrict__ g_scales, const float* __restrict__ g_rots,
                                const float* __restrict__ g_opac, float* __restrict__ d_xyz,
                                float* __restrict__ d_scaling, float* __restrict__ d_rotation,
                                float* __restrict__ d_opacity, float* __restrict__ d_h, float* __restrict__ d_p,
                                const float* __restrict__ g_reg, float reg_weight) {
  const int r = blockIdx.x * GB + threadIdx.x;
  if (r >= N) return;
  const float* hr = h + (size_t)r * 11;
  const float* pr = p + (size_t)r * 6;
  float dh[11], dp[6];
#pragma unroll
  for (int k = 0; k < 11; ++k) dh[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) dp[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float gm = g_means ? g_means[3 * r + k] : 0.f;
    const float th = tanhf(pr[3 + k] * 0.2f);
    const float ps = th * 0.25f + 1.0f;
    d_xyz[3 * r + k] = gm;
    dh[k] = gm * ps * 1e-2f;
    dp[3 + k] = gm * (hr[k] * 1e-2f) * 0.25f * (1.f - th * th) * 0.2f;
    const float gs = g_scales ? g_scales[3 * r + k] : 0.f;
    const float sg = gs * sigmoid_f(scaling[3 * r + k] + hr[8 + k]);      // d softplus = sigmoid
    d_scaling[3 * r + k] = sg;
    dh[8 + k] = sg;
  }
  float q[4], n2 = 0.f, dot = 0.f, gr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    q[k] = rotation[4 * r + k] + hr[3 + k];
    n2 += q[k] * q[k];
    gr[k] = g_rots ? g_rots[4 * r + k] : 0.f;
  }
  const float nrm = sqrtf(n2);
  const float inv = 1.0f / fmaxf(nrm, 1e-12f);
#pragma unroll
  for (int k = 0; k < 4; ++k) dot += gr[k] * q[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // y = q / max(|q|, eps):  dq = (g - y <g, y>) / |q|   (for |q| > eps)
    const float dq = nrm > 1e-12f ? (gr[k] - q[k] * inv * dot * inv) * inv : gr[k] * inv;
    d_rotation[4 * r + k] = dq;
    dh[3 + k] = dq;
  }
  const float so = sigmoid_f(opacity[r]);
  d_opacity[r] = (g_opac ? g_opac[r] : 0.f) * so * (1.f - so);
  if (g_reg) {
    const float go = g_reg[0] * reg_weight;
    const float w_xyz = go * 1e-2f / (3.f * N), w_rot = go / (4.f * N), w_opa = go / (float)N, w_sc = go / (3.f * N);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      // d_xyz term: |h_k| * 1e-2 * p_scale_k  (the in-place scaled displacement): gradient to h_k and to p_{3+k}
      const float th = tanhf(pr[3 + k] * 0.2f);
      dh[k] += w_xyz * sgn(hr[k]) * (th * 0.25f + 1.0f);
      dp[3 + k] += w_xyz * fabsf(hr[k]) * 0.25f * (1.f - th * th) * 0.2f;
      dh[8 + k] += w_sc * sgn(hr[8 + k]);
      dp[k] += w_xyz * sgn(pr[k]);
    }
#pragma unroll
    for (int k = 3; k < 7; ++k) dh[k] += w_rot * sgn(hr[k]);
    dh[7] += w_opa * sgn(hr[7]);
  }
#pragma unroll
  for (int k = 0; k < 11; ++k) d_h[(size_t)r * 11 + k] = dh[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) d_p[(size_t)r * 6 + k] = dp[k];
}

// ---- regulariser: mean|h[:, :3]*1e-2| + mean|h[:,3:7]| + mean|h[:,7:8]| + mean|h[:,8:11]| + mean|p[:, :3]*1e-2| ----------
__global__ void __launch_bounds__(GB)
motion_l1_reg_forward_kernel(int N, const float* __restrict__ h, const float* __restrict__ p,
                             float* __restrict__ partial) {
  __shared__ float s_red[4];
  float acc = 0.f;
  const float w_xyz = 1e-2f / (3.f * N), w_rot = 1.f / (4.f * N), w_opa = 1.f / (float)N, w_sc = 1.f / (3.f * N);
  for (int r = blockIdx.x * GB + threadIdx.x; r < N; r += gridDim.x * GB) {
    const float* hr = h + (size_t)r * 11;
    const float* pr = p + (size_t)r * 6;
    acc += w_xyz * (fabsf(hr[0]) + fabsf(hr[1]) + fabsf(hr[2]));
    acc += w_rot * (fabsf(hr[3]) + fabsf(hr[4]) + fabsf(hr[5]) + fabsf(hr[6]));
    acc += w_opa * fabsf(hr[7]);
    acc += w_sc * (fabsf(hr[8]) + fabsf(hr[9]) + fabsf(hr[10]));
    acc += w_xyz * (fabsf(pr[0]) + fabsf(pr[1]) + fabsf(pr[2]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

__global__ void __launch_bounds__(GB)
motion_l1_reg_backward_kernel(int N, const float* __restrict__ h, const float* __restrict__ p,
                              const float* __restrict__ g, float* __restrict__ d_h, float* __restrict__ d_p) {
  const int r = blockIdx.x * GB + threadIdx.x;
  if (r >= N) return;
  const float go = g[0];
  const float w_xyz = go * 1e-2f / (3.f * N), w_rot = go / (4.f * N), w_opa = go / (float)N, w_sc = go / (3.f * N);
  const float* hr = h + (size_t)r * 11;
  const float* pr = p + (size_t)r * 6;
  float* dh = d_h + (size_t)r * 11;
  float* dp = d_p + (size_t)r * 6;
  dh[0] = w_xyz * sgn(hr[0]); dh[1] = w_xyz * sgn(hr[1]); dh[2] = w_xyz * sgn(hr[2]);
  dh[3] = w_rot * sgn(hr[3]); dh[4] = w_rot * sgn(hr[4]); dh[5] = w_rot * sgn(hr[5]); dh[6] = w_rot * sgn(hr[6]);
  dh[7] = w_opa * sgn(hr[7]);
  dh[8] = w_sc * sgn(hr[8]); dh[9] = w_sc * sgn(hr[9]); dh[10] = w_sc * sgn(hr[10]);
  dp[0] = w_xyz * sgn(pr[0]); dp[1] = w_xyz * sgn(pr[1]); dp[2] = w_xyz * sgn(pr[2]);
  dp[3] = 0.f; dp[4] = 0.f; dp[5] = 0.f;
}

// ---- densification statistics (train_face.py:626-629, scene/gaussian_model.py add_densification_stats) ------------
//   max_radii2D[vis] = max(max_radii2D[vis], radii[vis]); xyz_gradient_accum[vis] += ||viewspace_grad[vis, :2]||;
//   denom[vis] += 1          with vis = radii > 0
__global__ void __launch_bounds__(GB)
densify_stats_kernel(int N, float* __restrict__ vs_grad, const float* __restrict__ vs_add /*[N,3] or null*/,
                     const int32_t* __restrict__ radii, float* __restrict__ max_radii2D, float* __restrict__ grad_accum,
                     float* __restrict__ denom) {
  const int i = blockIdx.x * GB + threadIdx.x;
  if (i >= N) return;
  float gx = vs_grad[3 * i], gy = vs_grad[3 * i + 1];
  if (vs_add) {
    // a second producer's share of the screen-space gradient (the auxiliary image's backward) is added HERE and the
    // sum written back, instead of by an elementwise launch on the tail of the step
    gx += vs_add[3 * i]; gy += vs_add[3 * i + 1];
    vs_grad[3 * i] = gx; vs_grad[3 * i + 1] = gy; vs_grad[3 * i + 2] += vs_add[3 * i + 2];
  }
  const int r = radii[i];
  if (r <= 0) return;
  max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
  grad_accum[i] += sqrtf(gx * gx + gy * gy);
  denom[i] += 1.f;
}

inline int row_blocks(int N) { return std::max(1, std::min(2048, (N + GB - 1) / GB)); }

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

int instag_motion_glue_forward(const float* enc_x, const float* aud, const float* eye_pre, const float* enc_a,
                               const float* enc_e, float* h_in, float* amb, int32_t N, int32_t KX, int32_t KA,
                               int32_t KE, instag_stream_t stream) {
  INSTAG_REQUIRE(enc_x && aud && eye_pre && enc_a && enc_e && h_in && amb, "motion_glue_forward: NULL tensor");
  INSTAG_REQUIRE(KA >= 1 && KA <= 32 && KE >= 1 && KE <= 8 && KX >= 1, "motion_glue: widths out of range");
  if (N == 0) return INSTAG_OK;
  const GlueDims d{N, KX, KA, KE};
  // every thread stays on one column of h_in: grid stride a multiple of K
  auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
  const int K = KX + KA + KE;
  const int unit = K / gcd(K, GB);
  const int blocks = std::max(unit, row_blocks(N * 8) / unit * unit);
  motion_glue_forward_kernel<<<blocks, GB, 0, (hipStream_t)stream>>>(d, enc_x, aud, eye_pre, enc_a, enc_e, h_in, amb);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

static int glue_bwd_blocks(int N, int KX, int KA, int KE) {
  // every thread must stay on one column of the [N, K] gradient: grid stride a multiple of K; a workgroup must
  // cover every column (GB >= K) so that each row of the partial sums is fully written
  auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
  const int K = KX + KA + KE;
  const int unit = K / gcd(K, GB);
  return std::max(unit, std::min(1024, row_blocks(N * 8)) / unit * unit);
}

int instag_motion_glue_backward_num_partials(int32_t N, int32_t KX, int32_t KA, int32_t KE) {
  return glue_bwd_blocks(N, KX, KA, KE);
}

int instag_motion_glue_backward(const float* d_h_in, const float* d_amb, const float* aud, const float* eye_pre,
                                const float* enc_a, const float* enc_e, const float* amb, float* d_enc_x,
                                float* d_aud, float* d_eye_pre, float* col_partials, int32_t N,
                                int32_t KX, int32_t KA, int32_t KE, instag_stream_t stream) {
  INSTAG_REQUIRE(d_h_in && aud && eye_pre && enc_a && enc_e && amb && d_enc_x && d_aud && d_eye_pre && col_partials,
                 "motion_glue_backward: NULL tensor");
  INSTAG_REQUIRE(KA >= 1 && KA <= 32 && KE >= 1 && KE <= 8 && KX >= 1 && KX + KA + KE <= GB,
                 "motion_glue: widths out of range");
  if (N == 0) return INSTAG_OK;
  const GlueDims d{N, KX, KA, KE};
  motion_glue_backward_kernel<<<glue_bwd_blocks(N, KX, KA, KE), GB, 0, (hipStream_t)stream>>>(
      d, d_h_in, d_amb, aud, eye_pre, enc_a, enc_e, amb, d_enc_x, d_aud, d_eye_pre, col_partials);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_deform_activate_forward(const float* xyz, const float* scaling, const float* rotation,
                                   const float* opacity, const float* h, const float* p, float* means3D,
                                   float* scales, float* rotations, float* opac, float* reg_partials,
                                   float reg_weight, int32_t N, instag_stream_t stream) {
  INSTAG_REQUIRE(xyz && scaling && rotation && opacity && h && p && means3D && scales && rotations && opac,
                 "deform_activate_forward: NULL tensor");
  if (N == 0) return INSTAG_OK;
  deform_activate_forward_kernel<<<(N + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(
      N, xyz, scaling, rotation, opacity, h, p, means3D, scales, rotations, opac, reg_partials, reg_weight);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_deform_activate_backward(const float* scaling, const float* rotation, const float* opacity, const float* h,
                                    const float* p, const float* g_means, const float* g_scales, const float* g_rots,
                                    const float* g_opac, float* d_xyz, float* d_scaling, float* d_rotation,
                                    float* d_opacity, float* d_h, float* d_p, const float* g_reg, float reg_weight,
                                    int32_t N, instag_stream_t stream) {
  INSTAG_REQUIRE(scaling && rotation && opacity && h && p && d_xyz && d_scaling && d_rotation && d_opacity && d_h && d_p,
                 "deform_activate_backward: NULL tensor");
  if (N == 0) return INSTAG_OK;
  deform_activate_backward_kernel<<<(N + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(
      N, scaling, rotation, opacity, h, p, g_means, g_scales, g_rots, g_opac, d_xyz, d_scaling, d_rotation, d_opacity,
      d_h, d_p, g_reg, reg_weight);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_deform_activate_num_reg_partials(int32_t N) { return (N + GB - 1) / GB; }

int instag_fuse_compose_forward(const float* face, const float* a_face, const float* mouth, const float* a_mouth,
                                const float* bg, const float* scene, float* image, float* mouth_image, int32_t H,
                                int32_t W, instag_stream_t stream) {
  INSTAG_REQUIRE(face && a_face && mouth && a_mouth && bg && image && mouth_image, "fuse_compose_forward: NULL tensor");
  INSTAG_REQUIRE(H >= 1 && W >= 1, "fuse_compose: empty image");
  const int HW = H * W;
  fuse_compose_forward_kernel<<<(HW + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(HW, face, a_face, mouth, a_mouth, bg,
                                                                                 scene, image, mouth_image);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_fuse_compose_backward(const float* g_image, const float* g_mouth_image, const float* a_face,
                                 const float* mouth_image, const float* bg, const float* scene, float* d_face,
                                 float* d_a_face, float* d_mouth, float* d_a_mouth, int32_t H, int32_t W,
                                 instag_stream_t stream) {
  INSTAG_REQUIRE(a_face && mouth_image && bg && d_face && d_a_face && d_mouth && d_a_mouth,
                 "fuse_compose_backward: NULL tensor");
  INSTAG_REQUIRE(H >= 1 && W >= 1, "fuse_compose: empty image");
  const int HW = H * W;
  fuse_compose_backward_kernel<<<(HW + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(
      HW, g_image, g_mouth_image, a_face, mouth_image, bg, scene, d_face, d_a_face, d_mouth, d_a_mouth);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}



int instag_abs_mean_num_partials(int32_t N) { return std::max(1, std::min(AM_MAX_PARTIALS, (N + GB - 1) / GB)); }

int instag_abs_mean_forward(const float* x, int32_t N, int32_t stride, int32_t ncols, float scale, float* partials,
                            instag_stream_t stream) {
  INSTAG_REQUIRE(x && partials, "abs_mean_forward: NULL tensor");
  INSTAG_REQUIRE(N >= 1 && ncols >= 1 && ncols <= stride, "abs_mean: need N >= 1 and 1 <= ncols <= stride");
  abs_mean_forward_kernel<<<instag_abs_mean_num_partials(N), GB, 0, (hipStream_t)stream>>>(N, stride, ncols, scale, x, partials);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_abs_mean_backward(const float* x, const float* g, int32_t N, int32_t stride, int32_t ncols, float scale,
                             float* dx, instag_stream_t stream) {
  INSTAG_REQUIRE(x && g && dx, "abs_mean_backward: NULL tensor");
  INSTAG_REQUIRE(N >= 1 && ncols >= 1 && ncols <= stride, "abs_mean: need N >= 1 and 1 <= ncols <= stride");
  INSTAG_REQUIRE((long long)N * stride <= 0x7fffffffll, "abs_mean: tensor too large");
  abs_mean_backward_kernel<<<(N * stride + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(N, stride, ncols, scale, x, g, dx);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_mouth_glue_forward(const float* enc_x, const float* enc_a, const float* move, float* in_sigma,
                              float* in_scaler, int32_t N, int32_t KX, int32_t KA, int32_t KM, instag_stream_t stream) {
  INSTAG_REQUIRE(enc_x && enc_a && move && in_sigma && in_scaler, "mouth_glue_forward: NULL tensor");
  INSTAG_REQUIRE(KX >= 1 && KA >= 1 && KA <= 32 && KM >= 1, "mouth_glue: need KX >= 1, 1 <= KA <= 32, KM >= 1");
  if (N <= 0) return INSTAG_OK;
  const long long total = (long long)N * (2 * KX + KA + 2 * KM);
  const int blocks = (int)std::max(1ll, std::min(2048ll, (total + GB - 1) / GB));
  mouth_glue_forward_kernel<<<blocks, GB, 0, (hipStream_t)stream>>>(N, KX, KA, KM, enc_x, enc_a, move, in_sigma, in_scaler);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_mouth_glue_backward_num_partials(int32_t N) { return std::max(1, std::min(MG_MAX_PARTIALS, (N + GB - 1) / GB)); }

int instag_mouth_glue_backward(const float* d_sigma, const float* d_scaler, float* d_enc_x, float* col_partials,
                               int32_t N, int32_t KX, int32_t KA, int32_t KM, instag_stream_t stream) {
  INSTAG_REQUIRE(d_enc_x && col_partials, "mouth_glue_backward: NULL tensor");
  INSTAG_REQUIRE(KX >= 1 && KA >= 1 && KA <= 32 && KM >= 1, "mouth_glue: need KX >= 1, 1 <= KA <= 32, KM >= 1");
  INSTAG_REQUIRE(N >= 1, "mouth_glue_backward: N must be >= 1");
  mouth_glue_backward_kernel<<<instag_mouth_glue_backward_num_partials(N), GB, 0, (hipStream_t)stream>>>(
      N, KX, KA, KM, d_sigma, d_scaler, d_enc_x, col_partials);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}



int instag_mouth_activate_forward(const float* xyz, const float* scaling, const float* rotation, const float* opacity,
                                  const float* h, const float* hs, float sx, float sy, float sz, float* means3D,
                                  float* scales, float* rotations, float* opac, int32_t N, instag_stream_t stream) {
  INSTAG_REQUIRE(xyz && scaling && rotation && opacity && h && hs && means3D && scales && rotations && opac,
                 "mouth_activate_forward: NULL tensor");
  if (N <= 0) return INSTAG_OK;
  mouth_activate_forward_kernel<<<(N + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(
      N, xyz, scaling, rotation, opacity, h, hs, sx, sy, sz, means3D, scales, rotations, opac);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_mouth_activate_backward(const float* scaling, const float* rotation, const float* opacity, const float* h,
                                   const float* hs, float sx, float sy, float sz, const float* g_means,
                                   const float* g_scales, const float* g_rots, const float* g_opac, float* d_xyz,
                                   float* d_scaling, float* d_rotation, float* d_opacity, float* d_h, float* d_hs,
                                   int32_t N, instag_stream_t stream) {
  INSTAG_REQUIRE(scaling && rotation && opacity && h && hs && d_xyz && d_scaling && d_rotation && d_opacity && d_h && d_hs,
                 "mouth_activate_backward: NULL tensor");
  if (N <= 0) return INSTAG_OK;
  mouth_activate_backward_kernel<<<(N + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(
      N, scaling, rotation, opacity, h, hs, sx, sy, sz, g_means, g_scales, g_rots, g_opac, d_xyz, d_scaling, d_rotation,
      d_opacity, d_h, d_hs);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}



int instag_motion_l1_reg_num_partials(int32_t N) { return std::max(1, std::min(256, (N + GB - 1) / GB)); }

int instag_motion_l1_reg_forward(const float* h, const float* p, float* partial, int32_t N, instag_stream_t stream) {
  INSTAG_REQUIRE(h && p && partial, "motion_l1_reg_forward: NULL tensor");
  INSTAG_REQUIRE(N >= 1, "motion_l1_reg: N must be >= 1");
  motion_l1_reg_forward_kernel<<<instag_motion_l1_reg_num_partials(N), GB, 0, (hipStream_t)stream>>>(N, h, p, partial);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_motion_l1_reg_backward(const float* h, const float* p, const float* g, float* d_h, float* d_p, int32_t N,
                                  instag_stream_t stream) {
  INSTAG_REQUIRE(h && p && g && d_h && d_p, "motion_l1_reg_backward: NULL tensor");
  INSTAG_REQUIRE(N >= 1, "motion_l1_reg: N must be >= 1");
  motion_l1_reg_backward_kernel<<<(N + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(N, h, p, g, d_h, d_p);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_densify_stats(const float* viewspace_grad, const int32_t* radii, float* max_radii2D, float* grad_accum,
                         float* denom, int32_t N, instag_stream_t stream) {
  INSTAG_REQUIRE(viewspace_grad && radii && max_radii2D && grad_accum && denom, "densify_stats: NULL tensor");
  if (N == 0) return INSTAG_OK;
  densify_stats_kernel<<<(N + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(N, const_cast<float*>(viewspace_grad), nullptr,
                                                                         radii, max_radii2D, grad_accum, denom);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int instag_densify_stats_add(float* viewspace_grad, const float* grad_add, const int32_t* radii, float* max_radii2D,
                             float* grad_accum, float* denom, int32_t N, instag_stream_t stream) {
  INSTAG_REQUIRE(viewspace_grad && radii && max_radii2D && grad_accum && denom, "densify_stats: NULL tensor");
  INSTAG_REQUIRE(grad_add != viewspace_grad, "densify_stats_add: grad_add must not alias viewspace_grad");
  if (N == 0) return INSTAG_OK;
  densify_stats_kernel<<<(N + GB - 1) / GB, GB, 0, (hipStream_t)stream>>>(N, viewspace_grad, grad_add, radii, max_radii2D,
                                                                         grad_accum, denom);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // extern "C"
