// Real spherical-harmonics basis encoder (degree 1..8) for gfx950.
//
// Replaces shencoder/src/shencoder.cu of the reference (kernel_sh :28-355, kernel_sh_backward
// :359-382).  The reference transcribes one Cartesian polynomial per basis function and per
// partial derivative (~450 lines of literals).  Here the same polynomials are evaluated from their
// structure
//     Y_l^{+m} = N_lm * Q_lm(z) * Re (x+iy)^m,   Y_l^{-m} = N_lm * Q_lm(z) * Im (x+iy)^m,
//     Q_lm = d^m/dz^m P_l,   N_lm = (-1)^m sqrt(2) K_l^m  (N_l0 = K_l^0),   index = l*l + l + m,
// with  d/dz Q_lm = Q_l,m+1,  d/dx,d/dy of (x+iy)^m = m (x+iy)^(m-1) * (1, i); the table of
// Legendre-derivative coefficients is built at compile time (constexpr) and the kernel is fully
// unrolled per degree, so every coefficient is an immediate.  Partial derivatives are those of the
// polynomials in free (x,y,z), exactly what shencoder.cu:130-350 writes.
#include "common.hpp"

namespace instag {
namespace {

constexpr int SH_MAXD = 8;

struct ShTable {
  float q[SH_MAXD][SH_MAXD + 1][SH_MAXD];  // q[l][m][k]: coefficient of z^k in Q_lm (m = l+1 row is all zero)
  float norm[SH_MAXD][SH_MAXD];            // N_lm
};

constexpr double csqrt(double v) {
  double r = v > 1 ? v : 1.0;
  for (int i = 0; i < 200; ++i) r = 0.5 * (r + v / r);
  return r;
}
constexpr double cfact(int n) { double r = 1; for (int i = 2; i <= n; ++i) r *= i; return r; }

constexpr ShTable make_sh_table() {
  ShTable T{};
  double P[SH_MAXD][SH_MAXD] = {};
  P[0][0] = 1.0;
  if (SH_MAXD > 1) P[1][1] = 1.0;
  for (int n = 1; n + 1 < SH_MAXD; ++n) {  // (n+1) P_{n+1} = (2n+1) z P_n - n P_{n-1}
    for (int k = 0; k < SH_MAXD; ++k) {
      double v = -n * P[n - 1][k];
      if (k > 0) v += (2 * n + 1) * P[n][k - 1];
      P[n + 1][k] = v / (n + 1);
    }
  }
  const double pi = 3.14159265358979323846;
  for (int l = 0; l < SH_MAXD; ++l) {
    double cur[SH_MAXD] = {};
    for (int k = 0; k < SH_MAXD; ++k) cur[k] = P[l][k];
    for (int m = 0; m <= l + 1 && m <= SH_MAXD; ++m) {
      for (int k = 0; k < SH_MAXD; ++k) T.q[l][m][k] = (float)cur[k];
      double nxt[SH_MAXD] = {};
      for (int k = 1; k < SH_MAXD; ++k) nxt[k - 1] = k * cur[k];
      for (int k = 0; k < SH_MAXD; ++k) cur[k] = nxt[k];
    }
    for (int m = 0; m <= l; ++m) {
      const double K = csqrt((2 * l + 1) / (4 * pi) * cfact(l - m) / cfact(l + m));
      T.norm[l][m] = (float)(m == 0 ? K : ((m & 1) ? -1.0 : 1.0) * csqrt(2.0) * K);
    }
  }
  return T;
}

__device__ constexpr ShTable kSh = make_sh_table();

template <int L, int M>
__device__ __forceinline__ float q_eval(float z) {  // Horner, degree L-M, immediates only
  if constexpr (M > L) {
    return 0.f;
  } else {
    float r = kSh.q[L][M][L - M];
#pragma unroll
    for (int k = L - M - 1; k >= 0; --k) r = r * z + kSh.q[L][M][k];
    return r;
  }
}

template <int C, int L, int M>
__device__ __forceinline__ void emit(const float* cm, const float* sm, float z, float* __restrict__ out,
                                     float* __restrict__ dx, float* __restrict__ dy, float* __restrict__ dz) {
  constexpr int C2 = C * C;
  (void)C2;
  const float n = kSh.norm[L][M];
  const float q = n * q_eval<L, M>(z);
  const float qd = n * q_eval<L, M + 1>(z);
  if constexpr (M == 0) {
    out[L * L + L] = q;
    if (dx) { dx[L * L + L] = 0.f; dy[L * L + L] = 0.f; dz[L * L + L] = qd; }
  } else {
    out[L * L + L + M] = q * cm[M];
    out[L * L + L - M] = q * sm[M];
    if (dx) {
      const float qm = q * (float)M;
      dx[L * L + L + M] = qm * cm[M - 1];
      dx[L * L + L - M] = qm * sm[M - 1];
      dy[L * L + L + M] = -qm * sm[M - 1];
      dy[L * L + L - M] = qm * cm[M - 1];
      dz[L * L + L + M] = qd * cm[M];
      dz[L * L + L - M] = qd * sm[M];
    }
  }
  if constexpr (M < L) emit<C, L, M + 1>(cm, sm, z, out, dx, dy, dz);
}

template <int C, int L>
__device__ __forceinline__ void emit_bands(const float* cm, const float* sm, float z, float* out, float* dx,
                                           float* dy, float* dz) {
  emit<C, L, 0>(cm, sm, z, out, dx, dy, dz);
  if constexpr (L + 1 < C) emit_bands<C, L + 1>(cm, sm, z, out, dx, dy, dz);
}

template <int C>
__global__ void __launch_bounds__(256)
sh_forward_kernel(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B,
                  float* __restrict__ dy_dx) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  constexpr int C2 = C * C;
  const float x = inputs[3 * b], y = inputs[3 * b + 1], z = inputs[3 * b + 2];
  float cm[C], sm[C];
  cm[0] = 1.f; sm[0] = 0.f;
#pragma unroll
  for (int m = 1; m < C; ++m) {
    cm[m] = x * cm[m - 1] - y * sm[m - 1];
    sm[m] = x * sm[m - 1] + y * cm[m - 1];
  }
  float out[C2], dx[C2], dy[C2], dz[C2];
  if (dy_dx) emit_bands<C, 0>(cm, sm, z, out, dx, dy, dz);
  else emit_bands<C, 0>(cm, sm, z, out, nullptr, nullptr, nullptr);
  float* o = outputs + (size_t)b * C2;
#pragma unroll
  for (int i = 0; i < C2; ++i) o[i] = out[i];
  if (dy_dx) {
    float* d = dy_dx + (size_t)b * 3 * C2;
#pragma unroll
    for (int i = 0; i < C2; ++i) { d[i] = dx[i]; d[C2 + i] = dy[i]; d[2 * C2 + i] = dz[i]; }
  }
}

__global__ void __launch_bounds__(256)
sh_backward_kernel(const float* __restrict__ grad, uint32_t B, uint32_t C2, const float* __restrict__ dy_dx,
                   float* __restrict__ grad_inputs) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t b = t / 3;
  if (b >= B) return;
  const uint32_t d = t - b * 3;
  const float* g = grad + (size_t)b * C2;
  const float* dd = dy_dx + (size_t)b * 3 * C2 + (size_t)d * C2;
  float r = 0.f;
  for (uint32_t ch = 0; ch < C2; ++ch) r += g[ch] * dd[ch];
  grad_inputs[t] += r;
}

template <int C>
int run_sh(const float* inputs, float* outputs, uint32_t B, float* dy_dx, hipStream_t s) {
  ProfScope p(K_SH_FWD, s);
  sh_forward_kernel<C><<<div_up<uint32_t>(B, 256), 256, 0, s>>>(inputs, outputs, B, dy_dx);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

int instag_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx,
                             instag_stream_t stream) {
  INSTAG_REQUIRE(inputs && outputs, "sh_encode_forward: NULL tensor");
  INSTAG_REQUIRE(D == 3, "SH encoder only support input dim == 3");
  INSTAG_REQUIRE(C >= 1 && C <= 8, "SH encoder only supports degree in [1, 8]");
  if (B == 0) return INSTAG_OK;
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 1: return run_sh<1>(inputs, outputs, B, dy_dx, s);
    case 2: return run_sh<2>(inputs, outputs, B, dy_dx, s);
    case 3: return run_sh<3>(inputs, outputs, B, dy_dx, s);
    case 4: return run_sh<4>(inputs, outputs, B, dy_dx, s);
    case 5: return run_sh<5>(inputs, outputs, B, dy_dx, s);
    case 6: return run_sh<6>(inputs, outputs, B, dy_dx, s);
    case 7: return run_sh<7>(inputs, outputs, B, dy_dx, s);
    default: return run_sh<8>(inputs, outputs, B, dy_dx, s);
  }
}

int instag_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                              const float* dy_dx, float* grad_inputs, instag_stream_t stream) {
  (void)inputs;
  INSTAG_REQUIRE(grad && dy_dx && grad_inputs, "sh_encode_backward: NULL tensor");
  INSTAG_REQUIRE(D == 3, "SH encoder only support input dim == 3");
  INSTAG_REQUIRE(C >= 1 && C <= 8, "SH encoder only supports degree in [1, 8]");
  if (B == 0) return INSTAG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_SH_BWD, s);
  sh_backward_kernel<<<div_up<uint32_t>(B * 3, 256), 256, 0, s>>>(grad, B, C * C, dy_dx, grad_inputs);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // extern "C"
