// Multiresolution grid ("hashgrid") encoder for gfx950.
//
// Replaces gridencoder/src/gridencoder.cu of the reference:
//   kernel_grid :87-245, kernel_grid_backward :248-340, kernel_input_backward :343-369,
//   kernel_grad_tv :506-610 (same index function :50-84, same level scale exp2f(l*S)*H-1).
//
// MI355X design (not the reference's (B/512, L) launch of per-(point,level) threads):
//  * one thread owns a point for ALL levels: the input is read once, not L times;
//  * consecutive levels whose tables fit the LDS budget are staged into LDS once per workgroup
//    (InsTaG's face planes are 9,464 floats = 37.9 KB for all 12 levels, the mouth planes
//    17 KB per level) and the 2^D corner gathers become LDS reads; levels too large for LDS
//    fall back to global gathers;
//  * backward accumulates the table gradient in an LDS-private copy with LDS float atomics and
//    flushes each non-zero cell with ONE global atomic per workgroup, instead of one contended
//    global atomic per (point, corner) -- 100k points x 4 corners into <=1600 cells otherwise;
//  * the input gradient (kernel_input_backward) is fused into the same pass.
#include <cstdlib>

#include "common.hpp"

namespace instag {
namespace {

constexpr int GRID_BLOCK = 256;
constexpr uint32_t LDS_BUDGET_FLOATS = 16384;  // 64 KB -> 2 workgroups per CU

template <uint32_t D>
__device__ __forceinline__ uint32_t fast_hash(const uint32_t pos_grid[D]) {
  constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
  uint32_t result = 0;
#pragma unroll
  for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * primes[i];
  return result;
}

// index of a grid vertex inside its level, in units of vertices (caller multiplies by C)
template <uint32_t D>
__device__ __forceinline__ uint32_t grid_index(uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                               uint32_t resolution, const uint32_t pos_grid[D]) {
  uint32_t stride = 1, index = 0;
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) {
    if (stride <= hashmap_size) {
      index += pos_grid[d] * stride;
      stride *= align_corners ? resolution : (resolution + 1);
    }
  }
  if (gridtype == 0 && stride > hashmap_size) index = fast_hash<D>(pos_grid);
  return index % hashmap_size;
}

struct LevelGeom {
  float scale;
  uint32_t resolution, hashmap_size;
};
__device__ __forceinline__ LevelGeom level_geom(const int32_t* __restrict__ offsets, uint32_t level, float S, uint32_t H) {
  LevelGeom g;
  g.hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
  g.scale = exp2f(level * S) * H - 1.0f;
  g.resolution = (uint32_t)ceilf(g.scale) + 1;
  return g;
}

template <uint32_t D>
__device__ __forceinline__ bool load_point(const float* __restrict__ inputs, uint32_t b, float x[D]) {
  bool oob = false;
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) {
    x[d] = inputs[b * D + d];
    if (x[d] < 0.f || x[d] > 1.f) oob = true;
  }
  return oob;
}

template <uint32_t D>
__device__ __forceinline__ void locate(const float x[D], float scale, bool align_corners, uint32_t interp,
                                       float pos[D], float pos_deriv[D], uint32_t pos_grid[D]) {
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) {
    pos[d] = x[d] * scale + (align_corners ? 0.0f : 0.5f);
    const float fl = floorf(pos[d]);
    pos_grid[d] = (uint32_t)fl;
    pos[d] -= fl;
    if (interp == 1) {
      pos_deriv[d] = 6.f * pos[d] * (1.0f - pos[d]);
      pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
    } else {
      pos_deriv[d] = 1.0f;
    }
  }
}

// One level of one point.  `tab` points at the level's table (LDS or global).
template <uint32_t D, uint32_t C, typename TabPtr>
__device__ __forceinline__ void encode_level(TabPtr tab, const LevelGeom& lg, const float x[D], bool oob,
                                             uint32_t gridtype, bool align_corners, uint32_t interp,
                                             float* __restrict__ out, float* __restrict__ dy_dx_row) {
  if (oob) {
#pragma unroll
    for (uint32_t ch = 0; ch < C; ++ch) out[ch] = 0.f;
    if (dy_dx_row) {
#pragma unroll
      for (uint32_t i = 0; i < D * C; ++i) dy_dx_row[i] = 0.f;
    }
    return;
  }
  float pos[D], pos_deriv[D];
  uint32_t pos_grid[D];
  locate<D>(x, lg.scale, align_corners, interp, pos, pos_deriv, pos_grid);
  float results[C];
#pragma unroll
  for (uint32_t ch = 0; ch < C; ++ch) results[ch] = 0.f;
#pragma unroll
  for (uint32_t idx = 0; idx < (1u << D); ++idx) {
    float w = 1.f;
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
      if ((idx & (1u << d)) == 0) { w *= 1.f - pos[d]; pg[d] = pos_grid[d]; }
      else { w *= pos[d]; pg[d] = pos_grid[d] + 1; }
    }
    const uint32_t index = grid_index<D>(gridtype, align_corners, lg.hashmap_size, lg.resolution, pg) * C;
#pragma unroll
    for (uint32_t ch = 0; ch < C; ++ch) results[ch] += w * tab[index + ch];
  }
#pragma unroll
  for (uint32_t ch = 0; ch < C; ++ch) out[ch] = results[ch];
  if (dy_dx_row) {
#pragma unroll
    for (uint32_t gd = 0; gd < D; ++gd) {
      float rg[C];
#pragma unroll
      for (uint32_t ch = 0; ch < C; ++ch) rg[ch] = 0.f;
#pragma unroll
      for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
        float w = lg.scale;
        uint32_t pg[D];
#pragma unroll
        for (uint32_t nd = 0; nd < D - 1; ++nd) {
          const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
          if ((idx & (1u << nd)) == 0) { w *= 1.f - pos[d]; pg[d] = pos_grid[d]; }
          else { w *= pos[d]; pg[d] = pos_grid[d] + 1; }
        }
        pg[gd] = pos_grid[gd];
        const uint32_t il = grid_index<D>(gridtype, align_corners, lg.hashmap_size, lg.resolution, pg) * C;
        pg[gd] = pos_grid[gd] + 1;
        const uint32_t ir = grid_index<D>(gridtype, align_corners, lg.hashmap_size, lg.resolution, pg) * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) rg[ch] += w * (tab[ir + ch] - tab[il + ch]) * pos_deriv[gd];
      }
#pragma unroll
      for (uint32_t ch = 0; ch < C; ++ch) dy_dx_row[gd * C + ch] = rg[ch];
    }
  }
}

// Group consecutive levels [level, lend) whose tables fit the LDS budget (0 levels -> global path).
__device__ __forceinline__ uint32_t group_end(const int32_t* __restrict__ offsets, uint32_t level, uint32_t L, uint32_t C) {
  const uint32_t base = (uint32_t)offsets[level];
  uint32_t lend = level;
  while (lend < L && ((uint32_t)offsets[lend + 1] - base) * C <= LDS_BUDGET_FLOATS) ++lend;
  return lend;
}

template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(GRID_BLOCK)
grid_forward_kernel(const float* __restrict__ inputs, const float* __restrict__ grid,
                    const int32_t* __restrict__ offsets, float* __restrict__ outputs, uint32_t B, uint32_t L,
                    float S, uint32_t H, float* __restrict__ dy_dx, uint32_t gridtype, bool align_corners,
                    uint32_t interp) {
  extern __shared__ __align__(16) float s_tab[];
  const uint32_t per_block = (B + gridDim.x - 1) / gridDim.x;
  const uint32_t b0 = blockIdx.x * per_block;
  const uint32_t b1 = min(B, b0 + per_block);
  uint32_t level = 0;
  while (level < L) {
    const uint32_t lend = group_end(offsets, level, L, C);
    if (lend == level) {  // table larger than LDS: gather from global memory
      const LevelGeom lg = level_geom(offsets, level, S, H);
      const float* tab = grid + (size_t)(uint32_t)offsets[level] * C;
      for (uint32_t b = b0 + threadIdx.x; b < b1; b += GRID_BLOCK) {
        float x[D];
        const bool oob = load_point<D>(inputs, b, x);
        float out[C], row[D * C];
        encode_level<D, C>(tab, lg, x, oob, gridtype, align_corners, interp, out, dy_dx ? row : nullptr);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) outputs[(size_t)level * B * C + (size_t)b * C + ch] = out[ch];
        if (dy_dx) {
#pragma unroll
          for (uint32_t i = 0; i < D * C; ++i) dy_dx[(size_t)b * L * D * C + level * D * C + i] = row[i];
        }
      }
      ++level;
      continue;
    }
    const uint32_t base = (uint32_t)offsets[level];
    const uint32_t n = ((uint32_t)offsets[lend] - base) * C;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += GRID_BLOCK) s_tab[i] = grid[(size_t)base * C + i];
    __syncthreads();
    for (uint32_t b = b0 + threadIdx.x; b < b1; b += GRID_BLOCK) {
      float x[D];
      const bool oob = load_point<D>(inputs, b, x);
      for (uint32_t l = level; l < lend; ++l) {
        const LevelGeom lg = level_geom(offsets, l, S, H);
        const float* tab = s_tab + ((uint32_t)offsets[l] - base) * C;
        float out[C], row[D * C];
        encode_level<D, C>(tab, lg, x, oob, gridtype, align_corners, interp, out, dy_dx ? row : nullptr);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) outputs[(size_t)l * B * C + (size_t)b * C + ch] = out[ch];
        if (dy_dx) {
#pragma unroll
          for (uint32_t i = 0; i < D * C; ++i) dy_dx[(size_t)b * L * D * C + l * D * C + i] = row[i];
        }
      }
    }
    level = lend;
  }
}

// LDS: the accumulators are 64-bit fixed-point integers (LDS float atomics run about one lane at a time on gfx950,
// integer ones at full rate; see the tri-plane kernel below) scaled by `to_fixed`; otherwise global float atomics.
template <uint32_t D, uint32_t C, bool LDS>
__device__ __forceinline__ void scatter_level(float* acc, unsigned long long* acc_fixed, double to_fixed,
                                              const LevelGeom& lg, const float x[D], uint32_t gridtype,
                                              bool align_corners, uint32_t interp, const float g[C]) {
  float pos[D], pos_deriv[D];
  uint32_t pos_grid[D];
  locate<D>(x, lg.scale, align_corners, interp, pos, pos_deriv, pos_grid);
#pragma unroll
  for (uint32_t idx = 0; idx < (1u << D); ++idx) {
    float w = 1.f;
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
      if ((idx & (1u << d)) == 0) { w *= 1.f - pos[d]; pg[d] = pos_grid[d]; }
      else { w *= pos[d]; pg[d] = pos_grid[d] + 1; }
    }
    const uint32_t index = grid_index<D>(gridtype, align_corners, lg.hashmap_size, lg.resolution, pg) * C;
#pragma unroll
    for (uint32_t ch = 0; ch < C; ++ch) {
      if (LDS) atomicAdd(&acc_fixed[index + ch], (unsigned long long)__double2ll_rn((double)w * (double)g[ch] * to_fixed));
      else atomicAdd(&acc[index + ch], w * g[ch]);
    }
  }
}

constexpr uint32_t LDS_BWD_ENTRIES = 16384;      // 128 KB of 64-bit accumulators: one workgroup per CU

__device__ __forceinline__ uint32_t group_end_bwd(const int32_t* __restrict__ offsets, uint32_t level, uint32_t L, uint32_t C) {
  const uint32_t base = (uint32_t)offsets[level];
  uint32_t lend = level;
  while (lend < L && ((uint32_t)offsets[lend + 1] - base) * C <= LDS_BWD_ENTRIES) ++lend;
  return lend;
}

template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(GRID_BLOCK)
grid_backward_kernel(const float* __restrict__ grad, const float* __restrict__ inputs,
                     const int32_t* __restrict__ offsets, float* __restrict__ grad_grid, uint32_t B, uint32_t L,
                     float S, uint32_t H, const float* __restrict__ dy_dx, float* __restrict__ grad_inputs,
                     uint32_t gridtype, bool align_corners, uint32_t interp) {
  extern __shared__ __align__(16) unsigned long long s_acc[];
  __shared__ float s_wmax[GRID_BLOCK / 64];
  const uint32_t per_block = (B + gridDim.x - 1) / gridDim.x;
  const uint32_t b0 = blockIdx.x * per_block;
  const uint32_t b1 = min(B, b0 + per_block);

  // fixed-point scale of this workgroup: |sum into one entry| <= (#points) * max|grad| < 2^62
  float gmax = 0.f;
  for (uint32_t l = 0; l < L; ++l)
    for (uint32_t i = b0 * C + threadIdx.x; i < b1 * C; i += GRID_BLOCK) gmax = fmaxf(gmax, fabsf(grad[(size_t)l * B * C + i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
  if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = gmax;
  __syncthreads();
  gmax = s_wmax[0];
#pragma unroll
  for (int w = 1; w < GRID_BLOCK / 64; ++w) gmax = fmaxf(gmax, s_wmax[w]);
  const bool usable = gmax > 0.f && gmax < INFINITY;
  int shift = 0;
  if (usable) {
    int npts_log2 = 1;
    while ((1u << npts_log2) < per_block) ++npts_log2;
    shift = 60 - (ilogbf(gmax) + 1) - npts_log2;
  }
  const double to_fixed = ldexp(1.0, shift), to_float = ldexp(1.0, -shift);

  // fused kernel_input_backward: grad_inputs[b,d] += sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]
  if (dy_dx && grad_inputs) {
    for (uint32_t b = b0 + threadIdx.x; b < b1; b += GRID_BLOCK) {
      float r[D];
#pragma unroll
      for (uint32_t d = 0; d < D; ++d) r[d] = 0.f;
      for (uint32_t l = 0; l < L; ++l) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) {
          const float gv = grad[(size_t)l * B * C + (size_t)b * C + ch];
#pragma unroll
          for (uint32_t d = 0; d < D; ++d) r[d] += gv * dy_dx[(size_t)b * L * D * C + l * D * C + d * C + ch];
        }
      }
#pragma unroll
      for (uint32_t d = 0; d < D; ++d) grad_inputs[(size_t)b * D + d] += r[d];
    }
  }

  uint32_t level = 0;
  while (level < L) {
    const uint32_t lend = usable ? group_end_bwd(offsets, level, L, C) : level;
    if (lend == level) {
      const LevelGeom lg = level_geom(offsets, level, S, H);
      float* acc = grad_grid + (size_t)(uint32_t)offsets[level] * C;
      for (uint32_t b = b0 + threadIdx.x; b < b1; b += GRID_BLOCK) {
        float x[D];
        if (load_point<D>(inputs, b, x)) continue;
        float g[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) g[ch] = grad[(size_t)level * B * C + (size_t)b * C + ch];
        scatter_level<D, C, false>(acc, nullptr, 0.0, lg, x, gridtype, align_corners, interp, g);
      }
      ++level;
      continue;
    }
    const uint32_t base = (uint32_t)offsets[level];
    const uint32_t n = ((uint32_t)offsets[lend] - base) * C;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += GRID_BLOCK) s_acc[i] = 0ull;
    __syncthreads();
    for (uint32_t b = b0 + threadIdx.x; b < b1; b += GRID_BLOCK) {
      float x[D];
      if (load_point<D>(inputs, b, x)) continue;
      for (uint32_t l = level; l < lend; ++l) {
        const LevelGeom lg = level_geom(offsets, l, S, H);
        float g[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) g[ch] = grad[(size_t)l * B * C + (size_t)b * C + ch];
        scatter_level<D, C, true>(nullptr, s_acc + ((uint32_t)offsets[l] - base) * C, to_fixed, lg, x, gridtype,
                                  align_corners, interp, g);
      }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += GRID_BLOCK) {
      const long long a = (long long)s_acc[i];
      if (a != 0) atomicAdd(&grad_grid[(size_t)base * C + i], (float)((double)a * to_float));
    }
    level = lend;
  }
}

// ---- total-variation gradient (gridencoder/grid.py:165-185; semantics of gridencoder.cu:506-610) --------------------
// For every sample and level the reference finds the vertex v = floor(x * scale + 0.5) and adds
//     weight / (2D) * r * rsqrt(q + 1e-9),   r = sum_n (e[v] - e[n]),  q = sum_n (e[v] - e[n])^2  over v's axis neighbours
// into grad[v] with one float atomic per sample and channel.  The term depends on the vertex only, so here:
//   * levels whose vertices map one-to-one onto table entries (no hashing / tiling wrap): pass 1 only COUNTS the samples
//     per entry (integer atomics), pass 2 walks the table, computes the term once per touched entry and adds
//     count * term with a plain store -- per cell, not per sample;
//   * hashed (or wrapped) levels, where several vertices share an entry: the per-sample term is accumulated in 64-bit
//     fixed point (integer atomics: order-independent, so deterministic) and converted in pass 2.
// No float atomics; bitwise reproducible.
template <uint32_t D>
__device__ __forceinline__ bool level_is_dense(const LevelGeom& lg, bool align_corners) {
  uint64_t n = 1;
  const uint64_t side = align_corners ? lg.resolution : lg.resolution + 1;
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) { n *= side; if (n > lg.hashmap_size) return false; }
  return true;
}

// term of vertex `pos_grid` for every channel
template <uint32_t D, uint32_t C>
__device__ __forceinline__ void tv_term(const float* __restrict__ tab, const LevelGeom& lg, uint32_t gridtype,
                                        bool align_corners, uint32_t (&pos_grid)[D], uint32_t index, float w,
                                        float (&term)[C]) {
  float r[C], q[C];
#pragma unroll
  for (uint32_t ch = 0; ch < C; ++ch) { r[ch] = 0.f; q[ch] = 0.f; }
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) {
    const uint32_t cur = pos_grid[d];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      if (side == 0 ? cur < lg.resolution : cur > 0) {
        pos_grid[d] = side == 0 ? cur + 1 : cur - 1;
        const uint32_t in = grid_index<D>(gridtype, align_corners, lg.hashmap_size, lg.resolution, pos_grid) * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) {
          const float gv = tab[index + ch] - tab[in + ch];
          r[ch] += gv; q[ch] += gv * gv;
        }
      }
    }
    pos_grid[d] = cur;
  }
#pragma unroll
  for (uint32_t ch = 0; ch < C; ++ch) term[ch] = w * r[ch] * rsqrtf(q[ch] + 1e-9f);
}

template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(GRID_BLOCK)
grid_tv_count_kernel(const float* __restrict__ inputs, const float* __restrict__ grid,
                     const int32_t* __restrict__ offsets, float weight, double to_fixed, uint32_t B, float S, uint32_t H,
                     uint32_t gridtype, bool align_corners, uint32_t* __restrict__ counts,
                     unsigned long long* __restrict__ acc) {
  const uint32_t b = blockIdx.x * GRID_BLOCK + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  float x[D];
  if (load_point<D>(inputs, b, x)) return;
  const LevelGeom lg = level_geom(offsets, level, S, H);
  const uint32_t off = (uint32_t)offsets[level];
  uint32_t pos_grid[D];
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) pos_grid[d] = (uint32_t)floorf(x[d] * lg.scale + (align_corners ? 0.0f : 0.5f));
  const uint32_t idx = grid_index<D>(gridtype, align_corners, lg.hashmap_size, lg.resolution, pos_grid);
  if (level_is_dense<D>(lg, align_corners)) {
    atomicAdd(&counts[off + idx], 1u);
  } else {
    float term[C];
    tv_term<D, C>(grid + (size_t)off * C, lg, gridtype, align_corners, pos_grid, idx * C, weight / (2 * D), term);
#pragma unroll
    for (uint32_t ch = 0; ch < C; ++ch)
      atomicAdd(&acc[((size_t)off + idx) * C + ch], (unsigned long long)__double2ll_rn((double)term[ch] * to_fixed));
  }
}

template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(GRID_BLOCK)
grid_tv_apply_kernel(const float* __restrict__ grid, float* __restrict__ grad, const int32_t* __restrict__ offsets,
                     float weight, double to_float, float S, uint32_t H, uint32_t gridtype, bool align_corners,
                     const uint32_t* __restrict__ counts, const unsigned long long* __restrict__ acc) {
  const uint32_t level = blockIdx.y;
  const LevelGeom lg = level_geom(offsets, level, S, H);
  const uint32_t off = (uint32_t)offsets[level];
  const bool dense = level_is_dense<D>(lg, align_corners);
  const uint32_t side = align_corners ? lg.resolution : lg.resolution + 1;
  for (uint32_t e = blockIdx.x * GRID_BLOCK + threadIdx.x; e < lg.hashmap_size; e += gridDim.x * GRID_BLOCK) {
    float total[C];
    bool any = false;
    if (dense) {
      const uint32_t cnt = counts[off + e];
      if (cnt != 0) {
        uint32_t pos_grid[D], rem = e;                 // entry -> vertex: index = sum_d pos[d] * side^d
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) { pos_grid[d] = rem % side; rem /= side; }
        float term[C];
        tv_term<D, C>(grid + (size_t)off * C, lg, gridtype, align_corners, pos_grid, e * C, weight / (2 * D), term);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) total[ch] = (float)cnt * term[ch];
        any = true;
      }
    } else {
#pragma unroll
      for (uint32_t ch = 0; ch < C; ++ch) {
        const long long a = (long long)acc[((size_t)off + e) * C + ch];
        total[ch] = (float)((double)a * to_float);
        any = any || a != 0;
      }
    }
    if (any) {
#pragma unroll
      for (uint32_t ch = 0; ch < C; ++ch) grad[((size_t)off + e) * C + ch] += total[ch];
    }
  }
}

inline unsigned fwd_blocks(uint32_t B) { return (unsigned)std::min<uint32_t>(div_up<uint32_t>(B, GRID_BLOCK), 2048u); }
inline unsigned bwd_blocks(uint32_t B) {
  constexpr unsigned cap = 256u;      // one workgroup per CU: the LDS-private table copy is flushed once per workgroup
  return (unsigned)std::min<uint32_t>(div_up<uint32_t>(B, GRID_BLOCK), cap);
}

template <uint32_t D, uint32_t C>
int run_forward(const float* inputs, const float* emb, const int32_t* offsets, float* outputs, uint32_t B,
                uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype, bool align, uint32_t interp,
                hipStream_t s) {
  ProfScope p(K_GRID_FWD, s);
  grid_forward_kernel<D, C><<<fwd_blocks(B), GRID_BLOCK, LDS_BUDGET_FLOATS * sizeof(float), s>>>(
      inputs, emb, offsets, outputs, B, L, S, H, dy_dx, gridtype, align, interp);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}
template <uint32_t D, uint32_t C>
int run_backward(const float* grad, const float* inputs, const int32_t* offsets, float* grad_emb, uint32_t B,
                 uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs, uint32_t gridtype,
                 bool align, uint32_t interp, hipStream_t s) {
  ProfScope p(K_GRID_BWD, s);
  if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(grid_backward_kernel<D, C>),
                                   (int)(LDS_BWD_ENTRIES * sizeof(unsigned long long)))) return rc;
  grid_backward_kernel<D, C><<<bwd_blocks(B), GRID_BLOCK, LDS_BWD_ENTRIES * sizeof(unsigned long long), s>>>(
      grad, inputs, offsets, grad_emb, B, L, S, H, dy_dx, grad_inputs, gridtype, align, interp);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}
template <uint32_t D, uint32_t C>
int run_tv(const float* inputs, const float* emb, float* grad, const int32_t* offsets, float weight, uint32_t B,
           uint32_t L, float S, uint32_t H, uint32_t gridtype, bool align, uint32_t* counts, unsigned long long* acc,
           hipStream_t s) {
  // |term| < |weight|: 2^36 fixed-point units per |weight| leave room for 2^26 samples in one entry
  const double scale = weight != 0.f ? 68719476736.0 / fabs((double)weight) : 1.0;
  dim3 g(div_up<uint32_t>(B, GRID_BLOCK), L, 1);
  grid_tv_count_kernel<D, C><<<g, GRID_BLOCK, 0, s>>>(inputs, emb, offsets, weight, scale, B, S, H, gridtype, align,
                                                      counts, acc);
  INSTAG_CHECK_LAUNCH();
  dim3 g2(256, L, 1);
  grid_tv_apply_kernel<D, C><<<g2, GRID_BLOCK, 0, s>>>(emb, grad, offsets, weight, 1.0 / scale, S, H, gridtype, align,
                                                       counts, acc);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

#define DISPATCH_DC(D, C, FN, ...)                                                            \
  switch (D) {                                                                                \
    case 2: switch (C) { case 1: return FN<2, 1>(__VA_ARGS__); case 2: return FN<2, 2>(__VA_ARGS__); \
                         case 4: return FN<2, 4>(__VA_ARGS__); case 8: return FN<2, 8>(__VA_ARGS__); } break; \
    case 3: switch (C) { case 1: return FN<3, 1>(__VA_ARGS__); case 2: return FN<3, 2>(__VA_ARGS__); \
                         case 4: return FN<3, 4>(__VA_ARGS__); case 8: return FN<3, 8>(__VA_ARGS__); } break; \
    case 4: switch (C) { case 1: return FN<4, 1>(__VA_ARGS__); case 2: return FN<4, 2>(__VA_ARGS__); \
                         case 4: return FN<4, 4>(__VA_ARGS__); case 8: return FN<4, 8>(__VA_ARGS__); } break; \
    case 5: switch (C) { case 1: return FN<5, 1>(__VA_ARGS__); case 2: return FN<5, 2>(__VA_ARGS__); \
                         case 4: return FN<5, 4>(__VA_ARGS__); case 8: return FN<5, 8>(__VA_ARGS__); } break; \
  }

}  // namespace
}  // namespace instag

using namespace instag;

extern "C" {

int instag_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx,
                               uint32_t gridtype, int align_corners, uint32_t interp, instag_stream_t stream) {
  INSTAG_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward: NULL tensor");
  INSTAG_REQUIRE(L >= 1 && L <= 64, "GridEncoding: L must be in [1,64]");
  if (B == 0) return INSTAG_OK;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DC(D, C, run_forward, inputs, embeddings, offsets, outputs, B, L, S, H, dy_dx, gridtype,
              align_corners != 0, interp, s);
  set_error("GridEncoding: D must be 2..5 and C must be 1, 2, 4, or 8.");
  return INSTAG_E_ARG;
}

int instag_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs,
                                uint32_t gridtype, int align_corners, uint32_t interp, instag_stream_t stream) {
  (void)embeddings;                     // (part of the reference's signature; the table gradient does not need the table)
  INSTAG_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward: NULL tensor");
  INSTAG_REQUIRE(L >= 1 && L <= 64, "GridEncoding: L must be in [1,64]");
  if (B == 0) return INSTAG_OK;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DC(D, C, run_backward, grad, inputs, offsets, grad_embeddings, B, L, S, H, dy_dx, grad_inputs,
              gridtype, align_corners != 0, interp, s);
  set_error("GridEncoding: D must be 2..5 and C must be 1, 2, 4, or 8.");
  return INSTAG_E_ARG;
}

size_t instag_grid_total_variation_workspace_bytes(uint32_t total_params, uint32_t C) {
  return align_up((size_t)total_params * sizeof(uint32_t), 256) + (size_t)total_params * C * sizeof(unsigned long long);
}

int instag_grid_total_variation(const float* inputs, const float* embeddings, float* grad, const int32_t* offsets,
                                float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                uint32_t gridtype, int align_corners, uint32_t total_params, void* workspace,
                                size_t workspace_bytes, instag_stream_t stream) {
  INSTAG_REQUIRE(inputs && embeddings && grad && offsets, "grad_total_variation: NULL tensor");
  INSTAG_REQUIRE(L >= 1 && L <= 64, "GridEncoding: L must be in [1,64]");
  const size_t need = instag_grid_total_variation_workspace_bytes(total_params, C);
  if (workspace == nullptr || workspace_bytes < need) { set_error("grad_total_variation: workspace too small"); return INSTAG_E_SPACE; }
  if (B == 0) return INSTAG_OK;
  hipStream_t s = (hipStream_t)stream;
  INSTAG_CHECK_HIP(hipMemsetAsync(workspace, 0, need, s));
  uint32_t* counts = (uint32_t*)workspace;
  unsigned long long* acc = (unsigned long long*)((char*)workspace + align_up((size_t)total_params * sizeof(uint32_t), 256));
  DISPATCH_DC(D, C, run_tv, inputs, embeddings, grad, offsets, weight, B, L, S, H, gridtype, align_corners != 0, counts,
              acc, s);
  set_error("GridEncoding: D must be 2..5 and C must be 1, 2, 4, or 8.");
  return INSTAG_E_ARG;
}

}  // extern "C"

// =================================================================================================
// Tri-plane encoder: the three 2-D GridEncoders of a motion field (xy, yz, xz planes; identical
// configuration: D=2, C=1, tables that fit LDS) evaluated in ONE pass over the points.
//
// Replaces, for one network, scene/motion_net.py:244-258 (split_xyz + three GridEncoder.forward calls +
// torch.cat) including the input mapping (x+bound)/(2 bound) of gridencoder/grid.py:149 and the
// [L,B,C]->[B,L*C] permute of :57: xyz [N,3] goes in, the concatenated feature row [N, 3L] comes out.
// Backward recomputes the interpolation from the LDS-resident table instead of storing dy_dx
// (96 B/point/plane), accumulates the table gradient in an LDS-private copy and returns d/dxyz.
// =================================================================================================
namespace instag {
namespace {

constexpr uint32_t TP_MAX_L = 16;
// the LDS kernels keep a plane's table (4 B / entry) and its gradient (8 B / entry) on chip; larger tables are read
// and accumulated in place (triplane_global_*)
inline bool tp_fits_lds(uint32_t total_params) { return (size_t)12 * total_params <= 156 * 1024; }

struct TriPlaneArgs {
  const float* xyz;           // [N,3]
  const float* tables[3];     // each [T,1]
  const int32_t* offsets;     // [L+1], shared by the three planes
  uint32_t N, L, H;
  float S, bound;
  const float* shift;         // optional [N, shift_stride]: the point is xyz + shift_scale * shift[:, :3]
  uint32_t shift_stride;      //   (the universal field is evaluated at xyz + p_xyz, gaussian_renderer/__init__.py:196-197)
  float shift_scale;
};

__device__ __forceinline__ void tp_point(const TriPlaneArgs& a, uint32_t b, float p[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = a.xyz[3 * b + k];
  if (a.shift) {
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] += a.shift_scale * a.shift[(size_t)b * a.shift_stride + k];
  }
}

__device__ __forceinline__ void plane_coords(int plane, const float p[3], float x[2]) {
  // xy = (x,y), yz = (y,z), xz = (x,z)   (motion_net.py:244-247)
  x[0] = plane == 1 ? p[1] : p[0];
  x[1] = plane == 0 ? p[1] : p[2];
}

// per-level constants, computed once per workgroup (every level of a tri-plane table is dense: index = x + y*(res+1))
struct TpLevel { float scale; uint32_t stride, offset; };

__device__ __forceinline__ void tp_levels(const TriPlaneArgs& a, TpLevel* s_lv) {
  if (threadIdx.x < a.L) {
    const LevelGeom lg = level_geom(a.offsets, threadIdx.x, a.S, a.H);
    s_lv[threadIdx.x] = TpLevel{lg.scale, lg.resolution + 1, (uint32_t)a.offsets[threadIdx.x]};
  }
}

// table -> LDS with 16-byte loads, several in flight per thread (a scalar loop with a run-time trip count waits
// for every load in turn: ~1 us per iteration from L2)
template <int THREADS, bool ZERO_ACC>
__device__ __forceinline__ void tp_stage_table(float* __restrict__ s_tab, float* __restrict__ s_acc,
                                               const float* __restrict__ tab, uint32_t T) {
  if ((T & 3u) == 0 && (reinterpret_cast<uintptr_t>(tab) & 15) == 0) {
    const float4* src = reinterpret_cast<const float4*>(tab);
    float4* dst = reinterpret_cast<float4*>(s_tab);
    float4* acc = reinterpret_cast<float4*>(s_acc);
    const uint32_t n4 = T >> 2;
#pragma unroll 8
    for (uint32_t i = threadIdx.x; i < n4; i += THREADS) {
      dst[i] = src[i];
      if (ZERO_ACC) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
#pragma unroll 8
    for (uint32_t i = threadIdx.x; i < T; i += THREADS) {
      s_tab[i] = tab[i];
      if (ZERO_ACC) s_acc[i] = 0.f;
    }
  }
}

__global__ void __launch_bounds__(GRID_BLOCK)
triplane_forward_kernel(TriPlaneArgs a, float* __restrict__ out /*[N,3L]*/) {
  extern __shared__ __align__(16) float s_tab[];
  __shared__ TpLevel s_lv[TP_MAX_L];
  const uint32_t per_block = (a.N + gridDim.x - 1) / gridDim.x;
  const uint32_t b0 = blockIdx.x * per_block, b1 = min(a.N, b0 + per_block);
  const uint32_t T = (uint32_t)a.offsets[a.L];
  const float inv2b = 1.0f / (2.0f * a.bound);
  tp_levels(a, s_lv);
  for (int plane = 0; plane < 3; ++plane) {
    __syncthreads();
    tp_stage_table<GRID_BLOCK, false>(s_tab, nullptr, a.tables[plane], T);
    __syncthreads();
    for (uint32_t b = b0 + threadIdx.x; b < b1; b += GRID_BLOCK) {
      float p[3];
      tp_point(a, b, p);
      float xw[2];
      plane_coords(plane, p, xw);
      const float x0 = (xw[0] + a.bound) * inv2b, x1 = (xw[1] + a.bound) * inv2b;
      const bool oob = x0 < 0.f || x0 > 1.f || x1 < 0.f || x1 > 1.f;
      float* o = out + (size_t)b * 3 * a.L + plane * a.L;
#pragma unroll 4
      for (uint32_t l = 0; l < a.L; ++l) {
        float v = 0.f;
        if (!oob) {
          const TpLevel lv = s_lv[l];
          const float px = x0 * lv.scale + 0.5f, py = x1 * lv.scale + 0.5f;
          const float flx = floorf(px), fly = floorf(py);
          const float fx = px - flx, fy = py - fly;
          const float* tab = s_tab + lv.offset + (uint32_t)flx + (uint32_t)fly * lv.stride;
          const float v00 = tab[0], v10 = tab[1], v01 = tab[lv.stride], v11 = tab[lv.stride + 1];
          v = ((1.f - fx) * (1.f - fy)) * v00 + (fx * (1.f - fy)) * v10 + ((1.f - fx) * fy) * v01 + (fx * fy) * v11;
        }
        o[l] = v;
      }
    }
  }
}

// one (point, plane) work item of the forward: the plane's L levels, written as 16-byte stores when L % 4 == 0
__device__ __forceinline__ void tp_forward_item(const TriPlaneArgs& a, const TpLevel* s_lv, const float* __restrict__ tabp,
                                                uint32_t b, uint32_t plane, float inv2b, bool vec,
                                                float* __restrict__ out) {
  float p[3];
  tp_point(a, b, p);
  float xw[2];
  plane_coords((int)plane, p, xw);
  const float x0 = (xw[0] + a.bound) * inv2b, x1 = (xw[1] + a.bound) * inv2b;
  const bool oob = x0 < 0.f || x0 > 1.f || x1 < 0.f || x1 > 1.f;
  float* o = out + ((size_t)b * 3 + plane) * a.L;
  for (uint32_t l4 = 0; l4 < a.L; l4 += 4) {
    float v[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
      const uint32_t l = min(l4 + q, a.L - 1);
      const TpLevel lv = s_lv[l];
      const float px = x0 * lv.scale + 0.5f, py = x1 * lv.scale + 0.5f;
      const float flx = floorf(px), fly = floorf(py);
      const float fx = px - flx, fy = py - fly;
      // (out-of-range points read cell 0: the value is discarded)
      const float* tab = tabp + lv.offset + (oob ? 0u : (uint32_t)flx + (uint32_t)fly * lv.stride);
      const float v00 = tab[0], v10 = tab[1], v01 = tab[lv.stride], v11 = tab[lv.stride + 1];
      const float r = ((1.f - fx) * (1.f - fy)) * v00 + (fx * (1.f - fy)) * v10 + ((1.f - fx) * fy) * v01 + (fx * fy) * v11;
      v[q] = oob ? 0.f : r;
    }
    if (vec) {
      *reinterpret_cast<float4*>(o + l4) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q)
        if (l4 + q < a.L) o[l4 + q] = v[q];
    }
  }
}

// All three tables resident in LDS at once (3T floats; the face fields' 3 x 37.9 KB), one work item per (point, plane):
// one staging phase and one barrier per workgroup instead of three, three times as many threads in flight, and a
// thread's 12 levels leave as three 16-byte stores into an output row that consecutive threads write contiguously
// (work item w of a workgroup owns out[(b0*3 + w) * L ...]).
constexpr int TPF_BLOCK = 1024;

__global__ void __launch_bounds__(TPF_BLOCK)
triplane_forward_all_kernel(TriPlaneArgs a, float* __restrict__ out /*[N,3L]*/) {
  extern __shared__ __align__(16) float s_tab[];          // [3][T]
  __shared__ TpLevel s_lv[TP_MAX_L];
  const uint32_t per_block = (a.N + gridDim.x - 1) / gridDim.x;
  const uint32_t b0 = blockIdx.x * per_block, b1 = min(a.N, b0 + per_block);
  const uint32_t T = (uint32_t)a.offsets[a.L];
  const float inv2b = 1.0f / (2.0f * a.bound);
  tp_levels(a, s_lv);
  // (16-byte loads, several in flight per thread: tp_stage_table)
#pragma unroll
  for (int plane = 0; plane < 3; ++plane) tp_stage_table<TPF_BLOCK, false>(s_tab + plane * T, nullptr, a.tables[plane], T);
  __syncthreads();
  if (b0 >= b1) return;
  const uint32_t items = 3u * (b1 - b0);
  const bool vec = (a.L & 3u) == 0;
  for (uint32_t w = threadIdx.x; w < items; w += TPF_BLOCK) {
    const uint32_t pt = w / 3u, plane = w - 3u * pt;
    tp_forward_item(a, s_lv, s_tab + plane * T, b0 + pt, plane, inv2b, vec, out);
  }
}

// Tables too large for LDS (the mouth field, scene/motion_net.py:346-478: 46,600 entries per plane): the same work
// items with the tables read in place -- 3 x 186 KB stay resident in L2 -- in one launch instead of the generic
// encoder's three plus the slicing and concatenation around them.
constexpr int TPG_BLOCK = 256;

__global__ void __launch_bounds__(TPG_BLOCK)
triplane_global_forward_kernel(TriPlaneArgs a, float* __restrict__ out /*[N,3L]*/) {
  __shared__ TpLevel s_lv[TP_MAX_L];
  tp_levels(a, s_lv);
  __syncthreads();
  const uint32_t w = blockIdx.x * TPG_BLOCK + threadIdx.x;
  if (w >= 3u * a.N) return;
  const uint32_t pt = w / 3u, plane = w - 3u * pt;
  tp_forward_item(a, s_lv, a.tables[plane], pt, plane, 1.0f / (2.0f * a.bound), (a.L & 3u) == 0, out);
}

// Backward, stage 1: plane by plane, the plane's table (T floats) and the gradient of the workgroup's points
// (T 64-bit fixed-point accumulators) live in LDS.  LDS FLOAT atomics run at roughly one lane at a time on gfx950
// (measured here: 14.4 M ds_add_f32 cost 80 us, the same number of ds_add_u64 nothing beside the loads), so the
// scatter adds integers: every contribution is scaled by a per-workgroup power of two chosen from the largest
// |grad| so that no sum can overflow (resolution <= 2^-40 of that maximum: finer than an fp32 running sum), the
// total is converted back once.  Integer addition is associative, so the result does not depend on the order of
// the adds: together with the fixed-order slice reduction of stage 2 the table gradient is bitwise reproducible.
// The slices are stored with plain coalesced stores (no global atomics: with ~400 points per workgroup nearly
// every cell of every level is touched, an atomic flush would cost 3T global atomics per workgroup).
constexpr int TP_BWD_BLOCK = 512;
constexpr uint32_t TP_BWD_MAX_BLOCKS = 256;

inline unsigned tp_bwd_blocks(uint32_t N) {
  return (unsigned)std::max<uint32_t>(1u, std::min<uint32_t>(div_up<uint32_t>(N, TP_BWD_BLOCK), TP_BWD_MAX_BLOCKS));
}

__global__ void __launch_bounds__(TP_BWD_BLOCK)
triplane_backward_kernel(TriPlaneArgs a, const float* __restrict__ grad /*[N,3L]*/, float* __restrict__ dxyz /*[N,3] or null*/,
                         float* __restrict__ dshift /*[N, shift_stride] or null*/, float* __restrict__ ws /*[gridDim.x][3T]*/,
                         const float* __restrict__ dxyz_add /*[N,3] or null: added to dxyz*/,
                         const float* __restrict__ dshift_add /*[N, shift_stride] or null: added to dshift*/) {
  extern __shared__ __align__(16) unsigned long long s_mem64[];      // [T] i64 gradient | [T] f32 table of the plane
  __shared__ TpLevel s_lv[TP_MAX_L];
  __shared__ float s_wmax[TP_BWD_BLOCK / 64];
  const uint32_t per_block = (a.N + gridDim.x - 1) / gridDim.x;
  const uint32_t b0 = blockIdx.x * per_block, b1 = min(a.N, b0 + per_block);
  const uint32_t T = (uint32_t)a.offsets[a.L];
  unsigned long long* s_acc = s_mem64;
  float* s_tab = reinterpret_cast<float*>(s_mem64 + T);
  const float inv2b = 1.0f / (2.0f * a.bound);
  tp_levels(a, s_lv);
  // fixed-point scale of this workgroup: |sum into one cell| <= (#points) * max|grad| < 2^62
  float gmax = 0.f;
  for (uint32_t b = b0 + threadIdx.x; b < b1; b += TP_BWD_BLOCK) {
    const float* __restrict__ g = grad + (size_t)b * 3 * a.L;
#pragma unroll 4
    for (uint32_t k = 0; k < 3 * a.L; ++k) gmax = fmaxf(gmax, fabsf(g[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
  if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = gmax;
  __syncthreads();
  gmax = s_wmax[0];
#pragma unroll
  for (int w = 1; w < TP_BWD_BLOCK / 64; ++w) gmax = fmaxf(gmax, s_wmax[w]);
  const bool usable = gmax > 0.f && gmax < INFINITY;      // all-zero (or non-finite) gradients: nothing to scatter
  int shift = 0;
  if (usable) {
    int npts_log2 = 1;
    while ((1u << npts_log2) < per_block) ++npts_log2;
    shift = 60 - (ilogbf(gmax) + 1) - npts_log2;
  }
  const double to_fixed = ldexp(1.0, shift), to_float = ldexp(1.0, -shift);
  float* slice = ws + (size_t)blockIdx.x * 3 * T;
  for (int plane = 0; plane < 3; ++plane) {
    __syncthreads();                                   // previous plane's slice fully stored
    const float* __restrict__ tab = a.tables[plane];
    tp_stage_table<TP_BWD_BLOCK, false>(s_tab, nullptr, tab, T);
#pragma unroll 8
    for (uint32_t i = threadIdx.x; i < T; i += TP_BWD_BLOCK) s_acc[i] = 0ull;
    __syncthreads();
    for (uint32_t b = b0 + threadIdx.x; b < b1; b += TP_BWD_BLOCK) {
      float p[3];
      tp_point(a, b, p);
      float xw[2];
      plane_coords(plane, p, xw);
      const float x0 = (xw[0] + a.bound) * inv2b, x1 = (xw[1] + a.bound) * inv2b;
      float gx = 0.f, gy = 0.f;
      if (!(x0 < 0.f || x0 > 1.f || x1 < 0.f || x1 > 1.f)) {
        const float* __restrict__ g = grad + (size_t)b * 3 * a.L + plane * a.L;
#pragma unroll 4
        for (uint32_t l = 0; l < a.L; ++l) {
          const TpLevel lv = s_lv[l];
          const float gl = g[l];
          const float px = x0 * lv.scale + 0.5f, py = x1 * lv.scale + 0.5f;
          const float flx = floorf(px), fly = floorf(py);
          const float fx = px - flx, fy = py - fly;
          const uint32_t i00 = lv.offset + (uint32_t)flx + (uint32_t)fly * lv.stride;
          const uint32_t i10 = i00 + 1, i01 = i00 + lv.stride, i11 = i01 + 1;
          if (usable) {
            const double gs = (double)gl * to_fixed;
            atomicAdd(&s_acc[i00], (unsigned long long)__double2ll_rn((double)((1.f - fx) * (1.f - fy)) * gs));
            atomicAdd(&s_acc[i10], (unsigned long long)__double2ll_rn((double)(fx * (1.f - fy)) * gs));
            atomicAdd(&s_acc[i01], (unsigned long long)__double2ll_rn((double)((1.f - fx) * fy) * gs));
            atomicAdd(&s_acc[i11], (unsigned long long)__double2ll_rn((double)(fx * fy) * gs));
          }
          if (dxyz) {
            const float v00 = s_tab[i00], v10 = s_tab[i10], v01 = s_tab[i01], v11 = s_tab[i11];
            // dy_dx of kernel_grid (gridencoder.cu:201-244) for D=2, linear interpolation
            gx += gl * lv.scale * ((1.f - fy) * (v10 - v00) + fy * (v11 - v01));
            gy += gl * lv.scale * ((1.f - fx) * (v01 - v00) + fx * (v11 - v10));
          }
        }
      }
      if (dxyz) {
        // the point is owned by this thread in all three planes: plain read-modify-write
        float* d = dxyz + (size_t)b * 3;
        gx *= inv2b; gy *= inv2b;
        if (plane == 0) { d[0] = gx; d[1] = gy; d[2] = 0.f; }
        else if (plane == 1) { d[1] += gx; d[2] += gy; }
        else {
          float d0 = d[0] + gx, d1 = d[1], d2 = d[2] + gy;
          if (dshift) {               // d/d shift[:, :3] = shift_scale * d/d point, the other columns get no gradient here
            float* ds = dshift + (size_t)b * a.shift_stride;
            const float* da = dshift_add ? dshift_add + (size_t)b * a.shift_stride : nullptr;
            ds[0] = a.shift_scale * d0 + (da ? da[0] : 0.f);
            ds[1] = a.shift_scale * d1 + (da ? da[1] : 0.f);
            ds[2] = a.shift_scale * d2 + (da ? da[2] : 0.f);
            for (uint32_t k = 3; k < a.shift_stride; ++k) ds[k] = da ? da[k] : 0.f;
          }
          if (dxyz_add) {             // the gradient of the position's other consumers, summed here (no extra launch)
            d0 += dxyz_add[(size_t)b * 3]; d1 += dxyz_add[(size_t)b * 3 + 1]; d2 += dxyz_add[(size_t)b * 3 + 2];
          }
          d[0] = d0; d[1] = d1; d[2] = d2;
        }
      }
    }
    __syncthreads();
#pragma unroll 8
    for (uint32_t i = threadIdx.x; i < T; i += TP_BWD_BLOCK)
      slice[plane * T + i] = (float)((double)(long long)s_acc[i] * to_float);
  }
}

// The same stage 1 for the common shape (LC levels known at compile time, at most one point per thread: N <= 256 x 512):
// the thread's point and its 3 x LC gradient values are loaded ONCE (the values also give the fixed-point scale) and
// stay in registers across the three plane phases, and the position gradient is assembled in registers and stored once.
// The general kernel re-reads them per plane and level -- about fifteen dependent global round trips per thread in a
// kernel whose LDS work takes ~30 us.  Same arithmetic, same summation order: identical bits.
template <int LC>
__global__ void __launch_bounds__(TP_BWD_BLOCK)
triplane_backward_cached_kernel(TriPlaneArgs a, const float* __restrict__ grad /*[N,3LC]*/,
                                float* __restrict__ dxyz /*[N,3] or null*/,
                                float* __restrict__ dshift /*[N, shift_stride] or null*/,
                                float* __restrict__ ws /*[gridDim.x][3T]*/, const float* __restrict__ dxyz_add,
                                const float* __restrict__ dshift_add) {
  extern __shared__ __align__(16) unsigned long long s_mem64[];      // [T] i64 gradient | [T] f32 table of the plane
  __shared__ TpLevel s_lv[TP_MAX_L];
  __shared__ float s_wmax[TP_BWD_BLOCK / 64];
  const uint32_t per_block = (a.N + gridDim.x - 1) / gridDim.x;      // <= TP_BWD_BLOCK (host)
  const uint32_t b0 = blockIdx.x * per_block, b1 = min(a.N, b0 + per_block);
  const uint32_t T = (uint32_t)a.offsets[LC];
  unsigned long long* s_acc = s_mem64;
  float* s_tab = reinterpret_cast<float*>(s_mem64 + T);
  const float inv2b = 1.0f / (2.0f * a.bound);
  tp_levels(a, s_lv);
  const uint32_t b = b0 + threadIdx.x;
  const bool mine = threadIdx.x < per_block && b < b1;
  float p[3] = {0.f, 0.f, 0.f};
  float gc[3 * LC];
  float gmax = 0.f;
  if (mine) {
    tp_point(a, b, p);
    const float* __restrict__ g = grad + (size_t)b * 3 * LC;
    if ((3 * LC) % 4 == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0) {
#pragma unroll
      for (int k = 0; k < 3 * LC / 4; ++k) {
        const float4 t = reinterpret_cast<const float4*>(g)[k];
        gc[4 * k] = t.x; gc[4 * k + 1] = t.y; gc[4 * k + 2] = t.z; gc[4 * k + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3 * LC; ++k) gc[k] = g[k];
    }
#pragma unroll
    for (int k = 0; k < 3 * LC; ++k) gmax = fmaxf(gmax, fabsf(gc[k]));
  } else {
#pragma unroll
    for (int k = 0; k < 3 * LC; ++k) gc[k] = 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
  if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = gmax;
  __syncthreads();
  gmax = s_wmax[0];
#pragma unroll
  for (int w = 1; w < TP_BWD_BLOCK / 64; ++w) gmax = fmaxf(gmax, s_wmax[w]);
  const bool usable = gmax > 0.f && gmax < INFINITY;
  int shift = 0;
  if (usable) {
    int npts_log2 = 1;
    while ((1u << npts_log2) < per_block) ++npts_log2;
    shift = 60 - (ilogbf(gmax) + 1) - npts_log2;
  }
  const double to_fixed = ldexp(1.0, shift), to_float = ldexp(1.0, -shift);
  float* slice = ws + (size_t)blockIdx.x * 3 * T;
  float dsum[3] = {0.f, 0.f, 0.f};                    // d/d point, in the general kernel's order of additions
#pragma unroll
  for (int plane = 0; plane < 3; ++plane) {
    __syncthreads();                                   // previous plane's slice fully stored
    tp_stage_table<TP_BWD_BLOCK, false>(s_tab, nullptr, a.tables[plane], T);
#pragma unroll 8
    for (uint32_t i = threadIdx.x; i < T; i += TP_BWD_BLOCK) s_acc[i] = 0ull;
    __syncthreads();
    if (mine) {
      float xw[2];
      plane_coords(plane, p, xw);
      const float x0 = (xw[0] + a.bound) * inv2b, x1 = (xw[1] + a.bound) * inv2b;
      float gx = 0.f, gy = 0.f;
      if (!(x0 < 0.f || x0 > 1.f || x1 < 0.f || x1 > 1.f)) {
#pragma unroll
        for (int l = 0; l < LC; ++l) {
          const TpLevel lv = s_lv[l];
          const float gl = gc[plane * LC + l];
          const float px = x0 * lv.scale + 0.5f, py = x1 * lv.scale + 0.5f;
          const float flx = floorf(px), fly = floorf(py);
          const float fx = px - flx, fy = py - fly;
          const uint32_t i00 = lv.offset + (uint32_t)flx + (uint32_t)fly * lv.stride;
          const uint32_t i10 = i00 + 1, i01 = i00 + lv.stride, i11 = i01 + 1;
          if (usable) {
            const double gs = (double)gl * to_fixed;
            atomicAdd(&s_acc[i00], (unsigned long long)__double2ll_rn((double)((1.f - fx) * (1.f - fy)) * gs));
            atomicAdd(&s_acc[i10], (unsigned long long)__double2ll_rn((double)(fx * (1.f - fy)) * gs));
            atomicAdd(&s_acc[i01], (unsigned long long)__double2ll_rn((double)((1.f - fx) * fy) * gs));
            atomicAdd(&s_acc[i11], (unsigned long long)__double2ll_rn((double)(fx * fy) * gs));
          }
          if (dxyz) {
            const float v00 = s_tab[i00], v10 = s_tab[i10], v01 = s_tab[i01], v11 = s_tab[i11];
            gx += gl * lv.scale * ((1.f - fy) * (v10 - v00) + fy * (v11 - v01));
            gy += gl * lv.scale * ((1.f - fx) * (v01 - v00) + fx * (v11 - v10));
          }
        }
      }
      gx *= inv2b; gy *= inv2b;
      // xy = (x,y), yz = (y,z), xz = (x,z): d[0] = gx0 + gx2, d[1] = gy0 + gx1, d[2] = gy1 + gy2
      if (plane == 0) { dsum[0] = gx; dsum[1] = gy; dsum[2] = 0.f; }
      else if (plane == 1) { dsum[1] += gx; dsum[2] += gy; }
      else { dsum[0] += gx; dsum[2] += gy; }
    }
    __syncthreads();
#pragma unroll 8
    for (uint32_t i = threadIdx.x; i < T; i += TP_BWD_BLOCK)
      slice[plane * T + i] = (float)((double)(long long)s_acc[i] * to_float);
  }
  if (mine && dxyz) {
    float d0 = dsum[0], d1 = dsum[1], d2 = dsum[2];
    if (dshift) {
      float* ds = dshift + (size_t)b * a.shift_stride;
      const float* da = dshift_add ? dshift_add + (size_t)b * a.shift_stride : nullptr;
      ds[0] = a.shift_scale * d0 + (da ? da[0] : 0.f);
      ds[1] = a.shift_scale * d1 + (da ? da[1] : 0.f);
      ds[2] = a.shift_scale * d2 + (da ? da[2] : 0.f);
      for (uint32_t k = 3; k < a.shift_stride; ++k) ds[k] = da ? da[k] : 0.f;
    }
    if (dxyz_add) {
      d0 += dxyz_add[(size_t)b * 3]; d1 += dxyz_add[(size_t)b * 3 + 1]; d2 += dxyz_add[(size_t)b * 3 + 2];
    }
    float* d = dxyz + (size_t)b * 3;
    d[0] = d0; d[1] = d1; d[2] = d2;
  }
}

// stage 2: dtab[plane][i] = sum over workgroup slices, fixed order.  32 cells x 8 slice-groups per workgroup.
__global__ void __launch_bounds__(256)
triplane_reduce_kernel(const float* __restrict__ ws, uint32_t nslices, uint32_t T, float* __restrict__ dtab0,
                       float* __restrict__ dtab1, float* __restrict__ dtab2) {
  __shared__ float s_part[8][32];
  const uint32_t cell = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
  const uint32_t per = (nslices + 7) / 8;
  const uint32_t s0 = grp * per, s1 = min(nslices, s0 + per);
  float acc = 0.f;
  if (cell < 3 * T) {
#pragma unroll 8
    for (uint32_t sl = s0; sl < s1; ++sl) acc += ws[(size_t)sl * 3 * T + cell];
  }
  s_part[grp][threadIdx.x & 31] = acc;
  __syncthreads();
  if (threadIdx.x < 32 && cell < 3 * T) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += s_part[k][threadIdx.x];
    const uint32_t plane = cell / T, i = cell - plane * T;
    (plane == 0 ? dtab0 : (plane == 1 ? dtab1 : dtab2))[i] = v;
  }
}

// The same sums with 16-byte loads (T % 4 == 0, 16-byte aligned slices): four cells per thread, 128 cells x 8
// slice-groups per workgroup; per cell the additions are the same, in the same order.
__global__ void __launch_bounds__(256)
triplane_reduce4_kernel(const float* __restrict__ ws, uint32_t nslices, uint32_t T, float* __restrict__ dtab0,
                        float* __restrict__ dtab1, float* __restrict__ dtab2) {
  __shared__ float4 s_part[8][32];
  const uint32_t quad = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;       // cells 4*quad .. 4*quad+3
  const uint32_t nquads = 3 * T / 4;
  const uint32_t per = (nslices + 7) / 8;
  const uint32_t s0 = grp * per, s1 = min(nslices, s0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (quad < nquads) {
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(ws);
#pragma unroll 8
    for (uint32_t sl = s0; sl < s1; ++sl) {
      const float4 t = w4[(size_t)sl * nquads + quad];
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
  }
  s_part[grp][threadIdx.x & 31] = acc;
  __syncthreads();
  if (threadIdx.x < 32 && quad < nquads) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 t = s_part[k][threadIdx.x];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const uint32_t cell = 4 * quad, plane = cell / T, i = cell - plane * T;       // T % 4 == 0: a quad stays in one plane
    *reinterpret_cast<float4*>((plane == 0 ? dtab0 : (plane == 1 ? dtab1 : dtab2)) + i) = v;
  }
}

inline int launch_triplane_reduce(const float* ws, uint32_t nslices, uint32_t T, float* d0, float* d1, float* d2,
                                  hipStream_t s) {
  const bool vec = (T & 3u) == 0 && ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(d0) |
                                      reinterpret_cast<uintptr_t>(d1) | reinterpret_cast<uintptr_t>(d2)) & 15) == 0;
  if (vec) triplane_reduce4_kernel<<<div_up<uint32_t>(3 * T / 4, 32), 256, 0, s>>>(ws, nslices, T, d0, d1, d2);
  else triplane_reduce_kernel<<<div_up<uint32_t>(3 * T, 32), 256, 0, s>>>(ws, nslices, T, d0, d1, d2);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

// Backward for tables read in place.  A wave owns 21 points: lanes 3i, 3i+1, 3i+2 hold the three planes of point i
// (lane 63 idles), so the position gradient is assembled with two shuffles and stored by one lane -- no atomics, no
// zeroed buffer, the same summation order as the LDS kernel.  The table gradient is scattered with global float
// atomics onto tables the caller (instag_triplane_backward) zeroed: the summation order is not fixed, as in the
// reference (gridencoder.cu:300-330 kernel_grid_backward's atomicAdd).
constexpr uint32_t TPG_PTS_PER_WAVE = 21;

__global__ void __launch_bounds__(256)
triplane_zero_kernel(float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2, uint32_t T) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < T) { d0[i] = 0.f; d1[i] = 0.f; d2[i] = 0.f; }
}

template <bool ATOMIC_TABLES>
__global__ void __launch_bounds__(TPG_BLOCK)
triplane_global_backward_kernel(TriPlaneArgs a, const float* __restrict__ grad /*[N,3L]*/,
                                float* __restrict__ dxyz /*[N,3] or null*/,
                                float* __restrict__ dshift /*[N, shift_stride] or null*/, float* __restrict__ dtab0,
                                float* __restrict__ dtab1, float* __restrict__ dtab2,
                                const float* __restrict__ dxyz_add, const float* __restrict__ dshift_add) {
  __shared__ TpLevel s_lv[TP_MAX_L];
  tp_levels(a, s_lv);
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * TPG_BLOCK + threadIdx.x) >> 6;
  const uint32_t pt = lane / 3u, plane = lane - 3u * pt;
  const uint32_t b = wave * TPG_PTS_PER_WAVE + pt;
  const bool live = lane < 3u * TPG_PTS_PER_WAVE && b < a.N;
  const float inv2b = 1.0f / (2.0f * a.bound);
  float gx = 0.f, gy = 0.f;
  if (live) {
    float p[3];
    tp_point(a, b, p);
    float xw[2];
    plane_coords((int)plane, p, xw);
    const float x0 = (xw[0] + a.bound) * inv2b, x1 = (xw[1] + a.bound) * inv2b;
    if (!(x0 < 0.f || x0 > 1.f || x1 < 0.f || x1 > 1.f)) {
      const float* __restrict__ g = grad + ((size_t)b * 3 + plane) * a.L;
      const float* __restrict__ tab = a.tables[plane];
      float* __restrict__ dtab = plane == 0 ? dtab0 : (plane == 1 ? dtab1 : dtab2);
#pragma unroll 4
      for (uint32_t l = 0; l < a.L; ++l) {
        const TpLevel lv = s_lv[l];
        const float gl = g[l];
        const float px = x0 * lv.scale + 0.5f, py = x1 * lv.scale + 0.5f;
        const float flx = floorf(px), fly = floorf(py);
        const float fx = px - flx, fy = py - fly;
        const uint32_t i00 = lv.offset + (uint32_t)flx + (uint32_t)fly * lv.stride;
        const uint32_t i10 = i00 + 1, i01 = i00 + lv.stride, i11 = i01 + 1;
        if (ATOMIC_TABLES && gl != 0.f) {
          atomicAdd(&dtab[i00], ((1.f - fx) * (1.f - fy)) * gl);
          atomicAdd(&dtab[i10], (fx * (1.f - fy)) * gl);
          atomicAdd(&dtab[i01], ((1.f - fx) * fy) * gl);
          atomicAdd(&dtab[i11], (fx * fy) * gl);
        }
        if (dxyz) {
          const float v00 = tab[i00], v10 = tab[i10], v01 = tab[i01], v11 = tab[i11];
          gx += gl * lv.scale * ((1.f - fy) * (v10 - v00) + fy * (v11 - v01));
          gy += gl * lv.scale * ((1.f - fx) * (v01 - v00) + fx * (v11 - v10));
        }
      }
    }
  }
  if (!dxyz) return;                                   // uniform
  gx *= inv2b; gy *= inv2b;
  const float gx1 = __shfl_down(gx, 1), gy1 = __shfl_down(gy, 1), gx2 = __shfl_down(gx, 2), gy2 = __shfl_down(gy, 2);
  if (live && plane == 0) {
    // xy = (x,y), yz = (y,z), xz = (x,z)
    float d0 = gx + gx2, d1 = gy + gx1, d2 = gy1 + gy2;
    if (dshift) {
      float* ds = dshift + (size_t)b * a.shift_stride;
      const float* da = dshift_add ? dshift_add + (size_t)b * a.shift_stride : nullptr;
      ds[0] = a.shift_scale * d0 + (da ? da[0] : 0.f);
      ds[1] = a.shift_scale * d1 + (da ? da[1] : 0.f);
      ds[2] = a.shift_scale * d2 + (da ? da[2] : 0.f);
      for (uint32_t k = 3; k < a.shift_stride; ++k) ds[k] = da ? da[k] : 0.f;
    }
    if (dxyz_add) {
      d0 += dxyz_add[(size_t)b * 3]; d1 += dxyz_add[(size_t)b * 3 + 1]; d2 += dxyz_add[(size_t)b * 3 + 2];
    }
    float* d = dxyz + (size_t)b * 3;
    d[0] = d0; d[1] = d1; d[2] = d2;
  }
}


// Table gradient for tables read in place, one (plane, level) and one chunk of points per workgroup: the level's
// cells (4,225 for the mouth field, 34 KB of 64-bit fixed-point accumulators) live in LDS exactly as in
// triplane_backward_kernel, each workgroup stores its slice of the chunk's [3T] gradient with plain stores and
// triplane_reduce_kernel adds the chunks in a fixed order: no global atomics (2.9 M contended float atomics at 20k
// clustered points cost 148 us), no zeroed buffer, bitwise reproducible.
constexpr int TPL_BLOCK = 512;
constexpr uint32_t TPL_MAX_LEVEL_CELLS = 16 * 1024;            // 128 KB of accumulators
constexpr uint32_t TPL_CHUNK = 4096, TPL_MAX_CHUNKS = 32;

inline uint32_t tpl_chunk_points(uint32_t N) { return std::max(TPL_CHUNK, div_up<uint32_t>(N, TPL_MAX_CHUNKS)); }
inline uint32_t tpl_chunks(uint32_t N) { return std::max(1u, div_up<uint32_t>(N, tpl_chunk_points(N))); }

__global__ void __launch_bounds__(TPL_BLOCK)
triplane_level_backward_kernel(TriPlaneArgs a, const float* __restrict__ grad /*[N,3L]*/,
                               float* __restrict__ ws /*[chunks][3T]*/, uint32_t chunk_pts) {
  extern __shared__ __align__(16) unsigned long long s_lvl[];
  __shared__ float s_wmax[TPL_BLOCK / 64];
  const uint32_t plane = blockIdx.x / a.L, level = blockIdx.x - plane * a.L;
  const uint32_t T = (uint32_t)a.offsets[a.L];
  const uint32_t off0 = (uint32_t)a.offsets[level], cells = (uint32_t)a.offsets[level + 1] - off0;
  const LevelGeom lg = level_geom(a.offsets, level, a.S, a.H);
  const uint32_t stride = lg.resolution + 1;
  const uint32_t b0 = blockIdx.y * chunk_pts, b1 = min(a.N, b0 + chunk_pts);
  const float inv2b = 1.0f / (2.0f * a.bound);
  const float* __restrict__ gcol = grad + (size_t)plane * a.L + level;
  float gmax = 0.f;
  for (uint32_t b = b0 + threadIdx.x; b < b1; b += TPL_BLOCK) gmax = fmaxf(gmax, fabsf(gcol[(size_t)b * 3 * a.L]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
  if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = gmax;
  for (uint32_t i = threadIdx.x; i < cells; i += TPL_BLOCK) s_lvl[i] = 0ull;
  __syncthreads();
  gmax = s_wmax[0];
#pragma unroll
  for (int w = 1; w < TPL_BLOCK / 64; ++w) gmax = fmaxf(gmax, s_wmax[w]);
  const bool usable = gmax > 0.f && gmax < INFINITY;
  int shift = 0;
  if (usable) {
    int npts_log2 = 1;
    while ((1u << npts_log2) < chunk_pts) ++npts_log2;
    shift = 60 - (ilogbf(gmax) + 1) - npts_log2;
  }
  const double to_fixed = ldexp(1.0, shift), to_float = ldexp(1.0, -shift);
  if (usable) {
    for (uint32_t b = b0 + threadIdx.x; b < b1; b += TPL_BLOCK) {
      float p[3];
      tp_point(a, b, p);
      float xw[2];
      plane_coords((int)plane, p, xw);
      const float x0 = (xw[0] + a.bound) * inv2b, x1 = (xw[1] + a.bound) * inv2b;
      if (x0 < 0.f || x0 > 1.f || x1 < 0.f || x1 > 1.f) continue;
      const float gl = gcol[(size_t)b * 3 * a.L];
      const float px = x0 * lg.scale + 0.5f, py = x1 * lg.scale + 0.5f;
      const float flx = floorf(px), fly = floorf(py);
      const float fx = px - flx, fy = py - fly;
      const uint32_t i00 = (uint32_t)flx + (uint32_t)fly * stride;
      const double gs = (double)gl * to_fixed;
      atomicAdd(&s_lvl[i00], (unsigned long long)__double2ll_rn((double)((1.f - fx) * (1.f - fy)) * gs));
      atomicAdd(&s_lvl[i00 + 1], (unsigned long long)__double2ll_rn((double)(fx * (1.f - fy)) * gs));
      atomicAdd(&s_lvl[i00 + stride], (unsigned long long)__double2ll_rn((double)((1.f - fx) * fy) * gs));
      atomicAdd(&s_lvl[i00 + stride + 1], (unsigned long long)__double2ll_rn((double)(fx * fy) * gs));
    }
  }
  __syncthreads();
  float* __restrict__ slice = ws + ((size_t)blockIdx.y * 3 + plane) * T + off0;
  for (uint32_t i = threadIdx.x; i < cells; i += TPL_BLOCK)
    slice[i] = (float)((double)(long long)s_lvl[i] * to_float);
}

}  // namespace
}  // namespace instag

extern "C" {

int instag_triplane_forward(const float* xyz, const float* table_xy, const float* table_yz, const float* table_xz,
                            const int32_t* offsets, float* out, const float* shift, uint32_t shift_stride,
                            float shift_scale, uint32_t N, uint32_t L, float S, uint32_t H,
                            float bound, uint32_t total_params, instag_stream_t stream) {
  using namespace instag;
  INSTAG_REQUIRE(xyz && table_xy && table_yz && table_xz && offsets && out, "triplane_forward: NULL tensor");
  INSTAG_REQUIRE(L >= 1 && L <= TP_MAX_L, "triplane: L must be in [1,16]");
  INSTAG_REQUIRE(shift == nullptr || shift_stride >= 3, "triplane: shift needs at least 3 columns");
  if (N == 0) return INSTAG_OK;
  TriPlaneArgs a{xyz, {table_xy, table_yz, table_xz}, offsets, N, L, H, S, bound, shift, shift_stride, shift_scale};
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_GRID_FWD, s);
  const size_t all_bytes = (size_t)3 * total_params * sizeof(float);
  if (!tp_fits_lds(total_params)) {
    INSTAG_REQUIRE(N <= 0x7fffffffu / 3u, "triplane: N too large");
    triplane_global_forward_kernel<<<div_up<uint32_t>(3u * N, TPG_BLOCK), TPG_BLOCK, 0, s>>>(a, out);
  } else if (all_bytes <= 150 * 1024) {
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(triplane_forward_all_kernel), 150 * 1024)) return rc;
    const unsigned blocks = std::max(1u, std::min(256u, div_up<uint32_t>(3u * N, TPF_BLOCK)));
    triplane_forward_all_kernel<<<blocks, TPF_BLOCK, all_bytes, s>>>(a, out);
  } else {
    triplane_forward_kernel<<<fwd_blocks(N), GRID_BLOCK, total_params * sizeof(float), s>>>(a, out);
  }
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

size_t instag_triplane_backward_workspace_bytes(uint32_t N, uint32_t total_params) {
  if (!instag::tp_fits_lds(total_params))                // tables read in place: one [3T] slice per chunk of points
    return (size_t)instag::tpl_chunks(N) * 3 * total_params * sizeof(float);
  return (size_t)instag::tp_bwd_blocks(N) * 3 * total_params * sizeof(float);
}

int instag_triplane_backward(const float* grad, const float* xyz, const float* table_xy, const float* table_yz,
                             const float* table_xz, const int32_t* offsets, float* dxyz, float* dtable_xy,
                             float* dtable_yz, float* dtable_xz, void* workspace, size_t workspace_bytes,
                             const float* shift, uint32_t shift_stride, float shift_scale, float* dshift, uint32_t N,
                             uint32_t L, float S, uint32_t H, float bound, uint32_t total_params,
                             const float* dxyz_add, const float* dshift_add, instag_stream_t stream) {
  using namespace instag;
  INSTAG_REQUIRE(dxyz_add == nullptr || (dxyz != nullptr && dxyz_add != dxyz), "triplane_backward: dxyz_add needs a distinct dxyz");
  INSTAG_REQUIRE(dshift_add == nullptr || (dshift != nullptr && dshift_add != dshift),
                 "triplane_backward: dshift_add needs a distinct dshift");
  INSTAG_REQUIRE(grad && xyz && table_xy && table_yz && table_xz && offsets && dtable_xy && dtable_yz && dtable_xz,
                 "triplane_backward: NULL tensor");
  INSTAG_REQUIRE(L >= 1 && L <= TP_MAX_L, "triplane: L must be in [1,16]");
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) {
    for (float* d : {dtable_xy, dtable_yz, dtable_xz})
      INSTAG_CHECK_HIP(hipMemsetAsync(d, 0, (size_t)total_params * sizeof(float), s));
    return INSTAG_OK;
  }
  if (!tp_fits_lds(total_params)) {
    INSTAG_REQUIRE(shift == nullptr || shift_stride >= 3, "triplane: shift needs at least 3 columns");
    INSTAG_REQUIRE(dshift == nullptr || (shift != nullptr && dxyz != nullptr), "triplane_backward: dshift needs shift and dxyz");
    INSTAG_REQUIRE(N <= 0x7fffffffu / 3u, "triplane: N too large");
    TriPlaneArgs a{xyz, {table_xy, table_yz, table_xz}, offsets, N, L, H, S, bound, shift, shift_stride, shift_scale};
    ProfScope p(K_GRID_BWD, s);
    const uint32_t waves = div_up<uint32_t>(N, TPG_PTS_PER_WAVE);
    const unsigned pt_blocks = div_up<uint32_t>(waves, TPG_BLOCK / 64);
    // largest level of the table (host copy of the offsets is not available: bound it by the densest possible level)
    const uint32_t finest = (uint32_t)std::ceil(std::exp2((float)(L - 1) * std::max(S, 0.f)) * (float)H) + 1;
    const uint32_t max_cells = ((finest + 1) * (finest + 1) + 7u) / 8u * 8u;
    if (max_cells <= TPL_MAX_LEVEL_CELLS) {
      const uint32_t chunks = tpl_chunks(N), chunk_pts = tpl_chunk_points(N);
      const size_t need = (size_t)chunks * 3 * total_params * sizeof(float);
      if (!workspace || workspace_bytes < need) { set_error("triplane_backward: workspace too small"); return INSTAG_E_SPACE; }
      if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(triplane_level_backward_kernel),
                                       TPL_MAX_LEVEL_CELLS * sizeof(unsigned long long))) return rc;
      triplane_level_backward_kernel<<<dim3(3 * L, chunks), TPL_BLOCK, (size_t)max_cells * sizeof(unsigned long long), s>>>(
          a, grad, (float*)workspace, chunk_pts);
      INSTAG_CHECK_LAUNCH();
      if (int rc = launch_triplane_reduce((const float*)workspace, chunks, total_params, dtable_xy, dtable_yz, dtable_xz, s))
        return rc;
      if (dxyz) {
        triplane_global_backward_kernel<false><<<pt_blocks, TPG_BLOCK, 0, s>>>(a, grad, dxyz, dshift, dtable_xy, dtable_yz,
                                                                              dtable_xz, dxyz_add, dshift_add);
        INSTAG_CHECK_LAUNCH();
      }
      return INSTAG_OK;
    }
    // a level too large for LDS: scatter with global float atomics onto zeroed tables (order not fixed, as the
    // reference's kernel_grid_backward)
    triplane_zero_kernel<<<div_up<uint32_t>(total_params, 256), 256, 0, s>>>(dtable_xy, dtable_yz, dtable_xz, total_params);
    INSTAG_CHECK_LAUNCH();
    triplane_global_backward_kernel<true><<<pt_blocks, TPG_BLOCK, 0, s>>>(a, grad, dxyz, dshift, dtable_xy, dtable_yz,
                                                                         dtable_xz, dxyz_add, dshift_add);
    INSTAG_CHECK_LAUNCH();
    return INSTAG_OK;
  }
  const unsigned blocks = tp_bwd_blocks(N);
  const size_t need = (size_t)blocks * 3 * total_params * sizeof(float);
  if (!workspace || workspace_bytes < need) { set_error("triplane_backward: workspace too small"); return INSTAG_E_SPACE; }
  if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(triplane_backward_kernel), 156 * 1024)) return rc;
  INSTAG_REQUIRE(shift == nullptr || shift_stride >= 3, "triplane: shift needs at least 3 columns");
  INSTAG_REQUIRE(dshift == nullptr || (shift != nullptr && dxyz != nullptr), "triplane_backward: dshift needs shift and dxyz");
  TriPlaneArgs a{xyz, {table_xy, table_yz, table_xz}, offsets, N, L, H, S, bound, shift, shift_stride, shift_scale};
  ProfScope p(K_GRID_BWD, s);
  if (L == 12 && div_up<uint32_t>(N, blocks) <= (uint32_t)TP_BWD_BLOCK) {
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(triplane_backward_cached_kernel<12>), 156 * 1024)) return rc;
    triplane_backward_cached_kernel<12><<<blocks, TP_BWD_BLOCK, (size_t)12 * total_params, s>>>(
        a, grad, dxyz, dshift, (float*)workspace, dxyz_add, dshift_add);
  } else {
    triplane_backward_kernel<<<blocks, TP_BWD_BLOCK, (size_t)12 * total_params, s>>>(a, grad, dxyz, dshift,
                                                                                      (float*)workspace, dxyz_add,
                                                                                      dshift_add);
  }
  INSTAG_CHECK_LAUNCH();
  return launch_triplane_reduce((const float*)workspace, blocks, total_params, dtable_xy, dtable_yz, dtable_xz, s);
}

}  // extern "C"
