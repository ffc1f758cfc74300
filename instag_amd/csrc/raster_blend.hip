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
// among the Gaussian's kept tiles).  The reduction of a Gaussian's gradient over the 256 pixels of the tile runs
// as two small fp32 GEMMs on the matrix cores (see blend_backward_kernel) and the row is written once
// with plain 16-byte stores.  The per-Gaussian sum over its rows happens in raster_backward.hip.
#include <cstdlib>
#include <cstring>

#include "raster_internal.hpp"

namespace instag {
namespace {

using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int BLOCK = 256;
constexpr int NCH = 8;  // r g b depth nx ny nz extra
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_MIN = 0.0001f;

// AUX: a second colour set [N,3] is blended over the same instances with the same alpha / T (the reference renders
// it as a separate rasterizer call on detached geometry: gaussian_renderer/__init__.py:243-258, the attention map)
//
// A tile's duration is its longest wave's instruction count: at C3's ~1.6 workgroups per CU a wave has its SIMD almost
// to itself and issues one instruction (vector, scalar or LDS alike) every ~4.5 cycles, so the inner loop is written
// for FEW INSTRUCTIONS per (Gaussian, pixel):
//  * the conic is rescaled when a record is staged (-0.5 log2e A, -0.5 log2e C, -log2e B), so the exponent comes out in
//    the log2 domain: two packed multiplies, one multiply, one add, one fma, then v_exp_f32 directly;
//  * a pixel that is finished moves to (1e18, 1e18): every later exponent is hugely negative, alpha = 0 < 1/255, the
//    Gaussian does not count -- no `done` predicate in the loop;
//  * a Gaussian that does not count has alpha' = 0, so test_T = T exactly and the weight is 0 without any select, and
//    since T >= T_MIN holds while a pixel is unfinished, "some pixel stops within these four Gaussians" is one compare
//    of the fourth running product plus ONE wave-uniform branch per four Gaussians; the rare path (each lane takes it
//    once per tile) redoes the four steps with the per-lane stop logic.
// The running products are the same sequence of fp32 operations as one-at-a-time blending: T, n_contrib are unchanged.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float FAR_PIXEL = 1e18f;

#ifdef BLEND_DBG2
// (diagnostic build only: who claimed which segment slot, and what a wait for a post that gave up was looking at)
__device__ uint32_t g_seg_owner[1 << 16];
__device__ uint32_t g_stall_log[8 * 512 + 8];
__device__ uint32_t* g_stall_host;            // (pinned host memory: readable while a kernel hangs)
#endif
#ifdef BLEND_DBG
// (diagnostic build only: when each tile's workgroup started and ended, and where it ran -- scripts/probes/blend_tile_profile.py)
__device__ uint32_t g_blend_dbg[4096 * 8];
#endif

template <bool AUX>
__global__ void __launch_bounds__(BLOCK)
blend_forward_kernel(Camera c, const int32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                     const float* __restrict__ rec2d, uint32_t* __restrict__ n_contrib,
                     float* __restrict__ final_T, float* __restrict__ out_color,
                     float* __restrict__ out_depth, float* __restrict__ out_normal,
                     float* __restrict__ out_alpha, float* __restrict__ out_extra,
                     const float* __restrict__ aux_colors, float* __restrict__ out_aux,
                     uint32_t* __restrict__ seg_queue, uint32_t* __restrict__ seg_count,
                     float* __restrict__ seg_state, uint32_t* __restrict__ tile_rounds) {
  // One LDS array per read of the inner loop, each read a whole ds_read_b128 (4 LDS cycles per wave) or ds_read_b64
  // (2): the LDS array, shared by every wave of the CU, is the forward pass's scarcest resource -- a record read as
  // 64-bit halves of neighbouring float4 (what the compiler makes of an array of structs) costs twice as much.
  __shared__ float4 s_geo[BLOCK];                  // x y A' C'
  __shared__ float2 s_bo[BLOCK];                   // B' opacity
  __shared__ float4 s_c0[BLOCK], s_c1[BLOCK];      // r g b depth | nx ny nz extra
  __shared__ float4 s_c2[AUX ? BLOCK : 1];         // aux r g b -
  const int tile = blockIdx.x;
  const int tx = tile % c.grid_x, ty = tile / c.grid_x;
  const int tid = threadIdx.x;
  const int pxi = tx * TILE_X + (tid & 15), pyi = ty * TILE_Y + (tid >> 4);
  const bool inside = pxi < c.W && pyi < c.H;
  const int start = ranges[2 * tile], end = ranges[2 * tile + 1];
  const int rounds = (end - start + BLOCK - 1) / BLOCK;
  int toDo = end - start;
#ifdef BLEND_DBG
  const uint64_t dbg_t0 = wall_clock64();
#endif
  // every SEG_LEN entries of the list are a segment with a slot of its own (common.hpp BinningLayout): the per-pixel
  // state after each segment the tile walks is kept there, so that the backward pass can start anywhere
  static_assert(BLOCK % SEG_LEN == 0 && SEG_LEN % 4 == 0, "a batch is a whole number of segments");
  constexpr int SUBS = BLOCK / SEG_LEN;
  const int slot0 = start / SEG_LEN + tile;
  int walked = 0;
  bool wave_done = false;

  f32x2 pix = inside ? f32x2{(float)pxi, (float)pyi} : f32x2{FAR_PIXEL, FAR_PIXEL};
  wave_done = __builtin_amdgcn_ballot_w64(inside) == 0;
  float T = 1.0f;
  uint32_t last_contributor = 0;
  // channel accumulators as float pairs: the eight FMAs per Gaussian become four v_pk_fma_f32 (the record keeps
  // (r,g) (b,depth) (nx,ny) (nz,extra) in adjacent registers)
  f32x2 acc2[NCH / 2];
#pragma unroll
  for (int k = 0; k < NCH / 2; ++k) acc2[k] = f32x2{0.f, 0.f};
  f32x2 xacc2 = {0.f, 0.f}, xacc3 = {0.f, 0.f};

  // the records of batch i+1 are fetched while batch i is being blended; a slot past the end of the list holds a record
  // with opacity 0 (never counts), so the loop below runs whole groups of four
  float4 nrec0, nrec1, nrec2, nrec3;
  float nauxb = 0.f;
  auto fetch = [&](int pos) {
    nrec0 = make_float4(0.f, 0.f, 0.f, 0.f); nrec1 = nrec0; nrec2 = nrec0; nrec3 = nrec0; nauxb = 0.f;
    if (start + pos < end) {
      const uint32_t gid = point_list[start + pos];
      const float4* r = reinterpret_cast<const float4*>(rec2d + (size_t)gid * REC_FLOATS);
      const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
      nrec0 = make_float4(r0.x, r0.y, (-0.5f * LOG2E) * r0.z, (-0.5f * LOG2E) * r1.x);
      nrec1 = make_float4(-LOG2E * r0.w, r1.y, r1.z, r1.w);
      nrec2 = r2;
      nrec3 = make_float4(r3.x, r3.y, 0.f, 0.f);
      if (AUX) {
        nrec3.z = aux_colors[3 * (size_t)gid]; nrec3.w = aux_colors[3 * (size_t)gid + 1];
        nauxb = aux_colors[3 * (size_t)gid + 2];
      }
    }
  };
  fetch(tid);
  for (int i = 0; i < rounds; ++i, toDo -= BLOCK) {
    if (__syncthreads_count(pix.x > 0.5f * FAR_PIXEL) == BLOCK) break;
    s_geo[tid] = nrec0; s_bo[tid] = make_float2(nrec1.x, nrec1.y);
    s_c0[tid] = make_float4(nrec1.z, nrec1.w, nrec2.x, nrec2.y);
    s_c1[tid] = make_float4(nrec2.z, nrec2.w, nrec3.x, nrec3.y);
    if (AUX) s_c2[tid] = make_float4(nrec3.z, nrec3.w, nauxb, 0.f);
    __syncthreads();
    fetch((i + 1) * BLOCK + tid);
    const int groups = (min(BLOCK, toDo) + 3) >> 2;
    for (int h = 0; h < SUBS && h * (SEG_LEN / 4) < groups; ++h) {
      const int g_end = min(groups, (h + 1) * (SEG_LEN / 4));
      for (int g = h * (SEG_LEN / 4); g < g_end && !wave_done; ++g) {
        float al[4], w[4];
        bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 a = s_geo[4 * g + k];
          const float2 b = s_bo[4 * g + k];
          const f32x2 d = f32x2{a.x, a.y} - pix;
          const f32x2 u = (d * f32x2{a.z, a.w}) * d;
          const float p2 = __builtin_fmaf(b.x, d.x * d.y, u.x + u.y);
          const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(p2));
          hit[k] = !(p2 > 0.0f) && !(alpha < ALPHA_MIN);
          al[k] = hit[k] ? alpha : 0.f;
        }
        const float t1 = T * (1.0f - al[0]);
        const float t2 = t1 * (1.0f - al[1]);
        const float t3 = t2 * (1.0f - al[2]);
        float t4 = t3 * (1.0f - al[3]);
        w[0] = al[0] * T; w[1] = al[1] * t1; w[2] = al[2] * t2; w[3] = al[3] * t3;
        uint32_t code = hit[0] ? 1u : 0u;
        code = hit[1] ? 2u : code;
        code = hit[2] ? 3u : code;
        code = hit[3] ? 4u : code;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(t4 < T_MIN) != 0, 0)) {
          // some pixel of this wave finishes within these four: redo them with the per-lane stop rule (a lane that
          // does not stop gets the same numbers again)
          bool dead = false;
          float Tc = T;
          code = 0u;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float a = dead ? 0.f : al[k];
            const float tt = Tc * (1.0f - a);
            const bool stop = tt < T_MIN;
            dead = dead || stop;
            a = stop ? 0.f : a;
            code = (a > 0.f) ? (uint32_t)(k + 1) : code;
            w[k] = a * Tc;
            Tc = stop ? Tc : tt;
          }
          t4 = Tc;
          if (dead) pix = f32x2{FAR_PIXEL, FAR_PIXEL};
          // a wave whose 64 pixels have all finished leaves the walk: what it would still read from LDS and issue
          // is taken from the tiles that share the CU with this one
          wave_done = __builtin_amdgcn_ballot_w64(pix.x < 0.5f * FAR_PIXEL) == 0;
        }
        T = t4;
        last_contributor = code ? (uint32_t)(i * BLOCK + 4 * g) + code : last_contributor;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 c0 = s_c0[4 * g + k];
          const float4 c1 = s_c1[4 * g + k];
          const f32x2 w2 = {w[k], w[k]};
          acc2[0] = __builtin_elementwise_fma(f32x2{c0.x, c0.y}, w2, acc2[0]);
          acc2[1] = __builtin_elementwise_fma(f32x2{c0.z, c0.w}, w2, acc2[1]);
          acc2[2] = __builtin_elementwise_fma(f32x2{c1.x, c1.y}, w2, acc2[2]);
          acc2[3] = __builtin_elementwise_fma(f32x2{c1.z, c1.w}, w2, acc2[3]);
          if (AUX) {
            const float4 c2 = s_c2[4 * g + k];
            xacc2 = __builtin_elementwise_fma(f32x2{c2.x, c2.y}, w2, xacc2);
            xacc3 = __builtin_elementwise_fma(f32x2{c2.z, c2.w}, w2, xacc3);   // (.w: a 12-byte read costs two 16-byte ones)
          }
        }
      }
      float* ck = seg_state + (size_t)(slot0 + i * SUBS + h) * (SEG_FLOATS * TILE_PIX) + tid;
      ck[0] = T;
#pragma unroll
      for (int k = 0; k < NCH / 2; ++k) { ck[(1 + 2 * k) * TILE_PIX] = acc2[k].x; ck[(2 + 2 * k) * TILE_PIX] = acc2[k].y; }
      ck[9 * TILE_PIX] = xacc2.x; ck[10 * TILE_PIX] = xacc2.y; ck[11 * TILE_PIX] = xacc3.x;
      walked = i * SUBS + h + 1;
    }
  }
  {
    // work list of the backward pass: the segments in front of the tile's last contributor
    __shared__ uint32_t s_last[BLOCK / 64], s_base;
    uint32_t m = inside ? last_contributor : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((tid & 63) == 0) s_last[tid >> 6] = m;
    __syncthreads();
    const uint32_t tile_last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    const uint32_t nseg = (tile_last + SEG_LEN - 1) / SEG_LEN;
    if (tid == 0) {
      tile_rounds[tile] = (uint32_t)walked;
      s_base = nseg ? atomicAdd(seg_count, nseg) : 0u;
#ifdef BLEND_DBG
      if (tile < 4096) {
        const uint64_t t1 = wall_clock64();
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        uint32_t* d = g_blend_dbg + 8 * tile;
        d[0] = (uint32_t)dbg_t0; d[1] = (uint32_t)(dbg_t0 >> 32); d[2] = (uint32_t)t1; d[3] = (uint32_t)(t1 >> 32);
        d[4] = hw; d[5] = xcc; d[6] = tile_last; d[7] = (uint32_t)(end - start);
      }
#endif
    }
    __syncthreads();
    for (uint32_t k = tid; k < nseg; k += BLOCK) {
      seg_queue[2 * (s_base + k)] = (uint32_t)tile;
      seg_queue[2 * (s_base + k) + 1] = k;
    }
  }
  const float acc[NCH] = {acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y, acc2[2].x, acc2[2].y, acc2[3].x, acc2[3].y};
  const float xacc[3] = {xacc2.x, xacc2.y, xacc3.x};
  if (inside) {
    const size_t P = (size_t)c.H * c.W;
    const size_t pix_i = (size_t)pyi * c.W + pxi;
    final_T[pix_i] = T;
    n_contrib[pix_i] = last_contributor;
    out_color[pix_i] = acc[0] + T * c.bg[0];
    out_color[P + pix_i] = acc[1] + T * c.bg[1];
    out_color[2 * P + pix_i] = acc[2] + T * c.bg[2];
    out_depth[pix_i] = acc[3];
    out_normal[pix_i] = acc[4];
    out_normal[P + pix_i] = acc[5];
    out_normal[2 * P + pix_i] = acc[6];
    out_alpha[pix_i] = 1.0f - T;
    if (out_extra) out_extra[pix_i] = acc[7];
    if (AUX) {
      out_aux[pix_i] = xacc[0] + T * c.bg[0];
      out_aux[P + pix_i] = xacc[1] + T * c.bg[1];
      out_aux[2 * P + pix_i] = xacc[2] + T * c.bg[2];
    }
  }
}

// ---- forward, one SEGMENT at a time ------------------------------------------------------------------------------------
// The kernel above walks a tile's list front to back in one workgroup: its duration is the longest tile's chain (C3: a
// few dozen tiles of the 650 populated ones walk 700-960 entries -- rays through hair never saturate -- and run alone
// for the last 60 of 110 us, scripts/probes/blend_tile_profile.py).  Here the unit of work is one SEG_LEN-entry segment:
// workgroups CLAIM the segments of a tile in order (one atomic), so several workgroups -- the tile's own plus helper
// workgroups started for tiles that walked far in the previous launch (walk hints) -- walk one tile's list on several
// CUs at a time.  What a segment needs from the segments in front of it is one number per pixel, the transmittance in
// front of it, and the arithmetic is arranged so that this number does not depend on WHO computed what, or when:
//   * inside a segment the recurrence runs on the LOCAL transmittance t (1 at the segment's first entry) and the local
//     weights alpha_k t_(k-1); a pixel stops at the first entry with fl(P t_k) < T_MIN, P = the transmittance in front
//     of the segment;
//   * a segment's product t_seg (0 for a pixel that stopped in it) is posted per pixel; P of segment s is the ordered
//     product fl(fl(t_seg(0) t_seg(1)) ...) -- `chain_step` below, used by every reader alike;
//   * the channel sums in front of segment s+1 are acc(s) = fma(P(s), local sums of s, acc(s-1)), formed by ONE
//     workgroup per tile (the one that finds the tile complete) from the posted local sums.
// A workgroup whose segment's predecessors are all posted walks it once, with the stop rule; otherwise it first runs a
// transmittance-only pass (no colours, no stop rule: 0.6 of a walk), posts t_seg, waits for the predecessors' posts --
// their owners are running and wait for nothing themselves -- and walks again with P known.  A tile nobody helps with
// is therefore walked exactly once, as before; results are bit-identical whoever takes part.
#ifndef FWD_HELPERS_N
#define FWD_HELPERS_N 6
#endif
#ifndef FWD_LONG_SEGS_N
#define FWD_LONG_SEGS_N 3
#endif
constexpr int FWD_HELPERS = FWD_HELPERS_N;            // helper workgroups per tile in a launch with walk hints
constexpr uint32_t FWD_LONG_SEGS = FWD_LONG_SEGS_N;     // a tile that walked this many segments lately gets helpers
constexpr uint32_t FWD_HINT_UNIT = 8;     // walk hints are kept in 1/8 segments (they decay by one unit per call)
constexpr int SYNC_CLAIM = 0, SYNC_DONE = 1, SYNC_INV_DEAD = 2, SYNC_RESOLVED = 3;

// What workgroups of different CUs hand each other inside the launch (the per-segment planes, the flags) goes through
// agent-scope relaxed atomics -- write-through stores, cache-bypassing loads -- ordered by the workgroup barrier's
// wait for outstanding memory operations; release / acquire fences write back or invalidate whole caches and cost
// the kernel 4x its duration (0.56 ms against 0.12).
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Before a flag or a count tells other workgroups that this one's stores are there: every wave waits for the
// acknowledgement of its own (write-through) stores, then the workgroup meets.  __syncthreads() alone does not wait
// for outstanding global stores -- a flag written after it can overtake them (seen: a captured step whose loss differed
// from the eager one in the fifth digit once in a few steps).
__device__ __forceinline__ void fwd_publish_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__device__ __forceinline__ void chain_step(float& P, bool& alive, float tseg) {
  // the transmittance in front of the next segment, and whether the pixel is still unfinished there
  const float next = P * tseg;
  if (alive) {
    if (next < T_MIN) alive = false;
    else P = next;
  }
}

// the per-pixel state in front of a segment, as the workgroup that adds a tile up carries it from segment to segment
struct FwdSums {
  float P, Tf, acc[NCH], xacc[3];
  uint32_t last;
  bool alive;
};

__device__ __forceinline__ void fold_segment(FwdSums& r, int q, float T_after, const float* loc /*[NCH]*/,
                                             const float* xloc /*[3]*/, float tseg, uint32_t last_local) {
  if (r.alive) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) r.acc[k] = __builtin_fmaf(r.P, loc[k], r.acc[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) r.xacc[k] = __builtin_fmaf(r.P, xloc[k], r.xacc[k]);
    r.Tf = T_after;
    r.last = last_local ? (uint32_t)(q * SEG_LEN) + last_local : r.last;
  }
  chain_step(r.P, r.alive, tseg);
}

template <bool AUX>
#ifdef FWD_CAP128
__global__ void __launch_bounds__(BLOCK, 4)
#else
// (three workgroups per CU at ~140 registers.  Held to 128 -- four per CU, every tile's own workgroup resident from the
// start -- the inner loop waits for its LDS reads one by one: 112 us against 104)
__global__ void __launch_bounds__(BLOCK)
#endif
blend_forward_claim_kernel(Camera c, const int32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                           const float* __restrict__ rec2d, uint32_t* __restrict__ n_contrib,
                           float* __restrict__ final_T, float* __restrict__ out_color,
                           float* __restrict__ out_depth, float* __restrict__ out_normal,
                           float* __restrict__ out_alpha, float* __restrict__ out_extra,
                           const float* __restrict__ aux_colors, float* __restrict__ out_aux,
                           uint32_t* __restrict__ seg_queue, uint32_t* __restrict__ seg_count,
                           float* seg_state, uint32_t* __restrict__ tile_rounds, uint32_t* tile_sync,
                           uint32_t* seg_flag, uint32_t* walk_hints, uint32_t* stalls, int ntiles, int share_all) {
  // two sets of record arrays: an unshared tile's workgroup stages segment s + 1 into the other set while slower waves
  // still read segment s -- ONE workgroup barrier per segment, and "is any pixel unfinished" rides on it (a flag per
  // wave instead of __syncthreads_or: the voting barriers cost the short tiles 10 % of their walk)
  __shared__ float4 s_geo[2 * SEG_LEN];            // x y A' C'   (one LDS array per read of the inner loop, see above)
  __shared__ float2 s_bo[2 * SEG_LEN];             // B' opacity
  __shared__ float4 s_c0[2 * SEG_LEN], s_c1[2 * SEG_LEN];  // r g b depth | nx ny nz extra
  __shared__ float4 s_c2[AUX ? 2 * SEG_LEN : 1];   // aux r g b -
  __shared__ uint32_t s_word[2];
  __shared__ uint32_t s_alive[2][BLOCK / 64];
  const int tid = threadIdx.x;
  // The tiles' own workgroups come first in block order, the helpers behind them: at 128 registers all 1,024 of C3's
  // tiles are resident at once, a third of them are empty and leave within a microsecond, and the helpers take those
  // slots (helpers in FRONT delay every tile's start by the dispatch of 3 x tiles blocks that mostly leave: +6 us)
#ifdef FWD_HELPERS_FIRST
  const int nhelp = (int)gridDim.x - ntiles;
  const bool helper = (int)blockIdx.x < nhelp;
  const int tile = helper ? (int)blockIdx.x / FWD_HELPERS : (int)blockIdx.x - nhelp;
#else
  const bool helper = (int)blockIdx.x >= ntiles;
  const int tile = helper ? ((int)blockIdx.x - ntiles) / FWD_HELPERS : (int)blockIdx.x;
#endif
  // (any content of the hint will do: the tile's own workgroup and its helpers read the same word -- nobody writes it
  // before the tile is complete -- and helpers only take work that is there)
  // share_all (tests): every tile long enough is shared, hints or not
  // ONE thread reads the hint for the workgroup: the word is rewritten when the tile is complete -- by whoever adds it up,
  // possibly while a late helper (or the tile's own workgroup, at more tiles than the chip holds) is starting -- and
  // threads that each read it for themselves could disagree about `direct`: some would leave, the others wait for them at
  // the next barrier for ever (seen as a launch that never ends, once the helpers were made to start early)
  if (tid == 0) s_word[0] = walk_hints != nullptr ? ld_agent(&walk_hints[tile]) : 0u;
  __syncthreads();
  const uint32_t hint = s_word[0];
  __syncthreads();
  const bool helped = share_all != 0 || hint >= FWD_LONG_SEGS * FWD_HINT_UNIT;
  const int start = ranges[2 * tile], end = ranges[2 * tile + 1];
  const int nsegs = (end - start + SEG_LEN - 1) / SEG_LEN;
  // direct: the tile has no helpers -- its workgroup walks the segments in order without claims, flags or a second look
  // at what it stored (the same arithmetic, segment by segment, as the tiles that are shared)
  const bool direct = !(helped && nsegs >= (int)FWD_LONG_SEGS);
  uint32_t* sync = tile_sync + 8 * (size_t)tile;
  if (helper && direct) return;
  if (direct && walk_hints != nullptr && tid == 0) {
    // This workgroup writes the new hint when it is done; a helper that starts after that reads the NEW value and may
    // take the tile for a shared one: it then finds every segment "finished" and the tile added up already
    atomicMax(&sync[SYNC_INV_DEAD], 0xFFFFFFFFu);
    atomicExch(&sync[SYNC_RESOLVED], 1u);
  }
  const int tx = tile % c.grid_x, ty = tile / c.grid_x;
  const int pxi = tx * TILE_X + (tid & 15), pyi = ty * TILE_Y + (tid >> 4);
  const bool inside = pxi < c.W && pyi < c.H;
  const int slot0 = start / SEG_LEN + tile;
  constexpr size_t SLOT = (size_t)SEG_FLOATS * TILE_PIX;
  const f32x2 pix_in = inside ? f32x2{(float)pxi, (float)pyi} : f32x2{FAR_PIXEL, FAR_PIXEL};
#ifdef BLEND_DBG
  const uint64_t dbg_t0 = wall_clock64();
#endif

  // the records of the segment this workgroup will probably take next are fetched while it walks the current one
  float4 nrec0, nrec1, nrec2, nrec3;
  float nauxb = 0.f;
  int fetched = -1;
  auto fetch = [&](int seg) {
    fetched = seg;
    nrec0 = make_float4(0.f, 0.f, 0.f, 0.f); nrec1 = nrec0; nrec2 = nrec0; nrec3 = nrec0; nauxb = 0.f;
    const int pos = start + seg * SEG_LEN + tid;
    if (tid < SEG_LEN && seg < nsegs && pos < end) {
      const uint32_t gid = point_list[pos];
      const float4* r = reinterpret_cast<const float4*>(rec2d + (size_t)gid * REC_FLOATS);
      const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
      nrec0 = make_float4(r0.x, r0.y, (-0.5f * LOG2E) * r0.z, (-0.5f * LOG2E) * r1.x);
      nrec1 = make_float4(-LOG2E * r0.w, r1.y, r1.z, r1.w);
      nrec2 = r2;
      nrec3 = make_float4(r3.x, r3.y, 0.f, 0.f);
      if (AUX) {
        nrec3.z = aux_colors[3 * (size_t)gid]; nrec3.w = aux_colors[3 * (size_t)gid + 1];
        nauxb = aux_colors[3 * (size_t)gid + 2];
      }
    }
  };
  // what this workgroup knows of the chain: P / alive in front of segment `known`
  float P = 1.0f;
  bool alive = inside;
  int known = 0;
  FwdSums sums;                                        // (direct tiles, and whoever adds a shared tile up)
  sums.P = 1.0f; sums.Tf = 1.0f; sums.last = 0u; sums.alive = inside;
#pragma unroll
  for (int k = 0; k < NCH; ++k) sums.acc[k] = 0.f;
  sums.xacc[0] = sums.xacc[1] = sums.xacc[2] = 0.f;
  int walked = 0;
  if (!helper) fetch(0);

  for (int next_direct = 0;; ++next_direct) {
    int seg = next_direct;
    const int lds0 = direct ? (seg & 1) * SEG_LEN : 0;
    if (!direct) __syncthreads();                     // (s_word, the record arrays: the previous round is done with them)
    if (!direct) {
      if (tid == 0) {
        s_word[0] = atomicAdd(&sync[SYNC_CLAIM], 1u);
        s_word[1] = ~ld_agent(&sync[SYNC_INV_DEAD]);
      }
      __syncthreads();
      seg = (int)s_word[0];
    }
    const int slot = slot0 + seg;
#ifdef BLEND_DBG2
    if (!direct && tid == 0 && slot < (1 << 16)) g_seg_owner[slot] = (helper ? 0x80000000u : 0u) | ((uint32_t)blockIdx.x + 1u);
#endif
    if (seg >= nsegs || (!direct && (uint32_t)seg >= s_word[1])) {
      // nothing left to take: the list ends here (or the tile is known to be finished in front of this segment).  A
      // claimed segment is always posted -- a workgroup that took a later one before the tile was known to be
      // finished may be waiting for it (every pixel is finished there: it will not look at the numbers)
      if (!direct && tid == 0) {
        if (seg >= nsegs) atomicMax(&sync[SYNC_INV_DEAD], ~(uint32_t)nsegs);
        else st_agent(&seg_flag[slot], 1u);
      }
      walked = nsegs;
      break;
    }
    if (fetched != seg) fetch(seg);
    if (tid < SEG_LEN) {
      s_geo[lds0 + tid] = nrec0; s_bo[lds0 + tid] = make_float2(nrec1.x, nrec1.y);
      s_c0[lds0 + tid] = make_float4(nrec1.z, nrec1.w, nrec2.x, nrec2.y);
      s_c1[lds0 + tid] = make_float4(nrec2.z, nrec2.w, nrec3.x, nrec3.y);
      if (AUX) s_c2[lds0 + tid] = make_float4(nrec3.z, nrec3.w, nauxb, 0.f);
    }
    bool all_posted = true;
    if (direct) {
      const uint32_t wave_alive = __builtin_amdgcn_ballot_w64(alive) != 0 ? 1u : 0u;   // (all lanes vote)
      if ((tid & 63) == 0) s_alive[seg & 1][tid >> 6] = wave_alive;
      __syncthreads();                                 // (the records are staged)
    } else {
      // are the segments in front of this one posted (those this workgroup has not folded into P yet)?
      bool posted = true;
      for (int q = known + tid; q < seg; q += BLOCK) posted = posted && ld_agent(&seg_flag[slot0 + q]) != 0u;
      all_posted = __syncthreads_and(posted) != 0;     // (also: the records are staged)
    }
    fetch(seg + 1);
    const int groups = (min(SEG_LEN, end - start - seg * SEG_LEN) + 3) >> 2;
    float* st = seg_state + (size_t)slot * SLOT + tid;

    if (!all_posted) {
      // ---- transmittance-only pass: t_seg of every pixel, as if none stopped ---------------------------------------
      float t = 1.0f;
      if (__builtin_amdgcn_ballot_w64(inside) != 0) {
        for (int g = 0; g < groups; ++g) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 a = s_geo[lds0 + 4 * g + k];
            const float2 b = s_bo[lds0 + 4 * g + k];
            const f32x2 d = f32x2{a.x, a.y} - pix_in;
            const f32x2 u = (d * f32x2{a.z, a.w}) * d;
            const float p2 = __builtin_fmaf(b.x, d.x * d.y, u.x + u.y);
            const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(p2));
            const bool hit = !(p2 > 0.0f) && !(alpha < ALPHA_MIN);
            t = t * (1.0f - (hit ? alpha : 0.f));
          }
        }
      }
      st_agent(&st[12 * TILE_PIX], t);
      fwd_publish_barrier();
      if (tid == 0) {
        st_agent(&seg_flag[slot], 1u);
        // the predecessors' owners are running and wait for nothing: their posts come (bounded all the same)
        for (int q = known; q < seg; ++q) {
          uint32_t spins = 0;
          while (ld_agent(&seg_flag[slot0 + q]) == 0u) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 15)) {                      // (~30 ms) gave up: the result is wrong, say so (sort_stalls())
              if (stalls) atomicAdd(stalls, 1u);
#ifdef BLEND_DBG2
              {
                const uint32_t r = atomicAdd(&g_stall_log[0], 1u);
                if (r < 512u && g_stall_host != nullptr) {
                  uint32_t* d = g_stall_host + 8 + 8 * r;
                  __hip_atomic_store(&g_stall_host[0], r + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                  d[0] = (uint32_t)tile; d[1] = (uint32_t)seg; d[2] = (uint32_t)q;
                  d[3] = (slot0 + q) < (1 << 16) ? g_seg_owner[slot0 + q] : 0u;
                  d[4] = (helper ? 0x80000000u : 0u) | ((uint32_t)blockIdx.x + 1u);
                  d[5] = (uint32_t)nsegs; d[6] = ld_agent(&sync[SYNC_CLAIM]); d[7] = ~ld_agent(&sync[SYNC_INV_DEAD]);
                }
              }
#endif
              break;
            }
          }
        }
      }
      __syncthreads();
    }
    if (!direct) {
      for (int q = known; q < seg; ++q)
        chain_step(P, alive, ld_agent(seg_state + (size_t)(slot0 + q) * SLOT + 12 * TILE_PIX + tid));
      known = seg;
    }
    const bool any_alive = direct ? (s_alive[seg & 1][0] | s_alive[seg & 1][1] | s_alive[seg & 1][2] | s_alive[seg & 1][3]) != 0u
                                  : __syncthreads_or(alive) != 0;
    if (!any_alive) {
      // every pixel finished in front of this segment: the tile's walk ends here (posted all the same, see above)
      if (!direct && tid == 0) {
        atomicMax(&sync[SYNC_INV_DEAD], ~(uint32_t)seg);
        if (all_posted) st_agent(&seg_flag[slot], 1u);
      }
      walked = seg;
      break;
    }

    // ---- the walk proper: local transmittance, local sums, the stop rule against P -----------------------------------
    f32x2 pix = alive ? pix_in : f32x2{FAR_PIXEL, FAR_PIXEL};
    const float Pe = alive ? P : 1.0f;                 // (a finished pixel never trips the stop test)
    bool wave_done = __builtin_amdgcn_ballot_w64(alive) == 0;
    float T = 1.0f;
    uint32_t last_local = 0;
    f32x2 acc2[NCH / 2];
#pragma unroll
    for (int k = 0; k < NCH / 2; ++k) acc2[k] = f32x2{0.f, 0.f};
    f32x2 xacc2 = {0.f, 0.f}, xacc3 = {0.f, 0.f};
    for (int g = 0; g < groups && !wave_done; ++g) {
      float al[4], w[4];
      bool hit[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 a = s_geo[lds0 + 4 * g + k];
        const float2 b = s_bo[lds0 + 4 * g + k];
        const f32x2 d = f32x2{a.x, a.y} - pix;
        const f32x2 u = (d * f32x2{a.z, a.w}) * d;
        const float p2 = __builtin_fmaf(b.x, d.x * d.y, u.x + u.y);
        const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(p2));
        hit[k] = !(p2 > 0.0f) && !(alpha < ALPHA_MIN);
        al[k] = hit[k] ? alpha : 0.f;
      }
      const float t1 = T * (1.0f - al[0]);
      const float t2 = t1 * (1.0f - al[1]);
      const float t3 = t2 * (1.0f - al[2]);
      float t4 = t3 * (1.0f - al[3]);
      w[0] = al[0] * T; w[1] = al[1] * t1; w[2] = al[2] * t2; w[3] = al[3] * t3;
      uint32_t code = hit[0] ? 1u : 0u;
      code = hit[1] ? 2u : code;
      code = hit[2] ? 3u : code;
      code = hit[3] ? 4u : code;
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(Pe * t4 < T_MIN) != 0, 0)) {
        // some pixel of this wave finishes within these four: redo them with the per-lane stop rule (a lane that
        // does not stop gets the same numbers again)
        bool dead = false;
        float Tc = T;
        code = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float a = dead ? 0.f : al[k];
          const float tt = Tc * (1.0f - a);
          const bool stop = Pe * tt < T_MIN;
          dead = dead || stop;
          a = stop ? 0.f : a;
          code = (a > 0.f) ? (uint32_t)(k + 1) : code;
          w[k] = a * Tc;
          Tc = stop ? Tc : tt;
        }
        t4 = Tc;
        if (dead) pix = f32x2{FAR_PIXEL, FAR_PIXEL};
        wave_done = __builtin_amdgcn_ballot_w64(pix.x < 0.5f * FAR_PIXEL) == 0;
      }
      T = t4;
      last_local = code ? (uint32_t)(4 * g) + code : last_local;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 c0 = s_c0[lds0 + 4 * g + k];
        const float4 c1 = s_c1[lds0 + 4 * g + k];
        const f32x2 w2 = {w[k], w[k]};
        acc2[0] = __builtin_elementwise_fma(f32x2{c0.x, c0.y}, w2, acc2[0]);
        acc2[1] = __builtin_elementwise_fma(f32x2{c0.z, c0.w}, w2, acc2[1]);
        acc2[2] = __builtin_elementwise_fma(f32x2{c1.x, c1.y}, w2, acc2[2]);
        acc2[3] = __builtin_elementwise_fma(f32x2{c1.z, c1.w}, w2, acc2[3]);
        if (AUX) {
          const float4 c2 = s_c2[lds0 + 4 * g + k];
          xacc2 = __builtin_elementwise_fma(f32x2{c2.x, c2.y}, w2, xacc2);
          xacc3 = __builtin_elementwise_fma(f32x2{c2.z, c2.w}, w2, xacc3);
        }
      }
    }
    // what the segment leaves behind: the transmittance after it (where the pixel stopped, if it did), the LOCAL sums,
    // t_seg (0 for a pixel that stopped here or before), the last contributor
    const bool stopped = alive && !(pix.x < 0.5f * FAR_PIXEL);
    const float tseg = (alive && !stopped) ? T : 0.0f;
    const float loc[NCH] = {acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y, acc2[2].x, acc2[2].y, acc2[3].x, acc2[3].y};
    const float xloc[3] = {xacc2.x, xacc2.y, xacc3.x};
    if (direct) {
      // the tile is this workgroup's alone: the running sums stay in registers, the state behind the segment is stored
      // in the form the backward pass reads
      fold_segment(sums, seg, P * T, loc, xloc, tseg, alive ? last_local : 0u);
      P = sums.P; alive = sums.alive;
      st[0] = sums.Tf;
#pragma unroll
      for (int k = 0; k < NCH; ++k) st[(1 + k) * TILE_PIX] = sums.acc[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) st[(9 + k) * TILE_PIX] = sums.xacc[k];
      continue;
    }
    st_agent(&st[0], P * T);
#pragma unroll
    for (int k = 0; k < NCH; ++k) st_agent(&st[(1 + k) * TILE_PIX], loc[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) st_agent(&st[(9 + k) * TILE_PIX], xloc[k]);
    if (all_posted) st_agent(&st[12 * TILE_PIX], tseg);   // (else: posted by the transmittance pass -- same verdict for the chain)
    st_agent(&st[13 * TILE_PIX], __uint_as_float(alive ? last_local : 0u));
    chain_step(P, alive, tseg);
    known = seg + 1;
    fwd_publish_barrier();
    if (tid == 0) {
      if (all_posted) st_agent(&seg_flag[slot], 1u);
      atomicAdd(&sync[SYNC_DONE], 1u);
    }
  }

  if (!direct) {
    // ---- whoever finds the tile complete (every segment in front of the first finished one walked) adds it up -----
    __syncthreads();
    if (tid == 0) {
      // (this workgroup's own count / end-of-walk updates have been performed before it looks at the others': of the
      // two workgroups that make the tile complete, at least one then sees both updates)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t done = atomicAdd(&sync[SYNC_DONE], 0u);
      const uint32_t dead = ~atomicMax(&sync[SYNC_INV_DEAD], 0u);
      s_word[0] = (done == dead && atomicCAS(&sync[SYNC_RESOLVED], 0u, 1u) == 0u) ? 1u : 0u;
      s_word[1] = dead;
    }
    __syncthreads();
    if (s_word[0] == 0u) return;
    walked = (int)s_word[1];
    for (int q = 0; q < walked; ++q) {
      float* sq = seg_state + (size_t)(slot0 + q) * SLOT + tid;
      float v[SEG_FLOATS];
#pragma unroll
      for (int k = 0; k < SEG_FLOATS; ++k) v[k] = ld_agent(sq + k * TILE_PIX);
      fold_segment(sums, q, v[0], v + 1, v + 9, v[12], __float_as_uint(v[13]));
      // the state after segment q as the backward pass reads it: T, the sums so far
      sq[0] = sums.Tf;
#pragma unroll
      for (int k = 0; k < NCH; ++k) sq[(1 + k) * TILE_PIX] = sums.acc[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) sq[(9 + k) * TILE_PIX] = sums.xacc[k];
    }
  }

  {
    __shared__ uint32_t s_last[BLOCK / 64], s_base;
    uint32_t m = inside ? sums.last : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((tid & 63) == 0) s_last[tid >> 6] = m;
    __syncthreads();
    const uint32_t tile_last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    const uint32_t nseg = (tile_last + SEG_LEN - 1) / SEG_LEN;
    if (tid == 0) {
      tile_rounds[tile] = (uint32_t)walked;
      // the hint follows the walk up at once and down slowly: steps of a training run alternate between views, and a
      // tile that walked far in one of the last FWD_HINT_UNIT * (length - 2) calls is worth its helpers' first look
      if (walk_hints) st_agent(&walk_hints[tile], max((uint32_t)walked * FWD_HINT_UNIT, hint > 0u && hint < (1u << 20) ? hint - 1u : 0u));
      s_base = nseg ? atomicAdd(seg_count, nseg) : 0u;
#ifdef BLEND_DBG
      if (tile < 4096) {
        const uint64_t t1 = wall_clock64();
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        uint32_t* d = g_blend_dbg + 8 * tile;
        d[0] = (uint32_t)dbg_t0; d[1] = (uint32_t)(dbg_t0 >> 32); d[2] = (uint32_t)t1; d[3] = (uint32_t)(t1 >> 32);
        d[4] = hw; d[5] = xcc; d[6] = tile_last; d[7] = (uint32_t)(end - start);
      }
#endif
    }
    __syncthreads();
    for (uint32_t k = tid; k < nseg; k += BLOCK) {
      seg_queue[2 * (s_base + k)] = (uint32_t)tile;
      seg_queue[2 * (s_base + k) + 1] = k;
    }
  }
  if (inside) {
    const float Tf = sums.Tf;
    const size_t Pn = (size_t)c.H * c.W;
    const size_t pix_i = (size_t)pyi * c.W + pxi;
    final_T[pix_i] = Tf;
    n_contrib[pix_i] = sums.last;
    out_color[pix_i] = sums.acc[0] + Tf * c.bg[0];
    out_color[Pn + pix_i] = sums.acc[1] + Tf * c.bg[1];
    out_color[2 * Pn + pix_i] = sums.acc[2] + Tf * c.bg[2];
    out_depth[pix_i] = sums.acc[3];
    out_normal[pix_i] = sums.acc[4];
    out_normal[Pn + pix_i] = sums.acc[5];
    out_normal[2 * Pn + pix_i] = sums.acc[6];
    out_alpha[pix_i] = 1.0f - Tf;
    if (out_extra) out_extra[pix_i] = sums.acc[7];
    if (AUX) {
      out_aux[pix_i] = sums.xacc[0] + Tf * c.bg[0];
      out_aux[Pn + pix_i] = sums.xacc[1] + Tf * c.bg[1];
      out_aux[2 * Pn + pix_i] = sums.xacc[2] + Tf * c.bg[2];
    }
  }
}

// ---- backward ------------------------------------------------------------------------------------------
// Per tile the back-to-front recurrence over the depth-sorted list is inherently sequential per pixel,
// and the kernel's duration is the critical path of the longest tile.  The per-Gaussian step is therefore
// kept minimal: phase A (one pixel per lane) only advances the recurrence and emits two numbers per
// (Gaussian, pixel) -- w = alpha*T and t = G*dL/dalpha -- into LDS; the 14 gradient components of a
// Gaussian are sums over the tile's 256 pixels of w*dL/dpixel[ch] and of t times the pixel-coordinate
// moments (1, x, y, x^2, xy, y^2), i.e. two small GEMMs [16 Gaussians x 256 pixels] x [256 pixels x 16],
// which phase B runs on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) instead of 14 x 6
// cross-lane DPP adds per Gaussian per wave.  One wave per (matrix, half of every pixel quarter); the per-pixel
// feature fragments (B operands) are fixed for the whole tile and live in 32 VGPRs.
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BB = 16;          // Gaussians per backward batch (one MFMA row block): 40 KB of LDS per workgroup, so that
                                // three to four workgroups fit a CU and one's MFMA phase overlaps another's VALU phase
constexpr int SB = 32;          // records staged per round (two batches)
constexpr int WROW = 68;        // floats per (pixel-quarter, Gaussian) row of the w / t matrices (64 + 4 pad)

// FULL: depth / normal / extra channels carry gradient too; else rgb only.
// AUX (rgb only) -- how much of the auxiliary colour set's image is differentiated in THIS launch:
//   1: all of it -- alpha, T and w are shared, the aux image adds its own dL/dalpha recurrence (Q_x) and one more small
//      GEMM; its gradients (aux colours, and the screen-space mean's share, which the reference's second rasterizer
//      call sends to means2D only) leave in the row's depth / normal / extra slots, which an rgb-only pass does not use;
//   2: the aux COLOURS' gradient only: it is sum_p w dL/d(aux pixel), i.e. three more feature columns of the w product,
//      which has ten idle ones -- free.  The screen-space mean's share then comes from an XONLY launch beside this one.
// XONLY: a pass over `color_override` colours that only produces the screen-space mean's gradient (rows: dx, dy): no w
//   product at all, half the matrix work of a colour pass.
template <bool FULL, int AUX, bool XONLY>
__device__ __forceinline__ void
blend_backward_segment(const int tile, const int seg, const Camera& c, const int32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const uint32_t* __restrict__ slot_list, const float* __restrict__ rec2d, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ final_T, const float* __restrict__ dL_dcolor,
                      const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dnormal,
                      const float* __restrict__ dL_dalpha_img, const float* __restrict__ dL_dextra,
                      float* __restrict__ inst_grad, const float* __restrict__ color_override /*[N,3] or null*/,
                      const float* __restrict__ aux_colors /*[N,3], AUX*/, const float* __restrict__ dL_daux /*[3,H,W], AUX*/,
                      const float* __restrict__ seg_state, const uint32_t* __restrict__ tile_rounds,
                      uint8_t* __restrict__ row_flag) {
  static_assert(!(FULL && AUX != 0), "the auxiliary gradients use the row slots of the depth / normal / extra gradients");
  static_assert(!(XONLY && (FULL || AUX != 0)), "XONLY is a pass of its own");
  constexpr bool AUXW = AUX != 0, AUXX = AUX == 1;
  constexpr int NMAT = XONLY ? 1 : (AUXX ? 3 : 2);      // w x dL/dpixel, t x moments, (t_aux x moments)
  constexpr int MW = 0, MT = XONLY ? 0 : 1, MX = 2;
  // Records are staged 32 at a time -- two batches -- so that their two dependent global loads are issued a full two
  // batches (~3 us) before they are needed
  // staged records, one LDS array per read of phase A (whole ds_read_b128 / ds_read_b64, see blend_forward_kernel)
  __shared__ float4 s_geo[SB];                  // x y A' C'   (conic in the log2 domain, as in the forward pass)
  __shared__ float2 s_bo[SB];                   // B' opacity
  __shared__ float4 s_c0[SB];                   // r g b depth
  __shared__ float4 s_c1[FULL ? SB : 1];        // nx ny nz extra
  __shared__ float4 s_con[SB];                  // conic A B C and the instance's gradient row (row assembly only)
  __shared__ float4 s_axc[AUXX ? SB : 1];              // auxiliary colours of the staged records
  __shared__ __align__(16) float s_W[XONLY ? 4 : 4 * BB * WROW];   // [pixel quarter kk][gaussian][64 pixels + pad]
  __shared__ __align__(16) float s_T[4 * BB * WROW];
  __shared__ __align__(16) float s_X[AUXX ? 4 * BB * WROW : 4];
  // [gaussian][matrix][wave = 16 steps of every quarter][feature < 8], one float of padding per Gaussian: the row
  // assembly reads one Gaussian per lane, and a stride of 64 or 96 floats put all sixteen lanes on one bank
  // (SQ_LDS_BANK_CONFLICT was 18 % of the kernel's LDS cycles, profiles/r02_pmc_lds_c3.csv)
  constexpr int RES_STRIDE = NMAT * 32 + 1;
  __shared__ float s_res_flat[BB * RES_STRIDE];
  auto s_res = [&](int g, int m, int w, int f) -> float& { return s_res_flat[g * RES_STRIDE + (m * 4 + w) * 8 + f]; };
  int* s_max = reinterpret_cast<int*>(&s_res_flat[0]);   // (used once, before the first batch)
  // One SEGMENT (SEG_LEN list entries of one tile) per call, not a whole tile: the forward pass left the per-pixel
  // state after every segment (seg_state), so a segment's back-to-front walk starts from the state behind it instead
  // of waiting for the walk over everything behind it.  The kernel's duration used to be the longest tile's chain (C3:
  // 1279 entries = 80 batches); now no chain is longer than SEG_LEN / 16 batches and the segments fill the chip evenly.
  const int list_start = ranges[2 * tile], list_end = ranges[2 * tile + 1];
  const int slot0 = list_start / SEG_LEN + tile;
  const int slot = slot0 + seg;
  const int tx = tile % c.grid_x, ty = tile / c.grid_x;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int pxi = tx * TILE_X + (tid & 15), pyi = ty * TILE_Y + (tid >> 4);
  const bool inside = pxi < c.W && pyi < c.H;
  const float pxf = (float)pxi, pyf = (float)pyi;
  const int start = list_start + seg * SEG_LEN, end = min(list_end, start + SEG_LEN);
  const size_t P = (size_t)c.H * c.W;
  const size_t pix = (size_t)pyi * c.W + pxi;

  // (positions are relative to the segment from here on: <= 0 means the pixel finished in front of it)
  const int last_contributor = inside ? (int)n_contrib[pix] - seg * SEG_LEN : 0;
  const float T_final = inside ? final_T[pix] : 0.f;
  float T = T_final;
  float dpix[NCH];
  dpix[0] = (inside && dL_dcolor) ? dL_dcolor[pix] : 0.f;
  dpix[1] = (inside && dL_dcolor) ? dL_dcolor[P + pix] : 0.f;
  dpix[2] = (inside && dL_dcolor) ? dL_dcolor[2 * P + pix] : 0.f;
  dpix[3] = (FULL && inside && dL_ddepth) ? dL_ddepth[pix] : 0.f;
  dpix[4] = (FULL && inside && dL_dnormal) ? dL_dnormal[pix] : 0.f;
  dpix[5] = (FULL && inside && dL_dnormal) ? dL_dnormal[P + pix] : 0.f;
  dpix[6] = (FULL && inside && dL_dnormal) ? dL_dnormal[2 * P + pix] : 0.f;
  dpix[7] = (FULL && inside && dL_dextra) ? dL_dextra[pix] : 0.f;
  float dpx[3] = {0.f, 0.f, 0.f};                      // dL/d(aux image) at this pixel
  if (AUXW && inside) { dpx[0] = dL_daux[pix]; dpx[1] = dL_daux[P + pix]; dpx[2] = dL_daux[2 * P + pix]; }
  const float dalpha_img = (inside && dL_dalpha_img) ? dL_dalpha_img[pix] : 0.f;
  bool live = dalpha_img != 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) live = live || dpix[k] != 0.f;
  if (AUXW) live = live || dpx[0] != 0.f || dpx[1] != 0.f || dpx[2] != 0.f;
  // only the first max(last_contributor) Gaussians of the list reached any pixel of this tile; a pixel whose
  // incoming gradient is exactly zero contributes nothing (a masked loss leaves most tiles of an image untouched)
  int m = live ? last_contributor : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if (lane == 0) s_max[wave] = m;
  __syncthreads();
  const int n = min(end - start, max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
  // entries behind the last contributor of every pixel get no gradient row: row_flag says which rows exist
  if (n <= 0) return;

  // d(T_final)/d(alpha_i) = -T_final/(1-alpha_i); T_final enters image (+bg) and alpha image (-1); the aux image has
  // the same background term and no alpha image
  const float tail = (c.bg[0] * dpix[0] + c.bg[1] * dpix[1] + c.bg[2] * dpix[2]) - dalpha_img;
  const float tail_x = c.bg[0] * dpx[0] + c.bg[1] * dpx[1] + c.bg[2] * dpx[2];

  // ---- B-operand fragments of phase B (fixed per tile).  MFMA step s of pixel quarter kk = lane>>4 covers pixel
  // p = 64 kk + s; feature f = lane & 15.  Wave q runs steps 16q .. 16q+15 of EVERY matrix product: the w matrix times
  // dL/dpixel[f] (main channels, then the three aux channels), the t (and t_aux) matrix times the moments
  // (1, lx, ly, lx^2, lx ly, ly^2) of the pixel's in-tile coordinates.
  float bfragW[16], bfragM[16];
  {
    float* s_F = XONLY ? s_T : s_W;         // staging: dL/dpixel of all 256 pixels, [pixel][12]
    constexpr int NF = 12;
#pragma unroll
    for (int k = 0; k < NCH; ++k) s_F[tid * NF + k] = dpix[k];
    if (AUXW) {                             // (rgb-only pass: the aux channels sit behind the three main ones)
      s_F[tid * NF + 3] = dpx[0]; s_F[tid * NF + 4] = dpx[1]; s_F[tid * NF + 5] = dpx[2];
    }
    __syncthreads();
    const int kk = lane >> 4, f = lane & 15;
    constexpr int NFW = AUXW ? 6 : NCH;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {
      const int p = 64 * kk + 16 * wave + s_;
      bfragW[s_] = (!XONLY && f < NFW) ? s_F[p * NF + f] : 0.f;
      const float lx = (float)(p & 15), ly = (float)(p >> 4);
      bfragM[s_] = f == 0 ? 1.f : f == 1 ? lx : f == 2 ? ly : f == 3 ? lx * lx : f == 4 ? lx * ly : f == 5 ? ly * ly : 0.f;
    }
  }

  // Serial per-pixel state of the back-to-front recurrence, reduced to two scalars: T (transmittance in front
  // of the current Gaussian) and Q = sum over the Gaussians behind of w_i * <c_i, dL/dpixel>.  With
  // cd_j = <c_j, dL/dpixel>:   dL/dalpha_j = T_j cd_j - (Q_j + T_final*tail) / (1 - alpha_j)
  // (the "colour accumulated behind" of the published kernel is Q / T_{j+1} contracted with dL/dpixel).
  // The aux image runs the same recurrence with its own colours and upstream gradient: (Q_x, tail_x).
  float Q = 0.f, Qx = 0.f;
  const float tf_tail = T_final * tail, tf_tail_x = T_final * tail_x;
  if (live && last_contributor > n) {
    // the pixel's walk began behind this segment: T in front of the segment's last entry + 1 is the forward pass's own
    // value, and Q there is <dL/dpixel, colour accumulated behind> = <dL/dpixel, final accumulators - accumulators
    // after this segment> (the last segment the tile walked holds the final ones)
    const float* ck = seg_state + (size_t)slot * (SEG_FLOATS * TILE_PIX) + tid;
    const float* cf = seg_state + (size_t)(slot0 + (int)tile_rounds[tile] - 1) * (SEG_FLOATS * TILE_PIX) + tid;
    T = ck[0];
    const int ch0 = color_override ? 9 : 1;
#pragma unroll
    for (int k = 0; k < (FULL ? NCH : 3); ++k) Q += dpix[k] * (cf[(ch0 + k) * TILE_PIX] - ck[(ch0 + k) * TILE_PIX]);
    if (AUXX) {
#pragma unroll
      for (int k = 0; k < 3; ++k) Qx += dpx[k] * (cf[(9 + k) * TILE_PIX] - ck[(9 + k) * TILE_PIX]);
    }
  }
  Q += tf_tail; Qx += tf_tail_x;               // (the recurrences below carry Q + T_final * tail)
  const f32x2 pix2 = {pxf, pyf};

  const float tile_x0 = (float)(tx * TILE_X), tile_y0 = (float)(ty * TILE_Y);
  const int rounds = (n + SB - 1) / SB;
  // the records of round i+1 are fetched (two dependent global loads) while round i is being processed
  float4 nrec0 = make_float4(0.f, 0.f, 0.f, 0.f), nrec1 = nrec0, nrec2 = nrec0, nrec3 = nrec0, naxc = nrec0;
  auto fetch = [&](int pos) {
    const uint32_t gid = point_list[start + pos];
    const float4* r = reinterpret_cast<const float4*>(rec2d + (size_t)gid * REC_FLOATS);
    nrec0 = r[0]; nrec1 = r[1]; nrec2 = r[2]; nrec3 = r[3];
    nrec3.z = __uint_as_float(slot_list[start + pos]);   // the instance's gradient row
    if (color_override) {
      nrec1.z = color_override[3 * (size_t)gid]; nrec1.w = color_override[3 * (size_t)gid + 1];
      nrec2.x = color_override[3 * (size_t)gid + 2];
    }
    if (AUXX) naxc = make_float4(aux_colors[3 * (size_t)gid], aux_colors[3 * (size_t)gid + 1], aux_colors[3 * (size_t)gid + 2], 0.f);
  };
  if (tid < min(SB, n)) fetch((n - 1) - tid);
  for (int si = 0; si < rounds; ++si) {
    __syncthreads();                          // previous round fully consumed (s_rec, s_W/s_T, s_res, s_F)
    const int sbase = n - 1 - si * SB;        // list index of round element 0 (walks backwards)
    const int scnt = min(SB, n - si * SB);
    if (tid < scnt) {
      s_geo[tid] = make_float4(nrec0.x, nrec0.y, (-0.5f * LOG2E) * nrec0.z, (-0.5f * LOG2E) * nrec1.x);
      s_bo[tid] = make_float2(-LOG2E * nrec0.w, nrec1.y);
      s_c0[tid] = make_float4(nrec1.z, nrec1.w, nrec2.x, nrec2.y);
      if (FULL) s_c1[tid] = make_float4(nrec2.z, nrec2.w, nrec3.x, nrec3.y);
      s_con[tid] = make_float4(nrec0.z, nrec0.w, nrec1.x, nrec3.z);
      if (AUXX) s_axc[tid] = naxc;
    }
    __syncthreads();
    if (tid < min(SB, sbase - SB + 1)) fetch(sbase - SB - tid);
   for (int off = 0; off < scnt; off += BB) {
    const int base = sbase - off;             // list index of batch element 0
    const int cnt = min(BB, scnt - off);
    // ---- phase A: advance the per-pixel recurrence, emit w and t (branch-free, unrolled) ------------------
#pragma unroll 4
    for (int j = 0; j < cnt; ++j) {
      const int idx = base - j;
      const float4 a = s_geo[off + j];
      const float2 bo = s_bo[off + j];
      const float4 c0 = s_c0[off + j];
      const f32x2 d = f32x2{a.x, a.y} - pix2;
      const f32x2 u = (d * f32x2{a.z, a.w}) * d;
      const float p2 = __builtin_fmaf(bo.x, d.x * d.y, u.x + u.y);
      const float G = __builtin_amdgcn_exp2f(p2);
      const float alpha = fminf(0.99f, bo.y * G);
      const bool valid = (idx < last_contributor) && !(p2 > 0.0f) && !(alpha < ALPHA_MIN);
      float cd = c0.x * dpix[0] + c0.y * dpix[1] + c0.z * dpix[2];
      if (FULL) {
        const float4 c1 = s_c1[off + j];
        cd += c0.w * dpix[3] + c1.x * dpix[4] + c1.y * dpix[5] + c1.z * dpix[6] + c1.w * dpix[7];
      }
      const float alpha_e = valid ? alpha : 0.f;
      const float inv1ma = __builtin_amdgcn_rcpf(1.0f - alpha_e);    // exactly 1 when not contributing
      T = T * inv1ma;
      const float w = alpha_e * T;
      const float dL_dalpha = T * cd - inv1ma * Q;
      Q = __builtin_fmaf(w, cd, Q);
      if (!XONLY) s_W[(wave * BB + j) * WROW + lane] = w;
      s_T[(wave * BB + j) * WROW + lane] = valid ? G * dL_dalpha : 0.f;
      if (AUXX) {
        const float4 ax = s_axc[off + j];
        const float cdx = ax.x * dpx[0] + ax.y * dpx[1] + ax.z * dpx[2];
        const float dLx = T * cdx - inv1ma * Qx;
        Qx = __builtin_fmaf(w, cdx, Qx);
        s_X[(wave * BB + j) * WROW + lane] = valid ? G * dLx : 0.f;
      }
    }
    __syncthreads();
    // ---- phase B: [16 Gaussians x 256 pixels] x [256 pixels x 16 features] on the matrix cores, per matrix ------------
    {
      const int rowo = ((lane >> 4) * BB + (lane & 15)) * WROW + 16 * wave;
      f32x4 accW = {0.f, 0.f, 0.f, 0.f}, accT = accW, accX = accW;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 t4 = *reinterpret_cast<const float4*>(s_T + rowo + 4 * s4);
        if (!XONLY) {
          const float4 w4 = *reinterpret_cast<const float4*>(s_W + rowo + 4 * s4);
          accW = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, bfragW[4 * s4 + 0], accW, 0, 0, 0);
          accT = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.x, bfragM[4 * s4 + 0], accT, 0, 0, 0);
          accW = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, bfragW[4 * s4 + 1], accW, 0, 0, 0);
          accT = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.y, bfragM[4 * s4 + 1], accT, 0, 0, 0);
          accW = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, bfragW[4 * s4 + 2], accW, 0, 0, 0);
          accT = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.z, bfragM[4 * s4 + 2], accT, 0, 0, 0);
          accW = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, bfragW[4 * s4 + 3], accW, 0, 0, 0);
          accT = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.w, bfragM[4 * s4 + 3], accT, 0, 0, 0);
        } else {
          accT = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.x, bfragM[4 * s4 + 0], accT, 0, 0, 0);
          accT = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.y, bfragM[4 * s4 + 1], accT, 0, 0, 0);
          accT = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.z, bfragM[4 * s4 + 2], accT, 0, 0, 0);
          accT = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.w, bfragM[4 * s4 + 3], accT, 0, 0, 0);
        }
        if (AUXX) {
          const float4 x4 = *reinterpret_cast<const float4*>(s_X + rowo + 4 * s4);
          accX = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.x, bfragM[4 * s4 + 0], accX, 0, 0, 0);
          accX = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.y, bfragM[4 * s4 + 1], accX, 0, 0, 0);
          accX = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.z, bfragM[4 * s4 + 2], accX, 0, 0, 0);
          accX = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.w, bfragM[4 * s4 + 3], accX, 0, 0, 0);
        }
      }
      // D[g = 4*(lane>>4) + r][f = lane&15]
      if ((lane & 15) < 8) {              // (no product uses more than eight feature columns)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (!XONLY) s_res(4 * (lane >> 4) + r, MW, wave, lane & 15) = accW[r];
          s_res(4 * (lane >> 4) + r, MT, wave, lane & 15) = accT[r];
          if (AUXX) s_res(4 * (lane >> 4) + r, MX, wave, lane & 15) = accX[r];
        }
      }
    }
    __syncthreads();
    // ---- one 64-byte gradient row per (tile, Gaussian) instance -------------------------------------------------
    if (tid < cnt) {
      const float4 ra = s_geo[off + tid], con = s_con[off + tid];
      const float2 rbo = s_bo[off + tid];
      float Dw[NCH], Dt[6];
#pragma unroll
      for (int k = 0; k < NCH; ++k)
        Dw[k] = XONLY ? 0.f : (s_res(tid, MW, 0, k) + s_res(tid, MW, 1, k)) + (s_res(tid, MW, 2, k) + s_res(tid, MW, 3, k));
#pragma unroll
      for (int k = 0; k < 6; ++k)
        Dt[k] = (s_res(tid, MT, 0, k) + s_res(tid, MT, 1, k)) + (s_res(tid, MT, 2, k) + s_res(tid, MT, 3, k));
      const float X = ra.x - tile_x0, Y = ra.y - tile_y0;
      const float A = con.x, B = con.y, Cc = con.z, op = rbo.y;
      const float S0 = Dt[0], Sx = Dt[1], Sy = Dt[2], Sxx = Dt[3], Sxy = Dt[4], Syy = Dt[5];
      const float tdx = X * S0 - Sx, tdy = Y * S0 - Sy;
      const float tdxx = X * X * S0 - 2.f * X * Sx + Sxx;
      const float tdxy = X * Y * S0 - X * Sy - Y * Sx + Sxy;
      const float tdyy = Y * Y * S0 - 2.f * Y * Sy + Syy;
      float4 r4[4];
      if (XONLY) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        r4[0] = make_float4(-op * (A * tdx + B * tdy), -op * (Cc * tdy + B * tdx), 0.f, 0.f);
        r4[1] = z; r4[2] = z; r4[3] = z;
      } else {
        r4[0] = make_float4(-op * (A * tdx + B * tdy), -op * (Cc * tdy + B * tdx), -0.5f * op * tdxx, -op * tdxy);
        r4[1] = make_float4(-0.5f * op * tdyy, S0, Dw[0], Dw[1]);
        r4[2] = make_float4(Dw[2], Dw[3], Dw[4], Dw[5]);       // AUX: slots 9..11 = dL/d(aux colour)
        r4[3] = make_float4(Dw[6], Dw[7], 0.f, 0.f);
        if (AUXW) r4[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (AUXX) {
          // slots 12, 13: the aux image's share of dL/d(screen-space mean)
          float Dx[3];
#pragma unroll
          for (int k = 0; k < 3; ++k)
            Dx[k] = (s_res(tid, NMAT - 1, 0, k) + s_res(tid, NMAT - 1, 1, k)) +
                    (s_res(tid, NMAT - 1, 2, k) + s_res(tid, NMAT - 1, 3, k));
          const float xdx = X * Dx[0] - Dx[1], xdy = Y * Dx[0] - Dx[2];
          r4[3] = make_float4(-op * (A * xdx + B * xdy), -op * (Cc * xdy + B * xdx), 0.f, 0.f);
        }
      }
      const uint32_t row = __float_as_uint(con.w);
      float4* dst = reinterpret_cast<float4*>(inst_grad + (size_t)row * REC_FLOATS);
      dst[0] = r4[0]; dst[1] = r4[1]; dst[2] = r4[2]; dst[3] = r4[3];
      row_flag[row] = 1;
    }
   }
  }
}

// Persistent over the work list the forward pass built: (tile, segment) pairs, seg_count of them.
template <bool FULL, int AUX, bool XONLY>
__global__ void __launch_bounds__(BLOCK, AUX == 1 ? 2 : 4)   // (waves per SIMD the LDS footprint allows: 128 / 256 registers)
blend_backward_kernel(Camera c, const int32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const uint32_t* __restrict__ slot_list, const float* __restrict__ rec2d, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ final_T, const float* __restrict__ dL_dcolor,
                      const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dnormal,
                      const float* __restrict__ dL_dalpha_img, const float* __restrict__ dL_dextra,
                      float* __restrict__ inst_grad, const float* __restrict__ color_override,
                      const float* __restrict__ aux_colors, const float* __restrict__ dL_daux,
                      const uint32_t* __restrict__ seg_queue, const uint32_t* __restrict__ seg_count,
                      const float* __restrict__ seg_state, const uint32_t* __restrict__ tile_rounds,
                      uint8_t* __restrict__ row_flag) {
  const uint32_t count = *seg_count;
  for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
    blend_backward_segment<FULL, AUX, XONLY>((int)seg_queue[2 * item], (int)seg_queue[2 * item + 1], c, ranges, point_list,
                                             slot_list, rec2d, n_contrib, final_T, dL_dcolor, dL_ddepth, dL_dnormal,
                                             dL_dalpha_img, dL_dextra, inst_grad, color_override, aux_colors, dL_daux,
                                             seg_state, tile_rounds, row_flag);
    __syncthreads();                      // the next segment reuses the LDS arrays
  }
}

}  // namespace

int launch_blend_forward(const Camera& c, const int32_t* ranges, const uint32_t* point_list,
                         const float* rec2d, uint32_t* n_contrib, float* final_T, float* out_color,
                         float* out_depth, float* out_normal, float* out_alpha, float* out_extra,
                         const float* aux_colors, float* out_aux, uint32_t* seg_queue, uint32_t* seg_count,
                         float* seg_state, uint32_t* tile_rounds, uint32_t* tile_sync, uint32_t* seg_flag,
                         uint32_t* walk_hints, int64_t instances, hipStream_t s) {
  const int tiles = c.grid_x * c.grid_y;
  if (tiles == 0) return INSTAG_OK;
  // Which forward kernel: the segment-wise one pays where tiles walk far (its helpers cut those chains); on a small
  // scene -- the mouth's 20k Gaussians -- every tile is short and its bookkeeping costs 5 %.  By the instance count, so
  // that the eager and the captured form of one scene run the same arithmetic.  INSTAG_BLEND_FWD=tile / segment forces
  // one (tests run both); the tile kernel also runs when nothing cleared the claim words (an empty frame).
  const char* e_fwd = getenv("INSTAG_BLEND_FWD");
  const bool by_segment = e_fwd && strcmp(e_fwd, "tile") == 0 ? false
                          : e_fwd && strcmp(e_fwd, "segment") == 0 ? true : instances >= 500000;
  { ProfScope calibration(K_EMPTY_BRACKET, s); }
  ProfScope p(K_BLEND_FWD, s);
  if (by_segment && tile_sync != nullptr) {
    // helper workgroups (behind the tiles' own in dispatch order) only with walk hints
    // INSTAG_BLEND_FWD_SHARE_ALL=1 (tests): helpers for every tile of three or more segments, with or without hints
    const char* e_all = getenv("INSTAG_BLEND_FWD_SHARE_ALL");
    const int share_all = (e_all && e_all[0] == '1') ? 1 : 0;
#ifdef FWD_NO_HELP
    walk_hints = nullptr;
#endif
    const int grid = (walk_hints || share_all) ? tiles * (1 + FWD_HELPERS) : tiles;
    uint32_t* stalls = sort_stalls_device_ptr();
    if (aux_colors)
      blend_forward_claim_kernel<true><<<grid, BLOCK, 0, s>>>(c, ranges, point_list, rec2d, n_contrib, final_T, out_color,
                                                              out_depth, out_normal, out_alpha, out_extra, aux_colors,
                                                              out_aux, seg_queue, seg_count, seg_state, tile_rounds,
                                                              tile_sync, seg_flag, walk_hints, stalls, tiles, share_all);
    else
      blend_forward_claim_kernel<false><<<grid, BLOCK, 0, s>>>(c, ranges, point_list, rec2d, n_contrib, final_T, out_color,
                                                               out_depth, out_normal, out_alpha, out_extra, nullptr,
                                                               nullptr, seg_queue, seg_count, seg_state, tile_rounds,
                                                               tile_sync, seg_flag, walk_hints, stalls, tiles, share_all);
    INSTAG_CHECK_LAUNCH();
    return INSTAG_OK;
  }
  if (aux_colors)
    blend_forward_kernel<true><<<tiles, BLOCK, 0, s>>>(c, ranges, point_list, rec2d, n_contrib, final_T, out_color,
                                                       out_depth, out_normal, out_alpha, out_extra, aux_colors, out_aux,
                                                       seg_queue, seg_count, seg_state, tile_rounds);
  else
    blend_forward_kernel<false><<<tiles, BLOCK, 0, s>>>(c, ranges, point_list, rec2d, n_contrib, final_T, out_color,
                                                        out_depth, out_normal, out_alpha, out_extra, nullptr, nullptr,
                                                        seg_queue, seg_count, seg_state, tile_rounds);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int launch_blend_backward(const Camera& c, const int32_t* ranges, const uint32_t* point_list,
                          const uint32_t* slot_list, const float* rec2d, const uint32_t* n_contrib, const float* final_T,
                          const float* dL_dcolor, const float* dL_ddepth, const float* dL_dnormal,
                          const float* dL_dalpha, const float* dL_dextra, float* inst_grad,
                          const float* color_override, const float* aux_colors, const float* dL_daux, int aux_mode,
                          const uint32_t* seg_queue, const uint32_t* seg_count, const float* seg_state,
                          const uint32_t* tile_rounds, uint32_t seg_slots, uint8_t* row_flag, hipStream_t s) {
  // aux_mode: 0 none, 1 whole aux image in this launch, 2 aux colours' gradient only, 3 mean-only pass over
  // `color_override` (rows carry dx, dy)
  const int tiles = c.grid_x * c.grid_y;
  if (tiles == 0) return INSTAG_OK;
  const bool full = dL_ddepth || dL_dnormal || dL_dextra;
  if ((aux_mode == 1 || aux_mode == 2) && (full || aux_colors == nullptr || dL_daux == nullptr)) {
    set_error("blend backward: the auxiliary gradients need an rgb-only main pass, aux_colors and dL_daux");
    return INSTAG_E_ARG;
  }
  if (aux_mode == 3 && (full || color_override == nullptr)) {
    set_error("blend backward: the mean-only pass takes its colours from color_override");
    return INSTAG_E_ARG;
  }
  // at most 16 workgroups per CU: enough to fill every slot the LDS footprint allows four times over; a workgroup
  // takes further segments of the list in strides of the grid
  const uint32_t grid = std::min<uint32_t>(seg_slots, 4096u);
  ProfScope p(aux_mode == 3 ? K_BLEND_BWD_MEAN : K_BLEND_BWD, s);
#define INSTAG_BB(F, A, X, d, n, e, ax, dax)                                                                            \
  blend_backward_kernel<F, A, X><<<grid, BLOCK, 0, s>>>(c, ranges, point_list, slot_list, rec2d, n_contrib, final_T,     \
                                                       dL_dcolor, d, n, dL_dalpha, e, inst_grad, color_override, ax,    \
                                                       dax, seg_queue, seg_count, seg_state, tile_rounds, row_flag)
  if (aux_mode == 1) INSTAG_BB(false, 1, false, nullptr, nullptr, nullptr, aux_colors, dL_daux);
  else if (aux_mode == 2) INSTAG_BB(false, 2, false, nullptr, nullptr, nullptr, aux_colors, dL_daux);
  else if (aux_mode == 3) INSTAG_BB(false, 0, true, nullptr, nullptr, nullptr, nullptr, nullptr);
  else if (full) INSTAG_BB(true, 0, false, dL_ddepth, dL_dnormal, dL_dextra, nullptr, nullptr);
  else INSTAG_BB(false, 0, false, nullptr, nullptr, nullptr, nullptr, nullptr);
#undef INSTAG_BB
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // namespace instag

#ifdef BLEND_DBG
extern "C" int instag_debug_blend_timing(uint32_t* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(instag::g_blend_dbg), sizeof(uint32_t) * (size_t)n_words);
}
#endif

#ifdef BLEND_DBG2
extern "C" int instag_debug_blend_stalls(uint32_t* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(instag::g_stall_log), sizeof(uint32_t) * (size_t)n_words);
}
extern "C" int instag_debug_blend_stall_host(uint32_t* pinned_device_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(instag::g_stall_host), &pinned_device_ptr, sizeof(pinned_device_ptr));
}
#endif
