// Shared host/device helpers for libinstag_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include <mutex>
#include <set>
#include <string>
#include <utility>

#include "instag_hip.h"

namespace instag {

void set_error(const std::string& msg);

#define INSTAG_CHECK_HIP(expr)                                                                  \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      ::instag::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
      return INSTAG_E_HIP;                                                                      \
    }                                                                                           \
  } while (0)

#define INSTAG_REQUIRE(cond, msg)                                                               \
  do {                                                                                          \
    if (!(cond)) {                                                                              \
      ::instag::set_error(std::string(msg));                                                    \
      return INSTAG_E_ARG;                                                                      \
    }                                                                                           \
  } while (0)

#define INSTAG_CHECK_LAUNCH() INSTAG_CHECK_HIP(hipGetLastError())

// ---- per-kernel event timing (bench.py roofline leg) -----------------------------------------
enum ProfKernel {
  K_PREPROCESS = 0, K_DUPLICATE = 1, K_SORT = 2, K_RANGES = 3, K_BLEND_FWD = 4, K_BLEND_BWD = 5,
  K_PREPROCESS_BWD = 6, K_GRID_FWD = 7, K_GRID_BWD = 8, K_SH_FWD = 9, K_SH_BWD = 10,
  K_MLP_FWD = 11, K_MLP_BWD = 12, K_MLP_WGRAD = 13, K_LOSS_FWD = 14, K_LOSS_BWD = 15, K_BLEND_BWD_MEAN = 16,
  K_EMPTY_BRACKET = 17,      // two event records with nothing between them: what a bracket itself costs
};
struct ProfScope {
  ProfScope(int kernel, hipStream_t stream);
  ~ProfScope();
  int kernel_;
  hipStream_t stream_;
  hipEvent_t start_ = nullptr;
  int graph_slot_ = -1;        // >= 0: the pair comes from the graph pool (the launch is being captured)
};

template <typename T>
static inline T div_up(T a, T b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device); callable from any thread (forward runs
// on the caller's thread, backward on autograd's) -- the attribute belongs to the device's copy of the code object
static inline int set_max_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  INSTAG_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({kernel, dev})) return INSTAG_OK;
  INSTAG_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({kernel, dev});
  return INSTAG_OK;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- rasterizer buffer layouts (device memory, opaque to the caller) ----------------------------
constexpr int TILE_X = 16;
constexpr int TILE_Y = 16;
constexpr int REC_FLOATS = 16;  // 64-byte per-Gaussian blend record

// rec2d[g][0..15]:
//  0 x   1 y   2 conA  3 conB  4 conC  5 opacity  6 r  7 g  8 b  9 depth  10 nx  11 ny  12 nz  13 extra
//  14 (u32) exclusive instance offset   15 (u32) rect_min_x | rect_min_y<<10 | rect_w<<20
enum RecSlot { R_X = 0, R_Y, R_CA, R_CB, R_CC, R_OP, R_R, R_G, R_B, R_DEPTH, R_NX, R_NY, R_NZ, R_EXTRA,
               R_OFFSET, R_RECT };

// flags[g]: bit0..2 rgb clamped, bit3..4 normal axis, bit5 normal flipped, bit6 tx clamped, bit7 ty clamped
struct GeomLayout {
  size_t rec2d, cov3d, tiles_touched, point_offsets, flags, cull_thr;
  // depth sort: keys ping-pong K0 -> KA -> K0 -> KA, values (index) -> VA -> VB -> VA -> VB = `order`
  size_t depth_key, depth_key_alt, order_a, order;
  size_t dsort_zero, dsort_zero_words;      // [16 words: one ticket per pass][4 passes x blocks x 256 look-back words]
  size_t dsort_digit_base;                  // [4][256]
  size_t dsort_partials;                    // [blocks][4][256]
  size_t scan_state, scan_state_words;      // u64 [1 + blocks] as 32-bit words; cleared by the preprocess kernel
  uint32_t dsort_blocks, dkey_blocks;
  size_t total;
};
GeomLayout geom_layout(int32_t N);

struct ImageLayout {
  size_t ranges, n_contrib, final_T;
  size_t tile_rounds;                        // u32 per tile: segments of its list the forward blend walked
  size_t total;
};
ImageLayout image_layout(int32_t H, int32_t W);

struct BinningLayout {
  // instance sort: keys K0 = keys_unsorted -> (ktmp) -> keys, values likewise (pair form only)
  size_t keys_unsorted, vals_unsorted, keys, vals, ktmp, vtmp, gid_unsorted, point_list;
  size_t tsort_zero, tsort_zero_words;      // [16 words: tickets][3 passes x blocks x 256 look-back words]
  size_t tsort_digit_base;                  // [3][256]
  size_t tsort_partials;                    // [1024][3][256]
  size_t sort_count;                        // u32: instances actually binned
  uint32_t tsort_blocks;
  // Blend segments (SEG_LEN list entries of one tile).  Segment k of tile t keeps the per-pixel blend state AFTER it in
  // seg_state[start_t / SEG_LEN + t + k] (unique, < seg_slots): [SEG_FLOATS][TILE_PIX] = T, the 8 channel
  // accumulators, the 3 aux ones.  The forward blend appends every segment that holds a contributor to seg_queue
  // ((tile, k) pairs; seg_count of them) -- the work list of the backward blend.  row_flag[set][slot] != 0 marks the
  // gradient rows a backward blend wrote (two sets: the main and the auxiliary pass may run side by side); the
  // reducing kernel clears what it consumes.  seg_count and row_flag follow tsort_zero and are cleared with it.
  size_t seg_state, seg_slots, seg_queue, seg_count, row_flag, row_flag_stride;
  // segment-wise forward blend: per tile 8 words {next segment to claim, segments walked, ~(first finished segment),
  // resolved, helpers joined, -, -, -}, per segment slot a "posted" flag -- both inside the region the duplicate kernel clears
  size_t fwd_sync, seg_flag;
  size_t total;
};
constexpr int SEG_LEN = 128;                 // list entries per blend segment (divides the forward blend's batch of 256)
constexpr int SEG_FLOATS = 14;               // + t_seg, the segment's last contributor (raster_blend.hip: segment-wise forward)
constexpr int TILE_PIX = TILE_X * TILE_Y;
BinningLayout binning_layout(int64_t R, int32_t H, int32_t W);

}  // namespace instag
