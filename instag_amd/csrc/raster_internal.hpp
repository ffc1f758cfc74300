// Internal launcher declarations shared by the rasterizer translation units.
#pragma once
#include "common.hpp"

namespace instag {

struct Camera {  // kernel-side view of instag_raster_args' scalar part
  int32_t N, M, sh_degree, E, H, W;
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  int32_t grid_x, grid_y;
  const float *bg, *view, *proj, *campos;
};

Camera make_camera(const instag_raster_args* a);

// digit p of a tile id = (tile >> (p * bits_per)) & ((1 << nbits[p]) - 1): the instance sort's passes (<= 8 bits each)
struct TilePasses {
  int npass, bits_per, nbits[3];
};
inline TilePasses tile_passes(int tiles) {
  int bits = 1;
  while ((1 << bits) < tiles) ++bits;
  TilePasses tp;
  tp.npass = (bits + 7) / 8;
  tp.bits_per = (bits + tp.npass - 1) / tp.npass;
  for (int p = 0; p < 3; ++p) tp.nbits[p] = p < tp.npass ? std::min(tp.bits_per, bits - p * tp.bits_per) : 0;
  return tp;
}

// raster_preprocess.hip (built with -ffp-contract=off: bit-exact against the oracle)
int launch_preprocess(const Camera& c, const instag_raster_args* a, float* rec2d, float* cov3d,
                      uint32_t* tiles_touched, uint32_t* flags, float* cull_thr, int32_t* radii,
                      uint32_t* zero_words, uint32_t n_zero_words, hipStream_t s);
uint32_t depth_key_blocks(int32_t N);
int launch_depth_keys(const Camera& c, const float* means3D, uint32_t* depth_key, uint32_t* partials,
                      uint32_t* zero_words, uint32_t n_zero_words, hipStream_t s);
int launch_export_keys(int64_t R, const uint32_t* tile_keys, const uint32_t* point_list, const float* rec2d,
                       uint64_t* keys64, bool packed, hipStream_t s);
uint32_t duplicate_blocks(int32_t N);
int launch_duplicate(const Camera& c, float* rec2d, const uint32_t* order,
                     const uint32_t* point_offsets, const uint32_t* flags, const float* cull_thr, uint32_t* keys,
                     uint32_t* vals, uint32_t* gid_unsorted, uint32_t capacity, int32_t* ranges,
                     bool packed, int32_t* status, uint32_t* sort_count, const TilePasses& tp, uint32_t* partials,
                     uint32_t* zero_words, uint32_t n_zero_words, hipStream_t s);
int launch_ranges(int64_t R, const uint32_t* count_ptr, const uint32_t* keys_sorted, uint32_t* slots_sorted,
                  const uint32_t* gid_unsorted, uint32_t* point_list, int32_t* ranges, uint32_t ntiles, bool packed,
                  hipStream_t s);

// raster_sort.hip: stable radix passes (one launch each), digit histogram scan, instance-offset scan
constexpr int SORT_IPT_DEPTH = 4;     // x 1,024 threads = 4,096 keys per block: 25 blocks for 100k Gaussians
constexpr int SORT_IPT_TILE = 8;      // x 1,024 threads = 8,192 keys per block
constexpr int HIST_SLICES = 8;        // second-level partial histograms of the instance sort
uint32_t sort_blocks(uint32_t count_max, int ipt);
int launch_radix_pass(int ipt, bool has_values, bool write_keys, const uint32_t* keys_in, uint32_t* keys_out,
                      const uint32_t* vals_in, uint32_t* vals_out, const uint32_t* count_ptr, uint32_t count_max,
                      int shift, int nbits, const uint32_t* hist, int n_hist, int hist_stride, uint32_t* ticket,
                      uint32_t* lookback, hipStream_t s, uint64_t* stamps = nullptr);
int read_sort_stalls(uint32_t* host_out, hipStream_t s, bool synchronize);
int clear_sort_stalls(hipStream_t s);
uint32_t* sort_stalls_device_ptr();
int launch_hist_reduce(const uint32_t* partials, int nblk, int npass, int slices, uint32_t* out, hipStream_t s);
int launch_scan_counts(int N, const uint32_t* order, const uint32_t* tiles_touched, uint32_t* point_offsets,
                       uint64_t* state, hipStream_t s);

// Instance sort as a KEY-ONLY 32-bit radix sort when the slot index fits below the tile id: key = tile << 21 | slot
// (R < 2^21 instances, < 2^11 - 1 tiles; config C3 qualifies).  Half the bytes per radix pass of a pair sort.
constexpr int PACK_SHIFT = 21;
inline bool use_packed_keys(int64_t R, int tiles) { return R < ((int64_t)1 << PACK_SHIFT) && tiles < (1 << 11) - 1; }

// raster_blend.hip
int launch_blend_forward(const Camera& c, const int32_t* ranges, const uint32_t* point_list,
                         const float* rec2d, uint32_t* n_contrib, float* final_T, float* out_color,
                         float* out_depth, float* out_normal, float* out_alpha, float* out_extra,
                         const float* aux_colors, float* out_aux, uint32_t* seg_queue, uint32_t* seg_count,
                         float* seg_state, uint32_t* tile_rounds, uint32_t* tile_sync /* zeroed, or NULL */,
                         uint32_t* seg_flag /* zeroed */, uint32_t* walk_hints /* persistent, or NULL */,
                         int64_t instances, hipStream_t s);
int launch_blend_backward(const Camera& c, const int32_t* ranges, const uint32_t* point_list,
                          const uint32_t* slot_list, const float* rec2d, const uint32_t* n_contrib, const float* final_T,
                          const float* dL_dcolor, const float* dL_ddepth, const float* dL_dnormal,
                          const float* dL_dalpha, const float* dL_dextra, float* inst_grad,
                          const float* color_override, const float* aux_colors, const float* dL_daux, int aux_mode,
                          const uint32_t* seg_queue, const uint32_t* seg_count, const float* seg_state,
                          const uint32_t* tile_rounds, uint32_t seg_slots, uint8_t* row_flag, hipStream_t s);

// raster_backward.hip
int launch_preprocess_backward(const Camera& c, const instag_raster_args* a, const float* rec2d,
                               const float* cov3d, const uint32_t* tiles_touched, const uint32_t* flags,
                               const int32_t* radii, const float* inst_grad, uint8_t* row_flag, uint32_t capacity,
                               float* dL_dmeans3D,
                               float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                               float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                               float* dL_dcov3D, float* dL_dextra, float* dL_dshs_rest, float* dL_daux_colors,
                               hipStream_t s);
int launch_aux_backward_reduce(const Camera& c, const float* rec2d, const uint32_t* tiles_touched, const int32_t* radii,
                               const float* inst_grad, uint8_t* row_flag, uint32_t capacity, float* dL_daux_colors,
                               float* dL_dmeans2D, bool accumulate_means2D, hipStream_t s);

}  // namespace instag
