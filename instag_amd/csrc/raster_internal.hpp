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

// raster_preprocess.hip (built with -ffp-contract=off: bit-exact against the oracle)
int launch_preprocess(const Camera& c, const instag_raster_args* a, float* rec2d, float* cov3d,
                      uint32_t* tiles_touched, uint32_t* flags, float* cull_thr, uint32_t* depth_key,
                      uint32_t* order_in, int32_t* radii, hipStream_t s);
int launch_export_keys(int64_t R, const uint32_t* tile_keys, const uint32_t* point_list, const float* rec2d,
                       uint64_t* keys64, bool packed, hipStream_t s);
int launch_duplicate(const Camera& c, float* rec2d, const uint32_t* order,
                     const uint32_t* point_offsets, const uint32_t* flags, const float* cull_thr, uint32_t* keys,
                     uint32_t* vals, uint32_t* gid_unsorted, uint32_t capacity, bool pad, int32_t* ranges,
                     bool packed, int32_t* status, hipStream_t s);
int launch_ranges(int64_t R, const uint32_t* keys_sorted, uint32_t* slots_sorted, const uint32_t* gid_unsorted,
                  uint32_t* point_list, int32_t* ranges, uint32_t ntiles, bool packed, hipStream_t s);

// Instance sort as a KEY-ONLY 32-bit radix sort when the slot index fits below the tile id: key = tile << 21 | slot
// (R < 2^21 instances, < 2^11 - 1 tiles; config C3 qualifies).  Half the bytes per radix pass of a pair sort.
constexpr int PACK_SHIFT = 21;
inline bool use_packed_keys(int64_t R, int tiles) { return R < ((int64_t)1 << PACK_SHIFT) && tiles < (1 << 11) - 1; }

// raster_blend.hip
int launch_blend_forward(const Camera& c, const int32_t* ranges, const uint32_t* point_list,
                         const float* rec2d, uint32_t* n_contrib, float* final_T, float* out_color,
                         float* out_depth, float* out_normal, float* out_alpha, float* out_extra,
                         const float* aux_colors, float* out_aux, hipStream_t s);
int launch_blend_backward(const Camera& c, const int32_t* ranges, const uint32_t* point_list,
                          const uint32_t* slot_list, const float* rec2d, const uint32_t* n_contrib, const float* final_T,
                          const float* dL_dcolor, const float* dL_ddepth, const float* dL_dnormal,
                          const float* dL_dalpha, const float* dL_dextra, float* inst_grad,
                          const float* color_override, hipStream_t s);

// raster_backward.hip
int launch_preprocess_backward(const Camera& c, const instag_raster_args* a, const float* rec2d,
                               const float* cov3d, const uint32_t* tiles_touched, const uint32_t* flags,
                               const int32_t* radii, const float* inst_grad, uint32_t capacity,
                               float* dL_dmeans3D,
                               float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                               float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                               float* dL_dcov3D, float* dL_dextra, float* dL_dshs_rest, hipStream_t s);
int launch_aux_backward_reduce(const Camera& c, const float* rec2d, const uint32_t* tiles_touched, const int32_t* radii,
                               const float* inst_grad, uint32_t capacity, float* dL_daux_colors, float* dL_dmeans2D,
                               hipStream_t s);

}  // namespace instag
