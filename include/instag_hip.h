/*
 * instag_hip.h -- C ABI of libinstag_hip.so: the MI355X (gfx950) implementation of the
 * InsTaG 3D-Gaussian-Splatting render/train hot path.
 *
 * Plain pointers and sizes only; every pointer is DEVICE memory unless marked (host).
 * Every function enqueues work on the given HIP stream (`instag_stream_t` is a
 * `hipStream_t`) and returns 0 on success or a non-zero code; `instag_last_error()`
 * returns a human-readable message for the calling thread.  The caller allocates every
 * output / scratch buffer (the ownership rule of the reference's Python autograd
 * Functions: gridencoder/grid.py:47-54,77-84, shencoder/sphere_harmonics.py:25-32,50-51).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference checkout).
 */
#ifndef INSTAG_HIP_H
#define INSTAG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* instag_stream_t; /* hipStream_t */

#define INSTAG_OK 0
#define INSTAG_E_ARG 1     /* invalid argument (shape / unsupported D, C, degree ...) */
#define INSTAG_E_HIP 2     /* a HIP runtime call or kernel launch failed */
#define INSTAG_E_SPACE 3   /* caller-provided workspace too small */

const char* instag_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int instag_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Multiresolution grid encoder.
 * Replaces gridencoder/src/gridencoder.h:12-15 / bindings.cpp:5-7:
 *   grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp)
 *   grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, ...)
 *   grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners)
 * inputs [B,D] fp32 in [0,1]; embeddings [sO,C] fp32; offsets [L+1] int32; outputs [L,B,C];
 * dy_dx [B, L*D*C] or NULL; grad [L,B,C]; grad_embeddings [sO,C] and grad_inputs [B,D] arrive
 * zero-filled and are accumulated into.  D in {2,3}, C in {1,2,4,8}.
 * grad_total_variation adds, for every table entry hit by a sample, the normalised sum of its differences to its axis
 * neighbours (gridencoder.cu:506-610) into `grad` [sO,C]; `workspace` (device, instag_grid_total_variation_workspace_bytes
 * bytes, cleared by the call) holds per-entry sample counts and fixed-point sums: no float atomics, reproducible.
 * ------------------------------------------------------------------------------------------ */
int instag_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets,
                               float* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                               uint32_t H, float* dy_dx, uint32_t gridtype, int align_corners,
                               uint32_t interp, instag_stream_t stream);
int instag_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                uint32_t C, uint32_t L, float S, uint32_t H, const float* dy_dx,
                                float* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                                instag_stream_t stream);
size_t instag_grid_total_variation_workspace_bytes(uint32_t total_params, uint32_t C);
int instag_grid_total_variation(const float* inputs, const float* embeddings, float* grad,
                                const int32_t* offsets, float weight, uint32_t B, uint32_t D, uint32_t C,
                                uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                uint32_t total_params, void* workspace, size_t workspace_bytes,
                                instag_stream_t stream);

/* Tri-plane encoder: the three identically configured 2-D, C=1 grid encoders of a motion field (planes xy, yz, xz;
 * scene/motion_net.py:214-216,244-258) in one pass: xyz [N,3] in [-bound,bound] -> out [N,3L] = cat(enc_xy, enc_yz,
 * enc_xz), including the (x+bound)/(2 bound) mapping (gridencoder/grid.py:149).  Each plane's table is
 * [total_params,1] with shared `offsets` [L+1], every level dense.  backward: grad [N,3L] -> dxyz [N,3] (written, may
 * be NULL) and the three table gradients (written, not accumulated).  workspace:
 * instag_triplane_backward_workspace_bytes(N, total_params) bytes of scratch.
 * Tables of up to 13,312 entries per plane (the face fields) are staged in LDS whole; larger tables (the mouth field,
 * 46,600 entries) are read in place and their gradient is accumulated level by level in LDS.  Either way the sums are
 * 64-bit fixed point, added up in a fixed order: bitwise reproducible.  Only a single level of more than 16,384 cells
 * falls back to global float atomics (summation order not fixed, as in gridencoder.cu:300-330).
 * shift (optional, [N, shift_stride >= 3]): the encoders are evaluated at xyz + shift_scale * shift[:, :3] (the universal
 * field sits behind the personalised alignment, gaussian_renderer/__init__.py:196-197); backward then also writes
 * dshift [N, shift_stride] = (shift_scale * d/dpoint, 0, ...) when it is non-NULL (needs dxyz).
 * dxyz_add [N,3] / dshift_add [N, shift_stride] (optional, distinct from the outputs): gradients that the position's /
 * the shift's OTHER consumers produced, added into dxyz / dshift by the same kernel -- the host code passes the position
 * through the encoder (instag_amd/gridencoder.py tri_plane_encode(passthrough=True)), so autograd never launches a
 * separate add for them. */
int instag_triplane_forward(const float* xyz, const float* table_xy, const float* table_yz,
                            const float* table_xz, const int32_t* offsets, float* out, const float* shift,
                            uint32_t shift_stride, float shift_scale, uint32_t N, uint32_t L,
                            float S, uint32_t H, float bound, uint32_t total_params, instag_stream_t stream);
size_t instag_triplane_backward_workspace_bytes(uint32_t N, uint32_t total_params);
int instag_triplane_backward(const float* grad, const float* xyz, const float* table_xy, const float* table_yz,
                             const float* table_xz, const int32_t* offsets, float* dxyz, float* dtable_xy,
                             float* dtable_yz, float* dtable_xz, void* workspace, size_t workspace_bytes,
                             const float* shift, uint32_t shift_stride, float shift_scale, float* dshift,
                             uint32_t N, uint32_t L, float S, uint32_t H, float bound, uint32_t total_params,
                             const float* dxyz_add, const float* dshift_add, instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Spherical-harmonics encoder.  Replaces shencoder/src/shencoder.h:8-9:
 *   sh_encode_forward(inputs, outputs, B, D, C, dy_dx)
 *   sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs)
 * inputs [B,3]; outputs [B,C*C]; dy_dx [B,3*C*C] or NULL; C = degree in [1,8]; D must be 3.
 * grad_inputs [B,3] is accumulated into (arrives zero-filled).
 * ------------------------------------------------------------------------------------------ */
int instag_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                             float* dy_dx, instag_stream_t stream);
int instag_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                              const float* dy_dx, float* grad_inputs, instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Differentiable tile rasterizer.  Replaces the `_C.rasterize_gaussians` /
 * `_C.rasterize_gaussians_backward` pair behind `diff_gauss.GaussianRasterizer`
 * (absent third-party module; call sites gaussian_renderer/__init__.py:58-73,111-121).
 *
 * Forward is split in two calls because the number of (tile, Gaussian) instances R is only
 * known after the per-Gaussian stage: stage1 returns R to the HOST (it synchronises the
 * stream once), the caller sizes the binning buffer with instag_raster_binning_bytes(R, H, W),
 * stage2 does duplicate -> sort -> tile ranges -> blend.
 * ------------------------------------------------------------------------------------------ */
typedef struct instag_raster_args {
  /* sizes */
  int32_t N;             /* Gaussians */
  int32_t M;             /* SH coefficients per Gaussian in `shs` (0 when colors_precomp is used) */
  int32_t sh_degree;     /* active degree, (sh_degree+1)^2 <= M, 0..3 */
  int32_t E;             /* extra attribute channels: 0 or 1 */
  int32_t image_height, image_width;
  float tanfovx, tanfovy, scale_modifier;
  int32_t prefiltered, debug;
  /* execution hint (in the struct's former padding): 1 = keep every launch of this call on `stream`.  By default the
     depth sort of the Gaussians is forked onto a second stream beside the preprocess kernel; a caller that is itself
     running on a stream FORKED inside a stream capture must set this -- a fork of a fork inside one capture crashes
     hipStreamEndCapture on ROCm 7.2 (scripts/probes/infer_capture_probe2.py) */
  int32_t single_stream;
  /* camera (device): bg[3], viewmatrix[16], projmatrix[16] (row-vector convention, i.e. the
     transposed matrices of scene/cameras.py:61-64), campos[3] */
  const float *bg, *viewmatrix, *projmatrix, *campos;
  /* per-Gaussian inputs (device). Exactly one of shs / colors_precomp, and exactly one of
     (scales, rotations) / cov3Ds_precomp, is non-NULL. */
  const float* means3D;        /* [N,3] */
  const float* shs;            /* [N,M,3] */
  const float* colors_precomp; /* [N,3] */
  const float* opacities;      /* [N,1] */
  const float* scales;         /* [N,3] */
  const float* rotations;      /* [N,4] (r,x,y,z), already normalised */
  const float* cov3Ds_precomp; /* [N,6] xx,xy,xz,yy,yz,zz */
  const float* extra_attrs;    /* [N,E] or NULL */
  /* optional split SH storage (scene/gaussian_model.py keeps _features_dc [N,1,3] and _features_rest [N,M-1,3] as
     separate parameters and concatenates them on every call, :183-186): when shs_rest != NULL, `shs` holds only the
     DC coefficient [N,1,3] and shs_rest the other M-1; M is still the total count.  NULL = `shs` is [N,M,3]. */
  const float* shs_rest;
  /* optional (NULL: none): device uint32[tiles] (tiles = ceil(W/16) * ceil(H/16)) that PERSISTS between the calls of one
     render slot, any initial content.  The forward blend keeps every tile's recent walk length there (1/8 of a
     128-entry segment per unit: up at once, down one unit per call) and starts helper workgroups for the tiles that
     walked far lately; results do not depend on it. */
  uint32_t* walk_hints;
} instag_raster_args;

size_t instag_raster_geom_bytes(int32_t N);
size_t instag_raster_image_bytes(int32_t image_height, int32_t image_width);
size_t instag_raster_binning_bytes(int64_t R, int32_t H, int32_t W);
/* scratch for backward: per-instance gradient rows */
size_t instag_raster_backward_workspace_bytes(int32_t N, int64_t R);

/* stage 1: preprocess + scan.  radii [N] int32 (output).  *num_rendered (host) receives R. */
int instag_raster_forward_stage1(const instag_raster_args* a, void* geom, size_t geom_bytes,
                                 int32_t* radii, int64_t* num_rendered /* (host) */,
                                 instag_stream_t stream);
/* stage 2: outputs color [3,H,W], depth [1,H,W], normal [3,H,W], alpha [1,H,W], extra [E,H,W] (NULL if E==0).
 * aux_colors [N,3] / out_aux [3,H,W] (both NULL or both set): a second colour set blended over the SAME instances
 * with the same alpha and transmittance, out_aux = sum c_aux alpha T + T_final bg.  It equals the colour image
 * of a second rasterizer call with colors_precomp = aux_colors on the same (detached) geometry, which is how the
 * reference renders the attention map (gaussian_renderer/__init__.py:243-258), without preprocessing, binning
 * and sorting twice. */
int instag_raster_forward_stage2(const instag_raster_args* a, void* geom, size_t geom_bytes,
                                 void* binning, size_t binning_bytes, void* image, size_t image_bytes,
                                 int64_t R, float* out_color, float* out_depth, float* out_normal,
                                 float* out_alpha, float* out_extra, const float* aux_colors, float* out_aux,
                                 instag_stream_t stream);
/* Sync-free forward (hipGraph-capturable): stage1 + stage2 in one call with a caller-chosen instance
 * capacity instead of the host round trip.  binning / backward workspace are sized for `capacity`
 * (instag_raster_binning_bytes(capacity, H, W), instag_raster_backward_workspace_bytes(N, capacity)) and
 * backward is called with R = capacity.  status (device int32[4], zero-initialised by the caller):
 *   [0] = instances this call needed (R);
 *   [1] = STICKY overflow flag: set to 1 by any call with R > capacity, never cleared by the library -- a caller that
 *         replays a captured step checks it now and then and clears it itself;
 *   [2] = largest R seen since the caller last cleared it (what to size the next capacity from);
 *   [3] = instances this call binned: R, or, on overflow, the instances of the Gaussians (in depth order) in front of
 *         the first one whose instances would cross the capacity.  Dropped Gaussians are not rendered and get zero
 *         gradients; nothing is read or written outside the buffers. */
int instag_raster_forward_capacity(const instag_raster_args* a, void* geom, size_t geom_bytes,
                                   void* binning, size_t binning_bytes, void* image, size_t image_bytes,
                                   int64_t capacity, int32_t* radii, int32_t* status, float* out_color,
                                   float* out_depth, float* out_normal, float* out_alpha,
                                   float* out_extra, const float* aux_colors, float* out_aux,
                                   instag_stream_t stream);
/* backward.  dL_dout_* may be NULL (treated as zero).  Gradient outputs may be NULL when not
 * needed; non-NULL ones are fully written (not accumulated).  dL_dmeans2D is [N,3]
 * (x,y in NDC units = pixel gradient * 0.5*(W,H); z = 0), the quantity
 * scene/gaussian_model.py:684 consumes.
 * aux_colors [N,3] + dL_dout_aux [3,H,W] (both or neither): the auxiliary image of the forward call is differentiated
 * too -- dL_daux_colors [N,3] is written and the aux image's share of the screen-space mean gradient is ADDED into
 * dL_dmeans2D (only there: the reference renders the attention map from detached geometry, gaussian_renderer/
 * __init__.py:256-268).  With an rgb-only main pass (no depth / normal / extra gradient) this costs no extra launch:
 * the blend kernel carries both images through one alpha / T recurrence.
 * aux_colors_only != 0 (rgb-only main pass): only dL_daux_colors is produced here -- three more feature columns of a
 * matrix product that has idle ones, i.e. free -- and dL_dmeans2D gets no aux share: the caller runs
 * instag_raster_aux_backward(dL_daux_colors = NULL) beside this call for that (the two launches overlap; at 512x512
 * that is faster than one launch that walks every tile's list with 1.3x the work per entry). */
int instag_raster_backward(const instag_raster_args* a, const void* geom, size_t geom_bytes,
                           const void* binning, size_t binning_bytes, const void* image,
                           size_t image_bytes, int64_t R, const int32_t* radii,
                           const float* dL_dout_color, const float* dL_dout_depth,
                           const float* dL_dout_normal, const float* dL_dout_alpha,
                           const float* dL_dout_extra, void* workspace, size_t workspace_bytes,
                           float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                           float* dL_dcolors_precomp, float* dL_dopacities, float* dL_dscales,
                           float* dL_drotations, float* dL_dcov3Ds_precomp, float* dL_dextra_attrs,
                           float* dL_dshs_rest /* [N,M-1,3], with split SH storage: then dL_dshs is [N,1,3] */,
                           const float* aux_colors, const float* dL_dout_aux, float* dL_daux_colors,
                           int32_t aux_colors_only, instag_stream_t stream);

/* backward of the auxiliary colour set over the forward's state: dL_dout_aux [3,H,W] -> dL_daux_colors [N,3] and
 * the aux image's contribution to dL_dmeans2D [N,3] (same convention as above; either may be NULL).  The geometry
 * and the opacities receive NO gradient from the aux image (the reference detaches them for that pass).
 * workspace: instag_raster_backward_workspace_bytes(N, R) bytes, distinct from a concurrently running
 * instag_raster_backward's. */
int instag_raster_aux_backward(const instag_raster_args* a, const void* geom, size_t geom_bytes,
                               const void* binning, size_t binning_bytes, const void* image,
                               size_t image_bytes, int64_t R, const int32_t* radii, const float* aux_colors,
                               const float* dL_dout_aux, void* workspace, size_t workspace_bytes,
                               float* dL_daux_colors, float* dL_dmeans2D, instag_stream_t stream);

/* Debug / test access to the integer state (bit-exact parity checks against the oracle).
 * Copies are enqueued on `stream`; destination pointers are DEVICE memory. */
int instag_raster_debug_export(const void* geom, size_t geom_bytes, const void* binning,
                               size_t binning_bytes, const void* image, size_t image_bytes,
                               int32_t N, int64_t R, int32_t image_height, int32_t image_width,
                               uint32_t* tiles_touched /*[N]*/, uint32_t* point_offsets /*[N]*/,
                               uint64_t* keys_sorted /*[R]*/, uint32_t* point_list /*[R]*/,
                               int32_t* ranges /*[tiles,2]*/, uint32_t* n_contrib /*[H*W]*/,
                               float* final_T /*[H*W]*/, float* rec2d /*[N,16]*/,
                               instag_stream_t stream);

/* Per-Gaussian flag word of the forward state (device copy on `stream`): bits 0-2 SH clamp, 3-4 normal axis, 5 normal
 * sign, 6-7 cov2D clamp, bits 16-31 = HEIGHT of the Gaussian's tile rectangle.  Together with rec2d[15] (min x |
 * min y << 10 | width << 20) this is the bounding rectangle of the published binning, compared bit for bit with the
 * oracle's cull="rect" mode.  Only entries of Gaussians with radii > 0 are defined. */
int instag_raster_debug_export_flags(const void* geom, size_t geom_bytes, int32_t N, uint32_t* flags /*[N]*/,
                                     instag_stream_t stream);

/* Diagnostics (scripts/bench_sort.py): the depth sort of the Gaussians alone, on `stream`.  stamps (device, may be
 * NULL): uint64[4 passes][instag_debug_depth_sort_blocks(N)][8], 100 MHz timestamps of every block's phases.
 * order_out (device, may be NULL): uint32[N], Gaussian indices in (depth bits, index) order. */
int instag_debug_depth_sort(const instag_raster_args* a, void* geom, size_t geom_bytes, uint64_t* stamps,
                            uint32_t* order_out, instag_stream_t stream);
uint32_t instag_debug_depth_sort_blocks(int32_t N);

/* The binning sorts and the instance-offset scan order their workgroups by decoupled look-back with a bounded spin.
 * A look-back that gives up continues with a wrong prefix -- a mis-sorted list, a wrong image -- so every such event
 * bumps a sticky device counter.  instag_raster_forward_stage1 reads it at its synchronisation point and fails with
 * INSTAG_E_HIP; capacity-mode callers (no synchronisation per step) read it here: *count = events since the last clear.
 * synchronize = 0: `count` must be pinned host memory, the copy is only enqueued on `stream`. */
int instag_raster_sort_stalls(uint32_t* count, int32_t synchronize, instag_stream_t stream);
int instag_raster_sort_stalls_clear(instag_stream_t stream);
/* Test hook: runs the offset scan over 8 items with its ticket counter pre-set so that the one workgroup looks back at
 * a predecessor that never publishes; returns after the bounded spin has expired (about a second). */
int instag_debug_scan_stall_probe(instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Bias-free ReLU MLP over N rows on the f32 matrix cores (exact fp32).
 * Replaces the per-Gaussian `MLP` modules of scene/motion_net.py:152-173 (sigma_net, aud_ch_att_net,
 * eye_att_net, align_net; instantiated at :234-238 and :600-604), i.e. for NL layers
 *     y = W_NL * relu(... relu(W_1 x))          W_l row-major [out_l, in_l] (torch.nn.Linear.weight)
 * x [N,K0], hidden width H, y [N,O]; NL in {2,3}; K0 <= 96, H <= 64, O <= 32.
 * forward optionally stores the post-ReLU hidden activations a1 [N,H] (and a2 [N,H] for NL==3) for
 * backward (pass NULL for inference).  backward consumes dy [N,O] and writes the pre-activation
 * gradients dz1 [N,H] (dz2 [N,H]) and, when dx != NULL, dx [N,K0]; the weight gradients are then
 *     dW_1 = dz1^T x,  dW_2 = dz2^T a1 (NL==3) or dy^T a1 (NL==2),  dW_3 = dy^T a2
 * each computed with instag_linear_weight_grad (dz [N,O], in [N,K] -> dw [O,K]; O <= 64, K <= 96;
 * workspace of instag_linear_weight_grad_workspace_bytes; deterministic).
 * ------------------------------------------------------------------------------------------ */
int instag_mlp_forward(const float* x, const float* w1, const float* w2, const float* w3, float* y,
                       float* a1, float* a2, int32_t N, int32_t K0, int32_t H, int32_t O, int32_t NL,
                       instag_stream_t stream);
int instag_mlp_backward(const float* dy, const float* a1, const float* a2, const float* w1,
                        const float* w2, const float* w3, float* dz1, float* dz2, float* dx, int32_t N,
                        int32_t K0, int32_t H, int32_t O, int32_t NL, instag_stream_t stream);
/* as instag_mlp_backward with dx = dX + dx_add ([N,K0], may be dx itself or NULL): when the MLP's input feeds other
 * consumers too (scene/motion_net.py:281-306: enc_x goes to both attention MLPs and into sigma_net's input), their
 * gradient is summed inside this kernel instead of by an elementwise launch in between */
int instag_mlp_backward_add(const float* dy, const float* a1, const float* a2, const float* w1,
                            const float* w2, const float* w3, float* dz1, float* dz2, float* dx,
                            const float* dx_add, int32_t N, int32_t K0, int32_t H, int32_t O, int32_t NL,
                            instag_stream_t stream);
/* sigma_net's backward with the glue operator's backward (instag_motion_glue_backward) as its epilogue, for the universal
 * field's widths (K0 = 36 + 32 + 6 = enc_x | enc_a * aud | enc_e * relu(eye_pre), scene/motion_net.py:291-306): the
 * [N,74] input gradient is never stored; d_enc_x, d_aud, d_eye_pre leave from the accumulator registers and the per-frame
 * vectors' column sums as one row [KA + KE] per workgroup (num_partials rows, added up in order by the caller).
 * d_amb [N,3] may be NULL. */
int instag_mlp_backward_glue_supported(int32_t K0, int32_t H, int32_t O, int32_t KX, int32_t KA, int32_t KE);
int instag_mlp_backward_glue_num_partials(int32_t N);
int instag_mlp_backward_glue(const float* dy, const float* a1, const float* a2, const float* w1, const float* w2,
                             const float* w3, float* dz1, float* dz2, const float* aud, const float* eye_pre,
                             const float* enc_a, const float* enc_e, const float* amb, const float* d_amb,
                             float* d_enc_x, float* d_aud, float* d_eye_pre, float* col_partials, int32_t N,
                             int32_t H, int32_t O, instag_stream_t stream);
/* ... and its forward with the glue operator's forward (instag_motion_glue_forward) in front of the first layer: the
 * input rows are formed in registers; h_in [N,74] (read by the first layer's weight gradient) and amb [N,3] are still
 * written once.  a1 / a2 may be NULL (no gradient wanted). */
int instag_mlp_forward_glue(const float* enc_x, const float* aud, const float* eye_pre, const float* enc_a,
                            const float* enc_e, const float* w1, const float* w2, const float* w3, float* y, float* a1,
                            float* a2, float* h_in, float* amb, int32_t N, int32_t H, int32_t O,
                            instag_stream_t stream);
/* Two 2-layer MLPs over the SAME input x in one launch: (MLP_a(x), MLP_b(x)) -- the universal field's aud_ch_att_net and
 * eye_att_net both read the tri-plane features (scene/motion_net.py:281-290).  backward: dx = W_a1^T dz1a + W_b1^T dz1b
 * (+ dx_add, which may alias dx).  instag_mlp2_supported: 1 when the shape pair has a kernel (36 -> 32 -> 32 with
 * 36 -> 16 -> 6, in groups of eight features); otherwise launch the heads one by one. */
int instag_mlp2_supported(int32_t K0, int32_t HA, int32_t OA, int32_t HB, int32_t OB);
int instag_mlp2_forward(const float* x, const float* wa1, const float* wa2, const float* wb1, const float* wb2,
                        float* ya, float* yb, float* a1a, float* a1b, int32_t N, int32_t K0, int32_t HA, int32_t OA,
                        int32_t HB, int32_t OB, instag_stream_t stream);
int instag_mlp2_backward(const float* dya, const float* dyb, const float* a1a, const float* a1b, const float* wa1,
                         const float* wa2, const float* wb1, const float* wb2, float* dz1a, float* dz1b, float* dx,
                         const float* dx_add, int32_t N, int32_t K0, int32_t HA, int32_t OA, int32_t HB, int32_t OB,
                         instag_stream_t stream);
size_t instag_linear_weight_grad_workspace_bytes(int32_t N, int32_t O, int32_t K);
int instag_linear_weight_grad(const float* dz, const float* in, float* dw, void* workspace,
                              size_t workspace_bytes, int32_t N, int32_t O, int32_t K,
                              instag_stream_t stream);
/* Several weight gradients in one launch (all with the same N): workspace = sum over jobs of
 * instag_linear_weight_grad_workspace_bytes rounded up to 256 B.  At most 16 jobs. */
typedef struct {
  const float* dz; /* [N,O] */
  const float* in; /* [N,K] */
  float* dw;       /* [O,K] (written) */
  int32_t N, O, K;
} instag_wgrad_job;
int instag_linear_weight_grad_batched(const instag_wgrad_job* jobs, int32_t n_jobs, void* workspace,
                                      size_t workspace_bytes, instag_stream_t stream);
/* The same with ONE job (index glue_job) whose input rows are not stored: row r of its `in` is
 * cat(in[r] (KX = K - KA - KE values), aud[r] * enc_a (KA), relu(eye_pre[r]) * enc_e (KE)) -- sigma_net's input as the glue
 * forms it (scene/motion_net.py:291-306).  instag_mlp_forward_glue then takes h_in = NULL and does not write those rows
 * (30 MB at 100k Gaussians).  65 <= K <= 96. */
int instag_linear_weight_grad_batched_glue(const instag_wgrad_job* jobs, int32_t n_jobs, int32_t glue_job,
                                           const float* aud, const float* eye_pre, const float* enc_a,
                                           const float* enc_e, int32_t KA, int32_t KE, void* workspace,
                                           size_t workspace_bytes, instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused per-Gaussian glue (csrc/glue.hip).
 *  motion_glue  (scene/motion_net.py:291-306, :679-692): h_in [N,KX+KA+KE] = cat(enc_x, enc_a*aud, enc_e*relu(eye_pre)),
 *               amb [N,3] = (||aud||, ||relu(eye_pre)||, 0); backward returns d_enc_x, d_aud, d_eye_pre and
 *               col_partials [instag_motion_glue_backward_num_partials][KA+KE]: per-workgroup partial sums of
 *               (d_enc_a | d_enc_e), to be summed over rows by the caller in a fixed order.  KA <= 32, KE <= 8,
 *               KX+KA+KE <= 256.
 *  deform_activate (gaussian_renderer/__init__.py:200-235, personalized=False, align=True): h [N,11] = UMF head output,
 *               p [N,6] = PMF align head output -> means3D, scales, rotations, opacity.
 *  motion_l1_reg (train_face.py:510-514): mean|h[:, :3]*1e-2| + mean|h[:,3:7]| + mean|h[:,7:8]| + mean|h[:,8:11]| +
 *               mean|p[:, :3]*1e-2| as per-workgroup partial sums; backward takes the upstream scalar gradient g (device).
 * ------------------------------------------------------------------------------------------ */
int instag_motion_glue_forward(const float* enc_x, const float* aud, const float* eye_pre, const float* enc_a,
                               const float* enc_e, float* h_in, float* amb, int32_t N, int32_t KX, int32_t KA,
                               int32_t KE, instag_stream_t stream);
int instag_motion_glue_backward_num_partials(int32_t N, int32_t KX, int32_t KA, int32_t KE);
int instag_motion_glue_backward(const float* d_h_in, const float* d_amb, const float* aud, const float* eye_pre,
                                const float* enc_a, const float* enc_e, const float* amb, float* d_enc_x,
                                float* d_aud, float* d_eye_pre, float* col_partials, int32_t N,
                                int32_t KX, int32_t KA, int32_t KE, instag_stream_t stream);
/* reg_partials (may be NULL): instag_deform_activate_num_reg_partials(N) per-workgroup partial sums of
 * reg_weight * motion_l1_reg(h, p) (same terms as instag_motion_l1_reg_*), to be added up by the consumer, e.g. as the
 * `extra` array of instag_face_loss_forward.  backward: g_reg (device scalar, may be NULL) is the upstream gradient of
 * that sum; its sign terms are added into d_h / d_p. */
int instag_deform_activate_num_reg_partials(int32_t N);
int instag_deform_activate_forward(const float* xyz, const float* scaling, const float* rotation,
                                   const float* opacity, const float* h, const float* p, float* means3D,
                                   float* scales, float* rotations, float* opac, float* reg_partials,
                                   float reg_weight, int32_t N, instag_stream_t stream);
int instag_deform_activate_backward(const float* scaling, const float* rotation, const float* opacity,
                                    const float* h, const float* p, const float* g_means, const float* g_scales,
                                    const float* g_rots, const float* g_opac, float* d_xyz, float* d_scaling,
                                    float* d_rotation, float* d_opacity, float* d_h, float* d_p, const float* g_reg,
                                    float reg_weight, int32_t N, instag_stream_t stream);
/* mouth_activate (gaussian_renderer/__init__.py:404-420 with scene/motion_net.py:446-452): h [N,7] = the mouth field's
 * sigma_net output, hs [N,1] = its scaler_net output, (sx, sy, sz) = the per-axis displacement scale (1e-2/5, 1e-2,
 * 1e-2/5) -> means3D = xyz + ((h[:, :3] * s) * sigmoid(hs)) * 2, scales = softplus(scaling), rotations =
 * normalize(rotation), opacity = sigmoid(opacity).  backward writes d_h[:, 3:7] = 0 (the mouth render does not apply
 * the predicted rotation). */
int instag_mouth_activate_forward(const float* xyz, const float* scaling, const float* rotation, const float* opacity,
                                  const float* h, const float* hs, float sx, float sy, float sz, float* means3D,
                                  float* scales, float* rotations, float* opac, int32_t N, instag_stream_t stream);
int instag_mouth_activate_backward(const float* scaling, const float* rotation, const float* opacity, const float* h,
                                   const float* hs, float sx, float sy, float sz, const float* g_means,
                                   const float* g_scales, const float* g_rots, const float* g_opac, float* d_xyz,
                                   float* d_scaling, float* d_rotation, float* d_opacity, float* d_h, float* d_hs,
                                   int32_t N, instag_stream_t stream);
/* abs_mean (train_mouth.py:203, `p_xyz.abs().mean()`): mean |x[:, :ncols] * scale| over the N rows of x [N, stride] as
 * instag_abs_mean_num_partials(N) per-workgroup partial sums (to be added up by the consumer, e.g. the `extra` array of
 * instag_face_loss_forward); backward: g = upstream gradient of that sum (device scalar) -> dx [N, stride], zero in the
 * columns beyond ncols. */
int instag_abs_mean_num_partials(int32_t N);
int instag_abs_mean_forward(const float* x, int32_t N, int32_t stride, int32_t ncols, float scale, float* partials,
                            instag_stream_t stream);
int instag_abs_mean_backward(const float* x, const float* g, int32_t N, int32_t stride, int32_t ncols, float scale,
                             float* dx, instag_stream_t stream);
/* mouth_glue (scene/motion_net.py:437-444): in_sigma [N, KX+KA+KM] = [enc_x | enc_a | move], in_scaler [N, KX+KM] =
 * [enc_x | move]; enc_a [KA <= 32] and move [KM] are per-frame vectors.  backward: d_enc_x [N,KX] = the enc_x columns
 * of both gradients added (either may be NULL), col_partials [instag_mouth_glue_backward_num_partials(N)][KA] =
 * per-workgroup column sums of d_in_sigma's enc_a block, to be summed by the caller in order. */
int instag_mouth_glue_forward(const float* enc_x, const float* enc_a, const float* move, float* in_sigma,
                              float* in_scaler, int32_t N, int32_t KX, int32_t KA, int32_t KM, instag_stream_t stream);
int instag_mouth_glue_backward_num_partials(int32_t N);
int instag_mouth_glue_backward(const float* d_sigma, const float* d_scaler, float* d_enc_x, float* col_partials,
                               int32_t N, int32_t KX, int32_t KA, int32_t KM, instag_stream_t stream);
/* fuse_compose (train_fuse_con.py:102-121): face, mouth [3,H,W] rendered over bg [3], a_face, a_mouth [1,H,W], scene
 * [3,H,W] or NULL (= black) -> mouth_image = mouth - bg (1 - a_mouth) + scene (1 - a_mouth), image = face - bg (1 - a_face)
 * + mouth_image (1 - a_face).  backward: g_image / g_mouth_image (either may be NULL) -> gradients of the four inputs. */
int instag_fuse_compose_forward(const float* face, const float* a_face, const float* mouth, const float* a_mouth,
                                const float* bg, const float* scene, float* image, float* mouth_image, int32_t H,
                                int32_t W, instag_stream_t stream);
int instag_fuse_compose_backward(const float* g_image, const float* g_mouth_image, const float* a_face,
                                 const float* mouth_image, const float* bg, const float* scene, float* d_face,
                                 float* d_a_face, float* d_mouth, float* d_a_mouth, int32_t H, int32_t W,
                                 instag_stream_t stream);
int instag_motion_l1_reg_num_partials(int32_t N);
int instag_motion_l1_reg_forward(const float* h, const float* p, float* partial, int32_t N,
                                 instag_stream_t stream);
int instag_motion_l1_reg_backward(const float* h, const float* p, const float* g, float* d_h, float* d_p,
                                  int32_t N, instag_stream_t stream);
/* Densification statistics of one step (train_face.py:626-629, GaussianModel.add_densification_stats), in place,
 * for the Gaussians with radii > 0: max_radii2D = max(max_radii2D, radii), grad_accum += ||viewspace_grad[:, :2]||,
 * denom += 1.  viewspace_grad [N,3], radii int32 [N], the three statistics float [N]. */
int instag_densify_stats(const float* viewspace_grad, const int32_t* radii, float* max_radii2D, float* grad_accum,
                         float* denom, int32_t N, instag_stream_t stream);
/* As instag_densify_stats with a second producer's share of the screen-space gradient: viewspace_grad += grad_add
 * ([N,3], may be NULL) for every Gaussian first (the sum is written back), then the statistics from the sum. */
int instag_densify_stats_add(float* viewspace_grad, const float* grad_add, const int32_t* radii, float* max_radii2D,
                             float* grad_accum, float* denom, int32_t N, instag_stream_t stream);

/* The k largest (descending) and k smallest (ascending) VALUES of v [N], 1 <= k <= 64 (csrc/select.hip): the two
 * torch.topk selections behind the mouth branch's jaw-movement feature, gaussian_renderer/__init__.py:341-349. */
size_t instag_extreme_values_workspace_bytes(int32_t N, int32_t k);
int instag_extreme_values(const float* v, int32_t N, int32_t k, float* largest, float* smallest, void* workspace,
                          size_t workspace_bytes, instag_stream_t stream);
/* The mouth field's jaw-movement feature (gaussian_renderer/__init__.py:341-349): out[3] = [max, min, max - min] * 1e2,
 * max / min = scale * the k-th largest / smallest of v[i * stride + offset], i < N (a column of the face field's head
 * output).  kmax <= min(64, N) candidates per side; k (1-based, clamped to [1, kmax]) is read from k_dev (int64 on the
 * device) or, when that is NULL, k_host.  workspace: instag_jaw_feature_workspace_bytes(N, kmax). */
size_t instag_jaw_feature_workspace_bytes(int32_t N, int32_t kmax);
int instag_jaw_feature(const float* v, int32_t N, int32_t stride, int32_t offset, float scale, int32_t kmax,
                       const int64_t* k_dev, int32_t k_host, float* out, void* workspace, size_t workspace_bytes,
                       instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * `simple_knn._C.distCUDA2` provider (scene/gaussian_model.py:20,246; the package itself is an absent third-party
 * submodule): out[i] = mean of the squared distances from point i to its three nearest OTHER points
 * (over min(3, N-1) neighbours when the cloud is smaller; 0 for a single point).  points [N,3], out [N].
 * ------------------------------------------------------------------------------------------ */
int instag_knn3_mean_dist2(const float* points, float* out, int32_t N, instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-frame conditioning codes of one motion network, one workgroup per pass (csrc/audio.hip):
 *   enc_a [dim_aud] = AudioAttNet(AudioNet(a))             scene/motion_net.py:29-64, :67-99
 *   enc_e [6]       = cat(exp_encode_net(e[:5]), e[5:6])   scene/motion_net.py:152-173 (MLP 5->16->5, bias-free)
 * a [8, dim_in, 16] (8 audio windows of 16 samples), e [6] or NULL (then enc_e NULL: network without the
 * expression branch).  `params` / `grads` are HOST arrays of 26 DEVICE pointers in state_dict order:
 *   0-7   audio_net.encoder_conv.{0,2,4,6}.{weight,bias}   conv1d k3 s2 p1: dim_in->mid->mid->64->64
 *   8-11  audio_net.encoder_fc1.{0,2}.{weight,bias}        64->64 (LeakyReLU 0.02) ->dim_aud
 *   12-21 audio_att_net.attentionConvNet.{0,2,4,6,8}.{weight,bias}   conv1d k3 s1 p1: dim_aud->16->8->4->2->1
 *   22-23 audio_att_net.attentionNet.0.{weight,bias}       linear 8->8 (+ softmax)
 *   24-25 exp_encode_net.net.{0,1}.weight                  [16,5], [5,16] (may be NULL when e is NULL)
 * forward keeps every activation in `saved` (instag_frame_code_saved_floats floats); backward OVERWRITES
 * every grads[i] (same shapes as params[i]); no gradient flows to a or e.  Deterministic (no atomics).
 * arrivals (device uint32, ZERO before the first call, owned by ONE network: calls that may overlap on different
 * streams need words of their own): the forward runs as one workgroup per audio window, the last one to arrive runs
 * the attention stage and leaves the word zero again.  NULL selects the single-workgroup form.
 * ------------------------------------------------------------------------------------------ */
int64_t instag_frame_code_saved_floats(int32_t dim_in, int32_t mid, int32_t dim_aud);
int instag_frame_code_forward(const float* a, const float* e, const float* const* params, float* enc_a,
                              float* enc_e, float* saved, int32_t dim_in, int32_t mid, int32_t dim_aud,
                              uint32_t* arrivals, instag_stream_t stream);
/* backward: with a workspace of instag_frame_code_backward_workspace_bytes() the eight audio windows run on eight
 * workgroups (each writes a row of parameter gradients there, a second launch adds the rows up in a fixed order);
 * workspace NULL selects the single-workgroup form. */
size_t instag_frame_code_backward_workspace_bytes(int32_t dim_in, int32_t mid, int32_t dim_aud);
int instag_frame_code_backward(const float* a, const float* e, const float* const* params, const float* saved,
                               const float* d_enc_a, const float* d_enc_e, float* const* grads, int32_t dim_in,
                               int32_t mid, int32_t dim_aud, void* workspace, size_t workspace_bytes,
                               instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused L1 + SSIM image loss.  Replaces utils/loss_utils.py l1_loss :26-27 and ssim :42-72 (11x11
 * Gaussian window sigma 1.5, zero padding, C1=0.01^2, C2=0.03^2, mean over C*H*W) as used at
 * train_face.py:450-456.  img1/img2 [C,H,W].  forward writes per-workgroup partial sums
 * (instag_l1_ssim_num_partials of them each; l1 = sum(partial_l1)/(C*H*W), ssim likewise) and the
 * derivative maps [3,C,H,W]; backward writes d(loss)/d(img1) given the upstream gradients of the two
 * scalar means (device scalars, NULL = 0).
 * ------------------------------------------------------------------------------------------ */
int instag_l1_ssim_num_partials(int32_t C, int32_t H, int32_t W);
int instag_l1_ssim_forward(const float* img1, const float* img2, int32_t C, int32_t H, int32_t W,
                           float* maps, float* partial_ssim, float* partial_l1, instag_stream_t stream);
int instag_l1_ssim_backward(const float* img1, const float* img2, const float* maps, const float* g_ssim,
                            const float* g_l1, int32_t C, int32_t H, int32_t W, float* dimg1,
                            instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused loss block of the face branch (csrc/ssim.hip), train_face.py:415-416 (gt_white), :426-456
 * (hair / mouth masking, L1 + DSSIM), :508-575 (alpha and attention terms):
 *   head = face | hair;  gt_white = (head & ~mouth) ? gt : bg;  under FLAG_HAIR_TO_BG (hair_mask_iter)
 *   hair pixels of BOTH images become bg (no gradient there)
 *   loss = L1 + w_dssim (1 - SSIM)
 *        + w_alpha (mean((1-alpha) head) + mean(alpha ~head))                      [FLAG_ALPHA]
 *        + w_attn_hair (mean(attn[1][hair]) + mean(attn[0][hair]))                 [FLAG_HAIR_ATTN]
 *        + w_attn_lips mean(attn[1, r0:r1, c0:c1])   lips_rect = (r0, r1, c0, c1)  [FLAG_LIPS]
 *        + w_extra * sum(extra[0:n_extra])           (extra: optional device array of partial sums, e.g. the
 *                                                     regulariser partials of instag_deform_activate_forward)
 * image, gt, attn [3,H,W]; alpha [1,H,W]; masks [H,W] bytes (0 / non-zero); bg [3]; lips_rect int32[4] on the
 * device (it changes per frame under graph replay).  An empty hair mask contributes 0 (the reference
 * yields NaN there).  forward writes maps [3,3,H,W] (SSIM derivative maps), partials
 * (instag_face_loss_num_partials floats) and out[5] = (loss, L1, SSIM, 1/#hair, 1/lips area).
 * backward takes the upstream gradients of loss and of the separately returned L1 (device scalars, either
 * may be NULL) and writes d_image [3,H,W], d_alpha [1,H,W] (may be NULL), d_attn [3,H,W] (may be NULL);
 * d loss / d extra = w_extra * g_loss is left to the caller.
 * ------------------------------------------------------------------------------------------ */
#define INSTAG_FACE_LOSS_HAIR_TO_BG 1
#define INSTAG_FACE_LOSS_ALPHA 2
#define INSTAG_FACE_LOSS_HAIR_ATTN 4
#define INSTAG_FACE_LOSS_LIPS 8
/* the mouth branch's loss block instead (train_mouth.py:186-221): image_green = (lips ^ mouth) ? bg : image,
 * gt_green = mouth ? gt : bg, the alpha terms (INSTAG_FACE_LOSS_ALPHA) over the lips rectangle = rows
 * [lips_rect[0], lips_rect[1]) x columns [lips_rect[2], lips_rect[3]); face_mask / hair_mask may be NULL, lips_rect is
 * required, none of the three face-branch flags may be set */
#define INSTAG_FACE_LOSS_MOUTH 16
#define INSTAG_FACE_LOSS_PLAIN 32     /* whole-frame L1 + DSSIM of image against gt (train_fuse_con.py:176-181): no masks, bg or rectangle */
typedef struct {
  int32_t H, W, flags;
  float w_dssim, w_alpha, w_attn_hair, w_attn_lips, w_extra;
} instag_face_loss_cfg;
int64_t instag_face_loss_num_partials(int32_t H, int32_t W);
int instag_face_loss_forward(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                             const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                             const float* bg, const float* alpha, const float* attn, const int32_t* lips_rect,
                             const float* extra, int32_t n_extra, float* maps, float* partials, float* out,
                             instag_stream_t stream);
int instag_face_loss_backward(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                              const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                              const float* bg, const int32_t* lips_rect, const float* maps, const float* out,
                              const float* g_loss, const float* g_l1, float* d_image, float* d_alpha, float* d_attn,
                              instag_stream_t stream);
/* The same pair with the scalar stage DEFERRED into the backward launch: the forward writes `maps` and `partials` only
 * (one launch), the backward's first workgroup adds the partial sums up (the same additions in the same order as the
 * forward's scalar stage: same bits) and writes out[5]; the other workgroups derive 1 / #hair pixels and 1 / lips area,
 * the only scalars the gradients use, themselves.  For callers that run the backward right behind the forward and read
 * the loss value afterwards (a train step): one launch less on the step's critical chain.  `extra` / `n_extra` as given
 * to the forward. */
int instag_face_loss_forward_deferred(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                                      const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                                      const float* bg, const float* alpha, const float* attn, const int32_t* lips_rect,
                                      const float* extra, int32_t n_extra, float* maps, float* partials,
                                      instag_stream_t stream);
int instag_face_loss_backward_deferred(const instag_face_loss_cfg* cfg, const float* image, const float* gt,
                                       const uint8_t* face_mask, const uint8_t* hair_mask, const uint8_t* mouth_mask,
                                       const float* bg, const int32_t* lips_rect, const float* maps,
                                       const float* partials, const float* extra, int32_t n_extra, float* out,
                                       const float* g_loss, const float* g_l1, float* d_image, float* d_alpha,
                                       float* d_attn, instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Monocular geometry priors of the face branch (csrc/prior.hip); replaces train_face.py:458-504 with
 * utils/loss_utils.py:17-20 (normalize):
 *   loss = w_normal * mean_{(face|hair) ^ mouth} sum_c (1 - gt_normal[c] * normal[c])
 *        + w_depth  * mean_{face ^ mouth} | normalize(depth) - normalize(gt_depth) |        (when use_depth)
 * normal / gt_normal [3,H,W], depth / gt_depth [H,W], masks uint8 [H,W].  Workspaces (floats): stat 8H, parts 4H,
 * rowb 2H; out[5] = [loss, S_depth, N_sel, S_normal, N_m].  backward: g_loss = device scalar. */
int instag_geometry_prior_forward(const float* normal, const float* depth, const float* gt_normal,
                                  const float* gt_depth, const uint8_t* face_mask, const uint8_t* hair_mask,
                                  const uint8_t* mouth_mask, int32_t H, int32_t W, int32_t use_depth, float w_normal,
                                  float w_depth, float* stat, float* parts, float* out, instag_stream_t stream);
int instag_geometry_prior_backward(const float* g_loss, const float* depth, const float* gt_normal,
                                   const float* gt_depth, const uint8_t* face_mask, const uint8_t* hair_mask,
                                   const uint8_t* mouth_mask, int32_t H, int32_t W, int32_t use_depth, float w_normal,
                                   float w_depth, const float* stat, const float* out, float* rowb, float* d_normal,
                                   float* d_depth, instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multi-tensor Adam / AdamW in one launch (csrc/adam.hip); replaces motion_optimizer.step() and
 * gaussians.optimizer.step() of train_face.py:781-788.  tensors: device array of 48-byte records
 * {float* p, const float* g (NULL = skip), float* m, float* v, int64 n, int32 group, int32 pad}; groups: device array
 * of 24-byte records {beta1, beta2, eps, weight_decay, int32 decoupled (1 = AdamW), int32 pad}; lrs: device
 * float[n_groups]; chunks: device int32[n_chunks][2] = (tensor index, chunk index), instag_adam_chunk_elems()
 * elements per chunk; step: device float[n_tensors], per-tensor step counters, incremented by the call (for tensors with a
 * gradient) before use.
 * ------------------------------------------------------------------------------------------ */
int instag_adam_chunk_elems(void);
/* the same with the gradient pointers as a HOST array (uint64[n_tensors], 0 = no gradient this step; n_tensors <=
 * instag_adam_grads_max()): they travel in the kernel arguments and the `g` field of the device records is ignored, so
 * no copy precedes the launch and the device table is uploaded only when the parameter set changes */
int instag_adam_grads_max(void);
int instag_adam_step_grads(const void* tensors, const void* host_grads, int32_t n_tensors, const void* groups,
                           const float* lrs, const int32_t* chunks, int32_t n_chunks, float* step,
                           instag_stream_t stream);
/* The same in ONE launch (no counter launch in front): tickets = int32[n_tensors] in device memory, zero before the first
 * call and left zero by every call; the workgroup of each tensor that finishes last stores the tensor's new step count.
 * chunks must list every chunk of every tensor it names exactly once (a step may be made of several calls over
 * disjoint sets of tensors). */
int instag_adam_step_grads_ticketed(const void* tensors, const void* host_grads, int32_t n_tensors, const void* groups,
                                    const float* lrs, const int32_t* chunks, int32_t n_chunks, float* step,
                                    int32_t* tickets, instag_stream_t stream);
int instag_adam_step(const void* tensors, int32_t n_tensors, const void* groups, const float* lrs,
                     const int32_t* chunks, int32_t n_chunks, float* step, instag_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing (bench.py roofline leg).  When enabled, the launcher brackets the named
 * kernel with hipEvents on the launch stream; instag_prof_read synchronises those events and
 * returns accumulated milliseconds and launch count since the last reset.
 * Kernel ids: 0 preprocess, 1 duplicate, 2 sort, 3 ranges, 4 blend_fwd, 5 blend_bwd,
 *             6 preprocess_bwd, 7 grid_fwd, 8 grid_bwd, 9 sh_fwd, 10 sh_bwd,
 *             11 mlp_fwd, 12 mlp_bwd, 13 mlp_weight_grad, 14 loss_fwd, 15 loss_bwd,
 *             16 blend_bwd mean-only launch (the auxiliary image's d/dmean pass; 5 = every other blend_bwd launch),
 *             17 an EMPTY bracket (recorded in front of every blend_fwd launch): the time between two event records with
 *                nothing between them, i.e. what the bracket adds to every kernel's figure.
 *
 * Inside a hipGraph: a launch issued while its stream is being CAPTURED is bracketed with EXTERNAL event-record nodes
 * (hipEventRecordWithFlags(hipEventRecordExternal)) taken from a pool the caller sized with instag_prof_graph_begin --
 * events cannot be created while a capture is open.  Every replay of the captured graph re-records them;
 * instag_prof_graph_collect (after the replay has been synchronised) adds each pair's elapsed time to its kernel's
 * totals, so the durations are those of the kernels as they run in the replayed graph, next to their concurrent
 * branches.  instag_prof_graph_end destroys the pool (the graphs that captured its events must be gone by then).
 * ------------------------------------------------------------------------------------------ */
#define INSTAG_PROF_KERNELS 18
int instag_prof_enable(int kernel_mask_or_minus1);
int instag_prof_reset(void);
int instag_prof_read(int kernel_id, double* total_ms /* (host) */, int64_t* launches /* (host) */);
int instag_prof_graph_begin(int32_t max_pairs);
int instag_prof_graph_pairs_used(void);
int instag_prof_graph_collect(void);
int instag_prof_graph_end(void);

#ifdef __cplusplus
}
#endif
#endif /* INSTAG_HIP_H */
