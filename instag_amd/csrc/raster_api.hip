// C-ABI entry points of the rasterizer (see include/instag_hip.h) + buffer layouts, scan and sort.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <mutex>
#include <vector>

#include "raster_internal.hpp"

namespace instag {

// ---- thread-local error string ------------------------------------------------------------------
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

// ---- per-kernel event timing ------------------------------------------------------------------------
namespace {
struct ProfState {
  std::mutex mu;
  int mask = 0;
  struct Pair { hipEvent_t a, b; };
  std::vector<Pair> pending[INSTAG_PROF_KERNELS];
  double total_ms[INSTAG_PROF_KERNELS] = {0};
  int64_t launches[INSTAG_PROF_KERNELS] = {0};
  // pool for launches that are being captured into a graph (instag_prof_graph_*)
  struct GraphPair { hipEvent_t a = nullptr, b = nullptr; int kernel = -1; bool closed = false; };
  std::vector<GraphPair> graph_pool;
  int graph_used = 0;
};
ProfState& prof() { static ProfState p; return p; }
}  // namespace

// An event-record NODE in the graph `stream` is being captured into (the event is re-recorded by every replay).
// hipEventRecordWithFlags(hipEventRecordExternal) is the direct way; under a capture torch opened it returns
// hipErrorInvalidValue on ROCm 7.0/7.2 (a plain HIP program gets hipSuccess: scripts/probes/event_node_probe.hip), so the
// node is otherwise added by hand: capture info -> hipGraphAddEventRecordNode behind the stream's current dependencies
// -> the node becomes the stream's dependency set.
static hipError_t record_external(hipEvent_t ev, hipStream_t stream) {
  static bool direct_ok = true;
  if (direct_ok) {
    if (hipEventRecordWithFlags(ev, stream, hipEventRecordExternal) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    direct_ok = false;
  }
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  hipError_t e = hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, &deps, &ndeps);
  if (e != hipSuccess) return e;
  if (st != hipStreamCaptureStatusActive || graph == nullptr) return hipErrorIllegalState;
  hipGraphNode_t node = nullptr;
  e = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev);
  if (e != hipSuccess) return e;
  return hipStreamUpdateCaptureDependencies(stream, &node, 1, hipStreamSetCaptureDependencies);
}

ProfScope::ProfScope(int kernel, hipStream_t stream) : kernel_(kernel), stream_(stream) {
  ProfState& p = prof();
  if (!(p.mask & (1 << kernel))) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const hipError_t qe = hipStreamIsCapturing(stream_, &cs);
  static const bool debug = getenv("INSTAG_PROF_DEBUG") != nullptr;
  if (debug)
    fprintf(stderr, "[instag prof] kernel %d stream %p: hipStreamIsCapturing -> %s, status %d, pool %zu used %d\n", kernel,
            (void*)stream_, hipGetErrorString(qe), (int)cs, p.graph_pool.size(), p.graph_used);
  if (qe != hipSuccess) { (void)hipGetLastError(); return; }
  if (cs != hipStreamCaptureStatusNone) {
    // captured launch: external event-record nodes from the pool (no event may be created while a capture is open)
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.graph_used >= (int)p.graph_pool.size()) return;
    ProfState::GraphPair& g = p.graph_pool[p.graph_used];
    const hipError_t err = record_external(g.a, stream_);
    if (err != hipSuccess) {
      fprintf(stderr, "[instag prof] external event record (start, kernel %d, slot %d, stream %p) failed: %s\n", kernel,
              p.graph_used, (void*)stream_, hipGetErrorString(err));
      (void)hipGetLastError();
      return;
    }
    g.kernel = kernel;
    graph_slot_ = p.graph_used++;
    return;
  }
  if (hipEventCreate(&start_) != hipSuccess) { start_ = nullptr; return; }
  (void)hipEventRecord(start_, stream_);
}
ProfScope::~ProfScope() {
  if (graph_slot_ >= 0) {
    ProfState& p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    ProfState::GraphPair& g = p.graph_pool[graph_slot_];
    const hipError_t err = record_external(g.b, stream_);
    g.closed = err == hipSuccess;
    if (!g.closed) {
      fprintf(stderr, "[instag prof] external event record (stop, kernel %d, slot %d, stream %p) failed: %s\n", kernel_,
              graph_slot_, (void*)stream_, hipGetErrorString(err));
      (void)hipGetLastError();
    }
    return;
  }
  if (!start_) return;
  hipEvent_t stop;
  if (hipEventCreate(&stop) != hipSuccess) { (void)hipEventDestroy(start_); return; }
  (void)hipEventRecord(stop, stream_);
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  p.pending[kernel_].push_back({start_, stop});
}

// ---- second stream for the depth sort ------------------------------------------------------------------------------
// The depth sort needs means3D and the view matrix only, so it runs beside the preprocess kernel: forked from the
// caller's stream with an event, joined with another.  Stream and events belong to the calling THREAD (forward runs on
// the caller's thread, backward on autograd's: no sharing, no lock) and are created on a call that is not being captured
// into a graph (stream / event creation is not capturable); until then the sort simply stays on the caller's stream.
namespace {
struct SideLane {
  int device = -1;
  hipStream_t caller = nullptr, stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
// a few lanes per thread, one per caller stream (frames streamed through several streams at once keep their depth
// sorts apart); round-robin replacement is never needed in practice: callers use a handful of streams
constexpr int MAX_LANES = 8;
thread_local SideLane g_lanes[MAX_LANES];

SideLane* side_lane(hipStream_t caller) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  SideLane* free_slot = nullptr;
  for (SideLane& l : g_lanes) {
    if (l.stream != nullptr && l.device == dev && l.caller == caller) return &l;
    if (l.stream == nullptr && free_slot == nullptr) free_slot = &l;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(caller, &cs) != hipSuccess) return nullptr;
  if (cs != hipStreamCaptureStatusNone) {
    // a capturing caller (torch captures on a stream of its own, which never made an eager call) cannot create a lane:
    // it borrows one that an eager call on this device created.  Only a capture's origin stream gets here
    // (instag_raster_args.single_stream), and a thread issues one call at a time, so the lane is not in use.
    for (SideLane& l : g_lanes)
      if (l.stream != nullptr && l.device == dev) return &l;
    return nullptr;
  }
  if (free_slot == nullptr) return nullptr;        // all lanes taken: the sort stays on the caller's stream
  SideLane l;
  if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&l.join, hipEventDisableTiming) != hipSuccess) {
    (void)hipStreamDestroy(l.stream);
    return nullptr;
  }
  l.device = dev;
  l.caller = caller;
  *free_slot = l;
  return free_slot;
}
}  // namespace

// ---- layouts ----------------------------------------------------------------------------------------
GeomLayout geom_layout(int32_t N) {
  GeomLayout L;
  const size_t n = (size_t)(N > 0 ? N : 1);
  size_t o = 0;
  L.rec2d = o; o = align_up(o + n * REC_FLOATS * sizeof(float), 256);
  L.cov3d = o; o = align_up(o + n * 6 * sizeof(float), 256);
  L.tiles_touched = o; o = align_up(o + n * sizeof(uint32_t), 256);
  L.point_offsets = o; o = align_up(o + n * sizeof(uint32_t), 256);
  L.flags = o; o = align_up(o + n * sizeof(uint32_t), 256);
  L.cull_thr = o; o = align_up(o + n * sizeof(float), 256);
  L.depth_key = o; o = align_up(o + n * sizeof(uint32_t), 256);
  L.depth_key_alt = o; o = align_up(o + n * sizeof(uint32_t), 256);
  L.order_a = o; o = align_up(o + n * sizeof(uint32_t), 256);
  L.order = o; o = align_up(o + n * sizeof(uint32_t), 256);
  L.dsort_blocks = sort_blocks((uint32_t)n, SORT_IPT_DEPTH);
  L.dsort_zero_words = 16 + (size_t)4 * L.dsort_blocks * 256;
  L.dsort_zero = o; o = align_up(o + L.dsort_zero_words * sizeof(uint32_t), 256);
  L.dsort_digit_base = o; o = align_up(o + (size_t)HIST_SLICES * 4 * 256 * sizeof(uint32_t), 256);
  L.dkey_blocks = depth_key_blocks((int32_t)n);
  L.dsort_partials = o; o = align_up(o + (size_t)L.dkey_blocks * 4 * 256 * sizeof(uint32_t), 256);
  L.scan_state_words = 2 * (1 + (size_t)div_up<size_t>(n, 1024));
  L.scan_state = o; o = align_up(o + L.scan_state_words * sizeof(uint32_t), 256);
  L.total = o;
  return L;
}

ImageLayout image_layout(int32_t H, int32_t W) {
  ImageLayout L;
  const size_t tiles = (size_t)div_up(W, TILE_X) * div_up(H, TILE_Y);
  const size_t P = (size_t)H * W;
  size_t o = 0;
  L.ranges = o; o = align_up(o + tiles * 2 * sizeof(int32_t), 256);
  L.n_contrib = o; o = align_up(o + P * sizeof(uint32_t), 256);
  L.final_T = o; o = align_up(o + P * sizeof(float), 256);
  L.tile_rounds = o; o = align_up(o + tiles * sizeof(uint32_t), 256);
  L.total = o;
  return L;
}

BinningLayout binning_layout(int64_t R, int32_t H, int32_t W) {
  BinningLayout L;
  const size_t r = (size_t)(R > 0 ? R : 1);
  const size_t tiles = (size_t)div_up(W, TILE_X) * div_up(H, TILE_Y);
  size_t o = 0;
  L.keys_unsorted = o; o = align_up(o + r * sizeof(uint32_t), 256);
  L.vals_unsorted = o; o = align_up(o + r * sizeof(uint32_t), 256);
  L.keys = o; o = align_up(o + r * sizeof(uint32_t), 256);
  L.vals = o; o = align_up(o + r * sizeof(uint32_t), 256);
  L.ktmp = o; o = align_up(o + r * sizeof(uint32_t), 256);
  L.vtmp = o; o = align_up(o + r * sizeof(uint32_t), 256);
  L.gid_unsorted = o; o = align_up(o + r * sizeof(uint32_t), 256);
  L.point_list = o; o = align_up(o + r * sizeof(uint32_t), 256);
  L.tsort_blocks = sort_blocks((uint32_t)r, SORT_IPT_TILE);
  // [tickets + look-back words][seg_count + pad][row flags, two sets]: one region, cleared by the duplicate kernel
  const size_t sort_words = 16 + (size_t)3 * L.tsort_blocks * 256;
  L.row_flag_stride = align_up(r, 16);
  L.seg_slots = r / SEG_LEN + tiles + 2;
  L.tsort_zero_words = sort_words + 4 + 2 * L.row_flag_stride / 4 + 8 * tiles + L.seg_slots;
  L.tsort_zero = o;
  L.seg_count = o + sort_words * sizeof(uint32_t);
  L.row_flag = L.seg_count + 4 * sizeof(uint32_t);
  L.fwd_sync = L.row_flag + 2 * L.row_flag_stride;
  L.seg_flag = L.fwd_sync + 8 * tiles * sizeof(uint32_t);
  o = align_up(o + L.tsort_zero_words * sizeof(uint32_t), 256);
  L.tsort_digit_base = o; o = align_up(o + (size_t)HIST_SLICES * 3 * 256 * sizeof(uint32_t), 256);
  L.tsort_partials = o; o = align_up(o + (size_t)1024 * 3 * 256 * sizeof(uint32_t), 256);
  L.sort_count = o; o = align_up(o + sizeof(uint32_t), 256);
  L.seg_queue = o; o = align_up(o + 2 * L.seg_slots * sizeof(uint32_t), 256);
  L.seg_state = o; o = align_up(o + L.seg_slots * SEG_FLOATS * TILE_PIX * sizeof(float), 256);
  L.total = o;
  return L;
}

static int validate(const instag_raster_args* a) {
  INSTAG_REQUIRE(a != nullptr, "raster args is NULL");
  INSTAG_REQUIRE(a->N >= 0, "N must be >= 0");
  INSTAG_REQUIRE(a->image_height > 0 && a->image_width > 0, "image size must be positive");
  INSTAG_REQUIRE(a->image_height <= 16368 && a->image_width <= 16368, "image larger than 16368 px not supported");
  INSTAG_REQUIRE((a->shs != nullptr) != (a->colors_precomp != nullptr),
                 "Please provide excatly one of either SHs or precomputed colors!");
  INSTAG_REQUIRE(((a->scales != nullptr && a->rotations != nullptr) && a->cov3Ds_precomp == nullptr) ||
                     ((a->scales == nullptr && a->rotations == nullptr) && a->cov3Ds_precomp != nullptr),
                 "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
  INSTAG_REQUIRE(a->E == 0 || a->E == 1, "extra_attrs: only 0 or 1 channel is supported");
  INSTAG_REQUIRE(a->sh_degree >= 0 && a->sh_degree <= 3, "sh_degree must be in [0,3]");
  if (a->shs) INSTAG_REQUIRE((a->sh_degree + 1) * (a->sh_degree + 1) <= a->M, "shs has fewer coefficients than sh_degree needs");
  INSTAG_REQUIRE(a->shs_rest == nullptr || (a->shs != nullptr && a->M >= 2), "shs_rest needs shs (the DC term) and M >= 2");
  INSTAG_REQUIRE(a->bg && a->viewmatrix && a->projmatrix && a->campos, "camera pointers must not be NULL");
  INSTAG_REQUIRE(a->N == 0 || (a->means3D && a->opacities), "means3D / opacities must not be NULL");
  return INSTAG_OK;
}

// preprocess || depth sort of the Gaussians -> instance offsets in depth order.
// Emitting the instances in depth order makes the (tile, depth) sort a STABLE sort by tile id alone
// (10-12 key bits instead of 42-44): same final order, a third of the radix passes on half the bytes.
static int depth_sort(const Camera& c, const instag_raster_args* a, char* gb, const GeomLayout& L, hipStream_t s,
                      uint64_t* stamps = nullptr) {
  uint32_t* zero = (uint32_t*)(gb + L.dsort_zero);
  uint32_t* digit_base = (uint32_t*)(gb + L.dsort_digit_base);
  uint32_t* partials = (uint32_t*)(gb + L.dsort_partials);
  uint32_t* K0 = (uint32_t*)(gb + L.depth_key);
  uint32_t* KA = (uint32_t*)(gb + L.depth_key_alt);
  uint32_t* VA = (uint32_t*)(gb + L.order_a);
  uint32_t* VB = (uint32_t*)(gb + L.order);
  ProfScope p(K_SORT, s);
  if (int e = launch_depth_keys(c, a->means3D, K0, partials, zero, (uint32_t)L.dsort_zero_words, s)) return e;
  // per-block digit histograms [blocks][4][256]: the pass kernels sum them directly while they are few
  const uint32_t* hist = partials;
  int n_hist = (int)L.dkey_blocks;
  if (n_hist > 320) {
    if (int e = launch_hist_reduce(partials, n_hist, 4, HIST_SLICES, digit_base, s)) return e;
    hist = digit_base; n_hist = HIST_SLICES;
  }
  const uint32_t n = (uint32_t)a->N;
  const uint32_t* kin[4] = {K0, KA, K0, KA};
  uint32_t* kout[4] = {KA, K0, KA, nullptr};
  const uint32_t* vin[4] = {nullptr, VA, VB, VA};
  uint32_t* vout[4] = {VA, VB, VA, VB};
  for (int pass = 0; pass < 4; ++pass) {
    if (int e = launch_radix_pass(SORT_IPT_DEPTH, true, pass < 3, kin[pass], kout[pass], vin[pass], vout[pass], nullptr,
                                  n, 8 * pass, 8, hist + 256 * pass, n_hist, 4 * 256, zero + pass,
                                  zero + 16 + (size_t)pass * L.dsort_blocks * 256, s,
                                  stamps ? stamps + (size_t)pass * L.dsort_blocks * 8 : nullptr)) return e;
  }
  return INSTAG_OK;
}

static int per_gaussian_stage(const instag_raster_args* a, const Camera& c, char* gb, const GeomLayout& L, int32_t* radii,
                              hipStream_t s) {
  uint32_t* tiles_touched = (uint32_t*)(gb + L.tiles_touched);
  uint32_t* point_offsets = (uint32_t*)(gb + L.point_offsets);
  SideLane* lane = a->single_stream ? nullptr : side_lane(s);
  if (lane != nullptr) {
    INSTAG_CHECK_HIP(hipEventRecord(lane->fork, s));
    INSTAG_CHECK_HIP(hipStreamWaitEvent(lane->stream, lane->fork, 0));
    if (int e = depth_sort(c, a, gb, L, lane->stream)) return e;
    INSTAG_CHECK_HIP(hipEventRecord(lane->join, lane->stream));
  } else {
    if (int e = depth_sort(c, a, gb, L, s)) return e;
  }
  if (int e = launch_preprocess(c, a, (float*)(gb + L.rec2d), (float*)(gb + L.cov3d), tiles_touched,
                                (uint32_t*)(gb + L.flags), (float*)(gb + L.cull_thr), radii,
                                (uint32_t*)(gb + L.scan_state), (uint32_t)L.scan_state_words, s)) return e;
  if (lane != nullptr) INSTAG_CHECK_HIP(hipStreamWaitEvent(s, lane->join, 0));
  return launch_scan_counts(a->N, (const uint32_t*)(gb + L.order), tiles_touched, point_offsets,
                            (uint64_t*)(gb + L.scan_state), s);
}

}  // namespace instag

using namespace instag;

extern "C" {

const char* instag_last_error(void) { return g_err.c_str(); }
int instag_abi_version(void) { return 10; }

size_t instag_raster_geom_bytes(int32_t N) { return geom_layout(N).total; }
size_t instag_raster_image_bytes(int32_t H, int32_t W) { return image_layout(H, W).total; }
size_t instag_raster_binning_bytes(int64_t R, int32_t H, int32_t W) { return binning_layout(R, H, W).total; }
size_t instag_raster_backward_workspace_bytes(int32_t N, int64_t R) {
  (void)N;
  return align_up((size_t)(R > 0 ? R : 1) * REC_FLOATS * sizeof(float), 256);
}

int instag_raster_forward_stage1(const instag_raster_args* a, void* geom, size_t geom_bytes, int32_t* radii,
                                 int64_t* num_rendered, instag_stream_t stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (int e = validate(a)) return e;
  INSTAG_REQUIRE(num_rendered != nullptr, "num_rendered is NULL");
  const GeomLayout L = geom_layout(a->N);
  if (geom_bytes < L.total) { set_error("geom buffer too small"); return INSTAG_E_SPACE; }
  *num_rendered = 0;
  if (a->N == 0) return INSTAG_OK;
  char* gb = (char*)geom;
  const Camera c = make_camera(a);
  uint32_t* point_offsets = (uint32_t*)(gb + L.point_offsets);
  if (int e = per_gaussian_stage(a, c, gb, L, radii, s)) return e;
  uint32_t r32 = 0, stalls = 0;
  INSTAG_CHECK_HIP(hipMemcpyAsync(&r32, point_offsets + (a->N - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  if (int e = read_sort_stalls(&stalls, s, /*synchronize=*/true)) return e;
  if (stalls != 0) {
    // (sticky: also reports a stall of an EARLIER call's tile sort, which has no synchronisation point of its own)
    set_error("rasterizer binning: a sort / scan look-back gave up waiting for its predecessor (" +
              std::to_string(stalls) + " event(s)); the lists of the affected call are wrong");
    return INSTAG_E_HIP;
  }
  *num_rendered = (int64_t)r32;
  return INSTAG_OK;
}

// duplicate -> sort -> ranges -> blend over `R` instance slots.  In capacity mode R is the caller's capacity; how many
// of the slots are in use is a device word written by the duplicate kernel and read by the sort and the range kernel.
static int forward_tail(const instag_raster_args* a, void* geom, size_t geom_bytes, void* binning,
                        size_t binning_bytes, void* image, size_t image_bytes, int64_t R,
                        float* out_color, float* out_depth, float* out_normal, float* out_alpha, float* out_extra,
                        const float* aux_colors, float* out_aux, int32_t* status, hipStream_t s) {
  INSTAG_REQUIRE(out_color && out_depth && out_normal && out_alpha, "output images must not be NULL");
  INSTAG_REQUIRE((aux_colors == nullptr) == (out_aux == nullptr), "aux_colors and out_aux go together");
  INSTAG_REQUIRE(R >= 0 && R < (int64_t)1 << 30, "instance count out of range (2^30 - 1 at most)");
  const GeomLayout GL = geom_layout(a->N);
  const ImageLayout IL = image_layout(a->image_height, a->image_width);
  const BinningLayout BL = binning_layout(R, a->image_height, a->image_width);
  if (geom_bytes < GL.total) { set_error("geom buffer too small"); return INSTAG_E_SPACE; }
  if (image_bytes < IL.total) { set_error("image buffer too small"); return INSTAG_E_SPACE; }
  if (binning_bytes < BL.total) { set_error("binning buffer too small"); return INSTAG_E_SPACE; }
  char *gb = (char*)geom, *bb = (char*)binning, *ib = (char*)image;
  const Camera c = make_camera(a);
  const int tiles = c.grid_x * c.grid_y;
  int32_t* ranges = (int32_t*)(ib + IL.ranges);
  uint32_t* K0 = (uint32_t*)(bb + BL.keys_unsorted);
  uint32_t* V0 = (uint32_t*)(bb + BL.vals_unsorted);
  uint32_t* K1 = (uint32_t*)(bb + BL.keys);
  uint32_t* V1 = (uint32_t*)(bb + BL.vals);
  uint32_t* K2 = (uint32_t*)(bb + BL.ktmp);
  uint32_t* V2 = (uint32_t*)(bb + BL.vtmp);
  uint32_t* gid_u = (uint32_t*)(bb + BL.gid_unsorted);
  uint32_t* point_list = (uint32_t*)(bb + BL.point_list);
  uint32_t* zero = (uint32_t*)(bb + BL.tsort_zero);
  uint32_t* digit_base = (uint32_t*)(bb + BL.tsort_digit_base);
  uint32_t* partials = (uint32_t*)(bb + BL.tsort_partials);
  uint32_t* sort_count = (uint32_t*)(bb + BL.sort_count);
  const bool packed = use_packed_keys(R, tiles);
  if (R > 0 && a->N > 0) {
    const TilePasses tp = tile_passes(tiles);
    if (int e = launch_duplicate(c, (float*)(gb + GL.rec2d), (const uint32_t*)(gb + GL.order),
                                 (const uint32_t*)(gb + GL.point_offsets), (const uint32_t*)(gb + GL.flags),
                                 (const float*)(gb + GL.cull_thr), K0, V0, gid_u, (uint32_t)R, ranges, packed, status,
                                 sort_count, tp, partials, zero, (uint32_t)BL.tsort_zero_words, s)) return e;
    {
      ProfScope p(K_SORT, s);
      if (int e = launch_hist_reduce(partials, (int)duplicate_blocks(a->N), tp.npass, HIST_SLICES, digit_base, s))
        return e;
      // K0 -> [K2 ->] [K1 -> K2 ->] K1: the last pass always lands in `keys` / `vals`
      const uint32_t *kin = K0, *vin = packed ? nullptr : V0;
      for (int pass = 0; pass < tp.npass; ++pass) {
        const bool to_final = ((tp.npass - 1 - pass) % 2) == 0;
        uint32_t* kout = to_final ? K1 : K2;
        uint32_t* vout = packed ? nullptr : (to_final ? V1 : V2);
        if (int e = launch_radix_pass(SORT_IPT_TILE, !packed, true, kin, kout, vin, vout, sort_count, (uint32_t)R,
                                      (packed ? PACK_SHIFT : 0) + pass * tp.bits_per, tp.nbits[pass],
                                      digit_base + 256 * pass, HIST_SLICES, tp.npass * 256, zero + pass,
                                      zero + 16 + (size_t)pass * BL.tsort_blocks * 256, s)) return e;
        kin = kout; vin = vout;
      }
    }
    if (int e = launch_ranges(R, sort_count, K1, V1, gid_u, point_list, ranges, (uint32_t)tiles, packed, s)) return e;
  } else {
    INSTAG_CHECK_HIP(hipMemsetAsync(ranges, 0, (size_t)tiles * 2 * sizeof(int32_t), s));
  }
  // (R == 0: no duplicate kernel ran, nothing cleared the claim words -- the whole-tile kernel handles empty lists)
  return launch_blend_forward(c, ranges, point_list, (const float*)(gb + GL.rec2d), (uint32_t*)(ib + IL.n_contrib),
                              (float*)(ib + IL.final_T), out_color, out_depth, out_normal, out_alpha,
                              a->E > 0 ? out_extra : nullptr, aux_colors, out_aux, (uint32_t*)(bb + BL.seg_queue),
                              (uint32_t*)(bb + BL.seg_count), (float*)(bb + BL.seg_state),
                              (uint32_t*)(ib + IL.tile_rounds), R > 0 ? (uint32_t*)(bb + BL.fwd_sync) : nullptr,
                              (uint32_t*)(bb + BL.seg_flag), a->walk_hints, R, s);
}

int instag_raster_forward_stage2(const instag_raster_args* a, void* geom, size_t geom_bytes, void* binning,
                                 size_t binning_bytes, void* image, size_t image_bytes, int64_t R,
                                 float* out_color, float* out_depth, float* out_normal, float* out_alpha,
                                 float* out_extra, const float* aux_colors, float* out_aux,
                                 instag_stream_t stream_) {
  if (int e = validate(a)) return e;
  return forward_tail(a, geom, geom_bytes, binning, binning_bytes, image, image_bytes, R, out_color,
                      out_depth, out_normal, out_alpha, out_extra, aux_colors, out_aux, nullptr, (hipStream_t)stream_);
}

int instag_raster_forward_capacity(const instag_raster_args* a, void* geom, size_t geom_bytes, void* binning,
                                   size_t binning_bytes, void* image, size_t image_bytes, int64_t capacity,
                                   int32_t* radii, int32_t* status, float* out_color, float* out_depth,
                                   float* out_normal, float* out_alpha, float* out_extra,
                                   const float* aux_colors, float* out_aux, instag_stream_t stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (int e = validate(a)) return e;
  INSTAG_REQUIRE(status != nullptr && radii != nullptr, "status / radii is NULL");
  INSTAG_REQUIRE(capacity >= 1, "capacity must be >= 1");
  const GeomLayout L = geom_layout(a->N);
  if (geom_bytes < L.total) { set_error("geom buffer too small"); return INSTAG_E_SPACE; }
  char* gb = (char*)geom;
  const Camera c = make_camera(a);
  if (a->N > 0) {
    if (int e = per_gaussian_stage(a, c, gb, L, radii, s)) return e;
  } else {
    INSTAG_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), s));       // [0] = instances needed = 0
  }
  // the status words are written by the duplicate kernel
  return forward_tail(a, geom, geom_bytes, binning, binning_bytes, image, image_bytes, capacity, out_color,
                      out_depth, out_normal, out_alpha, out_extra, aux_colors, out_aux, status, s);
}

int instag_raster_backward(const instag_raster_args* a, const void* geom, size_t geom_bytes, const void* binning,
                           size_t binning_bytes, const void* image, size_t image_bytes, int64_t R,
                           const int32_t* radii, const float* dL_dout_color, const float* dL_dout_depth,
                           const float* dL_dout_normal, const float* dL_dout_alpha, const float* dL_dout_extra,
                           void* workspace, size_t workspace_bytes, float* dL_dmeans3D, float* dL_dmeans2D,
                           float* dL_dshs, float* dL_dcolors_precomp, float* dL_dopacities, float* dL_dscales,
                           float* dL_drotations, float* dL_dcov3Ds_precomp, float* dL_dextra_attrs,
                           float* dL_dshs_rest, const float* aux_colors, const float* dL_dout_aux,
                           float* dL_daux_colors, int32_t aux_colors_only, instag_stream_t stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (int e = validate(a)) return e;
  INSTAG_REQUIRE(radii != nullptr || a->N == 0, "radii is NULL");
  INSTAG_REQUIRE((aux_colors == nullptr) == (dL_dout_aux == nullptr), "aux_colors and dL_dout_aux go together");
  INSTAG_REQUIRE(dL_daux_colors == nullptr || aux_colors != nullptr, "dL_daux_colors needs aux_colors");
  const GeomLayout GL = geom_layout(a->N);
  const ImageLayout IL = image_layout(a->image_height, a->image_width);
  const BinningLayout BL = binning_layout(R, a->image_height, a->image_width);
  if (geom_bytes < GL.total || image_bytes < IL.total || binning_bytes < BL.total) {
    set_error("state buffer too small"); return INSTAG_E_SPACE;
  }
  const size_t need = instag_raster_backward_workspace_bytes(a->N, R);
  if (workspace_bytes < need || workspace == nullptr) { set_error("backward workspace too small"); return INSTAG_E_SPACE; }
  if (a->N == 0) return INSTAG_OK;
  const char *gb = (const char*)geom, *bb = (const char*)binning, *ib = (const char*)image;
  const Camera c = make_camera(a);
  float* inst_grad = (float*)workspace;
  const float* g_extra = a->E > 0 ? dL_dout_extra : nullptr;
  const bool full = dL_dout_depth || dL_dout_normal || g_extra;
  // the auxiliary image rides along the main pass when that is an rgb-only pass (its gradients use the row slots of
  // the depth / normal / extra channels); otherwise it gets its own blend launch over the same state, below
  const bool fused_aux = aux_colors != nullptr && !full;
  INSTAG_REQUIRE(!aux_colors_only || fused_aux, "aux_colors_only needs aux_colors and an rgb-only main pass");
  if (R > 0) {
    if (int e = launch_blend_backward(c, (const int32_t*)(ib + IL.ranges), (const uint32_t*)(bb + BL.point_list),
                                      (const uint32_t*)(bb + BL.vals), (const float*)(gb + GL.rec2d), (const uint32_t*)(ib + IL.n_contrib),
                                      (const float*)(ib + IL.final_T), dL_dout_color, dL_dout_depth,
                                      dL_dout_normal, dL_dout_alpha, g_extra, inst_grad, nullptr,
                                      fused_aux ? aux_colors : nullptr, fused_aux ? dL_dout_aux : nullptr,
                                      fused_aux ? (aux_colors_only ? 2 : 1) : 0,
                                      (const uint32_t*)(bb + BL.seg_queue), (const uint32_t*)(bb + BL.seg_count),
                                      (const float*)(bb + BL.seg_state), (const uint32_t*)(ib + IL.tile_rounds),
                                      (uint32_t)BL.seg_slots, (uint8_t*)const_cast<char*>(bb + BL.row_flag) + 0 * BL.row_flag_stride, s)) return e;
  }
  if (int e = launch_preprocess_backward(c, a, (const float*)(gb + GL.rec2d), (const float*)(gb + GL.cov3d),
                                         (const uint32_t*)(gb + GL.tiles_touched), (const uint32_t*)(gb + GL.flags),
                                         radii, inst_grad, (uint8_t*)const_cast<char*>(bb + BL.row_flag), (uint32_t)R, dL_dmeans3D,
                                         dL_dmeans2D, dL_dshs,
                                         dL_dcolors_precomp, dL_dopacities, dL_dscales, dL_drotations, dL_dcov3Ds_precomp,
                                         a->E > 0 ? dL_dextra_attrs : nullptr, dL_dshs_rest,
                                         fused_aux ? dL_daux_colors : nullptr, s)) return e;
  if (aux_colors != nullptr && !fused_aux) {
    // all-channel main pass: the aux image's backward reuses the workspace once the main rows have been reduced
    if (R > 0) {
      if (int e = launch_blend_backward(c, (const int32_t*)(ib + IL.ranges), (const uint32_t*)(bb + BL.point_list),
                                        (const uint32_t*)(bb + BL.vals), (const float*)(gb + GL.rec2d),
                                        (const uint32_t*)(ib + IL.n_contrib), (const float*)(ib + IL.final_T), dL_dout_aux,
                                        nullptr, nullptr, nullptr, nullptr, inst_grad, aux_colors, nullptr, nullptr, 0,
                                        (const uint32_t*)(bb + BL.seg_queue), (const uint32_t*)(bb + BL.seg_count),
                                      (const float*)(bb + BL.seg_state), (const uint32_t*)(ib + IL.tile_rounds),
                                      (uint32_t)BL.seg_slots, (uint8_t*)const_cast<char*>(bb + BL.row_flag) + 0 * BL.row_flag_stride, s))
        return e;
    }
    return launch_aux_backward_reduce(c, (const float*)(gb + GL.rec2d), (const uint32_t*)(gb + GL.tiles_touched), radii,
                                      inst_grad, (uint8_t*)const_cast<char*>(bb + BL.row_flag), (uint32_t)R,
                                      dL_daux_colors, dL_dmeans2D, /*accumulate=*/true, s);
  }
  return INSTAG_OK;
}

int instag_raster_aux_backward(const instag_raster_args* a, const void* geom, size_t geom_bytes, const void* binning,
                               size_t binning_bytes, const void* image, size_t image_bytes, int64_t R,
                               const int32_t* radii, const float* aux_colors, const float* dL_dout_aux,
                               void* workspace, size_t workspace_bytes, float* dL_daux_colors, float* dL_dmeans2D,
                               instag_stream_t stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (int e = validate(a)) return e;
  INSTAG_REQUIRE(aux_colors && dL_dout_aux, "aux_backward: aux_colors / dL_dout_aux is NULL");
  INSTAG_REQUIRE(radii != nullptr || a->N == 0, "radii is NULL");
  const GeomLayout GL = geom_layout(a->N);
  const ImageLayout IL = image_layout(a->image_height, a->image_width);
  const BinningLayout BL = binning_layout(R, a->image_height, a->image_width);
  if (geom_bytes < GL.total || image_bytes < IL.total || binning_bytes < BL.total) {
    set_error("state buffer too small"); return INSTAG_E_SPACE;
  }
  const size_t need = instag_raster_backward_workspace_bytes(a->N, R);
  if (workspace_bytes < need || workspace == nullptr) { set_error("backward workspace too small"); return INSTAG_E_SPACE; }
  if (a->N == 0) return INSTAG_OK;
  const char *gb = (const char*)geom, *bb = (const char*)binning, *ib = (const char*)image;
  const Camera c = make_camera(a);
  float* inst_grad = (float*)workspace;
  if (R > 0) {
    if (int e = launch_blend_backward(c, (const int32_t*)(ib + IL.ranges), (const uint32_t*)(bb + BL.point_list),
                                      (const uint32_t*)(bb + BL.vals), (const float*)(gb + GL.rec2d),
                                      (const uint32_t*)(ib + IL.n_contrib), (const float*)(ib + IL.final_T), dL_dout_aux,
                                      nullptr, nullptr, nullptr, nullptr, inst_grad, aux_colors, nullptr, nullptr,
                                      dL_daux_colors == nullptr ? 3 : 0,          // no colour gradient wanted: mean-only pass
                                      (const uint32_t*)(bb + BL.seg_queue), (const uint32_t*)(bb + BL.seg_count),
                                      (const float*)(bb + BL.seg_state), (const uint32_t*)(ib + IL.tile_rounds),
                                      (uint32_t)BL.seg_slots, (uint8_t*)const_cast<char*>(bb + BL.row_flag) + 1 * BL.row_flag_stride, s))
      return e;
  }
  return launch_aux_backward_reduce(c, (const float*)(gb + GL.rec2d), (const uint32_t*)(gb + GL.tiles_touched), radii,
                                    inst_grad, (uint8_t*)const_cast<char*>(bb + BL.row_flag) + BL.row_flag_stride,
                                    (uint32_t)R, dL_daux_colors, dL_dmeans2D, /*accumulate=*/false, s);
}

int instag_raster_debug_export(const void* geom, size_t geom_bytes, const void* binning, size_t binning_bytes,
                               const void* image, size_t image_bytes, int32_t N, int64_t R, int32_t H, int32_t W,
                               uint32_t* tiles_touched, uint32_t* point_offsets, uint64_t* keys_sorted,
                               uint32_t* point_list, int32_t* ranges, uint32_t* n_contrib, float* final_T,
                               float* rec2d, instag_stream_t stream_) {
  hipStream_t s = (hipStream_t)stream_;
  const GeomLayout GL = geom_layout(N);
  const ImageLayout IL = image_layout(H, W);
  const BinningLayout BL = binning_layout(R, H, W);
  const char *gb = (const char*)geom, *bb = (const char*)binning, *ib = (const char*)image;
  const size_t tiles = (size_t)div_up(W, TILE_X) * div_up(H, TILE_Y), P = (size_t)H * W;
  auto cp = [&](void* dst, const void* src, size_t n) -> hipError_t {
    if (!dst || n == 0) return hipSuccess;
    return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s);
  };
  if (geom) {
    if (geom_bytes < GL.total) { set_error("geom buffer too small"); return INSTAG_E_SPACE; }
    INSTAG_CHECK_HIP(cp(tiles_touched, gb + GL.tiles_touched, (size_t)N * 4));
    INSTAG_CHECK_HIP(cp(point_offsets, gb + GL.point_offsets, (size_t)N * 4));
    INSTAG_CHECK_HIP(cp(rec2d, gb + GL.rec2d, (size_t)N * REC_FLOATS * 4));
  }
  if (binning && R > 0) {
    if (binning_bytes < BL.total) { set_error("binning buffer too small"); return INSTAG_E_SPACE; }
    if (keys_sorted) {
      if (int e = launch_export_keys(R, (const uint32_t*)(bb + BL.keys), (const uint32_t*)(bb + BL.point_list),
                                     (const float*)(gb + GL.rec2d), keys_sorted,
                                     use_packed_keys(R, (int)tiles), s)) return e;
    }
    INSTAG_CHECK_HIP(cp(point_list, bb + BL.point_list, (size_t)R * 4));
  }
  if (image) {
    if (image_bytes < IL.total) { set_error("image buffer too small"); return INSTAG_E_SPACE; }
    INSTAG_CHECK_HIP(cp(ranges, ib + IL.ranges, tiles * 8));
    INSTAG_CHECK_HIP(cp(n_contrib, ib + IL.n_contrib, P * 4));
    INSTAG_CHECK_HIP(cp(final_T, ib + IL.final_T, P * 4));
  }
  return INSTAG_OK;
}

int instag_raster_debug_export_flags(const void* geom, size_t geom_bytes, int32_t N, uint32_t* flags,
                                     instag_stream_t stream_) {
  const GeomLayout GL = geom_layout(N);
  INSTAG_REQUIRE(geom != nullptr && flags != nullptr, "debug_export_flags: NULL pointer");
  if (geom_bytes < GL.total) { set_error("geom buffer too small"); return INSTAG_E_SPACE; }
  if (N > 0)
    INSTAG_CHECK_HIP(hipMemcpyAsync(flags, (const char*)geom + GL.flags, (size_t)N * 4, hipMemcpyDeviceToDevice,
                                    (hipStream_t)stream_));
  return INSTAG_OK;
}

/* diagnostics: the depth sort alone on the caller's stream, with per-block phase timestamps (100 MHz ticks, [4 passes]
 * [blocks][8]: block start, keys loaded, ranked, local scans done, look-back done, -, stores issued) */
int instag_debug_depth_sort(const instag_raster_args* a, void* geom, size_t geom_bytes, uint64_t* stamps,
                            uint32_t* order_out, instag_stream_t stream_) {
  INSTAG_REQUIRE(a != nullptr && a->N > 0, "debug_depth_sort: N must be positive");
  const GeomLayout L = geom_layout(a->N);
  if (geom_bytes < L.total) { set_error("geom buffer too small"); return INSTAG_E_SPACE; }
  const Camera c = make_camera(a);
  if (int e = depth_sort(c, a, (char*)geom, L, (hipStream_t)stream_, stamps)) return e;
  if (order_out)
    INSTAG_CHECK_HIP(hipMemcpyAsync(order_out, (char*)geom + L.order, (size_t)a->N * 4, hipMemcpyDeviceToDevice,
                                    (hipStream_t)stream_));
  return INSTAG_OK;
}

uint32_t instag_debug_depth_sort_blocks(int32_t N) { return geom_layout(N).dsort_blocks; }

int instag_raster_sort_stalls(uint32_t* count, int32_t synchronize, instag_stream_t stream_) {
  INSTAG_REQUIRE(count != nullptr, "sort_stalls: count is NULL");
  return read_sort_stalls(count, (hipStream_t)stream_, synchronize != 0);
}

int instag_raster_sort_stalls_clear(instag_stream_t stream_) { return clear_sort_stalls((hipStream_t)stream_); }

int instag_debug_scan_stall_probe(instag_stream_t stream_) {
  hipStream_t s = (hipStream_t)stream_;
  constexpr int N = 8;
  // [order N][tiles_touched N][point_offsets N] u32, then the scan state (u64: ticket, look-back words)
  uint32_t host[3 * N];
  for (int i = 0; i < N; ++i) { host[i] = (uint32_t)i; host[N + i] = 1u; host[2 * N + i] = 0u; }
  const uint64_t state_host[4] = {1ull, 0ull, 0ull, 0ull};      // ticket = 1: the block becomes block 1, block 0 never runs
  char* dev = nullptr;
  INSTAG_CHECK_HIP(hipMalloc((void**)&dev, sizeof(host) + sizeof(state_host)));
  hipError_t err = hipMemcpyAsync(dev, host, sizeof(host), hipMemcpyHostToDevice, s);
  if (err == hipSuccess) err = hipMemcpyAsync(dev + sizeof(host), state_host, sizeof(state_host), hipMemcpyHostToDevice, s);
  int rc = INSTAG_OK;
  if (err == hipSuccess)
    rc = launch_scan_counts(N, (const uint32_t*)dev, (const uint32_t*)dev + N, (uint32_t*)dev + 2 * N,
                            (uint64_t*)(dev + sizeof(host)), s);
  if (err == hipSuccess) err = hipStreamSynchronize(s);
  (void)hipFree(dev);
  if (err != hipSuccess) { set_error(std::string("scan_stall_probe: ") + hipGetErrorString(err)); return INSTAG_E_HIP; }
  return rc;
}

int instag_prof_enable(int mask) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  p.mask = mask;
  return INSTAG_OK;
}

static void prof_drain(ProfState& p, int k) {
  for (auto& pr : p.pending[k]) {
    float ms = 0.f;
    if (hipEventSynchronize(pr.b) == hipSuccess && hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
      p.total_ms[k] += ms;
      p.launches[k] += 1;
    }
    (void)hipEventDestroy(pr.a);
    (void)hipEventDestroy(pr.b);
  }
  p.pending[k].clear();
}

int instag_prof_reset(void) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  for (int k = 0; k < INSTAG_PROF_KERNELS; ++k) {
    prof_drain(p, k);
    p.total_ms[k] = 0;
    p.launches[k] = 0;
  }
  return INSTAG_OK;
}

int instag_prof_graph_end(void);

int instag_prof_graph_begin(int32_t max_pairs) {
  INSTAG_REQUIRE(max_pairs > 0 && max_pairs <= 4096, "prof_graph_begin: max_pairs out of range");
  if (int e = instag_prof_graph_end()) return e;
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  p.graph_pool.resize(max_pairs);
  for (auto& g : p.graph_pool) {
    INSTAG_CHECK_HIP(hipEventCreate(&g.a));
    INSTAG_CHECK_HIP(hipEventCreate(&g.b));
  }
  p.graph_used = 0;
  return INSTAG_OK;
}

int instag_prof_graph_pairs_used(void) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  int n = 0;
  for (int i = 0; i < p.graph_used; ++i) n += p.graph_pool[i].closed ? 1 : 0;
  return n;
}

int instag_prof_graph_collect(void) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  for (int i = 0; i < p.graph_used; ++i) {
    const ProfState::GraphPair& g = p.graph_pool[i];
    if (!g.closed || g.kernel < 0) continue;
    float ms = 0.f;
    if (hipEventSynchronize(g.b) == hipSuccess && hipEventElapsedTime(&ms, g.a, g.b) == hipSuccess) {
      p.total_ms[g.kernel] += ms;
      p.launches[g.kernel] += 1;
    } else {
      (void)hipGetLastError();
      set_error("prof_graph_collect: a captured event pair could not be read (was the graph replayed and synchronised?)");
      return INSTAG_E_HIP;
    }
  }
  return INSTAG_OK;
}

int instag_prof_graph_end(void) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  for (auto& g : p.graph_pool) {
    if (g.a) (void)hipEventDestroy(g.a);
    if (g.b) (void)hipEventDestroy(g.b);
  }
  p.graph_pool.clear();
  p.graph_used = 0;
  return INSTAG_OK;
}

int instag_prof_read(int k, double* total_ms, int64_t* launches) {
  INSTAG_REQUIRE(k >= 0 && k < INSTAG_PROF_KERNELS, "kernel id out of range");
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  prof_drain(p, k);
  if (total_ms) *total_ms = p.total_ms[k];
  if (launches) *launches = p.launches[k];
  return INSTAG_OK;
}

}  // extern "C"
