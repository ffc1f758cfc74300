// Stable LSD radix sort passes and the instance-offset scan of the rasterizer's binning stage, written for its two
// jobs: (1) Gaussians by view depth (32-bit float bits, N keys), (2) instances by tile id (the few bits of the tile
// index, R keys in depth order).  Both replace rocPRIM calls of round 1 (10 + 12 launches and 5 memset nodes per
// forward, merge sort below 1 M keys) with 4 + 2 launches and no memset node:
//
//   * one launch per digit pass: decoupled look-back over per-(block, digit) counters (flag and count in ONE 32-bit
//     word, agent-scope relaxed atomics), block order by a ticket drawn at block start, so a block only ever waits for
//     blocks that are already running -- no assumption about dispatch order, every spin ends;
//   * the look-back words and tickets must be zero at launch: they are cleared by the kernel that runs BEFORE the first
//     pass anyway (depth-key kernel / duplicate kernel), one private set per pass, so nothing is cleared between passes;
//   * the global digit histograms come from per-block partial histograms written with plain stores by those same
//     kernels (no global atomics, no zero-initialised counters), summed and scanned by one small launch;
//   * the number of keys is read from DEVICE memory (capacity mode: the instance count never reaches the host), so
//     unused capacity is neither padded nor sorted.
//
// Ranking inside a block is the wave64 "match" scheme: log2(radix) ballots give every lane the set of lanes holding
// the same digit; the lowest such lane bumps the wave's LDS counter once for all of them.  Keys are then reordered
// through LDS so that every digit's run leaves as one contiguous store.  Stable and deterministic.
#include "raster_internal.hpp"

namespace instag {
namespace {

constexpr int SORT_THREADS = 256;
constexpr int PASS_THREADS = 1024;
constexpr int RADIX = 256;
constexpr int SPIN_LIMIT = 1 << 21;
constexpr int LB_CHUNK = 8;
constexpr uint32_t LB_PARTIAL = 1u << 30, LB_COMPLETE = 2u << 30, LB_FLAGS = 3u << 30, LB_VALUE = ~LB_FLAGS;

// A look-back that gave up (SPIN_LIMIT polls without the predecessor publishing) continues with a wrong prefix: the
// lists it produces are mis-sorted.  That must never pass silently: every such event bumps this sticky counter, which
// the host reads at its synchronising calls (instag_raster_forward_stage1, instag_raster_sort_stalls).
__device__ uint32_t g_sort_stalls = 0;

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_agent64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent64(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive scan of one value per thread over the 256 threads of the block (s_w: 4 words of LDS)
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) base += (w < wave) ? s_w[w] : 0u;
  if (total) *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return base + inc - v;
}

// exclusive scan over the 256 digits, one per thread of the first four waves (every thread of the block calls this)
__device__ __forceinline__ uint32_t digits_exclusive_scan(uint32_t v, uint32_t* s_w) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (wave < 4 && lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < 3; ++w) base += (w < wave) ? s_w[w] : 0u;
  return base + inc - v;
}

// One digit pass.  Block = PASS_THREADS threads (16 waves: four per SIMD hide each other's ballot / LDS latencies; with
// one wave per SIMD the ranking alone took 0.36 us per key and wave), IPT keys per thread.
//   keys_in/vals_in -> keys_out/vals_out (vals_in == nullptr with HAS_VALUES: the value is the key's input index)
//   count = *count_ptr (device), clamped to count_max; shift / nbits: the digit
//   hist[j * hist_stride + d], j < n_hist: partial histograms of this pass's digit (every block sums them and scans
//   the 256 totals itself: no separate launch);  ticket, lookback[grid][256]: zero at launch
template <int IPT, bool HAS_VALUES, bool WRITE_KEYS>
__global__ void __launch_bounds__(PASS_THREADS)
radix_pass_kernel(const uint32_t* __restrict__ keys_in, uint32_t* __restrict__ keys_out,
                  const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ vals_out,
                  const uint32_t* __restrict__ count_ptr, uint32_t count_max, int shift, int nbits,
                  const uint32_t* __restrict__ hist, int n_hist, int hist_stride, uint32_t* __restrict__ ticket,
                  uint32_t* __restrict__ lookback, uint64_t* __restrict__ stamps) {
  constexpr int TILE = PASS_THREADS * IPT;
  constexpr int WAVES = PASS_THREADS / 64;
  // diagnostics (scripts/bench_sort.py): 100 MHz timestamps of the block's phases; null in every production call
#define INSTAG_STAMP(k) do { if (stamps != nullptr && threadIdx.x == 0) stamps[(size_t)s_bid * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  __shared__ uint32_t s_hist[WAVES][RADIX];  // per-wave digit counts, then per-wave exclusive offsets
  __shared__ uint32_t s_dstart[RADIX];       // start of the digit's run inside the block-sorted tile
  __shared__ uint32_t s_gbase[RADIX];        // global position of the run minus s_dstart
  __shared__ uint32_t s_keys[TILE];
  __shared__ uint32_t s_vals[HAS_VALUES ? TILE : 1];
  __shared__ uint32_t s_w[4], s_w2[4];
  __shared__ uint32_t s_bid;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int d = tid; d < WAVES * RADIX; d += PASS_THREADS) (&s_hist[0][0])[d] = 0;
  if (tid == 0) s_bid = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint32_t bid = s_bid;
  INSTAG_STAMP(0);
  uint32_t count = count_ptr ? *count_ptr : count_max;
  count = min(count, count_max);
  const uint64_t base = (uint64_t)bid * TILE;
  if (base >= count) return;                         // (whole block: bid is uniform)
  const uint32_t valid = (uint32_t)min((uint64_t)TILE, (uint64_t)count - base);
  const uint32_t mask = (1u << nbits) - 1u;

  // all loads unconditional (index clamped into the tile) and issued back to back: predicated loads behind branches
  // made the compiler wait for every single one
  uint32_t key[IPT], val[IPT], rank[IPT];
  const uint32_t* kin = keys_in + base;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const uint32_t idx = (uint32_t)wave * (64 * IPT) + i * 64 + lane;     // wave-striped: stable order = (wave, i, lane)
    key[i] = kin[min(idx, valid - 1u)];
  }
  if (HAS_VALUES) {
    if (vals_in != nullptr) {
      const uint32_t* vin = vals_in + base;
#pragma unroll
      for (int i = 0; i < IPT; ++i) val[i] = vin[min((uint32_t)wave * (64 * IPT) + i * 64 + lane, valid - 1u)];
    } else {
#pragma unroll
      for (int i = 0; i < IPT; ++i) val[i] = (uint32_t)base + (uint32_t)wave * (64 * IPT) + i * 64 + lane;
    }
  }
  // total count of digit `tid` over all keys (threads 0..255): sum of the partial histograms, HB independent loads per
  // round trip (the first batch shares its round trip with the key loads above)
  uint32_t gtotal = 0;
  if (tid < RADIX) {
    constexpr int HB = 32;
    const uint32_t* hp = hist + tid;
    for (int j0 = 0; j0 < n_hist; j0 += HB) {
      uint32_t h[HB];
#pragma unroll
      for (int j = 0; j < HB; ++j) h[j] = (j0 + j < n_hist) ? hp[(size_t)(j0 + j) * hist_stride] : 0u;
#pragma unroll
      for (int j = 0; j < HB; ++j) gtotal += h[j];
    }
  }
  INSTAG_STAMP(1);
  // A digit that EVERY key shares makes the pass the identity permutation: the tile is copied as it stands -- no
  // ranking, no look-back chain through the blocks in front (most of a pass's 9-15 us at 100k keys).  Every block sums
  // the same histograms, so the whole grid takes the same way.  The depth sort's last pass is the case: view-space z
  // in [0.5, 2) -- a head at arm's length -- has one top byte (0x3F).
  if (__syncthreads_or(tid < RADIX && gtotal == count) != 0) {
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const uint32_t idx = (uint32_t)wave * (64 * IPT) + i * 64 + lane;
      if (idx < valid) {
        if (WRITE_KEYS) keys_out[base + idx] = key[i];
        if (HAS_VALUES) vals_out[base + idx] = val[i];
      }
    }
    return;
  }
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const uint32_t idx = (uint32_t)wave * (64 * IPT) + i * 64 + lane;
    const bool ok = idx < valid;
    const uint32_t d = (key[i] >> shift) & mask;
    uint64_t peers = __builtin_amdgcn_ballot_w64(ok);
    for (int b = 0; b < nbits; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __builtin_amdgcn_ballot_w64(bit);
      peers &= bit ? m : ~m;
    }
    // every lane reads its digit's counter (one LDS read, lanes of a group hit the same word), then the group's lowest
    // lane bumps it for the whole group: LDS operations of a wave execute in program order, so the next key's read
    // sees this write
    const uint32_t prev = ok ? s_hist[wave][d] : 0u;
    if (ok && lane == __builtin_ctzll(peers)) s_hist[wave][d] = prev + (uint32_t)__builtin_popcountll(peers);
    rank[i] = prev + (uint32_t)__builtin_popcountll(peers & lt_mask);
  }
  __syncthreads();
  INSTAG_STAMP(2);
  // thread d < 256 owns digit d: wave offsets, the digit's run inside the tile, its global position by look-back
  {
    uint32_t cnt = 0;
    if (tid < RADIX) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = s_hist[w][tid];
        s_hist[w][tid] = cnt;
        cnt += c;
      }
    }
    const uint32_t dstart = digits_exclusive_scan(cnt, s_w);
    // global start of the digit = exclusive scan over the digits of the summed partial histograms
    const uint32_t gstart = digits_exclusive_scan(gtotal, s_w2);
    uint32_t excl = 0;
    INSTAG_STAMP(3);
    if (tid < RADIX && (uint32_t)tid <= mask) {
      uint32_t* mine = lookback + (size_t)bid * RADIX + tid;
      st_agent(mine, LB_PARTIAL | cnt);
      // look back LB_CHUNK blocks per round trip: the loads of a chunk are independent of each other; a block that has
      // not published yet drew its ticket before this one, so it is running and publishes without waiting for anybody
      // (the poll count is bounded all the same, about a second: a wave must never spin for ever)
      uint32_t b = bid;
      bool done = false;
      while (b > 0 && !done) {
        uint32_t st[LB_CHUNK];
#pragma unroll
        for (int j = 0; j < LB_CHUNK; ++j)
          st[j] = (uint32_t)j < b ? ld_agent(lookback + (size_t)(b - 1 - j) * RADIX + tid) : LB_COMPLETE;
#pragma unroll
        for (int j = 0; j < LB_CHUNK; ++j) {
          if (!done) {
            if ((uint32_t)j < b) {
              const uint32_t* p = lookback + (size_t)(b - 1 - j) * RADIX + tid;
              for (int polls = 0; (st[j] & LB_FLAGS) == 0u && polls < SPIN_LIMIT; ++polls) {
                __builtin_amdgcn_s_sleep(1);
                st[j] = ld_agent(p);
              }
              if ((st[j] & LB_FLAGS) == 0u) atomicAdd(&g_sort_stalls, 1u);     // gave up: the result is wrong, say so
            }
            excl += st[j] & LB_VALUE;
            done = (st[j] & LB_FLAGS) == LB_COMPLETE;
          }
        }
        b = b > (uint32_t)LB_CHUNK ? b - LB_CHUNK : 0u;
      }
      st_agent(mine, LB_COMPLETE | (excl + cnt));
    }
    if (tid < RADIX) {
      s_dstart[tid] = dstart;
      s_gbase[tid] = gstart + excl - dstart;
    }
  }
  __syncthreads();
  INSTAG_STAMP(4);
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const uint32_t idx = (uint32_t)wave * (64 * IPT) + i * 64 + lane;
    if (idx < valid) {
      const uint32_t d = (key[i] >> shift) & mask;
      const uint32_t p = s_dstart[d] + s_hist[wave][d] + rank[i];
      s_keys[p] = key[i];
      if (HAS_VALUES) s_vals[p] = val[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const uint32_t p = (uint32_t)k * PASS_THREADS + tid;
    if (p < valid) {
      const uint32_t kv = s_keys[p];
      const uint32_t dst = s_gbase[(kv >> shift) & mask] + p;
      if (WRITE_KEYS) keys_out[dst] = kv;
      if (HAS_VALUES) vals_out[dst] = s_vals[p];
    }
  }
  INSTAG_STAMP(5);
#undef INSTAG_STAMP
}

// second-level partial histograms: out[j][p][d] = sum of partials[b][p][d] over the j-th slice of the blocks b
// (grid = (npass, slices); the pass kernels sum the few slices themselves)
__global__ void __launch_bounds__(SORT_THREADS)
hist_reduce_kernel(const uint32_t* __restrict__ partials, int nblk, int npass, int per_slice,
                   uint32_t* __restrict__ out) {
  const int p = blockIdx.x, j = blockIdx.y, d = threadIdx.x;
  const int b0 = j * per_slice, b1 = min(nblk, b0 + per_slice);
  uint32_t sum = 0;
  constexpr int HB = 16;                      // independent loads per round trip
  for (int c0 = b0; c0 < b1; c0 += HB) {
    uint32_t h[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) h[k] = (c0 + k < b1) ? partials[((size_t)(c0 + k) * npass + p) * RADIX + d] : 0u;
#pragma unroll
    for (int k = 0; k < HB; ++k) sum += h[k];
  }
  out[((size_t)j * npass + p) * RADIX + d] = sum;
}

// Inclusive scan, in depth order, of the kept-tile counts: point_offsets[i] = sum_{j <= i} tiles_touched[order[j]].
// 1024 items per block, single-word look-back (64-bit: flag << 62 | value), ticket order.  state: [0] ticket (as u64),
// [1 ..] look-back words; zero at launch (cleared by the preprocess kernel).
constexpr int SCAN_IPT = 4;
constexpr uint64_t SB_PARTIAL = 1ull << 62, SB_COMPLETE = 2ull << 62, SB_FLAGS = 3ull << 62;

__global__ void __launch_bounds__(SORT_THREADS)
scan_counts_kernel(int N, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tiles_touched,
                   uint32_t* __restrict__ point_offsets, uint64_t* __restrict__ state) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_bid;
  __shared__ uint64_t s_excl;
  const int tid = threadIdx.x;
  if (tid == 0) s_bid = (uint32_t)atomicAdd(reinterpret_cast<unsigned long long*>(state), 1ull);
  __syncthreads();
  const uint32_t bid = s_bid;
  const int base = (int)bid * SORT_THREADS * SCAN_IPT + tid * SCAN_IPT;      // blocked: thread owns 4 consecutive ranks
  uint32_t v[SCAN_IPT], o[SCAN_IPT], tsum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) o[k] = order[min(base + k, N - 1)];        // unconditional, back to back
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) v[k] = tiles_touched[o[k]];
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    v[k] = (base + k < N) ? v[k] : 0u;
    tsum += v[k];
  }
  uint32_t total = 0;
  const uint32_t texcl = block_exclusive_scan_256(tsum, s_w, &total);
  if (tid < 64) {
    // the first wave looks back 64 blocks per round trip (lane l reads block bid-1-l of the current window)
    uint64_t* lb = state + 1;
    if (tid == 0) st_agent64(lb + bid, SB_PARTIAL | total);
    uint64_t excl = 0;
    int64_t b = (int64_t)bid;               // blocks [0, b) are still to be accounted for
    while (b > 0) {
      const int64_t pb = b - 1 - tid;
      uint64_t st = pb >= 0 ? ld_agent64(lb + pb) : SB_COMPLETE;
      int fc = 64;
      bool settled = false;
      for (int polls = 0; polls < SPIN_LIMIT; ++polls) {
        const uint64_t cm = __builtin_amdgcn_ballot_w64((st & SB_FLAGS) == SB_COMPLETE);
        const uint64_t em = __builtin_amdgcn_ballot_w64((st & SB_FLAGS) == 0ull);
        fc = cm ? __builtin_ctzll(cm) : 64;
        const uint64_t need = fc >= 63 ? ~0ull : ((2ull << fc) - 1ull);
        if ((em & need) == 0ull) { settled = true; break; }
        __builtin_amdgcn_s_sleep(1);
        if ((st & SB_FLAGS) == 0ull) st = ld_agent64(lb + pb);
      }
      if (!settled && tid == 0) atomicAdd(&g_sort_stalls, 1u);                 // gave up: the offsets are wrong, say so
      uint64_t v = (tid <= fc) ? (st & ~SB_FLAGS) : 0ull;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      excl += v;
      if (fc < 64) break;
      b -= 64;
    }
    if (tid == 0) {
      st_agent64(lb + bid, SB_COMPLETE | (excl + total));
      s_excl = excl;
    }
  }
  __syncthreads();
  uint32_t run = (uint32_t)s_excl + texcl;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    run += v[k];
    if (base + k < N) point_offsets[base + k] = run;
  }
}

template <int IPT, bool HAS_VALUES, bool WRITE_KEYS>
int launch_pass_t(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                  const uint32_t* count_ptr, uint32_t count_max, int shift, int nbits, const uint32_t* hist,
                  int n_hist, int hist_stride, uint32_t* ticket, uint32_t* lookback, uint64_t* stamps, hipStream_t s) {
  const uint32_t blocks = div_up<uint32_t>(count_max, PASS_THREADS * IPT);
  if (blocks == 0) return INSTAG_OK;
  radix_pass_kernel<IPT, HAS_VALUES, WRITE_KEYS><<<blocks, PASS_THREADS, 0, s>>>(
      keys_in, keys_out, vals_in, vals_out, count_ptr, count_max, shift, nbits, hist, n_hist, hist_stride, ticket,
      lookback, stamps);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // namespace

uint32_t sort_blocks(uint32_t count_max, int ipt) { return div_up<uint32_t>(count_max, PASS_THREADS * ipt); }

int launch_radix_pass(int ipt, bool has_values, bool write_keys, const uint32_t* keys_in, uint32_t* keys_out,
                      const uint32_t* vals_in, uint32_t* vals_out, const uint32_t* count_ptr, uint32_t count_max,
                      int shift, int nbits, const uint32_t* hist, int n_hist, int hist_stride, uint32_t* ticket,
                      uint32_t* lookback, hipStream_t s, uint64_t* stamps) {
#define INSTAG_PASS(I, V, K)                                                                                         \
  if (ipt == I && has_values == V && write_keys == K)                                                                \
    return launch_pass_t<I, V, K>(keys_in, keys_out, vals_in, vals_out, count_ptr, count_max, shift, nbits,          \
                                  hist, n_hist, hist_stride, ticket, lookback, stamps, s);
  INSTAG_PASS(SORT_IPT_DEPTH, true, true)
  INSTAG_PASS(SORT_IPT_DEPTH, true, false)
  INSTAG_PASS(SORT_IPT_TILE, false, true)
  INSTAG_PASS(SORT_IPT_TILE, true, true)
#undef INSTAG_PASS
  set_error("launch_radix_pass: unsupported variant");
  return INSTAG_E_ARG;
}

int launch_hist_reduce(const uint32_t* partials, int nblk, int npass, int slices, uint32_t* out, hipStream_t s) {
  if (npass <= 0 || nblk <= 0) return INSTAG_OK;
  hist_reduce_kernel<<<dim3(npass, slices), SORT_THREADS, 0, s>>>(partials, nblk, npass, div_up(nblk, slices), out);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

int read_sort_stalls(uint32_t* host_out, hipStream_t s, bool synchronize) {
  INSTAG_CHECK_HIP(hipMemcpyFromSymbolAsync(host_out, HIP_SYMBOL(g_sort_stalls), sizeof(uint32_t), 0,
                                            hipMemcpyDeviceToHost, s));
  if (synchronize) INSTAG_CHECK_HIP(hipStreamSynchronize(s));
  return INSTAG_OK;
}

uint32_t* sort_stalls_device_ptr() {
  // (other kernels of the rasterizer that wait for a neighbour -- the segment-wise forward blend -- count into the same
  // sticky word; the address is looked up once per device, outside any capture: warm-up calls come first)
  static uint32_t* cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (cached[dev] == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sort_stalls)) != hipSuccess) return nullptr;
    cached[dev] = (uint32_t*)p;
  }
  return cached[dev];
}

int clear_sort_stalls(hipStream_t s) {
  const uint32_t zero = 0;
  INSTAG_CHECK_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sort_stalls), &zero, sizeof(uint32_t), 0, hipMemcpyHostToDevice, s));
  INSTAG_CHECK_HIP(hipStreamSynchronize(s));
  return INSTAG_OK;
}

int launch_scan_counts(int N, const uint32_t* order, const uint32_t* tiles_touched, uint32_t* point_offsets,
                       uint64_t* state, hipStream_t s) {
  if (N <= 0) return INSTAG_OK;
  scan_counts_kernel<<<div_up(N, SORT_THREADS * SCAN_IPT), SORT_THREADS, 0, s>>>(N, order, tiles_touched,
                                                                                 point_offsets, state);
  INSTAG_CHECK_LAUNCH();
  return INSTAG_OK;
}

}  // namespace instag
