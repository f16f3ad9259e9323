// gsx_intersect.hip — tile binning for gfx950: per-Gaussian tile counts, exclusive offsets,
// (camera | tile | depth-bits) key emission, stable radix sort, per-tile offsets.
//
// Replaces gsplat::intersect_tile / gsplat::intersect_offset (reference: gsplat/Intersect.cpp:15-137,
// kernels gsplat/IntersectTile.cu:23-114 (count + emit), :206-252 (offsets), CUB radix sort :290-342,
// at::cumsum gsplat/Intersect.cpp:75).
//
// All integer / bit work, bit-exact by construction:
//   key   = (cid << (32 + tile_n_bits)) | (tile_id << 32) | float_bits(depth)     (int64)
//   value = flatten index c*N + n                                                 (int32)
//   tile_n_bits = bit_width(n_tiles), cam_n_bits = bit_width(C)   (== floor(log2)+1 upstream)
// Sorting is the device-wide LSD radix sort of rocPRIM restricted to the low
// 32+tile_n_bits+cam_n_bits bits: stable, so equal (tile, depth) keys keep ascending flatten index,
// exactly like CUB's SortPairs upstream.  The scan is rocPRIM's single-pass decoupled look-back.
// The per-tile offsets are a lower_bound per tile (8 160 tiles @1080p) instead of the reference's
// one-thread-per-intersection boundary detection with serial gap filling; same output.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "gsx_device.hpp"

namespace gsx {

void set_error(const char* msg);
int check_launch(const char* what);
const char* test_switch(const char* name);

constexpr int ISECT_BLOCK = 256;

GSX_DEV uint32_t f2u_sat(float v) { return v > 0.f ? (v >= 4294967296.f ? 0xFFFFFFFFu : (uint32_t)v) : 0u; }
static inline uint32_t bit_width_u32(uint32_t v) { uint32_t n = 0; while (v) { ++n; v >>= 1; } return n; }

// tile AABB of one projected Gaussian: [x0,x1) x [y0,y1) in tile units (IntersectTile.cu:54-76)
GSX_DEV bool tile_rect(const float* __restrict__ means2d, const int32_t* __restrict__ radii, size_t idx, float tile_size,
                       uint32_t tw, uint32_t th, uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1) {
    const int2 r = reinterpret_cast<const int2*>(radii)[idx];
    const float rx = (float)r.x, ry = (float)r.y;
    if (rx <= 0.f || ry <= 0.f) return false;
    const float2 m = reinterpret_cast<const float2*>(means2d)[idx];
    const float trx = rx / tile_size, try_ = ry / tile_size;
    const float tx = m.x / tile_size, ty = m.y / tile_size;
    x0 = min(f2u_sat(floorf(tx - trx)), tw);
    y0 = min(f2u_sat(floorf(ty - try_)), th);
    x1 = min(f2u_sat(ceilf(tx + trx)), tw);
    y1 = min(f2u_sat(ceilf(ty + try_)), th);
    return true;
}

__global__ __launch_bounds__(ISECT_BLOCK) void isect_count_kernel(uint32_t total, const float* __restrict__ means2d,
                                                                  const int32_t* __restrict__ radii, float tile_size,
                                                                  uint32_t tw, uint32_t th,
                                                                  int32_t* __restrict__ tiles_per_gauss) {
    const uint32_t idx = blockIdx.x * ISECT_BLOCK + threadIdx.x;
    if (idx >= total) return;
    uint32_t x0, y0, x1, y1;
    int32_t n = 0;
    if (tile_rect(means2d, radii, idx, tile_size, tw, th, x0, y0, x1, y1)) n = (int32_t)((y1 - y0) * (x1 - x0));
    tiles_per_gauss[idx] = n;
}

__global__ __launch_bounds__(ISECT_BLOCK) void isect_fill_kernel(uint32_t total, uint32_t N,
                                                                 const float* __restrict__ means2d,
                                                                 const int32_t* __restrict__ radii,
                                                                 const float* __restrict__ depths,
                                                                 const int64_t* __restrict__ cum, float tile_size,
                                                                 uint32_t tw, uint32_t th, uint32_t tile_n_bits,
                                                                 int64_t* __restrict__ isect_ids,
                                                                 int32_t* __restrict__ flatten_ids,
                                                                 const int64_t* __restrict__ camera_ids) {
    const uint32_t idx = blockIdx.x * ISECT_BLOCK + threadIdx.x;
    if (idx >= total) return;
    uint32_t x0, y0, x1, y1;
    if (!tile_rect(means2d, radii, idx, tile_size, tw, th, x0, y0, x1, y1)) return;
    const int64_t cid = camera_ids ? camera_ids[idx] : (int64_t)(idx / N);   // packed [nnz] layout: IntersectTile.cu:85-93
    const int64_t cid_enc = cid << (32 + tile_n_bits);
    const int64_t depth_enc = (int64_t)__float_as_uint(depths[idx]);
    int64_t cur = (idx == 0) ? 0 : cum[idx - 1];
    for (uint32_t i = y0; i < y1; ++i)
        for (uint32_t j = x0; j < x1; ++j) {
            const int64_t tile_id = (int64_t)i * tw + j;
            isect_ids[cur] = cid_enc | (tile_id << 32) | depth_enc;
            flatten_ids[cur] = (int32_t)idx;
            ++cur;
        }
}

__global__ __launch_bounds__(ISECT_BLOCK) void isect_offset_kernel(int64_t n_isects, const int64_t* __restrict__ isect_ids,
                                                                   uint32_t C, uint32_t n_tiles, uint32_t tile_n_bits,
                                                                   int32_t* __restrict__ offsets) {
    const uint32_t t = blockIdx.x * ISECT_BLOCK + threadIdx.x;
    if (t >= C * n_tiles) return;
    const int64_t want = ((int64_t)(t / n_tiles) << tile_n_bits) | (int64_t)(t % n_tiles);
    int64_t lo = 0, hi = n_isects;  // first index whose (cam,tile) >= want
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((isect_ids[mid] >> 32) < want) lo = mid + 1; else hi = mid;
    }
    offsets[t] = (int32_t)lo;
}

// ---- binned path (sort == true): tile-major binning with LDS histograms + a per-tile LDS sort -----------------------------
// The reference sorts ALL intersections by a 46-bit (camera | tile | depth) key: six device-wide radix passes over 12 B
// pairs (~0.33 ms at 3.4 M intersections on MI355X, latency- not bandwidth-bound).  The same total order
// (camera, tile, depth bits, flatten index) is produced here with ONE scatter of the data and a sort that never leaves LDS:
//   1. count    BIN_NB blocks per camera, each owning every BIN_NB-th chunk of 1024 consecutive Gaussians, histogram their tile rectangles in
//               LDS (one 32-bit counter per tile: 32 KB at 1080p, 127 KB at 4K) and store the histogram;
//   2. prefix   per tile, an exclusive prefix over the blocks (-> each block's first slot inside the tile) and the tile total;
//   3. scan     exclusive scan of the C*tiles totals = the reference's isect_offsets (+ the grand total = n_isects);
//   4. scatter  the same blocks reload (tile offset + block prefix) as LDS cursors, claim slots with returning LDS atomics and
//               write 64-bit keys (depth bits << idx_bits | flatten index) into their tile's segment — unordered inside it;
//   5. sort     one wave per tile up to 1024 keys: a bitonic network on the keys in REGISTERS (tile_sort_wave_regs_kernel: exchanges through
//               DPP and the LDS crossbar, no LDS memory; round 4 — the LDS merge sort it replaces still serves calls that want isect_ids);
//               a 256-thread block up to 4096 keys, a 1024-thread block with 132 KB of LDS up to 16384: merge sort of the segment in LDS; larger segments: 16384-key LDS-sorted chunks + merge-path passes by many
//               blocks (giant_* kernels).  Then flatten_ids = low bits, isect_ids (on request) = (camera|tile) << 32 | depth bits.
//               Keys are unique, so the result is exactly the stable sort upstream.
//      Frames with heavy tiles: the ranked variant (4-byte depth ranks as keys, bitmap sort for tiles above 4096 keys), further down.
// No global atomics anywhere (device-scope atomics resolve at the memory side on MI355X: ~14 G/s measured, 0.24 ms for the
// 3.4 M increments of a naive tile counter).  Tile grids above 36 K tiles per camera (LDS) use the device-wide sort instead.
constexpr uint32_t BIN_NB_MAX = 1024;     // Gaussian slices (blocks) per camera: bin_nb() below, a multiple of 32
constexpr uint32_t BIN_MAX_TILES = 36864; // 144 KB of LDS counters
constexpr int TSORT_CAP = 4096;           // keys sorted in LDS per block (32 KB)
constexpr int TSORT_BIG_CAP = 16384;      // keys sorted in LDS by a 1024-thread block (132 KB): the heavy tiles of dense scenes
constexpr int TSORT_WAVE_CAP = 1024;      // keys sorted by one wave without block barriers (8 KB); measured faster than a block up to here
constexpr int BIN_BLOCK = 1024;           // count / scatter: 16 waves share one LDS counter array
constexpr uint32_t BIN_WIDE = 16;         // count / scatter: tile rectangles above this many tiles are walked by the whole wave

__global__ __launch_bounds__(BIN_BLOCK) void bin_count_kernel(uint32_t N, uint32_t per_block, const float* __restrict__ means2d,
                                                                const int32_t* __restrict__ radii, float tile_size, uint32_t tw,
                                                                uint32_t th, int32_t* __restrict__ tiles_per_gauss,
                                                                uint32_t* __restrict__ block_hist, unsigned long long* __restrict__ agg, uint32_t n_agg) {
    const uint32_t BIN_NB = gridDim.x;
    extern __shared__ uint32_t s_hist[];
    const uint32_t n_tiles = tw * th, c = blockIdx.y, b = blockIdx.x;
    // the aggregate words of bin_prefix_scan_kernel's look-back start at "not published" (0): cleared here, one launch earlier, instead of by a memset launch
    if (agg != nullptr && b == 0u && c == 0u)
        for (uint32_t t = threadIdx.x; t < n_agg; t += BIN_BLOCK) agg[t] = 0ull;
    for (uint32_t t = threadIdx.x; t < n_tiles; t += BIN_BLOCK) s_hist[t] = 0u;
    __syncthreads();
    // Block b takes the chunks b, b + BIN_NB, b + 2 BIN_NB, ... of BIN_BLOCK consecutive Gaussians (not one contiguous slice): with the
    // Gaussians stored in a spatially coherent order (layout.py) a contiguous slice of near, screen-filling Gaussians would own most of
    // a heavy frame's keys (measured on the trained garden stand-in: 1.05 vs 0.93 ms against a shuffled model); a chunk is still
    // spatially compact, so its keys still land in few tile segments.  bin_scatter_kernel walks the same chunks.
    (void)per_block;
    const uint32_t n1 = N;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = b * BIN_BLOCK; base < n1; base += BIN_NB * BIN_BLOCK) {   // (wave-uniform trip count: the wide rectangles below are shared by the wave)
        const uint32_t n = base + threadIdx.x;
        const size_t idx = (size_t)c * N + n;
        uint32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        const bool hit = n < n1 && tile_rect(means2d, radii, idx, tile_size, tw, th, x0, y0, x1, y1);
        const uint32_t cnt = hit ? (y1 - y0) * (x1 - x0) : 0u;
        if (cnt != 0u && cnt <= BIN_WIDE)
            for (uint32_t i = y0; i < y1; ++i)
                for (uint32_t j = x0; j < x1; ++j) atomicAdd(&s_hist[i * tw + j], 1u);
        // a rectangle of more than BIN_WIDE tiles is walked by all 64 lanes (one thread looping over the 4 000 tiles of a
        // screen-filling Gaussian held its whole block for hundreds of microseconds)
        for (uint64_t wide = __ballot(cnt > BIN_WIDE); wide != 0ull; wide &= wide - 1ull) {
            const int src = __builtin_ctzll(wide);
            const uint32_t wx0 = __builtin_amdgcn_readlane(x0, src), wy0 = __builtin_amdgcn_readlane(y0, src);
            const uint32_t ww = __builtin_amdgcn_readlane(x1, src) - wx0, total = __builtin_amdgcn_readlane(cnt, src);
            for (uint32_t k = lane; k < total; k += 64u) atomicAdd(&s_hist[(wy0 + k / ww) * tw + wx0 + k % ww], 1u);
        }
        if (tiles_per_gauss && n < n1) tiles_per_gauss[idx] = (int32_t)cnt;
    }
    __syncthreads();
    uint32_t* out = block_hist + ((size_t)c * BIN_NB + b) * n_tiles;
    for (uint32_t t = threadIdx.x; t < n_tiles; t += BIN_BLOCK) out[t] = s_hist[t];
}

// block_hist[c][b][t] -> exclusive prefix over b; tile_counts[c*n_tiles + t] = total.
// 32 tiles x 8 groups of 32 blocks per workgroup: every thread keeps its 32 counts in registers, the 8 group sums of a tile meet
// in LDS (8 160 tiles alone would be 32 workgroups of strictly serial 256-step columns).
__global__ __launch_bounds__(BIN_NB_MAX) void bin_prefix_kernel(uint32_t C, uint32_t n_tiles, uint32_t* __restrict__ block_hist,
                                                                 uint32_t* __restrict__ tile_counts) {
    const uint32_t BIN_NB = blockDim.x;   // BIN_NB / 32 groups of 32 slices, one thread per (tile, group)
    __shared__ uint32_t s_sum[BIN_NB_MAX / 32][32];
    const uint32_t tl = threadIdx.x & 31u, grp = threadIdx.x >> 5;
    const uint32_t g = blockIdx.x * 32u + tl;
    const bool ok = g < C * n_tiles;
    const uint32_t c = ok ? g / n_tiles : 0u, t = ok ? g - c * n_tiles : 0u;
    uint32_t* col = block_hist + ((size_t)c * BIN_NB + grp * 32u) * n_tiles + t;
    uint32_t v[32], sum = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        v[k] = ok ? col[(size_t)k * n_tiles] : 0u;
        sum += v[k];
    }
    s_sum[grp][tl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (uint32_t k = 0; k < BIN_NB / 32; ++k) {
        const uint32_t sv = s_sum[k][tl];
        run += k < grp ? sv : 0u;
        total += sv;
    }
    if (!ok) return;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        col[(size_t)k * n_tiles] = run;
        run += v[k];
    }
    if (grp == 0) tile_counts[g] = total;
}

// Round 6: bin_prefix_kernel and bin_scan_kernel in ONE launch.  The scan of the C*tiles totals was a single 1024-thread block behind the prefix kernel:
// 10.5 us of launch + latency for 32 KB.  Here every workgroup (32 tiles) publishes the sum of its 32 totals and its largest total in one 8-byte word and
// takes the sum of its predecessors' words (a look-back over ALL of them: a few hundred words, every lane a few): the exclusive offsets of its tiles follow
// from a 32-lane scan, and the last workgroup — which then holds the grand total and the largest segment — writes what bin_scan wrote (offsets[n],
// max_count, the guarded verdict, the host's word).  The words travel as RELAXED device-scope atomics (they resolve at the memory side: no release / L2
// write-back, which costs ~3.5 us per block on this part — why rounds 5 and 6 first left the two launches apart); nothing but the word itself
// has to be visible to the other workgroups.  Workgroups are dispatched in order, so a predecessor is always running or done when its successor waits.
//   word: bit 63 = published | bits 62..36 = min(largest total, 2^27 - 1) | bits 35..0 = min(sum, 2^36 - 1)
__global__ __launch_bounds__(BIN_NB_MAX) void bin_prefix_scan_kernel(uint32_t C, uint32_t n_tiles, uint32_t* __restrict__ block_hist, unsigned long long* __restrict__ agg,
                                                                      int32_t* __restrict__ offsets, uint32_t* __restrict__ max_count, int64_t capacity, int64_t seg_bound,
                                                                      int32_t* __restrict__ lists_status, unsigned long long* __restrict__ host_word) {
    const uint32_t BIN_NB = blockDim.x;
    __shared__ uint32_t s_sum[BIN_NB_MAX / 32][32];
    const uint32_t tl = threadIdx.x & 31u, grp = threadIdx.x >> 5;
    const uint32_t nseg = C * n_tiles;
    const uint32_t g = blockIdx.x * 32u + tl;
    const bool ok = g < nseg;
    const uint32_t c = ok ? g / n_tiles : 0u, t = ok ? g - c * n_tiles : 0u;
    uint32_t* col = block_hist + ((size_t)c * BIN_NB + grp * 32u) * n_tiles + t;
    uint32_t v[32], sum = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        v[k] = ok ? col[(size_t)k * n_tiles] : 0u;
        sum += v[k];
    }
    s_sum[grp][tl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (uint32_t k = 0; k < BIN_NB / 32; ++k) {
        const uint32_t sv = s_sum[k][tl];
        run += k < grp ? sv : 0u;
        total += sv;
    }
    // ---- wave 0 first (its word is what the successors wait for): lanes 0..31 (grp 0) hold the 32 tile totals of this workgroup
    if (threadIdx.x < 64u) {
        const uint32_t lane = threadIdx.x;
        const unsigned long long mine = (lane < 32u && ok) ? (unsigned long long)total : 0ull;
        unsigned long long incl = mine;
        uint32_t mx = (uint32_t)mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(incl, o);
            if ((int)lane >= o) incl += up;
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        }
        const unsigned long long wg_sum = __shfl(incl, 63);
        if (lane == 0u) {
            const unsigned long long word = (1ull << 63) | ((unsigned long long)min(mx, (1u << 27) - 1u) << 36) | min(wg_sum, (1ull << 36) - 1ull);
            __hip_atomic_store(agg + blockIdx.x, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // look-back: the words of all predecessors (lane l takes l, l + 64, ...)
        unsigned long long before = 0ull;
        uint32_t gmax = mx;
        for (uint32_t j = lane; j < blockIdx.x; j += 64u) {
            unsigned long long w;
            do { w = __hip_atomic_load(agg + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w >> 63) == 0ull);
            before += w & ((1ull << 36) - 1ull);
            gmax = max(gmax, (uint32_t)((w >> 36) & ((1u << 27) - 1u)));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            before += __shfl_xor(before, o);
            gmax = max(gmax, (uint32_t)__shfl_xor((int)gmax, o));
        }
        // exclusive offsets of this workgroup's tiles; saturating like bin_scan (a frame beyond 2^31 - 1 intersections: offsets INT32_MAX, total -1)
        const unsigned long long excl = before + incl - mine;
        if (lane < 32u && ok) offsets[g] = (int32_t)(excl > 0x7FFFFFFFull ? 0x7FFFFFFFull : excl);
        if (blockIdx.x == gridDim.x - 1u && lane == 0u) {   // the last workgroup holds the grand total and the largest segment
            const unsigned long long grand = before + wg_sum;
            offsets[nseg] = grand > 0x7FFFFFFFull ? -1 : (int32_t)grand;
            if (max_count != nullptr) *max_count = gmax;
            if (lists_status != nullptr) {
                const bool fits = grand <= (unsigned long long)capacity && (seg_bound <= 0 || (int64_t)gmax <= seg_bound);
                *lists_status = fits ? (int32_t)grand : -1;
            }
            if (host_word != nullptr) {
                const unsigned long long lo = grand > 0x7FFFFFFFull ? 0xFFFFFFFFull : grand;
                __hip_atomic_store(host_word, lo | ((unsigned long long)(max_count != nullptr ? gmax : 0u) << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (ok) {   // the per-slice prefixes of the tile (what bin_scatter adds to the tile's offset)
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            col[(size_t)k * n_tiles] = run;
            run += v[k];
        }
    }
}

// K = uint64_t: keys (depth bits << idx_bits | flatten index); K = uint32_t: the Gaussian's rank (ranked variant), read from `ranks`.
template <typename K>
__global__ __launch_bounds__(BIN_BLOCK) void bin_scatter_kernel(uint32_t N, uint32_t per_block, const float* __restrict__ means2d,
                                                                  const int32_t* __restrict__ radii, const float* __restrict__ depths,
                                                                  const uint32_t* __restrict__ ranks, float tile_size, uint32_t tw, uint32_t th,
                                                                  uint32_t idx_bits, const int32_t* __restrict__ tile_offsets,
                                                                  const uint32_t* __restrict__ block_hist, K* __restrict__ keys,
                                                                  uint32_t capacity) {
    extern __shared__ uint32_t s_cur[];
    const uint32_t BIN_NB = gridDim.x;
    const uint32_t n_tiles = tw * th, c = blockIdx.y, b = blockIdx.x;
    const uint32_t* pre = block_hist + ((size_t)c * BIN_NB + b) * n_tiles;
    const int32_t* off = tile_offsets + (size_t)c * n_tiles;
    for (uint32_t t = threadIdx.x; t < n_tiles; t += BIN_BLOCK) s_cur[t] = (uint32_t)off[t] + pre[t];
    __syncthreads();
    (void)per_block;
    const uint32_t n1 = N;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = b * BIN_BLOCK; base < n1; base += BIN_NB * BIN_BLOCK) {   // the chunks bin_count_kernel gave this block
        const uint32_t n = base + threadIdx.x;
        const size_t idx = (size_t)c * N + n;
        uint32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        const bool hit = n < n1 && tile_rect(means2d, radii, idx, tile_size, tw, th, x0, y0, x1, y1);
        const uint32_t cnt = hit ? (y1 - y0) * (x1 - x0) : 0u;
        uint64_t key = 0ull;   // (held in 64 bits either way: the wide path broadcasts both halves)
        if (hit) key = sizeof(K) == 8 ? ((uint64_t)__float_as_uint(depths[idx]) << idx_bits) | (uint64_t)idx : (uint64_t)ranks[idx];
        if (cnt != 0u && cnt <= BIN_WIDE)
            for (uint32_t i = y0; i < y1; ++i)
                for (uint32_t j = x0; j < x1; ++j) {
                    const uint32_t pos = atomicAdd(&s_cur[i * tw + j], 1u);
                    if (pos < capacity) keys[pos] = (K)key;  // capacity < n_isects only when an optimistic caller under-estimated: it re-runs
                }
        for (uint64_t wide = __ballot(cnt > BIN_WIDE); wide != 0ull; wide &= wide - 1ull) {   // wide rectangles: all 64 lanes (see bin_count_kernel)
            const int src = __builtin_ctzll(wide);
            const uint32_t wx0 = __builtin_amdgcn_readlane(x0, src), wy0 = __builtin_amdgcn_readlane(y0, src);
            const uint32_t ww = __builtin_amdgcn_readlane(x1, src) - wx0, total = __builtin_amdgcn_readlane(cnt, src);
            const uint64_t wkey = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), src) << 32) |
                                  (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, src);   // (readlane returns int)
            for (uint32_t k0 = lane; k0 < total; k0 += 256u) {   // four returning LDS atomics in flight per lane
                uint32_t pos[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t k = k0 + 64u * u;
                    pos[u] = k < total ? atomicAdd(&s_cur[(wy0 + k / ww) * tw + wx0 + k % ww], 1u) : 0xFFFFFFFFu;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u)
                    if (pos[u] < capacity) keys[pos[u]] = (K)wkey;   // (capacity <= 2^31 - 1)
            }
        }
    }
}

// One wave per segment of up to TSORT_WAVE_CAP keys, no s_barrier (a single wavefront's LDS operations are ordered).
// Merge sort instead of a bitonic network: the network moves all m keys through LDS log2(m)(log2(m)+1)/2 times (45 times at
// m = 512: the kernel was LDS-bandwidth bound), the merge sort log2(64) + 1 times — every lane sorts its E keys in registers,
// then six merge passes; in each pass a lane finds its merge-path split by binary search and produces E consecutive outputs.
// LDS layout: one pad key after every 32 (index i lives at i + (i >> 5)).  A lane's keys sit E apart from its neighbour's and
// the merge cursors of neighbouring lanes about E/2 apart: without the pad, lanes 32/E (or 64/E) apart hit the same banks
// (8 B keys, 64 banks x 4 B) — 72 % of the LDS cycles of this kernel were bank-conflict cycles at E = 8..16.
GSX_DEV int spad(int i) { return i + (i >> 5); }
// (4-byte keys — the ranked variant below — : one pad key after every 64, the same two-bank shift)
template <typename K> GSX_DEV int spad_of(int i) { return sizeof(K) == 8 ? i + (i >> 5) : i + (i >> 6); }

template <int NT>
GSX_DEV void group_sync() {
    if (NT == 64) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // one wavefront: its LDS operations are ordered
    else __syncthreads();
}

// Sort NT * E keys in LDS with NT threads (NT = 64: one wave, no barrier; NT = 256: a block).  Thread t first sorts keys
// [t E, (t+1) E) in registers, then log2(NT) merge passes: it finds the merge-path split of its E outputs by binary search and
// merges them sequentially.
template <int E, int NT, typename K = uint64_t>
GSX_DEV void merge_sort_lds(K* s, int t) {
    constexpr K kMax = ~(K)0;
    K r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = s[spad_of<K>(t * E + e)];
    // Batcher's odd-even merge sort network on the thread's own keys (E a power of two: 1 / 5 / 19 / 63 compare-exchanges for
    // E = 2 / 4 / 8 / 16; all indices are compile-time constants after unrolling)
#pragma unroll
    for (int p = 1; p < E; p <<= 1)
#pragma unroll
        for (int k = p; k >= 1; k >>= 1)
#pragma unroll
            for (int j = k % p; j + k < E; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < E && (i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                        const K x = r[i + j], y = r[i + j + k];
                        r[i + j] = x < y ? x : y;
                        r[i + j + k] = x < y ? y : x;
                    }
#pragma unroll
    for (int e = 0; e < E; ++e) s[spad_of<K>(t * E + e)] = r[e];
    group_sync<NT>();
    for (int run = E; run < NT * E; run <<= 1) {
        const int o = t * E;
        const int pair0 = o & ~(2 * run - 1);
        const int d = o - pair0;  // diagonal of this thread's first output inside the pair of runs
        const int A = pair0, B = pair0 + run;  // run starts
        int lo = max(0, d - run), hi = min(d, run);
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s[spad_of<K>(A + mid)] <= s[spad_of<K>(B + d - 1 - mid)]) lo = mid + 1; else hi = mid;
        }
        int ai = lo, bi = d - lo;
        K a = ai < run ? s[spad_of<K>(A + ai)] : kMax, b = bi < run ? s[spad_of<K>(B + bi)] : kMax;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool ta = a <= b;
            r[e] = ta ? a : b;
            const int nxt = ta ? ++ai : ++bi;
            const K v = nxt < run ? s[spad_of<K>((ta ? A : B) + nxt)] : kMax;
            a = ta ? v : a;
            b = ta ? b : v;
        }
        group_sync<NT>();  // every thread has finished reading this pass
#pragma unroll
        for (int e = 0; e < E; ++e) s[spad_of<K>(t * E + e)] = r[e];
        group_sync<NT>();
    }
}

// What a sorted key stands for.  KeyDepthIdx: (depth bits << idx_bits) | flatten index, 8 bytes — self-contained.  KeyRank: the
// position of the Gaussian in the frame-wide (depth bits, flatten index) order, 4 bytes — see "ranked variant" further down.
struct KeyDepthIdx {
    using T = uint64_t;
    static constexpr bool kDeferred = false;   // the sort kernels write flatten_ids / isect_ids themselves
    uint32_t idx_bits;
    GSX_DEV int32_t id(T k) const { return (int32_t)(k & ((1ull << idx_bits) - 1ull)); }
    GSX_DEV int64_t depth_bits(T k, int32_t) const { return (int64_t)(k >> idx_bits); }
};
struct KeyRank {
    using T = uint32_t;
    static constexpr bool kDeferred = true;    // the sort kernels leave sorted ranks in place; ranked_finalize_kernel turns them into ids
    const uint32_t* __restrict__ order;   // rank -> flatten index
    const float* __restrict__ depths;
    GSX_DEV int32_t id(T k) const { return (int32_t)order[k]; }
    GSX_DEV int64_t depth_bits(T, int32_t id) const { return (int64_t)__float_as_uint(depths[id]); }
};

template <class KT>
__global__ __launch_bounds__(256) void tile_sort_wave_kernel(uint32_t n_segments, uint32_t n_tiles, uint32_t tile_n_bits, KT kt,
                                                             const int32_t* __restrict__ tile_offsets, typename KT::T* __restrict__ keys,
                                                             int32_t* __restrict__ flatten_ids, int64_t* __restrict__ isect_ids,
                                                             int64_t capacity) {
    using K = typename KT::T;
    __shared__ K s_all[4][TSORT_WAVE_CAP + TSORT_WAVE_CAP / 32];
    const uint32_t seg = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (seg >= n_segments) return;
    const int64_t begin = tile_offsets[seg];
    const int n = (int)(tile_offsets[seg + 1] - tile_offsets[seg]);
    if (n <= 0 || n > TSORT_WAVE_CAP || begin + n > capacity) return;
    K* s_keys = s_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    int m = 128;
    while (m < n) m <<= 1;
    for (int i = lane; i < m; i += 64) s_keys[spad_of<K>(i)] = i < n ? keys[begin + i] : ~(K)0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    switch (m) {
    case 128: merge_sort_lds<2, 64, K>(s_keys, lane); break;
    case 256: merge_sort_lds<4, 64, K>(s_keys, lane); break;
    case 512: merge_sort_lds<8, 64, K>(s_keys, lane); break;
    default: merge_sort_lds<16, 64, K>(s_keys, lane); break;
    }
    const int64_t cam_tile = (((int64_t)(seg / n_tiles) << tile_n_bits) | (int64_t)(seg % n_tiles)) << 32;
    for (int i = lane; i < n; i += 64) {
        const K k = s_keys[spad_of<K>(i)];
        if (KT::kDeferred) {
            keys[begin + i] = k;   // in place: the whole segment was read before the sort
        } else {
            const int32_t id = kt.id(k);
            flatten_ids[begin + i] = id;
            if (isect_ids) isect_ids[begin + i] = cam_tile | kt.depth_bits(k, id);
        }
    }
}

// ---- one wave per segment, keys in REGISTERS: bitonic network over 64 lanes x E keys, exchanges through DPP / the LDS crossbar ----
// The LDS merge sort above moves every key through LDS memory log2(64) + 1 times and half of its LDS cycles are bank conflicts (cursor
// positions are data dependent).  A bitonic network needs no memory at all: lane l holds E keys; an exchange at element distance >= E is
// "same register, lane l ^ j" — quad_perm / row_ror DPP moves for j = 1, 2, 8, ds_swizzle / ds_bpermute (the LDS crossbar: no bank, no
// conflict) for j = 4, 16, 32 — and below E it is a compare-exchange between two registers of the lane.  21 cross-lane steps and
// 6 log2(E) + log2(E)(log2(E)+1)/2 local ones; the direction of a step is a compile-time constant or a lane bit.  The input may
// sit in the registers in ANY order (it is unsorted), so the keys are loaded striped (coalesced); the output is in blocked order
// (element lane E + e) and takes one trip through LDS to leave coalesced.
template <int M> GSX_DEV uint32_t lane_xor_u32(uint32_t v, uint32_t lane) {
    if (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
    if (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);        // quad_perm [2,3,0,1]
    if (M == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);                         // lane ^ 4
    if (M == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);       // row_ror:8 = lane ^ 8 inside the row
    if (M == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                        // lane ^ 16
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), (int)v);                  // lane ^ 32
}
template <int M> GSX_DEV uint64_t lane_xor(uint64_t v, uint32_t lane) {
    return ((uint64_t)lane_xor_u32<M>((uint32_t)(v >> 32), lane) << 32) | (uint64_t)lane_xor_u32<M>((uint32_t)v, lane);
}
template <int M> GSX_DEV uint32_t lane_xor(uint32_t v, uint32_t lane) { return lane_xor_u32<M>(v, lane); }

// (64-bit keys compare with v_cmp_lt_u64: 6.5 cycles per wave instruction on gfx950 against 2 x 4.3 for a v_sub_co / v_subb_co borrow chain,
//  tools/valu_probe.hip; a key step = 2 moves (DPP 4.3 each, ds_swizzle 9.0) + the compare + 2 v_cndmask (4.6) ~ 25-34 cycles)
template <typename K> GSX_DEV bool key_lt(K a, K b) { return a < b; }

// one cross-lane step: partner = lane ^ JL; `keep_min` (per lane): this lane keeps the smaller key of each pair
template <int E, int JL, typename K> GSX_DEV void bitonic_cross(K (&r)[E], uint32_t lane, bool keep_min) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const K o = lane_xor<JL>(r[e], lane);
        const bool take = key_lt(o, r[e]) == keep_min;   // (equal keys are the padding: either choice is the same value)
        r[e] = take ? o : r[e];
    }
}
// the local steps of a stage: element distances JE = E/2 .. 1 inside the lane, all in direction `asc` (per lane)
template <int E, typename K> GSX_DEV void bitonic_local(K (&r)[E], bool asc) {
#pragma unroll
    for (int je = E / 2; je >= 1; je >>= 1)
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((e & je) == 0) {
                const K x = r[e], y = r[e | je];
                const bool sw = key_lt(y, x) == asc;
                r[e] = sw ? y : x;
                r[e | je] = sw ? x : y;
            }
}
template <int E, typename K> GSX_DEV void bitonic_sort_regs(K (&r)[E], uint32_t lane) {
    // stages of size k (elements) <= E: inside the lane, direction by the element index (compile time) — and by lane bit 0 at k = E
#pragma unroll
    for (int k = 2; k <= E; k <<= 1)
#pragma unroll
        for (int je = k / 2; je >= 1; je >>= 1)
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((e & je) == 0) {
                    const bool asc = k < E ? ((e & k) == 0) : ((lane & 1u) == 0u);
                    const K x = r[e], y = r[e | je];
                    const bool sw = key_lt(y, x) == asc;
                    r[e] = sw ? y : x;
                    r[e | je] = sw ? x : y;
                }
    // stages of 2, 4, ..., 64 lanes (k = 2E .. 64E elements): direction = lane bit KL (the last stage: ascending everywhere)
#define GSX_BITONIC_STAGE(KL, ...)                                                        \
    {                                                                                     \
        const bool asc = KL >= 64 || (lane & (uint32_t)KL) == 0u;                         \
        __VA_ARGS__                                                                       \
        bitonic_local<E>(r, asc);                                                         \
    }
#define GSX_X(JL) bitonic_cross<E, JL>(r, lane, asc == ((lane & (uint32_t)JL) == 0u));
    GSX_BITONIC_STAGE(2, GSX_X(1))
    GSX_BITONIC_STAGE(4, GSX_X(2) GSX_X(1))
    GSX_BITONIC_STAGE(8, GSX_X(4) GSX_X(2) GSX_X(1))
    GSX_BITONIC_STAGE(16, GSX_X(8) GSX_X(4) GSX_X(2) GSX_X(1))
    GSX_BITONIC_STAGE(32, GSX_X(16) GSX_X(8) GSX_X(4) GSX_X(2) GSX_X(1))
    GSX_BITONIC_STAGE(64, GSX_X(32) GSX_X(16) GSX_X(8) GSX_X(4) GSX_X(2) GSX_X(1))
#undef GSX_X
#undef GSX_BITONIC_STAGE
}

// (k = E needs the lane's direction at the END of the local sort: for E == 1 there is no local stage)
// (isect_ids != nullptr — KeyDepthIdx only, round 6: the sorted depth bits take a second trip through the same LDS words and leave as
//  (camera | tile) << 32 | depth bits, what gsplat::intersect_tile returns to a reference build; the LDS merge sort used to serve those calls)
template <int E, class KT>
GSX_DEV void tile_sort_regs_segment(const KT& kt, typename KT::T* __restrict__ keys, int32_t* __restrict__ flatten_ids, int64_t begin, int n,
                                    uint32_t lane, uint32_t* s_out, int64_t* __restrict__ isect_ids = nullptr, int64_t cam_tile = 0) {
    using K = typename KT::T;
    K r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {   // striped: coalesced, and the network does not care where an unsorted key starts
        const int i = e * 64 + (int)lane;
        r[e] = i < n ? keys[begin + i] : ~(K)0;
    }
    bitonic_sort_regs<E>(r, lane);
    // blocked (lane E + e) -> striped through LDS (pad: one word after every 32), then coalesced stores
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = (int)lane * E + e;
        s_out[i + (i >> 5)] = KT::kDeferred ? (uint32_t)r[e] : (uint32_t)kt.id(r[e]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + (int)lane;
        if (i < n) {
            const uint32_t v = s_out[i + (i >> 5)];
            if (KT::kDeferred) keys[begin + i] = (K)v;   // in place: the whole segment was read before the sort
            else flatten_ids[begin + i] = (int32_t)v;
        }
    }
    if constexpr (!KT::kDeferred) if (isect_ids != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = (int)lane * E + e;
            s_out[i + (i >> 5)] = (uint32_t)kt.depth_bits(r[e], 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = e * 64 + (int)lane;
            if (i < n) isect_ids[begin + i] = cam_tile | (int64_t)s_out[i + (i >> 5)];
        }
    }
}

template <class KT>
__global__ __launch_bounds__(256) void tile_sort_wave_regs_kernel(uint32_t n_segments, KT kt, const int32_t* __restrict__ tile_offsets,
                                                                  typename KT::T* __restrict__ keys, int32_t* __restrict__ flatten_ids,
                                                                  int64_t capacity, int64_t* __restrict__ isect_ids, uint32_t n_tiles, uint32_t tile_n_bits) {
    __shared__ uint32_t s_all[4][TSORT_WAVE_CAP + TSORT_WAVE_CAP / 32];
    const uint32_t seg = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (seg >= n_segments) return;
    const int64_t begin = tile_offsets[seg];
    const int n = (int)(tile_offsets[seg + 1] - tile_offsets[seg]);
    if (n <= 0 || n > TSORT_WAVE_CAP || begin + n > capacity) return;
    uint32_t* s_out = s_all[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t cam_tile = isect_ids ? ((((int64_t)(seg / n_tiles) << tile_n_bits) | (int64_t)(seg % n_tiles)) << 32) : 0;
    if (n <= 64) tile_sort_regs_segment<1>(kt, keys, flatten_ids, begin, n, lane, s_out, isect_ids, cam_tile);
    else if (n <= 128) tile_sort_regs_segment<2>(kt, keys, flatten_ids, begin, n, lane, s_out, isect_ids, cam_tile);
    else if (n <= 256) tile_sort_regs_segment<4>(kt, keys, flatten_ids, begin, n, lane, s_out, isect_ids, cam_tile);
    else if (n <= 512) tile_sort_regs_segment<8>(kt, keys, flatten_ids, begin, n, lane, s_out, isect_ids, cam_tile);
    else tile_sort_regs_segment<16>(kt, keys, flatten_ids, begin, n, lane, s_out, isect_ids, cam_tile);
}

static bool wave_sort_merge_forced() { const char* e = test_switch("GSX_WAVE_SORT"); return e != nullptr && strcmp(e, "merge") == 0; }   // read per launch

template <class KT>
__global__ __launch_bounds__(ISECT_BLOCK) void tile_sort_kernel(uint32_t n_tiles, uint32_t tile_n_bits, KT kt,
                                                                const int32_t* __restrict__ tile_offsets, typename KT::T* __restrict__ keys,
                                                                int32_t* __restrict__ flatten_ids, int64_t* __restrict__ isect_ids,
                                                                int64_t capacity) {
    using K = typename KT::T;
    __shared__ K s_keys[TSORT_CAP + TSORT_CAP / 32];
    const uint32_t seg = blockIdx.x;
    const int64_t begin = tile_offsets[seg];
    const int n = (int)(tile_offsets[seg + 1] - tile_offsets[seg]);
    if (n <= TSORT_WAVE_CAP || begin + n > capacity) return;  // small segments: tile_sort_wave_kernel
    if (n > TSORT_CAP) return;                                // heavy / giant segments: tile_sort_big_kernel + giant_*, or tile_sort_bitmap_kernel
    const int64_t cam_tile = (((int64_t)(seg / n_tiles) << tile_n_bits) | (int64_t)(seg % n_tiles)) << 32;
    const int t = threadIdx.x;
    const int m = n <= 2048 ? 2048 : 4096;
    for (int i = t; i < m; i += ISECT_BLOCK) s_keys[spad_of<K>(i)] = i < n ? keys[begin + i] : ~(K)0;
    __syncthreads();
    if (m == 2048) merge_sort_lds<8, ISECT_BLOCK, K>(s_keys, t);
    else merge_sort_lds<16, ISECT_BLOCK, K>(s_keys, t);
    for (int i = t; i < n; i += ISECT_BLOCK) {
        const K k = s_keys[spad_of<K>(i)];
        if (KT::kDeferred) {
            keys[begin + i] = k;
        } else {
            const int32_t id = kt.id(k);
            flatten_ids[begin + i] = id;
            if (isect_ids) isect_ids[begin + i] = cam_tile | kt.depth_bits(k, id);
        }
    }
}

// Heavy tiles (4096 < keys <= 16384): one 1024-thread block sorts the whole segment in 132 KB of LDS.  (Their first implementation —
// 4096-key LDS chunks + rank merges through global memory by ONE 256-thread block — was latency bound: dependent global loads in
// every binary-search step made a 6 000-key tile cost over a millisecond, and dense scenes have hundreds of such tiles per frame:
// garden-like stand-in, 185 cameras: 2.35 ms per frame = 47 % of the GPU time of a training iteration went there.)
// Persistent grid: 256 blocks walk the segments with stride 256 (heavy tiles are neighbours in tile order: the stride spreads them
// over the CUs); a frame without heavy tiles costs 32 offset reads per block.
__global__ __launch_bounds__(1024) void tile_sort_big_kernel(uint32_t n_segments, uint32_t n_tiles, uint32_t tile_n_bits, uint32_t idx_bits,
                                                             const int32_t* __restrict__ tile_offsets, const uint64_t* __restrict__ keys,
                                                             int32_t* __restrict__ flatten_ids, int64_t* __restrict__ isect_ids,
                                                             int64_t capacity) {
    extern __shared__ uint64_t s_big[];
    const int t = threadIdx.x;
    const uint64_t idx_mask = (1ull << idx_bits) - 1ull;
    for (uint32_t seg = blockIdx.x; seg < n_segments; seg += gridDim.x) {
        const int64_t begin = tile_offsets[seg];
        const int n = (int)(tile_offsets[seg + 1] - tile_offsets[seg]);
        if (n <= TSORT_CAP || n > TSORT_BIG_CAP || begin + n > capacity) continue;   // (block-uniform)
        const int m = n <= 8192 ? 8192 : 16384;
        __syncthreads();   // the previous segment's keys have been read out
        for (int i = t; i < m; i += 1024) s_big[spad(i)] = i < n ? keys[begin + i] : ~0ull;
        __syncthreads();
        if (m == 8192) merge_sort_lds<8, 1024>(s_big, t);
        else merge_sort_lds<16, 1024>(s_big, t);
        const int64_t cam_tile = (((int64_t)(seg / n_tiles) << tile_n_bits) | (int64_t)(seg % n_tiles)) << 32;
        for (int i = t; i < n; i += 1024) {
            const uint64_t k = s_big[spad(i)];
            flatten_ids[begin + i] = (int32_t)(k & idx_mask);
            if (isect_ids) isect_ids[begin + i] = cam_tile | (int64_t)(k >> idx_bits);
        }
    }
}

// Giant segments (more than TSORT_BIG_CAP keys: the densest tiles of a garden-like scene hold 20 000 .. 60 000 keys, and nearly
// every frame of such a scene has a few).  They are sorted by many blocks: the segment is cut into chunks of TSORT_BIG_CAP keys,
// every chunk is sorted in LDS (giant_chunk_sort_kernel), then ceil(log2(chunks)) merge passes ping-pong between the two key
// buffers (giant_merge_kernel).  A merge pass is parallel over WINDOWS of TSORT_BIG_CAP output keys: the block finds the merge-path
// split of its window's two diagonals by a 64-way search (one wave per diagonal: three dependent global loads for a 262 144-key
// run instead of eighteen), stages the two input pieces in LDS, and merges them there exactly as merge_sort_lds does.  The last
// pass of a segment writes flatten_ids / isect_ids directly.  Keys are unique (they carry the Gaussian index), so the result is
// THE sorted order: bit-identical to the device-wide radix sort.
// giant_list_kernel enumerates the (segment, chunk) pairs once; chunk c of a segment is also window c of each of its passes.
GSX_DEV int ceil_log2_i(int v) { int p = 0; while ((1 << p) < v) ++p; return p; }

__global__ __launch_bounds__(1024) void giant_list_kernel(uint32_t n_segments, const int32_t* __restrict__ tile_offsets, int64_t capacity,
                                                          int4* __restrict__ list, uint32_t* __restrict__ count) {
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    for (uint32_t seg = threadIdx.x; seg < n_segments; seg += 1024u) {
        const int begin = tile_offsets[seg];
        const int n = tile_offsets[seg + 1] - begin;
        if (n <= TSORT_BIG_CAP || (int64_t)begin + n > capacity) continue;
        const int k = (n + TSORT_BIG_CAP - 1) / TSORT_BIG_CAP;
        const uint32_t base = atomicAdd(&s_n, (uint32_t)k);
        for (int c = 0; c < k; ++c) list[base + c] = make_int4(begin, n, c, (int)seg);
    }
    __syncthreads();
    if (threadIdx.x == 0) *count = s_n;
}

__global__ __launch_bounds__(1024) void giant_chunk_sort_kernel(const int4* __restrict__ list, const uint32_t* __restrict__ count,
                                                                uint64_t* __restrict__ keys) {
    extern __shared__ uint64_t s_big[];
    const int t = threadIdx.x;
    const uint32_t cnt = *count;
    for (uint32_t w = blockIdx.x; w < cnt; w += gridDim.x) {
        const int4 d = list[w];
        const int c0 = d.z * TSORT_BIG_CAP, cn = min(TSORT_BIG_CAP, d.y - c0);
        uint64_t* chunk = keys + (int64_t)d.x + c0;
        const int m = cn <= 8192 ? 8192 : 16384;
        __syncthreads();   // the previous chunk has been stored
        for (int i = t; i < m; i += 1024) s_big[spad(i)] = i < cn ? chunk[i] : ~0ull;
        __syncthreads();
        if (m == 8192) merge_sort_lds<8, 1024>(s_big, t);
        else merge_sort_lds<16, 1024>(s_big, t);
        for (int i = t; i < cn; i += 1024) chunk[i] = s_big[spad(i)];
    }
}

// merge-path split of diagonal d of two sorted runs in global memory, searched by a whole wave: number of outputs among the first d
// that come from A (A wins ties; there are none).  All 64 lanes must call it together.
GSX_DEV int merge_split_wave(const uint64_t* __restrict__ A, int la, const uint64_t* __restrict__ B, int lb, int d, int lane) {
    int lo = max(0, d - lb), hi = min(d, la);
    while (lo < hi) {
        const int64_t span = hi - lo;
        const int mid = lo + (int)(span * (lane + 1) / 65);   // non-decreasing in the lane, inside [lo, hi)
        const bool below = A[mid] <= B[d - 1 - mid];          // true for the first `cnt` lanes, false after
        const int cnt = __popcll(__ballot(below));
        const int last_true = lo + (int)(span * cnt / 65), first_false = lo + (int)(span * (cnt + 1) / 65);
        const int nlo = cnt > 0 ? last_true + 1 : lo, nhi = cnt < 64 ? first_false : hi;
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

__global__ __launch_bounds__(1024) void giant_merge_kernel(const int4* __restrict__ list, const uint32_t* __restrict__ count, int pass,
                                                           const uint64_t* __restrict__ src_buf, uint64_t* __restrict__ dst_buf, uint32_t n_tiles,
                                                           uint32_t tile_n_bits, uint32_t idx_bits, int32_t* __restrict__ flatten_ids,
                                                           int64_t* __restrict__ isect_ids) {
    extern __shared__ uint64_t s_big[];
    __shared__ int s_split[2];
    constexpr int E = TSORT_BIG_CAP / 1024;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t idx_mask = (1ull << idx_bits) - 1ull;
    const uint32_t cnt = *count;
    for (uint32_t w = blockIdx.x; w < cnt; w += gridDim.x) {
        const int4 d = list[w];
        const int n = d.y;
        const int passes = ceil_log2_i((n + TSORT_BIG_CAP - 1) / TSORT_BIG_CAP);
        if (pass >= passes) continue;   // (block-uniform) this segment is finished
        const bool last = pass == passes - 1;
        const uint32_t run = (uint32_t)TSORT_BIG_CAP << pass;             // pass < 17: run <= 2^30
        const uint32_t o = (uint32_t)d.z * TSORT_BIG_CAP;
        const uint32_t pair0 = o & ~(2u * run - 1u);
        const int la = (int)min(run, (uint32_t)n - pair0), lb = (int)min(run, (uint32_t)n - pair0 - (uint32_t)la);
        const int d0 = (int)(o - pair0), d1 = min(d0 + TSORT_BIG_CAP, la + lb);
        const uint64_t* A = src_buf + (int64_t)d.x + pair0;
        const uint64_t* B = A + la;
        if (wave < 2) {
            const int dd = wave == 0 ? d0 : d1;
            const int a = merge_split_wave(A, la, B, lb, dd, lane);
            if (lane == 0) s_split[wave] = a;
        }
        __syncthreads();   // (also: the previous window has been stored)
        const int a0 = s_split[0], a1 = s_split[1];
        const int na = a1 - a0, b0 = d0 - a0, W = d1 - d0, nb = W - na;
        for (int i = t; i < W; i += 1024) s_big[spad(i)] = i < na ? A[a0 + i] : B[b0 + (i - na)];
        __syncthreads();
        uint64_t r[E];
        {
            const int ot = t * E;   // this thread's first output inside the window
            int lo = max(0, min(ot, W) - nb), hi = min(min(ot, W), na);
            const int dt = min(ot, W);
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_big[spad(mid)] <= s_big[spad(na + dt - 1 - mid)]) lo = mid + 1; else hi = mid;
            }
            int ai = lo, bi = dt - lo;
            uint64_t a = ai < na ? s_big[spad(ai)] : ~0ull, b = bi < nb ? s_big[spad(na + bi)] : ~0ull;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool ta = a <= b;
                r[e] = ta ? a : b;
                const int nxt = ta ? ++ai : ++bi;
                const uint64_t v = nxt < (ta ? na : nb) ? s_big[spad((ta ? 0 : na) + nxt)] : ~0ull;
                a = ta ? v : a;
                b = ta ? b : v;
            }
        }
        __syncthreads();   // every thread has finished reading the staged runs
#pragma unroll
        for (int e = 0; e < E; ++e) s_big[spad(t * E + e)] = r[e];
        __syncthreads();
        const int64_t out0 = (int64_t)d.x + pair0 + d0;
        if (last) {
            const int64_t cam_tile = (((int64_t)((uint32_t)d.w / n_tiles) << tile_n_bits) | (int64_t)((uint32_t)d.w % n_tiles)) << 32;
            for (int i = t; i < W; i += 1024) {
                const uint64_t k = s_big[spad(i)];
                flatten_ids[out0 + i] = (int32_t)(k & idx_mask);
                if (isect_ids) isect_ids[out0 + i] = cam_tile | (int64_t)(k >> idx_bits);
            }
        } else {
            for (int i = t; i < W; i += 1024) dst_buf[out0 + i] = s_big[spad(i)];
        }
        __syncthreads();
    }
}

// ---- ranked variant: for frames whose tiles are heavy (a trained garden-like scene: 37 M intersections over 4 293 tiles, 60 % of
// them in tiles above 16 384 keys) the per-tile LDS merge sorts are LDS-bandwidth bound (~64 us per 16 384 64-bit keys per CU).
// Here the frame's Gaussians are first ranked ONCE by (depth bits, flatten index) — a stable 32-bit radix sort of C*N pairs, about
// 50 us at 1 M — and the per-tile keys are those ranks: unique integers below C*N.  A tile above 4096 keys is then sorted by
// writing its ranks into a C*N-bit bitmap in LDS (128 KB at 1 M) and reading the set bits back in order: no comparison, no merge
// passes, any segment size (tile_sort_bitmap_kernel, ~5 us + 3 ps per key per CU).  Lighter tiles take the merge sorts above with
// 4-byte keys.  flatten_ids = order[rank]; the result is the same total order, bit for bit.

GSX_DEV uint32_t rank_key_of(const int32_t* __restrict__ radii, const float* __restrict__ depths, uint32_t i) {
    const int2 r = reinterpret_cast<const int2*>(radii)[i];
    return (r.x > 0 && r.y > 0) ? __float_as_uint(depths[i]) : 0xFFFFFFFFu;   // (culled Gaussians: behind every visible one)
}

__global__ __launch_bounds__(ISECT_BLOCK) void rank_keys_kernel(uint32_t total, const int32_t* __restrict__ radii, const float* __restrict__ depths,
                                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * ISECT_BLOCK + threadIdx.x;
    if (i >= total) return;
    keys[i] = rank_key_of(radii, depths, i);
    vals[i] = i;
}

__global__ __launch_bounds__(ISECT_BLOCK) void rank_invert_kernel(uint32_t total, const uint32_t* __restrict__ order, uint32_t* __restrict__ ranks) {
    const uint32_t r = blockIdx.x * ISECT_BLOCK + threadIdx.x;
    if (r < total) ranks[order[r]] = r;
}

// ---- depth ranks without a library sort: a stable LSD radix sort of the C*N (depth bits, flatten index) pairs in four 8-bit digit
// passes, three launches per pass, sized for the ~1 M pairs of a frame (the whole problem lives in L2 / Infinity Cache: launch count and
// latency are the cost, not bytes).  A block owns RS_TILE consecutive pairs, a wave 64-pair rounds of them in index order.
//   rs_hist:    LDS histogram of the block's digits -> hist[digit][block]   (pass 0 builds the keys from radii / depths on the fly)
//   rs_scan:    one wave per digit: exclusive scan of its row over the blocks (DPP wave scan), row total -> total[digit]
//   rs_scatter: every block scans total[] itself, ranks its pairs stably inside each wave with ballots ("match any" over the digit's
//               bits: the lanes below me with my digit), sums the waves' counts, and stores each pair at
//               digit base + blocks before + waves before + rank in the wave.  The last pass writes order[] and ranks[] directly.
// Measured at 1 M pairs (tools/rank_sort_bench.py): 100 us (hist 7 + scan 5 + scatter 13 per pass) against 155 us for rocPRIM's pair
// sort + key / inversion passes; 11-bit digits (three passes): 139 us — 2048 output streams per block cost more than the pass they save;
// 4096-pair tiles: 109 us; counting the next pass's block histogram with global atomics inside the scatter: 2.2 ms.
#ifndef GSX_RS_BITS
#define GSX_RS_BITS 8
#endif
#ifndef GSX_RS_ROUNDS
#define GSX_RS_ROUNDS 8
#endif
constexpr uint32_t RS_BLOCK = 256, RS_WAVES = RS_BLOCK / 64, RS_ROUNDS = GSX_RS_ROUNDS, RS_TILE = RS_BLOCK * RS_ROUNDS;
constexpr uint32_t RS_BITS = GSX_RS_BITS, RS_BINS = 1u << RS_BITS, RS_PASSES = (32u + RS_BITS - 1u) / RS_BITS;
// a digit's row of the block histogram table: an odd number of 256 B units, so that one block's column (stride = row) spreads over the
// memory channels instead of camping on a few
__host__ __device__ inline uint32_t rs_row(uint32_t nblk) { return (((nblk + 63u) / 64u) | 1u) * 64u; }

template <bool FIRST>
GSX_DEV uint32_t rs_load_key(const uint32_t* __restrict__ keys, const int32_t* __restrict__ radii, const float* __restrict__ depths, uint32_t i) {
    return FIRST ? rank_key_of(radii, depths, i) : keys[i];
}

template <bool FIRST>
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(uint32_t total, uint32_t nblk, uint32_t shift, const uint32_t* __restrict__ keys,
                                                           const int32_t* __restrict__ radii, const float* __restrict__ depths,
                                                           uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_hist[RS_BINS];
    for (uint32_t d = threadIdx.x; d < RS_BINS; d += RS_BLOCK) s_hist[d] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE + threadIdx.x;
#pragma unroll
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        const uint32_t i = base + r * RS_BLOCK;
        if (i < total) atomicAdd(&s_hist[(rs_load_key<FIRST>(keys, radii, depths, i) >> shift) & (RS_BINS - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < RS_BINS; d += RS_BLOCK) hist[(size_t)d * rs_row(nblk) + blockIdx.x] = s_hist[d];
}

// row `digit` of hist (nblk <= 64 * RS_SCAN_PER_LANE entries) -> exclusive prefix in place, row total -> total[digit]
constexpr uint32_t RS_SCAN_PER_LANE = 8;
__global__ __launch_bounds__(RS_BLOCK) void rs_scan_kernel(uint32_t nblk, uint32_t* __restrict__ hist, uint32_t* __restrict__ total) {
    const uint32_t lane = threadIdx.x & 63u, digit = blockIdx.x * RS_WAVES + (threadIdx.x >> 6);
    uint32_t* row = hist + (size_t)digit * rs_row(nblk);
    uint32_t v[RS_SCAN_PER_LANE], sum = 0u;
#pragma unroll
    for (uint32_t k = 0; k < RS_SCAN_PER_LANE; ++k) {
        const uint32_t b = lane * RS_SCAN_PER_LANE + k;
        v[k] = b < nblk ? row[b] : 0u;
        sum += v[k];
    }
    const uint32_t incl = wave_incl_scan_u32(sum);
    uint32_t run = incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < RS_SCAN_PER_LANE; ++k) {
        const uint32_t b = lane * RS_SCAN_PER_LANE + k;
        if (b < nblk) row[b] = run;
        run += v[k];
    }
    if (lane == 63u) total[digit] = incl;
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(RS_BLOCK) void rs_scatter_kernel(uint32_t total, uint32_t nblk, uint32_t shift, const uint32_t* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in, const int32_t* __restrict__ radii,
                                                              const float* __restrict__ depths, const uint32_t* __restrict__ hist,
                                                              const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, uint32_t* __restrict__ ranks) {
    __shared__ uint32_t s_cnt[RS_WAVES][RS_BINS];   // per wave: digit counts, then the counts of the waves before it
    __shared__ uint32_t s_base[RS_BINS];            // first slot of this block's pairs of a digit
    __shared__ uint32_t s_wsum[RS_WAVES];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t d = threadIdx.x; d < RS_WAVES * RS_BINS; d += RS_BLOCK) (&s_cnt[0][0])[d] = 0u;
    // a wave's pairs are consecutive: rounds of 64 in index order (stability = block, wave, round, lane order)
    const uint32_t first = blockIdx.x * RS_TILE + wave * (64u * RS_ROUNDS) + lane;
    uint32_t key[RS_ROUNDS], val[RS_ROUNDS], lrank[RS_ROUNDS];
#pragma unroll
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        const uint32_t i = first + r * 64u;
        key[r] = i < total ? rs_load_key<FIRST>(keys_in, radii, depths, i) : 0u;
        val[r] = FIRST ? i : (i < total ? vals_in[i] : 0u);
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        const bool valid = first + r * 64u < total;
        const uint32_t d = (key[r] >> shift) & (RS_BINS - 1u);
        uint64_t peers = __builtin_amdgcn_ballot_w64(valid);   // the valid lanes of the round with my digit
#pragma unroll
        for (uint32_t bit = 0; bit < RS_BITS; ++bit) {
            const bool one = (d >> bit) & 1u;
            const uint64_t m = __builtin_amdgcn_ballot_w64(one);
            peers &= one ? m : ~m;
        }
        if (valid) {
            lrank[r] = s_cnt[wave][d] + (uint32_t)__builtin_popcountll(peers & below);
            if ((peers >> lane) == 1ull) s_cnt[wave][d] += (uint32_t)__builtin_popcountll(peers);   // the highest peer books the round
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the row is private to the wave: LDS operations of a wave stay in order
    }
    __syncthreads();
    {   // thread t: digits 8t .. 8t+7 — digit bases (scan of the digit totals), blocks before, waves before
        constexpr uint32_t PER = RS_BINS / RS_BLOCK;
        const uint32_t d0 = threadIdx.x * PER;
        uint32_t tot[PER], sum = 0u;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) { tot[k] = digit_total[d0 + k]; sum += tot[k]; }
        const uint32_t incl = wave_incl_scan_u32(sum);
        if (lane == 63u) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < wave; ++w) run += s_wsum[w];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t d = d0 + k;
            s_base[d] = run + hist[(size_t)d * rs_row(nblk) + blockIdx.x];
            run += tot[k];
            uint32_t before = 0u;
#pragma unroll
            for (uint32_t w = 0; w < RS_WAVES; ++w) { const uint32_t c = s_cnt[w][d]; s_cnt[w][d] = before; before += c; }
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        if (first + r * 64u >= total) continue;
        const uint32_t d = (key[r] >> shift) & (RS_BINS - 1u);
        const uint32_t dst = s_base[d] + s_cnt[wave][d] + lrank[r];
        if (LAST) { vals_out[dst] = val[r]; ranks[val[r]] = dst; }
        else { keys_out[dst] = key[r]; vals_out[dst] = val[r]; }
    }
}


// Persistent grid, one 1024-thread block per CU; dynamic LDS: n_words (a multiple of 4096) 32-bit words of bitmap, then 16 staging
// rows of BITMAP_STAGE ranks (one per wave).  The sorted ranks overwrite the segment's keys in place (every key has been read by
// then); ranked_finalize_kernel turns them into flatten_ids afterwards, at full-chip parallelism — a gather inside this kernel's
// emission loop put one global-load latency into each of a wave's 32 dependent steps.
constexpr uint32_t BITMAP_STAGE = 480;
__global__ __launch_bounds__(1024) void tile_sort_bitmap_kernel(uint32_t n_segments, uint32_t n_words, const int32_t* __restrict__ tile_offsets,
                                                                uint32_t* __restrict__ keys, int64_t capacity) {
    extern __shared__ uint32_t s_bits[];
    __shared__ uint32_t s_wave_total[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t per_wave = n_words / 16u;   // (a multiple of 256)
    uint32_t* stage = s_bits + n_words + wave * BITMAP_STAGE;
    for (uint32_t seg = blockIdx.x; seg < n_segments; seg += gridDim.x) {
        const int64_t begin = tile_offsets[seg];
        const int n = (int)(tile_offsets[seg + 1] - tile_offsets[seg]);
        if (n <= TSORT_CAP || begin + n > capacity) continue;   // (block-uniform)
        uint32_t* seg_keys = keys + begin;
        __syncthreads();   // the previous segment has been read out
        for (uint32_t i = t; i < n_words / 4u; i += 1024u) reinterpret_cast<uint4*>(s_bits)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (int i0 = (int)t; i0 < n; i0 += 8 * 1024) {   // eight loads in flight per thread
            uint32_t r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = i0 + u * 1024 < n ? seg_keys[i0 + u * 1024] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r[u] != 0xFFFFFFFFu) atomicOr(&s_bits[r[u] >> 5], 1u << (r[u] & 31u));
        }
        __syncthreads();
        // every wave owns a contiguous range of words: its population count, then the block-wide exclusive prefix over the waves
        uint32_t mine = 0u;
        for (uint32_t w = wave * per_wave + lane; w < (wave + 1u) * per_wave; w += 64u) mine += (uint32_t)__popc(s_bits[w]);
        mine = wave_incl_scan_u32(mine);
        if (lane == 63u) s_wave_total[wave] = mine;
        __syncthreads();
        uint32_t pos = 0u;
        for (uint32_t w = 0; w < wave; ++w) pos += s_wave_total[w];
        // emission, 256 words (4 per lane, lane-major within each 64-word group) per step
        for (uint32_t w0 = wave * per_wave; w0 < (wave + 1u) * per_wave; w0 += 256u) {
            uint32_t word[4], cnt[4], incl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                word[u] = s_bits[w0 + 64u * u + lane];
                cnt[u] = (uint32_t)__popc(word[u]);
            }
            const uint32_t s01 = wave_incl_scan_u32(cnt[0] | (cnt[1] << 16)), s23 = wave_incl_scan_u32(cnt[2] | (cnt[3] << 16));   // (<= 2048 each)
            incl[0] = s01 & 0xFFFFu; incl[1] = s01 >> 16; incl[2] = s23 & 0xFFFFu; incl[3] = s23 >> 16;
            const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)s01, 63), t23 = (uint32_t)__builtin_amdgcn_readlane((int)s23, 63);
            const uint32_t base[4] = {0u, t01 & 0xFFFFu, (t01 & 0xFFFFu) + (t01 >> 16), (t01 & 0xFFFFu) + (t01 >> 16) + (t23 & 0xFFFFu)};
            const uint32_t tot = base[3] + (t23 >> 16);
            if (tot <= BITMAP_STAGE) {   // (wave-uniform) ranks to the wave's staging row in order, then 64 consecutive outputs per store
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint32_t j = base[u] + incl[u] - cnt[u], bits = word[u];
                    const uint32_t rank0 = (w0 + 64u * u + lane) << 5;
                    while (bits != 0u) {
                        stage[j++] = rank0 + (uint32_t)__builtin_ctz(bits);
                        bits &= bits - 1u;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (uint32_t i = lane; i < tot; i += 64u) seg_keys[pos + i] = stage[i];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            } else {   // more than 480 of the step's 8192 bits set (a tile holding > 6 % of the frame's Gaussians): every lane writes its own runs
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint32_t out = pos + base[u] + incl[u] - cnt[u], bits = word[u];
                    const uint32_t rank0 = (w0 + 64u * u + lane) << 5;
                    while (bits != 0u) {
                        seg_keys[out++] = rank0 + (uint32_t)__builtin_ctz(bits);
                        bits &= bits - 1u;
                    }
                }
            }
            pos += tot;
        }
    }
}

// sorted ranks -> flatten_ids (and isect_ids on request: the segment of a position by binary search over the offsets)
__global__ __launch_bounds__(ISECT_BLOCK) void ranked_finalize_kernel(int64_t capacity, uint32_t n_segments, uint32_t n_tiles, uint32_t tile_n_bits,
                                                                      const int32_t* __restrict__ tile_offsets, const uint32_t* __restrict__ sorted,
                                                                      KeyRank kt, int32_t* __restrict__ flatten_ids, int64_t* __restrict__ isect_ids) {
    const int64_t p = (int64_t)blockIdx.x * ISECT_BLOCK + threadIdx.x;
    if (p >= capacity || p >= (int64_t)tile_offsets[n_segments]) return;
    const uint32_t r = sorted[p];
    const int32_t id = kt.id(r);
    flatten_ids[p] = id;
    if (isect_ids) {
        uint32_t lo = 0, hi = n_segments;   // last segment whose offset is <= p
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((int64_t)tile_offsets[mid] <= p) lo = mid; else hi = mid;
        }
        const int64_t cam_tile = (((int64_t)(lo / n_tiles) << tile_n_bits) | (int64_t)(lo % n_tiles)) << 32;
        isect_ids[p] = cam_tile | kt.depth_bits(r, id);
    }
}

// exclusive scan of n counters by one 1024-thread block (n = C*tiles + 1: a few thousand entries; the generic device scan
// costs three launches for them).  out[i] = sum(in[0..i)), in[n-1] is ignored and out[n-1] = grand total.
__global__ __launch_bounds__(1024) void bin_scan_kernel(uint32_t n, const uint32_t* __restrict__ in, int32_t* __restrict__ out,
                                                        uint32_t* __restrict__ max_count, int64_t capacity, int64_t seg_bound,
                                                        int32_t* __restrict__ lists_status, unsigned long long* __restrict__ host_word) {
    // exclusive scan of n - 1 counts, out[n - 1] = total.  Runs in 64 bits: a frame with more than 2^31 - 1 intersections does not
    // wrap silently — every offset saturates at INT32_MAX and the total is written as -1, which both consumers reject (the fill's
    // n_isects guard in the C ABI, the shim's TORCH_CHECK): such a scene needs the device-wide sort's int64 scan.
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    __shared__ uint32_t s_max;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_carry = 0ull; s_max = 0u; }
    uint32_t my_max = 0u;
    __syncthreads();
    // eight counters per thread and iteration: the 8 161 offsets of a 1080p frame are ONE pass of load -> scan -> store (with four, the second
    // pass's loads waited behind the first one's barriers: 10.7 us for a kernel that moves 64 KB)
    constexpr uint32_t PER = 8u;
    const bool aligned16 = (reinterpret_cast<uintptr_t>(in) & 15u) == 0u;   // (a caller's workspace pointer decides)
    for (uint32_t base = 0; base < n; base += 1024u * PER) {
        const uint32_t i0 = base + threadIdx.x * PER;
        unsigned long long v[PER];
        if (i0 + PER < n && aligned16) {   // whole and inside the counts
            const uint4 a = *reinterpret_cast<const uint4*>(in + i0), b = *reinterpret_cast<const uint4*>(in + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < PER; ++k) v[k] = (i0 + k < n - 1u) ? (unsigned long long)in[i0 + k] : 0ull;
        }
        unsigned long long mine = 0ull;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) { mine += v[k]; my_max = max(my_max, (uint32_t)v[k]); }
        unsigned long long incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(incl, o);
            if ((int)lane >= o) incl += t;
        }
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        unsigned long long wave_base = s_carry;
        for (uint32_t w = 0; w < wave; ++w) wave_base += s_wave[w];
        unsigned long long run = wave_base + incl - mine;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            if (i0 + k < n) out[i0 + k] = (i0 + k == n - 1u && run > 0x7FFFFFFFull) ? -1 : (int32_t)(run > 0x7FFFFFFFull ? 0x7FFFFFFFull : run);
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023u) s_carry = run;
        __syncthreads();
    }
    if (max_count != nullptr) {   // the largest segment: callers route frames with giant segments to the device-wide sort
        atomicMax(&s_max, my_max);
        __syncthreads();
        if (threadIdx.x == 0) *max_count = s_max;
    }
    // Guarded protocol (gsx_intersect_bin_count_guarded): the consumers of the lists learn ON THE DEVICE whether the fill that was launched
    // with `capacity` slots and merge passes for segments up to `seg_bound` keys produced complete lists: *lists_status = the total, or -1.
    if (lists_status != nullptr && threadIdx.x == 0) {   // (s_carry and s_max are final: written before the barriers above)
        const unsigned long long total = s_carry;
        const bool ok = total <= (unsigned long long)capacity && (seg_bound <= 0 || (int64_t)s_max <= seg_bound);
        *lists_status = ok ? (int32_t)total : -1;
    }
    // The host's copy of (n_isects | largest segment << 32): ONE 8-byte store into the caller's pinned memory through its device alias,
    // instead of two 4-byte copy launches behind this kernel (each ~5 us of stream time between the count and the key scatter).
    if (host_word != nullptr && threadIdx.x == 0) {
        const unsigned long long total = s_carry;
        const unsigned long long lo = total > 0x7FFFFFFFull ? 0xFFFFFFFFull : total;
        __hip_atomic_store(host_word, lo | ((unsigned long long)(max_count != nullptr ? s_max : 0u) << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct I32ToI64 {
    __host__ __device__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t scan_temp_bytes(uint32_t total) {
    size_t bytes = 0;
    auto in = rocprim::make_transform_iterator((const int32_t*)nullptr, I32ToI64());
    (void)rocprim::inclusive_scan(nullptr, bytes, in, (int64_t*)nullptr, (size_t)total, rocprim::plus<int64_t>(), 0, false);
    return bytes;
}
static size_t sort_temp_bytes(int64_t n, unsigned end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr,
                                    (int32_t*)nullptr, (size_t)n, 0u, end_bit, 0, false);
    return bytes;
}

}  // namespace gsx

using namespace gsx;

extern "C" size_t gsx_intersect_count_workspace_bytes(uint32_t C, uint32_t N) {
    return align_up(scan_temp_bytes(C * N), 256) + 256;
}

extern "C" int gsx_intersect_tile_count(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii,
                                        uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
                                        int32_t* tiles_per_gauss, int64_t* cum_tiles_per_gauss, int64_t* n_isects_dev,
                                        int64_t* n_isects_host_pinned, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint64_t total64 = (uint64_t)C * N;
    if (total64 > 0x7FFFFFFFull) { set_error("intersect_tile: C*N must fit int32 (flatten ids are int32)"); return GSX_ERR_INVALID_ARGUMENT; }
    const uint32_t total = (uint32_t)total64;
    if (bit_width_u32(tile_width * tile_height) + bit_width_u32(C) > 32) {  // Intersect.cpp:50
        set_error("intersect_tile: tile_n_bits + cam_n_bits must be <= 32");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    if (total == 0) {
        if (n_isects_dev) (void)hipMemsetAsync(n_isects_dev, 0, 8, st);
        if (n_isects_host_pinned) *n_isects_host_pinned = 0;
        return GSX_OK;
    }
    if (!means2d || !radii || !tiles_per_gauss || !cum_tiles_per_gauss || tile_size == 0) {
        set_error("intersect_tile_count: null pointer / zero tile size");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    size_t temp = scan_temp_bytes(total);
    if (workspace_bytes < temp || (temp && !workspace)) { set_error("intersect_tile_count: workspace too small"); return GSX_ERR_WORKSPACE_TOO_SMALL; }
    hipLaunchKernelGGL(isect_count_kernel, dim3((total + ISECT_BLOCK - 1) / ISECT_BLOCK), dim3(ISECT_BLOCK), 0, st, total,
                       means2d, radii, (float)tile_size, tile_width, tile_height, tiles_per_gauss);
    auto in = rocprim::make_transform_iterator((const int32_t*)tiles_per_gauss, I32ToI64());
    if (rocprim::inclusive_scan(workspace, temp, in, cum_tiles_per_gauss, (size_t)total, rocprim::plus<int64_t>(), st, false) !=
        hipSuccess) {
        set_error("intersect_tile_count: scan failed");
        return GSX_ERR_LAUNCH_FAILED;
    }
    if (n_isects_dev) (void)hipMemcpyAsync(n_isects_dev, cum_tiles_per_gauss + (total - 1), 8, hipMemcpyDeviceToDevice, st);
    if (n_isects_host_pinned) (void)hipMemcpyAsync(n_isects_host_pinned, cum_tiles_per_gauss + (total - 1), 8, hipMemcpyDeviceToHost, st);
    return check_launch("intersect_tile_count");
}

extern "C" size_t gsx_intersect_fill_workspace_bytes(uint32_t C, uint32_t N, int64_t n_isects, int sort) {
    (void)N;
    if (!sort || n_isects <= 0) return 256;
    // unsorted keys + unsorted values + rocPRIM temporary storage
    return align_up((size_t)n_isects * 8, 256) + align_up((size_t)n_isects * 4, 256) + align_up(sort_temp_bytes(n_isects, 64), 256) + 256;
}

extern "C" int gsx_intersect_tile_fill(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii,
                                       const float* depths, const int64_t* cum_tiles_per_gauss, uint32_t tile_size,
                                       uint32_t tile_width, uint32_t tile_height, int sort, int64_t n_isects,
                                       int64_t* isect_ids, int32_t* flatten_ids, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    return gsx_intersect_tile_fill_packed(C, N, 0, nullptr, means2d, radii, depths, cum_tiles_per_gauss, tile_size, tile_width, tile_height, sort,
                                          n_isects, isect_ids, flatten_ids, workspace, workspace_bytes, stream);
}

// camera_ids != NULL: the packed layout of the reference (Intersect.cpp:31-38, IntersectTile.cu:85-88): the arrays hold nnz (camera, Gaussian)
// pairs, camera_ids[nnz] (int64) names each pair's camera, flatten_ids index the nnz pairs; N is ignored.
extern "C" int gsx_intersect_tile_fill_packed(uint32_t C, uint32_t N, uint32_t nnz, const int64_t* camera_ids, const float* means2d,
                                              const int32_t* radii, const float* depths, const int64_t* cum_tiles_per_gauss,
                                              uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int sort, int64_t n_isects,
                                              int64_t* isect_ids, int32_t* flatten_ids, void* workspace, size_t workspace_bytes,
                                              void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_isects <= 0) return GSX_OK;
    if (n_isects > 0x7FFFFFFFll) { set_error("intersect_tile: n_isects must fit int32 (tile offsets are int32)"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!means2d || !radii || !depths || !cum_tiles_per_gauss || !isect_ids || !flatten_ids || tile_size == 0) {
        set_error("intersect_tile_fill: null pointer / zero tile size");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    const uint32_t total = camera_ids ? nnz : C * N;
    if (camera_ids) N = 1;
    const uint32_t tile_n_bits = bit_width_u32(tile_width * tile_height), cam_n_bits = bit_width_u32(C);
    if (tile_n_bits + cam_n_bits > 32) { set_error("intersect_tile: tile_n_bits + cam_n_bits must be <= 32"); return GSX_ERR_INVALID_ARGUMENT; }
    int64_t* keys_out = isect_ids;
    int32_t* vals_out = flatten_ids;
    size_t temp = 0;
    char* ws = (char*)workspace;
    if (sort) {
        temp = sort_temp_bytes(n_isects, 32 + tile_n_bits + cam_n_bits);
        const size_t need = align_up((size_t)n_isects * 8, 256) + align_up((size_t)n_isects * 4, 256) + temp;
        if (!workspace || workspace_bytes < need) { set_error("intersect_tile_fill: workspace too small"); return GSX_ERR_WORKSPACE_TOO_SMALL; }
        keys_out = (int64_t*)ws;
        vals_out = (int32_t*)(ws + align_up((size_t)n_isects * 8, 256));
    }
    hipLaunchKernelGGL(isect_fill_kernel, dim3((total + ISECT_BLOCK - 1) / ISECT_BLOCK), dim3(ISECT_BLOCK), 0, st, total, N,
                       means2d, radii, depths, cum_tiles_per_gauss, (float)tile_size, tile_width, tile_height, tile_n_bits,
                       keys_out, vals_out, camera_ids);
    if (sort) {
        void* tmp = ws + align_up((size_t)n_isects * 8, 256) + align_up((size_t)n_isects * 4, 256);
        if (rocprim::radix_sort_pairs(tmp, temp, (const uint64_t*)keys_out, (uint64_t*)isect_ids, (const int32_t*)vals_out,
                                      flatten_ids, (size_t)n_isects, 0u, 32 + tile_n_bits + cam_n_bits, st, false) != hipSuccess) {
            set_error("intersect_tile_fill: radix sort failed");
            return GSX_ERR_LAUNCH_FAILED;
        }
    }
    return check_launch("intersect_tile_fill");
}

extern "C" int gsx_intersect_offset(int64_t n_isects, const int64_t* isect_ids, uint32_t C, uint32_t tile_width,
                                    uint32_t tile_height, int32_t* offsets, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint32_t n_tiles = tile_width * tile_height;
    if (C * n_tiles == 0) return GSX_OK;
    if (!offsets || (n_isects > 0 && !isect_ids)) { set_error("intersect_offset: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(isect_offset_kernel, dim3((C * n_tiles + ISECT_BLOCK - 1) / ISECT_BLOCK), dim3(ISECT_BLOCK), 0, st,
                       n_isects, isect_ids, C, n_tiles, bit_width_u32(n_tiles), offsets);
    return check_launch("intersect_offset");
}


// ---- binned path C ABI -----------------------------------------------------------------------------------------------------
// Slices per camera: 256 (one 1024-thread block per CU); GSX_BIN_NB overrides it for experiments (a multiple of 32 up to 1024; read once).
static uint32_t bin_nb() {
    static const uint32_t nb = [] {
        const char* e = test_switch("GSX_BIN_NB");
        const long v = e ? atol(e) : 256;
        return (v >= 32 && v <= (long)BIN_NB_MAX && v % 32 == 0) ? (uint32_t)v : 256u;
    }();
    return nb;
}
static size_t bin_hist_bytes(uint32_t C, uint32_t n_tiles) { return align_up((size_t)C * bin_nb() * n_tiles * 4, 256); }

extern "C" int gsx_intersect_bin_supported(uint32_t tile_width, uint32_t tile_height) {
    return (uint64_t)tile_width * tile_height <= BIN_MAX_TILES;
}

extern "C" size_t gsx_intersect_bin_count_workspace_bytes(uint32_t C, uint32_t tile_width, uint32_t tile_height) {
    const uint32_t nseg = C * tile_width * tile_height;
    return bin_hist_bytes(C, tile_width * tile_height) + align_up((size_t)(nseg + 1) * 4, 256) + 256;
}

extern "C" int gsx_intersect_bin_count(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, uint32_t tile_size,
                                       uint32_t tile_width, uint32_t tile_height, int32_t* tiles_per_gauss, int32_t* tile_offsets,
                                       int64_t* n_isects_host_pinned, void* workspace, size_t workspace_bytes, void* stream) {
    return gsx_intersect_bin_count_guarded(C, N, means2d, radii, tile_size, tile_width, tile_height, tiles_per_gauss, tile_offsets,
                                           n_isects_host_pinned, workspace, workspace_bytes, 0, 0, nullptr, stream);
}

extern "C" int gsx_intersect_bin_count_guarded(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, uint32_t tile_size,
                                               uint32_t tile_width, uint32_t tile_height, int32_t* tiles_per_gauss, int32_t* tile_offsets,
                                               int64_t* n_isects_host_pinned, void* workspace, size_t workspace_bytes, int64_t capacity,
                                               int64_t max_segment, int32_t* lists_status, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint64_t total64 = (uint64_t)C * N;
    if (total64 > 0x7FFFFFFFull) { set_error("intersect_bin_count: C*N must fit int32 (flatten ids are int32)"); return GSX_ERR_INVALID_ARGUMENT; }
    const uint32_t n_tiles = tile_width * tile_height, nseg = C * n_tiles;
    if (!gsx_intersect_bin_supported(tile_width, tile_height)) { set_error("intersect_bin_count: more than 36864 tiles per camera (use intersect_tile)"); return GSX_ERR_UNSUPPORTED; }
    if (bit_width_u32(n_tiles) + bit_width_u32(C) > 32) { set_error("intersect_bin_count: tile_n_bits + cam_n_bits must be <= 32"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!tile_offsets || tile_size == 0 || (total64 && (!means2d || !radii))) { set_error("intersect_bin_count: null pointer / zero tile size"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!workspace || workspace_bytes < gsx_intersect_bin_count_workspace_bytes(C, tile_width, tile_height)) {
        set_error("intersect_bin_count: workspace too small");
        return GSX_ERR_WORKSPACE_TOO_SMALL;
    }
    if (nseg == 0 || N == 0) {
        (void)hipMemsetAsync(tile_offsets, 0, (size_t)(nseg + 1) * 4, st);
        if (lists_status) (void)hipMemsetAsync(lists_status, 0, 4, st);   // complete (empty) lists
        if (n_isects_host_pinned) *n_isects_host_pinned = 0;
        return check_launch("intersect_bin_count(empty)");
    }
    uint32_t* hist = (uint32_t*)workspace;
    uint32_t* counts = (uint32_t*)((char*)workspace + bin_hist_bytes(C, n_tiles));
    const uint32_t per_block = (N + bin_nb() - 1) / bin_nb();
    const size_t lds = (size_t)n_tiles * 4;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)bin_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // round 6: prefix + scan in one launch (bin_prefix_scan_kernel; its per-workgroup aggregate words live where the tile totals of the two-launch form
    // did); GSX_BIN_SCAN=separate (test switch) keeps bin_prefix_kernel + bin_scan_kernel: second implementation in the tests, A/B
    const uint32_t n_pblocks = (nseg + 31) / 32;
    const bool separate_scan = [] { const char* e = test_switch("GSX_BIN_SCAN"); return e != nullptr && strcmp(e, "separate") == 0; }();   // read per launch: the tests switch it
    unsigned long long* agg = separate_scan ? nullptr : (unsigned long long*)counts;   // n_pblocks x 8 B <= (nseg + 1) x 4 B
    hipLaunchKernelGGL(bin_count_kernel, dim3(bin_nb(), C), dim3(BIN_BLOCK), lds, st, N, per_block, means2d, radii, (float)tile_size, tile_width,
                       tile_height, tiles_per_gauss, hist, agg, n_pblocks);
    if (separate_scan) hipLaunchKernelGGL(bin_prefix_kernel, dim3(n_pblocks), dim3(bin_nb()), 0, st, C, n_tiles, hist, counts);
    // offsets[t] = intersections before (camera, tile) t; offsets[nseg] = n_isects
    // (the largest segment lands in the slack word behind the counts; it travels to the host in the upper half of the pinned word)
    uint32_t* max_count = (uint32_t*)((char*)workspace + bin_hist_bytes(C, n_tiles) + align_up((size_t)(nseg + 1) * 4, 256));
    // the pinned host word (low 32 bits: n_isects, 0xFFFFFFFF = more than 2^31 - 1; high 32 bits: keys of the largest (camera, tile) segment)
    // is written by bin_scan itself when the memory has a device alias (hipHostMalloc / hipHostRegister: torch's pinned tensors), else copied
    unsigned long long* host_alias = nullptr;
    if (n_isects_host_pinned) {
        // all ones = "not written yet" (no frame produces it: the high half is a segment size <= 2^31): a host that would rather not put
        // an event into the stream polls the word — its HIGH half changes last on either path below
        *n_isects_host_pinned = -1;
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, n_isects_host_pinned, 0) == hipSuccess && dp != nullptr) host_alias = (unsigned long long*)dp;
        else (void)hipGetLastError();
    }
    if (separate_scan)
        hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, st, nseg + 1, (const uint32_t*)counts, tile_offsets, max_count, capacity, max_segment,
                           lists_status, host_alias);
    else
        hipLaunchKernelGGL(bin_prefix_scan_kernel, dim3(n_pblocks), dim3(bin_nb()), 0, st, C, n_tiles, hist, agg, tile_offsets, max_count, capacity, max_segment,
                           lists_status, host_alias);
    if (n_isects_host_pinned && host_alias == nullptr) {
        (void)hipMemcpyAsync(n_isects_host_pinned, tile_offsets + nseg, 4, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync((char*)n_isects_host_pinned + 4, max_count, 4, hipMemcpyDeviceToHost, st);
    }
    return check_launch("intersect_bin_count");
}

// (segment, chunk) list of the giant segments: sum of ceil(n_s / 16384) over segments with n_s > 16384 is below n_isects / 8192
static size_t giant_list_bytes(int64_t n_isects) { return align_up(((size_t)n_isects / (TSORT_BIG_CAP / 2) + 2) * sizeof(int4), 256); }

extern "C" size_t gsx_intersect_bin_fill_workspace_bytes(uint32_t C, uint32_t tile_width, uint32_t tile_height, int64_t n_isects) {
    (void)C; (void)tile_width; (void)tile_height;
    if (n_isects <= 0) return 256;
    return 2 * align_up((size_t)n_isects * 8, 256) + giant_list_bytes(n_isects) + 256;
}

// `count_workspace` is the workspace gsx_intersect_bin_count filled (its per-block prefixes are consumed here).
extern "C" int gsx_intersect_bin_fill(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
                                      uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, const int32_t* tile_offsets,
                                      int64_t n_isects, int64_t max_segment, const void* count_workspace, int32_t* flatten_ids,
                                      int64_t* isect_ids, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_isects <= 0) return GSX_OK;
    if (n_isects > 0x7FFFFFFFll) { set_error("intersect_bin_fill: n_isects must fit int32"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!means2d || !radii || !depths || !tile_offsets || !count_workspace || !flatten_ids || tile_size == 0) {
        set_error("intersect_bin_fill: null pointer / zero tile size");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    if (!gsx_intersect_bin_supported(tile_width, tile_height)) { set_error("intersect_bin_fill: more than 36864 tiles per camera"); return GSX_ERR_UNSUPPORTED; }
    const uint32_t total = C * N, n_tiles = tile_width * tile_height, nseg = C * n_tiles;
    if (!workspace || workspace_bytes < gsx_intersect_bin_fill_workspace_bytes(C, tile_width, tile_height, n_isects)) {
        set_error("intersect_bin_fill: workspace too small");
        return GSX_ERR_WORKSPACE_TOO_SMALL;
    }
    const uint32_t idx_bits = total > 1 ? bit_width_u32(total - 1) : 1;
    uint64_t* keys = (uint64_t*)workspace;
    uint64_t* keys_alt = (uint64_t*)((char*)workspace + align_up((size_t)n_isects * 8, 256));
    const uint32_t per_block = (N + bin_nb() - 1) / bin_nb();
    const size_t lds = (size_t)n_tiles * 4;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)bin_scatter_kernel<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bin_scatter_kernel<uint64_t>, dim3(bin_nb(), C), dim3(BIN_BLOCK), lds, st, N, per_block, means2d, radii, depths,
                       (const uint32_t*)nullptr, (float)tile_size, tile_width, tile_height, idx_bits, tile_offsets, (const uint32_t*)count_workspace,
                       keys, (uint32_t)n_isects);
    const KeyDepthIdx kt{idx_bits};
    // segments up to 1024 keys: one wave each — the register bitonic network (round 6: isect_ids too; GSX_WAVE_SORT=merge (test switch) forces
    // the LDS merge sort: second implementation in the tests)
    if (!wave_sort_merge_forced())
        hipLaunchKernelGGL(tile_sort_wave_regs_kernel<KeyDepthIdx>, dim3((nseg + 3) / 4), dim3(256), 0, st, nseg, kt, tile_offsets, keys, flatten_ids, n_isects,
                           isect_ids, n_tiles, bit_width_u32(n_tiles));
    else
        hipLaunchKernelGGL(tile_sort_wave_kernel<KeyDepthIdx>, dim3((nseg + 3) / 4), dim3(256), 0, st, nseg, n_tiles, bit_width_u32(n_tiles), kt,
                           tile_offsets, keys, flatten_ids, isect_ids, n_isects);
    // (`max_segment` > 0 is the caller's bound on the largest segment: the kernels for larger segments are not launched at all — a frame
    // that outgrows the bound is refilled / rendered again, as documented for the giant-segment passes)
    const int64_t seg_cap = std::min<int64_t>(max_segment > 0 ? max_segment : n_isects, n_isects);
    if (seg_cap > TSORT_WAVE_CAP)
        hipLaunchKernelGGL(tile_sort_kernel<KeyDepthIdx>, dim3(nseg), dim3(ISECT_BLOCK), 0, st, n_tiles, bit_width_u32(n_tiles), kt, tile_offsets,
                           keys, flatten_ids, isect_ids, n_isects);
    if (seg_cap > TSORT_CAP) {   // a segment above 4096 keys needs at least that many intersections
        const size_t big_lds = (size_t)(TSORT_BIG_CAP + TSORT_BIG_CAP / 32) * 8;
        static const bool attr_set = [&] {
            return hipFuncSetAttribute((const void*)tile_sort_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds) == hipSuccess;
        }();
        (void)attr_set;
        hipLaunchKernelGGL(tile_sort_big_kernel, dim3(256), dim3(1024), big_lds, st, nseg, n_tiles, bit_width_u32(n_tiles), idx_bits, tile_offsets,
                           (const uint64_t*)keys, flatten_ids, isect_ids, n_isects);
    }
    // giant segments (above 16384 keys): chunk sorts + merge passes by many blocks; the number of passes follows from the caller's
    // bound on the largest segment (unknown: every intersection could sit in one tile)
    const int64_t seg_bound = std::min<int64_t>(max_segment > 0 ? max_segment : n_isects, n_isects);
    if (seg_bound > TSORT_BIG_CAP) {
        const size_t big_lds = (size_t)(TSORT_BIG_CAP + TSORT_BIG_CAP / 32) * 8;
        static const bool attr_set = [&] {
            return hipFuncSetAttribute((const void*)giant_chunk_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds) == hipSuccess &&
                   hipFuncSetAttribute((const void*)giant_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds) == hipSuccess;
        }();
        (void)attr_set;
        int4* list = (int4*)((char*)workspace + 2 * align_up((size_t)n_isects * 8, 256));
        uint32_t* list_count = (uint32_t*)((char*)list + giant_list_bytes(n_isects));
        const int64_t chunks = (seg_bound + TSORT_BIG_CAP - 1) / TSORT_BIG_CAP;
        int passes = 0;
        while ((1ll << passes) < chunks) ++passes;
        const uint32_t grid = 256;   // one block per CU (132 KB of LDS each); the blocks walk the list with that stride
        hipLaunchKernelGGL(giant_list_kernel, dim3(1), dim3(1024), 0, st, nseg, tile_offsets, n_isects, list, list_count);
        hipLaunchKernelGGL(giant_chunk_sort_kernel, dim3(grid), dim3(1024), big_lds, st, (const int4*)list, (const uint32_t*)list_count, keys);
        for (int p = 0; p < passes; ++p)
            hipLaunchKernelGGL(giant_merge_kernel, dim3(grid), dim3(1024), big_lds, st, (const int4*)list, (const uint32_t*)list_count, p,
                               (const uint64_t*)((p & 1) ? keys_alt : keys), (p & 1) ? keys : keys_alt, n_tiles, bit_width_u32(n_tiles), idx_bits,
                               flatten_ids, isect_ids);
    }
    return check_launch("intersect_bin_fill");
}

// ---- ranked variant (see tile_sort_bitmap_kernel): C ABI ------------------------------------------------------------------
static constexpr uint32_t RANK_MAX_WORDS = 32768;   // 128 KB of LDS bitmap (+ 30 KB of staging rows): 1 048 576 Gaussians, all cameras together
static uint32_t rank_words(uint32_t total) { return (uint32_t)align_up((size_t)(total + 31u) / 32u, 4096); }   // 16 waves x steps of 256 words

extern "C" int gsx_intersect_ranked_supported(uint32_t C, uint32_t N) {
    return (uint64_t)C * N > 0 && (uint64_t)C * N <= (uint64_t)RANK_MAX_WORDS * 32u ? 1 : 0;
}

// merge_sort_limit = 0: rocPRIM's default sorts up to 1 M items with a block sort + 10 merge passes (21 launches, 150 us at 1 M);
// the Onesweep radix path is one histogram + four digit passes
using RankSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

static size_t rank_sort_temp_bytes(uint32_t total) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<RankSortConfig>(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (size_t)total, 0u, 32u, 0, false);
    return align_up(bytes, 256);
}

static uint32_t rs_blocks(uint32_t total) { return (total + RS_TILE - 1u) / RS_TILE; }
static size_t rs_hist_bytes(uint32_t total) { return align_up((size_t)RS_BINS * rs_row(rs_blocks(total)) * 4, 256); }
static size_t rs_table_bytes(uint32_t total) { return rs_hist_bytes(total) + align_up((size_t)RS_BINS * 4, 256); }

extern "C" size_t gsx_intersect_depth_ranks_workspace_bytes(uint32_t C, uint32_t N) {
    const uint32_t total = C * N;
    const size_t lib = rank_sort_temp_bytes(total), own = align_up((size_t)total * 4, 256) + rs_table_bytes(total);
    return 3 * align_up((size_t)total * 4, 256) + (lib > own ? lib : own) + 256;
}

extern "C" int gsx_intersect_depth_ranks(uint32_t C, uint32_t N, const int32_t* radii, const float* depths, uint32_t* ranks, uint32_t* order,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!gsx_intersect_ranked_supported(C, N)) { set_error("intersect_depth_ranks: C*N must be in [1, 1048576]"); return GSX_ERR_UNSUPPORTED; }
    if (!radii || !depths || !ranks || !order) { set_error("intersect_depth_ranks: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!workspace || workspace_bytes < gsx_intersect_depth_ranks_workspace_bytes(C, N)) {
        set_error("intersect_depth_ranks: workspace too small");
        return GSX_ERR_WORKSPACE_TOO_SMALL;
    }
    const uint32_t total = C * N;
    const size_t stride = align_up((size_t)total * 4, 256);
    uint32_t* keys_a = (uint32_t*)workspace;
    uint32_t* keys_b = (uint32_t*)((char*)workspace + stride);
    uint32_t* vals_a = (uint32_t*)((char*)workspace + 2 * stride);
    void* rest = (char*)workspace + 3 * stride;
    const char* sw = test_switch("GSX_RANK_SORT");
    if (sw && !strcmp(sw, "rocprim")) {
        // second implementation (tests, A/B): the library's stable pair sort + an inversion pass
        size_t temp_bytes = rank_sort_temp_bytes(total);
        const uint32_t grid = (total + ISECT_BLOCK - 1) / ISECT_BLOCK;
        hipLaunchKernelGGL(rank_keys_kernel, dim3(grid), dim3(ISECT_BLOCK), 0, st, total, radii, depths, keys_a, vals_a);
        if (rocprim::radix_sort_pairs<RankSortConfig>(rest, temp_bytes, (const uint32_t*)keys_a, keys_b, (const uint32_t*)vals_a, order, (size_t)total, 0u, 32u,
                                                      st, false) != hipSuccess) {
            set_error("intersect_depth_ranks: radix sort failed");
            return GSX_ERR_LAUNCH_FAILED;
        }
        hipLaunchKernelGGL(rank_invert_kernel, dim3(grid), dim3(ISECT_BLOCK), 0, st, total, (const uint32_t*)order, ranks);
        return check_launch("intersect_depth_ranks");
    }
    // stable LSD radix sort over the 32 depth bits: equal depths keep ascending flatten index, as the 64-bit keys' low bits do
    uint32_t* vals_b = (uint32_t*)rest;
    uint32_t* hist = (uint32_t*)((char*)rest + stride);
    uint32_t* digit_total = (uint32_t*)((char*)hist + rs_hist_bytes(total));
    const uint32_t nblk = rs_blocks(total);
    const dim3 gb(nblk), gs(RS_BINS / RS_WAVES), blk(RS_BLOCK);
    const uint32_t *kin = nullptr, *vin = nullptr;
    uint32_t *kout = keys_a, *vout = vals_a;
    for (uint32_t pass = 0; pass < RS_PASSES; ++pass) {
        const uint32_t shift = pass * RS_BITS;
        const bool first = pass == 0, last = pass + 1 == RS_PASSES;
        if (first) hipLaunchKernelGGL(rs_hist_kernel<true>, gb, blk, 0, st, total, nblk, shift, kin, radii, depths, hist);
        else hipLaunchKernelGGL(rs_hist_kernel<false>, gb, blk, 0, st, total, nblk, shift, kin, radii, depths, hist);
        hipLaunchKernelGGL(rs_scan_kernel, gs, blk, 0, st, nblk, hist, digit_total);
#define GSX_RS_SCATTER(F, L, KO, VO, RK) hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_scatter_kernel<F, L>), gb, blk, 0, st, total, nblk, shift, kin, vin, radii, depths, \
                                                             (const uint32_t*)hist, (const uint32_t*)digit_total, KO, VO, RK)
        if (last) GSX_RS_SCATTER(false, true, (uint32_t*)nullptr, order, ranks);
        else if (first) GSX_RS_SCATTER(true, false, kout, vout, (uint32_t*)nullptr);
        else GSX_RS_SCATTER(false, false, kout, vout, (uint32_t*)nullptr);
#undef GSX_RS_SCATTER
        kin = kout; vin = vout;
        kout = kout == keys_a ? keys_b : keys_a;
        vout = vout == vals_a ? vals_b : vals_a;
    }
    static_assert(RS_PASSES >= 2 && RS_PASSES * RS_BITS >= 32 && RS_BINS % RS_BLOCK == 0 && RS_BINS % RS_WAVES == 0, "digit split");
    static_assert((uint64_t)64 * RS_SCAN_PER_LANE * RS_TILE >= (uint64_t)RANK_MAX_WORDS * 32u, "rs_scan_kernel: one wave scans a digit's row");
    return check_launch("intersect_depth_ranks");
}

extern "C" size_t gsx_intersect_bin_fill_ranked_workspace_bytes(int64_t n_isects) {
    return n_isects <= 0 ? 256 : align_up((size_t)n_isects * 4, 256) + 256;
}

extern "C" int gsx_intersect_bin_fill_ranked(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
                                             uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, const int32_t* tile_offsets,
                                             int64_t n_isects, const void* count_workspace, const uint32_t* ranks, const uint32_t* order,
                                             int32_t* flatten_ids, int64_t* isect_ids, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_isects <= 0) return GSX_OK;
    if (n_isects > 0x7FFFFFFFll) { set_error("intersect_bin_fill_ranked: n_isects must fit int32"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!means2d || !radii || !depths || !tile_offsets || !count_workspace || !ranks || !order || !flatten_ids || tile_size == 0) {
        set_error("intersect_bin_fill_ranked: null pointer / zero tile size");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    if (!gsx_intersect_bin_supported(tile_width, tile_height) || !gsx_intersect_ranked_supported(C, N)) {
        set_error("intersect_bin_fill_ranked: more than 36864 tiles per camera or more than 1048576 Gaussians");
        return GSX_ERR_UNSUPPORTED;
    }
    if (!workspace || workspace_bytes < gsx_intersect_bin_fill_ranked_workspace_bytes(n_isects)) {
        set_error("intersect_bin_fill_ranked: workspace too small");
        return GSX_ERR_WORKSPACE_TOO_SMALL;
    }
    const uint32_t total = C * N, n_tiles = tile_width * tile_height, nseg = C * n_tiles;
    uint32_t* keys = (uint32_t*)workspace;
    const uint32_t per_block = (N + bin_nb() - 1) / bin_nb();
    const size_t lds = (size_t)n_tiles * 4;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)bin_scatter_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bin_scatter_kernel<uint32_t>, dim3(bin_nb(), C), dim3(BIN_BLOCK), lds, st, N, per_block, means2d, radii, depths, ranks,
                       (float)tile_size, tile_width, tile_height, 0u, tile_offsets, (const uint32_t*)count_workspace, keys, (uint32_t)n_isects);
    const KeyRank kt{order, depths};
    if (!wave_sort_merge_forced())   // (deferred keys: the sorted ranks stay in place, ranked_finalize_kernel turns them into ids / isect_ids)
        hipLaunchKernelGGL(tile_sort_wave_regs_kernel<KeyRank>, dim3((nseg + 3) / 4), dim3(256), 0, st, nseg, kt, tile_offsets, keys, flatten_ids, n_isects, (int64_t*)nullptr, n_tiles, bit_width_u32(n_tiles));
    else
        hipLaunchKernelGGL(tile_sort_wave_kernel<KeyRank>, dim3((nseg + 3) / 4), dim3(256), 0, st, nseg, n_tiles, bit_width_u32(n_tiles), kt, tile_offsets,
                           keys, flatten_ids, isect_ids, n_isects);
    hipLaunchKernelGGL(tile_sort_kernel<KeyRank>, dim3(nseg), dim3(ISECT_BLOCK), 0, st, n_tiles, bit_width_u32(n_tiles), kt, tile_offsets,
                       keys, flatten_ids, isect_ids, n_isects);
    if (n_isects > TSORT_CAP) {
        const uint32_t n_words = rank_words(total);
        const size_t bitmap_lds = (size_t)n_words * 4 + (size_t)16 * BITMAP_STAGE * 4;
        if (bitmap_lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)tile_sort_bitmap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bitmap_lds);
        hipLaunchKernelGGL(tile_sort_bitmap_kernel, dim3(256), dim3(1024), bitmap_lds, st, nseg, n_words, tile_offsets, keys, n_isects);
    }
    hipLaunchKernelGGL(ranked_finalize_kernel, dim3((uint32_t)((n_isects + ISECT_BLOCK - 1) / ISECT_BLOCK)), dim3(ISECT_BLOCK), 0, st, n_isects, nseg,
                       n_tiles, bit_width_u32(n_tiles), tile_offsets, (const uint32_t*)keys, kt, flatten_ids, isect_ids);
    return check_launch("intersect_bin_fill_ranked");
}
