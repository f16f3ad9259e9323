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
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "gsx_device.hpp"

namespace gsx {

void set_error(const char* msg);
int check_launch(const char* what);

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
                                                                 int32_t* __restrict__ flatten_ids) {
    const uint32_t idx = blockIdx.x * ISECT_BLOCK + threadIdx.x;
    if (idx >= total) return;
    uint32_t x0, y0, x1, y1;
    if (!tile_rect(means2d, radii, idx, tile_size, tw, th, x0, y0, x1, y1)) return;
    const int64_t cid = idx / N;
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
    hipStream_t st = (hipStream_t)stream;
    if (n_isects <= 0) return GSX_OK;
    if (n_isects > 0x7FFFFFFFll) { set_error("intersect_tile: n_isects must fit int32 (tile offsets are int32)"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!means2d || !radii || !depths || !cum_tiles_per_gauss || !isect_ids || !flatten_ids || tile_size == 0) {
        set_error("intersect_tile_fill: null pointer / zero tile size");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    const uint32_t total = C * N;
    const uint32_t tile_n_bits = bit_width_u32(tile_width * tile_height), cam_n_bits = bit_width_u32(C);
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
                       keys_out, vals_out);
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
